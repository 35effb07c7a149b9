#!/usr/bin/env python3
"""Regression fixtures at the .proof byte level: proofs of commitment-free circuits (assembled by the product's host code, no GPU
needed) produced by the ORACLE under fixed RNG seeds, for every encoding dialect.  They do not pin parity with the reference
(it stores no proof bytes); they freeze today's oracle output so that both the oracle (CPU suite) and
the HIP path (GPU suite) are compared with committed bytes, not only with each other.
    python tests/golden/gen_proof_fixtures.py     ->  tests/golden/proofs.json
"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
import bulletproofs_gadgets_amd as bpg          # host-side assembly only (Prover(None, ...))
import oracle_lib as O

RANGE_VALUE_BE = "0522a64d7b931e"                # reference src/utils.rs:46-90 (56-bit value)
LEAF_BE = "0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749"       # merkle_tree_gadget.rs:126-131 (leaf W1)


def to_oracle(inst):
    return O.FlatCircuit(inst.n, inst.m, inst.aL or None, inst.aR or None, inst.aO or None, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)


def circuits():
    """name -> (transcript label, builder(prover), generator capacity)"""
    x = bpg.be_to_scalar(bytes.fromhex(RANGE_VALUE_BE))
    leaf = bpg.be_to_scalar(bytes.fromhex(LEAF_BE))

    def merkle4(p):
        probe = bpg.Prover(None, bpg.Transcript(b"probe"))
        bpg.MerkleTree256(bytes(32), [leaf] * 4, [], "((I I) (I I))").prove(probe, [], [])
        root = probe.instance().aO[-32:]
        bpg.MerkleTree256(root, [leaf] * 4, [], "((I I) (I I))").prove(p, [], [])
    return {
        "range56": (b"RangeProof", lambda p: bpg.range_proof(p, x, 56, x), 64),
        "range8": (b"RangeProof", lambda p: bpg.range_proof(p, bpg.be_to_scalar(b"\xa5"), 8, bpg.be_to_scalar(b"\xa5")), 8),
        "merkle4": (b"MerkleTree", merkle4, 8192),
    }


def seeds():
    return [bytes(32), bytes(range(32)), hashlib.sha256(b"bpg proof fixture").digest()]


def build(name):
    label, fn, cap = circuits()[name]
    t = bpg.Transcript(label)
    p = bpg.Prover(None, t)
    fn(p)
    return p.instance(), t.state, cap


def main():
    out = {"_provenance": "oracle/ (C restatement) via tests/oracle_lib.py; circuits assembled by the product's host code; see the docstring of gen_proof_fixtures.py",
           "proofs": []}
    for name in circuits():
        inst, state, cap = build(name)
        gens = O.Gens(cap)
        for flags in (0, O.FLAG_COMPACT_1PHASE, O.FLAG_NO_1PHASE_DOMSEP, O.FLAG_COMPACT_1PHASE | O.FLAG_NO_1PHASE_DOMSEP, O.FLAG_EXPANDED_BLINDING):
            for seed in seeds()[: (3 if name != "merkle4" else 1)]:
                rc, proof, _ = O.prove(gens, state, to_oracle(inst), b"", seed, flags | O.FLAG_FAST_MSM)
                assert rc == 0
                rec = {"circuit": name, "n": inst.n, "q": inst.q, "capacity": cap, "flags": flags, "seed": seed.hex(), "len": len(proof),
                       "sha256": hashlib.sha256(proof).hexdigest()}
                if len(proof) <= 1024:
                    rec["proof"] = proof.hex()
                out["proofs"].append(rec)
    with open(os.path.join(HERE, "proofs.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("%d proofs" % len(out["proofs"]))


if __name__ == "__main__":
    main()
