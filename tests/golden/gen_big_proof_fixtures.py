#!/usr/bin/env python3
"""Proof fixtures at BASELINE.json's full sizes, produced by the ORACLE in the build container (no GPU involved):
    cfg4_merkle512   the reference's own 2^20 circuit (src/merkle_tree/merkle_tree_gadget.rs:473-545: 512 x leaf W1, n = 993,384), 2 seeds
    merkle256_seed5  a full 256-leaf tree with seeded leaves, N = 2^19 (the circuit of test_fold_profiles_agree_at_half_a_million_multipliers)
    cfg3_mimc67      BASELINE.json config 3: HASH over 2,130 bytes, 67 absorbed blocks, N = 2^16
    cfg4b_path20     SURVEY.md section 8 cfg 4b: BASELINE.json's "depth 20" read literally - one Merkle authentication path through the FILE driver
                     (workloads.merkle_path_files -> cli.prover(assemble_only)), n = 39,852, N = 2^16
for all four encoding dialects.  The circuits are assembled by the product's HOST code (workloads.py on an assembly-only prover); every Pedersen
commitment is made by the oracle (O.pedersen_commit) and handed in through bpg_prover_commit_precomputed, so the recorded transcript state is
the oracle's too.  One oracle prove per distinct transcript (flags 0 and NO_1PHASE_DOMSEP); the COMPACT encodings are the same proofs with the
version byte in front and the three identity phase-2 points left out - the derivation is checked against a real oracle prove of
that dialect on the 2^16 case.  Records go to tests/golden/proofs_big.json: {circuit, n, q, m, capacity, seed, flags, len, sha256, head, state_sha256}.

    python tests/golden/gen_big_proof_fixtures.py [--jobs 6]          about 4 minutes of wall time on 8 cores (150 s per 2^20 prove on one)

Regression fixtures, not a pin against the reference (it stores no proof bytes)."""
import argparse, hashlib, json, multiprocessing, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)

SEEDS = [bytes(range(32)), hashlib.sha256(b"bpg proof fixture").digest()]
# circuit -> seeds proved at each of the two transcripts
PLAN = {"cfg4_merkle512": SEEDS, "merkle256_seed5": SEEDS[:1], "cfg3_mimc67": SEEDS[:1], "cfg4b_path20": SEEDS[:1]}


def oracle_prover_class():
    """Prover(None, transcript) whose commitments come from the oracle: the assembly runs in the product's host library, no device anywhere."""
    import bulletproofs_gadgets_amd as bpg
    import oracle_lib as O

    class OracleCommitProver(bpg.Prover):
        def __init__(self, ctx, transcript):
            super().__init__(None, transcript)

        def commit(self, v, v_blinding):
            com = O.pedersen_commit(v, v_blinding)
            return com, self.commit_precomputed(v, v_blinding, com)

        def commit_many(self, vs, blindings):
            pairs = [self.commit(v, r) for v, r in zip(vs, blindings)]
            return [c for c, _ in pairs], [x for _, x in pairs]

        def gadget_setup(self, g, witnesses, blindings):        # Gadget::setup (reference src/gadget.rs:18-38): preprocess, then one commit per derived scalar
            derived = g.preprocess(witnesses)
            coms, vars_ = self.commit_many(derived, list(blindings)[:len(derived)])
            return coms, list(zip(derived, vars_))
    return OracleCommitProver


def build(name, ctx=None):
    """The circuit of a fixture: on a device context (GPU tests) or, with ctx None, on the assembly-only prover with oracle commitments."""
    from bulletproofs_gadgets_amd import workloads
    kw = {} if ctx is not None else {"prover_cls": oracle_prover_class()}
    if name == "cfg4_merkle512":
        return workloads.merkle_full_tree(ctx, leaves=512, seed=None, **kw)
    if name == "merkle256_seed5":
        return workloads.merkle_full_tree(ctx, leaves=256, seed=5, **kw)
    if name == "cfg3_mimc67":
        return workloads.mimc_preimage(ctx, nbytes=2130, seed=0, **kw)
    if name == "cfg4b_path20":
        return path20(ctx, **kw)
    raise KeyError(name)


def path20(ctx, prover_cls=None):
    """cfg 4b through the file driver, assembled but not proved.  The transcript label is the NAME argument of the driver (prover.rs:49-52): the files
    are written to a scratch directory and the driver runs there under the fixed relative name "path20"."""
    import tempfile
    from bulletproofs_gadgets_amd import cli, workloads
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            n = workloads.merkle_path_files("path20", depth=20)
            p, t = cli.prover("path20", ctx=ctx, seed=b"cfg4b", rng_seed=bytes(32), quiet=True, two_pass=False, prover_cls=prover_cls, assemble_only=True)
        finally:
            os.chdir(cwd)
    assert p.get_num_multiplications() == n == 39852
    coms = [p.commitment(i) for i in range(p.num_committed())]
    return workloads.Assembled(p, t, coms, 1 << 16, None)


def compact_of(proof14):
    """the D11v encoding of a one-phase proof given its 14-point encoding"""
    assert proof14[96:192] == bytes(96)
    return b"\x00" + proof14[:96] + proof14[192:]


def record(name, inst, cap, state, seed, flags, proof):
    return {"circuit": name, "n": inst.n, "q": inst.q, "m": inst.m, "capacity": cap, "seed": seed.hex(), "flags": flags, "len": len(proof),
            "sha256": hashlib.sha256(proof).hexdigest(), "head": proof[:64].hex(), "state_sha256": hashlib.sha256(state).hexdigest()}


def job(args):
    name, seed_hex, flags = args
    import oracle_lib as O
    t0 = time.time()
    a = build(name)
    inst, state = a.prover.instance(), a.transcript.state
    oc = O.FlatCircuit(inst.n, inst.m, inst.aL, inst.aR, inst.aO, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)
    assert O.satisfied(oc, inst.v)
    gens = O.Gens(a.gens_capacity)
    seed = bytes.fromhex(seed_hex)
    rc, proof, _ = O.prove(gens, state, oc, inst.v_blinding, seed, flags | O.FLAG_FAST_MSM)
    assert rc == 0
    recs = [record(name, inst, a.gens_capacity, state, seed, flags, proof),
            record(name, inst, a.gens_capacity, state, seed, flags | O.FLAG_COMPACT_1PHASE, compact_of(proof))]
    if name == "cfg3_mimc67":       # the derivation of the compact dialect against the oracle itself
        rc, pc, _ = O.prove(gens, state, oc, inst.v_blinding, seed, flags | O.FLAG_COMPACT_1PHASE | O.FLAG_FAST_MSM)
        assert rc == 0 and pc == compact_of(proof)
    print("%s seed %s flags %d: %.0f s" % (name, seed_hex[:8], flags, time.time() - t0), flush=True)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--only", default=None, help="one circuit name")
    a = ap.parse_args()
    import oracle_lib as O
    O.lib()                                                  # build the oracle once, before the workers start
    jobs = [(name, s.hex(), flags) for name, seeds in PLAN.items() if a.only in (None, name) for s in seeds for flags in (0, O.FLAG_NO_1PHASE_DOMSEP)]
    jobs.sort(key=lambda j: 0 if j[0] == "cfg4_merkle512" else 1)      # longest first
    with multiprocessing.get_context("spawn").Pool(min(a.jobs, len(jobs))) as pool:
        out = [r for recs in pool.map(job, jobs, chunksize=1) for r in recs]
    out.sort(key=lambda r: (r["circuit"], r["seed"], r["flags"]))
    path = os.path.join(HERE, "proofs_big.json")
    if a.only and os.path.exists(path):
        keep = [r for r in json.load(open(path))["proofs"] if r["circuit"] != a.only]
        out = sorted(keep + out, key=lambda r: (r["circuit"], r["seed"], r["flags"]))
    with open(path, "w") as f:
        json.dump({"_provenance": "oracle/ (C restatement) run in the build container by tests/golden/gen_big_proof_fixtures.py; circuits assembled by the "
                                  "product's host code with oracle commitments; regression fixtures, not a pin against the reference", "proofs": out}, f, indent=1)
    print("%d records -> %s" % (len(out), path))


if __name__ == "__main__":
    main()
