"""Committed proof bytes (tests/golden/proofs.json, written by tests/golden/gen_proof_fixtures.py): the oracle must still
produce them (CPU suite) and the HIP path must produce the same bytes through the C ABI (GPU suite).  Regression fixtures -
the reference itself stores no proof bytes."""
import hashlib
import json
import pathlib
import pytest
import bulletproofs_gadgets_amd as bpg
import oracle_lib as O
import gen_proof_fixtures as G

FIX = json.loads((pathlib.Path(__file__).parent / "golden" / "proofs.json").read_text())["proofs"]
NAMES = sorted({r["circuit"] for r in FIX})


def _check(rec, proof):
    assert len(proof) == rec["len"] == O.proof_size(rec["n"], rec["flags"])
    assert hashlib.sha256(proof).hexdigest() == rec["sha256"]
    if "proof" in rec:
        assert proof.hex() == rec["proof"]


def test_fixture_covers_every_dialect_and_the_generator_is_committed():
    assert len(FIX) == 35 and {r["flags"] for r in FIX} == {0, 1, 2, 3, 4}
    assert (pathlib.Path(__file__).parent / "golden" / "gen_proof_fixtures.py").exists()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_committed_proofs(name):
    inst, state, cap = G.build(name)
    gens = O.Gens(cap)
    recs = [r for r in FIX if r["circuit"] == name]
    assert recs and all((r["n"], r["q"], r["capacity"]) == (inst.n, inst.q, cap) for r in recs)
    for k, r in enumerate(recs):
        # alternate the oracle's two MSM strategies: upstream's Straus / Pippenger and the fast bucket method give the same bytes
        rc, proof, _ = O.prove(gens, state, G.to_oracle(inst), b"", bytes.fromhex(r["seed"]), r["flags"] | (O.FLAG_FAST_MSM if (k & 1) or inst.n > 1000 else 0))
        assert rc == 0
        _check(r, proof)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_committed_proofs(name):
    ctx = bpg.Context(0)
    inst, state, cap = G.build(name)
    ctx.gens_ensure(cap)
    res = ctx.upload(inst)
    for r in [r for r in FIX if r["circuit"] == name]:
        proof, _ = res.prove(state, b"", bytes.fromhex(r["seed"]), r["flags"])
        _check(r, proof)
        flat, _ = ctx.prove_flat(inst, state, b"", bytes.fromhex(r["seed"]), r["flags"])          # host-buffer entry point of the C ABI
        assert flat == proof
