"""The product's host-side R1CS assembly (csrc/host/r1cs.hpp, gadgets.hpp behind the C ABI) against tests/golden/assembly.json, which was
written by an INDEPENDENT pure-Python restatement of the reference's gadgets (tests/golden/pyref_r1cs.py, gen_assembly_fixtures.py):
constraint order and the merged coefficient of every variable in every constraint (these decide the z^j weights, hence the proof bytes),
the witness vectors a_L, a_R, a_O, the committed values, the commitment bytes and the transcript after the last "V".  No GPU: the
commitments come from the CPU oracle through bpg_prover_commit_precomputed (the GPU suite checks k_pedersen against the same oracle)."""
import hashlib
import json
import pathlib
import shutil
import struct

import pytest
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import cli
import oracle_lib as O
import assembly_cases as AC
import pyref_r1cs as PR

HERE = pathlib.Path(__file__).resolve().parent
FIX = json.loads((HERE / "golden" / "assembly.json").read_text())
L = AC.L


class HostProver(bpg.Prover):
    """Assembly-only product prover (no device context); Pedersen commitments by the oracle, registered with commit_precomputed."""
    def __init__(self, ctx, transcript):
        super().__init__(None, transcript)

    def commit(self, v, v_blinding):
        com = O.pedersen_commit((int.from_bytes(v, "little") % L).to_bytes(32, "little"), v_blinding)
        return com, self.commit_precomputed(v, v_blinding, com)

    def commit_many(self, vs, blindings):
        out = [self.commit(v, b) for v, b in zip(vs, blindings)]
        return [c for c, _ in out], [x for _, x in out]

    def prove(self, *a, **k):
        return b""

    def start_blinding(self, *a, **k):           # the file driver's two-pass flow (commitments first, then assembly) runs as it does on the GPU
        self.started_blinding = True


@pytest.fixture(autouse=True)
def setup_without_a_device(monkeypatch):
    """Gadget::setup = preprocess + one commit per derived scalar (src/gadget.rs:18-38); here the commits go through HostProver.commit"""
    def setup(self, prover, witnesses, blindings):
        derived = self.preprocess(witnesses)
        coms, out = [], []
        for s, b in zip(derived, blindings):
            com, v = prover.commit(s, b)
            coms.append(com); out.append((s, v))
        assert len(out) == len(derived)
        return coms, out
    monkeypatch.setattr(bpg.Gadget, "setup", setup)


class ProductApi:
    Transcript, Prover = bpg.Transcript, HostProver
    BoundsCheck, MimcHash256, MerkleTree256 = bpg.BoundsCheck, bpg.MimcHash256, bpg.MerkleTree256
    commit, commit_single, commit_all_single = staticmethod(bpg.commit), staticmethod(bpg.commit_single), staticmethod(bpg.commit_all_single)
    mimc_hash, be_to_scalar = staticmethod(bpg.mimc_hash), staticmethod(bpg.be_to_scalar)


def product_summary(p, transcript):
    inst = p.instance()
    rp = struct.unpack("<%dQ" % (inst.q + 1), inst.row_ptr)
    nnz = rp[-1]
    tv = struct.unpack("<%dI" % nnz, inst.term_var) if nnz else ()
    tc = struct.unpack("<%dI" % nnz, inst.term_coef) if nnz else ()
    coef = [int.from_bytes(inst.coef[32 * i:32 * i + 32], "little") for i in range(len(inst.coef) // 32)]
    rows = ([(tv[k], coef[tc[k]]) for k in range(rp[r], rp[r + 1])] for r in range(inst.q))
    sc = lambda b: [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(len(b) // 32)]
    return {"n": inst.n, "q": inst.q, "m": inst.m, "constraints_sha256": PR.digest_rows(rows),
            "witness_sha256": PR.digest_scalars(sc(inst.aL), sc(inst.aR), sc(inst.aO)),
            "committed_sha256": PR.digest_scalars(sc(inst.v), sc(inst.v_blinding)),
            "transcript_state_sha256": hashlib.sha256(transcript.state).hexdigest()}


KEYS = ("n", "q", "m", "constraints_sha256", "witness_sha256", "committed_sha256", "transcript_state_sha256")


@pytest.mark.parametrize("name", sorted(AC.CASES))
def test_gadget_assembly_matches_the_independent_restatement(name):
    p, t, coms = AC.build(ProductApi, name)
    got, want = product_summary(p, t), FIX[name]
    assert [c.hex() for c in coms] == want["commitments"]
    for k in KEYS:
        assert got[k] == want[k], (name, k)


def test_readme_example_assembly_matches_the_independent_restatement(tmp_path, monkeypatch):
    """example.gadgets (all seven gadget kinds, 9 lines; reference README / example.*) through the product's file driver, commitments by the
    oracle: same n = 14,988, q = 30,007, m = 33, same constraints, witness, .coms text and transcript as the Python restatement."""
    for ext in (".gadgets", ".inst", ".wtns"):
        shutil.copy(HERE / "golden" / "resources" / ("example" + ext), tmp_path / ("example" + ext))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(cli, "Prover", HostProver)
    monkeypatch.setattr(cli, "BulletproofGens", lambda ctx, cap: None)
    want = FIX["example_gadgets"]
    p, _ = cli.prover("example", ctx=object(), seed=want["blinding_seed"].encode(), rng_seed=bytes(32), quiet=True)
    got = product_summary(p, p.transcript)
    for k in KEYS:
        assert got[k] == want[k], k
    assert p.started_blinding            # two passes here (the fixture was written by a single pass in the reference's order): nothing may differ
    assert hashlib.sha256((tmp_path / "example.coms").read_bytes()).hexdigest() == want["coms_file_sha256"]
    coms_two_pass = (tmp_path / "example.coms").read_bytes()
    p1, _ = cli.prover("example", ctx=object(), seed=want["blinding_seed"].encode(), rng_seed=bytes(32), quiet=True, two_pass=False)
    assert product_summary(p1, p1.transcript) == got and (tmp_path / "example.coms").read_bytes() == coms_two_pass
    assert (got["n"], got["q"], got["m"]) == (14988, 30007, 33)


def test_the_restatement_detects_a_reordered_constraint():
    """the digest is sensitive to what matters: swapping two constraints or changing one coefficient changes it; the order of terms inside a
    constraint and a split coefficient do not"""
    rows = [[(1, 5), (2, 7)], [(3, 1)], [(1, 2), (1, 3)]]
    d = PR.digest_rows(rows)
    assert PR.digest_rows([[(2, 7), (1, 5)], [(3, 1)], [(1, 5)]]) == d
    assert PR.digest_rows([[(3, 1)], [(1, 5), (2, 7)], [(1, 5)]]) != d
    assert PR.digest_rows([[(1, 5), (2, 8)], [(3, 1)], [(1, 5)]]) != d
