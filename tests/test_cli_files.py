"""File-driver subset (BOUND / HASH / MERKLE): the reference's CI integration test is prover -> verifier exit code 0 on
tests/resources/* (.github/workflows/integration_tests.yml:19-58). Same here on the fixture copies, plus circuit sizes from
SURVEY.md Appendix C and rejection after tampering."""
import shutil
import pathlib
import pytest
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import cli
import oracle_lib as O

RES = pathlib.Path(__file__).resolve().parent / "golden" / "resources"


def test_tree_parser():
    assert cli.parse_tree("((W0 I1) (I3 W1))") == (["I1", "I3"], ["W0", "W1"], "((W I) (I W))")
    assert cli.parse_tree("(W1 I3)") == (["I3"], ["W1"], "(W I)")
    assert cli.parse_tree("(((W2 W3) I0) W9)")[2] == "(((W W) I) W)"
    with pytest.raises(ValueError):
        cli.parse_tree("(W0 I1")
    # nesting is bounded - 64 levels in both drivers and in the library (Pattern::MAX_DEPTH): twenty thousand opening brackets were a stack overflow of the
    # native driver under the fuzzer (tests/hostcheck), a RecursionError here
    deep = lambda d: "(" * d + "W0" + " I1)" * d
    assert cli.parse_tree(deep(60))[2].count("(") == 60
    for bad in (deep(80), "(" * 20000 + "W0"):
        with pytest.raises(ValueError):
            cli.parse_tree(bad)
    assert cli.round_pow2(14988) == 16384 and cli.round_pow2(128) == 128 and cli.round_pow2(1) == 1


@pytest.fixture(scope="module")
def ctx():
    return bpg.Context(0)


# expected (multipliers, constraints, commitments) derived per SURVEY.md Appendix C
CASES = {
    "bounds_check": None,
    "mimc_hash": None,
    "merkle_tree": None,
    "equality": None,
    "inequality": None,
    "less_than": (3 * 379, 3 * 763, None),
    "set_membership": None,
    "or": None, "or2": None, "or4": None, "or5": (4721, 10531, 29),   # OR blocks: clauses 37 x 6 x 1 x 5 explicit constraints -> 1110 products
    "or3": None,                             # nested OR over EQUALS clauses (reference tests/resources/or3.*)
    "example": (14988, 30007, 33),          # SURVEY.md section 8 cfg 1: the reference's README example, all nine lines
    "example_subset": (16 + 972 + (972 + 1944) * 2 + 2 * 972 + 3 * 1944, 35 + 1946 + (1946 + 3889) * 2 + 2 * 1946 + 11665, None),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_prover_then_verifier(ctx, tmp_path, name):
    for ext in ("gadgets", "inst", "wtns"):
        shutil.copy(RES / ("%s.%s" % (name, ext)), tmp_path / ("%s.%s" % (name, ext)))
    stem = str(tmp_path / name)
    p, proof = cli.prover(stem, ctx=ctx, seed=b"cli-test", rng_seed=bytes(32), quiet=True)
    if CASES[name]:
        assert (p.get_num_multiplications(), p.num_constraints()) == CASES[name][:2]
        if CASES[name][2] is not None:
            assert p.num_committed() == CASES[name][2]
    assert (tmp_path / (name + ".proof")).read_bytes() == proof
    if p.get_num_multiplications() <= (1 << 14):
        # byte parity with the oracle PROVER on what the driver assembled (cfg 1: the README example at N = 2^14 takes the oracle ~2 s)
        inst = p.instance()
        pre = bpg.Transcript(stem.encode())
        pre.append_message(b"dom-sep", b"r1cs v1")                              # Prover::new
        for k in range(inst.m):
            pre.append_message(b"V", bytes.fromhex((tmp_path / (name + ".coms")).read_text().splitlines()[k].split("0x")[1]))
        oc_p = O.FlatCircuit(inst.n, inst.m, inst.aL, inst.aR, inst.aO, inst.row_ptr, inst.term_var, inst.term_coef, inst.coef)
        rc, want, _ = O.prove(O.Gens(cli.round_pow2(inst.n)), pre.state, oc_p, inst.v_blinding, bytes(32), O.FLAG_FAST_MSM)
        assert rc == 0 and proof == want, "%s: the driver's proof differs from the oracle prover's" % name
    if name == "example":
        import hashlib, json
        fix = json.loads((RES.parent / "assembly.json").read_text())["example_gadgets"]
        p2, _ = cli.prover(stem, ctx=ctx, seed=fix["blinding_seed"].encode(), rng_seed=bytes(32), quiet=True)
        # the .coms file the GPU driver writes under the fixture's blinding seed is the one the independent Python restatement wrote
        assert hashlib.sha256((tmp_path / (name + ".coms")).read_bytes()).hexdigest() == fix["coms_file_sha256"]
        p, proof = cli.prover(stem, ctx=ctx, seed=b"cli-test", rng_seed=bytes(32), quiet=True)
    coms = (tmp_path / (name + ".coms")).read_text().splitlines()
    assert len(coms) == p.num_committed() and all(l.startswith(("C", "D")) and " = 0x" in l for l in coms)
    # verifier side: assembled only from .gadgets/.inst/.coms
    assert cli.verifier(stem, ctx=ctx, quiet=True)
    v, tv = cli.assemble_verifier(stem)
    assert v.get_num_vars() == p.get_num_multiplications()
    vi = v.instance()
    og = O.Gens(cli.round_pow2(vi.n))
    oc = O.FlatCircuit(vi.n, vi.m, None, None, None, vi.row_ptr, vi.term_var, vi.term_coef, vi.coef)
    assert O.verify(og, tv.state, oc, vi.commitments, proof) == 0           # the independent oracle verifier agrees
    # tampering with the proof or with one commitment line must flip the verdict
    bad = bytearray(proof); bad[33] ^= 1
    (tmp_path / (name + ".proof")).write_bytes(bytes(bad))
    assert not cli.verifier(stem, ctx=ctx, quiet=True)
    (tmp_path / (name + ".proof")).write_bytes(proof)
    lines = (tmp_path / (name + ".coms")).read_text().splitlines(keepends=True)
    if len(lines) < 2 or lines[0].split(" = ")[1] == lines[-1].split(" = ")[1]:
        return
    lines[0], lines[-1] = lines[0].split(" = ")[0] + " = " + lines[-1].split(" = ")[1], lines[-1].split(" = ")[0] + " = " + lines[0].split(" = ")[1]
    (tmp_path / (name + ".coms")).write_text("".join(lines))
    assert not cli.verifier(stem, ctx=ctx, quiet=True)


@pytest.mark.gpu
def test_depth20_merkle_path_cfg4b(ctx, tmp_path):
    """SURVEY.md section 8 cfg 4b: BASELINE.json's "depth 20" wording read literally - one authentication path through the CLI."""
    from bulletproofs_gadgets_amd import workloads
    stem = str(tmp_path / "path20")
    n = workloads.merkle_path_files(stem, depth=20)
    assert n == 39852
    p, proof = cli.prover(stem, ctx=ctx, seed=b"cfg4b", rng_seed=bytes(32), quiet=True)
    assert p.get_num_multiplications() == n and p.num_committed() == 4 and len(proof) == 1536
    assert p.num_constraints() == 20 * 3888 + 1 + 1946          # 2 per multiplier, the root equality, the leaf image equality x2
    assert cli.verifier(stem, ctx=ctx, quiet=True)
    v, tv = cli.assemble_verifier(stem)
    vi = v.instance()
    oc = O.FlatCircuit(vi.n, vi.m, None, None, None, vi.row_ptr, vi.term_var, vi.term_coef, vi.coef)
    gens = O.Gens(65536)
    assert O.verify(gens, tv.state, oc, vi.commitments, proof) == 0
    # the oracle PROVER on the same statement (the driver's assembly with the GPU's commitments, the same blindings and seed) gives the same bytes
    pa, ta = cli.prover(stem, ctx=ctx, seed=b"cfg4b", rng_seed=bytes(32), quiet=True, two_pass=False, assemble_only=True)
    pi = pa.instance()
    rc, want, _ = O.prove(gens, ta.state, O.FlatCircuit(pi.n, pi.m, pi.aL, pi.aR, pi.aO, pi.row_ptr, pi.term_var, pi.term_coef, pi.coef), pi.v_blinding,
                          bytes(32), O.FLAG_FAST_MSM)
    assert rc == 0 and want == proof
    # a wrong sibling must make the prover's own circuit unsatisfied -> the proof does not verify
    lines = open(stem + ".inst").read().splitlines()
    lines[7] = lines[7][:-2] + ("00" if lines[7][-2:] != "00" else "01")
    open(stem + ".inst", "w").write("\n".join(lines) + "\n")
    assert not cli.verifier(stem, ctx=ctx, quiet=True)
