"""Native file drivers bpg_prover / bpg_verifier (csrc/cli_main.cpp: the reference's src/bin/prover.rs and src/bin/verifier.rs restated in
C++ over the C ABI): same files as the Python driver under the same blinding stream, exit codes of the reference's verifier."""
import os
import pathlib
import shutil
import subprocess
import pytest
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import cli
from bulletproofs_gadgets_amd import build as bpg_build

RES = pathlib.Path(__file__).resolve().parent / "golden" / "resources"
CASES = ["bounds_check", "mimc_hash", "merkle_tree", "equality", "inequality", "less_than", "set_membership", "or", "or2", "or3", "or4", "or5", "example"]


def test_native_cli_is_built_and_prints_usage():
    bins = bpg_build.build_cli()
    assert all(b.exists() for b in bins)
    r = subprocess.run([str(bins[0])], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.fixture(scope="module")
def ctx():
    return bpg.Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_native_prover_and_verifier_match_python_driver(ctx, tmp_path, name):
    prover_bin, verifier_bin = bpg_build.build_cli()
    a, b = tmp_path / "native", tmp_path / "python"
    for d in (a, b):
        d.mkdir()
        for ext in ("gadgets", "inst", "wtns"):
            shutil.copy(RES / ("%s.%s" % (name, ext)), d / ("%s.%s" % (name, ext)))
    env = dict(os.environ, BPG_CLI_SEED="cli-test", BPG_CLI_RNG_SEED="00" * 32)
    # the transcript label is the NAME argument (prover.rs:49-52): run both drivers with the same relative name
    r = subprocess.run([str(prover_bin), name], cwd=a, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cwd = os.getcwd()
    try:
        os.chdir(b)
        p, proof = cli.prover(name, ctx=ctx, seed=b"cli-test", rng_seed=bytes(32), quiet=True)
    finally:
        os.chdir(cwd)
    assert int(r.stdout.strip()) == p.num_constraints()                      # prover.rs:89
    assert (a / (name + ".coms")).read_text() == (b / (name + ".coms")).read_text()
    assert (a / (name + ".proof")).read_bytes() == proof
    # both drivers ran in two passes (commitments, early blinding stream, then assembly): the reference's single pass gives the same files
    c = tmp_path / "single"
    c.mkdir()
    for ext in ("gadgets", "inst", "wtns"):
        shutil.copy(RES / ("%s.%s" % (name, ext)), c / ("%s.%s" % (name, ext)))
    r1 = subprocess.run([str(prover_bin), name], cwd=c, env=dict(env, BPG_CLI_TWO_PASS="0"), capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r1.stdout == r.stdout, r1.stderr
    assert (c / (name + ".coms")).read_bytes() == (a / (name + ".coms")).read_bytes() and (c / (name + ".proof")).read_bytes() == proof
    try:
        os.chdir(b)
        _, proof1 = cli.prover(name, ctx=ctx, seed=b"cli-test", rng_seed=bytes(32), quiet=True, two_pass=False)
    finally:
        os.chdir(cwd)
    assert proof1 == proof and (b / (name + ".coms")).read_bytes() == (a / (name + ".coms")).read_bytes()
    v = subprocess.run([str(verifier_bin), name], cwd=a, capture_output=True, text=True, timeout=300)
    assert (v.returncode, v.stdout.strip()) == (0, "true"), v.stderr       # verifier.rs:91-100
    # the Python verifier accepts the native proof as well
    try:
        os.chdir(a)
        assert cli.verifier(name, ctx=ctx, quiet=True)
    finally:
        os.chdir(cwd)
    bad = bytearray(proof); bad[40] ^= 1
    (a / (name + ".proof")).write_bytes(bytes(bad))
    v = subprocess.run([str(verifier_bin), name], cwd=a, capture_output=True, text=True, timeout=300)
    assert (v.returncode, v.stdout.strip()) == (1, "false")
    # generic two-argument form
    (a / (name + ".proof")).write_bytes(proof)
    v = subprocess.run([str(prover_bin), "verifier", name], cwd=a, capture_output=True, text=True, timeout=300)
    assert (v.returncode, v.stdout.strip()) == (0, "true")


@pytest.mark.gpu
def test_native_prover_reports_malformed_input(tmp_path):
    prover_bin, _ = bpg_build.build_cli()
    (tmp_path / "bad.gadgets").write_text("FROBNICATE W0\n")
    (tmp_path / "bad.inst").write_text("")
    (tmp_path / "bad.wtns").write_text("W0 = 0x01\n")
    r = subprocess.run([str(prover_bin), "bad"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 101 and "unknown gadget line" in r.stderr      # the reference unwrap()s: panic, exit code 101


def _random_gadget_files(stem, seed):
    """A seeded random .gadgets/.inst/.wtns triple whose statements hold: every gadget kind, and OR blocks (also nested) that mix
    false clauses with a true one."""
    import random
    from bulletproofs_gadgets_amd import workloads
    rnd = random.Random(seed)
    inst, wtns, lines = [], [], []

    def I(b):
        inst.append(b); return "I%d" % (len(inst) - 1)

    def W(b):
        wtns.append(b); return "W%d" % (len(wtns) - 1)

    def rb(n, first_max=0x0f):
        return bytes([rnd.randrange(1, first_max + 1)]) + bytes(rnd.randrange(256) for _ in range(n - 1))

    def true_line():
        k = rnd.randrange(7)
        if k == 0:
            v = rnd.randrange(50, 200); return "BOUND %s %s %s" % (W(bytes([v])), I(bytes([rnd.randrange(0, v + 1)])), I(bytes([rnd.randrange(v, 256)])))
        if k == 1:
            pre = rb(rnd.choice([5, 20, 31, 40])); return "HASH %s %s" % (I(bpg.scalar_to_be(bpg.mimc_hash(pre))), W(pre))
        if k == 2:
            v = rb(rnd.choice([8, 32, 45])); return rnd.choice(["EQUALS %s %s" % (W(v), I(v)), "EQUALS %s %s" % (I(v), W(v)), "EQUALS %s %s" % (W(v), W(v))])
        if k == 3:
            v = rb(rnd.choice([8, 32])); u = bytes([v[0] ^ 1]) + v[1:]; return "UNEQUAL %s %s" % (W(v), rnd.choice([I, W])(u))
        if k == 4:
            a = rnd.randrange(1, 1 << 60); b = a + rnd.randrange(1, 1 << 60); return "LESS_THAN %s %s" % (W(a.to_bytes(9, "big")), W(b.to_bytes(9, "big")))
        if k == 5:
            m = rb(4); others = [rb(4) for _ in range(3)]
            names = [I(others[0]), W(others[1]), I(m), W(others[2])]; rnd.shuffle(names)
            return "SET_MEMBER %s %s" % (W(m), " ".join(names))
        # MERKLE over ((W I) (I W)) with the root computed by an assembly-only prover
        leaves = [rb(12) for _ in range(4)]
        hashed = [bpg.mimc_hash(x) for x in leaves]
        probe = bpg.Prover(None, bpg.Transcript(b"probe"))
        bpg.MerkleTree256(bytes(32), hashed, [], "((I I) (I I))").prove(probe, [], [])
        root = probe.instance().aO[-32:]
        return "MERKLE %s ((%s %s) (%s %s))" % (I(root[::-1]), W(leaves[0]), I(leaves[1]), I(leaves[2]), W(leaves[3]))

    def false_line():
        v = rb(8); return "EQUALS %s %s" % (W(v), I(bytes([v[0] ^ 1]) + v[1:]))

    def or_block(depth):
        out = ["OR", "["]
        clauses = rnd.randrange(2, 4); good = rnd.randrange(clauses)
        for c in range(clauses):
            out.append("{")
            if c == good:
                out.append(true_line() if depth or rnd.random() < 0.6 else "\n".join(or_block(depth + 1)))
                if rnd.random() < 0.4: out.append(true_line())
            else:
                out.append(false_line())
                if rnd.random() < 0.3: out.append(true_line())
            out.append("}")
        out.append("]")
        return out

    for _ in range(rnd.randrange(2, 5)):
        lines += [true_line()] if rnd.random() < 0.7 else or_block(0)
    text = "\n".join(lines) + "\n"
    open(stem + ".gadgets", "w").write(text)
    open(stem + ".inst", "w").write("".join("I%d = 0x%s\n" % (k, b.hex()) for k, b in enumerate(inst)))
    open(stem + ".wtns", "w").write("".join("W%d = 0x%s\n" % (k, b.hex()) for k, b in enumerate(wtns)))
    return text


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_gadget_files_native_and_python_agree(ctx, tmp_path, seed):
    prover_bin, verifier_bin = bpg_build.build_cli()
    a, b = tmp_path / "native", tmp_path / "python"
    a.mkdir(); b.mkdir()
    text = _random_gadget_files(str(a / "rnd"), 4242 + seed)
    for ext in ("gadgets", "inst", "wtns"):
        shutil.copy(a / ("rnd." + ext), b / ("rnd." + ext))
    env = dict(os.environ, BPG_CLI_SEED="fuzz", BPG_CLI_RNG_SEED="11" * 32)
    r = subprocess.run([str(prover_bin), "rnd"], cwd=a, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (text, r.stderr)
    cwd = os.getcwd()
    try:
        os.chdir(b)
        p, proof = cli.prover("rnd", ctx=ctx, seed=b"fuzz", rng_seed=bytes([0x11]) * 32, quiet=True)
        assert cli.verifier("rnd", ctx=ctx, quiet=True), text
    finally:
        os.chdir(cwd)
    assert (a / "rnd.coms").read_text() == (b / "rnd.coms").read_text(), text
    assert (a / "rnd.proof").read_bytes() == proof, text
    v = subprocess.run([str(verifier_bin), "rnd"], cwd=a, capture_output=True, text=True, timeout=300)
    assert (v.returncode, v.stdout.strip()) == (0, "true"), (text, v.stderr)
    # make the statement false: break the first instance value -> the verifier must say false (or the prover's circuit is unsatisfied)
    inst_lines = (a / "rnd.inst").read_text().splitlines()
    if inst_lines:
        name, val = inst_lines[0].split(" = 0x")
        inst_lines[0] = "%s = 0x%s" % (name, val[:-1] + ("0" if val[-1] != "0" else "1"))
        (a / "rnd.inst").write_text("\n".join(inst_lines) + "\n")
        v = subprocess.run([str(verifier_bin), "rnd"], cwd=a, capture_output=True, text=True, timeout=300)
        assert v.returncode in (0, 1, 101), (v.returncode, v.stderr[-3000:])               # 0 only if the changed value sits in a false OR clause or is unused
