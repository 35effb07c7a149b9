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
