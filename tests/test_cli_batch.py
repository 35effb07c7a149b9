"""File-level batches (BASELINE.json config 5 as a product; the reference's own batch is its CI workflow, prover then verifier over twelve
stems: .github/workflows/integration_tests.yml:19-58): `bpg_prover --batch FILE --gpus N` (native: the command starts its ranks itself) and
`python -m bulletproofs_gadgets_amd.cli prover --batch FILE --gpus N` (one rank per GPU over torch.distributed, proof bytes gathered) write the
same .coms / .proof files as one run per stem, in one process and in two."""
import hashlib
import os
import pathlib
import shutil
import subprocess
import sys
import pytest
from bulletproofs_gadgets_amd import cli
from bulletproofs_gadgets_amd import build as bpg_build

ROOT = pathlib.Path(__file__).resolve().parent.parent
RES = ROOT / "tests" / "golden" / "resources"
# the twelve stems of the reference's workflow, in its order, and the README example
STEMS = ["bounds_check", "equality", "inequality", "less_than", "merkle_tree", "mimc_hash", "set_membership", "or", "or2", "or3", "or4", "or5", "example"]
ENV = dict(BPG_CLI_SEED="cli-test", BPG_CLI_RNG_SEED="00" * 32)


def _stage(d):
    d.mkdir()
    for s in STEMS:
        for ext in ("gadgets", "inst", "wtns"):
            shutil.copy(RES / ("%s.%s" % (s, ext)), d / ("%s.%s" % (s, ext)))
    (d / "batch.txt").write_text("# the reference's integration workflow\n" + "\n".join(STEMS[:7]) + "\n\n" + "\n".join(STEMS[7:]) + "\n")
    return d


def _files(d):
    return {s + e: (d / (s + e)).read_bytes() for s in STEMS for e in (".coms", ".proof")}


def test_batch_file_parsing_and_usage(tmp_path):
    f = tmp_path / "b.txt"
    f.write_text("a/b\n\n  # comment\n c \n")
    assert cli.read_batch(str(f)) == ["a/b", "c"]
    assert cli.main(["prover", "--batch"]) == 2 and cli.main(["prover", "--batch", str(f), "--gpus"]) == 2
    prover_bin, _ = bpg_build.build_cli()
    r = subprocess.run([str(prover_bin), "--batch", str(f), "--frobnicate", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown option" in r.stderr


def test_batch_without_a_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = _stage(tmp_path / "x")
    prover_bin, _ = bpg_build.build_cli()
    r = subprocess.run([str(prover_bin), "--batch", "batch.txt", "--gpus", "2"], cwd=d, capture_output=True, text=True, timeout=120)
    assert r.returncode == 101 and r.stdout.count("FAILED") == len(STEMS) and "no AMD GPU" in r.stderr     # no CPU path


@pytest.mark.gpu
def test_batch_drivers_write_the_files_of_one_run_per_stem(tmp_path):
    prover_bin, verifier_bin = bpg_build.build_cli()
    env = dict(os.environ, **ENV)
    # (a) the reference's way: one prover process per stem
    a = _stage(tmp_path / "per_stem")
    counts = {}
    for s in STEMS:
        r = subprocess.run([str(prover_bin), s], cwd=a, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        counts[s] = int(r.stdout.strip())
    want = _files(a)
    # (b) native batch, one rank, one worker thread; (c) one rank, four workers (the default); (d) two ranks (two processes on this card, two workers each)
    for name, extra in (("native1", ["--workers", "1"]), ("native1x4", []), ("native2", ["--gpus", "2", "--workers", "2"])):
        d = _stage(tmp_path / name)
        r = subprocess.run([str(prover_bin), "--batch", "batch.txt"] + extra, cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert _files(d) == want, name
        lines = r.stdout.strip().splitlines()
        assert [l.split(":")[0] for l in lines] == STEMS                                  # summary in file order
        for s, l in zip(STEMS, lines):
            assert l == "%s: %d constraints, %d-byte proof" % (s, counts[s], len(want[s + ".proof"]))
        rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt"] + extra, cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert rv.returncode == 0 and rv.stdout.strip().splitlines() == ["%s: true" % s for s in STEMS], rv.stdout + rv.stderr
    # a tampered proof in the batch: that stem is reported false, the command exits 1 like the reference's verifier
    bad = bytearray((d / "less_than.proof").read_bytes()); bad[40] ^= 1
    (d / "less_than.proof").write_bytes(bytes(bad))
    rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt", "--gpus", "2"], cwd=d, env=env, capture_output=True, text=True, timeout=600)
    assert rv.returncode == 1 and "less_than: false" in rv.stdout and rv.stdout.count(": true") == len(STEMS) - 1
    # (e) the Python driver: one rank, then two ranks over torch.distributed (gloo here: both ranks share this box's one GPU; RCCL needs a GPU per rank)
    penv = dict(env, PYTHONPATH=str(ROOT) + os.pathsep + env.get("PYTHONPATH", ""), BPG_BATCH_BACKEND="gloo")
    for name, extra in (("python1", []), ("python2", ["--gpus", "2"])):
        d = _stage(tmp_path / name)
        r = subprocess.run([sys.executable, "-m", "bulletproofs_gadgets_amd.cli", "prover", "--batch", "batch.txt"] + extra, cwd=d, env=penv,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert _files(d) == want, name
        lines = [l for l in r.stdout.strip().splitlines() if ": " in l and "constraints" in l]
        assert len(lines) == len(STEMS)
        for s, l in zip(STEMS, lines):                                                   # the gathered bytes are the files' bytes
            assert l == "%s: %d constraints, %d-byte proof, sha256 %s" % (s, counts[s], len(want[s + ".proof"]), hashlib.sha256(want[s + ".proof"]).hexdigest()[:16])
        rv = subprocess.run([sys.executable, "-m", "bulletproofs_gadgets_amd.cli", "verifier", "--batch", "batch.txt"] + extra, cwd=d, env=penv,
                            capture_output=True, text=True, timeout=900)
        assert rv.returncode == 0 and [l for l in rv.stdout.strip().splitlines() if l.endswith(": true")] == ["%s: true" % s for s in STEMS], rv.stdout + rv.stderr[-3000:]


@pytest.mark.gpu
def test_cfg5_batch_of_full_size_merkle_stems(tmp_path):
    """BASELINE.json config 5 as files: independent 2^20-multiplier proofs from a batch of .gadgets stems (256-leaf MiMC Merkle trees, every leaf a
    witness hashed by hash_witness: n = 744,552, N = 2^20).  `bpg_prover --batch` with its worker threads (one engine context each, all
    commitments of a stem in one launch, chains drawn beside the assembly) writes the files of one `bpg_prover NAME` run per stem; `bpg_verifier
    --batch` accepts them and rejects a tampered one."""
    from bulletproofs_gadgets_amd import workloads
    prover_bin, verifier_bin = bpg_build.build_cli()
    env = dict(os.environ, **ENV)
    names = ["tree%d" % k for k in range(4)]
    dirs = {}
    for mode in ("batch", "per_stem"):
        d = tmp_path / mode
        d.mkdir()
        for k, nm in enumerate(names):
            n = workloads.merkle_tree_files(str(d / nm), leaves=256, seed=k)
        (d / "batch.txt").write_text("\n".join(names) + "\n")
        dirs[mode] = d
    assert n == 744552
    r = subprocess.run([str(prover_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines() == ["%s: 1489617 constraints, 1792-byte proof" % nm for nm in names]
    for nm in names:
        r = subprocess.run([str(prover_bin), nm], cwd=dirs["per_stem"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip() == "1489617", r.stdout + r.stderr
        for ext in (".coms", ".proof"):
            assert (dirs["batch"] / (nm + ext)).read_bytes() == (dirs["per_stem"] / (nm + ext)).read_bytes(), nm + ext
    assert len({(dirs["batch"] / (nm + ".proof")).read_bytes() for nm in names}) == 4          # four different statements
    rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert rv.returncode == 0 and rv.stdout.strip().splitlines() == ["%s: true" % nm for nm in names], rv.stdout + rv.stderr
    bad = bytearray((dirs["batch"] / "tree2.proof").read_bytes()); bad[100] ^= 1
    (dirs["batch"] / "tree2.proof").write_bytes(bytes(bad))
    rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert rv.returncode == 1 and "tree2: false" in rv.stdout and rv.stdout.count(": true") == 3
