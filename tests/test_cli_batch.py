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
    r = subprocess.run([str(prover_bin), "--batch", str(f), "--gpus"], capture_output=True, text=True)          # a trailing option without its value
    assert r.returncode == 2 and "needs a value" in r.stderr
    # --gpus N re-executes this image per rank: refused under a profiler, whose preloaded library has initialised the GPU before main()
    r = subprocess.run([str(prover_bin), "--batch", str(f), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(os.environ, ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"))
    assert r.returncode == 2 and "profile ONE rank" in r.stderr
    r = subprocess.run([str(prover_bin)], capture_output=True, text=True)
    assert r.returncode == 2 and "--batch FILE" in r.stderr


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
    """BASELINE.json config 5 in its own shape: EIGHT independent 2^20-multiplier proofs from a batch of .gadgets stems (256-leaf MiMC Merkle
    trees, every leaf a witness hashed by hash_witness: n = 744,552, N = 2^20).  `bpg_prover --batch` with its worker threads (one engine
    context each, all commitments of a stem in one launch, chains drawn beside the assembly) writes the files of one `bpg_prover NAME` run per
    stem - on one rank and on two ranks of two workers (two processes on this card) - `bpg_verifier --batch` accepts them and rejects a
    tampered one; two of the proofs are checked by the ORACLE verifier on the verifier-side assembly of their files, and one .coms file equals
    the one the independent Python restatement (tests/golden/pyref_r1cs.py under the driver's orchestration) writes for the same stem."""
    import sys as _sys
    _sys.path.insert(0, str(ROOT / "tests" / "golden"))
    import bulletproofs_gadgets_amd as bpg
    from bulletproofs_gadgets_amd import workloads
    import oracle_lib as O
    import gen_assembly_fixtures as GA
    prover_bin, verifier_bin = bpg_build.build_cli()
    env = dict(os.environ, **ENV)
    names = ["tree%d" % k for k in range(8)]
    dirs = {}
    for mode in ("batch", "batch2", "per_stem"):
        d = tmp_path / mode
        d.mkdir()
        for k, nm in enumerate(names):
            n = workloads.merkle_tree_files(str(d / nm), leaves=256, seed=k)
        (d / "batch.txt").write_text("\n".join(names) + "\n")
        dirs[mode] = d
    assert n == 744552
    summary = ["%s: 1489617 constraints, 1792-byte proof" % nm for nm in names]
    r = subprocess.run([str(prover_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines() == summary
    # two ranks (two processes sharing this box's one GPU), two workers each: eight 2^20 contexts' worth of buffers on one card
    r = subprocess.run([str(prover_bin), "--batch", "batch.txt", "--gpus", "2", "--workers", "2"], cwd=dirs["batch2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().splitlines() == summary
    for nm in names:
        r = subprocess.run([str(prover_bin), nm], cwd=dirs["per_stem"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip() == "1489617", r.stdout + r.stderr
        for ext in (".coms", ".proof"):
            want = (dirs["per_stem"] / (nm + ext)).read_bytes()
            assert (dirs["batch"] / (nm + ext)).read_bytes() == want, nm + ext
            assert (dirs["batch2"] / (nm + ext)).read_bytes() == want, nm + ext + " (two ranks)"
    assert len({(dirs["batch"] / (nm + ".proof")).read_bytes() for nm in names}) == 8          # eight different statements
    rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert rv.returncode == 0 and rv.stdout.strip().splitlines() == ["%s: true" % nm for nm in names], rv.stdout + rv.stderr
    # the oracle verifier (CPU, one 2N-term MSM) on the verifier-side assembly of two of the stems: .gadgets + .inst + .coms only
    ctx = bpg.Context(0)
    ctx.gens_ensure(1 << 20)
    G, Hh = ctx.gens_export(0, 1 << 20)
    og = O.Gens(compressed=(G, Hh))
    assert og.export(0, 2048) == O.Gens(2048).export(0, 2048)
    for nm in (names[0], names[5]):
        cwd = os.getcwd(); os.chdir(dirs["batch"])               # the transcript label is the stem as the prover was given it: "tree0", not a path
        try:
            v, tv = cli.assemble_verifier(nm)
        finally:
            os.chdir(cwd)
        vi = v.instance()
        assert (vi.n, vi.m) == (744552, 1024)
        oc = O.FlatCircuit(vi.n, vi.m, None, None, None, vi.row_ptr, vi.term_var, vi.term_coef, vi.coef)
        proof = (dirs["batch"] / (nm + ".proof")).read_bytes()
        assert O.verify(og, tv.state, oc, vi.commitments, proof) == 0, nm
        bad = bytearray(proof); bad[70] ^= 4
        assert O.verify(og, tv.state, oc, vi.commitments, bytes(bad)) != 0
    ctx.close()
    # the .coms file of one stem against the independent Python restatement of gadgets + driver (same blinding seed): 1,024 commitment lines
    _, coms_text = GA.example_through_pyref(dirs["batch"] / "tree3", seed=ENV["BPG_CLI_SEED"].encode())
    assert coms_text == (dirs["batch"] / "tree3.coms").read_text() and len(coms_text.splitlines()) == 1024
    bad = bytearray((dirs["batch"] / "tree2.proof").read_bytes()); bad[100] ^= 1
    (dirs["batch"] / "tree2.proof").write_bytes(bytes(bad))
    rv = subprocess.run([str(verifier_bin), "--batch", "batch.txt"], cwd=dirs["batch"], env=env, capture_output=True, text=True, timeout=600)
    assert rv.returncode == 1 and "tree2: false" in rv.stdout and rv.stdout.count(": true") == 7


@pytest.mark.gpu
def test_a_broken_stem_fails_alone(tmp_path):
    """One stem of a batch cannot be proved (its .wtns is missing, another's .gadgets is garbled): that stem is reported FAILED with the reason, every
    other stem is still proved - also with ONE worker thread, which used to stop at the first failure - and the command exits 101, the exit code of
    the reference's panic for the run of that stem (one prover run per stem: .github/workflows/integration_tests.yml:19-58)."""
    prover_bin, verifier_bin = bpg_build.build_cli()
    env = dict(os.environ, **ENV)
    ref = _stage(tmp_path / "ref")
    r = subprocess.run([str(prover_bin), "--batch", "batch.txt"], cwd=ref, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    want = _files(ref)
    for extra in (["--workers", "1"], ["--workers", "3"], ["--gpus", "2", "--workers", "1"]):
        d = _stage(tmp_path / ("broken" + "_".join(extra)))
        (d / "equality.wtns").unlink()
        (d / "or3.gadgets").unlink()                               # (the staged copies keep the read-only mode of the fixtures)
        (d / "or3.gadgets").write_text("BOUND W0 I0 I1\nFROBNICATE W0\n")
        r = subprocess.run([str(prover_bin), "--batch", "batch.txt"] + extra, cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 101, r.stdout + r.stderr
        lines = dict(l.split(": ", 1) for l in r.stdout.strip().splitlines())
        assert list(lines) == STEMS
        assert lines["equality"].startswith("FAILED (") and "equality.wtns" in lines["equality"], lines["equality"]
        assert lines["or3"].startswith("FAILED ("), lines["or3"]
        for s in STEMS:
            if s in ("equality", "or3"):
                assert not (d / (s + ".proof")).exists()
                continue
            assert "constraints" in lines[s], (s, lines[s])
            for e in (".coms", ".proof"):
                assert (d / (s + e)).read_bytes() == want[s + e], s + e
        assert "equality" in r.stderr or "or3" in r.stderr          # the first error text on stderr names its stem
    # the Python driver: same behaviour, and with two ranks the rank that owns a broken stem still joins the gather the other one waits in
    penv = dict(env, PYTHONPATH=str(ROOT) + os.pathsep + env.get("PYTHONPATH", ""), BPG_BATCH_BACKEND="gloo")
    for extra in ([], ["--gpus", "2"]):
        d = _stage(tmp_path / ("pybroken" + "_".join(extra)))
        (d / "equality.wtns").unlink()
        (d / "or3.gadgets").unlink()
        (d / "or3.gadgets").write_text("BOUND W0 I0 I1\nFROBNICATE W0\n")
        r = subprocess.run([sys.executable, "-m", "bulletproofs_gadgets_amd.cli", "prover", "--batch", "batch.txt"] + extra, cwd=d, env=penv,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 101, r.stdout[-2000:] + r.stderr[-3000:]
        lines = dict(l.split(": ", 1) for l in r.stdout.strip().splitlines() if ": " in l and l.split(": ")[0] in STEMS)
        assert list(lines) == STEMS and lines["equality"] == "FAILED" and lines["or3"] == "FAILED"
        for s in STEMS:
            if s not in ("equality", "or3"):
                assert "constraints" in lines[s] and (d / (s + ".proof")).read_bytes() == want[s + ".proof"], s
