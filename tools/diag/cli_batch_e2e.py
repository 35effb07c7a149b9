#!/usr/bin/env python3
"""BASELINE.json config 5 as files: a batch of 8 independent 2^20-multiplier .gadgets stems (256-leaf MiMC Merkle trees with seeded leaves, every leaf
a witness hashed by hash_witness: n = 744,552 -> N = 2^20) through the native batch driver on ONE GPU - `bpg_prover --batch` (one process; W worker threads with an engine
context each: HIP start-up, generators and fold tables once; a proof is 0.5 s of host work and 30 ms of GPU, so workers multiply the rate) against the reference's shape, one `bpg_prover NAME` process per stem
(src/bin/prover.rs:47-100; .github/workflows/integration_tests.yml:19-58) - and `bpg_verifier --batch`.  Same .coms / .proof bytes both ways.
usage: cli_batch_e2e.py [stems=8] [leaves=256]"""
import json, os, pathlib, subprocess, sys, tempfile, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from bulletproofs_gadgets_amd import workloads, build
stems = int(sys.argv[1]) if len(sys.argv) > 1 else 8
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prover_bin, verifier_bin = build.build_cli()
env = dict(os.environ, BPG_CLI_SEED="batch-e2e", BPG_CLI_RNG_SEED="33" * 32)
out = {"stems": stems, "leaves": leaves}
files = {}
for mode in ("batch_w1", "batch_w4", "batch_w8", "per_stem"):
    d = pathlib.Path(tempfile.mkdtemp())
    names = ["tree%d" % k for k in range(stems)]
    for k, nm in enumerate(names):
        out["n"] = workloads.merkle_tree_files(str(d / nm), leaves=leaves, seed=k)
    (d / "batch.txt").write_text("\n".join(names) + "\n")
    t0 = time.perf_counter()
    if mode.startswith("batch"):
        r = subprocess.run([str(prover_bin), "--batch", "batch.txt", "--workers", mode.split("_w")[1]], cwd=d, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out["summary"] = r.stdout.strip().splitlines()[:2]
    else:
        for nm in names:
            r = subprocess.run([str(prover_bin), nm], cwd=d, env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr
    dt = time.perf_counter() - t0
    out[mode] = {"wall_s": round(dt, 3), "s_per_proof": round(dt / stems, 3)}
    files[mode] = [((d / (nm + ".coms")).read_bytes(), (d / (nm + ".proof")).read_bytes()) for nm in names]
    if mode == "batch_w4":
        t0 = time.perf_counter()
        v = subprocess.run([str(verifier_bin), "--batch", "batch.txt"], cwd=d, env=env, capture_output=True, text=True, timeout=900)
        assert v.returncode == 0 and v.stdout.count(": true") == stems, v.stdout + v.stderr
        out["verify_batch"] = {"wall_s": round(time.perf_counter() - t0, 3)}
out["identical_files"] = all(files[m] == files["per_stem"] for m in files)
print(json.dumps(out, indent=1))
