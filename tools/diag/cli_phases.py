#!/usr/bin/env python3
"""Phase times of the native file driver on 2^20-multiplier stems (BPG_CLI_TIMING=1): one stem alone, then a batch of 4 with 4 workers.
usage: cli_phases.py [leaves=256]"""
import os, pathlib, subprocess, sys, tempfile, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from bulletproofs_gadgets_amd import workloads, build
leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prover_bin, _ = build.build_cli()
env = dict(os.environ, BPG_CLI_SEED="phases", BPG_CLI_RNG_SEED="44" * 32, BPG_CLI_TIMING="1")
d = pathlib.Path(tempfile.mkdtemp())
names = ["tree%d" % k for k in range(4)]
for k, nm in enumerate(names):
    n = workloads.merkle_tree_files(str(d / nm), leaves=leaves, seed=k)
(d / "batch.txt").write_text("\n".join(names) + "\n")
print("n =", n, "files:", {e: (d / ("tree0." + e)).stat().st_size for e in ("gadgets", "inst", "wtns")})
for label, cmd in (("one stem", [str(prover_bin), "tree0"]), ("one stem again", [str(prover_bin), "tree1"]),
                   ("batch of 4, 1 worker", [str(prover_bin), "--batch", "batch.txt", "--workers", "1"]),
                   ("batch of 4, 4 workers", [str(prover_bin), "--batch", "batch.txt", "--workers", "4"])):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=600)
    dt = time.perf_counter() - t0
    print("==", label, "rc", r.returncode, "wall %.3f s" % dt)
    print(r.stderr[-3000:])
