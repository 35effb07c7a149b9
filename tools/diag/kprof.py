#!/usr/bin/env python3
"""Per-kernel HIP-event totals of one resident proof (profile mode 2), averaged over a few proofs, under the current BPG_* knobs.
usage: kprof.py [leaves=512] [proofs=3] [prefetch=0|1]   -> one JSON line: {knobs, wall_ms, gpu_ms, kernels: {name: [count, ms per proof]}}"""
import json, os, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 512
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prefetch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=leaves)
inst, state = a.prover.instance(), a.transcript.state
ctx.gens_ensure(a.gens_capacity)
res = ctx.upload(inst)
seed = lambda i: bytes([i + 1]) * 32
ref = res.prove(state, inst.v_blinding, seed(0), 0)[0]
ctx.profile_set(2)
t0 = time.perf_counter()
if prefetch:
    ctx.blinding_begin(state, inst.v_blinding, seed(0), inst.n)
for i in range(proofs):
    if prefetch and i + 1 < proofs:
        ctx.blinding_begin(state, inst.v_blinding, seed(i + 1), inst.n)
    p = res.prove(state, inst.v_blinding, seed(i), 0)[0]
    if i == 0:
        assert p == ref
wall = (time.perf_counter() - t0) / proofs * 1e3
k = ctx.profile_report()
ctx.profile_set(0)
assert res.verify(state, b"".join(a.commitments), p) == 0
kern = {n: [v["count"] // proofs, round(v["total_ms"] / proofs, 4)] for n, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"])}
print(json.dumps({"knobs": {e: os.environ[e] for e in os.environ if e.startswith("BPG_") or e.startswith("GPU_MAX")}, "leaves": leaves, "prefetch": prefetch,
                  "wall_ms": round(wall, 2), "gpu_ms": round(sum(v[1] for v in kern.values()), 3), "kernels": kern}))
