import sys, time
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=8, seed=7); inst = a.prover.instance(); st = a.transcript.state
ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
for i in range(3):
    t0 = time.perf_counter(); res.prove(st, inst.v_blinding, bytes([i]) * 32, 0); print("proof %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
