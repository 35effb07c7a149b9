import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.bounds_check_64(ctx, seed=0); inst = a.prover.instance(); state = a.transcript.state
ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
for i in range(12): res.prove(state, inst.v_blinding, bytes(32), 0)
ts = []
for i in range(30):
    t0 = time.perf_counter(); res.prove(state, inst.v_blinding, bytes(32), 0); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort(); print("cfg2 steady: median %.2f ms min %.2f max %.2f" % (ts[15], ts[0], ts[-1]))
