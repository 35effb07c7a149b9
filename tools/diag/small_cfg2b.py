import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.bounds_check_64(ctx, seed=0); inst = a.prover.instance(); state = a.transcript.state
ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
out = []
for i in range(24):
    t0 = time.perf_counter(); _, _, tm = res.prove(state, inst.v_blinding, bytes(32), 0, timings=True); dt = time.perf_counter() - t0
    out.append("%.1f(aiao %.1f s %.1f poly %.1f ipa %.1f)" % (dt * 1e3, tm["msm_aiao"], tm["msm_s"], tm["poly"], tm["ipa"]))
print(" ".join(out))
out = []
for i in range(24):
    t0 = time.perf_counter(); res.prove(state, inst.v_blinding, bytes(32), 0); dt = time.perf_counter() - t0
    out.append("%.1f" % (dt * 1e3))
print("no timings:", " ".join(out))
