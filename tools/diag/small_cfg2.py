import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.bounds_check_64(ctx, seed=0); inst = a.prover.instance(); state = a.transcript.state
ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
for i in range(3):
    t0 = time.perf_counter(); _, _, tm = res.prove(state, inst.v_blinding, bytes(32), 0, timings=True); dt = time.perf_counter() - t0
    print("wall %.2f ms" % (dt * 1e3), {k: round(v, 2) for k, v in tm.items()})
ctx.profile_set(2); res.prove(state, inst.v_blinding, bytes(32), 0); rep = ctx.profile_report(); ctx.profile_set(0)
ks = sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])
print("kernels total %.2f ms: " % sum(v["total_ms"] for _, v in ks) + "  ".join("%s %.2f x%d" % (k.replace("k_", ""), v["total_ms"], v["count"]) for k, v in ks[:16]))
