import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=512, seed=None)
inst = a.prover.instance(); state = a.transcript.state
ctx.gens_ensure(a.gens_capacity)
res = ctx.upload(inst)
for flags in (0, 4):
    res.prove(state, inst.v_blinding, bytes(32), flags)
    _, _, tm = res.prove(state, inst.v_blinding, bytes(32), flags, timings=True)
    print("flags", flags, {k: round(v, 2) for k, v in tm.items()})
    ctx.profile_set(2); res.prove(state, inst.v_blinding, bytes(32), flags); rep = ctx.profile_report(); ctx.profile_set(0)
    ks = sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])
    print("   kernels total %.2f ms: " % sum(v["total_ms"] for _, v in ks) + "  ".join("%s %.2f" % (k.replace("k_", ""), v["total_ms"]) for k, v in ks[:14]))
