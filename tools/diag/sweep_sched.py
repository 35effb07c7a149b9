import sys, time, os
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=512, seed=None)
inst = a.prover.instance(); state = a.transcript.state
ctx.gens_ensure(a.gens_capacity)
res = ctx.upload(inst)
for i in range(2): res.prove(state, inst.v_blinding, bytes(32), 4)
ts = []
for i in range(5):
    t0 = time.perf_counter(); res.prove(state, inst.v_blinding, bytes([i]) * 32, 4); ts.append((time.perf_counter() - t0) * 1e3)
_, _, tm = res.prove(state, inst.v_blinding, bytes(32), 4, timings=True)
print("TT_LG=%s GROUP=%s: expanded-mode proof median %.2f ms  ipa %.2f (msm %.2f fold %.2f)" % (os.environ.get("BPG_TT_LG"), os.environ.get("BPG_FOLD_GROUP"), sorted(ts)[2], tm["ipa"], tm["ipa_msm"], tm["ipa_fold"]), flush=True)
