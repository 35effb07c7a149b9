#!/usr/bin/env python3
"""One 2^20 proof ALONE on the device (the reference's shape: one prover process per proof, src/bin/prover.rs:47-100): phase times of the library's own
clock (median of `reps` proofs, chain drawn inside the call), the sum of the kernels' HIP-event durations, launches per proof - under the one-shot and the
serving profile, with blocking and with spinning stream waits (the first context of a process decides how its threads wait: one process per setting).
usage: lone_proof.py oneshot|serving blocking|spin [reps=5] [leaf_seed]      (tools/diag/lone_proof.sh runs the four settings)"""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
profile, blocking = sys.argv[1], sys.argv[2] == "blocking"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
leaf_seed = int(sys.argv[4]) if len(sys.argv) > 4 else None
out = {}
if True:
    if True:
        c = bpg.Context(0, profile=profile, blocking_sync=blocking)
        a = workloads.merkle_full_tree(c, leaves=512, seed=leaf_seed)
        inst, state = a.prover.instance(), a.transcript.state
        c.gens_ensure(a.gens_capacity)
        res = c.upload(inst)
        for k in range(2):
            res.prove(state, inst.v_blinding, bytes([k]) * 32, 0)
        tms = [res.prove(state, inst.v_blinding, bytes([10 + k]) * 32, 0, timings=True)[2] for k in range(reps)]
        med = {k: sorted(t[k] for t in tms)[reps // 2] for k in tms[0]}
        c.profile_set(2)
        res.prove(state, inst.v_blinding, bytes([99]) * 32, 0)
        rep = c.profile_report()
        kern = {k: v for k, v in rep.items() if not k.startswith("_")}
        out["%s/%s" % (profile, "blocking" if blocking else "spin")] = {
            "phase_ms": {k: round(v, 3) for k, v in med.items()}, "kernel_ms_sum": round(sum(v["total_ms"] for v in kern.values()), 3),
            "launches": int(sum(v["count"] for v in kern.values())),
            "top": {k: round(v["total_ms"], 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"])[:14]}}
        res.free(); c.close()
print(json.dumps(out, indent=1))
