#!/usr/bin/env python3
"""Workload for the instruction-count passes (tools/diag/insts.sh): `proofs` resident 2^20 proofs on ONE stream with the chain drawn ahead, under the
current BPG_* knobs (BPG_PROFILE=serving BPG_FOLD_ADAPT=2 = the kernel variants of the concurrent mix), then the field-multiplication
microbenchmark (k_bench_fe_mul) as the yardstick of VALU issue.  usage: insts_workload.py [leaves=512] [proofs=2]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 512
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=leaves)
inst, state = a.prover.instance(), a.transcript.state
ctx.gens_ensure(a.gens_capacity)
res = ctx.upload(inst)
seed = lambda i: bytes([i + 1]) * 32
for i in range(proofs):
    ctx.blinding_begin(state, inst.v_blinding, seed(i), inst.n)
    p = res.prove(state, inst.v_blinding, seed(i), 0)[0]
assert res.verify(state, b"".join(a.commitments), p) == 0
try:
    sched = ctx.schedule()
except KeyError:            # a library built before round 4 (A/B runs through BPG_LIB_PATH)
    sched = None
print("fe_mul_per_s", ctx.bench_fe_mul(2000), "proofs", proofs, "schedule", sched)
