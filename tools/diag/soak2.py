#!/usr/bin/env python3
"""Soak of the headline arrangement: P proving streams (contexts, threads) share W chain-worker threads and prove STEPS 2^20 proofs; every proof
is verified on the GPU and a sample is compared with stand-alone proofs of the same seeds.  usage: soak2.py [steps=60] [streams=2] [workers=12]"""
import pathlib, sys, threading, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lanes = []
for p in range(P):
    c = bpg.Context(0); c.set_chain_workers(max(1, -(-W // P)))
    if p == 0:
        a = workloads.merkle_full_tree(c, leaves=512); inst, state = a.prover.instance(), a.transcript.state
    c.gens_ensure(a.gens_capacity)
    lanes.append((c, c.upload(inst)))
seeds = [bytes([i & 255, i >> 8, 7]) + bytes(29) for i in range(steps)]
outs, errs = [None] * steps, []
def work(p):
    try:
        c, r = lanes[p]; mine = seeds[p::P]; ahead = max(1, -(-W // P)); queued = 0
        for i, s in enumerate(mine):
            while queued < len(mine) and queued <= i + ahead:
                c.blinding_begin(state, inst.v_blinding, mine[queued], inst.n); queued += 1
            outs[p + i * P] = r.prove(state, inst.v_blinding, s, 0)[0]
    except Exception as e:
        errs.append(repr(e))
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(p,)) for p in range(P)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
assert not errs, errs
coms = b"".join(a.commitments)
bad = [i for i, pr in enumerate(outs) if lanes[0][1].verify(state, coms, pr) != 0]
assert not bad, bad
assert len(set(outs)) == steps
for i in (0, 1, steps // 2, steps - 1):
    assert lanes[0][1].prove(state, inst.v_blinding, seeds[i], 0)[0] == outs[i], i
print("soak ok: %d proofs in %.2f s (%.1f ms per proof), all verified, samples equal to stand-alone proofs" % (steps, dt, dt / steps * 1e3))
