#!/usr/bin/env python3
"""Soak: `threads` proving streams (engine contexts on a shared chain pool) prove the 2^16 MiMC-preimage circuit (cfg 3) and the 2^20 Merkle circuit for
`seconds`, cycling through a few seeds whose proofs were made once, alone, at the start: every proof must equal its reference bytes whatever else is
in flight (the shared-device kernel variants switch on and off as proofs of the other threads come and go), and neither host RSS nor device memory
may creep.  usage: soak.py [seconds=60] [threads=6]"""
import os, pathlib, sys, threading, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rss = lambda: int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e9
dev = lambda: (lambda f, t: (t - f) / 1e9)(*torch.cuda.mem_get_info(0))
ctx0 = bpg.Context(0, profile="serving", blocking_sync=True)
cases = []
for mk in (lambda: workloads.mimc_preimage(ctx0, nbytes=2130, seed=0, label=b"MiMCHash"), lambda: workloads.merkle_full_tree(ctx0, leaves=512, seed=None)):
    a = mk(); inst = a.prover.instance(); state = a.transcript.state
    ctx0.gens_ensure(a.gens_capacity)
    res = ctx0.upload(inst)
    seeds = [bytes([k + 1]) * 32 for k in range(4)]
    refs = [res.prove(state, inst.v_blinding, s, 0)[0] for s in seeds]
    assert res.verify(state, b"".join(a.commitments), refs[0]) == 0
    res.free()
    cases.append((a, inst, state, seeds, refs))
pool = bpg.ChainPool([1] * min(nthreads, 8) + [8])
ctxs = [bpg.Context(0, profile="serving", blocking_sync=True) for _ in range(nthreads)]
ress = []
for c in ctxs:
    c.attach_chain_pool(pool, 2)
    c.gens_ensure(1 << 20)
    ress.append([c.upload(inst) for (_, inst, _, _, _) in cases])
count = [0] * nthreads; bad = []; stop = time.time() + seconds
def work(k):
    i = k
    while time.time() < stop and not bad:
        ci = 0 if (i % 5) else 1                              # four small proofs, then a 2^20 one
        a, inst, state, seeds, refs = cases[ci]
        s = i % len(seeds)
        if i % 3 == 0:
            ctxs[k].blinding_begin(state, inst.v_blinding, seeds[s], inst.n)      # sometimes with the chain drawn ahead on the pool
        p = ress[k][ci].prove(state, inst.v_blinding, seeds[s], 0)[0]
        if p != refs[s]:
            bad.append((k, i, ci, s))
        count[k] += 1; i += nthreads
r0, d0 = None, None
th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
for t in th: t.start()
time.sleep(min(10.0, seconds / 4)); r0, d0 = rss(), dev()       # after warm-up: every workspace has been sized
for t in th: t.join()
r1, d1 = rss(), dev()
print("soak: %d proofs in %.0f s on %d streams (%s per stream); mismatches: %s; host RSS %.2f -> %.2f GB; device memory %.2f -> %.2f GB"
      % (sum(count), seconds, nthreads, count, bad or "none", r0, r1, d0, d1))
sys.exit(1 if bad or r1 > r0 + 0.5 or d1 > d0 + 1.0 else 0)
