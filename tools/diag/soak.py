"""Soak: many proofs with different seeds, every one verified on the GPU; a few contexts in flight. Catches rare races."""
import sys, time, threading, hashlib
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
def job(k, leaves, count, out):
    ctx = bpg.Context(0)
    a = workloads.merkle_full_tree(ctx, leaves=leaves, seed=None if k == 0 else k)
    inst = a.prover.instance(); state = a.transcript.state
    ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
    coms = b"".join(a.commitments)
    bad = 0; digest = hashlib.sha256()
    for i in range(count):
        seed = hashlib.sha256(b"soak%d-%d" % (k, i)).digest()
        proof, _ = res.prove(state, inst.v_blinding, seed, 0)
        digest.update(proof)
        if res.verify(state, coms, proof) != 0: bad += 1
        if i % 7 == 0:                                   # determinism: same seed -> same bytes
            again, _ = res.prove(state, inst.v_blinding, seed, 0)
            if again != proof: bad += 1000
        if i % 3 == 0:                                   # speculative blinding stream: started early, started for another seed, started and abandoned
            ctx.blinding_begin(state, inst.v_blinding, seed, a.gens_capacity)
            if i % 2: time.sleep(0.002 * (i % 5))
            early, _ = res.prove(state, inst.v_blinding, seed, 0)
            if early != proof: bad += 100000
            ctx.blinding_begin(state, inst.v_blinding, hashlib.sha256(seed).digest(), a.gens_capacity)
            wrong, _ = res.prove(state, inst.v_blinding, seed, 0)
            if wrong != proof: bad += 1000000
            ctx.blinding_begin(state, inst.v_blinding, seed, a.gens_capacity)     # left running: the next prove (other seed) or begin drops it
    out[k] = (bad, digest.hexdigest()[:16])
for leaves, count, nthreads in ((32, 120, 4), (512, 12, 3)):
    out = {}
    th = [threading.Thread(target=job, args=(k, leaves, count, out)) for k in range(nthreads)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    print("leaves=%d: %d threads x %d proofs in %.1f s -> %s" % (leaves, nthreads, count, time.time() - t0, out), flush=True)
    assert len(out) == nthreads and all(v[0] == 0 for v in out.values())
print("soak ok")
