"""End to end at 2^20 (commit, assemble, flatten, upload, prove) with and without the speculative blinding stream."""
import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads as W
ctx = bpg.Context(0)
leaves = 512
leaf_be = [bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")] * leaves
root = bpg.be_to_scalar(bytes.fromhex("038c137beec8e2edfb5c48cbd063f04e569139d2221a4eb7befb85aa1bf8ba40"))
ctx.gens_ensure(1 << 20)
pattern = W.full_tree_pattern(leaves)
seed = bytes(range(32))
ref = None
for early in (False, True, False, True, True):
    t0 = time.perf_counter()
    t = bpg.Transcript(b"MerkleTree"); p = bpg.Prover(ctx, t)
    scalars, wcoms, wvars = W.commit_all_single(p, leaf_be, [W.blinding("x", i) for i in range(leaves)])
    if early:
        p.start_blinding(seed, 1 << 20)
    t1 = time.perf_counter()
    bpg.MerkleTree256(root, [], W.vars_to_lc(wvars), pattern).prove(p, [], [])
    t2 = time.perf_counter()
    proof = p.prove(bpg.BulletproofGens(ctx, 1 << 20), seed)
    t3 = time.perf_counter()
    ref = ref or proof
    print("early=%d  commit %.3f  assembly %.3f  prove (flatten + upload + proof) %.3f  total %.3f s  same bytes %s" % (early, t1 - t0, t2 - t1, t3 - t2, t3 - t0, proof == ref), flush=True)
