"""Where the untimed host set-up of the 2^20 workload goes: commitments, gadget assembly, flattening, upload."""
import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads as W
ctx = bpg.Context(0)
leaves = 512
leaf_be = [bytes.fromhex("0522a64d7b931e21760cf955a15fcc793e8a52b42a56ab03afddec8beb668749")] * leaves
root = bpg.be_to_scalar(bytes.fromhex("038c137beec8e2edfb5c48cbd063f04e569139d2221a4eb7befb85aa1bf8ba40"))
for rep in range(3):
    t0 = time.perf_counter()
    t = bpg.Transcript(b"MerkleTree"); p = bpg.Prover(ctx, t)
    scalars, wcoms, wvars = W.commit_all_single(p, leaf_be, [W.blinding("x", i) for i in range(leaves)])
    t1 = time.perf_counter()
    pattern = W.full_tree_pattern(leaves)
    g = bpg.MerkleTree256(root, [], W.vars_to_lc(wvars), pattern)
    t2 = time.perf_counter()
    g.prove(p, [], [])
    t3 = time.perf_counter()
    inst = p.instance()
    t4 = time.perf_counter()
    res = ctx.upload(inst)
    t5 = time.perf_counter()
    print("commit %.3f  gadget-construct %.3f  assembly %.3f  instance() %.3f  upload %.3f  (n=%d)" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, inst.n), flush=True)
