import sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
a = workloads.merkle_full_tree(ctx, leaves=512, seed=None)
inst = a.prover.instance(); state = a.transcript.state
for workers in (8, 12):
    pool = bpg.ProverPool(0, workers=workers, gens_capacity=a.gens_capacity)
    items = [(inst, state, inst.v_blinding, bytes([i + 1]) * 32, 0) for i in range(2 * workers)]
    pool.prove_batch(items[:workers])                      # warm-up: workspaces
    t0 = time.perf_counter(); out = pool.prove_batch(items); dt = time.perf_counter() - t0
    print("pool workers=%d: %d proofs of q=%d in %.2f s -> %.1f M constraints/s (upload per item included)" % (workers, len(items), inst.q, dt, inst.q * len(items) / dt / 1e6), flush=True)
    pool.close()
