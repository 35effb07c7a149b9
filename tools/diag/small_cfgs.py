"""prove / verify times of the small BASELINE.json configurations (parity cases, not bench lines)"""
import sys, time
import pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bulletproofs_gadgets_amd as bpg
from bulletproofs_gadgets_amd import workloads
ctx = bpg.Context(0)
for name, mk in (("cfg2 bounds_check_64", lambda: workloads.bounds_check_64(ctx, seed=0)),
                 ("8-leaf merkle 2^14 (cfg 1 size)", lambda: workloads.merkle_full_tree(ctx, leaves=8, seed=7)),
                 ("cfg3 mimc_preimage 2^16", lambda: workloads.mimc_preimage(ctx, nbytes=2130, seed=0, label=b"MiMCHash")),
                 ("32-leaf merkle 2^16", lambda: workloads.merkle_full_tree(ctx, leaves=32, seed=7))):
    a = mk(); inst = a.prover.instance(); state = a.transcript.state
    ctx.gens_ensure(a.gens_capacity); res = ctx.upload(inst)
    for i in range(8): res.prove(state, inst.v_blinding, bytes(32), 0)
    ts = []
    for i in range(9):
        t0 = time.perf_counter(); proof, _ = res.prove(state, inst.v_blinding, bytes([i]) * 32, 0); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[4]
    _, _, tm = res.prove(state, inst.v_blinding, bytes(32), 0, timings=True)
    coms = b"".join(a.commitments)
    t0 = time.perf_counter()
    for i in range(5): ok = res.verify(state, coms, proof)
    dv = (time.perf_counter() - t0) / 5
    print("%-26s n=%d N=%d q=%d: prove %.2f ms (%.0f constraints/s; rng %.2f ipa %.2f)  verify %.2f ms ok=%s" % (name, inst.n, a.gens_capacity, inst.q, dt * 1e3, inst.q / dt, tm["rng_host"], tm["ipa"], dv * 1e3, ok == 0), flush=True)
    res.free()
