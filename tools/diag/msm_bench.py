"""MSM microbenchmark through the C ABI test hook bpg_msm_gens: per-kernel HIP-event times at several sizes."""
import sys, time, os, hashlib
sys.path.insert(0, "/root/repo")
import bulletproofs_gadgets_amd as bpg
ctx = bpg.Context(0)
cap = 1 << 20
ctx.gens_ensure(cap)
sizes = [int(x) for x in sys.argv[1:]] or [1 << 10, 1 << 13, 1 << 16, 1 << 18, 1 << 20]
for cnt in sizes:
    rnd = hashlib.shake_256(b"msm-bench%d" % cnt).digest(64 * cnt)
    s = [bytes(rnd[32 * i:32 * i + 31]) + bytes([rnd[32 * i + 31] & 15]) for i in range(cnt)]
    t = [bytes(rnd[32 * (cnt + i):32 * (cnt + i) + 31]) + bytes([rnd[32 * (cnt + i) + 31] & 15]) for i in range(cnt)]
    ctx.msm_gens(0, s, t)
    t0 = time.perf_counter()
    for _ in range(3):
        out = ctx.msm_gens(0, s, t)
    wall = (time.perf_counter() - t0) / 3
    ctx.profile_set(2)
    ctx.msm_gens(0, s, t)
    rep = ctx.profile_report()
    ctx.profile_set(0)
    ks = sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])
    tot = sum(v["total_ms"] for k, v in ks)
    print("terms=%d (2 x %d) wall %.3f ms (incl. upload)  kernels %.3f ms  result %s" % (2 * cnt, cnt, wall * 1e3, tot, out.hex()[:16]))
    print("   " + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"]) for k, v in ks if v["total_ms"] > 0.002), flush=True)
