#!/usr/bin/env python3
"""Chain throughput of ONE host thread: TranscriptRng draws per second with 1, 2, 4, 8 generators in lockstep (merlin.hpp strobe_rng_bulk64_x8)."""
import ctypes as C, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bulletproofs_gadgets_amd as bpg
t = bpg.Transcript(b"lanes"); vb = bytes(32); count = 200000
for lanes in (1, 2, 4, 8):
    seeds = b"".join(bytes([v + 1]) * 32 for v in range(lanes)); out = C.create_string_buffer(lanes * 64 * count)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        rc = bpg.lib().bpg_rng_draws_multi(t.state, C.c_uint64(1), vb, C.c_uint32(lanes), seeds, (C.c_uint64 * lanes)(*([0] * lanes)), C.c_uint64(count), out)
        dt = time.perf_counter() - t0
        assert rc == 0
        best = dt if best is None else min(best, dt)
    print("%d lanes: %.1f ns per draw and lane, %.1f ns per draw; a 2^20 proof's chain (1,986,776 draws): %.0f ms per lane group, %.1f ms per chain"
          % (lanes, best / count * 1e9, best / count / lanes * 1e9, best / count * 1986776 * 1e3, best / count / lanes * 1986776 * 1e3), flush=True)
