// standalone micro-benchmark (diagnostic, not part of the product): issue cost per wave64 instruction on gfx950 for the operations a big-integer
// multiplication can be built from.  Every kernel runs REPS x 64 instructions of one kind per loop trip, 8 waves per SIMD resident.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bench_issue tools/diag/bench_issue.hip && /tmp/bench_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int V> __global__ void __launch_bounds__(256) kb(uint32_t *out, uint32_t iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = t * 2654435761u + 1, b = t ^ 0x9e3779b9u, c = t + 77, d = ~t;
    uint64_t p = t, q = t + 1, r = t + 2, s = t + 3, cc; uint32_t c2 = t, e0 = t * 3, e1 = t * 5;
    double fp = 1.0 + t * 1e-9, fq = 1.5 + t * 1e-9, fr = 0.5 + t * 1e-9, fs = 0.25 + t * 1e-9; const double fa = 1.0000001, fb = 1e-30;
    float gp = 1.0f + t * 1e-6f, gq = 1.5f, gr = 0.5f, gs = 0.25f; const float ga = 1.0001f, gb = 1e-20f;
    for (uint32_t i = 0; i < iters; i++) {
        // floating-point candidates for a limb product (round 4): is a double-precision FMA cheaper to issue than v_mad_u64_u32?
        if (V == 21) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                                           : "+v"(fp), "+v"(fq), "+v"(fr), "+v"(fs) : "v"(fa), "v"(fb));) }
        if (V == 22) { REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                                           : "+v"(fp), "+v"(fq), "+v"(fr), "+v"(fs) : "v"(fb));) }
        if (V == 23) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                           : "+v"(gp), "+v"(gq), "+v"(gr), "+v"(gs) : "v"(ga), "v"(gb));) }
        if (V == 24) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4"
                                           : "+v"(fp), "+v"(fq), "+v"(fr), "+v"(fs) : "v"(fa));) }
        if (V == 0) { REP16(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_mad_u64_u32 %1, %4, %6, %7, %1\n\tv_mad_u64_u32 %2, %4, %7, %8, %2\n\tv_mad_u64_u32 %3, %4, %8, %5, %3"
                                          : "+v"(p), "+v"(q), "+v"(r), "+v"(s), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }                  // 4 independent accumulators
        if (V == 1) { REP16(asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %5, %2, %0"
                                          : "+v"(p), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }                                              // one accumulator (a column)
        if (V == 2) { REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 3) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %5, vcc\n\tv_addc_co_u32 %2, vcc, %2, %6, vcc\n\tv_addc_co_u32 %3, vcc, %3, %7, vcc"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s) : "vcc");) }
        if (V == 4) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 29\n\tv_alignbit_b32 %1, %2, %1, 29\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_alignbit_b32 %3, %0, %3, 29"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (V == 5) { REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n\tv_mad_u32_u24 %1, %5, %6, %1\n\tv_mad_u32_u24 %2, %6, %7, %2\n\tv_mad_u32_u24 %3, %7, %4, %3"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 6) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %5\n\tv_mul_hi_u32 %2, %2, %6\n\tv_mul_hi_u32 %3, %3, %7"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 7) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %5\n\tv_mul_lo_u32 %2, %2, %6\n\tv_mul_lo_u32 %3, %3, %7"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 8) { REP16(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n\tv_dot4_u32_u8 %1, %5, %6, %1\n\tv_dot4_u32_u8 %2, %6, %7, %2\n\tv_dot4_u32_u8 %3, %7, %4, %3"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 9) { REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %5\n\tv_mul_hi_u32_u24 %2, %2, %6\n\tv_mul_hi_u32_u24 %3, %3, %7"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 10) { REP16(asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
                                           : "+v"(p), "+v"(a), "=&s"(cc) : "v"(b), "v"(c), "v"(d));) }                                          // the product's column pattern: 2 MAD + 2 carry
        if (V == 15) { REP16(asm volatile("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2\n\ts_nop 0"
                                           : "+v"(p), "+v"(a), "=&s"(cc) : "v"(b), "v"(c), "v"(d));) }                                          // the same with one s_nop per four instructions (counted as 4)
        if (V == 16) { REP16(asm volatile("v_mad_u64_u32 %0, %2, %4, %5, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2\n\tv_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2\n\tv_mov_b32 %3, %1"
                                           : "+v"(p), "+v"(a), "=&s"(cc), "+v"(c2) : "v"(b), "v"(c), "v"(d));) }                                  // the same with one v_mov per four instructions (counted as 4)
        // a radix-2^29 column: k MADs on one accumulator, then limb = lo & M29, carry = {hi >> 29 : alignbit(hi, lo, 29)} (counted as 4 "instructions" per group so that the
        // printed figure x 4 = ns per column)
        if (V == 17) { REP16(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_mad_u64_u32 %0, %4, %7, %8, %0\n\tv_mad_u64_u32 %0, %4, %8, %5, %0\n\tv_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
                                           "v_and_b32 %1, 0x1fffffff, %2\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_lshrrev_b32 %3, 29, %3"
                                           : "+v"(p), "+v"(c2), "+v"(e0), "+v"(e1), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }      // 5 MADs + 3 plain
        if (V == 18) { REP16(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_mad_u64_u32 %0, %4, %7, %8, %0\n\tv_mad_u64_u32 %0, %4, %8, %5, %0\n\tv_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
                                           "v_mad_u64_u32 %0, %4, %6, %8, %0\n\tv_mad_u64_u32 %0, %4, %7, %5, %0\n\tv_mad_u64_u32 %0, %4, %8, %6, %0\n\tv_mad_u64_u32 %0, %4, %5, %5, %0\n\t"
                                           "v_and_b32 %1, 0x1fffffff, %2\n\tv_alignbit_b32 %2, %3, %2, 29\n\tv_lshrrev_b32 %3, 29, %3"
                                           : "+v"(p), "+v"(c2), "+v"(e0), "+v"(e1), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }      // 9 MADs + 3 plain
        if (V == 19) { REP16(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_add_u32 %1, %1, %5\n\tv_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_add_u32 %2, %2, %6\n\tv_mad_u64_u32 %0, %4, %7, %8, %0\n\tv_add_u32 %3, %3, %7\n\tv_mad_u64_u32 %0, %4, %8, %5, %0\n\tv_add_u32 %1, %1, %8"
                                           : "+v"(p), "+v"(c2), "+v"(e0), "+v"(e1), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }      // MAD, add alternating (8 instr, counted 4)
        if (V == 20) { REP16(asm volatile("v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_add_u32 %3, %3, %7\n\tv_add_u32 %1, %1, %8"
                                           : "+v"(p), "+v"(c2), "+v"(e0), "+v"(e1), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }      // MAD + 2 adds (6 instr, counted 4)
        if (V == 11) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_lshl_add_u64 %3, %3, 0, %0"
                                           : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
        if (V == 12) { REP16(asm volatile("v_dot2_u32_u16 %0, %4, %5, %0\n\tv_dot2_u32_u16 %1, %5, %6, %1\n\tv_dot2_u32_u16 %2, %6, %7, %2\n\tv_dot2_u32_u16 %3, %7, %4, %3"
                                           : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 13) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n\tv_pk_mul_lo_u16 %1, %1, %5\n\tv_pk_mul_lo_u16 %2, %2, %6\n\tv_pk_mul_lo_u16 %3, %3, %7"
                                           : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((uint32_t)p), "v"((uint32_t)q), "v"((uint32_t)r), "v"((uint32_t)s));) }
        if (V == 14) { REP16(asm volatile("v_mad_i64_i32 %0, %4, %5, %6, %0\n\tv_mad_i64_i32 %1, %4, %6, %7, %1\n\tv_mad_i64_i32 %2, %4, %7, %8, %2\n\tv_mad_i64_i32 %3, %4, %8, %5, %3"
                                           : "+v"(p), "+v"(q), "+v"(r), "+v"(s), "=&s"(cc) : "v"(a), "v"(b), "v"(c), "v"(d));) }
    }
    out[t] = (uint32_t)(fp + fq + fr + fs) + (uint32_t)(gp + gq + gr + gs) + c2 + e0 + e1 + a + b + c + d + (uint32_t)p + (uint32_t)q + (uint32_t)r + (uint32_t)s + (uint32_t)((p ^ q ^ r ^ s) >> 32);
#endif
}

template <int V> static void run(uint32_t *d_out, const char *name) {
    const uint32_t blocks = 256 * 8, iters = 4000;       // 8 waves per SIMD
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, d_out, 8u);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 4 * iters * 64;          // per-wave instructions issued
    const double per_simd_ns = ms * 1e6 / (wave_instr / 1024.0);         // 1024 SIMDs
    std::printf("{\"op\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instruction_per_simd\": %.3f}\n", name, ms, per_simd_ns);
}

int main() {
    uint32_t *d_out; (void)hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4);
    run<21>(d_out, "v_fma_f64 x4 independent");
    run<22>(d_out, "v_add_f64 x4 independent");
    run<24>(d_out, "v_mul_f64 x4 independent");
    run<23>(d_out, "v_fma_f32 x4 independent");
    run<0>(d_out, "v_mad_u64_u32 x4 independent");
    run<1>(d_out, "v_mad_u64_u32 one accumulator");
    run<10>(d_out, "v_mad_u64_u32 + v_addc (column pattern)");
    run<15>(d_out, "column pattern + s_nop 0 per 4 (time per 4 counted instructions / 4)");
    run<16>(d_out, "column pattern + v_mov per 4 (time per 4 counted instructions / 4)");
    run<14>(d_out, "v_mad_i64_i32 x4 independent");
    run<17>(d_out, "radix-2^29 column: 5 MADs + and, alignbit, lshr (x 4 = ns per column)");
    run<18>(d_out, "radix-2^29 column: 9 MADs + and, alignbit, lshr (x 4 = ns per column)");
    run<19>(d_out, "4 x (MAD, add) (x 4 = ns per 8 instructions)");
    run<20>(d_out, "2 x (MAD, add, add) (x 4 = ns per 6 instructions)");
    run<2>(d_out, "v_add_u32");
    run<3>(d_out, "v_addc_co_u32 chain (vcc)");
    run<4>(d_out, "v_alignbit_b32");
    run<11>(d_out, "v_lshl_add_u64");
    run<5>(d_out, "v_mad_u32_u24");
    run<9>(d_out, "v_mul_hi_u32_u24");
    run<6>(d_out, "v_mul_hi_u32");
    run<7>(d_out, "v_mul_lo_u32");
    run<8>(d_out, "v_dot4_u32_u8");
    run<12>(d_out, "v_dot2_u32_u16");
    run<13>(d_out, "v_pk_mul_lo_u16");
    return 0;
}
