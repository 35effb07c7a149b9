// GF(2^255-19) for gfx950: 8 x 32-bit saturated limbs, values kept anywhere in [0, 2^256) ("weakly reduced",
// 2^256 = 38 mod p).  Products go through 64-bit multiply-adds (v_mad_u64_u32); 8 VGPRs per element keeps an
// extended point at 32 VGPRs.  Replaces curve25519-dalek's FieldElement (not vendored in the reference;
// Cargo.toml:8) underneath the calls at reference src/bin/prover.rs:53-54,92-93.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BPG_HD __host__ __device__ __forceinline__
#define BPG_UNROLL _Pragma("unroll")
#else
#define BPG_HD inline
#define BPG_UNROLL
#endif

namespace bpg {

struct alignas(16) fe { uint32_t v[8]; };

BPG_HD fe fe_zero() { fe r; BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
BPG_HD fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }

// r = lo + 38 * hi for a 512-bit product t[0..15]
BPG_HD fe fe_fold512(const uint32_t t[16]) {
    fe r; uint64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) {
        c += (uint64_t)t[i] + (uint64_t)t[i + 8] * 38u;
        r.v[i] = (uint32_t)c; c >>= 32;
    }
    // c < 39 : fold again
    uint64_t d = (uint64_t)r.v[0] + c * 38u;
    r.v[0] = (uint32_t)d; d >>= 32;
    BPG_UNROLL for (int i = 1; i < 8; i++) { d += r.v[i]; r.v[i] = (uint32_t)d; d >>= 32; }
    r.v[0] += 38u * (uint32_t)d;      // a second wrap leaves a tiny value: no further carry
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 device paths (inline asm).  The portable C++ forms further down are what the host-side unit tests compile;
// tests/test_gpu_parity.py::test_device_field_ops checks these against big-integer arithmetic on the GPU itself.
//
// Multiply: product scanning with a 96-bit column accumulator (lo = 64-bit VGPR pair, hi = collected carries).
// v_mad_u64_u32 delivers its carry-out in an SGPR pair and one v_addc collects it: two VALU instructions per limb product
// and no zero-extension moves (the row-wise C++ form costs ~190 v_mov_b32 per multiplication because every 32-bit limb has
// to be widened into the MAD's 64-bit addend pair).  Only column 0 (no carry-in) is overflow-free; in every other column
// the first product initialises hi with its carry (a column's carry-in can reach ~2^35, and (2^32-1)^2 + 2^35 > 2^64).
// One asm block per product column: lo += sum x_j * y_j, hi = the carries out of bit 64.  The compiler pads every asm statement
// that writes an SGPR with an s_nop; a block per column instead of one per limb product leaves 15 of them in a multiplication
// instead of 64, which is worth +13 % where one wave per SIMD runs a dependent chain (Horner tails, compressions) and nothing at
// two waves or more (round-1 microbenchmark; the issue costs behind it: tools/diag/bench_issue.hip, profiles/r03_issue_costs.txt).
#define BPG_MAC(X, Y, HI) "v_mad_u64_u32 %0, %2, %" X ", %" Y ", %0\n\tv_addc_co_u32_e64 %1, %2, 0, " HI ", %2\n\t"
__device__ __forceinline__ void fe_col1(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0));
}
__device__ __forceinline__ void fe_col2(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1));
}
__device__ __forceinline__ void fe_col3(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2));
}
__device__ __forceinline__ void fe_col4(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") BPG_MAC("9", "10", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3));
}
__device__ __forceinline__ void fe_col5(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") BPG_MAC("9", "10", "%1") BPG_MAC("11", "12", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4));
}
__device__ __forceinline__ void fe_col6(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") BPG_MAC("9", "10", "%1") BPG_MAC("11", "12", "%1") BPG_MAC("13", "14", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5));
}
__device__ __forceinline__ void fe_col7(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") BPG_MAC("9", "10", "%1") BPG_MAC("11", "12", "%1") BPG_MAC("13", "14", "%1") BPG_MAC("15", "16", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6));
}
__device__ __forceinline__ void fe_col8(uint64_t &lo, uint32_t &hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3, uint32_t x4, uint32_t y4, uint32_t x5, uint32_t y5, uint32_t x6, uint32_t y6, uint32_t x7, uint32_t y7) {
    uint64_t c; asm(BPG_MAC("3", "4", "0") BPG_MAC("5", "6", "%1") BPG_MAC("7", "8", "%1") BPG_MAC("9", "10", "%1") BPG_MAC("11", "12", "%1") BPG_MAC("13", "14", "%1") BPG_MAC("15", "16", "%1") BPG_MAC("17", "18", "%1") : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7));
}
#undef BPG_MAC
// r = t[0..7] + 38 * t[8..15] (mod p, weakly reduced): 8 MADs and three v_addc chains (__builtin_addc lowers to v_addc_co_u32)
__device__ __forceinline__ fe fe_fold512_dev(const uint32_t t[16]) {
    uint64_t p[8];
    {   // the eight products t[8+i] * 38 in one asm block (one trailing s_nop instead of eight)
        uint64_t c;
        asm("v_mad_u64_u32 %0, %8, %9, 38, 0\n\tv_mad_u64_u32 %1, %8, %10, 38, 0\n\tv_mad_u64_u32 %2, %8, %11, 38, 0\n\tv_mad_u64_u32 %3, %8, %12, 38, 0\n\t"
            "v_mad_u64_u32 %4, %8, %13, 38, 0\n\tv_mad_u64_u32 %5, %8, %14, 38, 0\n\tv_mad_u64_u32 %6, %8, %15, 38, 0\n\tv_mad_u64_u32 %7, %8, %16, 38, 0"
            : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&s"(c)
            : "v"(t[8]), "v"(t[9]), "v"(t[10]), "v"(t[11]), "v"(t[12]), "v"(t[13]), "v"(t[14]), "v"(t[15]));
    }
    fe r; uint32_t c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; i++) { r.v[i] = __builtin_addc(t[i], (uint32_t)p[i], c, &co); c = co; }
    uint32_t top = c;
    c = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) { r.v[i] = __builtin_addc(r.v[i], (uint32_t)(p[i - 1] >> 32), c, &co); c = co; }
    top += (uint32_t)(p[7] >> 32) + c;                       // <= 1 + 37 + 1
    const uint32_t f = top * 38u;
    c = 0;
    r.v[0] = __builtin_addc(r.v[0], f, 0u, &co); c = co;
#pragma unroll
    for (int i = 1; i < 8; i++) { r.v[i] = __builtin_addc(r.v[i], 0u, c, &co); c = co; }
    r.v[0] += 38u * c;                                       // a second wrap leaves a tiny value: no further carry
    return r;
}
__device__ __forceinline__ fe fe_mul(const fe &a, const fe &b) {
    uint32_t t[16];
    uint64_t lo = 0; uint32_t hi;
    fe_col1(lo, hi, a.v[0], b.v[0]); t[0] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[0], b.v[1], a.v[1], b.v[0]); t[1] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[0], b.v[2], a.v[1], b.v[1], a.v[2], b.v[0]); t[2] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col4(lo, hi, a.v[0], b.v[3], a.v[1], b.v[2], a.v[2], b.v[1], a.v[3], b.v[0]); t[3] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col5(lo, hi, a.v[0], b.v[4], a.v[1], b.v[3], a.v[2], b.v[2], a.v[3], b.v[1], a.v[4], b.v[0]); t[4] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col6(lo, hi, a.v[0], b.v[5], a.v[1], b.v[4], a.v[2], b.v[3], a.v[3], b.v[2], a.v[4], b.v[1], a.v[5], b.v[0]); t[5] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col7(lo, hi, a.v[0], b.v[6], a.v[1], b.v[5], a.v[2], b.v[4], a.v[3], b.v[3], a.v[4], b.v[2], a.v[5], b.v[1], a.v[6], b.v[0]); t[6] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col8(lo, hi, a.v[0], b.v[7], a.v[1], b.v[6], a.v[2], b.v[5], a.v[3], b.v[4], a.v[4], b.v[3], a.v[5], b.v[2], a.v[6], b.v[1], a.v[7], b.v[0]); t[7] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col7(lo, hi, a.v[1], b.v[7], a.v[2], b.v[6], a.v[3], b.v[5], a.v[4], b.v[4], a.v[5], b.v[3], a.v[6], b.v[2], a.v[7], b.v[1]); t[8] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col6(lo, hi, a.v[2], b.v[7], a.v[3], b.v[6], a.v[4], b.v[5], a.v[5], b.v[4], a.v[6], b.v[3], a.v[7], b.v[2]); t[9] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col5(lo, hi, a.v[3], b.v[7], a.v[4], b.v[6], a.v[5], b.v[5], a.v[6], b.v[4], a.v[7], b.v[3]); t[10] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col4(lo, hi, a.v[4], b.v[7], a.v[5], b.v[6], a.v[6], b.v[5], a.v[7], b.v[4]); t[11] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[5], b.v[7], a.v[6], b.v[6], a.v[7], b.v[5]); t[12] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[6], b.v[7], a.v[7], b.v[6]); t[13] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col1(lo, hi, a.v[7], b.v[7]); t[14] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    t[15] = (uint32_t)lo;
    return fe_fold512_dev(t);
}
__device__ __forceinline__ fe fe_sq(const fe &a) {
    // a^2 = 2 * U + D with U = sum_{i<j} a_i a_j 2^(32(i+j)) (28 products, column scanning as in fe_mul),
    // doubled as one 512-bit shift (v_alignbit per limb), and D = sum a_i^2 2^(64 i) added with one carry chain.
    uint32_t u[16];
    u[0] = 0;
    uint64_t lo = 0; uint32_t hi;
    fe_col1(lo, hi, a.v[0], a.v[1]); u[1] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col1(lo, hi, a.v[0], a.v[2]); u[2] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[0], a.v[3], a.v[1], a.v[2]); u[3] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[0], a.v[4], a.v[1], a.v[3]); u[4] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[0], a.v[5], a.v[1], a.v[4], a.v[2], a.v[3]); u[5] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[0], a.v[6], a.v[1], a.v[5], a.v[2], a.v[4]); u[6] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col4(lo, hi, a.v[0], a.v[7], a.v[1], a.v[6], a.v[2], a.v[5], a.v[3], a.v[4]); u[7] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[1], a.v[7], a.v[2], a.v[6], a.v[3], a.v[5]); u[8] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col3(lo, hi, a.v[2], a.v[7], a.v[3], a.v[6], a.v[4], a.v[5]); u[9] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[3], a.v[7], a.v[4], a.v[6]); u[10] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col2(lo, hi, a.v[4], a.v[7], a.v[5], a.v[6]); u[11] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col1(lo, hi, a.v[5], a.v[7]); u[12] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    fe_col1(lo, hi, a.v[6], a.v[7]); u[13] = (uint32_t)lo; lo = (lo >> 32) | ((uint64_t)hi << 32);
    u[14] = (uint32_t)lo; u[15] = (uint32_t)(lo >> 32);
    uint64_t dd[8];
    {   // the eight squares a_i^2, both halves each, in one asm block
        uint64_t cs;
        asm("v_mad_u64_u32 %0, %8, %9, %9, 0\n\tv_mad_u64_u32 %1, %8, %10, %10, 0\n\tv_mad_u64_u32 %2, %8, %11, %11, 0\n\tv_mad_u64_u32 %3, %8, %12, %12, 0\n\t"
            "v_mad_u64_u32 %4, %8, %13, %13, 0\n\tv_mad_u64_u32 %5, %8, %14, %14, 0\n\tv_mad_u64_u32 %6, %8, %15, %15, 0\n\tv_mad_u64_u32 %7, %8, %16, %16, 0"
            : "=&v"(dd[0]), "=&v"(dd[1]), "=&v"(dd[2]), "=&v"(dd[3]), "=&v"(dd[4]), "=&v"(dd[5]), "=&v"(dd[6]), "=&v"(dd[7]), "=&s"(cs)
            : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]));
    }
    uint32_t t[16], c = 0, co;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = dd[i];
        const uint32_t e0 = (2 * i == 0) ? 0u : __builtin_amdgcn_alignbit(u[2 * i], u[2 * i - 1], 31);     // limb 2i of 2U
        const uint32_t e1 = __builtin_amdgcn_alignbit(u[2 * i + 1], u[2 * i], 31);                        // limb 2i+1 of 2U
        t[2 * i] = __builtin_addc(e0, (uint32_t)d, c, &co); c = co;
        t[2 * i + 1] = __builtin_addc(e1, (uint32_t)(d >> 32), c, &co); c = co;
    }
    return fe_fold512_dev(t);
}
__device__ __forceinline__ fe fe_add(const fe &a, const fe &b) {
    fe r; uint32_t t;
    asm("v_add_co_u32_e32 %0, vcc, %9, %17\n\t"
        "v_addc_co_u32_e32 %1, vcc, %10, %18, vcc\n\tv_addc_co_u32_e32 %2, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %12, %20, vcc\n\tv_addc_co_u32_e32 %4, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %14, %22, vcc\n\tv_addc_co_u32_e32 %6, vcc, %15, %23, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, 38, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\tv_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\tv_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, 38, vcc\n\t"
        "v_add_u32_e32 %0, %0, %8"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(t)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]) : "vcc");
    return r;
}
__device__ __forceinline__ fe fe_sub(const fe &a, const fe &b) {
    // a - b, then take 38 back off for each borrow (a - b + 2^256 = a - b + 38 mod p)
    fe r; uint32_t t;
    asm("v_sub_co_u32_e32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32_e32 %1, vcc, %10, %18, vcc\n\tv_subb_co_u32_e32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %12, %20, vcc\n\tv_subb_co_u32_e32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %14, %22, vcc\n\tv_subb_co_u32_e32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, 38, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, %8\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\tv_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\tv_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\tv_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
        "v_subbrev_co_u32_e32 %7, vcc, 0, %7, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, 38, vcc\n\t"
        "v_sub_u32_e32 %0, %0, %8"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(t)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]) : "vcc");
    return r;
}
#else
BPG_HD fe fe_mul(const fe &a, const fe &b) {
    uint32_t t[16];
    BPG_UNROLL for (int i = 0; i < 16; i++) t[i] = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        BPG_UNROLL for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)a.v[i] * b.v[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)x; carry = x >> 32;
        }
        t[i + 8] = (uint32_t)carry;
    }
    return fe_fold512(t);
}

BPG_HD fe fe_sq(const fe &a) {
    // off-diagonal products once, doubled, plus the diagonal
    uint32_t t[16];
    BPG_UNROLL for (int i = 0; i < 16; i++) t[i] = 0;
    BPG_UNROLL for (int i = 0; i < 7; i++) {
        uint64_t carry = 0;
        BPG_UNROLL for (int j = i + 1; j < 8; j++) {
            uint64_t x = (uint64_t)a.v[i] * a.v[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)x; carry = x >> 32;
        }
        t[i + 8] = (uint32_t)carry;
    }
    // double
    uint32_t top = 0;
    BPG_UNROLL for (int i = 1; i < 16; i++) { uint32_t n = t[i] >> 31; t[i] = (t[i] << 1) | top; top = n; }
    // add squares
    uint64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t s = (uint64_t)a.v[i] * a.v[i];
        c += (uint64_t)t[2 * i] + (uint32_t)s; t[2 * i] = (uint32_t)c; c >>= 32;
        c += (uint64_t)t[2 * i + 1] + (s >> 32); t[2 * i + 1] = (uint32_t)c; c >>= 32;
    }
    return fe_fold512(t);
}
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
BPG_HD fe fe_add(const fe &a, const fe &b) {
    fe r; uint64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    uint64_t d = (uint64_t)r.v[0] + 38u * c;
    r.v[0] = (uint32_t)d; d >>= 32;
    BPG_UNROLL for (int i = 1; i < 8; i++) { d += r.v[i]; r.v[i] = (uint32_t)d; d >>= 32; }
    r.v[0] += 38u * (uint32_t)d;
    return r;
}

BPG_HD fe fe_sub(const fe &a, const fe &b) {
    fe r; int64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (int64_t)a.v[i] - (int64_t)b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    // c is 0 or -1 : a - b + 2^256 = a - b + 38 (mod p), so take 38 back off
    int64_t d = (int64_t)r.v[0] + 38 * c;
    r.v[0] = (uint32_t)d; d >>= 32;
    BPG_UNROLL for (int i = 1; i < 8; i++) { d += r.v[i]; r.v[i] = (uint32_t)d; d >>= 32; }
    r.v[0] -= 38u * (uint32_t)(-d);   // second borrow leaves a value just below 2^256: no further borrow
    return r;
}
#endif

BPG_HD fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }

BPG_HD fe fe_mul_small(const fe &a, uint32_t k) {   // k < 2^26
    fe r; uint64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] * k; r.v[i] = (uint32_t)c; c >>= 32; }
    uint64_t d = (uint64_t)r.v[0] + c * 38u;     // c < 2^26 : 38c < 2^32
    r.v[0] = (uint32_t)d; d >>= 32;
    BPG_UNROLL for (int i = 1; i < 8; i++) { d += r.v[i]; r.v[i] = (uint32_t)d; d >>= 32; }
    r.v[0] += 38u * (uint32_t)d;
    return r;
}

// canonical representative in [0, p)
BPG_HD fe fe_freeze(const fe &a) {
    fe r = a;
    // fold bit 255 twice (value < 2^256 -> < 2^255 + 19 -> < 2^255 + small)
    BPG_UNROLL for (int k = 0; k < 2; k++) {
        uint32_t top = r.v[7] >> 31; r.v[7] &= 0x7fffffffu;
        uint64_t c = (uint64_t)r.v[0] + 19u * top; r.v[0] = (uint32_t)c; c >>= 32;
        BPG_UNROLL for (int i = 1; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    }
    // now r < 2^255; subtract p iff r >= p  <=>  r + 19 >= 2^255
    fe s; uint64_t c = (uint64_t)r.v[0] + 19u; s.v[0] = (uint32_t)c; c >>= 32;
    BPG_UNROLL for (int i = 1; i < 8; i++) { c += r.v[i]; s.v[i] = (uint32_t)c; c >>= 32; }
    uint32_t ge_p = s.v[7] >> 31;
    s.v[7] &= 0x7fffffffu;
    uint32_t m = 0u - ge_p;
    BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (r.v[i] & ~m) | (s.v[i] & m);
    return r;
}

BPG_HD fe fe_frombytes(const uint8_t *s) {   // ignores bit 255, like dalek FieldElement::from_bytes
    fe r;
    BPG_UNROLL for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
    r.v[7] &= 0x7fffffffu;
    return r;
}
BPG_HD fe fe_fromwords(const uint32_t *w) { fe r; BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = w[i]; r.v[7] &= 0x7fffffffu; return r; }

BPG_HD void fe_tobytes(uint8_t *s, const fe &a) {
    fe r = fe_freeze(a);
    BPG_UNROLL for (int i = 0; i < 8; i++) { s[4 * i] = (uint8_t)r.v[i]; s[4 * i + 1] = (uint8_t)(r.v[i] >> 8); s[4 * i + 2] = (uint8_t)(r.v[i] >> 16); s[4 * i + 3] = (uint8_t)(r.v[i] >> 24); }
}

BPG_HD uint32_t fe_isnegative(const fe &a) { return fe_freeze(a).v[0] & 1u; }
BPG_HD uint32_t fe_iszero(const fe &a) { fe r = fe_freeze(a); uint32_t o = 0; BPG_UNROLL for (int i = 0; i < 8; i++) o |= r.v[i]; return o == 0; }
BPG_HD uint32_t fe_eq(const fe &a, const fe &b) { return fe_iszero(fe_sub(a, b)); }
BPG_HD fe fe_select(const fe &a, const fe &b, uint32_t pick_b) {   // pick_b in {0,1}
    uint32_t m = 0u - pick_b; fe r;
    BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (a.v[i] & ~m) | (b.v[i] & m);
    return r;
}
BPG_HD fe fe_cneg(const fe &a, uint32_t neg) { return fe_select(a, fe_neg(a), neg); }
BPG_HD fe fe_abs(const fe &a) { return fe_cneg(a, fe_isnegative(a)); }

BPG_HD fe fe_sqn(fe a, int n) { for (int i = 0; i < n; i++) a = fe_sq(a); return a; }

// z^(2^250-1) and z^11
BPG_HD void fe_pow22501(fe &t19, fe &t3, const fe &z) {
    fe t0 = fe_sq(z);
    fe t1 = fe_sqn(t0, 2);
    fe t2 = fe_mul(z, t1);
    t3 = fe_mul(t0, t2);
    fe t4 = fe_sq(t3);
    fe t5 = fe_mul(t2, t4);
    fe t7 = fe_mul(fe_sqn(t5, 5), t5);
    fe t9 = fe_mul(fe_sqn(t7, 10), t7);
    fe t11 = fe_mul(fe_sqn(t9, 20), t9);
    fe t13 = fe_mul(fe_sqn(t11, 10), t7);
    fe t15 = fe_mul(fe_sqn(t13, 50), t13);
    fe t17 = fe_mul(fe_sqn(t15, 100), t15);
    t19 = fe_mul(fe_sqn(t17, 50), t13);
}
BPG_HD fe fe_invert(const fe &z) { fe t19, t3; fe_pow22501(t19, t3, z); return fe_mul(fe_sqn(t19, 5), t3); }
BPG_HD fe fe_pow22523(const fe &z) { fe t19, t3; fe_pow22501(t19, t3, z); return fe_mul(fe_sqn(t19, 2), z); }

#define BPG_FE(w0, w1, w2, w3, w4, w5, w6, w7) fe{{w0, w1, w2, w3, w4, w5, w6, w7}}
BPG_HD fe FE_D() { return BPG_FE(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }
BPG_HD fe FE_D2() { return BPG_FE(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
BPG_HD fe FE_D4() { return BPG_FE(0x4d65e2b2u, 0xd7ad3728u, 0x050762adu, 0x01c02935u, 0xdde7a260u, 0x331d01e5u, 0xadbff9ceu, 0x480db3b8u); }      // 4d
BPG_HD fe FE_INV_D() { return BPG_FE(0xcdc9f843u, 0x25e0f276u, 0x4279542eu, 0x0b5dd698u, 0xcdb9cf66u, 0x2b162114u, 0x14d5ce43u, 0x40907ed2u); }      // 1 / d
BPG_HD fe FE_INV2() { return BPG_FE(0xfffffff7u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu); }    // 1/2 = (p + 1) / 2
BPG_HD fe FE_SQRTM1() { return BPG_FE(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); }
BPG_HD fe FE_SQRT_AD_MINUS_ONE() { return BPG_FE(0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu); }
BPG_HD fe FE_INVSQRT_A_MINUS_D() { return BPG_FE(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); }
BPG_HD fe FE_ONE_MINUS_D_SQ() { return BPG_FE(0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u); }
BPG_HD fe FE_D_MINUS_ONE_SQ() { return BPG_FE(0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au); }

}  // namespace bpg
