// Diagnostic prototype (not part of the product): GF(2^255-19) in radix 2^25.5 - ten limbs of 26/25 bits, the layout of ref10 - on
// gfx950.  No carry-outs: a column of ten 32x32 products stays below 2^64, so every limb product is ONE v_mad_u64_u32 and the
// v_addc that the saturated 8 x 32 layout pays per product disappears.  Measured against fe_mul / fe_sq by bench_fe.hip.
#pragma once
#include <stdint.h>
namespace bpg10 {
struct fe10 { uint32_t v[10]; };
#define M26 0x3ffffffu
#define M25 0x1ffffffu
__host__ __device__ inline fe10 fe10_from8(const uint32_t w[8]) {            // 255-bit value in 8 words -> 10 limbs
    fe10 r; const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
    for (int i = 0; i < 10; i++) {
        const int o = off[i], wi = o >> 5, sh = o & 31;
        uint64_t two = (uint64_t)w[wi] | ((uint64_t)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
        r.v[i] = (uint32_t)(two >> sh) & ((i & 1) ? M25 : M26);
    }
    return r;
}
__host__ __device__ inline void fe10_to8(uint32_t w[8], const fe10 &a) {     // limbs must be carried (26/25 bits); value < 2^255
    const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
    uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 10; i++) { const int o = off[i], wi = o >> 5, sh = o & 31; uint64_t v = (uint64_t)a.v[i] << sh; acc[wi] += v & 0xffffffffu; acc[wi + 1] += v >> 32; }
    uint64_t c = 0;
    for (int i = 0; i < 8; i++) { c += acc[i]; w[i] = (uint32_t)c; c >>= 32; }
}
// h = f * g.  Inputs may be lazily reduced (limbs < 2^27); output limbs < 2^26 / 2^25 (+ a small excess in limb 1).
__device__ __forceinline__ fe10 fe10_mul(const fe10 &f, const fe10 &g) {
    const uint32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    const uint32_t g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4], g5 = g.v[5], g6 = g.v[6], g7 = g.v[7], g8 = g.v[8], g9 = g.v[9];
    const uint32_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4, g5_19 = 19 * g5, g6_19 = 19 * g6, g7_19 = 19 * g7, g8_19 = 19 * g8, g9_19 = 19 * g9;
    const uint32_t f1_2 = 2 * f1, f3_2 = 2 * f3, f5_2 = 2 * f5, f7_2 = 2 * f7, f9_2 = 2 * f9;
#define P(a, b) ((uint64_t)(a) * (b))
    fe10 r; uint64_t h, c;
    h = P(f0, g0) + P(f1_2, g9_19) + P(f2, g8_19) + P(f3_2, g7_19) + P(f4, g6_19) + P(f5_2, g5_19) + P(f6, g4_19) + P(f7_2, g3_19) + P(f8, g2_19) + P(f9_2, g1_19);
    r.v[0] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0, g1) + P(f1, g0) + P(f2, g9_19) + P(f3, g8_19) + P(f4, g7_19) + P(f5, g6_19) + P(f6, g5_19) + P(f7, g4_19) + P(f8, g3_19) + P(f9, g2_19);
    r.v[1] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0, g2) + P(f1_2, g1) + P(f2, g0) + P(f3_2, g9_19) + P(f4, g8_19) + P(f5_2, g7_19) + P(f6, g6_19) + P(f7_2, g5_19) + P(f8, g4_19) + P(f9_2, g3_19);
    r.v[2] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0, g3) + P(f1, g2) + P(f2, g1) + P(f3, g0) + P(f4, g9_19) + P(f5, g8_19) + P(f6, g7_19) + P(f7, g6_19) + P(f8, g5_19) + P(f9, g4_19);
    r.v[3] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0, g4) + P(f1_2, g3) + P(f2, g2) + P(f3_2, g1) + P(f4, g0) + P(f5_2, g9_19) + P(f6, g8_19) + P(f7_2, g7_19) + P(f8, g6_19) + P(f9_2, g5_19);
    r.v[4] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0, g5) + P(f1, g4) + P(f2, g3) + P(f3, g2) + P(f4, g1) + P(f5, g0) + P(f6, g9_19) + P(f7, g8_19) + P(f8, g7_19) + P(f9, g6_19);
    r.v[5] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0, g6) + P(f1_2, g5) + P(f2, g4) + P(f3_2, g3) + P(f4, g2) + P(f5_2, g1) + P(f6, g0) + P(f7_2, g9_19) + P(f8, g8_19) + P(f9_2, g7_19);
    r.v[6] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0, g7) + P(f1, g6) + P(f2, g5) + P(f3, g4) + P(f4, g3) + P(f5, g2) + P(f6, g1) + P(f7, g0) + P(f8, g9_19) + P(f9, g8_19);
    r.v[7] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0, g8) + P(f1_2, g7) + P(f2, g6) + P(f3_2, g5) + P(f4, g4) + P(f5_2, g3) + P(f6, g2) + P(f7_2, g1) + P(f8, g0) + P(f9_2, g9_19);
    r.v[8] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0, g9) + P(f1, g8) + P(f2, g7) + P(f3, g6) + P(f4, g5) + P(f5, g4) + P(f6, g3) + P(f7, g2) + P(f8, g1) + P(f9, g0);
    r.v[9] = (uint32_t)h & M25; c = h >> 25;
    h = (uint64_t)r.v[0] + c * 19;                                   // c < 2^39: 64-bit product by a constant
    r.v[0] = (uint32_t)h & M26; r.v[1] += (uint32_t)(h >> 26);
#undef P
    return r;
}
__device__ __forceinline__ fe10 fe10_sq(const fe10 &f) {
    const uint32_t f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    const uint32_t f0_2 = 2 * f0, f1_2 = 2 * f1, f2_2 = 2 * f2, f3_2 = 2 * f3, f4_2 = 2 * f4, f5_2 = 2 * f5, f6_2 = 2 * f6, f7_2 = 2 * f7;
    const uint32_t f5_38 = 38 * f5, f6_19 = 19 * f6, f7_38 = 38 * f7, f8_19 = 19 * f8, f9_38 = 38 * f9;
#define P(a, b) ((uint64_t)(a) * (b))
    fe10 r; uint64_t h, c;
    h = P(f0, f0) + P(f1_2, f9_38) + P(f2_2, f8_19) + P(f3_2, f7_38) + P(f4_2, f6_19) + P(f5, f5_38);
    r.v[0] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0_2, f1) + P(f2, f9_38) + P(f3_2, f8_19) + P(f4, f7_38) + P(f5_2, f6_19);
    r.v[1] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0_2, f2) + P(f1_2, f1) + P(f3_2, f9_38) + P(f4_2, f8_19) + P(f5_2, f7_38) + P(f6, f6_19);
    r.v[2] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0_2, f3) + P(f1_2, f2) + P(f4, f9_38) + P(f5_2, f8_19) + P(f6, f7_38);
    r.v[3] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0_2, f4) + P(f1_2, f3_2) + P(f2, f2) + P(f5_2, f9_38) + P(f6_2, f8_19) + P(f7, f7_38);
    r.v[4] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0_2, f5) + P(f1_2, f4) + P(f2_2, f3) + P(f6, f9_38) + P(f7_2, f8_19);
    r.v[5] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0_2, f6) + P(f1_2, f5_2) + P(f2_2, f4) + P(f3_2, f3) + P(f7_2, f9_38) + P(f8, f8_19);
    r.v[6] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0_2, f7) + P(f1_2, f6) + P(f2_2, f5) + P(f3_2, f4) + P(f8, f9_38);
    r.v[7] = (uint32_t)h & M25; c = h >> 25;
    h = c + P(f0_2, f8) + P(f1_2, f7_2) + P(f2_2, f6) + P(f3_2, f5_2) + P(f4, f4) + P(f9, f9_38);
    r.v[8] = (uint32_t)h & M26; c = h >> 26;
    h = c + P(f0_2, f9) + P(f1_2, f8) + P(f2_2, f7) + P(f3_2, f6) + P(f4_2, f5);
    r.v[9] = (uint32_t)h & M25; c = h >> 25;
    h = (uint64_t)r.v[0] + c * 19;
    r.v[0] = (uint32_t)h & M26; r.v[1] += (uint32_t)(h >> 26);
#undef P
    return r;
}
// lazy add (no carries) and sub (a + 2p - b: stays non-negative for carried b, limbs < 2^27 afterwards)
__device__ __forceinline__ fe10 fe10_add(const fe10 &a, const fe10 &b) { fe10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r; }
__device__ __forceinline__ fe10 fe10_sub(const fe10 &a, const fe10 &b) { fe10 r;
    r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];
    return r; }
// one carry pass: limbs back to 26/25 bits (+ a small excess in limb 0); needed before a lazy sum or difference feeds a multiplication twice
__device__ __forceinline__ fe10 fe10_carry(const fe10 &a) { fe10 r = a; uint32_t c;
#pragma unroll
    for (int i = 0; i < 9; i++) { c = r.v[i] >> ((i & 1) ? 25 : 26); r.v[i] &= (i & 1) ? M25 : M26; r.v[i + 1] += c; }
    c = r.v[9] >> 25; r.v[9] &= M25; r.v[0] += 19 * c;
    return r; }
}  // namespace bpg10
