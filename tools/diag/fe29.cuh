// Diagnostic prototype: GF(2^255-19) in radix 2^29 - nine unsigned limbs, 261 bits, 2^261 = 1216 (mod p) - on gfx950.
// A column of nine 32x32 products of operands below 2^30.4 stays below 2^64, so every limb product is ONE v_mad_u64_u32 with no carry-out:
// 92 MADs + ~60 plain 32-bit operations per multiplication against 72 MADs + 94 carry-chained v_addc + 25 moves for the saturated 8 x 32 layout.
#pragma once
#include <stdint.h>
namespace bpg29 {
struct fe29 { uint32_t v[9]; };
#define M29 0x1fffffffu
#define F29_FOLD 1216u                      // 2^261 mod p = 19 * 2^6

// "tight": every limb <= 2^29 + 2^11.  Multiplication needs 9 * max(a_i) * max(b_j) < 2^64 - 2^41.
__device__ __forceinline__ uint64_t mac(uint64_t h, uint32_t a, uint32_t b) { return h + (uint64_t)a * b; }
__device__ __forceinline__ uint64_t shr29(uint64_t h) {      // two 32-bit operations instead of a 64-bit shift
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    return ((uint64_t)(hi >> 29) << 32) | __builtin_amdgcn_alignbit(hi, lo, 29);
}
__device__ __forceinline__ fe29 fe29_mul(const fe29 &f, const fe29 &g) {
    const uint32_t *a = f.v, *b = g.v;
    // high columns 9..16 first (carry-in 0): u[k] = limb of weight 2^(29(k+9)), u[8] = the carry out of column 16 (< 2^32)
    uint32_t u[9]; uint64_t h = 0;
#pragma unroll
    for (int k = 9; k <= 16; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) h = mac(h, a[i], b[k - i]);
        u[k - 9] = (uint32_t)h & M29; h = shr29(h);
    }
    u[8] = (uint32_t)h;
    // low columns 0..8 with the high half folded in: + 1216 * u[k]
    fe29 r; h = 0;
#pragma unroll
    for (int k = 0; k <= 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) h = mac(h, a[i], b[k - i]);
        h = mac(h, u[k], F29_FOLD);
        r.v[k] = (uint32_t)h & M29; h = shr29(h);
    }
    // carry out of column 8 (weight 2^261, < 2^37): fold it as two 29-bit pieces, then ripple twice
    const uint32_t c0 = (uint32_t)h & M29, c1 = (uint32_t)(h >> 29);
    uint64_t t = mac((uint64_t)r.v[0], c0, F29_FOLD);
    r.v[0] = (uint32_t)t & M29; t = shr29(t);
    t = mac(t + r.v[1], c1, F29_FOLD);
    r.v[1] = (uint32_t)t & M29;
    r.v[2] += (uint32_t)(t >> 29);
    return r;
}
__device__ __forceinline__ fe29 fe29_sq(const fe29 &f) {
    const uint32_t *a = f.v;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = 2 * a[i];
    uint32_t u[9]; uint64_t h = 0;
#pragma unroll
    for (int k = 9; k <= 16; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) h = mac(h, d[i], a[k - i]);
        if ((k & 1) == 0) h = mac(h, a[k / 2], a[k / 2]);
        u[k - 9] = (uint32_t)h & M29; h = shr29(h);
    }
    u[8] = (uint32_t)h;
    fe29 r; h = 0;
#pragma unroll
    for (int k = 0; k <= 8; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) h = mac(h, d[i], a[k - i]);
        if ((k & 1) == 0) h = mac(h, a[k / 2], a[k / 2]);
        h = mac(h, u[k], F29_FOLD);
        r.v[k] = (uint32_t)h & M29; h = shr29(h);
    }
    const uint32_t c0 = (uint32_t)h & M29, c1 = (uint32_t)(h >> 29);
    uint64_t t = mac((uint64_t)r.v[0], c0, F29_FOLD);
    r.v[0] = (uint32_t)t & M29; t = shr29(t);
    t = mac(t + r.v[1], c1, F29_FOLD);
    r.v[1] = (uint32_t)t & M29;
    r.v[2] += (uint32_t)(t >> 29);
    return r;
}
__device__ __forceinline__ fe29 fe29_add(const fe29 &a, const fe29 &b) { fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r; }
// a - b + 128 p (limbs 2^30 - 2 resp. 2^30 - 2432): needs every limb of b <= 2^30 - 2432; limbs of the result < limbs of a + 2^30
__device__ __forceinline__ fe29 fe29_sub(const fe29 &a, const fe29 &b) { fe29 r;
    r.v[0] = a.v[0] + (0x40000000u - 2432u) - b.v[0];
#pragma unroll
    for (int i = 1; i < 9; i++) r.v[i] = a.v[i] + (0x40000000u - 2u) - b.v[i];
    return r; }
// one parallel carry step (no ripple): limbs below 2^32 -> limbs <= 2^29 + 8 (limb 0: + 1216 * 7)
__device__ __forceinline__ fe29 fe29_carry(const fe29 &a) { fe29 r;
    r.v[0] = (a.v[0] & M29) + F29_FOLD * (a.v[8] >> 29);
#pragma unroll
    for (int i = 1; i < 9; i++) r.v[i] = (a.v[i] & M29) + (a.v[i - 1] >> 29);
    return r; }
// 8 x 32 saturated words (any value below 2^256) <-> nine 29-bit limbs
__device__ __forceinline__ fe29 fe29_from8(const uint32_t w[8]) { fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int o = 29 * i, wi = o >> 5, sh = o & 31;
        const uint32_t lo = w[wi], hi = wi + 1 < 8 ? w[wi + 1] : 0u;
        r.v[i] = (sh == 0 ? lo : __builtin_amdgcn_alignbit(hi, lo, sh)) & M29;
    }
    return r; }
// tight limbs -> 8 words holding the same residue (value below 2^256 after folding the bits from 256 up)
__device__ __forceinline__ void fe29_to8(uint32_t w[8], const fe29 &a) {
    fe29 c = a; uint32_t k;
#pragma unroll
    for (int i = 0; i < 8; i++) { k = c.v[i] >> 29; c.v[i] &= M29; c.v[i + 1] += k; }       // full ripple: limbs 0..7 < 2^29, limb 8 < 2^29 + small
    // bits 256.. of the value sit in limb 8 from bit 24 up (8 * 29 = 232): fold them with 2^256 = 38
    const uint32_t top = c.v[8] >> 24; c.v[8] &= 0xffffffu;
    uint64_t acc = (uint64_t)c.v[0] + 38u * (uint64_t)top;
    c.v[0] = (uint32_t)acc & M29; k = (uint32_t)(acc >> 29);
#pragma unroll
    for (int i = 1; i < 9; i++) { c.v[i] += k; k = c.v[i] >> 29; if (i < 8) c.v[i] &= M29; }
    uint64_t bits = 0; int have = 0, wi = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        bits |= (uint64_t)c.v[i] << have; have += 29;
        if (have >= 32 && wi < 8) { w[wi++] = (uint32_t)bits; bits >>= 32; have -= 32; }
    }
    if (wi < 8) w[wi] = (uint32_t)bits;
}
}  // namespace bpg29
