// Host-side GF(2^255-19) and Ristretto255 point arithmetic in radix 2^51 (five 64-bit limbs, 128-bit products): the SERIAL epilogues of
// the prove path run here, on the latency-optimised core, instead of on one GPU wave:
//   * the Horner recombination of an MSM's window sums, sum_j 2^off(j) * S_j: ~254 dependent doublings of ONE point (0.35 ms on a wave
//     that issues one instruction every 4-5 cycles, ~35 us on a 5 GHz core), after the GPU has reduced millions of additions to W points;
//   * RFC 9496 encoding (R1CSProof::to_bytes of a point = one inverse square root, a chain of 252 dependent squarings).
// Everything data-parallel stays on the GPU (hip/*.cuh); device points arrive as four 8 x u32 "weakly reduced" coordinates (hip/fe.cuh).
// Same group elements, hence the same bytes: RFC 9496 encodings are canonical.  Replaces the corresponding curve25519-dalek calls made
// inside bulletproofs' Prover::prove (reference call site src/bin/prover.rs:93,97; crates not vendored).
#pragma once
#include <cstdint>
#include <cstring>

namespace bpg {
namespace h51 {

typedef unsigned __int128 u128;
struct fe51 { uint64_t v[5]; };
static const uint64_t M51 = (1ull << 51) - 1;

inline fe51 fe_zero() { return fe51{{0, 0, 0, 0, 0}}; }
inline fe51 fe_one() { return fe51{{1, 0, 0, 0, 0}}; }
// 256-bit little-endian integer below 2^256 (device form) -> limbs; bit 255 stays in the top limb (weight 2^255 = 2^204 * 2^51)
inline fe51 fe_from_words(const uint32_t w[8]) {
    const uint64_t a0 = w[0] | ((uint64_t)w[1] << 32), a1 = w[2] | ((uint64_t)w[3] << 32), a2 = w[4] | ((uint64_t)w[5] << 32), a3 = w[6] | ((uint64_t)w[7] << 32);
    fe51 r;
    r.v[0] = a0 & M51;
    r.v[1] = ((a0 >> 51) | (a1 << 13)) & M51;
    r.v[2] = ((a1 >> 38) | (a2 << 26)) & M51;
    r.v[3] = ((a2 >> 25) | (a3 << 39)) & M51;
    r.v[4] = a3 >> 12;                                     // 52 bits
    return r;
}
inline fe51 fe_carry(fe51 a) {                              // limbs below 2^63 -> below 2^51 + 2^13
    uint64_t c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    c = a.v[4] >> 51; a.v[4] &= M51; a.v[0] += 19 * c;
    return a;
}
inline fe51 fe_add(const fe51 &a, const fe51 &b) { fe51 r; for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i]; return fe_carry(r); }
inline fe51 fe_sub(const fe51 &a, const fe51 &b) {          // a + 8p - b, inputs below 2^53
    fe51 r;
    r.v[0] = a.v[0] + 0x3fffffffffff68ull - b.v[0];         // 8 * (2^51 - 19)
    for (int i = 1; i < 5; i++) r.v[i] = a.v[i] + 0x3ffffffffffff8ull - b.v[i];   // 8 * (2^51 - 1)
    return fe_carry(r);
}
inline fe51 fe_neg(const fe51 &a) { return fe_sub(fe_zero(), a); }
inline fe51 fe_mul(const fe51 &a, const fe51 &b) {          // inputs below 2^54
    const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    const uint64_t b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3], b4 = b.v[4];
    const uint64_t b1_19 = 19 * b1, b2_19 = 19 * b2, b3_19 = 19 * b3, b4_19 = 19 * b4;
    u128 r0 = (u128)a0 * b0 + (u128)a1 * b4_19 + (u128)a2 * b3_19 + (u128)a3 * b2_19 + (u128)a4 * b1_19;
    u128 r1 = (u128)a0 * b1 + (u128)a1 * b0 + (u128)a2 * b4_19 + (u128)a3 * b3_19 + (u128)a4 * b2_19;
    u128 r2 = (u128)a0 * b2 + (u128)a1 * b1 + (u128)a2 * b0 + (u128)a3 * b4_19 + (u128)a4 * b3_19;
    u128 r3 = (u128)a0 * b3 + (u128)a1 * b2 + (u128)a2 * b1 + (u128)a3 * b0 + (u128)a4 * b4_19;
    u128 r4 = (u128)a0 * b4 + (u128)a1 * b3 + (u128)a2 * b2 + (u128)a3 * b1 + (u128)a4 * b0;
    fe51 r;
    r1 += (uint64_t)(r0 >> 51); r.v[0] = (uint64_t)r0 & M51;
    r2 += (uint64_t)(r1 >> 51); r.v[1] = (uint64_t)r1 & M51;
    r3 += (uint64_t)(r2 >> 51); r.v[2] = (uint64_t)r2 & M51;
    r4 += (uint64_t)(r3 >> 51); r.v[3] = (uint64_t)r3 & M51;
    const uint64_t c = (uint64_t)(r4 >> 51); r.v[4] = (uint64_t)r4 & M51;
    r.v[0] += 19 * c;
    const uint64_t c2 = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c2;
    return r;
}
inline fe51 fe_sq(const fe51 &a) {
    const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    const uint64_t d0 = 2 * a0, d1 = 2 * a1, d2 = 2 * a2, a3_19 = 19 * a3, a4_19 = 19 * a4;
    u128 r0 = (u128)a0 * a0 + (u128)d1 * a4_19 + (u128)d2 * a3_19;
    u128 r1 = (u128)d0 * a1 + (u128)d2 * a4_19 + (u128)a3 * a3_19;
    u128 r2 = (u128)d0 * a2 + (u128)a1 * a1 + (u128)(2 * a3) * a4_19;
    u128 r3 = (u128)d0 * a3 + (u128)d1 * a2 + (u128)a4 * a4_19;
    u128 r4 = (u128)d0 * a4 + (u128)d1 * a3 + (u128)a2 * a2;
    fe51 r;
    r1 += (uint64_t)(r0 >> 51); r.v[0] = (uint64_t)r0 & M51;
    r2 += (uint64_t)(r1 >> 51); r.v[1] = (uint64_t)r1 & M51;
    r3 += (uint64_t)(r2 >> 51); r.v[2] = (uint64_t)r2 & M51;
    r4 += (uint64_t)(r3 >> 51); r.v[3] = (uint64_t)r3 & M51;
    const uint64_t c = (uint64_t)(r4 >> 51); r.v[4] = (uint64_t)r4 & M51;
    r.v[0] += 19 * c;
    const uint64_t c2 = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c2;
    return r;
}
inline fe51 fe_sqn(fe51 a, int n) { for (int i = 0; i < n; i++) a = fe_sq(a); return a; }
// canonical 32-byte encoding
inline void fe_tobytes(uint8_t s[32], const fe51 &a) {
    fe51 t = fe_carry(fe_carry(a));
    // t < 2^255 + small; add 19 and see whether bit 255 appears: t >= p
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
    t.v[4] &= M51;
    const uint64_t w0 = t.v[0] | (t.v[1] << 51), w1 = (t.v[1] >> 13) | (t.v[2] << 38), w2 = (t.v[2] >> 26) | (t.v[3] << 25), w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    std::memcpy(s, &w0, 8); std::memcpy(s + 8, &w1, 8); std::memcpy(s + 16, &w2, 8); std::memcpy(s + 24, &w3, 8);
}
inline bool fe_isnegative(const fe51 &a) { uint8_t s[32]; fe_tobytes(s, a); return s[0] & 1; }
inline bool fe_eq(const fe51 &a, const fe51 &b) { uint8_t s[32], t[32]; fe_tobytes(s, a); fe_tobytes(t, b); return std::memcmp(s, t, 32) == 0; }
inline fe51 fe_cneg(const fe51 &a, bool neg) { return neg ? fe_neg(a) : a; }
inline fe51 fe_abs(const fe51 &a) { return fe_cneg(a, fe_isnegative(a)); }
// z^(2^252 - 3)
inline fe51 fe_pow22523(const fe51 &z) {
    fe51 t0 = fe_sq(z);
    fe51 t1 = fe_sqn(t0, 2);
    fe51 t2 = fe_mul(z, t1);
    fe51 t3 = fe_mul(t0, t2);
    fe51 t4 = fe_sq(t3);
    fe51 t5 = fe_mul(t2, t4);
    fe51 t7 = fe_mul(fe_sqn(t5, 5), t5);
    fe51 t9 = fe_mul(fe_sqn(t7, 10), t7);
    fe51 t11 = fe_mul(fe_sqn(t9, 20), t9);
    fe51 t13 = fe_mul(fe_sqn(t11, 10), t7);
    fe51 t15 = fe_mul(fe_sqn(t13, 50), t13);
    fe51 t17 = fe_mul(fe_sqn(t15, 100), t15);
    fe51 t19 = fe_mul(fe_sqn(t17, 50), t13);
    return fe_mul(fe_sqn(t19, 2), z);
}
inline fe51 fe_const(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7) {
    const uint32_t w[8] = {w0, w1, w2, w3, w4, w5, w6, w7}; return fe_from_words(w);
}
// the constants of hip/fe.cuh (same words)
inline const fe51 &FE_D2() { static const fe51 c = fe_const(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); return c; }
inline const fe51 &FE_SQRTM1() { static const fe51 c = fe_const(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); return c; }
inline const fe51 &FE_INVSQRT_A_MINUS_D() { static const fe51 c = fe_const(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); return c; }

// r = sqrt(u/v) or sqrt(i*u/v) (RFC 9496 SQRT_RATIO_M1), as hip/ge.cuh fe_sqrt_ratio_i
inline bool fe_sqrt_ratio_i(fe51 &r, const fe51 &u, const fe51 &v) {
    const fe51 v3 = fe_mul(fe_sq(v), v);
    const fe51 v7 = fe_mul(fe_sq(v3), v);
    fe51 rr = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    const fe51 check = fe_mul(v, fe_sq(rr));
    const fe51 negu = fe_neg(u);
    const bool correct = fe_eq(check, u), flipped = fe_eq(check, negu), flipped_i = fe_eq(check, fe_mul(negu, FE_SQRTM1()));
    if (flipped || flipped_i) rr = fe_mul(rr, FE_SQRTM1());
    r = fe_abs(rr);
    return correct || flipped;
}

struct pt { fe51 X, Y, Z, T; };                              // extended twisted Edwards, a = -1
inline pt pt_identity() { return pt{fe_zero(), fe_one(), fe_one(), fe_zero()}; }
inline pt pt_from_device(const uint32_t w[32]) { return pt{fe_from_words(w), fe_from_words(w + 8), fe_from_words(w + 16), fe_from_words(w + 24)}; }
inline pt pt_dbl(const pt &p) {                              // 4S + 4M, as hip/ge.cuh ge_dbl
    const fe51 XX = fe_sq(p.X), YY = fe_sq(p.Y); fe51 ZZ2 = fe_sq(p.Z); ZZ2 = fe_add(ZZ2, ZZ2);
    const fe51 S = fe_sq(fe_add(p.X, p.Y));
    const fe51 YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX);
    const fe51 cX = fe_sub(S, YpX), cT = fe_sub(ZZ2, YmX);
    return pt{fe_mul(cX, cT), fe_mul(YpX, YmX), fe_mul(YmX, cT), fe_mul(cX, YpX)};
}
inline pt pt_add(const pt &p, const pt &q) {                 // 9M, as hip/ge.cuh ge_add
    const fe51 A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const fe51 B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const fe51 C = fe_mul(fe_mul(p.T, q.T), FE_D2());
    fe51 D = fe_mul(p.Z, q.Z); D = fe_add(D, D);
    const fe51 E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    return pt{fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
}
// RFC 9496 4.3.2 Encode, as hip/ge.cuh ge_compress
inline void pt_compress(uint8_t out[32], const pt &p) {
    const fe51 u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    const fe51 u2 = fe_mul(p.X, p.Y);
    fe51 inv; (void)fe_sqrt_ratio_i(inv, fe_one(), fe_mul(u1, fe_sq(u2)));
    const fe51 i1 = fe_mul(inv, u1), i2 = fe_mul(inv, u2);
    const fe51 zinv = fe_mul(fe_mul(i1, i2), p.T);
    const fe51 iX = fe_mul(p.X, FE_SQRTM1()), iY = fe_mul(p.Y, FE_SQRTM1());
    const fe51 ench = fe_mul(i1, FE_INVSQRT_A_MINUS_D());
    const bool rotate = fe_isnegative(fe_mul(p.T, zinv));
    const fe51 X = rotate ? iY : p.X; fe51 Y = rotate ? iX : p.Y; const fe51 den = rotate ? ench : i2;
    Y = fe_cneg(Y, fe_isnegative(fe_mul(X, zinv)));
    fe_tobytes(out, fe_abs(fe_mul(den, fe_sub(p.Z, Y))));
}
// sum_j 2^off(j) * S_j with off(j) = j * 254 / W (hip/k_msm.cuh msm_off): the recombination of an MSM's W window sums
inline pt pt_horner(const uint32_t *wsum /* W x 32 words, device ge_ext */, uint32_t W) {
    pt acc = pt_from_device(wsum + (size_t)(W - 1) * 32);
    for (int32_t win = (int32_t)W - 2; win >= 0; win--) {
        const uint32_t shift = ((uint32_t)(win + 1) * 254u) / W - ((uint32_t)win * 254u) / W;
        for (uint32_t k = 0; k < shift; k++) acc = pt_dbl(acc);
        acc = pt_add(acc, pt_from_device(wsum + (size_t)win * 32));
    }
    return acc;
}

}  // namespace h51
}  // namespace bpg
