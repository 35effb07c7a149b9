// Scalars mod l = 2^252 + 27742317777372353535851937790883648493 for gfx950.
// Device-resident scalar vectors are kept in MONTGOMERY form (x * 2^256 mod l, canonical < l) as 8 x u32, so that
// the Hadamard / inner-product / fold kernels cost one Montgomery product per multiply.  Conversions happen at the
// byte boundary (sc_from_bytes / sc_to_words).  Replaces curve25519-dalek Scalar arithmetic (not vendored;
// reference Cargo.toml:8; semantics per reference src/conversions.rs:18 `Scalar::from_bits`).
#pragma once
#include "fe.cuh"

namespace bpg {

struct alignas(16) scm { uint32_t v[8]; };    // Montgomery form

#define BPG_SCM(w0, w1, w2, w3, w4, w5, w6, w7) scm{{w0, w1, w2, w3, w4, w5, w6, w7}}
BPG_HD scm SC_L() { return BPG_SCM(0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u); }
BPG_HD scm SC_R1() { return BPG_SCM(0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu); }   // 1 in Montgomery form
BPG_HD scm SC_RR() { return BPG_SCM(0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu); }
BPG_HD scm SC_RRR() { return BPG_SCM(0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u); }
#define BPG_SC_NINV 0x12547e1bu

BPG_HD scm sc_zero() { scm r; BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
BPG_HD scm sc_plain_one() { scm r = sc_zero(); r.v[0] = 1; return r; }

// t (9 limbs, value < 2l) -> t mod l
BPG_HD scm sc_cond_sub(const uint32_t t[9]) {
    const scm L = SC_L();
    uint32_t s[8]; int64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (int64_t)t[i] - (int64_t)L.v[i]; s[i] = (uint32_t)c; c >>= 32; }
    c += t[8];                                  // c == -1  <=>  t < l
    uint32_t keep = (uint32_t)(c >> 63) & 1u;   // 1: keep t
    uint32_t m = 0u - keep; scm r;
    BPG_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (t[i] & m) | (s[i] & ~m);
    return r;
}

// Montgomery product a*b/2^256 mod l; needs a*b < 2^256 * l (true when one operand < l); output canonical
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 device path (round 5).  The portable row-wise form below compiles to ~520 VALU instructions (every 32-bit limb widened into the MAD's 64-bit
// addend pair); this one is product scanning twice over, on the column blocks of fe.cuh: first the 512-bit product t = a * b (64 multiply-adds), then
// the Montgomery reduction column by column - u_k = t_k + sum_{i+j=k} m_i l_j, m_k = u_k * (-1/l) mod 2^32 - where l has only five non-zero limbs
// (l_4 = l_5 = l_6 = 0), so 40 more multiply-adds.  One VALU pair per limb product (v_mad_u64_u32 + the v_addc that collects its carry): ~330
// instructions.  tests/test_gpu_parity.py::test_device_field_ops (op 6) checks it against Python integers on the GPU.
__device__ __forceinline__ void sc_mac0(uint64_t &lo, uint32_t &hi, uint32_t x) {                    // lo += x, hi = carry
    uint64_t c; asm("v_mad_u64_u32 %0, %2, %3, 1, %0\n\tv_addc_co_u32_e64 %1, %2, 0, 0, %2" : "+v"(lo), "=&v"(hi), "=&s"(c) : "v"(x));
}
__device__ __forceinline__ void sc_mac(uint64_t &lo, uint32_t &hi, uint32_t x, uint32_t y) {         // (hi, lo) += x * y
    uint64_t c; asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2" : "+v"(lo), "+v"(hi), "=&s"(c) : "v"(x), "v"(y));
}
__device__ __forceinline__ scm sc_mont_mul(const scm &a, const scm &b) {
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
    // Montgomery reduction, product scanning: column k of t + m * l, with l_j != 0 only for j in {0, 1, 2, 3, 7}
    const uint32_t L0 = 0x5cf5d3edu, L1 = 0x5812631au, L2 = 0xa2f79cd6u, L3 = 0x14def9deu, L7 = 0x10000000u;
    uint32_t m[8], r[9];
    lo = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        sc_mac0(lo, hi, t[k]);
        if (k >= 1) sc_mac(lo, hi, m[k - 1], L1);
        if (k >= 2) sc_mac(lo, hi, m[k - 2], L2);
        if (k >= 3) sc_mac(lo, hi, m[k - 3], L3);
        if (k >= 7) sc_mac(lo, hi, m[k - 7], L7);
        m[k] = (uint32_t)lo * BPG_SC_NINV;
        sc_mac(lo, hi, m[k], L0);                                         // the low word of the column is now zero
        lo = (lo >> 32) | ((uint64_t)hi << 32);
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
        sc_mac0(lo, hi, t[k]);
        if (k - 1 <= 7) sc_mac(lo, hi, m[k - 1], L1);
        if (k - 2 <= 7) sc_mac(lo, hi, m[k - 2], L2);
        if (k - 3 <= 7) sc_mac(lo, hi, m[k - 3], L3);
        if (k - 7 <= 7) sc_mac(lo, hi, m[k - 7], L7);
        r[k - 8] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
    }
    r[8] = (uint32_t)lo;                                                  // (t + m l) / 2^256 < 2 l
    return sc_cond_sub(r);
}
#else
BPG_HD scm sc_mont_mul(const scm &a, const scm &b) {
    const scm L = SC_L();
    uint32_t t[10];
    BPG_UNROLL for (int i = 0; i < 10; i++) t[i] = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        BPG_UNROLL for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)x; c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c; t[8] = (uint32_t)x; t[9] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * BPG_SC_NINV;
        c = ((uint64_t)m * L.v[0] + t[0]) >> 32;
        BPG_UNROLL for (int j = 1; j < 4; j++) {
            uint64_t y = (uint64_t)m * L.v[j] + t[j] + c;
            t[j - 1] = (uint32_t)y; c = y >> 32;
        }
        // limbs 4..6 of l are zero
        BPG_UNROLL for (int j = 4; j < 7; j++) { uint64_t y = (uint64_t)t[j] + c; t[j - 1] = (uint32_t)y; c = y >> 32; }
        { uint64_t y = (uint64_t)m * L.v[7] + t[7] + c; t[6] = (uint32_t)y; c = y >> 32; }
        uint64_t y = (uint64_t)t[8] + c; t[7] = (uint32_t)y;
        t[8] = t[9] + (uint32_t)(y >> 32); t[9] = 0;
    }
    return sc_cond_sub(t);
}
#endif

BPG_HD scm sc_add(const scm &a, const scm &b) {
    uint32_t t[9]; uint64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    t[8] = (uint32_t)c;
    return sc_cond_sub(t);
}

BPG_HD scm sc_sub(const scm &a, const scm &b) {
    const scm L = SC_L();
    scm r; int64_t c = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { c += (int64_t)a.v[i] - (int64_t)b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    uint32_t m = (uint32_t)(c >> 63);           // all ones when a < b
    uint64_t d = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) { d += (uint64_t)r.v[i] + (L.v[i] & m); r.v[i] = (uint32_t)d; d >>= 32; }
    return r;
}

BPG_HD scm sc_neg(const scm &a) { return sc_sub(sc_zero(), a); }
BPG_HD uint32_t sc_iszero(const scm &a) { uint32_t o = 0; BPG_UNROLL for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }

// any 256-bit little-endian integer (8 words) -> Montgomery form of (x mod l)
BPG_HD scm sc_from_words(const uint32_t *w) { scm x; BPG_UNROLL for (int i = 0; i < 8; i++) x.v[i] = w[i]; return sc_mont_mul(x, SC_RR()); }
// 512-bit little-endian integer (16 words) -> Montgomery form of (x mod l)   [Scalar::from_bytes_mod_order_wide]
BPG_HD scm sc_from_wide_words(const uint32_t *w) {
    scm lo, hi;
    BPG_UNROLL for (int i = 0; i < 8; i++) { lo.v[i] = w[i]; hi.v[i] = w[8 + i]; }
    return sc_add(sc_mont_mul(lo, SC_RR()), sc_mont_mul(hi, SC_RRR()));
}
// Montgomery form -> canonical integer words
// = a / 2^256 mod l: the Montgomery REDUCTION alone (a product by one spends 64 multiply-adds on zeros: ~330 instructions, this is ~190; every term of every
// multiscalar sum and every thread of the table-driven tail converts its scalar).  Eight steps, each adds m * l with m = t0 * (-1/l) so that the low limb
// vanishes; l has five non-zero limbs.  a < l, so the result is below l + 1: one conditional subtraction.
BPG_HD void sc_to_words(uint32_t *w, const scm &a) {
    const uint32_t L0 = 0x5cf5d3edu, L1 = 0x5812631au, L2 = 0xa2f79cd6u, L3 = 0x14def9deu, L7 = 0x10000000u;
    uint32_t t[9];
    BPG_UNROLL for (int i = 0; i < 8; i++) t[i] = a.v[i];
    t[8] = 0;
    BPG_UNROLL for (int i = 0; i < 8; i++) {
        const uint32_t m = t[0] * BPG_SC_NINV;
        uint64_t c = ((uint64_t)t[0] + (uint64_t)m * L0) >> 32;
        c += (uint64_t)t[1] + (uint64_t)m * L1; t[0] = (uint32_t)c; c >>= 32;
        c += (uint64_t)t[2] + (uint64_t)m * L2; t[1] = (uint32_t)c; c >>= 32;
        c += (uint64_t)t[3] + (uint64_t)m * L3; t[2] = (uint32_t)c; c >>= 32;
        c += t[4]; t[3] = (uint32_t)c; c >>= 32;
        c += t[5]; t[4] = (uint32_t)c; c >>= 32;
        c += t[6]; t[5] = (uint32_t)c; c >>= 32;
        c += (uint64_t)t[7] + (uint64_t)m * L7; t[6] = (uint32_t)c; c >>= 32;
        c += t[8]; t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
    }
    const scm r = sc_cond_sub(t);
    BPG_UNROLL for (int i = 0; i < 8; i++) w[i] = r.v[i];
}

}  // namespace bpg
