// Host-side Fiat-Shamir transcript (product code): Keccak-f[1600] -> STROBE-128 -> Merlin v1.0 -> TranscriptRng.
// Mirrors merlin::Transcript as the reference drives it (src/bin/prover.rs:52 Transcript::new(filename),
// src/cs_buffer.rs:90-92 transcript()), plus the labels of dalek bulletproofs' TranscriptProtocol.
// The transcript stays on the host: the prover's challenge chain is serial and each step is ~1 us, while the
// 2n+8 TranscriptRng draws of one proof (one permutation per 64-byte draw) are the only sizeable serial cost and
// are overlapped with the A_I/A_O multiscalar kernels (see engine.hip).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "scalar.hpp"

namespace bpg {

static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600], scalar: fully unrolled round body on 25 lane variables.
static inline void keccak_f1600_scalar(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    uint64_t a00 = s[0], a01 = s[1], a02 = s[2], a03 = s[3], a04 = s[4], a05 = s[5], a06 = s[6], a07 = s[7], a08 = s[8], a09 = s[9],
             a10 = s[10], a11 = s[11], a12 = s[12], a13 = s[13], a14 = s[14], a15 = s[15], a16 = s[16], a17 = s[17], a18 = s[18],
             a19 = s[19], a20 = s[20], a21 = s[21], a22 = s[22], a23 = s[23], a24 = s[24];
    for (int r = 0; r < 24; r++) {
        uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                 c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
        // theta + rho + pi : b[y][2x+3y] = rot(a[x][y])
        uint64_t b00 = a00 ^ d0, b01 = rotl64(a06 ^ d1, 44), b02 = rotl64(a12 ^ d2, 43), b03 = rotl64(a18 ^ d3, 21), b04 = rotl64(a24 ^ d4, 14);
        uint64_t b05 = rotl64(a03 ^ d3, 28), b06 = rotl64(a09 ^ d4, 20), b07 = rotl64(a10 ^ d0, 3), b08 = rotl64(a16 ^ d1, 45), b09 = rotl64(a22 ^ d2, 61);
        uint64_t b10 = rotl64(a01 ^ d1, 1), b11 = rotl64(a07 ^ d2, 6), b12 = rotl64(a13 ^ d3, 25), b13 = rotl64(a19 ^ d4, 8), b14 = rotl64(a20 ^ d0, 18);
        uint64_t b15 = rotl64(a04 ^ d4, 27), b16 = rotl64(a05 ^ d0, 36), b17 = rotl64(a11 ^ d1, 10), b18 = rotl64(a17 ^ d2, 15), b19 = rotl64(a23 ^ d3, 56);
        uint64_t b20 = rotl64(a02 ^ d2, 62), b21 = rotl64(a08 ^ d3, 55), b22 = rotl64(a14 ^ d4, 39), b23 = rotl64(a15 ^ d0, 41), b24 = rotl64(a21 ^ d1, 2);
        // chi
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        a00 ^= RC[r];
    }
    s[0] = a00; s[1] = a01; s[2] = a02; s[3] = a03; s[4] = a04; s[5] = a05; s[6] = a06; s[7] = a07; s[8] = a08; s[9] = a09;
    s[10] = a10; s[11] = a11; s[12] = a12; s[13] = a13; s[14] = a14; s[15] = a15; s[16] = a16; s[17] = a17; s[18] = a18; s[19] = a19;
    s[20] = a20; s[21] = a21; s[22] = a22; s[23] = a23; s[24] = a24;
}

#if defined(__x86_64__)
// Keccak-f[1600] with AVX-512F: one zmm per plane y (lanes x = 0..4 in elements 0..4).  theta and rho act inside planes,
// pi is split in two: "pi1" permutes each plane so that element y' of register j holds B[x' = j][y'], which lets chi run
// ACROSS registers and leaves the state transposed (register = x, element = y); "pi2" transposes back with
// unpack / two-source permutes.  About 38 vector instructions per round against ~150 scalar ones; the TranscriptRng
// chain of a 2^20-gate proof is two million dependent permutations, so this is the prover's serial floor.
__attribute__((target("avx512f"))) static inline void keccak_f1600_avx512(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    const __mmask8 m5 = 0x1f;
    __m512i P0 = _mm512_maskz_loadu_epi64(m5, s), P1 = _mm512_maskz_loadu_epi64(m5, s + 5), P2 = _mm512_maskz_loadu_epi64(m5, s + 10),
            P3 = _mm512_maskz_loadu_epi64(m5, s + 15), P4 = _mm512_maskz_loadu_epi64(m5, s + 20);
    const __m512i prev = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), next = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);
    // rho offsets r[x][y], one vector per plane y
    const __m512i rho0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), rho1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0),
                  rho2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), rho3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0),
                  rho4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
    // pi1: Q_j[y'] = P_j[(j + 3 y') mod 5]
    const __m512i pi0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), pi1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7),
                  pi2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7), pi3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7),
                  pi4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
    // pi2 (5x5 transpose) index vectors; +8 selects the second source
    const __m512i t_s1 = _mm512_setr_epi64(0, 1, 2, 3, 4, 5, 8 + 0, 8 + 2), t_s2 = _mm512_setr_epi64(0, 1, 2, 3, 4, 5, 8 + 1, 8 + 3),
                  t_bg = _mm512_setr_epi64(0, 1, 8 + 0, 8 + 1, 6, 5, 6, 7), t_km = _mm512_setr_epi64(2, 3, 8 + 2, 8 + 3, 7, 5, 6, 7),
                  t_s3 = _mm512_setr_epi64(4, 5, 8 + 4, 8 + 5, 4, 5, 6, 7);
    for (int r = 0; r < 24; r++) {
        // theta
        __m512i C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
        __m512i D = _mm512_xor_si512(_mm512_permutexvar_epi64(prev, C), _mm512_rol_epi64(_mm512_permutexvar_epi64(next, C), 1));
        // theta + rho
        P0 = _mm512_rolv_epi64(_mm512_xor_si512(P0, D), rho0); P1 = _mm512_rolv_epi64(_mm512_xor_si512(P1, D), rho1);
        P2 = _mm512_rolv_epi64(_mm512_xor_si512(P2, D), rho2); P3 = _mm512_rolv_epi64(_mm512_xor_si512(P3, D), rho3);
        P4 = _mm512_rolv_epi64(_mm512_xor_si512(P4, D), rho4);
        // pi1
        __m512i Q0 = _mm512_permutexvar_epi64(pi0, P0), Q1 = _mm512_permutexvar_epi64(pi1, P1), Q2 = _mm512_permutexvar_epi64(pi2, P2),
                Q3 = _mm512_permutexvar_epi64(pi3, P3), Q4 = _mm512_permutexvar_epi64(pi4, P4);
        // chi across registers: R_x = Q_x ^ (~Q_{x+1} & Q_{x+2})   (0xD2 = a ^ (~b & c))
        __m512i R0 = _mm512_ternarylogic_epi64(Q0, Q1, Q2, 0xD2), R1 = _mm512_ternarylogic_epi64(Q1, Q2, Q3, 0xD2),
                R2 = _mm512_ternarylogic_epi64(Q2, Q3, Q4, 0xD2), R3 = _mm512_ternarylogic_epi64(Q3, Q4, Q0, 0xD2),
                R4 = _mm512_ternarylogic_epi64(Q4, Q0, Q1, 0xD2);
        // iota: A[0][0] is element 0 of R0
        R0 = _mm512_xor_si512(R0, _mm512_maskz_set1_epi64(0x01, (long long)RC[r]));
        // pi2: transpose back, P_y[x] = R_x[y]
        __m512i a0 = _mm512_unpacklo_epi64(R0, R1), a1 = _mm512_unpacklo_epi64(R2, R3);
        __m512i b0 = _mm512_unpackhi_epi64(R0, R1), b1 = _mm512_unpackhi_epi64(R2, R3);
        a0 = _mm512_permutex2var_epi64(a0, t_s1, R4);      // elements 6,7 <- R4[0], R4[2]
        b0 = _mm512_permutex2var_epi64(b0, t_s2, R4);      // elements 6,7 <- R4[1], R4[3]
        P0 = _mm512_permutex2var_epi64(a0, t_bg, a1);
        P1 = _mm512_permutex2var_epi64(b0, t_bg, b1);
        P2 = _mm512_permutex2var_epi64(a0, t_km, a1);
        P3 = _mm512_permutex2var_epi64(b0, t_km, b1);
        P4 = _mm512_mask_blend_epi64(0x10, _mm512_permutex2var_epi64(a0, t_s3, a1), R4);
    }
    _mm512_mask_storeu_epi64(s, m5, P0); _mm512_mask_storeu_epi64(s + 5, m5, P1); _mm512_mask_storeu_epi64(s + 10, m5, P2);
    _mm512_mask_storeu_epi64(s + 15, m5, P3); _mm512_mask_storeu_epi64(s + 20, m5, P4);
}
// Keccak-f[1600] on 64-bit lanes held one per XMM register (AVX-512VL): 32 vector registers hold the whole state plus the
// theta vector (no spills, which is what limits the 16-GPR scalar code), and VPTERNLOGQ does chi and the 5-way column
// parity in one or two instructions each: ~95 operations per round against ~130 (+ spill traffic) for the scalar code.
#define KX_X3(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0x96)
#define KX_CHI(a, b, c) _mm_ternarylogic_epi64(a, b, c, 0xD2)
#define KX_RX(a, d, n) _mm_rol_epi64(_mm_xor_si128(a, d), n)
#define KX_LD(s, i) _mm_loadl_epi64(reinterpret_cast<const __m128i *>((s) + (i)))
#define KX_ST(s, i, v) _mm_storel_epi64(reinterpret_cast<__m128i *>((s) + (i)), v)
#define KX_DECLARE_LOAD(s) \
    __m128i a00 = KX_LD(s, 0), a01 = KX_LD(s, 1), a02 = KX_LD(s, 2), a03 = KX_LD(s, 3), a04 = KX_LD(s, 4), a05 = KX_LD(s, 5), a06 = KX_LD(s, 6), \
            a07 = KX_LD(s, 7), a08 = KX_LD(s, 8), a09 = KX_LD(s, 9), a10 = KX_LD(s, 10), a11 = KX_LD(s, 11), a12 = KX_LD(s, 12), a13 = KX_LD(s, 13), \
            a14 = KX_LD(s, 14), a15 = KX_LD(s, 15), a16 = KX_LD(s, 16), a17 = KX_LD(s, 17), a18 = KX_LD(s, 18), a19 = KX_LD(s, 19), a20 = KX_LD(s, 20), \
            a21 = KX_LD(s, 21), a22 = KX_LD(s, 22), a23 = KX_LD(s, 23), a24 = KX_LD(s, 24)
#define KX_STORE(s) \
    KX_ST(s, 0, a00); KX_ST(s, 1, a01); KX_ST(s, 2, a02); KX_ST(s, 3, a03); KX_ST(s, 4, a04); KX_ST(s, 5, a05); KX_ST(s, 6, a06); KX_ST(s, 7, a07); \
    KX_ST(s, 8, a08); KX_ST(s, 9, a09); KX_ST(s, 10, a10); KX_ST(s, 11, a11); KX_ST(s, 12, a12); KX_ST(s, 13, a13); KX_ST(s, 14, a14); KX_ST(s, 15, a15); \
    KX_ST(s, 16, a16); KX_ST(s, 17, a17); KX_ST(s, 18, a18); KX_ST(s, 19, a19); KX_ST(s, 20, a20); KX_ST(s, 21, a21); KX_ST(s, 22, a22); KX_ST(s, 23, a23); \
    KX_ST(s, 24, a24)
#define KX_24_ROUNDS(RC) \
    for (int r_ = 0; r_ < 24; r_++) { \
        const __m128i c0 = KX_X3(KX_X3(a00, a05, a10), a15, a20), c1 = KX_X3(KX_X3(a01, a06, a11), a16, a21), \
                      c2 = KX_X3(KX_X3(a02, a07, a12), a17, a22), c3 = KX_X3(KX_X3(a03, a08, a13), a18, a23), \
                      c4 = KX_X3(KX_X3(a04, a09, a14), a19, a24); \
        const __m128i d0 = _mm_xor_si128(c4, _mm_rol_epi64(c1, 1)), d1 = _mm_xor_si128(c0, _mm_rol_epi64(c2, 1)), \
                      d2 = _mm_xor_si128(c1, _mm_rol_epi64(c3, 1)), d3 = _mm_xor_si128(c2, _mm_rol_epi64(c4, 1)), \
                      d4 = _mm_xor_si128(c3, _mm_rol_epi64(c0, 1)); \
        __m128i b0, b1, b2, b3, b4; \
        b0 = _mm_xor_si128(a00, d0); b1 = KX_RX(a06, d1, 44); b2 = KX_RX(a12, d2, 43); b3 = KX_RX(a18, d3, 21); b4 = KX_RX(a24, d4, 14); \
        const __m128i n00 = _mm_xor_si128(KX_CHI(b0, b1, b2), KX_LD(RC, r_)), \
                      n01 = KX_CHI(b1, b2, b3), n02 = KX_CHI(b2, b3, b4), n03 = KX_CHI(b3, b4, b0), n04 = KX_CHI(b4, b0, b1); \
        b0 = KX_RX(a03, d3, 28); b1 = KX_RX(a09, d4, 20); b2 = KX_RX(a10, d0, 3); b3 = KX_RX(a16, d1, 45); b4 = KX_RX(a22, d2, 61); \
        const __m128i n05 = KX_CHI(b0, b1, b2), n06 = KX_CHI(b1, b2, b3), n07 = KX_CHI(b2, b3, b4), n08 = KX_CHI(b3, b4, b0), n09 = KX_CHI(b4, b0, b1); \
        b0 = KX_RX(a01, d1, 1); b1 = KX_RX(a07, d2, 6); b2 = KX_RX(a13, d3, 25); b3 = KX_RX(a19, d4, 8); b4 = KX_RX(a20, d0, 18); \
        const __m128i n10 = KX_CHI(b0, b1, b2), n11 = KX_CHI(b1, b2, b3), n12 = KX_CHI(b2, b3, b4), n13 = KX_CHI(b3, b4, b0), n14 = KX_CHI(b4, b0, b1); \
        b0 = KX_RX(a04, d4, 27); b1 = KX_RX(a05, d0, 36); b2 = KX_RX(a11, d1, 10); b3 = KX_RX(a17, d2, 15); b4 = KX_RX(a23, d3, 56); \
        const __m128i n15 = KX_CHI(b0, b1, b2), n16 = KX_CHI(b1, b2, b3), n17 = KX_CHI(b2, b3, b4), n18 = KX_CHI(b3, b4, b0), n19 = KX_CHI(b4, b0, b1); \
        b0 = KX_RX(a02, d2, 62); b1 = KX_RX(a08, d3, 55); b2 = KX_RX(a14, d4, 39); b3 = KX_RX(a15, d0, 41); b4 = KX_RX(a21, d1, 2); \
        const __m128i n20 = KX_CHI(b0, b1, b2), n21 = KX_CHI(b1, b2, b3), n22 = KX_CHI(b2, b3, b4), n23 = KX_CHI(b3, b4, b0), n24 = KX_CHI(b4, b0, b1); \
        a00 = n00; a01 = n01; a02 = n02; a03 = n03; a04 = n04; a05 = n05; a06 = n06; a07 = n07; a08 = n08; a09 = n09; \
        a10 = n10; a11 = n11; a12 = n12; a13 = n13; a14 = n14; a15 = n15; a16 = n16; a17 = n17; a18 = n18; a19 = n19; \
        a20 = n20; a21 = n21; a22 = n22; a23 = n23; a24 = n24; \
    }
static const uint64_t KECCAK_RC_XMM[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
__attribute__((target("avx512f,avx512vl"))) static inline void keccak_f1600_xmm(uint64_t s[25]) {
    KX_DECLARE_LOAD(s);
    KX_24_ROUNDS(KECCAK_RC_XMM)
    KX_STORE(s);
}
// TranscriptRng steady state, `count` draws of 64 bytes with the sponge kept in registers between draws.  Per draw
// (STROBE meta_ad(le32(64)) + prf(64) with the position at 64): bytes 64..73 and 167 take fixed XOR constants, one
// permutation, then the first 64 bytes are the output and are zeroed (PRF).  Caller has folded the old pos_begin into byte 64.
__attribute__((target("avx512f,avx512vl"))) static inline void strobe_rng_bulk64_xmm(uint64_t s[25], uint8_t *dest, size_t count,
                                                                                     uint64_t k8, uint64_t k9, uint64_t k20) {
    KX_DECLARE_LOAD(s);
    const __m128i x8 = _mm_cvtsi64_si128((long long)k8), x9 = _mm_cvtsi64_si128((long long)k9), x20 = _mm_cvtsi64_si128((long long)k20);
    for (size_t i = 0; i < count; i++) {
        a08 = _mm_xor_si128(a08, x8); a09 = _mm_xor_si128(a09, x9); a20 = _mm_xor_si128(a20, x20);
        KX_24_ROUNDS(KECCAK_RC_XMM)
        uint64_t *o = reinterpret_cast<uint64_t *>(dest + 64 * i);
        KX_ST(o, 0, a00); KX_ST(o, 1, a01); KX_ST(o, 2, a02); KX_ST(o, 3, a03); KX_ST(o, 4, a04); KX_ST(o, 5, a05); KX_ST(o, 6, a06); KX_ST(o, 7, a07);
        a00 = a01 = a02 = a03 = a04 = a05 = a06 = a07 = _mm_setzero_si128();
    }
    KX_STORE(s);
}

// EIGHT independent sponges at once: the same lanes-in-registers round body on ZMM registers, lane v of every register belonging to
// sponge v.  On Zen 5 a 512-bit VPTERNLOGQ / VPROLQ issues at the rate of a 128-bit one (profiles/r02_zen5_vector_width.txt), so eight TranscriptRng
// chains of eight different proofs cost one core about what one chain costs it - the serial chain of ONE proof gets no shorter, a core's
// chain THROUGHPUT goes up eightfold.  Same bytes per sponge as strobe_rng_bulk64_xmm (tests/test_device_arith_host.py).
#define KZ_X3(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0x96)
#define KZ_CHI(a, b, c) _mm512_ternarylogic_epi64(a, b, c, 0xD2)
#define KZ_RX(a, d, n) _mm512_rol_epi64(_mm512_xor_si512(a, d), n)
#define KZ_GATHER(i) _mm512_set_epi64((long long)st[7][i], (long long)st[6][i], (long long)st[5][i], (long long)st[4][i], (long long)st[3][i], (long long)st[2][i], (long long)st[1][i], (long long)st[0][i])
#define KZ_SCATTER(i, v) do { _mm512_store_si512(tmp, v); for (int l_ = 0; l_ < 8; l_++) st[l_][i] = tmp[l_]; } while (0)
#define KZ_24_ROUNDS(RC) \
    for (int r_ = 0; r_ < 24; r_++) { \
        const __m512i c0 = KZ_X3(KZ_X3(a00, a05, a10), a15, a20), c1 = KZ_X3(KZ_X3(a01, a06, a11), a16, a21), \
                      c2 = KZ_X3(KZ_X3(a02, a07, a12), a17, a22), c3 = KZ_X3(KZ_X3(a03, a08, a13), a18, a23), \
                      c4 = KZ_X3(KZ_X3(a04, a09, a14), a19, a24); \
        const __m512i d0 = _mm512_xor_si512(c4, _mm512_rol_epi64(c1, 1)), d1 = _mm512_xor_si512(c0, _mm512_rol_epi64(c2, 1)), \
                      d2 = _mm512_xor_si512(c1, _mm512_rol_epi64(c3, 1)), d3 = _mm512_xor_si512(c2, _mm512_rol_epi64(c4, 1)), \
                      d4 = _mm512_xor_si512(c3, _mm512_rol_epi64(c0, 1)); \
        __m512i b0, b1, b2, b3, b4; \
        b0 = _mm512_xor_si512(a00, d0); b1 = KZ_RX(a06, d1, 44); b2 = KZ_RX(a12, d2, 43); b3 = KZ_RX(a18, d3, 21); b4 = KZ_RX(a24, d4, 14); \
        const __m512i n00 = _mm512_xor_si512(KZ_CHI(b0, b1, b2), _mm512_set1_epi64((long long)(RC)[r_])), \
                      n01 = KZ_CHI(b1, b2, b3), n02 = KZ_CHI(b2, b3, b4), n03 = KZ_CHI(b3, b4, b0), n04 = KZ_CHI(b4, b0, b1); \
        b0 = KZ_RX(a03, d3, 28); b1 = KZ_RX(a09, d4, 20); b2 = KZ_RX(a10, d0, 3); b3 = KZ_RX(a16, d1, 45); b4 = KZ_RX(a22, d2, 61); \
        const __m512i n05 = KZ_CHI(b0, b1, b2), n06 = KZ_CHI(b1, b2, b3), n07 = KZ_CHI(b2, b3, b4), n08 = KZ_CHI(b3, b4, b0), n09 = KZ_CHI(b4, b0, b1); \
        b0 = KZ_RX(a01, d1, 1); b1 = KZ_RX(a07, d2, 6); b2 = KZ_RX(a13, d3, 25); b3 = KZ_RX(a19, d4, 8); b4 = KZ_RX(a20, d0, 18); \
        const __m512i n10 = KZ_CHI(b0, b1, b2), n11 = KZ_CHI(b1, b2, b3), n12 = KZ_CHI(b2, b3, b4), n13 = KZ_CHI(b3, b4, b0), n14 = KZ_CHI(b4, b0, b1); \
        b0 = KZ_RX(a04, d4, 27); b1 = KZ_RX(a05, d0, 36); b2 = KZ_RX(a11, d1, 10); b3 = KZ_RX(a17, d2, 15); b4 = KZ_RX(a23, d3, 56); \
        const __m512i n15 = KZ_CHI(b0, b1, b2), n16 = KZ_CHI(b1, b2, b3), n17 = KZ_CHI(b2, b3, b4), n18 = KZ_CHI(b3, b4, b0), n19 = KZ_CHI(b4, b0, b1); \
        b0 = KZ_RX(a02, d2, 62); b1 = KZ_RX(a08, d3, 55); b2 = KZ_RX(a14, d4, 39); b3 = KZ_RX(a15, d0, 41); b4 = KZ_RX(a21, d1, 2); \
        const __m512i n20 = KZ_CHI(b0, b1, b2), n21 = KZ_CHI(b1, b2, b3), n22 = KZ_CHI(b2, b3, b4), n23 = KZ_CHI(b3, b4, b0), n24 = KZ_CHI(b4, b0, b1); \
        a00 = n00; a01 = n01; a02 = n02; a03 = n03; a04 = n04; a05 = n05; a06 = n06; a07 = n07; a08 = n08; a09 = n09; \
        a10 = n10; a11 = n11; a12 = n12; a13 = n13; a14 = n14; a15 = n15; a16 = n16; a17 = n17; a18 = n18; a19 = n19; \
        a20 = n20; a21 = n21; a22 = n22; a23 = n23; a24 = n24; \
    }
// `count` steady-state TranscriptRng draws (see strobe_rng_bulk64_xmm) for each of eight sponges st[v] -> dest[v]
__attribute__((target("avx512f"))) static inline void strobe_rng_bulk64_x8(uint64_t *const st[8], uint8_t *const dest[8], size_t count,
                                                                            uint64_t k8, uint64_t k9, uint64_t k20) {
    alignas(64) uint64_t tmp[8];
    alignas(64) uint64_t out[8][8];
    __m512i a00 = KZ_GATHER(0), a01 = KZ_GATHER(1), a02 = KZ_GATHER(2), a03 = KZ_GATHER(3), a04 = KZ_GATHER(4), a05 = KZ_GATHER(5), a06 = KZ_GATHER(6),
            a07 = KZ_GATHER(7), a08 = KZ_GATHER(8), a09 = KZ_GATHER(9), a10 = KZ_GATHER(10), a11 = KZ_GATHER(11), a12 = KZ_GATHER(12), a13 = KZ_GATHER(13),
            a14 = KZ_GATHER(14), a15 = KZ_GATHER(15), a16 = KZ_GATHER(16), a17 = KZ_GATHER(17), a18 = KZ_GATHER(18), a19 = KZ_GATHER(19), a20 = KZ_GATHER(20),
            a21 = KZ_GATHER(21), a22 = KZ_GATHER(22), a23 = KZ_GATHER(23), a24 = KZ_GATHER(24);
    const __m512i x8 = _mm512_set1_epi64((long long)k8), x9 = _mm512_set1_epi64((long long)k9), x20 = _mm512_set1_epi64((long long)k20);
    for (size_t i = 0; i < count; i++) {
        a08 = _mm512_xor_si512(a08, x8); a09 = _mm512_xor_si512(a09, x9); a20 = _mm512_xor_si512(a20, x20);
        KZ_24_ROUNDS(KECCAK_RC_XMM)
        _mm512_store_si512(out[0], a00); _mm512_store_si512(out[1], a01); _mm512_store_si512(out[2], a02); _mm512_store_si512(out[3], a03);
        _mm512_store_si512(out[4], a04); _mm512_store_si512(out[5], a05); _mm512_store_si512(out[6], a06); _mm512_store_si512(out[7], a07);
        for (int v = 0; v < 8; v++) {                        // word j of sponge v is lane v of register j
            uint64_t *o = reinterpret_cast<uint64_t *>(dest[v] + 64 * i);
            for (int j = 0; j < 8; j++) o[j] = out[j][v];
        }
        a00 = a01 = a02 = a03 = a04 = a05 = a06 = a07 = _mm512_setzero_si512();
    }
    KZ_SCATTER(0, a00); KZ_SCATTER(1, a01); KZ_SCATTER(2, a02); KZ_SCATTER(3, a03); KZ_SCATTER(4, a04); KZ_SCATTER(5, a05); KZ_SCATTER(6, a06);
    KZ_SCATTER(7, a07); KZ_SCATTER(8, a08); KZ_SCATTER(9, a09); KZ_SCATTER(10, a10); KZ_SCATTER(11, a11); KZ_SCATTER(12, a12); KZ_SCATTER(13, a13);
    KZ_SCATTER(14, a14); KZ_SCATTER(15, a15); KZ_SCATTER(16, a16); KZ_SCATTER(17, a17); KZ_SCATTER(18, a18); KZ_SCATTER(19, a19); KZ_SCATTER(20, a20);
    KZ_SCATTER(21, a21); KZ_SCATTER(22, a22); KZ_SCATTER(23, a23); KZ_SCATTER(24, a24);
}
static inline bool keccak_have_x8() { static const bool ok = __builtin_cpu_supports("avx512f"); return ok; }

// Which one is fastest depends on the core (EPYC 9575F / Zen 5: lanes-in-XMM 154 ns, scalar 177 ns, planes-in-ZMM 228 ns, the
// cross-lane permutes having a long latency there; Xeon: the vector forms win by more), so a ~2 ms calibration at first
// use picks the implementation.  All three compute the same permutation (tests/test_host_logic.py); the choice never
// changes an output.  keccak_impl(): 0 scalar, 1 planes-in-ZMM, 2 lanes-in-XMM.
static inline int keccak_impl() {
    static const int impl = [] {
        if (!__builtin_cpu_supports("avx512f") || !__builtin_cpu_supports("avx512vl")) return 0;
        auto time_it = [](void (*f)(uint64_t *)) {
            uint64_t st[25];
            for (int i = 0; i < 25; i++) st[i] = 0x9e3779b97f4a7c15ULL * (uint64_t)(i + 1);
            uint64_t best = ~0ULL;
            for (int rep = 0; rep < 3; rep++) {
                uint64_t t0 = __builtin_ia32_rdtsc();
                for (int r = 0; r < 1500; r++) f(st);
                uint64_t dt = __builtin_ia32_rdtsc() - t0;
                if (dt < best) best = dt;
            }
            return best;
        };
        const uint64_t t[3] = {time_it([](uint64_t *st) { keccak_f1600_scalar(st); }), time_it([](uint64_t *st) { keccak_f1600_avx512(st); }),
                               time_it([](uint64_t *st) { keccak_f1600_xmm(st); })};
        int b = 0; for (int k = 1; k < 3; k++) if (t[k] < t[b]) b = k;
        return b;
    }();
    return impl;
}
static inline bool keccak_have_avx512() { return keccak_impl() != 0; }
static inline void keccak_f1600_host(uint64_t s[25]) {
    const int impl = keccak_impl();
    if (impl == 2) keccak_f1600_xmm(s); else if (impl == 1) keccak_f1600_avx512(s); else keccak_f1600_scalar(s);
}
#else
static inline int keccak_impl() { return 0; }
static inline bool keccak_have_x8() { return false; }
static inline bool keccak_have_avx512() { return false; }
static inline void keccak_f1600_host(uint64_t s[25]) { keccak_f1600_scalar(s); }
#endif

// SHAKE256 squeeze helper for the generator chains, and SHA3-512 for the Pedersen blinding base.
class Shake256 {
public:
    Shake256() { std::memset(st_, 0, sizeof st_); pos_ = 0; squeezing_ = false; }
    void absorb(const uint8_t *d, size_t n) { uint8_t *b = bytes(); for (size_t i = 0; i < n; i++) { b[pos_++] ^= d[i]; if (pos_ == 136) { keccak_f1600_host(st_); pos_ = 0; } } }
    void squeeze(uint8_t *out, size_t n) {
        uint8_t *b = bytes();
        if (!squeezing_) { b[pos_] ^= 0x1f; b[135] ^= 0x80; keccak_f1600_host(st_); pos_ = 0; squeezing_ = true; }
        while (n) {                                           // whole-block copies: the generator chains squeeze 64 bytes per generator
            if (pos_ == 136) { keccak_f1600_host(st_); pos_ = 0; }
            const size_t take = n < 136 - pos_ ? n : 136 - pos_;
            std::memcpy(out, b + pos_, take);
            out += take; pos_ += take; n -= take;
        }
    }
private:
    uint8_t *bytes() { return reinterpret_cast<uint8_t *>(st_); }
    uint64_t st_[25]; size_t pos_; bool squeezing_;
};

static inline void sha3_512_host(uint8_t out[64], const uint8_t *in, size_t n) {
    uint64_t st[25]; std::memset(st, 0, sizeof st); uint8_t *b = reinterpret_cast<uint8_t *>(st); size_t pos = 0;
    for (size_t i = 0; i < n; i++) { b[pos++] ^= in[i]; if (pos == 72) { keccak_f1600_host(st); pos = 0; } }
    b[pos] ^= 0x06; b[71] ^= 0x80; keccak_f1600_host(st); std::memcpy(out, st, 64);
}

// STROBE-128 restricted to the operations Merlin uses. The 203-byte wire image (200 state bytes, pos, pos_begin,
// cur_flags) is what crosses the C ABI (include/bpg.h: transcript_state).
class Strobe128 {
public:
    static constexpr int R = 166;
    enum : uint8_t { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };
    Strobe128() { std::memset(st_, 0, sizeof st_); pos_ = pos_begin_ = cur_flags_ = 0; }
    explicit Strobe128(const char *proto) : Strobe128() {
        uint8_t *b = bytes();
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        std::memcpy(b, hdr, 6); std::memcpy(b + 6, "STROBEv1.0.2", 12);
        keccak_f1600_host(st_);
        meta_ad(reinterpret_cast<const uint8_t *>(proto), std::strlen(proto), false);
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
    void ad(const uint8_t *d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
    void prf(uint8_t *d, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(d, n); }
    void key(const uint8_t *d, size_t n, bool more) { begin_op(FLAG_A | FLAG_C, more); overwrite(d, n); }
    // `count` TranscriptRng draws of 64 bytes each (meta_ad(le32(64)) + prf(64)); same bytes as the generic path
    void rng_draws64(uint8_t *dest, size_t count) {
        while (count && pos_ != 64) {                        // reach the steady state through the generic operations
            const uint8_t l4[4] = {64, 0, 0, 0}; meta_ad(l4, 4, false); prf(dest, 64, false); dest += 64; count--;
        }
        if (!count) return;
        // at pos 64: header {old pos_begin, M|A} + 64,0,0,0 + header {65, I|A|C} fill bytes 64..71; run_f pads at 72, 73 and R+1
        const uint64_t k8 = (0x12ULL << 8) | (64ULL << 16) | (65ULL << 48) | (0x07ULL << 56), k9 = 71ULL | (0x04ULL << 8), k20 = 0x80ULL << 56;
        st_[8] ^= pos_begin_;
#if defined(__x86_64__)
        if (keccak_impl() == 2) strobe_rng_bulk64_xmm(st_, dest, count, k8, k9, k20);
        else
#endif
        for (size_t i = 0; i < count; i++) {
            st_[8] ^= k8; st_[9] ^= k9; st_[20] ^= k20;
            keccak_f1600_host(st_);
            std::memcpy(dest + 64 * i, st_, 64); std::memset(st_, 0, 64);
        }
        pos_ = 64; pos_begin_ = 0; cur_flags_ = FLAG_I | FLAG_A | FLAG_C;
    }
    // `count` TranscriptRng draws for each of `lanes` (<= 8) independent sponges in lockstep: same bytes as rng_draws64 on each
    static void rng_draws64_multi(Strobe128 *const s[], uint8_t *const dest[], uint32_t lanes, size_t count) {
#if defined(__x86_64__)
        if (lanes > 1 && lanes <= 8 && keccak_have_x8() && count > 1) {
            // every sponge reaches the steady state (pos 64) through the generic operations: at most one draw each; all take the same number
            size_t head = 0;
            for (uint32_t v = 0; v < lanes; v++) if (s[v]->pos_ != 64) head = 1;
            for (uint32_t v = 0; v < lanes; v++) for (size_t i = 0; i < head; i++) { const uint8_t l4[4] = {64, 0, 0, 0}; s[v]->meta_ad(l4, 4, false); s[v]->prf(dest[v] + 64 * i, 64, false); }
            bool steady = true;
            for (uint32_t v = 0; v < lanes; v++) steady = steady && s[v]->pos_ == 64;
            if (steady) {
                const uint64_t k8 = (0x12ULL << 8) | (64ULL << 16) | (65ULL << 48) | (0x07ULL << 56), k9 = 71ULL | (0x04ULL << 8), k20 = 0x80ULL << 56;
                uint64_t idle_state[25]; alignas(64) static thread_local uint8_t idle_out[64 * 4096];
                std::memset(idle_state, 0, sizeof idle_state);
                uint64_t *st[8]; uint8_t *dst[8];
                for (uint32_t v = 0; v < 8; v++) {
                    if (v < lanes) { s[v]->st_[8] ^= s[v]->pos_begin_; st[v] = s[v]->st_; dst[v] = dest[v] + 64 * head; }
                    else { st[v] = idle_state; dst[v] = idle_out; }
                }
                size_t left = count - head;
                while (left) {                                   // idle lanes write into a 4096-draw scratch: chunk the call accordingly
                    const size_t chunk = left < 4096 ? left : 4096;
                    strobe_rng_bulk64_x8(st, dst, chunk, k8, k9, k20);
                    for (uint32_t v = 0; v < lanes; v++) dst[v] += 64 * chunk;
                    left -= chunk;
                }
                for (uint32_t v = 0; v < lanes; v++) { s[v]->pos_ = 64; s[v]->pos_begin_ = 0; s[v]->cur_flags_ = FLAG_I | FLAG_A | FLAG_C; }
                return;
            }
            for (uint32_t v = 0; v < lanes; v++) s[v]->rng_draws64(dest[v] + 64 * head, count - head);
            return;
        }
#endif
        for (uint32_t v = 0; v < lanes; v++) s[v]->rng_draws64(dest[v], count);
    }
    void export_state(uint8_t out[203]) const { std::memcpy(out, st_, 200); out[200] = pos_; out[201] = pos_begin_; out[202] = cur_flags_; }
    void import_state(const uint8_t in[203]) { std::memcpy(st_, in, 200); pos_ = in[200]; pos_begin_ = in[201]; cur_flags_ = in[202]; }
private:
    uint8_t *bytes() { return reinterpret_cast<uint8_t *>(st_); }
    void run_f() { uint8_t *b = bytes(); b[pos_] ^= pos_begin_; b[pos_ + 1] ^= 0x04; b[R + 1] ^= 0x80; keccak_f1600_host(st_); pos_ = 0; pos_begin_ = 0; }
    // fast paths: an operation that stays inside the current block touches the state with memcpy/memset-sized moves
    void absorb(const uint8_t *d, size_t n) {
        uint8_t *b = bytes();
        if (pos_ + n < (size_t)R) { for (size_t i = 0; i < n; i++) b[pos_ + i] ^= d[i]; pos_ = (uint8_t)(pos_ + n); return; }
        for (size_t i = 0; i < n; i++) { b[pos_++] ^= d[i]; if (pos_ == R) run_f(); }
    }
    void overwrite(const uint8_t *d, size_t n) {
        uint8_t *b = bytes();
        if (pos_ + n < (size_t)R) { std::memcpy(b + pos_, d, n); pos_ = (uint8_t)(pos_ + n); return; }
        for (size_t i = 0; i < n; i++) { b[pos_++] = d[i]; if (pos_ == R) run_f(); }
    }
    void squeeze(uint8_t *d, size_t n) {
        uint8_t *b = bytes();
        if (pos_ + n < (size_t)R) { std::memcpy(d, b + pos_, n); std::memset(b + pos_, 0, n); pos_ = (uint8_t)(pos_ + n); return; }
        for (size_t i = 0; i < n; i++) { d[i] = b[pos_]; b[pos_++] = 0; if (pos_ == R) run_f(); }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t old = pos_begin_;
        pos_begin_ = static_cast<uint8_t>(pos_ + 1); cur_flags_ = flags;
        const uint8_t hdr[2] = {old, flags};
        absorb(hdr, 2);
        if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
    }
    uint64_t st_[25]; uint8_t pos_, pos_begin_, cur_flags_;
};

class TranscriptRng {
public:
    explicit TranscriptRng(const Strobe128 &s) : s_(s) {}
    void fill_bytes(uint8_t *dest, size_t n) { uint8_t l4[4]; le32(l4, n); s_.meta_ad(l4, 4, false); s_.prf(dest, n, false); }
    void fill_draws64(uint8_t *dest, size_t count) { s_.rng_draws64(dest, count); }      // count x fill_bytes(.., 64), bulk
    static void fill_draws64_multi(TranscriptRng *const r[], uint8_t *const dest[], uint32_t lanes, size_t count) {      // lanes <= 8 generators in lockstep
        Strobe128 *s[8];
        for (uint32_t v = 0; v < lanes && v < 8; v++) s[v] = &r[v]->s_;
        Strobe128::rng_draws64_multi(s, dest, lanes, count);
    }
    Scalar random_scalar() { uint8_t b[64]; fill_bytes(b, 64); return Scalar::from_wide(b); }
    static void le32(uint8_t b[4], size_t n) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }
private:
    Strobe128 s_;
};

class Transcript {
public:
    Transcript() {}
    Transcript(const uint8_t *label, size_t n) : s_("Merlin v1.0") { append_message("dom-sep", label, n); }
    explicit Transcript(const std::string &label) : Transcript(reinterpret_cast<const uint8_t *>(label.data()), label.size()) {}
    static Transcript from_state(const uint8_t st[203]) { Transcript t; t.s_.import_state(st); return t; }
    void export_state(uint8_t out[203]) const { s_.export_state(out); }

    void append_message(const char *label, const uint8_t *msg, size_t n) {
        uint8_t l4[4]; TranscriptRng::le32(l4, n);
        s_.meta_ad(reinterpret_cast<const uint8_t *>(label), std::strlen(label), false);
        s_.meta_ad(l4, 4, true);
        s_.ad(msg, n, false);
    }
    void append_u64(const char *label, uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); append_message(label, b, 8); }
    void challenge_bytes(const char *label, uint8_t *out, size_t n) {
        uint8_t l4[4]; TranscriptRng::le32(l4, n);
        s_.meta_ad(reinterpret_cast<const uint8_t *>(label), std::strlen(label), false);
        s_.meta_ad(l4, 4, true);
        s_.prf(out, n, false);
    }
    // dalek bulletproofs TranscriptProtocol
    void r1cs_domain_sep() { append_message("dom-sep", reinterpret_cast<const uint8_t *>("r1cs v1"), 7); }
    void r1cs_1phase_domain_sep() { append_message("dom-sep", reinterpret_cast<const uint8_t *>("r1cs-1phase"), 11); }
    void innerproduct_domain_sep(uint64_t n) { append_message("dom-sep", reinterpret_cast<const uint8_t *>("ipp v1"), 6); append_u64("n", n); }
    void append_scalar(const char *label, const Scalar &s) { append_message(label, s.as_bytes(), 32); }
    void append_point(const char *label, const uint8_t p[32]) { append_message(label, p, 32); }
    Scalar challenge_scalar(const char *label) { uint8_t b[64]; challenge_bytes(label, b, 64); return Scalar::from_wide(b); }

    // build_rng().rekey_with_witness_bytes("v_blinding", ..)*.finalize(seed)
    TranscriptRng build_rng(const std::vector<Scalar> &v_blinding, const uint8_t seed[32]) const {
        Strobe128 s = s_;
        for (const Scalar &vb : v_blinding) {
            uint8_t l4[4]; TranscriptRng::le32(l4, 32);
            s.meta_ad(reinterpret_cast<const uint8_t *>("v_blinding"), 10, false);
            s.meta_ad(l4, 4, true);
            s.key(vb.as_bytes(), 32, false);
        }
        s.meta_ad(reinterpret_cast<const uint8_t *>("rng"), 3, false);
        s.key(seed, 32, false);
        return TranscriptRng(s);
    }
private:
    Strobe128 s_;
};

}  // namespace bpg
