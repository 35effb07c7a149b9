// inner-product argument: expanded scalars, grouped generator folds, table-driven tail - part of kernels.cuh (included from there, in this order; see its header for the kernel map and the data layout)
#pragma once

namespace bpg {

// ------------------------------------------------------------------------------------------------ inner-product rounds
// Generators are kept UNSCALED: actual G_p = Gamma * gf(p) * Gst[p], actual H_p = Eta * y^-p * gf(p) * Hst[p], where
// gf(p) = u_ch for p >= n in the first round (G_factors / H_factors of the R1CS padding) and 1 otherwise.
// Rounds are GROUPED: the stored generators are folded once per group of r rounds (k_fold_points), and sub-round j of a
// group works on the group-start tables of size M with expanded scalars.  With challenges u_1..u_j since the group start,
// M_j = M / 2^j, h = M_j / 2, and cG[t] = Gamma * prod_k (bit_k(t) ? u_k : u_k^-1), cH[t] = Eta * prod_k (bit_k(t) ? u_k^-1 : u_k)
// (bit_k(t) = bit j-k of t; k_tt_advance maintains both tables), the virtual folded generators are
//     G^(j)[i'] = sum_t cG[t] gf(p) Gst[p],   H^(j)[i'] = sum_t cH[t] y^-p gf(p) Hst[p],   p = i' + t*M_j
// so that for e = t*h + i (i < h):
//     L: a_lo[i] cG[t] gf on Gst[t*M_j + h + i],        b_hi[i] cH[t] y^-p gf on Hst[t*M_j + i]
//     R: a_hi[i] cG[t] gf on Gst[t*M_j + i],            b_lo[i] cH[t] y^-p gf on Hst[t*M_j + h + i]
// (j = 0 is the plain round).  Also accumulates c_L = <a_lo, b_hi>, c_R = <a_hi, b_lo> per block.
__global__ void __launch_bounds__(256) k_ipa_prep(const scm *__restrict__ a, const scm *__restrict__ b, const scm *__restrict__ yinvpow,
                                                  const scm *__restrict__ cG, const scm *__restrict__ cH, scm u_ch, uint32_t first_group, uint32_t n,
                                                  uint32_t lgh, uint32_t j,
                                                  scm *__restrict__ sLG, scm *__restrict__ sLH, scm *__restrict__ sRG, scm *__restrict__ sRH,
                                                  scm *__restrict__ partial /* gridDim.x * 2 */) {
    __shared__ scm lds[256];
    scm cL = sc_zero(), cR = sc_zero();
    const uint32_t h = 1u << lgh, count = h << j;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
        const uint32_t t = e >> lgh, i = e & (h - 1);
        const uint32_t plo = (t << (lgh + 1)) | i, phi = plo | h;
        const scm alo = a[i], ahi = a[h + i], blo = b[i], bhi = b[h + i];
        const scm g = cG[t], eh = cH[t];
        const bool padlo = first_group && plo >= n, padhi = first_group && phi >= n;
        scm v;
        v = sc_mont_mul(alo, g); if (padhi) v = sc_mont_mul(v, u_ch); sLG[e] = v;
        v = sc_mont_mul(sc_mont_mul(bhi, eh), yinvpow[plo]); if (padlo) v = sc_mont_mul(v, u_ch); sLH[e] = v;
        v = sc_mont_mul(ahi, g); if (padlo) v = sc_mont_mul(v, u_ch); sRG[e] = v;
        v = sc_mont_mul(sc_mont_mul(blo, eh), yinvpow[phi]); if (padhi) v = sc_mont_mul(v, u_ch); sRH[e] = v;
        if (t == 0) { cL = sc_add(cL, sc_mont_mul(alo, bhi)); cR = sc_add(cR, sc_mont_mul(ahi, blo)); }
    }
    scm r;
    r = block_sum_256(cL, lds); if (threadIdx.x == 0) partial[blockIdx.x * 2 + 0] = r;
    r = block_sum_256(cR, lds); if (threadIdx.x == 0) partial[blockIdx.x * 2 + 1] = r;
}
__global__ void k_set2(scm *__restrict__ c, uint32_t stride, scm v0, scm v1) { if (threadIdx.x == 0 && blockIdx.x == 0) { c[0] = v0; c[stride] = v1; } }
// (the Q = w*B term of L and R becomes a scalar on the fixed base B: c_L * w, c_R * w come out of k_reduce_partials_scaled, k_scalars.cuh)

__global__ void __launch_bounds__(256) k_ipa_fold_scalars(scm *__restrict__ a, scm *__restrict__ b, scm u, scm uinv, uint32_t h) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    a[i] = sc_add(sc_mont_mul(a[i], u), sc_mont_mul(uinv, a[h + i]));
    b[i] = sc_add(sc_mont_mul(b[i], uinv), sc_mont_mul(u, b[h + i]));
}

// ------------------------------------------------------------------------------------------------ table-driven IPA tail
// Once a round is down to M0 generators per side (M0 = 2^14 by default) the remaining lg M0 rounds are latency-bound:
// a generator fold is 253 dependent doublings (about 1.15 ms on one wave however few points there are) and the
// bucket-method MSM ends in another ~250.  Instead the generators are frozen at that level and every later L_k, R_k is
// computed over the SAME 2*M0 base points with expanded scalars (the verifier's s-vector idea): with challenges u_1..u_j
// drawn since the freeze, the virtual folded generator i' of size M_j = M0 / 2^j is
//     G^(j)[i'] = Gamma_0 * sum_t prod_k (bit_k(t) ? u_k : u_k^-1) * gf(p) * Gst[p],      p = i' + t*M_j
//     H^(j)[i'] = Eta_0   * sum_t prod_k (bit_k(t) ? u_k^-1 : u_k) * y^-p * gf(p) * Hst[p]
// (bit_k(t) = bit j-k of t; the y^-M_k factors of the H fold scalars collapse into y^-p).  Each base point then carries
// exactly one scalar per round and lands in exactly one of L_k / R_k.  A one-off table of k * 2^(4w) * P for every base
// point (w < 64 windows, k = 1..8, projective Niels, 64 KB per point) turns each of those scalar multiplications into 64
// table additions and no doubling at all; 8 threads share a point, partial sums go through an LDS tree.
#define TT_WINDOWS 64
#define TT_MULTS 8

// bases[p * 64 + w] = 2^(4w) * P_p for the 2*M0 + 1 base points G[0..M0), H[0..M0), B  (the only 252-doubling chain of the tail).
// One dependent chain per point, so the chain is shortened the way k_msm_horner does it: a block of four waves owns 64 points
// and wave k computes the k-th of the four independent field products of every doubling step (first the four squarings, then
// the four products); the operands travel through LDS in a word-major layout [coordinate][limb][lane] (no bank conflicts).
struct CoopLds { uint32_t c[4][8][64]; uint32_t s[4][8][64]; };
__device__ __forceinline__ fe coop_ld(const uint32_t (&a)[8][64], uint32_t lane) { fe r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = a[j][lane];
    return r; }
__device__ __forceinline__ void coop_st(uint32_t (&a)[8][64], uint32_t lane, const fe &x) {
#pragma unroll
    for (int j = 0; j < 8; j++) a[j][lane] = x.v[j]; }
__device__ __forceinline__ void coop_dbl(CoopLds &L, uint32_t wv, uint32_t lane) {
    // L.c = (X, Y, Z, T) of this lane's point -> doubled point in L.c
    const fe in = (wv == 3) ? fe_add(coop_ld(L.c[0], lane), coop_ld(L.c[1], lane)) : coop_ld(L.c[wv], lane);          // X, Y, Z, X+Y
    const fe sq = fe_sq(in);
    __syncthreads();
    coop_st(L.s[wv], lane, sq);                                     // XX, YY, ZZ, (X+Y)^2
    __syncthreads();
    const fe XX = coop_ld(L.s[0], lane), YY = coop_ld(L.s[1], lane);
    const fe YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX);
    fe a, b;
    if (wv == 1) { a = YpX; b = YmX; }                              // Y3 = YpX * YmX
    else {
        const fe ZZ = coop_ld(L.s[2], lane);
        const fe cT = fe_sub(fe_add(ZZ, ZZ), YmX), cX = fe_sub(coop_ld(L.s[3], lane), YpX);
        if (wv == 0) { a = cX; b = cT; }                            // X3 = cX * cT
        else if (wv == 2) { a = YmX; b = cT; }                      // Z3 = YmX * cT
        else { a = cX; b = YpX; }                                   // T3 = cX * YpX
    }
    coop_st(L.c[wv], lane, fe_mul(a, b));
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_tt_bases(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, const ge_niels *__restrict__ B,
                                                  ge_ext *__restrict__ bases, uint32_t M0) {
    __shared__ CoopLds L;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t count = 2 * M0 + 1;
    const uint32_t p = blockIdx.x * 64 + lane;
    const bool live = p < count;
    const uint32_t q_idx = live ? p : count - 1;                    // idle lanes shadow the last point: whole block stays in step
    if (wv == 0) {
        const ge_niels q = q_idx < M0 ? G[q_idx] : (q_idx < 2 * M0 ? H[q_idx - M0] : B[0]);
        const ge_ext e = ge_madd(ge_identity(), q);
        coop_st(L.c[0], lane, e.X); coop_st(L.c[1], lane, e.Y); coop_st(L.c[2], lane, e.Z); coop_st(L.c[3], lane, e.T);
    }
    __syncthreads();
    fe *dst = reinterpret_cast<fe *>(bases + (size_t)q_idx * TT_WINDOWS) + wv;       // wave k stores coordinate k (ge_ext = X, Y, Z, T)
    for (uint32_t w = 0; w < TT_WINDOWS; w++) {
        if (live) dst[4 * w] = coop_ld(L.c[wv], lane);
        if (w + 1 < TT_WINDOWS) { coop_dbl(L, wv, lane); coop_dbl(L, wv, lane); coop_dbl(L, wv, lane); coop_dbl(L, wv, lane); }
    }
}
// table[i * 8 + k] = (k + 1) * bases[i], i = p * 64 + w
__global__ void __launch_bounds__(256) k_tt_multiples(const ge_ext *__restrict__ bases, ge_pniels *__restrict__ table, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ge_pniels *dst = table + (size_t)i * TT_MULTS;
    const ge_ext b1 = bases[i];
    const ge_pniels n1 = ge_to_pniels(b1);
    dst[0] = n1;
    const ge_ext b2 = ge_dbl(b1); dst[1] = ge_to_pniels(b2);
    const ge_ext b3 = ge_add_pniels_signed(b2, n1, 0); dst[2] = ge_to_pniels(b3);
    const ge_ext b4 = ge_dbl(b2); dst[3] = ge_to_pniels(b4);
    const ge_ext b5 = ge_add_pniels_signed(b4, n1, 0); dst[4] = ge_to_pniels(b5);
    const ge_ext b6 = ge_dbl(b3); dst[5] = ge_to_pniels(b6);
    const ge_ext b7 = ge_add_pniels_signed(b6, n1, 0); dst[6] = ge_to_pniels(b7);
    const ge_ext b8 = ge_dbl(b4); dst[7] = ge_to_pniels(b8);
}
// per-base-point constant factors: fG[p] = gf(p), fH[p] = y^-p * gf(p); c tables start as {Gamma_0}, {Eta_0}
__global__ void __launch_bounds__(256) k_tt_factors(const scm *__restrict__ yinvpow, scm u_ch, uint32_t first_round, uint32_t n, uint32_t M0,
                                                    scm Gamma0, scm Eta0, scm *__restrict__ fG, scm *__restrict__ fH, scm *__restrict__ c0) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) { c0[0] = Gamma0; c0[M0] = Eta0; }
    if (p >= M0) return;
    const bool pad = first_round && p >= n;
    fG[p] = pad ? u_ch : SC_R1();
    fH[p] = pad ? sc_mont_mul(yinvpow[p], u_ch) : yinvpow[p];
}
// after challenge u: fold the scalar vectors (2h -> h) and extend the coefficient tables (cnt -> 2*cnt entries per side)
__global__ void __launch_bounds__(256) k_tt_advance(scm *__restrict__ a, scm *__restrict__ b, scm u, scm uinv, uint32_t h,
                                                    const scm *__restrict__ cprev, scm *__restrict__ cnext, uint32_t cnt, uint32_t M0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < h) {
        a[i] = sc_add(sc_mont_mul(a[i], u), sc_mont_mul(uinv, a[h + i]));
        b[i] = sc_add(sc_mont_mul(b[i], uinv), sc_mont_mul(u, b[h + i]));
    }
    if (i < cnt) {
        const scm g = cprev[i], e = cprev[M0 + i];
        cnext[2 * i] = sc_mont_mul(g, uinv); cnext[2 * i + 1] = sc_mont_mul(g, u);
        cnext[M0 + 2 * i] = sc_mont_mul(e, u); cnext[M0 + 2 * i + 1] = sc_mont_mul(e, uinv);
    }
}
// signed 4-bit digits without a carry chain: nibble w of (s + 0x88..8) minus 8 lies in [-8, 7]
__device__ __forceinline__ void tt_biased_words(uint32_t w[8], const scm &s) {
    sc_to_words(w, s);
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t t = (uint64_t)w[k] + 0x88888888ull + carry; w[k] = (uint32_t)t; carry = t >> 32; }
}
// sub-round j of the tail (h = M0 >> (j+1)): blockIdx.y = 0 accumulates L, 1 accumulates R; thread = (base point, 8 windows)
__global__ void __launch_bounds__(256) k_tt_round(const ge_pniels *__restrict__ table, const scm *__restrict__ a, const scm *__restrict__ b,
                                                  const scm *__restrict__ fG, const scm *__restrict__ fH, const scm *__restrict__ c,
                                                  uint32_t lgM0, uint32_t j, ge_ext *__restrict__ partial /* [2][gridDim.x] */, uint32_t quad) {
    __shared__ ge_ext lds[256];
    const uint32_t cls = blockIdx.y, tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M0 = 1u << lgM0, e = tid >> 3, g = tid & 7u;
    ge_ext acc = ge_identity();
    if (e < M0) {
        const bool isH = e >= (M0 >> 1);
        const uint32_t e2 = isH ? e - (M0 >> 1) : e;
        const uint32_t lgh = lgM0 - j - 1, h = 1u << lgh;
        const uint32_t t = e2 >> lgh, i = e2 & (h - 1);
        const bool hi = (cls == 0) != isH;                         // L: G_hi and H_lo;  R: G_lo and H_hi
        const uint32_t p = (t << (lgh + 1)) | (hi ? h : 0u) | i;
        const uint32_t sidx = hi ? i : (h | i);                    // the scalar of the opposite half
        scm s = isH ? b[sidx] : a[sidx];
        s = sc_mont_mul(s, isH ? fH[p] : fG[p]);
        s = sc_mont_mul(s, c[(isH ? M0 : 0u) + t]);
        uint32_t w[8]; tt_biased_words(w, s);
        const ge_pniels *tbl = table + ((size_t)(isH ? M0 : 0u) + p) * (TT_WINDOWS * TT_MULTS) + (size_t)g * 8 * TT_MULTS;
        const uint32_t word = w[g];
#pragma unroll 1
        for (uint32_t k = 0; k < 8; k++) {
            const int32_t d = (int32_t)((word >> (4 * k)) & 15u) - 8;
            if (d == 0) continue;
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            acc = ge_add_pniels_signed(acc, tbl[k * TT_MULTS + mag - 1], neg);
        }
    }
    ge_block_sum_store(acc, lds, quad, partial + cls * gridDim.x + blockIdx.x);
}
// Wide tables for a tail that starts on the ORIGINAL generators (circuits up to 2^14 multipliers freeze at round 0): those never change, so a
// table of k * 2^(8w) * P (w < 32 windows of 8 bits, k = 1..128, 512 KB per point, built once per device) halves the additions of every round:
// 32 per base point instead of 64, 4 per thread instead of 8.  Same layout of threads and partials as k_tt_round.
#define TT8_WINDOWS 32
#define TT8_MULTS 128
__global__ void __launch_bounds__(256) k_tt_bases8(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, ge_ext *__restrict__ bases, uint32_t M0) {
    __shared__ CoopLds L;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t count = 2 * M0;
    const uint32_t p = blockIdx.x * 64 + lane;
    const bool live = p < count;
    const uint32_t q_idx = live ? p : count - 1;
    if (wv == 0) {
        const ge_niels q = q_idx < M0 ? G[q_idx] : H[q_idx - M0];
        const ge_ext e = ge_madd(ge_identity(), q);
        coop_st(L.c[0], lane, e.X); coop_st(L.c[1], lane, e.Y); coop_st(L.c[2], lane, e.Z); coop_st(L.c[3], lane, e.T);
    }
    __syncthreads();
    fe *dst = reinterpret_cast<fe *>(bases + (size_t)q_idx * TT8_WINDOWS) + wv;
    for (uint32_t w = 0; w < TT8_WINDOWS; w++) {
        if (live) dst[4 * w] = coop_ld(L.c[wv], lane);
        if (w + 1 < TT8_WINDOWS) for (int k = 0; k < 8; k++) coop_dbl(L, wv, lane);
    }
}
// table[i * 128 + k] = (k + 1) * bases[i], i = p * 32 + w
__global__ void __launch_bounds__(256) k_tt_multiples8(const ge_ext *__restrict__ bases, ge_pniels *__restrict__ table, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ge_pniels *dst = table + (size_t)i * TT8_MULTS;
    ge_ext cur = bases[i];
    const ge_pniels n1 = ge_to_pniels(cur);
    dst[0] = n1;
#pragma unroll 1
    for (uint32_t k = 1; k < TT8_MULTS; k++) { cur = ge_add_pniels_signed(cur, n1, 0); dst[k] = ge_to_pniels(cur); }
}
// signed 8-bit digits without a carry chain: byte w of (s + 0x80..80) minus 128 lies in [-128, 127]
__device__ __forceinline__ void tt8_biased_words(uint32_t w[8], const scm &s) {
    sc_to_words(w, s);
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t t = (uint64_t)w[k] + 0x80808080ull + carry; w[k] = (uint32_t)t; carry = t >> 32; }
}
__global__ void __launch_bounds__(256) k_tt_round8(const ge_pniels *__restrict__ table, const scm *__restrict__ a, const scm *__restrict__ b,
                                                   const scm *__restrict__ fG, const scm *__restrict__ fH, const scm *__restrict__ c,
                                                   uint32_t lgM0, uint32_t j, ge_ext *__restrict__ partial /* [2][gridDim.x] */, uint32_t quad) {
    __shared__ ge_ext lds[256];
    const uint32_t cls = blockIdx.y, tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M0 = 1u << lgM0, e = tid >> 3, g = tid & 7u;
    ge_ext acc = ge_identity();
    if (e < M0) {
        const bool isH = e >= (M0 >> 1);
        const uint32_t e2 = isH ? e - (M0 >> 1) : e;
        const uint32_t lgh = lgM0 - j - 1, h = 1u << lgh;
        const uint32_t t = e2 >> lgh, i = e2 & (h - 1);
        const bool hi = (cls == 0) != isH;                         // L: G_hi and H_lo;  R: G_lo and H_hi
        const uint32_t p = (t << (lgh + 1)) | (hi ? h : 0u) | i;
        const uint32_t sidx = hi ? i : (h | i);
        scm s = isH ? b[sidx] : a[sidx];
        s = sc_mont_mul(s, isH ? fH[p] : fG[p]);
        s = sc_mont_mul(s, c[(isH ? M0 : 0u) + t]);
        uint32_t w[8]; tt8_biased_words(w, s);
        const ge_pniels *tbl = table + ((size_t)(isH ? M0 : 0u) + p) * (TT8_WINDOWS * TT8_MULTS) + (size_t)g * 4 * TT8_MULTS;
        const uint32_t word = w[g];
#pragma unroll 1
        for (uint32_t k = 0; k < 4; k++) {
            const int32_t d = (int32_t)((word >> (8 * k)) & 255u) - 128;
            if (d == 0) continue;
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            acc = ge_add_pniels_signed(acc, tbl[k * TT8_MULTS + mag - 1], neg);
        }
    }
    ge_block_sum_store(acc, lds, quad, partial + cls * gridDim.x + blockIdx.x);
}
// A_I, A_O, S of a circuit whose generators already have window tables (N <= 2^14: the tables of the frozen IPA tail are those of
// the original generators and live with the context): blockIdx.y = 0: <a_L,G> + <a_R,H>, 1: <a_O,G>, 2: <s_L,G> + <s_R,H>; same thread
// layout as k_tt_round (8 threads per base point, 8 windows each), block partials to partial[3][gridDim.x].
__global__ void __launch_bounds__(256) k_tt_commit3(const ge_pniels *__restrict__ table, const scm *__restrict__ aL, const scm *__restrict__ aR,
                                                    const scm *__restrict__ aO, const scm *__restrict__ sL, const scm *__restrict__ sR,
                                                    uint32_t n, uint32_t M0, ge_ext *__restrict__ partial, uint32_t quad) {
    __shared__ ge_ext lds[256];
    const uint32_t cls = blockIdx.y, tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = tid >> 3, g = tid & 7u;
    ge_ext acc = ge_identity();
    const bool isH = e >= M0;
    const uint32_t p = isH ? e - M0 : e;
    if (e < 2 * M0 && p < n && !(cls == 1 && isH)) {
        const scm s = cls == 0 ? (isH ? aR[p] : aL[p]) : (cls == 1 ? aO[p] : (isH ? sR[p] : sL[p]));
        uint32_t w[8]; tt_biased_words(w, s);
        const ge_pniels *tbl = table + ((size_t)(isH ? M0 : 0u) + p) * (TT_WINDOWS * TT_MULTS) + (size_t)g * 8 * TT_MULTS;
        const uint32_t word = w[g];
#pragma unroll 1
        for (uint32_t k = 0; k < 8; k++) {
            const int32_t d = (int32_t)((word >> (4 * k)) & 15u) - 8;
            if (d == 0) continue;
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            acc = ge_add_pniels_signed(acc, tbl[k * TT_MULTS + mag - 1], neg);
        }
    }
    ge_block_sum_store(acc, lds, quad, partial + cls * gridDim.x + blockIdx.x);
}
// block k: out[k] = sum of partial[k][0..nblk) + blind[k] * (fixed base whose window table is tableX)
__global__ void __launch_bounds__(256) k_tt_commit3_finish(const ge_ext *__restrict__ partial, uint32_t nblk, const scm *__restrict__ blind,
                                                           const ge_pniels *__restrict__ tableX, ge_ext *__restrict__ out, uint32_t quad) {
    __shared__ ge_ext lds[256];
    const uint32_t cls = blockIdx.x;
    ge_ext acc = ge_identity();                              // as in k_tt_finish: first point as it is, the blinding term on the last threads
    uint32_t s0 = threadIdx.x;
    if (s0 < nblk) { acc = partial[cls * nblk + s0]; s0 += 256; }
    for (; s0 < nblk; s0 += 256) acc = ge_add(acc, partial[cls * nblk + s0]);
    const uint32_t win = 255u - threadIdx.x;
    if (win < TT_WINDOWS) {
        uint32_t w[8]; tt_biased_words(w, blind[cls]);
        const int32_t d = (int32_t)((w[win >> 3] >> (4 * (win & 7u))) & 15u) - 8;
        if (d != 0) {
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            const ge_pniels q = tableX[win * TT_MULTS + mag - 1];
            acc = threadIdx.x >= nblk ? ge_from_pniels_signed(q, neg) : ge_add_pniels_signed(acc, q, neg);
        }
    }
    ge_block_sum_store(acc, lds, quad, out + cls);
}
// block 0 -> L, block 1 -> R: sum the block partials, add (c * w) * B with c = <a_lo, b_hi> resp. <a_hi, b_lo>; the point goes to the host, which encodes it
__global__ void __launch_bounds__(256) k_tt_finish(const ge_ext *__restrict__ partial, uint32_t nblk, const scm *__restrict__ a, const scm *__restrict__ b,
                                                   uint32_t h, scm wq, const ge_pniels *__restrict__ tableB, ge_ext *__restrict__ out, uint32_t quad) {
    __shared__ ge_ext lds[256];
    __shared__ scm slds[256];
    const uint32_t cls = blockIdx.x;
    scm ip = sc_zero();
    for (uint32_t i = threadIdx.x; i < h; i += 256) ip = sc_add(ip, cls == 0 ? sc_mont_mul(a[i], b[h + i]) : sc_mont_mul(a[h + i], b[i]));
    const scm cw = sc_mont_mul(block_sum_256(ip, slds), wq);
    // a thread's first point is taken as it is (no addition to the identity), and the 64 windows of (c w) B go to the LAST threads of the block, which hold no
    // partial when nblk <= 192: two dependent additions less ahead of the tree
    ge_ext acc = ge_identity();
    uint32_t s0 = threadIdx.x;
    if (s0 < nblk) { acc = partial[cls * nblk + s0]; s0 += 256; }
    for (; s0 < nblk; s0 += 256) acc = ge_add(acc, partial[cls * nblk + s0]);
    const uint32_t win = 255u - threadIdx.x;
    if (win < TT_WINDOWS) {
        uint32_t w[8]; tt_biased_words(w, cw);
        const int32_t d = (int32_t)((w[win >> 3] >> (4 * (win & 7u))) & 15u) - 8;
        if (d != 0) {
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            const ge_pniels q = tableB[win * TT_MULTS + mag - 1];
            acc = threadIdx.x >= nblk ? ge_from_pniels_signed(q, neg) : ge_add_pniels_signed(acc, q, neg);
        }
    }
    ge_block_sum_store(acc, lds, quad, out + cls);
}

// Pedersen commitments v*B + r*B_blinding from the window tables of the two fixed bases (k_tt_bases / k_tt_multiples on
// {B, B_blinding} at context creation): one wave per commitment, lane w adds the two table entries of 4-bit window w, the
// 64 partial sums meet in an LDS tree and lane 0 compresses - 8 dependent point additions instead of 255 doublings.
// v, r are plain 256-bit little-endian integers below 2^255 (v may be an unreduced Scalar::from_bits value), so the signed
// recoding (digit in [-8, 8], carry into the next window) never carries out of window 63.
__device__ __forceinline__ int32_t ped_digit(const uint32_t w[8], uint32_t win) {
    uint32_t carry = 0; int32_t d = 0;
    for (uint32_t j = 0; j <= win; j++) {
        const uint32_t nib = ((w[j >> 3] >> (4 * (j & 7u))) & 15u) + carry;
        carry = nib > 8u; d = (int32_t)nib - (carry ? 16 : 0);
    }
    return d;
}
__global__ void __launch_bounds__(64) k_pedersen(const uint32_t *__restrict__ v, const uint32_t *__restrict__ r,
                                                 const ge_pniels *__restrict__ table /* [2][64][8] */, uint8_t *__restrict__ out, uint32_t count) {
    __shared__ ge_ext lds[64];
    const uint32_t i = blockIdx.x, win = threadIdx.x;
    if (i >= count) return;
    uint32_t vw[8], rw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { vw[k] = v[8 * (size_t)i + k]; rw[k] = r[8 * (size_t)i + k]; }
    ge_ext acc = ge_identity();
    const int32_t dv = ped_digit(vw, win), dr = ped_digit(rw, win);
    if (dv != 0) acc = ge_add_pniels_signed(acc, table[(size_t)win * TT_MULTS + (dv < 0 ? -dv : dv) - 1], dv < 0);
    if (dr != 0) acc = ge_add_pniels_signed(acc, table[(size_t)(TT_WINDOWS + win) * TT_MULTS + (dr < 0 ? -dr : dr) - 1], dr < 0);
    lds[win] = acc; __syncthreads();
    for (uint32_t d = 32; d > 0; d >>= 1) {
        if (win < d) lds[win] = ge_add(lds[win], lds[win + d]);
        __syncthreads();
    }
    if (win == 0) ge_compress(out + 32 * (size_t)i, lds[0]);
}

// Generator fold of one group of r rounds: out[i] = tab[i] + sum_{t=1}^{2^r - 1} s_t * tab[i + t*Mr], i < Mr (Straus: one
// shared chain of doublings, all scalars wave-uniform, so the add/skip branch never diverges).  Threads [0,Mr) fold G,
// threads [Mr,2Mr) fold H.  Class B scalars (s_t * u_ch) apply to the padding generators p = i + t*Mr >= n of the first
// group.  naf holds, per (class, t), the non-adjacent form as two 256-bit masks (nz, neg): [4][nterms][16] words, class =
// 2*isH + isB.  A wave whose lanes agree on the class of every term takes the scalar path (s_cbranch on the digit: the
// addition is skipped, not masked); the few waves that straddle a class boundary take the per-lane path.
struct FoldGroup { uint32_t Mr, nterms, first_group, n; int32_t top; };
__global__ void __launch_bounds__(256) k_fold_points(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H,
                                                     ge_ext *__restrict__ out /* 2*Mr */, const uint32_t *__restrict__ naf, const FoldGroup fg) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < 2 * fg.Mr;
    if (!live) t = 2 * fg.Mr - 1;                           // keep whole waves converged; the store is guarded
    const bool isH = t >= fg.Mr;
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
    // bit q of bmask: term q+1 is a padding generator for this lane
    uint32_t bmask = 0;
    if (fg.first_group) for (uint32_t q = 0; q < fg.nterms; q++) if (i + (q + 1) * fg.Mr >= fg.n) bmask |= 1u << q;
    const uint32_t key = (isH ? 0x80000000u : 0u) | bmask;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    ge_ext acc = ge_identity();
    if (__ballot(key != key0) == 0ull) {
        const uint32_t hsel = (key0 >> 31) * 2u;
        for (int k = fg.top; k >= 0; k--) {
            acc = ge_dbl(acc);                              // doubling the identity above the top digit is harmless
            for (uint32_t q = 0; q < fg.nterms; q++) {
                const uint32_t *d = naf + ((size_t)(hsel + ((key0 >> q) & 1u)) * fg.nterms + q) * 16;     // scalar loads
                if ((d[k >> 5] >> (k & 31)) & 1u) acc = ge_madd_signed(acc, tab[i + (q + 1) * fg.Mr], (d[8 + (k >> 5)] >> (k & 31)) & 1u);
            }
        }
    } else {
        const uint32_t hsel = isH ? 2u : 0u;
        for (int k = fg.top; k >= 0; k--) {
            acc = ge_dbl(acc);
            for (uint32_t q = 0; q < fg.nterms; q++) {
                const uint32_t *d = naf + ((size_t)(hsel + ((bmask >> q) & 1u)) * fg.nterms + q) * 16;
                if ((d[k >> 5] >> (k & 31)) & 1u) acc = ge_madd_signed(acc, tab[i + (q + 1) * fg.Mr], (d[8 + (k >> 5)] >> (k & 31)) & 1u);
            }
        }
    }
    if (live) out[t] = ge_madd(acc, tab[i]);
}

// Same fold with the 2^r - 1 addends of every lane held in registers (NT * 24 VGPRs: 168 for r = 3, which leaves one wave per
// SIMD; the kernel is a single dependent chain per lane anyway): each table point is read from memory exactly once instead
// of once per non-zero NAF digit (~84 times), which was 165x the algorithmic traffic out of the Infinity Cache.
// (addends are 15 named variables, not an array: the compiler keeps an indexed local array in scratch memory)
#define BPG_FOLD_VARS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_fold_points_reg(
        const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, ge_ext *__restrict__ out /* 2*Mr */, const uint32_t *__restrict__ naf, const FoldGroup fg) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < 2 * fg.Mr;
    if (!live) t = 2 * fg.Mr - 1;
    const bool isH = t >= fg.Mr;
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
#define BPG_FOLD_LOAD(j) ge_niels p##j; if (j <= NT) p##j = tab[i + (size_t)j * fg.Mr];
    BPG_FOLD_VARS(BPG_FOLD_LOAD)
#undef BPG_FOLD_LOAD
    uint32_t bmask = 0;
    if (fg.first_group) for (uint32_t q = 0; q < (uint32_t)NT; q++) if (i + (q + 1) * fg.Mr >= fg.n) bmask |= 1u << q;
    const uint32_t key = (isH ? 0x80000000u : 0u) | bmask;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    const bool uniform = __ballot(key != key0) == 0ull;
    const uint32_t hsel = isH ? 2u : 0u;
    ge_ext acc = ge_identity();
    for (int k = fg.top; k >= 0; k--) {
        acc = ge_dbl(acc);
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)NT; q++) {
            // wave-uniform digit (scalar loads, s_cbranch) when every lane agrees on the class of every term
            const uint32_t cls = uniform ? (uint32_t)__builtin_amdgcn_readfirstlane(hsel + ((bmask >> q) & 1u)) : hsel + ((bmask >> q) & 1u);
            const uint32_t *d = naf + ((size_t)cls * NT + q) * 16;
            const uint32_t nz = (d[k >> 5] >> (k & 31)) & 1u, ng = (d[8 + (k >> 5)] >> (k & 31)) & 1u;
            if (uniform ? (__builtin_amdgcn_readfirstlane(nz) != 0) : (nz != 0)) {
                ge_niels Q = p1;
                switch (q) {                                   // q is wave-uniform: scalar branches, 24 moves
#define BPG_FOLD_PICK(j) case j - 1: if (j <= NT) Q = p##j; break;
                    BPG_FOLD_VARS(BPG_FOLD_PICK)
#undef BPG_FOLD_PICK
                    default: break;
                }
                acc = ge_madd_signed(acc, Q, ng);
            }
        }
    }
    if (live) out[t] = ge_madd(acc, tab[i]);
}

// Fold of a group whose tables are the ORIGINAL generators (the first group of every proof above the table tail): the generators never
// change, so their odd multiples odd[m-1][p] = (2m+1) * P_p, m = 1 .. 2^(w-2) - 1, are built once per capacity (k_odd_start / k_odd_step,
// affine Niels) and the shared scalars are recoded in width-w NAF: one addition per w+1 bits instead of one per 3 (36 instead of 84 per
// term at w = 6), each addend read from memory when its digit comes up - lane i of a wave reads point i of the same table: coalesced, and
// no addend lives in registers.  dig: [4 classes][nterms][256] signed odd digits (int8, 0 = none), class = 2*isH + isB as in k_fold_points.
// Split scalars (parts > 1): the generators never change, so the odd multiples of 2^(j*L) * P_p are tabulated as well (j < parts, L = ceil(254 / parts))
// and a scalar s = sum_j s_j 2^(j*L) becomes `parts` short scalars on different tables: ONE chain of L doublings instead of 253 for the same
// number of additions.  Table (part j, multiple 2m+1) sits at odd[(j * NM + m - 1) * tab] (NM = 2^(w-2); j = 0, m = 0 is the generator table itself).
// dig: [4 classes][parts * nterms][256] with entry part * nterms + q holding the width-w NAF of part `part` of term q's scalar.
struct FoldWnaf { uint32_t Mr, nterms, first_group, n, cap; int32_t top; uint32_t parts, NM; };
__device__ __forceinline__ int32_t fold_wnaf_digit(const uint32_t *__restrict__ dig32, uint32_t cls, uint32_t nterms, uint32_t q, int k) {
    const uint32_t w = dig32[(cls * nterms + q) * 64u + ((uint32_t)k >> 2)];
    return (int32_t)(int8_t)(w >> (8u * ((uint32_t)k & 3u)));
}
__global__ void __launch_bounds__(256) k_fold_points_wnaf(const ge_niels *__restrict__ gens /* [G | H], 2*cap */, const ge_niels *__restrict__ odd /* [m-1][2*cap] */,
                                                          ge_ext *__restrict__ out /* 2*Mr */, const uint32_t *__restrict__ dig32, const FoldWnaf fg) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < 2 * fg.Mr;
    if (!live) t = 2 * fg.Mr - 1;                           // keep whole waves converged; the store is guarded
    const bool isH = t >= fg.Mr;
    const uint32_t i = isH ? t - fg.Mr : t;
    const uint32_t base = (isH ? fg.cap : 0u) + i;           // index of P_i in every multiple table
    const size_t tab = (size_t)2 * fg.cap;
    uint32_t bmask = 0;
    if (fg.first_group) for (uint32_t q = 0; q < fg.nterms; q++) if (i + (q + 1) * fg.Mr >= fg.n) bmask |= 1u << q;
    const uint32_t key = (isH ? 0x80000000u : 0u) | bmask;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    const uint32_t hsel = isH ? 2u : 0u;
    ge_ext acc = ge_identity();
    const uint32_t nq = fg.nterms * fg.parts;
    if (__ballot(key != key0) == 0ull) {                    // every lane agrees on the class of every term: scalar digit loads, scalar branches
        const uint32_t hs = (key0 >> 31) * 2u;
        for (int k = fg.top; k >= 0; k--) {
            acc = ge_dbl(acc);
            for (uint32_t part = 0, e = 0; part < fg.parts; part++) {
                for (uint32_t q = 0; q < fg.nterms; q++, e++) {
                    const int32_t d = __builtin_amdgcn_readfirstlane(fold_wnaf_digit(dig32, hs + ((key0 >> q) & 1u), nq, e, k));
                    if (d != 0) {
                        const uint32_t mag = (uint32_t)(d < 0 ? -d : d), idx = part * fg.NM + (mag >> 1);
                        const ge_niels *T = idx ? odd + (size_t)(idx - 1) * tab : gens;
                        const ge_niels Q = T[base + (q + 1) * fg.Mr];
                        // the digit is the wave's: a scalar branch on its sign instead of the two selects and the negation of ge_madd_signed (43 of ~1,480
                        // instructions per addition)
                        if (d < 0) acc = ge_msub(acc, Q); else acc = ge_madd(acc, Q);
                    }
                }
            }
        }
    } else {
        for (int k = fg.top; k >= 0; k--) {
            acc = ge_dbl(acc);
            for (uint32_t part = 0, e = 0; part < fg.parts; part++) {
                for (uint32_t q = 0; q < fg.nterms; q++, e++) {
                    const int32_t d = fold_wnaf_digit(dig32, hsel + ((bmask >> q) & 1u), nq, e, k);
                    if (d != 0) {
                        const uint32_t mag = (uint32_t)(d < 0 ? -d : d), idx = part * fg.NM + (mag >> 1);
                        const ge_niels *T = idx ? odd + (size_t)(idx - 1) * tab : gens;
                        acc = ge_madd_signed(acc, T[base + (q + 1) * fg.Mr], d < 0);
                    }
                }
            }
        }
    }
    if (live) out[t] = ge_madd(acc, gens[base]);
}
// odd multiples of the generators: cur = P, dbl = 2P; each step cur += dbl gives the next odd multiple (normalised by k_normalize_niels)
__global__ void __launch_bounds__(256) k_odd_start(const ge_niels *__restrict__ gens, ge_ext *__restrict__ cur, ge_ext *__restrict__ dbl, uint32_t count) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const ge_ext e = ge_madd(ge_identity(), gens[p]);
    const ge_ext d = ge_dbl(e);
    cur[p] = ge_add(e, d); dbl[p] = d;                       // 3P
}
// the same from extended points (the 2^(j*L) multiples of the generators): cur = 3Q, dbl = 2Q
__global__ void __launch_bounds__(256) k_odd_start_ext(const ge_ext *__restrict__ base, ge_ext *__restrict__ cur, ge_ext *__restrict__ dbl, uint32_t count) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const ge_ext e = base[p];
    const ge_ext d = ge_dbl(e);
    cur[p] = ge_add(e, d); dbl[p] = d;
}
// pts[p] = 2^times * src[p]   (src = Niels generators when from_niels, else pts itself)
__global__ void __launch_bounds__(256) k_dbl_times(const ge_niels *__restrict__ gens, ge_ext *__restrict__ pts, uint32_t count, uint32_t times, uint32_t from_niels) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    ge_ext e = from_niels ? ge_madd(ge_identity(), gens[p]) : pts[p];
    for (uint32_t k = 0; k < times; k++) e = ge_dbl(e);
    pts[p] = e;
}
__global__ void __launch_bounds__(256) k_odd_step(ge_ext *__restrict__ cur, const ge_ext *__restrict__ dbl, uint32_t count) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    cur[p] = ge_add(cur[p], dbl[p]);
}

// The fold of a small table (at most 64 K outputs) with FOUR LANES PER OUTPUT (k_points.cuh quad_*): 16 outputs per wave, every lane keeps
// one coordinate of each of the (at most seven) addends in registers, plain NAF digits.  The chain per output is 253 doublings + 84 additions
// per term as in k_fold_points_reg, but each step costs ~510 instructions on the critical wave instead of ~1,350, and 2 * Mr * 4 lanes fill
// two waves per SIMD where k_fold_points_split needs four waves per 64 outputs that each repeat the doublings.
#define BPG_QFOLD_VARS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
__global__ void __launch_bounds__(256) k_fold_points_quad(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, ge_ext *__restrict__ out /* 2*Mr */,
                                                          const uint32_t *__restrict__ naf, const FoldGroup fg) {
    const uint32_t r = threadIdx.x & 3u;
    uint32_t t = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = t < 2 * fg.Mr;
    if (!live) t = 2 * fg.Mr - 1;
    const bool isH = t >= fg.Mr;
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
    const uint32_t NT = fg.nterms;                           // <= 7
#define BPG_QFOLD_LOAD(j) fe q##j = fe_zero(); if (j <= NT) q##j = quad_load_niels(tab + i + (size_t)j * fg.Mr, r);
    BPG_QFOLD_VARS(BPG_QFOLD_LOAD)
#undef BPG_QFOLD_LOAD
    uint32_t bmask = 0;
    if (fg.first_group) for (uint32_t q = 0; q < NT; q++) if (i + (q + 1) * fg.Mr >= fg.n) bmask |= 1u << q;
    const uint32_t key = (isH ? 0x80000000u : 0u) | bmask;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    const bool uniform = __ballot(key != key0) == 0ull;
    const uint32_t hsel = isH ? 2u : 0u;
    fe c = fe_zero(); c.v[0] = (r == 1u || r == 2u) ? 1u : 0u;       // identity: (0, 1, 1, 0)
    for (int k = fg.top; k >= 0; k--) {
        c = quad_dbl(c, r);
#pragma unroll 1
        for (uint32_t q = 0; q < NT; q++) {
            const uint32_t cls = uniform ? (uint32_t)__builtin_amdgcn_readfirstlane(hsel + ((bmask >> q) & 1u)) : hsel + ((bmask >> q) & 1u);
            const uint32_t *d = naf + ((size_t)cls * NT + q) * 16;
            const uint32_t nz = (d[k >> 5] >> (k & 31)) & 1u, ng = (d[8 + (k >> 5)] >> (k & 31)) & 1u;
            if (uniform ? (__builtin_amdgcn_readfirstlane(nz) != 0) : (nz != 0)) {
                fe Q = q1;
                switch (q) {                                   // q is wave-uniform
#define BPG_QFOLD_PICK(j) case j - 1: Q = q##j; break;
                    BPG_QFOLD_VARS(BPG_QFOLD_PICK)
#undef BPG_QFOLD_PICK
                    default: break;
                }
                c = quad_madd(c, Q, ng, r);
            }
        }
    }
    c = quad_madd(c, quad_load_niels(tab + i, r), 0u, r);
    if (live) reinterpret_cast<fe *>(out + t)[r] = c;        // ge_ext = {X, Y, Z, T}
}

// The same fold with width-4 NAF digits against the odd multiples 3P, 5P, 7P of every addend (round 5, second session): 51 additions per 253-bit scalar
// instead of 84 - 253 doublings + 357 additions + ~45 operations to make the multiples, against 253 + 588, on a chain that is the whole kernel.  The multiples are made
// by the quad itself (P -> 2P -> 3P -> 5P -> 7P), stored as projective operands of quad_madd in a scratch table [entry][output][coordinate] (lanes of a wave read and
// write consecutive addresses) and read back by the lanes that wrote them; 1P stays in registers.  The host turns the shared scalars into ONE list of steps per class
// (G, H): step = doublings before the addition | term << 8 | multiple << 11 | sign << 13; the operand of the next step is loaded while the current one is computed.
// Groups after the first only (no padding class); a block of 64 outputs lies in G or in H (Mr is a multiple of 64).
#define QW_MAXSTEPS 1024
struct FoldQuadW { uint32_t Mr, nterms, nsteps[2], tail[2]; };
__global__ void __launch_bounds__(256) k_fold_points_quadw(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, ge_ext *__restrict__ out /* 2*Mr */,
                                                           const uint32_t *__restrict__ steps /* [2][QW_MAXSTEPS] */, fe *tabq /* [3*nterms][2*Mr][4] */, const FoldQuadW fg) {
    const uint32_t r = threadIdx.x & 3u;
    const uint32_t t = blockIdx.x * 64 + (threadIdx.x >> 2);              // 2*Mr is a multiple of 64: every quad is live
    const uint32_t isH = (blockIdx.x * 64u >= fg.Mr) ? 1u : 0u;           // block-uniform
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
    const uint32_t NT = fg.nterms;                                        // <= 7
#define BPG_QFOLD_LOAD(j) fe q##j = fe_zero(); if (j <= NT) q##j = quad_load_niels(tab + i + (size_t)j * fg.Mr, r);
    BPG_QFOLD_VARS(BPG_QFOLD_LOAD)
#undef BPG_QFOLD_LOAD
    const size_t stride = (size_t)2 * fg.Mr * 4;                          // field elements per table entry
    fe *mine = tabq + (size_t)t * 4 + r;
#define BPG_QFOLD_BUILD(j) if (j <= NT) { \
        const fe P = quad_madd(quad_identity(r), q##j, 0u, r), D = quad_dbl(P, r); \
        fe M = quad_madd(D, q##j, 0u, r);  mine[(size_t)(3 * (j - 1) + 0) * stride] = quad_to_operand(M, r); \
        M = quad_add(M, D, r);             mine[(size_t)(3 * (j - 1) + 1) * stride] = quad_to_operand(M, r); \
        M = quad_add(M, D, r);             mine[(size_t)(3 * (j - 1) + 2) * stride] = quad_to_operand(M, r); }
    BPG_QFOLD_VARS(BPG_QFOLD_BUILD)
#undef BPG_QFOLD_BUILD
    const uint32_t *st = steps + isH * QW_MAXSTEPS;
    const uint32_t ns = fg.nsteps[isH];
    auto operand = [&](uint32_t s) -> fe {                                // s is the wave's
        const uint32_t q = (s >> 8) & 7u, m = (s >> 11) & 3u;
        if (m) return mine[(size_t)(3 * q + m - 1) * stride];
        fe Q = q1;
        switch (q) {
#define BPG_QFOLD_PICK(j) case j - 1: Q = q##j; break;
            BPG_QFOLD_VARS(BPG_QFOLD_PICK)
#undef BPG_QFOLD_PICK
            default: break;
        }
        return Q;
    };
    fe c = quad_identity(r);
    uint32_t snext = ns ? (uint32_t)__builtin_amdgcn_readfirstlane(st[0]) : 0u;
    fe nxt = ns ? operand(snext) : fe_zero();
    for (uint32_t k = 0; k < ns; k++) {
        const uint32_t s = snext;
        const fe op = nxt;
        if (k + 1 < ns) { snext = (uint32_t)__builtin_amdgcn_readfirstlane(st[k + 1]); nxt = operand(snext); }
        for (uint32_t d = s & 255u; d > 0; d--) c = quad_dbl(c, r);
        c = quad_madd(c, op, (s >> 13) & 1u, r);
    }
    for (uint32_t d = fg.tail[isH]; d > 0; d--) c = quad_dbl(c, r);
    c = quad_madd(c, quad_load_niels(tab + i, r), 0u, r);
    reinterpret_cast<fe *>(out + t)[r] = c;                               // ge_ext = {X, Y, Z, T}
}

// The same steps with ONE lane per output and every operand read from memory when its step comes up (the shape for a device that other proofs share: fewest
// instructions - 253 doublings + ~51 additions per term + 44 multiplications per term for its multiples, against 253 + 84 per term with the addends in
// registers: -15 % at seven terms).  Lane i of a wave reads entry e of output i: [entry][output] layout, consecutive lanes read consecutive points; 1P is the
// group-start table itself (halved affine Niels, 7M), 3P, 5P, 7P are projective Niels operands (8M).  The operand of the next step is loaded while the current one
// is added.  A block of 256 outputs lies in G or in H (Mr is a multiple of 256).
struct fold_operand { fe a, b, z, c; };                       // (ypx, ymx, Z, t2d) of a projective multiple; (ypx, ymx, -, t2d) of a table point (halved)
__global__ void __launch_bounds__(256) k_fold_points_regw(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H, ge_ext *__restrict__ out /* 2*Mr */,
                                                          const uint32_t *__restrict__ steps /* [2][QW_MAXSTEPS] */, ge_pniels *tabw /* [3*nterms][2*Mr] */, const FoldQuadW fg) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;                    // 2*Mr is a multiple of 256: every lane is live
    const uint32_t isH = (blockIdx.x * 256u >= fg.Mr) ? 1u : 0u;          // block-uniform
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
    const size_t stride = (size_t)2 * fg.Mr;
    ge_pniels *mine = tabw + t;
#pragma unroll 1
    for (uint32_t j = 0; j < fg.nterms; j++) {
        const ge_niels q = tab[i + (size_t)(j + 1) * fg.Mr];
        const ge_ext D = ge_dbl(ge_from_niels(q));
        ge_ext M = ge_madd(D, q);  mine[(size_t)(3 * j + 0) * stride] = ge_to_pniels(M);
        M = ge_add(M, D);          mine[(size_t)(3 * j + 1) * stride] = ge_to_pniels(M);
        M = ge_add(M, D);          mine[(size_t)(3 * j + 2) * stride] = ge_to_pniels(M);
    }
    const uint32_t *st = steps + isH * QW_MAXSTEPS;
    const uint32_t ns = fg.nsteps[isH];
    auto operand = [&](uint32_t s) -> fold_operand {                      // s is the wave's
        const uint32_t q = (s >> 8) & 7u, m = (s >> 11) & 3u;
        fold_operand o;
        if (m) { const ge_pniels p = mine[(size_t)(3 * q + m - 1) * stride]; o.a = p.ypx; o.b = p.ymx; o.z = p.Z; o.c = p.t2d; }
        else { const ge_niels p = tab[i + (size_t)(q + 1) * fg.Mr]; o.a = p.ypx; o.b = p.ymx; o.z = fe_one(); o.c = p.t2d; }
        return o;
    };
    ge_ext acc = ge_identity();
    uint32_t snext = ns ? (uint32_t)__builtin_amdgcn_readfirstlane(st[0]) : 0u;
    fold_operand nxt; nxt.a = nxt.b = nxt.z = nxt.c = fe_zero();
    if (ns) nxt = operand(snext);
    for (uint32_t k = 0; k < ns; k++) {
        const uint32_t s = snext;
        const fold_operand op = nxt;
        if (k + 1 < ns) { snext = (uint32_t)__builtin_amdgcn_readfirstlane(st[k + 1]); nxt = operand(snext); }
        for (uint32_t d = s & 255u; d > 0; d--) acc = ge_dbl(acc);
        const uint32_t neg = (s >> 13) & 1u;                              // the wave's: scalar branches, no selects
        if ((s >> 11) & 3u) {
            ge_pniels p; p.Z = op.z;
            if (neg) { p.ypx = op.b; p.ymx = op.a; p.t2d = fe_neg(op.c); } else { p.ypx = op.a; p.ymx = op.b; p.t2d = op.c; }
            acc = ge_add_pniels_signed(acc, p, 0u);
        } else {
            ge_niels p; p.ypx = op.a; p.ymx = op.b; p.t2d = op.c;
            if (neg) acc = ge_msub(acc, p); else acc = ge_madd(acc, p);
        }
    }
    for (uint32_t d = fg.tail[isH]; d > 0; d--) acc = ge_dbl(acc);
    out[t] = ge_madd(acc, tab[i]);
}

// Latency variant of the same fold for small tables (2*Mr <= 64 K outputs: one wave per SIMD at most, so the kernel is one
// dependent chain of 253 doublings + nterms * 84 additions whatever it does): the terms of an output are dealt to the FOUR
// waves of a block (term q goes to wave q mod 4), each wave runs its own chain of doublings over its <= 4 addends with
// wave-uniform digits, and the four partial sums meet in LDS.  4x the doublings, on hardware that would idle otherwise.
__global__ void __launch_bounds__(256) k_fold_points_split(const ge_niels *__restrict__ G, const ge_niels *__restrict__ H,
                                                           ge_ext *__restrict__ out /* 2*Mr */, const uint32_t *__restrict__ naf, const FoldGroup fg) {
    __shared__ ge_ext lds[256];
    const uint32_t sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint32_t t = blockIdx.x * 64 + lane;
    const bool live = t < 2 * fg.Mr;
    if (!live) t = 2 * fg.Mr - 1;
    const bool isH = t >= fg.Mr;
    const uint32_t i = isH ? t - fg.Mr : t;
    const ge_niels *tab = isH ? H : G;
    const uint32_t nslots = fg.nterms > sub ? (fg.nterms - sub + 3) / 4 : 0;          // terms q = sub + 4*j, j < nslots (<= 4)
    ge_niels p0 = ge_niels_identity(), p1 = p0, p2 = p0, p3 = p0;
    if (nslots > 0) p0 = tab[i + (size_t)(sub + 1) * fg.Mr];
    if (nslots > 1) p1 = tab[i + (size_t)(sub + 5) * fg.Mr];
    if (nslots > 2) p2 = tab[i + (size_t)(sub + 9) * fg.Mr];
    if (nslots > 3) p3 = tab[i + (size_t)(sub + 13) * fg.Mr];
    uint32_t bmask = 0;                                          // bit j: slot j is a padding generator for this lane
    if (fg.first_group) for (uint32_t j = 0; j < nslots; j++) if (i + (size_t)(sub + 4 * j + 1) * fg.Mr >= fg.n) bmask |= 1u << j;
    const uint32_t key = (isH ? 0x80000000u : 0u) | bmask;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    const bool uniform = __ballot(key != key0) == 0ull;
    const uint32_t hsel = isH ? 2u : 0u;
    ge_ext acc = ge_identity();
    if (nslots) for (int k = fg.top; k >= 0; k--) {
        acc = ge_dbl(acc);
#pragma unroll 1
        for (uint32_t j = 0; j < nslots; j++) {
            const uint32_t q = sub + 4 * j;
            const uint32_t cls = uniform ? (uint32_t)__builtin_amdgcn_readfirstlane(hsel + ((bmask >> j) & 1u)) : hsel + ((bmask >> j) & 1u);
            const uint32_t *d = naf + ((size_t)cls * fg.nterms + q) * 16;
            const uint32_t nz = (d[k >> 5] >> (k & 31)) & 1u, ng = (d[8 + (k >> 5)] >> (k & 31)) & 1u;
            if (uniform ? (__builtin_amdgcn_readfirstlane(nz) != 0) : (nz != 0)) {
                const ge_niels Q = j == 0 ? p0 : (j == 1 ? p1 : (j == 2 ? p2 : p3));
                acc = ge_madd_signed(acc, Q, ng);
            }
        }
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0 && live) {
        ge_ext r = ge_add(ge_add(lds[lane], lds[64 + lane]), ge_add(lds[128 + lane], lds[192 + lane]));
        out[t] = ge_madd(r, tab[i]);
    }
}

}  // namespace bpg
