// generator derivation, normalisation, compression - part of kernels.cuh (included from there, in this order; see its header for the kernel map and the data layout)
#pragma once

namespace bpg {

// ------------------------------------------------------------------------------------------------ four lanes per point
// A latency-bound chain of point operations (the generator fold of a small table: 253 dependent doublings per output) is shortened by giving
// a point to FOUR lanes: a doubling and an addition are two rounds of four independent field products; lane r of a quad (r = lane & 3)
// computes product r of each round and holds coordinate r of the running point (X, Y, Z, T); operands move inside the quad with DPP
// quad_perm (no LDS, no barrier).  About 510 instructions per doubling on the critical wave instead of 1,350.
template <int K> __device__ __forceinline__ fe quad_get(const fe &x) {          // every lane of a quad reads lane K's value
    fe r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.v[j], K * 0x55, 0xf, 0xf, true);
    return r;
}
__device__ __forceinline__ fe quad_swap01(const fe &x) {                         // lanes 0 and 1 of every quad exchange, 2 and 3 keep theirs
    fe r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.v[j], 0xE1, 0xf, 0xf, true);   // quad_perm:[1,0,2,3]
    return r;
}
__device__ __forceinline__ fe quad_pick(uint32_t r, const fe &v0, const fe &v1, const fe &v2, const fe &v3) {
    return fe_select(fe_select(v0, v1, r & 1u), fe_select(v2, v3, r & 1u), r >> 1);
}
// c = coordinate r of P  ->  coordinate r of 2P
__device__ __forceinline__ fe quad_dbl(const fe &c, uint32_t r) {
    const fe in = fe_select(c, fe_add(quad_get<0>(c), quad_get<1>(c)), r == 3u);      // X, Y, Z, X+Y
    const fe sq = fe_sq(in);
    const fe XX = quad_get<0>(sq), YY = quad_get<1>(sq), ZZ = quad_get<2>(sq), SS = quad_get<3>(sq);
    const fe YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX);
    const fe cX = fe_sub(SS, YpX), cT = fe_sub(fe_add(ZZ, ZZ), YmX);
    // X3 = cX * cT, Y3 = YpX * YmX, Z3 = YmX * cT, T3 = cX * YpX
    return fe_mul(quad_pick(r, cX, YpX, YmX, cX), quad_pick(r, cT, YmX, cT, YpX));
}
// mixed addition with a (halved, ge.cuh) affine Niels operand spread over the quad: qv = ((y - x)/2, (y + x)/2, dxy, 1)[r]; neg subtracts instead
// (the first two exchange, the third changes sign).  c = coordinate r of P -> coordinate r of P +- Q
__device__ __forceinline__ fe quad_madd(const fe &c, const fe &qv, uint32_t neg, uint32_t r) {
    fe qs = fe_select(qv, quad_swap01(qv), neg);
    qs = fe_select(qs, fe_neg(qs), neg & (uint32_t)(r == 2u));
    const fe X1 = quad_get<0>(c), Y1 = quad_get<1>(c);
    const fe lhs = quad_pick(r, fe_sub(Y1, X1), fe_add(Y1, X1), quad_get<3>(c), quad_get<2>(c));      // (Y1 - X1, Y1 + X1, T1, Z1)[r]
    const fe p = fe_mul(lhs, qs);                                                                   // A, B, C, D = Z1
    const fe A = quad_get<0>(p), B = quad_get<1>(p), C = quad_get<2>(p), D = quad_get<3>(p);
    const fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    // X3 = E * F, Y3 = G * H, Z3 = F * G, T3 = E * H
    return fe_mul(quad_pick(r, E, G, F, E), quad_pick(r, F, H, G, H));
}
// full addition (ge_add, 9M) of two points that are both spread over the quad: c, e = coordinate r of P, Q -> coordinate r of P + Q.  Three rounds of
// products: (A, B, T1 T2, Z1 Z2), then the constants (1, 1, 2d, 2) - C = 2d T1 T2 and D = 2 Z1 Z2 in one round - then the four output products.
// About 750 instructions on the wave against ~1,650 of ge_add.  Every lane of the quad must be active (DPP reads inside the quad).
__device__ __forceinline__ fe quad_add(const fe &c, const fe &e, uint32_t r) {
    const fe X1 = quad_get<0>(c), Y1 = quad_get<1>(c), X2 = quad_get<0>(e), Y2 = quad_get<1>(e);
    const fe lhs = quad_pick(r, fe_sub(Y1, X1), fe_add(Y1, X1), quad_get<3>(c), quad_get<2>(c));      // (Y1 - X1, Y1 + X1, T1, Z1)[r]
    const fe rhs = quad_pick(r, fe_sub(Y2, X2), fe_add(Y2, X2), quad_get<3>(e), quad_get<2>(e));
    fe k = fe_zero(); k.v[0] = r == 3u ? 2u : 1u;
    const fe p = fe_mul(fe_mul(lhs, rhs), fe_select(k, FE_D2(), r == 2u));
    const fe A = quad_get<0>(p), B = quad_get<1>(p), C = quad_get<2>(p), D = quad_get<3>(p);
    const fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    return fe_mul(quad_pick(r, E, G, F, E), quad_pick(r, F, H, G, H));
}
// coordinate r of P -> lane r's operand of quad_madd for the PROJECTIVE point P: ((Y - X)/2, (Y + X)/2, d T, Z)[r] (one round of products by constants);
// with it quad_madd computes D = Z1 Z2 where an affine operand gives D = Z1 - the same halved formulas, 8M
__device__ __forceinline__ fe quad_to_operand(const fe &c, uint32_t r) {
    const fe X = quad_get<0>(c), Y = quad_get<1>(c);
    const fe v = quad_pick(r, fe_sub(Y, X), fe_add(Y, X), quad_get<3>(c), quad_get<2>(c));
    return fe_mul(v, quad_pick(r, FE_INV2(), FE_INV2(), FE_D(), fe_one()));
}
__device__ __forceinline__ fe quad_identity(uint32_t r) { fe c = fe_zero(); c.v[0] = (r == 1u || r == 2u) ? 1u : 0u; return c; }      // (0, 1, 1, 0)[r]
__device__ __forceinline__ fe quad_load_ext(const ge_ext *p, uint32_t r) { return reinterpret_cast<const fe *>(p)[r]; }              // ge_ext = {X, Y, Z, T}
__device__ __forceinline__ fe fe_shfl_down(const fe &a, uint32_t d) { fe r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = __shfl_down(a.v[j], d, 64);
    return r; }
// Sum of the 256 points of a block, one per thread (the end of every table-driven kernel of the tail: rounds 1-4 ran a binary tree through LDS, eight dependent
// ge_add of 4.2 us each with most of the block idle).  The points go through LDS into quad layout - slot s = threads 4s .. 4s+3 takes points 4s .. 4s+3, three
// additions - then four shuffle levels inside each wave and two across the waves: nine quad additions of about half the length.  lds: 256 points; the block's
// threads all call it; coordinate r of the sum is returned in threads r = 0 .. 3.
__device__ __forceinline__ fe ge_block_sum_quad(const ge_ext &acc, ge_ext *lds) {
    lds[threadIdx.x] = acc;
    __syncthreads();
    const uint32_t r = threadIdx.x & 3u, slot = threadIdx.x >> 2, s16 = slot & 15u, wv = threadIdx.x >> 6;
    const fe *L = reinterpret_cast<const fe *>(lds);
    fe c = L[(4 * slot + 0) * 4 + r];
#pragma unroll 1
    for (uint32_t k = 1; k < 4; k++) c = quad_add(c, L[(4 * slot + k) * 4 + r], r);
    for (uint32_t d = 8; d > 0; d >>= 1) { const fe o = fe_shfl_down(c, 4u * d); c = quad_add(c, o, r); }
    __syncthreads();                                            // every slot has read its points: the first 16 field elements are reused
    fe *Wt = reinterpret_cast<fe *>(lds);
    if (s16 == 0) Wt[wv * 4 + r] = c;
    __syncthreads();
    if (wv == 0) {
        fe v = s16 < 4 ? Wt[s16 * 4 + r] : quad_identity(r);
        for (uint32_t d = 2; d > 0; d >>= 1) { const fe o = fe_shfl_down(v, 4u * d); v = quad_add(v, o, r); }
        c = v;
    }
    return c;
}
// the block's sum stored to *dst: in quad layout for a proof alone (the shorter chain), by the binary LDS tree on a shared device (9 waves of ge_add against 30 of
// quad_add: fewer instructions - 0.04 G per proof in the tail kernels); `quad` is the launch's
__device__ __forceinline__ void ge_block_sum_store(const ge_ext &acc, ge_ext *lds, uint32_t quad, ge_ext *dst) {
    if (quad) {
        const fe sum = ge_block_sum_quad(acc, lds);
        if (threadIdx.x < 4) reinterpret_cast<fe *>(dst)[threadIdx.x] = sum;
    } else {
        lds[threadIdx.x] = acc; __syncthreads();
        for (uint32_t d = 128; d > 0; d >>= 1) {
            if (threadIdx.x < d) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
            __syncthreads();
        }
        if (threadIdx.x == 0) *dst = lds[0];
    }
}
__device__ __forceinline__ fe quad_load_niels(const ge_niels *p, uint32_t r) {     // lane r's operand of quad_madd
    const fe *f = reinterpret_cast<const fe *>(p);                                 // ge_niels = {(y + x)/2, (y - x)/2, dxy}
    return r == 3u ? fe_one() : f[r == 0u ? 1 : (r == 1u ? 0 : 2)];
}

// ------------------------------------------------------------------------------------------------ generators
// one thread per generator: 64 uniform bytes -> Ristretto point (two Elligator maps + add), extended coordinates
__global__ void __launch_bounds__(256) k_gens_derive(const uint32_t *__restrict__ uniform, ge_ext *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[16];
    const uint4 *src = reinterpret_cast<const uint4 *>(uniform + 16 * (size_t)i);
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 q = src[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    out[i] = ge_from_uniform_words(w);
}

// extended -> affine Niels with one field inversion per NORM_K points (Montgomery's trick inside a thread).
// Thread t handles points t, t+T, t+2T, ... so that loads and stores stay coalesced.
#define NORM_K 8
__global__ void __launch_bounds__(256) k_normalize_niels(const ge_ext *__restrict__ in, ge_niels *__restrict__ out, uint32_t count) {
    uint32_t T = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    fe pre[NORM_K];
    fe acc = fe_one();
#pragma unroll
    for (int k = 0; k < NORM_K; k++) {
        uint32_t idx = t + k * T;
        pre[k] = acc;
        if (idx < count) acc = fe_mul(acc, in[idx].Z);
    }
    fe inv = fe_mul(fe_invert(acc), FE_INV2());                // 1 / (2 Z_1 .. Z_K): the halved Niels form wants 1 / (2 Z_k), and the 1/2 rides along for nothing
#pragma unroll
    for (int k = NORM_K - 1; k >= 0; k--) {
        uint32_t idx = t + k * T;
        if (idx < count) {
            ge_ext p = in[idx];
            fe zinv_half = fe_mul(inv, pre[k]);
            inv = fe_mul(inv, p.Z);                            // 1 / (2 Z_1 .. Z_(k-1))
            out[idx] = ge_to_niels_halfinv(p, zinv_half);
        }
    }
}

__global__ void __launch_bounds__(64) k_compress_niels(const ge_niels *__restrict__ in, uint8_t *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ge_compress(out + 32 * (size_t)i, ge_madd(ge_identity(), in[i]));
}

// bases[0] = B, bases[1] = B_blinding = from_uniform(SHA3-512(compress(B))) (hash computed on the host), bases[2] = B + B_blinding
__global__ void k_init_bases(const uint32_t *__restrict__ hash64, ge_niels *__restrict__ bases) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ge_ext B;
    B.X = BPG_FE(0x918de5d2u, 0x2f4183e0u, 0xa8a67c6cu, 0x40971ffau, 0x6803537au, 0xdd5bff85u, 0x8cfe80c3u, 0x1063e2ccu);
    B.Y = BPG_FE(0xf533ad9bu, 0xcc7edf80u, 0x4253df49u, 0x5d14c8bau, 0x0fc4ed5bu, 0x061b3d57u, 0xe44c3c7fu, 0x159a6849u);
    B.Z = fe_one(); B.T = fe_mul(B.X, B.Y);
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = hash64[i];
    ge_ext Bb = ge_from_uniform_words(w);
    ge_ext S = ge_add(B, Bb);
    bases[0] = ge_to_niels(B, fe_one());
    bases[1] = ge_to_niels(Bb, fe_invert(Bb.Z));
    bases[2] = ge_to_niels(S, fe_invert(S.Z));
}

// unit-test hook for the device field arithmetic (the inline-asm paths cannot be compiled for the host):
// op 0 mul, 1 sq, 2 add, 3 sub, 4 invert, 5 chain (mixed ops on weakly reduced intermediates); inputs are raw 256-bit values;
// op 6: the SCALAR Montgomery product sc_mont_mul(a, b) = a b / 2^256 mod l (sc.cuh device path), raw words out
__global__ void __launch_bounds__(64) k_test_fe(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint8_t *__restrict__ out, uint32_t n, uint32_t op) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x, y, r;
#pragma unroll
    for (int k = 0; k < 8; k++) { x.v[k] = a[8 * i + k]; y.v[k] = b[8 * i + k]; }
    if (op == 6) {
        scm sa, sb;
#pragma unroll
        for (int k = 0; k < 8; k++) { sa.v[k] = x.v[k]; sb.v[k] = y.v[k]; }
        const scm sr = sc_mont_mul(sa, sb);
        uint32_t *o = reinterpret_cast<uint32_t *>(out + 32 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = sr.v[k];
        return;
    }
    switch (op) {
    case 0: r = fe_mul(x, y); break;
    case 1: r = fe_sq(x); break;
    case 2: r = fe_add(x, y); break;
    case 3: r = fe_sub(x, y); break;
    case 4: r = fe_invert(x); break;
    default:
        for (int k = 0; k < 25; k++) { fe t = fe_sub(fe_mul(x, y), fe_add(x, y)); x = fe_sq(fe_sub(y, t)); y = fe_add(t, fe_neg(x)); }
        r = fe_add(x, y); break;
    }
    fe_tobytes(out + 32 * (size_t)i, r);
}

// integer-VALU roofline probe: 4 independent chains of field multiplications per thread, nothing but registers
__global__ void __launch_bounds__(256) k_bench_fe_mul(fe *__restrict__ out, uint32_t iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(), b = FE_SQRTM1(), c = FE_D2(), d = FE_ONE_MINUS_D_SQ();
    a.v[0] ^= t; b.v[1] ^= t; c.v[2] ^= t; d.v[3] ^= t;
    for (uint32_t i = 0; i < iters; i++) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
    out[t] = fe_add(fe_add(a, b), fe_add(c, d));
}

}  // namespace bpg
