// HIP kernels of the Bulletproofs R1CS prove path for gfx950 (MI355X).  Integer VALU work (v_mad_u64_u32 chains);
// no MFMA - this is 255-bit modular arithmetic, not a dense contraction.  Hot-path rows of SURVEY.md section 8(a):
//   a7  BulletproofGens::new            k_gens_derive + k_normalize_niels
//   a1-a3,a6  Pedersen commits          k_pedersen (window tables of B and B_blinding: k_tt_bases, k_tt_multiples)
//   a9  A_I, A_O, S multiscalar muls    k_msm_plain, k_msm_tile<0/1>, k_msm_tile_prefix, k_scan_*, k_bucket_chunks, k_bucket_combine(_heavy),
//                                       k_bucket_reduce, k_window_sums, k_msm_horner (bucket method: LDS tile histograms -> scan -> scatter ->
//                                       balanced bucket sweep -> reductions)
//   a10 vector-polynomial phase         k_exp_table, k_flatten, k_poly_t, k_poly_eval, k_reduce_partials
//   a11 inner-product argument          above 2^14 generators: k_ipa_prep, the MSM kernels, k_tt_advance, k_fold_points / k_fold_points_reg<NT> /
//                                       k_fold_points_split once per group of rounds; below: k_tt_bases, k_tt_multiples, k_tt_factors, then
//                                       k_tt_advance, k_tt_round, k_tt_finish per round
//   f1  Verifier::verify                k_decompress, k_flatten_const, k_ipa_s, k_verify_scalars + one MSM
// Data layout in HBM: scalars = 8 x u32 Montgomery form, 32 B each, AoS (lane i <-> element i: 2 x 16 B coalesced
// loads); generator tables = affine Niels (y+x, y-x, 2dxy), 96 B per point, G at [0,N) and H at [N,2N); window tables =
// projective Niels (y+x, y-x, z, 2dt), 128 B per entry.
#pragma once
#include <hip/hip_runtime.h>
#include "ge.cuh"
#include "sc.cuh"

namespace bpg {

#define BPG_MAX_SEGS 8
struct MsmSegs {
    const scm *sc[BPG_MAX_SEGS];
    const ge_niels *pts[BPG_MAX_SEGS];
    uint32_t len[BPG_MAX_SEGS];
    uint32_t start[BPG_MAX_SEGS + 1];   // prefix of len
    uint32_t msm[BPG_MAX_SEGS];
    uint32_t lgblk[BPG_MAX_SEGS];       // element e of the segment is point ((e >> lgblk) << (lgblk+1)) | (e & (2^lgblk - 1)): every
    uint32_t nseg;                      // other block of 2^lgblk points (grouped IPA rounds); 31 = contiguous
};
__device__ __forceinline__ uint32_t msm_point_index(const MsmSegs &S, uint32_t seg, uint32_t e) {
    const uint32_t lg = S.lgblk[seg];
    return ((e >> lg) << (lg + 1)) | (e & ((1u << lg) - 1u));
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
    fe inv = fe_invert(acc);
#pragma unroll
    for (int k = NORM_K - 1; k >= 0; k--) {
        uint32_t idx = t + k * T;
        if (idx < count) {
            ge_ext p = in[idx];
            fe zinv = fe_mul(inv, pre[k]);
            inv = fe_mul(inv, p.Z);
            out[idx] = ge_to_niels(p, zinv);
        }
    }
}

__global__ void __launch_bounds__(64) k_compress(const ge_ext *__restrict__ in, uint8_t *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ge_compress(out + 32 * (size_t)i, in[i]);
}
// out = in[0] + .. + in[count-1]  (a handful of partial MSM results)
__global__ void __launch_bounds__(64) k_sum_points(const ge_ext *__restrict__ in, uint32_t count, ge_ext *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ge_ext acc = count ? in[0] : ge_identity();
    for (uint32_t k = 1; k < count; k++) acc = ge_add(acc, in[k]);
    *out = acc;
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
// op 0 mul, 1 sq, 2 add, 3 sub, 4 invert, 5 chain (mixed ops on weakly reduced intermediates); inputs are raw 256-bit values
__global__ void __launch_bounds__(64) k_test_fe(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint8_t *__restrict__ out, uint32_t n, uint32_t op) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x, y, r;
#pragma unroll
    for (int k = 0; k < 8; k++) { x.v[k] = a[8 * i + k]; y.v[k] = b[8 * i + k]; }
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

// ------------------------------------------------------------------------------------------------ scalar vectors
__global__ void __launch_bounds__(256) k_sc_from_bytes(const uint32_t *__restrict__ in, scm *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[8];
    const uint4 *src = reinterpret_cast<const uint4 *>(in + 8 * (size_t)i);
    uint4 a = src[0], b = src[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    out[i] = sc_from_words(w);
}
// 64-byte TranscriptRng draws -> scalars (Scalar::random = from_bytes_mod_order_wide)
__global__ void __launch_bounds__(256) k_sc_from_wide(const uint32_t *__restrict__ in, scm *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[16];
    const uint4 *src = reinterpret_cast<const uint4 *>(in + 16 * (size_t)i);
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 q = src[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    out[i] = sc_from_wide_words(w);
}
// BPG_FLAG_EXPANDED_BLINDING (include/bpg.h): scalar j = SHAKE256("bpg blinding v1" || K || le64(j))[0..64) mod l, one Keccak-f[1600]
// per thread.  The 87 message bytes fill lanes 0..10: lanes 0..8 and the low 7 bytes of lane 9 are the same for every j (head[]).
__device__ __forceinline__ uint64_t kk_rol(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
struct BlindHead { uint64_t lane[10]; };
__global__ void __launch_bounds__(256) k_blind_expand(const BlindHead head, scm *__restrict__ out, uint32_t count) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    uint64_t a[25];
#pragma unroll
    for (int k = 0; k < 25; k++) a[k] = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) a[k] = head.lane[k];
    a[9] |= (uint64_t)(j & 0xffu) << 56;                      // byte 79 = low byte of le64(j)
    a[10] = (uint64_t)(j >> 8) | (0x1fULL << 56);             // bytes 80..86 = the rest of j, byte 87 = SHAKE padding
    a[16] = 0x80ULL << 56;                                    // byte 135 = end of the 136-byte rate
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL,
        0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL,
        0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL,
        0x8000000080008008ULL};
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20], c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21], c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22],
                 c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23], c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        const uint64_t d0 = c4 ^ kk_rol(c1, 1), d1 = c0 ^ kk_rol(c2, 1), d2 = c1 ^ kk_rol(c3, 1), d3 = c2 ^ kk_rol(c4, 1), d4 = c3 ^ kk_rol(c0, 1);
        const uint64_t b00 = a[0] ^ d0, b01 = kk_rol(a[6] ^ d1, 44), b02 = kk_rol(a[12] ^ d2, 43), b03 = kk_rol(a[18] ^ d3, 21), b04 = kk_rol(a[24] ^ d4, 14);
        const uint64_t b05 = kk_rol(a[3] ^ d3, 28), b06 = kk_rol(a[9] ^ d4, 20), b07 = kk_rol(a[10] ^ d0, 3), b08 = kk_rol(a[16] ^ d1, 45), b09 = kk_rol(a[22] ^ d2, 61);
        const uint64_t b10 = kk_rol(a[1] ^ d1, 1), b11 = kk_rol(a[7] ^ d2, 6), b12 = kk_rol(a[13] ^ d3, 25), b13 = kk_rol(a[19] ^ d4, 8), b14 = kk_rol(a[20] ^ d0, 18);
        const uint64_t b15 = kk_rol(a[4] ^ d4, 27), b16 = kk_rol(a[5] ^ d0, 36), b17 = kk_rol(a[11] ^ d1, 10), b18 = kk_rol(a[17] ^ d2, 15), b19 = kk_rol(a[23] ^ d3, 56);
        const uint64_t b20 = kk_rol(a[2] ^ d2, 62), b21 = kk_rol(a[8] ^ d3, 55), b22 = kk_rol(a[14] ^ d4, 39), b23 = kk_rol(a[15] ^ d0, 41), b24 = kk_rol(a[21] ^ d1, 2);
        a[0] = b00 ^ (~b01 & b02) ^ RC[r]; a[1] = b01 ^ (~b02 & b03); a[2] = b02 ^ (~b03 & b04); a[3] = b03 ^ (~b04 & b00); a[4] = b04 ^ (~b00 & b01);
        a[5] = b05 ^ (~b06 & b07); a[6] = b06 ^ (~b07 & b08); a[7] = b07 ^ (~b08 & b09); a[8] = b08 ^ (~b09 & b05); a[9] = b09 ^ (~b05 & b06);
        a[10] = b10 ^ (~b11 & b12); a[11] = b11 ^ (~b12 & b13); a[12] = b12 ^ (~b13 & b14); a[13] = b13 ^ (~b14 & b10); a[14] = b14 ^ (~b10 & b11);
        a[15] = b15 ^ (~b16 & b17); a[16] = b16 ^ (~b17 & b18); a[17] = b17 ^ (~b18 & b19); a[18] = b18 ^ (~b19 & b15); a[19] = b19 ^ (~b15 & b16);
        a[20] = b20 ^ (~b21 & b22); a[21] = b21 ^ (~b22 & b23); a[22] = b22 ^ (~b23 & b24); a[23] = b23 ^ (~b24 & b20); a[24] = b24 ^ (~b20 & b21);
    }
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { w[2 * k] = (uint32_t)a[k]; w[2 * k + 1] = (uint32_t)(a[k] >> 32); }
    out[j] = sc_from_wide_words(w);
}
__global__ void __launch_bounds__(256) k_sc_to_bytes(const scm *__restrict__ in, uint32_t *__restrict__ out, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[8]; sc_to_words(w, in[i]);
#pragma unroll
    for (int k = 0; k < 8; k++) out[8 * (size_t)i + k] = w[k];
}

// out[i] = base^i for i < count (Montgomery form). Thread t walks i = t, t+T, ... multiplying by base^T; T = 2^lgT.
__global__ void __launch_bounds__(256) k_exp_table(scm base, scm *__restrict__ out, uint32_t count, uint32_t lgT) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t T = 1u << lgT;
    if (t >= T) return;
    scm cur = SC_R1(), sq = base;
    for (uint32_t b = 0; b < lgT; b++) {           // cur = base^t ; sq ends as base^T
        if ((t >> b) & 1u) cur = sc_mont_mul(cur, sq);
        sq = sc_mont_mul(sq, sq);
    }
    for (uint32_t i = t; i < count; i += T) { out[i] = cur; cur = sc_mont_mul(cur, sq); }
}

// block-wide sum of one scalar per thread (256 threads) through LDS; result valid in thread 0
__device__ __forceinline__ scm block_sum_256(scm v, scm *lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) lds[threadIdx.x] = sc_add(lds[threadIdx.x], lds[threadIdx.x + s]);
        __syncthreads();
    }
    scm r = lds[0];
    __syncthreads();
    return r;
}

// out[k] = sum over parts of partial[p * stride + k], k < nsum   (one block per output scalar)
__global__ void __launch_bounds__(256) k_reduce_partials(const scm *__restrict__ partial, uint32_t parts, uint32_t stride, scm *__restrict__ out) {
    __shared__ scm lds[256];
    uint32_t k = blockIdx.x;
    scm acc = sc_zero();
    for (uint32_t p = threadIdx.x; p < parts; p += 256) acc = sc_add(acc, partial[(size_t)p * stride + k]);
    scm r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[k] = r;
}

// ------------------------------------------------------------------------------------------------ circuit upload: CSR -> CSC on the device
// The caller's constraint list is row-major (one row per constraint); k_flatten wants it column-major (one column per variable).
// Column of a term: left / right / output multiplier i -> i, n+i, 2n+i; committed j -> 3n+j; the constant terms (Variable::One) form
// the last column 3n+m, which can hold O(q) entries: it is laid out by a scan over the rows, never through one hot atomic.
__device__ __forceinline__ uint32_t csc_col(uint32_t pv, uint32_t n, uint32_t m) {
    const uint32_t kind = pv >> 29, idx = pv & 0x1fffffffu;
    return kind <= 2 ? kind * n + idx : (kind == 3 ? 3 * n + idx : 3 * n + m);
}
__global__ void __launch_bounds__(256) k_csc_count(const uint64_t *__restrict__ row_ptr, const uint32_t *__restrict__ term_var, uint32_t q, uint32_t n, uint32_t m,
                                                   uint32_t *__restrict__ counts /* 3n+m, zeroed */, uint32_t *__restrict__ rowconst /* q */) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= q) return;
    uint32_t rc = 0;
    for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
        const uint32_t col = csc_col(term_var[k], n, m);
        if (col == 3 * n + m) rc++; else atomicAdd(&counts[col], 1u);
    }
    rowconst[r] = rc;
}
// cursor[] = running positions of the variable columns (k_scan_apply), rowconst_start[] = exclusive scan of rowconst
__global__ void __launch_bounds__(256) k_csc_fill(const uint64_t *__restrict__ row_ptr, const uint32_t *__restrict__ term_var, const uint32_t *__restrict__ term_coef,
                                                  uint32_t q, uint32_t n, uint32_t m, uint32_t *__restrict__ cursor, const uint32_t *__restrict__ rowconst_start,
                                                  const uint32_t *__restrict__ var_total /* starts[3n+m] */, uint32_t *__restrict__ ent_row, uint32_t *__restrict__ ent_coef) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= q) return;
    uint32_t cpos = *var_total + rowconst_start[r];
    for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
        const uint32_t col = csc_col(term_var[k], n, m);
        const uint32_t pos = col == 3 * n + m ? cpos++ : atomicAdd(&cursor[col], 1u);
        ent_row[pos] = r; ent_coef[pos] = term_coef[k];
    }
}
// col_ptr (64-bit, 3n+m+2 entries) from the two scans: variable columns, then the constant column
__global__ void __launch_bounds__(256) k_csc_colptr(const uint32_t *__restrict__ starts /* 3n+m+1 */, const uint32_t *__restrict__ rowconst_start /* q+1 */,
                                                    uint32_t nvar, uint32_t q, uint64_t *__restrict__ col_ptr, uint32_t *__restrict__ totals /* [0] var, [1] all */) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= nvar) col_ptr[c] = starts[c];                      // col_ptr[nvar] = first entry of the constant column
    if (c == 0) { const uint32_t all = starts[nvar] + rowconst_start[q]; col_ptr[nvar + 1] = all; totals[0] = starts[nvar]; totals[1] = all; }
}

// flattened_constraints(z): column-major gather. Column c (0..3n+m): w[c] = sum_e coef[ent_coef[e]] * z^(ent_row[e]+1);
// columns [3n, 3n+m) are the committed variables and come out negated (wV).
__global__ void __launch_bounds__(256) k_flatten(const uint64_t *__restrict__ col_ptr, const uint32_t *__restrict__ ent_row,
                                                 const uint32_t *__restrict__ ent_coef, const scm *__restrict__ coef,
                                                 const scm *__restrict__ zpow /* z^j, j >= 0 */, scm *__restrict__ w,
                                                 uint32_t ncols, uint32_t first_neg_col) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    scm acc = sc_zero();
    for (uint64_t e = col_ptr[c]; e < col_ptr[c + 1]; e++) acc = sc_add(acc, sc_mont_mul(coef[ent_coef[e]], zpow[ent_row[e] + 1]));
    w[c] = (c >= first_neg_col) ? sc_neg(acc) : acc;
}

// w_c = - sum over the constant terms (Variable::One) of coef * z^(row+1): the one column of the constraint matrix that holds
// O(q) entries, so it gets a grid-wide reduction instead of a k_flatten thread (verifier only; the prover never needs it)
__global__ void __launch_bounds__(256) k_flatten_const(const uint32_t *__restrict__ ent_row, const uint32_t *__restrict__ ent_coef,
                                                       const scm *__restrict__ coef, const scm *__restrict__ zpow, uint64_t e0, uint64_t e1,
                                                       scm *__restrict__ partial) {
    __shared__ scm lds[256];
    scm acc = sc_zero();
    for (uint64_t e = e0 + blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (uint64_t)gridDim.x * blockDim.x)
        acc = sc_add(acc, sc_mont_mul(coef[ent_coef[e]], zpow[ent_row[e] + 1]));
    scm r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = sc_neg(r);
}

// t1..t6 partial sums of <l(X), r(X)>:  l1 = aL + y^-i wR, l2 = aO, l3 = sL ; r0 = wO - y^i, r1 = y^i aR + wL, r3 = y^i sR
__global__ void __launch_bounds__(256) k_poly_t(const scm *__restrict__ aL, const scm *__restrict__ aR, const scm *__restrict__ aO,
                                                const scm *__restrict__ sL, const scm *__restrict__ sR,
                                                const scm *__restrict__ wL, const scm *__restrict__ wR, const scm *__restrict__ wO,
                                                const scm *__restrict__ ypow, const scm *__restrict__ yinvpow,
                                                scm *__restrict__ partial /* gridDim.x * 6 */, uint32_t n) {
    __shared__ scm lds[256];
    scm t1 = sc_zero(), t2 = t1, t3 = t1, t4 = t1, t5 = t1, t6 = t1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        scm y = ypow[i];
        scm l1 = sc_add(aL[i], sc_mont_mul(yinvpow[i], wR[i]));
        scm l2 = aO[i], l3 = sL[i];
        scm r0 = sc_sub(wO[i], y);
        scm r1 = sc_add(sc_mont_mul(y, aR[i]), wL[i]);
        scm r3 = sc_mont_mul(y, sR[i]);
        t1 = sc_add(t1, sc_mont_mul(l1, r0));
        t2 = sc_add(t2, sc_add(sc_mont_mul(l1, r1), sc_mont_mul(l2, r0)));
        t3 = sc_add(t3, sc_add(sc_mont_mul(l2, r1), sc_mont_mul(l3, r0)));
        t4 = sc_add(t4, sc_add(sc_mont_mul(l1, r3), sc_mont_mul(l3, r1)));
        t5 = sc_add(t5, sc_mont_mul(l2, r3));
        t6 = sc_add(t6, sc_mont_mul(l3, r3));
    }
    scm r;
    r = block_sum_256(t1, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 0] = r;
    r = block_sum_256(t2, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 1] = r;
    r = block_sum_256(t3, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 2] = r;
    r = block_sum_256(t4, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 3] = r;
    r = block_sum_256(t5, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 4] = r;
    r = block_sum_256(t6, lds); if (threadIdx.x == 0) partial[blockIdx.x * 6 + 5] = r;
}

// l(x) = x (l1 + x (l2 + x l3)),  r(x) = r0 + x (r1 + x^2 r3); padding i in [n, N): l = 0, r = -y^i
__global__ void __launch_bounds__(256) k_poly_eval(const scm *__restrict__ aL, const scm *__restrict__ aR, const scm *__restrict__ aO,
                                                   const scm *__restrict__ sL, const scm *__restrict__ sR,
                                                   const scm *__restrict__ wL, const scm *__restrict__ wR, const scm *__restrict__ wO,
                                                   const scm *__restrict__ ypow, const scm *__restrict__ yinvpow, scm x,
                                                   scm *__restrict__ lv, scm *__restrict__ rv, uint32_t n, uint32_t N) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    scm y = ypow[i];
    if (i >= n) { lv[i] = sc_zero(); rv[i] = sc_neg(y); return; }
    scm l1 = sc_add(aL[i], sc_mont_mul(yinvpow[i], wR[i]));
    scm r0 = sc_sub(wO[i], y);
    scm r1 = sc_add(sc_mont_mul(y, aR[i]), wL[i]);
    scm r3 = sc_mont_mul(y, sR[i]);
    scm l = sc_mont_mul(x, sc_add(l1, sc_mont_mul(x, sc_add(aO[i], sc_mont_mul(x, sL[i])))));
    scm r = sc_add(r0, sc_mont_mul(x, sc_add(r1, sc_mont_mul(x, sc_mont_mul(x, r3)))));
    lv[i] = l; rv[i] = r;
}

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
// c[k] *= w   (k < 2): the Q = w*B term of L and R becomes a scalar on the fixed base B
__global__ void k_scale2(scm *__restrict__ c, scm w) { if (threadIdx.x < 2 && blockIdx.x == 0) c[threadIdx.x] = sc_mont_mul(c[threadIdx.x], w); }

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
                                                  uint32_t lgM0, uint32_t j, ge_ext *__restrict__ partial /* [2][gridDim.x] */) {
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
    lds[threadIdx.x] = acc; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[cls * gridDim.x + blockIdx.x] = lds[0];
}
// A_I, A_O, S of a circuit whose generators already have window tables (N <= 2^14: the tables of the frozen IPA tail are those of
// the original generators and live with the context): blockIdx.y = 0: <a_L,G> + <a_R,H>, 1: <a_O,G>, 2: <s_L,G> + <s_R,H>; same thread
// layout as k_tt_round (8 threads per base point, 8 windows each), block partials to partial[3][gridDim.x].
__global__ void __launch_bounds__(256) k_tt_commit3(const ge_pniels *__restrict__ table, const scm *__restrict__ aL, const scm *__restrict__ aR,
                                                    const scm *__restrict__ aO, const scm *__restrict__ sL, const scm *__restrict__ sR,
                                                    uint32_t n, uint32_t M0, ge_ext *__restrict__ partial) {
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
    lds[threadIdx.x] = acc; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[cls * gridDim.x + blockIdx.x] = lds[0];
}
// block k: out[k] = sum of partial[k][0..nblk) + blind[k] * (fixed base whose window table is tableX)
__global__ void __launch_bounds__(256) k_tt_commit3_finish(const ge_ext *__restrict__ partial, uint32_t nblk, const scm *__restrict__ blind,
                                                           const ge_pniels *__restrict__ tableX, ge_ext *__restrict__ out) {
    __shared__ ge_ext lds[256];
    const uint32_t cls = blockIdx.x;
    ge_ext acc = ge_identity();
    for (uint32_t s = threadIdx.x; s < nblk; s += 256) acc = ge_add(acc, partial[cls * nblk + s]);
    if (threadIdx.x < TT_WINDOWS) {
        uint32_t w[8]; tt_biased_words(w, blind[cls]);
        const int32_t d = (int32_t)((w[threadIdx.x >> 3] >> (4 * (threadIdx.x & 7u))) & 15u) - 8;
        if (d != 0) {
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            acc = ge_add_pniels_signed(acc, tableX[threadIdx.x * TT_MULTS + mag - 1], neg);
        }
    }
    lds[threadIdx.x] = acc; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[cls] = lds[0];
}
// block 0 -> L, block 1 -> R: sum the block partials, add (c * w) * B with c = <a_lo, b_hi> resp. <a_hi, b_lo>, compress
__global__ void __launch_bounds__(256) k_tt_finish(const ge_ext *__restrict__ partial, uint32_t nblk, const scm *__restrict__ a, const scm *__restrict__ b,
                                                   uint32_t h, scm wq, const ge_pniels *__restrict__ tableB, uint8_t *__restrict__ out) {
    __shared__ ge_ext lds[256];
    __shared__ scm slds[256];
    const uint32_t cls = blockIdx.x;
    scm ip = sc_zero();
    for (uint32_t i = threadIdx.x; i < h; i += 256) ip = sc_add(ip, cls == 0 ? sc_mont_mul(a[i], b[h + i]) : sc_mont_mul(a[h + i], b[i]));
    const scm cw = sc_mont_mul(block_sum_256(ip, slds), wq);
    ge_ext acc = ge_identity();
    for (uint32_t s = threadIdx.x; s < nblk; s += 256) acc = ge_add(acc, partial[cls * nblk + s]);
    if (threadIdx.x < TT_WINDOWS) {
        uint32_t w[8]; tt_biased_words(w, cw);
        const int32_t d = (int32_t)((w[threadIdx.x >> 3] >> (4 * (threadIdx.x & 7u))) & 15u) - 8;
        if (d != 0) {
            const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            acc = ge_add_pniels_signed(acc, tableB[threadIdx.x * TT_MULTS + mag - 1], neg);
        }
    }
    lds[threadIdx.x] = acc; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) ge_compress(out + 32 * cls, lds[0]);
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

// ------------------------------------------------------------------------------------------------ verifier
// compressed points -> affine Niels (Z = 1 after decoding, so no inversion); ok[i] = 0 for invalid encodings
__global__ void __launch_bounds__(64) k_decompress(const uint8_t *__restrict__ in, ge_niels *__restrict__ out, uint32_t *__restrict__ ok, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    ge_ext p; uint32_t good = ge_decompress(p, in + 32 * (size_t)i);
    out[i] = ge_to_niels(p, fe_one());
    ok[i] = good;
}
// s_i = prod_k (bit_{lgN-1-k}(i) ? u_k : u_k^-1), the inner-product verification scalars (dalek verification_scalars)
struct IpaChallenges { scm u[32]; scm uinv[32]; };
__global__ void __launch_bounds__(256) k_ipa_s(const IpaChallenges *__restrict__ ch, scm *__restrict__ s, uint32_t lgN, uint32_t N) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    scm acc = SC_R1();
    for (uint32_t k = 0; k < lgN; k++) acc = sc_mont_mul(acc, ((i >> (lgN - 1 - k)) & 1u) ? ch->u[k] : ch->uinv[k]);
    s[i] = acc;
}
// g_i = gf(i) * (x * y^-i * wR_i - a * s_i) ; h_i = gf(i) * (y^-i * (x * wL_i + wO_i - b * s_{N-1-i}) - 1) ; delta partials = y^-i wR_i wL_i
__global__ void __launch_bounds__(256) k_verify_scalars(const scm *__restrict__ wL, const scm *__restrict__ wR, const scm *__restrict__ wO,
                                                        const scm *__restrict__ yinvpow, const scm *__restrict__ s, scm x, scm a, scm b, scm u_ch,
                                                        scm *__restrict__ g, scm *__restrict__ h, scm *__restrict__ partial, uint32_t n, uint32_t N) {
    __shared__ scm lds[256];
    scm delta = sc_zero();
    const scm one = SC_R1();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const bool real = i < n;
        scm yi = yinvpow[i];
        scm ywr = real ? sc_mont_mul(yi, wR[i]) : sc_zero();
        scm gi = sc_sub(sc_mont_mul(x, ywr), sc_mont_mul(a, s[i]));
        scm t = sc_neg(sc_mont_mul(b, s[N - 1 - i]));
        if (real) { t = sc_add(t, sc_add(sc_mont_mul(x, wL[i]), wO[i])); delta = sc_add(delta, sc_mont_mul(ywr, wL[i])); }
        scm hi = sc_sub(sc_mont_mul(yi, t), one);
        if (!real) { gi = sc_mont_mul(gi, u_ch); hi = sc_mont_mul(hi, u_ch); }
        g[i] = gi; h[i] = hi;
    }
    scm r = block_sum_256(delta, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ------------------------------------------------------------------------------------------------ multiscalar multiplication
// Window j of W covers bits [off(j), off(j+1)) with off(j) = j*254/W: near-equal widths, so that the top window keeps
// (almost) a full width of entropy - with fixed c-bit windows the last one holds only 253 mod c bits and a handful of
// buckets would receive every term.  Signed digits: digit j in (-2^(wd-1), 2^(wd-1)], wd = width of window j.
__device__ __forceinline__ uint32_t msm_off(uint32_t j, uint32_t W) { return (j * 254u) / W; }
__device__ __forceinline__ int32_t msm_digit(const uint32_t w[8], uint32_t W, uint32_t win, uint32_t &carry) {
    uint32_t off = msm_off(win, W), wd = msm_off(win + 1, W) - off, wi = off >> 5, sh = off & 31;
    uint64_t two = (uint64_t)(wi < 8 ? w[wi] : 0u) | ((uint64_t)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
    uint32_t raw = (uint32_t)((two >> sh) & ((1u << wd) - 1u)) + carry;
    if (raw > (1u << (wd - 1))) { carry = 1; return (int32_t)raw - (int32_t)(1u << wd); }
    carry = 0; return (int32_t)raw;
}
__device__ __forceinline__ uint32_t msm_find_seg(const MsmSegs &S, uint32_t g) {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 1; k < BPG_MAX_SEGS; k++) if (k < S.nseg && g >= S.start[k]) s = k;
    return s;
}

// Sorting the (term, window) entries by bucket without global atomics (device-scope atomics on MI355X resolve beyond the
// per-XCD L2 and were the slowest part of the MSM): the terms of each MSM are cut into tiles of 2^lgTile terms; block
// (tile, window) histograms its tile in LDS (LDS atomics) and writes the row H[msm*W+window][tile][0..nb) with plain coalesced
// stores; k_msm_tile_prefix turns the rows of one key column into exclusive prefixes over the tiles and emits the bucket
// totals; after the usual scan of the totals, block (tile, window) reloads its row (+ bucket start) into LDS as cursors and
// scatters its entries.  key = (msm * W + window) * nb + (|digit| - 1); entry = sign << 31 | seg << 27 | index-in-segment.
// Signed digits come from a carry-free recoding: with bias = sum_j 2^(off(j)+wd(j)-1) added to the scalar once, digit j is
// field_j(s + bias) - 2^(wd(j)-1), in [-2^(wd-1), 2^(wd-1)).
struct MsmPlan {
    uint32_t nmsm, W, nb, lgTile, tmax, lgCH;
    uint32_t term_start[5];      // first global term of MSM m (term_start[nmsm] = total)
    uint32_t tile_start[5];      // first tile of MSM m
    uint32_t bias[8];
};
__device__ __forceinline__ int32_t msm_digit_biased(const uint32_t w[8], uint32_t W, uint32_t win) {
    const uint32_t off = msm_off(win, W), wd = msm_off(win + 1, W) - off, wi = off >> 5, sh = off & 31;
    const uint64_t two = (uint64_t)w[wi] | ((uint64_t)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
    return (int32_t)((uint32_t)(two >> sh) & ((1u << wd) - 1u)) - (int32_t)(1u << (wd - 1));
}
__device__ __forceinline__ void msm_biased_words(uint32_t w[8], const scm &sc, const MsmPlan &P) {
    sc_to_words(w, sc);
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t t = (uint64_t)w[k] + P.bias[k] + carry; w[k] = (uint32_t)t; carry = t >> 32; }
}
// biased plain words of every term, once per MSM (the tile kernels run W times over the same scalars)
__global__ void __launch_bounds__(256) k_msm_plain(MsmSegs S, MsmPlan P, uint32_t total, uint4 *__restrict__ plain) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const uint32_t s = msm_find_seg(S, g), i = g - S.start[s];
    uint32_t w[8]; msm_biased_words(w, S.sc[s][i], P);
    plain[2 * (size_t)g] = make_uint4(w[0], w[1], w[2], w[3]);
    plain[2 * (size_t)g + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// pass 1 also notes, for every chunk of 2^lgCH sorted entries, the key of the chunk's first entry (chunk_key): k_bucket_chunks starts
// from it instead of searching starts[]
template <int PASS>
__global__ void __launch_bounds__(256) k_msm_tile(MsmSegs S, MsmPlan P, const uint4 *__restrict__ plain, uint32_t *__restrict__ H,
                                                  const uint32_t *__restrict__ starts, uint32_t *__restrict__ entries, uint32_t *__restrict__ chunk_key) {
    extern __shared__ uint32_t tile_lds[];                   // nb counters (pass 0) or cursors (pass 1)
    const uint32_t T = blockIdx.x, win = blockIdx.y;
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 1; k < 4; k++) if (k < P.nmsm && T >= P.tile_start[k]) m = k;
    const uint32_t t = T - P.tile_start[m];
    const uint32_t g0 = P.term_start[m] + (t << P.lgTile);
    const uint32_t g1 = min(g0 + (1u << P.lgTile), P.term_start[m + 1]);
    const uint32_t mw = m * P.W + win;
    uint32_t *row = H + ((size_t)mw * P.tmax + t) * P.nb;
    if (PASS == 0) for (uint32_t b = threadIdx.x; b < P.nb; b += 256) tile_lds[b] = 0;
    else for (uint32_t b = threadIdx.x; b < P.nb; b += 256) tile_lds[b] = row[b] + starts[(size_t)mw * P.nb + b];
    __syncthreads();
    // the window's bits sit in one or two of the eight words: read only the 16-byte half (or both halves) that holds them
    const uint32_t off = msm_off(win, P.W), wd = msm_off(win + 1, P.W) - off, wi = off >> 5, sh = off & 31;
    for (uint32_t g = g0 + threadIdx.x; g < g1; g += 256) {
        const uint32_t *pw = reinterpret_cast<const uint32_t *>(plain + 2 * (size_t)g);
        const uint64_t two = (uint64_t)pw[wi] | ((uint64_t)(wi + 1 < 8 ? pw[wi + 1] : 0u) << 32);
        const int32_t d = (int32_t)((uint32_t)(two >> sh) & ((1u << wd) - 1u)) - (int32_t)(1u << (wd - 1));
        if (d == 0) continue;
        const uint32_t neg = d < 0, mag = neg ? (uint32_t)(-d) : (uint32_t)d;
        if (PASS == 0) atomicAdd(&tile_lds[mag - 1], 1u);
        else {
            const uint32_t s = msm_find_seg(S, g), i = g - S.start[s];
            const uint32_t pos = atomicAdd(&tile_lds[mag - 1], 1u); entries[pos] = (neg << 31) | (s << 27) | i;
            if ((pos & ((1u << P.lgCH) - 1u)) == 0) chunk_key[pos >> P.lgCH] = mw * P.nb + mag - 1;
        }
    }
    if (PASS == 0) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < P.nb; b += 256) row[b] = tile_lds[b];
    }
}
// one thread per key: H[mw][t][b] <- sum_{t' < t} H[mw][t'][b], counts[key] <- column total
__global__ void __launch_bounds__(256) k_msm_tile_prefix(MsmPlan P, uint32_t *__restrict__ H, uint32_t *__restrict__ counts, uint32_t nkeys,
                                                         uint32_t *__restrict__ heavy_count) {
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key == 0) *heavy_count = 0;                          // list of k_bucket_combine, filled later on this stream
    if (key >= nkeys) return;
    const uint32_t mw = key / P.nb, b = key - mw * P.nb, m = mw / P.W;
    const uint32_t nt = P.tile_start[m + 1] - P.tile_start[m];
    uint32_t *col = H + (size_t)mw * P.tmax * P.nb + b;
    uint32_t run = 0;
    for (uint32_t t = 0; t < nt; t++) { const uint32_t v = col[(size_t)t * P.nb]; col[(size_t)t * P.nb] = run; run += v; }
    counts[key] = run;
}

// exclusive scan of counts[0..nkeys) in three launches (chunk = 2048 keys per block)
#define SCAN_CHUNK 2048
__global__ void __launch_bounds__(256) k_scan_blocksums(const uint32_t *__restrict__ counts, uint32_t nkeys, uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t lds[256];
    uint32_t base = blockIdx.x * SCAN_CHUNK, s = 0;
    for (uint32_t k = threadIdx.x; k < SCAN_CHUNK; k += 256) if (base + k < nkeys) s += counts[base + k];
    lds[threadIdx.x] = s; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) lds[threadIdx.x] += lds[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = lds[0];
}
__global__ void k_scan_top(uint32_t *__restrict__ blocksum, uint32_t nblocks) {   // single thread: nblocks <= a few thousand
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0;
    for (uint32_t b = 0; b < nblocks; b++) { uint32_t v = blocksum[b]; blocksum[b] = run; run += v; }
    blocksum[nblocks] = run;
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ counts, uint32_t nkeys, const uint32_t *__restrict__ blocksum,
                                                    uint32_t *__restrict__ starts, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t lds[256];
    uint32_t base = blockIdx.x * SCAN_CHUNK;
    uint32_t v[8], s = 0;                       // thread owns 8 consecutive keys
#pragma unroll
    for (int k = 0; k < 8; k++) { uint32_t idx = base + threadIdx.x * 8 + k; v[k] = idx < nkeys ? counts[idx] : 0; s += v[k]; }
    lds[threadIdx.x] = s; __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {     // inclusive Hillis-Steele scan
        uint32_t x = threadIdx.x >= d ? lds[threadIdx.x - d] : 0; __syncthreads();
        lds[threadIdx.x] += x; __syncthreads();
    }
    uint32_t run = blocksum[blockIdx.x] + lds[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t idx = base + threadIdx.x * 8 + k;
        if (idx < nkeys) { starts[idx] = run; cursor[idx] = run; }
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) starts[nkeys] = blocksum[gridDim.x];
}

// Balanced bucket sweep.  The entry list is sorted by bucket (starts[]); thread c adds the points of the fixed-size chunk
// [c*CH, (c+1)*CH) whatever buckets it crosses, so a bucket that received thousands of terms (identical scalars: the -y^h
// padding terms of the first IPA round, repeated witness values, range-proof bits) is spread over many threads instead of
// serialising one.  A bucket that lies inside one chunk is stored directly; a bucket that crosses chunk boundaries leaves
// one partial per chunk (slotA = piece at the chunk's beginning, slotB = piece at its end) for k_bucket_combine.
// bucket that holds sorted entry e: the k >= klo with starts[k] <= e < starts[k+1]  (upper_bound - 1)
__device__ __forceinline__ uint32_t msm_bucket_of(const uint32_t *__restrict__ starts, uint32_t nkeys, uint32_t e, uint32_t klo) {
    uint32_t lo = klo, hi = nkeys + 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= e) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
__global__ void __launch_bounds__(256) k_bucket_chunks(MsmSegs S, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ entries,
                                                       const uint32_t *__restrict__ chunk_key, ge_ext *__restrict__ buckets,
                                                       ge_ext *__restrict__ slotA, ge_ext *__restrict__ slotB, uint32_t nkeys, uint32_t lgCH) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e0 = c << lgCH;
    const uint32_t M = starts[nkeys];                      // true entry count (zero digits were skipped)
    if (e0 >= M) return;
    const uint32_t e1 = (e0 + (1u << lgCH) < M) ? e0 + (1u << lgCH) : M;
    uint32_t k = chunk_key[c], kstart = starts[k], kend = starts[k + 1], seg_begin = e0;
    uint32_t kend2 = starts[k + 2 <= nkeys ? k + 2 : nkeys];   // end of the next bucket, loaded one boundary ahead of its use
    uint32_t ent = entries[e0];
    ge_ext acc = ge_identity();
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t sg = (ent >> 27) & 7u, neg = ent >> 31;
        const ge_niels q = S.pts[sg][msm_point_index(S, sg, ent & 0x07ffffffu)];
        if (e + 1 < e1) ent = entries[e + 1];
        if (e >= kend) {
            if ((kstart >> lgCH) == ((kend - 1) >> lgCH)) buckets[k] = acc;
            else { if (seg_begin == e0) slotA[c] = acc; /* a piece ending inside the chunk cannot also end it */ }
            acc = ge_identity(); seg_begin = e;
            k++; kstart = kend; kend = kend2;
            if (e >= kend) {                                 // a run of empty buckets (half of a 15-bit window is structurally
                k = msm_bucket_of(starts, nkeys, e, k + 1);  // empty): search instead of walking it with dependent loads
                kstart = starts[k]; kend = starts[k + 1];
            }
            kend2 = starts[k + 2 <= nkeys ? k + 2 : nkeys];
        }
        acc = ge_madd_signed(acc, q, neg);
    }
    if ((kstart >> lgCH) == ((kend - 1) >> lgCH)) buckets[k] = acc;
    else { if (seg_begin == e0) slotA[c] = acc; if (kend >= e1) slotB[c] = acc; }
}

// one thread per bucket: identity for empty buckets, nothing for single-chunk buckets, slotB[c0] + slotA[c0+1..c1] for a
// bucket spread over a few chunks; a bucket spread over more than HEAVY_CHUNKS chunks (thousands of identical scalars: the
// -y^h padding terms of the first IPA round, repeated witness values) goes on the heavy list for k_bucket_combine_heavy
#define HEAVY_CHUNKS 32
__global__ void __launch_bounds__(256) k_bucket_combine(const uint32_t *__restrict__ starts, ge_ext *__restrict__ buckets,
                                                        const ge_ext *__restrict__ slotA, const ge_ext *__restrict__ slotB,
                                                        uint32_t nkeys, uint32_t lgCH, uint32_t *__restrict__ heavy /* [0] = count, then keys */) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkeys) return;
    const uint32_t s0 = starts[k], s1 = starts[k + 1];
    if (s0 == s1) { buckets[k] = ge_identity(); return; }
    const uint32_t c0 = s0 >> lgCH, c1 = (s1 - 1) >> lgCH;
    if (c0 == c1) return;
    if (c1 - c0 > HEAVY_CHUNKS) { heavy[1 + atomicAdd(&heavy[0], 1u)] = k; return; }
    ge_ext acc = slotB[c0];
    for (uint32_t c = c0 + 1; c <= c1; c++) acc = ge_add(acc, slotA[c]);
    buckets[k] = acc;
}
// one wave per heavy bucket (grid-stride over the list): lane-strided partial sums, then a 6-level tree through LDS
__global__ void __launch_bounds__(256) k_bucket_combine_heavy(const uint32_t *__restrict__ starts, ge_ext *__restrict__ buckets,
                                                              const ge_ext *__restrict__ slotA, const ge_ext *__restrict__ slotB,
                                                              uint32_t lgCH, const uint32_t *__restrict__ heavy) {
    __shared__ ge_ext lds[256];
    const uint32_t count = heavy[0], lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    ge_ext *L = lds + wv * 64;
    for (uint32_t it = blockIdx.x * 4 + wv; it < count; it += gridDim.x * 4) {      // wave-uniform trip count; no block barrier inside
        const uint32_t k = heavy[1 + it];
        const uint32_t c0 = starts[k] >> lgCH, c1 = (starts[k + 1] - 1) >> lgCH;
        ge_ext acc = lane == 0 ? slotB[c0] : ge_identity();
        for (uint32_t c = c0 + 1 + lane; c <= c1; c += 64) acc = ge_add(acc, slotA[c]);
        L[lane] = acc;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t d = 32; d > 0; d >>= 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < d) L[lane] = ge_add(L[lane], L[lane + d]);
        }
        if (lane == 0) buckets[k] = L[0];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// per (msm, window, segment of SEG buckets): sum_b (b+1) * bucket[b] over the segment -> partial
__global__ void __launch_bounds__(64) k_bucket_reduce(const ge_ext *__restrict__ buckets, ge_ext *__restrict__ partial,
                                                     uint32_t nb, uint32_t seg, uint32_t nseg_per_win, uint32_t total) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint32_t win = t / nseg_per_win, sg = t % nseg_per_win;
    uint32_t lo = sg * seg;
    const ge_ext *B = buckets + (size_t)win * nb;
    ge_ext run = ge_identity(), acc = ge_identity();
    for (int32_t b = (int32_t)(lo + seg) - 1; b >= (int32_t)lo; b--) { run = ge_add(run, B[b]); acc = ge_add(acc, run); }
    // acc = sum (b - lo + 1) B_b ; add lo * run
    ge_ext m = ge_identity();
    for (int32_t k = 15; k >= 0; k--) { m = ge_dbl(m); if ((lo >> k) & 1u) m = ge_add(m, run); }   // lo < nb <= 2^15
    partial[t] = ge_add(acc, m);
}

// window sums: one block per (msm, window) adds that window's segment partials (strided loads, LDS tree)
__global__ void __launch_bounds__(256) k_window_sums(const ge_ext *__restrict__ partial, ge_ext *__restrict__ wsum, uint32_t nseg_per_win) {
    __shared__ ge_ext lds[256];
    const ge_ext *P = partial + (size_t)blockIdx.x * nseg_per_win;
    ge_ext acc = ge_identity();
    for (uint32_t s = threadIdx.x; s < nseg_per_win; s += 256) acc = ge_add(acc, P[s]);
    lds[threadIdx.x] = acc; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d && threadIdx.x + d < nseg_per_win) lds[threadIdx.x] = ge_add(lds[threadIdx.x], lds[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) wsum[blockIdx.x] = lds[0];
}

// Horner over the window sums, result = sum_j 2^off(j) * S_j: about 250 DEPENDENT doublings, the serial tail of every MSM.
// One block of 4 waves per MSM; the four independent field products of each doubling / addition step are computed by the
// four waves concurrently (each wave is on its own SIMD, its lanes all hold the same value) and exchanged through LDS.
// A single lane needs ~2,600 instructions per doubling; here each wave issues ~1/4 of that between two barriers.
struct HornerLds { fe c[4]; fe s[4]; };
__device__ __forceinline__ void horner_dbl(HornerLds &L, uint32_t wv) {
    // L.c = (X, Y, Z, T) -> doubled point in L.c
    fe in = (wv == 3) ? fe_add(L.c[0], L.c[1]) : L.c[wv];          // X, Y, Z, X+Y
    fe sq = fe_sq(in);
    __syncthreads();
    L.s[wv] = sq;                                                  // XX, YY, ZZ, (X+Y)^2
    __syncthreads();
    fe XX = L.s[0], YY = L.s[1];
    fe YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX);
    fe a, b;
    if (wv == 1) { a = YpX; b = YmX; }                              // Y3 = YpX * YmX
    else {
        fe ZZ2 = fe_add(L.s[2], L.s[2]);
        fe cT = fe_sub(ZZ2, YmX), cX = fe_sub(L.s[3], YpX);
        if (wv == 0) { a = cX; b = cT; }                            // X3 = cX * cT
        else if (wv == 2) { a = YmX; b = cT; }                      // Z3 = YmX * cT
        else { a = cX; b = YpX; }                                   // T3 = cX * YpX
    }
    fe r = fe_mul(a, b);
    L.c[wv] = r;
    __syncthreads();
}
__device__ __forceinline__ void horner_add(HornerLds &L, const ge_ext &q, uint32_t wv) {
    // L.c += q   (extended + extended, unified formulas)
    fe X1 = L.c[0], Y1 = L.c[1];
    fe p;
    if (wv == 0) p = fe_mul(fe_sub(Y1, X1), fe_sub(q.Y, q.X));      // A
    else if (wv == 1) p = fe_mul(fe_add(Y1, X1), fe_add(q.Y, q.X)); // B
    else if (wv == 2) p = fe_mul(fe_mul(L.c[3], q.T), FE_D2());     // C
    else { p = fe_mul(L.c[2], q.Z); p = fe_add(p, p); }             // D
    __syncthreads();
    L.s[wv] = p;
    __syncthreads();
    fe A = L.s[0], B = L.s[1], C = L.s[2], D = L.s[3];
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    fe r = (wv == 0) ? fe_mul(E, F) : (wv == 1) ? fe_mul(G, H) : (wv == 2) ? fe_mul(F, G) : fe_mul(E, H);
    L.c[wv] = r;
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_msm_horner(const ge_ext *__restrict__ wsum, ge_ext *__restrict__ result, uint32_t W) {
    __shared__ HornerLds L;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave index, uniform
    const ge_ext *S = wsum + (size_t)blockIdx.x * W;
    if (threadIdx.x == 0) { ge_ext t = S[W - 1]; L.c[0] = t.X; L.c[1] = t.Y; L.c[2] = t.Z; L.c[3] = t.T; }
    __syncthreads();
    for (int32_t win = (int32_t)W - 2; win >= 0; win--) {
        const uint32_t shift = msm_off(win + 1, W) - msm_off(win, W);
        for (uint32_t k = 0; k < shift; k++) horner_dbl(L, wv);
        const ge_ext q = S[win];
        horner_add(L, q, wv);
    }
    if (threadIdx.x == 0) { ge_ext t; t.X = L.c[0]; t.Y = L.c[1]; t.Z = L.c[2]; t.T = L.c[3]; result[blockIdx.x] = t; }
}

}  // namespace bpg
