// scalar-vector kernels of the polynomial phase and the circuit upload (CSR -> CSC on the device) - part of kernels.cuh (included from there, in this order; see its header for the kernel map and the data layout)
#pragma once

namespace bpg {

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
// 64-byte TranscriptRng draws -> scalars (Scalar::random = from_bytes_mod_order_wide).
// Canary: the device slab the draws of a proof are uploaded into is reused from proof to proof; when it changes owner the first draw of every
// uploaded block is overwritten with BLIND_POISON words (k_blind_poison).  A draw that still reads as poison was never uploaded - a dropped copy that
// reported no error - and the scalar would be built from something the host did not draw (in the worst case the previous proof's s_L, s_R, which
// together with this proof would leak the witness): it raises *stale, and prove() refuses to emit the proof.  A real draw equals the pattern with
// probability 2^-512.
#define BLIND_POISON 0xa5c3a5c3u
__global__ void __launch_bounds__(256) k_blind_poison(uint32_t *__restrict__ raw, uint32_t nblocks, uint32_t draws_per_block) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * 16u) return;
    raw[(size_t)(t >> 4) * draws_per_block * 16u + (t & 15u)] = BLIND_POISON;
}
__global__ void __launch_bounds__(256) k_sc_from_wide(const uint32_t *__restrict__ in, scm *__restrict__ out, uint32_t count, uint32_t *__restrict__ stale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[16];
    const uint4 *src = reinterpret_cast<const uint4 *>(in + 16 * (size_t)i);
#pragma unroll
    for (int k = 0; k < 4; k++) { uint4 q = src[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) diff |= w[k] ^ BLIND_POISON;
    if (diff == 0) atomicOr(stale, 1u);
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
// out[i] = base^i for i < count (Montgomery form), for up to three tables in one launch (blockIdx.y picks the table: y^i, y^-i and z^i of a proof).
// Thread t walks i = t, t+T, ... multiplying by base^T; T = 2^lgT.
struct ExpTables { scm base[3]; scm *out[3]; uint32_t count[3], lgT[3]; };
__global__ void __launch_bounds__(256) k_exp_table(ExpTables E) {
    const uint32_t k = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lgT = E.lgT[k], T = 1u << lgT, count = E.count[k];
    if (t >= T) return;
    scm *__restrict__ out = E.out[k];
    scm cur = SC_R1(), sq = E.base[k];
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
// the same with every sum multiplied by w (the c_L * w, c_R * w scalars of Q = w * B in an inner-product round: one launch instead of two)
__global__ void __launch_bounds__(256) k_reduce_partials_scaled(const scm *__restrict__ partial, uint32_t parts, uint32_t stride, scm *__restrict__ out, scm w) {
    __shared__ scm lds[256];
    uint32_t k = blockIdx.x;
    scm acc = sc_zero();
    for (uint32_t p = threadIdx.x; p < parts; p += 256) acc = sc_add(acc, partial[(size_t)p * stride + k]);
    scm r = block_sum_256(acc, lds);
    if (threadIdx.x == 0) out[k] = sc_mont_mul(r, w);
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

}  // namespace bpg
