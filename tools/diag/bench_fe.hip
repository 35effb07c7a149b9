// standalone micro-benchmark of field-multiplication variants on gfx950 (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bulletproofs_gadgets_amd/csrc/hip/fe.cuh"
#include "fe10.cuh"
#if defined(__HIP_DEVICE_COMPILE__)
#include "fe10_asm.cuh"
#else
__host__ __device__ static inline bpg10::fe10 fe10_mul_asm(const bpg10::fe10 &a, const bpg10::fe10 &) { return a; }
#endif
using namespace bpg;
using namespace bpg10;
#if defined(__HIP_DEVICE_COMPILE__)
#include "fe_cols.cuh"
#else
__host__ __device__ static inline fe fe_mul_cols(const fe &a, const fe &b) { return fe_mul(a, b); }
#endif
// variant: two interleaved column accumulators per step
__device__ __forceinline__ void mac2(uint64_t &loA, uint32_t &hiA, uint32_t a0, uint32_t b0, uint64_t &loB, uint32_t &hiB, uint32_t a1, uint32_t b1) {
    uint64_t c0, c1;
    asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\tv_mad_u64_u32 %2, %5, %8, %9, %2\n\tv_addc_co_u32_e64 %1, %4, 0, %1, %4\n\tv_addc_co_u32_e64 %3, %5, 0, %3, %5"
        : "+v"(loA), "+v"(hiA), "+v"(loB), "+v"(hiB), "=&s"(c0), "=&s"(c1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
__device__ __forceinline__ void mac1(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t b) {
    uint64_t c; asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2" : "+v"(lo), "+v"(hi), "=&s"(c) : "v"(a), "v"(b));
}
// columns k (A) and k+8 (B) are independent until the final carry merge: low half and high half computed side by side
__device__ __forceinline__ fe fe_mul_pair(const fe &a, const fe &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return fe_mul(a, b);
#else
    uint32_t t[16];
    uint64_t loA = 0, loB = 0; uint32_t hiA = 0, hiB = 0;
    // pass: column k (k = 0..7) has k+1 products, column k+8 has 7-k products (k+8 <= 14)
    uint32_t tl[8], th[8]; uint64_t carryA = 0, carryB = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        loA = carryA; hiA = 0; loB = carryB; hiB = 0;
        const int nA = k + 1, nB = 7 - k;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool doA = j < nA, doB = j < nB;
            if (doA && doB) mac2(loA, hiA, a.v[j], b.v[k - j], loB, hiB, a.v[k + 1 + j], b.v[7 - j]);
            else if (doA) mac1(loA, hiA, a.v[j], b.v[k - j]);
            else if (doB) mac1(loB, hiB, a.v[k + 1 + j], b.v[7 - j]);
        }
        tl[k] = (uint32_t)loA; carryA = (loA >> 32) | ((uint64_t)hiA << 32);
        th[k] = (uint32_t)loB; carryB = (loB >> 32) | ((uint64_t)hiB << 32);
    }
    // merge: high half += carryA (64-bit) at limb 8; th[7] is column 15 = carryB of column 14 ... (th[k] = column k+8)
    uint64_t c = carryA;
#pragma unroll
    for (int k = 0; k < 8; k++) { c += th[k]; t[8 + k] = (uint32_t)c; c >>= 32; t[k] = tl[k]; }
    return fe_fold512_dev(t);
#endif
}
template <int V> __global__ void __launch_bounds__(256) kb(fe *out, uint32_t iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(), b = FE_SQRTM1(), c = FE_D2(), d = FE_ONE_MINUS_D_SQ();
    a.v[0] ^= t; b.v[1] ^= t; c.v[2] ^= t; d.v[3] ^= t;
    for (uint32_t i = 0; i < iters; i++) {
        if (V == 0) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
        if (V == 1) { a = fe_mul(a, b); a = fe_mul(a, c); a = fe_mul(a, d); a = fe_mul(a, b); }          // one dependent chain
        if (V == 2) { a = fe_mul_pair(a, b); b = fe_mul_pair(b, c); c = fe_mul_pair(c, d); d = fe_mul_pair(d, a); }
        if (V == 3) { a = fe_mul_pair(a, b); a = fe_mul_pair(a, c); a = fe_mul_pair(a, d); a = fe_mul_pair(a, b); }
        if (V == 6) { a = fe_mul_cols(a, b); b = fe_mul_cols(b, c); c = fe_mul_cols(c, d); d = fe_mul_cols(d, a); }
        if (V == 4) { a = fe_sq(a); b = fe_sq(b); c = fe_sq(c); d = fe_sq(d); }
        if (V == 5) { a = fe_add(a, b); b = fe_sub(b, c); c = fe_add(c, d); d = fe_sub(d, a); }
    }
    out[t] = fe_add(fe_add(a, b), fe_add(c, d));
}
template <int V> __global__ void __launch_bounds__(256) kb10(fe *out, uint32_t iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a8 = FE_D(), b8 = FE_SQRTM1(), c8 = FE_D2(), d8 = FE_ONE_MINUS_D_SQ();
    a8.v[0] ^= t; b8.v[1] ^= t; c8.v[2] ^= t; d8.v[3] ^= t;
    a8 = fe_freeze(a8); b8 = fe_freeze(b8); c8 = fe_freeze(c8); d8 = fe_freeze(d8);
    fe10 a = fe10_from8(a8.v), b = fe10_from8(b8.v), c = fe10_from8(c8.v), d = fe10_from8(d8.v);
    for (uint32_t i = 0; i < iters; i++) {
        if (V == 0) { a = fe10_mul(a, b); b = fe10_mul(b, c); c = fe10_mul(c, d); d = fe10_mul(d, a); }
        if (V == 1) { a = fe10_mul_asm(a, b); b = fe10_mul_asm(b, c); c = fe10_mul_asm(c, d); d = fe10_mul_asm(d, a); }
        if (V == 4) { a = fe10_sq(a); b = fe10_sq(b); c = fe10_sq(c); d = fe10_sq(d); }
        if (V == 5) { a = fe10_add(a, b); b = fe10_sub(b, c); c = fe10_add(c, d); d = fe10_sub(d, a);
                      if ((i & 1) == 1) { a = fe10_mul(a, a); b = fe10_mul(b, b); c = fe10_mul(c, c); d = fe10_mul(d, d); } }   // lazy sums need a carrying op now and then
        if (V == 9) { fe10 x = fe10_mul(a, b); fe10 y = fe10_sq(x); a = fe10_carry(fe10_sub(fe10_add(x, y), a)); b = fe10_mul(fe10_sub(b, x), y); }  // mixed check
    }
    // carry, pack, canonical
    fe10 s4 = fe10_mul(fe10_add(fe10_add(a, b), fe10_add(c, d)), fe10_from8(fe_one().v));
    // full carry of limb 1 excess
    uint32_t cy = s4.v[1] >> 25; s4.v[1] &= M25; s4.v[2] += cy; cy = s4.v[2] >> 26; s4.v[2] &= M26; s4.v[3] += cy;
    fe o; fe10_to8(o.v, s4);
    out[t] = fe_freeze(o);
}
template <int V> __global__ void __launch_bounds__(256) kb8check(fe *out, uint32_t iters) {                 // the same arithmetic in the product's layout
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(), b = FE_SQRTM1(), c = FE_D2(), d = FE_ONE_MINUS_D_SQ();
    a.v[0] ^= t; b.v[1] ^= t; c.v[2] ^= t; d.v[3] ^= t;
    for (uint32_t i = 0; i < iters; i++) {
        if (V == 0) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
        if (V == 1) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
        if (V == 4) { a = fe_sq(a); b = fe_sq(b); c = fe_sq(c); d = fe_sq(d); }
        if (V == 9) { fe x = fe_mul(a, b); fe y = fe_sq(x); a = fe_sub(fe_add(x, y), a); b = fe_mul(fe_sub(b, x), y); }
    }
    out[t] = fe_freeze(fe_add(fe_add(a, b), fe_add(c, d)));
}
template <int V> double run10(fe *buf, uint32_t iters, int blocks, double per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kb10<V>, dim3(blocks), dim3(256), 0, 0, buf, 8u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kb10<V>, dim3(blocks), dim3(256), 0, 0, buf, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * iters * per_iter / (ms * 1e-3);
}
template <int V> int check10(fe *buf) {
    fe *h0 = new fe[64], *h2 = new fe[64];
    hipLaunchKernelGGL(kb8check<V>, dim3(1), dim3(64), 0, 0, buf, 7u); hipMemcpy(h0, buf, 64 * sizeof(fe), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kb10<V>, dim3(1), dim3(64), 0, 0, buf, 7u); hipMemcpy(h2, buf, 64 * sizeof(fe), hipMemcpyDeviceToHost);
    int same = 1; for (int i = 0; i < 64; i++) for (int k = 0; k < 8; k++) same &= h0[i].v[k] == h2[i].v[k];
    return same;
}
template <int V> double run(fe *buf, uint32_t iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, buf, 8u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, buf, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * iters * 4.0 / (ms * 1e-3);
}
int main() {
    fe *buf; hipMalloc(&buf, 256 * 16 * 256 * sizeof(fe));
    // correctness cross-check of the pair variant against fe_mul on a few values happens through equal outputs of V0/V2 chains
    fe *h0 = new fe[64], *h2 = new fe[64];
    hipLaunchKernelGGL(kb<0>, dim3(1), dim3(64), 0, 0, buf, 5u); hipMemcpy(h0, buf, 64 * sizeof(fe), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kb<2>, dim3(1), dim3(64), 0, 0, buf, 5u); hipMemcpy(h2, buf, 64 * sizeof(fe), hipMemcpyDeviceToHost);
    int same = 1; for (int i = 0; i < 64; i++) { fe x = fe_freeze(h0[i]), y = fe_freeze(h2[i]); for (int k = 0; k < 8; k++) same &= x.v[k] == y.v[k]; }
    printf("pair variant equals fe_mul: %d\n", same);
    hipLaunchKernelGGL(kb<6>, dim3(1), dim3(64), 0, 0, buf, 5u); hipMemcpy(h2, buf, 64 * sizeof(fe), hipMemcpyDeviceToHost);
    same = 1; for (int i = 0; i < 64; i++) { fe x = fe_freeze(h0[i]), y = fe_freeze(h2[i]); for (int k = 0; k < 8; k++) same &= x.v[k] == y.v[k]; }
    printf("column-block variant equals fe_mul: %d\n", same);
    printf("radix-2^25.5 prototype equals the 8x32 layout: mul %d  sq %d  mixed %d  hand-scheduled mul %d\n", check10<0>(buf), check10<4>(buf), check10<9>(buf), check10<1>(buf));
    for (int blocks : {256 * 4, 256 * 8, 256 * 16})
        printf("blocks %d: fe10 mul4 %.3e  sq4 %.3e  addsub(+1 mul per 4) %.3e  mul4-asm %.3e ops/s\n", blocks, run10<0>(buf, 2000, blocks, 4.0), run10<4>(buf, 2000, blocks, 4.0), run10<5>(buf, 2000, blocks, 4.0), run10<1>(buf, 2000, blocks, 4.0));
    for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
        printf("blocks %d: mul4chains %.3e  mul1chain %.3e  pair4 %.3e  pair1 %.3e  sq4 %.3e  addsub %.3e  cols4 %.3e\n", blocks,
               run<0>(buf, 2000, blocks), run<1>(buf, 2000, blocks), run<2>(buf, 2000, blocks), run<3>(buf, 2000, blocks), run<4>(buf, 2000, blocks), run<5>(buf, 20000, blocks), run<6>(buf, 2000, blocks));
    }
    return 0;
}
