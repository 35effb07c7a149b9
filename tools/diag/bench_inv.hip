// standalone micro-benchmark (diagnostic, not part of the product): what ONE field inversion costs a wave, in field multiplications - the constant behind
// "why not a 6-multiplication batched-affine bucket addition" (DESIGN.md section 8).  A wave64 issues an inversion for all of its lanes or for one at the
// same price, so a batch of B affine additions per lane shares 1 inversion + 3 (B - 1) products (Montgomery's trick): the addition costs
// 2M + 1S (lambda, lambda^2, y3) + 3M (its share of the trick) + inv / B against the 7M of the extended mixed addition the sweep uses.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bench_inv tools/diag/bench_inv.hip && /tmp/bench_inv
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bulletproofs_gadgets_amd/csrc/hip/ge.cuh"
using namespace bpg;

__global__ void __launch_bounds__(256) k_mul(fe *out, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(), b = FE_SQRTM1(); a.v[0] ^= t; b.v[1] ^= t;
    for (uint32_t i = 0; i < iters; i++) { a = fe_mul(a, b); b = fe_mul(b, a); }
    out[t] = fe_add(a, b);
}
__global__ void __launch_bounds__(256) k_inv(fe *out, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(); a.v[0] ^= t;
    for (uint32_t i = 0; i < iters; i++) a = fe_add(fe_invert(a), FE_D2());
    out[t] = a;
}
__global__ void __launch_bounds__(256) k_madd(ge_ext *out, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    ge_ext p = ge_identity(); ge_niels q; q.ypx = FE_D(); q.ymx = FE_SQRTM1(); q.t2d = FE_D2(); q.ypx.v[0] ^= t;
    for (uint32_t i = 0; i < iters; i++) p = ge_madd(p, q);          // (not a curve point: the instruction stream is what is timed)
    out[t] = p;
}
template <class K, class T> static double run(K kern, T *buf, uint32_t iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, buf, 2u);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, buf, iters);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return (double)ms;
}
int main() {
    void *buf; (void)hipMalloc(&buf, (size_t)256 * 8 * 256 * sizeof(ge_ext));
    const double tm = run(k_mul, (fe *)buf, 4000) / (2.0 * 4000), ti = run(k_inv, (fe *)buf, 40) / 40.0, ta = run(k_madd, (ge_ext *)buf, 1200) / 1200.0;
    const double inv_fm = ti / tm, madd_fm = ta / tm;
    std::printf("{\"ms_per_mul_all_lanes\": %.6f, \"inversion_in_multiplications\": %.1f, \"mixed_addition_in_multiplications\": %.2f,\n", tm, inv_fm, madd_fm);
    std::printf(" \"batched_affine_addition_in_multiplications\": {");
    const int Bs[] = {8, 32, 64, 128, 256, 1024, 4096};
    for (int k = 0; k < 7; k++) std::printf("%s\"B=%d\": %.2f", k ? ", " : "", Bs[k], 2.0 + 0.85 + 3.0 + inv_fm / Bs[k] + (madd_fm - 7.0) * 6.0 / 8.0);
    std::printf("},\n \"note\": \"2M + 1S (0.85M) + 3M of Montgomery's trick + inversion / B + six of the eight additions and subtractions of the mixed addition; the sweep's mixed addition is the "
                "mixed_addition_in_multiplications figure; state per pending addition: 32 B of prefix product + 64 B of coordinates, B of them per lane\"}\n");
    return 0;
}
