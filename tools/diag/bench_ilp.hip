// standalone micro-benchmark (diagnostic, not part of the product): what a DEPENDENT point addition costs a wave that has its SIMD to itself, and whether
// independent work in the same thread (two, four chains side by side in the source) overlaps with it - the question behind the latency chains of a lone proof
// (window epilogue, small folds, tail trees: DESIGN.md section 9).  One wave per SIMD (1,024 waves), every lane runs `iters` additions per chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bench_ilp tools/diag/bench_ilp.hip && /tmp/bench_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bulletproofs_gadgets_amd/csrc/hip/ge.cuh"
using namespace bpg;

template <int C> __global__ void __launch_bounds__(64) k_chain(ge_ext *out, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    ge_ext p[C], q = ge_identity(); q.X = FE_D(); q.T = FE_D2(); q.X.v[0] ^= t;
    for (int c = 0; c < C; c++) { p[c] = ge_identity(); p[c].Y.v[1] ^= (uint32_t)c + t; }
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < C; c++) p[c] = ge_add(p[c], q);          // (not curve points: the instruction stream is what is timed)
    }
    ge_ext r = p[0];
    for (int c = 1; c < C; c++) r = ge_add(r, p[c]);
    out[t] = r;
}
template <int C> static double run(ge_ext *buf, uint32_t waves_per_simd, uint32_t iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const dim3 grid(1024 * waves_per_simd), block(64);
    hipLaunchKernelGGL(k_chain<C>, grid, block, 0, 0, buf, 2u);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_chain<C>, grid, block, 0, 0, buf, iters);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return (double)ms * 1e3 / iters;        // us per loop trip (C additions)
}
int main() {
    ge_ext *buf; (void)hipMalloc(&buf, (size_t)1024 * 8 * 64 * sizeof(ge_ext));
    std::printf("{\"us_per_addition_per_chain\": {\n");
    for (uint32_t w = 1; w <= 4; w *= 2) {
        const double c1 = run<1>(buf, w, 400), c2 = run<2>(buf, w, 400), c4 = run<4>(buf, w, 200);
        std::printf("  \"%u wave(s) per SIMD\": {\"1 chain\": %.3f, \"2 chains in one thread\": %.3f, \"4 chains in one thread\": %.3f}%s\n", w, c1, c2 / 2, c4 / 4, w < 4 ? "," : "");
    }
    std::printf(" },\n \"note\": \"time of one loop trip / chains: below the 1-chain figure means the independent additions of a thread overlap\"}\n");
    return 0;
}
