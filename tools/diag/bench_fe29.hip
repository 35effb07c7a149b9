// standalone micro-benchmark (diagnostic, not part of the product): the saturated 8 x 32 field layout of the product against the radix-2^29
// prototype of fe29.cuh - multiplication, squaring and the mixed point addition of the bucket sweep - with the results compared.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bench_fe29 tools/diag/bench_fe29.hip && /tmp/bench_fe29
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../bulletproofs_gadgets_amd/csrc/hip/ge.cuh"
#if defined(__HIP_DEVICE_COMPILE__)
#include "fe29.cuh"
using namespace bpg29;
#endif
using namespace bpg;

#if defined(__HIP_DEVICE_COMPILE__)
struct ge29 { fe29 X, Y, Z, T; };
__device__ __forceinline__ fe29 sel29(const fe29 &a, const fe29 &b, uint32_t pick_b) { fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r; }
// p tight, q = (y+x, y-x, 2dxy) tight; neg subtracts (operands exchanged, F and G exchanged)
__device__ __forceinline__ ge29 ge29_madd_signed(const ge29 &p, const fe29 &ypx, const fe29 &ymx, const fe29 &t2d, uint32_t neg) {
    const fe29 qp = sel29(ypx, ymx, neg), qm = sel29(ymx, ypx, neg);
    const fe29 A = fe29_mul(fe29_sub(p.Y, p.X), qm);
    const fe29 B = fe29_mul(fe29_add(p.Y, p.X), qp);
    const fe29 C = fe29_mul(p.T, t2d);
    const fe29 D = fe29_add(p.Z, p.Z);
    const fe29 E = fe29_carry(fe29_sub(B, A)), H = fe29_add(B, A);
    const fe29 Fm = fe29_carry(fe29_sub(D, C)), Gp = fe29_add(D, C);
    const fe29 F = sel29(Fm, Gp, neg), G = sel29(Gp, Fm, neg);
    ge29 r; r.X = fe29_mul(E, F); r.Y = fe29_mul(G, H); r.T = fe29_mul(E, H); r.Z = fe29_mul(F, G);
    return r;
}
#endif

// V: 0 fe_mul, 1 fe29_mul, 2 fe_sq, 3 fe29_sq, 4 ge_madd_signed, 5 ge29_madd_signed; four independent chains per thread for 0..3
template <int V> __global__ void __launch_bounds__(256) kb(uint32_t *out /* 8 words per thread */, uint32_t iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = FE_D(), b = FE_SQRTM1(), c = FE_D2(), d = FE_ONE_MINUS_D_SQ();
    a.v[0] ^= t; b.v[1] ^= t; c.v[2] ^= t; d.v[3] ^= t;
    a.v[7] &= 0x7fffffffu; b.v[7] &= 0x7fffffffu; c.v[7] &= 0x7fffffffu; d.v[7] &= 0x7fffffffu;
    fe res;
    if (V == 0 || V == 2) {
        for (uint32_t i = 0; i < iters; i++) {
            if (V == 0) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
            else { a = fe_sq(a); b = fe_sq(b); c = fe_sq(c); d = fe_sq(d); }
        }
        res = fe_add(fe_add(a, b), fe_add(c, d));
    } else if (V == 1 || V == 3) {
        fe29 a9 = fe29_from8(a.v), b9 = fe29_from8(b.v), c9 = fe29_from8(c.v), d9 = fe29_from8(d.v);
        for (uint32_t i = 0; i < iters; i++) {
            if (V == 1) { a9 = fe29_mul(a9, b9); b9 = fe29_mul(b9, c9); c9 = fe29_mul(c9, d9); d9 = fe29_mul(d9, a9); }
            else { a9 = fe29_sq(a9); b9 = fe29_sq(b9); c9 = fe29_sq(c9); d9 = fe29_sq(d9); }
        }
        const fe29 s = fe29_carry(fe29_add(fe29_add(a9, b9), fe29_add(c9, d9)));
        fe29_to8(res.v, s);
    } else if (V == 4) {
        ge_ext p; p.X = a; p.Y = b; p.Z = fe_one(); p.T = fe_mul(a, b);       // not a curve point: the formulas are polynomial identities either way
        ge_niels q; q.ypx = c; q.ymx = d; q.t2d = fe_mul(c, d);
        for (uint32_t i = 0; i < iters; i++) p = ge_madd_signed(p, q, (i >> 1) & 1u);
        res = fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T));
    } else {
        const fe t0 = fe_mul(a, b), t1 = fe_mul(c, d);
        ge29 p; p.X = fe29_from8(a.v); p.Y = fe29_from8(b.v); p.Z = fe29_from8(fe_one().v); p.T = fe29_from8(t0.v);
        const fe29 ypx = fe29_from8(c.v), ymx = fe29_from8(d.v), t2d = fe29_from8(t1.v);
        for (uint32_t i = 0; i < iters; i++) p = ge29_madd_signed(p, ypx, ymx, t2d, (i >> 1) & 1u);
        const fe29 s = fe29_carry(fe29_add(fe29_add(p.X, p.Y), fe29_add(p.Z, p.T)));
        fe29_to8(res.v, s);
    }
    uint8_t bytes[32]; fe_tobytes(bytes, res);
    for (int k = 0; k < 8; k++) out[8 * (size_t)t + k] = (uint32_t)bytes[4 * k] | ((uint32_t)bytes[4 * k + 1] << 8) | ((uint32_t)bytes[4 * k + 2] << 16) | ((uint32_t)bytes[4 * k + 3] << 24);
#endif
}

template <int V> static double run(uint32_t *d_out, uint32_t *h_out, uint32_t blocks, uint32_t iters, const char *name, double ops_per_iter) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, d_out, 4u);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kb<V>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h_out, d_out, (size_t)blocks * 256 * 32, hipMemcpyDeviceToHost);
    const double rate = (double)blocks * 256 * iters * ops_per_iter / (ms * 1e-3);
    std::printf("{\"variant\": \"%s\", \"ms\": %.3f, \"ops_per_s\": %.4g}\n", name, ms, rate);
    return rate;
}

int main() {
    const uint32_t blocks = 256 * 16, iters = 2000;
    uint32_t *d_out; (void)hipMalloc(&d_out, (size_t)blocks * 256 * 32);
    uint32_t *h0 = new uint32_t[(size_t)blocks * 256 * 8], *h1 = new uint32_t[(size_t)blocks * 256 * 8];
    auto same = [&](const char *what) { const bool ok = std::memcmp(h0, h1, (size_t)blocks * 256 * 32) == 0; std::printf("{\"check\": \"%s\", \"equal\": %s}\n", what, ok ? "true" : "false"); return ok; };
    const double m8 = run<0>(d_out, h0, blocks, iters, "fe_mul 8x32", 4), m9 = run<1>(d_out, h1, blocks, iters, "fe29_mul", 4);
    bool ok = same("mul");
    const double s8 = run<2>(d_out, h0, blocks, iters, "fe_sq 8x32", 4), s9 = run<3>(d_out, h1, blocks, iters, "fe29_sq", 4);
    ok &= same("sq");
    const double a8 = run<4>(d_out, h0, blocks, iters / 4, "ge_madd_signed 8x32", 1), a9 = run<5>(d_out, h1, blocks, iters / 4, "ge29_madd_signed", 1);
    ok &= same("madd");
    std::printf("{\"mul_ratio\": %.3f, \"sq_ratio\": %.3f, \"madd_ratio\": %.3f, \"all_equal\": %s}\n", m9 / m8, s9 / s8, a9 / a8, ok ? "true" : "false");
    return ok ? 0 : 1;
}
