// FETCH_SIZE calibration for the access pattern of the bucket sweep (k_bucket_chunks): every lane reads whole 96-byte rows (six 16-byte
// loads) at uniformly random positions of a 201 MB table - the affine-Niels generator table at capacity 2^20.  The guide
// (MI355X_MICROARCH.md, HBM) calibrates FETCH_SIZE only for wide coalesced reads (x2 on gfx950) and says: calibrate other shapes on a
// known byte count.  This program launches
//   k_calib_stream    : 16 B per lane, lane-contiguous, the whole table once            known bytes = table bytes
//   k_calib_gather96  : ROWS random 96-byte rows, one row per lane per iteration          known bytes = ROWS * 96
//   k_calib_gather128 : the same rows padded to 128 bytes (128-byte aligned), 96 B read   known bytes = ROWS * 96 (128 B lines touched: ROWS)
// and prints the known byte counts as JSON; run it under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and feed both to
// tools/calib/fetch_factor.py, which writes profiles/<tag>_fetch_calibration.json (factor = known bytes / (FETCH_SIZE * 1024)).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) k_calib_fill(uint4 *t, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_calib_stream(const uint4 *__restrict__ t, size_t n16, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// STRIDE16 = row pitch in 16-byte units (6: packed 96-byte rows, 8: rows padded to 128 bytes); six loads per row either way
template <int STRIDE16>
__global__ void __launch_bounds__(256) k_calib_gather(const uint4 *__restrict__ t, uint32_t nrows, uint32_t iters, uint32_t *__restrict__ out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < iters; k++) {
        const uint32_t r = mix(tid * 2654435761u + k * 40503u + 12345u) % nrows;
        const uint4 *p = t + (size_t)r * STRIDE16;
#pragma unroll
        for (int j = 0; j < 6; j++) { const uint4 v = p[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    // argv[1] = table size in units of the 2^21-row generator table (default 1; 17 = the 3.4 GB a pre-shifted table per window would take:
    // the gather rate once the table no longer fits the 256 MB Infinity Cache)
    const uint32_t mult = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1u;
    const uint32_t nrows = (mult ? mult : 1u) << 21;                // 2 * 2^20 points per unit
    const size_t bytes96 = (size_t)nrows * 96, bytes128 = (size_t)nrows * 128;
    uint4 *t96, *t128; uint32_t *out;
    const uint32_t blocks = 256 * 32, threads = 256, iters = 8;     // 2^21 lanes * 8 rows = 16.8 M rows
    CHK(hipMalloc(&t96, bytes96)); CHK(hipMalloc(&t128, bytes128)); CHK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipLaunchKernelGGL(k_calib_fill, dim3(4096), dim3(256), 0, 0, t96, bytes96 / 16);
    hipLaunchKernelGGL(k_calib_fill, dim3(4096), dim3(256), 0, 0, t128, bytes128 / 16);
    CHK(hipDeviceSynchronize());
    hipEvent_t e[8]; for (auto &x : e) CHK(hipEventCreate(&x));
    float ms[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; rep++) {                              // three launches each: the PMC pass averages them
        CHK(hipEventRecord(e[0], 0));
        hipLaunchKernelGGL(k_calib_stream, dim3(blocks), dim3(threads), 0, 0, t96, bytes96 / 16, out);
        CHK(hipEventRecord(e[1], 0));
        hipLaunchKernelGGL(k_calib_gather<6>, dim3(blocks), dim3(threads), 0, 0, t96, nrows, iters, out);
        CHK(hipEventRecord(e[2], 0));
        hipLaunchKernelGGL(k_calib_gather<8>, dim3(blocks), dim3(threads), 0, 0, t128, nrows, iters, out);
        CHK(hipEventRecord(e[3], 0));
        CHK(hipDeviceSynchronize());
        for (int k = 0; k < 3; k++) CHK(hipEventElapsedTime(&ms[k], e[k], e[k + 1]));
    }
    const double rows = (double)blocks * threads * iters;
    std::printf("{\"table_bytes\": %zu, \"rows_gathered\": %.0f, \"known_bytes\": {\"k_calib_stream\": %zu, \"k_calib_gather<6>\": %.0f, \"k_calib_gather<8>\": %.0f}, "
                "\"ms\": {\"k_calib_stream\": %.4f, \"k_calib_gather<6>\": %.4f, \"k_calib_gather<8>\": %.4f}, "
                "\"GBps\": {\"k_calib_stream\": %.1f, \"k_calib_gather<6>\": %.1f, \"k_calib_gather<8>\": %.1f}}\n",
                bytes96, rows, bytes96, rows * 96, rows * 96, ms[0], ms[1], ms[2], bytes96 / ms[0] / 1e6, rows * 96 / ms[1] / 1e6, rows * 96 / ms[2] / 1e6);
    return 0;
}
