// Throughput of the Keccak vector operations at 128 / 256 / 512 bits on the core this runs on: does a 512-bit VPTERNLOGQ / VPROLQ cost what a 128-bit one
// does?  (If so, eight independent TranscriptRng chains in the eight 64-bit lanes of ZMM registers cost what one chain costs in XMM registers.)
//   g++ -O2 -mavx512f -mavx512vl zen5_wide.cpp && ./a.out
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <immintrin.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define RUN(NAME, TYPE, INIT, TERN, ROL, SINK) { \
    TYPE a0 = INIT(1), a1 = INIT(2), a2 = INIT(3), a3 = INIT(4), a4 = INIT(5), a5 = INIT(6), a6 = INIT(7), a7 = INIT(8), k = INIT(0x123456789abcdefLL); \
    const long N = 20000000; double t0, t1; \
    t0 = now(); for (long i = 0; i < N; i++) { a0 = TERN(a0, k, k); a1 = TERN(a1, k, k); a2 = TERN(a2, k, k); a3 = TERN(a3, k, k); a4 = TERN(a4, k, k); a5 = TERN(a5, k, k); a6 = TERN(a6, k, k); a7 = TERN(a7, k, k); \
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); } t1 = now(); \
    std::printf("%s ternlog  %.2f ops/ns\n", NAME, 8.0 * N / (t1 - t0) / 1e9); \
    t0 = now(); for (long i = 0; i < N; i++) { a0 = ROL(a0); a1 = ROL(a1); a2 = ROL(a2); a3 = ROL(a3); a4 = ROL(a4); a5 = ROL(a5); a6 = ROL(a6); a7 = ROL(a7); \
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); } t1 = now(); \
    std::printf("%s rol      %.2f ops/ns\n", NAME, 8.0 * N / (t1 - t0) / 1e9); \
    t0 = now(); for (long i = 0; i < N; i++) { a0 = TERN(a0, k, k); a1 = ROL(a1); a2 = TERN(a2, k, k); a3 = TERN(a3, k, k); a4 = ROL(a4); a5 = TERN(a5, k, k); a6 = TERN(a6, k, k); a7 = TERN(a7, k, k); \
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); } t1 = now(); \
    std::printf("%s 3 tern : 1 rol  %.2f ops/ns\n", NAME, 8.0 * N / (t1 - t0) / 1e9); \
    sink ^= SINK(a0) ^ SINK(a1) ^ SINK(a2) ^ SINK(a3) ^ SINK(a4) ^ SINK(a5) ^ SINK(a6) ^ SINK(a7); }
int main() {
    long long sink = 0;
    RUN("xmm", __m128i, _mm_set1_epi64x, [](__m128i a, __m128i b, __m128i c) { return _mm_ternarylogic_epi64(a, b, c, 0x96); }, [](__m128i a) { return _mm_rol_epi64(a, 13); }, _mm_cvtsi128_si64)
    RUN("ymm", __m256i, _mm256_set1_epi64x, [](__m256i a, __m256i b, __m256i c) { return _mm256_ternarylogic_epi64(a, b, c, 0x96); }, [](__m256i a) { return _mm256_rol_epi64(a, 13); }, [](__m256i a) { return _mm_cvtsi128_si64(_mm256_castsi256_si128(a)); })
    RUN("zmm", __m512i, _mm512_set1_epi64, [](__m512i a, __m512i b, __m512i c) { return _mm512_ternarylogic_epi64(a, b, c, 0x96); }, [](__m512i a) { return _mm512_rol_epi64(a, 13); }, [](__m512i a) { return _mm_cvtsi128_si64(_mm512_castsi512_si128(a)); })
    std::printf("(%lld)\n", sink);
    return 0;
}
