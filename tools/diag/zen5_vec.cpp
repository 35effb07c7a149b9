// Port microbenchmark for the host Keccak chain: throughput (ops/ns) of the vector operations the lanes-in-XMM permutation is made of,
// of the scalar ones, and of both together, on the core it runs on.   g++ -O2 -mavx512f -mavx512vl -mbmi2 zen5_vec.cpp && ./a.out
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <immintrin.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define REP8(x) x x x x x x x x
int main() {
    const long N = 20000000;
    __m128i a0 = _mm_set1_epi64x(1), a1 = _mm_set1_epi64x(2), a2 = _mm_set1_epi64x(3), a3 = _mm_set1_epi64x(4), a4 = _mm_set1_epi64x(5), a5 = _mm_set1_epi64x(6),
            a6 = _mm_set1_epi64x(7), a7 = _mm_set1_epi64x(8), k = _mm_set1_epi64x(0x123456789abcdefLL);
    uint64_t g0 = 1, g1 = 2, g2 = 3, g3 = 4, g4 = 5, g5 = 6, g6 = 7, g7 = 8, gk = 0x9e3779b97f4a7c15ULL;
    double t0, t1;
#define VEC8(OP) asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#define TERN(r) "vpternlogq $0x96, %8, %8, " #r "\n\t"
#define ROL(r) "vprolq $13, " #r ", " #r "\n\t"
#define XOR(r) "vpxorq %8, " #r ", " #r "\n\t"
    t0 = now(); for (long i = 0; i < N; i++) { VEC8(TERN) } t1 = now(); printf("vpternlogq xmm  %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { VEC8(ROL) } t1 = now(); printf("vprolq xmm      %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { VEC8(XOR) } t1 = now(); printf("vpxorq xmm      %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { VEC8(TERN) VEC8(ROL) VEC8(TERN) } t1 = now(); printf("2 tern : 1 rol   %.2f ops/ns\n", 24.0 * N / (t1 - t0) / 1e9);
#define GPR8(OP) asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7) : "+r"(g0), "+r"(g1), "+r"(g2), "+r"(g3), "+r"(g4), "+r"(g5), "+r"(g6), "+r"(g7) : "r"(gk));
#define GXOR(r) "xorq %8, " #r "\n\t"
#define GROR(r) "rorxq $13, " #r ", " #r "\n\t"
#define GANDN(r) "andnq %8, " #r ", " #r "\n\t"
    t0 = now(); for (long i = 0; i < N; i++) { GPR8(GXOR) } t1 = now(); printf("xor gpr         %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { GPR8(GROR) } t1 = now(); printf("rorx gpr        %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { GPR8(GANDN) } t1 = now(); printf("andn gpr        %.2f ops/ns\n", 8.0 * N / (t1 - t0) / 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { VEC8(TERN) GPR8(GXOR) VEC8(ROL) GPR8(GROR) VEC8(TERN) GPR8(GANDN) } t1 = now();
    printf("vector + gpr interleaved: %.2f vector ops/ns + %.2f gpr ops/ns\n", 24.0 * N / (t1 - t0) / 1e9, 24.0 * N / (t1 - t0) / 1e9);
    // dependent chains: latency
    t0 = now(); for (long i = 0; i < N; i++) { asm volatile(REP8("vpternlogq $0x96, %1, %1, %0\n\t") : "+v"(a0) : "v"(k)); } t1 = now(); printf("vpternlogq latency %.2f ns\n", (t1 - t0) / (8.0 * N) * 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { asm volatile(REP8("vprolq $13, %0, %0\n\t") : "+v"(a0)); } t1 = now(); printf("vprolq latency     %.2f ns\n", (t1 - t0) / (8.0 * N) * 1e9);
    t0 = now(); for (long i = 0; i < N; i++) { asm volatile(REP8("rorxq $13, %0, %0\n\t") : "+r"(g0)); } t1 = now(); printf("rorx latency       %.2f ns\n", (t1 - t0) / (8.0 * N) * 1e9);
    t0 = now(); for (long i = 0; i < N / 4; i++) { asm volatile(REP8("vmovq %0, %1\n\tvmovq %1, %0\n\t") : "+r"(g0), "+v"(a0)); } t1 = now(); printf("gpr->xmm->gpr round trip %.2f ns\n", (t1 - t0) / (8.0 * N / 4) * 1e9);
    printf("(%llu %lld)\n", (unsigned long long)(g0 ^ g1 ^ g2 ^ g3 ^ g4 ^ g5 ^ g6 ^ g7), (long long)_mm_cvtsi128_si64(_mm_xor_si128(_mm_xor_si128(a0, a1), _mm_xor_si128(a2, _mm_xor_si128(a3, _mm_xor_si128(a4, _mm_xor_si128(a5, _mm_xor_si128(a6, a7))))))));
    return 0;
}
