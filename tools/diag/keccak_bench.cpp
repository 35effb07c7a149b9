// Host microbenchmark: ns per Keccak-f[1600] (scalar / AVX-512) and per TranscriptRng draw, on the CPU it runs on.
#include <chrono>
#include <cstdio>
#include <vector>
#include "../../bulletproofs_gadgets_amd/csrc/host/merlin.hpp"
using namespace bpg;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint64_t st[25]; for (int i = 0; i < 25; i++) st[i] = 0x9e3779b97f4a7c15ULL * (i + 1);
    const int N = 2000000;
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); for (int i = 0; i < N; i++) keccak_f1600_scalar(st); double t1 = now();
        printf("scalar   %.1f ns/perm  (%llx)\n", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
#if defined(__x86_64__)
        if (__builtin_cpu_supports("avx512f")) { t0 = now(); for (int i = 0; i < N; i++) keccak_f1600_avx512(st); t1 = now();
            printf("avx512   %.1f ns/perm  (%llx)\n", (t1 - t0) / N * 1e9, (unsigned long long)st[0]); }
#endif
    }
    Transcript T(std::string("bench"));
    std::vector<Scalar> vb; uint8_t seed[32] = {1};
    TranscriptRng rng = T.build_rng(vb, seed);
    uint8_t buf[64]; uint64_t acc = 0;
    double t0 = now(); for (int i = 0; i < N; i++) { rng.fill_bytes(buf, 64); acc += buf[0]; } double t1 = now();
    printf("rng draw (fill_bytes 64) %.1f ns  (%llu)  dispatch=%s\n", (t1 - t0) / N * 1e9, (unsigned long long)acc, keccak_have_avx512() ? "avx512" : "scalar");
    t0 = now(); for (int i = 0; i < N; i++) { Scalar s = rng.random_scalar(); acc += s.w[0]; } t1 = now();
    printf("rng random_scalar        %.1f ns  (%llu)\n", (t1 - t0) / N * 1e9, (unsigned long long)acc);
    return 0;
}
