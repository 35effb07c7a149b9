// verifier kernels - part of kernels.cuh (included from there, in this order; see its header for the kernel map and the data layout)
#pragma once

namespace bpg {

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

}  // namespace bpg
