// Host-side scalar type for constraint assembly and Fiat-Shamir bookkeeping (product code, C++).
// Mirrors curve25519_dalek::scalar::Scalar as the reference uses it:
//   * 32 little-endian bytes; from_bits() keeps an UNREDUCED 255-bit value (reference src/conversions.rs:18,43);
//   * as_bytes() returns those bytes verbatim (src/utils.rs:13, src/mimc_hash/mimc.rs:80);
//   * every arithmetic result is fully reduced mod l.
// Arithmetic: 4 x 64-bit limbs, Montgomery multiplication with R = 2^256.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace bpg {

class Scalar {
public:
    uint64_t w[4];

    Scalar() : w{0, 0, 0, 0} {}
    static Scalar zero() { return Scalar(); }
    static Scalar one() { Scalar s; s.w[0] = 1; return s; }
    static Scalar from_u64(uint64_t x) { Scalar s; s.w[0] = x; return s; }
    // Scalar::from_bits: clears bit 255 only, no reduction
    static Scalar from_bits(const uint8_t b[32]) { Scalar s; std::memcpy(s.w, b, 32); s.w[3] &= 0x7fffffffffffffffULL; return s; }
    // Scalar::from_bytes_mod_order
    static Scalar from_bytes_mod_order(const uint8_t b[32]) { Scalar s; std::memcpy(s.w, b, 32); return s.reduced(); }
    // Scalar::from_bytes_mod_order_wide
    static Scalar from_wide(const uint8_t b[64]) {
        Scalar lo, hi; std::memcpy(lo.w, b, 32); std::memcpy(hi.w, b + 32, 32);
        return add_canon(mont(lo, R1()), mont(hi, RR()));
    }
    void to_bytes(uint8_t out[32]) const { std::memcpy(out, w, 32); }
    const uint8_t *as_bytes() const { return reinterpret_cast<const uint8_t *>(w); }

    bool is_canonical() const { return !geq_l(w, 0); }
    Scalar reduced() const { return mont(mont(*this, RR()), one()); }
    bool is_zero_mod_l() const { Scalar r = reduced(); return (r.w[0] | r.w[1] | r.w[2] | r.w[3]) == 0; }
    bool operator==(const Scalar &o) const { return std::memcmp(w, o.w, 32) == 0; }   // byte equality, like dalek's Eq on bytes
    bool operator!=(const Scalar &o) const { return !(*this == o); }

    // dalek reduces after add/sub even for unreduced inputs (scalar.rs impl Add)
    Scalar operator+(const Scalar &o) const { return add_canon(reduced_if_needed(), o.reduced_if_needed()); }
    Scalar operator-(const Scalar &o) const { return sub_canon(reduced_if_needed(), o.reduced_if_needed()); }
    Scalar operator-() const { return sub_canon(Scalar(), reduced_if_needed()); }
    Scalar operator*(const Scalar &o) const {
        // a*b < 2^255 * 2^255 < R*l needs one operand < l: reduce the left one if it is not
        return mont(mont(reduced_if_needed(), o), RR());
    }
    Scalar &operator+=(const Scalar &o) { *this = *this + o; return *this; }
    Scalar &operator-=(const Scalar &o) { *this = *this - o; return *this; }
    Scalar &operator*=(const Scalar &o) { *this = *this * o; return *this; }

    Scalar invert() const {   // a^(l-2)
        static const uint64_t e[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
        Scalar am = mont(reduced_if_needed(), RR()), acc = R1();
        for (int i = 252; i >= 0; i--) {
            acc = mont(acc, acc);
            if ((e[i >> 6] >> (i & 63)) & 1) acc = mont(acc, am);
        }
        return mont(acc, one());
    }

    static void batch_invert(std::vector<Scalar> &v) {
        std::vector<Scalar> pre(v.size());
        Scalar acc = one();
        for (size_t i = 0; i < v.size(); i++) { pre[i] = acc; acc = acc * v[i]; }
        Scalar inv = acc.invert();
        for (size_t i = v.size(); i-- > 0;) { Scalar t = inv * pre[i]; inv = inv * v[i]; v[i] = t; }
    }

private:
    typedef unsigned __int128 u128;
    static const uint64_t *Lw() { static const uint64_t l[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL}; return l; }
    static Scalar R1() { Scalar s; s.w[0] = 0xd6ec31748d98951dULL; s.w[1] = 0xc6ef5bf4737dcf70ULL; s.w[2] = 0xfffffffffffffffeULL; s.w[3] = 0x0fffffffffffffffULL; return s; }
    static Scalar RR() { Scalar s; s.w[0] = 0xa40611e3449c0f01ULL; s.w[1] = 0xd00e1ba768859347ULL; s.w[2] = 0xceec73d217f5be65ULL; s.w[3] = 0x0399411b7c309a3dULL; return s; }

    static bool geq_l(const uint64_t a[4], uint64_t top) {
        if (top) return true;
        const uint64_t *l = Lw();
        for (int i = 3; i >= 0; i--) { if (a[i] > l[i]) return true; if (a[i] < l[i]) return false; }
        return true;
    }
    Scalar reduced_if_needed() const { return is_canonical() ? *this : reduced(); }
    static Scalar csub(const uint64_t a[4], uint64_t top) {
        Scalar r;
        if (!geq_l(a, top)) { std::memcpy(r.w, a, 32); return r; }
        const uint64_t *l = Lw(); u128 bw = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - l[i] - (uint64_t)bw; r.w[i] = (uint64_t)d; bw = (d >> 64) & 1; }
        return r;
    }
    static Scalar add_canon(const Scalar &a, const Scalar &b) {
        uint64_t t[4]; u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)a.w[i] + b.w[i]; t[i] = (uint64_t)c; c >>= 64; }
        return csub(t, (uint64_t)c);
    }
    static Scalar sub_canon(const Scalar &a, const Scalar &b) {
        Scalar r; u128 bw = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)a.w[i] - b.w[i] - (uint64_t)bw; r.w[i] = (uint64_t)d; bw = (d >> 64) & 1; }
        if (bw) { const uint64_t *l = Lw(); u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.w[i] + l[i]; r.w[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    // a*b/R mod l, needs a*b < R*l
    static Scalar mont(const Scalar &a, const Scalar &b) {
        const uint64_t *l = Lw();
        const uint64_t ninv = 0xd2b51da312547e1bULL;
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { u128 x = (u128)a.w[j] * b.w[i] + t[j] + (uint64_t)c; t[j] = (uint64_t)x; c = x >> 64; }
            u128 x = (u128)t[4] + (uint64_t)c; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
            uint64_t m = t[0] * ninv;
            c = ((u128)m * l[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) { u128 y = (u128)m * l[j] + t[j] + (uint64_t)c; t[j - 1] = (uint64_t)y; c = y >> 64; }
            u128 y = (u128)t[4] + (uint64_t)c; t[3] = (uint64_t)y; t[4] = t[5] + (uint64_t)(y >> 64); t[5] = 0;
        }
        return csub(t, t[4]);
    }
};

}  // namespace bpg
