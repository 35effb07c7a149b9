/* ORACLE (test infrastructure). GF(2^255-19) in radix 2^51; see fe.h for provenance. */
#include "fe.h"
#include <string.h>

typedef unsigned __int128 u128;
#define M51 ((1ULL << 51) - 1)

const fe FE_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};
const fe FE_D2 = {{0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL, 0x2406d9dc56dffULL}};
const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL, 0xd5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};
const fe FE_SQRT_AD_MINUS_ONE = {{0x7f6a0497b2e1bULL, 0x1836f0a97afd2ULL, 0x7d747f6be7638ULL, 0x456079e7e6498ULL, 0x376931bf2b834ULL}};
const fe FE_INVSQRT_A_MINUS_D = {{0xfdaa805d40eaULL, 0x2eb482e57d339ULL, 0x7610274bc58ULL, 0x6510b613dc8ffULL, 0x786c8905cfaffULL}};
const fe FE_ONE_MINUS_D_SQ = {{0x409c1945fc176ULL, 0x719abc6a1fc4fULL, 0x1c37f90b20684ULL, 0x6bccca55eedfULL, 0x29072a8b2b3eULL}};
const fe FE_D_MINUS_ONE_SQ = {{0x55aaa44ed4d20ULL, 0x59603c3332635ULL, 0x26d3baf4a7928ULL, 0x120a66e6997a9ULL, 0x5968b37af66c2ULL}};

void fe_0(fe *h) { memset(h, 0, sizeof *h); }
void fe_1(fe *h) { memset(h, 0, sizeof *h); h->v[0] = 1; }
void fe_copy(fe *h, const fe *f) { *h = *f; }

static inline uint64_t load64(const uint8_t *s) {
    uint64_t r; memcpy(&r, s, 8); return r;   /* little-endian host */
}

void fe_frombytes(fe *h, const uint8_t s[32]) {
    h->v[0] = load64(s) & M51;
    h->v[1] = (load64(s + 6) >> 3) & M51;
    h->v[2] = (load64(s + 12) >> 6) & M51;
    h->v[3] = (load64(s + 19) >> 1) & M51;
    h->v[4] = (load64(s + 24) >> 12) & M51;
}

/* carry chain leaving every limb < 2^51 + tiny */
static inline void fe_weak(fe *h) {
    uint64_t c;
    c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
    c = h->v[1] >> 51; h->v[1] &= M51; h->v[2] += c;
    c = h->v[2] >> 51; h->v[2] &= M51; h->v[3] += c;
    c = h->v[3] >> 51; h->v[3] &= M51; h->v[4] += c;
    c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += c * 19;
}

void fe_tobytes(uint8_t s[32], const fe *f) {
    fe h = *f;
    fe_weak(&h); fe_weak(&h);
    /* q = 1 iff h >= p */
    uint64_t q = (h.v[0] + 19) >> 51;
    q = (h.v[1] + q) >> 51; q = (h.v[2] + q) >> 51; q = (h.v[3] + q) >> 51; q = (h.v[4] + q) >> 51;
    h.v[0] += 19 * q;
    uint64_t c;
    c = h.v[0] >> 51; h.v[0] &= M51; h.v[1] += c;
    c = h.v[1] >> 51; h.v[1] &= M51; h.v[2] += c;
    c = h.v[2] >> 51; h.v[2] &= M51; h.v[3] += c;
    c = h.v[3] >> 51; h.v[3] &= M51; h.v[4] += c;
    h.v[4] &= M51;
    uint64_t w0 = h.v[0] | (h.v[1] << 51);
    uint64_t w1 = (h.v[1] >> 13) | (h.v[2] << 38);
    uint64_t w2 = (h.v[2] >> 26) | (h.v[3] << 25);
    uint64_t w3 = (h.v[3] >> 39) | (h.v[4] << 12);
    memcpy(s, &w0, 8); memcpy(s + 8, &w1, 8); memcpy(s + 16, &w2, 8); memcpy(s + 24, &w3, 8);
}

void fe_add(fe *h, const fe *f, const fe *g) {
    for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
    fe_weak(h);
}

void fe_sub(fe *h, const fe *f, const fe *g) {
    /* f + 16p - g keeps every limb positive for g limbs < 2^54 */
    h->v[0] = f->v[0] + 36028797018963664ULL - g->v[0];
    h->v[1] = f->v[1] + 36028797018963952ULL - g->v[1];
    h->v[2] = f->v[2] + 36028797018963952ULL - g->v[2];
    h->v[3] = f->v[3] + 36028797018963952ULL - g->v[3];
    h->v[4] = f->v[4] + 36028797018963952ULL - g->v[4];
    fe_weak(h);
}

void fe_neg(fe *h, const fe *f) { fe z; fe_0(&z); fe_sub(h, &z, f); }

void fe_mul(fe *h, const fe *f, const fe *g) {
    const uint64_t *a = f->v, *b = g->v;
    uint64_t b1 = b[1] * 19, b2 = b[2] * 19, b3 = b[3] * 19, b4 = b[4] * 19;
    u128 c0 = (u128)a[0] * b[0] + (u128)a[4] * b1 + (u128)a[3] * b2 + (u128)a[2] * b3 + (u128)a[1] * b4;
    u128 c1 = (u128)a[1] * b[0] + (u128)a[0] * b[1] + (u128)a[4] * b2 + (u128)a[3] * b3 + (u128)a[2] * b4;
    u128 c2 = (u128)a[2] * b[0] + (u128)a[1] * b[1] + (u128)a[0] * b[2] + (u128)a[4] * b3 + (u128)a[3] * b4;
    u128 c3 = (u128)a[3] * b[0] + (u128)a[2] * b[1] + (u128)a[1] * b[2] + (u128)a[0] * b[3] + (u128)a[4] * b4;
    u128 c4 = (u128)a[4] * b[0] + (u128)a[3] * b[1] + (u128)a[2] * b[2] + (u128)a[1] * b[3] + (u128)a[0] * b[4];
    c1 += (uint64_t)(c0 >> 51); uint64_t r0 = (uint64_t)c0 & M51;
    c2 += (uint64_t)(c1 >> 51); uint64_t r1 = (uint64_t)c1 & M51;
    c3 += (uint64_t)(c2 >> 51); uint64_t r2 = (uint64_t)c2 & M51;
    c4 += (uint64_t)(c3 >> 51); uint64_t r3 = (uint64_t)c3 & M51;
    uint64_t carry = (uint64_t)(c4 >> 51); uint64_t r4 = (uint64_t)c4 & M51;
    r0 += carry * 19;
    r1 += r0 >> 51; r0 &= M51;
    h->v[0] = r0; h->v[1] = r1; h->v[2] = r2; h->v[3] = r3; h->v[4] = r4;
}

void fe_sq(fe *h, const fe *f) { fe_mul(h, f, f); }

static void fe_sqn(fe *h, const fe *f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

/* t19 = z^(2^250-1), t3 = z^11 : shared prefix of invert and pow22523 */
static void fe_pow22501(fe *t19, fe *t3, const fe *z) {
    fe t0, t1, t2, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15, t16, t17, t18;
    fe_sq(&t0, z);                 /* 2 */
    fe_sqn(&t1, &t0, 2);           /* 8 */
    fe_mul(&t2, z, &t1);           /* 9 */
    fe_mul(t3, &t0, &t2);          /* 11 */
    fe_sq(&t4, t3);                /* 22 */
    fe_mul(&t5, &t2, &t4);         /* 31 = 2^5-1 */
    fe_sqn(&t6, &t5, 5);
    fe_mul(&t7, &t6, &t5);         /* 2^10-1 */
    fe_sqn(&t8, &t7, 10);
    fe_mul(&t9, &t8, &t7);         /* 2^20-1 */
    fe_sqn(&t10, &t9, 20);
    fe_mul(&t11, &t10, &t9);       /* 2^40-1 */
    fe_sqn(&t12, &t11, 10);
    fe_mul(&t13, &t12, &t7);       /* 2^50-1 */
    fe_sqn(&t14, &t13, 50);
    fe_mul(&t15, &t14, &t13);      /* 2^100-1 */
    fe_sqn(&t16, &t15, 100);
    fe_mul(&t17, &t16, &t15);      /* 2^200-1 */
    fe_sqn(&t18, &t17, 50);
    fe_mul(t19, &t18, &t13);       /* 2^250-1 */
}

void fe_invert(fe *out, const fe *z) {
    fe t19, t3, t20;
    fe_pow22501(&t19, &t3, z);
    fe_sqn(&t20, &t19, 5);         /* 2^255-32 */
    fe_mul(out, &t20, &t3);        /* 2^255-21 */
}

void fe_pow22523(fe *out, const fe *z) {
    fe t19, t3, t20;
    fe_pow22501(&t19, &t3, z);
    fe_sqn(&t20, &t19, 2);         /* 2^252-4 */
    fe_mul(out, z, &t20);          /* 2^252-3 */
}

int fe_isnegative(const fe *f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }

int fe_iszero(const fe *f) {
    uint8_t s[32]; fe_tobytes(s, f);
    uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i];
    return r == 0;
}

int fe_eq(const fe *f, const fe *g) {
    uint8_t a[32], b[32]; fe_tobytes(a, f); fe_tobytes(b, g);
    uint8_t r = 0; for (int i = 0; i < 32; i++) r |= a[i] ^ b[i];
    return r == 0;
}

void fe_cmov(fe *f, const fe *g, unsigned b) {
    uint64_t m = (uint64_t)0 - (uint64_t)(b & 1);
    for (int i = 0; i < 5; i++) f->v[i] ^= m & (f->v[i] ^ g->v[i]);
}

void fe_cneg(fe *f, unsigned b) { fe n; fe_neg(&n, f); fe_cmov(f, &n, b); }
void fe_abs(fe *f) { fe_cneg(f, (unsigned)fe_isnegative(f)); }

int fe_sqrt_ratio_i(fe *r, const fe *u, const fe *v) {
    fe v3, v7, t, rr, check, negu, negui, rp;
    fe_sq(&v3, v); fe_mul(&v3, &v3, v);
    fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
    fe_mul(&t, u, &v7); fe_pow22523(&t, &t);
    fe_mul(&rr, u, &v3); fe_mul(&rr, &rr, &t);
    fe_sq(&check, &rr); fe_mul(&check, &check, v);
    fe_neg(&negu, u); fe_mul(&negui, &negu, &FE_SQRTM1);
    int correct = fe_eq(&check, u);
    int flipped = fe_eq(&check, &negu);
    int flipped_i = fe_eq(&check, &negui);
    fe_mul(&rp, &rr, &FE_SQRTM1);
    fe_cmov(&rr, &rp, (unsigned)(flipped | flipped_i));
    fe_abs(&rr);
    *r = rr;
    return correct | flipped;
}
