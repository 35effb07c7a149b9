/* ORACLE (test infrastructure). Arithmetic mod l with 4x64 Montgomery (R = 2^256); see sc.h. */
#include "sc.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

static const uint64_t Lq[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
static const sc SC_R  = {{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}};
static const sc SC_RR = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};
#define LFACTOR 0xd2b51da312547e1bULL
const sc SC_ZERO = {{0, 0, 0, 0}};
const sc SC_ONE = {{1, 0, 0, 0}};

/* r = a - l if a >= l (a given with an extra top limb), else a */
static void cond_sub_l(sc *r, const uint64_t a[4], uint64_t top) {
    uint64_t t[4]; u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - Lq[i] - (uint64_t)bw;
        t[i] = (uint64_t)d; bw = (d >> 64) & 1;
    }
    /* borrow out means a < l (when top == 0) */
    uint64_t keep = (top == 0 && bw) ? 1 : 0;
    for (int i = 0; i < 4; i++) r->v[i] = keep ? a[i] : t[i];
}

/* Montgomery product a*b/R mod l; requires a*b < R*l; result canonical */
static void mont_mul(sc *r, const sc *a, const sc *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 x = (u128)a->v[j] * b->v[i] + t[j] + (uint64_t)carry;
            t[j] = (uint64_t)x; carry = x >> 64;
        }
        u128 x = (u128)t[4] + (uint64_t)carry; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * LFACTOR;
        carry = ((u128)m * Lq[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            u128 y = (u128)m * Lq[j] + t[j] + (uint64_t)carry;
            t[j - 1] = (uint64_t)y; carry = y >> 64;
        }
        u128 y = (u128)t[4] + (uint64_t)carry; t[3] = (uint64_t)y;
        t[4] = t[5] + (uint64_t)(y >> 64); t[5] = 0;
    }
    cond_sub_l(r, t, t[4]);
}

void sc_frombytes_raw(sc *r, const uint8_t s[32]) { memcpy(r->v, s, 32); }
void sc_tobytes(uint8_t s[32], const sc *a) { memcpy(s, a->v, 32); }
void sc_from_u64(sc *r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }

void sc_reduce(sc *r, const sc *a) {
    sc t; mont_mul(&t, a, &SC_RR);     /* a*R mod l   (a < 2^256, RR < l) */
    mont_mul(r, &t, &SC_ONE);          /* a mod l */
}

void sc_frombytes_mod_order(sc *r, const uint8_t s[32]) { sc t; sc_frombytes_raw(&t, s); sc_reduce(r, &t); }

void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t t[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
    cond_sub_l(r, t, (uint64_t)c);
}

void sc_sub(sc *r, const sc *a, const sc *b) {
    uint64_t t[4]; u128 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)bw;
        t[i] = (uint64_t)d; bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)t[i] + Lq[i]; t[i] = (uint64_t)c; c >>= 64; }
    }
    memcpy(r->v, t, 32);
}

void sc_neg(sc *r, const sc *a) { sc_sub(r, &SC_ZERO, a); }

void sc_mul(sc *r, const sc *a, const sc *b) {
    /* dalek Scalar::mul: montgomery_reduce(a*b) then montgomery_reduce(ab * RR); needs a*b < R*l,
       true for canonical inputs (every oracle entry point reduces what it ingests) */
    sc t; mont_mul(&t, a, b); mont_mul(r, &t, &SC_RR);
}

#define sc_mul_fast sc_mul

void sc_muladd(sc *r, const sc *a, const sc *b, const sc *c) { sc t; sc_mul_fast(&t, a, b); sc_add(r, &t, c); }

void sc_frombytes_wide(sc *r, const uint8_t s[64]) {
    /* dalek: lo*R/R + hi*R^2/R = lo + hi*2^256 (mod l) */
    sc lo, hi, a, b;
    memcpy(lo.v, s, 32); memcpy(hi.v, s + 32, 32);
    mont_mul(&a, &lo, &SC_R);
    mont_mul(&b, &hi, &SC_RR);
    sc_add(r, &a, &b);
}

void sc_invert(sc *r, const sc *a) {
    /* a^(l-2), Montgomery ladder of squarings in the Montgomery domain */
    static const uint64_t e[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
    sc am, acc;
    mont_mul(&am, a, &SC_RR);
    acc = SC_R;
    for (int i = 252; i >= 0; i--) {
        mont_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) mont_mul(&acc, &acc, &am);
    }
    mont_mul(r, &acc, &SC_ONE);
}

int sc_iszero(const sc *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int sc_eq(const sc *a, const sc *b) { return memcmp(a->v, b->v, 32) == 0; }

void sc_batch_invert(sc *v, size_t n) {
    if (!n) return;
    sc *pre = (sc *)malloc(n * sizeof(sc));
    sc acc = SC_ONE;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; sc_mul_fast(&acc, &acc, &v[i]); }
    sc inv; sc_invert(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        sc t; sc_mul_fast(&t, &inv, &pre[i]);
        sc_mul_fast(&inv, &inv, &v[i]);
        v[i] = t;
    }
    free(pre);
}
