/* ORACLE (test infrastructure, not product code).
 * Field GF(2^255-19), radix 2^51, unsigned __int128 products.
 * Restates curve25519-dalek 1.x backend/serial/u64/field.rs (crate not vendored under
 * /root/reference; pinned by Cargo.toml:8 `curve25519-dalek = "1"`). */
#ifndef ORACLE_FE_H
#define ORACLE_FE_H
#include <stdint.h>

typedef struct { uint64_t v[5]; } fe;

void fe_0(fe *h);
void fe_1(fe *h);
void fe_copy(fe *h, const fe *f);
void fe_frombytes(fe *h, const uint8_t s[32]);      /* ignores bit 255 */
void fe_tobytes(uint8_t s[32], const fe *h);        /* canonical encoding */
void fe_add(fe *h, const fe *f, const fe *g);
void fe_sub(fe *h, const fe *f, const fe *g);
void fe_neg(fe *h, const fe *f);
void fe_mul(fe *h, const fe *f, const fe *g);
void fe_sq(fe *h, const fe *f);
void fe_invert(fe *out, const fe *z);
void fe_pow22523(fe *out, const fe *z);             /* z^((p-5)/8) */
int  fe_isnegative(const fe *f);
int  fe_iszero(const fe *f);
int  fe_eq(const fe *f, const fe *g);
void fe_cmov(fe *f, const fe *g, unsigned b);       /* f = b ? g : f, constant time */
void fe_cneg(fe *f, unsigned b);
void fe_abs(fe *f);
/* r = sqrt(u/v) if square, else sqrt(i*u/v); returns 1 when u/v was square (RFC 9496 SQRT_RATIO_M1) */
int  fe_sqrt_ratio_i(fe *r, const fe *u, const fe *v);

extern const fe FE_D, FE_D2, FE_SQRTM1, FE_SQRT_AD_MINUS_ONE, FE_INVSQRT_A_MINUS_D,
                FE_ONE_MINUS_D_SQ, FE_D_MINUS_ONE_SQ;
#endif
