/* ORACLE (test infrastructure). See ge.h for provenance. */
#include "ge.h"
#include <string.h>

typedef struct { fe X, Y, Z, T; } completed;

static void completed_to_ext(ge *r, const completed *c) {
    fe_mul(&r->X, &c->X, &c->T);
    fe_mul(&r->Y, &c->Y, &c->Z);
    fe_mul(&r->Z, &c->Z, &c->T);
    fe_mul(&r->T, &c->X, &c->Y);
}

void ge_identity(ge *p) { fe_0(&p->X); fe_1(&p->Y); fe_1(&p->Z); fe_0(&p->T); }

void ge_to_pn(ge_pn *r, const ge *p) {
    fe_add(&r->YpX, &p->Y, &p->X);
    fe_sub(&r->YmX, &p->Y, &p->X);
    r->Z = p->Z;
    fe_mul(&r->T2d, &p->T, &FE_D2);
}

void ge_add_pn(ge *r, const ge *p, const ge_pn *q) {
    fe ypx, ymx, pp, mm, tt2d, zz, zz2; completed c;
    fe_add(&ypx, &p->Y, &p->X); fe_sub(&ymx, &p->Y, &p->X);
    fe_mul(&pp, &ypx, &q->YpX); fe_mul(&mm, &ymx, &q->YmX);
    fe_mul(&tt2d, &p->T, &q->T2d); fe_mul(&zz, &p->Z, &q->Z);
    fe_add(&zz2, &zz, &zz);
    fe_sub(&c.X, &pp, &mm); fe_add(&c.Y, &pp, &mm);
    fe_add(&c.Z, &zz2, &tt2d); fe_sub(&c.T, &zz2, &tt2d);
    completed_to_ext(r, &c);
}

void ge_sub_pn(ge *r, const ge *p, const ge_pn *q) {
    fe ypx, ymx, pm, mp, tt2d, zz, zz2; completed c;
    fe_add(&ypx, &p->Y, &p->X); fe_sub(&ymx, &p->Y, &p->X);
    fe_mul(&pm, &ypx, &q->YmX); fe_mul(&mp, &ymx, &q->YpX);
    fe_mul(&tt2d, &p->T, &q->T2d); fe_mul(&zz, &p->Z, &q->Z);
    fe_add(&zz2, &zz, &zz);
    fe_sub(&c.X, &pm, &mp); fe_add(&c.Y, &pm, &mp);
    fe_sub(&c.Z, &zz2, &tt2d); fe_add(&c.T, &zz2, &tt2d);
    completed_to_ext(r, &c);
}

void ge_add(ge *r, const ge *p, const ge *q) { ge_pn t; ge_to_pn(&t, q); ge_add_pn(r, p, &t); }
void ge_sub(ge *r, const ge *p, const ge *q) { ge_pn t; ge_to_pn(&t, q); ge_sub_pn(r, p, &t); }

void ge_neg(ge *r, const ge *p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }

void ge_double(ge *r, const ge *p) {
    fe xx, yy, zz2, xpy, xpy2, ypx, ymx; completed c;
    fe_sq(&xx, &p->X); fe_sq(&yy, &p->Y);
    fe_sq(&zz2, &p->Z); fe_add(&zz2, &zz2, &zz2);
    fe_add(&xpy, &p->X, &p->Y); fe_sq(&xpy2, &xpy);
    fe_add(&ypx, &yy, &xx); fe_sub(&ymx, &yy, &xx);
    fe_sub(&c.X, &xpy2, &ypx); c.Y = ypx; c.Z = ymx; fe_sub(&c.T, &zz2, &ymx);
    completed_to_ext(r, &c);
}

int ge_eq(const ge *p, const ge *q) {
    fe a, b, c, d;
    fe_mul(&a, &p->X, &q->Y); fe_mul(&b, &p->Y, &q->X);
    fe_mul(&c, &p->X, &q->X); fe_mul(&d, &p->Y, &q->Y);
    return fe_eq(&a, &b) | fe_eq(&c, &d);
}

int ge_is_identity(const ge *p) { ge id; ge_identity(&id); return ge_eq(p, &id); }

void ge_compress(uint8_t s[32], const ge *p) {
    fe u1, u2, t, inv, i1, i2, zinv, denInv, iX, iY, ench, X, Y, tmp;
    fe_add(&u1, &p->Z, &p->Y); fe_sub(&t, &p->Z, &p->Y); fe_mul(&u1, &u1, &t);
    fe_mul(&u2, &p->X, &p->Y);
    fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
    fe one; fe_1(&one);
    fe_sqrt_ratio_i(&inv, &one, &t);
    fe_mul(&i1, &inv, &u1); fe_mul(&i2, &inv, &u2);
    fe_mul(&zinv, &i1, &i2); fe_mul(&zinv, &zinv, &p->T);
    denInv = i2;
    fe_mul(&iX, &p->X, &FE_SQRTM1); fe_mul(&iY, &p->Y, &FE_SQRTM1);
    fe_mul(&ench, &i1, &FE_INVSQRT_A_MINUS_D);
    fe_mul(&tmp, &p->T, &zinv);
    unsigned rotate = (unsigned)fe_isnegative(&tmp);
    X = p->X; Y = p->Y;
    fe_cmov(&X, &iY, rotate); fe_cmov(&Y, &iX, rotate); fe_cmov(&denInv, &ench, rotate);
    fe_mul(&tmp, &X, &zinv);
    fe_cneg(&Y, (unsigned)fe_isnegative(&tmp));
    fe_sub(&tmp, &p->Z, &Y); fe_mul(&tmp, &tmp, &denInv);
    fe_abs(&tmp);
    fe_tobytes(s, &tmp);
}

int ge_decompress(ge *p, const uint8_t b[32]) {
    fe s, ss, u1, u2, u2s, v, t, I, Dx, Dy, x, y, one;
    uint8_t chk[32];
    fe_frombytes(&s, b); fe_tobytes(chk, &s);
    if (memcmp(chk, b, 32) != 0) return 0;           /* non-canonical (also catches bit 255) */
    if (fe_isnegative(&s)) return 0;
    fe_1(&one);
    fe_sq(&ss, &s);
    fe_sub(&u1, &one, &ss); fe_add(&u2, &one, &ss);
    fe_sq(&u2s, &u2);
    fe_sq(&t, &u1); fe_mul(&t, &t, &FE_D); fe_neg(&t, &t); fe_sub(&v, &t, &u2s);
    fe_mul(&t, &v, &u2s);
    int ok = fe_sqrt_ratio_i(&I, &one, &t);
    fe_mul(&Dx, &I, &u2);
    fe_mul(&Dy, &I, &Dx); fe_mul(&Dy, &Dy, &v);
    fe_add(&x, &s, &s); fe_mul(&x, &x, &Dx); fe_abs(&x);
    fe_mul(&y, &u1, &Dy);
    fe_mul(&t, &x, &y);
    if (!ok || fe_isnegative(&t) || fe_iszero(&y)) return 0;
    p->X = x; p->Y = y; fe_1(&p->Z); p->T = t;
    return 1;
}

void ge_elligator(ge *p, const fe *r0) {
    fe r, Ns, c, D, t, s, sp, Nt, ssq, one, minus_one; completed cp;
    fe_1(&one); fe_neg(&minus_one, &one);
    fe_sq(&r, r0); fe_mul(&r, &r, &FE_SQRTM1);
    fe_add(&Ns, &r, &one); fe_mul(&Ns, &Ns, &FE_ONE_MINUS_D_SQ);
    c = minus_one;
    fe_mul(&t, &FE_D, &r); fe_sub(&D, &c, &t);
    fe_add(&t, &r, &FE_D); fe_mul(&D, &D, &t);
    int was_sq = fe_sqrt_ratio_i(&s, &Ns, &D);
    fe_mul(&sp, &s, r0); fe_abs(&sp); fe_neg(&sp, &sp);
    fe_cmov(&s, &sp, (unsigned)!was_sq);
    fe_cmov(&c, &r, (unsigned)!was_sq);
    fe_sub(&t, &r, &one); fe_mul(&Nt, &c, &t); fe_mul(&Nt, &Nt, &FE_D_MINUS_ONE_SQ); fe_sub(&Nt, &Nt, &D);
    fe_sq(&ssq, &s);
    fe_add(&cp.X, &s, &s); fe_mul(&cp.X, &cp.X, &D);
    fe_mul(&cp.Z, &Nt, &FE_SQRT_AD_MINUS_ONE);
    fe_sub(&cp.Y, &one, &ssq);
    fe_add(&cp.T, &one, &ssq);
    completed_to_ext(p, &cp);
}

void ge_from_uniform_bytes(ge *p, const uint8_t b[64]) {
    fe r1, r2; ge P1, P2;
    fe_frombytes(&r1, b); fe_frombytes(&r2, b + 32);
    ge_elligator(&P1, &r1); ge_elligator(&P2, &r2);
    ge_add(p, &P1, &P2);
}

void ge_scalarmult(ge *r, const sc *k, const ge *p) {
    ge acc; ge_identity(&acc);
    ge_pn pp; ge_to_pn(&pp, p);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) ge_double(&acc, &acc);
        if ((k->v[i / 64] >> (i % 64)) & 1) { ge_add_pn(&acc, &acc, &pp); started = 1; }
    }
    *r = acc;
}

void ge_basepoint(ge *p) {
    static const uint8_t B[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                  0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge_decompress(p, B);
}
