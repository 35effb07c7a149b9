// Ristretto255 group arithmetic for gfx950 on top of fe.cuh: extended twisted-Edwards points (a = -1),
// HALVED affine "Niels" operands ((y+x)/2, (y-x)/2, dxy; 96 bytes) for the generator tables, RFC 9496 encode and one-way map.
// Halved (round 5): with the usual (y+x, y-x, 2dxy) a mixed addition needs D = 2 Z1; every product of the addition is linear in the operand, so
// halving the operand halves A, B, C and with D = Z1 the result is (X3, Y3, Z3, T3) / 4 - the same point, one field addition (19 instructions of
// ~1,640 per bucket entry) less.  The halves cost nothing to make: the batched normalisation folds 1/2 into its one inversion (k_normalize_niels).
// Replaces curve25519-dalek RistrettoPoint / EdwardsPoint (not vendored; reference Cargo.toml:8) under
// PedersenGens::commit, BulletproofGens::new and Prover::prove (reference src/bin/prover.rs:53,92-93).
#pragma once
#include "fe.cuh"

namespace bpg {

struct ge_ext { fe X, Y, Z, T; };
struct ge_niels { fe ypx, ymx, t2d; };      // affine, halved: (y + x) / 2, (y - x) / 2, d x y

BPG_HD ge_ext ge_identity() { ge_ext r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }
BPG_HD ge_niels ge_niels_identity() { ge_niels r; r.ypx = FE_INV2(); r.ymx = FE_INV2(); r.t2d = fe_zero(); return r; }

// mixed addition, 7M (halved operand: D = Z1, the result comes out scaled by 1/4)
BPG_HD ge_ext ge_madd(const ge_ext &p, const ge_niels &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe B = fe_mul(fe_add(p.Y, p.X), q.ypx);
    fe C = fe_mul(p.T, q.t2d);
    const fe &D = p.Z;
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge_ext r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
BPG_HD ge_ext ge_msub(const ge_ext &p, const ge_niels &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), q.ypx);
    fe B = fe_mul(fe_add(p.Y, p.X), q.ymx);
    fe C = fe_mul(p.T, q.t2d);
    const fe &D = p.Z;
    fe E = fe_sub(B, A), F = fe_add(D, C), G = fe_sub(D, C), H = fe_add(B, A);
    ge_ext r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
// sign-selected mixed addition without divergence: neg in {0,1}
BPG_HD ge_ext ge_madd_signed(const ge_ext &p, const ge_niels &q, uint32_t neg) {
    ge_niels s;
    s.ypx = fe_select(q.ypx, q.ymx, neg);
    s.ymx = fe_select(q.ymx, q.ypx, neg);
    s.t2d = fe_cneg(q.t2d, neg);
    return ge_madd(p, s);
}

// full addition, 9M
BPG_HD ge_ext ge_add(const ge_ext &p, const ge_ext &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, q.T), FE_D2());
    fe D = fe_mul(p.Z, q.Z); D = fe_add(D, D);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge_ext r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
BPG_HD ge_ext ge_neg(const ge_ext &p) { ge_ext r; r.X = fe_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fe_neg(p.T); return r; }

// doubling, 4S + 4M
BPG_HD ge_ext ge_dbl(const ge_ext &p) {
    fe XX = fe_sq(p.X), YY = fe_sq(p.Y), ZZ2 = fe_sq(p.Z); ZZ2 = fe_add(ZZ2, ZZ2);
    fe S = fe_sq(fe_add(p.X, p.Y));
    fe YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX);
    fe cX = fe_sub(S, YpX), cT = fe_sub(ZZ2, YmX);       // completed point (cX, YpX, YmX, cT)
    ge_ext r; r.X = fe_mul(cX, cT); r.Y = fe_mul(YpX, YmX); r.Z = fe_mul(YmX, cT); r.T = fe_mul(cX, YpX);
    return r;
}

// projective Niels operand (Y+X, Y-X, Z, 2dT; 128 bytes): table entries that are never normalised (IPA tail tables)
struct ge_pniels { fe ypx, ymx, Z, t2d; };
BPG_HD ge_pniels ge_to_pniels(const ge_ext &p) {
    ge_pniels r; r.ypx = fe_add(p.Y, p.X); r.ymx = fe_sub(p.Y, p.X); r.Z = p.Z; r.t2d = fe_mul(p.T, FE_D2());
    return r;
}
// extended + projective Niels, 8M; neg in {0,1} subtracts instead (operand swap, no divergence)
BPG_HD ge_ext ge_add_pniels_signed(const ge_ext &p, const ge_pniels &q, uint32_t neg) {
    fe qp = fe_select(q.ypx, q.ymx, neg), qm = fe_select(q.ymx, q.ypx, neg), qt = fe_cneg(q.t2d, neg);
    fe A = fe_mul(fe_sub(p.Y, p.X), qm);
    fe B = fe_mul(fe_add(p.Y, p.X), qp);
    fe C = fe_mul(p.T, qt);
    fe D = fe_mul(p.Z, q.Z); D = fe_add(D, D);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge_ext r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}

// the affine (halved) Niels point as an extended point: x = (y+x)/2 - (y-x)/2, y likewise, z = 1, t = x y = (d x y) / d: one product
BPG_HD ge_ext ge_from_niels(const ge_niels &q) {
    ge_ext r; r.X = fe_sub(q.ypx, q.ymx); r.Y = fe_add(q.ypx, q.ymx); r.Z = fe_one(); r.T = fe_mul(q.t2d, FE_INV_D());
    return r;
}
// +-q as an extended point without an addition to the identity: (Y+X) -+ (Y-X) = 2X, 2Y, 2Z and 2dT / d = 2T - the point (2X : 2Y : 2Z : 2T), one product
BPG_HD ge_ext ge_from_pniels_signed(const ge_pniels &q, uint32_t neg) {
    ge_ext r; r.X = fe_cneg(fe_sub(q.ypx, q.ymx), neg); r.Y = fe_add(q.ypx, q.ymx); r.Z = fe_add(q.Z, q.Z); r.T = fe_cneg(fe_mul(q.t2d, FE_INV_D()), neg);
    return r;
}

// extended -> halved affine Niels given 1 / (2 Z): x/2 = X / (2Z), y/2 = Y / (2Z), d x y = 4d (x/2)(y/2)
BPG_HD ge_niels ge_to_niels_halfinv(const ge_ext &p, const fe &zinv_half) {
    fe x = fe_mul(p.X, zinv_half), y = fe_mul(p.Y, zinv_half);
    ge_niels r; r.ypx = fe_add(y, x); r.ymx = fe_sub(y, x); r.t2d = fe_mul(fe_mul(x, y), FE_D4());
    return r;
}
// ... given 1/Z
BPG_HD ge_niels ge_to_niels(const ge_ext &p, const fe &zinv) { return ge_to_niels_halfinv(p, fe_mul(zinv, FE_INV2())); }

// r = sqrt(u/v) or sqrt(i*u/v); returns 1 when u/v is square (RFC 9496 SQRT_RATIO_M1)
BPG_HD uint32_t fe_sqrt_ratio_i(fe &r, const fe &u, const fe &v) {
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe rr = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(rr));
    fe negu = fe_neg(u);
    uint32_t correct = fe_eq(check, u);
    uint32_t flipped = fe_eq(check, negu);
    uint32_t flipped_i = fe_eq(check, fe_mul(negu, FE_SQRTM1()));
    rr = fe_select(rr, fe_mul(rr, FE_SQRTM1()), flipped | flipped_i);
    r = fe_abs(rr);
    return correct | flipped;
}

// RFC 9496 4.3.2 Encode
BPG_HD void ge_compress(uint8_t *out, const ge_ext &p) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe inv; fe_sqrt_ratio_i(inv, fe_one(), fe_mul(u1, fe_sq(u2)));
    fe i1 = fe_mul(inv, u1), i2 = fe_mul(inv, u2);
    fe zinv = fe_mul(fe_mul(i1, i2), p.T);
    fe iX = fe_mul(p.X, FE_SQRTM1()), iY = fe_mul(p.Y, FE_SQRTM1());
    fe ench = fe_mul(i1, FE_INVSQRT_A_MINUS_D());
    uint32_t rotate = fe_isnegative(fe_mul(p.T, zinv));
    fe X = fe_select(p.X, iY, rotate), Y = fe_select(p.Y, iX, rotate), den = fe_select(i2, ench, rotate);
    Y = fe_cneg(Y, fe_isnegative(fe_mul(X, zinv)));
    fe s = fe_abs(fe_mul(den, fe_sub(p.Z, Y)));
    fe_tobytes(out, s);
}

// RFC 9496 4.3.1 Decode: returns 1 and the point when s is a canonical, non-negative encoding of a group element
BPG_HD uint32_t ge_decompress(ge_ext &p, const uint8_t *in) {
    fe s = fe_frombytes(in);
    uint8_t chk[32]; fe_tobytes(chk, s);
    uint32_t canonical = 1;
    for (int i = 0; i < 32; i++) canonical &= (chk[i] == in[i]);         // also rejects bit 255 set
    const fe one = fe_one();
    fe ss = fe_sq(s);
    fe u1 = fe_sub(one, ss), u2 = fe_add(one, ss);
    fe u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(FE_D(), fe_sq(u1))), u2s);
    fe I; uint32_t ok = fe_sqrt_ratio_i(I, one, fe_mul(v, u2s));
    fe Dx = fe_mul(I, u2), Dy = fe_mul(fe_mul(I, Dx), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), Dx));
    fe y = fe_mul(u1, Dy);
    fe t = fe_mul(x, y);
    p.X = x; p.Y = y; p.Z = one; p.T = t;
    return canonical & (fe_isnegative(s) ^ 1u) & ok & (fe_isnegative(t) ^ 1u) & (fe_iszero(y) ^ 1u);
}

// RFC 9496 4.3.4 one-way map (dalek elligator_ristretto_flavor)
BPG_HD ge_ext ge_elligator(const fe &r0) {
    const fe one = fe_one();
    fe r = fe_mul(FE_SQRTM1(), fe_sq(r0));
    fe Ns = fe_mul(fe_add(r, one), FE_ONE_MINUS_D_SQ());
    fe c = fe_neg(one);
    fe D = fe_mul(fe_sub(c, fe_mul(FE_D(), r)), fe_add(r, FE_D()));
    fe s; uint32_t was_sq = fe_sqrt_ratio_i(s, Ns, D);
    fe sp = fe_neg(fe_abs(fe_mul(s, r0)));
    s = fe_select(s, sp, was_sq ^ 1u);
    c = fe_select(c, r, was_sq ^ 1u);
    fe Nt = fe_sub(fe_mul(fe_mul(c, fe_sub(r, one)), FE_D_MINUS_ONE_SQ()), D);
    fe ss = fe_sq(s);
    fe cX = fe_mul(fe_add(s, s), D), cZ = fe_mul(Nt, FE_SQRT_AD_MINUS_ONE()), cY = fe_sub(one, ss), cT = fe_add(one, ss);
    ge_ext p; p.X = fe_mul(cX, cT); p.Y = fe_mul(cY, cZ); p.Z = fe_mul(cZ, cT); p.T = fe_mul(cX, cY);
    return p;
}
BPG_HD ge_ext ge_from_uniform_words(const uint32_t *w) {   // 16 words = 64 uniform bytes
    return ge_add(ge_elligator(fe_fromwords(w)), ge_elligator(fe_fromwords(w + 8)));
}

}  // namespace bpg
