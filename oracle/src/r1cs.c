/* ORACLE (test infrastructure). Generators, R1CS prover, inner-product argument and verifier.
 * Restates dalek bulletproofs (develop 2019-H2) generators.rs, r1cs/prover.rs::prove, r1cs/verifier.rs::verify,
 * inner_product_proof.rs::{create, verification_scalars}, transcript.rs - the code the reference reaches at
 * src/bin/prover.rs:53-54,92-97 and src/bin/verifier.rs:89-90 (crate not vendored; see oracle.h). */
#include "oracle.h"
#include "ge.h"
#include "msm.h"
#include "merlin.h"
#include "keccak.h"
#include <stdlib.h>
#include <string.h>

struct orc_gens { uint64_t cap; ge *G, *H; };

static ge PED_B, PED_BB;
static int ped_ready = 0;
static void ped_init(void) {
    if (ped_ready) return;
    ge_basepoint(&PED_B);
    uint8_t c[32], h[64];
    ge_compress(c, &PED_B);
    sha3_512(h, c, 32);                          /* PedersenGens::default: hash_from_bytes::<Sha3_512>(B.compress()) */
    ge_from_uniform_bytes(&PED_BB, h);
    ped_ready = 1;
}

orc_gens *orc_gens_new(uint64_t cap) {
    orc_gens *g = (orc_gens *)calloc(1, sizeof *g);
    g->cap = cap;
    g->G = (ge *)malloc((cap ? cap : 1) * sizeof(ge));
    g->H = (ge *)malloc((cap ? cap : 1) * sizeof(ge));
    for (int which = 0; which < 2; which++) {
        /* GeneratorsChain::new(label = 'G'|'H' || u32le(party 0)) : SHAKE256("GeneratorsChain" || label) */
        shake256_ctx s; shake256_init(&s);
        uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), 0, 0, 0, 0};
        shake256_absorb(&s, (const uint8_t *)"GeneratorsChain", 15);
        shake256_absorb(&s, label, 5);
        ge *dst = which ? g->H : g->G;
        for (uint64_t i = 0; i < cap; i++) { uint8_t u[64]; shake256_squeeze(&s, u, 64); ge_from_uniform_bytes(&dst[i], u); }
    }
    return g;
}

orc_gens *orc_gens_from_compressed(uint64_t cap, const uint8_t *G, const uint8_t *H) {
    orc_gens *g = (orc_gens *)calloc(1, sizeof *g);
    g->cap = cap;
    g->G = (ge *)malloc((cap ? cap : 1) * sizeof(ge));
    g->H = (ge *)malloc((cap ? cap : 1) * sizeof(ge));
    for (uint64_t i = 0; i < cap; i++)
        if (!ge_decompress(&g->G[i], G + 32 * i) || !ge_decompress(&g->H[i], H + 32 * i)) { orc_gens_free(g); return NULL; }
    return g;
}

void orc_gens_free(orc_gens *g) { if (!g) return; free(g->G); free(g->H); free(g); }

void orc_gens_export(const orc_gens *g, uint64_t first, uint64_t count, uint8_t *Go, uint8_t *Ho) {
    for (uint64_t i = 0; i < count; i++) { ge_compress(Go + 32 * i, &g->G[first + i]); ge_compress(Ho + 32 * i, &g->H[first + i]); }
}

void orc_pedersen_bases(uint8_t B[32], uint8_t Bb[32]) { ped_init(); ge_compress(B, &PED_B); ge_compress(Bb, &PED_BB); }

static void pedersen_commit(ge *out, const sc *v, const sc *r) {
    ped_init();
    sc s[2] = {*v, *r}; ge p[2] = {PED_B, PED_BB};
    msm_straus_ct(out, s, p, 2);                 /* PedersenGens::commit = multiscalar_mul(&[v, r], &[B, B_blinding]) */
}

void orc_pedersen_commit(uint8_t out[32], const uint8_t v[32], const uint8_t r[32]) {
    sc a, b; sc_frombytes_mod_order(&a, v); sc_frombytes_mod_order(&b, r);
    ge c; pedersen_commit(&c, &a, &b); ge_compress(out, &c);
}

int orc_msm(uint8_t out[32], uint64_t n, const uint8_t *scalars, const uint8_t *points, int algo) {
    sc *s = (sc *)malloc((n ? n : 1) * sizeof(sc)); ge *p = (ge *)malloc((n ? n : 1) * sizeof(ge));
    int ok = 1;
    for (uint64_t i = 0; i < n; i++) { sc_frombytes_mod_order(&s[i], scalars + 32 * i); ok &= ge_decompress(&p[i], points + 32 * i); }
    if (ok) { ge r; if (algo == 0) msm_straus_ct(&r, s, p, n); else msm_vartime(&r, s, p, n); ge_compress(out, &r); }
    free(s); free(p);
    return ok ? ORC_OK : ORC_ERR_FORMAT;
}

/* ------------------------------------------------------------------------------------------------ helpers */
static uint64_t next_pow2(uint64_t n) { uint64_t p = 1; while (p < n) p <<= 1; return p; }   /* 0 -> 1 like Rust */

static void append_point(merlin_transcript *t, const char *label, const ge *p, uint8_t out[32]) {
    ge_compress(out, p); merlin_append(t, label, out, 32);
}
static void append_scalar(merlin_transcript *t, const char *label, const sc *s) {
    uint8_t b[32]; sc_tobytes(b, s); merlin_append(t, label, b, 32);
}
static void inner_product(sc *out, const sc *a, const sc *b, size_t n) {
    sc acc = SC_ZERO;
    for (size_t i = 0; i < n; i++) sc_muladd(&acc, &a[i], &b[i], &acc);
    *out = acc;
}

/* flattened_constraints(z): wL,wR,wO (n), wV (m), wc; constraint j (0-based) weighted by z^(j+1) */
static void flatten(const orc_circuit *c, const sc *z, sc *wL, sc *wR, sc *wO, sc *wV, sc *wc) {
    sc *coef = (sc *)malloc((c->ncoef ? c->ncoef : 1) * sizeof(sc));
    for (uint64_t i = 0; i < c->ncoef; i++) sc_frombytes_mod_order(&coef[i], c->coef + 32 * i);
    for (uint64_t i = 0; i < c->n; i++) wL[i] = wR[i] = wO[i] = SC_ZERO;
    for (uint64_t i = 0; i < c->m; i++) wV[i] = SC_ZERO;
    *wc = SC_ZERO;
    sc ez = *z;
    for (uint64_t j = 0; j < c->q; j++) {
        for (uint64_t k = c->row_ptr[j]; k < c->row_ptr[j + 1]; k++) {
            uint32_t kind = c->term_var[k] >> 29, idx = c->term_var[k] & 0x1fffffffu;
            sc t; sc_mul(&t, &ez, &coef[c->term_coef[k]]);
            switch (kind) {
            case ORC_VAR_MUL_LEFT: sc_add(&wL[idx], &wL[idx], &t); break;
            case ORC_VAR_MUL_RIGHT: sc_add(&wR[idx], &wR[idx], &t); break;
            case ORC_VAR_MUL_OUT: sc_add(&wO[idx], &wO[idx], &t); break;
            case ORC_VAR_COMMITTED: sc_sub(&wV[idx], &wV[idx], &t); break;
            default: sc_sub(wc, wc, &t); break;            /* Variable::One (verifier only) */
            }
        }
        sc_mul(&ez, &ez, z);
    }
    free(coef);
}

static int circuit_ok(const orc_circuit *c) {
    if (c->row_ptr[0] != 0 || c->row_ptr[c->q] != c->nnz) return 0;
    for (uint64_t k = 0; k < c->nnz; k++) {
        uint32_t kind = c->term_var[k] >> 29, idx = c->term_var[k] & 0x1fffffffu;
        if (c->term_coef[k] >= c->ncoef) return 0;
        if (kind <= ORC_VAR_MUL_OUT) { if (idx >= c->n) return 0; }
        else if (kind == ORC_VAR_COMMITTED) { if (idx >= c->m) return 0; }
        else if (kind != ORC_VAR_ONE) return 0;
    }
    return 1;
}

int orc_r1cs_satisfied(const orc_circuit *c, const uint8_t *v) {
    if (!circuit_ok(c)) return 0;
    sc *coef = (sc *)malloc((c->ncoef ? c->ncoef : 1) * sizeof(sc));
    for (uint64_t i = 0; i < c->ncoef; i++) sc_frombytes_mod_order(&coef[i], c->coef + 32 * i);
    int ok = 1;
    for (uint64_t i = 0; i < c->n && ok; i++) {
        sc a, b, o, p; sc_frombytes_mod_order(&a, c->aL + 32 * i); sc_frombytes_mod_order(&b, c->aR + 32 * i);
        sc_frombytes_mod_order(&o, c->aO + 32 * i); sc_mul(&p, &a, &b); ok &= sc_eq(&p, &o);
    }
    for (uint64_t j = 0; j < c->q && ok; j++) {
        sc acc = SC_ZERO;
        for (uint64_t k = c->row_ptr[j]; k < c->row_ptr[j + 1]; k++) {
            uint32_t kind = c->term_var[k] >> 29, idx = c->term_var[k] & 0x1fffffffu;
            sc x;
            switch (kind) {
            case ORC_VAR_MUL_LEFT: sc_frombytes_mod_order(&x, c->aL + 32 * (size_t)idx); break;
            case ORC_VAR_MUL_RIGHT: sc_frombytes_mod_order(&x, c->aR + 32 * (size_t)idx); break;
            case ORC_VAR_MUL_OUT: sc_frombytes_mod_order(&x, c->aO + 32 * (size_t)idx); break;
            case ORC_VAR_COMMITTED: sc_frombytes_mod_order(&x, v + 32 * (size_t)idx); break;
            default: x = SC_ONE; break;
            }
            sc_muladd(&acc, &coef[c->term_coef[k]], &x, &acc);
        }
        ok &= sc_iszero(&acc);
    }
    free(coef);
    return ok;
}

/* ------------------------------------------------------------------------------------------------ IPP */
/* InnerProductProof::create. G,H,a,b are consumed. L/R points appended to out (32 B each, L then R per round). */
static void ipp_create(merlin_transcript *T, const ge *Q, const sc *Gf, const sc *Hf, ge *G, ge *H, sc *a, sc *b,
                       size_t n, uint8_t *out, sc *a_out, sc *b_out) {
    merlin_append(T, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(T, "n", n);
    int first = 1;
    sc *s = (sc *)malloc((n + 1) * sizeof(sc));
    ge *p = (ge *)malloc((n + 1) * sizeof(ge));
    while (n != 1) {
        n /= 2;
        sc *aL = a, *aR = a + n, *bL = b, *bR = b + n;
        ge *GL = G, *GR = G + n, *HL = H, *HR = H + n;
        sc cL, cR; inner_product(&cL, aL, bR, n); inner_product(&cR, aR, bL, n);
        ge Lp, Rp;
        for (size_t i = 0; i < n; i++) {
            if (first) { sc_mul(&s[i], &aL[i], &Gf[n + i]); sc_mul(&s[n + i], &bR[i], &Hf[i]); }
            else { s[i] = aL[i]; s[n + i] = bR[i]; }
            p[i] = GR[i]; p[n + i] = HL[i];
        }
        s[2 * n] = cL; p[2 * n] = *Q;
        msm_vartime(&Lp, s, p, 2 * n + 1);
        for (size_t i = 0; i < n; i++) {
            if (first) { sc_mul(&s[i], &aR[i], &Gf[i]); sc_mul(&s[n + i], &bL[i], &Hf[n + i]); }
            else { s[i] = aR[i]; s[n + i] = bL[i]; }
            p[i] = GL[i]; p[n + i] = HR[i];
        }
        s[2 * n] = cR; p[2 * n] = *Q;
        msm_vartime(&Rp, s, p, 2 * n + 1);
        append_point(T, "L", &Lp, out); out += 32;
        append_point(T, "R", &Rp, out); out += 32;
        sc u, ui; merlin_challenge_scalar(T, "u", &u); sc_invert(&ui, &u);
        for (size_t i = 0; i < n; i++) {
            sc t1, t2;
            sc_mul(&t1, &aL[i], &u); sc_mul(&t2, &ui, &aR[i]); sc_add(&aL[i], &t1, &t2);
            sc_mul(&t1, &bL[i], &ui); sc_mul(&t2, &u, &bR[i]); sc_add(&bL[i], &t1, &t2);
            sc k[2]; ge pp[2];
            if (first) { sc_mul(&k[0], &ui, &Gf[i]); sc_mul(&k[1], &u, &Gf[n + i]); } else { k[0] = ui; k[1] = u; }
            pp[0] = GL[i]; pp[1] = GR[i]; msm_vartime(&GL[i], k, pp, 2);
            if (first) { sc_mul(&k[0], &u, &Hf[i]); sc_mul(&k[1], &ui, &Hf[n + i]); } else { k[0] = u; k[1] = ui; }
            pp[0] = HL[i]; pp[1] = HR[i]; msm_vartime(&HL[i], k, pp, 2);
        }
        first = 0;
    }
    *a_out = a[0]; *b_out = b[0];
    free(s); free(p);
}

/* ------------------------------------------------------------------------------------------------ prove */
int orc_r1cs_prove(const orc_gens *g, uint8_t tstate[203], const orc_circuit *c, const uint8_t *v_blinding,
                   const uint8_t seed[32], uint32_t flags, uint8_t *proof, uint64_t *proof_len) {
    ped_init();
    if (!circuit_ok(c)) return ORC_ERR_ARG;
    const size_t n = c->n, m = c->m;
    const size_t N = next_pow2(n);
    size_t lgN = 0; while (((size_t)1 << lgN) < N) lgN++;
    const int compact = (flags & ORC_FLAG_COMPACT_1PHASE) != 0;
    const size_t need = (compact ? 1 + 11 * 32 : 14 * 32) + (2 * lgN + 2) * 32;
    if (*proof_len < need) return ORC_ERR_ARG;
    if (g->cap < N) return ORC_ERR_GENS_LENGTH;
    void (*big_msm)(ge *, const sc *, const ge *, size_t) = (flags & ORC_FLAG_FAST_MSM) ? msm_vartime : msm_straus_ct;

    merlin_transcript T; memcpy(&T.s, tstate, 203);
    merlin_append_u64(&T, "m", m);

    sc *vb = (sc *)malloc((m ? m : 1) * sizeof(sc));
    merlin_rng rng; merlin_rng_begin(&rng, &T);
    for (size_t i = 0; i < m; i++) {
        sc_frombytes_mod_order(&vb[i], v_blinding + 32 * i);
        uint8_t b[32]; sc_tobytes(b, &vb[i]);
        merlin_rng_rekey(&rng, "v_blinding", b, 32);
    }
    merlin_rng_finalize(&rng, seed);

    sc *aL = (sc *)malloc((N) * sizeof(sc)), *aR = (sc *)malloc(N * sizeof(sc)), *aO = (sc *)malloc(N * sizeof(sc));
    for (size_t i = 0; i < n; i++) {
        sc_frombytes_mod_order(&aL[i], c->aL + 32 * i); sc_frombytes_mod_order(&aR[i], c->aR + 32 * i);
        sc_frombytes_mod_order(&aO[i], c->aO + 32 * i);
    }
    sc ib, ob, sb;
    merlin_rng_scalar(&rng, &ib); merlin_rng_scalar(&rng, &ob); merlin_rng_scalar(&rng, &sb);
    sc *sL = (sc *)malloc(N * sizeof(sc)), *sR = (sc *)malloc(N * sizeof(sc));
    if (flags & ORC_FLAG_EXPANDED_BLINDING) {
        /* opt-in dialect of the product (include/bpg.h BPG_FLAG_EXPANDED_BLINDING), restated independently: one 64-byte draw K,
         * then scalar j of s_L || s_R is SHAKE256("bpg blinding v1" || K || le64(j)) read as 64 bytes and reduced mod l */
        uint8_t K[64]; merlin_rng_fill(&rng, K, 64);
        for (size_t j = 0; j < 2 * n; j++) {
            shake256_ctx sh; shake256_init(&sh);
            shake256_absorb(&sh, (const uint8_t *)"bpg blinding v1", 15);
            shake256_absorb(&sh, K, 64);
            uint8_t idx[8]; for (int b = 0; b < 8; b++) idx[b] = (uint8_t)((uint64_t)j >> (8 * b));
            shake256_absorb(&sh, idx, 8);
            uint8_t wide[64]; shake256_squeeze(&sh, wide, 64);
            sc_frombytes_wide(j < n ? &sL[j] : &sR[j - n], wide);
        }
    } else {
        for (size_t i = 0; i < n; i++) merlin_rng_scalar(&rng, &sL[i]);
        for (size_t i = 0; i < n; i++) merlin_rng_scalar(&rng, &sR[i]);
    }

    /* A_I, A_O, S */
    sc *ms = (sc *)malloc((2 * N + 2) * sizeof(sc)); ge *mp = (ge *)malloc((2 * N + 2) * sizeof(ge));
    ge AI, AO, S;
    ms[0] = ib; mp[0] = PED_BB;
    for (size_t i = 0; i < n; i++) { ms[1 + i] = aL[i]; mp[1 + i] = g->G[i]; ms[1 + n + i] = aR[i]; mp[1 + n + i] = g->H[i]; }
    big_msm(&AI, ms, mp, 2 * n + 1);
    ms[0] = ob;
    for (size_t i = 0; i < n; i++) ms[1 + i] = aO[i];
    big_msm(&AO, ms, mp, n + 1);
    ms[0] = sb;
    for (size_t i = 0; i < n; i++) { ms[1 + i] = sL[i]; ms[1 + n + i] = sR[i]; }
    big_msm(&S, ms, mp, 2 * n + 1);

    uint8_t *out = proof;
    if (compact) *out++ = 0;                                   /* ONE_PHASE_COMMITMENTS version byte */
    append_point(&T, "A_I1", &AI, out); out += 32;
    append_point(&T, "A_O1", &AO, out); out += 32;
    append_point(&T, "S1", &S, out); out += 32;
    if (!(flags & ORC_FLAG_NO_1PHASE_DOMSEP)) merlin_append(&T, "dom-sep", (const uint8_t *)"r1cs-1phase", 11);
    uint8_t ident[32]; memset(ident, 0, 32);
    merlin_append(&T, "A_I2", ident, 32); merlin_append(&T, "A_O2", ident, 32); merlin_append(&T, "S2", ident, 32);
    if (!compact) { memset(out, 0, 96); out += 96; }

    sc y, z; merlin_challenge_scalar(&T, "y", &y); merlin_challenge_scalar(&T, "z", &z);
    sc *wL = (sc *)malloc(N * sizeof(sc)), *wR = (sc *)malloc(N * sizeof(sc)), *wO = (sc *)malloc(N * sizeof(sc));
    sc *wV = (sc *)malloc((m ? m : 1) * sizeof(sc)), wc;
    flatten(c, &z, wL, wR, wO, wV, &wc);

    sc yinv; sc_invert(&yinv, &y);
    sc *eyi = (sc *)malloc(N * sizeof(sc));
    { sc e = SC_ONE; for (size_t i = 0; i < N; i++) { eyi[i] = e; sc_mul(&e, &e, &yinv); } }
    sc *l1 = (sc *)malloc(N * sizeof(sc)), *l2 = aO, *l3 = sL;
    sc *r0 = (sc *)malloc(N * sizeof(sc)), *r1 = (sc *)malloc(N * sizeof(sc)), *r3 = (sc *)malloc(N * sizeof(sc));
    sc ey = SC_ONE;
    for (size_t i = 0; i < n; i++) {
        sc t; sc_mul(&t, &eyi[i], &wR[i]); sc_add(&l1[i], &aL[i], &t);
        sc_sub(&r0[i], &wO[i], &ey);
        sc_mul(&t, &ey, &aR[i]); sc_add(&r1[i], &t, &wL[i]);
        sc_mul(&r3[i], &ey, &sR[i]);
        sc_mul(&ey, &ey, &y);
    }
    sc t1, t2, t3, t4, t5, t6, tmp;
    inner_product(&t1, l1, r0, n);
    inner_product(&t2, l1, r1, n); inner_product(&tmp, l2, r0, n); sc_add(&t2, &t2, &tmp);
    inner_product(&t3, l2, r1, n); inner_product(&tmp, l3, r0, n); sc_add(&t3, &t3, &tmp);
    inner_product(&t4, l1, r3, n); inner_product(&tmp, l3, r1, n); sc_add(&t4, &t4, &tmp);
    inner_product(&t5, l2, r3, n);
    inner_product(&t6, l3, r3, n);

    sc tb1, tb3, tb4, tb5, tb6;
    merlin_rng_scalar(&rng, &tb1); merlin_rng_scalar(&rng, &tb3); merlin_rng_scalar(&rng, &tb4);
    merlin_rng_scalar(&rng, &tb5); merlin_rng_scalar(&rng, &tb6);
    ge Tp;
    pedersen_commit(&Tp, &t1, &tb1); append_point(&T, "T_1", &Tp, out); out += 32;
    pedersen_commit(&Tp, &t3, &tb3); append_point(&T, "T_3", &Tp, out); out += 32;
    pedersen_commit(&Tp, &t4, &tb4); append_point(&T, "T_4", &Tp, out); out += 32;
    pedersen_commit(&Tp, &t5, &tb5); append_point(&T, "T_5", &Tp, out); out += 32;
    pedersen_commit(&Tp, &t6, &tb6); append_point(&T, "T_6", &Tp, out); out += 32;

    sc u, x; merlin_challenge_scalar(&T, "u", &u); merlin_challenge_scalar(&T, "x", &x);
    sc tb2 = SC_ZERO;
    for (size_t i = 0; i < m; i++) sc_muladd(&tb2, &wV[i], &vb[i], &tb2);

    /* t(x), tau(x): x*(c1 + x*(c2 + ... + x*c6)) */
    sc tx, txb;
    { const sc *tc[6] = {&t1, &t2, &t3, &t4, &t5, &t6}; sc acc = SC_ZERO;
      for (int k = 5; k >= 0; k--) { sc_add(&acc, &acc, tc[k]); sc_mul(&acc, &acc, &x); } tx = acc; }
    { const sc *tc[6] = {&tb1, &tb2, &tb3, &tb4, &tb5, &tb6}; sc acc = SC_ZERO;
      for (int k = 5; k >= 0; k--) { sc_add(&acc, &acc, tc[k]); sc_mul(&acc, &acc, &x); } txb = acc; }

    sc *lv = (sc *)malloc(N * sizeof(sc)), *rv = (sc *)malloc(N * sizeof(sc));
    for (size_t i = 0; i < n; i++) {
        sc acc;                                 /* l = x*(l1 + x*(l2 + x*l3)) */
        sc_mul(&acc, &x, &l3[i]); sc_add(&acc, &acc, &l2[i]); sc_mul(&acc, &acc, &x); sc_add(&acc, &acc, &l1[i]);
        sc_mul(&lv[i], &acc, &x);
        sc_mul(&acc, &x, &r3[i]); sc_mul(&acc, &acc, &x); sc_add(&acc, &acc, &r1[i]); sc_mul(&acc, &acc, &x);
        sc_add(&rv[i], &acc, &r0[i]);           /* r = r0 + x*(r1 + x*(0 + x*r3)) */
    }
    for (size_t i = n; i < N; i++) { lv[i] = SC_ZERO; sc_neg(&rv[i], &ey); sc_mul(&ey, &ey, &y); }

    sc eb; sc_mul(&eb, &x, &sb); sc_add(&eb, &eb, &ob); sc_mul(&eb, &eb, &x); sc_add(&eb, &eb, &ib); sc_mul(&eb, &eb, &x);
    append_scalar(&T, "t_x", &tx); append_scalar(&T, "t_x_blinding", &txb); append_scalar(&T, "e_blinding", &eb);
    sc_tobytes(out, &tx); out += 32; sc_tobytes(out, &txb); out += 32; sc_tobytes(out, &eb); out += 32;

    sc w; merlin_challenge_scalar(&T, "w", &w);
    ge Q; ge_scalarmult(&Q, &w, &PED_B);
    sc *Gf = (sc *)malloc(N * sizeof(sc)), *Hf = (sc *)malloc(N * sizeof(sc));
    for (size_t i = 0; i < N; i++) { Gf[i] = i < n ? SC_ONE : u; sc_mul(&Hf[i], &eyi[i], &Gf[i]); }
    ge *Gv = (ge *)malloc(N * sizeof(ge)), *Hv = (ge *)malloc(N * sizeof(ge));
    memcpy(Gv, g->G, N * sizeof(ge)); memcpy(Hv, g->H, N * sizeof(ge));
    sc ipa, ipb;
    ipp_create(&T, &Q, Gf, Hf, Gv, Hv, lv, rv, N, out, &ipa, &ipb);
    out += 64 * lgN;
    sc_tobytes(out, &ipa); out += 32; sc_tobytes(out, &ipb); out += 32;
    *proof_len = (uint64_t)(out - proof);
    memcpy(tstate, &T.s, 203);

    free(vb); free(aL); free(aR); free(aO); free(sL); free(sR); free(ms); free(mp); free(wL); free(wR); free(wO); free(wV);
    free(eyi); free(l1); free(r0); free(r1); free(r3); free(lv); free(rv); free(Gf); free(Hf); free(Gv); free(Hv);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------ verify */
int orc_r1cs_verify(const orc_gens *g, uint8_t tstate[203], const orc_circuit *c, const uint8_t *V,
                    const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags) {
    ped_init();
    if (!circuit_ok(c)) return ORC_ERR_ARG;
    const size_t n = c->n, m = c->m, N = next_pow2(n);
    size_t lgN = 0; while (((size_t)1 << lgN) < N) lgN++;
    const int compact = (flags & ORC_FLAG_COMPACT_1PHASE) != 0;
    const size_t need = (compact ? 1 + 11 * 32 : 14 * 32) + (2 * lgN + 2) * 32;
    if (proof_len != need) return ORC_ERR_FORMAT;
    if (g->cap < N) return ORC_ERR_GENS_LENGTH;
    const uint8_t *in = proof;
    uint8_t ident[32]; memset(ident, 0, 32);
    if (compact) { if (*in++ != 0) return ORC_ERR_FORMAT; }
    const uint8_t *pAI = in, *pAO = in + 32, *pS = in + 64; in += 96;
    const uint8_t *pAI2 = ident, *pAO2 = ident, *pS2 = ident;
    if (!compact) { pAI2 = in; pAO2 = in + 32; pS2 = in + 64; in += 96; }
    const uint8_t *pT[5]; for (int i = 0; i < 5; i++) { pT[i] = in; in += 32; }
    sc tx, txb, eb;
    /* R1CSProof::from_bytes rejects non-canonical scalars */
    { sc r; sc_frombytes_raw(&tx, in); sc_reduce(&r, &tx); if (!sc_eq(&r, &tx)) return ORC_ERR_FORMAT; in += 32;
      sc_frombytes_raw(&txb, in); sc_reduce(&r, &txb); if (!sc_eq(&r, &txb)) return ORC_ERR_FORMAT; in += 32;
      sc_frombytes_raw(&eb, in); sc_reduce(&r, &eb); if (!sc_eq(&r, &eb)) return ORC_ERR_FORMAT; in += 32; }
    const uint8_t *pLR = in; in += 64 * lgN;
    sc ipa, ipb;
    { sc r; sc_frombytes_raw(&ipa, in); sc_reduce(&r, &ipa); if (!sc_eq(&r, &ipa)) return ORC_ERR_FORMAT; in += 32;
      sc_frombytes_raw(&ipb, in); sc_reduce(&r, &ipb); if (!sc_eq(&r, &ipb)) return ORC_ERR_FORMAT; in += 32; }

    merlin_transcript T; memcpy(&T.s, tstate, 203);
    merlin_append_u64(&T, "m", m);
    /* validate_and_append_point: identity encodings are rejected for first-phase and T points */
    if (!memcmp(pAI, ident, 32) || !memcmp(pAO, ident, 32) || !memcmp(pS, ident, 32)) return ORC_ERR_VERIFY;
    merlin_append(&T, "A_I1", pAI, 32); merlin_append(&T, "A_O1", pAO, 32); merlin_append(&T, "S1", pS, 32);
    if (!(flags & ORC_FLAG_NO_1PHASE_DOMSEP)) merlin_append(&T, "dom-sep", (const uint8_t *)"r1cs-1phase", 11);
    merlin_append(&T, "A_I2", pAI2, 32); merlin_append(&T, "A_O2", pAO2, 32); merlin_append(&T, "S2", pS2, 32);
    sc y, z; merlin_challenge_scalar(&T, "y", &y); merlin_challenge_scalar(&T, "z", &z);
    static const char *tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int i = 0; i < 5; i++) { if (!memcmp(pT[i], ident, 32)) return ORC_ERR_VERIFY; merlin_append(&T, tl[i], pT[i], 32); }
    sc u, x; merlin_challenge_scalar(&T, "u", &u); merlin_challenge_scalar(&T, "x", &x);
    append_scalar(&T, "t_x", &tx); append_scalar(&T, "t_x_blinding", &txb); append_scalar(&T, "e_blinding", &eb);
    sc w; merlin_challenge_scalar(&T, "w", &w);

    sc *wL = (sc *)malloc(N * sizeof(sc)), *wR = (sc *)malloc(N * sizeof(sc)), *wO = (sc *)malloc(N * sizeof(sc));
    sc *wV = (sc *)malloc((m ? m : 1) * sizeof(sc)), wc;
    flatten(c, &z, wL, wR, wO, wV, &wc);
    for (size_t i = n; i < N; i++) wL[i] = wR[i] = wO[i] = SC_ZERO;

    /* verification_scalars */
    merlin_append(&T, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&T, "n", N);
    sc *ch = (sc *)malloc((lgN ? lgN : 1) * sizeof(sc)), *chi = (sc *)malloc((lgN ? lgN : 1) * sizeof(sc));
    int rc = ORC_OK;
    for (size_t k = 0; k < lgN; k++) {
        const uint8_t *Lp = pLR + 64 * k, *Rp = Lp + 32;
        if (!memcmp(Lp, ident, 32) || !memcmp(Rp, ident, 32)) rc = ORC_ERR_VERIFY;
        merlin_append(&T, "L", Lp, 32); merlin_append(&T, "R", Rp, 32);
        merlin_challenge_scalar(&T, "u", &ch[k]); chi[k] = ch[k];
    }
    sc allinv = SC_ONE;
    if (lgN) { sc_batch_invert(chi, lgN); for (size_t k = 0; k < lgN; k++) sc_mul(&allinv, &allinv, &chi[k]); }
    sc *usq = (sc *)malloc((lgN ? lgN : 1) * sizeof(sc)), *uisq = (sc *)malloc((lgN ? lgN : 1) * sizeof(sc));
    for (size_t k = 0; k < lgN; k++) { sc_mul(&usq[k], &ch[k], &ch[k]); sc_mul(&uisq[k], &chi[k], &chi[k]); }
    sc *s = (sc *)malloc(N * sizeof(sc));
    s[0] = allinv;
    for (size_t i = 1; i < N; i++) {
        size_t lg = 0; while (((size_t)2 << lg) <= i) lg++;
        size_t k = (size_t)1 << lg;
        sc_mul(&s[i], &s[i - k], &usq[(lgN - 1) - lg]);
    }

    sc yinv; sc_invert(&yinv, &y);
    sc *eyi = (sc *)malloc(N * sizeof(sc));
    { sc e = SC_ONE; for (size_t i = 0; i < N; i++) { eyi[i] = e; sc_mul(&e, &e, &yinv); } }
    sc *ynwR = (sc *)malloc(N * sizeof(sc));
    for (size_t i = 0; i < N; i++) sc_mul(&ynwR[i], &wR[i], &eyi[i]);
    sc delta; inner_product(&delta, ynwR, wL, n);

    merlin_rng rng; merlin_rng_begin(&rng, &T); merlin_rng_finalize(&rng, seed);
    sc r; merlin_rng_scalar(&rng, &r);
    sc xx, rxx, xxx; sc_mul(&xx, &x, &x); sc_mul(&rxx, &r, &xx); sc_mul(&xxx, &x, &xx);

    size_t total = 6 + m + 5 + 2 + 2 * N + 2 * lgN;
    sc *ks = (sc *)malloc(total * sizeof(sc)); ge *ps = (ge *)malloc(total * sizeof(ge));
    size_t k = 0; int ok = 1;
    ks[k] = x; ok &= ge_decompress(&ps[k++], pAI);
    ks[k] = xx; ok &= ge_decompress(&ps[k++], pAO);
    ks[k] = xxx; ok &= ge_decompress(&ps[k++], pS);
    sc_mul(&ks[k], &u, &x); ok &= ge_decompress(&ps[k++], pAI2);
    sc_mul(&ks[k], &u, &xx); ok &= ge_decompress(&ps[k++], pAO2);
    sc_mul(&ks[k], &u, &xxx); ok &= ge_decompress(&ps[k++], pS2);
    for (size_t i = 0; i < m; i++) { sc_mul(&ks[k], &wV[i], &rxx); ok &= ge_decompress(&ps[k++], V + 32 * i); }
    { sc t; sc_mul(&ks[k], &r, &x); ok &= ge_decompress(&ps[k++], pT[0]);
      sc_mul(&ks[k], &rxx, &x); ok &= ge_decompress(&ps[k++], pT[1]);
      sc_mul(&ks[k], &rxx, &xx); ok &= ge_decompress(&ps[k++], pT[2]);
      sc_mul(&ks[k], &rxx, &xxx); ok &= ge_decompress(&ps[k++], pT[3]);
      sc_mul(&t, &rxx, &xx); sc_mul(&ks[k], &t, &xx); ok &= ge_decompress(&ps[k++], pT[4]); }
    { /* B: w*(t_x - a*b) + r*(xx*(wc + delta) - t_x) */
      sc ab, t1, t2; sc_mul(&ab, &ipa, &ipb); sc_sub(&t1, &tx, &ab); sc_mul(&t1, &t1, &w);
      sc_add(&t2, &wc, &delta); sc_mul(&t2, &t2, &xx); sc_sub(&t2, &t2, &tx); sc_mul(&t2, &t2, &r);
      sc_add(&ks[k], &t1, &t2); ps[k++] = PED_B;
      /* B_blinding: -e_blinding - r*t_x_blinding */
      sc_mul(&t1, &r, &txb); sc_add(&t1, &t1, &eb); sc_neg(&ks[k], &t1); ps[k++] = PED_BB; }
    for (size_t i = 0; i < N; i++) {   /* G: u_or_1 * (x*yneg_wR_i - a*s_i) */
        sc t1, t2; sc_mul(&t1, &x, &ynwR[i]); sc_mul(&t2, &ipa, &s[i]); sc_sub(&t1, &t1, &t2);
        if (i >= n) sc_mul(&t1, &t1, &u);
        ks[k] = t1; ps[k++] = g->G[i];
    }
    for (size_t i = 0; i < N; i++) {   /* H: u_or_1 * (y^-i * (x*wL_i + wO_i - b*s_{N-1-i}) - 1) */
        sc t1, t2; sc_mul(&t1, &x, &wL[i]); sc_add(&t1, &t1, &wO[i]); sc_mul(&t2, &ipb, &s[N - 1 - i]); sc_sub(&t1, &t1, &t2);
        sc_mul(&t1, &t1, &eyi[i]); sc_sub(&t1, &t1, &SC_ONE);
        if (i >= n) sc_mul(&t1, &t1, &u);
        ks[k] = t1; ps[k++] = g->H[i];
    }
    for (size_t i = 0; i < lgN; i++) { ks[k] = usq[i]; ok &= ge_decompress(&ps[k++], pLR + 64 * i); }
    for (size_t i = 0; i < lgN; i++) { ks[k] = uisq[i]; ok &= ge_decompress(&ps[k++], pLR + 64 * i + 32); }
    if (!ok) rc = ORC_ERR_VERIFY;
    if (rc == ORC_OK) {
        ge chk; msm_vartime(&chk, ks, ps, k);
        if (!ge_is_identity(&chk)) rc = ORC_ERR_VERIFY;
    }
    memcpy(tstate, &T.s, 203);
    free(wL); free(wR); free(wO); free(wV); free(ch); free(chi); free(usq); free(uisq); free(s); free(eyi); free(ynwR);
    free(ks); free(ps);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ primitives for tests */
void orc_transcript_init(uint8_t ts[203], const uint8_t *label, uint64_t len) { merlin_transcript t; merlin_init(&t, label, len); memcpy(ts, &t.s, 203); }
void orc_transcript_append(uint8_t ts[203], const char *label, const uint8_t *msg, uint64_t len) {
    merlin_transcript t; memcpy(&t.s, ts, 203); merlin_append(&t, label, msg, len); memcpy(ts, &t.s, 203);
}
void orc_transcript_append_u64(uint8_t ts[203], const char *label, uint64_t v) {
    merlin_transcript t; memcpy(&t.s, ts, 203); merlin_append_u64(&t, label, v); memcpy(ts, &t.s, 203);
}
void orc_transcript_challenge(uint8_t ts[203], const char *label, uint8_t *out, uint64_t len) {
    merlin_transcript t; memcpy(&t.s, ts, 203); merlin_challenge_bytes(&t, label, out, len); memcpy(ts, &t.s, 203);
}
void orc_transcript_challenge_scalar(uint8_t ts[203], const char *label, uint8_t out[32]) {
    merlin_transcript t; memcpy(&t.s, ts, 203); sc s; merlin_challenge_scalar(&t, label, &s); sc_tobytes(out, &s); memcpy(ts, &t.s, 203);
}
void orc_rng_scalars(const uint8_t ts[203], uint64_t m, const uint8_t *vb, const uint8_t seed[32], uint64_t count, uint8_t *out) {
    merlin_transcript t; memcpy(&t.s, ts, 203);
    merlin_rng r; merlin_rng_begin(&r, &t);
    for (uint64_t i = 0; i < m; i++) merlin_rng_rekey(&r, "v_blinding", vb + 32 * i, 32);
    merlin_rng_finalize(&r, seed);
    for (uint64_t i = 0; i < count; i++) { sc s; merlin_rng_scalar(&r, &s); sc_tobytes(out + 32 * i, &s); }
}
void orc_sc_wide(uint8_t out[32], const uint8_t in[64]) { sc s; sc_frombytes_wide(&s, in); sc_tobytes(out, &s); }
void orc_sc_reduce(uint8_t out[32], const uint8_t in[32]) { sc s; sc_frombytes_mod_order(&s, in); sc_tobytes(out, &s); }
#define BIN(name, fn) void name(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) { \
    sc x, y, r; sc_frombytes_mod_order(&x, a); sc_frombytes_mod_order(&y, b); fn(&r, &x, &y); sc_tobytes(out, &r); }
BIN(orc_sc_mul, sc_mul) BIN(orc_sc_add, sc_add) BIN(orc_sc_sub, sc_sub)
void orc_sc_invert(uint8_t out[32], const uint8_t a[32]) { sc x, r; sc_frombytes_mod_order(&x, a); sc_invert(&r, &x); sc_tobytes(out, &r); }
void orc_from_uniform(uint8_t out[32], const uint8_t in[64]) { ge p; ge_from_uniform_bytes(&p, in); ge_compress(out, &p); }
int orc_point_mul(uint8_t out[32], const uint8_t k[32], const uint8_t p[32]) {
    ge P, R; sc s; if (!ge_decompress(&P, p)) return ORC_ERR_FORMAT;
    sc_frombytes_mod_order(&s, k); ge_scalarmult(&R, &s, &P); ge_compress(out, &R); return ORC_OK;
}
int orc_point_add(uint8_t out[32], const uint8_t p[32], const uint8_t q[32]) {
    ge P, Q, R; if (!ge_decompress(&P, p) || !ge_decompress(&Q, q)) return ORC_ERR_FORMAT;
    ge_add(&R, &P, &Q); ge_compress(out, &R); return ORC_OK;
}
void orc_sha3_512(uint8_t out[64], const uint8_t *in, uint64_t len) { sha3_512(out, in, len); }
void orc_shake256(uint8_t *out, uint64_t outlen, const uint8_t *in, uint64_t len) {
    shake256_ctx c; shake256_init(&c); shake256_absorb(&c, in, len); shake256_squeeze(&c, out, outlen);
}
