/* ORACLE (test infrastructure). Multiscalar multiplication with the algorithms upstream uses, so that the
 * CPU baseline is not a strawman:
 *   msm_straus_ct  = RistrettoPoint::multiscalar_mul        (curve25519-dalek 1.x scalar_mul/straus.rs, radix 16,
 *                    8-entry table per point, constant-time table scan)  -> A_I, A_O, S, T_k, Pedersen commits
 *   msm_vartime    = RistrettoPoint::vartime_multiscalar_mul (straus.rs width-5 NAF below 190 terms, else
 *                    pippenger.rs with w = 6/7/8)                         -> L_k, R_k, generator folds, verifier
 */
#include "msm.h"
#include <stdlib.h>
#include <string.h>

static void pn_cmov(ge_pn *a, const ge_pn *b, unsigned m) {
    fe_cmov(&a->YpX, &b->YpX, m); fe_cmov(&a->YmX, &b->YmX, m); fe_cmov(&a->Z, &b->Z, m); fe_cmov(&a->T2d, &b->T2d, m);
}
static void pn_identity(ge_pn *a) { fe_1(&a->YpX); fe_1(&a->YmX); fe_1(&a->Z); fe_0(&a->T2d); }
static void pn_cneg(ge_pn *a, unsigned m) {
    ge_pn n; n.YpX = a->YmX; n.YmX = a->YpX; n.Z = a->Z; fe_neg(&n.T2d, &a->T2d);
    pn_cmov(a, &n, m);
}

static void radix16(int8_t out[64], const sc *k) {
    uint8_t b[32]; sc_tobytes(b, k);
    for (int i = 0; i < 32; i++) { out[2 * i] = (int8_t)(b[i] & 15); out[2 * i + 1] = (int8_t)((b[i] >> 4) & 15); }
    for (int i = 0; i < 63; i++) {
        int8_t carry = (int8_t)((out[i] + 8) >> 4);
        out[i] = (int8_t)(out[i] - (carry << 4)); out[i + 1] = (int8_t)(out[i + 1] + carry);
    }
}

#define CT_CHUNK 2048
void msm_straus_ct(ge *out, const sc *scalars, const ge *points, size_t n) {
    ge total; ge_identity(&total);
    ge_pn *tab = (ge_pn *)malloc((size_t)CT_CHUNK * 8 * sizeof(ge_pn));
    int8_t *dig = (int8_t *)malloc((size_t)CT_CHUNK * 64);
    /* Chunking only bounds memory (upstream keeps one 1.25 KiB table per point alive); it adds 252 doublings per
       2048 terms (<0.2%) and leaves the group element unchanged. */
    for (size_t base = 0; base < n; base += CT_CHUNK) {
        size_t cnt = n - base < CT_CHUNK ? n - base : CT_CHUNK;
        for (size_t i = 0; i < cnt; i++) {
            ge acc = points[base + i]; ge_pn p1; ge_to_pn(&p1, &acc);
            tab[8 * i] = p1;
            for (int j = 1; j < 8; j++) { ge_add_pn(&acc, &acc, &p1); ge_to_pn(&tab[8 * i + j], &acc); }
            radix16(dig + 64 * i, &scalars[base + i]);
        }
        ge Q; ge_identity(&Q);
        for (int j = 63; j >= 0; j--) {
            for (int d = 0; d < 4; d++) ge_double(&Q, &Q);
            for (size_t i = 0; i < cnt; i++) {
                int8_t x = dig[64 * i + j];
                unsigned neg = (unsigned)(x < 0);
                unsigned ax = (unsigned)(neg ? -x : x);
                ge_pn t; pn_identity(&t);
                for (unsigned e = 1; e <= 8; e++) pn_cmov(&t, &tab[8 * i + e - 1], (unsigned)(ax == e));
                pn_cneg(&t, neg);
                ge_add_pn(&Q, &Q, &t);
            }
        }
        ge_add(&total, &total, &Q);
    }
    free(tab); free(dig);
    *out = total;
}

static void naf(int8_t out[256], const sc *k, int w) {
    uint64_t x[5] = {k->v[0], k->v[1], k->v[2], k->v[3], 0};
    memset(out, 0, 256);
    uint64_t width = 1ULL << w, mask = width - 1, carry = 0;
    int pos = 0;
    while (pos < 256) {
        int idx = pos / 64, bit = pos % 64;
        uint64_t buf = (bit < 64 - w) ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
        uint64_t window = carry + (buf & mask);
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < width / 2) { carry = 0; out[pos] = (int8_t)window; }
        else { carry = 1; out[pos] = (int8_t)((int64_t)window - (int64_t)width); }
        pos += w;
    }
}

static void straus_vartime(ge *out, const sc *scalars, const ge *points, size_t n) {
    int8_t *nafs = (int8_t *)malloc(n * 256);
    ge_pn *tab = (ge_pn *)malloc(n * 8 * sizeof(ge_pn));
    for (size_t i = 0; i < n; i++) {
        naf(nafs + 256 * i, &scalars[i], 5);
        ge a2, acc = points[i]; ge_double(&a2, &acc);
        ge_pn d2; ge_to_pn(&d2, &a2);
        ge_to_pn(&tab[8 * i], &acc);
        for (int j = 1; j < 8; j++) { ge_add_pn(&acc, &acc, &d2); ge_to_pn(&tab[8 * i + j], &acc); }
    }
    ge r; ge_identity(&r);
    int started = 0;
    for (int b = 255; b >= 0; b--) {
        if (started) ge_double(&r, &r);
        for (size_t i = 0; i < n; i++) {
            int8_t d = nafs[256 * i + b];
            if (d > 0) { ge_add_pn(&r, &r, &tab[8 * i + d / 2]); started = 1; }
            else if (d < 0) { ge_sub_pn(&r, &r, &tab[8 * i + (-d) / 2]); started = 1; }
        }
    }
    free(nafs); free(tab);
    *out = r;
}

static void pippenger(ge *out, const sc *scalars, const ge *points, size_t n) {
    int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    int ndig = (256 + w - 1) / w + (w == 8 ? 1 : 0);
    size_t nb = (size_t)1 << (w - 1);
    int16_t *dig = (int16_t *)calloc(n * (size_t)ndig, sizeof(int16_t));
    ge_pn *pts = (ge_pn *)malloc(n * sizeof(ge_pn));
    for (size_t i = 0; i < n; i++) {
        ge_to_pn(&pts[i], &points[i]);
        uint64_t x[5] = {scalars[i].v[0], scalars[i].v[1], scalars[i].v[2], scalars[i].v[3], 0};
        uint64_t radix = 1ULL << w, mask = radix - 1, carry = 0;
        int cnt = (256 + w - 1) / w;
        for (int d = 0; d < cnt; d++) {
            int off = d * w, idx = off / 64, bit = off % 64;
            uint64_t buf = (bit < 64 - w || idx == 3) ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
            uint64_t coef = carry + (buf & mask);
            carry = (coef + radix / 2) >> w;
            dig[i * (size_t)ndig + d] = (int16_t)((int64_t)coef - (int64_t)(carry << w));
        }
        if (w == 8) dig[i * (size_t)ndig + cnt] = (int16_t)(dig[i * (size_t)ndig + cnt] + (int16_t)carry);
        else dig[i * (size_t)ndig + cnt - 1] = (int16_t)(dig[i * (size_t)ndig + cnt - 1] + (int16_t)(carry << w));
    }
    ge *buckets = (ge *)malloc(nb * sizeof(ge));
    ge total; ge_identity(&total);
    for (int d = ndig - 1; d >= 0; d--) {
        for (size_t b = 0; b < nb; b++) ge_identity(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int16_t x = dig[i * (size_t)ndig + d];
            if (x > 0) ge_add_pn(&buckets[x - 1], &buckets[x - 1], &pts[i]);
            else if (x < 0) ge_sub_pn(&buckets[-x - 1], &buckets[-x - 1], &pts[i]);
        }
        ge run = buckets[nb - 1], sum = buckets[nb - 1];
        for (size_t b = nb - 1; b-- > 0;) { ge_add(&run, &run, &buckets[b]); ge_add(&sum, &sum, &run); }
        if (d != ndig - 1) for (int k = 0; k < w; k++) ge_double(&total, &total);
        ge_add(&total, &total, &sum);
    }
    free(buckets); free(pts); free(dig);
    *out = total;
}

void msm_vartime(ge *out, const sc *scalars, const ge *points, size_t n) {
    if (n < 190) straus_vartime(out, scalars, points, n);
    else pippenger(out, scalars, points, n);
}
