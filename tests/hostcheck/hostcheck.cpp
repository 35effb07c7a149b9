// TEST HARNESS: compiles the product's __host__ __device__ arithmetic headers (csrc/hip/{fe,sc,ge}.cuh) for the
// host with g++ so that tests/test_device_arith_host.py can compare them with the oracle without a GPU.
// Not part of the product library; the product has no CPU execution path.
#include "../../bulletproofs_gadgets_amd/csrc/hip/ge.cuh"
#include "../../bulletproofs_gadgets_amd/csrc/hip/sc.cuh"
#include "../../bulletproofs_gadgets_amd/csrc/host/fe51.hpp"
#include <string.h>
using namespace bpg;

static fe load_raw(const uint8_t *s) { fe r; memcpy(r.v, s, 32); return r; }   // keeps bit 255: exercises weak forms
static void store_raw(uint8_t *s, const fe &a) { memcpy(s, a.v, 32); }

extern "C" {
// op: 0 mul 1 sq 2 add 3 sub 4 neg 5 invert 6 pow22523 7 freeze ; inputs are raw 256-bit values (any), output canonical
void hc_fe_op(int op, uint8_t *out, const uint8_t *a, const uint8_t *b) {
    fe x = load_raw(a), y = load_raw(b), r;
    switch (op) {
    case 0: r = fe_mul(x, y); break;
    case 1: r = fe_sq(x); break;
    case 2: r = fe_add(x, y); break;
    case 3: r = fe_sub(x, y); break;
    case 4: r = fe_neg(x); break;
    case 5: r = fe_invert(x); break;
    case 6: r = fe_pow22523(x); break;
    default: r = x; break;
    }
    fe_tobytes(out, r);
}
void hc_fe_chain(uint8_t *out, const uint8_t *a, const uint8_t *b, int rounds) {   // stresses weakly reduced intermediates
    fe x = load_raw(a), y = load_raw(b);
    for (int i = 0; i < rounds; i++) { fe t = fe_sub(fe_mul(x, y), fe_add(x, y)); x = fe_sq(fe_sub(y, t)); y = fe_add(t, fe_neg(x)); }
    fe_tobytes(out, fe_add(x, y));
}
void hc_sc_from_bytes(uint8_t *out_plain, const uint8_t *in) { uint32_t w[8]; memcpy(w, in, 32); scm m = sc_from_words(w); sc_to_words(w, m); memcpy(out_plain, w, 32); }
void hc_sc_from_wide(uint8_t *out_plain, const uint8_t *in) { uint32_t w[16]; memcpy(w, in, 64); scm m = sc_from_wide_words(w); uint32_t o[8]; sc_to_words(o, m); memcpy(out_plain, o, 32); }
// op: 0 mul 1 add 2 sub 3 neg
void hc_sc_op(int op, uint8_t *out, const uint8_t *a, const uint8_t *b) {
    uint32_t w[8]; memcpy(w, a, 32); scm x = sc_from_words(w); memcpy(w, b, 32); scm y = sc_from_words(w); scm r;
    switch (op) { case 0: r = sc_mont_mul(x, y); break; case 1: r = sc_add(x, y); break; case 2: r = sc_sub(x, y); break; default: r = sc_neg(x); break; }
    sc_to_words(w, r); memcpy(out, w, 32);
}
void hc_from_uniform(uint8_t *out, const uint8_t *in64) { uint32_t w[16]; memcpy(w, in64, 64); ge_compress(out, ge_from_uniform_words(w)); }
// double-and-add k*P + (P as niels via inversion) exercising dbl / madd / msub / add / to_niels / compress
void hc_scalarmul_uniform(uint8_t *out, const uint8_t *k32, const uint8_t *in64) {
    uint32_t w[16]; memcpy(w, in64, 64);
    ge_ext P = ge_from_uniform_words(w);
    ge_niels Pn = ge_to_niels(P, fe_invert(P.Z));
    ge_ext acc = ge_identity();
    for (int i = 255; i >= 0; i--) {
        acc = ge_dbl(acc);
        if ((k32[i / 8] >> (i % 8)) & 1) acc = ge_madd(acc, Pn);
    }
    // (acc - P) + P through msub / full add
    acc = ge_add(ge_msub(acc, Pn), P);
    acc = ge_madd_signed(ge_madd_signed(acc, Pn, 1), Pn, 0);
    ge_compress(out, acc);
}
// ---- host/fe51.hpp (the serial epilogues of the product: Horner recombination and point encoding on the host)
// op as hc_fe_op (0 mul 1 sq 2 add 3 sub 4 neg 6 pow22523 7 freeze), inputs raw 256-bit values as the device hands them over
void hc_fe51_op(int op, uint8_t *out, const uint8_t *a, const uint8_t *b) {
    uint32_t wa[8], wb[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);
    h51::fe51 x = h51::fe_from_words(wa), y = h51::fe_from_words(wb), r;
    switch (op) {
    case 0: r = h51::fe_mul(x, y); break;
    case 1: r = h51::fe_sq(x); break;
    case 2: r = h51::fe_add(x, y); break;
    case 3: r = h51::fe_sub(x, y); break;
    case 4: r = h51::fe_neg(x); break;
    case 6: r = h51::fe_pow22523(x); break;
    default: r = x; break;
    }
    h51::fe_tobytes(out, r);
}
// the same point through the device-side formulas (portable forms) and through fe51: k*P built with ge_dbl / ge_madd, then encoded by both;
// also 2^shift * P + Q through pt_dbl / pt_add against ge_dbl / ge_add.  Returns 1 when every pair of encodings agrees.
int hc_fe51_point_check(uint8_t *out_dev, uint8_t *out_h51, const uint8_t *k32, const uint8_t *in64, int shift) {
    uint32_t w[16]; memcpy(w, in64, 64);
    ge_ext P = ge_from_uniform_words(w);
    ge_niels Pn = ge_to_niels(P, fe_invert(P.Z));
    ge_ext acc = ge_identity();
    for (int i = 255; i >= 0; i--) { acc = ge_dbl(acc); if ((k32[i / 8] >> (i % 8)) & 1) acc = ge_madd(acc, Pn); }
    uint32_t raw[32]; memcpy(raw, &acc, 128);
    h51::pt hp = h51::pt_from_device(raw);
    ge_compress(out_dev, acc); h51::pt_compress(out_h51, hp);
    int ok = memcmp(out_dev, out_h51, 32) == 0;
    ge_ext d = acc; h51::pt hd = hp;
    for (int i = 0; i < shift; i++) { d = ge_dbl(d); hd = h51::pt_dbl(hd); }
    d = ge_add(d, P); memcpy(raw, &P, 128); hd = h51::pt_add(hd, h51::pt_from_device(raw));
    uint8_t e1[32], e2[32]; ge_compress(e1, d); h51::pt_compress(e2, hd);
    ok &= memcmp(e1, e2, 32) == 0;
    // Horner over W = 3 "window sums" (acc, P, d): sum_j 2^off(j) S_j with off = j * 254 / 3
    ge_ext S[3] = {acc, P, d};
    uint32_t sw[96]; memcpy(sw, S, 384);
    h51::pt hh = h51::pt_horner(sw, 3);
    ge_ext r = S[2];
    for (int win = 1; win >= 0; win--) { int sh = ((win + 1) * 254) / 3 - (win * 254) / 3; for (int i = 0; i < sh; i++) r = ge_dbl(r); r = ge_add(r, S[win]); }
    ge_compress(e1, r); h51::pt_compress(e2, hh);
    ok &= memcmp(e1, e2, 32) == 0;
    return ok;
}
}
