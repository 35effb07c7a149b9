/* ORACLE (test infrastructure). See merlin.h for provenance. */
#include "merlin.h"
#include "keccak.h"
#include <string.h>

#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void run_f(strobe128 *s) {
    s->st[s->pos] ^= s->pos_begin;
    s->st[s->pos + 1] ^= 0x04;
    s->st[STROBE_R + 1] ^= 0x80;
    uint64_t w[25]; memcpy(w, s->st, 200);
    keccak_f1600(w);
    memcpy(s->st, w, 200);
    s->pos = 0; s->pos_begin = 0;
}

static void absorb(strobe128 *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) run_f(s); }
}
static void overwrite(strobe128 *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] = d[i]; if (s->pos == STROBE_R) run_f(s); }
}
static void squeeze(strobe128 *s, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == STROBE_R) run_f(s); }
}
static void begin_op(strobe128 *s, uint8_t flags, int more) {
    if (more) return;                          /* caller guarantees cur_flags == flags */
    uint8_t old_begin = s->pos_begin;
    s->pos_begin = (uint8_t)(s->pos + 1);
    s->cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(s, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) run_f(s);
}
static void meta_ad(strobe128 *s, const uint8_t *d, size_t n, int more) { begin_op(s, FLAG_M | FLAG_A, more); absorb(s, d, n); }
static void ad(strobe128 *s, const uint8_t *d, size_t n, int more) { begin_op(s, FLAG_A, more); absorb(s, d, n); }
static void prf(strobe128 *s, uint8_t *d, size_t n, int more) { begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); squeeze(s, d, n); }
static void key(strobe128 *s, const uint8_t *d, size_t n, int more) { begin_op(s, FLAG_A | FLAG_C, more); overwrite(s, d, n); }

static void strobe_new(strobe128 *s, const char *proto) {
    memset(s, 0, sizeof *s);
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(s->st, hdr, 6); memcpy(s->st + 6, "STROBEv1.0.2", 12);
    uint64_t w[25]; memcpy(w, s->st, 200); keccak_f1600(w); memcpy(s->st, w, 200);
    meta_ad(s, (const uint8_t *)proto, strlen(proto), 0);
}

static void le32(uint8_t b[4], size_t n) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }

void merlin_append(merlin_transcript *t, const char *label, const uint8_t *msg, size_t len) {
    uint8_t l4[4]; le32(l4, len);
    meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    meta_ad(&t->s, l4, 4, 1);
    ad(&t->s, msg, len, 0);
}

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t len) {
    strobe_new(&t->s, "Merlin v1.0");
    merlin_append(t, "dom-sep", label, len);
}

void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t v) {
    uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    merlin_append(t, label, b, 8);
}

void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t len) {
    uint8_t l4[4]; le32(l4, len);
    meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    meta_ad(&t->s, l4, 4, 1);
    prf(&t->s, out, len, 0);
}

void merlin_challenge_scalar(merlin_transcript *t, const char *label, sc *out) {
    uint8_t b[64]; merlin_challenge_bytes(t, label, b, 64); sc_frombytes_wide(out, b);
}

void merlin_rng_begin(merlin_rng *r, const merlin_transcript *t) { r->s = t->s; }
void merlin_rng_rekey(merlin_rng *r, const char *label, const uint8_t *w, size_t len) {
    uint8_t l4[4]; le32(l4, len);
    meta_ad(&r->s, (const uint8_t *)label, strlen(label), 0);
    meta_ad(&r->s, l4, 4, 1);
    key(&r->s, w, len, 0);
}
void merlin_rng_finalize(merlin_rng *r, const uint8_t seed[32]) {
    meta_ad(&r->s, (const uint8_t *)"rng", 3, 0);
    key(&r->s, seed, 32, 0);
}
void merlin_rng_fill(merlin_rng *r, uint8_t *out, size_t len) {
    uint8_t l4[4]; le32(l4, len);
    meta_ad(&r->s, l4, 4, 0);
    prf(&r->s, out, len, 0);
}
void merlin_rng_scalar(merlin_rng *r, sc *out) { uint8_t b[64]; merlin_rng_fill(r, b, 64); sc_frombytes_wide(out, b); }
