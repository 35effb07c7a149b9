/* ORACLE (test infrastructure). FIPS 202 permutations and sponges, byte-wise and unoptimised on purpose
 * (the product carries its own, separately written, implementation). */
#include "keccak.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ ROL(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PILN[i];
            bc[0] = st[j]; st[j] = ROL(t, ROTC[i]); t = bc[0];
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[r];
    }
}

static void sponge_absorb(uint64_t st[25], size_t *pos, size_t rate, const uint8_t *in, size_t len) {
    uint8_t *b = (uint8_t *)st;
    for (size_t i = 0; i < len; i++) {
        b[(*pos)++] ^= in[i];
        if (*pos == rate) { keccak_f1600(st); *pos = 0; }
    }
}

void sha3_512(uint8_t out[64], const uint8_t *in, size_t len) {
    uint64_t st[25]; size_t pos = 0; memset(st, 0, sizeof st);
    sponge_absorb(st, &pos, 72, in, len);
    uint8_t *b = (uint8_t *)st;
    b[pos] ^= 0x06; b[71] ^= 0x80;
    keccak_f1600(st);
    memcpy(out, st, 64);
}

void shake256_init(shake256_ctx *c) { memset(c, 0, sizeof *c); }
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len) { sponge_absorb(c->st, &c->pos, 136, in, len); }
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len) {
    uint8_t *b = (uint8_t *)c->st;
    if (!c->squeezing) {
        b[c->pos] ^= 0x1f; b[135] ^= 0x80;
        keccak_f1600(c->st); c->pos = 0; c->squeezing = 1;
    }
    for (size_t i = 0; i < len; i++) {
        if (c->pos == 136) { keccak_f1600(c->st); c->pos = 0; }
        out[i] = b[c->pos++];
    }
}
