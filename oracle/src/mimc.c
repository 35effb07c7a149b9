/* ORACLE (test infrastructure). Native MiMC-x^3 sponge hash over the scalar field.
 * Restates reference src/mimc_hash/mimc.rs: mimc_encryption :7-23, mimc_sponge_1 :26-40, mimc_hash :61-75,
 * pad :77-97, with src/conversions.rs:26-30 (be_to_scalars) and :33-46 (le_to_scalar / from_bits). */
#include "oracle.h"
#include "sc.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t RC769[486][4] = {
#include "mimc_rc769.inc"
};

static void encrypt(sc *state, const sc *key) {      /* mimc.rs:7-23 */
    sc s = *state;
    for (int i = 0; i < 486; i++) {
        sc c, t, t2; memcpy(c.v, RC769[i], 32);
        sc_add(&t, key, &c); sc_add(&t, &s, &t);
        sc_mul(&t2, &t, &t); sc_mul(&s, &t2, &t);
    }
    sc_add(state, &s, key);
}

void orc_mimc_sponge(uint8_t out[32], const uint8_t *blocks, uint64_t nblocks) {   /* mimc.rs:26-40 */
    sc state = SC_ZERO, zero = SC_ZERO;
    for (uint64_t i = 0; i < nblocks; i++) {
        sc b; sc_frombytes_mod_order(&b, blocks + 32 * i);
        sc_add(&state, &state, &b);
        encrypt(&state, &zero);
    }
    sc_tobytes(out, &state);
}

void orc_mimc_hash(uint8_t out[32], const uint8_t *pre, uint64_t len) {            /* mimc.rs:61-97 */
    uint64_t padded = (len + 31) / 32 * 32;
    uint64_t nb = padded / 32;
    uint8_t *buf = (uint8_t *)calloc(padded + 32, 1);
    for (uint64_t i = 0; i < len; i++) buf[i] = pre[len - 1 - i];                   /* be_to_scalars: reverse, zero-pad */
    for (uint64_t i = 0; i < nb; i++) buf[32 * i + 31] &= 0x7f;                    /* Scalar::from_bits */
    uint8_t *last = buf + 32 * (nb - 1);
    int l = 32; while (l > 0 && last[l - 1] == 0) l--;                              /* remove_zero_padding! */
    if (l < 32) { uint8_t k = (uint8_t)(32 - l); for (int i = l; i < 32; i++) last[i] = k; last[31] &= 0x7f; }
    else { memset(buf + padded, 32, 32); nb++; }
    orc_mimc_sponge(out, buf, nb);
    free(buf);
}
