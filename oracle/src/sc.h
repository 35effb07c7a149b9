/* ORACLE (test infrastructure). Scalars mod l = 2^252 + 27742317777372353535851937790883648493.
 * Restates curve25519-dalek 1.x scalar.rs semantics (crate not vendored; Cargo.toml:8):
 *   - from_bits keeps 255 unreduced bits (reference call sites src/conversions.rs:18,43);
 *   - every arithmetic result is fully reduced; Mul = two Montgomery multiplications. */
#ifndef ORACLE_SC_H
#define ORACLE_SC_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t v[4]; } sc;   /* canonical (< l) unless documented otherwise */

void sc_frombytes_raw(sc *r, const uint8_t s[32]);          /* no reduction (from_bits callers mask bit 255) */
void sc_frombytes_mod_order(sc *r, const uint8_t s[32]);    /* any 256-bit value, reduced */
void sc_frombytes_wide(sc *r, const uint8_t s[64]);         /* Scalar::from_bytes_mod_order_wide */
void sc_tobytes(uint8_t s[32], const sc *a);
void sc_from_u64(sc *r, uint64_t x);
void sc_reduce(sc *r, const sc *a);                         /* a < 2^256 -> a mod l */
void sc_add(sc *r, const sc *a, const sc *b);               /* inputs < l */
void sc_sub(sc *r, const sc *a, const sc *b);
void sc_neg(sc *r, const sc *a);
void sc_mul(sc *r, const sc *a, const sc *b);               /* canonical inputs, output < l */
void sc_muladd(sc *r, const sc *a, const sc *b, const sc *c); /* a*b + c */
void sc_invert(sc *r, const sc *a);
int  sc_iszero(const sc *a);
int  sc_eq(const sc *a, const sc *b);
void sc_batch_invert(sc *v, size_t n);                      /* in place, all non-zero */
extern const sc SC_ZERO, SC_ONE;
#endif
