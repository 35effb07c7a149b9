/* ORACLE (test infrastructure). Ristretto255 over the twisted Edwards curve -x^2+y^2 = 1+d x^2 y^2.
 * Restates curve25519-dalek 1.x edwards.rs / ristretto.rs / backend/serial/curve_models
 * (not vendored; Cargo.toml:8) = RFC 9496.  Reference call sites: src/bin/prover.rs:53 (PedersenGens),
 * src/lalrpop/assignment_parser.rs:145 (CompressedRistretto::from_slice). */
#ifndef ORACLE_GE_H
#define ORACLE_GE_H
#include "fe.h"
#include "sc.h"

typedef struct { fe X, Y, Z, T; } ge;           /* extended */
typedef struct { fe YpX, YmX, Z, T2d; } ge_pn;  /* projective Niels ("cached") */

void ge_identity(ge *p);
void ge_add(ge *r, const ge *p, const ge *q);
void ge_sub(ge *r, const ge *p, const ge *q);
void ge_neg(ge *r, const ge *p);
void ge_double(ge *r, const ge *p);
void ge_to_pn(ge_pn *r, const ge *p);
void ge_add_pn(ge *r, const ge *p, const ge_pn *q);
void ge_sub_pn(ge *r, const ge *p, const ge_pn *q);
int  ge_eq(const ge *p, const ge *q);           /* Ristretto equality */
int  ge_is_identity(const ge *p);
void ge_compress(uint8_t s[32], const ge *p);
int  ge_decompress(ge *p, const uint8_t s[32]); /* 1 on success */
void ge_elligator(ge *p, const fe *r0);
void ge_from_uniform_bytes(ge *p, const uint8_t b[64]);
void ge_scalarmult(ge *r, const sc *k, const ge *p);   /* variable time, simple double-and-add */
void ge_basepoint(ge *p);
#endif
