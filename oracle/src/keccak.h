/* ORACLE (test infrastructure). Keccak-f[1600], SHA3-512, SHAKE256 (FIPS 202). */
#ifndef ORACLE_KECCAK_H
#define ORACLE_KECCAK_H
#include <stdint.h>
#include <stddef.h>
void keccak_f1600(uint64_t st[25]);
void sha3_512(uint8_t out[64], const uint8_t *in, size_t len);
typedef struct { uint64_t st[25]; size_t pos; int squeezing; } shake256_ctx;
void shake256_init(shake256_ctx *c);
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len);
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len);
#endif
