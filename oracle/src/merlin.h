/* ORACLE (test infrastructure). Merlin v1.0 transcripts over STROBE-128 and the TranscriptRng.
 * Restates merlin 1.x strobe.rs / transcript.rs (crate not vendored; Cargo.toml:10 `merlin = "1.1.0"`).
 * Reference call sites: src/bin/prover.rs:52 (Transcript::new(filename)), :68, :211. */
#ifndef ORACLE_MERLIN_H
#define ORACLE_MERLIN_H
#include <stdint.h>
#include <stddef.h>
#include "sc.h"

typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } strobe128;   /* 203 bytes, packed */
typedef struct { strobe128 s; } merlin_transcript;
typedef struct { strobe128 s; } merlin_rng;

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t len);
void merlin_append(merlin_transcript *t, const char *label, const uint8_t *msg, size_t len);
void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t v);
void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t len);
void merlin_challenge_scalar(merlin_transcript *t, const char *label, sc *out);
/* TranscriptRngBuilder */
void merlin_rng_begin(merlin_rng *r, const merlin_transcript *t);
void merlin_rng_rekey(merlin_rng *r, const char *label, const uint8_t *w, size_t len);
void merlin_rng_finalize(merlin_rng *r, const uint8_t seed[32]);
void merlin_rng_fill(merlin_rng *r, uint8_t *out, size_t len);
void merlin_rng_scalar(merlin_rng *r, sc *out);
#endif
