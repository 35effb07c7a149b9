/* ORACLE public C API (loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY).
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE.  A single-threaded CPU restatement of the Bulletproofs R1CS
 * prove/verify path that the reference reaches through un-vendored crates (Cargo.toml:8-20):
 *   Prover::new / commit / prove     reference call sites src/bin/prover.rs:52-54,92-97, src/gadget.rs:31,
 *                                    src/commitments.rs:27,39
 *   Verifier::verify                 src/bin/verifier.rs:89-90
 * Algorithms follow dalek bulletproofs (develop, 2019-H2) r1cs/prover.rs, r1cs/verifier.rs,
 * inner_product_proof.rs, generators.rs, transcript.rs and curve25519-dalek 1.x scalar_mul/{straus,pippenger}.rs
 * as summarised in SURVEY.md Appendix A.
 *
 * PARITY STATUS: "parity unpinned" at the .proof/.coms byte level - the reference stores no proof bytes and
 * randomises its blindings with thread_rng (src/gadget.rs:31).  What IS pinned (tests/test_oracle_*.py):
 * RFC 9496 vectors, the dalek Pedersen base, Merlin's published vector, every MiMC/Merkle KAT in the reference.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>

#define ORC_OK 0
#define ORC_ERR_GENS_LENGTH 1       /* R1CSError::InvalidGeneratorsLength */
#define ORC_ERR_FORMAT 2            /* R1CSError::FormatError */
#define ORC_ERR_VERIFY 3            /* R1CSError::VerificationError */
#define ORC_ERR_ARG 4

/* dialect flags; shared numbering with include/bpg.h */
#define ORC_FLAG_COMPACT_1PHASE 1u  /* v2.0.0 encoding: version byte + 11 points */
#define ORC_FLAG_NO_1PHASE_DOMSEP 2u
#define ORC_FLAG_EXPANDED_BLINDING 4u /* NOT upstream: s_L, s_R = SHAKE256("bpg blinding v1" || K || le64(j)) with K one TranscriptRng draw (include/bpg.h) */
#define ORC_FLAG_FAST_MSM 0x100u    /* oracle-only: Pippenger instead of upstream's constant-time Straus (same group elements) */

/* variable encoding inside constraint terms: kind << 29 | index */
#define ORC_VAR_MUL_LEFT 0u
#define ORC_VAR_MUL_RIGHT 1u
#define ORC_VAR_MUL_OUT 2u
#define ORC_VAR_COMMITTED 3u
#define ORC_VAR_ONE 4u

typedef struct orc_gens orc_gens;

typedef struct {
    uint64_t n, q, m, nnz, ncoef;
    const uint8_t *aL, *aR, *aO;   /* n x 32, may be NULL for verify */
    const uint64_t *row_ptr;       /* q + 1 */
    const uint32_t *term_var;      /* nnz */
    const uint32_t *term_coef;     /* nnz, index into coef */
    const uint8_t *coef;           /* ncoef x 32 */
} orc_circuit;

orc_gens *orc_gens_new(uint64_t capacity);
orc_gens *orc_gens_from_compressed(uint64_t capacity, const uint8_t *G, const uint8_t *H);
void orc_gens_free(orc_gens *g);
void orc_gens_export(const orc_gens *g, uint64_t first, uint64_t count, uint8_t *G_out, uint8_t *H_out);

void orc_pedersen_bases(uint8_t B[32], uint8_t B_blinding[32]);
void orc_pedersen_commit(uint8_t out[32], const uint8_t v[32], const uint8_t r[32]);
int  orc_msm(uint8_t out[32], uint64_t n, const uint8_t *scalars, const uint8_t *points, int algo); /* 0 ct-straus 1 vartime */

int orc_r1cs_prove(const orc_gens *g, uint8_t tstate[203], const orc_circuit *c, const uint8_t *v_blinding,
                   const uint8_t seed[32], uint32_t flags, uint8_t *proof, uint64_t *proof_len);
int orc_r1cs_verify(const orc_gens *g, uint8_t tstate[203], const orc_circuit *c, const uint8_t *V,
                    const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags);
/* 1 when a_L o a_R = a_O and every constraint evaluates to 0 on (aL,aR,aO,v) */
int orc_r1cs_satisfied(const orc_circuit *c, const uint8_t *v);

/* primitives exposed for unit tests */
void orc_transcript_init(uint8_t tstate[203], const uint8_t *label, uint64_t len);
void orc_transcript_append(uint8_t tstate[203], const char *label, const uint8_t *msg, uint64_t len);
void orc_transcript_append_u64(uint8_t tstate[203], const char *label, uint64_t v);
void orc_transcript_challenge(uint8_t tstate[203], const char *label, uint8_t *out, uint64_t len);
void orc_transcript_challenge_scalar(uint8_t tstate[203], const char *label, uint8_t out[32]);
void orc_rng_scalars(const uint8_t tstate[203], uint64_t m, const uint8_t *v_blinding, const uint8_t seed[32],
                     uint64_t count, uint8_t *out);
void orc_sc_wide(uint8_t out[32], const uint8_t in[64]);
void orc_sc_reduce(uint8_t out[32], const uint8_t in[32]);
void orc_sc_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_sc_add(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_sc_sub(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_sc_invert(uint8_t out[32], const uint8_t a[32]);
void orc_from_uniform(uint8_t out[32], const uint8_t in[64]);
int  orc_point_mul(uint8_t out[32], const uint8_t k[32], const uint8_t p[32]);
int  orc_point_add(uint8_t out[32], const uint8_t p[32], const uint8_t q[32]);
void orc_sha3_512(uint8_t out[64], const uint8_t *in, uint64_t len);
void orc_shake256(uint8_t *out, uint64_t outlen, const uint8_t *in, uint64_t len);
/* MiMC (reference src/mimc_hash/mimc.rs:61-97); output little-endian scalar bytes */
void orc_mimc_hash(uint8_t out[32], const uint8_t *preimage, uint64_t len);
void orc_mimc_sponge(uint8_t out[32], const uint8_t *blocks, uint64_t nblocks);
#endif
