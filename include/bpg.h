/* bpg.h - C ABI of the MI355X-native Bulletproofs R1CS prove path (libbpg_hip.so).
 *
 * Drop-in boundary for MarcKloter/bulletproofs_gadgets: the reference reaches this path through Rust crates
 * (bulletproofs fork, curve25519-dalek, merlin; reference Cargo.toml:8-20) that have no FFI today.  A Rust host
 * keeps the mini-language parser and the R1CS assembly and binds the functions of PART 1 (see INTEGRATION.md for
 * the `extern "C"` block); PART 2 is the same assembly surface offered natively (C++ inside the library) for hosts
 * without a Rust toolchain - the repo's Python harness, tests and bench.py drive it through ctypes.
 *
 * Conventions: scalars and compressed points are 32-byte little-endian encodings; the caller allocates every
 * output; every function returns a bpg_status (0 = ok) and never unwinds across the boundary; a context (and the
 * objects created from it) is used by one host thread at a time, different contexts are independent.
 * There is NO CPU fallback: bpg_ctx_create fails with BPG_ERR_DEVICE when no AMD GPU is visible.
 *
 * SIDE CHANNELS - read before proving with secrets on shared hardware.  Upstream computes A_I, A_O, S, T_k and every Pedersen commitment with
 * a CONSTANT-TIME multiscalar multiplication (Straus, fixed table lookups) because their scalars are secret: the witness a_L, a_R, a_O, the
 * blinding vectors s_L, s_R and the blinding factors.  This library does not: those scalars go through the bucket method (digit-indexed
 * scatter and gather) and, in the table-driven paths, through digit-indexed table reads with zero digits skipped; the host's Horner
 * recombination and the scalar arithmetic of host/scalar.hpp are variable-time as well.  Memory access pattern and running time therefore depend
 * on secret data.  The proof bytes are the same; the posture is that of a prover on a machine its operator trusts.
 */
#ifndef BPG_H
#define BPG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t bpg_status;
#define BPG_OK 0
#define BPG_ERR_INVALID_GENERATORS_LENGTH 1   /* R1CSError::InvalidGeneratorsLength */
#define BPG_ERR_FORMAT 2                      /* R1CSError::FormatError */
#define BPG_ERR_VERIFICATION 3                /* R1CSError::VerificationError */
#define BPG_ERR_INVALID_ARGUMENT 4
#define BPG_ERR_MISSING_ASSIGNMENT 5          /* R1CSError::MissingAssignment (reference src/cs_buffer.rs:104) */
#define BPG_ERR_GADGET 6                      /* R1CSError::GadgetError        (reference src/cs_buffer.rs:100) */
#define BPG_ERR_DEVICE 7
#define BPG_ERR_INTERNAL 8

/* dialect flags of the proof encoding / transcript (the fork's revision is unpinned) */
#define BPG_FLAG_COMPACT_1PHASE 1u            /* v2.0.0 encoding: version byte 0x00 + 11 points */
#define BPG_FLAG_NO_1PHASE_DOMSEP 2u          /* omit the "r1cs-1phase" domain separator */
/* Prover-only, opt-in, NOT upstream's derivation (off by default; never used for the headline benchmark): the 2n blinding scalars
 * s_L || s_R are expanded on the GPU instead of being drawn one by one from Merlin's serial TranscriptRng (which is 87 % of a
 * 2^20 proof).  After the three blinding scalars of A_I, A_O, S one more 64-byte block K is drawn from the TranscriptRng; scalar j is
 * from_bytes_mod_order_wide(SHAKE256("bpg blinding v1" || K || le64(j))[0..64)).  K depends on the transcript, the witness blindings
 * and the external 32 random bytes exactly as every upstream draw does.  The proof is an ordinary proof for any verifier. */
#define BPG_FLAG_EXPANDED_BLINDING 4u

/* variable encoding inside constraint terms: kind << 29 | index   (bulletproofs::r1cs::Variable) */
#define BPG_VAR_MULTIPLIER_LEFT 0u
#define BPG_VAR_MULTIPLIER_RIGHT 1u
#define BPG_VAR_MULTIPLIER_OUTPUT 2u
#define BPG_VAR_COMMITTED 3u
#define BPG_VAR_ONE 4u
#define BPG_TRANSCRIPT_STATE_BYTES 203        /* STROBE-128: 200 state bytes, pos, pos_begin, cur_flags */

typedef struct bpg_ctx bpg_ctx;
typedef struct bpg_circuit bpg_circuit;

/* Flattened R1CS instance = the state a bulletproofs::r1cs::Prover holds when prove() is called:
 * a_L, a_R, a_O (n x 32 B, reduced mod l) and the constraint list in CSR form with a de-duplicated coefficient table. */
typedef struct {
    uint64_t n, q, m, nnz, ncoef;
    const uint8_t *aL, *aR, *aO;
    const uint64_t *row_ptr;      /* q + 1 */
    const uint32_t *term_var;     /* nnz : kind << 29 | index */
    const uint32_t *term_coef;    /* nnz : index into coef */
    const uint8_t *coef;          /* ncoef x 32 */
} bpg_r1cs_instance;

/* milliseconds; filled when a non-NULL pointer is passed to a prove call.  FROZEN at nine doubles (72 bytes): the library writes exactly these and
 * the struct never grows (a caller-owned struct without a size field cannot).  Whether a proof took the shared-device kernel variants is reported by
 * bpg_ctx_last_shared_variants() and in bpg_profile_report's "_schedule". */
typedef struct {
    double rng_host, msm_aiao, msm_s, poly, ipa, total, ipa_msm, ipa_fold, ipa_sync;
} bpg_timings;

/* ---------------------------------------------------------------------------------------------------- PART 1: hot path */
/* ABI version of this header.  Rules: structs the CALLER allocates either carry a struct_size (bpg_config: fields are only ever added at the end and
 * read when struct_size covers them) or are frozen (bpg_timings, bpg_r1cs_instance, bpg_batch_item, bpg_term, bpg_lc); a field never changes type or
 * meaning; BPG_ABI_VERSION grows with every addition.  A host checks bpg_abi_version() >= the BPG_ABI_VERSION it was compiled against. */
#define BPG_ABI_VERSION 5u
uint32_t bpg_abi_version(void);
const char *bpg_strerror(bpg_status s);
const char *bpg_last_error(void);                                /* message of the calling thread's last failure */

/* replaces PedersenGens::default() + device selection            (reference src/bin/prover.rs:53).
 * bpg_ctx_create(device, out) = bpg_ctx_create_ex(device, NULL, out): the ONE-SHOT profile - what a process that proves once and exits wants
 * (the reference's prover binary, src/bin/prover.rs:47-100): at most 4 GB of precomputed tables beside the generators (3.0 GB at 2^20, 7 ms).
 * bpg_ctx_create_ex takes the choices a host has: a zeroed bpg_config with struct_size set means "defaults"; a field left at 0 / NULL falls
 * back to the environment variable named beside it, then to the profile's default.  Every setting gives the same proof bytes. */
#define BPG_PROFILE_DEFAULT 0u   /* BPG_PROFILE=oneshot|serving if set, else one-shot */
#define BPG_PROFILE_ONESHOT 1u   /* first generator fold on width-5 NAF tables of scalars cut in two (15 tables: 3.0 GB at 2^20); no 8-bit tail tables; table budget 4 GB */
#define BPG_PROFILE_SERVING 2u   /* a long-lived prover: width-8 NAF on scalars cut in four (51.5 GB of tables at 2^20, built once per device in 0.12 s,
                                    shared by the contexts of the process: -2.4 ms per 2^20 proof), 8-bit tail tables for circuits up to 2^14
                                    multipliers (17.2 GB); table budget 96 GB */
typedef struct {
    uint32_t struct_size;        /* size of this struct in bytes as the caller compiled it: lets the struct grow */
    uint32_t profile;            /* BPG_PROFILE_* */
    double table_budget_gb;      /* cumulative HBM all precomputed generator multiples of this process on the device may take (BPG_TABLE_GB); what does
                                    not fit is replaced by the next smaller table set, in the end by kernels that need none; 0 = profile default */
    uint32_t chain_workers;      /* bpg_ctx_set_chain_workers at creation (BPG_CHAIN_WORKERS); 0 = default 1 */
    uint32_t chain_lanes;        /* bpg_ctx_set_chain_lanes at creation (BPG_CHAIN_LANES); 0 = default 1 */
    int32_t blocking_sync;       /* 1: host threads sleep in stream waits (hipDeviceScheduleBlockingSync; device-wide, first context of the process
                                    decides) - for hosts that run many proving threads beside their chain threads; 0 (and 2, which one revision of this
                                    header used for it): spin; -1 = BPG_SYNC_BLOCKING (1 / 0) if set, else spin.  A zeroed struct therefore spins, which
                                    is also what an unset environment gives */
    const char *gens_cache_dir;  /* directory of the on-disk generator cache (BPG_GENS_CACHE_DIR); NULL = no cache */
} bpg_config;
int32_t bpg_device_count(void);          /* AMD GPUs visible to the process (0: none - every bpg_ctx_create then fails with BPG_ERR_DEVICE) */
bpg_status bpg_ctx_create(int32_t device, bpg_ctx **out);
bpg_status bpg_ctx_create_ex(int32_t device, const bpg_config *config /* NULL = defaults */, bpg_ctx **out);
void bpg_ctx_destroy(bpg_ctx *ctx);
bpg_status bpg_pedersen_bases(bpg_ctx *ctx, uint8_t B[32], uint8_t B_blinding[32]);

/* replaces BulletproofGens::new(capacity, 1)                     (reference src/bin/prover.rs:92); tables stay in HBM.
   Contexts of one process on one device share the tables of a capacity (derived once, immutable, freed with the last context using them). */
bpg_status bpg_gens_ensure(bpg_ctx *ctx, uint64_t capacity);
bpg_status bpg_gens_export(bpg_ctx *ctx, uint64_t first, uint64_t count, uint8_t *G_out, uint8_t *H_out);

/* replaces the point part of Prover::commit(v, v_blinding)       (reference src/gadget.rs:31, src/commitments.rs:27,39):
 * out[i] = compress(v[i]*B + blind[i]*B_blinding); v may be an unreduced Scalar::from_bits value */
bpg_status bpg_pedersen_commit(bpg_ctx *ctx, uint64_t k, const uint8_t *v, const uint8_t *blind, uint8_t *out);

/* replaces Prover::prove(&bp_gens) + R1CSProof::to_bytes()        (reference src/bin/prover.rs:93,97).
 * transcript_state: Merlin state after Transcript::new(label), Prover::new and every "V" append; updated in place.
 * rng_seed replaces the 32 bytes upstream draws from thread_rng(). proof_len: in = capacity, out = bytes written. */
bpg_status bpg_r1cs_upload(bpg_ctx *ctx, const bpg_r1cs_instance *inst, bpg_circuit **out);
void bpg_r1cs_free(bpg_ctx *ctx, bpg_circuit *c);
bpg_status bpg_r1cs_prove_resident(bpg_ctx *ctx, bpg_circuit *c, uint8_t transcript_state[BPG_TRANSCRIPT_STATE_BYTES],
                                   uint64_t m, const uint8_t *v_blinding, const uint8_t rng_seed[32], uint32_t flags,
                                   uint8_t *proof_out, uint64_t *proof_len, bpg_timings *timings);
bpg_status bpg_r1cs_prove(bpg_ctx *ctx, const bpg_r1cs_instance *inst, uint8_t transcript_state[BPG_TRANSCRIPT_STATE_BYTES],
                          uint64_t m, const uint8_t *v_blinding, const uint8_t rng_seed[32], uint32_t flags,
                          uint8_t *proof_out, uint64_t *proof_len);
uint64_t bpg_proof_size(uint64_t n_multipliers, uint32_t flags);

/* replaces R1CSProof::from_bytes + Verifier::verify(&proof, &pc_gens, &bp_gens)   (reference src/bin/verifier.rs:64,89-90).
 * inst: the verifier-side instance (aL/aR/aO NULL, constraints as assembled with None assignments); transcript_state: Merlin
 * state after Verifier::new and every "V" append (updated in place); V: the m commitments; seed replaces the verifier's
 * thread_rng() draw. Returns BPG_OK, BPG_ERR_VERIFICATION, BPG_ERR_FORMAT or BPG_ERR_INVALID_GENERATORS_LENGTH. */
bpg_status bpg_r1cs_verify(bpg_ctx *ctx, const bpg_r1cs_instance *inst, uint8_t transcript_state[BPG_TRANSCRIPT_STATE_BYTES],
                           uint64_t m, const uint8_t *V, const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags);

/* the same on an uploaded circuit (prover-side or verifier-side upload): a verifier that checks many proofs of one circuit
 * keeps the constraint matrix in HBM. */
bpg_status bpg_r1cs_verify_resident(bpg_ctx *ctx, bpg_circuit *circuit, uint8_t transcript_state[BPG_TRANSCRIPT_STATE_BYTES],
                                    uint64_t m, const uint8_t *V, const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags);

/* A batch of INDEPENDENT proofs on one GPU.  One proof keeps the device busy for ~45 ms of 340 (the rest is the host's serial Merlin
 * TranscriptRng chain, upstream-exact), so a pool of `workers` engine contexts + host threads proves items concurrently: the chain of
 * one proof overlaps the kernels of the others (8 workers: ~5x the single-proof rate on a 2^20 circuit).  Each item is what
 * bpg_r1cs_prove takes; items are independent (own transcript, witness, seed); results are byte-identical to proving them one by
 * one.  status_out[i] receives the bpg_status of item i; the call returns BPG_OK when every item succeeded, else the first failure. */
typedef struct bpg_pool bpg_pool;
typedef struct {
    const bpg_r1cs_instance *inst;
    uint8_t *transcript_state;            /* 203 B, updated in place */
    uint64_t m; const uint8_t *v_blinding;
    const uint8_t *rng_seed;              /* 32 B */
    uint32_t flags;
    uint8_t *proof_out; uint64_t *proof_len;   /* in = capacity, out = bytes written */
} bpg_batch_item;
bpg_status bpg_pool_create(int32_t device, uint32_t workers, uint64_t gens_capacity, bpg_pool **out);
bpg_status bpg_pool_create_ex(int32_t device, uint32_t workers, uint64_t gens_capacity, const bpg_config *config /* of every context; NULL = defaults */, bpg_pool **out);
void bpg_pool_destroy(bpg_pool *pool);
bpg_status bpg_pool_prove(bpg_pool *pool, uint64_t count, const bpg_batch_item *items, bpg_status *status_out);

/* measurement hooks (bench.py): HIP events on the engine's own stream. mode 0 off, 1 = dominant kernel only, 2 = all kernels;
 * report = JSON text {kernel: {count, total_ms, alg_bytes, device_bytes, field_mults}} accumulated since the last set. */
bpg_status bpg_profile_set(bpg_ctx *ctx, int32_t mode);
bpg_status bpg_profile_report(bpg_ctx *ctx, char *out, uint64_t cap);
bpg_status bpg_bench_fe_mul(bpg_ctx *ctx, uint32_t iters, double *mults_per_second);

/* test hook: device field arithmetic on n pairs of raw 256-bit values; op 0 mul, 1 sq, 2 add, 3 sub, 4 invert, 5 mixed chain; canonical output;
 * op 6: the scalar-field Montgomery product a b / 2^256 mod l of the device (one operand below l), eight raw words out */
bpg_status bpg_test_fe_ops(bpg_ctx *ctx, int32_t op, uint64_t n, const uint8_t *a, const uint8_t *b, uint8_t *out);
/* test hook: the next blinding stream started on ctx (bpg_blinding_begin) records a failed upload of its first block, as a failing hipMemcpyAsync
 * would: the prove that adopts it must fail with BPG_ERR_DEVICE instead of reading a stale device slab */
bpg_status bpg_test_fail_next_upload(bpg_ctx *ctx);
/* test hook: the copies of the next blinding stream started on ctx are skipped WITHOUT an error (a dropped DMA): the device slab keeps the pattern its
 * blocks were marked with when the slab changed owner, the conversion kernel notices, and the prove that adopts the stream fails with BPG_ERR_DEVICE */
bpg_status bpg_test_drop_next_upload(bpg_ctx *ctx);
/* diagnostics: bytes of precomputed generator multiples (fold tables + wide tail tables) this process holds on the context's device */
uint64_t bpg_table_bytes(bpg_ctx *ctx);
/* diagnostics: 1 when the last bpg_r1cs_prove* / bpg_prover_prove on ctx took the kernel variants for a shared device (another prove() was in flight on the
 * device when it was entered; decided once per proof), else 0 - replay a failing proof with BPG_FOLD_ADAPT=2 (always) or 0 (never) */
int32_t bpg_ctx_last_shared_variants(bpg_ctx *ctx);
/* test hook: compress(sum s_i*G[first+i] + t_i*H[first+i]) through the bucket-method MSM kernels */
bpg_status bpg_msm_gens(bpg_ctx *ctx, uint64_t first, uint64_t count, const uint8_t *s, const uint8_t *t, uint8_t out[32]);

/* ---------------------------------------------------------------------------------------------------- PART 2: host mirror
 * merlin::Transcript, bulletproofs::r1cs::{Prover, Verifier}, and the reference's Gadget trait with BoundsCheck,
 * MimcHash256 and MerkleTree256 (reference src/gadget.rs:6-59 and the gadget modules), implemented in C++. */
typedef struct bpg_transcript bpg_transcript;
typedef struct bpg_prover bpg_prover;
typedef struct bpg_verifier bpg_verifier;
typedef struct bpg_gadget bpg_gadget;
typedef struct { uint32_t var; uint8_t coeff[32]; } bpg_term;       /* (Variable, Scalar); var = kind << 29 | index */
typedef struct { const bpg_term *terms; uint64_t n; } bpg_lc;        /* bulletproofs::r1cs::LinearCombination */

bpg_status bpg_transcript_new(const uint8_t *label, uint64_t len, bpg_transcript **out);        /* Transcript::new */
void bpg_transcript_free(bpg_transcript *t);
bpg_status bpg_transcript_append_message(bpg_transcript *t, const char *label, const uint8_t *msg, uint64_t len);
bpg_status bpg_transcript_challenge_bytes(bpg_transcript *t, const char *label, uint8_t *out, uint64_t len);
bpg_status bpg_transcript_state(const bpg_transcript *t, uint8_t out[BPG_TRANSCRIPT_STATE_BYTES]);

bpg_status bpg_prover_new(bpg_ctx *ctx, bpg_transcript *t, bpg_prover **out);                   /* Prover::new */
void bpg_prover_free(bpg_prover *p);
/* test hook: from now on the commitments of p are 32 hash bytes of (value, blinding) made on the host - NOT group elements - so that a device-less
 * prover (bpg_prover_new with ctx = NULL) can run a file driver's parsing and gadget assembly under sanitizers / a fuzzer; bpg_prover_prove stays
 * refused (BPG_ERR_DEVICE).  Refused with BPG_ERR_INVALID_ARGUMENT on a prover that HAS a device context: such a prover never hands out anything
 * but Pedersen commitments */
bpg_status bpg_test_prover_stub_commitments(bpg_prover *p);
bpg_status bpg_prover_commit(bpg_prover *p, const uint8_t v[32], const uint8_t blind[32], uint8_t com_out[32], uint32_t *var_out);
bpg_status bpg_prover_commit_many(bpg_prover *p, uint64_t k, const uint8_t *v, const uint8_t *blind, uint8_t *coms_out, uint32_t *vars_out);
/* Prover::commit for a host that computes its Pedersen commitments elsewhere (its own PedersenGens, or a prover created without a device
 * context): registers (v, blind) as the next committed variable and appends the given 32-byte commitment to the transcript as "V". */
bpg_status bpg_prover_commit_precomputed(bpg_prover *p, const uint8_t v[32], const uint8_t blind[32], const uint8_t com[32], uint32_t *var_out);
/* Extension - every commitment of a prover in ONE kernel launch (the reference commits one value at a time, src/gadget.rs:27-35,
 * src/lalrpop/assignment_parser.rs:152-169; a 512-leaf tree makes a thousand of them).  While deferral is on, the commit calls above and
 * Gadget::setup register their variables and return ALL-ZERO commitment bytes; the flush computes every pending commitment at once and
 * appends them to the transcript in the order they were made, which is the transcript the per-call path gives.  Prove, the start of the
 * blinding chain and the instance export flush first; a host that reads the transcript itself flushes before it does.  The commitment
 * of committed variable `index` (0-based, in commit order) can be read once it has been flushed. */
bpg_status bpg_prover_defer_commitments(bpg_prover *p, int32_t on);     /* turning it off flushes */
bpg_status bpg_prover_flush_commitments(bpg_prover *p);
bpg_status bpg_prover_commitment(bpg_prover *p, uint64_t index, uint8_t out[32]);
uint64_t bpg_prover_num_constraints(const bpg_prover *p);                                       /* fork getter, prover.rs:89 */
uint64_t bpg_prover_num_multiplications(const bpg_prover *p);                                   /* fork getter, prover.rs:92 */
uint64_t bpg_prover_num_committed(const bpg_prover *p);
/* ConstraintSystem methods (trait visible at reference src/cs_buffer.rs:89-113) */
bpg_status bpg_prover_multiply(bpg_prover *p, const bpg_lc *left, const bpg_lc *right, uint32_t vars_out[3]);
bpg_status bpg_prover_allocate_multiplier(bpg_prover *p, int32_t has_assignment, const uint8_t l[32], const uint8_t r[32], uint32_t vars_out[3]);
bpg_status bpg_prover_allocate(bpg_prover *p, int32_t has_assignment, const uint8_t s[32], uint32_t *var_out);
bpg_status bpg_prover_constrain(bpg_prover *p, const bpg_lc *lc);
/* borrowed view of the assembled instance (valid until the prover is next mutated or freed) */
bpg_status bpg_prover_instance(bpg_prover *p, bpg_r1cs_instance *out, const uint8_t **v_out, const uint8_t **v_blinding_out);
/* Extension (no upstream counterpart; the proof bytes do not change): start drawing the blinding scalars of the coming prove() now.
 * Upstream's Prover::prove builds its TranscriptRng from the transcript after the last commitment (+ the "m" suffix), the commitment
 * blindings and thread_rng() - not from the constraints - and then draws 2n + 3 scalars serially (0.30 s of a 0.34 s proof at n = 2^20).
 * Called once every commitment has been made, this starts that chain on a host thread while the caller keeps assembling constraints;
 * bpg_prover_prove / bpg_r1cs_prove(_resident) on the same context use the stream iff transcript state, blindings and rng_seed are still
 * the same and n <= max_multipliers, and silently draw afresh otherwise (another commitment, another seed, BPG_FLAG_EXPANDED_BLINDING).
 * Streams of one context are drawn in the order of the calls by the context's chain worker - ONE AT A TIME with its single default thread
 * (bpg_ctx_set_chain_workers adds threads); workers + 1 are alive at most (one more call retires the oldest), so a sequence of proofs can have the chain of proof i+1 drawn while the kernels of proof i
 * run: call bpg_blinding_begin for proof i+1, then prove proof i.  max_multipliers sizes a pinned host buffer of 128 bytes per multiplier.
 * A deterministic rng_seed is for tests and benchmarks; production callers pass 32 fresh random bytes per proof (upstream: thread_rng()). */
bpg_status bpg_prover_start_blinding(bpg_prover *p, const uint8_t rng_seed[32], uint64_t max_multipliers);
bpg_status bpg_blinding_begin(bpg_ctx *ctx, const uint8_t transcript_state[BPG_TRANSCRIPT_STATE_BYTES] /* after every "V" append */, uint64_t m,
                              const uint8_t *v_blinding /* m x 32 */, const uint8_t rng_seed[32], uint64_t max_multipliers);
/* Threads of the context's chain worker (default 1, or BPG_CHAIN_WORKERS): with `workers` threads that many queued blinding streams are drawn
 * side by side and workers + 1 may be alive, so a host that proves a SEQUENCE of independent proofs keeps bpg_blinding_begin `workers` proofs
 * ahead of the proof it is proving and the GPU, not one host core's Keccak chain, sets the pace.  Streams in flight are dropped by the call.
 * Each alive stream pins 128 bytes per multiplier of host memory. */
bpg_status bpg_ctx_set_chain_workers(bpg_ctx *ctx, uint32_t workers);
/* streams EACH chain thread draws in lockstep, 1..8 (default 1): with AVX-512 the sponges of up to eight queued streams sit in the 64-bit lanes of ZMM
 * registers and cost one core about what one costs it (EPYC 9575F: 193 ns per draw of all eight against 152 ns for one), so one thread keeps
 * up to eight chains going - a chain alone gets ~25 % slower, a core's chain throughput six times higher; workers * lanes + 1 streams may be alive.
 * Streams in flight are dropped by this call, like bpg_ctx_set_chain_workers. */
bpg_status bpg_ctx_set_chain_lanes(bpg_ctx *ctx, uint32_t lanes);
/* A chain pool: host threads that draw the blinding chains of EVERY context attached to it, thread k up to lanes[k] (1..8) of them in lockstep.  A
 * host that proves on several contexts of a GPU (proving streams) sizes ONE pool for its cores instead of a worker per context: e.g. 14 threads with
 * one lane (a chain alone takes 0.30 s at 2^20) and one thread with 6 lanes (0.38 s each) draw twenty chains at once on 15 cores.  An attached
 * context may have max_streams blinding streams alive (bpg_blinding_begin retires the oldest beyond that); bpg_ctx_set_chain_workers / _lanes detach.
 * The pool should outlive its contexts' use of it: detach (pool = NULL) or destroy the contexts first.  (bpg_chain_pool_destroy lets the pool finish
 * what is queued; a context that is still attached afterwards goes back to its own chain worker at its next bpg_blinding_begin.) */
typedef struct bpg_chain_pool bpg_chain_pool;
bpg_status bpg_chain_pool_create(uint32_t threads, const uint32_t *lanes /* threads entries, NULL = 1 each */, bpg_chain_pool **out);
void bpg_chain_pool_destroy(bpg_chain_pool *pool);
bpg_status bpg_ctx_attach_chain_pool(bpg_ctx *ctx, bpg_chain_pool *pool /* NULL = detach */, uint32_t max_streams);
int32_t bpg_chain_cpu(bpg_ctx *ctx);   /* diagnostics: host core the chain worker last ran on, -1 = no stream drawn yet */
bpg_status bpg_prover_prove(bpg_prover *p, uint64_t gens_capacity, const uint8_t rng_seed[32], uint32_t flags,
                            uint8_t *proof_out, uint64_t *proof_len, bpg_timings *timings);

bpg_status bpg_verifier_new(bpg_transcript *t, bpg_verifier **out);                             /* Verifier::new */
void bpg_verifier_free(bpg_verifier *v);
bpg_status bpg_verifier_commit(bpg_verifier *v, const uint8_t com[32], uint32_t *var_out);       /* Verifier::commit */
uint64_t bpg_verifier_num_vars(const bpg_verifier *v);                                          /* fork getter, verifier.rs:89 */
bpg_status bpg_verifier_instance(bpg_verifier *v, bpg_r1cs_instance *out, const uint8_t **commitments_out);
/* Verifier::verify(&proof, &pc_gens, &bp_gens) on the GPU of ctx */
bpg_status bpg_verifier_verify(bpg_verifier *v, bpg_ctx *ctx, uint64_t gens_capacity, const uint8_t *proof, uint64_t proof_len,
                               const uint8_t seed[32], uint32_t flags);

bpg_status bpg_bounds_check_new(const uint8_t *min_be, uint64_t min_len, const uint8_t *max_be, uint64_t max_len, bpg_gadget **out);
bpg_status bpg_mimc_hash256_new(const bpg_lc *image, bpg_gadget **out);
/* pattern: the tree syntax of the .gadgets grammar with W / I leaves, e.g. "((W I) (I W))"; at most 64 levels of nesting (BPG_ERR_INVALID_ARGUMENT beyond) */
bpg_status bpg_merkle_tree256_new(const bpg_lc *root, const bpg_lc *instance_vars, uint64_t n_inst, const bpg_lc *witness_vars,
                                  uint64_t n_wit, const char *pattern, bpg_gadget **out);
/* the remaining gadgets of the reference; assignment pointers may be NULL on the verifier side */
bpg_status bpg_equality_new(const bpg_lc *right_hand, uint64_t n, bpg_gadget **out);
bpg_status bpg_inequality_new(const bpg_lc *right_hand, uint64_t n, const uint8_t *right_assignment /* n x 32 or NULL */, bpg_gadget **out);
bpg_status bpg_less_than_new(const bpg_lc *left, const uint8_t *left_assignment, const bpg_lc *right, const uint8_t *right_assignment, bpg_gadget **out);
bpg_status bpg_set_membership_new(const bpg_lc *value, const uint8_t *value_assignment, const bpg_lc *instance_vars, uint64_t n_inst,
                                  const uint8_t *instance_assignments /* n_inst x 32 or NULL */, bpg_gadget **out);
void bpg_gadget_free(bpg_gadget *g);
/* Gadget::setup: derived = preprocess(witnesses); one commitment each. *n_derived: in = capacity, out = count */
bpg_status bpg_gadget_setup(bpg_gadget *g, bpg_prover *p, const uint8_t *witness_scalars, uint64_t n_wit, const uint8_t *blindings,
                            uint64_t n_blind, uint8_t *coms_out, uint8_t *derived_scalars_out, uint32_t *derived_vars_out, uint64_t *n_derived);
/* Gadget::preprocess alone: the derived scalars a host commits itself (bpg_prover_commit / bpg_prover_commit_precomputed), in order */
bpg_status bpg_gadget_preprocess(bpg_gadget *g, const uint8_t *witness_scalars, uint64_t n_wit, uint8_t *derived_scalars_out, uint64_t *n_derived);
bpg_status bpg_gadget_prove(bpg_gadget *g, bpg_prover *p, const uint32_t *vars, uint64_t n_vars, const uint8_t *derived_scalars,
                            const uint32_t *derived_vars, uint64_t n_derived);
bpg_status bpg_gadget_verify(bpg_gadget *g, bpg_verifier *v, const uint32_t *vars, uint64_t n_vars, const uint32_t *derived_vars, uint64_t n_derived);
/* OR blocks (reference src/cs_buffer.rs, src/or/or_conjunction.rs): a recording constraint system whose multiplier numbering
 * starts at the parent's current multiplier count; rewind() closes a clause; bpg_or_* replays the clauses into the parent. */
typedef struct bpg_buffer bpg_buffer;
bpg_status bpg_buffer_new(uint64_t first_multiplier, int32_t prover_side, bpg_buffer **out);
void bpg_buffer_free(bpg_buffer *b);
bpg_status bpg_buffer_rewind(bpg_buffer *b);
uint64_t bpg_buffer_next_multiplier(const bpg_buffer *b);
bpg_status bpg_gadget_prove_buffered(bpg_gadget *g, bpg_buffer *b, const uint32_t *vars, uint64_t n_vars, const uint8_t *derived_scalars,
                                     const uint32_t *derived_vars, uint64_t n_derived);
bpg_status bpg_gadget_verify_buffered(bpg_gadget *g, bpg_buffer *b, const uint32_t *vars, uint64_t n_vars, const uint32_t *derived_vars, uint64_t n_derived);
bpg_status bpg_or_prover(bpg_prover *main, const bpg_buffer *b);
bpg_status bpg_or_verifier(bpg_verifier *main, const bpg_buffer *b);
bpg_status bpg_or_buffer(bpg_buffer *parent, const bpg_buffer *b);
/* utils::range_proof(cs, x, n, x_assignment) on a prover (assignment given) or a verifier (none) */
bpg_status bpg_range_proof_prove(bpg_prover *p, const bpg_lc *x, uint32_t n_bits, const uint8_t assignment[32]);
bpg_status bpg_range_proof_verify(bpg_verifier *v, const bpg_lc *x, uint32_t n_bits);
/* mimc::mimc_hash(preimage) -> Scalar bytes (little-endian); conversions */
bpg_status bpg_mimc_hash(const uint8_t *preimage, uint64_t len, uint8_t out[32]);
bpg_status bpg_be_to_scalars(const uint8_t *be, uint64_t len, uint8_t *out, uint64_t *n_out);   /* conversions::be_to_scalars */
/* test hook: `count` 64-byte TranscriptRng draws (merlin build_rng().rekey_with_witness_bytes("v_blinding")*.finalize(seed)), after
 * `skip` draws through the generic STROBE operations; bulk != 0 uses the prover's in-register bulk path. Same bytes either way. */
bpg_status bpg_rng_draws(const uint8_t transcript_state[203], uint64_t m, const uint8_t *v_blinding, const uint8_t rng_seed[32],
                         uint64_t skip, uint64_t count, int32_t bulk, uint8_t *out);
/* the same for `lanes` (1..8) generators with seeds rng_seeds[lanes][32] drawn in lockstep (csrc/host/merlin.hpp: eight sponges in the lanes of ZMM
   registers); lane v skips skip[v] draws first; out = [lanes][count][64] */
bpg_status bpg_rng_draws_multi(const uint8_t transcript_state[203], uint64_t m, const uint8_t *v_blinding, uint32_t lanes, const uint8_t *rng_seeds,
                               const uint64_t *skip, uint64_t count, uint8_t *out);
/* host Keccak-f[1600] self-check: runs the scalar and (when the CPU has AVX-512F+VL) both vector implementations on `rounds` chained
 * states derived from seed; *impl_out = the active one (0 scalar, 1 planes-in-ZMM, 2 lanes-in-XMM; chosen by a start-up calibration).
 * Fails with BPG_ERR_INTERNAL on a mismatch. */
bpg_status bpg_keccak_selftest(uint64_t seed, uint32_t rounds, int32_t *impl_out, double *ns_per_permutation);
/* host scalar arithmetic (curve25519_dalek::Scalar semantics), exposed for tests: op 0 add, 1 sub, 2 mul, 3 invert, 4 reduce, 5 from_wide(a = 64 B) */
bpg_status bpg_scalar_op(int32_t op, const uint8_t *a, const uint8_t *b, uint8_t out[32]);

#ifdef __cplusplus
}
#endif
#endif
