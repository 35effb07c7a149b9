// extern "C" surface declared in include/bpg.h.  Translates C handles to the C++ host mirror (host/*.hpp) and the
// HIP engine (engine.hip); every exception is mapped to a bpg_status and a thread-local message.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstring>
#include <thread>
#include <memory>
#include <string>
#include "../../include/bpg.h"
#include "engine.hpp"
#include "host/gadgets.hpp"

using namespace bpg;

// ---------------------------------------------------------------------------------------- pieces of the host mirror
namespace bpg {

const std::vector<Scalar> &mimc_round_constants() {
    static const uint64_t RC[486][4] = {
#include "host/mimc_rc769.inc"
    };
    static const std::vector<Scalar> v = [] {
        std::vector<Scalar> out(486);
        for (int i = 0; i < 486; i++) out[i] = Scalar::from_bits(reinterpret_cast<const uint8_t *>(RC[i]));   // Scalar::from_bits(*constant), mimc.rs:69-71
        return out;
    }();
    return v;
}

std::pair<std::vector<uint8_t>, Variable> Prover::commit(const Scalar &v, const Scalar &v_blinding) {
    std::vector<uint8_t> com;
    std::vector<Variable> vars = commit_many({v}, {v_blinding}, com);
    return {com, vars[0]};
}

std::vector<Variable> Prover::commit_many(const std::vector<Scalar> &v, const std::vector<Scalar> &blind, std::vector<uint8_t> &coms_out) {
    if (v.size() != blind.size()) throw std::invalid_argument("commit_many: size mismatch");
    const size_t k = v.size();
    coms_out.assign(k * 32, 0);
    std::vector<Variable> vars;
    if (!k) return vars;
    if (!engine_ && !stub_commitments_) throw DeviceError("this prover has no device context: Pedersen commitments need the GPU engine");
    for (size_t i = 0; i < k; i++) {
        const uint32_t idx = (uint32_t)v_.size();
        v_.push_back(v[i]); vb_.push_back(blind[i].is_canonical() ? blind[i] : blind[i].reduced());
        vars.push_back(Variable{Variable::Committed, idx});
    }
    V_.resize(v_.size() * 32, 0);
    if (!deferred_) {
        const size_t before = v_.size() - k;
        try { flush_commitments(); }
        catch (...) { if (flushed_ <= before) { v_.resize(before); vb_.resize(before); V_.resize(before * 32); } throw; }     // a failed call registers nothing
        std::memcpy(coms_out.data(), &V_[32 * before], k * 32);
    }
    return vars;
}

// every commitment registered since the last flush: one k_pedersen launch, then the "V" appends in order (the per-call path's transcript)
void Prover::flush_commitments() {
    const size_t k = v_.size() - flushed_;
    if (!k) return;
    if (!engine_ && !stub_commitments_) throw DeviceError("this prover has no device context: Pedersen commitments need the GPU engine");
    std::vector<uint8_t> vb(k * 32), bb(k * 32);
    for (size_t i = 0; i < k; i++) { v_[flushed_ + i].to_bytes(&vb[32 * i]); vb_[flushed_ + i].to_bytes(&bb[32 * i]); }
    V_.resize(v_.size() * 32, 0);
    if (stub_commitments_) {      // test hook: hash bytes in place of group elements (sanitizer fuzzing of the drivers without a device)
        for (size_t i = 0; i < k; i++) {
            Shake256 sh; sh.absorb(reinterpret_cast<const uint8_t *>("bpg stub commitment"), 19); sh.absorb(&vb[32 * i], 32); sh.absorb(&bb[32 * i], 32);
            sh.squeeze(&V_[32 * (flushed_ + i)], 32);
        }
    } else
    engine_->pedersen_commit(k, vb.data(), bb.data(), &V_[32 * flushed_]);
    for (size_t i = 0; i < k; i++) t_->append_point("V", &V_[32 * (flushed_ + i)]);
    flushed_ = v_.size();
}

Variable Prover::commit_precomputed(const Scalar &v, const Scalar &v_blinding, const uint8_t com[32]) {
    flush_commitments();
    const uint32_t idx = (uint32_t)v_.size();
    v_.push_back(v); vb_.push_back(v_blinding.is_canonical() ? v_blinding : v_blinding.reduced());
    V_.insert(V_.end(), com, com + 32);
    t_->append_point("V", com);
    flushed_ = v_.size();
    return Variable{Variable::Committed, idx};
}

std::vector<uint8_t> Prover::prove(uint64_t gens_capacity, const uint8_t rng_seed[32], uint32_t flags) {
    if (!engine_) throw DeviceError("this prover has no device context: prove() needs the GPU engine");
    if (stub_commitments_) throw DeviceError("this prover holds stub (hash) commitments: prove() is refused");      // (test hook; cannot be set on a prover with an engine)
    flush_commitments();
    engine_->gens_ensure(gens_capacity);
    uint64_t N = 1; while (N < aL_.size()) N <<= 1;
    if (gens_capacity < N) throw R1CSException(R1CSError::InvalidGeneratorsLength, "generator capacity below padded circuit size");
    FlatCircuit f = flatten();
    DeviceCircuit *dc = engine_->upload(f);
    try {
        std::vector<uint8_t> proof = engine_->prove(dc, *t_, vb_, rng_seed, flags, nullptr);
        engine_->free_circuit(dc);
        return proof;
    } catch (...) { engine_->free_circuit(dc); throw; }
}

void Prover::start_blinding(const uint8_t rng_seed[32], uint64_t max_multipliers) {
    if (!engine_) throw DeviceError("this prover has no device context: start_blinding() needs the GPU engine");
    flush_commitments();
    engine_->blinding_begin(*t_, vb_, rng_seed, max_multipliers);
}

}  // namespace bpg

// ---------------------------------------------------------------------------------------- handles
struct bpg_ctx { Engine *engine; };
struct bpg_circuit { DeviceCircuit *dc; uint64_t n, m; };
struct bpg_transcript { Transcript t; };
struct bpg_prover { Prover *p; FlatCircuit flat; std::vector<uint8_t> v_bytes, vb_bytes; };
struct bpg_verifier { Verifier *v; FlatCircuit flat; };
struct bpg_gadget { std::unique_ptr<Gadget> g; };
struct bpg_buffer { OpBuffer b; };

static thread_local std::string g_last_error;

template <class F> static bpg_status guard(F &&f) {
    try { f(); g_last_error.clear(); return BPG_OK; }
    catch (const R1CSException &e) { g_last_error = e.what(); return (bpg_status)e.code; }
    catch (const DeviceError &e) { g_last_error = e.what(); return BPG_ERR_DEVICE; }
    catch (const std::invalid_argument &e) { g_last_error = e.what(); return BPG_ERR_INVALID_ARGUMENT; }
    catch (const std::out_of_range &e) { g_last_error = e.what(); return BPG_ERR_INVALID_ARGUMENT; }
    catch (const std::bad_alloc &) { g_last_error = "out of host memory"; return BPG_ERR_INTERNAL; }
    catch (const std::exception &e) { g_last_error = e.what(); return BPG_ERR_INTERNAL; }
    catch (...) { g_last_error = "unknown error"; return BPG_ERR_INTERNAL; }
}
#define REQUIRE(cond) do { if (!(cond)) throw std::invalid_argument("null or invalid argument: " #cond); } while (0)

static LinearCombination lc_from(const bpg_lc *lc) {
    REQUIRE(lc && (lc->n == 0 || lc->terms));
    LinearCombination out;
    for (uint64_t i = 0; i < lc->n; i++) {
        if ((lc->terms[i].var >> 29) > BPG_VAR_ONE) throw std::invalid_argument("linear combination holds an unknown variable kind");
        out.terms.emplace_back(Variable::unpack(lc->terms[i].var), Scalar::from_bits(lc->terms[i].coeff));
    }
    return out;
}
static void view(const FlatCircuit &f, bpg_r1cs_instance *o) {
    o->n = f.n; o->m = f.m; o->q = f.row_ptr.size() - 1; o->nnz = f.term_var.size(); o->ncoef = f.coef.size() / 32;
    o->aL = f.aL.empty() ? nullptr : f.aL.data(); o->aR = f.aR.empty() ? nullptr : f.aR.data(); o->aO = f.aO.empty() ? nullptr : f.aO.data();
    o->row_ptr = f.row_ptr.data(); o->term_var = f.term_var.data(); o->term_coef = f.term_coef.data(); o->coef = f.coef.data();
}
static FlatView as_view(const bpg_r1cs_instance *i, bool need_witness, bool drop_witness = false) {
    REQUIRE(i && i->row_ptr && (i->nnz == 0 || (i->term_var && i->term_coef)) && (i->ncoef == 0 || i->coef));
    const bool has_w = i->aL && i->aR && i->aO;
    REQUIRE(i->n == 0 || has_w || !need_witness);
    FlatView v; v.n = i->n; v.m = i->m; v.q = i->q; v.nnz = i->nnz; v.ncoef = i->ncoef;
    if (has_w && !drop_witness) { v.aL = i->aL; v.aR = i->aR; v.aO = i->aO; }
    v.row_ptr = i->row_ptr; v.term_var = i->term_var; v.term_coef = i->term_coef; v.coef = i->coef;
    return v;
}
extern "C" {

const char *bpg_strerror(bpg_status s) {
    switch (s) {
    case BPG_OK: return "ok";
    case BPG_ERR_INVALID_GENERATORS_LENGTH: return "invalid generators length";
    case BPG_ERR_FORMAT: return "format error";
    case BPG_ERR_VERIFICATION: return "verification error";
    case BPG_ERR_INVALID_ARGUMENT: return "invalid argument";
    case BPG_ERR_MISSING_ASSIGNMENT: return "missing assignment";
    case BPG_ERR_GADGET: return "gadget error";
    case BPG_ERR_DEVICE: return "device error";
    default: return "internal error";
    }
}
const char *bpg_last_error(void) { return g_last_error.c_str(); }
uint32_t bpg_abi_version(void) { return BPG_ABI_VERSION; }

static EngineConfig engine_config(const bpg_config *c) {
    EngineConfig e;
    if (!c) return e;
    // struct_size tells which fields the caller's header knows; the layout only ever grows at the end
    // a field is read when the caller's struct holds ALL of it (offset + size <= struct_size)
    const size_t sz = c->struct_size;
#define BPG_CFG_HAS(field) (sz >= offsetof(bpg_config, field) + sizeof(c->field))
    if (!BPG_CFG_HAS(profile)) throw std::invalid_argument("bpg_config.struct_size not set");
    e.profile = c->profile;
    if (BPG_CFG_HAS(table_budget_gb)) e.table_budget_gb = c->table_budget_gb;
    if (BPG_CFG_HAS(chain_workers)) e.chain_workers = c->chain_workers;
    if (BPG_CFG_HAS(chain_lanes)) e.chain_lanes = c->chain_lanes;
    if (BPG_CFG_HAS(blocking_sync)) e.blocking_sync = c->blocking_sync;
    if (BPG_CFG_HAS(gens_cache_dir) && c->gens_cache_dir) e.gens_cache_dir = c->gens_cache_dir;
#undef BPG_CFG_HAS
    return e;
}
bpg_status bpg_ctx_create_ex(int32_t device, const bpg_config *config, bpg_ctx **out) {
    return guard([&] { REQUIRE(out); *out = nullptr; Engine *e = new Engine(device, engine_config(config)); *out = new bpg_ctx{e}; });
}
bpg_status bpg_ctx_create(int32_t device, bpg_ctx **out) { return bpg_ctx_create_ex(device, nullptr, out); }
int32_t bpg_device_count(void) { return Engine::device_count(); }
bpg_status bpg_test_fail_next_upload(bpg_ctx *ctx) { return guard([&] { REQUIRE(ctx); ctx->engine->test_fail_next_upload(); }); }
bpg_status bpg_test_drop_next_upload(bpg_ctx *ctx) { return guard([&] { REQUIRE(ctx); ctx->engine->test_drop_next_upload(); }); }
uint64_t bpg_table_bytes(bpg_ctx *ctx) { return ctx ? ctx->engine->table_bytes() : 0; }
int32_t bpg_ctx_last_shared_variants(bpg_ctx *ctx) { return ctx && ctx->engine->last_shared_variants() ? 1 : 0; }
void bpg_ctx_destroy(bpg_ctx *ctx) { if (ctx) { delete ctx->engine; delete ctx; } }
bpg_status bpg_pedersen_bases(bpg_ctx *ctx, uint8_t B[32], uint8_t Bb[32]) { return guard([&] { REQUIRE(ctx && B && Bb); ctx->engine->pedersen_bases(B, Bb); }); }
bpg_status bpg_gens_ensure(bpg_ctx *ctx, uint64_t capacity) { return guard([&] { REQUIRE(ctx); ctx->engine->gens_ensure(capacity); }); }
bpg_status bpg_gens_export(bpg_ctx *ctx, uint64_t first, uint64_t count, uint8_t *G, uint8_t *H) {
    return guard([&] { REQUIRE(ctx && (count == 0 || (G && H))); ctx->engine->gens_export(first, count, G, H); });
}
bpg_status bpg_pedersen_commit(bpg_ctx *ctx, uint64_t k, const uint8_t *v, const uint8_t *blind, uint8_t *out) {
    return guard([&] { REQUIRE(ctx && (k == 0 || (v && blind && out))); ctx->engine->pedersen_commit(k, v, blind, out); });
}
bpg_status bpg_test_fe_ops(bpg_ctx *ctx, int32_t op, uint64_t n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    return guard([&] { REQUIRE(ctx && op >= 0 && op <= 6 && (n == 0 || (a && b && out))); ctx->engine->test_fe_ops(op, n, a, b, out); });
}
bpg_status bpg_msm_gens(bpg_ctx *ctx, uint64_t first, uint64_t count, const uint8_t *s, const uint8_t *t, uint8_t out[32]) {
    return guard([&] { REQUIRE(ctx && out && (count == 0 || (s && t))); ctx->engine->msm_gens(first, count, s, t, out); });
}

bpg_status bpg_profile_set(bpg_ctx *ctx, int32_t mode) { return guard([&] { REQUIRE(ctx && mode >= 0 && mode <= 2); ctx->engine->profile_set(mode); }); }
bpg_status bpg_profile_report(bpg_ctx *ctx, char *out, uint64_t cap) {
    return guard([&] { REQUIRE(ctx && out && cap); std::string r = ctx->engine->profile_report(); if (r.size() + 1 > cap) throw std::invalid_argument("profile_report: buffer too small"); std::memcpy(out, r.c_str(), r.size() + 1); });
}
bpg_status bpg_bench_fe_mul(bpg_ctx *ctx, uint32_t iters, double *out) { return guard([&] { REQUIRE(ctx && out && iters); *out = ctx->engine->bench_fe_mul(iters); }); }

uint64_t bpg_proof_size(uint64_t n, uint32_t flags) {
    uint64_t N = 1, lg = 0; while (N < n) { N <<= 1; lg++; }
    return ((flags & BPG_FLAG_COMPACT_1PHASE) ? 1 + 11 * 32 : 14 * 32) + (2 * lg + 2) * 32;
}

bpg_status bpg_r1cs_upload(bpg_ctx *ctx, const bpg_r1cs_instance *inst, bpg_circuit **out) {
    return guard([&] {
        REQUIRE(ctx && out); *out = nullptr;
        const FlatView f = as_view(inst, true);                  // no host copy: the instance goes to the device as it is
        DeviceCircuit *dc = ctx->engine->upload(f);
        *out = new bpg_circuit{dc, f.n, f.m};
    });
}
void bpg_r1cs_free(bpg_ctx *ctx, bpg_circuit *c) { if (ctx && c) { ctx->engine->free_circuit(c->dc); delete c; } }

static void copy_timings(const ProveTimings &t, bpg_timings *o) {
    if (!o) return;
    o->rng_host = t.rng_host; o->msm_aiao = t.msm_aiao; o->msm_s = t.msm_s; o->poly = t.poly; o->ipa = t.ipa; o->total = t.total;
    o->ipa_msm = t.ipa_msm; o->ipa_fold = t.ipa_fold; o->ipa_sync = t.ipa_sync;
}

bpg_status bpg_r1cs_prove_resident(bpg_ctx *ctx, bpg_circuit *c, uint8_t ts[BPG_TRANSCRIPT_STATE_BYTES], uint64_t m, const uint8_t *v_blinding,
                                   const uint8_t seed[32], uint32_t flags, uint8_t *proof_out, uint64_t *proof_len, bpg_timings *timings) {
    return guard([&] {
        REQUIRE(ctx && c && ts && seed && proof_out && proof_len && (m == 0 || v_blinding));
        if (m != c->m) throw std::invalid_argument("prove: m does not match the uploaded circuit");
        if (*proof_len < bpg_proof_size(c->n, flags)) throw std::invalid_argument("prove: proof buffer too small");
        uint64_t N = 1; while (N < c->n) N <<= 1;
        if (ctx->engine->gens_capacity() < N) throw R1CSException(R1CSError::InvalidGeneratorsLength, "generator capacity below padded circuit size (call bpg_gens_ensure)");
        Transcript T = Transcript::from_state(ts);
        std::vector<Scalar> vb(m);
        for (uint64_t i = 0; i < m; i++) vb[i] = Scalar::from_bytes_mod_order(v_blinding + 32 * i);
        ProveTimings tm;
        std::vector<uint8_t> proof = ctx->engine->prove(c->dc, T, vb, seed, flags, timings ? &tm : nullptr);
        std::memcpy(proof_out, proof.data(), proof.size()); *proof_len = proof.size();
        T.export_state(ts);
        copy_timings(tm, timings);
    });
}

bpg_status bpg_r1cs_prove(bpg_ctx *ctx, const bpg_r1cs_instance *inst, uint8_t ts[BPG_TRANSCRIPT_STATE_BYTES], uint64_t m, const uint8_t *v_blinding,
                          const uint8_t seed[32], uint32_t flags, uint8_t *proof_out, uint64_t *proof_len) {
    bpg_circuit *c = nullptr;
    bpg_status s = bpg_r1cs_upload(ctx, inst, &c);
    if (s != BPG_OK) return s;
    s = bpg_r1cs_prove_resident(ctx, c, ts, m, v_blinding, seed, flags, proof_out, proof_len, nullptr);
    std::string keep = g_last_error;
    bpg_r1cs_free(ctx, c);
    g_last_error = keep;
    return s;
}

bpg_status bpg_r1cs_verify(bpg_ctx *ctx, const bpg_r1cs_instance *inst, uint8_t ts[BPG_TRANSCRIPT_STATE_BYTES], uint64_t m, const uint8_t *V,
                           const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags) {
    return guard([&] {
        REQUIRE(ctx && ts && proof && seed && (m == 0 || V));
        const FlatView f = as_view(inst, false, true);           // verifier side: no assignments
        if (f.m != m) throw std::invalid_argument("verify: m does not match the instance");
        DeviceCircuit *dc = ctx->engine->upload(f);
        Transcript T = Transcript::from_state(ts);
        R1CSError e;
        try { e = ctx->engine->verify(dc, T, V, proof, proof_len, seed, flags); } catch (...) { ctx->engine->free_circuit(dc); throw; }
        ctx->engine->free_circuit(dc);
        T.export_state(ts);
        if (e != R1CSError::None) throw R1CSException(e, e == R1CSError::VerificationError ? "proof rejected" : e == R1CSError::FormatError ? "malformed proof" : "generator capacity below padded circuit size");
    });
}

bpg_status bpg_r1cs_verify_resident(bpg_ctx *ctx, bpg_circuit *c, uint8_t ts[BPG_TRANSCRIPT_STATE_BYTES], uint64_t m, const uint8_t *V,
                                    const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags) {
    return guard([&] {
        REQUIRE(ctx && c && ts && proof && seed && (m == 0 || V));
        if (m != c->m) throw std::invalid_argument("verify: m does not match the uploaded circuit");
        Transcript T = Transcript::from_state(ts);
        const R1CSError e = ctx->engine->verify(c->dc, T, V, proof, proof_len, seed, flags);
        T.export_state(ts);
        if (e != R1CSError::None) throw R1CSException(e, e == R1CSError::VerificationError ? "proof rejected" : e == R1CSError::FormatError ? "malformed proof" : "generator capacity below padded circuit size");
    });
}

// ---------------------------------------------------------------------------------------- batch pool
struct bpg_pool { std::vector<bpg_ctx *> ctxs; };
bpg_status bpg_pool_create(int32_t device, uint32_t workers, uint64_t gens_capacity, bpg_pool **out) { return bpg_pool_create_ex(device, workers, gens_capacity, nullptr, out); }
bpg_status bpg_pool_create_ex(int32_t device, uint32_t workers, uint64_t gens_capacity, const bpg_config *config, bpg_pool **out) {
    if (!out) return BPG_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (workers < 1 || workers > 64) { g_last_error = "pool: 1..64 workers"; return BPG_ERR_INVALID_ARGUMENT; }
    bpg_pool *pool = new bpg_pool();
    for (uint32_t k = 0; k < workers; k++) {
        bpg_ctx *c = nullptr;
        bpg_status s = bpg_ctx_create_ex(device, config, &c);
        if (s == BPG_OK && gens_capacity) s = bpg_gens_ensure(c, gens_capacity);
        if (s != BPG_OK) { std::string keep = g_last_error; if (c) bpg_ctx_destroy(c); bpg_pool_destroy(pool); g_last_error = keep; return s; }
        pool->ctxs.push_back(c);
    }
    *out = pool;
    return BPG_OK;
}
void bpg_pool_destroy(bpg_pool *pool) {
    if (!pool) return;
    for (bpg_ctx *c : pool->ctxs) bpg_ctx_destroy(c);
    delete pool;
}
bpg_status bpg_pool_prove(bpg_pool *pool, uint64_t count, const bpg_batch_item *items, bpg_status *status_out) {
    if (!pool || (count && !items)) return BPG_ERR_INVALID_ARGUMENT;
    std::atomic<uint64_t> next(0);
    std::vector<bpg_status> st(count, BPG_OK);
    std::vector<std::string> msg(count);
    auto work = [&](bpg_ctx *ctx) {
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= count) return;
            const bpg_batch_item &it = items[i];
            st[i] = (it.inst && it.transcript_state && it.rng_seed && it.proof_out && it.proof_len)
                        ? bpg_r1cs_prove(ctx, it.inst, it.transcript_state, it.m, it.v_blinding, it.rng_seed, it.flags, it.proof_out, it.proof_len)
                        : BPG_ERR_INVALID_ARGUMENT;
            if (st[i] != BPG_OK) msg[i] = g_last_error;            // g_last_error is per thread
        }
    };
    std::vector<std::thread> th;
    const size_t nthreads = std::min<uint64_t>(pool->ctxs.size(), count);
    for (size_t k = 1; k < nthreads; k++) th.emplace_back(work, pool->ctxs[k]);
    if (nthreads) work(pool->ctxs[0]);
    for (std::thread &t : th) t.join();
    bpg_status first = BPG_OK;
    for (uint64_t i = 0; i < count; i++) {
        if (status_out) status_out[i] = st[i];
        if (first == BPG_OK && st[i] != BPG_OK) { first = st[i]; g_last_error = "item " + std::to_string(i) + ": " + msg[i]; }
    }
    return first;
}

// ---------------------------------------------------------------------------------------- transcript
bpg_status bpg_transcript_new(const uint8_t *label, uint64_t len, bpg_transcript **out) {
    return guard([&] { REQUIRE(out && (len == 0 || label)); *out = new bpg_transcript{Transcript(label, len)}; });
}
void bpg_transcript_free(bpg_transcript *t) { delete t; }
bpg_status bpg_transcript_append_message(bpg_transcript *t, const char *label, const uint8_t *msg, uint64_t len) {
    return guard([&] { REQUIRE(t && label && (len == 0 || msg)); t->t.append_message(label, msg, len); });
}
bpg_status bpg_transcript_challenge_bytes(bpg_transcript *t, const char *label, uint8_t *out, uint64_t len) {
    return guard([&] { REQUIRE(t && label && out); t->t.challenge_bytes(label, out, len); });
}
bpg_status bpg_transcript_state(const bpg_transcript *t, uint8_t out[BPG_TRANSCRIPT_STATE_BYTES]) { return guard([&] { REQUIRE(t && out); t->t.export_state(out); }); }

// ---------------------------------------------------------------------------------------- prover / verifier
bpg_status bpg_prover_new(bpg_ctx *ctx, bpg_transcript *t, bpg_prover **out) {
    // ctx may be NULL: an assembly-only prover (multiply / constrain / witness synthesis); commit() and prove() then fail with BPG_ERR_DEVICE
    return guard([&] { REQUIRE(t && out); bpg_prover *p = new bpg_prover(); p->p = new Prover(ctx ? ctx->engine : nullptr, &t->t); *out = p; });
}
void bpg_prover_free(bpg_prover *p) { if (p) { delete p->p; delete p; } }
bpg_status bpg_test_prover_stub_commitments(bpg_prover *p) { return guard([&] { REQUIRE(p); p->p->test_stub_commitments(); }); }
bpg_status bpg_prover_commit(bpg_prover *p, const uint8_t v[32], const uint8_t blind[32], uint8_t com_out[32], uint32_t *var_out) {
    return bpg_prover_commit_many(p, 1, v, blind, com_out, var_out);
}
bpg_status bpg_prover_commit_many(bpg_prover *p, uint64_t k, const uint8_t *v, const uint8_t *blind, uint8_t *coms_out, uint32_t *vars_out) {
    return guard([&] {
        REQUIRE(p && (k == 0 || (v && blind && coms_out)));
        std::vector<Scalar> vs(k), bs(k);
        for (uint64_t i = 0; i < k; i++) { vs[i] = Scalar::from_bits(v + 32 * i); bs[i] = Scalar::from_bytes_mod_order(blind + 32 * i); }
        std::vector<uint8_t> coms;
        std::vector<Variable> vars = p->p->commit_many(vs, bs, coms);
        if (k) std::memcpy(coms_out, coms.data(), k * 32);
        if (vars_out) for (uint64_t i = 0; i < k; i++) vars_out[i] = vars[i].packed();
    });
}
bpg_status bpg_prover_defer_commitments(bpg_prover *p, int32_t on) { return guard([&] { REQUIRE(p); p->p->defer_commitments(on != 0); }); }
bpg_status bpg_prover_flush_commitments(bpg_prover *p) { return guard([&] { REQUIRE(p); p->p->flush_commitments(); }); }
bpg_status bpg_prover_commitment(bpg_prover *p, uint64_t index, uint8_t out[32]) {
    return guard([&] {
        REQUIRE(p && out);
        if (index >= p->p->num_flushed()) throw std::invalid_argument("commitment: no such committed variable (or its commitment is still deferred)");
        std::memcpy(out, p->p->commitment(index), 32);
    });
}
bpg_status bpg_prover_commit_precomputed(bpg_prover *p, const uint8_t v[32], const uint8_t blind[32], const uint8_t com[32], uint32_t *var_out) {
    return guard([&] {
        REQUIRE(p && v && blind && com);
        const Variable var = p->p->commit_precomputed(Scalar::from_bits(v), Scalar::from_bytes_mod_order(blind), com);
        if (var_out) *var_out = var.packed();
    });
}
uint64_t bpg_prover_num_constraints(const bpg_prover *p) { return p ? p->p->num_constraints() : 0; }
uint64_t bpg_prover_num_multiplications(const bpg_prover *p) { return p ? p->p->get_num_multiplications() : 0; }
uint64_t bpg_prover_num_committed(const bpg_prover *p) { return p ? p->p->num_committed() : 0; }

static void put_vars(const MulVars &mv, uint32_t out[3]) { if (out) { out[0] = mv.l.packed(); out[1] = mv.r.packed(); out[2] = mv.o.packed(); } }
bpg_status bpg_prover_multiply(bpg_prover *p, const bpg_lc *left, const bpg_lc *right, uint32_t vars_out[3]) {
    return guard([&] { REQUIRE(p); put_vars(p->p->multiply(lc_from(left), lc_from(right)), vars_out); });
}
bpg_status bpg_prover_allocate_multiplier(bpg_prover *p, int32_t has, const uint8_t l[32], const uint8_t r[32], uint32_t vars_out[3]) {
    return guard([&] {
        REQUIRE(p && (!has || (l && r)));
        Scalar a, b; if (has) { a = Scalar::from_bits(l); b = Scalar::from_bits(r); }
        put_vars(p->p->allocate_multiplier(has != 0, a, b), vars_out);
    });
}
bpg_status bpg_prover_allocate(bpg_prover *p, int32_t has, const uint8_t s[32], uint32_t *var_out) {
    return guard([&] { REQUIRE(p && (!has || s)); Variable v = p->p->allocate(has ? OptScalar(Scalar::from_bits(s)) : OptScalar()); if (var_out) *var_out = v.packed(); });
}
bpg_status bpg_prover_constrain(bpg_prover *p, const bpg_lc *lc) { return guard([&] { REQUIRE(p); p->p->constrain(lc_from(lc)); }); }

bpg_status bpg_prover_instance(bpg_prover *p, bpg_r1cs_instance *out, const uint8_t **v_out, const uint8_t **vb_out) {
    return guard([&] {
        REQUIRE(p && out);
        p->p->flush_commitments();
        p->flat = p->p->flatten();
        view(p->flat, out);
        const size_t m = p->p->num_committed();
        p->v_bytes.resize(m * 32 + 1); p->vb_bytes.resize(m * 32 + 1);
        for (size_t i = 0; i < m; i++) { p->p->v()[i].to_bytes(&p->v_bytes[32 * i]); p->p->v_blinding()[i].to_bytes(&p->vb_bytes[32 * i]); }
        if (v_out) *v_out = p->v_bytes.data();
        if (vb_out) *vb_out = p->vb_bytes.data();
    });
}

bpg_status bpg_prover_start_blinding(bpg_prover *p, const uint8_t seed[32], uint64_t max_multipliers) {
    return guard([&] { REQUIRE(p && seed); p->p->start_blinding(seed, max_multipliers); });
}

bpg_status bpg_blinding_begin(bpg_ctx *ctx, const uint8_t transcript_state[203], uint64_t m, const uint8_t *v_blinding, const uint8_t seed[32], uint64_t max_multipliers) {
    return guard([&] {
        REQUIRE(ctx && transcript_state && seed && (v_blinding || m == 0));
        std::vector<Scalar> vb(m);
        for (uint64_t i = 0; i < m; i++) vb[i] = Scalar::from_bytes_mod_order(v_blinding + 32 * i);
        ctx->engine->blinding_begin(Transcript::from_state(transcript_state), vb, seed, max_multipliers);
    });
}

bpg_status bpg_ctx_set_chain_workers(bpg_ctx *ctx, uint32_t workers) { return guard([&] { REQUIRE(ctx); ctx->engine->set_chain_workers(workers); }); }
bpg_status bpg_ctx_set_chain_lanes(bpg_ctx *ctx, uint32_t lanes) { return guard([&] { REQUIRE(ctx); ctx->engine->set_chain_lanes(lanes); }); }
int32_t bpg_chain_cpu(bpg_ctx *ctx) { return ctx ? ctx->engine->chain_cpu() : -1; }
struct bpg_chain_pool { ChainPool *pool; };
bpg_status bpg_chain_pool_create(uint32_t threads, const uint32_t *lanes, bpg_chain_pool **out) {
    return guard([&] {
        REQUIRE(out); *out = nullptr;
        REQUIRE(threads >= 1 && threads <= 256);
        std::vector<uint32_t> l(threads, 1u);
        if (lanes) l.assign(lanes, lanes + threads);
        *out = new bpg_chain_pool{new ChainPool(l)};
    });
}
void bpg_chain_pool_destroy(bpg_chain_pool *pool) { if (pool) { delete pool->pool; delete pool; } }
bpg_status bpg_ctx_attach_chain_pool(bpg_ctx *ctx, bpg_chain_pool *pool, uint32_t max_streams) {
    return guard([&] { REQUIRE(ctx); ctx->engine->attach_chain_pool(pool ? pool->pool : nullptr, pool ? max_streams : 1u); });
}

bpg_status bpg_prover_prove(bpg_prover *p, uint64_t gens_capacity, const uint8_t seed[32], uint32_t flags, uint8_t *proof_out, uint64_t *proof_len, bpg_timings *timings) {
    return guard([&] {
        REQUIRE(p && seed && proof_out && proof_len);
        if (*proof_len < bpg_proof_size(p->p->get_num_multiplications(), flags)) throw std::invalid_argument("prove: proof buffer too small");
        (void)timings;
        std::vector<uint8_t> proof = p->p->prove(gens_capacity, seed, flags);
        std::memcpy(proof_out, proof.data(), proof.size()); *proof_len = proof.size();
    });
}

bpg_status bpg_verifier_new(bpg_transcript *t, bpg_verifier **out) {
    return guard([&] { REQUIRE(t && out); bpg_verifier *v = new bpg_verifier(); v->v = new Verifier(&t->t); *out = v; });
}
void bpg_verifier_free(bpg_verifier *v) { if (v) { delete v->v; delete v; } }
bpg_status bpg_verifier_commit(bpg_verifier *v, const uint8_t com[32], uint32_t *var_out) {
    return guard([&] { REQUIRE(v && com); Variable x = v->v->commit(com); if (var_out) *var_out = x.packed(); });
}
uint64_t bpg_verifier_num_vars(const bpg_verifier *v) { return v ? v->v->get_num_vars() : 0; }
bpg_status bpg_verifier_instance(bpg_verifier *v, bpg_r1cs_instance *out, const uint8_t **commitments_out) {
    return guard([&] { REQUIRE(v && out); v->flat = v->v->flatten(); view(v->flat, out); if (commitments_out) *commitments_out = v->v->commitments().data(); });
}

bpg_status bpg_verifier_verify(bpg_verifier *v, bpg_ctx *ctx, uint64_t gens_capacity, const uint8_t *proof, uint64_t proof_len, const uint8_t seed[32], uint32_t flags) {
    return guard([&] {
        REQUIRE(v && ctx && proof && seed);
        ctx->engine->gens_ensure(gens_capacity);
        uint64_t N = 1; while (N < v->v->get_num_vars()) N <<= 1;
        if (gens_capacity < N) throw R1CSException(R1CSError::InvalidGeneratorsLength, "generator capacity below padded circuit size");
        FlatCircuit f = v->v->flatten();
        DeviceCircuit *dc = ctx->engine->upload(f);
        R1CSError e;
        try { e = ctx->engine->verify(dc, *v->v->transcript(), v->v->commitments().data(), proof, proof_len, seed, flags); } catch (...) { ctx->engine->free_circuit(dc); throw; }
        ctx->engine->free_circuit(dc);
        if (e != R1CSError::None) throw R1CSException(e, e == R1CSError::VerificationError ? "proof rejected" : e == R1CSError::FormatError ? "malformed proof" : "generator capacity below padded circuit size");
    });
}

// ---------------------------------------------------------------------------------------- gadgets
bpg_status bpg_bounds_check_new(const uint8_t *min_be, uint64_t min_len, const uint8_t *max_be, uint64_t max_len, bpg_gadget **out) {
    return guard([&] {
        REQUIRE(out && min_be && max_be);
        if (min_len > 32 || max_len > 32) throw std::invalid_argument("the given vector is longer than 32 bytes");
        *out = new bpg_gadget{std::unique_ptr<Gadget>(new BoundsCheck(Bytes(min_be, min_be + min_len), Bytes(max_be, max_be + max_len)))};
    });
}
bpg_status bpg_mimc_hash256_new(const bpg_lc *image, bpg_gadget **out) {
    return guard([&] { REQUIRE(out); *out = new bpg_gadget{std::unique_ptr<Gadget>(new MimcHash256(lc_from(image)))}; });
}
bpg_status bpg_merkle_tree256_new(const bpg_lc *root, const bpg_lc *inst, uint64_t n_inst, const bpg_lc *wit, uint64_t n_wit, const char *pattern, bpg_gadget **out) {
    return guard([&] {
        REQUIRE(out && pattern && (n_inst == 0 || inst) && (n_wit == 0 || wit));
        std::vector<LinearCombination> iv, wv;
        for (uint64_t i = 0; i < n_inst; i++) iv.push_back(lc_from(&inst[i]));
        for (uint64_t i = 0; i < n_wit; i++) wv.push_back(lc_from(&wit[i]));
        *out = new bpg_gadget{std::unique_ptr<Gadget>(new MerkleTree256(lc_from(root), iv, wv, Pattern::parse(pattern)))};
    });
}
static std::vector<LinearCombination> lcs_from(const bpg_lc *a, uint64_t n) { std::vector<LinearCombination> o; for (uint64_t i = 0; i < n; i++) o.push_back(lc_from(&a[i])); return o; }
static std::vector<Scalar> scalars_from(const uint8_t *p, uint64_t n) { std::vector<Scalar> o; if (p) for (uint64_t i = 0; i < n; i++) o.push_back(Scalar::from_bits(p + 32 * i)); return o; }
bpg_status bpg_equality_new(const bpg_lc *right, uint64_t n, bpg_gadget **out) {
    return guard([&] { REQUIRE(out && (n == 0 || right)); *out = new bpg_gadget{std::unique_ptr<Gadget>(new Equality(lcs_from(right, n)))}; });
}
bpg_status bpg_inequality_new(const bpg_lc *right, uint64_t n, const uint8_t *ra, bpg_gadget **out) {
    return guard([&] { REQUIRE(out && (n == 0 || right)); *out = new bpg_gadget{std::unique_ptr<Gadget>(new Inequality(lcs_from(right, n), ra != nullptr, scalars_from(ra, n)))}; });
}
bpg_status bpg_less_than_new(const bpg_lc *left, const uint8_t *la, const bpg_lc *right, const uint8_t *ra, bpg_gadget **out) {
    return guard([&] {
        REQUIRE(out);
        *out = new bpg_gadget{std::unique_ptr<Gadget>(new LessThan(lc_from(left), la ? OptScalar(Scalar::from_bits(la)) : OptScalar(), lc_from(right),
                                                                   ra ? OptScalar(Scalar::from_bits(ra)) : OptScalar()))};
    });
}
bpg_status bpg_set_membership_new(const bpg_lc *value, const uint8_t *va, const bpg_lc *inst, uint64_t n_inst, const uint8_t *ia, bpg_gadget **out) {
    return guard([&] {
        REQUIRE(out && (n_inst == 0 || inst));
        *out = new bpg_gadget{std::unique_ptr<Gadget>(new SetMembership(lc_from(value), va ? OptScalar(Scalar::from_bits(va)) : OptScalar(), lcs_from(inst, n_inst),
                                                                        ia != nullptr || n_inst == 0, scalars_from(ia, n_inst)))};
    });
}
void bpg_gadget_free(bpg_gadget *g) { delete g; }

bpg_status bpg_gadget_setup(bpg_gadget *g, bpg_prover *p, const uint8_t *wit, uint64_t n_wit, const uint8_t *blind, uint64_t n_blind,
                            uint8_t *coms_out, uint8_t *derived_scalars_out, uint32_t *derived_vars_out, uint64_t *n_derived) {
    return guard([&] {
        REQUIRE(g && p && n_derived && (n_wit == 0 || wit) && (n_blind == 0 || blind));
        std::vector<Scalar> w(n_wit), b(n_blind);
        for (uint64_t i = 0; i < n_wit; i++) w[i] = Scalar::from_bits(wit + 32 * i);
        for (uint64_t i = 0; i < n_blind; i++) b[i] = Scalar::from_bytes_mod_order(blind + 32 * i);
        if (g->g->preprocess(w).size() > *n_derived) throw std::invalid_argument("setup: output capacity too small");
        auto r = g->g->setup(*p->p, w, b);
        const size_t k = r.second.size();
        if (k) { REQUIRE(coms_out && derived_scalars_out && derived_vars_out); std::memcpy(coms_out, r.first.data(), k * 32); }
        for (size_t i = 0; i < k; i++) { r.second[i].first.v.to_bytes(derived_scalars_out + 32 * i); derived_vars_out[i] = r.second[i].second.packed(); }
        *n_derived = k;
    });
}
bpg_status bpg_gadget_preprocess(bpg_gadget *g, const uint8_t *wit, uint64_t n_wit, uint8_t *derived_scalars_out, uint64_t *n_derived) {
    return guard([&] {
        REQUIRE(g && n_derived && (n_wit == 0 || wit));
        std::vector<Scalar> w(n_wit);
        for (uint64_t i = 0; i < n_wit; i++) w[i] = Scalar::from_bits(wit + 32 * i);
        const std::vector<Scalar> d = g->g->preprocess(w);
        if (d.size() > *n_derived) throw std::invalid_argument("preprocess: output capacity too small");
        if (!d.empty()) REQUIRE(derived_scalars_out);
        for (size_t i = 0; i < d.size(); i++) d[i].to_bytes(derived_scalars_out + 32 * i);
        *n_derived = d.size();
    });
}
static std::vector<Variable> unpack_vars(const uint32_t *v, uint64_t n) { std::vector<Variable> o; for (uint64_t i = 0; i < n; i++) o.push_back(Variable::unpack(v[i])); return o; }
bpg_status bpg_gadget_prove(bpg_gadget *g, bpg_prover *p, const uint32_t *vars, uint64_t n_vars, const uint8_t *dsc, const uint32_t *dvars, uint64_t n_derived) {
    return guard([&] {
        REQUIRE(g && p && (n_vars == 0 || vars) && (n_derived == 0 || (dsc && dvars)));
        Derived d;
        for (uint64_t i = 0; i < n_derived; i++) d.emplace_back(OptScalar(Scalar::from_bits(dsc + 32 * i)), Variable::unpack(dvars[i]));
        g->g->prove(*p->p, unpack_vars(vars, n_vars), d);
    });
}
bpg_status bpg_gadget_verify(bpg_gadget *g, bpg_verifier *v, const uint32_t *vars, uint64_t n_vars, const uint32_t *dvars, uint64_t n_derived) {
    return guard([&] {
        REQUIRE(g && v && (n_vars == 0 || vars) && (n_derived == 0 || dvars));
        g->g->verify(*v->v, unpack_vars(vars, n_vars), unpack_vars(dvars, n_derived));
    });
}
bpg_status bpg_buffer_new(uint64_t first, int32_t prover_side, bpg_buffer **out) { return guard([&] { REQUIRE(out); *out = new bpg_buffer{OpBuffer(first, prover_side != 0)}; }); }
void bpg_buffer_free(bpg_buffer *b) { delete b; }
bpg_status bpg_buffer_rewind(bpg_buffer *b) { return guard([&] { REQUIRE(b); b->b.rewind(); }); }
uint64_t bpg_buffer_next_multiplier(const bpg_buffer *b) { return b ? b->b.next_multiplier() : 0; }
bpg_status bpg_gadget_prove_buffered(bpg_gadget *g, bpg_buffer *b, const uint32_t *vars, uint64_t n_vars, const uint8_t *dsc, const uint32_t *dvars, uint64_t n_derived) {
    return guard([&] {
        REQUIRE(g && b && (n_vars == 0 || vars) && (n_derived == 0 || (dsc && dvars)));
        Derived d;
        for (uint64_t i = 0; i < n_derived; i++) d.emplace_back(OptScalar(Scalar::from_bits(dsc + 32 * i)), Variable::unpack(dvars[i]));
        g->g->prove(b->b, unpack_vars(vars, n_vars), d);
    });
}
bpg_status bpg_gadget_verify_buffered(bpg_gadget *g, bpg_buffer *b, const uint32_t *vars, uint64_t n_vars, const uint32_t *dvars, uint64_t n_derived) {
    return guard([&] { REQUIRE(g && b && (n_vars == 0 || vars) && (n_derived == 0 || dvars)); g->g->verify(b->b, unpack_vars(vars, n_vars), unpack_vars(dvars, n_derived)); });
}
bpg_status bpg_or_prover(bpg_prover *main, const bpg_buffer *b) { return guard([&] { REQUIRE(main && b); or_conjunction(*main->p, b->b); }); }
bpg_status bpg_or_verifier(bpg_verifier *main, const bpg_buffer *b) { return guard([&] { REQUIRE(main && b); or_conjunction(*main->v, b->b); }); }
bpg_status bpg_or_buffer(bpg_buffer *parent, const bpg_buffer *b) { return guard([&] { REQUIRE(parent && b); or_conjunction(parent->b, b->b); }); }
bpg_status bpg_range_proof_prove(bpg_prover *p, const bpg_lc *x, uint32_t n_bits, const uint8_t a[32]) {
    return guard([&] { REQUIRE(p && a && n_bits <= 255); range_proof(*p->p, lc_from(x), (uint8_t)n_bits, OptScalar(Scalar::from_bits(a))); });
}
bpg_status bpg_range_proof_verify(bpg_verifier *v, const bpg_lc *x, uint32_t n_bits) {
    return guard([&] { REQUIRE(v && n_bits <= 255); range_proof(*v->v, lc_from(x), (uint8_t)n_bits, OptScalar()); });
}
bpg_status bpg_mimc_hash(const uint8_t *pre, uint64_t len, uint8_t out[32]) {
    return guard([&] { REQUIRE(pre && out && len > 0); mimc_hash(Bytes(pre, pre + len)).to_bytes(out); });
}
bpg_status bpg_be_to_scalars(const uint8_t *be, uint64_t len, uint8_t *out, uint64_t *n_out) {
    return guard([&] {
        REQUIRE(n_out && (len == 0 || be));
        std::vector<Scalar> s = be_to_scalars(Bytes(be, be + len));
        if (s.size() > *n_out) throw std::invalid_argument("be_to_scalars: output capacity too small");
        for (size_t i = 0; i < s.size(); i++) s[i].to_bytes(out + 32 * i);
        *n_out = s.size();
    });
}
bpg_status bpg_rng_draws(const uint8_t transcript_state[203], uint64_t m, const uint8_t *v_blinding, const uint8_t rng_seed[32],
                         uint64_t skip, uint64_t count, int32_t bulk, uint8_t *out) {
    return guard([&] {
        REQUIRE(transcript_state && rng_seed && out && (m == 0 || v_blinding));
        Transcript t = Transcript::from_state(transcript_state);
        std::vector<Scalar> vb(m);
        for (uint64_t i = 0; i < m; i++) std::memcpy(vb[i].w, v_blinding + 32 * i, 32);
        TranscriptRng rng = t.build_rng(vb, rng_seed);
        uint8_t tmp[64];
        for (uint64_t i = 0; i < skip; i++) rng.fill_bytes(tmp, 64);
        if (bulk) rng.fill_draws64(out, count);
        else for (uint64_t i = 0; i < count; i++) rng.fill_bytes(out + 64 * i, 64);
    });
}
bpg_status bpg_rng_draws_multi(const uint8_t transcript_state[203], uint64_t m, const uint8_t *v_blinding, uint32_t lanes, const uint8_t *rng_seeds,
                               const uint64_t *skip, uint64_t count, uint8_t *out) {
    return guard([&] {
        REQUIRE(transcript_state && rng_seeds && skip && out && lanes >= 1 && lanes <= 8 && (m == 0 || v_blinding));
        Transcript t = Transcript::from_state(transcript_state);
        std::vector<Scalar> vb(m);
        for (uint64_t i = 0; i < m; i++) std::memcpy(vb[i].w, v_blinding + 32 * i, 32);
        std::vector<TranscriptRng> rngs;
        for (uint32_t v = 0; v < lanes; v++) {
            rngs.push_back(t.build_rng(vb, rng_seeds + 32 * v));
            uint8_t tmp[64];
            for (uint64_t i = 0; i < skip[v]; i++) rngs.back().fill_bytes(tmp, 64);
        }
        TranscriptRng *r[8]; uint8_t *dst[8];
        for (uint32_t v = 0; v < lanes; v++) { r[v] = &rngs[v]; dst[v] = out + (size_t)v * 64 * count; }
        // in two calls, so that the second starts from the state the first left behind
        const uint64_t first = count / 3;
        TranscriptRng::fill_draws64_multi(r, dst, lanes, first);
        for (uint32_t v = 0; v < lanes; v++) dst[v] += 64 * first;
        TranscriptRng::fill_draws64_multi(r, dst, lanes, count - first);
    });
}
bpg_status bpg_keccak_selftest(uint64_t seed, uint32_t rounds, int32_t *impl_out, double *ns_out) {
    return guard([&] {
        uint64_t a[25], b[25], c[25], d[25];
        for (int i = 0; i < 25; i++) a[i] = b[i] = c[i] = d[i] = seed * 0x9e3779b97f4a7c15ULL + (uint64_t)i * 0xd1342543de82ef95ULL + (seed >> (i & 31));
#if defined(__x86_64__)
        const bool vec = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
#else
        const bool vec = false;
#endif
        for (uint32_t r = 0; r < rounds; r++) {
            keccak_f1600_scalar(a); keccak_f1600_host(b);
            if (std::memcmp(a, b, sizeof a) != 0) throw std::runtime_error("keccak: active and scalar implementations disagree");
#if defined(__x86_64__)
            if (vec) {
                keccak_f1600_avx512(c); keccak_f1600_xmm(d);
                if (std::memcmp(a, c, sizeof a) != 0) throw std::runtime_error("keccak: planes-in-ZMM and scalar implementations disagree");
                if (std::memcmp(a, d, sizeof a) != 0) throw std::runtime_error("keccak: lanes-in-XMM and scalar implementations disagree");
            }
#endif
            a[r % 25] ^= r; b[r % 25] ^= r; c[r % 25] ^= r; d[r % 25] ^= r;
        }
        if (impl_out) *impl_out = keccak_impl();
        if (ns_out) {
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 200000; r++) keccak_f1600_host(b);
            *ns_out = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 200000.0 + (b[0] == 1 ? 1e-9 : 0);
        }
    });
}
bpg_status bpg_scalar_op(int32_t op, const uint8_t *a, const uint8_t *b, uint8_t out[32]) {
    return guard([&] {
        REQUIRE(a && out);
        Scalar r;
        if (op == 5) r = Scalar::from_wide(a);
        else {
            Scalar x; std::memcpy(x.w, a, 32);
            Scalar y; if (b) std::memcpy(y.w, b, 32);
            switch (op) { case 0: r = x + y; break; case 1: r = x - y; break; case 2: r = x * y; break; case 3: r = x.invert(); break; case 4: r = x.reduced(); break; default: throw std::invalid_argument("bad op"); }
        }
        r.to_bytes(out);
    });
}

}  // extern "C"
