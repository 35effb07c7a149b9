// Engine implementation: device buffers, generator derivation, the bucket-method MSM pipeline and the prove
// pipeline (dalek bulletproofs r1cs/prover.rs::prove + inner_product_proof.rs::create re-designed for one GPU;
// reference call site src/bin/prover.rs:92-93).  Fiat-Shamir stays on the host (merlin.hpp); every challenge is a
// 32..96-byte device->host copy followed by a few hundred bytes host->device.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "engine.hpp"
#include "host/fe51.hpp"
#include "host/chain.hpp"
#include "hip/kernels.cuh"

namespace bpg {

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw DeviceError(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

// launch on the engine stream, bracketed by HIP events when the profile asks for this kernel
#define BPG_LAUNCH_ID(I, id, kernel, grid, block, ...) do { (I).prof_begin(id); hipLaunchKernelGGL(kernel, grid, block, 0, (I).st, __VA_ARGS__); (I).prof_end(id); } while (0)
#define BPG_LAUNCH(I, kernel, grid, block, ...) BPG_LAUNCH_ID(I, KID_##kernel, kernel, grid, block, __VA_ARGS__)
#define BPG_LAUNCH_LDS(I, id, kernel, grid, block, lds, ...) do { (I).prof_begin(id); hipLaunchKernelGGL(kernel, grid, block, lds, (I).st, __VA_ARGS__); (I).prof_end(id); } while (0)

namespace {

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIPCHK(hipFree(p));
        p = nullptr; cap = 0;
        HIPCHK(hipMalloc(&p, bytes)); cap = bytes;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIPCHK(hipHostFree(p));
        p = nullptr; cap = 0;
        HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); cap = bytes;
    }
    void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Generator tables of one (device, capacity), shared by every context of the process on that device: the affine-Niels table [G | H] and, built on
// first use, the odd multiples for the width-w NAF fold (one set per w).  Immutable once published, so contexts on different streams read them
// freely; two proving streams then gather from ONE 201 MB table that the Infinity Cache holds, instead of two that evict each other.  A context that
// needs a larger capacity moves to another generation; a generation is freed with its last context.
// Bytes of precomputed multiples (fold tables + wide tail tables) held on each device by ALL generations of this process: what the table budget
// of a context (bpg_config.table_budget_gb) is compared with, so that a process serving mixed circuit sizes cannot pin more than it was given.
static std::mutex g_table_bytes_mutex;
static std::map<int, uint64_t> g_table_bytes;
static uint64_t table_bytes_held(int device) { std::lock_guard<std::mutex> lk(g_table_bytes_mutex); return g_table_bytes[device]; }
static void table_bytes_add(int device, int64_t delta) { std::lock_guard<std::mutex> lk(g_table_bytes_mutex); g_table_bytes[device] = (uint64_t)((int64_t)g_table_bytes[device] + delta); }
// Proofs this process has in flight per device.  Some steps come in two variants with the same results: one that finishes soonest on an idle
// device (four lanes per output in the small folds, sweep chunks fitted to whole rounds of resident blocks, 15-bit windows) and one with the
// fewest instructions (one lane per output, 64-entry chunks, 16-bit windows); a proof that shares the device with others takes the second
// (Impl::shared_variants; measured at 2^20, profiles/r03_tail_start.txt: 1.9 ms per proof sustained, and 1.7 ms more for a proof alone).
// BPG_FOLD_ADAPT=0 pins the first set, 2 the second.
static std::atomic<int> g_proving[64];
struct ProvingGuard {
    std::atomic<int> &c;
    explicit ProvingGuard(int device) : c(g_proving[(unsigned)device & 63u]) { c.fetch_add(1, std::memory_order_relaxed); }
    ~ProvingGuard() { c.fetch_sub(1, std::memory_order_relaxed); }
};
static bool device_shared(int device) { return g_proving[(unsigned)device & 63u].load(std::memory_order_relaxed) > 1; }
struct SharedTables {
    int device = 0; uint64_t cap = 0;
    DevBuf gens;
    std::mutex m;                                   // guards odd, wide, refused (construction on first use)
    std::map<uint32_t, DevBuf> wide;                // M0 -> 8-bit window tables of G[0..M0), H[0..M0) for a tail that starts on the original generators (k_tt_round8)
    std::map<uint32_t, DevBuf> odd;                 // (w | parts << 8) -> [parts * 2^(w-2) - 1][2*cap] Niels points: (2m+1) * 2^(j*L) * P, see FoldWnaf
    std::map<uint64_t, uint64_t> refused;           // table key (odd: w | parts << 8; wide: 1 << 32 | M0) -> bytes held on the device when its allocation failed:
                                                    // not tried again until the device holds less (no failing 50 GB hipMalloc per proof)
    ~SharedTables() {
        (void)hipSetDevice(device); gens.release();
        uint64_t held = 0;
        for (auto &kv : odd) { held += kv.second.cap; kv.second.release(); }
        for (auto &kv : wide) { held += kv.second.cap; kv.second.release(); }
        table_bytes_add(device, -(int64_t)held);
    }
};
static std::mutex g_tables_mutex;                   // held across a derivation: contexts created side by side derive once
static std::map<std::pair<int, uint64_t>, std::weak_ptr<SharedTables>> g_tables;

inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// host Scalar (plain, canonical) <-> device Montgomery form
const Scalar &R1_plain() { static Scalar r = [] { Scalar s; s.w[0] = 0xd6ec31748d98951dULL; s.w[1] = 0xc6ef5bf4737dcf70ULL; s.w[2] = 0xfffffffffffffffeULL; s.w[3] = 0x0fffffffffffffffULL; return s; }(); return r; }
const Scalar &Rinv_plain() { static Scalar r = R1_plain().invert(); return r; }
scm to_scm(const Scalar &s) { Scalar m = s * R1_plain(); scm o; std::memcpy(o.v, m.w, 32); return o; }
Scalar from_scm(const scm &m) { Scalar s; std::memcpy(s.w, m.v, 32); return s * Rinv_plain(); }

// non-adjacent form of a canonical scalar; returns index of the top non-zero digit (-1 for zero)
int32_t naf256(const Scalar &s, int8_t d[256]) {
    uint64_t k[5] = {s.w[0], s.w[1], s.w[2], s.w[3], 0};
    std::memset(d, 0, 256);
    int32_t top = -1;
    for (int i = 0; i < 256; i++) {
        if ((k[0] | k[1] | k[2] | k[3] | k[4]) == 0) break;
        if (k[0] & 1) {
            int dig = 2 - (int)(k[0] & 3);           // +1 or -1
            d[i] = (int8_t)dig; top = i;
            if (dig == 1) k[0] -= 1;                  // low bit set, no borrow
            else { for (int j = 0; j < 5; j++) { if (++k[j] != 0) break; } }
        }
        for (int j = 0; j < 4; j++) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    return top;
}

// width-w non-adjacent form of a canonical scalar: odd digits in (-2^(w-1), 2^(w-1)), at most one non-zero in any w consecutive positions;
// returns the index of the top non-zero digit (-1 for zero).  Scalars are < 2^253, so position 255 is never reached.
int32_t wnaf256(const Scalar &s, uint32_t w, int8_t d[256]) {
    uint64_t k[5] = {s.w[0], s.w[1], s.w[2], s.w[3], 0};
    std::memset(d, 0, 256);
    int32_t top = -1;
    const int64_t full = 1ll << w, half = 1ll << (w - 1);
    for (int i = 0; i < 256; i++) {
        if ((k[0] | k[1] | k[2] | k[3] | k[4]) == 0) break;
        if (k[0] & 1) {
            int64_t dig = (int64_t)(k[0] & (uint64_t)(full - 1));
            if (dig >= half) dig -= full;
            d[i] = (int8_t)dig; top = i;
            if (dig > 0) { k[0] -= (uint64_t)dig; }                                                   // low bits cleared, no borrow
            else { uint64_t add = (uint64_t)(-dig); for (int j = 0; j < 5; j++) { uint64_t t = k[j] + add; add = t < add ? 1 : 0; k[j] = t; if (!add) break; } }
        }
        for (int j = 0; j < 4; j++) k[j] = (k[j] >> 1) | (k[j + 1] << 63);
        k[4] >>= 1;
    }
    return top;
}

// bits [from, from + count) of a canonical scalar as a scalar of its own (count <= 128)
Scalar scalar_bits(const Scalar &s, uint32_t from, uint32_t count) {
    Scalar r = Scalar::zero();
    for (uint32_t k = 0; k < count && from + k < 256; k++) {
        const uint32_t b = from + k;
        if ((s.w[b >> 6] >> (b & 63)) & 1ull) r.w[k >> 6] |= 1ull << (k & 63);
    }
    return r;
}

uint32_t ceil_log2(uint64_t x) { uint32_t l = 0; while ((1ULL << l) < x) l++; return l; }

// On-disk cache of the generator tables (reference src/bin/prover.rs:92 re-derives them on every run).  Opt-in:
// BPG_GENS_CACHE_DIR names a directory; the file gens_<capacity>.bpg holds a header and the affine Niels table [G | H] exactly as it lives
// in HBM.  A file is used only if its header, length and checksum agree AND a sample of its points equals freshly derived ones.  The checksum
// catches corruption, not an adversary: whoever can write the file chooses the generators (a table with known discrete-log relations lets
// verify() accept forged proofs), so the cache trusts the FILE SYSTEM: directory and file must belong to the calling user and be writable by
// nobody else (gens_cache_trusted), files are written through an O_EXCL | O_NOFOLLOW temporary and renamed.
struct GensCacheHeader { char magic[8]; uint64_t version, capacity, bytes, checksum; };
static const char kGensMagic[8] = {'B', 'P', 'G', 'G', 'E', 'N', 'S', '1'};
uint64_t gens_checksum(const uint8_t *p, size_t n) {            // four interleaved multiply-rotate lanes over 8-byte words (about 10 GB/s): corruption, not adversaries
    uint64_t h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) for (int k = 0; k < 4; k++) { uint64_t w; std::memcpy(&w, p + i + 8 * k, 8); h[k] = (h[k] ^ w) * 0x100000001b3ull; h[k] = (h[k] << 29) | (h[k] >> 35); }
    for (; i < n; i++) h[0] = (h[0] ^ p[i]) * 0x100000001b3ull;
    return h[0] ^ (h[1] * 3) ^ (h[2] * 5) ^ (h[3] * 7) ^ (uint64_t)n;
}
// the cache directory (and, when it exists, the file) is owned by this user and not writable by group or others
bool gens_cache_trusted(const std::string &dir, const std::string &file) {
    struct stat st;
    if (::stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != ::geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;
    if (::lstat(file.c_str(), &st) != 0) return true;                       // nothing there yet
    return S_ISREG(st.st_mode) && st.st_uid == ::geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}
std::string gens_cache_path(const std::string &dir, uint64_t cap) {
    if (dir.empty()) return std::string();
    return dir + "/gens_" + std::to_string(cap) + ".bpg";
}

}  // namespace

struct DeviceCircuit {
    uint64_t n = 0, m = 0, q = 0, ncols = 0, nnz = 0;
    bool has_witness = false;
    uint64_t const_begin = 0;      // first entry of the constant-terms column (the last one)
    DevBuf aL, aR, aO, col_ptr, ent_row, ent_coef, coef;
    // equal-scalar merging of A_I and A_O (hip/k_merge.cuh), built at the first prove() of this witness: terms of <a_L, G> + <a_R, H> (of <a_O, G>) that carry
    // the same scalar -> one term on the sum of their generators (pts, affine Niels) with that scalar (sc); skipA / skipB mark the terms that were merged away
    // (one bit per multiplier: a_L and a_R for A_I, a_O for A_O)
    struct MergeSet { uint32_t groups = 0, skipped = 0; DevBuf skipA, skipB, sc, pts; };
    bool merge_tried = false;
    MergeSet mI, mO;
};

// kernel ids for the optional HIP-event profile (bpg_profile_*)
#define BPG_KERNELS(X) X(k_gens_derive) X(k_normalize_niels) X(k_compress_niels) X(k_pedersen) X(k_sc_from_bytes) \
    X(k_sc_from_wide) X(k_blind_poison) X(k_exp_table) X(k_reduce_partials) X(k_flatten) X(k_flatten_const) X(k_poly_t) X(k_poly_eval) X(k_ipa_prep) \
    X(k_ipa_fold_scalars) X(k_fold_points) X(k_fold_points_reg) X(k_fold_points_split) X(k_fold_points_wnaf) X(k_fold_points_quad) X(k_fold_points_quadw) X(k_fold_points_regw) X(k_odd_start) X(k_odd_start_ext) X(k_dbl_times) X(k_odd_step) X(k_msm_digits) X(k_msm_scatter1) X(k_msm_sort2) X(k_scan_blocksums) \
    X(k_scan_apply) X(k_bucket_chunks) X(k_bucket_combine) X(k_bucket_combine_heavy) X(k_bucket_reduce) X(k_window_sums) X(k_window_sums_quad) X(k_decompress) X(k_ipa_s) X(k_verify_scalars) X(k_bench_fe_mul) \
    X(k_tt_bases) X(k_tt_multiples) X(k_tt_bases8) X(k_tt_multiples8) X(k_tt_round8) X(k_tt_factors) X(k_tt_advance) X(k_tt_round) X(k_tt_finish) X(k_blind_expand) X(k_tt_commit3) X(k_tt_commit3_finish) X(k_csc_count) X(k_csc_fill) X(k_csc_colptr) X(k_merge_insert) X(k_merge_plan) X(k_merge_groups) X(k_merge_members) X(k_merge_sum)
enum KernelId {
#define X(n) KID_##n,
    BPG_KERNELS(X)
#undef X
    KID_COUNT
};
static const char *const kKernelNames[KID_COUNT] = {
#define X(n) #n,
    BPG_KERNELS(X)
#undef X
};

struct Engine::Impl {
    hipStream_t st = nullptr;
    // profiling: mode 0 off, 1 = the generator-fold kernels and the bucket sweep only (a few launches per proof: cheap enough for
    // timed regions), 2 = every kernel
    int prof_mode = 0;
    struct ProfRec { int id; hipEvent_t a, b; };
    std::vector<ProfRec> prof_open;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[KID_COUNT] = {0};
    uint64_t prof_count[KID_COUNT] = {0};
    double prof_alg_bytes[KID_COUNT] = {0}, prof_act_bytes[KID_COUNT] = {0}, prof_fm[KID_COUNT] = {0};
    bool prof_on(int id) const { return prof_mode == 2 || (prof_mode == 1 && (id == KID_k_fold_points || id == KID_k_fold_points_reg || id == KID_k_fold_points_split || id == KID_k_fold_points_wnaf || id == KID_k_fold_points_quad || id == KID_k_fold_points_quadw || id == KID_k_fold_points_regw || id == KID_k_bucket_chunks)); }
    hipEvent_t prof_event() { if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; } hipEvent_t e; HIPCHK(hipEventCreate(&e)); return e; }
    void prof_begin(int id) { if (!prof_on(id)) return; ProfRec r{id, prof_event(), prof_event()}; HIPCHK(hipEventRecord(r.a, st)); prof_open.push_back(r); }
    void prof_end(int id) { if (!prof_on(id)) return; HIPCHK(hipEventRecord(prof_open.back().b, st)); }
    void prof_note(int id, double alg_bytes, double act_bytes, double fm) { if (!prof_on(id)) return; prof_alg_bytes[id] += alg_bytes; prof_act_bytes[id] += act_bytes; prof_fm[id] += fm; }
    void prof_collect() {
        if (prof_open.empty()) return;
        HIPCHK(hipStreamSynchronize(st));
        for (ProfRec &r : prof_open) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, r.a, r.b)); prof_ms[r.id] += ms; prof_count[r.id]++; prof_pool.push_back(r.a); prof_pool.push_back(r.b); }
        prof_open.clear();
    }
    void prof_reset() { prof_collect(); for (int i = 0; i < KID_COUNT; i++) { prof_ms[i] = 0; prof_count[i] = 0; prof_alg_bytes[i] = prof_act_bytes[i] = prof_fm[i] = 0; } }
    std::shared_ptr<SharedTables> shared;       // the generation of generator tables this context works on
    DevBuf gens;                                // view of shared->gens (not owned)
    DevBuf bases, scratch_ext, comp, small_in, small_sc;
    // One arena for the large per-stream buffers whose lifetimes never overlap in stream order (round 5: 20 proving streams held 2.9 GB each):
    //   an MSM:        [digits | entries1] (dead once k_msm_sort2 has run) overlaid by the sweep's partial sums (slots); entries behind them
    //   poly phase:    flattened weights, powers of z and y (between the S sums and the first round of the inner-product argument)
    //   IPA tail:      the window tables of the folded generators (0.54 GB; no MSM runs once the tail has started)
    // Everything is queued on ONE stream, so a later phase's kernels start after the earlier phase's have finished.  Growing the arena frees it first
    // (hipFree synchronises the device), exactly as growing any DevBuf does.
    DevBuf arena;
    uint8_t *arena_at(size_t off) const { return arena.as<uint8_t>() + off; }
    // Waiting for the stream inside a proof (a round of the inner-product argument ends with one: the host needs L, R for the transcript).  With blocking waits
    // (bpg_config.blocking_sync = 1: a host with many proving threads beside its chain threads) a wake-up costs tens of microseconds, twenty-odd times per
    // proof; a proof that has the device to ITSELF has a core to spare, so it polls first (a proof alone: 1.5 ms; the mix never polls) and blocks only
    // when the work is long.
    bool blocking_waits = false;
    void wait_stream() {
        if (blocking_waits && !shared_now) {
            const double until = now_ms() + 3.0;
            do {
                const hipError_t e = hipStreamQuery(st);
                if (e == hipSuccess) return;
                if (e != hipErrorNotReady) HIPCHK(e);
                for (int k = 0; k < 64; k++) __builtin_ia32_pause();
            } while (now_ms() < until);
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
    // MSM workspace
    DevBuf counts, starts, cursor, blocksum, buckets, partial, msm_result, open_keys, medium, wsums, wq_stage, wq_tickets, tile_hist, heavy, plain, starts1;   // tile_hist, plain: workspace of upload(); digits, the entry lists and the sweep's partial sums live in the arena
    uint32_t sweep_blocks_resident = 1024;   // blocks of k_bucket_chunks the device holds at once: 4 per CU of the device the context is created on (BPG_SWEEP_RESIDENT overrides)
    uint32_t msm_cmax = 15;         // widest window of a proof ALONE on the device (BPG_MSM_CMAX sets both caps)
    uint32_t msm_cmax_shared = 16;  // ... and while other proofs share the device: 16 windows instead of 17 per term, twice the buckets (digits are 16-bit)
    uint32_t rseg = 8;              // buckets per thread of k_bucket_reduce, a power of two (BPG_RSEG)
    uint32_t lgch = 0;              // BPG_LGCH: pins the sweep's chunk length to 2^lgch entries (0 = fitted, see msm())
    bool gens_share = true;         // BPG_GENS_SHARE=0: this context derives (or loads) and keeps generator tables of its own
    uint32_t profile = 1;           // 1 one-shot, 2 serving (what bpg_config / BPG_PROFILE settled on)
    bool shared_now = false;        // sampled ONCE per prove()/verify(): does this call take the shared-device variants (shared_variants())
    uint32_t msm_cmin = 2;          // BPG_MSM_CMIN: narrowest window (tests: wide windows on small sums)
    uint32_t merge_equal = 1;       // BPG_MERGE: 0 A_I and A_O term by term; 1 equal scalars grouped once per uploaded witness (at its first proof); 2 grouped afresh in EVERY
                                    // proof (what a host that proves each witness once pays: the measurement behind bench.py's `merge_per_proof`); same bytes
    uint32_t merged_last = 0, merged_skipped_last = 0;   // witness of the last prove(): groups of equal scalars in A_I and A_O, and the terms they replace (0: none, or the table-driven path)
    uint32_t msm_skipped_terms = 0; // terms of the next msm() call whose skip bit is set (they make no entries): the window width and the entry bound are sized for the rest
    uint32_t msm_alg_discount = 0;  // terms of the next msm() call that are not terms of the sum it computes (merged-point terms stand in for terms that were skipped): roofline bookkeeping only
    void merge_witness(DeviceCircuit *c, const ge_niels *Gtab, const ge_niels *Htab);
    void merge_build(DeviceCircuit::MergeSet &M, const scm *A, const ge_niels *PA, uint32_t nA, const scm *B, const ge_niels *PB, uint32_t nB);
    double merge_ms_last = 0;       // host wall time of the last merge_witness() that did something (BPG_MERGE=1: to the end of its kernels; 2: to the end of the group count's read-back)
    // prove buffers
    DevBuf sLR, yinvpow, lv, rv, red_partial, red_out, raw_rng, extras;      // (y^i, z^j and the flattened weights: in the arena)
    DevBuf stale_flag;              // one word, zero unless k_sc_from_wide met a poisoned (never uploaded) draw: checked before a proof leaves prove()
    DevBuf ipa_s, ipa_tabA, ipa_tabB, naf, qsteps, vfy_in, vfy_pts, vfy_ok, vfy_sc, vfy_ch;
    // table-driven IPA tail (kernels.cuh k_tt_*): frozen-generator window tables, per-point factors, coefficient tables
    DevBuf tt_bases, tt_table, tt_f, tt_c, tt_partial, grp_c, ped_table, s_parts;
    // tt_table holds the tables of the ORIGINAL generators G[0..M0), H[0..M0) when tt_orig_M0 != 0: they survive across proofs (a circuit
    // with N <= 2^tt_orig_lg freezes its generators at round 0) and also serve A_I, A_O, S (k_tt_commit3)
    uint32_t tt_orig_M0 = 0; const void *tt_orig_gens = nullptr;
    ge_pniels *tt_table_p = nullptr;            // the window tables in use: tt_table (original generators) or the arena (folded ones), set by tt_build
    void tt_build(const ge_niels *G, const ge_niels *H, const ge_niels *B, uint32_t M0, bool original) {
        const uint32_t npts = 2 * M0 + 1;
        if (original && tt_orig_M0 == M0 && tt_orig_gens == gens.p) { tt_table_p = tt_table.as<ge_pniels>(); return; }
        tt_orig_M0 = 0;
        const size_t bb = al256((size_t)npts * TT_WINDOWS * sizeof(ge_ext)), tb = (size_t)npts * TT_WINDOWS * TT_MULTS * sizeof(ge_pniels);
        ge_ext *basesp;
        if (original) {     // tables of the ORIGINAL generators outlive the proof (and serve A_I, A_O, S of the next one): buffers of their own
            tt_bases.ensure(bb); tt_table.ensure(tb);
            basesp = tt_bases.as<ge_ext>(); tt_table_p = tt_table.as<ge_pniels>();
        } else {            // tables of FOLDED generators live for the tail of one proof: in the arena, where the MSM workspace of the rounds before was
            arena.ensure(bb + tb);
            basesp = reinterpret_cast<ge_ext *>(arena_at(0)); tt_table_p = reinterpret_cast<ge_pniels *>(arena_at(bb));
        }
        tt_partial.ensure((size_t)3 * cdiv((uint64_t)M0 * 16, 256) * sizeof(ge_ext));
        BPG_LAUNCH((*this), k_tt_bases, dim3(cdiv(npts, 64)), dim3(256), G, H, B, basesp, M0);
        BPG_LAUNCH((*this), k_tt_multiples, dim3(cdiv((uint64_t)npts * TT_WINDOWS, 256)), dim3(256), basesp, tt_table_p, npts * TT_WINDOWS);
        if (original) { tt_orig_M0 = M0; tt_orig_gens = gens.p; }
    }
    // 8-bit window tables of the original generators (kernels.cuh k_tt_round8): shared per device like the fold tables, built on first use
    // Budgets.  table_budget bounds the CUMULATIVE bytes of precomputed multiples (fold tables + wide tail tables, every capacity) this process
    // holds on the device (bpg_config.table_budget_gb / BPG_TABLE_GB; profile default: one-shot 4 GB, serving 96 GB); the two per-kind caps are
    // diagnostics (BPG_TT_WIDE_GB, BPG_FOLD_TABLE_GB).
    uint64_t table_budget = 4ull << 30;
    uint64_t tt_wide_budget = 0;                // widest single 8-bit tail table (17.2 GB at M0 = 2^14); 0 = never (the one-shot profile)
    std::string gens_cache_dir;                 // bpg_config.gens_cache_dir / BPG_GENS_CACHE_DIR
    // Reserve `bytes` of the device's table budget (shared->m held).  Check and reservation happen under ONE lock of the per-device byte count, so
    // two generations of different capacity (each under its own shared->m) cannot both pass the check and overshoot the budget together; the
    // caller gives the bytes back (table_bytes_add(-bytes)) when its allocation fails.
    bool table_reserve(uint64_t key, uint64_t bytes) const {
        std::lock_guard<std::mutex> lk(g_table_bytes_mutex);
        uint64_t &held = g_table_bytes[shared->device];
        auto it = shared->refused.find(key);
        if (it != shared->refused.end() && held >= it->second) return false;      // failed with this much (or less) held: do not try again
        if (held + bytes > table_budget) return false;
        held += bytes;
        return true;
    }
    const ge_pniels *wide_ensure(uint32_t M0) {
        const uint64_t bytes = (uint64_t)2 * M0 * TT8_WINDOWS * TT8_MULTS * sizeof(ge_pniels);
        if (M0 < 64 || bytes > tt_wide_budget) return nullptr;
        std::lock_guard<std::mutex> lk(shared->m);
        auto it = shared->wide.find(M0);
        if (it != shared->wide.end()) return it->second.as<ge_pniels>();
        const uint64_t key = (1ull << 32) | M0;
        if (!table_reserve(key, bytes)) return nullptr;                          // the 4-bit tables of the context do
        DevBuf table, bases8;
        try { table.ensure(bytes); bases8.ensure((size_t)2 * M0 * TT8_WINDOWS * sizeof(ge_ext)); }
        catch (const std::exception &) {
            (void)hipGetLastError(); table.release(); bases8.release();
            table_bytes_add(shared->device, -(int64_t)bytes); shared->refused[key] = table_bytes_held(shared->device); return nullptr;
        }
        try {
            BPG_LAUNCH((*this), k_tt_bases8, dim3(cdiv(2 * M0, 64)), dim3(256), gens.as<ge_niels>(), gens.as<ge_niels>() + gens_cap, bases8.as<ge_ext>(), M0);
            BPG_LAUNCH((*this), k_tt_multiples8, dim3(cdiv((uint64_t)2 * M0 * TT8_WINDOWS, 256)), dim3(256), bases8.as<ge_ext>(), table.as<ge_pniels>(), 2 * M0 * TT8_WINDOWS);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        } catch (...) {   // a table that was not built is not published: give its memory and its share of the budget back (as odd_ensure does)
            (void)hipStreamSynchronize(st); bases8.release(); table.release(); table_bytes_add(shared->device, -(int64_t)bytes);
            throw;
        }
        bases8.release();
        shared->wide[M0] = table;                                               // (its bytes were reserved above)
        return table.as<ge_pniels>();
    }
    PinBuf h_naf, h_qsteps;
    // odd multiples (2m+1) * 2^(j*L) * P of the original generators for the width-w NAF fold of the first group (k_fold_points_wnaf, scalars
    // cut into `fold_parts` pieces of L bits); built on first use for the device's generator tables and shared with them
    DevBuf gens_odd;                 // view of shared->odd[fold_wnaf | fold_parts << 8] (not owned)
    uint32_t fold_wnaf = 5;          // width of the NAF the first fold recodes its scalars in (BPG_FOLD_WNAF; one-shot profile 5, serving 8; 0 = register kernels)
    uint32_t fold_parts = 2;         // the scalars of the first fold are cut into this many parts on tables of 2^(j*L) * P (BPG_FOLD_PARTS: 1, 2, 4 or 8; one-shot 2, serving 4)
    uint64_t fold_table_budget = 64ull << 30;       // per-kind cap of the fold tables (BPG_FOLD_TABLE_GB); the cumulative bound is table_budget
    uint32_t eff_wnaf = 0, eff_parts = 0;            // what odd_ensure settled on for the current capacity
    uint32_t fold_part_bits() const { return (254 + eff_parts - 1) / eff_parts; }
    // false: no table of any width fits the budget (or the device): the caller folds with the register kernels
    bool odd_ensure() {
        eff_wnaf = fold_wnaf; eff_parts = fold_parts;
        auto table_bytes = [&](uint32_t w, uint32_t parts) { return ((uint64_t)parts * (1u << (w - 2)) - 1) * 2 * gens_cap * sizeof(ge_niels); };
        auto smaller = [&]() { if (eff_parts > 1) { eff_parts /= 2; return true; } if (eff_wnaf > 3) { eff_wnaf--; return true; } return false; };
        std::lock_guard<std::mutex> lk(shared->m);
        DevBuf odd;
        for (;;) {   // the widest profile that is already built, or fits the budgets and the device (other tenants): the next smaller one otherwise
            const uint32_t key = eff_wnaf | (eff_parts << 8);
            auto it = shared->odd.find(key);
            if (it != shared->odd.end()) { gens_odd = it->second; return true; }
            const uint64_t bytes = table_bytes(eff_wnaf, eff_parts);
            if (bytes <= fold_table_budget && table_reserve(key, bytes)) {
                try { odd.ensure(bytes); break; }
                catch (const std::exception &) { (void)hipGetLastError(); odd.release(); table_bytes_add(shared->device, -(int64_t)bytes); shared->refused[key] = table_bytes_held(shared->device); }
            }
            if (!smaller()) { gens_odd = DevBuf(); return false; }
        }
        const uint32_t key_built = eff_wnaf | (eff_parts << 8);
        const uint32_t NM = 1u << (eff_wnaf - 2), cnt = (uint32_t)(2 * gens_cap), L = fold_part_bits();
        DevBuf dbl, base;
        try {
        scratch_ext.ensure((size_t)cnt * sizeof(ge_ext));
        dbl.ensure((size_t)cnt * sizeof(ge_ext));
        if (eff_parts > 1) base.ensure((size_t)cnt * sizeof(ge_ext));
        const dim3 grid(cdiv(cnt, 256)), ngrid(cdiv(cdiv(cnt, NORM_K), 256));
        for (uint32_t part = 0; part < eff_parts; part++) {
            ge_niels *tab0 = odd.as<ge_niels>() + ((ptrdiff_t)part * NM - 1) * (ptrdiff_t)cnt;          // table (part, m) = tab0 + m * cnt; (0, 0) is the generator table
            if (part == 0) BPG_LAUNCH((*this), k_odd_start, grid, dim3(256), gens.as<ge_niels>(), scratch_ext.as<ge_ext>(), dbl.as<ge_ext>(), cnt);
            else {
                BPG_LAUNCH((*this), k_dbl_times, grid, dim3(256), gens.as<ge_niels>(), base.as<ge_ext>(), cnt, L, part == 1 ? 1u : 0u);      // 2^(part*L) * P
                BPG_LAUNCH((*this), k_normalize_niels, ngrid, dim3(256), base.as<ge_ext>(), tab0, cnt);
                BPG_LAUNCH((*this), k_odd_start_ext, grid, dim3(256), base.as<ge_ext>(), scratch_ext.as<ge_ext>(), dbl.as<ge_ext>(), cnt);
            }
            for (uint32_t m = 1; m < NM; m++) {
                if (m > 1) BPG_LAUNCH((*this), k_odd_step, grid, dim3(256), scratch_ext.as<ge_ext>(), dbl.as<ge_ext>(), cnt);
                BPG_LAUNCH((*this), k_normalize_niels, ngrid, dim3(256), scratch_ext.as<ge_ext>(), tab0 + (size_t)m * cnt, cnt);
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        } catch (...) {   // a table that was not built is not published: give its memory and its share of the budget back
            (void)hipStreamSynchronize(st); dbl.release(); base.release();
            const uint64_t held = odd.cap; odd.release(); table_bytes_add(shared->device, -(int64_t)held);
            throw;
        }
        dbl.release(); base.release();
        shared->odd[key_built] = odd;                                           // (its bytes were reserved above)
        gens_odd = odd;
        return true;
    }
    uint32_t fold_split_max = 65536; // folds with at most this many outputs use the 4-wave latency variant (BPG_FOLD_SPLIT overrides; 0 = never)
    uint32_t fold_adapt = 1;        // a proof that shares the device with others takes the register fold kernels throughout and sweeps in chunks of 64 (BPG_FOLD_ADAPT: 0 never, 1 when shared, 2 always)
    bool shared_variants() const { return fold_adapt == 2 || (fold_adapt == 1 && device_shared(shared->device)); }
    bool fold_quad = true;          // small folds: four lanes per output (BPG_FOLD_QUAD=0: the four-wave split kernel)
    bool fold_quad_w = true;        // ... with width-4 NAF against multiples the quads make themselves, in the groups after the first (BPG_FOLD_QUAD_W=0: plain NAF, addends in registers)
    bool fold_reg_w = true;         // the same steps with one lane per output where the register kernels would run (BPG_FOLD_REG_W=0: plain NAF, addends in registers)
    bool window_quad = true;        // window sums of a proof alone: four lanes per point, several blocks per window (BPG_WINDOW_QUAD=0: k_window_sums always)
    uint32_t window_quad_blocks = 288;   // ... at most this many blocks of four waves per launch (about one wave per SIMD on 256 CUs; BPG_WINDOW_QUAD_BLOCKS)
    uint32_t fold_group = 3;        // rounds per generator fold (BPG_FOLD_GROUP overrides, 1..5)
    uint32_t tt_lg = 12;            // freeze the FOLDED generators once a round is down to 2^tt_lg per side: their window tables are built per proof (BPG_TT_LG; 0 = never)
    uint32_t tt_orig_lg = 14;       // a circuit of N <= 2^tt_orig_lg freezes the ORIGINAL generators at round 0: those tables are built once and also serve A_I, A_O, S
                                    // (BPG_TT_ORIG_LG; BPG_TT_LG sets both).  Measured at 2^20 (profiles/r03_tail_start.txt): 12 beats 14 alone and in flight
    PinBuf h_raw, h_small;
    // Speculative blinding streams (Engine::blinding_begin): the leading draws of Prover::prove's TranscriptRng, produced on the context's
    // chain worker (ONE host thread, FIFO) before the circuit is known - or, for a sequence of proofs, while the previous proof's kernels run.
    // snaps[k] = generator state before draw k * SNAP (after the three leading blinding scalars).  Up to two streams are alive per context
    // (the one being consumed and the next), each with its own pinned slab.
    using BlindStream = bpg::BlindStream;                    // host/chain.hpp: the stream, its worker loops and the publication protocol (no HIP in there)
    using ChainWorker = bpg::ChainWorker;
    std::shared_ptr<ChainWorker> chain;                     // the context's own worker, or the ChainPool it is attached to
    uint32_t pool_streams = 0;                              // attached to a pool: blinding streams this context may have alive
    uint32_t chain_workers = 1;                             // threads of the chain worker (bpg_ctx_set_chain_workers / BPG_CHAIN_WORKERS); alive streams <= workers * lanes + 1
    uint32_t chain_lanes = 1;                               // streams each thread draws in lockstep (bpg_ctx_set_chain_lanes / BPG_CHAIN_LANES, 1..8)
    std::deque<std::shared_ptr<BlindStream>> blinds;        // alive streams, oldest first
    std::vector<std::shared_ptr<BlindStream>> slab_owner;   // last stream that wrote each pinned slab (workers + 1 slabs)
    std::vector<PinBuf> h_blind;
    struct SlabDev { DevBuf d; hipStream_t copy_st = nullptr; std::vector<hipEvent_t> ev; };
    std::vector<std::unique_ptr<SlabDev>> slab_dev;         // device side of each slab: the uploaded draws, the copy stream, one event per block
    int last_chain_cpu = -1;
    int test_fail_upload = 0;       // test hooks: 1 the next stream's upload reports an error, 2 its copies are silently dropped
    static void blind_stop(const std::shared_ptr<BlindStream> &b) {       // returns once the worker no longer touches b's slab
        b->stop.store(true, std::memory_order_relaxed);
    }
    void blind_retire(const std::shared_ptr<BlindStream> &b) {
        blind_stop(b);
        if (chain) {   // still queued (never started)?  take it out; else wait for the worker to leave it
            std::unique_lock<std::mutex> lk(chain->mu);
            for (auto it = chain->pending.begin(); it != chain->pending.end(); ++it) if (it->get() == b.get()) { chain->pending.erase(it); b->finished.store(true); break; }
        }
        while (!b->finished.load(std::memory_order_acquire)) std::this_thread::yield();
        const int c = b->cpu.load(std::memory_order_relaxed); if (c >= 0) last_chain_cpu = c;
    }
    void blind_cancel() {
        while (!blinds.empty()) { blind_retire(blinds.front()); blinds.pop_front(); }
    }
    void chain_shutdown() {
        blind_cancel();
        if (!chain) return;
        if (!chain->pool) chain->stop();                    // a pool's threads go on serving the other contexts
        chain.reset(); pool_streams = 0;
    }
    uint64_t gens_cap = 0;
    // BPG_GENS_CACHE_DIR (see gens_cache_path): load = read + checksum + upload + compare 2 x 64 sampled points with points derived afresh from
    // the SHAKE256 stream (k_gens_derive on 128 generators); anything that does not agree falls back to the full derivation
    bool gens_load_cached(const std::string &path, uint64_t cap, DevBuf &out);
    void adopt(const std::shared_ptr<SharedTables> &sp) { shared = sp; gens = sp->gens; gens_cap = sp->cap; gens_odd = DevBuf(); }
    void gens_store_cached(const std::string &path, uint64_t cap);

    // An MSM runs on the GPU down to its W window sums per result; those (W x 128 B) travel to a pinned slot and the serial recombination
    // sum_j 2^off(j) S_j (~254 dependent doublings of one point) and the point encoding run on the host (host/fe51.hpp).  msm() queues the
    // kernels and the copy and returns a ticket; msm_points() is called after the stream has been synchronised.
    struct MsmTicket { uint32_t slot, nmsm, W; };
    static constexpr uint32_t WS_SLOTS = 8, WS_SLOT_BYTES = 4 * 128 * 128;     // nmsm <= 4, W <= 127 (c >= 2), 128 B per point
    PinBuf h_wsums; uint32_t ws_next = 0;
    MsmTicket msm(const MsmSegs &S, uint32_t nmsm);
    std::vector<h51::pt> msm_points(const MsmTicket &t) const {
        std::vector<h51::pt> out(t.nmsm);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(h_wsums.as<uint8_t>() + (size_t)t.slot * WS_SLOT_BYTES);
        for (uint32_t m = 0; m < t.nmsm; m++) out[m] = h51::pt_horner(w + (size_t)m * t.W * 32, t.W);
        return out;
    }
    // Host -> device copy of caller-owned pageable memory through two pinned bounce slots.  A direct hipMemcpyAsync from pageable memory
    // lets the runtime pin the caller's pages for the DMA; several contexts uploading the SAME arrays from different threads (a pool
    // proving many witnesses of one circuit) then pin and unpin the same pages concurrently, which faulted the GPU.
    PinBuf stage; hipEvent_t stage_ev[2] = {nullptr, nullptr};
    void h2d(void *dst, const void *src, size_t bytes) {
        const size_t SLOT = 8u << 20;
        stage.ensure(2 * SLOT);
        for (int k = 0; k < 2; k++) if (!stage_ev[k]) HIPCHK(hipEventCreateWithFlags(&stage_ev[k], hipEventDisableTiming));
        const uint8_t *s8 = static_cast<const uint8_t *>(src); uint8_t *d8 = static_cast<uint8_t *>(dst);
        int slot = 0;
        for (size_t off = 0; off < bytes; off += SLOT, slot ^= 1) {
            const size_t len = std::min(SLOT, bytes - off);
            HIPCHK(hipEventSynchronize(stage_ev[slot]));                 // the previous copy out of this slot has finished
            std::memcpy(stage.as<uint8_t>() + (size_t)slot * SLOT, s8 + off, len);
            HIPCHK(hipMemcpyAsync(d8 + off, stage.as<uint8_t>() + (size_t)slot * SLOT, len, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(stage_ev[slot], st));
        }
    }
    void inner_product(Transcript &T, std::vector<uint8_t> &proof, uint64_t n, uint64_t N, const Scalar &yinv, const Scalar &u_ch, const Scalar &w,
                       const ge_niels *Gtab, const ge_niels *Htab, const ge_niels *Bn, ProveTimings *tm, double &t0);
};

// Environment knobs are read ONCE, here, before anything touches the device: a value that does not parse or lies outside its range is an error
// of the call that creates the context (std::invalid_argument -> BPG_ERR_INVALID_ARGUMENT), never a silent default and never a fault later on
// the hot path (round 3 read BPG_RSEG with atoi on every MSM call and divided by it).
namespace {
bool env_present(const char *name) { const char *e = std::getenv(name); return e && *e; }
long env_int_strict(const char *name, long lo, long hi) {
    const char *e = std::getenv(name);
    char *end = nullptr; errno = 0;
    const long v = std::strtol(e, &end, 10);
    if (errno || end == e || *end != '\0' || v < lo || v > hi)
        throw std::invalid_argument(std::string(name) + "=" + e + ": expected an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
    return v;
}
double env_double_strict(const char *name, double lo, double hi) {
    const char *e = std::getenv(name);
    char *end = nullptr; errno = 0;
    const double v = std::strtod(e, &end);
    if (errno || end == e || *end != '\0' || !(v >= lo) || !(v <= hi))
        throw std::invalid_argument(std::string(name) + "=" + e + ": expected a number in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
    return v;
}
template <class T> void env_set(const char *name, long lo, long hi, T &out) { if (env_present(name)) out = (T)env_int_strict(name, lo, hi); }
}  // namespace

Engine::Engine(int device, const EngineConfig &cfg) : device_(device) {
    if (cfg.profile > 2) throw std::invalid_argument("bpg_config.profile: 0 (default), 1 (one-shot) or 2 (serving)");
    if (cfg.table_budget_gb < 0 || cfg.table_budget_gb > 4096) throw std::invalid_argument("bpg_config.table_budget_gb: 0 (default) .. 4096");
    if (cfg.chain_workers > 64) throw std::invalid_argument("bpg_config.chain_workers: 0 (default), 1..64");
    if (cfg.chain_lanes > 8) throw std::invalid_argument("bpg_config.chain_lanes: 0 (default), 1..8");
    if (cfg.blocking_sync < -1 || cfg.blocking_sync > 2) throw std::invalid_argument("bpg_config.blocking_sync: -1 (environment, else spin), 0 (spin), 1 (blocking)");
    // What the host chose (bpg_config): the struct first, then the environment variable, then the profile's default.  Everything is settled
    // in this local Impl before the first HIP call, so a bad knob costs nothing and leaks nothing.
    std::unique_ptr<Impl> K(new Impl());
    int blocking = cfg.blocking_sync == 2 ? 0 : cfg.blocking_sync;          // -1 unset, 0 spin, 1 blocking (2: what one header revision called spin)
    if (blocking < 0 && env_present("BPG_SYNC_BLOCKING")) blocking = env_int_strict("BPG_SYNC_BLOCKING", 0, 1) ? 1 : 0;
    uint32_t profile = cfg.profile;
    if (profile == 0) { if (const char *e = std::getenv("BPG_PROFILE")) profile = (!std::strcmp(e, "serving") || !std::strcmp(e, "2")) ? 2u : 1u; else profile = 1u; }
    // one-shot (the default of a bare bpg_ctx_create): width-5 NAF fold tables on scalars cut in two (15 tables, 3.0 GB at 2^20; round 5: the same bytes as the
    // width-6 tables of whole scalars it replaces, 127 doublings instead of 253 for 42 instead of 36 additions per scalar: the first fold 5.6 -> 5.0 ms), no 8-bit
    // tail tables, 4 GB of tables in all; serving: width-8 NAF on scalars cut in four (51.5 GB at 2^20, 0.12 s), 8-bit tail tables (17.2 GB at 2^14), 96 GB
    if (profile == 2) { K->fold_wnaf = 8; K->fold_parts = 4; K->tt_wide_budget = 24ull << 30; K->table_budget = 96ull << 30; }
    else { K->fold_wnaf = 5; K->fold_parts = 2; K->tt_wide_budget = 0; K->table_budget = 4ull << 30; }
    K->profile = profile;
    {
        double gb = cfg.table_budget_gb;
        if (!(gb > 0) && env_present("BPG_TABLE_GB")) gb = env_double_strict("BPG_TABLE_GB", 0.0, 4096.0);
        if (gb > 0) K->table_budget = (uint64_t)(gb * (double)(1ull << 30));
    }
    K->chain_workers = cfg.chain_workers ? cfg.chain_workers : 1u; if (!cfg.chain_workers) env_set("BPG_CHAIN_WORKERS", 1, 64, K->chain_workers);
    K->chain_lanes = cfg.chain_lanes ? cfg.chain_lanes : 1u; if (!cfg.chain_lanes) env_set("BPG_CHAIN_LANES", 1, 8, K->chain_lanes);
    K->gens_cache_dir = cfg.gens_cache_dir;
    if (K->gens_cache_dir.empty()) { if (const char *e = std::getenv("BPG_GENS_CACHE_DIR")) K->gens_cache_dir = e; }
    // tuning knobs (diagnostics and the schedule tests; every setting gives the same bytes)
    if (env_present("BPG_MSM_CMAX")) K->msm_cmax = K->msm_cmax_shared = (uint32_t)env_int_strict("BPG_MSM_CMAX", 4, 16);
    env_set("BPG_MSM_CMIN", 2, 16, K->msm_cmin);
    env_set("BPG_MERGE", 0, 2, K->merge_equal);
    bool resident_set = false;
    if (env_present("BPG_SWEEP_RESIDENT")) { K->sweep_blocks_resident = (uint32_t)env_int_strict("BPG_SWEEP_RESIDENT", 64, 65536); resident_set = true; }
    if (env_present("BPG_RSEG")) {
        const long v = env_int_strict("BPG_RSEG", 1, 1024);
        if (v & (v - 1)) throw std::invalid_argument("BPG_RSEG: the segment length of the bucket reduction is a power of two in [1, 1024]");
        K->rseg = (uint32_t)v;
    }
    env_set("BPG_LGCH", 2, 10, K->lgch);
    env_set("BPG_FOLD_SPLIT", 0, 1 << 24, K->fold_split_max);
    if (env_present("BPG_FOLD_QUAD")) K->fold_quad = env_int_strict("BPG_FOLD_QUAD", 0, 1) != 0;
    if (env_present("BPG_FOLD_QUAD_W")) K->fold_quad_w = env_int_strict("BPG_FOLD_QUAD_W", 0, 1) != 0;
    if (env_present("BPG_FOLD_REG_W")) K->fold_reg_w = env_int_strict("BPG_FOLD_REG_W", 0, 1) != 0;
    if (env_present("BPG_WINDOW_QUAD")) K->window_quad = env_int_strict("BPG_WINDOW_QUAD", 0, 1) != 0;
    env_set("BPG_WINDOW_QUAD_BLOCKS", 1, 65536, K->window_quad_blocks);
    env_set("BPG_FOLD_ADAPT", 0, 2, K->fold_adapt);
    env_set("BPG_FOLD_GROUP", 1, 5, K->fold_group);
    if (env_present("BPG_TT_WIDE_GB")) K->tt_wide_budget = (uint64_t)(env_double_strict("BPG_TT_WIDE_GB", 0.0, 4096.0) * (double)(1ull << 30));
    if (env_present("BPG_FOLD_TABLE_GB")) K->fold_table_budget = (uint64_t)(env_double_strict("BPG_FOLD_TABLE_GB", 0.0, 4096.0) * (double)(1ull << 30));
    if (env_present("BPG_FOLD_PARTS")) { const long v = env_int_strict("BPG_FOLD_PARTS", 1, 8); if (v & (v - 1)) throw std::invalid_argument("BPG_FOLD_PARTS: 1, 2, 4 or 8"); K->fold_parts = (uint32_t)v; }
    if (env_present("BPG_FOLD_WNAF")) { const long v = env_int_strict("BPG_FOLD_WNAF", 0, 8); if (v == 1 || v == 2) throw std::invalid_argument("BPG_FOLD_WNAF: 0 (register kernels) or 3..8"); K->fold_wnaf = (uint32_t)v; }
    if (env_present("BPG_TT_LG")) K->tt_lg = K->tt_orig_lg = (uint32_t)env_int_strict("BPG_TT_LG", 0, 20);
    env_set("BPG_TT_ORIG_LG", 0, 20, K->tt_orig_lg);
    if (env_present("BPG_GENS_SHARE")) K->gens_share = env_int_strict("BPG_GENS_SHARE", 0, 1) != 0;

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw DeviceError("no HIP device available: the bpg engine has no CPU path");
    if (device < 0 || device >= count) throw DeviceError("invalid device ordinal");
    HIPCHK(hipSetDevice(device));
    if (blocking == 1) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); K->blocking_waits = true; }
    {   // blocks of the sweep the device holds at once: 4 per CU of THIS device (a partitioned MI355X shows fewer CUs); blocks of k_window_sums_quad per launch: one
        // wave per SIMD and an eighth more (288 on 256 CUs)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) {
            if (!resident_set) K->sweep_blocks_resident = (uint32_t)cus * 4u;
            if (!env_present("BPG_WINDOW_QUAD_BLOCKS")) K->window_quad_blocks = (uint32_t)cus + (uint32_t)cus / 8u;
        }
    }
    impl_ = K.release();
    try { init_device(); }
    catch (...) { if (impl_->st) (void)hipStreamDestroy(impl_->st); delete impl_; impl_ = nullptr; throw; }
}

void Engine::init_device() {
    HIPCHK(hipStreamCreate(&impl_->st));
    stream_ = impl_->st;
    // Pedersen bases: B_blinding = from_uniform(SHA3-512(compress(B)))  (PedersenGens::default, reference src/bin/prover.rs:53)
    static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    uint8_t h[64]; sha3_512_host(h, Bc, 32);
    impl_->small_in.ensure(4096); impl_->bases.ensure(3 * sizeof(ge_niels));
    impl_->stale_flag.ensure(64); HIPCHK(hipMemsetAsync(impl_->stale_flag.p, 0, 64, impl_->st));
    HIPCHK(hipMemcpyAsync(impl_->small_in.p, h, 64, hipMemcpyHostToDevice, impl_->st));
    hipLaunchKernelGGL(k_init_bases, dim3(1), dim3(64), 0, impl_->st, impl_->small_in.as<uint32_t>(), impl_->bases.as<ge_niels>());
    HIPCHK(hipGetLastError());
    // window tables of B and B_blinding for k_pedersen (64 windows x 8 multiples each, 128 KB)
    impl_->tt_bases.ensure((size_t)3 * TT_WINDOWS * sizeof(ge_ext));
    impl_->ped_table.ensure((size_t)3 * TT_WINDOWS * TT_MULTS * sizeof(ge_pniels));
    hipLaunchKernelGGL(k_tt_bases, dim3(1), dim3(256), 0, impl_->st, impl_->bases.as<ge_niels>(), impl_->bases.as<ge_niels>() + 1, impl_->bases.as<ge_niels>(),
                       impl_->tt_bases.as<ge_ext>(), 1u);
    hipLaunchKernelGGL(k_tt_multiples, dim3(cdiv(3 * TT_WINDOWS, 256)), dim3(256), 0, impl_->st, impl_->tt_bases.as<ge_ext>(), impl_->ped_table.as<ge_pniels>(),
                       (uint32_t)(3 * TT_WINDOWS));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(impl_->st));
}

Engine::~Engine() {
    if (!impl_) return;
    // order: (1) no host thread of ours issues work for this context any more (the chain threads have left its streams), (2) everything this
    // context queued has finished - the engine stream AND the slabs' copy streams, whose uploads read the pinned slabs - (3) only then memory goes
    impl_->chain_shutdown();
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(impl_->st);
    for (auto &sd : impl_->slab_dev) if (sd->copy_st) (void)hipStreamSynchronize(sd->copy_st);
    DevBuf *bufs[] = {&impl_->bases, &impl_->scratch_ext, &impl_->comp, &impl_->small_in, &impl_->small_sc, &impl_->counts,
                      &impl_->starts, &impl_->cursor, &impl_->blocksum, &impl_->arena, &impl_->buckets, &impl_->partial, &impl_->msm_result,
                      &impl_->sLR, &impl_->yinvpow, &impl_->lv, &impl_->rv, &impl_->red_partial,
                      &impl_->red_out, &impl_->raw_rng, &impl_->extras, &impl_->ipa_s, &impl_->ipa_tabA, &impl_->ipa_tabB, &impl_->naf, &impl_->qsteps, &impl_->wsums, &impl_->wq_stage, &impl_->wq_tickets, &impl_->vfy_in, &impl_->vfy_pts, &impl_->vfy_ok, &impl_->vfy_sc, &impl_->vfy_ch,
                      &impl_->stale_flag, &impl_->tile_hist, &impl_->heavy, &impl_->plain, &impl_->open_keys, &impl_->medium, &impl_->tt_bases, &impl_->tt_table, &impl_->tt_f, &impl_->tt_c, &impl_->tt_partial, &impl_->grp_c, &impl_->ped_table, &impl_->s_parts, &impl_->starts1};
    for (DevBuf *b : bufs) b->release();
    impl_->shared.reset();                                   // the generator tables go with their last context
    impl_->h_raw.release(); impl_->h_small.release(); impl_->h_naf.release(); impl_->h_qsteps.release(); impl_->stage.release(); for (PinBuf &b : impl_->h_blind) b.release();
    for (auto &sd : impl_->slab_dev) {
        if (sd->copy_st) { (void)hipStreamSynchronize(sd->copy_st); (void)hipStreamDestroy(sd->copy_st); }
        for (hipEvent_t e : sd->ev) (void)hipEventDestroy(e);
        sd->d.release();
    }
    for (int k = 0; k < 2; k++) if (impl_->stage_ev[k]) (void)hipEventDestroy(impl_->stage_ev[k]);
    (void)hipStreamDestroy(impl_->st);
    delete impl_;
}

int Engine::device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }
void Engine::profile_set(int mode) { HIPCHK(hipSetDevice(device_)); impl_->prof_reset(); impl_->prof_mode = mode; }
std::string Engine::profile_report() {
    HIPCHK(hipSetDevice(device_));
    impl_->prof_collect();
    // the effective schedule of this context first (what the knobs, the profile and the table budget settled on): bench.py derives its term
    // counts from THIS, not from its own reading of the environment
    char sch[896];
    std::snprintf(sch, sizeof sch, "{\"_schedule\": {\"profile\": %u, \"tt_lg\": %u, \"tt_orig_lg\": %u, \"fold_group\": %u, \"fold_wnaf\": %u, \"fold_parts\": %u, "
                  "\"eff_wnaf\": %u, \"eff_parts\": %u, \"fold_adapt\": %u, \"fold_split_max\": %u, \"fold_quad\": %u, \"msm_cmax\": %u, \"msm_cmax_shared\": %u, \"msm_cmin\": %u, "
                  "\"rseg\": %u, \"lgch\": %u, \"sweep_blocks_resident\": %u, \"shared_variants_last\": %u, \"merge_equal\": %u, \"merged_last\": %u, \"merged_skipped_last\": %u, \"merge_ms_last\": %.3f, \"table_budget\": %llu, \"table_bytes\": %llu}",
                  impl_->profile, impl_->tt_lg, impl_->tt_orig_lg, impl_->fold_group, impl_->fold_wnaf, impl_->fold_parts, impl_->eff_wnaf, impl_->eff_parts,
                  impl_->fold_adapt, impl_->fold_split_max, (unsigned)impl_->fold_quad, impl_->msm_cmax, impl_->msm_cmax_shared, impl_->msm_cmin, impl_->rseg, impl_->lgch,
                  impl_->sweep_blocks_resident, (unsigned)impl_->shared_now, (unsigned)impl_->merge_equal, impl_->merged_last, impl_->merged_skipped_last, impl_->merge_ms_last, (unsigned long long)impl_->table_budget, (unsigned long long)table_bytes_held(device_));
    std::string out = sch;
    bool first = false;
    for (int i = 0; i < KID_COUNT; i++) {
        if (!impl_->prof_count[i]) continue;
        char buf[512];
        std::snprintf(buf, sizeof buf, "%s\"%s\": {\"count\": %llu, \"total_ms\": %.6f, \"alg_bytes\": %.0f, \"device_bytes\": %.0f, \"field_mults\": %.0f}",
                      first ? "" : ", ", kKernelNames[i], (unsigned long long)impl_->prof_count[i], impl_->prof_ms[i], impl_->prof_alg_bytes[i],
                      impl_->prof_act_bytes[i], impl_->prof_fm[i]);
        out += buf; first = false;
    }
    return out + "}";
}
// throughput of dependent-free field multiplications (the binding roofline of this path): returns multiplications / second
double Engine::bench_fe_mul(uint32_t iters) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    const uint32_t blocks = 256 * 16, threads = 256;
    I.small_sc.ensure((size_t)blocks * threads * sizeof(fe));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_bench_fe_mul, dim3(blocks), dim3(threads), 0, I.st, I.small_sc.as<fe>(), 8u);     // warm-up
    HIPCHK(hipEventRecord(a, I.st));
    hipLaunchKernelGGL(k_bench_fe_mul, dim3(blocks), dim3(threads), 0, I.st, I.small_sc.as<fe>(), iters);
    HIPCHK(hipEventRecord(b, I.st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return (double)blocks * threads * iters * 4.0 / (ms * 1e-3);
}

void Engine::test_fe_ops(int op, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    if (!n) return;
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    I.small_sc.ensure(2 * n * 32); I.comp.ensure(n * 32);
    HIPCHK(hipMemcpyAsync(I.small_sc.p, a, n * 32, hipMemcpyHostToDevice, I.st));
    HIPCHK(hipMemcpyAsync(I.small_sc.as<uint8_t>() + n * 32, b, n * 32, hipMemcpyHostToDevice, I.st));
    hipLaunchKernelGGL(k_test_fe, dim3(cdiv(n, 64)), dim3(64), 0, I.st, I.small_sc.as<uint32_t>(), I.small_sc.as<uint32_t>() + 8 * n, I.comp.as<uint8_t>(), (uint32_t)n, (uint32_t)op);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, I.comp.p, n * 32, hipMemcpyDeviceToHost, I.st));
    HIPCHK(hipStreamSynchronize(I.st));
}

void Engine::synchronize() { HIPCHK(hipSetDevice(device_)); HIPCHK(hipStreamSynchronize(impl_->st)); }

// ------------------------------------------------------------------------------------------------ generators
void Engine::gens_ensure(uint64_t capacity) {
    if (capacity == 0 || (capacity & (capacity - 1))) throw std::invalid_argument("generator capacity must be a power of two");
    if (capacity > (1ULL << 24)) throw std::invalid_argument("generator capacity above 2^24 is not supported");
    if (capacity <= gens_cap_) return;
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    // GeneratorsChain: SHAKE256("GeneratorsChain" || 'G'|'H' || u32le(party = 0)), 64 bytes per generator (host squeeze,
    // a serial XOF), then 2*capacity Elligator maps + one batched normalisation on the device.
    const uint64_t cap = capacity;
    std::lock_guard<std::mutex> tables_lock(g_tables_mutex);
    const bool share = I.gens_share;
    if (share) {   // another context of this device already holds tables of this capacity: adopt them
        auto it = g_tables.find({device_, cap});
        if (it != g_tables.end()) {
            if (std::shared_ptr<SharedTables> sp = it->second.lock()) { I.adopt(sp); gens_cap_ = cap; return; }
            g_tables.erase(it);
        }
    }
    auto publish = [&](DevBuf fresh) {
        auto sp = std::make_shared<SharedTables>();
        sp->device = device_; sp->cap = cap; sp->gens = fresh;
        if (share) g_tables[{device_, cap}] = sp;
        I.adopt(sp); gens_cap_ = cap;
    };
    const std::string cache_file = gens_cache_path(I.gens_cache_dir, cap);
    const bool cache_ok = !cache_file.empty() && gens_cache_trusted(I.gens_cache_dir, cache_file);       // else: derive, never read or write the cache
    if (cache_ok) { DevBuf loaded; if (I.gens_load_cached(cache_file, cap, loaded)) { publish(loaded); return; } }
    I.h_raw.ensure(2 * cap * 64);
    {   // the two chains are independent XOF streams: squeeze them on two threads; the streams are prefixes of one another across
        // capacities, so a process-wide cache keeps the longest one squeezed so far (contexts of a batch share it)
        struct Chain { Shake256 sh; std::vector<uint8_t> bytes; bool started = false; };
        static Chain chains[2];
        static std::mutex chain_mutex;
        std::lock_guard<std::mutex> lock(chain_mutex);
        auto extend = [&](int which) {
            Chain &c = chains[which];
            if (!c.started) {
                const uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), 0, 0, 0, 0};
                c.sh.absorb(reinterpret_cast<const uint8_t *>("GeneratorsChain"), 15);
                c.sh.absorb(label, 5);
                c.started = true;
            }
            const size_t have = c.bytes.size(), want = (size_t)cap * 64;
            if (want > have) { c.bytes.resize(want); c.sh.squeeze(c.bytes.data() + have, want - have); }
        };
        (void)keccak_impl();                                  // calibrate once before the threads start
        std::thread th([&] { extend(1); });
        extend(0);
        th.join();
        for (int which = 0; which < 2; which++) std::memcpy(I.h_raw.as<uint8_t>() + (size_t)which * cap * 64, chains[which].bytes.data(), (size_t)cap * 64);
    }
    I.raw_rng.ensure(2 * cap * 64);
    I.scratch_ext.ensure(2 * cap * sizeof(ge_ext));
    DevBuf fresh; fresh.ensure(2 * cap * sizeof(ge_niels));
    HIPCHK(hipMemcpyAsync(I.raw_rng.p, I.h_raw.p, 2 * cap * 64, hipMemcpyHostToDevice, I.st));
    const uint32_t cnt = (uint32_t)(2 * cap);
    BPG_LAUNCH(I, k_gens_derive, dim3(cdiv(cnt, 256)), dim3(256), I.raw_rng.as<uint32_t>(), I.scratch_ext.as<ge_ext>(), cnt);
    BPG_LAUNCH(I, k_normalize_niels, dim3(cdiv(cdiv(cnt, NORM_K), 256)), dim3(256), I.scratch_ext.as<ge_ext>(), fresh.as<ge_niels>(), cnt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(I.st));
    publish(fresh);
    I.raw_rng.release(); I.scratch_ext.release(); I.h_raw.release();      // one-off derivation buffers (0.4 GB of device memory at 2^20); the prove path sizes its own
    if (cache_ok) I.gens_store_cached(cache_file, cap);
}

bool Engine::Impl::gens_load_cached(const std::string &path, uint64_t cap, DevBuf &out) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    FILE *f = fd >= 0 ? ::fdopen(fd, "rb") : nullptr;
    if (!f) { if (fd >= 0) ::close(fd); return false; }
    const size_t bytes = (size_t)2 * cap * sizeof(ge_niels);
    GensCacheHeader hd;
    bool ok = std::fread(&hd, sizeof hd, 1, f) == 1 && std::memcmp(hd.magic, kGensMagic, 8) == 0 && hd.version == 2 && hd.capacity == cap && hd.bytes == bytes;
    PinBuf host;
    if (ok) { host.ensure(bytes); ok = std::fread(host.p, 1, bytes, f) == bytes && std::fgetc(f) == EOF; }
    std::fclose(f);
    if (ok) ok = gens_checksum(host.as<uint8_t>(), bytes) == hd.checksum;
    if (!ok) { host.release(); return false; }
    DevBuf fresh; fresh.ensure(bytes);
    HIPCHK(hipMemcpyAsync(fresh.p, host.p, bytes, hipMemcpyHostToDevice, st));
    // sample: the first 64 generators of G and of H, derived afresh (the head of each SHAKE256 chain: 4 KB) and normalised the usual way
    const uint32_t SAMPLE = (uint32_t)std::min<uint64_t>(64, cap);
    uint8_t raw[2 * 64 * 64];
    for (int which = 0; which < 2; which++) {
        Shake256 sh; const uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), 0, 0, 0, 0};
        sh.absorb(reinterpret_cast<const uint8_t *>("GeneratorsChain"), 15); sh.absorb(label, 5);
        sh.squeeze(raw + (size_t)which * SAMPLE * 64, (size_t)SAMPLE * 64);
    }
    small_in.ensure(sizeof raw); scratch_ext.ensure((size_t)2 * SAMPLE * sizeof(ge_ext)); comp.ensure((size_t)4 * SAMPLE * 32);
    DevBuf smp; smp.ensure((size_t)2 * SAMPLE * sizeof(ge_niels));
    HIPCHK(hipMemcpyAsync(small_in.p, raw, (size_t)2 * SAMPLE * 64, hipMemcpyHostToDevice, st));
    BPG_LAUNCH((*this), k_gens_derive, dim3(cdiv(2 * SAMPLE, 256)), dim3(256), small_in.as<uint32_t>(), scratch_ext.as<ge_ext>(), 2 * SAMPLE);
    BPG_LAUNCH((*this), k_normalize_niels, dim3(1), dim3(256), scratch_ext.as<ge_ext>(), smp.as<ge_niels>(), 2 * SAMPLE);
    // compare encodings (the affine Niels form is unique up to the representative of each coordinate; the encoding is canonical)
    BPG_LAUNCH((*this), k_compress_niels, dim3(cdiv(SAMPLE, 64)), dim3(64), smp.as<ge_niels>(), comp.as<uint8_t>(), SAMPLE);
    BPG_LAUNCH((*this), k_compress_niels, dim3(cdiv(SAMPLE, 64)), dim3(64), smp.as<ge_niels>() + SAMPLE, comp.as<uint8_t>() + 32 * SAMPLE, SAMPLE);
    BPG_LAUNCH((*this), k_compress_niels, dim3(cdiv(SAMPLE, 64)), dim3(64), fresh.as<ge_niels>(), comp.as<uint8_t>() + 64 * SAMPLE, SAMPLE);
    BPG_LAUNCH((*this), k_compress_niels, dim3(cdiv(SAMPLE, 64)), dim3(64), fresh.as<ge_niels>() + cap, comp.as<uint8_t>() + 96 * SAMPLE, SAMPLE);
    HIPCHK(hipGetLastError());
    std::vector<uint8_t> enc((size_t)4 * SAMPLE * 32);
    HIPCHK(hipMemcpyAsync(enc.data(), comp.p, enc.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    host.release(); smp.release();
    if (std::memcmp(enc.data(), enc.data() + (size_t)2 * SAMPLE * 32, (size_t)2 * SAMPLE * 32) != 0) { fresh.release(); return false; }
    out = fresh;
    return true;
}
void Engine::Impl::gens_store_cached(const std::string &path, uint64_t cap) {
    const size_t bytes = (size_t)2 * cap * sizeof(ge_niels);
    std::vector<uint8_t> host(bytes);
    HIPCHK(hipMemcpyAsync(host.data(), gens.p, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    GensCacheHeader hd; std::memcpy(hd.magic, kGensMagic, 8); hd.version = 2; hd.capacity = cap; hd.bytes = bytes; hd.checksum = gens_checksum(host.data(), bytes);
    std::string tmp = path + ".tmp.XXXXXX";
    const int fd = ::mkstemp(&tmp[0]);                          // O_CREAT | O_EXCL, mode 0600, never through a symbolic link
    FILE *f = fd >= 0 ? ::fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) { ::close(fd); std::remove(tmp.c_str()); } return; }      // a cache that cannot be written is not an error
    const bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1 && std::fwrite(host.data(), 1, bytes, f) == bytes;
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

void Engine::gens_export(uint64_t first, uint64_t count, uint8_t *G_out, uint8_t *H_out) {
    if (first + count > gens_cap_) throw std::invalid_argument("gens_export: range beyond capacity");
    if (!count) return;
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    I.comp.ensure(count * 32);
    for (int which = 0; which < 2; which++) {
        const ge_niels *src = I.gens.as<ge_niels>() + (which ? gens_cap_ : 0) + first;
        BPG_LAUNCH(I, k_compress_niels, dim3(cdiv(count, 64)), dim3(64), src, I.comp.as<uint8_t>(), (uint32_t)count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(which ? H_out : G_out, I.comp.p, count * 32, hipMemcpyDeviceToHost, I.st));
        HIPCHK(hipStreamSynchronize(I.st));
    }
}

void Engine::pedersen_bases(uint8_t B[32], uint8_t Bb[32]) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    I.comp.ensure(64);
    BPG_LAUNCH(I, k_compress_niels, dim3(1), dim3(64), I.bases.as<ge_niels>(), I.comp.as<uint8_t>(), 2u);
    HIPCHK(hipGetLastError());
    uint8_t out[64];
    HIPCHK(hipMemcpyAsync(out, I.comp.p, 64, hipMemcpyDeviceToHost, I.st));
    HIPCHK(hipStreamSynchronize(I.st));
    std::memcpy(B, out, 32); std::memcpy(Bb, out + 32, 32);
}

void Engine::pedersen_commit(size_t k, const uint8_t *v, const uint8_t *blind, uint8_t *out) {
    if (!k) return;
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    // v: keep the caller's 255-bit integer (from_bits semantics); blind: reduce on the host so that bit 255 never matters
    std::vector<uint8_t> hv(k * 32), hr(k * 32);
    for (size_t i = 0; i < k; i++) {
        Scalar a = Scalar::from_bits(v + 32 * i); a.to_bytes(&hv[32 * i]);
        Scalar b = Scalar::from_bytes_mod_order(blind + 32 * i); b.to_bytes(&hr[32 * i]);
    }
    I.small_sc.ensure(2 * k * 32); I.comp.ensure(k * 32);
    HIPCHK(hipMemcpyAsync(I.small_sc.p, hv.data(), k * 32, hipMemcpyHostToDevice, I.st));
    HIPCHK(hipMemcpyAsync(I.small_sc.as<uint8_t>() + k * 32, hr.data(), k * 32, hipMemcpyHostToDevice, I.st));
    BPG_LAUNCH(I, k_pedersen, dim3((uint32_t)k), dim3(64), I.small_sc.as<uint32_t>(), I.small_sc.as<uint32_t>() + k * 8,
                       I.ped_table.as<ge_pniels>(), I.comp.as<uint8_t>(), (uint32_t)k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, I.comp.p, k * 32, hipMemcpyDeviceToHost, I.st));
    HIPCHK(hipStreamSynchronize(I.st));
}

// ------------------------------------------------------------------------------------------------ MSM pipeline
Engine::Impl::MsmTicket Engine::Impl::msm(const MsmSegs &S, uint32_t nmsm) {
    const uint32_t total = S.start[S.nseg];
    if (nmsm < 1 || nmsm > 4) throw std::logic_error("msm: 1..4 results per call");
    const uint32_t live = total - std::min(total, msm_skipped_terms);       // terms that can make entries (the others were merged away: MsmSegs::skip)
    msm_skipped_terms = 0;
    uint32_t per = live / nmsm; if (per < 1) per = 1;
    const int cap = (int)(shared_now ? msm_cmax_shared : msm_cmax);
    int cc = (int)ceil_log2(per) - 4; if (cc < (int)msm_cmin) cc = (int)msm_cmin; if (cc > cap) cc = cap;
    uint32_t maxseg = 1; for (uint32_t k = 0; k < S.nseg; k++) maxseg = std::max(maxseg, S.len[k]);
    // two-level sort: entry = sign | fb fine bits | 4 segment bits | index in segment -> 27 - fb index bits; at most 512 coarse bins
    uint32_t fb = 0;
    {
        const uint32_t lgseg = ceil_log2(maxseg);
        if (lgseg > 27) throw std::invalid_argument("msm: segment too long");
        const uint32_t fbmax = std::min<uint32_t>(7, 27 - lgseg);
        if (cc - 1 > (int)fbmax + 9) cc = (int)fbmax + 10;
        fb = std::min<uint32_t>(fbmax, (uint32_t)cc - 1);
    }
    // W near-equal windows over 254 bits (window j starts at bit j * 254 / W: MsmPlan::off); the widest has cmax bits -> 2^(cmax-1) buckets per window
    const uint32_t W = (254 + (uint32_t)cc - 1) / (uint32_t)cc, cmax = (254 + W - 1) / W, nb = 1u << (cmax - 1);
    if (fb > cmax - 1) fb = cmax - 1;
    const uint32_t nkeys = nmsm * W * nb;
    const uint32_t seg = std::min(rseg, nb), nsegpw = nb / seg;        // both powers of two (rseg is validated at context creation)
    // tiling plan: the segments of one MSM are contiguous; tiles never span two MSMs
    MsmPlan P; std::memset(&P, 0, sizeof P);
    P.nmsm = nmsm; P.W = W; P.nb = nb; P.fb = fb; P.CB = nb >> fb;
    {
        const uint32_t lg = 12;                                 // k_msm_scatter1 stages one tile of entries in LDS (MSM_TILE1_MAX)
        P.lgTile = lg;
        uint32_t k = 0;
        for (uint32_t m = 0; m < nmsm; m++) {
            P.term_start[m] = k < S.nseg ? S.start[k] : total;
            while (k < S.nseg && S.msm[k] == m) k++;
        }
        if (k != S.nseg) throw std::logic_error("msm: segments must be grouped by result in ascending order");
        P.term_start[nmsm] = total;
        for (uint32_t m = 0; m < nmsm; m++) {
            const uint32_t nt = cdiv(P.term_start[m + 1] - P.term_start[m], 1u << lg);
            P.tile_start[m + 1] = P.tile_start[m] + nt; if (nt > P.tmax) P.tmax = nt;
        }
        if (P.tmax == 0) P.tmax = 1;
        for (uint32_t j = 0; j < W; j++) { const uint32_t bit = ((j + 1) * 254u) / W - 1; P.bias[bit >> 5] |= 1u << (bit & 31); }
        for (uint32_t j = 0; j <= W; j++) P.off[j] = (uint8_t)((j * 254u) / W);
    }
    const uint32_t ntiles = P.tile_start[nmsm];
    starts.ensure((size_t)(nkeys + 1) * 4);
    buckets.ensure((size_t)nkeys * sizeof(ge_ext));
    partial.ensure((size_t)2 * nmsm * W * nsegpw * sizeof(ge_ext));       // acc and run of every segment
    // balanced sweep: CH sorted entries per thread.  About 32, adjusted so that the launch's blocks fill the device a whole number of times: the
    // sweep keeps sweep_blocks_resident blocks of 256 threads on the CUs at once (4 waves per SIMD at its register count), and 4.25 rounds of
    // blocks cost what 5 do.  BPG_LGCH pins a power of two instead (diagnostics).
    const uint64_t Mub = (uint64_t)live * W;                    // upper bound of the entry count (zero digits are skipped)
    uint32_t CH = 32;
    if (lgch) CH = 1u << lgch;
    else if (shared_now && Mub >= (uint64_t)sweep_blocks_resident * 256 * 64) CH = 64;      // other proofs fill the device and this sweep is long: longer chunks, half the boundary pieces to combine (18.7 against 19.2 ms per proof sustained)
    else {
        const uint64_t slots = (uint64_t)sweep_blocks_resident * 256;
        uint64_t rounds = (Mub + slots * 16) / (slots * 32);    // nearest whole number of rounds at 32 entries per thread
        if (rounds < 1) rounds = 1;
        CH = (uint32_t)std::max<uint64_t>(4, (Mub + slots * rounds - 1) / (slots * rounds));
    }
    const uint32_t nchunks = cdiv(Mub ? Mub : 1, CH);
    // arena layout of this call: [digits | entries1] overlaid by the sweep's partial sums, then entries
    const size_t b_digits = al256((size_t)(total ? total : 1) * W * 2), b_e1 = al256((size_t)(live ? live : 1) * W * 4), b_entries = b_e1;
    const size_t b_slots = al256((size_t)nchunks * 2 * sizeof(ge_ext));
    const size_t b_front = std::max(b_digits + b_e1, b_slots);
    arena.ensure(b_front + b_entries);
    uint16_t *digits_p = reinterpret_cast<uint16_t *>(arena_at(0));
    uint32_t *entries1_p = reinterpret_cast<uint32_t *>(arena_at(b_digits)), *entries_p = reinterpret_cast<uint32_t *>(arena_at(b_front));
    ge_ext *slots_p = reinterpret_cast<ge_ext *>(arena_at(0));
    heavy.ensure(((size_t)nchunks / HEAVY_CHUNKS + 2) * 4); medium.ensure(((size_t)nchunks / 2 + 2) * 4);      // a bucket on the medium list crosses at least two boundaries
    {
        // (kernels.cuh, "two-level sort"): digits once, coarse partition with coalesced runs, fine counting sort inside each coarse bin
        const uint64_t nflat64 = (uint64_t)nmsm * W * P.CB * P.tmax;
        if (nflat64 >= (1ull << 31)) throw std::invalid_argument("msm: too many tiles");
        const uint32_t nflat = (uint32_t)nflat64, nblk1 = cdiv(nflat, SCAN_CHUNK), K = nmsm * W * P.CB;
        counts.ensure((size_t)(nflat + 1) * 4); starts1.ensure((size_t)(nflat + 1) * 4); cursor.ensure((size_t)(nflat + 1) * 4);
        blocksum.ensure((size_t)(nblk1 + 1) * 4);
        HIPCHK(hipMemsetAsync(counts.p, 0, (size_t)nflat * 4, st));        // tiles an MSM does not have (tmax is the longest MSM's count)
        if ((size_t)W * P.CB * 4 > 64 * 1024) throw std::logic_error("msm: coarse histograms exceed the LDS of a block");
        BPG_LAUNCH_LDS((*this), KID_k_msm_digits, k_msm_digits, dim3(ntiles ? ntiles : 1), dim3(512), (size_t)W * P.CB * 4, S, P, total, digits_p, counts.as<uint32_t>(), heavy.as<uint32_t>(), medium.as<uint32_t>());
        BPG_LAUNCH((*this), k_scan_blocksums, dim3(nblk1), dim3(256), counts.as<uint32_t>(), nflat, blocksum.as<uint32_t>());
        BPG_LAUNCH((*this), k_scan_apply, dim3(nblk1), dim3(256), counts.as<uint32_t>(), nflat, blocksum.as<uint32_t>(), starts1.as<uint32_t>(), cursor.as<uint32_t>());
        if (ntiles) BPG_LAUNCH((*this), k_msm_scatter1, dim3(ntiles, W), dim3(256), S, P, digits_p, total, starts1.as<uint32_t>(), entries1_p);
        BPG_LAUNCH((*this), k_msm_sort2, dim3(K), dim3(256), P, starts1.as<uint32_t>(), nflat, entries1_p, starts.as<uint32_t>(), entries_p);
    }
    {
        open_keys.ensure((size_t)nchunks * 4);
        ge_ext *slotA = slots_p, *slotB = slotA + nchunks;
        // the true entry count is starts[nkeys] (device side); threads past it exit immediately
        BPG_LAUNCH((*this), k_bucket_chunks, dim3(cdiv(nchunks, 256)), dim3(256), S, starts.as<uint32_t>(), entries_p,
                   buckets.as<ge_ext>(), slotA, slotB, open_keys.as<uint32_t>(), nkeys, CH);
        // roofline bookkeeping.  Algorithmic bytes: the information content of the MSM this launch sweeps, one scalar + one
        // point = 64 B per TERM, counted once however many windows the term is cut into.  Device bytes: every (term, window) entry is a
        // 4-byte index and a 96-byte affine Niels point.  Work: one mixed addition (7 field multiplications) per entry; Mub counts zero
        // digits too (probability 2^-c each for full-width scalars).
        prof_note(KID_k_bucket_chunks, 64.0 * (double)(total - std::min(total, msm_alg_discount)), 100.0 * (double)Mub, 7.0 * (double)Mub);
        msm_alg_discount = 0;
        // joining the pieces of buckets that cross chunk boundaries: one thread per boundary where chunks are at least as long as the average bucket
        // (the shared-device shape: 64-entry chunks, ~32 entries per bucket), one thread per bucket where buckets are longer (a proof alone)
        if ((uint64_t)CH * nkeys >= Mub)
            BPG_LAUNCH((*this), k_bucket_combine, dim3(cdiv(nchunks, 256)), dim3(256), starts.as<uint32_t>(), buckets.as<ge_ext>(), slotA, slotB, open_keys.as<uint32_t>(), nkeys, CH, heavy.as<uint32_t>(), medium.as<uint32_t>());
        else
            BPG_LAUNCH_ID((*this), KID_k_bucket_combine, k_bucket_combine_per_bucket, dim3(cdiv(nkeys, 256)), dim3(256), starts.as<uint32_t>(), buckets.as<ge_ext>(), slotA, slotB, nkeys, CH, heavy.as<uint32_t>());
        BPG_LAUNCH((*this), k_bucket_combine_heavy, dim3(512), dim3(256), starts.as<uint32_t>(), buckets.as<ge_ext>(), slotA, slotB, CH, heavy.as<uint32_t>(), medium.as<uint32_t>());
    }
    const uint32_t nred = nmsm * W * nsegpw;
    BPG_LAUNCH((*this), k_bucket_reduce, dim3(cdiv(nred, 64)), dim3(64), buckets.as<ge_ext>(), starts.as<uint32_t>(), partial.as<ge_ext>(), nb, seg, nsegpw, nred);
    wsums.ensure((size_t)nmsm * W * sizeof(ge_ext));
    // a window's block: as many threads as it has segments, at most 512 for a proof alone (shortest chain) and 256 while the device is shared (fewest additions)
    if (!shared_now && window_quad) {
        // a proof alone: four lanes per point, a window spread over nblk blocks of 64 slots so that the launch is about one wave per SIMD (k_msm.cuh k_window_sums_quad);
        // per = segments per slot, nblk = blocks per window (at most 64: one slot each in the last block's second stage)
        const uint32_t nwin = nmsm * W;
        uint32_t lgper = 0;
        auto nblk_of = [&](uint32_t lp) { return std::max<uint32_t>(1u, nsegpw >> (6 + lp)); };
        while (nblk_of(lgper) > 1 && ((uint64_t)nwin * nblk_of(lgper) > window_quad_blocks || nblk_of(lgper) > 64)) lgper++;
        const uint32_t nblk = nblk_of(lgper);
        wq_stage.ensure((size_t)nwin * nblk * 2 * sizeof(ge_ext));
        if (!wq_tickets.p) { wq_tickets.ensure(1024 * 4); HIPCHK(hipMemsetAsync(wq_tickets.p, 0, 1024 * 4, st)); }
        if (nwin > 1024) throw std::logic_error("msm: too many windows for the ticket array");
        BPG_LAUNCH((*this), k_window_sums_quad, dim3(nblk, nwin), dim3(256), partial.as<ge_ext>(), wsums.as<ge_ext>(), wq_stage.as<ge_ext>(), wq_tickets.as<uint32_t>(), nsegpw, nred,
                   ceil_log2(seg), lgper);
    } else {
    const uint32_t wthreads = std::max<uint32_t>(64, std::min<uint32_t>(nsegpw, shared_now ? 256u : 512u));
    BPG_LAUNCH((*this), k_window_sums, dim3(nmsm * W), dim3(wthreads), partial.as<ge_ext>(), wsums.as<ge_ext>(), nsegpw, nred, ceil_log2(seg));
    }
    HIPCHK(hipGetLastError());
    if ((size_t)nmsm * W * sizeof(ge_ext) > WS_SLOT_BYTES) throw std::logic_error("msm: window sums exceed the host slot");
    h_wsums.ensure((size_t)WS_SLOTS * WS_SLOT_BYTES);
    MsmTicket t{ws_next, nmsm, W};
    ws_next = (ws_next + 1) % WS_SLOTS;
    HIPCHK(hipMemcpyAsync(h_wsums.as<uint8_t>() + (size_t)t.slot * WS_SLOT_BYTES, wsums.p, (size_t)nmsm * W * sizeof(ge_ext), hipMemcpyDeviceToHost, st));
    return t;
}

namespace {
void seg_push(MsmSegs &S, const scm *sc, const ge_niels *pts, uint32_t len, uint32_t msm, uint32_t lgblk = 31, const uint32_t *skip = nullptr) {
    if (!len) return;
    if (S.nseg >= BPG_MAX_SEGS) throw std::logic_error("too many MSM segments");
    uint32_t k = S.nseg++;
    S.sc[k] = sc; S.pts[k] = pts; S.len[k] = len; S.msm[k] = msm; S.lgblk[k] = lgblk; S.skip[k] = skip;
    S.start[k + 1] = S.start[k] + len;
}
MsmSegs seg_new() { MsmSegs S; std::memset(&S, 0, sizeof S); return S; }
}  // namespace

void Engine::msm_gens(uint64_t first, uint64_t count, const uint8_t *s, const uint8_t *t, uint8_t out[32]) {
    if (first + count > gens_cap_) throw std::invalid_argument("msm_gens: range beyond capacity");
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    I.shared_now = I.shared_variants();
    I.small_sc.ensure(2 * count * 32 + 64); I.sLR.ensure(2 * count * sizeof(scm) + 64);
    I.msm_result.ensure(4 * sizeof(ge_ext)); I.comp.ensure(128);
    if (count) {
        HIPCHK(hipMemcpyAsync(I.small_sc.p, s, count * 32, hipMemcpyHostToDevice, I.st));
        HIPCHK(hipMemcpyAsync(I.small_sc.as<uint8_t>() + count * 32, t, count * 32, hipMemcpyHostToDevice, I.st));
        BPG_LAUNCH(I, k_sc_from_bytes, dim3(cdiv(2 * count, 256)), dim3(256), I.small_sc.as<uint32_t>(), I.sLR.as<scm>(), (uint32_t)(2 * count));
    }
    MsmSegs S = seg_new();
    seg_push(S, I.sLR.as<scm>(), I.gens.as<ge_niels>() + first, (uint32_t)count, 0);
    seg_push(S, I.sLR.as<scm>() + count, I.gens.as<ge_niels>() + gens_cap_ + first, (uint32_t)count, 0);
    const Impl::MsmTicket tk = I.msm(S, 1);
    HIPCHK(hipStreamSynchronize(I.st));
    h51::pt_compress(out, I.msm_points(tk)[0]);
}

// ------------------------------------------------------------------------------------------------ circuit upload
DeviceCircuit *Engine::upload(const FlatView &c) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    const uint64_t n = c.n, m = c.m, q = c.q, nnz = c.nnz, ncoef = c.ncoef;
    const bool has_witness = !(!c.aL && !c.aR && !c.aO && n > 0);
    if (has_witness && n > 0 && (!c.aL || !c.aR || !c.aO)) throw std::invalid_argument("upload: witness vectors must hold n scalars");
    if (!c.row_ptr || c.row_ptr[0] != 0 || c.row_ptr[q] != nnz || (nnz && (!c.term_var || !c.term_coef)) || (ncoef && !c.coef)) throw std::invalid_argument("upload: malformed CSR");
    if (n >= (1u << 27)) throw std::invalid_argument("upload: too many multipliers");
    // validation on the host (one sequential pass), transposition CSR (by constraint) -> CSC (by variable) on the device (k_csc_*);
    // columns: [0,n) left, [n,2n) right, [2n,3n) output, [3n,3n+m) committed, 3n+m = the constant terms (only the verifier's w_c needs them)
    const uint64_t ncols = 3 * n + m + 1, nvar = ncols - 1;
    if (q >= (1ull << 32) || nnz >= (1ull << 32) || ncols >= (1ull << 32)) throw std::invalid_argument("upload: circuit too large");
    for (uint64_t k = 0; k < nnz; k++) {
        const uint32_t pv = c.term_var[k], kind = pv >> 29, idx = pv & 0x1fffffffu;
        if (c.term_coef[k] >= ncoef) throw std::invalid_argument("upload: coefficient index out of range");
        if (kind <= 2) { if (idx >= n) throw std::invalid_argument("upload: multiplier index out of range"); }
        else if (kind == 3) { if (idx >= m) throw std::invalid_argument("upload: committed index out of range"); }
        else if (kind != 4) throw std::invalid_argument("upload: bad variable kind");
    }
    for (uint64_t r = 0; r < q; r++) if (c.row_ptr[r] > c.row_ptr[r + 1]) throw std::invalid_argument("upload: malformed CSR");
    DeviceCircuit *d = new DeviceCircuit();
    d->n = n; d->m = m; d->q = q; d->ncols = ncols; d->has_witness = has_witness;
    try {
        d->aL.ensure((n ? n : 1) * sizeof(scm)); d->aR.ensure((n ? n : 1) * sizeof(scm)); d->aO.ensure((n ? n : 1) * sizeof(scm));
        d->col_ptr.ensure((ncols + 1) * 8); d->ent_row.ensure((nnz ? nnz : 1) * 4); d->ent_coef.ensure((nnz ? nnz : 1) * 4);
        d->coef.ensure((ncoef ? ncoef : 1) * sizeof(scm));
        {
            // the CSR arrays travel as they are; workspace: the MSM sort buffers (no MSM runs on this context during an upload)
            I.arena.ensure((nnz ? nnz : 1) * 8);                                // term_var | term_coef
            I.plain.ensure((q + 2) * 8 + 64);                                   // row_ptr
            I.counts.ensure((std::max<uint64_t>(nvar, q) + 2) * 4); I.starts.ensure((std::max<uint64_t>(nvar, q) + 2) * 4);
            I.cursor.ensure((std::max<uint64_t>(nvar, q) + 2) * 4); I.heavy.ensure((q + 2) * 4); I.tile_hist.ensure((q + 2) * 4 + 64);
            const uint32_t nb1 = cdiv(nvar ? nvar : 1, SCAN_CHUNK), nb2 = cdiv(q ? q : 1, SCAN_CHUNK);
            I.blocksum.ensure((size_t)(std::max(nb1, nb2) + 2) * 4);
            uint32_t *tv = I.arena.as<uint32_t>(), *tc = tv + (nnz ? nnz : 1);
            uint64_t *rp = I.plain.as<uint64_t>();
            uint32_t *rowconst = I.heavy.as<uint32_t>(), *rowconst_start = I.tile_hist.as<uint32_t>();
            if (nnz) {
                I.h2d(tv, c.term_var, nnz * 4);
                I.h2d(tc, c.term_coef, nnz * 4);
            }
            I.h2d(rp, c.row_ptr, (q + 1) * 8);
            HIPCHK(hipMemsetAsync(I.counts.p, 0, (nvar + 1) * 4, I.st));
            HIPCHK(hipMemsetAsync(rowconst, 0, (q + 1) * 4, I.st));
            if (q) BPG_LAUNCH(I, k_csc_count, dim3(cdiv(q, 256)), dim3(256), rp, tv, (uint32_t)q, (uint32_t)n, (uint32_t)m, I.counts.as<uint32_t>(), rowconst);
            // exclusive scans: variable columns -> starts / cursor, constant terms per row -> rowconst_start
            BPG_LAUNCH(I, k_scan_blocksums, dim3(nb1), dim3(256), I.counts.as<uint32_t>(), (uint32_t)nvar, I.blocksum.as<uint32_t>());
            BPG_LAUNCH(I, k_scan_apply, dim3(nb1), dim3(256), I.counts.as<uint32_t>(), (uint32_t)nvar, I.blocksum.as<uint32_t>(), I.starts.as<uint32_t>(), I.cursor.as<uint32_t>());
            BPG_LAUNCH(I, k_scan_blocksums, dim3(nb2), dim3(256), rowconst, (uint32_t)q, I.blocksum.as<uint32_t>());
            BPG_LAUNCH(I, k_scan_apply, dim3(nb2), dim3(256), rowconst, (uint32_t)q, I.blocksum.as<uint32_t>(), rowconst_start, I.counts.as<uint32_t>() /* scratch */);
            I.extras.ensure(16 * sizeof(scm));
            uint32_t *totals = reinterpret_cast<uint32_t *>(I.extras.as<scm>() + 8);
            BPG_LAUNCH(I, k_csc_colptr, dim3(cdiv(nvar + 1, 256)), dim3(256), I.starts.as<uint32_t>(), rowconst_start, (uint32_t)nvar, (uint32_t)q, d->col_ptr.as<uint64_t>(), totals);
            if (q) BPG_LAUNCH(I, k_csc_fill, dim3(cdiv(q, 256)), dim3(256), rp, tv, tc, (uint32_t)q, (uint32_t)n, (uint32_t)m, I.cursor.as<uint32_t>(), rowconst_start,
                              I.starts.as<uint32_t>() + nvar, d->ent_row.as<uint32_t>(), d->ent_coef.as<uint32_t>());
            HIPCHK(hipGetLastError());
            uint32_t h_tot[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h_tot, totals, 8, hipMemcpyDeviceToHost, I.st));
            HIPCHK(hipStreamSynchronize(I.st));
            d->const_begin = h_tot[0]; d->nnz = h_tot[1];
            if (d->nnz != nnz) throw std::logic_error("upload: transposition lost entries");
        }
        const size_t maxn = std::max<uint64_t>(n, ncoef);
        I.small_sc.ensure((maxn ? maxn : 1) * 32);
        const uint8_t *src[4] = {c.aL, c.aR, c.aO, c.coef};
        DevBuf *dst[4] = {&d->aL, &d->aR, &d->aO, &d->coef};
        const uint64_t cnts[4] = {n, n, n, ncoef};
        for (int k = 0; k < 4; k++) {
            if (!cnts[k] || (k < 3 && !has_witness)) continue;
            I.h2d(I.small_sc.p, src[k], cnts[k] * 32);
            BPG_LAUNCH(I, k_sc_from_bytes, dim3(cdiv(cnts[k], 256)), dim3(256), I.small_sc.as<uint32_t>(), dst[k]->as<scm>(), (uint32_t)cnts[k]);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(I.st));
        }
    } catch (...) { free_circuit(d); throw; }
    return d;
}

void Engine::free_circuit(DeviceCircuit *c) {
    if (!c) return;
    (void)hipSetDevice(device_);
    DevBuf *b[] = {&c->aL, &c->aR, &c->aO, &c->col_ptr, &c->ent_row, &c->ent_coef, &c->coef, &c->mI.skipA, &c->mI.skipB, &c->mI.sc, &c->mI.pts, &c->mO.skipA, &c->mO.skipB, &c->mO.sc, &c->mO.pts};
    for (DevBuf *x : b) x->release();
    delete c;
}

// ------------------------------------------------------------------------------------------------ equal-scalar merging of A_I, A_O (hip/k_merge.cuh)
// Built once per uploaded witness, at its first prove() on the bucket-method path (the generator tables must exist; upload() may precede them).  Cost at
// n = 993,384: a hash-table pass over the scalars, two scans over the table, one point addition per merged-away term and a batched normalisation.
void Engine::Impl::merge_build(DeviceCircuit::MergeSet &M, const scm *A, const ge_niels *PA, uint32_t nA, const scm *B, const ge_niels *PB, uint32_t nB) {
    const uint32_t nterms = nA + nB;
    M.groups = 0; M.skipped = 0;
    if (nterms < 2) return;
    MergeTerms T; T.A = A; T.B = B; T.PA = PA; T.PB = PB; T.nA = nA; T.nterms = nterms;
    const uint32_t lgslots = ceil_log2((uint64_t)2 * nterms), slots = 1u << lgslots;        // load <= 1/2
    const uint32_t nblk = cdiv(slots, SCAN_CHUNK);
    // workspace out of the arena (no MSM of this context is in flight: prove() calls this before its first launch)
    const size_t words = (size_t)7 * slots + 2 * (size_t)nterms + (size_t)nterms / 2 + 8;
    arena.ensure(words * 4); blocksum.ensure((size_t)(nblk + 1) * 4);
    uint32_t *rep = arena.as<uint32_t>(), *count = rep + slots, *msize = count + slots, *gcount = msize + slots, *moff = gcount + slots, *goff = moff + slots + 1,
             *fill = goff + slots + 1, *slot_of = fill + slots, *members = slot_of + nterms, *gslot = members + nterms;
    const uint32_t wA = (nA + 31) / 32, wB = (nB + 31) / 32;
    M.skipA.ensure((size_t)std::max(wA, 1u) * 4); M.skipB.ensure((size_t)std::max(wB, 1u) * 4);
    HIPCHK(hipMemsetAsync(rep, 0xff, (size_t)slots * 4, st));
    HIPCHK(hipMemsetAsync(count, 0, (size_t)slots * 4, st));
    HIPCHK(hipMemsetAsync(M.skipA.p, 0, (size_t)std::max(wA, 1u) * 4, st)); HIPCHK(hipMemsetAsync(M.skipB.p, 0, (size_t)std::max(wB, 1u) * 4, st));
    BPG_LAUNCH((*this), k_merge_insert, dim3(cdiv(nterms, 256)), dim3(256), T, rep, count, slot_of, lgslots);
    BPG_LAUNCH((*this), k_merge_plan, dim3(cdiv(slots, 256)), dim3(256), count, msize, gcount, slots);
    auto scan = [&](uint32_t *in, uint32_t *out) {        // exclusive scan of in[0..slots) -> out[0..slots], out[slots] = total; `fill` is the scan's scratch cursor
        BPG_LAUNCH((*this), k_scan_blocksums, dim3(nblk), dim3(256), in, slots, blocksum.as<uint32_t>());
        BPG_LAUNCH((*this), k_scan_apply, dim3(nblk), dim3(256), in, slots, blocksum.as<uint32_t>(), out, fill);
    };
    scan(msize, moff); scan(gcount, goff);
    HIPCHK(hipGetLastError());
    uint32_t tot[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&tot[0], moff + slots, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[1], goff + slots, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t skipped = tot[0], groups = tot[1];
    if (groups == 0 || skipped > nterms || groups > nterms / 2) return;        // (a group has at least two of the terms)
    HIPCHK(hipMemsetAsync(fill, 0, (size_t)slots * 4, st));
    M.sc.ensure((size_t)groups * sizeof(scm)); M.pts.ensure((size_t)groups * sizeof(ge_niels));
    scratch_ext.ensure((size_t)groups * sizeof(ge_ext));
    BPG_LAUNCH((*this), k_merge_groups, dim3(cdiv(slots, 256)), dim3(256), count, goff, gslot, slots);
    BPG_LAUNCH((*this), k_merge_members, dim3(cdiv(nterms, 256)), dim3(256), T, slot_of, count, moff, fill, members, M.skipA.as<uint32_t>(), M.skipB.as<uint32_t>());
    BPG_LAUNCH((*this), k_merge_sum, dim3(cdiv(groups, 64)), dim3(64), T, count, moff, goff, gslot, members, groups, scratch_ext.as<ge_ext>(), M.sc.as<scm>());
    BPG_LAUNCH((*this), k_normalize_niels, dim3(cdiv(cdiv(groups, NORM_K), 256)), dim3(256), scratch_ext.as<ge_ext>(), M.pts.as<ge_niels>(), groups);
    HIPCHK(hipGetLastError());
    if (merge_equal != 2) HIPCHK(hipStreamSynchronize(st));                // (redone in every proof: the sums that follow are queued on the same stream, no wait)
    M.groups = groups; M.skipped = skipped;
}
void Engine::Impl::merge_witness(DeviceCircuit *c, const ge_niels *Gtab, const ge_niels *Htab) {
    if (c->merge_tried) return;
    c->merge_tried = true;
    const uint32_t n = (uint32_t)c->n;
    if (!merge_equal || n < 2) return;
    const double t0 = now_ms();
    merge_build(c->mI, c->aL.as<scm>(), Gtab, n, c->aR.as<scm>(), Htab, n);
    merge_build(c->mO, c->aO.as<scm>(), Gtab, n, nullptr, nullptr, 0);
    merge_ms_last = now_ms() - t0;
}

// ------------------------------------------------------------------------------------------------ inner-product argument
// InnerProductProof::create (dalek inner_product_proof.rs) on l(x) = lv, r(x) = rv with G_factors = (1.., u..), H_factors = y^-i * G_factors
// and Q = w * B: appends L_k, R_k (lg N pairs) and the final a, b to `proof`.  Rounds above 2^tt_lg generators: grouped folds with
// bucket-method MSMs; from there on: frozen generators and window tables (kernels.cuh).
void Engine::Impl::inner_product(Transcript &T, std::vector<uint8_t> &proof, uint64_t n, uint64_t N, const Scalar &yinv, const Scalar &u_ch,
                                 const Scalar &w, const ge_niels *Gtab, const ge_niels *Htab, const ge_niels *Bn, ProveTimings *tm, double &t0) {
    Impl &I = *this;
    const uint32_t lgN = ceil_log2(N);
    auto lap = [&](double *slot) { if (tm) { I.wait_stream(); double t1 = now_ms(); *slot += t1 - t0; t0 = t1; } };
    T.innerproduct_domain_sep(N);
    std::vector<Scalar> yinv_pow2(lgN + 1);
    yinv_pow2[0] = yinv; for (uint32_t k = 1; k <= lgN; k++) yinv_pow2[k] = yinv_pow2[k - 1] * yinv_pow2[k - 1];
    scm *a = I.lv.as<scm>(), *b = I.rv.as<scm>();
    if (lgN) {
        const uint64_t half = N / 2;
        I.ipa_s.ensure(4 * half * sizeof(scm));
        I.ipa_tabA.ensure(2 * half * sizeof(ge_niels)); I.ipa_tabB.ensure((half > 1 ? half : 2) * sizeof(ge_niels));
        I.naf.ensure(4096);
    }
    Scalar Gamma = Scalar::one(), Eta = Scalar::one();
    const ge_niels *Gst = Gtab, *Hst = Htab;
    const scm w_m = to_scm(w), uch_m = to_scm(u_ch);
    uint64_t mcur = N;
    // table-driven tail state (kernels.cuh "table-driven IPA tail")
    const uint32_t quad_sums = I.shared_now ? 0u : 1u;                 // the block sums of the tail kernels: quad layout for a proof alone (k_points.cuh ge_block_sum_store)
    bool tt_on = false; uint32_t tt_j = 0, tt_lgM0 = 0, tt_cur = 0; Scalar tt_u, tt_uinv; const ge_pniels *tt_wide = nullptr;
    // grouped-fold state
    const uint32_t GRP_STRIDE = 64;
    uint32_t g_j = 0, g_r = 1, g_cur = 0, g_index = 0; uint64_t g_M = N; bool g_first = true; std::vector<Scalar> g_us;
    for (uint32_t round = 0; round < lgN; round++) {
        const uint64_t h = mcur / 2;
        const bool first = round == 0;
        if (!tt_on && ((I.tt_lg > 0 && mcur <= (1ull << I.tt_lg)) || (first && I.tt_orig_lg > 0 && mcur <= (1ull << I.tt_orig_lg)))) {
            // freeze the generators at this level: window tables for G[0..M0), H[0..M0) and B
            tt_on = true; tt_lgM0 = ceil_log2(mcur); tt_j = 0; tt_cur = 0;
            const uint32_t M0 = (uint32_t)mcur;
            I.tt_f.ensure((size_t)2 * M0 * sizeof(scm)); I.tt_c.ensure((size_t)4 * M0 * sizeof(scm));
            I.tt_build(Gst, Hst, Bn, M0, Gst == Gtab && Hst == Htab);          // no-op when this context already holds them (N <= 2^tt_orig_lg)
            tt_wide = (Gst == Gtab && Hst == Htab && Gtab == I.gens.as<ge_niels>()) ? I.wide_ensure(M0) : nullptr;       // original generators: 8-bit windows, built once per device
            BPG_LAUNCH(I, k_tt_factors, dim3(cdiv(M0, 256)), dim3(256), I.yinvpow.as<scm>(), uch_m, (uint32_t)first, (uint32_t)n, M0, to_scm(Gamma), to_scm(Eta),
                       I.tt_f.as<scm>(), I.tt_f.as<scm>() + M0, I.tt_c.as<scm>());
        }
        if (tt_on) {
            const uint32_t M0 = 1u << tt_lgM0;
            scm *c0 = I.tt_c.as<scm>() + (size_t)tt_cur * 2 * M0, *c1 = I.tt_c.as<scm>() + (size_t)(tt_cur ^ 1u) * 2 * M0;
            if (tt_j > 0) {     // apply the previous round's challenge: fold a, b (2*mcur -> mcur) and double the coefficient tables
                BPG_LAUNCH(I, k_tt_advance, dim3(cdiv(std::max<uint64_t>(mcur, 1ull << (tt_j - 1)), 256)), dim3(256), a, b, to_scm(tt_u), to_scm(tt_uinv), (uint32_t)mcur,
                           c0, c1, 1u << (tt_j - 1), M0);
                tt_cur ^= 1u; std::swap(c0, c1);
            }
            const uint32_t nblk = cdiv((uint64_t)M0 * 8, 256);
            if (tt_wide) BPG_LAUNCH(I, k_tt_round8, dim3(nblk, 2), dim3(256), tt_wide, a, b, I.tt_f.as<scm>(), I.tt_f.as<scm>() + M0, c0, tt_lgM0, tt_j, I.tt_partial.as<ge_ext>(), quad_sums);
            else BPG_LAUNCH(I, k_tt_round, dim3(nblk, 2), dim3(256), I.tt_table_p, a, b, I.tt_f.as<scm>(), I.tt_f.as<scm>() + M0, c0, tt_lgM0, tt_j,
                       I.tt_partial.as<ge_ext>(), quad_sums);
            BPG_LAUNCH(I, k_tt_finish, dim3(2), dim3(256), I.tt_partial.as<ge_ext>(), nblk, a, b, (uint32_t)h, w_m,
                       I.tt_table_p + (size_t)2 * M0 * TT_WINDOWS * TT_MULTS, I.msm_result.as<ge_ext>(), quad_sums);
            HIPCHK(hipGetLastError());
            uint8_t lr[64];
            uint32_t *hp = reinterpret_cast<uint32_t *>(I.h_small.as<uint8_t>() + 16384);       // L, R as extended points; encoded on the host
            HIPCHK(hipMemcpyAsync(hp, I.msm_result.p, 2 * sizeof(ge_ext), hipMemcpyDeviceToHost, st));
            I.wait_stream();
            h51::pt_compress(lr, h51::pt_from_device(hp)); h51::pt_compress(lr + 32, h51::pt_from_device(hp + 32));
            lap(tm ? &tm->ipa_msm : nullptr);
            T.append_point("L", lr); T.append_point("R", lr + 32);
            proof.insert(proof.end(), lr, lr + 64);
            tt_u = T.challenge_scalar("u"); tt_uinv = tt_u.invert();
            tt_j++;
            mcur = h;
            if (round + 1 == lgN) {   // last round: only the scalar fold remains
                BPG_LAUNCH(I, k_ipa_fold_scalars, dim3(cdiv(h, 256)), dim3(256), a, b, to_scm(tt_u), to_scm(tt_uinv), (uint32_t)h);
                HIPCHK(hipGetLastError());
            }
            continue;
        }
        // ---- grouped fold rounds (kernels.cuh k_ipa_prep / k_fold_points): sub-round g_j of a group of g_r rounds on tables of size g_M
        if (g_j == 0) {
            g_M = mcur; g_first = first;
            uint32_t left = ceil_log2(mcur);                               // rounds until the tables would be a single point
            if (I.tt_lg > 0 && left > I.tt_lg) left -= I.tt_lg;            // ... or until the tail freezes them
            g_r = std::min<uint32_t>(I.fold_group, left);
            g_us.clear();
            I.grp_c.ensure(4 * GRP_STRIDE * sizeof(scm));
            g_cur = 0;
            hipLaunchKernelGGL(k_set2, dim3(1), dim3(64), 0, st, I.grp_c.as<scm>(), (uint32_t)GRP_STRIDE, to_scm(Gamma), to_scm(Eta));
        }
        scm *c0 = I.grp_c.as<scm>() + (size_t)g_cur * 2 * GRP_STRIDE;
        const uint64_t cnt = g_M / 2;                                       // expanded scalars per side: h * 2^g_j
        scm *sLG = I.ipa_s.as<scm>(), *sLH = sLG + cnt, *sRG = sLH + cnt, *sRH = sRG + cnt;
        const uint32_t blocks = std::min<uint32_t>(cdiv(cnt, 256), 1024);
        const uint32_t lgh = ceil_log2(h);
        I.red_partial.ensure((size_t)blocks * 2 * sizeof(scm) + 4096);
        BPG_LAUNCH(I, k_ipa_prep, dim3(blocks), dim3(256), a, b, I.yinvpow.as<scm>(), c0, c0 + GRP_STRIDE, uch_m,
                           (uint32_t)g_first, (uint32_t)n, lgh, g_j, sLG, sLH, sRG, sRH, I.red_partial.as<scm>());
        BPG_LAUNCH_ID(I, KID_k_reduce_partials, k_reduce_partials_scaled, dim3(2), dim3(256), I.red_partial.as<scm>(), blocks, 2u, I.extras.as<scm>() + 3, w_m);
        Impl::MsmTicket tk;
        {
            MsmSegs S = seg_new();
            const uint32_t lgblk = g_j ? lgh : 31u;                         // every other block of h points of the group-start tables
            seg_push(S, sLG, Gst + h, (uint32_t)cnt, 0, lgblk);
            seg_push(S, sLH, Hst, (uint32_t)cnt, 0, lgblk);
            seg_push(S, I.extras.as<scm>() + 3, Bn, 1, 0);
            seg_push(S, sRG, Gst, (uint32_t)cnt, 1, lgblk);
            seg_push(S, sRH, Hst + h, (uint32_t)cnt, 1, lgblk);
            seg_push(S, I.extras.as<scm>() + 4, Bn, 1, 1);
            tk = I.msm(S, 2);
        }
        uint8_t lr[64];
        I.wait_stream();
        { const std::vector<h51::pt> LR = I.msm_points(tk); h51::pt_compress(lr, LR[0]); h51::pt_compress(lr + 32, LR[1]); }
        lap(tm ? &tm->ipa_msm : nullptr);
        T.append_point("L", lr); T.append_point("R", lr + 32);
        proof.insert(proof.end(), lr, lr + 64);
        const Scalar u = T.challenge_scalar("u"), uinv = u.invert();
        g_us.push_back(u);
        {   // fold a, b and extend the coefficient tables cG, cH (2^g_j -> 2^(g_j+1) entries)
            scm *c1 = I.grp_c.as<scm>() + (size_t)(g_cur ^ 1u) * 2 * GRP_STRIDE;
            BPG_LAUNCH(I, k_tt_advance, dim3(cdiv(std::max<uint64_t>(h, 1ull << g_j), 256)), dim3(256), a, b, to_scm(u), to_scm(uinv), (uint32_t)h,
                       c0, c1, 1u << g_j, (uint32_t)GRP_STRIDE);
            g_cur ^= 1u;
        }
        g_j++;
        if (g_j == g_r) {
            // generator fold of the whole group: Gst'[i] = Gst[i] + sum_{t>=1} sG_t Gst[i + t*Mr], same for H, with
            // sG_t = prod_k (u_k^2)^bit_k(t), sH_t = prod_k (u_k^-2 y^-(g_M/2^k))^bit_k(t), bit_k(t) = bit (g_r - k) of t
            const uint32_t Mr = (uint32_t)(g_M >> g_r), nterms = (1u << g_r) - 1u;
            std::vector<Scalar> fG(g_r), fH(g_r);
            for (uint32_t k = 1; k <= g_r; k++) {
                const Scalar &uk = g_us[k - 1]; const Scalar ukinv = uk.invert();
                fG[k - 1] = uk * uk; fH[k - 1] = ukinv * ukinv * yinv_pow2[ceil_log2(g_M >> k)];
            }
            ge_niels *dst = (g_index & 1) ? I.ipa_tabB.as<ge_niels>() : I.ipa_tabA.as<ge_niels>();
            I.scratch_ext.ensure((size_t)2 * Mr * sizeof(ge_ext));         // the folded points before their normalisation (2^18 of them after the first group of three rounds at 2^20)
            // the group-start tables are the original generators: width-w NAF against their precomputed odd multiples (k_fold_points_wnaf) -
            // unless no table fits the budget of this device, in which case the register kernels below fold them
            const bool use_wnaf = I.fold_wnaf >= 3 && Gst == Gtab && Hst == Htab && Gtab == I.gens.as<ge_niels>() && 2 * Mr > I.fold_split_max
                                  && I.odd_ensure();
            if (use_wnaf) {
                const uint32_t parts = I.eff_parts, L = I.fold_part_bits(), nq = nterms * parts;
                const size_t dbytes = (size_t)4 * nq * 256;
                I.h_naf.ensure(dbytes); I.naf.ensure(dbytes);
                int8_t *hd = I.h_naf.as<int8_t>();
                std::memset(hd, 0, dbytes);
                int32_t top = -1; double adds_fm = 0;
                for (uint32_t q = 0; q < nterms; q++) {
                    const uint32_t t = q + 1;
                    Scalar sg = Scalar::one(), sh = Scalar::one();
                    for (uint32_t k = 1; k <= g_r; k++) if ((t >> (g_r - k)) & 1u) { sg = sg * fG[k - 1]; sh = sh * fH[k - 1]; }
                    const Scalar cls_s[4] = {sg, sg * u_ch, sh, sh * u_ch};
                    const uint64_t lo = (uint64_t)t * Mr, nB = !g_first ? 0 : (lo >= n ? Mr : (lo + Mr > n ? lo + Mr - n : 0));
                    for (int cls = 0; cls < 4; cls++) {
                        if ((cls & 1) && !g_first) continue;
                        for (uint32_t part = 0; part < parts; part++) {
                            int8_t *d = hd + ((size_t)cls * nq + (size_t)part * nterms + q) * 256;
                            const int32_t tp = wnaf256(scalar_bits(cls_s[cls], part * L, L), I.eff_wnaf, d);
                            if (tp > top) top = tp;
                            int adds = 0; for (int k = 0; k < 256; k++) adds += d[k] != 0;
                            adds_fm += 7.0 * adds * ((cls & 1) ? (double)nB : (double)(Mr - nB));
                        }
                    }
                }
                HIPCHK(hipMemcpyAsync(I.naf.p, hd, dbytes, hipMemcpyHostToDevice, st));
                FoldWnaf fw; fw.Mr = Mr; fw.nterms = nterms; fw.first_group = g_first; fw.n = (uint32_t)n; fw.cap = (uint32_t)gens_cap; fw.top = top;
                fw.parts = parts; fw.NM = 1u << (I.eff_wnaf - 2);
                BPG_LAUNCH(I, k_fold_points_wnaf, dim3(cdiv(2 * Mr, 256)), dim3(256), I.gens.as<ge_niels>(), I.gens_odd.as<ge_niels>(), I.scratch_ext.as<ge_ext>(),
                           I.naf.as<uint32_t>(), fw);
                I.prof_note(KID_k_fold_points_wnaf, 32.0 * (2.0 * g_M + 2.0 * Mr), 96.0 * (2.0 * Mr + adds_fm / 7.0) + 128.0 * 2 * Mr, 2.0 * Mr * (8.0 * (top + 1) + 7.0) + adds_fm);
            } else {
            I.h_naf.ensure((size_t)4 * nterms * 16 * 4); I.naf.ensure((size_t)4 * nterms * 16 * 4);
            uint32_t *hn = I.h_naf.as<uint32_t>();
            std::memset(hn, 0, (size_t)4 * nterms * 16 * 4);
            int32_t top = -1; double adds_fm = 0;
            for (uint32_t q = 0; q < nterms; q++) {
                const uint32_t t = q + 1;
                Scalar sg = Scalar::one(), sh = Scalar::one();
                for (uint32_t k = 1; k <= g_r; k++) if ((t >> (g_r - k)) & 1u) { sg = sg * fG[k - 1]; sh = sh * fH[k - 1]; }
                const Scalar cls_s[4] = {sg, sg * u_ch, sh, sh * u_ch};
                // lanes of term t that are padding generators (first group): i + t*Mr >= n
                const uint64_t lo = (uint64_t)t * Mr, nB = !g_first ? 0 : (lo >= n ? Mr : (lo + Mr > n ? lo + Mr - n : 0));
                for (int cls = 0; cls < 4; cls++) {
                    if ((cls & 1) && !g_first) continue;
                    int8_t dg[256]; const int32_t tp = naf256(cls_s[cls], dg);
                    if (tp > top) top = tp;
                    uint32_t *d = hn + ((size_t)cls * nterms + q) * 16; int adds = 0;
                    for (int k = 0; k < 256; k++) { if (dg[k]) { d[k >> 5] |= 1u << (k & 31); adds++; } if (dg[k] < 0) d[8 + (k >> 5)] |= 1u << (k & 31); }
                    adds_fm += 7.0 * adds * ((cls & 1) ? (double)nB : (double)(Mr - nB));
                }
            }
            HIPCHK(hipMemcpyAsync(I.naf.p, hn, (size_t)4 * nterms * 16 * 4, hipMemcpyHostToDevice, st));
            FoldGroup fg; fg.Mr = Mr; fg.nterms = nterms; fg.first_group = g_first; fg.n = (uint32_t)n; fg.top = top;
            int fold_kid = KID_k_fold_points;
            {   // addends in registers when the group size has an instantiation (r = 1..4), from memory otherwise (r = 5)
                const dim3 grid(cdiv(2 * Mr, 256)), block(256);
                ge_ext *fo = I.scratch_ext.as<ge_ext>(); const uint32_t *nf = I.naf.as<uint32_t>();
                const bool regs = nterms == 1 || nterms == 3 || nterms == 7 || nterms == 15;
                const uint32_t split_max = (regs && I.shared_now) ? 0 : I.fold_split_max;    // other proofs fill the device: fewest instructions
                // width-4 NAF steps against odd multiples that the fold kernel makes itself (k_ipa.cuh k_fold_points_quadw / _regw): one list of steps per class (G, H),
                // the multiples in the arena - no multiscalar sum of this context is in flight during a fold
                FoldQuadW fq; std::memset(&fq, 0, sizeof fq); fq.Mr = Mr; fq.nterms = nterms;
                auto make_steps = [&]() {
                    I.h_qsteps.ensure((size_t)2 * QW_MAXSTEPS * 4); I.qsteps.ensure((size_t)2 * QW_MAXSTEPS * 4);
                    uint32_t *hs = I.h_qsteps.as<uint32_t>();
                    double adds_w = 0, dbls_w = 0;
                    for (uint32_t cls = 0; cls < 2; cls++) {
                        std::vector<std::array<int8_t, 256>> dg(nterms);
                        int32_t tp = -1;
                        for (uint32_t q = 0; q < nterms; q++) {
                            const uint32_t t = q + 1;
                            Scalar sc1 = Scalar::one();
                            for (uint32_t k = 1; k <= g_r; k++) if ((t >> (g_r - k)) & 1u) sc1 = sc1 * (cls ? fH[k - 1] : fG[k - 1]);
                            tp = std::max(tp, wnaf256(sc1, 4, dg[q].data()));
                        }
                        uint32_t ns = 0, pending = 0;
                        for (int32_t k = tp; k >= 0; k--) {
                            if (ns) pending++;                              // doubling the identity ahead of the first addition is skipped
                            for (uint32_t q = 0; q < nterms; q++) {
                                const int d = dg[q][k];
                                if (!d) continue;
                                if (ns >= QW_MAXSTEPS || pending > 255) throw std::logic_error("fold: step list overflow");
                                const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                                hs[cls * QW_MAXSTEPS + ns++] = pending | (q << 8) | ((mag >> 1) << 11) | ((d < 0 ? 1u : 0u) << 13);
                                dbls_w += pending; pending = 0;
                            }
                        }
                        fq.nsteps[cls] = ns; fq.tail[cls] = pending; adds_w += ns; dbls_w += pending;
                    }
                    HIPCHK(hipMemcpyAsync(I.qsteps.p, hs, (size_t)2 * QW_MAXSTEPS * 4, hipMemcpyHostToDevice, st));
                    I.arena.ensure((size_t)3 * nterms * 2 * Mr * sizeof(ge_pniels));
                    // bookkeeping below: field multiplications of the whole launch (8 per addition against a projective multiple; P, 2P, 3P, 5P, 7P and three conversions per term)
                    adds_fm = (adds_w * 8.0 + nterms * 2.0 * (7.0 + 8.0 + 7.0 + 2 * 9.0 + 3.0)) * Mr; top = (int32_t)(dbls_w / 2.0) - 1;
                };
                if (2 * Mr <= split_max && nterms >= 1 && nterms <= 7 && I.fold_quad && I.fold_quad_w && !g_first && Mr % 64 == 0) {
                    fold_kid = KID_k_fold_points_quadw;    // four lanes per output
                    make_steps();
                    BPG_LAUNCH_ID(I, fold_kid, k_fold_points_quadw, dim3(cdiv(2 * Mr, 64)), block, Gst, Hst, fo, I.qsteps.as<uint32_t>(), reinterpret_cast<fe *>(I.arena_at(0)), fq);
                } else if (2 * Mr <= split_max && nterms >= 1 && nterms <= 7 && I.fold_quad) {
                    fold_kid = KID_k_fold_points_quad;     // four lanes per output (kernels: k_points.cuh quad_*, k_ipa.cuh)
                    BPG_LAUNCH_ID(I, fold_kid, k_fold_points_quad, dim3(cdiv(2 * Mr, 64)), block, Gst, Hst, fo, nf, fg);
                } else if (2 * Mr <= split_max && nterms >= 3 && nterms <= 15) {
                    fold_kid = KID_k_fold_points_split;
                    BPG_LAUNCH_ID(I, fold_kid, k_fold_points_split, dim3(cdiv(2 * Mr, 64)), block, Gst, Hst, fo, nf, fg);
                } else if (2 * Mr > split_max && I.fold_reg_w && !g_first && nterms >= 1 && nterms <= 7 && Mr % 256 == 0) {
                    fold_kid = KID_k_fold_points_regw;     // one lane per output, every operand from memory: fewest instructions (a device shared with other proofs)
                    make_steps();
                    BPG_LAUNCH_ID(I, fold_kid, k_fold_points_regw, grid, block, Gst, Hst, fo, I.qsteps.as<uint32_t>(), reinterpret_cast<ge_pniels *>(I.arena_at(0)), fq);
                } else if (regs) {
                    fold_kid = KID_k_fold_points_reg;
                    if (nterms == 1) BPG_LAUNCH_ID(I, fold_kid, k_fold_points_reg<1>, grid, block, Gst, Hst, fo, nf, fg);
                    else if (nterms == 3) BPG_LAUNCH_ID(I, fold_kid, k_fold_points_reg<3>, grid, block, Gst, Hst, fo, nf, fg);
                    else if (nterms == 7) BPG_LAUNCH_ID(I, fold_kid, k_fold_points_reg<7>, grid, block, Gst, Hst, fo, nf, fg);
                    else BPG_LAUNCH_ID(I, fold_kid, k_fold_points_reg<15>, grid, block, Gst, Hst, fo, nf, fg);
                } else BPG_LAUNCH(I, k_fold_points, grid, block, Gst, Hst, fo, nf, fg);
            }
            // bookkeeping for the roofline: 2*g_M points read + 2*Mr written at 32 B (information content) resp. 96/128 B (device formats);
            // field multiplications: 8 per doubling, 7 per mixed addition (the split variant runs the doublings once per wave of a block)
            I.prof_note(fold_kid, 32.0 * (2.0 * g_M + 2.0 * Mr), 96.0 * 2 * g_M + 128.0 * 2 * Mr,
                        2.0 * Mr * (8.0 * (top + 1) * (fold_kid == KID_k_fold_points_split ? (nterms < 4 ? nterms : 4) : 1) + 7.0) + adds_fm);
            }
            BPG_LAUNCH(I, k_normalize_niels, dim3(cdiv(cdiv(2 * Mr, NORM_K), 256)), dim3(256), I.scratch_ext.as<ge_ext>(), dst, 2 * Mr);
            HIPCHK(hipGetLastError());
            I.wait_stream();                               // h_naf is reused by the next group
            Gst = dst; Hst = dst + Mr;
            for (const Scalar &uk : g_us) { Gamma = uk.invert() * Gamma; Eta = uk * Eta; }
            g_j = 0; g_index++;
        }
        mcur = h;
        lap(tm ? &tm->ipa_fold : nullptr);
    }
    scm ab[2];
    HIPCHK(hipMemcpyAsync(&ab[0], a, sizeof(scm), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&ab[1], b, sizeof(scm), hipMemcpyDeviceToHost, st));
    I.wait_stream();
    { uint8_t o[64]; from_scm(ab[0]).to_bytes(o); from_scm(ab[1]).to_bytes(o + 32); proof.insert(proof.end(), o, o + 64); }
}

// ------------------------------------------------------------------------------------------------ prove
// Speculative start of Prover::prove's TranscriptRng (include/bpg.h: bpg_blinding_begin).  The 2n + 3 leading draws depend on the transcript
// after the last commitment (+ the "m" suffix), the commitment blindings and the external seed - not on the constraints or on n - so a
// host thread can draw them while the caller still assembles the circuit.  prove() takes the stream over only when all three match.
void Engine::blinding_begin(const Transcript &after_commitments, const std::vector<Scalar> &v_blinding, const uint8_t seed[32], uint64_t max_multipliers) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    if (max_multipliers == 0) { I.blind_cancel(); return; }
    if (I.chain && I.chain->pool) {     // a pool that was destroyed under this context (its threads are gone): fall back to the context's own worker
        bool gone; { std::lock_guard<std::mutex> lk(I.chain->mu); gone = I.chain->quit; }
        if (gone) { I.blind_cancel(); I.chain.reset(); I.pool_streams = 0; }
    }
    using BS = Impl::BlindStream;
    const size_t max_alive = I.pool_streams ? I.pool_streams : (size_t)I.chain_workers * I.chain_lanes + 1;
    while (I.blinds.size() >= max_alive) { I.blind_retire(I.blinds.front()); I.blinds.pop_front(); }     // the oldest gives way
    auto b = std::make_shared<BS>();
    Transcript T = after_commitments;
    T.append_u64("m", v_blinding.size());
    T.export_state(b->state);
    std::memcpy(b->seed, seed, 32); b->vb = v_blinding;
    TranscriptRng rng = T.build_rng(v_blinding, seed);
    for (int k = 0; k < 3; k++) b->first[k] = rng.random_scalar();
    b->max_draws = ((2 * max_multipliers + BS::SNAP - 1) / BS::SNAP) * BS::SNAP;
    b->inject_fail = I.test_fail_upload; I.test_fail_upload = 0;
    // pinned slab: one that no alive stream owns; its previous owner must have left it (its uploads were synchronised by the prove() that used it)
    if (I.h_blind.size() < max_alive) { I.h_blind.resize(max_alive); I.slab_owner.resize(max_alive); while (I.slab_dev.size() < max_alive) I.slab_dev.emplace_back(std::make_unique<Impl::SlabDev>()); }
    int slot = -1;
    for (size_t k = 0; k < max_alive && slot < 0; k++) {
        bool used = false;
        for (const std::shared_ptr<BS> &x : I.blinds) used |= x->slot == (int)k;
        if (!used) slot = (int)k;
    }
    if (slot < 0) throw std::logic_error("blinding_begin: no free slab");
    b->slot = slot;
    if (I.slab_owner[slot]) { I.blind_retire(I.slab_owner[slot]); I.slab_owner[slot].reset(); }
    Impl::SlabDev &sd = *I.slab_dev[slot];
    if (!sd.copy_st) HIPCHK(hipStreamCreateWithFlags(&sd.copy_st, hipStreamNonBlocking));
    HIPCHK(hipStreamSynchronize(sd.copy_st));                  // no copy of the previous owner still reads the slab
    I.h_blind[slot].ensure(b->max_draws * 64);
    sd.d.ensure(b->max_draws * 64);
    while (sd.ev.size() < b->max_draws / BS::UP + 1) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); sd.ev.push_back(e); }
    // The device slab changes owner here.  Two things keep this proof from reading the previous owner's draws: BlindStream::err - an upload that
    // fails is recorded BEFORE its block is published and prove() refuses the stream - and a canary against a copy that is dropped WITHOUT an error:
    // the first draw of every block is overwritten with a pattern here (1 KB of stores for a 2^20 proof, on the copy stream, ahead of the uploads),
    // k_sc_from_wide raises stale_flag on a draw that still reads as the pattern, and prove() checks the flag before the proof leaves it.  (Round 3
    // overwrote the whole slab - a 128 MB memset per 2^20 proof - with a constant nothing ever checked for.)
    {
        const uint32_t nblk = (uint32_t)((b->max_draws + BS::UP - 1) / BS::UP);        // the last block may be a partial one
        if (nblk) hipLaunchKernelGGL(k_blind_poison, dim3(cdiv((uint64_t)nblk * 16, 256)), dim3(256), 0, sd.copy_st, sd.d.as<uint32_t>(), nblk, (uint32_t)BS::UP);
        HIPCHK(hipGetLastError());
    }
    b->raw = I.h_blind[slot].as<uint8_t>();
    b->device = device_; b->copy_st = sd.copy_st; b->d_raw = sd.d.as<uint8_t>(); b->ev = &sd.ev;
    // how a drawn block reaches the device (called on the chain thread; host/chain.hpp publishes the block afterwards): an asynchronous copy on the
    // slab's own copy stream, then the block's event; the threads of a pool serve contexts of different devices, hence the hipSetDevice
    b->upload = [](BlindStream &bs, uint64_t from, uint64_t to, uint64_t k) -> int {
        hipError_t e = hipSetDevice(bs.device);
        if (e == hipSuccess && bs.inject_fail != 2) e = hipMemcpyAsync(bs.d_raw + 64 * from, bs.raw + 64 * from, (to - from) * 64, hipMemcpyHostToDevice, static_cast<hipStream_t>(bs.copy_st));
        if (e == hipSuccess) e = hipEventRecord((*static_cast<std::vector<hipEvent_t> *>(bs.ev))[k], static_cast<hipStream_t>(bs.copy_st));
        return (int)e;
    };
    b->snaps.assign(b->max_draws / BS::SNAP + 1, rng);
    I.slab_owner[slot] = b;
    I.blinds.push_back(b);
    if (!I.chain) { I.chain = std::make_shared<Impl::ChainWorker>(); I.chain->lanes = I.chain_lanes; (void)keccak_impl(); }
    if (!I.chain->pool) while (I.chain->th.size() < I.chain_workers) I.chain->th.emplace_back(Impl::ChainWorker::run, I.chain.get(), I.chain->lanes);
    I.chain->push(b);
}
// ChainPool: chain threads shared by the contexts attached to it (engine.hpp)
struct ChainPool::Impl { std::shared_ptr<Engine::Impl::ChainWorker> w; };
ChainPool::ChainPool(const std::vector<uint32_t> &lanes_per_thread) : impl_(new Impl()) {
    if (lanes_per_thread.empty() || lanes_per_thread.size() > 256) { delete impl_; throw std::invalid_argument("chain pool: 1..256 threads"); }
    for (uint32_t l : lanes_per_thread) if (l < 1 || l > 8) { delete impl_; throw std::invalid_argument("chain pool: 1..8 lanes per thread"); }
    impl_->w = std::make_shared<Engine::Impl::ChainWorker>();
    impl_->w->pool = true;
    (void)keccak_impl();
    for (uint32_t l : lanes_per_thread) { impl_->w->th.emplace_back(Engine::Impl::ChainWorker::run, impl_->w.get(), l); capacity_ += l; }
}
ChainPool::~ChainPool() { impl_->w->stop(); delete impl_; }
void Engine::attach_chain_pool(ChainPool *pool, uint32_t max_streams) {
    impl_->chain_shutdown();                                  // streams in flight are dropped
    if (!pool) return;
    if (max_streams < 1 || max_streams > 64) throw std::invalid_argument("chain pool: 1..64 streams per context");
    impl_->chain = pool->impl_->w; impl_->pool_streams = max_streams;
}
void Engine::set_chain_workers(uint32_t n) {
    if (n < 1 || n > 64) throw std::invalid_argument("chain workers: 1..64");
    impl_->chain_shutdown();                                  // streams in flight are dropped; threads restart with the next begin
    impl_->chain_workers = n;
}
void Engine::set_chain_lanes(uint32_t n) {
    if (n < 1 || n > 8) throw std::invalid_argument("chain lanes: 1..8");
    impl_->chain_shutdown();
    impl_->chain_lanes = n;
}
void Engine::blinding_cancel() { impl_->blind_cancel(); }
void Engine::test_fail_next_upload() { impl_->test_fail_upload = 1; }
void Engine::test_drop_next_upload() { impl_->test_fail_upload = 2; }
uint64_t Engine::table_bytes() const { return table_bytes_held(device_); }

int Engine::chain_cpu() const { return impl_->last_chain_cpu; }
bool Engine::last_shared_variants() const { return impl_->shared_now; }

std::vector<uint8_t> Engine::prove(DeviceCircuit *c, Transcript &T, const std::vector<Scalar> &v_blinding,
                                   const uint8_t rng_seed[32], uint32_t flags, ProveTimings *tm) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    hipStream_t st = I.st;
    ProvingGuard in_flight(device_);
    // the kernel variants of this proof are chosen HERE, once: a proof that enters or leaves prove() on another thread must not flip the
    // schedule half-way through (ProveTimings::shared_variants reports the choice, so a failing proof can be replayed with BPG_FOLD_ADAPT=0 or 2)
    I.shared_now = I.shared_variants();
    if (tm) tm->shared_variants = I.shared_now ? 1u : 0u;
    const uint64_t n = c->n, m = c->m, q = c->q;
    if (v_blinding.size() != m) throw std::invalid_argument("prove: need one blinding factor per committed variable");
    if (!c->has_witness) throw R1CSException(R1CSError::MissingAssignment, "prove: the uploaded circuit carries no assignments (verifier-side instance)");
    uint64_t N = 1; while (N < n) N <<= 1;
    const uint32_t lgN = ceil_log2(N);
    if (gens_cap_ < N) throw R1CSException(R1CSError::InvalidGeneratorsLength, "generator capacity below padded circuit size");
    const bool compact = flags & 1u, no_1phase = flags & 2u;
    const double t_begin = now_ms();
    double t0 = t_begin;
    auto lap = [&](double *slot) { if (tm) { I.wait_stream(); double t1 = now_ms(); *slot += t1 - t0; t0 = t1; } };

    const ge_niels *Gtab = I.gens.as<ge_niels>(), *Htab = I.gens.as<ge_niels>() + gens_cap_;
    const ge_niels *Bn = I.bases.as<ge_niels>(), *Bbn = Bn + 1;

    // ---- transcript, RNG, first blindings
    T.append_u64("m", m);
    const bool expanded = (flags & 4u) != 0;          // BPG_FLAG_EXPANDED_BLINDING: no host chain to hide behind
    // a speculative blinding stream is used only when it was drawn from exactly this transcript state, these blindings and this seed
    std::shared_ptr<Impl::BlindStream> bs;
    if (!I.blinds.empty()) {
        uint8_t now[203]; T.export_state(now);
        for (const std::shared_ptr<Impl::BlindStream> &cand : I.blinds) {
            const Impl::BlindStream &b = *cand;
            bool ok = !expanded && 2 * n <= b.max_draws && std::memcmp(now, b.state, 203) == 0 && std::memcmp(rng_seed, b.seed, 32) == 0 && b.vb.size() == v_blinding.size();
            for (size_t i = 0; ok && i < b.vb.size(); i++) ok = std::memcmp(b.vb[i].as_bytes(), v_blinding[i].as_bytes(), 32) == 0;
            if (ok) { bs = cand; break; }
        }
        // no match: the alive streams may belong to later proofs of a sequence; they are bounded (two) and replaced by the next begin
    }
    // (first[] and snaps[0] were written by blinding_begin on this thread, before the worker saw the stream)
    TranscriptRng rng = T.build_rng(v_blinding, rng_seed);     // with a stream: replaced by the stream's generator once the 2n draws are in
    Scalar ib, ob, sb;
    if (bs) { ib = bs->first[0]; ob = bs->first[1]; sb = bs->first[2]; }
    else { ib = rng.random_scalar(); ob = rng.random_scalar(); sb = rng.random_scalar(); }

    // small device scalars: extras[0..2] = ib, ob, sb ; [3..4] = cL*w, cR*w (per round)
    I.extras.ensure(16 * sizeof(scm));
    I.h_small.ensure(1 << 16);
    {
        scm *hs = I.h_small.as<scm>();
        hs[0] = to_scm(ib); hs[1] = to_scm(ob); hs[2] = to_scm(sb);
        HIPCHK(hipMemcpyAsync(I.extras.p, hs, 3 * sizeof(scm), hipMemcpyHostToDevice, st));
    }
    I.msm_result.ensure(4 * sizeof(ge_ext)); I.comp.ensure(256);

    // ---- A_I, A_O (do not depend on s_L, s_R): launch, then draw the 2n RNG scalars on the host while they run
    const bool tabled = I.tt_orig_lg > 0 && N <= (1ull << I.tt_orig_lg) && n > 0;   // the generators have (or get) window tables: A_I, A_O, S are table sums
    if (tabled) I.tt_build(Gtab, Htab, Bn, (uint32_t)N, true);
    // the stream of this proof was drawn ahead (a sequence of proofs: its chain ran under the previous proof's kernels) and is complete
    const bool chain_ready = bs && bs->produced.load(std::memory_order_acquire) >= 2 * n;
    const bool merged = expanded || tabled || n < 4096 || chain_ready;   // nothing to hide behind: A_I, A_O, S in one pass after the draws (one serial tail, not four)
    Impl::MsmTicket tk_aiao{0, 0, 0}, tk_s[3]; uint32_t nparts = 0;
    // A_I's terms with equal scalars share their bucket entries (hip/k_merge.cuh): grouped once per uploaded witness, here at its first proof
    if (!tabled) { if (I.merge_equal == 2) c->merge_tried = false; I.merge_witness(c, Gtab, Htab); }
    I.merged_last = tabled ? 0u : c->mI.groups + c->mO.groups; I.merged_skipped_last = tabled ? 0u : c->mI.skipped + c->mO.skipped;
    auto push_AI = [&](MsmSegs &S) {        // <a_L, G> + <a_R, H> + i_blinding * B_blinding as result 0
        const bool mg = c->mI.groups != 0;
        seg_push(S, c->aL.as<scm>(), Gtab, (uint32_t)n, 0, 31, mg ? c->mI.skipA.as<uint32_t>() : nullptr);
        seg_push(S, c->aR.as<scm>(), Htab, (uint32_t)n, 0, 31, mg ? c->mI.skipB.as<uint32_t>() : nullptr);
        if (mg) { seg_push(S, c->mI.sc.as<scm>(), c->mI.pts.as<ge_niels>(), c->mI.groups, 0); I.msm_alg_discount += c->mI.groups; I.msm_skipped_terms += c->mI.skipped; }
        seg_push(S, I.extras.as<scm>() + 0, Bbn, 1, 0);
    };
    auto push_AO = [&](MsmSegs &S) {        // <a_O, G> + o_blinding * B_blinding as result 1
        const bool mg = c->mO.groups != 0;
        seg_push(S, c->aO.as<scm>(), Gtab, (uint32_t)n, 1, 31, mg ? c->mO.skipA.as<uint32_t>() : nullptr);
        if (mg) { seg_push(S, c->mO.sc.as<scm>(), c->mO.pts.as<ge_niels>(), c->mO.groups, 1); I.msm_alg_discount += c->mO.groups; I.msm_skipped_terms += c->mO.skipped; }
        seg_push(S, I.extras.as<scm>() + 1, Bbn, 1, 1);
    };
    if (!merged) {
        MsmSegs S = seg_new();
        push_AI(S); push_AO(S);
        tk_aiao = I.msm(S, 2);
    }
    const double t_rng0 = now_ms();
    if (!bs && !expanded) { I.h_raw.ensure((2 * n ? 2 * n : 1) * 64); I.raw_rng.ensure((2 * n ? 2 * n : 1) * 64); }      // the draws of a chain made inside this call; a blinding stream brings its own slabs
    I.sLR.ensure((2 * n ? 2 * n : 1) * sizeof(scm));
    scm *sL = I.sLR.as<scm>(), *sR = sL + n;
    // S = <s_L, G> + <s_R, H> + sb * B_blinding is accumulated in pieces as the draws arrive: <s_L, G> once s_L is complete,
    // the first 7/8 of <s_R, H> next, and only the last eighth (+ the blinding term) after the chain has ended.  (A last piece of 1/32 takes 0.05 ms off a lone
    // proof and nothing off a burst: thirteen chains end together, and their second pieces - a million terms each - then start 5 ms before the end instead of 19.)
    struct Piece { uint64_t a, b; } pieces[3] = {{0, n}, {n, n + (n - n / 8)}, {n + (n - n / 8), 2 * n}};
    if (n < (1u << 17)) { pieces[0] = {0, 0}; pieces[1] = {0, 0}; pieces[2] = {0, 2 * n}; }      // short chain: one MSM after it (each call has a ~1 ms serial tail)
    uint32_t next_piece = 0;
    auto launch_pieces = [&](uint64_t drawn) {
        while (next_piece < 3 && pieces[next_piece].b <= drawn) {
            const Piece pc = pieces[next_piece];
            const bool last = next_piece == 2;
            next_piece++;
            if (pc.b == pc.a && !last) continue;
            MsmSegs S = seg_new();
            if (pc.a < n) seg_push(S, sL + pc.a, Gtab + pc.a, (uint32_t)(std::min<uint64_t>(pc.b, n) - pc.a), 0);
            if (pc.b > n) { const uint64_t a2 = std::max<uint64_t>(pc.a, n) - n; seg_push(S, sR + a2, Htab + a2, (uint32_t)(pc.b - n - a2), 0); }
            if (last) seg_push(S, I.extras.as<scm>() + 2, Bbn, 1, 0);
            tk_s[nparts++] = I.msm(S, 1);
        }
    };
    if (expanded) {     // (include/bpg.h): one draw K, the 2n scalars are expanded from it on the device
        uint8_t msg[80]; std::memset(msg, 0, sizeof msg);
        std::memcpy(msg, "bpg blinding v1", 15);
        rng.fill_bytes(msg + 15, 64);
        BlindHead head; std::memcpy(head.lane, msg, 80); head.lane[9] &= 0x00ffffffffffffffULL;      // byte 79 belongs to the index
        if (n) BPG_LAUNCH(I, k_blind_expand, dim3(cdiv(2 * n, 256)), dim3(256), head, sL, (uint32_t)(2 * n));
    } else
    {   // s_L[0..n) then s_R[0..n): 64 uniform bytes each, drawn in slabs; each slab is uploaded and reduced mod l while the
        // host draws the next one (the copies queue behind the A_I/A_O kernels on the stream and overlap the serial chain)
        uint8_t *raw = bs ? bs->raw : I.h_raw.as<uint8_t>();
        const uint64_t slab = 1u << 16;
        if (bs && chain_ready && 2 * n > 0) {
            // the whole chain was drawn (and handed to the copy stream) before this proof started - a sequence of proofs with its chains drawn
            // ahead: wait for every block's event, then ONE conversion launch instead of one per 4 MB block (31 at 2^20)
            const uint64_t nblk = (2 * n + Impl::BlindStream::UP - 1) / Impl::BlindStream::UP;
            while (bs->uploaded_blocks.load(std::memory_order_acquire) < nblk) {
                if (bs->finished.load(std::memory_order_acquire) && bs->uploaded_blocks.load(std::memory_order_acquire) < nblk) break;     // stopped short: handled below
                std::this_thread::sleep_for(std::chrono::microseconds(40));
            }
            if (const int uerr = bs->err.load(std::memory_order_acquire)) {
                Impl::blind_stop(bs);
                for (auto it = I.blinds.begin(); it != I.blinds.end(); ++it) if (it->get() == bs.get()) { I.blinds.erase(it); break; }
                throw DeviceError(std::string("upload of the blinding draws failed: ") + hipGetErrorString((hipError_t)uerr));
            }
            if (bs->uploaded_blocks.load(std::memory_order_acquire) < nblk) throw DeviceError("the blinding stream ended before its draws were uploaded");
            for (uint64_t k = 0; k < nblk; k++) HIPCHK(hipStreamWaitEvent(st, (*static_cast<std::vector<hipEvent_t> *>(bs->ev))[k], 0));
            BPG_LAUNCH(I, k_sc_from_wide, dim3(cdiv(2 * n, 256)), dim3(256), reinterpret_cast<const uint32_t *>(bs->d_raw), sL, (uint32_t)(2 * n), I.stale_flag.as<uint32_t>());
        } else
        {
        // the draws are converted (64 uniform bytes -> a scalar mod l) when somebody needs them: before each piece of S and after the last draw - three launches
        // for a 2^20-gate proof where rounds 1-4 made one per slab (31), every one of them behind the chain and none of them needed so early
        uint64_t conv_from = 0;
        auto convert_to = [&](uint64_t hi) {
            if (hi <= conv_from) return;
            const uint32_t *src = bs ? reinterpret_cast<const uint32_t *>(bs->d_raw) : I.raw_rng.as<uint32_t>();
            BPG_LAUNCH(I, k_sc_from_wide, dim3(cdiv(hi - conv_from, 256)), dim3(256), src + 16 * conv_from, sL + conv_from, (uint32_t)(hi - conv_from), I.stale_flag.as<uint32_t>());
            conv_from = hi;
        };
        for (uint64_t i = 0; i < 2 * n; i += slab) {
            const uint64_t cnt = std::min<uint64_t>(slab, 2 * n - i);
            if (bs) {
                // drawn (or being drawn) and uploaded block by block by the chain worker: wait until block i / UP has been handed to the copy
                // stream, then make this stream wait for its event - no host copy, no copy on this stream
                const uint64_t k = i / Impl::BlindStream::UP;
                while (bs->uploaded_blocks.load(std::memory_order_acquire) <= k) std::this_thread::sleep_for(std::chrono::microseconds(40));
                if (const int uerr = bs->err.load(std::memory_order_acquire)) {
                    // the slab on the device may still hold an earlier proof's draws: s_L, s_R must never be built from it
                    Impl::blind_stop(bs);
                    for (auto it = I.blinds.begin(); it != I.blinds.end(); ++it) if (it->get() == bs.get()) { I.blinds.erase(it); break; }
                    throw DeviceError(std::string("upload of the blinding draws failed: ") + hipGetErrorString((hipError_t)uerr));
                }
                HIPCHK(hipStreamWaitEvent(st, (*static_cast<std::vector<hipEvent_t> *>(bs->ev))[k], 0));
            } else {
                rng.fill_draws64(raw + 64 * i, cnt);
                HIPCHK(hipMemcpyAsync(I.raw_rng.as<uint8_t>() + 64 * i, raw + 64 * i, cnt * 64, hipMemcpyHostToDevice, st));
            }
            if (!merged && i + cnt < 2 * n && next_piece < 3 && pieces[next_piece].b <= i + cnt) { convert_to(i + cnt); launch_pieces(i + cnt); }
        }
        convert_to(2 * n);
        }
    }
    if (bs) {   // take the generator back: the state before draw 2n is the last snapshot at or below it, advanced by the remainder
        const uint64_t K = (2 * n) / Impl::BlindStream::SNAP, rem = 2 * n - K * Impl::BlindStream::SNAP;
        while (bs->produced.load(std::memory_order_acquire) < K * Impl::BlindStream::SNAP) std::this_thread::sleep_for(std::chrono::microseconds(20));
        rng = bs->snaps[K];
        if (rem) { std::vector<uint8_t> skip(rem * 64); rng.fill_draws64(skip.data(), rem); }
        // done with the stream: the worker stops at its next snapshot and moves on to the next queued one; the pinned slab stays allocated
        // (the queued uploads still read it) and is not handed out again before this prove() has synchronised the stream
        Impl::blind_stop(bs);
        for (auto it = I.blinds.begin(); it != I.blinds.end(); ++it) if (it->get() == bs.get()) { I.blinds.erase(it); break; }
        { const int c = bs->cpu.load(std::memory_order_relaxed); if (c >= 0) I.last_chain_cpu = c; }
        bs.reset();
    }
    if (tm) tm->rng_host += now_ms() - t_rng0;
    lap(tm ? &tm->msm_aiao : nullptr);
    uint8_t pts[96];
    if (tabled) {
        const uint32_t M0 = (uint32_t)N, nblk = cdiv((uint64_t)M0 * 16, 256);
        BPG_LAUNCH(I, k_tt_commit3, dim3(nblk, 3), dim3(256), I.tt_table_p, c->aL.as<scm>(), c->aR.as<scm>(), c->aO.as<scm>(), sL, sR,
                   (uint32_t)n, M0, I.tt_partial.as<ge_ext>(), I.shared_now ? 0u : 1u);
        BPG_LAUNCH(I, k_tt_commit3_finish, dim3(3), dim3(256), I.tt_partial.as<ge_ext>(), nblk, I.extras.as<scm>(),
                   I.ped_table.as<ge_pniels>() + (size_t)TT_WINDOWS * TT_MULTS, I.msm_result.as<ge_ext>(), I.shared_now ? 0u : 1u);
        HIPCHK(hipGetLastError());
        uint32_t *hp = reinterpret_cast<uint32_t *>(I.h_small.as<uint8_t>() + 16384);           // three extended points; encoded on the host
        HIPCHK(hipMemcpyAsync(hp, I.msm_result.p, 3 * sizeof(ge_ext), hipMemcpyDeviceToHost, st));
        I.wait_stream();
        for (int k = 0; k < 3; k++) h51::pt_compress(pts + 32 * k, h51::pt_from_device(hp + 32 * k));
    } else if (merged) {
        MsmSegs S = seg_new();
        push_AI(S); push_AO(S);
        seg_push(S, sL, Gtab, (uint32_t)n, 2);
        seg_push(S, sR, Htab, (uint32_t)n, 2);
        seg_push(S, I.extras.as<scm>() + 2, Bbn, 1, 2);
        const Impl::MsmTicket tk = I.msm(S, 3);
        I.wait_stream();
        const std::vector<h51::pt> P3 = I.msm_points(tk);
        for (int k = 0; k < 3; k++) h51::pt_compress(pts + 32 * k, P3[k]);
    } else {
        launch_pieces(2 * n);
        I.wait_stream();
        const std::vector<h51::pt> AB = I.msm_points(tk_aiao);
        h51::pt Sp = I.msm_points(tk_s[0])[0];
        for (uint32_t k = 1; k < nparts; k++) Sp = h51::pt_add(Sp, I.msm_points(tk_s[k])[0]);
        h51::pt_compress(pts, AB[0]); h51::pt_compress(pts + 32, AB[1]); h51::pt_compress(pts + 64, Sp);
    }
    lap(tm ? &tm->msm_s : nullptr);

    std::vector<uint8_t> proof;
    proof.reserve(14 * 32 + (2 * lgN + 2) * 32 + 1);
    if (compact) proof.push_back(0);
    proof.insert(proof.end(), pts, pts + 96);
    T.append_point("A_I1", pts); T.append_point("A_O1", pts + 32); T.append_point("S1", pts + 64);
    if (!no_1phase) T.r1cs_1phase_domain_sep();
    const uint8_t ident[32] = {0};
    T.append_point("A_I2", ident); T.append_point("A_O2", ident); T.append_point("S2", ident);
    if (!compact) proof.insert(proof.end(), 96, 0);

    const Scalar y = T.challenge_scalar("y"), z = T.challenge_scalar("z");
    const Scalar yinv = y.invert();

    // ---- powers, flattened weights, t-polynomial
    // y^-i stays for the inner-product argument; y^i, z^j and the flattened weights are needed until k_poly_eval only: they take the arena between the
    // S sums (synchronised above) and the first multiscalar sum of the inner-product argument
    I.yinvpow.ensure(N * sizeof(scm));
    const size_t b_w = Impl::al256((c->ncols ? c->ncols : 1) * sizeof(scm)), b_z = Impl::al256((q + 2) * sizeof(scm)), b_y = Impl::al256(N * sizeof(scm));
    I.arena.ensure(b_w + b_z + b_y);
    scm *const wAll_p = reinterpret_cast<scm *>(I.arena_at(0)), *const zpow_p = reinterpret_cast<scm *>(I.arena_at(b_w)), *const ypow_p = reinterpret_cast<scm *>(I.arena_at(b_w + b_z));
    {   // y^i, y^-i, z^j: one launch
        ExpTables E; std::memset(&E, 0, sizeof E);
        uint32_t k = 0, lgmax = 0;
        auto add = [&](const Scalar &base, scm *out, uint64_t count) {
            uint32_t lgT = ceil_log2(count); if (lgT > 16) lgT = 16;
            E.base[k] = to_scm(base); E.out[k] = out; E.count[k] = (uint32_t)count; E.lgT[k] = lgT; lgmax = std::max(lgmax, lgT); k++;
        };
        add(y, ypow_p, N); add(yinv, I.yinvpow.as<scm>(), N); add(z, zpow_p, q + 1);
        BPG_LAUNCH(I, k_exp_table, dim3(cdiv(1u << lgmax, 256), k), dim3(256), E);
    }
    if (c->ncols > 1)      // every column but the last (constant terms: verifier only)
        BPG_LAUNCH(I, k_flatten, dim3(cdiv(c->ncols - 1, 256)), dim3(256), c->col_ptr.as<uint64_t>(), c->ent_row.as<uint32_t>(),
                           c->ent_coef.as<uint32_t>(), c->coef.as<scm>(), zpow_p, wAll_p, (uint32_t)(c->ncols - 1), (uint32_t)(3 * n));
    scm *wL = wAll_p, *wR = wL + n, *wO = wR + n, *wV = wO + n;
    const uint32_t pblocks = n ? std::min<uint32_t>(cdiv(n, 256), 1024) : 1;
    I.red_partial.ensure((size_t)pblocks * 6 * sizeof(scm) + 4096); I.red_out.ensure(16 * sizeof(scm));
    BPG_LAUNCH(I, k_poly_t, dim3(pblocks), dim3(256), c->aL.as<scm>(), c->aR.as<scm>(), c->aO.as<scm>(), sL, sR, wL, wR, wO,
                       ypow_p, I.yinvpow.as<scm>(), I.red_partial.as<scm>(), (uint32_t)n);
    BPG_LAUNCH(I, k_reduce_partials, dim3(6), dim3(256), I.red_partial.as<scm>(), pblocks, 6u, I.red_out.as<scm>());
    HIPCHK(hipGetLastError());
    scm h_t[6]; std::vector<scm> h_wV(m ? m : 1);
    HIPCHK(hipMemcpyAsync(h_t, I.red_out.p, 6 * sizeof(scm), hipMemcpyDeviceToHost, st));
    if (m) HIPCHK(hipMemcpyAsync(h_wV.data(), wV, m * sizeof(scm), hipMemcpyDeviceToHost, st));
    uint32_t *h_stale = reinterpret_cast<uint32_t *>(I.h_small.as<uint8_t>() + 32768);
    HIPCHK(hipMemcpyAsync(h_stale, I.stale_flag.p, 4, hipMemcpyDeviceToHost, st));
    I.wait_stream();
    if (*h_stale) {   // a blinding draw was read from a slab position no upload of this proof wrote (k_sc_from_wide): nothing of this proof may leave
        HIPCHK(hipMemsetAsync(I.stale_flag.p, 0, 4, st));
        throw DeviceError("a blinding draw was read before it was uploaded (stale device slab): proof withheld");
    }
    Scalar t[7], tb[7];
    for (int k = 0; k < 6; k++) t[k + 1] = from_scm(h_t[k]);
    tb[1] = rng.random_scalar(); tb[3] = rng.random_scalar(); tb[4] = rng.random_scalar(); tb[5] = rng.random_scalar(); tb[6] = rng.random_scalar();
    {   // T_1, T_3, T_4, T_5, T_6 = t_k * B + tau_k * B_blinding
        uint8_t vv[5 * 32], rr[5 * 32], out[5 * 32];
        const int idx[5] = {1, 3, 4, 5, 6};
        for (int k = 0; k < 5; k++) { t[idx[k]].to_bytes(vv + 32 * k); tb[idx[k]].to_bytes(rr + 32 * k); }
        pedersen_commit(5, vv, rr, out);
        static const char *labels[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
        for (int k = 0; k < 5; k++) T.append_point(labels[k], out + 32 * k);
        proof.insert(proof.end(), out, out + 160);
    }
    const Scalar u_ch = T.challenge_scalar("u"), x = T.challenge_scalar("x");
    for (uint64_t j = 0; j < m; j++) tb[2] += from_scm(h_wV[j]) * v_blinding[j];
    Scalar tx, txb;
    for (int k = 6; k >= 1; k--) { tx = (tx + t[k]) * x; txb = (txb + tb[k]) * x; }
    const Scalar eb = x * (ib + x * (ob + x * sb));
    T.append_scalar("t_x", tx); T.append_scalar("t_x_blinding", txb); T.append_scalar("e_blinding", eb);
    { uint8_t b[96]; tx.to_bytes(b); txb.to_bytes(b + 32); eb.to_bytes(b + 64); proof.insert(proof.end(), b, b + 96); }
    const Scalar w = T.challenge_scalar("w");

    I.lv.ensure(N * sizeof(scm)); I.rv.ensure(N * sizeof(scm));
    BPG_LAUNCH(I, k_poly_eval, dim3(cdiv(N, 256)), dim3(256), c->aL.as<scm>(), c->aR.as<scm>(), c->aO.as<scm>(), sL, sR, wL, wR, wO,
                       ypow_p, I.yinvpow.as<scm>(), to_scm(x), I.lv.as<scm>(), I.rv.as<scm>(), (uint32_t)n, (uint32_t)N);
    HIPCHK(hipGetLastError());
    lap(tm ? &tm->poly : nullptr);

    // ---- inner-product argument
    I.inner_product(T, proof, n, N, yinv, u_ch, w, Gtab, Htab, Bn, tm, t0);
    if (tm) { tm->ipa = tm->ipa_msm + tm->ipa_fold; tm->total += now_ms() - t_begin; }
    return proof;
}

// ------------------------------------------------------------------------------------------------ verify
// Verifier::verify (dalek r1cs/verifier.rs; reference call site src/bin/verifier.rs:89-90): Fiat-Shamir replay on the host,
// then ONE multiscalar multiplication of 2N + m + 2 lgN + 13 terms through the same bucket-method kernels; accept iff it is the identity.
R1CSError Engine::verify(DeviceCircuit *c, Transcript &T, const uint8_t *V, const uint8_t *proof, size_t proof_len, const uint8_t seed[32], uint32_t flags) {
    HIPCHK(hipSetDevice(device_));
    Impl &I = *impl_;
    hipStream_t st = I.st;
    I.shared_now = I.shared_variants();
    const uint64_t n = c->n, m = c->m, q = c->q;
    uint64_t N = 1; while (N < n) N <<= 1;
    const uint32_t lgN = ceil_log2(N);
    const bool compact = flags & 1u, no_1phase = flags & 2u;
    const size_t need = (compact ? 1 + 11 * 32 : 14 * 32) + (2 * (size_t)lgN + 2) * 32;
    if (proof_len != need) return R1CSError::FormatError;
    if (gens_cap_ < N) return R1CSError::InvalidGeneratorsLength;
    if (lgN > 32) return R1CSError::FormatError;
    const uint8_t *in = proof;
    static const uint8_t ident[32] = {0};
    if (compact) { if (*in++ != 0) return R1CSError::FormatError; }
    const uint8_t *pA[6] = {in, in + 32, in + 64, ident, ident, ident}; in += 96;
    if (!compact) { pA[3] = in; pA[4] = in + 32; pA[5] = in + 64; in += 96; }
    const uint8_t *pT[5]; for (int k = 0; k < 5; k++) { pT[k] = in; in += 32; }
    Scalar sc5[5];                                   // t_x, t_x_blinding, e_blinding, a, b : must be canonical (R1CSProof::from_bytes)
    const uint8_t *ps[5] = {in, in + 32, in + 64, proof + proof_len - 64, proof + proof_len - 32};
    for (int k = 0; k < 5; k++) { std::memcpy(sc5[k].w, ps[k], 32); if (!sc5[k].is_canonical()) return R1CSError::FormatError; }
    in += 96;
    const uint8_t *pLR = in;
    const Scalar &tx = sc5[0], &txb = sc5[1], &eb = sc5[2], &ipa = sc5[3], &ipb = sc5[4];
    auto is_ident = [](const uint8_t *p) { return std::memcmp(p, ident, 32) == 0; };

    T.append_u64("m", m);
    if (is_ident(pA[0]) || is_ident(pA[1]) || is_ident(pA[2])) return R1CSError::VerificationError;      // validate_and_append_point
    T.append_point("A_I1", pA[0]); T.append_point("A_O1", pA[1]); T.append_point("S1", pA[2]);
    if (!no_1phase) T.r1cs_1phase_domain_sep();
    T.append_point("A_I2", pA[3]); T.append_point("A_O2", pA[4]); T.append_point("S2", pA[5]);
    const Scalar y = T.challenge_scalar("y"), z = T.challenge_scalar("z");
    static const char *tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int k = 0; k < 5; k++) { if (is_ident(pT[k])) return R1CSError::VerificationError; T.append_point(tl[k], pT[k]); }
    const Scalar u_ch = T.challenge_scalar("u"), x = T.challenge_scalar("x");
    T.append_scalar("t_x", tx); T.append_scalar("t_x_blinding", txb); T.append_scalar("e_blinding", eb);
    const Scalar w = T.challenge_scalar("w");
    T.innerproduct_domain_sep(N);
    std::vector<Scalar> uk(lgN), ukinv(lgN);
    bool lr_ident = false;
    for (uint32_t k = 0; k < lgN; k++) {
        const uint8_t *L = pLR + 64 * k, *R = L + 32;
        lr_ident |= is_ident(L) || is_ident(R);
        T.append_point("L", L); T.append_point("R", R);
        uk[k] = T.challenge_scalar("u"); ukinv[k] = uk[k];
    }
    if (lr_ident) return R1CSError::VerificationError;
    if (lgN) Scalar::batch_invert(ukinv);
    TranscriptRng rng = T.build_rng({}, seed);
    const Scalar r = rng.random_scalar();
    const Scalar yinv = y.invert();

    // ---- device side
    const uint32_t npts = (uint32_t)(6 + m + 5 + 2 * lgN);
    std::vector<uint8_t> hpts((size_t)npts * 32);
    {
        size_t o = 0;
        for (int k = 0; k < 6; k++) { std::memcpy(&hpts[o], pA[k], 32); o += 32; }
        if (m) { std::memcpy(&hpts[o], V, m * 32); o += m * 32; }
        for (int k = 0; k < 5; k++) { std::memcpy(&hpts[o], pT[k], 32); o += 32; }
        for (uint32_t k = 0; k < lgN; k++) { std::memcpy(&hpts[o], pLR + 64 * k, 32); o += 32; }
        for (uint32_t k = 0; k < lgN; k++) { std::memcpy(&hpts[o], pLR + 64 * k + 32, 32); o += 32; }
    }
    I.red_partial.ensure((size_t)4096 * sizeof(scm)); I.red_out.ensure(16 * sizeof(scm));      // [0,1024): delta partials, [1024,1536): w_c partials
    I.vfy_in.ensure((size_t)npts * 32); I.vfy_pts.ensure((size_t)npts * sizeof(ge_niels)); I.vfy_ok.ensure((size_t)npts * 4);
    I.vfy_sc.ensure((size_t)(npts + 2) * sizeof(scm)); I.vfy_ch.ensure(sizeof(IpaChallenges));
    HIPCHK(hipMemcpyAsync(I.vfy_in.p, hpts.data(), hpts.size(), hipMemcpyHostToDevice, st));
    BPG_LAUNCH(I, k_decompress, dim3(cdiv(npts, 64)), dim3(64), I.vfy_in.as<uint8_t>(), I.vfy_pts.as<ge_niels>(), I.vfy_ok.as<uint32_t>(), npts);
    I.yinvpow.ensure(N * sizeof(scm));
    // powers of z, the flattened weights and the s vector are read by k_verify_scalars and earlier kernels only: in the arena, ahead of the one MSM
    const size_t b_w = Impl::al256(c->ncols * sizeof(scm)), b_z = Impl::al256((q + 2) * sizeof(scm)), b_y = Impl::al256(N * sizeof(scm));
    I.arena.ensure(b_w + b_z + b_y);
    scm *const wAll_p = reinterpret_cast<scm *>(I.arena_at(0)), *const zpow_p = reinterpret_cast<scm *>(I.arena_at(b_w)), *const ypow_p = reinterpret_cast<scm *>(I.arena_at(b_w + b_z));
    {   // y^-i, z^j: one launch
        ExpTables E; std::memset(&E, 0, sizeof E);
        uint32_t k = 0, lgmax = 0;
        auto add = [&](const Scalar &base, scm *out, uint64_t count) {
            uint32_t lgT = ceil_log2(count); if (lgT > 16) lgT = 16;
            E.base[k] = to_scm(base); E.out[k] = out; E.count[k] = (uint32_t)count; E.lgT[k] = lgT; lgmax = std::max(lgmax, lgT); k++;
        };
        add(yinv, I.yinvpow.as<scm>(), N); add(z, zpow_p, q + 1);
        BPG_LAUNCH(I, k_exp_table, dim3(cdiv(1u << lgmax, 256), k), dim3(256), E);
    }
    if (c->ncols > 1)
        BPG_LAUNCH(I, k_flatten, dim3(cdiv(c->ncols - 1, 256)), dim3(256), c->col_ptr.as<uint64_t>(), c->ent_row.as<uint32_t>(), c->ent_coef.as<uint32_t>(),
                   c->coef.as<scm>(), zpow_p, wAll_p, (uint32_t)(c->ncols - 1), (uint32_t)(3 * n));
    scm *wL = wAll_p, *wR = wL + n, *wO = wR + n, *wV = wO + n;      // wV[m] = w_c
    {   // w_c: grid-wide reduction over the constant-term entries
        const uint64_t e0 = c->const_begin, e1 = c->nnz;
        const uint32_t cb = std::max<uint32_t>(1, std::min<uint32_t>(cdiv(e1 - e0, 256), 512));
        BPG_LAUNCH(I, k_flatten_const, dim3(cb), dim3(256), c->ent_row.as<uint32_t>(), c->ent_coef.as<uint32_t>(), c->coef.as<scm>(), zpow_p, e0, e1,
                   I.red_partial.as<scm>() + 1024);
        BPG_LAUNCH(I, k_reduce_partials, dim3(1), dim3(256), I.red_partial.as<scm>() + 1024, cb, 1u, wV + m);
    }
    {
        I.h_small.ensure(1 << 16);
        IpaChallenges *hc = reinterpret_cast<IpaChallenges *>(I.h_small.as<uint8_t>() + 8192);
        for (uint32_t k = 0; k < lgN; k++) { hc->u[k] = to_scm(uk[k]); hc->uinv[k] = to_scm(ukinv[k]); }
        HIPCHK(hipMemcpyAsync(I.vfy_ch.p, hc, sizeof(IpaChallenges), hipMemcpyHostToDevice, st));
    }
    I.lv.ensure(N * sizeof(scm)); I.rv.ensure(N * sizeof(scm));
    scm *svec = ypow_p, *gsc = I.lv.as<scm>(), *hsc = I.rv.as<scm>();
    BPG_LAUNCH(I, k_ipa_s, dim3(cdiv(N, 256)), dim3(256), I.vfy_ch.as<IpaChallenges>(), svec, lgN, (uint32_t)N);
    const uint32_t blocks = std::min<uint32_t>(cdiv(N, 256), 1024);
    BPG_LAUNCH(I, k_verify_scalars, dim3(blocks), dim3(256), wL, wR, wO, I.yinvpow.as<scm>(), svec, to_scm(x), to_scm(ipa), to_scm(ipb), to_scm(u_ch),
               gsc, hsc, I.red_partial.as<scm>(), (uint32_t)n, (uint32_t)N);
    BPG_LAUNCH(I, k_reduce_partials, dim3(1), dim3(256), I.red_partial.as<scm>(), blocks, 1u, I.red_out.as<scm>());
    HIPCHK(hipGetLastError());
    scm h_delta; std::vector<scm> h_wV(m + 1); std::vector<uint32_t> h_ok(npts);
    HIPCHK(hipMemcpyAsync(&h_delta, I.red_out.p, sizeof(scm), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_wV.data(), wV, (m + 1) * sizeof(scm), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_ok.data(), I.vfy_ok.p, npts * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t k = 0; k < npts; k++) if (!h_ok[k]) return R1CSError::VerificationError;      // optional_multiscalar_mul: a point failed to decompress
    const Scalar delta = from_scm(h_delta), wc = from_scm(h_wV[m]);
    const Scalar xx = x * x, rxx = r * xx, xxx = x * xx;
    std::vector<scm> hs(npts + 2);
    {
        size_t o = 0;
        hs[o++] = to_scm(x); hs[o++] = to_scm(xx); hs[o++] = to_scm(xxx);
        hs[o++] = to_scm(u_ch * x); hs[o++] = to_scm(u_ch * xx); hs[o++] = to_scm(u_ch * xxx);
        for (uint64_t j = 0; j < m; j++) hs[o++] = to_scm(from_scm(h_wV[j]) * rxx);
        hs[o++] = to_scm(r * x); hs[o++] = to_scm(rxx * x); hs[o++] = to_scm(rxx * xx); hs[o++] = to_scm(rxx * xxx); hs[o++] = to_scm(rxx * xx * xx);
        for (uint32_t k = 0; k < lgN; k++) hs[o++] = to_scm(uk[k] * uk[k]);
        for (uint32_t k = 0; k < lgN; k++) hs[o++] = to_scm(ukinv[k] * ukinv[k]);
        hs[o++] = to_scm(w * (tx - ipa * ipb) + r * (xx * (wc + delta) - tx));      // B
        hs[o++] = to_scm(-eb - r * txb);                                              // B_blinding
    }
    HIPCHK(hipMemcpyAsync(I.vfy_sc.p, hs.data(), hs.size() * sizeof(scm), hipMemcpyHostToDevice, st));
    Impl::MsmTicket tk;
    {
        MsmSegs S = seg_new();
        seg_push(S, gsc, I.gens.as<ge_niels>(), (uint32_t)N, 0);
        seg_push(S, hsc, I.gens.as<ge_niels>() + gens_cap_, (uint32_t)N, 0);
        seg_push(S, I.vfy_sc.as<scm>(), I.vfy_pts.as<ge_niels>(), npts, 0);
        seg_push(S, I.vfy_sc.as<scm>() + npts, I.bases.as<ge_niels>(), 2, 0);
        tk = I.msm(S, 1);
    }
    HIPCHK(hipStreamSynchronize(st));
    uint8_t out[32];
    h51::pt_compress(out, I.msm_points(tk)[0]);
    return is_ident(out) ? R1CSError::None : R1CSError::VerificationError;
}

}  // namespace bpg
