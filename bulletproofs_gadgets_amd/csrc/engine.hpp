// HIP engine behind the C ABI (include/bpg.h): owns the device, the generator tables resident in HBM, the MSM
// workspace and the prove pipeline.  There is NO CPU execution path: construction throws when no GPU is present.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "host/scalar.hpp"
#include "host/merlin.hpp"
#include "host/r1cs.hpp"

namespace bpg {

struct DeviceError : std::runtime_error { explicit DeviceError(const std::string &m) : std::runtime_error(m) {} };

struct ProveTimings {       // milliseconds, host wall clock around each phase (stream synchronised at phase ends)
    double rng_host = 0, msm_aiao = 0, msm_s = 0, poly = 0, ipa = 0, total = 0;
    double ipa_msm = 0, ipa_fold = 0, ipa_sync = 0;
    uint32_t shared_variants = 0;   // 1: this proof took the shared-device kernel variants (decided once, when prove() was entered)
};

struct DeviceCircuit;       // HBM-resident flattened R1CS instance

// What a host chooses at context creation (include/bpg.h bpg_config).  A field left at its "unset" value falls back to the environment
// variable named beside it, then to the profile's default.
struct EngineConfig {
    uint32_t profile = 0;           // 0 unset (BPG_PROFILE, else one-shot), 1 one-shot, 2 serving
    double table_budget_gb = 0;     // cumulative HBM the precomputed generator multiples of this device may take (BPG_TABLE_GB); 0 unset
    uint32_t chain_workers = 0;     // 0 unset (BPG_CHAIN_WORKERS, else 1)
    uint32_t chain_lanes = 0;       // 0 unset (BPG_CHAIN_LANES, else 1)
    int32_t blocking_sync = -1;     // -1 unset (BPG_SYNC_BLOCKING=1/0, else spin), 1 blocking waits, 0 (or 2) spin waits
    std::string gens_cache_dir;     // empty unset (BPG_GENS_CACHE_DIR, else no cache)
};

// Chain threads shared by several contexts (include/bpg.h bpg_chain_pool_*): thread k draws up to lanes_per_thread[k] queued blinding streams in
// lockstep, whichever attached context queued them.  Must outlive the contexts attached to it.
class ChainPool {
public:
    explicit ChainPool(const std::vector<uint32_t> &lanes_per_thread);
    ~ChainPool();
    ChainPool(const ChainPool &) = delete;
    ChainPool &operator=(const ChainPool &) = delete;
    uint32_t capacity() const { return capacity_; }          // chains the pool draws side by side
    struct Impl;
private:
    friend class Engine;
    Impl *impl_;
    uint32_t capacity_ = 0;
};

class Engine {
public:
    explicit Engine(int device, const EngineConfig &cfg = EngineConfig());
    ~Engine();
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;

    int device() const { return device_; }
    static int device_count();      // AMD GPUs visible to the process (0 when there is none)
    // BulletproofGens::new(capacity, 1): derive (or extend) the G/H tables in HBM; capacity must be a power of two
    void gens_ensure(uint64_t capacity);
    uint64_t gens_capacity() const { return gens_cap_; }
    void gens_export(uint64_t first, uint64_t count, uint8_t *G_out, uint8_t *H_out);
    void pedersen_bases(uint8_t B[32], uint8_t B_blinding[32]);
    // k Pedersen commitments v_i*B + r_i*B_blinding (compressed). v may be unreduced (< 2^255); r any 256-bit value.
    void pedersen_commit(size_t k, const uint8_t *v, const uint8_t *blind, uint8_t *out);
    // multiscalar multiplication over generator-table slices, for tests: sum s_i * G[first+i] + t_i * H[first+i]
    void msm_gens(uint64_t first, uint64_t count, const uint8_t *s, const uint8_t *t, uint8_t out[32]);

    DeviceCircuit *upload(const FlatView &c);
    DeviceCircuit *upload(const FlatCircuit &c) { return upload(FlatView(c)); }
    void free_circuit(DeviceCircuit *c);
    // Prover::prove on a resident circuit. transcript: state after Prover::new + every "V" append (updated in place).
    std::vector<uint8_t> prove(DeviceCircuit *c, Transcript &transcript, const std::vector<Scalar> &v_blinding,
                               const uint8_t rng_seed[32], uint32_t flags, ProveTimings *timings = nullptr);
    // Speculative start of prove()'s TranscriptRng on a host thread (extension; include/bpg.h bpg_blinding_begin): the draws depend on the
    // transcript after the last commitment, v_blinding and the seed only.  prove() uses the stream iff all three still match, else discards it.
    void blinding_begin(const Transcript &after_commitments, const std::vector<Scalar> &v_blinding, const uint8_t seed[32], uint64_t max_multipliers);
    void blinding_cancel();
    // threads of the context's chain worker: that many queued streams are drawn side by side, workers + 1 may be alive (default 1)
    void set_chain_workers(uint32_t n);
    // draw this context's blinding streams on a shared pool instead of its own worker (nullptr: back to its own); max_streams of them may be alive
    void attach_chain_pool(ChainPool *pool, uint32_t max_streams);
    void set_chain_lanes(uint32_t n);        // streams each chain thread draws in lockstep (1..8; eight sponges in the lanes of ZMM registers)
    void test_fail_next_upload();   // test hook (bpg_test_fail_next_upload)
    void test_drop_next_upload();   // test hook (bpg_test_drop_next_upload): the copies of the next stream are skipped without an error
    uint64_t table_bytes() const;   // precomputed generator multiples held on this device by the process
    bool last_shared_variants() const;   // did the last prove()/verify() on this context take the shared-device kernel variants
    int chain_cpu() const;      // host core the chain worker last drew a stream on (-1: none yet); diagnostics for bench.py
    void test_fe_ops(int op, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out);   // unit-test hook (k_test_fe)
    // Verifier::verify on a resident (assignment-free) circuit. transcript: state after Verifier::new + every "V" append.
    R1CSError verify(DeviceCircuit *c, Transcript &transcript, const uint8_t *V, const uint8_t *proof, size_t proof_len,
                     const uint8_t seed[32], uint32_t flags);
    void synchronize();
    // HIP-event profile on the engine's own stream: mode 0 off, 1 = dominant kernel (k_fold_points) only, 2 = all kernels
    void profile_set(int mode);
    std::string profile_report();          // JSON: {kernel: {count, total_ms, alg_bytes, device_bytes, field_mults}}
    double bench_fe_mul(uint32_t iters);   // measured field-multiplication throughput (integer-VALU roofline)
    void *stream_handle() const { return stream_; }
    // names of the kernels launched by the last prove(), with counts (diagnostics for bench.py)
    struct Impl;
private:
    int device_;
    void *stream_ = nullptr;
    uint64_t gens_cap_ = 0;
    Impl *impl_ = nullptr;
    void init_device();             // second half of the constructor: everything that touches the GPU
};

}  // namespace bpg
