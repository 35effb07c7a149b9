// SANITIZER DRIVER (test infrastructure, not product): the host-side concurrency of the prove path - the chain threads and their lock-free
// publication protocol (csrc/host/chain.hpp), the Merlin transcript and the lockstep Keccak of TranscriptRng (csrc/host/merlin.hpp) - compiled
// for the host alone with -fsanitize=thread or -fsanitize=address,undefined (tests/hostcheck/Makefile).  The product code is the SAME headers;
// the only stand-in is the uploader: where engine.hip issues hipMemcpyAsync + an event per block, this driver copies the block into a host
// "device slab" and lets a consumer thread read it back the way Engine::prove does.
//   san_chain [rounds=3] [streams=12] [draws=40000]
// Exit code 0 and the line "san_chain ok" when every stream's bytes equal a single-threaded redraw of the same generator.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include "../../bulletproofs_gadgets_amd/csrc/host/chain.hpp"
using namespace bpg;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); failures++; } } while (0)

struct Slab { std::vector<uint8_t> host, dev; std::atomic<uint64_t> copied{0}; };

static std::shared_ptr<BlindStream> make_stream(Transcript &T, const uint8_t seed[32], uint64_t draws, Slab &slab, bool fail_upload) {
    auto b = std::make_shared<BlindStream>();
    T.export_state(b->state); std::memcpy(b->seed, seed, 32);
    TranscriptRng rng = T.build_rng({}, seed);
    for (int k = 0; k < 3; k++) b->first[k] = rng.random_scalar();
    b->max_draws = ((draws + BlindStream::SNAP - 1) / BlindStream::SNAP) * BlindStream::SNAP;
    slab.host.assign(b->max_draws * 64, 0xee); slab.dev.assign(b->max_draws * 64, 0xdd);
    b->raw = slab.host.data(); b->d_raw = slab.dev.data(); b->ev = &slab;
    b->snaps.assign(b->max_draws / BlindStream::SNAP + 1, rng);
    b->inject_fail = fail_upload;
    b->upload = [](BlindStream &bs, uint64_t from, uint64_t to, uint64_t) -> int {       // the product: hipMemcpyAsync + hipEventRecord
        std::memcpy(bs.d_raw + 64 * from, bs.raw + 64 * from, (to - from) * 64);
        static_cast<Slab *>(bs.ev)->copied.store(to, std::memory_order_release);
        return 0;
    };
    return b;
}

// what Engine::prove does with an adopted stream: block by block wait for uploaded_blocks, check err, read the block; then take the generator back
static void consume(const std::shared_ptr<BlindStream> &b, uint64_t need, std::vector<uint8_t> &got, TranscriptRng &rng_after, bool &refused) {
    refused = false; got.clear();
    for (uint64_t i = 0; i < need; i += BlindStream::UP) {
        const uint64_t k = i / BlindStream::UP, cnt = std::min<uint64_t>(BlindStream::UP, need - i);
        while (b->uploaded_blocks.load(std::memory_order_acquire) <= k) {
            if (b->finished.load(std::memory_order_acquire) && b->uploaded_blocks.load(std::memory_order_acquire) <= k) break;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (b->err.load(std::memory_order_acquire)) { refused = true; b->stop.store(true, std::memory_order_relaxed); return; }
        if (b->uploaded_blocks.load(std::memory_order_acquire) <= k) { refused = true; return; }
        (void)static_cast<Slab *>(b->ev)->copied.load(std::memory_order_acquire);       // the product waits for the block's event on its stream
        got.insert(got.end(), b->d_raw + 64 * i, b->d_raw + 64 * (i + cnt));
    }
    const uint64_t K = need / BlindStream::SNAP, rem = need - K * BlindStream::SNAP;
    while (b->produced.load(std::memory_order_acquire) < K * BlindStream::SNAP) std::this_thread::sleep_for(std::chrono::microseconds(20));
    rng_after = b->snaps[K];
    if (rem) { std::vector<uint8_t> skip(rem * 64); rng_after.fill_draws64(skip.data(), rem); }
    b->stop.store(true, std::memory_order_relaxed);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    const int nstreams = argc > 2 ? std::atoi(argv[2]) : 12;
    const uint64_t draws = argc > 3 ? (uint64_t)std::atoll(argv[3]) : 40000;
    (void)keccak_impl();
    // ---- Merlin's published vector and the equality of the bulk / lockstep generators with the generic STROBE path
    {
        Transcript t(reinterpret_cast<const uint8_t *>("test protocol"), 13);
        t.append_message("some label", reinterpret_cast<const uint8_t *>("some data"), 9);
        uint8_t c[32]; t.challenge_bytes("challenge", c, 32);
        static const uint8_t want[32] = {0xd5, 0xa2, 0x19, 0x72, 0xd0, 0xd5, 0xfe, 0x32, 0x0c, 0x0d, 0x26, 0x3f, 0xac, 0x7f, 0xff, 0xb8, 0x14, 0x5a, 0xa6, 0x40, 0xaf, 0x6e, 0x9b, 0xca, 0x17, 0x7c, 0x03, 0xc7, 0xef, 0xcf, 0x06, 0x15};      // merlin's published test vector
        CHECK(std::memcmp(c, want, 32) == 0);
    }
    for (int r = 0; r < rounds; r++) {
        // a pool like the headline's: single-lane threads and lockstep threads serving the same queue, streams of several "contexts"
        ChainWorker pool; pool.pool = true;
        const uint32_t lanes_per_thread[5] = {1, 1, 8, 3, 1};
        for (uint32_t l : lanes_per_thread) pool.th.emplace_back(ChainWorker::run, &pool, l);
        std::vector<Slab> slabs(nstreams);
        std::vector<std::shared_ptr<BlindStream>> streams;
        std::vector<Transcript> Ts;
        std::vector<std::array<uint8_t, 32>> seeds(nstreams);
        for (int k = 0; k < nstreams; k++) {
            Transcript T(reinterpret_cast<const uint8_t *>("san"), 3);
            T.append_u64("k", (uint64_t)(r * 1000 + k));
            Ts.push_back(T);
            for (int j = 0; j < 32; j++) seeds[k][j] = (uint8_t)(k * 7 + j + r);
        }
        for (int k = 0; k < nstreams; k++) streams.push_back(make_stream(Ts[k], seeds[k].data(), draws + (uint64_t)k * 1111, slabs[k], k == 5));
        std::vector<std::thread> consumers;
        std::vector<std::vector<uint8_t>> got(nstreams); std::vector<TranscriptRng> after(nstreams, streams[0]->snaps[0]); std::vector<char> refused(nstreams, 0);
        for (int k = 0; k < nstreams; k++) {
            pool.push(streams[k]);
            // some consumers need fewer draws than the stream offers (the circuit was smaller than the estimate), one cancels early
            const uint64_t need = (k % 3 == 1) ? draws / 2 + 17 : draws + (uint64_t)k * 1111;
            consumers.emplace_back([&, k, need] { bool rf = false; consume(streams[k], need, got[k], after[k], rf); refused[k] = rf; });
        }
        for (std::thread &t : consumers) t.join();
        for (int k = 0; k < nstreams; k++) while (!streams[k]->finished.load(std::memory_order_acquire)) std::this_thread::yield();
        pool.stop();
        for (int k = 0; k < nstreams; k++) {
            if (k == 5) { CHECK(refused[k]); continue; }                         // the injected upload failure must be seen before any byte is read
            CHECK(!refused[k]);
            TranscriptRng ref = Ts[k].build_rng({}, seeds[k].data());
            for (int j = 0; j < 3; j++) (void)ref.random_scalar();
            const uint64_t need = got[k].size() / 64;
            std::vector<uint8_t> want(need * 64);
            for (uint64_t i = 0; i < need; i++) ref.fill_bytes(&want[64 * i], 64);       // the generic path, one draw at a time
            CHECK(want == got[k]);
            uint8_t a[64], b[64]; ref.fill_bytes(a, 64); after[k].fill_bytes(b, 64);       // the generator handed back continues the same stream
            CHECK(std::memcmp(a, b, 64) == 0);
        }
    }
    if (failures) { std::fprintf(stderr, "san_chain: %d failures\n", failures); return 1; }
    std::puts("san_chain ok");
    return 0;
}
