// Chain threads: the serial TranscriptRng draws of Prover::prove (2n + 3 leading draws of 64 bytes, one Keccak-f each; reference call site
// src/bin/prover.rs:93 -> bulletproofs r1cs/prover.rs::prove -> merlin TranscriptRng) produced AHEAD of the proof on host threads and handed to
// the device block by block.  Host-only code (no HIP in this header): how a block reaches the device is the `upload` callback of the stream, so
// the worker loops and the lock-free publication protocol below are the SAME code in the product (engine.hip: hipMemcpyAsync + event per block)
// and in the sanitizer builds of tests/hostcheck (a memcpy into a stand-in slab under -fsanitize=thread).
//
// Publication protocol of one BlindStream (one producer = the chain thread that drew it, one consumer = the prove() that adopts it):
//   raw[0 .. 64 * produced)            bytes of the draws; written by the producer BEFORE produced.store(release); read after produced.load(acquire)
//   snaps[k], k <= produced / SNAP     generator state before draw k * SNAP; same ordering through `produced`
//   err                                first upload error, stored (release) BEFORE uploaded_blocks is published
//   uploaded_blocks                    blocks of UP draws handed to the device; the consumer reads err after uploaded_blocks.load(acquire)
//   stop                               consumer -> producer: stop at the next snapshot;  finished: producer -> consumer: the slab is no longer touched
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>
#include "merlin.hpp"
#include "scalar.hpp"

namespace bpg {

struct BlindStream {
    static constexpr uint64_t SNAP = 4096;
    std::atomic<uint64_t> produced{0};
    std::atomic<bool> stop{false}, finished{false};
    std::atomic<int> cpu{-1};
    uint8_t state[203]; uint8_t seed[32]; std::vector<Scalar> vb;
    Scalar first[3];
    std::vector<TranscriptRng> snaps;
    uint64_t max_draws = 0;
    int slot = 0; uint8_t *raw = nullptr;
    // the worker hands over what it has drawn, block by block (UP draws = 4 MB): upload(b, from, to, k) sends draws [from, to) = block k on its
    // way (the product: an asynchronous copy on the slab's own copy stream + the event of block k) and returns 0 or an error code
    static constexpr uint64_t UP = 65536;
    std::function<int(BlindStream &, uint64_t, uint64_t, uint64_t)> upload;
    int device = 0; void *copy_st = nullptr; uint8_t *d_raw = nullptr; void *ev = nullptr;      // the uploader's handles (opaque here)
    std::atomic<uint64_t> uploaded_blocks{0};
    // first error of this stream's uploads (0 = none), stored BEFORE uploaded_blocks is published: the device slab is reused from proof to
    // proof, so a block that was not copied would hand prove() the previous proof's draws - prove() checks and refuses
    std::atomic<int> err{0};
    int inject_fail = 0;                                        // test hooks: 1 = bpg_test_fail_next_upload (an upload reports an error), 2 = bpg_test_drop_next_upload (the copy is skipped, no error)
    void note(int e) { if (e != 0) { int want = 0; err.compare_exchange_strong(want, e, std::memory_order_release); } }
    // draws [up, to) -> device; returns the new `up`
    uint64_t hand_over(uint64_t up, uint64_t to) {
        if (to <= up) return up;
        const uint64_t k = up / UP;
        note(upload ? upload(*this, up, to, k) : 0);
        if (inject_fail == 1) note(999);                        // hipErrorUnknown
        uploaded_blocks.store(k + 1, std::memory_order_release);
        return to;
    }
};

struct ChainWorker {                                     // chain threads: a context's own (one by default) or a pool shared by several contexts
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv;
    std::deque<std::shared_ptr<BlindStream>> pending; bool quit = false;
    uint32_t lanes = 1;                                  // streams one thread of a context's own worker draws in lockstep (merlin.hpp strobe_rng_bulk64_x8); set before the threads start
    bool pool = false;                                   // a ChainPool shared by several contexts (its threads carry their own lane counts and outlive the contexts)
    void stop() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); for (std::thread &t : th) if (t.joinable()) t.join(); th.clear(); }
    void push(const std::shared_ptr<BlindStream> &b) { { std::lock_guard<std::mutex> lk(mu); pending.push_back(b); } cv.notify_one(); }
    // One thread, up to eight streams in lockstep: the sponges of eight proofs in the eight 64-bit lanes of ZMM registers cost a Zen 5 core
    // 193 ns per draw of all eight against 152 ns for one alone (tools/diag/chain_lanes.py): a chain still takes 0.3 - 0.4 s, a core's chain
    // THROUGHPUT goes up sixfold.  Streams join at 4,096-draw boundaries as they are queued and leave when they are complete or stopped.
    static void run_lanes(ChainWorker *w, uint32_t lanes) {
        struct Lane { std::shared_ptr<BlindStream> b; TranscriptRng rng; uint64_t pos, up; };
        std::vector<Lane> act;
        for (;;) {
            {   // take what is queued; wait only when there is nothing to draw
                std::unique_lock<std::mutex> lk(w->mu);
                if (act.empty()) { w->cv.wait(lk, [&] { return w->quit || !w->pending.empty(); }); if (w->pending.empty()) return; }
                while (act.size() < lanes && !w->pending.empty()) {
                    std::shared_ptr<BlindStream> b = w->pending.front(); w->pending.pop_front();
                    b->cpu.store(sched_getcpu(), std::memory_order_relaxed);
                    act.push_back(Lane{b, b->snaps[0], 0, 0});
                }
            }
            for (size_t k = 0; k < act.size();) {            // publish; retire what is complete or stopped
                Lane &L = act[k]; BlindStream &b = *L.b;
                if (L.pos) b.snaps[L.pos / BlindStream::SNAP] = L.rng;
                b.produced.store(L.pos, std::memory_order_release);
                if (L.pos >= b.max_draws || b.stop.load(std::memory_order_relaxed)) {
                    L.up = b.hand_over(L.up, L.pos);
                    b.finished.store(true, std::memory_order_release);
                    act.erase(act.begin() + (ptrdiff_t)k);
                } else k++;
            }
            if (act.empty()) continue;
            TranscriptRng *r[8]; uint8_t *dst[8];
            for (size_t k = 0; k < act.size(); k++) { r[k] = &act[k].rng; dst[k] = act[k].b->raw + 64 * act[k].pos; }
            TranscriptRng::fill_draws64_multi(r, dst, (uint32_t)act.size(), BlindStream::SNAP);
            for (Lane &L : act) { L.pos += BlindStream::SNAP; if (L.pos % BlindStream::UP == 0) L.up = L.b->hand_over(L.up, L.pos); }
        }
    }
    static void run(ChainWorker *w, uint32_t lanes) {
        if (lanes > 1) { run_lanes(w, lanes); return; }
        for (;;) {
            std::shared_ptr<BlindStream> b;
            { std::unique_lock<std::mutex> lk(w->mu); w->cv.wait(lk, [&] { return w->quit || !w->pending.empty(); }); if (w->pending.empty()) return; b = w->pending.front(); w->pending.pop_front(); }
            b->cpu.store(sched_getcpu(), std::memory_order_relaxed);
            TranscriptRng rng = b->snaps[0];
            uint64_t pos = 0, up = 0;                                               // up: draws handed to the device
            for (;;) {
                if (pos) b->snaps[pos / BlindStream::SNAP] = rng;                   // state before draw pos (snaps[0] was set by the caller)
                b->produced.store(pos, std::memory_order_release);                  // draws [0, pos) and snapshots up to pos are published
                if (pos >= b->max_draws || b->stop.load(std::memory_order_relaxed)) break;
                rng.fill_draws64(b->raw + 64 * pos, BlindStream::SNAP);
                pos += BlindStream::SNAP;
                if (pos % BlindStream::UP == 0) up = b->hand_over(up, pos);
            }
            up = b->hand_over(up, pos);                                             // the last, shorter block (or what was drawn before a stop)
            b->finished.store(true, std::memory_order_release);
        }
    }
};

}  // namespace bpg
