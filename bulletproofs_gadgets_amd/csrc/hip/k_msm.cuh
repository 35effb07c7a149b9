// bucket-method multiscalar multiplication - part of kernels.cuh (included from there, in this order; see its header for the kernel map and the data layout)
#pragma once

namespace bpg {

// ------------------------------------------------------------------------------------------------ multiscalar multiplication
// Window j of W covers bits [off(j), off(j+1)) with off(j) = j*254/W (MsmPlan::off, filled by the host): near-equal widths, so that the top window keeps
// (almost) a full width of entropy - with fixed c-bit windows the last one holds only 253 mod c bits and a handful of
// buckets would receive every term.  Signed digits: digit j in (-2^(wd-1), 2^(wd-1)], wd = width of window j.
__device__ __forceinline__ uint32_t msm_find_seg(const MsmSegs &S, uint32_t g) {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 1; k < BPG_MAX_SEGS; k++) if (k < S.nseg && g >= S.start[k]) s = k;
    return s;
}

// Sorting the (term, window) entries by bucket without global atomics (device-scope atomics on MI355X resolve beyond the per-XCD L2 and
// were the slowest part of the MSM): LDS histograms and LDS-staged runs only, see "two-level sort" below.
// key = (msm * W + window) * nb + (|digit| - 1); entry = sign << 31 | seg (4 bits) << 27 | index-in-segment.
// Signed digits come from a carry-free recoding: with bias = sum_j 2^(off(j)+wd(j)-1) added to the scalar once, digit j is
// field_j(s + bias) - 2^(wd(j)-1), in [-2^(wd-1), 2^(wd-1)).
struct MsmPlan {
    uint32_t nmsm, W, nb, lgTile, tmax;
    uint8_t off[132];            // off[j] = j * 254 / W, the first bit of window j, j <= W (the host fills it: the kernel divided by W twice per digit)
    uint32_t fb, CB;             // two-level sort: a bucket index splits into CB coarse bins x 2^fb fine slots (nb = CB << fb)
    uint32_t term_start[5];      // first global term of MSM m (term_start[nmsm] = total)
    uint32_t tile_start[5];      // first tile of MSM m
    uint32_t bias[8];
};
__device__ __forceinline__ int32_t msm_digit_biased(const uint32_t w[8], const MsmPlan &P, uint32_t win) {
    const uint32_t off = P.off[win], wd = P.off[win + 1] - off, wi = off >> 5, sh = off & 31;
    const uint64_t two = (uint64_t)w[wi] | ((uint64_t)(wi + 1 < 8 ? w[wi + 1] : 0u) << 32);
    return (int32_t)((uint32_t)(two >> sh) & ((1u << wd) - 1u)) - (int32_t)(1u << (wd - 1));
}
__device__ __forceinline__ void msm_biased_words(uint32_t w[8], const scm &sc, const MsmPlan &P) {
    sc_to_words(w, sc);
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint64_t t = (uint64_t)w[k] + P.bias[k] + carry; w[k] = (uint32_t)t; carry = t >> 32; }
}
// ------------------------------------------------------------------------------------------------ two-level sort
// A one-level counting sort (round 1; removed in round 4) writes every entry with its own 4-byte store at a position nobody else of the block
// writes near (a tile holds about one entry per bucket): 8x write amplification, and a histogram matrix of tiles x buckets that is read and
// written three times.  Two levels instead:
//   k_msm_digits    block (tile of 2^lgTile terms): every signed digit of every term once: dig[window][term] = digit + 2^15  (2 bytes, coalesced; digits of
//                   windows up to 16 bits wide lie in [-2^15, 2^15), 2^15 = no entry), and an LDS histogram per window over the CB coarse bins (top bits of
//                   the bucket index) while the digits are in registers
//   (scan)          exclusive scan of counts1[(msm*W + window)*CB + bin][tile] in that order: where each (bin, tile) run starts
//   k_msm_scatter1  block (tile, window): orders its entries by coarse bin in LDS and copies the runs out - consecutive lanes write
//                   consecutive addresses; entry = neg << 31 | fine << (31 - fb) | segment << (27 - fb) | index in segment
//   k_msm_sort2     block per coarse bin: counting sort by the fb fine bits inside the bin's own range (LDS counters, the range is written
//                   by this block only, so its lines are completed in the XCD's L2), emits starts[] per bucket
// The result is exactly what the one-level sort produces (entries grouped by bucket, starts[]); the sweep is unchanged.
// LDS counter increment that does not serialise when every lane of the wave holds the same key (identical scalars in consecutive terms:
// padding generators, repeated witness values, range-proof bits): one atomic per wave then.  Returns the lane's slot.
__device__ __forceinline__ uint32_t msm_lds_take(uint32_t *ctr, uint32_t key, bool active) {
    const uint64_t act = __ballot(active);
    if (act == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u, lead = (uint32_t)__ffsll((unsigned long long)act) - 1u;
    const uint32_t k0 = __shfl(key, lead, 64);
    if (__ballot(active && key == k0) == act) {
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(&ctr[k0], (uint32_t)__popcll(act));
        base = __shfl(base, lead, 64);
        return base + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
    }
    return active ? atomicAdd(&ctr[key], 1u) : 0u;
}
// stored digit -> |digit| (0 = no entry) and sign
__device__ __forceinline__ uint32_t msm_dig_mag(uint32_t x) { return x >= 32768u ? x - 32768u : 32768u - x; }
__device__ __forceinline__ uint32_t msm_dig_neg(uint32_t x) { return x < 32768u ? 1u : 0u; }
__device__ __forceinline__ void msm_tile_of(const MsmPlan &P, uint32_t T, uint32_t &m, uint32_t &t, uint32_t &g0, uint32_t &g1) {
    m = 0;
#pragma unroll
    for (uint32_t k = 1; k < 4; k++) if (k < P.nmsm && T >= P.tile_start[k]) m = k;
    t = T - P.tile_start[m];
    g0 = P.term_start[m] + (t << P.lgTile);
    g1 = min(g0 + (1u << P.lgTile), P.term_start[m + 1]);
}
#define MSM_CB_MAX 512
#define MSM_DIG_PER 8                    // terms per thread of k_msm_digits: a tile of 2^12 terms on 512 threads
// Digits and coarse histograms in one launch (rounds 2-4: k_msm_digits over the terms, then k_msm_count1 over (tile, window) re-reading the digits).  Block = one tile
// of 2^lgTile terms (tiles never span two sums): a thread converts its eight terms once, writes their W digits (coalesced per window) and counts each in the LDS
// histogram of its window - W x CB counters, dynamic LDS (17 x 128 for a 2^21-term sum).
__global__ void __launch_bounds__(512) k_msm_digits(MsmSegs S, MsmPlan P, uint32_t total, uint16_t *__restrict__ dig, uint32_t *__restrict__ counts1,
                                                    uint32_t *__restrict__ heavy_count, uint32_t *__restrict__ medium_count) {
    extern __shared__ uint32_t hist[];                       // [W][CB]
    if (blockIdx.x == 0 && threadIdx.x == 0) { *heavy_count = 0; *medium_count = 0; }     // lists of k_bucket_combine, filled later on this stream
    uint32_t m, t, g0, g1; msm_tile_of(P, blockIdx.x, m, t, g0, g1);
    if (g1 <= g0) return;                                    // a sum without terms has one empty tile (its counters were zeroed by the host); the whole block leaves
    const uint32_t nh = P.W * P.CB;
    for (uint32_t b = threadIdx.x; b < nh; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    // 512 threads, eight terms each (a tile is 2^12 terms).  The terms are converted first and kept in registers (8 x 8 words); then the WINDOWS are the outer loop:
    // a window's bit offset is the wave's (one scalar load per window - with the terms outside, every digit of every term waited for two of them, and a launch
    // of ten blocks took as long as one of five hundred: 75 us), its eight digits are stored (coalesced per window) and counted.  The counters are incremented
    // without their values coming back (no slot is taken here: k_msm_scatter1 takes them), so lanes with one key cost a wave 64 cycles, not a round trip each
    uint32_t w[MSM_DIG_PER][8];
    uint32_t livemask = 0, inmask = 0;
#pragma unroll
    for (uint32_t u = 0; u < MSM_DIG_PER; u++) {
        const uint32_t g = g0 + u * 512u + threadIdx.x;
        const bool in = g < g1;
        const uint32_t gg = in ? g : g0;
        const uint32_t s = msm_find_seg(S, gg), i = gg - S.start[s];
        const uint32_t *skip = S.skip[s];
        const bool live = in && !(skip && ((skip[i >> 5] >> (i & 31u)) & 1u));       // a skipped term was merged away (its scalar rides on another term of this sum): no entry in any window
        msm_biased_words(w[u], S.sc[s][i], P);
        if (in) inmask |= 1u << u;
        if (live) livemask |= 1u << u;
    }
    for (uint32_t j = 0; j < P.W; j++) {
        const uint32_t off = P.off[j], wd = P.off[j + 1] - off, wi = off >> 5, sh = off & 31u, half = 1u << (wd - 1u), mask = (1u << wd) - 1u;
        uint32_t *hj = hist + j * P.CB;
        uint16_t *dj = dig + (size_t)j * total + g0 + threadIdx.x;
#pragma unroll
        for (uint32_t u = 0; u < MSM_DIG_PER; u++) {
            uint32_t lo = w[u][0], hi = w[u][1];                                     // words wi, wi + 1 of the term (wi is the wave's: a chain of scalar compares, no indexed registers)
#pragma unroll
            for (uint32_t k = 1; k < 8; k++) { if (wi == k) { lo = w[u][k]; hi = k < 7 ? w[u][k + 1] : 0u; } }
            const uint64_t two = (uint64_t)lo | ((uint64_t)hi << 32);
            const uint32_t d = ((livemask >> u) & 1u) ? (((uint32_t)(two >> sh) & mask) - half + 32768u) : 32768u;
            if ((inmask >> u) & 1u) dj[u * 512u] = (uint16_t)d;
            const uint32_t mag = msm_dig_mag(d);
            if (mag) atomicAdd(&hj[(mag - 1u) >> P.fb], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nh; b += blockDim.x) {
        const uint32_t j = b / P.CB, bin = b - j * P.CB;
        counts1[((size_t)(m * P.W + j) * P.CB + bin) * P.tmax + t] = hist[b];
    }
}
#define MSM_TILE1_MAX 4096
#define MSM_TILE1_PER (MSM_TILE1_MAX / 256)
__global__ void __launch_bounds__(256) k_msm_scatter1(MsmSegs S, MsmPlan P, const uint16_t *__restrict__ dig, uint32_t total,
                                                      const uint32_t *__restrict__ starts1, uint32_t *__restrict__ entries1) {
    __shared__ uint32_t cnt[MSM_CB_MAX], lbase[MSM_CB_MAX], gbase[MSM_CB_MAX], staged[MSM_TILE1_MAX];
    __shared__ uint16_t sbin[MSM_TILE1_MAX];
    __shared__ uint32_t wsum[4];
    uint32_t m, t, g0, g1; msm_tile_of(P, blockIdx.x, m, t, g0, g1);
    const uint32_t win = blockIdx.y, mw = m * P.W + win;
    for (uint32_t b = threadIdx.x; b < P.CB; b += blockDim.x) { cnt[b] = 0; gbase[b] = starts1[((size_t)mw * P.CB + b) * P.tmax + t]; }
    __syncthreads();
    const uint16_t *D = dig + (size_t)win * total;
    // one pass of LDS atomics: the slot a term takes inside its coarse bin is kept (16 terms per thread, digit | slot << 16 in a register)
    uint32_t keep[MSM_TILE1_PER];
#pragma unroll
    for (uint32_t it = 0; it < MSM_TILE1_PER; it++) { const uint32_t g = g0 + it * 256u + threadIdx.x; keep[it] = g < g1 ? D[g] : 32768u; }   // all loads first
#pragma unroll
    for (uint32_t it = 0; it < MSM_TILE1_PER; it++) {
        const uint32_t d = keep[it], mag = msm_dig_mag(d);
        const uint32_t slot = msm_lds_take(cnt, mag ? (mag - 1u) >> P.fb : 0u, mag != 0u);
        keep[it] = d | (slot << 16);
    }
    __syncthreads();
    {   // exclusive prefix of cnt[0..CB) -> lbase; CB <= 512: two values per thread, wave scans, 4 wave totals
        const uint32_t a0 = 2 * threadIdx.x < P.CB ? cnt[2 * threadIdx.x] : 0u, a1 = 2 * threadIdx.x + 1 < P.CB ? cnt[2 * threadIdx.x + 1] : 0u;
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        uint32_t incl = a0 + a1;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t off = 0;
        for (uint32_t k = 0; k < wv; k++) off += wsum[k];
        const uint32_t excl = off + incl - (a0 + a1);
        if (2 * threadIdx.x < P.CB) lbase[2 * threadIdx.x] = excl;
        if (2 * threadIdx.x + 1 < P.CB) lbase[2 * threadIdx.x + 1] = excl + a0;
        __syncthreads();
    }
    const uint32_t fmask = (1u << P.fb) - 1u;
    const uint32_t sfirst = msm_find_seg(S, g0), slast = msm_find_seg(S, g1 - 1u);      // wave-uniform: the segments this tile touches (nearly always one)
#pragma unroll
    for (uint32_t it = 0; it < MSM_TILE1_PER; it++) {
        const uint32_t g = g0 + it * 256u + threadIdx.x;
        const uint32_t d = keep[it] & 0xffffu, mag = msm_dig_mag(d);
        if (mag) {
            const uint32_t bkt = mag - 1u, bin = bkt >> P.fb, pos = lbase[bin] + (keep[it] >> 16);
            uint32_t s = sfirst;
            for (uint32_t k = sfirst + 1; k <= slast; k++) if (g >= S.start[k]) s = k;
            const uint32_t i = g - S.start[s];
            staged[pos] = (msm_dig_neg(d) << 31) | ((bkt & fmask) << (31u - P.fb)) | (s << (27u - P.fb)) | i;
            sbin[pos] = (uint16_t)bin;
        }
    }
    __syncthreads();
    const uint32_t ntot = lbase[P.CB - 1] + cnt[P.CB - 1];     // number of entries of the tile
    for (uint32_t j = threadIdx.x; j < ntot; j += blockDim.x) { const uint32_t b = sbin[j]; entries1[gbase[b] + (j - lbase[b])] = staged[j]; }
}
// block per coarse bin k = (msm*W + window)*CB + bin: its entries are entries1[starts1[k*tmax], starts1[(k+1)*tmax]) (the last bin ends at
// the grand total).  Counting sort by the fine bits; the bin's range of `entries` is written by this block alone.
#define MSM_STASH 16384
#define SORT2_PER 36                    // entries per thread of a bin that k_msm_sort2 keeps in registers and stages in LDS (a power-of-two sum's 8,192-entry bins with room to spare)
__global__ void __launch_bounds__(256) k_msm_sort2(MsmPlan P, const uint32_t *__restrict__ starts1, uint32_t nflat, const uint32_t *__restrict__ entries1,
                                                   uint32_t *__restrict__ starts, uint32_t *__restrict__ entries) {
    __shared__ uint32_t cnt[128];
    __shared__ uint32_t cur[128];
    // 36 KB shared by the two shapes of a bin: up to 256 x SORT2_PER entries - the SORTED bin, staged here and copied out with consecutive lanes writing
    // consecutive addresses; a larger bin (up to MSM_STASH entries) - the 16-bit slot each entry took in its bucket, between the two passes over the coarse list
    __shared__ uint32_t shbuf[256 * SORT2_PER];
    uint16_t *const slot16 = reinterpret_cast<uint16_t *>(shbuf);
    static_assert(2 * 256 * SORT2_PER >= MSM_STASH, "the slot stash must fit the staging buffer");
    const uint32_t k = blockIdx.x, K = gridDim.x, nf = 1u << P.fb;
    const uint32_t s0 = starts1[(size_t)k * P.tmax], s1 = k + 1 < K ? starts1[(size_t)(k + 1) * P.tmax] : starts1[nflat];
    const bool stash = s1 - s0 <= MSM_STASH;
    for (uint32_t f = threadIdx.x; f < nf; f += blockDim.x) cnt[f] = 0;
    __syncthreads();
    const uint32_t fsh = 31u - P.fb, fmask = nf - 1u;
    // A bin of up to 256 x SORT2_PER entries is read ONCE - every load of a thread in flight together - and stays in registers with the slots its entries took;
    // the sorted bin goes through LDS, so that the 4-byte stores of a wave fall into four sectors instead of sixty-four (rounds 2-5, first session: every entry
    // its own sector of the bin's 32 KB range - the launch was bound by the L2's transaction rate, not by its atomics or its loads)
    const bool inregs = s1 - s0 <= 256u * SORT2_PER;                     // block-uniform
    uint32_t rv[SORT2_PER], rs[SORT2_PER / 2];
    if (inregs) {
#pragma unroll
        for (uint32_t u = 0; u < SORT2_PER; u++) { const uint32_t e = s0 + u * 256u + threadIdx.x; rv[u] = e < s1 ? entries1[e] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < SORT2_PER; u++) {
            const bool on = s0 + u * 256u + threadIdx.x < s1;
            const uint32_t slot = msm_lds_take(cnt, on ? (rv[u] >> fsh) & fmask : 0u, on);
            if (u & 1u) rs[u >> 1] |= slot << 16; else rs[u >> 1] = slot;
        }
    } else
    // four independent loads per thread and trip
    for (uint32_t e0 = s0 + threadIdx.x; e0 < s1 + 63u; e0 += 4u * blockDim.x) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) { const uint32_t e = e0 + u * blockDim.x; v[u] = e < s1 ? entries1[e] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t e = e0 + u * blockDim.x;
            const bool on = e < s1;
            const uint32_t slot = msm_lds_take(cnt, on ? (v[u] >> fsh) & fmask : 0u, on);
            if (on && stash) slot16[e - s0] = (uint16_t)slot;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                  // exclusive prefix over nf <= 128 counters: two per lane
        const uint32_t lane = threadIdx.x;
        const uint32_t a0 = 2 * lane < nf ? cnt[2 * lane] : 0u, a1 = 2 * lane + 1 < nf ? cnt[2 * lane + 1] : 0u;
        uint32_t incl = a0 + a1;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        const uint32_t excl = s0 + incl - (a0 + a1);
        if (2 * lane < nf) { cur[2 * lane] = excl; starts[(size_t)k * nf + 2 * lane] = excl; }
        if (2 * lane + 1 < nf) { cur[2 * lane + 1] = excl + a0; starts[(size_t)k * nf + 2 * lane + 1] = excl + a0; }
        if (k + 1 == K && lane == 0) starts[(size_t)K * nf] = s1;
    }
    __syncthreads();
    const uint32_t idxbits = 27u - P.fb, imask = (1u << idxbits) - 1u;
    if (inregs) {
#pragma unroll
        for (uint32_t u = 0; u < SORT2_PER; u++) {
            if (s0 + u * 256u + threadIdx.x < s1) {
                const uint32_t v = rv[u], f = (v >> fsh) & fmask, slot = (rs[u >> 1] >> (16u * (u & 1u))) & 0xffffu;
                shbuf[cur[f] - s0 + slot] = (v & 0x80000000u) | (((v >> idxbits) & 15u) << 27) | (v & imask);
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < s1 - s0; j += 256u) entries[s0 + j] = shbuf[j];
        return;
    }
    for (uint32_t e0 = s0 + threadIdx.x; e0 < s1 + 63u; e0 += 4u * blockDim.x) {
        uint32_t vv[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) { const uint32_t e = e0 + u * blockDim.x; vv[u] = e < s1 ? entries1[e] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t e = e0 + u * blockDim.x;
            const bool on = e < s1;
            const uint32_t v = vv[u], f = (v >> fsh) & fmask;
            const uint32_t pos = stash ? (on ? cur[f] + slot16[e - s0] : 0u) : msm_lds_take(cur, f, on);
            if (on) entries[pos] = (v & 0x80000000u) | (((v >> idxbits) & 15u) << 27) | (v & imask);
        }
    }
}

// exclusive scan of counts[0..nkeys) in two launches (chunk = 2048 keys per block): the block sums, then every block adds up the block sums before it
// (a few thousand values for the sums of a 2^20-gate proof, read from L2) and scans its own chunk - no serial pass over the block sums in between
// (rounds 1-4 had a one-wave launch for it: twelve launches per proof with nothing else to run beside them)
#define SCAN_CHUNK 2048
__global__ void __launch_bounds__(256) k_scan_blocksums(const uint32_t *__restrict__ counts, uint32_t nkeys, uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t lds[256];
    uint32_t base = blockIdx.x * SCAN_CHUNK, s = 0;
    for (uint32_t k = threadIdx.x; k < SCAN_CHUNK; k += 256) if (base + k < nkeys) s += counts[base + k];
    lds[threadIdx.x] = s; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) lds[threadIdx.x] += lds[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = lds[0];
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ counts, uint32_t nkeys, const uint32_t *__restrict__ blocksum,
                                                    uint32_t *__restrict__ starts, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t lds[256];
    uint32_t pre = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) pre += blocksum[b];
    lds[threadIdx.x] = pre; __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) lds[threadIdx.x] += lds[threadIdx.x + d]; __syncthreads(); }
    const uint32_t before = lds[0];
    __syncthreads();
    uint32_t base = blockIdx.x * SCAN_CHUNK;
    uint32_t v[8], s = 0;                       // thread owns 8 consecutive keys
#pragma unroll
    for (int k = 0; k < 8; k++) { uint32_t idx = base + threadIdx.x * 8 + k; v[k] = idx < nkeys ? counts[idx] : 0; s += v[k]; }
    lds[threadIdx.x] = s; __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {     // inclusive Hillis-Steele scan
        uint32_t x = threadIdx.x >= d ? lds[threadIdx.x - d] : 0; __syncthreads();
        lds[threadIdx.x] += x; __syncthreads();
    }
    uint32_t run = before + lds[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t idx = base + threadIdx.x * 8 + k;
        if (idx < nkeys) { starts[idx] = run; cursor[idx] = run; }
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) starts[nkeys] = run;      // the grand total
}

// Balanced bucket sweep.  The entry list is sorted by bucket (starts[]); thread c adds the points of the fixed-size chunk
// [c*CH, (c+1)*CH) whatever buckets it crosses, so a bucket that received thousands of terms (identical scalars: the -y^h
// padding terms of the first IPA round, repeated witness values, range-proof bits) is spread over many threads instead of
// serialising one.  A bucket that lies inside one chunk is stored directly; a bucket that crosses chunk boundaries leaves
// one partial per chunk (slotA = the piece of a bucket that began in an earlier chunk, slotB = the piece of a bucket that begins here and
// goes on) for k_bucket_combine.  CH is any length (not a power of two): the host picks it so that the blocks of a launch fill the CUs
// a whole number of times (engine.hip Impl::msm) - with 2^5 entries per chunk the 4,352 blocks of a 2^21-term launch were 4.25 rounds of the
// 1,024 blocks the device holds, and the last quarter round ran on a quarter of the machine.
#define MSM_NO_KEY 0xffffffffu
// bucket that holds sorted entry e: the k >= klo with starts[k] <= e < starts[k+1]  (upper_bound - 1)
__device__ __forceinline__ uint32_t msm_bucket_of(const uint32_t *__restrict__ starts, uint32_t nkeys, uint32_t e, uint32_t klo) {
    uint32_t lo = klo, hi = nkeys + 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= e) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
// q = +-P for the (halved) affine Niels point at `base` ((y+x)/2, (y-x)/2, dxy): a subtraction exchanges the first two (picked by ADDRESS: no select) and
// the sign of the third, which is the same as exchanging F and G of the addition formulas
__device__ __forceinline__ ge_ext ge_madd_swapped(const ge_ext &p, const fe &qp, const fe &qm, const fe &t2d, uint32_t neg) {
    fe A = fe_mul(fe_sub(p.Y, p.X), qm);
    fe B = fe_mul(fe_add(p.Y, p.X), qp);
    fe C = fe_mul(p.T, t2d);
    const fe &D = p.Z;                                         // halved Niels operand (ge.cuh): no doubling of Z
    fe E = fe_sub(B, A), F0 = fe_sub(D, C), G0 = fe_add(D, C), H = fe_add(B, A);
    const fe F = fe_select(F0, G0, neg), G = fe_select(G0, F0, neg);
    ge_ext r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
__global__ void __launch_bounds__(256) k_bucket_chunks(MsmSegs S, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ entries,
                                                       ge_ext *__restrict__ buckets, ge_ext *__restrict__ slotA, ge_ext *__restrict__ slotB,
                                                       uint32_t *__restrict__ open_key /* [chunk]: bucket whose first piece is slotB[chunk], or MSM_NO_KEY */,
                                                       uint32_t nkeys, uint32_t CH) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t e0w = (uint64_t)c * CH;
    const uint32_t M = starts[nkeys];                      // true entry count (zero digits were skipped)
    if (e0w >= M) return;
    const uint32_t e0 = (uint32_t)e0w;
    const uint32_t e1 = (e0w + CH < M) ? e0 + CH : M;
    uint32_t k = msm_bucket_of(starts, nkeys, e0, 0), kstart = starts[k], kend = starts[k + 1];
    uint32_t kend2 = starts[k + 2 <= nkeys ? k + 2 : nkeys];   // end of the next bucket, loaded one boundary ahead of its use
    uint32_t ent = entries[e0];
    ge_ext acc = ge_identity();
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t sg = (ent >> 27) & 15u, neg = ent >> 31;
        const fe *q = reinterpret_cast<const fe *>(S.pts[sg] + msm_point_index(S, sg, ent & 0x07ffffffu));
        const fe qp = q[neg], qm = q[neg ^ 1u], t2d = q[2];   // issued before the bookkeeping below
        if (e + 1 < e1) ent = entries[e + 1];
        if (e >= kend) {
            if (kstart >= e0) buckets[k] = acc;              // began in this chunk and ends in it
            else slotA[c] = acc;                             // the piece of a bucket that began in an earlier chunk (at most one per chunk: the first)
            acc = ge_identity();
            k++; kstart = kend; kend = kend2;
            if (e >= kend) {                                 // a run of empty buckets (half of a 15-bit window is structurally
                k = msm_bucket_of(starts, nkeys, e, k + 1);  // empty): search instead of walking it with dependent loads
                kstart = starts[k]; kend = starts[k + 1];
            }
            kend2 = starts[k + 2 <= nkeys ? k + 2 : nkeys];
        }
        acc = ge_madd_swapped(acc, qp, qm, t2d, neg);
    }
    uint32_t open = MSM_NO_KEY;
    if (kstart >= e0 && kend <= e1) buckets[k] = acc;
    else if (kstart < e0) slotA[c] = acc;                    // a bucket that began earlier (and may go on beyond this chunk)
    else { slotB[c] = acc; open = k; }                       // begins here and goes on: k_bucket_combine's thread c joins its pieces
    open_key[c] = open;
}

// Joining the pieces of the buckets that cross chunk boundaries.  One thread per CHUNK BOUNDARY (round 4; rounds 1-3 ran one thread per bucket,
// and since a wave executes the addition as soon as one of its lanes needs it, a million mostly idle bucket threads issued 2.5x the additions
// that were needed): thread c looks at the bucket that holds the LAST entry of chunk c.  If that bucket began in chunk c and goes on beyond it,
// the thread owns it (the sweep leaves the bucket's key in open_key[c]): slotB[c] + slotA[c+1 .. c1].  A bucket that began earlier belongs to the thread of the chunk it began in; a bucket that
// ends with the chunk needs nothing.  With 64-entry chunks and ~32 entries per bucket nearly every boundary is crossed by exactly one bucket:
// every lane does one addition.  A bucket spread over more than HEAVY_CHUNKS chunks (thousands of identical scalars: the -y^h padding terms of
// the first IPA round, repeated witness values) goes on the heavy list for k_bucket_combine_heavy.  Empty buckets are never written: the
// epilogue (k_bucket_reduce) takes the identity for a bucket whose range is empty.
#define HEAVY_CHUNKS 32
__global__ void __launch_bounds__(256) k_bucket_combine(const uint32_t *__restrict__ starts, ge_ext *__restrict__ buckets,
                                                        const ge_ext *__restrict__ slotA, const ge_ext *__restrict__ slotB,
                                                        const uint32_t *__restrict__ open_key, uint32_t nkeys, uint32_t CH,
                                                        uint32_t *__restrict__ heavy /* [0] = count, then keys */,
                                                        uint32_t *__restrict__ medium /* [0] = count, then keys: buckets with pieces beyond the second */) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = starts[nkeys];
    if ((uint64_t)c * CH >= M) return;                           // the sweep had no such chunk (open_key[c] was not written)
    const uint32_t k = open_key[c];                              // the bucket that began in chunk c and goes on beyond it, as the sweep saw it
    if (k == MSM_NO_KEY) return;
    const uint32_t c1 = (starts[k + 1] - 1) / CH;
    if (c1 - c > HEAVY_CHUNKS) { heavy[1 + atomicAdd(&heavy[0], 1u)] = k; return; }
    // exactly ONE addition per lane here: a bucket that crosses a second boundary (a few per cent of them) would make its whole wave run the
    // addition again, so its remaining pieces go on a compacted list that k_bucket_combine_heavy finishes with one thread per bucket
    buckets[k] = ge_add(slotB[c], slotA[c + 1]);
    const bool more = c1 > c + 1;
    const uint64_t vote = __ballot(more);
    if (vote) {                                                  // one counter update per wave
        const uint32_t lane = threadIdx.x & 63u, lead = (uint32_t)__ffsll((unsigned long long)vote) - 1u;
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(&medium[0], (uint32_t)__popcll(vote));
        base = __shfl(base, lead, 64);
        if (more) medium[1 + base + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull))] = k;
    }
}
// The same join with one thread per BUCKET: the better shape when buckets are longer than chunks (a proof alone on the device: 15-bit windows, 64
// entries per bucket, 34-entry chunks - every bucket crosses two or three boundaries, so every lane has its two additions, while two of three
// boundary threads would idle).  The host picks by the average bucket length (engine.hip Impl::msm).
__global__ void __launch_bounds__(256) k_bucket_combine_per_bucket(const uint32_t *__restrict__ starts, ge_ext *__restrict__ buckets,
                                                                   const ge_ext *__restrict__ slotA, const ge_ext *__restrict__ slotB,
                                                                   uint32_t nkeys, uint32_t CH, uint32_t *__restrict__ heavy /* [0] = count, then keys */) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nkeys) return;
    const uint32_t s0 = starts[k], s1 = starts[k + 1];
    if (s0 == s1) return;                                        // empty: never read (k_bucket_reduce takes the identity)
    const uint32_t c0 = s0 / CH, c1 = (s1 - 1) / CH;
    if (c0 == c1) return;                                        // inside one chunk: the sweep stored it
    if (c1 - c0 > HEAVY_CHUNKS) { heavy[1 + atomicAdd(&heavy[0], 1u)] = k; return; }
    ge_ext acc = slotB[c0];
    for (uint32_t c = c0 + 1; c <= c1; c++) acc = ge_add(acc, slotA[c]);
    buckets[k] = acc;
}
// one BLOCK per heavy bucket (grid-stride over the list): thread-strided partial sums, then the block sum in quad layout (k_points.cuh ge_block_sum_quad) - the
// 55,000 padding terms of a 2^20-gate proof's first inner-product round are one bucket of 1,600 pieces per window: 6 additions per thread and the tree, where a
// single wave (rounds 1-4) ran 25 and a tree of its own, 130 us on the critical path of the round; then the medium list, one thread per bucket
__global__ void __launch_bounds__(256) k_bucket_combine_heavy(const uint32_t *__restrict__ starts, ge_ext *__restrict__ buckets,
                                                              const ge_ext *__restrict__ slotA, const ge_ext *__restrict__ slotB,
                                                              uint32_t CH, const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ medium) {
    __shared__ ge_ext lds[256];
    const uint32_t count = heavy[0];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {                    // block-uniform trip count
        const uint32_t k = heavy[1 + it];
        const uint32_t c0 = starts[k] / CH, c1 = (starts[k + 1] - 1) / CH, P = c1 - c0 + 1;      // pieces: slotB[c0], slotA[c0 + 1 .. c1]
        ge_ext acc = ge_identity();
        uint32_t p = threadIdx.x;
        if (p < P) { acc = p ? slotA[c0 + p] : slotB[c0]; p += 256; }                // a thread's first piece as it is
        for (; p < P; p += 256) acc = ge_add(acc, slotA[c0 + p]);
        const fe sum = ge_block_sum_quad(acc, lds);
        if (threadIdx.x < 4) reinterpret_cast<fe *>(buckets + k)[threadIdx.x] = sum;
        __syncthreads();                                                             // lds is reused by the next bucket
    }
    // the medium list of k_bucket_combine: buckets[k] already holds the first two pieces; one thread adds the remaining ones
    const uint32_t mcount = medium[0];
    for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < mcount; it += gridDim.x * blockDim.x) {
        const uint32_t k = medium[1 + it];
        const uint32_t c0 = starts[k] / CH, c1 = (starts[k + 1] - 1) / CH;
        ge_ext acc = buckets[k];
        for (uint32_t c = c0 + 2; c <= c1; c++) acc = ge_add(acc, slotA[c]);
        buckets[k] = acc;
    }
}

// Epilogue of a window: S = sum_b (b+1) * bucket[b] over its nb buckets, in two levels of running sums and one weighted tree - no scalar
// multiplication by a bucket index anywhere (rounds 1-3 added lo * run per segment by double-and-add: 15 doublings + ~7 additions for every 8
// buckets, more field multiplications than the running sums themselves and a third of the instructions of the sweep it follows).
//   k_bucket_reduce  thread t = (window, segment of `seg` buckets starting at lo = sg*seg): run_t = sum B_b, acc_t = sum (b - lo + 1) B_b
//                    (2 additions per bucket); partial[t] = acc_t, partial[total + t] = run_t
//   k_window_sums    block per window (64..512 threads: two waves per SIMD, 256 VGPRs for three live points and an addition; per = segments per thread, a power of two):
//                      S = sum_t acc_t + seg * sum_t sg * run_t,    sg = tau * per + j for thread tau
//                    thread: Q = sum acc_t + seg * sum_j j * run_t (running sums again, 3 additions per segment), R = sum run_t;
//                    sum_tau tau * R_tau = sum_{tau >= 1} Suf_tau with Suf_tau = sum_{tau' >= tau} R_tau' - a suffix scan over the block
//                    (wave shuffles, then the wave totals through LDS) instead of multiplications by tau; S = sum_tau (Q_tau + seg*per*Suf_tau).
__global__ void __launch_bounds__(64) k_bucket_reduce(const ge_ext *__restrict__ buckets, const uint32_t *__restrict__ starts, ge_ext *__restrict__ partial,
                                                     uint32_t nb, uint32_t seg, uint32_t nseg_per_win, uint32_t total) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    uint32_t win = t / nseg_per_win, sg = t % nseg_per_win;
    uint32_t lo = sg * seg;
    const ge_ext *B = buckets + (size_t)win * nb;
    const uint32_t *S = starts + (size_t)win * nb;                 // bucket b of this window holds entries [S[b], S[b+1]): empty -> the identity, B[b] was never written
    uint32_t hi = S[lo + seg];
    uint32_t cur = S[lo + seg - 1];
    ge_ext run = cur != hi ? B[lo + seg - 1] : ge_identity(), acc = run;
    for (int32_t b = (int32_t)(lo + seg) - 2; b >= (int32_t)lo; b--) {
        hi = cur; cur = S[b];
        if (cur != hi) run = ge_add(run, B[b]);
        acc = ge_add(acc, run);
    }
    partial[t] = acc; partial[(size_t)total + t] = run;
}
__device__ __forceinline__ ge_ext ge_dbl_times(ge_ext p, uint32_t k) { for (uint32_t i = 0; i < k; i++) p = ge_dbl(p); return p; }
__device__ __forceinline__ ge_ext ge_shfl_down(const ge_ext &p, uint32_t d) {
    ge_ext r; r.X = fe_shfl_down(p.X, d); r.Y = fe_shfl_down(p.Y, d); r.Z = fe_shfl_down(p.Z, d); r.T = fe_shfl_down(p.T, d); return r; }
// lane l <- sum_{l' >= l} of the wave (lanes >= n hold the identity)
__device__ __forceinline__ ge_ext ge_wave_suffix(ge_ext s, uint32_t lane, uint32_t n) {
    for (uint32_t d = 1; d < n; d <<= 1) { const ge_ext o = ge_shfl_down(s, d); if (lane + d < n) s = ge_add(s, o); }
    return s; }
// lane 0 <- sum of lanes [0, n), n a power of two
__device__ __forceinline__ ge_ext ge_wave_sum(ge_ext s, uint32_t lane, uint32_t n) {
    for (uint32_t d = n >> 1; d > 0; d >>= 1) { const ge_ext o = ge_shfl_down(s, d); if (lane < d) s = ge_add(s, o); }
    return s; }
__global__ void __launch_bounds__(512) k_window_sums(const ge_ext *__restrict__ partial, ge_ext *__restrict__ wsum, uint32_t nseg_per_win, uint32_t total,
                                                     uint32_t lgseg) {
    __shared__ ge_ext ldsT[8], ldsH[8];
    const ge_ext *A = partial + (size_t)blockIdx.x * nseg_per_win, *Rn = A + total;
    const uint32_t nthr = blockDim.x, nw = nthr >> 6, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t per = nseg_per_win > nthr ? nseg_per_win / nthr : 1u, lgper = 31u - (uint32_t)__builtin_clz(per);
    const uint32_t t0 = threadIdx.x * per;
    ge_ext Q = ge_identity(), R = ge_identity();
    if (t0 < nseg_per_win) {
        // zero-based running sum from the top: Z = sum_j j * run_(t0+j), R = sum run, Q = sum acc
        ge_ext Z = ge_identity();
        R = Rn[t0 + per - 1]; Q = A[t0 + per - 1];
        if (per > 1) Z = R;
        for (uint32_t k = per - 1; k-- > 0;) {
            R = ge_add(R, Rn[t0 + k]); Q = ge_add(Q, A[t0 + k]);
            if (k > 0) Z = ge_add(Z, R);
        }
        if (per > 1) Q = ge_add(Q, ge_dbl_times(Z, lgseg));
    }
    // suffix sums of R over the block
    ge_ext S = ge_wave_suffix(R, lane, 64);
    if (nw > 1) {
        if (lane == 0) ldsT[wv] = S;
        __syncthreads();
        if (wv == 0) {
            ge_ext Tt = lane < nw ? ldsT[lane] : ge_identity();
            Tt = ge_wave_suffix(Tt, lane, nw);                            // inclusive; wave w needs the exclusive one: that of w + 1
            if (lane < nw) ldsH[lane] = Tt;
        }
        __syncthreads();
        if (wv + 1 < nw) S = ge_add(S, ldsH[wv + 1]);
    }
    ge_ext X = Q;
    if (threadIdx.x > 0 && t0 < nseg_per_win) X = ge_add(X, ge_dbl_times(S, lgseg + lgper));
    X = ge_wave_sum(X, lane, 64);
    if (nw > 1) {
        __syncthreads();                                                   // ldsT is reused
        if (lane == 0) ldsT[wv] = X;
        __syncthreads();
        if (wv == 0) { X = lane < nw ? ldsT[lane] : ge_identity(); X = ge_wave_sum(X, lane, nw); }
    }
    if (threadIdx.x == 0) wsum[blockIdx.x] = X;
}

// The same sums for a proof ALONE on the device (round 5, second session): k_window_sums is a chain of ~35 dependent point additions per window on 34 CUs (two waves
// per SIMD, 6.8 us per addition); here a point has FOUR lanes (k_points.cuh quad_*: ~750 instructions per addition on the wave instead of ~1,650) and a window
// is spread over several blocks, one wave per SIMD on the whole device.  Grid (nblk, windows), 256 threads = 64 slots; slot s of block beta owns the segments
// t = (beta * 64 + s) * per + i, i < per.  Stage A, per block: U = sum acc_t + seg * sum (t - t0) run_t and Rt = sum run_t (running sums in the slot, suffix scan
// and tree over the slots as in k_window_sums).  A window of one block is done; otherwise the block that finishes LAST (a ticket per window; agent-scope stores,
// fence, atomic) repeats the slot-level step on the nblk pairs: S = sum U_beta + seg * span * sum beta Rt_beta.  More instructions than k_window_sums: a proof that
// shares the device keeps that one.
// slots of a block, (Q_s, R_s) in quad layout -> slot 0 of wave 0: Q = sum_s Q_s + 2^shift * sum_s s R_s, R = sum_s R_s.  nw waves take part (1, or the
// block's 4: block-uniform), n16 live slots per wave (a power of two <= 16; the slots beyond hold the identity)
__device__ __forceinline__ void wq_combine(fe &Q, fe &R, uint32_t shift, uint32_t nw, uint32_t n16, fe (*ldsT)[4], fe (*ldsU)[4], uint32_t r, uint32_t wv, uint32_t s16) {
    fe S = R;                                                       // suffix sums of R over the wave's slots, then over the waves
    for (uint32_t d = 1; d < n16; d <<= 1) { const fe o = fe_shfl_down(S, 4u * d); const fe t = quad_add(S, o, r); S = fe_select(S, t, s16 + d < n16); }
    if (nw > 1) {
        if (s16 == 0) ldsT[wv][r] = S;
        __syncthreads();
        if (wv + 1 < nw) {                                          // wave-uniform
            fe h = ldsT[nw - 1][r];
            for (uint32_t w2 = nw - 2; w2 > wv; w2--) h = quad_add(h, ldsT[w2][r], r);
            S = quad_add(S, h, r);
        }
    }
    R = S;
    fe D = S;
    for (uint32_t i = 0; i < shift; i++) D = quad_dbl(D, r);
    const fe X = quad_add(Q, D, r);
    Q = fe_select(X, Q, wv == 0 && s16 == 0);                       // the block's first slot has weight 0
    for (uint32_t d = n16 >> 1; d > 0; d >>= 1) { const fe o = fe_shfl_down(Q, 4u * d); Q = quad_add(Q, o, r); }
    if (nw > 1) {
        if (s16 == 0) ldsU[wv][r] = Q;
        __syncthreads();
        if (wv == 0) {
            fe v = s16 < nw ? ldsU[s16][r] : quad_identity(r);
            for (uint32_t d = nw >> 1; d > 0; d >>= 1) { const fe o = fe_shfl_down(v, 4u * d); v = quad_add(v, o, r); }
            Q = v;
        }
    }
}
__device__ __forceinline__ void wq_store(ge_ext *p, uint32_t r, const fe &c) {           // coordinate r, visible to the other blocks of the kernel
    uint32_t *w = reinterpret_cast<uint32_t *>(reinterpret_cast<fe *>(p) + r);
#pragma unroll
    for (int j = 0; j < 8; j++) __hip_atomic_store(w + j, c.v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ fe wq_load(const ge_ext *p, uint32_t r) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<const fe *>(p) + r);
    fe c;
#pragma unroll
    for (int j = 0; j < 8; j++) c.v[j] = __hip_atomic_load(w + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return c;
}
__global__ void __launch_bounds__(256) k_window_sums_quad(const ge_ext *__restrict__ partial, ge_ext *__restrict__ wsum, ge_ext *stage /* [window][nblk][2] */,
                                                          uint32_t *tickets /* [window], zero between launches */, uint32_t nseg_per_win, uint32_t total,
                                                          uint32_t lgseg, uint32_t lgper) {
    __shared__ fe ldsT[4][4], ldsU[4][4];
    __shared__ uint32_t is_last;
    const uint32_t r = threadIdx.x & 3u, slot = threadIdx.x >> 2, wv = threadIdx.x >> 6, s16 = slot & 15u;
    const uint32_t win = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x, per = 1u << lgper;
    const ge_ext *A = partial + (size_t)win * nseg_per_win, *Rn = A + total;
    const uint32_t t0 = (blk * 64u + slot) * per;
    fe Q = quad_identity(r), R = Q;
    if (t0 < nseg_per_win) {                                        // the same for the four lanes of a slot
        R = quad_load_ext(Rn + t0 + per - 1, r); Q = quad_load_ext(A + t0 + per - 1, r);
        if (per > 1) {                                              // zero-based running sum from the top, as in k_window_sums
            fe Z = R;
            for (uint32_t k = per - 1; k-- > 0;) {
                R = quad_add(R, quad_load_ext(Rn + t0 + k, r), r); Q = quad_add(Q, quad_load_ext(A + t0 + k, r), r);
                if (k > 0) Z = quad_add(Z, R, r);
            }
            for (uint32_t i = 0; i < lgseg; i++) Z = quad_dbl(Z, r);
            Q = quad_add(Q, Z, r);
        }
    }
    wq_combine(Q, R, lgseg + lgper, 4u, 16u, ldsT, ldsU, r, wv, s16);
    if (nblk == 1) { if (threadIdx.x < 4) reinterpret_cast<fe *>(wsum + win)[r] = Q; return; }
    ge_ext *mine = stage + ((size_t)win * nblk + blk) * 2;
    if (threadIdx.x < 4) { wq_store(mine, r, Q); wq_store(mine + 1, r, R); }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(&tickets[win], 1u);
        is_last = t == nblk - 1u ? 1u : 0u;
        if (t == nblk - 1u) tickets[win] = 0;                       // every other block of the window has taken its ticket
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // stage B: the nblk (U, Rt) pairs of the window, one per slot; block beta's segments start at beta * 64 * per
    const uint32_t nwB = nblk > 16u ? 4u : 1u, n16B = nblk > 16u ? 16u : nblk;
    if (wv >= nwB) return;
    const ge_ext *all = stage + (size_t)win * nblk * 2;
    Q = quad_identity(r); R = Q;
    if (slot < nblk) { Q = wq_load(all + 2 * slot, r); R = wq_load(all + 2 * slot + 1, r); }
    wq_combine(Q, R, lgseg + lgper + 6u, nwB, n16B, ldsT, ldsU, r, wv, s16);
    if (threadIdx.x < 4) reinterpret_cast<fe *>(wsum + win)[r] = Q;
}

// The recombination of the W window sums, sum_j 2^off(j) * S_j (about 254 dependent doublings of one point), runs on the host
// (csrc/host/fe51.hpp pt_horner): a lone wave issues such a chain at half rate, a 5 GHz core is an order of magnitude faster on it.

}  // namespace bpg
