// equal-scalar merging of a witness commitment - part of kernels.cuh (included from there; see its header for the kernel map)
#pragma once

namespace bpg {

// ------------------------------------------------------------------------------------------------ equal-scalar merging
// A_I = <a_L, G> + <a_R, H> + i_blinding * B_blinding and A_O = <a_O, G> + o_blinding * B_blinding (Prover::prove, reference src/bin/prover.rs:93) are sums
// over FIXED points.  Terms that carry the same scalar can share their sixteen bucket entries: s*P + s*Q + s*R = s*(P + Q + R).  Gadget circuits are full
// of such equalities - a MiMC round multiplies (t, t) -> t^2 and (t^2, t) -> t^3 (reference src/mimc_hash/mimc_hash_gadget.rs:133-144), so
// a_L[2i] = a_R[2i] = a_R[2i+1] = t: three of A_I's four terms per round collapse into one; and the reference's own 2^20 circuit hashes 512 EQUAL leaves
// (src/merkle_tree/merkle_tree_gadget.rs:473-545), so every value of a level-k node occurs 2^(8-k) times over: its 2.98 M terms of A_I and A_O carry
// about 35,000 distinct scalars.  Nothing here knows about gadgets: the terms are grouped by VALUE with a hash table.
//   k_merge_insert   term t (scalar A[t] on PA[t] for t < nA, B[t-nA] on PB[t-nA] beyond), non-zero: finds / claims the slot of its value (open
//                    addressing, atomicCAS on the slot's representative term, full 32-byte compare on a hit), counts the slot's members
//   k_merge_plan     a slot with c >= 2 members becomes ceil(c / MERGE_GMAX) groups (a group is summed by one thread: at most 32 additions in a row, so that
//                    the grouping costs a proof a fraction of a millisecond even when it is redone for every proof; the groups of one value are
//                    separate terms with the same scalar, which the sweep adds up in their common buckets) and takes c member places
//   (two scans)      member places and group numbers of every slot
//   k_merge_groups   the slot of every group
//   k_merge_members  members enter their slot's list and set their bit in the skip masks (MsmSegs::skip: no entry in any window)
//   k_merge_sum      one thread per group: the sum of its generator points, and the group's scalar
//   (k_normalize_niels) the sums as affine Niels points, a segment of the MSM like any generator table
// Done once per uploaded witness (Engine::Impl::merge_witness), not per proof: the grouping depends on the witness alone.  The order in which members
// arrive in a list depends on the schedule, so the projective sums differ from run to run; the normalised points are the same group elements.
#define MERGE_EMPTY 0xffffffffu
#define MERGE_GMAX 32u
struct MergeTerms { const scm *A, *B; const ge_niels *PA, *PB; uint32_t nA, nterms; };
__device__ __forceinline__ const scm &merge_scalar(const MergeTerms &T, uint32_t t) { return t < T.nA ? T.A[t] : T.B[t - T.nA]; }
__device__ __forceinline__ uint32_t merge_hash(const scm &s) {
    uint32_t h = 0x9e3779b9u;
#pragma unroll
    for (int k = 0; k < 8; k++) { h ^= s.v[k]; h *= 0x85ebca6bu; h ^= h >> 15; }
    return h;
}
__device__ __forceinline__ bool merge_equal(const scm &a, const scm &b) {
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d |= a.v[k] ^ b.v[k];
    return d == 0;
}
__global__ void __launch_bounds__(256) k_merge_insert(MergeTerms T, uint32_t *__restrict__ rep /* [slots], MERGE_EMPTY */, uint32_t *__restrict__ count /* [slots], zero */,
                                                      uint32_t *__restrict__ slot_of /* [nterms] */, uint32_t lgslots) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T.nterms) return;
    const scm s = merge_scalar(T, t);
    if (sc_iszero(s)) { slot_of[t] = MERGE_EMPTY; return; }              // no entries anyway
    const uint32_t mask = (1u << lgslots) - 1u;
    uint32_t slot = merge_hash(s) & mask;
    for (;;) {                                                            // load <= 1/2: terminates
        uint32_t cur = __hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // hundreds of terms may carry this value: only the first few race for the slot
        if (cur == MERGE_EMPTY) cur = atomicCAS(&rep[slot], MERGE_EMPTY, t);
        if (cur == MERGE_EMPTY) break;                                    // claimed
        if (merge_equal(merge_scalar(T, cur), s)) break;                  // a slot only ever holds terms of one value, so any representative will do
        slot = (slot + 1u) & mask;
    }
    slot_of[t] = slot;
    atomicAdd(&count[slot], 1u);
}
__global__ void __launch_bounds__(256) k_merge_plan(const uint32_t *__restrict__ count, uint32_t *__restrict__ msize, uint32_t *__restrict__ gcount, uint32_t slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const uint32_t c = count[i];
    msize[i] = c >= 2u ? c : 0u;
    gcount[i] = c >= 2u ? (c + MERGE_GMAX - 1u) / MERGE_GMAX : 0u;
}
__global__ void __launch_bounds__(256) k_merge_groups(const uint32_t *__restrict__ count, const uint32_t *__restrict__ goff, uint32_t *__restrict__ gslot, uint32_t slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const uint32_t c = count[i];
    if (c < 2u) return;
    const uint32_t ng = (c + MERGE_GMAX - 1u) / MERGE_GMAX, g0 = goff[i];
    for (uint32_t k = 0; k < ng; k++) gslot[g0 + k] = i;
}
__global__ void __launch_bounds__(256) k_merge_members(MergeTerms T, const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ count, const uint32_t *__restrict__ moff,
                                                       uint32_t *__restrict__ fill /* [slots], zero */, uint32_t *__restrict__ members,
                                                       uint32_t *__restrict__ skipA, uint32_t *__restrict__ skipB) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T.nterms) return;
    const uint32_t slot = slot_of[t];
    if (slot == MERGE_EMPTY || count[slot] < 2u) return;
    members[moff[slot] + atomicAdd(&fill[slot], 1u)] = t;
    const uint32_t i = t < T.nA ? t : t - T.nA;
    atomicOr(&(t < T.nA ? skipA : skipB)[i >> 5], 1u << (i & 31u));
}
__global__ void __launch_bounds__(64) k_merge_sum(MergeTerms T, const uint32_t *__restrict__ count, const uint32_t *__restrict__ moff, const uint32_t *__restrict__ goff,
                                                  const uint32_t *__restrict__ gslot, const uint32_t *__restrict__ members, uint32_t groups,
                                                  ge_ext *__restrict__ sums, scm *__restrict__ mscal) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const uint32_t slot = gslot[g], m0 = moff[slot] + (g - goff[slot]) * MERGE_GMAX, m1 = min(moff[slot] + count[slot], m0 + MERGE_GMAX);
    ge_ext acc = ge_identity();
    for (uint32_t k = m0; k < m1; k++) {
        const uint32_t t = members[k];
        acc = ge_madd(acc, t < T.nA ? T.PA[t] : T.PB[t - T.nA]);
    }
    sums[g] = acc;
    mscal[g] = merge_scalar(T, members[m0]);
}

}  // namespace bpg
