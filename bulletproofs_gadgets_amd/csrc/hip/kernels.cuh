// HIP kernels of the Bulletproofs R1CS prove path for gfx950 (MI355X).  Integer VALU work (v_mad_u64_u32 chains);
// no MFMA - this is 255-bit modular arithmetic, not a dense contraction.  Hot-path rows of SURVEY.md section 8(a):
//   a7  BulletproofGens::new            k_gens_derive + k_normalize_niels
//   a1-a3,a6  Pedersen commits          k_pedersen (window tables of B and B_blinding: k_tt_bases, k_tt_multiples)
//   a9  A_I, A_O, S multiscalar muls    k_msm_digits (digits + coarse counts), k_scan_blocksums / _apply, k_msm_scatter1, k_msm_sort2 (two-level sort), k_bucket_chunks,
//                                       k_bucket_combine(_per_bucket, _heavy), k_bucket_reduce, k_window_sums (a shared device) / k_window_sums_quad (a proof alone)
//                                       (bucket method: digits -> coarse partition in LDS -> fine counting sort -> balanced bucket sweep -> reductions;
//                                       the 17 window sums are recombined and encoded on the host, host/fe51.hpp)
//       equal scalars of a_L, a_R        k_merge_insert, k_merge_flags, k_merge_members, k_merge_sum (k_merge.cuh): terms of A_I that carry the same value share one
//                                       bucket entry per window on the sum of their generators; once per uploaded witness
//   a10 vector-polynomial phase         k_exp_table, k_flatten, k_poly_t, k_poly_eval, k_reduce_partials
//   a11 inner-product argument          above 2^12 generators (a circuit of N <= 2^14: none): k_ipa_prep, the MSM kernels, k_tt_advance, one generator fold per group of rounds -
//                                       k_fold_points_wnaf on the original generators (tables of (2m+1) * 2^(64j) * P: k_odd_start(_ext), k_odd_step,
//                                       k_dbl_times); on folded ones k_fold_points_quadw (a proof alone) / k_fold_points_regw (a shared device): width-4 NAF
//                                       steps against multiples the kernel makes itself - and k_fold_points_quad / _split / _reg<NT> / k_fold_points for the
//                                       shapes those two do not take (first groups, outputs that do not fill whole blocks); below: k_tt_bases,
//                                       k_tt_multiples, k_tt_factors, then k_tt_advance, k_tt_round, k_tt_finish per round
//   f1  Verifier::verify                k_decompress, k_flatten_const, k_ipa_s, k_verify_scalars + one MSM
// The kernels live in k_points.cuh, k_scalars.cuh, k_ipa.cuh, k_verify.cuh and k_msm.cuh, included at the end of this file in that order.
// Data layout in HBM: scalars = 8 x u32 Montgomery form, 32 B each, AoS (lane i <-> element i: 2 x 16 B coalesced
// loads); generator tables = affine Niels (y+x, y-x, 2dxy), 96 B per point, G at [0,N) and H at [N,2N); window tables =
// projective Niels (y+x, y-x, z, 2dt), 128 B per entry.
#pragma once
#include <hip/hip_runtime.h>
#include "ge.cuh"
#include "sc.cuh"

namespace bpg {

#define BPG_MAX_SEGS 16                 // 4 segment bits in a sorted entry (sign | segment | 27 index bits)
struct MsmSegs {
    const scm *sc[BPG_MAX_SEGS];
    const ge_niels *pts[BPG_MAX_SEGS];
    const uint32_t *skip[BPG_MAX_SEGS]; // optional bit per term: 1 = the term takes no part (its scalar was merged into another segment's term, k_merge_*); nullptr = none
    uint32_t len[BPG_MAX_SEGS];
    uint32_t start[BPG_MAX_SEGS + 1];   // prefix of len
    uint32_t msm[BPG_MAX_SEGS];
    uint32_t lgblk[BPG_MAX_SEGS];       // element e of the segment is point ((e >> lgblk) << (lgblk+1)) | (e & (2^lgblk - 1)): every
    uint32_t nseg;                      // other block of 2^lgblk points (grouped IPA rounds); 31 = contiguous
};
__device__ __forceinline__ uint32_t msm_point_index(const MsmSegs &S, uint32_t seg, uint32_t e) {
    const uint32_t lg = S.lgblk[seg];
#ifdef BPG_DIAG_INDEX_MASK
    // DIAGNOSTIC BUILD ONLY (tools/diag/mask_build.sh; never defined for the product): every gathered point index is masked, so the sweep does the
    // same arithmetic on a table that fits a cache level - wrong sums, same instruction stream - to tell exposed memory time from issue time
    return (((e >> lg) << (lg + 1)) | (e & ((1u << lg) - 1u))) & (BPG_DIAG_INDEX_MASK);
#else
    return ((e >> lg) << (lg + 1)) | (e & ((1u << lg) - 1u));
#endif
}

}  // namespace bpg

#include "k_points.cuh"
#include "k_scalars.cuh"
#include "k_ipa.cuh"
#include "k_verify.cuh"
#include "k_msm.cuh"
#include "k_merge.cuh"
