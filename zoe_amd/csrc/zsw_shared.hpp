// zsw_shared.hpp — entry points of the one-profile-many-sequences role (zsw_shared.hip, zsw_align.hip).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

constexpr size_t SHARED_MAX_LDS = 60 * 1024;  // reads of up to ~6,800 rows (residues + strip boundary of every row in LDS)
inline uint32_t shared_max_rows() { return (uint32_t)((SHARED_MAX_LDS - 64) / 9) & ~3u; }

// sw_simd_score_ends(reference = read i, profile of d_pseq) for every item (striped.rs:153-336): out.ref_end / out.query_end =
// exclusive ends in the read / in the profile sequence. With rev_ref_end / rev_query_end (the forward ends): the reverse pass of
// sw_simd_score_ranges, out.ref_end / out.query_end = the inclusive starts.
hipError_t launch_shared_ends(const BatchDev& b, uint32_t max_rows, const uint8_t* d_pseq, uint32_t plen, const ScoringDev* d_sc,
                              const ResultRule& rule, const ScoreOut& out, const uint32_t* rev_ref_end, const uint32_t* rev_query_end,
                              hipStream_t stream, const uint32_t* n_items_dev = nullptr /* the number of b.items, on the device */);

// Pass 2 of sw_simd_align with the shared profile: <N lanes, nv = ceil(plen / N) vectors> striping over d_pseq, rows = the bases
// of read i, every row's flags kept (W = the longest read). d_score / d_ref_end / d_status: the shared ends pass.
size_t align_shared_ring_bytes(int N, uint32_t plen, uint32_t W, uint32_t grid, int S);
hipError_t align_pass2_shared(int N, const uint8_t* d_pseq, uint32_t plen, const BatchDev& b, const ScoringDev* d_sc, int S,
                              const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W, uint32_t maxc,
                              uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item, uint64_t* d_cig_start,
                              uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list, uint32_t* d_fb_count, int invert,
                              hipStream_t stream, bool half_ok /* every score of these reads fits 16 bits: the 8- and 16-bit instantiations */);

}  // namespace zsw
