// zsw_align.hpp — entry points of the alignment (traceback) kernels (zsw_align.hip).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

// Pass 2 for one <N lanes, nv vectors> group of reads (b.items lists them).
hipError_t align_pass2(int N, uint32_t nv, const BatchDev& b, const uint8_t* d_ref, uint32_t ref_len, const ScoringDev* d_sc,
                       int S, const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W,
                       uint32_t maxc, uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item,
                       uint64_t* d_cig_start, uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list,
                       uint32_t* d_fb_count, int invert, hipStream_t stream);
size_t align_ring_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid);
size_t align_lds_need(uint32_t nv, int S);
// count_only: per-block sums + total of n_ciglets; otherwise the packed write (and range inversion).
hipError_t align_finalize(zsw_alignment* d_aln, const uint8_t* d_status, uint32_t n, uint64_t* d_block_sums,
                          uint64_t* d_total, const uint64_t* d_cig_start, const uint32_t* d_cig_raw, int invert,
                          uint32_t* out_inc, uint8_t* out_op, uint64_t cap, bool count_only, hipStream_t stream);

}  // namespace zsw
