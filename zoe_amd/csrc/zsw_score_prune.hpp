// zsw_score_prune.hpp — launcher of the column-pruned score-only pass (zsw_score_prune.hip).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

constexpr int PR_CP = 24;   // query columns of the strip that sees every reference row
constexpr int PR_G2 = 4;    // lanes per read pair of the window kernel
constexpr int PR_C2 = 32;   // columns per lane of the window kernel: reads of up to PR_CP + PR_G2 * PR_C2 = 152 bases
constexpr int PR_BLK = 32;  // rows per block of the boundary maxima (windows are whole blocks)
constexpr int PR_M1 = 8;    // window rows kept above the anchor
constexpr int PR_M2 = 24;   // and below the row where an alignment without deletions ends
constexpr uint32_t PR_MIN_READS = 1u << 16;    // smaller batches cannot fill the chip with one pair per lane: the full pass is faster
constexpr uint32_t PR_CHUNK_READS = 2u << 20;  // reads per round of the two kernels (16 KiB of boundary stream per pair at 2 kb)

struct ScoreArgsV2;

size_t prune_workspace_bytes(uint32_t chunk_reads, uint32_t ref_len);
bool prune_applicable(const ScoringDev& s, uint32_t max_len, uint32_t ref_len, uint32_t limit);
hipError_t launch_score_pruned(const ScoreArgsV2& a2, uint32_t floor_strip, uint32_t floor_window, const ScoringDev& h_sc, uint8_t* work,
                               size_t work_bytes, uint32_t chunk_reads, uint32_t* fail_list, uint32_t* fail_count,
                               int mode /* as launch_score */, hipStream_t stream);

}  // namespace zsw
