// zsw_score_prune.hpp — launcher of the column-pruned first pass (zsw_score_prune.hip).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

// Read-length classes: the strip's columns CP see every reference row; the window kernel holds the other columns in G lanes x C
// columns per read pair. A longer read loses more score to its mismatches, so its strip is wider: the checks compare the junk
// that leaves the strip (~20-30) with maxw * CP minus those losses.
struct PruneClass {
    int cp, g, c;
    uint32_t max_len;  // cp + g * c
};
constexpr int PR_N_CLASSES = 3;
constexpr PruneClass kPruneClasses[PR_N_CLASSES] = {{24, 4, 32, 152}, {48, 8, 32, 304}, {48, 16, 22, 400}};
constexpr int PR_BLK = 32;  // rows per block of the boundary maxima (windows are whole blocks)
constexpr int PR_M1 = 8;    // window rows kept above the anchor
constexpr int PR_M2 = 16;   // and below the row where an alignment without deletions ends (+8 per 8 lanes of the class)
constexpr uint32_t PR_MIN_READS = 96u << 10;   // smaller batches cannot fill the chip with one pair per lane: the full pass is faster
                                               // (150 bp vs 2 kb: 65,536 reads 2.7 ms pruned / 2.4 ms full; 100,000 reads 2.9 / 4.4 ms)
constexpr uint32_t PR_BAIL_PERCENT = 70;       // handed back by the first chip-full of a large batch: above this the rest takes the full pass
constexpr uint32_t PR_CHUNK_READS = 2u << 20;  // reads per round of the two kernels, at most
constexpr size_t PR_WORK_BYTES = size_t(32) << 30;  // and as many as this much boundary stream holds (8 B per pair and reference row)

struct ScoreArgsV2;

size_t prune_workspace_bytes(uint32_t chunk_reads, uint32_t ref_len);
uint32_t prune_chunk_reads(uint32_t n_reads, uint32_t ref_len);
int prune_class_for(uint32_t max_len);
bool prune_applicable(const ScoringDev& s, uint32_t max_len, uint32_t ref_len, uint32_t limit);
hipError_t launch_score_pruned(const ScoreArgsV2& a2, int cls, uint32_t n_cls, uint32_t floor_strip, const uint32_t* floor_window,
                               const ScoringDev& h_sc, uint8_t* work, size_t work_bytes, uint32_t chunk_reads, uint32_t* fail_list,
                               uint32_t* fail_count, int mode /* as launch_score */, bool wide /* a2 carries the WIDE table */,
                               hipStream_t stream, uint32_t* est_failed = nullptr /* out: the handed-back reads to expect, from the
                               bail-out probe (0xffffffff: no probe ran) */);

}  // namespace zsw
