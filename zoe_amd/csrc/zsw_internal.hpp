// zsw_internal.hpp — shared declarations of the gfx950 striped-SW library (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/zoe_sw.h"

namespace zsw {

constexpr int MAX_S = 32;
constexpr uint32_t MIN2 = 0x80008000u;  // two packed i16 at T::MIN (= true score 0)

// How a true Smith-Waterman score s becomes MaybeAligned<u32> for the requested instantiation
// (score_to_maybe_aligned, striped.rs:610-633, composed over the tiers of or_else_overflowed,
// profile_set.rs:71-107). Tier k answers iff s < thr[k].
struct ResultRule {
    uint64_t thr[3];
    uint8_t tier_code[3];
    int n_tiers;
};

__host__ __device__ inline void apply_rule(const ResultRule& r, uint64_t s, uint32_t* score, uint8_t* status,
                                           uint8_t* tier) {
    for (int k = 0; k < r.n_tiers; ++k) {
        if (s < r.thr[k]) {
            *score = (uint32_t)s;
            *status = s == 0 ? ZSW_STATUS_UNMAPPED : ZSW_STATUS_SOME;
            if (s == 0) *score = 0;
            *tier = r.tier_code[k];
            return;
        }
    }
    *score = 0;
    *status = ZSW_STATUS_OVERFLOWED;
    *tier = r.tier_code[r.n_tiers - 1];
}

// Device-resident scoring tables derived from WeightMatrix + ByteIndexMap.
struct ScoringDev {
    uint8_t index_map[256];
    int32_t w[MAX_S * MAX_S];  // signed weights, row = reference residue
    int32_t S;
    int32_t gap_open, gap_extend;  // positive magnitudes (StripedProfile stores them negated, profile.rs:300-301)
};

// Inputs of one batched launch.
struct BatchDev {
    const uint8_t* bases;
    const uint64_t* offsets;  // may be null
    uint32_t fixed_len;
    uint32_t n_reads;
    const uint32_t* items;  // optional indirection: item j -> read index (bucketed launches); null = identity
    uint32_t n_items;
};

struct ScoreOut {
    uint32_t* score;
    uint8_t* status;
    uint8_t* tier;      // may be null
    uint32_t* ref_end;  // ends kernels only (may be null)
    uint32_t* query_end;
    uint32_t* fb_list;  // reads that need the exact 32-bit kernel
    uint32_t* fb_count;
    // optional (alignment's first pass): per read, the row from which the second pass may start with a zero state
    // (seed_safe_start, zsw_seed.hpp); the caller presets 0xffffffff = no certificate, the seeded window kernel fills the rest
    uint32_t* safe_row = nullptr;
    // optional (mode 3, the banded seeded pass only): 1 where the read's maximum sits in exactly one cell of its matrix — the ends
    // then do not depend on the tie rule (zsw_capi_shared.hip; the reverse pass of sw_simd_score_ranges as a second seeded pass).
    // The caller presets 0; kernels that do not know leave it.
    uint8_t* unique = nullptr;
    // the seeded pass does not score the reads it hands back (the caller computes every read without `unique` by other means)
    bool skip_handed_back = false;
    // the banded seeded pass stops after its first (narrow) tier: what fails there is handed back at once — for a caller whose own
    // way of computing those reads costs no more than the second tier would (the exact reverse kernel of sw_simd_score_ranges)
    bool narrow_only = false;
    // the seeded pass reads every read back to front (the reverse seeded pass of sw_simd_score_ranges: no reversed copy of the batch);
    // only with skip_handed_back — no other kernel knows
    bool reads_reversed = false;
};

struct KernelTimer;

// Side streams on which the length classes of a ragged batch run concurrently (forked from / joined to the caller's stream).
struct SideStreams {
    static constexpr int N = 8;
    hipStream_t s[N];
    hipEvent_t fork, join[N];
};

// launchers (zsw_score.hip)
struct ScoreWorkspace {
    int32_t* scratch;       // exact32 kernel rows: 2 * slots * scratch_len ints
    size_t slots;
    uint32_t scratch_len;
    uint32_t* bucket_items;  // n_reads entries (ragged batches: reads grouped by strip configuration)
    uint32_t* bucket_counts; // 32 counters + 32 cursors
    // reads longer than the widest strip configuration are scored tile by tile (zsw_score_v2.hpp, TILED): two row-boundary buffers
    // of tile_bytes / 2 each and one running (score, ref_end, query_end) per read. All null when the batch has no such read.
    uint2* tile_buf = nullptr;
    size_t tile_bytes = 0;
    uint4* tile_state = nullptr;
    SideStreams* side = nullptr;  // null: the length classes run one after the other on the caller's stream
    // column-pruned score-only pass (zsw_score_prune.hip): workspace of prune_workspace_bytes(prune_chunk, ref_len) and a
    // worklist of n_reads entries + counter for the reads it hands back; null when the batch does not qualify
    uint8_t* prune_work = nullptr;
    size_t prune_bytes = 0;
    uint32_t prune_chunk = 0;
    uint32_t* prune_fail_list = nullptr;
    uint32_t* prune_fail_count = nullptr;
    // seeded exact pass (zsw_score_seed.hip): the context's reference index and seed_workspace_bytes(n_reads) + slack of workspace;
    // shares the worklist above. null when the batch does not qualify
    const struct SeedIndex* seed = nullptr;
    uint8_t* seed_work = nullptr;
    size_t seed_bytes = 0;
    KernelTimer* window_timer = nullptr;  // events around the window kernel of the seeded pass (bench.py's roofline)
    uint2* seed_gtab = nullptr;   // 2 x (ref_len + 2 * SEED_GTAB_PAD) entries: the per-row score table for blocks without an LDS table, then
                                  // the same with every score doubled (the banded kernel's domain, zsw_score_band.hip)
    int32_t* band_dbg = nullptr;  // zsw_debug_band_records
    unsigned long long* chunk_keys = nullptr;  // one per read: row-chunked full pass over the reads the seeded pass hands back (long references)
    uint32_t debug = 0;           // ZSW_DEBUG_* bits of the context (zsw_debug_set) | its options: kernel-selection overrides
};

hipError_t launch_score(const ScoringDev* d_sc, const ScoringDev& h_sc, const BatchDev& b, uint32_t max_len,
                        const uint8_t* d_ref, uint32_t ref_len, const ResultRule& rule, const ScoreOut& out,
                        const ScoreWorkspace& ws, hipStream_t stream, KernelTimer* timer,
                        int mode /* 0 score, 1 +ref_end, 2 +both ends */);
// Reverse pass of sw_simd_score_ranges: per read, SW of reverse(read[..query_end]) against reverse(reference[..ref_end]);
// out.ref_end / out.query_end receive the inclusive starts. d_gtab: ref_len uint2 entries of scratch.
hipError_t launch_score_rev(const ScoringDev* d_sc, const ScoringDev& h_sc, const BatchDev& b, uint32_t max_len,
                            const uint8_t* d_ref, uint32_t ref_len, const ResultRule& rule, const ScoreOut& out,
                            const ScoreWorkspace& ws, const uint32_t* d_fwd_ref_end, const uint32_t* d_fwd_query_end,
                            const uint32_t* d_fwd_score, uint2* d_gtab, hipStream_t stream);
bool score_config_for(uint32_t max_len, int* G, int* C);

// zsw_filter.hip: sneaky_snake over (reference window, read) pairs
hipError_t launch_sneaky(const BatchDev& b, const uint8_t* d_ref, uint32_t R, const uint32_t* d_ref_start,
                         const uint32_t* d_ref_len, float threshold, uint8_t* d_out, hipStream_t stream);

}  // namespace zsw
