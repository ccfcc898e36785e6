// zsw_align.hpp — entry points of the alignment (traceback) kernels (zsw_align.hip).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

// Pass 2 for one <N lanes, nv vectors> group of reads (b.items lists them).
hipError_t align_pass2(int N, uint32_t nv, const BatchDev& b, const uint8_t* d_ref, uint32_t ref_len, const ScoringDev* d_sc,
                       int S, const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W,
                       uint32_t maxc, uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item,
                       uint64_t* d_cig_start, uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list,
                       uint32_t* d_fb_count, int invert, hipStream_t stream, const uint32_t* d_safe_row = nullptr);
size_t align_ring_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid);
size_t align_lds_need(uint32_t nv, int S);
// count_only: per-block sums + total of n_ciglets; otherwise the packed write (and range inversion).
hipError_t align_finalize(zsw_alignment* d_aln, const uint8_t* d_status, uint32_t n, uint64_t* d_block_sums,
                          uint64_t* d_total, const uint64_t* d_cig_start, const uint32_t* d_cig_raw, int invert,
                          uint32_t* out_inc, uint8_t* out_op, uint64_t cap, bool count_only, hipStream_t stream);

// The packed kernel (zsw_align_pk_kernel.hpp): two reads per lane group in 16-bit halves. It takes groups of 8..64 lanes and
// up to 32 vectors whose striped profile fits LDS, and reads whose score stays below 2^15 with head-room (signed packed max).
constexpr uint32_t ALIGN_PK_MAX_SCORE = 30000;
bool align_pk_supported(int N, uint32_t nv, int S);
uint32_t align_pk_grid(int N, uint32_t nv, int S, uint32_t count, uint32_t cu_count);  // blocks that fill the chip once
size_t align_pk_ring_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid);
hipError_t align_pass2_pk(int N, uint32_t nv, const BatchDev& b, const uint8_t* d_ref, uint32_t ref_len, const ScoringDev* d_sc,
                          int S, const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W,
                          uint32_t maxc, uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item,
                          uint64_t* d_cig_start, uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list,
                          uint32_t* d_fb_count, int invert, hipStream_t stream, const uint32_t* d_safe_row = nullptr);

// Device-side grouping (zsw_group.hip): read ids sorted by (N, packed/wide, nv, ref_end) + the table of group starts.
size_t group_temp_bytes(uint32_t n);
hipError_t group_reads(const BatchDev& b, const uint8_t* d_status, const uint8_t* d_tier, const uint32_t* d_ref_end, const uint32_t* d_score,
                       const uint32_t* d_safe_row /* may be null */, int lanes_w8, int lanes_w16, int lanes_w32, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* items_out,
                       void* temp, size_t temp_bytes, uint32_t* table, uint32_t* table_count, uint32_t cap, hipStream_t stream);

// Third pass of sw_align_3pass (zsw_threepass.hip). list == null: classify pass over all reads (resolves the no-gaps
// shortcut, queues the rest in dp_list); otherwise the DP pass over `list`.
struct ThreePassArgs {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    const uint32_t* score;  // ranges results, indexed by read id
    const uint32_t *rs, *re, *qs, *qe;
    const uint8_t* status;
    const uint32_t* list;  // read ids that need the DP (class 2), or null: classify pass over all reads
    const uint32_t* list_count;
    uint32_t* dp_list;     // classify pass: reads that need the DP
    uint32_t* dp_count;
    uint32_t* dp_need_max;  // classify pass: largest slot need (bytes) among them
    uint8_t* scratch;       // DP pass: slots * slot_bytes
    uint32_t slots;
    uint64_t slot_bytes;
    uint32_t* cig;          // ciglet pool (traceback order), maxc per read
    uint32_t maxc;
    uint64_t pool_base;
    int by_item;
    uint64_t* cig_start;
    uint32_t* cig_raw;
    zsw_alignment* aln;
    uint32_t* fb_list;      // ciglet overflow / slot too small -> rerun with larger resources
    uint32_t* fb_count;
    int invert;
    // The shared-profile role (profile_set.rs:552-560: one profile, built from `pseq`, reused for every sequence of the batch):
    // non-null = the roles of three_pass.rs:21-26 are `reference` = read i, `query` = pseq for every i (ranges from
    // zsw_score_ranges_shared_batch: rs/re index the read, qs/qe the profile sequence); `ref` / `ref_len` are then unused.
    const uint8_t* pseq = nullptr;
    uint32_t pseq_len = 0;
    // Certificate mode of the classify pass (sw_simd_align's second pass skipped, zsw_capi.hip run_align; host model
    // tests/models/align_gapless_cert.cpp): cert_ok[i] = both maxima of read i sit in one cell each (forward and reversed seeded
    // pass, mode 3). A read whose ranges have equal lengths n, whose diagonal adds up to its score S and whose S exceeds
    // cert_maxw * (n - 1) - 3 * cert_go, and whose two-run alternatives stay below it (classify pass), has exactly one optimal alignment, the diagonal: it is written here (cert_done[i] = 1) and
    // is what sw_simd_align returns at every <T, N>; every other read is left to the literal striped kernel (cert_done[i] = 0).
    const uint8_t* cert_ok = nullptr;
    uint8_t* cert_done = nullptr;
    int cert_maxw = 0, cert_go = 0, cert_ge = 0;  // (the one-gap certificate, tests/models/align_onegap_cert.cpp, needs gap_extend too)
    // The sweeps over the two-run alternatives cost a read several times the rest of the certificate, and one read in seven needs
    // them: with one thread per read the other lanes of its wavefront would wait. The classify launch therefore lists those reads
    // (sweep_list / sweep_count) and a second launch — list = that list, sweep_pass = true: classify code, sweeps included — takes
    // them, every lane busy.
    uint32_t* sweep_list = nullptr;
    uint32_t* sweep_count = nullptr;
    bool sweep_pass = false;
};

hipError_t launch_threepass(const ThreePassArgs& a, uint32_t grid, hipStream_t stream);

}  // namespace zsw
