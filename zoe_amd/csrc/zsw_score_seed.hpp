// zsw_score_seed.hpp — the seeded exact score pass: declarations shared by zsw_score_seed.hip (seed kernel, reference index,
// launcher), zsw_score_seed_m{0,1,2}.hip (the window kernel per MODE) and zsw_score.hip / zsw_capi.hip (callers).
#pragma once
#include "zsw_internal.hpp"
#include "zsw_seed.hpp"

namespace zsw {

constexpr int SEED_M1 = 24;   // window rows above the first row of the anchor diagonal: M1 + len / 8 (gap_open + (rows - Dn) *
constexpr int SEED_M1_PER8 = 1;  // gap_extend is the price of a path that comes down to the anchor from above the window)
constexpr int SEED_M2 = 16;   // and below its last row (the read's own deletions)
constexpr int SEED_DN = 4;    // diagonals left of the anchor that count as near (the read's own insertions)
constexpr int SEED_TOL = 8;    // anchor vote tolerance
constexpr int SEED_DM = 4;     // banded pass: diagonals right of the anchor that count as near (the read's own deletions)
constexpr int SEED_WD = 9;     // banded pass: diagonals kept below the anchor, SEED_WD + len * SEED_WD_PER16 / 16 (18 for 150 bases;
constexpr int SEED_WD_PER16 = 1;  // coming back from beyond them takes insertions: the gap and the inserted columns' potential)
// A narrow first band for short reads (two tiers): most reads lose little to their own errors and prove their score inside
// 8 diagonals above and 6 below the anchor, walked in 16-column strips; the reads that cannot are walked again in 48-column strips
// — a third as many strip boundaries, each of which costs a diverged read's proof some slack (zsw_score_band.hip) — and a band of
// 16 + len/16 above and 8 + len/32 below (a strip is a rectangle: 48 columns widen the band by themselves), and only what fails
// there is scored over all its cells. A batch that takes one tier only (few reads, or reads of more than SEED_NARROW_MAX_LEN bases) walks the
// full band (SEED_M1 + len/8, SEED_WD + len/16).
constexpr int SEED_NARROW_WU = 8, SEED_NARROW_WU_PER16 = 0, SEED_NARROW_WD = 6, SEED_NARROW_WD_PER32 = 0;
constexpr uint32_t SEED_BAIL_RATIO = 40;
constexpr uint32_t SEED_BAIL_BELOW = 16384;  // 0.4 ms of second-tier latency = the full pass of 13,000 reads
constexpr int SEED_SECOND_WU = 16, SEED_SECOND_WU_PER16 = 1, SEED_SECOND_WD = 8, SEED_SECOND_WD_PER32 = 1;
constexpr uint32_t SEED_NARROW_MAX_LEN = 640;  // beyond: a read's own indels drift further than the narrow band is wide
constexpr uint32_t SEED_NARROW_MIN_READS = 200000;  // below: two more launches cost more than the narrower band saves (length classes of a ragged batch)
constexpr int SEED_BAND_SLACK = 32;           // banded pass: two reads share a lane if their anchors are at most this far apart
constexpr uint32_t SEED_BAND_MAX_GRID = 1024;  // banded pass: persistent blocks (each lane owns a boundary buffer in HBM)
constexpr uint32_t SEED_KEY_BIAS = 1u << 16;  // sort key = anchor diagonal + bias (reads of up to 65,535 bases)
constexpr uint32_t SEED_MAX_LEN = 2432;       // the widest strip configuration
constexpr uint32_t SEED_MIN_LEN = 24;
constexpr int SEED_GTAB_PAD = 128;            // neutral entries on either side of the per-row table (lane skew: up to 64 + 2 rows)
constexpr uint32_t SEED_MIN_READS = 1024;     // smaller batches: the launches of the seeded pass cost more than the cells they save

// The reference index of a context: rebuilt when the reference or the scoring changes.
struct SeedIndex {
    bool valid = false;     // params / table describe the context's current reference and matrix
    bool usable = false;    // ... and the seeded pass can prune with them
    SeedParams params{};
    uint32_t* d_table = nullptr;  // 2 * 4^K entries: (first position + 1, last position + 1) per k-mer
    size_t table_bytes = 0;
};

struct SeedWindowArgs {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    uint32_t wtab[9][2];
    uint32_t ge2, gd2, floor0, K;
    ResultRule rule;
    ScoreOut out;
    SeedParams sp;
    uint32_t first;           // first item of the range
    uint32_t n;               // items of the range
    const uint32_t* order;    // items of the range sorted by key
    const uint32_t* keys;     // [item of the range]: anchor diagonal + key_bias, or fail_key
    const uint32_t* info;     // [item of the range]: t_all | d_fa << 16 | d_bl << 24
    const uint32_t* masks;    // [item of the range]: SeedRead::bl_mask
    const uint2* gtab;        // the 8 table bytes of every reference row (index SEED_GTAB_PAD + row), neutral rows in the pads
    uint32_t key_bias, fail_key;
    uint32_t* fail_list;      // global read ids
    uint32_t* fail_count;
    bool reversed = false;    // column c of a read is its base len - 1 - c
};

// banded pass (zsw_score_band.hip): one read pair per lane, strips of query columns
struct SeedBandArgs {
    BatchDev b;
    uint32_t ref_len;
    uint32_t ge2, gd2, floor0;         // of the DOUBLED scoring (the tables of gtab hold 2 * (score + gap_extend))
    ResultRule rule;
    ScoreOut out;
    SeedParams sp;
    uint32_t n;
    const uint32_t* order;
    const uint32_t* keys;
    const uint32_t* info;        // t_all | d_fa << 16 | d_bl << 24
    const uint32_t* band_masks;  // fa_mask | fb_mask << 16
    const uint8_t* band_dfb;     // d_fb
    const uint32_t* codes;       // [item][cs]: the read as 4-bit residue codes, 15 = padding
    uint32_t cs;                 // dwords per read in `codes`
    const uint2* gtab;
    uint2* bnd;                  // [block][nb][BLOCK]: the strip boundary of each lane (true scores: H of the last column, outgoing F)
    uint32_t nb;
    uint32_t grid;                     // blocks to launch (each owns nb * BLOCK entries of bnd)
    int wu0, wu_per16, wd0, wd_per32;  // this launch's band: wu0 + len * wu_per16 / 16 diagonals above the anchor, wd0 + len * wd_per32 / 32 below
    const uint32_t* n_dev;             // non-null: the number of items in `order` (a device-side count, at most n)
    uint32_t* next_pair;               // work queue: the next pair to hand out (zeroed before the launch)
    uint8_t* retry;                    // non-null (first tier): a read whose bounds fail sets retry[its position in `order`] instead of joining the list
    // first tier: counts the reads it accepts; second tier: if the first accepted fewer than one read per SEED_BAIL_RATIO that arrive
    // here (a batch ten per cent or more away from the reference: the wider strips would prove a few per cent of them for more than
    // the full pass of those few costs), the launch hands its items back without walking them
    uint32_t* accepted = nullptr;
    bool bail_check = false;
    uint32_t bail_below = 0;  // ... or with fewer items than this: a handful of reads costs the second tier one wavefront's whole walk (0.4 ms)
                              // and the full pass 30 ns each
    int* dbg;                          // non-null (tests): 8 ints per read — the walk's own values (maximum, oa, ob) and its geometry
    uint32_t key_bias, fail_key;
    uint32_t* fail_list;
    uint32_t* fail_count;
};
uint32_t seed_band_rows(const SeedParams& p, uint32_t max_len);
uint32_t seed_band_grid(uint32_t n, uint32_t grid_cap);
size_t seed_band_buffer_bytes(const SeedParams& p, uint32_t n, uint32_t max_len, uint32_t grid_cap);
// rebase_rows / limit: of the doubled scoring's drift domain (a strip's rows must fit one drift period, twice the largest score the range)
bool seed_band_applicable(const SeedParams& p, uint32_t max_len, uint32_t rebase_rows, uint32_t limit);
// narrow_strips: 16 columns per strip (the first tier) instead of 32
hipError_t launch_seed_band(const SeedBandArgs& a, int mode, bool narrow_strips, hipStream_t stream);

struct ScoreArgsV2;

// bytes of workspace for a range of n items
// band_grid_cap: most blocks the banded kernel may launch for this range (each owns a boundary buffer). A ragged batch shares
// SEED_BAND_MAX_GRID among its length classes in proportion to their reads (seed_band_class_cap), so that the classes' regions
// together stay within seed_workspace_bytes(n_reads, max_len) + SEED_BAND_CLASS_SLACK regions of an empty range.
constexpr uint32_t SEED_BAND_CLASS_MIN_GRID = 64;
inline uint32_t seed_band_class_cap(uint32_t n_class, uint32_t n_total) {
    return SEED_BAND_CLASS_MIN_GRID + (uint32_t)((uint64_t)SEED_BAND_MAX_GRID * n_class / (n_total ? n_total : 1));
}
size_t seed_workspace_bytes(uint32_t n, uint32_t max_len, uint32_t band_grid_cap = SEED_BAND_MAX_GRID);
// can reads of up to max_len bases be seeded with this index, given the packed kernels' score limit?
bool seed_applicable(const SeedIndex& ix, uint32_t max_len, uint32_t ref_len, uint32_t limit);
// narrow_min_reads: score-only calls of at least this many short reads walk a narrow band first (SEED_NARROW_*).
// Seeds, sorts and runs the window kernel (strip configuration G x C of a2's tables) over the items of a2.b; reads without an
// anchor and reads whose bounds fail are appended to fail_list (count at fail_count, not reset here).
// gtab: (ref_len + 2 * SEED_GTAB_PAD) uint2, filled by seed_build_gtab (once per call of launch_score, from a2's tables).
hipError_t seed_build_gtab(const ScoreArgsV2& a2, uint2* gtab, hipStream_t stream);
// band_tabs: non-null = the banded kernel may run: drift constants of the DOUBLED scoring (ge2, gd2, floor0, K, limit), whose
// per-row table is gtab_band (seed_build_gtab with those tables); band_dbg: zsw_debug_band_records.
hipError_t launch_score_seeded(const ScoreArgsV2& a2, int G, int C, uint32_t max_len, const SeedIndex& ix, uint8_t* work, size_t work_bytes,
                               uint2* gtab, uint32_t* fail_list, uint32_t* fail_count, int mode, const ScoreArgsV2* band_tabs, const uint2* gtab_band, int32_t* band_dbg,
                               uint32_t narrow_min_reads, uint32_t band_grid_cap, hipStream_t stream, KernelTimer* window_timer, bool narrow_only = false, bool reads_reversed = false);
// (Re)builds the index for a reference given as residue indices on the host.
hipError_t seed_index_update(SeedIndex* ix, const ScoringDev& sc, const uint8_t* h_ref, size_t ref_len);
void seed_index_release(SeedIndex* ix);

hipError_t launch_seed_window_m0(const SeedWindowArgs& a, int G, int C, hipStream_t stream);
hipError_t launch_seed_window_m1(const SeedWindowArgs& a, int G, int C, hipStream_t stream);
hipError_t launch_seed_window_m2(const SeedWindowArgs& a, int G, int C, hipStream_t stream);

}  // namespace zsw
