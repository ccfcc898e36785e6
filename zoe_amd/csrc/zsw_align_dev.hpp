// zsw_align_dev.hpp — device-side pieces shared by the alignment kernels (zsw_align.hip: 32-bit lanes, one read per lane
// group; zsw_align_pk*.hip: packed 16-bit lanes, two reads per lane group): launch arguments, the late-start bound and the
// traceback walk (backtrack.rs:290-342).
#pragma once
#include "zsw_align.hpp"

namespace zsw {

constexpr uint8_t BT_UP = 1, BT_UP_EXT = 2, BT_LEFT = 4, BT_LEFT_EXT = 8, BT_STOP = 16;  // backtrack.rs:18-34

struct AlignArgs {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    const uint32_t* score;    // pass-1 results, indexed by read id
    const uint32_t* ref_end;  // exclusive end (r_end + 1)
    const uint8_t* status;
    uint32_t nv, W, maxc;
    uint8_t* ring;           // [gridDim.x][64/N][W][nv][N] flag bytes
    uint32_t* cig;           // ciglet pool of this launch: (inc << 8 | op) in traceback order, maxc slots per read
    uint64_t pool_base;      // first pool slot of this launch
    int by_item;             // slot = pool_base + (by_item ? item : read id) * maxc
    uint64_t* cig_start;     // per read: device address of its first slot (pools differ between launches)
    uint32_t* cig_raw;       // per read: ciglets written to the pool (traceback order)
    zsw_alignment* aln;      // per read; n_ciglets = count after the optional inversion
    uint32_t* fb_list;
    uint32_t* fb_count;
    int invert;
    uint8_t* rows;  // GLOBAL_ROWS kernels: per-block H/E/profile/flag rows in HBM (null otherwise)
    uint32_t* next_item;  // packed kernel: work counter (zeroed before the launch); a wavefront takes its next reads from it
    // optional, per read: the row from which the recompute may start with a zero state (seed_safe_start, zsw_seed.hpp: the
    // first pass's k-mer certificate), 0xffffffff = none: the warm-up bound below applies
    const uint32_t* safe_row;
    // SHARED kernels (one profile, many sequences: zsw_shared.hpp): the profile is striped over prof_seq for every item and the
    // rows are the bases of read i; null otherwise
    const uint8_t* prof_seq;
    uint32_t prof_len;
};

// Late start of pass 2. The state after row r (the H and E rows) is a (max,+) function of earlier rows in which every
// positive term is the score of an alignment path; a path that spans `span` reference rows has at most L diagonal steps and
// at least span - L vertical gap steps, so its score is <= L*maxw - (span - L)*gap_extend and it is positive only while
// span < L + L*maxw/gap_extend. Rows older than that bound cannot influence the rows whose flags are kept, so the recompute
// may start that many rows before the first retained row with a zero state and still be bit-identical (gap_extend = 0: no
// bound, start at row 0).
__device__ __forceinline__ int max_weight(const int32_t* w, int S) {
    int maxw = 0;
    for (int i = 0; i < S * S; ++i) maxw = max(maxw, w[i]);
    return maxw;
}

__device__ __forceinline__ int warmup_rows(const int32_t* w, int S, int ge, int l_pad) {
    if (ge <= 0) return 0x3fffffff;
    const int maxw = max_weight(w, S);
    const long long b = (long long)l_pad + ((long long)l_pad * maxw) / ge + 2;
    return b > 0x3fffffff ? 0x3fffffff : (int)b;
}

// Rows of flags a read's traceback can visit, from its score. An alignment with d <= L diagonal steps and g > 0 deleted
// reference bases scores at most d*maxw - gap_open - (g - 1)*gap_extend, so a read that scored `score` has
// g <= (L*maxw - score - gap_open)/gap_extend + 1 and its walk spans at most L + g rows (+2: the terminating cell and the
// row the walk looks at before it stops). A read with few edits keeps ~L rows of flags instead of the launch's window W.
// The walk still checks the window it was given; a walk that leaves it falls back to the full-window rerun.
__device__ __forceinline__ int flag_rows_needed(int W, int len, int maxw, int32_t score, int go, int ge) {
    if (ge <= 0) return W;
    long long g = ((long long)len * maxw - (long long)score - go) / ge + 1;
    if (g < 0) g = 0;
    const long long need = (long long)len + g + 2;
    return need < W ? (int)need : W;
}

// bytes of one block's DP rows in the generic kernel: H, E (i32), residue codes and flags (u8) per vector and lane
__host__ __device__ inline size_t align_rows_bytes(uint32_t nv) { return (size_t)nv * 64 * (4 + 4 + 1 + 1); }
constexpr size_t ALIGN_LDS_LIMIT = 160 * 1024 - 9 * 1024;

// max(a - b, 0) for non-negative a, b: one v_sub_u32 with clamp
__device__ __forceinline__ int32_t subsat(int32_t a, int32_t b) {
    return (int32_t)__builtin_elementwise_sub_sat((uint32_t)a, (uint32_t)b);
}

__device__ __forceinline__ uint32_t read_len(const BatchDev& b, uint32_t id, uint64_t* off) {
    if (b.offsets) {
        *off = b.offsets[id];
        return (uint32_t)(b.offsets[id + 1] - *off);
    }
    *off = (uint64_t)id * b.fixed_len;
    return b.fixed_len;
}


// BackTrackable::to_alignment (backtrack.rs:290-342) with AlignmentStates::add_ciglet merging (state.rs:142-152),
// run by one lane per read. `cell(r, c)` returns the flag byte of DP cell (r, c) from the retained window.
template <typename CellFn>
__device__ __forceinline__ void traceback_emit(const AlignArgs& a, uint32_t id, uint32_t item, uint32_t len, int rend, int cend,
                                               int32_t best, CellFn cell, int window, uint32_t ref_len) {
    const uint64_t slot0 = a.pool_base + (uint64_t)(a.by_item ? item : id) * a.maxc;
    uint32_t* cig = a.cig + slot0;
    uint32_t ncig = 0, cur_op = 0, cur_inc = 0;
    bool overflow = cend == 0x7fffffff;
    auto push = [&](uint32_t inc, uint32_t op) {
        if (inc == 0) return;
        if (cur_inc && cur_op == op) {
            cur_inc += inc;
            return;
        }
        if (cur_inc) {
            if (ncig < a.maxc) cig[ncig] = (cur_inc << 8) | cur_op;
            else overflow = true;
            ++ncig;
        }
        cur_op = op;
        cur_inc = inc;
    };
    const int r_end1 = rend + 1, c_end1 = cend + 1;
    int r = r_end1, c = c_end1;
    uint32_t n_nons = 0;  // ciglets that are not soft clips (for the inverted count)
    if (!overflow) {
        push(len - (uint32_t)c, 'S');
        uint32_t f = cell(rend, cend);
        uint32_t op = 0;
        while (!(f & BT_STOP) && r > 0 && c > 0) {
            if (op == 'D' && (f & BT_UP_EXT)) {
                r -= 1;
            } else if (op == 'I' && (f & BT_LEFT_EXT)) {
                c -= 1;
            } else if (f & BT_UP) {
                op = 'D';
                r -= 1;
            } else if (f & BT_LEFT) {
                op = 'I';
                c -= 1;
            } else {
                op = 'M';
                r -= 1;
                c -= 1;
            }
            if (!(cur_inc && cur_op == op)) ++n_nons;
            push(1, op);
            if (r > 0 && c > 0) {
                if (r - 1 + window <= rend) {  // the walk left the retained window
                    overflow = true;
                    break;
                }
                f = cell(r - 1, c - 1);
            }
        }
        push((uint32_t)c, 'S');
        push(1, 0);  // flush the pending ciglet (the sentinel op 0 itself is never stored)
    }
    if (overflow || ncig > a.maxc) {
        const uint32_t k = atomicAdd(a.fb_count, 1u);
        a.fb_list[k] = id;
    } else {
        zsw_alignment out;
        out.score = (uint32_t)best;
        out.ref_start = (uint32_t)r;
        out.ref_end = (uint32_t)r_end1;
        out.query_start = (uint32_t)c;
        out.query_end = (uint32_t)c_end1;
        out.ref_len = ref_len;
        out.query_len = len;
        // forward count; inverted: clips re-derived from ref_range (output.rs:399-414)
        out.n_ciglets = a.invert ? n_nons + (r > 0 ? 1u : 0u) + (ref_len > (uint32_t)r_end1 ? 1u : 0u) : ncig;
        out.ciglet_offset = 0;  // filled by write_ciglets_kernel
        a.aln[id] = out;
        a.cig_start[id] = (uint64_t)(uintptr_t)cig;
        a.cig_raw[id] = ncig;
    }
}

}  // namespace zsw
