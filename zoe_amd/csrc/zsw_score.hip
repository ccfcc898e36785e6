// zsw_score.hip — score / score+ends kernels for gfx950 (MI355X).
//
// What is computed: for every read (the profile sequence, "query") the Smith-Waterman affine-gap
// local score against the context's reference, i.e. the value sw_simd_score returns
// (reference: src/alignment/sw/striped.rs:65-142) and, in ENDS mode, the (ref_end, query_end) of
// sw_simd_score_ends (striped.rs:213-336: first row holding the maximum, then first column).
// Both are invariant to Zoe's SIMD lane count N (they equal the Gotoh recurrence that
// src/alignment/sw/scalar.rs:55-122 states), so this kernel is free to parallelise differently:
//
//   * inter-read: each 32-bit lane carries TWO reads as packed i16 (v_pk_add_i16/v_pk_sub_i16 with
//     clamp = Zoe's saturating_add/sub, v_pk_max_i16); scores live at offset i16::MIN exactly as in
//     Zoe's signed profiles, so saturation at MIN is the zero floor of local alignment;
//   * a group of G adjacent lanes owns one read pair; lane g keeps columns [g*C, (g+1)*C) of H and E
//     in VGPRs and walks down the reference one row per step, skewed by g steps (anti-diagonal
//     wavefront at strip granularity); the only cross-lane traffic is the strip's last H and F, one
//     wave shuffle each per step;
//   * the reference row's four substitution scores (8 bytes) come from an LDS table built once per
//     block; the per-column score is ONE v_perm_b32 of that row by a per-column selector register
//     that encodes the two reads' residues — no per-cell memory access at all.
//
// 10 VALU instructions per packed cell pair; no MFMA (this is not a contraction); HBM traffic is the
// read bytes in and 4-5 bytes out per read. The binding roof is integer VALU issue.
#include <stdlib.h>

#include <algorithm>

#include "zsw_internal.hpp"
#include "zsw_score_v1.hpp"
#include "zsw_score_prune.hpp"
#include "zsw_score_seed.hpp"
#include "zsw_score_v2.hpp"
#include "zsw_timer.hpp"

namespace zsw {

// Exact 32-bit kernel: any alphabet size, any read length, no saturation below 2^31. One thread per
// read, H/E rows in global scratch ([column][slot], coalesced over threads). Used for reads whose
// packed-i16 score saturated, for alphabets the table kernels do not cover (S > 7) and for reads
// longer than the largest strip configuration. Follows scalar.rs:55-122 (same recurrence, same
// strict-greater scan order = first row, then first column).
__global__ __launch_bounds__(64) void exact32_kernel(BatchDev b, const uint32_t* list, const uint32_t* list_count,
                                                     const uint8_t* ref, uint32_t ref_len, const ScoringDev* sc,
                                                     ResultRule rule, ScoreOut out, int32_t* scratch, uint32_t slots,
                                                     uint32_t scratch_len, const uint32_t* rev_ref_end,
                                                     const uint32_t* rev_query_end) {
    __shared__ uint8_t lut[256];
    __shared__ int32_t w[MAX_S * MAX_S];
    for (int i = threadIdx.x; i < 256; i += 64) lut[i] = sc->index_map[i];
    for (int i = threadIdx.x; i < MAX_S * MAX_S; i += 64) w[i] = sc->w[i];
    __syncthreads();
    const int S = sc->S;
    const int go = sc->gap_open, ge = sc->gap_extend;
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    const uint32_t n = list ? *list_count : b.n_items;
    int32_t* Hrow = scratch + slot;
    int32_t* Erow = scratch + (size_t)slots * scratch_len + slot;
    for (uint32_t item = slot; item < n; item += slots) {
        const uint32_t id = list ? list[item] : (b.items ? b.items[item] : item);
        uint64_t off;
        uint32_t len;
        if (b.offsets) {
            off = b.offsets[id];
            len = (uint32_t)(b.offsets[id + 1] - off);
        } else {
            off = (uint64_t)id * b.fixed_len;
            len = b.fixed_len;
        }
        uint32_t rows = ref_len;
        if (rev_ref_end) {  // reverse pass of sw_simd_score_ranges: reversed prefixes of both sequences
            len = rev_query_end[id] <= len ? rev_query_end[id] : len;
            rows = len ? rev_ref_end[id] : 0;
        }
        if (len == 0 || len > scratch_len) {
            out.score[id] = 0;
            out.status[id] = len == 0 ? ZSW_STATUS_EMPTY : ZSW_STATUS_OVERFLOWED;
            if (out.tier) out.tier[id] = 0;
            if (out.ref_end) out.ref_end[id] = 0;
            if (out.query_end) out.query_end[id] = 0;
            continue;
        }
        for (uint32_t c = 0; c < len; ++c) {
            Hrow[(size_t)c * slots] = 0;
            Erow[(size_t)c * slots] = 0;
        }
        int64_t best = 0;
        uint32_t r_end = 0, c_end = 0;
        for (uint32_t r = 0; r < rows; ++r) {
            const int32_t* wr = &w[lut[ref[rev_ref_end ? rows - 1 - r : r]] * S];
            int32_t f = 0, diag = 0;
            for (uint32_t c = 0; c < len; ++c) {
                const int32_t up = Hrow[(size_t)c * slots];
                int32_t e = Erow[(size_t)c * slots];
                int32_t h = diag + wr[lut[b.bases[rev_ref_end ? off + (len - 1 - c) : off + c]]];
                h = max(max(h, e), max(f, 0));
                if (h > best) {
                    best = h;
                    r_end = r;
                    c_end = c;
                }
                Hrow[(size_t)c * slots] = h;
                e = max(max(e - ge, h - go), 0);
                f = max(max(f - ge, h - go), 0);
                Erow[(size_t)c * slots] = e;
                diag = up;
            }
        }
        uint32_t score;
        uint8_t status, tier;
        apply_rule(rule, (uint64_t)best, &score, &status, &tier);
        out.score[id] = score;
        out.status[id] = status;
        if (out.tier) out.tier[id] = tier;
        const bool some = status == ZSW_STATUS_SOME;
        if (rev_ref_end) {  // inclusive starts
            out.ref_end[id] = some ? rows - (r_end + 1) : 0;
            out.query_end[id] = some ? len - (c_end + 1) : 0;
        } else {
            if (out.ref_end) out.ref_end[id] = some ? r_end + 1 : 0;
            if (out.query_end) out.query_end[id] = some ? c_end + 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct Cfg {
    int G, C;
};
// strip configurations, ascending capacity G*C
#define ZSW_CFG_ENTRY(GV, CV) {GV, CV},
static const Cfg kCfgs[] = {ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CFG_ENTRY)};
#undef ZSW_CFG_ENTRY

constexpr int N_ASCENDING_CFGS = 20;  // the rest of kCfgs is for small batches only (score_config_for_batch)

bool score_config_for(uint32_t max_len, int* G, int* C) {
    for (int k = 0; k < N_ASCENDING_CFGS; ++k) {
        const Cfg& c = kCfgs[k];
        if ((uint32_t)(c.G * c.C) >= max_len) {
            *G = c.G;
            *C = c.C;
            return true;
        }
    }
    return false;
}

// Same, for a batch of n_items reads: a small batch cannot fill 1024 SIMDs with the narrowest strips (10 k reads of 150 bp are
// 312 wavefronts at G = 4), so it takes more lanes per read and fewer columns per lane. The score does not depend on the choice.
// Cost model: a step issues about 7.5*C + 25 instructions; a launch is latency-bound up to ~2 waves per SIMD, throughput-bound above.
static bool score_config_for_batch(uint32_t max_len, uint32_t n_items, int* G, int* C) {
    double best = 0;
    bool found = false;
    for (const Cfg& c : kCfgs) {
        if ((uint32_t)(c.G * c.C) < max_len) continue;
        const double waves = ((double)((n_items + 1) / 2) * c.G) / 64.0;
        const double cost = (7.5 * c.C + 25.0) * std::max(2048.0, waves);
        if (!found || cost < best * 0.97) {  // ties go to the narrower configuration
            best = cost;
            *G = c.G;
            *C = c.C;
            found = true;
        }
    }
    return found;
}

template <int G, int C>
static hipError_t launch_cfg(const ScoreArgs& a, bool fast, int mode, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
#define ZSW_LAUNCH(FASTV, MODEV) hipLaunchKernelGGL((score_kernel<G, C, FASTV, MODEV>), dim3(grid), dim3(BLOCK), 0, stream, a)
    if (fast) {
        if (mode == 0) ZSW_LAUNCH(true, 0);
        else if (mode == 1) ZSW_LAUNCH(true, 1);
        else ZSW_LAUNCH(true, 2);
    } else {
        if (mode == 0) ZSW_LAUNCH(false, 0);
        else if (mode == 1) ZSW_LAUNCH(false, 1);
        else ZSW_LAUNCH(false, 2);
    }
#undef ZSW_LAUNCH
    return hipGetLastError();
}

// Chooses the table form. FAST needs every query residue code >= 4 to score 0 against every
// reference residue (true for S <= 4, and for DNA matrices built with `ignoring = Some(b'N')`).
static bool fast_ok(const ScoringDev& s) {
    if (s.S > 8) return false;
    for (int r = 0; r < s.S; ++r)
        for (int q = 4; q < s.S; ++q)
            if (s.w[r * s.S + q] != 0) return false;
    return true;
}

static void build_tables(const ScoringDev& s, bool fast, ScoreArgs* a) {
    int bias = 0;
    for (int i = 0; i < s.S * s.S; ++i) bias = s.w[i] < -bias ? -s.w[i] : bias;
    auto pk = [](uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); };
    for (int r = 0; r < 9; ++r) {
        uint32_t lo = 0, hi = 0;
        if (fast) {
            int v[4] = {0, 0, 0, 0};
            if (r < s.S)
                for (int q = 0; q < 4 && q < s.S; ++q) v[q] = s.w[r * s.S + q];
            lo = pk((uint32_t)v[0], (uint32_t)v[1]);
            hi = pk((uint32_t)v[2], (uint32_t)v[3]);
        } else {
            uint8_t by[8];
            for (int q = 0; q < 8; ++q) by[q] = (uint8_t)bias;
            if (r < s.S)
                for (int q = 0; q < s.S && q < 7; ++q) by[q] = (uint8_t)(s.w[r * s.S + q] + bias);
            lo = by[0] | (by[1] << 8) | (by[2] << 16) | ((uint32_t)by[3] << 24);
            hi = by[4] | (by[5] << 8) | (by[6] << 16) | ((uint32_t)by[7] << 24);
        }
        a->wtab[r][0] = lo;
        a->wtab[r][1] = hi;
    }
    a->go2 = pk((uint32_t)s.gap_open, (uint32_t)s.gap_open);
    a->ge2 = pk((uint32_t)s.gap_extend, (uint32_t)s.gap_extend);
    a->bias2 = pk((uint32_t)bias, (uint32_t)bias);
}

template <int G, int C>
static hipError_t launch_cfg_v2(const ScoreArgsV2& a, int mode, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
    const dim3 g(grid, a.chunk_rows ? (a.ref_len + a.chunk_rows - 1) / a.chunk_rows : 1u);  // row-chunked launches: one grid row per chunk
    if (mode == 0) hipLaunchKernelGGL((score_kernel_v2<G, C, 0>), g, dim3(BLOCK), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((score_kernel_v2<G, C, 1>), g, dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((score_kernel_v2<G, C, 2>), g, dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

// Row-chunked launches (ScoreArgsV2::chunk_rows): the keys of the listed reads before, their results after.
__global__ void chunk_zero_kernel(const uint32_t* items, const uint32_t* n_dev, uint32_t n_max, unsigned long long* keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < min(*n_dev, n_max)) keys[items[i]] = 0ull;
}
__global__ void chunk_finalize_kernel(BatchDev b, const uint32_t* n_dev, const unsigned long long* keys, uint32_t limit, ResultRule rule, ScoreOut out, int mode) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*n_dev, b.n_items)) return;
    const uint32_t id = b.items[i];
    const uint32_t len = b.offsets ? (uint32_t)(b.offsets[id + 1] - b.offsets[id]) : b.fixed_len;
    const unsigned long long key = keys[id];
    const uint32_t s = (uint32_t)(key >> (CHUNK_ROW_BITS + CHUNK_COL_BITS));
    const uint32_t row1 = ((1u << CHUNK_ROW_BITS) - 1u) - (uint32_t)((key >> CHUNK_COL_BITS) & ((1u << CHUNK_ROW_BITS) - 1u));
    const uint32_t col1 = ((1u << CHUNK_COL_BITS) - 1u) - (uint32_t)(key & ((1u << CHUNK_COL_BITS) - 1u));
    if (len == 0) {
        out.score[id] = 0;
        out.status[id] = ZSW_STATUS_EMPTY;
        if (out.tier) out.tier[id] = 0;
        if (mode != 0 && out.ref_end) out.ref_end[id] = 0;
        if (mode == 2 && out.query_end) out.query_end[id] = 0;
    } else if (s >= limit) {  // near the representable limit: recompute exactly in 32 bits
        out.fb_list[atomicAdd(out.fb_count, 1u)] = id;
    } else {
        uint32_t score;
        uint8_t status, tier;
        apply_rule(rule, (uint64_t)s, &score, &status, &tier);
        out.score[id] = score;
        out.status[id] = status;
        if (out.tier) out.tier[id] = tier;
        const bool some = status == ZSW_STATUS_SOME;
        if (mode != 0 && out.ref_end) out.ref_end[id] = some ? row1 : 0;
        if (mode == 2 && out.query_end) out.query_end[id] = some ? col1 : 0;
    }
}

static hipError_t launch_table_cfg_v2(const ScoreArgsV2& a, int G, int C, int mode, hipStream_t stream) {
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: return launch_cfg_v2<GV, CV>(a, mode, stream);
        ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CASE)
#undef ZSW_CASE
    }
    return hipErrorInvalidValue;
}

// v2 needs: S <= 7; for query residues 0..3: s + ge in [-128, 127]; for residues >= 4: s + ge in [0, 255].
static bool v2_ok(const ScoringDev& s, uint32_t debug) {
    if (debug & ZSW_DEBUG_SCORE_V1) return false;
    if (s.S > 7) return false;
    for (int r = 0; r < s.S; ++r)
        for (int q = 0; q < s.S; ++q) {
            const int t = s.w[r * s.S + q] + s.gap_extend;
            if (q < 4 ? (t < -128 || t > 127) : (t < 0 || t > 255)) return false;
        }
    return true;
}

// Drift-domain constants for strip width G; returns false if the drifted range does not leave room for scores.
static bool v2_range_setup(const ScoringDev& s, int G, ScoreArgsV2* a) {
    const int ge = s.gap_extend, go = s.gap_open;
    a->ge2 = (uint32_t)ge * 0x00010001u;
    a->gd2 = (uint32_t)(go - ge) * 0x00010001u;
    uint32_t K = 2048;
    while (K > 16 && K * (uint32_t)ge > 8192) K /= 2;
    a->K = K;
    a->floor0 = 1152u + (uint32_t)G * (uint32_t)ge + 128u;
    const uint32_t dmax = a->floor0 + (K + 2) * (uint32_t)ge;
    if (dmax + 1024 > 0x7C00u - 512u) return false;
    a->limit = 0x7C00u - 512u - dmax;
    return true;
}

// Fills the v2 tables for strip width G (alphabets of up to 7 letters).
static bool build_tables_v2(const ScoringDev& s, int G, ScoreArgsV2* a) {
    const int ge = s.gap_extend;
    for (int r = 0; r < 9; ++r) {
        uint8_t by[8];
        for (int q = 0; q < 8; ++q) by[q] = (uint8_t)ge;  // true 0 (+ge): padding, unused slots and the neutral row
        if (r < s.S)
            for (int q = 0; q < s.S; ++q) {
                const int t = s.w[r * s.S + q] + ge;
                if (q < 4) by[2 * q + 1] = (uint8_t)(int8_t)t;
                else by[2 * (q - 3)] = (uint8_t)t;
            }
        a->wtab[r][0] = by[0] | (by[1] << 8) | (by[2] << 16) | ((uint32_t)by[3] << 24);
        a->wtab[r][1] = by[4] | (by[5] << 8) | (by[6] << 16) | ((uint32_t)by[7] << 24);
    }
    return v2_range_setup(s, G, a);
}

// The banded seeded kernel's tables: every weight and gap penalty doubled (strip width 1: a lane owns whole strips). false when the
// doubled weights do not fit the table bytes.
static bool doubled_tables(const ScoringDev& s, ScoreArgsV2* a) {
    ScoringDev d = s;
    for (int i = 0; i < s.S * s.S; ++i) d.w[i] = 2 * s.w[i];
    d.gap_open = 2 * s.gap_open;
    d.gap_extend = 2 * s.gap_extend;
    return v2_ok(d, 0) && build_tables_v2(d, 1, a);
}

// WIDE kernels: 8..32 letters, every score + ge must fit a signed byte.
static bool wide_ok(const ScoringDev& s, uint32_t debug) {
    if (debug & ZSW_DEBUG_NO_WIDE) return false;
    if (s.S <= 7 || s.S > 32) return false;
    for (int i = 0; i < s.S * s.S; ++i) {
        const int t = s.w[i] + s.gap_extend;
        if (t < -128 || t > 127) return false;
    }
    return true;
}

static void fill_wide_table(const ScoringDev& s, ScoreArgsV2* a) {
    const int ge = s.gap_extend;
    for (int r = 0; r < 33; ++r)
        for (int q = 0; q < WIDE_STRIDE; ++q)
            a->wide[r * WIDE_STRIDE + q] = (int8_t)((r < s.S && q < s.S) ? s.w[r * s.S + q] + ge : ge);
}

static bool build_tables_wide(const ScoringDev& s, int G, ScoreArgsV2* a) {
    fill_wide_table(s, a);
    return v2_range_setup(s, G, a);
}

// The 32-bit tile kernel takes any alphabet of up to 32 letters whose weights + gap_extend fit a signed byte, as long as the
// drifted values stay inside an i32 for this reference and read length.
static bool w32_ok(const ScoringDev& s, uint32_t ref_len, uint32_t max_len, uint32_t debug) {
    if ((debug & ZSW_DEBUG_NO_W32) || s.S > 32) return false;
    int maxw = 0;
    for (int i = 0; i < s.S * s.S; ++i) {
        const int t = s.w[i] + s.gap_extend;
        if (t < -128 || t > 127) return false;
        maxw = std::max(maxw, (int)s.w[i]);
    }
    const uint64_t top = ((uint64_t)ref_len + TILE_G + 4) * (uint64_t)s.gap_extend + (uint64_t)max_len * (uint64_t)maxw;
    return top < 0x7f000000ull;
}

static hipError_t launch_table_cfg(const ScoreArgs& a, int G, int C, bool fast, int mode, hipStream_t stream) {
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: return launch_cfg<GV, CV>(a, fast, mode, stream);
        ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CASE)
#undef ZSW_CASE
    }
    return hipErrorInvalidValue;
}

// ---- ragged batches: group the reads by the smallest strip configuration that holds them -------------------
// class k < NCLS: kCfgs[kBucketCfg[k]]; class NCLS: longer than every table configuration (exact 32-bit kernel)
static const int kBucketCfg[] = {0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19};  // (8,19) duplicates the capacity of (4,38)
constexpr int NCLS = 19;

struct BucketCaps {
    uint32_t cap[NCLS];
};

__device__ __forceinline__ int bucket_of(const BucketCaps& caps, uint32_t len) {
    int k = 0;
    while (k < NCLS && len > caps.cap[k]) ++k;
    return k;
}

__global__ void reverse_bytes_kernel(const uint8_t* in, uint32_t n, uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[n - 1 - i];
}

__global__ void iota_kernel(uint32_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ void bucket_count_kernel(const uint64_t* offsets, uint32_t n, BucketCaps caps, uint32_t* counts) {
    __shared__ uint32_t sh[NCLS + 1];
    if (threadIdx.x <= NCLS) sh[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&sh[bucket_of(caps, (uint32_t)(offsets[i + 1] - offsets[i]))], 1u);
    __syncthreads();
    if (threadIdx.x <= NCLS && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sh[threadIdx.x]);
}

__global__ void bucket_scatter_kernel(const uint64_t* offsets, uint32_t n, BucketCaps caps, uint32_t* cursors, uint32_t* items) {
    // one atomic per wavefront and class (a million single atomics on twenty cursors took 3.2 ms)
    const int lane = threadIdx.x & 63;
    const uint32_t per_sweep = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += per_sweep) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        const int k = valid ? bucket_of(caps, (uint32_t)(offsets[i + 1] - offsets[i])) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int kl = __shfl(k, leader, 64);
            const unsigned long long same = __ballot(valid && k == kl);
            if (k == kl) {
                uint32_t start = 0;
                if (lane == leader) start = atomicAdd(&cursors[kl], (uint32_t)__popcll(same));
                start = (uint32_t)__shfl((int)start, leader, 64);
                items[start + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = i;
            }
            todo &= ~same;
        }
    }
}

__global__ void add_count_kernel(const uint32_t* count, uint32_t* total) { atomicAdd(total, *count); }

hipError_t launch_score(const ScoringDev* d_sc, const ScoringDev& h_sc, const BatchDev& b, uint32_t max_len,
                        const uint8_t* d_ref, uint32_t ref_len, const ResultRule& rule, const ScoreOut& out,
                        const ScoreWorkspace& ws, hipStream_t stream, KernelTimer* timer, int mode) {
    hipError_t e = hipMemsetAsync(out.fb_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    int G = 0, C = 0;
    const bool wide = wide_ok(h_sc, ws.debug);
    const bool table_ok = h_sc.S <= 7 || fast_ok(h_sc) || wide;
    const uint32_t exact_grid = (uint32_t)(ws.slots / 64);
    ScoreArgs a;
    a.b = b;
    a.ref = d_ref;
    a.ref_len = ref_len;
    a.sc = d_sc;
    a.rule = rule;
    a.out = out;
    a.rev_ref_end = nullptr;
    a.rev_query_end = nullptr;
    a.rev_score = nullptr;
    a.gtab = nullptr;
    const bool fast = fast_ok(h_sc);
    if (table_ok && !wide) build_tables(h_sc, fast, &a);
    const bool use_v2 = table_ok && !wide && v2_ok(h_sc, ws.debug);
    ScoreArgsV2 a2;
    a2.b = b;
    a2.ref = d_ref;
    a2.ref_len = ref_len;
    a2.sc = d_sc;
    a2.rule = rule;
    a2.out = out;
    a2.tile_q0 = 0;
    a2.tile_in = nullptr;
    a2.tile_out = nullptr;
    a2.tile_state = nullptr;
    a2.rev_ref_end = nullptr;
    a2.rev_query_end = nullptr;
    auto launch_one = [&](const BatchDev& bb, int g, int c) -> hipError_t {
        if (wide) {
            if (build_tables_wide(h_sc, g, &a2)) {
                a2.b = bb;
                return launch_table_cfg_v2_wide(a2, g, c, mode, stream);
            }
            hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, bb, (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr, d_ref, ref_len, d_sc, rule, out, ws.scratch, (uint32_t)ws.slots,
                               ws.scratch_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
            return hipGetLastError();
        }
        if (use_v2 && build_tables_v2(h_sc, g, &a2)) {
            a2.b = bb;
            return launch_table_cfg_v2(a2, g, c, mode, stream);
        }
        a.b = bb;
        return launch_table_cfg(a, g, c, fast, mode, stream);
    };
    // Reads longer than the widest strip configuration: TILE_COLS query columns per launch, the strip boundary of every
    // reference row handed from tile to tile through HBM, as many reads per round as the boundary buffers hold.
    // Returns hipErrorNotSupported when the packed kernels cannot take the batch (the caller then uses the exact kernel).
    auto launch_tiled = [&](const BatchDev& bb, uint32_t longest) -> hipError_t {
        if (!ws.tile_buf || !ws.tile_state || !bb.items || (ws.debug & ZSW_DEBUG_NO_TILES)) return hipErrorNotSupported;
        if (!(wide || use_v2)) return hipErrorNotSupported;
        if (!(wide ? build_tables_wide(h_sc, TILE_G, &a2) : build_tables_v2(h_sc, TILE_G, &a2))) return hipErrorNotSupported;
        const size_t per_pair = (size_t)ref_len * sizeof(uint2);
        const size_t half = ws.tile_bytes / 2;
        const size_t fit = per_pair ? half / per_pair : (size_t)bb.n_items;
        if (fit == 0) return hipErrorNotSupported;
        const uint32_t chunk = (uint32_t)std::min<size_t>(2 * fit, 0x7ffffffeu);
        uint2* buf[2] = {ws.tile_buf, reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(ws.tile_buf) + half)};
        const uint32_t n_tiles = (longest + TILE_COLS - 1) / TILE_COLS;
        for (uint32_t first = 0; first < bb.n_items; first += chunk) {
            a2.b = bb;
            a2.b.items = bb.items + first;
            a2.b.n_items = std::min<uint32_t>(chunk, bb.n_items - first);
            a2.tile_state = ws.tile_state;
            for (uint32_t t = 0; t < n_tiles; ++t) {
                a2.tile_q0 = t * (uint32_t)TILE_COLS;
                a2.tile_in = t ? buf[(t - 1) & 1] : nullptr;
                a2.tile_out = t + 1 < n_tiles ? buf[t & 1] : nullptr;
                hipError_t te = launch_tile_v2(a2, wide, mode, stream);
                if (te != hipSuccess) return te;
            }
        }
        return hipSuccess;
    };
    // Reads whose score is beyond the packed range sit in the device-side worklist (usually empty). Batches that can hold such
    // scores at all (long reads) read the count back and run the 32-bit tile kernel over the list; everything else launches the
    // exact kernel on the worklist unconditionally, which keeps short-read calls asynchronous.
    auto finish_worklist = [&]() -> hipError_t {
        int maxw = 1;
        for (int i = 0; i < h_sc.S * h_sc.S; ++i) maxw = std::max(maxw, (int)h_sc.w[i]);
        if ((uint64_t)max_len * (uint64_t)maxw >= 16384 && ws.tile_buf && ws.tile_state && w32_ok(h_sc, ref_len, max_len, ws.debug)) {
            uint32_t cnt = 0;
            hipError_t we = hipMemcpyAsync(&cnt, out.fb_count, sizeof(cnt), hipMemcpyDeviceToHost, stream);
            if (we == hipSuccess) we = hipStreamSynchronize(stream);
            if (we != hipSuccess) return we;
            if (cnt == 0) return hipSuccess;
            fill_wide_table(h_sc, &a2);
            a2.ge2 = (uint32_t)h_sc.gap_extend * 0x00010001u;
            a2.gd2 = (uint32_t)(h_sc.gap_open - h_sc.gap_extend) * 0x00010001u;
            const size_t per_read = (size_t)ref_len * sizeof(uint2), half = ws.tile_bytes / 2;
            const size_t fit = per_read ? half / per_read : (size_t)cnt;
            if (fit > 0) {
                uint2* buf[2] = {ws.tile_buf, reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(ws.tile_buf) + half)};
                const uint32_t n_tiles = (max_len + TILE_COLS - 1) / TILE_COLS;
                for (uint32_t first = 0; first < cnt; first += (uint32_t)std::min<size_t>(fit, 0x7fffffffu)) {
                    a2.b = b;
                    a2.b.n_items = (uint32_t)std::min<size_t>(fit, cnt - first);
                    a2.tile_state = ws.tile_state;
                    for (uint32_t t = 0; t < n_tiles; ++t) {
                        a2.tile_q0 = t * (uint32_t)TILE_COLS;
                        a2.tile_in = t ? buf[(t - 1) & 1] : nullptr;
                        a2.tile_out = t + 1 < n_tiles ? buf[t & 1] : nullptr;
                        we = launch_tile_w32(a2, out.fb_list + first, mode, stream);
                        if (we != hipSuccess) return we;
                    }
                }
                return hipSuccess;
            }
        }
        hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, out.fb_list, out.fb_count, d_ref, ref_len, d_sc,
                           rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        return hipGetLastError();
    };
    auto exact_all = [&](const BatchDev& bb) {
        hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, bb, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, d_ref, ref_len, d_sc, rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        return hipGetLastError();
    };
    // Seeded exact pass (zsw_score_seed.hip) over the items of `bb` in strip configuration (g, c): seed + sort + window kernel, then
    // score_kernel_v2 over the reads it hands back (device-side list; `counter` = its count, zeroed here). The workspace region
    // [work_off, work_off + seed_workspace_bytes(n_items)) and fail_list + list_off belong to this call alone.
    bool gtab_built = false;
    auto seed_items = [&](const BatchDev& bb, int g, int c, uint32_t longest, size_t work_off, uint32_t list_off, uint32_t* counter,
                          uint32_t band_grid_cap) -> hipError_t {
        if (!use_v2 || !ws.seed || !ws.seed_work || !ws.prune_fail_list || !(ws.debug & ZSW_DEBUG_SCORE_PRUNE) || (ws.debug & ZSW_DEBUG_PRUNE_STRIP))
            return hipErrorNotSupported;
        if (bb.n_items < SEED_MIN_READS && !(ws.debug & ZSW_DEBUG_SCORE_PRUNE_ANY_SIZE)) return hipErrorNotSupported;
        if (g == 32 || work_off + seed_workspace_bytes(bb.n_items, longest, band_grid_cap) > ws.seed_bytes) return hipErrorNotSupported;
        ScoreArgsV2 ap = a2;
        if (!build_tables_v2(h_sc, g, &ap) || !seed_applicable(*ws.seed, longest, ref_len, ap.limit)) return hipErrorNotSupported;
        if (!ws.seed_gtab) return hipErrorNotSupported;
        hipError_t pe = hipSuccess;
        // the banded kernel works on doubled scores (zsw_seed.hpp: an odd value marks a path through a cell outside the band): its own
        // drift constants and per-row table, if the doubled weights fit the table bytes
        ScoreArgsV2 ab = a2;
        const bool band_ok = !(ws.debug & ZSW_DEBUG_SEED_NO_BAND) && doubled_tables(h_sc, &ab);
        uint2* const gtab_band = ws.seed_gtab + (ref_len + 2 * SEED_GTAB_PAD);
        if (!gtab_built) {  // the per-row score table, for blocks whose windows do not fit one LDS table (same bytes for every g)
            pe = seed_build_gtab(ap, ws.seed_gtab, stream);
            if (pe != hipSuccess) return pe;
            if (band_ok) {
                pe = seed_build_gtab(ab, gtab_band, stream);
                if (pe != hipSuccess) return pe;
            }
            gtab_built = true;
        }
        pe = hipMemsetAsync(counter, 0, 4, stream);
        if (pe != hipSuccess) return pe;
        ap.b = bb;
        pe = launch_score_seeded(ap, g, c, longest, *ws.seed, ws.seed_work + work_off, seed_workspace_bytes(bb.n_items, longest, band_grid_cap), ws.seed_gtab,
                                 ws.prune_fail_list + list_off, counter, mode, band_ok ? &ab : nullptr, gtab_band, ws.band_dbg,
                                 (ws.debug & ZSW_DEBUG_SEED_WIDE_BAND) ? 0xffffffffu : (ws.debug & ZSW_DEBUG_SCORE_PRUNE_ANY_SIZE) ? 0u : SEED_NARROW_MIN_READS, band_grid_cap, stream,
                                 ws.window_timer, out.narrow_only, out.reads_reversed && out.skip_handed_back);
        if (pe != hipSuccess) return pe;
        ap.b.items = ws.prune_fail_list + list_off;
        ap.n_items_dev = counter;
        // out.skip_handed_back (the shared-profile role's passes, the seeded reverse pass of the ranges): the caller recomputes every
        // read without the `unique` flag by other means, the handed-back reads among them — scoring them here would be thrown away
        if (!out.skip_handed_back) {
            // Against a long reference the few reads handed back are cut into chunks of rows, each an item of its own (a class of
            // 2,000 reads walking 30,000 rows each is a dozen blocks on 256 CUs): chunks of at least four times the rows a
            // positive path can span, so that the overlap costs a quarter more cells at most.
            int maxw = 0;
            for (int i = 0; i < h_sc.S * h_sc.S; ++i) maxw = std::max(maxw, (int)h_sc.w[i]);
            const uint64_t overlap = h_sc.gap_extend > 0 ? (uint64_t)longest + (uint64_t)longest * (uint64_t)maxw / (uint64_t)h_sc.gap_extend + 2 : ~0ull;
            const uint64_t rows = std::max<uint64_t>(4 * overlap, 2048);
            if (ws.chunk_keys && !(ws.debug & ZSW_DEBUG_NO_ROW_CHUNKS) && h_sc.gap_extend > 0 && rows * 2 <= ref_len && longest < (1u << CHUNK_COL_BITS) &&
                ref_len < (1u << CHUNK_ROW_BITS) - 2) {
                ap.chunk_rows = (uint32_t)rows;
                ap.chunk_overlap = (uint32_t)overlap;
                ap.chunk_keys = ws.chunk_keys;
                const uint32_t zgrid = (bb.n_items + 255) / 256;
                hipLaunchKernelGGL(chunk_zero_kernel, dim3(zgrid), dim3(256), 0, stream, ap.b.items, counter, bb.n_items, ws.chunk_keys);
                pe = launch_table_cfg_v2(ap, g, c, mode, stream);
                if (pe != hipSuccess) return pe;
                hipLaunchKernelGGL(chunk_finalize_kernel, dim3(zgrid), dim3(256), 0, stream, ap.b, counter, ws.chunk_keys, ap.limit, rule, out, std::min(mode, 2));
                pe = hipGetLastError();
            } else {
                // A short list is latency-bound in the class's configuration (10,000 reads hand back a few hundred: one wavefront per
                // 32 reads walking all R rows is 1.3 ms whatever the count), a long one throughput-bound in any other. The count
                // stays on the device, so the list is launched in up to three configurations — the ones the cost model picks for
                // 16 k, 128 k and many reads — and each launch returns at once unless the count lies in its range (measured, 1 M
                // reads: 21 k handed back 1.27 ms at 8 lanes per pair, 1.1 ms at 16; 110 k: 4.3 ms at 4 lanes, 4.0 ms at 8).
                const uint32_t cuts[2] = {49152u, 327680u}, probe[2] = {16384u, 131072u};
                uint32_t lo = 0;
                for (int k = 0; k < 3 && pe == hipSuccess; ++k) {
                    const uint32_t hi = k < 2 ? cuts[k] : 0xffffffffu;
                    int g2 = g, c2 = c;
                    ScoreArgsV2 ah = ap;
                    const bool other = k < 2 && lo < bb.n_items && score_config_for_batch(longest, probe[k], &g2, &c2) && g2 != g;
                    if (other) {
                        ah = a2;
                        if (!build_tables_v2(h_sc, g2, &ah)) continue;  // (the class's configuration serves this range too: lo stays)
                        ah.b = ap.b;
                        ah.b.n_items = std::min(bb.n_items, hi);  // (the grid: a longer list is not this launch's)
                        ah.n_items_dev = counter;
                    } else if (k < 2) {
                        continue;
                    }
                    ah.gate_lo = lo;
                    ah.gate_hi = hi;
                    if (lo < bb.n_items) pe = launch_table_cfg_v2(ah, other ? g2 : g, other ? c2 : c, mode, stream);
                    lo = hi;
                }
            }
        }
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, stream, counter, ws.prune_fail_count + 1);
        return hipGetLastError();
    };
    // Column-pruned pass over the items of `bb` (reads of at most kPruneClasses[cls].max_len bases), then score_kernel_v2 over the
    // reads it hands back (device-side list and count). hipErrorNotSupported: not switched on, or the batch does not qualify.
    auto prune_items = [&](const BatchDev& bb, int cls, uint32_t n_cls) -> hipError_t {
        // 5-letter tables: behind ZSW_DEBUG_PRUNE_STRIP (the seeded pass is their default); 8..32 letter alphabets (WIDE tables): the
        // default first pass — the seeded pass's k-mer argument does not hold for matrices whose substitutions score close to identities
        if (!(use_v2 || wide) || !ws.prune_work || !(ws.debug & ZSW_DEBUG_SCORE_PRUNE)) return hipErrorNotSupported;
        if (!wide && !(ws.debug & ZSW_DEBUG_PRUNE_STRIP)) return hipErrorNotSupported;
        if (bb.n_items < PR_MIN_READS && !(ws.debug & ZSW_DEBUG_SCORE_PRUNE_ANY_SIZE)) return hipErrorNotSupported;
        const int last = n_cls < bb.n_items ? cls + 1 : cls;  // the range may hold the next class too (same strip width)
        ScoreArgsV2 ap = a2;
        auto tables = [&](int g, ScoreArgsV2* t) { return wide ? build_tables_wide(h_sc, g, t) : build_tables_v2(h_sc, g, t); };
        if (!tables(1, &ap)) return hipErrorNotSupported;
        const uint32_t floor_strip = ap.floor0;
        uint32_t limit = ap.limit, floor_window[2] = {0, 0};
        for (int c = cls; c <= last; ++c) {
            if (!tables(kPruneClasses[c].g, &ap)) return hipErrorNotSupported;
            floor_window[c - cls] = ap.floor0;
            limit = std::min(limit, ap.limit);
        }
        if (!prune_applicable(h_sc, kPruneClasses[last].max_len, ref_len, limit)) return hipErrorNotSupported;
        int Gr = 0, Cr = 0;
        if (!score_config_for(kPruneClasses[last].max_len, &Gr, &Cr)) return hipErrorNotSupported;
        hipError_t pe = hipMemsetAsync(ws.prune_fail_count, 0, 4, stream);
        if (pe != hipSuccess) return pe;
        ap.b = bb;
        uint32_t est_failed = 0xffffffffu;
        pe = launch_score_pruned(ap, cls, n_cls, floor_strip, floor_window, h_sc, ws.prune_work, ws.prune_bytes, ws.prune_chunk,
                                 ws.prune_fail_list, ws.prune_fail_count, mode, wide, stream, &est_failed);
        if (pe != hipSuccess) return pe;
        // a short list of handed-back reads is latency-bound in the narrowest configuration (20,000 reads of 150 residues: 3.4 ms
        // at four lanes per pair, every block walking all R rows): with the probe's estimate it takes the small-batch configuration
        if (est_failed != 0xffffffffu) (void)score_config_for_batch(kPruneClasses[last].max_len, est_failed + est_failed / 4 + 1024, &Gr, &Cr);
        if (!tables(Gr, &a2)) return hipErrorInvalidValue;
        a2.b = bb;
        a2.b.items = ws.prune_fail_list;
        a2.n_items_dev = ws.prune_fail_count;
        pe = wide ? launch_table_cfg_v2_wide(a2, Gr, Cr, mode, stream) : launch_table_cfg_v2(a2, Gr, Cr, mode, stream);
        a2.n_items_dev = nullptr;
        if (pe != hipSuccess) return pe;
        hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, stream, ws.prune_fail_count, ws.prune_fail_count + 1);
        return hipGetLastError();
    };
    if (!table_ok) {  // alphabet outside the table kernels: exact kernel over the whole batch
        if (timer) timer->begin(stream);
        e = exact_all(b);
        if (timer) timer->end(stream);
        return e;
    }
    if (b.offsets && !b.items && b.n_items > 0) {
        // ragged: one launch per occupied length class (device-side histogram + scatter, counts read back once)
        BucketCaps caps;
        for (int k = 0; k < NCLS; ++k) caps.cap[k] = (uint32_t)(kCfgs[kBucketCfg[k]].G * kCfgs[kBucketCfg[k]].C);
        e = hipMemsetAsync(ws.bucket_counts, 0, 64 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        const uint32_t hgrid = std::min<uint32_t>(1024, (b.n_items + 255) / 256);
        hipLaunchKernelGGL(bucket_count_kernel, dim3(hgrid), dim3(256), 0, stream, b.offsets, b.n_items, caps, ws.bucket_counts);
        uint32_t counts[NCLS + 1];
        e = hipMemcpyAsync(counts, ws.bucket_counts, sizeof(counts), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        uint32_t starts[NCLS + 1], run = 0;
        for (int k = 0; k <= NCLS; ++k) {
            starts[k] = run;
            run += counts[k];
        }
        e = hipMemcpyAsync(ws.bucket_counts + 32, starts, sizeof(starts), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bucket_scatter_kernel, dim3(hgrid), dim3(256), 0, stream, b.offsets, b.n_items, caps,
                           ws.bucket_counts + 32, ws.bucket_items);
        if (timer) timer->begin(stream);
        // The length classes are independent launches and the small ones cannot fill the chip on their own (a class of
        // 70 k reads is two wavefronts per SIMD): they are spread over side streams, forked from and joined to `stream`.
        SideStreams* side = ws.side;  // owned by the context: two contexts never share fork/join events
        const bool fork = side != nullptr && !(ws.debug & ZSW_DEBUG_NO_SIDE_STREAMS);
        if (use_v2 && ws.seed && ws.seed_gtab && (ws.debug & ZSW_DEBUG_SCORE_PRUNE) && !(ws.debug & ZSW_DEBUG_PRUNE_STRIP)) {
            // the seeded pass's per-row score table, before the fork: every side stream reads it
            ScoreArgsV2 ag = a2;
            if (build_tables_v2(h_sc, 4, &ag)) {
                e = seed_build_gtab(ag, ws.seed_gtab, stream);
                if (e != hipSuccess) return e;
                ScoreArgsV2 ab = a2;
                if (!(ws.debug & ZSW_DEBUG_SEED_NO_BAND) && doubled_tables(h_sc, &ab)) {
                    e = seed_build_gtab(ab, ws.seed_gtab + (ref_len + 2 * SEED_GTAB_PAD), stream);
                    if (e != hipSuccess) return e;
                }
                gtab_built = true;
            }
        }
        if (fork) {
            e = hipEventRecord(side->fork, stream);
            if (e != hipSuccess) return e;
            for (int i = 0; i < SideStreams::N; ++i) {
                e = hipStreamWaitEvent(side->s[i], side->fork, 0);
                if (e != hipSuccess) return e;
            }
        }
        const hipStream_t main_stream = stream;
        int used = 0;
        // seeded exact pass per length class: each class has its own region of the workspace, its own part of the worklist (at
        // its items' offset) and its own counter, so every class runs its whole pipeline (seed, sort, window, full pass over the
        // reads handed back) on a side stream of its own turn. A handed-back read walks all R rows however few there are of them —
        // 15 ms per class against a 30 kb reference — so these launches must overlap.
        {
            size_t work_off = 0;
            uint32_t n_seedable = 0;  // the banded kernel's blocks are shared among the classes in proportion to their reads
            for (int k = 0; k < NCLS; ++k) n_seedable += counts[k];
            for (int k = NCLS - 1; k >= 0; --k) {
                if (!counts[k]) continue;
                BatchDev bk = b;
                bk.items = ws.bucket_items + starts[k];
                bk.n_items = counts[k];
                const Cfg& cf = kCfgs[kBucketCfg[k]];
                stream = fork ? side->s[used % SideStreams::N] : main_stream;
                const uint32_t grid_cap = seed_band_class_cap(counts[k], n_seedable);
                e = seed_items(bk, cf.G, cf.C, caps.cap[k], work_off, starts[k], ws.prune_fail_count + 2 + k, grid_cap);
                stream = main_stream;
                if (e == hipSuccess) ++used;
                if (e == hipSuccess) {
                    work_off += seed_workspace_bytes(counts[k], caps.cap[k], grid_cap);
                    counts[k] = 0;
                } else if (e != hipErrorNotSupported) {
                    return e;
                }
            }
        }
        // column-pruned pass (strip + window; ZSW_DEBUG_PRUNE_STRIP): consecutive length classes that share a pruning class form
        // one item range, and pruning classes with the same strip width share the strip launch
        {
            int k0[PR_N_CLASSES], k1[PR_N_CLASSES];
            uint32_t cnt[PR_N_CLASSES];
            for (int pc = 0; pc < PR_N_CLASSES; ++pc) {
                k0[pc] = k1[pc] = -1;
                cnt[pc] = 0;
                for (int k = 0; k < NCLS; ++k) {
                    const uint32_t cap = caps.cap[k];
                    if (cap <= kPruneClasses[pc].max_len && (pc == 0 || cap > kPruneClasses[pc - 1].max_len) && cap > (uint32_t)kPruneClasses[pc].cp + 40) {
                        if (k0[pc] < 0) k0[pc] = k;
                        k1[pc] = k;
                        cnt[pc] += counts[k];
                    }
                }
            }
            for (int pc = 0; pc < PR_N_CLASSES; ++pc) {
                if (k0[pc] < 0) continue;
                int last = pc;
                if (pc + 1 < PR_N_CLASSES && k0[pc + 1] == k1[pc] + 1 && kPruneClasses[pc + 1].cp == kPruneClasses[pc].cp && ref_len < (1u << 24)) last = pc + 1;
                BatchDev bp = b;
                bp.items = ws.bucket_items + starts[k0[pc]];
                bp.n_items = starts[k1[last]] + counts[k1[last]] - starts[k0[pc]];
                if (bp.n_items) {
                    e = prune_items(bp, pc, cnt[pc]);
                    if (e == hipSuccess) {
                        for (int k = k0[pc]; k <= k1[last]; ++k) counts[k] = 0;
                    } else if (e != hipErrorNotSupported) {
                        return e;
                    }
                }
                pc = last;
            }
        }
        for (int k = NCLS; k >= 0; --k) {  // longest class first
            if (!counts[k]) continue;
            BatchDev bk = b;
            bk.items = ws.bucket_items + starts[k];
            bk.n_items = counts[k];
            if (k == NCLS) {  // longer than every strip configuration; stays on the main stream (shares scratch with the pass below)
                e = launch_tiled(bk, max_len);
                if (e == hipErrorNotSupported) e = exact_all(bk);
            } else {
                stream = fork ? side->s[used++ % SideStreams::N] : main_stream;
                e = launch_one(bk, kCfgs[kBucketCfg[k]].G, kCfgs[kBucketCfg[k]].C);
                stream = main_stream;
            }
            if (e != hipSuccess) return e;
        }
        if (fork) {
            for (int i = 0; i < SideStreams::N; ++i) {
                e = hipEventRecord(side->join[i], side->s[i]);
                if (e != hipSuccess) return e;
                e = hipStreamWaitEvent(stream, side->join[i], 0);
                if (e != hipSuccess) return e;
            }
        }
        if (timer) timer->end(stream);
    } else {
        if (!score_config_for_batch(max_len, b.n_items, &G, &C)) {  // longer than every strip configuration
            if (timer) timer->begin(stream);
            BatchDev bl = b;
            if (!bl.items && ws.bucket_items) {  // the tiles address reads through an item list
                hipLaunchKernelGGL(iota_kernel, dim3((b.n_items + 255) / 256), dim3(256), 0, stream, ws.bucket_items, b.n_items);
                bl.items = ws.bucket_items;
            }
            e = launch_tiled(bl, max_len);
            if (e == hipErrorNotSupported) {
                e = exact_all(b);
                if (timer) timer->end(stream);
                return e;
            }
            if (timer) timer->end(stream);
            if (e != hipSuccess) return e;
            return finish_worklist();
        }
        // The seeded exact pass (zsw_score_seed.hip), or the column-pruned pass (zsw_score_prune.hip; ZSW_DEBUG_PRUNE_STRIP); the
        // reads they hand back are scored over all their cells. Otherwise the full pass. One timed interval either way.
        if (timer) timer->begin(stream);
        e = hipErrorNotSupported;
        {
            int Gs = 0, Cs = 0;
            if (score_config_for(max_len, &Gs, &Cs)) e = seed_items(b, Gs, Cs, max_len, 0, 0, ws.prune_fail_count, SEED_BAND_MAX_GRID);
        }
        const int cls = prune_class_for(max_len);
        if (e == hipErrorNotSupported && cls >= 0) e = prune_items(b, cls, b.n_items);
        if (e == hipErrorNotSupported) e = launch_one(b, G, C);
        if (timer) timer->end(stream);
        if (e != hipSuccess) return e;
    }
    return finish_worklist();
}

// ---- reverse pass of sw_simd_score_ranges -------------------------------------------------------------------
__global__ void gtab_kernel(const uint8_t* ref, uint32_t n, const ScoringDev* sc, ScoreArgs a, int wide, uint2* gtab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int idx = sc->index_map[ref[i]];
        gtab[i] = wide ? make_uint2((uint32_t)(idx * WIDE_STRIDE), 0u) : make_uint2(a.wtab[idx][0], a.wtab[idx][1]);
    }
}

template <int G, int C>
static hipError_t launch_cfg_rev(const ScoreArgs& a, bool fast, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
    if (fast) hipLaunchKernelGGL((score_kernel<G, C, true, 2, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((score_kernel<G, C, false, 2, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_score_rev(const ScoringDev* d_sc, const ScoringDev& h_sc, const BatchDev& b, uint32_t max_len,
                            const uint8_t* d_ref, uint32_t ref_len, const ResultRule& rule, const ScoreOut& out,
                            const ScoreWorkspace& ws, const uint32_t* d_fwd_ref_end, const uint32_t* d_fwd_query_end,
                            const uint32_t* d_fwd_score, uint2* d_gtab, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out.fb_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t exact_grid = (uint32_t)(ws.slots / 64);
    int G = 0, C = 0;
    const bool wide = wide_ok(h_sc, ws.debug);
    const bool table_ok = (h_sc.S <= 7 || fast_ok(h_sc) || wide) && score_config_for(max_len, &G, &C);
    if (!score_config_for(max_len, &G, &C) && ws.tile_buf && ws.tile_state && ws.bucket_items && w32_ok(h_sc, ref_len, max_len, ws.debug) && b.n_items) {
        // reads longer than every strip configuration: the 32-bit tile kernel over the reversed reference (held in d_gtab's bytes)
        uint8_t* d_rev = reinterpret_cast<uint8_t*>(d_gtab);
        if (ref_len) hipLaunchKernelGGL(reverse_bytes_kernel, dim3((ref_len + 255) / 256), dim3(256), 0, stream, d_ref, ref_len, d_rev);
        hipLaunchKernelGGL(iota_kernel, dim3((b.n_items + 255) / 256), dim3(256), 0, stream, ws.bucket_items, b.n_items);
        ScoreArgsV2 a2;
        a2.b = b;
        a2.ref = d_rev;
        a2.ref_len = ref_len;
        a2.sc = d_sc;
        a2.rule = rule;
        a2.out = out;
        a2.rev_ref_end = d_fwd_ref_end;
        a2.rev_query_end = d_fwd_query_end;
        a2.tile_state = ws.tile_state;
        fill_wide_table(h_sc, &a2);
        a2.ge2 = (uint32_t)h_sc.gap_extend * 0x00010001u;
        a2.gd2 = (uint32_t)(h_sc.gap_open - h_sc.gap_extend) * 0x00010001u;
        const size_t per_read = (size_t)ref_len * sizeof(uint2), half = ws.tile_bytes / 2;
        const size_t fit = per_read ? half / per_read : (size_t)b.n_items;
        if (fit > 0) {
            uint2* buf[2] = {ws.tile_buf, reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(ws.tile_buf) + half)};
            const uint32_t n_tiles = (max_len + TILE_COLS - 1) / TILE_COLS;
            for (uint32_t first = 0; first < b.n_items; first += (uint32_t)std::min<size_t>(fit, 0x7fffffffu)) {
                a2.b.n_items = (uint32_t)std::min<size_t>(fit, b.n_items - first);
                for (uint32_t t = 0; t < n_tiles; ++t) {
                    a2.tile_q0 = t * (uint32_t)TILE_COLS;
                    a2.tile_in = t ? buf[(t - 1) & 1] : nullptr;
                    a2.tile_out = t + 1 < n_tiles ? buf[t & 1] : nullptr;
                    e = launch_tile_w32(a2, ws.bucket_items + first, 2, stream);
                    if (e != hipSuccess) return e;
                }
            }
            return hipSuccess;
        }
    }
    if (!table_ok) {
        hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, d_ref, ref_len, d_sc, rule, out, ws.scratch, (uint32_t)ws.slots,
                           ws.scratch_len, d_fwd_ref_end, d_fwd_query_end);
        return hipGetLastError();
    }
    ScoreArgs a;
    a.b = b;
    a.ref = d_ref;
    a.ref_len = ref_len;
    a.sc = d_sc;
    a.rule = rule;
    a.out = out;
    a.rev_ref_end = d_fwd_ref_end;
    a.rev_query_end = d_fwd_query_end;
    a.rev_score = d_fwd_score;
    a.gtab = d_gtab;
    const bool fast = fast_ok(h_sc);
    if (wide) {
        for (int r = 0; r < 33; ++r)
            for (int q = 0; q < WIDE_STRIDE; ++q)
                a.wide[r * WIDE_STRIDE + q] = (int8_t)((r < h_sc.S && q < h_sc.S) ? h_sc.w[r * h_sc.S + q] : 0);
        a.go2 = ((uint32_t)h_sc.gap_open & 0xffffu) * 0x00010001u;
        a.ge2 = ((uint32_t)h_sc.gap_extend & 0xffffu) * 0x00010001u;
        a.bias2 = 0;
    } else {
        build_tables(h_sc, fast, &a);
    }
    if (ref_len) hipLaunchKernelGGL(gtab_kernel, dim3((ref_len + 255) / 256), dim3(256), 0, stream, d_ref, ref_len, d_sc, a, (int)wide, d_gtab);
    if (wide) {
        e = launch_cfg_rev_wide(a, G, C, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, out.fb_list, out.fb_count, d_ref, ref_len, d_sc,
                           rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, d_fwd_ref_end, d_fwd_query_end);
        return hipGetLastError();
    }
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: e = launch_cfg_rev<GV, CV>(a, fast, stream); break;
        ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CASE)
#undef ZSW_CASE
        default: e = hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, out.fb_list, out.fb_count, d_ref, ref_len, d_sc,
                       rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, d_fwd_ref_end, d_fwd_query_end);
    return hipGetLastError();
}

}  // namespace zsw
