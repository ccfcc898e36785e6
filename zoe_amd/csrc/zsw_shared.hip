// zsw_shared.hip — the one-profile-many-sequences role of the striped functions (sw/mod.rs:63-67: "the profile can be aligned
// against any number of different sequences"; SharedProfiles, profile_set.rs:552-560; Nucleotides::into_shared_profile,
// nucleotides/mod.rs:295-299): the PROFILE is built from one sequence the context holds (zsw_set_profile_sequence, typically
// the reference) and every read of the batch is the other sequence, the one sw_simd_* walks row by row.
//
// The score is the same number either way round (for the transposed matrix), so the score-only entry points go through the
// ordinary kernels. What differs is the tie rule of sw_simd_score_ends (striped.rs:296-321: first ROW holding the maximum, then
// the first column of that row) — rows are now positions of the READ, columns positions of the profile sequence — and with it
// the ranges and the cell the alignment's traceback starts from. shared_ends_kernel computes exactly that: one read per
// wavefront, lane l owns 32 consecutive columns of the profile sequence (2,048 per tile; longer sequences take more tiles, the
// strip boundary of every row waiting in LDS), the lanes walk the read's rows skewed by one step each, 32-bit values, the
// substitution score of a cell from an LDS copy of the weight matrix (row = residue of the read, the non-profile sequence:
// weights[ref_idx][query_idx], matrices/mod.rs:230-235). REV is the second pass of sw_simd_score_ranges (striped.rs:355-388) on
// the reversed prefixes. The alignment itself (flags of the <T, N> striping over the profile sequence, nv = ceil(len / N)
// vectors) is align_kernel<N, ., SHARED> in zsw_align.hip.
#include <algorithm>

#include "zsw_align_dev.hpp"
#include "zsw_shared.hpp"

namespace zsw {

namespace {

constexpr int SH_C = 32;              // columns per lane
constexpr int SH_TILE = 64 * SH_C;    // columns per tile
constexpr int SH_STRIDE = 36;         // bytes per row of the LDS weight table (rows start in different banks)

struct SharedArgs {
    BatchDev b;
    const uint8_t* pseq;
    uint32_t plen;
    const ScoringDev* sc;
    ResultRule rule;
    ScoreOut out;        // ref_end = row in the read + 1, query_end = column in the profile sequence + 1 (REV: the starts)
    uint32_t max_rows;   // longest read of the batch (LDS: residues + boundary of every row)
    const uint32_t* rev_ref_end;    // REV: forward ends (rows of the read / columns of the profile sequence used)
    const uint32_t* rev_query_end;
    const uint32_t* n_items_dev;    // non-null: the number of items is on the device (a list filled by an earlier kernel of the stream)
};

template <bool REV>
__global__ __launch_bounds__(64) void shared_ends_kernel(SharedArgs a) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint8_t lut[256];
    __shared__ int8_t wt[33 * SH_STRIDE];  // wt[r * SH_STRIDE + q] = w[r][q]; row / column 32: scores 0 (padding, rows outside the read)
    const int lane = threadIdx.x;
    const int S = a.sc->S;
    int32_t* Hb = reinterpret_cast<int32_t*>(smem);  // [max_rows]: H of the previous tile's last column, per row
    int32_t* Fb = Hb + a.max_rows;                   // [max_rows]: F leaving it
    uint8_t* rres = reinterpret_cast<uint8_t*>(Fb + a.max_rows);  // [max_rows]: residues of the read's rows
    for (int i = lane; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = lane; i < 33 * SH_STRIDE; i += 64) {
        const int r = i / SH_STRIDE, q = i % SH_STRIDE;
        wt[i] = (r < S && q < S) ? (int8_t)a.sc->w[r * S + q] : (int8_t)0;
    }
    __syncthreads();
    const int go = a.sc->gap_open, ge = a.sc->gap_extend;

    const uint32_t n_items = a.n_items_dev ? min(*a.n_items_dev, a.b.n_items) : a.b.n_items;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint32_t id = a.b.items ? a.b.items[item] : item;
        uint64_t off = 0;
        uint32_t rows = read_len(a.b, id, &off);
        uint32_t cols = a.plen;
        if (REV) {  // reverse(read[..ref_end]) against the profile of reverse(pseq[..query_end])
            const uint32_t re = a.rev_ref_end[id], qe = a.rev_query_end[id];
            rows = re <= rows ? re : rows;
            cols = qe <= cols ? qe : cols;
            if (cols == 0) rows = 0;
        }
        __syncthreads();  // the previous item's LDS rows are dead
        for (uint32_t r = lane; r < rows; r += 64) rres[r] = lut[a.b.bases[REV ? off + (rows - 1 - r) : off + r]];
        __syncthreads();
        int32_t bestv = 0, best_r = 0x7fffffff, best_c = 0x7fffffff;  // over the tiles so far
        const int n_tiles = (int)((cols + SH_TILE - 1) / SH_TILE);
        for (int tile = 0; tile < n_tiles && rows > 0; ++tile) {
            // this lane's columns: byte offsets of their residues in a table row (32: padding)
            uint32_t cq[SH_C];
#pragma unroll
            for (int c = 0; c < SH_C; ++c) {
                const uint32_t q = (uint32_t)tile * SH_TILE + (uint32_t)lane * SH_C + (uint32_t)c;
                cq[c] = q < cols ? (uint32_t)lut[a.pseq[REV ? cols - 1 - q : q]] : 32u;
            }
            int32_t H[SH_C], E[SH_C], snap[SH_C];
#pragma unroll
            for (int c = 0; c < SH_C; ++c) H[c] = E[c] = snap[c] = 0;
            int32_t lbest = 0, lrow = 0x7fffffff;
            int32_t Fout = 0, Hlast = 0, Hin_prev = 0;
            const int T = (int)rows + 63;
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
                const int row = t - lane;
                const bool in = row >= 0 && row < (int)rows;
                const uint32_t rbase = (in ? (uint32_t)rres[row] : 32u) * SH_STRIDE;
                int32_t Fin = __shfl_up(Fout, 1, 64);
                int32_t Hin = __shfl_up(Hlast, 1, 64);
                if (lane == 0) {  // the strip to the left belongs to the previous tile (or is the border)
                    Fin = (tile > 0 && in) ? Fb[row] : 0;
                    Hin = (tile > 0 && in) ? Hb[row] : 0;
                }
                int32_t hd = Hin_prev + (int32_t)wt[rbase + cq[0]];
                Hin_prev = Hin;
                int32_t F = Fin, rmax = 0;
#pragma unroll
                for (int c = 0; c < SH_C; ++c) {
                    int32_t hd_next = 0;
                    if (c + 1 < SH_C) hd_next = H[c] + (int32_t)wt[rbase + cq[c + 1 < SH_C ? c + 1 : c]];
                    const int32_t h = max(hd, max(E[c], F));  // E, F >= 0: the floor at T::MIN
                    H[c] = h;
                    const int32_t hg = subsat(h, go);
                    E[c] = max(subsat(E[c], ge), hg);
                    F = max(subsat(F, ge), hg);
                    rmax = max(rmax, h);
                    hd = hd_next;
                }
                Fout = F;
                Hlast = H[SH_C - 1];
                if (lane == 63 && in && tile + 1 < n_tiles) {  // hand the strip boundary of this row to the next tile
                    Hb[row] = Hlast;
                    Fb[row] = Fout;
                }
                if (in && rmax > lbest) {  // strictly greater: the first row holding the lane's maximum
                    lbest = rmax;
                    lrow = row;
#pragma unroll
                    for (int c = 0; c < SH_C; ++c) snap[c] = H[c];
                }
            }
            // the tile's (maximum, first row, first column of that row)
            int32_t tb = lbest;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) tb = max(tb, __shfl_xor(tb, d, 64));
            int32_t tr = lbest == tb ? lrow : 0x7fffffff;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) tr = min(tr, __shfl_xor(tr, d, 64));
            int32_t tc = 0x7fffffff;
            if (lbest == tb && lrow == tr) {
#pragma unroll
                for (int c = SH_C - 1; c >= 0; --c)
                    if (snap[c] == tb) tc = tile * SH_TILE + lane * SH_C + c;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) tc = min(tc, __shfl_xor(tc, d, 64));
            if (tb > bestv || (tb == bestv && tb > 0 && tr < best_r)) {  // same row: the earlier tile holds the earlier column
                bestv = tb;
                best_r = tr;
                best_c = tc;
            }
            __syncthreads();  // the boundary rows are complete before the next tile reads them
        }
        if (lane == 0) {
            const uint32_t full_rows = a.b.offsets ? (uint32_t)(a.b.offsets[id + 1] - a.b.offsets[id]) : a.b.fixed_len;
            if (full_rows == 0 && !REV) {  // sw_simd_score_ends on an empty `reference`: Unmapped (striped.rs:219-221)
                a.out.score[id] = 0;
                a.out.status[id] = ZSW_STATUS_UNMAPPED;
                if (a.out.tier) a.out.tier[id] = a.rule.tier_code[0];
                if (a.out.ref_end) a.out.ref_end[id] = 0;
                if (a.out.query_end) a.out.query_end[id] = 0;
            } else {
                uint32_t score;
                uint8_t status, tier;
                apply_rule(a.rule, (uint64_t)bestv, &score, &status, &tier);
                a.out.score[id] = score;
                a.out.status[id] = status;
                if (a.out.tier) a.out.tier[id] = tier;
                const bool some = status == ZSW_STATUS_SOME;
                if (REV) {  // inclusive starts (striped.rs:326-328)
                    a.out.ref_end[id] = some ? rows - (uint32_t)(best_r + 1) : 0;
                    a.out.query_end[id] = some ? cols - (uint32_t)(best_c + 1) : 0;
                } else {
                    if (a.out.ref_end) a.out.ref_end[id] = some ? (uint32_t)best_r + 1 : 0;
                    if (a.out.query_end) a.out.query_end[id] = some ? (uint32_t)best_c + 1 : 0;
                }
            }
        }
    }
}

}  // namespace

size_t shared_ends_lds(uint32_t max_rows) { return (size_t)max_rows * 9 + 64; }

hipError_t launch_shared_ends(const BatchDev& b, uint32_t max_rows, const uint8_t* d_pseq, uint32_t plen, const ScoringDev* d_sc,
                              const ResultRule& rule, const ScoreOut& out, const uint32_t* rev_ref_end, const uint32_t* rev_query_end,
                              hipStream_t stream, const uint32_t* n_items_dev) {
    if (b.n_items == 0) return hipSuccess;
    SharedArgs a;
    a.b = b;
    a.pseq = d_pseq;
    a.plen = plen;
    a.sc = d_sc;
    a.rule = rule;
    a.out = out;
    a.max_rows = (max_rows + 3) & ~3u;
    a.rev_ref_end = rev_ref_end;
    a.rev_query_end = rev_query_end;
    a.n_items_dev = n_items_dev;
    const size_t lds = shared_ends_lds(a.max_rows);
    if (lds > SHARED_MAX_LDS) return hipErrorNotSupported;
    // a device-side count: a grid that fills the chip strides over however many items there are
    const uint32_t grid = std::min<uint32_t>(b.n_items, n_items_dev ? 16384u : 1u << 20);
    if (rev_ref_end) hipLaunchKernelGGL(shared_ends_kernel<true>, dim3(grid), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL(shared_ends_kernel<false>, dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace zsw
