// zsw_score_w32.hip — the strip kernel in 32-bit lanes, tile by tile: reads whose true score does not fit the packed 16-bit
// kernels (>= ~28,000: whole-genome sized sequences, the `i32` tier of Zoe's cascade). Same parallelisation as score_kernel_v2
// (G = 64 lanes x C = 38 columns per tile, systolic skew, strip boundary through __shfl_up, tile boundary through HBM, row
// drift D_r = (r + G + 2) * gap_extend so that E needs no per-row subtraction), one read per lane group, plain 32-bit integer
// arithmetic (v_max3_i32), the substitution score of a cell from an LDS copy of the weight matrix. Results are the same
// (score, ref_end, query_end) the packed kernels produce: first row holding the maximum, then first column.
#include "zsw_score_v2.hpp"

namespace zsw {

namespace {

__device__ __forceinline__ int32_t max3i(int32_t a, int32_t b, int32_t c) { return max(max(a, b), c); }

template <int MODE>
__global__ __launch_bounds__(BLOCK, 2) void score_kernel_w32(ScoreArgsV2 a, const uint32_t* list) {
    constexpr int G = TILE_G, C = TILE_C;
    __shared__ uint16_t rpw[CH + G];     // byte offset of each staged row's table row
    __shared__ uint32_t wt32[33 * 9];    // the score table (score + gap_extend as signed bytes)
    __shared__ uint32_t lut32[64];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);
    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;  // = item of this lane group within the launch
    const bool valid = group < a.b.n_items;
    const uint32_t id = valid ? list[group] : 0;

    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    for (int i = tid; i < 33 * 9; i += BLOCK) wt32[i] = reinterpret_cast<const uint32_t*>(a.wide)[i];
    __syncthreads();
    const int8_t* wt = reinterpret_cast<const int8_t*>(wt32);

    uint64_t off = 0;
    uint32_t len = 0;
    if (valid) {
        if (a.b.offsets) {
            off = a.b.offsets[id];
            len = (uint32_t)(a.b.offsets[id + 1] - off);
        } else {
            off = (uint64_t)id * a.b.fixed_len;
            len = a.b.fixed_len;
        }
    }
    const int R = (int)a.ref_len;
    // Reverse pass: rows before this read's prefix behave like the neutral row (the state stays at the zero floor, which is
    // what starting the recurrence at that row means), the read is reverse(read[..query_end]).
    const bool rev = a.rev_ref_end != nullptr;
    int start_row = 0;
    if (rev) {
        const uint32_t qe_fwd = valid ? a.rev_query_end[id] : 0;
        len = min(len, qe_fwd);
        start_row = len ? R - (int)min(a.rev_ref_end[id], (uint32_t)R) : R;
    }
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t q = a.tile_q0 + (uint32_t)(g * C + c);
        sel[c] = q < len ? (uint32_t)lut[a.b.bases[rev ? off + (len - 1 - q) : off + q]] : (uint32_t)WIDE_PAD;
    }
    const int32_t ge = (int32_t)(a.ge2 & 0xffffu), gd = (int32_t)(a.gd2 & 0xffffu);
    // Reverse pass: no read of this block takes part before the smallest start row, so the block starts at the chunk holding it
    // (the state there is the zero floor either way; later tiles skip the same rows and never read their boundary).
    __shared__ int first_row;
    if (tid == 0) first_row = rev ? R : 0;
    __syncthreads();
    if (rev && g == 0) atomicMin(&first_row, start_row);
    __syncthreads();
    const int base0 = (first_row / CH) * CH;
    // D_r = (r + G + 2) * ge: row r of this lane at step t is r = t - g, so every D the lane touches is >= 0
    int32_t Dr = (base0 + G + 1 - g) * ge;  // D of row r-1 at the first step (r = base0 - g)
    int32_t H[C], E[C];
    int32_t snap[MODE == 2 ? C : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        H[c] = Dr;
        E[c] = Dr + ge;
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = 0;
    }
    int32_t snapD = 0, best = 0, rbest = 0;
    int32_t Fout = Dr, Hlast = Dr, Hin_prev = Dr;
    const int T = R + G - 1;
    uint2 bd = make_uint2(0u, 0u);
    if (g == 0 && valid && a.tile_in != nullptr && base0 < R) bd = a.tile_in[(size_t)group * (size_t)R + (size_t)base0];

    for (int base = base0; base < T; base += CH) {
        __syncthreads();
        for (int j = tid; j < CH + G - 1; j += BLOCK) {
            const int row = base - (G - 1) + j;
            int idx = WIDE_NEUTRAL;
            if (row >= 0 && row < R) idx = lut[a.ref[row]];
            rpw[j] = (uint16_t)(idx * WIDE_STRIDE);
        }
        __syncthreads();
        const int tend = (T < base + CH) ? T : base + CH;
        const int joff = (G - 1 - g) - base;
        uint32_t w = rpw[base + joff];
#pragma unroll 1
        for (int t = base; t < tend; ++t) {
            const uint32_t wn = rpw[t + 1 + joff];
            const int row = t - g;
            if (row < start_row) w = WIDE_NEUTRAL * WIDE_STRIDE;
            Dr += ge;                   // D_r
            const int32_t Dn = Dr + ge;  // D_{r+1}
            int32_t Fin = __shfl_up(Fout, 1, G);
            int32_t Hin = __shfl_up(Hlast, 1, G);
            if (g == 0) {
                Fin = Dr;
                Hin = Dr;
                if (valid && a.tile_in != nullptr && row >= 0 && row < R) {
                    Hin = (int32_t)bd.x;
                    Fin = (int32_t)bd.y;
                }
                if (valid && a.tile_in != nullptr && row + 1 < R) bd = a.tile_in[(size_t)group * (size_t)R + (size_t)(row + 1)];
            }
            int32_t hd = Hin_prev + (int32_t)wt[w + sel[0]];
            Hin_prev = Hin;
            int32_t F = Fin;
            int32_t rmax = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                int32_t hd_next = 0;
                if (c + 1 < C) hd_next = H[c] + (int32_t)wt[w + sel[c + 1 < C ? c + 1 : c]];
                const int32_t h = max3i(hd, E[c], F);
                H[c] = h;
                const int32_t hg = h - gd;
                E[c] = max3i(E[c], hg, Dn);
                F = max3i(F, hg, Dn) - ge;
                rmax = max(rmax, h);
                hd = hd_next;
            }
            Fout = F;
            Hlast = H[C - 1];
            if (g == G - 1 && valid && a.tile_out != nullptr && row >= 0 && row < R)  // lane groups past the list own no boundary rows
                a.tile_out[(size_t)group * (size_t)R + (size_t)row] = make_uint2((uint32_t)Hlast, (uint32_t)Fout);
            const int32_t tmax = rmax - Dr;  // true row maximum of the strip
            if (tmax > best) {
                best = tmax;
                rbest = row;
                if (MODE == 2) {
#pragma unroll
                    for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = H[c];
                    snapD = Dr;
                }
            }
            w = wn;
        }
    }

    // ---- per-read reduction over the G lanes (one wavefront) ----
    int gb = best;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) gb = max(gb, __shfl_xor(gb, d, G));
    uint32_t re = 0, qe = 0;
    if (MODE != 0) {
        int k = (best == gb) ? rbest : 0x7fffffff;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) k = min(k, __shfl_xor(k, d, G));
        re = (uint32_t)k + 1;
        if (MODE == 2) {
            int cfirst = 0x7fffffff;
#pragma unroll
            for (int c = C - 1; c >= 0; --c)
                if (snap[MODE == 2 ? c : 0] - snapD == gb) cfirst = (int)a.tile_q0 + g * C + c;
            if (!(best == gb && rbest == k)) cfirst = 0x7fffffff;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) cfirst = min(cfirst, __shfl_xor(cfirst, d, G));
            qe = (uint32_t)cfirst + 1;
        }
    }
    if (g != 0 || !valid) return;
    uint32_t best_u = (uint32_t)gb;
    if (a.tile_in != nullptr) {  // fold into the running result: larger score, then earlier row, then earlier column (= earlier tile)
        const uint4 prev = a.tile_state[id];
        if (prev.x > best_u || (prev.x == best_u && prev.y <= re)) {
            best_u = prev.x;
            re = prev.y;
            qe = prev.z;
        }
    }
    if (a.tile_out != nullptr) {
        a.tile_state[id] = make_uint4(best_u, re, qe, 0u);
        return;
    }
    if (len == 0) {
        a.out.score[id] = 0;
        a.out.status[id] = ZSW_STATUS_EMPTY;
        if (a.out.tier) a.out.tier[id] = 0;
        if (MODE != 0 && a.out.ref_end) a.out.ref_end[id] = 0;
        if (MODE == 2 && a.out.query_end) a.out.query_end[id] = 0;
        return;
    }
    uint32_t score;
    uint8_t status, tier;
    apply_rule(a.rule, (uint64_t)best_u, &score, &status, &tier);
    a.out.score[id] = score;
    a.out.status[id] = status;
    if (a.out.tier) a.out.tier[id] = tier;
    const bool some = status == ZSW_STATUS_SOME;
    if (rev) {  // inclusive 0-based starts (striped.rs:326-328); `re` counts rows of the whole reversed reference
        if (a.out.ref_end) a.out.ref_end[id] = some ? (uint32_t)R - re : 0;
        if (a.out.query_end) a.out.query_end[id] = some ? len - qe : 0;
        return;
    }
    if (MODE != 0 && a.out.ref_end) a.out.ref_end[id] = some ? re : 0;
    if (MODE == 2 && a.out.query_end) a.out.query_end[id] = some ? qe : 0;
}

}  // namespace

// One tile over the reads `list[0 .. n)`; a.b.n_items = n, a.b.items unused (the list is given explicitly).
hipError_t launch_tile_w32(const ScoreArgsV2& a, const uint32_t* list, int mode, hipStream_t stream) {
    const uint32_t per_block = BLOCK / TILE_G;
    const uint32_t grid = (a.b.n_items + per_block - 1) / per_block;
    if (grid == 0) return hipSuccess;
    if (mode == 0) hipLaunchKernelGGL(score_kernel_w32<0>, dim3(grid), dim3(BLOCK), 0, stream, a, list);
    else hipLaunchKernelGGL(score_kernel_w32<2>, dim3(grid), dim3(BLOCK), 0, stream, a, list);
    return hipGetLastError();
}

}  // namespace zsw
