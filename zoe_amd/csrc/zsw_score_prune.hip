// zsw_score_prune.hip — the score-only pass with exact column pruning (sw_simd_score, striped.rs:65-142: only the maximum
// of the DP matrix is returned, so cells that provably cannot lie on a path scoring more than a score already found need not
// be computed).
//
// Two kernels replace score_kernel_v2 for batches of short reads against a reference that fits one LDS row table:
//   1. prune_strip_kernel: query columns [0, CP) against EVERY reference row, one read pair per lane. Exact for these columns
//      (nothing flows into them from the right). It leaves, per read, the strip's own maximum, the boundary stream
//      (H[r][CP-1], F[r][CP]) of every row r — everything the columns to the right ever receive from the strip — the maxima of
//      the two per block of PR_BLK rows, and the row of the largest boundary H (the anchor: where the read's alignment
//      crosses column CP).
//   2. prune_window_kernel: columns [CP, L) for a window of rows around the anchor only (reads sorted by anchor, two per
//      lane group), fed with the exact boundary stream, starting from a zero state. Every value it computes is the score of a
//      real alignment, so S' = max(strip maximum, window maximum) is a lower bound of the read's score, and it is THE score
//      if no path that leaves the computed region can beat S'. A path through cell (r, c) scores at most H[r][c] +
//      maxw * (columns left of the read), which gives three checks, all against S':
//        V1  a path that starts right of the strip outside the window scores at most maxw * (len - CP);
//        V2  a path that crosses from the strip to the right in a row outside the window scores at most
//            max(H[r][CP-1] + maxw * (len - CP), F[r][CP] + maxw * (len - CP - 1)) — checked per block of rows;
//        V3  a path that leaves the window through its last rows scores at most max(H, E) + maxw * (len - c - 1) of the cell
//            it leaves from.
//      A read that fails a check goes to a list and is scored by score_kernel_v2 over all its cells, so the results are
//      those of the full pass for every input; the checks only decide how much work a read costs.
// On the synthetic 150 bp reads 95 % pass with CP = 24 and windows of ~190 rows: 24/150 of the columns over all rows plus
// 126/150 over a tenth of the rows.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "zsw_score_prune.hpp"
#include "zsw_score_v2.hpp"

namespace zsw {

namespace {

struct PruneArgs {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    uint32_t wtab[9][2];
    // WIDE kernels (alphabets of 8..32 letters): the score table of score_kernel_v2<.., WIDE> and the potential of a query column
    // by its residue, max(0, max_x w[x][q]) — what the column can add to a path at most (index WIDE_PAD: padding, 0)
    int8_t wide[33 * 36];
    uint8_t colpot[36];
    uint32_t ge2, gd2, floor0, K;
    ResultRule rule;
    ScoreOut out;
    uint32_t first;    // first read of the chunk
    uint32_t n;        // reads of the chunk
    uint32_t n_pairs;  // (n + 1) / 2
    uint32_t split;    // strip kernel: entries >= split of the chunk belong to the next length class (sort key bit 24)
    uint32_t maxw;
    uint32_t nblk;     // blocks of PR_BLK rows
    uint2* bnd;        // [pair][row_stride]: (H[r][CP-1], F[r][CP]), true scores, read A in the low halves
    uint32_t row_stride;  // ref_len rounded up to 8 rows (one 64-byte line)
    uint2* blk;        // [pair][nblk]: maxima of the two per block of rows
    uint32_t* best0;   // [pair]: the strip's own maximum
    uint32_t* anchor;  // [read of the chunk]: row of the largest boundary H
    const uint32_t* order;  // window kernel: reads of the chunk sorted by anchor
    uint32_t* fail_list;    // global read ids
    uint32_t* fail_count;
};

__device__ __forceinline__ void read_span(const BatchDev& b, uint32_t id, uint64_t* off, uint32_t* len) {
    if (b.offsets) {
        *off = b.offsets[id];
        *len = (uint32_t)(b.offsets[id + 1] - *off);
    } else {
        *off = (uint64_t)id * b.fixed_len;
        *len = b.fixed_len;
    }
}

// read id of entry k of the chunk (the batch may address its reads through an item list: length classes of a ragged batch)
__device__ __forceinline__ uint32_t read_id(const PruneArgs& a, uint32_t k) { return a.b.items ? a.b.items[a.first + k] : a.first + k; }

// selector of query column q of reads A and B for the v_perm lookup (zsw_score_v2.hpp (3)); WIDE: the two residues themselves
template <bool WIDE>
__device__ __forceinline__ uint32_t column_selector(const BatchDev& b, const uint8_t* lut, uint32_t q, uint64_t offA, uint32_t lenA,
                                                    uint64_t offB, uint32_t lenB) {
    uint32_t kA = PAD_K, kB = PAD_K;
    if (q < lenA) kA = lut[b.bases[offA + q]];
    if (q < lenB) kB = lut[b.bases[offB + q]];
    if (WIDE) return (kA == PAD_K ? (uint32_t)WIDE_PAD : kA) | ((kB == PAD_K ? (uint32_t)WIDE_PAD : kB) << 16);
    const uint32_t sA = kA < 4 ? (2 * kA + 1) | ((8 + kA) << 8) : (kA == PAD_K ? 0x0c00u : (2 * (kA - 3)) | 0x0c00u);
    const uint32_t sB = kB < 4 ? (2 * kB + 1) | ((8 + kB) << 8) : (kB == PAD_K ? 0x0c00u : (2 * (kB - 3)) | 0x0c00u);
    return sA | (sB << 16);
}

// WIDE, 24 columns: three waves per SIMD — at four (128 VGPRs) the two LDS lookups per column spill inside the row loop
template <int C, bool WIDE>
__global__ __launch_bounds__(BLOCK, (WIDE && C <= 24) ? 3 : min_waves(C, 0)) void prune_strip_kernel(PruneArgs a) {
    __shared__ uint2 rp[WIDE ? 1 : CH];
    __shared__ uint16_t rpw[WIDE ? CH : 1];       // WIDE: byte offset of each staged row's table row
    __shared__ uint32_t wt32[WIDE ? 33 * 9 : 1];  // WIDE: the score table
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    __shared__ __attribute__((aligned(16))) uint2 stage_all[BLOCK * 10];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);
    const int tid = threadIdx.x;
    const uint32_t pair = blockIdx.x * BLOCK + tid;
    const bool valid = pair < a.n_pairs;
    const bool validB = valid && 2 * pair + 1 < a.n;
    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    if (tid < 9) swt[tid] = make_uint2(a.wtab[tid][0], a.wtab[tid][1]);
    if (WIDE) {
        for (int i = tid; i < 33 * 9; i += BLOCK) wt32[WIDE ? i : 0] = reinterpret_cast<const uint32_t*>(a.wide)[i];
    }
    __syncthreads();
    const int8_t* wt = reinterpret_cast<const int8_t*>(wt32);
    // column score of the current row for both reads, as a packed pair of i16 (score + ge)
    auto lookup = [&](const uint2 ww, const uint32_t sl) -> uint32_t {
        if constexpr (WIDE) {
            const int sa = wt[ww.x + (sl & 0xffffu)], sb = wt[ww.x + (sl >> 16)];
            return __builtin_amdgcn_perm((uint32_t)sb, (uint32_t)sa, 0x05040100u);
        } else {
            return __builtin_amdgcn_perm(ww.y, ww.x, sl);
        }
    };
    uint64_t offA = 0, offB = 0;
    uint32_t lenA = 0, lenB = 0;
    if (valid) read_span(a.b, read_id(a, 2 * pair), &offA, &lenA);
    if (validB) read_span(a.b, read_id(a, 2 * pair + 1), &offB, &lenB);
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sel[c] = column_selector<WIDE>(a.b, lut, (uint32_t)c, offA, lenA, offB, lenB);

    const uint32_t ge2 = a.ge2, gd2 = a.gd2;
    const uint32_t ge1 = ge2 & 0xffffu;
    const uint32_t K = a.K;
    const uint32_t Kge2 = (K * ge1) * 0x00010001u;
    uint32_t Dr = (a.floor0 - ge1) * 0x00010001u;  // D of row -1
    uint32_t H[C], E[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        H[c] = Dr;
        E[c] = pk_addu(Dr, ge2);
    }
    uint32_t best = 0, mH = 0, mF = 0, ancH = 0;
    uint32_t ancA = 0, ancB = 0;
    const int R = (int)a.ref_len;
    uint2* blk = a.blk + (size_t)(valid ? pair : 0) * (size_t)a.nblk;
    constexpr int STAGE_STRIDE = 10;  // uint2 per lane: 8 rows + 2 of padding (80 bytes: 16-byte aligned, spreads the banks)
    const int lane = tid & 63;
    uint2* stage = stage_all + (tid >> 6) * 64 * STAGE_STRIDE;
    const uint32_t wave_pair0 = blockIdx.x * BLOCK + (uint32_t)(tid & ~63);

    for (int base = 0; base < R; base += CH) {
        __syncthreads();
        for (int j = tid; j < CH; j += BLOCK) {
            const int row = base + j;
            if (WIDE) rpw[WIDE ? j : 0] = (uint16_t)((row < R ? (int)lut[a.ref[row]] : WIDE_NEUTRAL) * WIDE_STRIDE);
            else rp[WIDE ? 0 : j] = swt[row < R ? (int)lut[a.ref[row]] : NEUTRAL];
        }
        __syncthreads();
        const int tend = (R < base + CH) ? R : base + CH;
        uint2 w = WIDE ? make_uint2(rpw[0], 0u) : rp[0];
#pragma unroll 1
        for (int t = base; t < tend; ++t) {
            const uint2 wn = WIDE ? make_uint2(rpw[WIDE ? (t + 1 - base) & (CH - 1) : 0], 0u) : rp[WIDE ? 0 : (t + 1 - base) & (CH - 1)];
            if (ge1 != 0 && t > 0 && (t & (int)(K - 1)) == 0) {  // re-base (every lane is in the same row)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    H[c] = pk_subu(H[c], Kge2);
                    E[c] = pk_subu(E[c], Kge2);
                }
                Dr = pk_subu(Dr, Kge2);
            }
            const uint32_t Dp = Dr;         // D_{r-1}: the (zero) H left of column 0 in the previous row
            Dr = pk_addu(Dr, ge2);          // D_r
            const uint32_t Dn = pk_addu(Dr, ge2);
            uint32_t hd = pk_addu(Dp, lookup(w, sel[0]));
            uint32_t F = Dr;
            uint32_t rmax = 0x04000400u;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                uint32_t hd_next = 0;
                if (c + 1 < C) hd_next = pk_addu(H[c], lookup(w, sel[c + 1 < C ? c + 1 : c]));
                const uint32_t h = pk_max3(hd, E[c], F);
                H[c] = h;
                const uint32_t hg = h - gd2;
                E[c] = pk_max3(E[c], hg, Dn);
                F = pk_max3(F, hg, Dn) - ge2;
                if (c & 1) rmax = pk_max3(rmax, H[c - (c & 1)], h);
                else if (c == C - 1) rmax = pk_max3(rmax, h, h);
                hd = hd_next;
            }
            const uint32_t tH = pk_subu(H[C - 1], Dr), tF = pk_subu(F, Dr);  // true scores leaving the strip in this row
            // eight rows per lane gather in LDS; then four lanes write one pair's 64 bytes, so every store instruction fills
            // sixteen whole lines (one 8-byte store per lane and row would touch 64 lines per instruction: 17 -> 12 ms per 2 M reads)
            stage[lane * STAGE_STRIDE + (t & 7)] = make_uint2(tH, tF);
            if ((t & 7) == 7 || t == R - 1) {
                __builtin_amdgcn_wave_barrier();
                const int r8 = t & ~7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pw = 16 * i + (lane >> 2), q = lane & 3;  // pair of the wavefront, quarter of its line
                    const uint4 v = *reinterpret_cast<const uint4*>(&stage[pw * STAGE_STRIDE + 2 * q]);
                    if (wave_pair0 + (uint32_t)pw < a.n_pairs)
                        *reinterpret_cast<uint4*>(a.bnd + (size_t)(wave_pair0 + (uint32_t)pw) * (size_t)a.row_stride + (size_t)(r8 + 2 * q)) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            mH = pk_maxu(mH, tH);
            mF = pk_maxu(mF, tF);
            const uint32_t na = pk_maxu(ancH, tH);
            const uint32_t ch = na ^ ancH;
            if (ch & 0xffffu) ancA = (uint32_t)t;
            if (ch >> 16) ancB = (uint32_t)t;
            ancH = na;
            best = pk_maxu(best, pk_subu(rmax, Dr));
            if ((t & (PR_BLK - 1)) == PR_BLK - 1 || t == R - 1) {
                if (valid) blk[t / PR_BLK] = make_uint2(mH, mF);
                mH = 0;
                mF = 0;
            }
            w = wn;
        }
    }
    if (valid) {
        a.best0[pair] = best;
        a.anchor[2 * pair] = ancA | (2 * pair >= a.split ? 1u << 24 : 0u);
        if (validB) a.anchor[2 * pair + 1] = ancB | (2 * pair + 1 >= a.split ? 1u << 24 : 0u);
    }
}

template <int CP, int G, int C, int MODE, bool WIDE>
__global__ __launch_bounds__(BLOCK, min_waves(C, MODE)) void prune_window_kernel(PruneArgs a) {
    constexpr int M2 = PR_M2 + 8 * (G / 8);  // longer reads: more room for deletions below the anchor
    __shared__ uint2 rp[WIDE ? 1 : CH + 2 * G];
    __shared__ uint16_t rpw[WIDE ? CH + 2 * G : 1];  // WIDE: byte offset of each staged row's table row
    __shared__ uint32_t wt32[WIDE ? 33 * 9 : 1];     // WIDE: the score table
    __shared__ uint8_t spot[WIDE ? 36 : 1];          // WIDE: potential of a column by its residue
    __shared__ int s_lo, s_hi;
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);
    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;
    const uint32_t itemA = 2 * group, itemB = 2 * group + 1;
    const bool validA = itemA < a.n, validB = itemB < a.n;
    const uint32_t ridA = validA ? a.order[itemA] : 0;      // read of the chunk
    const uint32_t ridB = validB ? a.order[itemB] : ridA;   // an absent B mirrors A's boundary; its columns are padding
    const int R = (int)a.ref_len;
    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    if (tid < 9) swt[tid] = make_uint2(a.wtab[tid][0], a.wtab[tid][1]);
    if (tid == 0) {
        s_lo = 0x7fffffff;
        s_hi = 0;
    }
    if (WIDE) {
        for (int i = tid; i < 33 * 9; i += BLOCK) wt32[WIDE ? i : 0] = reinterpret_cast<const uint32_t*>(a.wide)[i];
        if (tid < 36) spot[WIDE ? tid : 0] = a.colpot[tid];
    }
    __syncthreads();
    const int8_t* wt = reinterpret_cast<const int8_t*>(wt32);
    auto lookup = [&](const uint2 ww, const uint32_t sl) -> uint32_t {
        if constexpr (WIDE) {
            const int sa = wt[ww.x + (sl & 0xffffu)], sb = wt[ww.x + (sl >> 16)];
            return __builtin_amdgcn_perm((uint32_t)sb, (uint32_t)sa, 0x05040100u);
        } else {
            return __builtin_amdgcn_perm(ww.y, ww.x, sl);
        }
    };

    uint64_t offA = 0, offB = 0;
    uint32_t lenA = 0, lenB = 0;
    const uint32_t idA = validA ? read_id(a, ridA) : 0, idB = validB ? read_id(a, ridB) : 0;
    if (validA) read_span(a.b, idA, &offA, &lenA);
    if (validB) read_span(a.b, idB, &offB, &lenB);
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sel[c] = column_selector<WIDE>(a.b, lut, (uint32_t)(CP + g * C + c), offA, lenA, offB, lenB);

    // the window: blocks of rows around the two anchors
    const int rsA = validA ? (int)(a.anchor[ridA] & 0xffffffu) : 0, rsB = validB ? (int)(a.anchor[ridB] & 0xffffffu) : rsA;
    const int lo = min(rsA, rsB), hi = max(rsA, rsB);
    const int maxlen = (int)max(lenA, lenB);
    const int Rup = (R + PR_BLK - 1) / PR_BLK * PR_BLK;
    const int a0 = max(0, lo - PR_M1) / PR_BLK * PR_BLK;
    const int b0 = min(Rup, (hi + 1 + max(0, maxlen - CP) + M2 + PR_BLK - 1) / PR_BLK * PR_BLK);
    int Tw = validA ? b0 - a0 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) Tw = max(Tw, __shfl_xor(Tw, d, 64));  // the wavefront's groups walk equally many rows
    const int b1 = min(Rup, a0 + Tw);
    // the block's row table: rows [lo - (G - 1), hi + G) of the reference (real rows for the lanes' skewed last rows too:
    // V3 needs every lane's final state to sit in a real row, or past the end of the reference)
    if (g == 0 && validA) {
        atomicMin(&s_lo, a0);
        atomicMax(&s_hi, a0 + Tw);
    }
    __syncthreads();
    const int blo = s_lo, bhi = s_hi;
    const bool staged = bhi - blo + 2 * G <= CH + 2 * G && blo <= bhi;  // reads sorted by anchor: the windows of a block overlap
    if (staged) {
        for (int j = tid; j < bhi - blo + 2 * G; j += BLOCK) {
            const int row = blo - (G - 1) + j;
            if (WIDE) rpw[WIDE ? j : 0] = (uint16_t)(((row >= 0 && row < R) ? (int)lut[a.ref[row]] : WIDE_NEUTRAL) * WIDE_STRIDE);
            else rp[WIDE ? 0 : j] = swt[(row >= 0 && row < R) ? (int)lut[a.ref[row]] : NEUTRAL];
        }
    }
    __syncthreads();
    // table entry of this lane's row at step t: joff + t. Lane groups past the batch (validA false: the tail of the last block)
    // keep a0 = 0 while the block's table starts at blo: they read from entry 0 on, inside the table
    const int joff = validA ? a0 - blo + (G - 1) - g : 0;

    // the boundary streams of the two reads
    const uint2* bA = a.bnd + (size_t)(ridA >> 1) * (size_t)a.row_stride;
    const uint2* bB = a.bnd + (size_t)(ridB >> 1) * (size_t)a.row_stride;
    const uint32_t selH = ((ridA & 1) ? 0x0302u : 0x0100u) | (((ridB & 1) ? 0x0706u : 0x0504u) << 16);

    const uint32_t ge2 = a.ge2, gd2 = a.gd2;
    const uint32_t ge1 = ge2 & 0xffffu;
    const uint32_t K = a.K;
    const uint32_t Kge2 = (K * ge1) * 0x00010001u;
    uint32_t Dr = (a.floor0 - (uint32_t)(g + 1) * ge1) * 0x00010001u;
    uint32_t H[C], E[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        H[c] = Dr;
        E[c] = pk_addu(Dr, ge2);
    }
    uint32_t best = 0;
    uint32_t snap[MODE == 2 ? C : 1];  // MODE 2: the H row of each read's latest rise (as in score_kernel_v2)
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = 0;
    }
    uint32_t snapD = 0;
    int rA = 0, rB = 0;  // window row of the lane's latest rise
    uint32_t Fout = Dr, Hlast = Dr, Hin_prev = Dr;
    uint2 ldA = make_uint2(0u, 0u), ldB = make_uint2(0u, 0u);
    if (g == 0 && validA && a0 < R) {
        ldA = bA[a0];
        ldB = bB[a0];
    }
    const int T = staged ? Tw + G - 1 : 0;
    auto row_entry = [&](int j) -> uint2 { return WIDE ? make_uint2(rpw[WIDE ? j : 0], 0u) : rp[WIDE ? 0 : j]; };
    uint2 w = row_entry(staged ? joff : 0);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const uint2 wn = row_entry(joff + t + 1);
        const int row = t - g;  // row of the window
        const bool rebase = ge1 != 0 && row > 0 && (row & (int)(K - 1)) == 0;
        if (__ballot(rebase) != 0) {
            const uint32_t adj = rebase ? Kge2 : 0u;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                H[c] = pk_subu(H[c], adj);
                E[c] = pk_subu(E[c], adj);
            }
            Hin_prev = pk_subu(Hin_prev, adj);
            Dr = pk_subu(Dr, adj);
        }
        Dr = pk_addu(Dr, ge2);
        const uint32_t Dn = pk_addu(Dr, ge2);
        uint32_t Fin = (uint32_t)__shfl_up((int)Fout, 1, G);
        uint32_t Hin = (uint32_t)__shfl_up((int)Hlast, 1, G);
        if (g == 0) {
            const int grow = a0 + t;  // reference row
            Fin = Dr;
            Hin = Dr;
            if (validA && grow < R) {
                Hin = pk_addu(Dr, __builtin_amdgcn_perm(ldB.x, ldA.x, selH));
                Fin = pk_addu(Dr, __builtin_amdgcn_perm(ldB.y, ldA.y, selH));
            }
            if (validA && grow + 1 < R) {
                ldA = bA[grow + 1];
                ldB = bB[grow + 1];
            }
        }
        uint32_t hd = pk_addu(Hin_prev, lookup(w, sel[0]));
        Hin_prev = Hin;
        uint32_t F = Fin;
        uint32_t rmax = 0x04000400u;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t hd_next = 0;
            if (c + 1 < C) hd_next = pk_addu(H[c], lookup(w, sel[c + 1 < C ? c + 1 : c]));
            const uint32_t h = pk_max3(hd, E[c], F);
            H[c] = h;
            const uint32_t hg = h - gd2;
            E[c] = pk_max3(E[c], hg, Dn);
            F = pk_max3(F, hg, Dn) - ge2;
            if (c & 1) rmax = pk_max3(rmax, H[c - (c & 1)], h);
            else if (c == C - 1) rmax = pk_max3(rmax, h, h);
            hd = hd_next;
        }
        Fout = F;
        Hlast = H[C - 1];
        const uint32_t nb = pk_maxu(best, pk_subu(rmax, Dr));
        if (MODE != 0) {
            const uint32_t ch = nb ^ best;
            if (ch & 0xffffu) rA = row;
            if (ch >> 16) rB = row;
            if (MODE == 2) {
                const uint32_t m = ((ch & 0xffffu) ? 0xffffu : 0u) | ((ch >> 16) ? 0xffff0000u : 0u);
#pragma unroll
                for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = (H[c] & m) | (snap[MODE == 2 ? c : 0] & ~m);
                snapD = (Dr & m) | (snapD & ~m);
            }
        }
        best = nb;
        w = wn;
    }

    // ---- the read's score so far, and the three checks ----
    const int lbA = (int)(best & 0xffffu), lbB = (int)(best >> 16);
    int bA2 = lbA, bB2 = lbB;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        bA2 = max(bA2, __shfl_xor(bA2, d, G));
        bB2 = max(bB2, __shfl_xor(bB2, d, G));
    }
    // ends (MODE 1, 2): first row holding the window's maximum, then the first column of that row (striped.rs:296-321)
    uint32_t reA = 0, reB = 0, qeA = 0, qeB = 0;
    if (MODE != 0) {
        int kA = (lbA == bA2) ? rA : 0x7fffffff, kB = (lbB == bB2) ? rB : 0x7fffffff;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) {
            kA = min(kA, __shfl_xor(kA, d, G));
            kB = min(kB, __shfl_xor(kB, d, G));
        }
        reA = (uint32_t)(a0 + kA) + 1;
        reB = (uint32_t)(a0 + kB) + 1;
        if (MODE == 2) {
            int cA = 0x7fffffff, cB = 0x7fffffff;
            const int sdA = (int)(snapD & 0xffffu), sdB = (int)(snapD >> 16);
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t sv = snap[MODE == 2 ? c : 0];
                if ((int)(sv & 0xffffu) - sdA == bA2) cA = CP + g * C + c;
                if ((int)(sv >> 16) - sdB == bB2) cB = CP + g * C + c;
            }
            if (!(lbA == bA2 && rA == kA)) cA = 0x7fffffff;
            if (!(lbB == bB2 && rB == kB)) cB = 0x7fffffff;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) {
                cA = min(cA, __shfl_xor(cA, d, G));
                cB = min(cB, __shfl_xor(cB, d, G));
            }
            qeA = (uint32_t)cA + 1;
            qeB = (uint32_t)cB + 1;
        }
    }
    const uint32_t s0A = validA ? a.best0[ridA >> 1] : 0, s0B = validB ? a.best0[ridB >> 1] : 0;
    const int stripA = (int)((ridA & 1) ? s0A >> 16 : s0A & 0xffffu), stripB = (int)((ridB & 1) ? s0B >> 16 : s0B & 0xffffu);
    const int SA = max(bA2, stripA), SB = max(bB2, stripB);
    const int maxw = (int)a.maxw;
    const int remA = max(0, (int)lenA - CP), remB = max(0, (int)lenB - CP);
    // What the columns from a given one on can add to a path at most: maxw per column with the v_perm tables; WIDE: the columns'
    // own potentials by residue (padding: 0), summed from the right — inside the lane, then over the lanes to its right.
    int rightA = 0, rightB = 0, nfA = 0, nfB = 0, PcpA = maxw * remA, PcpB = maxw * remB, Pcp1A = maxw * max(0, remA - 1), Pcp1B = maxw * max(0, remB - 1);
    if constexpr (WIDE) {
        int totA = 0, totB = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            totA += (int)spot[sel[c] & 0xffffu];
            totB += (int)spot[sel[c] >> 16];
        }
        const int p0A = (int)spot[sel[0] & 0xffffu], p0B = (int)spot[sel[0] >> 16];
#pragma unroll
        for (int d = 1; d < G; ++d) {
            const int xa = __shfl_down(totA, d, G), xb = __shfl_down(totB, d, G);
            if (g + d < G) {
                rightA += xa;
                rightB += xb;
            }
        }
        nfA = __shfl_down(p0A, 1, G);
        nfB = __shfl_down(p0B, 1, G);
        if (g + 1 >= G) nfA = nfB = 0;
        PcpA = __shfl(totA + rightA, 0, G);
        PcpB = __shfl(totB + rightB, 0, G);
        Pcp1A = __shfl(totA + rightA - p0A, 0, G);
        Pcp1B = __shfl(totB + rightB - p0B, 0, G);
    }
    // V3: what can still leave this lane's last row (E holds the next row's E; Fout moves right into the next lane's columns)
    int v3A = 0, v3B = 0;
    if (a0 + Tw < R) {
        const int dA = (int)(Dr & 0xffffu), dB = (int)(Dr >> 16);
        if constexpr (WIDE) {
            int accA = rightA, accB = rightB;  // the columns right of column c
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t he = pk_maxu(H[c], pk_subu(E[c], ge2));
                v3A = max(v3A, (int)(he & 0xffffu) - dA + accA);
                v3B = max(v3B, (int)(he >> 16) - dB + accB);
                accA += (int)spot[sel[c] & 0xffffu];
                accB += (int)spot[sel[c] >> 16];
            }
            // the outgoing F opens the next lane's first column as a gap: the columns after that one
            v3A = max(v3A, (int)(Fout & 0xffffu) - dA + rightA - nfA);
            v3B = max(v3B, (int)(Fout >> 16) - dB + rightB - nfB);
            if (g > 0) {  // the diagonal from the left neighbour's last column, one row up, into the first cell below this lane's rows
                v3A = max(v3A, (int)(Hin_prev & 0xffffu) - dA + accA);
                v3B = max(v3B, (int)(Hin_prev >> 16) - dB + accB);
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int col = CP + g * C + c;
                const uint32_t he = pk_maxu(H[c], pk_subu(E[c], ge2));
                v3A = max(v3A, (int)(he & 0xffffu) - dA + maxw * max(0, (int)lenA - col - 1));
                v3B = max(v3B, (int)(he >> 16) - dB + maxw * max(0, (int)lenB - col - 1));
            }
            const int colr = CP + (g + 1) * C;
            v3A = max(v3A, (int)(Fout & 0xffffu) - dA + maxw * max(0, (int)lenA - colr - 1));
            v3B = max(v3B, (int)(Fout >> 16) - dB + maxw * max(0, (int)lenB - colr - 1));
            if (g > 0) {  // the diagonal from the left neighbour's last column, one row up, into the first cell below this lane's rows
                v3A = max(v3A, (int)(Hin_prev & 0xffffu) - dA + maxw * max(0, (int)lenA - (CP + g * C)));
                v3B = max(v3B, (int)(Hin_prev >> 16) - dB + maxw * max(0, (int)lenB - (CP + g * C)));
            }
        }
    }
    // V2: crossings in rows outside the window, block by block
    int v2A = 0, v2B = 0;
    if (validA) {
        const uint2* kA = a.blk + (size_t)(ridA >> 1) * (size_t)a.nblk;
        const uint2* kB = a.blk + (size_t)(ridB >> 1) * (size_t)a.nblk;
        const int k0 = a0 / PR_BLK, k1 = b1 / PR_BLK;  // blocks [k0, k1) are the window
        for (int k = g; k < (int)a.nblk; k += G) {
            if (k >= k0 && k < k1) continue;
            const uint2 xA = kA[k], xB = kB[k];
            const int hA = (int)((ridA & 1) ? xA.x >> 16 : xA.x & 0xffffu), fA = (int)((ridA & 1) ? xA.y >> 16 : xA.y & 0xffffu);
            const int hB = (int)((ridB & 1) ? xB.x >> 16 : xB.x & 0xffffu), fB = (int)((ridB & 1) ? xB.y >> 16 : xB.y & 0xffffu);
            if (remA > 0) v2A = max(v2A, max(hA + PcpA, fA + Pcp1A));
            if (remB > 0) v2B = max(v2B, max(hB + PcpB, fB + Pcp1B));
        }
    }
    int wA = max(v2A, v3A), wB = max(v2B, v3B);
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        wA = max(wA, __shfl_xor(wA, d, G));
        wB = max(wB, __shfl_xor(wB, d, G));
    }
    wA = max(wA, PcpA);  // V1
    wB = max(wB, PcpB);
    if (g < 2) {
        const bool second = g == 1;
        const bool v = second ? validB : validA;
        if (v) {
            const uint32_t id = second ? idB : idA;
            const uint32_t len = second ? lenB : lenA;
            const int S = second ? SB : SA, bound = second ? wB : wA;
            // score only: a path outside the computed cells matters if it can score MORE than S. With ends it also matters
            // if it can score S (it could end in an earlier row), and the maximum has to lie right of the strip, whose cells keep
            // no coordinates.
            const bool redo = !staged || (MODE == 0 ? bound > S : (bound >= S || (second ? stripB : stripA) >= S));
            if (len == 0) {
                a.out.score[id] = 0;
                a.out.status[id] = ZSW_STATUS_EMPTY;
                if (a.out.tier) a.out.tier[id] = 0;
                if (MODE != 0 && a.out.ref_end) a.out.ref_end[id] = 0;
                if (MODE == 2 && a.out.query_end) a.out.query_end[id] = 0;
            } else if (redo) {  // all cells for this read
                const uint32_t k = atomicAdd(a.fail_count, 1u);
                a.fail_list[k] = id;
            } else {
                uint32_t score;
                uint8_t status, tier;
                apply_rule(a.rule, (uint64_t)S, &score, &status, &tier);
                a.out.score[id] = score;
                a.out.status[id] = status;
                if (a.out.tier) a.out.tier[id] = tier;
                const bool some = status == ZSW_STATUS_SOME;
                if (MODE != 0 && a.out.ref_end) a.out.ref_end[id] = some ? (second ? reB : reA) : 0;
                if (MODE == 2 && a.out.query_end) a.out.query_end[id] = some ? (second ? qeB : qeA) : 0;
            }
        }
    }
}

__global__ void iota32_kernel(uint32_t* v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// bail-out: entries [first, n) of the batch join the list of reads to be scored over all their cells
__global__ void append_rest_kernel(const uint32_t* items, uint32_t first, uint32_t n, uint32_t* fail_list, const uint32_t* fail_count) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fail_list[*fail_count + (i - first)] = items ? items[i] : i;
}

__global__ void add_const_kernel(uint32_t* count, uint32_t v) { *count += v; }

size_t round256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

size_t prune_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n, 0, 32, (hipStream_t)0);
    return bytes;
}

size_t prune_workspace_bytes(uint32_t chunk_reads, uint32_t ref_len) {
    const size_t pairs = ((size_t)chunk_reads + 1) / 2, nblk = ((size_t)ref_len + PR_BLK - 1) / PR_BLK;
    return round256(pairs * (((size_t)ref_len + 7) & ~(size_t)7) * sizeof(uint2)) + round256(pairs * nblk * sizeof(uint2)) + round256(pairs * 4) +
           4 * round256((size_t)chunk_reads * 4 + 8) + round256(prune_sort_temp_bytes(chunk_reads)) + 256;
}

// reads per round of the two kernels: as many as PR_WORK_BYTES of boundary stream hold (8 bytes per pair and reference row)
uint32_t prune_chunk_reads(uint32_t n_reads, uint32_t ref_len) {
    const size_t per_pair = (((size_t)ref_len + 7) & ~(size_t)7) * sizeof(uint2) + ((size_t)ref_len / PR_BLK + 1) * sizeof(uint2) + 64;
    const size_t pairs = std::max<size_t>(PR_WORK_BYTES / per_pair, 1024);
    return (uint32_t)std::min<size_t>(std::min<size_t>(2 * pairs, PR_CHUNK_READS), std::max<uint32_t>(n_reads, 2));
}

int prune_class_for(uint32_t max_len) {
    for (int k = 0; k < PR_N_CLASSES; ++k)
        if (max_len <= kPruneClasses[k].max_len) return max_len > (uint32_t)kPruneClasses[k].cp + 40 ? k : -1;
    return -1;
}

bool prune_applicable(const ScoringDev& s, uint32_t max_len, uint32_t ref_len, uint32_t limit) {
    int maxw = 0;
    for (int i = 0; i < s.S * s.S; ++i) maxw = std::max(maxw, (int)s.w[i]);
    if (maxw <= 0 || ref_len == 0 || prune_class_for(max_len) < 0) return false;
    return (uint64_t)maxw * max_len + 8 < limit;  // no score can leave the packed range
}

namespace {

template <int CP, int G, int C, bool WIDE>
void launch_window_t(const PruneArgs& a, int mode, hipStream_t stream) {
    const uint32_t groups = (a.n + 1) / 2, per_block = BLOCK / G;
    const dim3 grid((groups + per_block - 1) / per_block);
    if (mode == 0) hipLaunchKernelGGL((prune_window_kernel<CP, G, C, 0, WIDE>), grid, dim3(BLOCK), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((prune_window_kernel<CP, G, C, 1, WIDE>), grid, dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((prune_window_kernel<CP, G, C, 2, WIDE>), grid, dim3(BLOCK), 0, stream, a);
}

template <int CP, int G, int C>
void launch_window(const PruneArgs& a, bool wide, int mode, hipStream_t stream) {
    if (wide) launch_window_t<CP, G, C, true>(a, mode, stream);
    else launch_window_t<CP, G, C, false>(a, mode, stream);
}

template <int CP>
void launch_strip(const PruneArgs& a, bool wide, dim3 grid, hipStream_t stream) {
    if (wide) hipLaunchKernelGGL((prune_strip_kernel<CP, true>), grid, dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((prune_strip_kernel<CP, false>), grid, dim3(BLOCK), 0, stream, a);
}

template <int CP>
hipError_t strip_blocks_per_cu(bool wide, int* per_cu) {
    return wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, prune_strip_kernel<CP, true>, BLOCK, 0)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, prune_strip_kernel<CP, false>, BLOCK, 0);
}

}  // namespace

// Scores items [0, a2.b.n_items) of the batch: the first n_cls of them are reads of class `cls`, the rest of class cls + 1 (same
// strip width: one strip launch serves both, so that a small class does not cost a round of its own). Reads that fail a check
// are appended to fail_list (device count in fail_count, zeroed by the caller). `a2` carries the v2 tables; floor_strip /
// floor_window[2] are the drift floors for G = 1 and for the two classes' G.
hipError_t launch_score_pruned(const ScoreArgsV2& a2, int cls, uint32_t n_cls, uint32_t floor_strip, const uint32_t* floor_window,
                               const ScoringDev& h_sc, uint8_t* work, size_t work_bytes, uint32_t chunk_reads, uint32_t* fail_list,
                               uint32_t* fail_count, int mode, bool wide, hipStream_t stream, uint32_t* est_failed) {
    const uint32_t n = a2.b.n_items, R = a2.ref_len;
    if (est_failed) *est_failed = 0xffffffffu;
    if (n == 0) return hipSuccess;
    if (cls < 0 || cls >= PR_N_CLASSES || chunk_reads < 2 || work_bytes < prune_workspace_bytes(chunk_reads, R)) return hipErrorNotSupported;
    if (n_cls < n && (cls + 1 >= PR_N_CLASSES || kPruneClasses[cls + 1].cp != kPruneClasses[cls].cp || R >= (1u << 24))) return hipErrorNotSupported;
    PruneArgs a;
    a.b = a2.b;
    a.ref = a2.ref;
    a.ref_len = R;
    a.sc = a2.sc;
    for (int r = 0; r < 9; ++r) {
        a.wtab[r][0] = a2.wtab[r][0];
        a.wtab[r][1] = a2.wtab[r][1];
    }
    for (int i = 0; i < 33 * 36; ++i) a.wide[i] = wide ? a2.wide[i] : (int8_t)0;
    for (int q = 0; q < 36; ++q) {  // what a query column holding residue q can add to a path at most
        int pot = 0;
        if (wide && q < h_sc.S)
            for (int r = 0; r < h_sc.S; ++r) pot = std::max(pot, (int)h_sc.w[r * h_sc.S + q]);
        a.colpot[q] = (uint8_t)std::min(pot, 255);
    }
    a.ge2 = a2.ge2;
    a.gd2 = a2.gd2;
    a.K = a2.K;
    a.rule = a2.rule;
    a.out = a2.out;
    int maxw = 0;
    for (int i = 0; i < h_sc.S * h_sc.S; ++i) maxw = std::max(maxw, (int)h_sc.w[i]);
    a.maxw = (uint32_t)maxw;
    a.nblk = (R + PR_BLK - 1) / PR_BLK;
    a.fail_list = fail_list;
    a.fail_count = fail_count;
    const size_t pairs_cap = ((size_t)chunk_reads + 1) / 2;
    uint8_t* p = work;
    a.row_stride = (R + 7) & ~7u;
    a.bnd = reinterpret_cast<uint2*>(p);
    p += round256(pairs_cap * a.row_stride * sizeof(uint2));
    a.blk = reinterpret_cast<uint2*>(p);
    p += round256(pairs_cap * a.nblk * sizeof(uint2));
    a.best0 = reinterpret_cast<uint32_t*>(p);
    p += round256(pairs_cap * 4);
    const size_t per = round256((size_t)chunk_reads * 4 + 8);
    a.anchor = reinterpret_cast<uint32_t*>(p);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(p + per);
    uint32_t* ids_in = reinterpret_cast<uint32_t*>(p + 2 * per);
    uint32_t* ids_out = reinterpret_cast<uint32_t*>(p + 3 * per);
    p += 4 * per;
    void* temp = p;
    size_t temp_bytes = prune_sort_temp_bytes(chunk_reads);
    int key_bits = 1;
    while ((1u << key_bits) < R + 1 && key_bits < 32) ++key_bits;
    if (n_cls < n) key_bits = 25;  // bit 24: the second class of the range

    uint32_t probe_reads = 0;
    // a round of the strip kernel should not spill a few blocks into a second wave of blocks: every block walks all R rows,
    // so 545 blocks on 512 slots take twice as long as 512. When the batch needs several rounds, a round is a whole multiple of
    // what the chip holds at once.
    {
        int dev = 0, cus = 0, per_cu = 0;
        hipError_t qe = hipGetDevice(&dev);
        if (qe == hipSuccess) qe = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (qe == hipSuccess)
            qe = kPruneClasses[cls].cp == 24 ? strip_blocks_per_cu<24>(wide, &per_cu) : strip_blocks_per_cu<48>(wide, &per_cu);
        if (qe != hipSuccess) return qe;
        const uint64_t quantum = (uint64_t)cus * (uint64_t)per_cu * 2 * BLOCK;  // reads the chip holds at once
        if (quantum > 0 && n > chunk_reads && chunk_reads >= quantum) chunk_reads = (uint32_t)(chunk_reads / quantum * quantum);
        // Bail-out probe: a batch of several chip-fulls runs one of them first and reads the number of handed-back reads back.
        // Strip + window cost about a fifth of the full pass, so above PR_BAIL_PERCENT handed back the pass no longer pays and
        // the rest of the batch is scored over all its cells at once (unrelated or strongly diverged reads: 1.2x -> 1.0x + the probe).
        if (quantum > 0 && quantum <= chunk_reads && n >= 2 * quantum) probe_reads = (uint32_t)quantum;
    }
    for (uint32_t first = 0; first < n; first += (first == 0 && probe_reads ? probe_reads : chunk_reads)) {
        const uint32_t this_chunk = first == 0 && probe_reads ? probe_reads : chunk_reads;
        if (probe_reads && first == probe_reads) {
            uint32_t failed = 0;
            hipError_t be = hipMemcpyAsync(&failed, fail_count, sizeof(failed), hipMemcpyDeviceToHost, stream);
            if (be == hipSuccess) be = hipStreamSynchronize(stream);
            if (be != hipSuccess) return be;
            if (est_failed) *est_failed = (uint32_t)std::min<uint64_t>((uint64_t)failed * n / probe_reads, n);
            if ((uint64_t)failed * 100 > (uint64_t)probe_reads * PR_BAIL_PERCENT) {
                if (est_failed) *est_failed = n;
                hipLaunchKernelGGL(append_rest_kernel, dim3((n - first + 255) / 256), dim3(256), 0, stream, a.b.items, first, n, fail_list, fail_count);
                hipLaunchKernelGGL(add_const_kernel, dim3(1), dim3(1), 0, stream, fail_count, n - first);
                return hipGetLastError();
            }
        }
        a.first = first;
        a.n = std::min<uint32_t>(this_chunk, n - first);
        a.n_pairs = (a.n + 1) / 2;
        a.split = first >= n_cls ? 0u : std::min<uint32_t>(a.n, n_cls - first);  // entries of the chunk that belong to `cls`
        a.floor0 = floor_strip;
        const dim3 sgrid((a.n_pairs + BLOCK - 1) / BLOCK);
        if (kPruneClasses[cls].cp == 24) launch_strip<24>(a, wide, sgrid, stream);
        else launch_strip<48>(a, wide, sgrid, stream);
        hipLaunchKernelGGL(iota32_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, ids_in, a.n);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t*)a.anchor, keys_out, (const uint32_t*)ids_in, ids_out,
                                                          (int)a.n, 0, key_bits, stream);
        if (e != hipSuccess) return e;
        // the window kernel per class: the sorted order holds the reads of `cls` first, then those of the next class
        const uint32_t n_chunk = a.n, n_first = a.split;
        for (int part = 0; part < 2; ++part) {
            const int c = cls + part;
            a.order = ids_out + (part ? n_first : 0);
            a.n = part ? n_chunk - n_first : n_first;
            if (a.n == 0) continue;
            a.floor0 = floor_window[part];
            if (c == 0) launch_window<24, 4, 32>(a, wide, mode, stream);
            else if (c == 1) launch_window<48, 8, 32>(a, wide, mode, stream);
            else launch_window<48, 16, 22>(a, wide, mode, stream);
        }
        a.n = n_chunk;
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace zsw
