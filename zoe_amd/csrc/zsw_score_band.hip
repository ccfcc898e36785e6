// zsw_score_band.hip — the banded form of the seeded exact pass (sw_simd_score / sw_simd_score_ends, striped.rs:65-142, 153-336:
// MODE 0 score only, 1 with the reference end, 2 with both ends, 3 = 2 + whether the maximum sits in exactly one cell: then the
// tie rule does not matter and the shared-profile role, whose rule runs over the other sequence first, can use the result —
// zsw_capi_shared.hip). seed_window_kernel computes every query column for all
// ~len + 60 rows around the anchor; the alignment itself occupies a band of a few diagonals. Here a lane owns one read pair
// (16-bit halves, as everywhere) and walks it strip by strip: strip k = query columns [kC, (k+1)C) in registers (score_kernel_v2's
// packed column loop) against the reference rows [dt + kC - Wu, dt + (k+1)C + Wd) only, the strip's last column (H, outgoing F,
// true scores) handed to the next strip through a per-lane buffer in global memory (8 bytes per row, written and re-read once).
// No lane talks to another. What lies outside the band is covered by the bounds of zsw_seed.hpp ("banded pass": fresh starts
// above / below the band, exits through a strip's right edge above the next strip's first row, exits through a strip's last
// row); host model against a layered Gotoh DP, one layer per class of paths: tests/models/seed_band.cpp. The launch's band
// (SeedBandArgs::wu0 ...) is either the full one or, for short reads, a narrow first tier whose failures are flagged in place
// (`retry`), selected in anchor order and walked again in the full band (launch_score_seeded, zsw_score_seed.hip); a read whose
// bounds fail in the last tier joins the same worklist as in the window kernel and is scored over all its cells.
// 150 bp: 5 strips x 63 (narrow) or 92 (full) rows x 32 columns per pair in one lane instead of 4 lanes x 211 steps x 38 columns.
#include <algorithm>

#include "zsw_score_seed.hpp"
#include "zsw_score_v2.hpp"
#include "zsw_timer.hpp"

namespace zsw {

namespace {

// a - b per 16-bit half, 0 where b > a (v_pk_sub_u16 with clamp)
__device__ __forceinline__ uint32_t pk_subu_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

constexpr int BC = 32;  // columns per strip (a multiple of 8: a strip's residue codes are whole dwords of the packed reads; MODE 3 keeps
                        // one bit per column in a 32-bit mask)

template <int C, int MINW, int MODE>
__global__ __launch_bounds__(BLOCK, MINW) void seed_band_kernel(SeedBandArgs a) {
    __shared__ uint16_t sel_lut[16];
    __shared__ uint16_t sq[BLOCK * 4 * (SEED_MAX_KMERS + 1)];  // per lane: seed_suffix_q of (read A, B) x (fa, fb), private slices
    const int tid = threadIdx.x;
    if (tid < 16) {
        const uint32_t k = (uint32_t)tid;
        sel_lut[tid] = (uint16_t)(k < 4 ? (2 * k + 1) | ((8 + k) << 8) : (k == 15 ? 0x0c00u : (2 * (k - 3)) | 0x0c00u));
    }
    __syncthreads();
    const int R = (int)a.ref_len;
    const uint32_t n_items = a.n_dev ? min(*a.n_dev, a.n) : a.n;
    const uint32_t n_pairs = (n_items + 1) / 2;
    const uint32_t ge2 = a.ge2, gd2 = a.gd2;
    const uint32_t ge1 = ge2 & 0xffffu;
    const int maxw = a.sp.maxw;
    uint16_t* q = sq + (size_t)tid * 4 * (SEED_MAX_KMERS + 1);
    uint2* bnd = a.bnd + (size_t)blockIdx.x * (size_t)a.nb * BLOCK + tid;  // row j of this lane: bnd[j * BLOCK]

    // Work queue: a wavefront takes the next 64 pairs of the anchor order (one atomic per wavefront), so that no lane waits for
    // a neighbour with one pair more. Every wavefront leaves at the first chunk past the last pair, or at the first chunk that
    // holds nothing but reads without an anchor (sorted: they are the tail of the order, and the seed kernel listed them).
    for (;;) {
        uint32_t base = 0;
        if ((tid & 63) == 0) base = atomicAdd(a.next_pair, 64u);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (base >= n_pairs) break;
        const uint32_t pair = base + (uint32_t)(tid & 63);
        const uint32_t itemA = 2 * pair, itemB = itemA + 1;
        uint32_t ridA = 0, keyA = a.fail_key;
        if (pair < n_pairs) {
            ridA = a.order[itemA];
            keyA = a.keys[ridA];
        }
        const bool active = keyA != a.fail_key;
        if (__ballot(active) == 0) break;
        if (active) {
        uint32_t ridB = itemB < n_items ? a.order[itemB] : ridA;
        uint32_t keyB = itemB < n_items ? a.keys[ridB] : a.fail_key;
        bool validB = keyB != a.fail_key;
        const int dtA = (int)keyA - (int)a.key_bias;
        int dtB = validB ? (int)keyB - (int)a.key_bias : dtA;
        const uint32_t idA = a.b.items ? a.b.items[ridA] : ridA;
        uint32_t idB = validB ? (a.b.items ? a.b.items[ridB] : ridB) : idA;
        if (validB && dtB - dtA > SEED_BAND_SLACK) {  // anchors too far apart to share a band: B waits for the next tier / takes the full pass
            if (a.retry) a.retry[itemB] = 1;
            else a.fail_list[atomicAdd(a.fail_count, 1u)] = idB;
            validB = false;
            dtB = dtA;
        }
        if (!validB) ridB = ridA;
        const uint32_t lenA = a.b.offsets ? (uint32_t)(a.b.offsets[idA + 1] - a.b.offsets[idA]) : a.b.fixed_len;
        const uint32_t lenB = validB ? (a.b.offsets ? (uint32_t)(a.b.offsets[idB + 1] - a.b.offsets[idB]) : a.b.fixed_len) : 0u;
        const int lenmax = (int)max(lenA, lenB);
        const int n_strips = (lenmax + C - 1) / C;
        const int wu = a.wu0 + lenmax * a.wu_per16 / 16, wd = a.wd0 + lenmax * a.wd_per32 / 32;
        const int dtmin = min(dtA, dtB), dtmax = max(dtA, dtB);
        // what the seed kernel left: potentials, k-mer bounds, masks; the suffix bounds of the two masks per read
        const uint32_t infoA = a.info[ridA], infoB = a.info[ridB];
        const uint32_t mkA = a.band_masks[ridA], mkB = a.band_masks[ridB];
        const int tallA = (int)(infoA & 0xffffu), tallB = (int)(infoB & 0xffffu);
        const int dfaA = (int)((infoA >> 16) & 0xffu), dfaB = (int)((infoB >> 16) & 0xffu);
        const int dfbA = (int)a.band_dfb[ridA], dfbB = (int)a.band_dfb[ridB];
        int mA, strA, c0A, mB, strB, c0B;
        seed_layout((int)lenA, a.sp.K, a.sp.spacer, &mA, &strA, &c0A);
        seed_layout((int)lenB, a.sp.K, a.sp.spacer, &mB, &strB, &c0B);
        {
            int tmp[SEED_MAX_KMERS + 1];
            seed_suffix_q(mA, a.sp.K, c0A, strA, (int)lenA, maxw, mkA & 0xffffu, seed_lambda(a.sp, strA), tmp);
            for (int i = 0; i <= SEED_MAX_KMERS; ++i) q[i] = (uint16_t)tmp[i <= mA ? i : mA];
            seed_suffix_q(mA, a.sp.K, c0A, strA, (int)lenA, maxw, mkA >> 16, seed_lambda(a.sp, strA), tmp);
            for (int i = 0; i <= SEED_MAX_KMERS; ++i) q[(SEED_MAX_KMERS + 1) + i] = (uint16_t)tmp[i <= mA ? i : mA];
            seed_suffix_q(mB, a.sp.K, c0B, strB, (int)lenB, maxw, mkB & 0xffffu, seed_lambda(a.sp, strB), tmp);
            for (int i = 0; i <= SEED_MAX_KMERS; ++i) q[2 * (SEED_MAX_KMERS + 1) + i] = (uint16_t)tmp[i <= mB ? i : mB];
            seed_suffix_q(mB, a.sp.K, c0B, strB, (int)lenB, maxw, mkB >> 16, seed_lambda(a.sp, strB), tmp);
            for (int i = 0; i <= SEED_MAX_KMERS; ++i) q[3 * (SEED_MAX_KMERS + 1) + i] = (uint16_t)tmp[i <= mB ? i : mB];
        }
        const int gup = seed_gap_up(a.sp, wu);
        const int gdnA = seed_gap_down(a.sp, wd, (int)lenA, tallA), gdnB = seed_gap_down(a.sp, wd, (int)lenB, tallB);

        const uint32_t* codeA = a.codes + (size_t)ridA * a.cs;
        const uint32_t* codeB = a.codes + (size_t)ridB * a.cs;
        uint32_t best = 0;          // true scores
        // ends (MODE 1, 2): first row holding the maximum, then the first column of that row (striped.rs:296-321). A strip walks its
        // rows in order; strips overlap in rows, so each strip keeps its own (maximum, first row, H row of that row) and the strips
        // merge by (higher maximum, then earlier row; the same row in two strips: the earlier strip holds the earlier column).
        int bestA = 0, bestB = 0, rowA = 0x7fffffff, rowB = 0x7fffffff, colA = 0x7fffffff, colB = 0x7fffffff;
        bool multA = false, multB = false;  // MODE 3: the maximum so far sits in more than one (real) cell
        int bndA = -1, bndB = -1;   // the exit bounds so far
        int prev_bot = 0;
#pragma unroll 1
        for (int k = 0; k < n_strips; ++k) {
            // selectors of the strip's columns from the packed 4-bit residue codes (15 = padding)
            uint32_t sel[C];
#pragma unroll
            for (int d = 0; d < C / 8; ++d) {
                const uint32_t di = (uint32_t)(k * (C / 8) + d);
                const uint32_t wa = di < a.cs ? codeA[di] : 0xffffffffu;
                const uint32_t wb = (validB && di < a.cs) ? codeB[di] : 0xffffffffu;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    sel[8 * d + j] = (uint32_t)sel_lut[(wa >> (4 * j)) & 15u] | ((uint32_t)sel_lut[(wb >> (4 * j)) & 15u] << 16);
            }
            const int top = max(0, min(R, dtmin + k * C - wu)), bot = max(0, min(R, dtmax + (k + 1) * C + wd));
            const bool has_next = k + 1 < n_strips;
            const int next_top = max(0, min(R, dtmin + (k + 1) * C - wu));  // first row of the next strip
            uint32_t Dr = (a.floor0 - ge1) * 0x00010001u;                   // D of the row before the strip's first row
            uint32_t H[C], E[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                H[c] = Dr;
                E[c] = pk_addu(Dr, ge2);
            }
            // the previous strip's last column: rows [top - 1, prev_bot) wait in bnd[row - (top - 1)] as true scores
            const bool have_left = k > 0;
            uint32_t Hin_prev = Dr;
            if (have_left && top >= 1 && top - 1 < prev_bot) Hin_prev = pk_addu(Dr, bnd[0].x);
            uint2 w = a.gtab[SEED_GTAB_PAD + top];
            uint2 bl = make_uint2(0u, 0u);
            if (have_left && top < prev_bot) bl = bnd[(size_t)1 * BLOCK];
            // exits through the right edge in the rows above the next strip: largest H / outgoing F of the last column, plain and
            // less gap_extend per diagonal beyond the first outside the band's edge (seed_band_upper)
            uint32_t uk = 0, ug = 0;
            uint32_t sbest = 0, snapD = 0;  // this strip's maximum (true scores) and, MODE 2, the H row and drift of each read's latest rise
            int srA = 0x7fffffff, srB = 0x7fffffff;
            uint32_t snap[MODE >= 2 ? C : 1];
            if (MODE >= 2) {
#pragma unroll
                for (int c = 0; c < C; ++c) snap[MODE >= 2 ? c : 0] = 0;
            }
            // MODE 3: the strip's real columns per read as bit masks (a padding column copies the value of its upper left
            // neighbour: a copy of the maximum is not a second cell holding it), and whether a row after the strip's latest rise
            // reached the same value again in a real column
            const int nrA = max(0, min(C, (int)lenA - k * C)), nrB = max(0, min(C, (int)lenB - k * C));
            const uint32_t realA = nrA >= 32 ? 0xffffffffu : ((1u << nrA) - 1u), realB = nrB >= 32 ? 0xffffffffu : ((1u << nrB) - 1u);
            bool smA = false, smB = false;
            const int e_top = (dtmin + (k + 1) * C - wu) - 1 - top;  // the first row's distance from the last row above the next strip
            uint32_t dec = min(ge1 * (uint32_t)max(e_top - 1, 0), 0xffffu) * 0x00010001u;
#pragma unroll 1
            for (int r = top; r < bot; ++r) {
                const uint2 wn = a.gtab[SEED_GTAB_PAD + r + 1];
                const bool left = have_left && r < prev_bot;
                uint2 bln = make_uint2(0u, 0u);
                if (have_left && r + 1 < prev_bot) bln = bnd[(size_t)(r + 1 - top + 1) * BLOCK];
                Dr = pk_addu(Dr, ge2);
                const uint32_t Dn = pk_addu(Dr, ge2);
                const uint32_t Hin = left ? pk_addu(Dr, bl.x) : Dr;
                uint32_t F = left ? pk_addu(Dr, bl.y) : Dr;
                uint32_t hd = pk_addu(Hin_prev, __builtin_amdgcn_perm(w.y, w.x, sel[0]));
                Hin_prev = Hin;
                uint32_t rmax = 0x04000400u;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    uint32_t hd_next = 0;
                    if (c + 1 < C) hd_next = pk_addu(H[c], __builtin_amdgcn_perm(w.y, w.x, sel[c + 1 < C ? c + 1 : c]));
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
                if (has_next) {
                    if (r >= next_top - 1) bnd[(size_t)(r - (next_top - 1)) * BLOCK] = make_uint2(tH, tF);
                    if (r < next_top) {
                        const uint32_t v = pk_maxu(tH, tF);
                        uk = pk_maxu(uk, v);
                        ug = pk_maxu(ug, pk_subu_sat(v, dec));
                    }
                }
                dec = pk_subu_sat(dec, ge2);
                const uint32_t tmax = pk_subu(rmax, Dr);
                if (MODE != 0) {
                    const uint32_t nsb = pk_maxu(sbest, tmax);
                    const uint32_t ch = nsb ^ sbest;
                    if (ch & 0xffffu) srA = r;
                    if (ch >> 16) srB = r;
                    if (MODE >= 2) {
                        const uint32_t m = ((ch & 0xffffu) ? 0xffffu : 0u) | ((ch >> 16) ? 0xffff0000u : 0u);
#pragma unroll
                        for (int c = 0; c < C; ++c) snap[MODE >= 2 ? c : 0] = (H[c] & m) | (snap[MODE >= 2 ? c : 0] & ~m);
                        snapD = (Dr & m) | (snapD & ~m);
                    }
                    if (MODE == 3) {
                        const uint32_t eq = tmax ^ nsb;  // a half is 0 where this row's maximum is the strip's
                        const bool tA = !(ch & 0xffffu) && !(eq & 0xffffu) && (nsb & 0xffffu) != 0;
                        const bool tB = !(ch >> 16) && !(eq >> 16) && (nsb >> 16) != 0;
                        if (ch & 0xffffu) smA = false;
                        if (ch >> 16) smB = false;
                        if (tA || tB) {  // rare: which columns hold it
                            const uint32_t target = pk_addu(nsb, Dr);
                            uint32_t hitA = 0, hitB = 0;
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                const uint32_t x = H[c] ^ target;
                                if (!(x & 0xffffu)) hitA |= 1u << c;
                                if (!(x >> 16)) hitB |= 1u << c;
                            }
                            if (tA && (hitA & realA)) smA = true;
                            if (tB && (hitB & realB)) smB = true;
                        }
                    }
                    sbest = nsb;
                }
                best = pk_maxu(best, tmax);
                w = wn;
                bl = bln;
            }
            // exits through the right edge above the next strip (they continue left of the band: fa mask, deletions to come back)
            if (has_next && next_top > top) {
                const int x = (k + 1) * C - 1;
                bndA = max(bndA, seed_band_upper(a.sp, (int)(uk & 0xffffu), (int)(ug & 0xffffu), x, (int)lenA, wu, mA, c0A, strA, q));
                bndB = max(bndB, seed_band_upper(a.sp, (int)(uk >> 16), (int)(ug >> 16), x, (int)lenB, wu, mB, c0B, strB, q + 2 * (SEED_MAX_KMERS + 1)));
            }
            // exits through the strip's last row (they continue below the band: fb mask, insertions to come back). Two packed
            // maxima over the strip's columns serve both reads (seed_band_lower): the exit value plus the columns up to the strip's
            // last at full potential, and the exit value less what the longer way back costs.
            if (bot < R && bot > top) {
                uint32_t mk = 0, mg = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const uint32_t he = pk_subu(pk_maxu(H[c], pk_subu(E[c], ge2)), Dr);
                    const int e = C - 1 - c;
                    mk = pk_maxu(mk, pk_addu(he, (uint32_t)(maxw * e) * 0x00010001u));
                    mg = pk_maxu(mg, pk_addu(pk_subu_sat(he, (uint32_t)(ge1 * (uint32_t)(e > 1 ? e - 1 : 0)) * 0x00010001u), (uint32_t)(e >= 1 ? maxw : 0) * 0x00010001u));
                }
                const int xl = (k + 1) * C - 1;
                if (k * C < (int)lenA)
                    bndA = max(bndA, seed_band_lower(a.sp, (int)(mk & 0xffffu), (int)(mg & 0xffffu), xl, (int)lenA, wd, tallA, mA, c0A, strA, q + (SEED_MAX_KMERS + 1)));
                if (k * C < (int)lenB)
                    bndB = max(bndB, seed_band_lower(a.sp, (int)(mk >> 16), (int)(mg >> 16), xl, (int)lenB, wd, tallB, mB, c0B, strB, q + 3 * (SEED_MAX_KMERS + 1)));
            }
            if (MODE != 0) {  // merge the strip's maximum into the read's
                const int sA = (int)(sbest & 0xffffu), sB = (int)(sbest >> 16);
                const bool eqA = sA == bestA && sA > 0, eqB = sB == bestB && sB > 0;  // MODE 3: another strip held this value already
                const bool upA = sA > bestA || (eqA && srA < rowA);
                const bool upB = sB > bestB || (eqB && srB < rowB);
                if (upA) {
                    bestA = sA;
                    rowA = srA;
                }
                if (upB) {
                    bestB = sB;
                    rowB = srB;
                }
                if (MODE >= 2 && (upA || upB || (MODE == 3 && (eqA || eqB)))) {
                    int cA = 0x7fffffff, cB = 0x7fffffff;
                    uint32_t hitA = 0, hitB = 0;
                    const int sdA = (int)(snapD & 0xffffu), sdB = (int)(snapD >> 16);
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        const uint32_t sv = snap[MODE >= 2 ? c : 0];
                        if ((int)(sv & 0xffffu) - sdA == sA) {
                            cA = k * C + c;
                            hitA |= 1u << c;
                        }
                        if ((int)(sv >> 16) - sdB == sB) {
                            cB = k * C + c;
                            hitB |= 1u << c;
                        }
                    }
                    if (upA) colA = cA;
                    if (upB) colB = cB;
                    if (MODE == 3) {
                        // cells of this strip holding its maximum: the real columns of the row of its latest rise, plus later rows
                        const int nA = __popc(hitA & realA) + (smA ? 1 : 0), nB = __popc(hitB & realB) + (smB ? 1 : 0);
                        if (sA > 0 && (upA && !eqA)) multA = nA > 1;   // a higher maximum: this strip's cells are all there are so far
                        else if (eqA && nA > 0) multA = true;          // the same value in two strips: two cells
                        if (sB > 0 && (upB && !eqB)) multB = nB > 1;
                        else if (eqB && nB > 0) multB = true;
                    }
                }
            }
            prev_bot = bot;
        }
        // fresh starts outside the band
        const int SA = (int)(best & 0xffffu), SB = (int)(best >> 16);
        if (dtmin + (n_strips - 1) * C - wu > 0) {
            bndA = max(bndA, tallA - min(dfaA, gup));
            bndB = max(bndB, tallB - min(dfaB, gup));
        }
        if (dtmax + C + wd < R) {
            bndA = max(bndA, tallA - min(dfbA, gdnA));
            bndB = max(bndB, tallB - min(dfbB, gdnB));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !validB) break;
            const uint32_t id = h ? idB : idA;
            const int S = h ? SB : SA, bound = h ? bndB : bndA;
            // score only: a path outside the band matters if it can score MORE than S; with ends also if it can score S (it could
            // end in an earlier row or column)
            const bool redo = MODE == 0 ? bound > S : bound >= S;
            if ((h ? lenB : lenA) == 0 || redo) {
                if (a.retry) a.retry[h ? itemB : itemA] = 1;
                else a.fail_list[atomicAdd(a.fail_count, 1u)] = id;
            } else {
                uint32_t score;
                uint8_t status, tier;
                apply_rule(a.rule, (uint64_t)S, &score, &status, &tier);
                a.out.score[id] = score;
                a.out.status[id] = status;
                if (a.out.tier) a.out.tier[id] = tier;
                const bool some = status == ZSW_STATUS_SOME;
                if (MODE != 0 && a.out.ref_end) a.out.ref_end[id] = some ? (uint32_t)(h ? rowB : rowA) + 1 : 0;
                if (MODE >= 2 && a.out.query_end) a.out.query_end[id] = some ? (uint32_t)(h ? colB : colA) + 1 : 0;
                if (MODE == 3 && a.out.unique) a.out.unique[id] = (some && !(h ? multB : multA)) ? 1 : 0;
                if (MODE != 0 && a.out.safe_row)  // sw_simd_align's second pass may start this late (or 0xffffffff: no certificate)
                    a.out.safe_row[id] = (uint32_t)seed_safe_start(a.sp, h ? tallB : tallA, h ? dfaB : dfaA, h ? dtB : dtA, S);
            }
        }
        }  // active
    }
}

}  // namespace

uint32_t seed_band_rows(const SeedParams& p, uint32_t max_len) { return (uint32_t)seed_rows_above(p, (int)max_len) + (uint32_t)seed_rows_below(p, (int)max_len) + SEED_BAND_SLACK + 4; }

uint32_t seed_band_grid(uint32_t n, uint32_t grid_cap) {
    const uint32_t pairs = (n + 1) / 2;
    return std::max<uint32_t>(1, std::min<uint32_t>((pairs + BLOCK - 1) / BLOCK, std::min(grid_cap, SEED_BAND_MAX_GRID)));
}

size_t seed_band_buffer_bytes(const SeedParams& p, uint32_t n, uint32_t max_len, uint32_t grid_cap) {
    return n ? (size_t)seed_band_grid(n, grid_cap) * BLOCK * (size_t)seed_band_rows(p, max_len) * sizeof(uint2) : 0;
}

bool seed_band_applicable(const SeedParams& p, uint32_t max_len, uint32_t rebase_rows) {
    // a strip's rows must fit one drift period of the packed domain (no re-basing inside a strip)
    return (uint32_t)BC + seed_band_rows(p, max_len) + 2 <= rebase_rows;
}

hipError_t launch_seed_band(const SeedBandArgs& a, int mode, hipStream_t stream) {
    // MODE 2 keeps a snapshot of the H row (32 more registers): two waves per SIMD
    if (mode == 0) hipLaunchKernelGGL((seed_band_kernel<BC, 3, 0>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((seed_band_kernel<BC, 3, 1>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    else if (mode == 2) hipLaunchKernelGGL((seed_band_kernel<BC, 2, 2>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((seed_band_kernel<BC, 2, 3>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

}  // namespace zsw
