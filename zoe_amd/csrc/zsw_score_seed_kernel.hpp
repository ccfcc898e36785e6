// zsw_score_seed_kernel.hpp — the window kernel of the seeded exact score pass (see zsw_seed.hpp for the argument, and
// zsw_score_seed.hip for the seed kernel and the launcher). score_kernel_v2's column loop (zsw_score_v2.hpp) over ALL query
// columns of a read pair but only the reference rows [a0, a0 + Tw) around the pair's anchor diagonals, from a zero state; then
// the three bounds decide whether the window's maximum is the read's score.
#pragma once
#include "zsw_score_seed.hpp"
#include "zsw_score_v2.hpp"

namespace zsw {

// the strip configurations of zsw_score_v2.hpp without (32, 5), which exists for small batches of the full pass only
#define ZSW_FOR_EACH_SEED_CONFIG(X)                                                                                             \
    X(4, 19) X(4, 22) X(4, 25) X(4, 28) X(4, 32) X(4, 35) X(4, 38) X(8, 19) X(8, 22) X(8, 25) X(8, 28) X(8, 32) X(8, 35) X(8, 38) \
    X(16, 22) X(16, 25) X(16, 32) X(16, 38) X(64, 19) X(64, 38)

// Waves per SIMD the register allocator must leave room for. Besides score_kernel_v2's state (H, E, selectors [+ snapshot row])
// this kernel keeps the window geometry and both reads' identities across the loop and needs H and E again for the exit bound:
// ~64 registers on top, so the widest strips run at two waves per SIMD (which still fills the 4-cycle issue slots: r03 PMC,
// seed_window_kernel<4,38,2> at 3.99 cycles per instruction with two) instead of spilling at three.
constexpr int seed_min_waves(int C, int MODE) {
    const int need = MODE == 2 ? 4 * C + 96 : 3 * C + 54;
    return need <= 80 ? 6 : need <= 96 ? 5 : need <= 128 ? 4 : need <= 168 ? 3 : 2;
}

template <int G, int C, int MODE>
__global__ __launch_bounds__(BLOCK, seed_min_waves(C, MODE)) void seed_window_kernel(SeedWindowArgs a) {
    // one raw buffer: the block's row table during the row loop, afterwards each lane's column of max(H, E) values for the exit
    // bound (C dwords per lane: a rolled loop over LDS instead of ~70 more live registers in an unrolled one)
    constexpr size_t RP_BYTES = (size_t)(CH + 2 * G) * sizeof(uint2), HE_BYTES = (size_t)C * BLOCK * sizeof(uint32_t);
    __shared__ __attribute__((aligned(16))) uint8_t sraw[RP_BYTES > HE_BYTES ? RP_BYTES : HE_BYTES];
    uint2* rp = reinterpret_cast<uint2*>(sraw);
    uint32_t* she = reinterpret_cast<uint32_t*>(sraw);
    __shared__ int s_lo, s_hi;
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    __shared__ int sq[(BLOCK / G) * 2 * (SEED_MAX_KMERS + 1)];  // exit bound: seed_suffix_q of the group's two reads
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);
    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;
    const uint32_t itemA = 2 * group, itemB = 2 * group + 1;
    // the order holds the reads without an anchor last (largest key): a block that starts with one has nothing to do
    {
        const uint32_t first = 2 * blockIdx.x * (BLOCK / G);
        if (first >= a.n || a.keys[a.order[first]] == a.fail_key) return;
    }
    uint32_t ridA = itemA < a.n ? a.order[itemA] : 0, ridB = itemB < a.n ? a.order[itemB] : 0;
    const uint32_t keyA = itemA < a.n ? a.keys[ridA] : a.fail_key, keyB = itemB < a.n ? a.keys[ridB] : a.fail_key;
    const bool validA = keyA != a.fail_key, validB = keyB != a.fail_key;  // sorted: a valid B implies a valid A
    if (!validB) ridB = ridA;  // an absent B mirrors A; its columns are padding
    const int R = (int)a.ref_len;
    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    if (tid < 9) swt[tid] = make_uint2(a.wtab[tid][0], a.wtab[tid][1]);
    if (tid == 0) {
        s_lo = 0x7fffffff;
        s_hi = 0;
    }
    __syncthreads();

    uint64_t offA = 0, offB = 0;
    uint32_t lenA = 0, lenB = 0;
    const uint32_t idA = validA ? (a.b.items ? a.b.items[a.first + ridA] : a.first + ridA) : 0;
    const uint32_t idB = validB ? (a.b.items ? a.b.items[a.first + ridB] : a.first + ridB) : 0;
    if (validA) {
        offA = a.b.offsets ? a.b.offsets[idA] : (uint64_t)idA * a.b.fixed_len;
        lenA = a.b.offsets ? (uint32_t)(a.b.offsets[idA + 1] - offA) : a.b.fixed_len;
    }
    if (validB) {
        offB = a.b.offsets ? a.b.offsets[idB] : (uint64_t)idB * a.b.fixed_len;
        lenB = a.b.offsets ? (uint32_t)(a.b.offsets[idB + 1] - offB) : a.b.fixed_len;
    }
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t q = (uint32_t)(g * C + c);
        uint32_t kA = PAD_K, kB = PAD_K;
        if (q < lenA) kA = lut[a.b.bases[offA + (a.reversed ? lenA - 1 - q : q)]];
        if (q < lenB) kB = lut[a.b.bases[offB + (a.reversed ? lenB - 1 - q : q)]];
        const uint32_t sA = kA < 4 ? (2 * kA + 1) | ((8 + kA) << 8) : (kA == PAD_K ? 0x0c00u : (2 * (kA - 3)) | 0x0c00u);
        const uint32_t sB = kB < 4 ? (2 * kB + 1) | ((8 + kB) << 8) : (kB == PAD_K ? 0x0c00u : (2 * (kB - 3)) | 0x0c00u);
        sel[c] = sA | (sB << 16);
    }

    // the window: rows M1 above the first row of the earlier anchor diagonal to M2 below the last row of the later one
    const int dtA = (int)keyA - (int)a.key_bias, dtB = validB ? (int)keyB - (int)a.key_bias : dtA;
    const int a0 = validA ? max(0, min(dtA - seed_rows_above(a.sp, (int)lenA), dtB - seed_rows_above(a.sp, (int)(validB ? lenB : lenA)))) : 0;
    const int b0 = validA ? min(R, max(dtA + (int)lenA, dtB + (int)(validB ? lenB : lenA)) + a.sp.M2) : 0;
    // rows of this lane group alone: the groups of a wavefront are independent (shuffles stay inside a group), so a group whose
    // window is shorter simply leaves the row loop earlier; a0 + Tw <= R keeps every row access inside the tables' padding
    const int Tw = validA ? b0 - a0 : 0;
    // the block's row table: rows [lo - (G - 1), hi + G) of the reference (real rows for the lanes' skewed last rows too)
    if (g == 0 && validA) {
        atomicMin(&s_lo, a0);
        atomicMax(&s_hi, a0 + Tw);
    }
    __syncthreads();
    const int blo = s_lo, bhi = s_hi;
    const bool staged = blo <= bhi && bhi - blo <= CH;  // reads sorted by anchor: the windows of a block overlap
    if (staged) {
        for (int j = tid; j < bhi - blo + 2 * G; j += BLOCK) {
            const int row = blo - (G - 1) + j;
            rp[j] = swt[(row >= 0 && row < R) ? (int)lut[a.ref[row]] : NEUTRAL];
        }
    }
    __syncthreads();
    const int joff = (validA && staged) ? a0 - blo + (G - 1) - g : 0;  // table entry of this lane's row at step t: joff + t

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
    const int T = validA ? Tw + G - 1 : 0;
    // the rows come from the block's LDS table, or — when the windows of the block's reads lie too far apart for one table (few
    // reads against a long reference) — from the per-row table in global memory (L2), one step ahead either way
    // (one loop over a generic pointer: two instantiations of the loop behind a lambda put the H / E / selector arrays into
    // scratch memory)
    const uint2* rows = staged ? static_cast<const uint2*>(rp) + joff : a.gtab + (SEED_GTAB_PAD + a0 - g);  // gtab[SEED_GTAB_PAD + r]: row r
    uint2 w = rows[0];
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const uint2 wn = rows[t + 1];
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
            Fin = Dr;
            Hin = Dr;
        }
        uint32_t hd = pk_addu(Hin_prev, __builtin_amdgcn_perm(w.y, w.x, sel[0]));
        Hin_prev = Hin;
        uint32_t F = Fin;
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

    // ---- the window's maximum per read ----
    const int lbA = (int)(best & 0xffffu), lbB = (int)(best >> 16);
    int SA = lbA, SB = lbB;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        SA = max(SA, __shfl_xor(SA, d, G));
        SB = max(SB, __shfl_xor(SB, d, G));
    }
    // ends (MODE 1, 2): first row holding the window's maximum, then the first column of that row (striped.rs:296-321)
    uint32_t reA = 0, reB = 0, qeA = 0, qeB = 0;
    if (MODE != 0) {
        int kA = (lbA == SA) ? rA : 0x7fffffff, kB = (lbB == SB) ? rB : 0x7fffffff;
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
                if ((int)(sv & 0xffffu) - sdA == SA) cA = g * C + c;
                if ((int)(sv >> 16) - sdB == SB) cB = g * C + c;
            }
            if (!(lbA == SA && rA == kA)) cA = 0x7fffffff;
            if (!(lbB == SB && rB == kB)) cB = 0x7fffffff;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) {
                cA = min(cA, __shfl_xor(cA, d, G));
                cB = min(cB, __shfl_xor(cB, d, G));
            }
            qeA = (uint32_t)cA + 1;
            qeB = (uint32_t)cB + 1;
        }
    }
    // ---- exit bound: what can still leave this lane's last rows (E holds the next row's E; Fout moves right into the next
    // lane's columns; Hin_prev is the left neighbour's H one row up, the diagonal into the first cell below this lane's rows),
    // plus what the columns to the right can add in the rows below the window (seed_exit_bound) ----
    const int maxw = a.sp.maxw;
    int mA, strA, c0A, mB, strB, c0B;
    seed_layout((int)lenA, a.sp.K, a.sp.spacer, &mA, &strA, &c0A);
    seed_layout((int)lenB, a.sp.K, a.sp.spacer, &mB, &strB, &c0B);
    int* qA = sq + (tid / G) * 2 * (SEED_MAX_KMERS + 1);
    int* qB = qA + (SEED_MAX_KMERS + 1);
    if (g == 0) seed_suffix_q(mA, a.sp.K, c0A, strA, (int)lenA, maxw, validA ? a.masks[ridA] : 0u, seed_lambda(a.sp, strA), qA);
    if (g == (G > 1 ? 1 : 0)) seed_suffix_q(mB, a.sp.K, c0B, strB, (int)lenB, maxw, validB ? a.masks[ridB] : 0u, seed_lambda(a.sp, strB), qB);
    __syncthreads();  // also: every wavefront of the block is done with the row table
    int v3A = -1, v3B = -1;
    if (a0 + Tw < R) {
        const int dA = (int)(Dr & 0xffffu), dB = (int)(Dr >> 16);
#pragma unroll
        for (int c = 0; c < C; ++c) she[c * BLOCK + tid] = pk_maxu(H[c], pk_subu(E[c], ge2));
        // the first sampled k-mer completely right of column x, as (index, its first column or len); x only grows below
        auto first_right = [](int x, int m, int c0, int stride, int len, int* i, int* ci) {
            *i = 0;
            if (m > 0 && x >= c0) *i = min(m, (x - c0) / stride + 1);
            *ci = *i < m ? c0 + *i * stride : len;
        };
        // x >= len - 1: no query column is left for the path to use below the window (a path that only trails a gap down there
        // scores less than its last computed cell, which the window's maximum covers) — and padding columns carry copies of H
        auto bound = [&](int v, int x, int len, int i, int ci, const int* q) { return x < len - 1 ? v + maxw * (ci - 1 - x) + q[i] : -1; };
        int iA, ciA, iB, ciB;
        if (g > 0) {  // the diagonal from the left neighbour's last column, one row up, into this lane's first column
            first_right(g * C - 1, mA, c0A, strA, (int)lenA, &iA, &ciA);
            first_right(g * C - 1, mB, c0B, strB, (int)lenB, &iB, &ciB);
            v3A = max(v3A, bound((int)(Hin_prev & 0xffffu) - dA, g * C - 1, (int)lenA, iA, ciA, qA));
            v3B = max(v3B, bound((int)(Hin_prev >> 16) - dB, g * C - 1, (int)lenB, iB, ciB, qB));
        }
        first_right(g * C, mA, c0A, strA, (int)lenA, &iA, &ciA);
        first_right(g * C, mB, c0B, strB, (int)lenB, &iB, &ciB);
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const int col = g * C + c;
            if (iA < mA && ciA <= col) {
                ++iA;
                ciA = iA < mA ? ciA + strA : (int)lenA;
            }
            if (iB < mB && ciB <= col) {
                ++iB;
                ciB = iB < mB ? ciB + strB : (int)lenB;
            }
            const uint32_t he = she[c * BLOCK + tid];
            v3A = max(v3A, bound((int)(he & 0xffffu) - dA, col, (int)lenA, iA, ciA, qA));
            v3B = max(v3B, bound((int)(he >> 16) - dB, col, (int)lenB, iB, ciB, qB));
        }
        const int colr = (g + 1) * C;  // the gap that runs on into the next lane's first column has consumed that column
        first_right(colr, mA, c0A, strA, (int)lenA, &iA, &ciA);
        first_right(colr, mB, c0B, strB, (int)lenB, &iB, &ciB);
        v3A = max(v3A, bound((int)(Fout & 0xffffu) - dA, colr, (int)lenA, iA, ciA, qA));
        v3B = max(v3B, bound((int)(Fout >> 16) - dB, colr, (int)lenB, iB, ciB, qB));
    }
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        v3A = max(v3A, __shfl_xor(v3A, d, G));
        v3B = max(v3B, __shfl_xor(v3B, d, G));
    }
    if (g < 2) {
        const bool second = g == 1;
        if (second ? validB : validA) {
            const uint32_t id = second ? idB : idA;
            const uint32_t len = second ? lenB : lenA;
            const uint32_t info = a.info[second ? ridB : ridA];
            const int S = second ? SB : SA;
            const SeedBounds sb = seed_bounds(a.sp, (int)(info & 0xffffu), (int)((info >> 16) & 0xffu), (int)(info >> 24), second ? dtB : dtA, a0,
                                              a0 + Tw, R);
            const int bound = max(max(sb.above, sb.below), second ? v3B : v3A);
            // score only: a path outside the computed cells matters if it can score MORE than S; with ends also if it can score S
            // (it could end in an earlier row or column)
            const bool redo = MODE == 0 ? bound > S : bound >= S;
            if (len == 0 || redo) {  // all cells for this read (an empty read gets its status there)
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
                if (MODE != 0 && a.out.safe_row)  // sw_simd_align's second pass may start this late (or 0xffffffff: no certificate)
                    a.out.safe_row[id] = (uint32_t)seed_safe_start(a.sp, (int)(info & 0xffffu), (int)((info >> 16) & 0xffu), second ? dtB : dtA, S);
            }
        }
    }
}

}  // namespace zsw
