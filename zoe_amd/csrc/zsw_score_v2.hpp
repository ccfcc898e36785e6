// zsw_score_v2.hpp — the drift-domain packed score kernel (score_kernel_v2), shared by zsw_score.hip (alphabets of up to
// 7 letters: the row's scores come from one v_perm_b32) and zsw_score_wide.hip (up to 32 letters: two LDS byte loads).
#pragma once
#include "zsw_internal.hpp"

namespace zsw {

constexpr int BLOCK = 256;
constexpr int CH = 2048;       // reference rows staged in LDS per chunk
constexpr int NEUTRAL = 8;     // table row used outside [0, R): scores 0 for every residue
constexpr int PAD_K = 255;     // residue code of a padded query column

// The strip configurations (G lanes per read pair x C columns per lane). Every launcher switch and the configuration table are
// generated from this one list: twenty of ascending capacity G*C (the first-fit choice and the length classes of ragged batches
// walk them in order; steps of 8-16 % between 76 and 400 columns keep the padded columns of a 75-400 bp batch near 6 %), then
// (32, 5), which only the batch-size-aware choice picks (short reads, too few of them to fill the chip with four lanes per pair).
// Measured and rejected: C = 41 / 44 with four and eight lanes instead of (8,19), (8,22), (16,22) — 8.5 instead of 9.0 TCUPS on
// the 75-400 bp batch (the longer column loops sit at the 168-VGPR boundary of three waves per SIMD); thirty configurations
// (capacity steps of 4-8 %, padded cells 1.035x): 8.4-8.5 TCUPS — ten more launches of smaller grids cost more than the padding
// they save; eight side streams instead of four: no change.
#define ZSW_FOR_EACH_STRIP_CONFIG(X)                                                                                            \
    X(4, 19) X(4, 22) X(4, 25) X(4, 28) X(4, 32) X(4, 35) X(4, 38) X(8, 19) X(8, 22) X(8, 25) X(8, 28) X(8, 32) X(8, 35) X(8, 38) \
    X(16, 22) X(16, 25) X(16, 32) X(16, 38) X(64, 19) X(64, 38) X(32, 5)
constexpr int WIDE_STRIDE = 36;  // bytes per table row of the WIDE kernels (9 dwords: rows start in different LDS banks)
constexpr int WIDE_PAD = 32, WIDE_NEUTRAL = 32;

// Waves per SIMD the register allocator must leave room for: H, E and the selectors take 3*C VGPRs
// (4*C with the MODE 2 snapshot row).
constexpr int min_waves(int C, int MODE) {
    // state registers (H, E, selectors [+ snapshot row]) plus ~40 temporaries; measured on the 150 bp configuration:
    // 3 vs 4 waves/SIMD is time-neutral for this VALU-issue-bound loop, register spills are not free
    const int need = (MODE == 2 ? 4 * C : 3 * C) + (MODE == 2 ? 44 : 36);
    return need <= 80 ? 6 : need <= 96 ? 5 : need <= 128 ? 4 : need <= 168 ? 3 : 2;
}

// ================================================================================================
// score_kernel_v2 — the same recurrence in 7.5 packed instructions per two cells instead of 10.
//
// (1) Row drift. Every quantity of row r is stored with offset D_r = FLOOR + r*ge (per packed half, u16):
//     E_{r+1} = max(E_r - ge, H_r - go, 0)  becomes  E~_{r+1} = max3(E~_r, H~_r - (go - ge), D_{r+1}) — the per-column decay
//     subtraction of E disappears; G~ = F~ + ge turns the F update into max3(F~, H~ - (go - ge), D_{r+1}) with the SAME
//     two operands, followed by F~ = G~ - ge. The +ge of H~_{r-1} -> row r is folded into the score table (t = s + ge).
// (2) Exact 3-input max. All stored values are kept inside [0x0400, 0x7BFF], where IEEE binary16 bit patterns are positive
//     normal numbers ordered exactly like the integers, so v_pk_maximum3_f16 is an exact packed integer max3 (no NaN, no
//     denormal, no -0 can occur). The zero floor of local alignment is the third operand D (true 0 of the row).
// (3) The table row holds signed 8-bit entries for A,C,G,T in bytes 1,3,5,7 (v_perm selectors 8..11 sign-extend exactly
//     those bytes) and non-negative entries (padding, N, ...) in the even bytes.
// The state is re-based every K rows so D stays small; a read whose true score nears the representable limit goes to the
// exact 32-bit kernel like in v1. Results are bit-identical to v1 (tests run both).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    // nested as max(max(a, b), c): with (E, hg, D) and (F, hg, D) the inner pairs differ, so the compiler cannot
    // share a max(hg, D) between them (that would cost a third instruction per column)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b)),
                                                                       __builtin_bit_cast(h2, c)));
}
__device__ __forceinline__ uint32_t pk_addu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ uint32_t pk_subu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ uint32_t pk_maxu(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

struct ScoreArgsV2 {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    uint32_t wtab[9][2];  // per reference residue: the 8 table bytes (see (3)); row 8 = neutral (all entries = ge)
    // WIDE kernels (alphabets of 8..32 letters): wide[r*WIDE_STRIDE + q] = (int8) score(ref residue r, query residue q) + ge,
    // column WIDE_PAD = padding, row 32 = neutral (both = ge, a true 0)
    int8_t wide[33 * 36];
    uint32_t ge2, gd2;    // packed ge and (go - ge)
    uint32_t floor0;      // FLOOR (one half)
    uint32_t K;           // re-base period in rows (power of two)
    uint32_t limit;       // true scores >= limit are recomputed by the exact kernel
    ResultRule rule;
    ScoreOut out;
    // TILED launches (reads longer than the widest strip configuration): this launch covers query columns
    // [tile_q0, tile_q0 + G*C); the strip boundary of every reference row travels between launches through HBM.
    uint32_t tile_q0;
    const uint2* tile_in;  // [pair][R] (last H, outgoing F) left by the previous tile's last lane, drift domain; null: first tile
    uint2* tile_out;       // the same, written by this tile's last lane; null: last tile
    uint4* tile_state;     // per read: (best true score | 0xffffffff = beyond the packed range, ref_end, query_end, -) so far
    // score_kernel_w32 as the reverse pass of sw_simd_score_ranges (null = forward): `ref` is then the REVERSED reference, a read
    // takes part from row ref_len - rev_ref_end[id] on with reverse(read[..rev_query_end[id]]), and the outputs are the starts.
    const uint32_t* rev_ref_end;
    const uint32_t* rev_query_end;
    // non-null: the number of items is read from the device (a worklist filled by an earlier kernel of the stream); the grid
    // is sized for b.n_items, blocks past the list return at once
    const uint32_t* n_items_dev = nullptr;
    // ... and the launch does nothing unless that number lies in [gate_lo, gate_hi): the same worklist is launched in several strip
    // configurations (many lanes per read for a short list, few for a long one), and the count picks the one that runs
    uint32_t gate_lo = 0, gate_hi = 0xffffffffu;
    // Row-chunked launch (reads handed back by the seeded pass against a long reference: a few thousand reads walking 30,000 rows
    // each fill a fraction of the chip): chunk_rows > 0 = blockIdx.y is a chunk of reference rows, rows [y * chunk_rows -
    // chunk_overlap, (y + 1) * chunk_rows) from a zero state; a path that spans more than chunk_overlap rows cannot be positive
    // (zsw_align_dev.hpp: warmup_rows), so the largest (score, then earliest row, then earliest column) over a read's chunks is
    // the read's result (host model: tests/models/chunk_rows.cpp). Every chunk folds its result into chunk_keys[read] with one
    // atomicMax (score << 36 | ~row << 12 | ~column); chunk_finalize_kernel turns the keys into the outputs.
    uint32_t chunk_rows = 0, chunk_overlap = 0;
    unsigned long long* chunk_keys = nullptr;
};

constexpr uint32_t CHUNK_ROW_BITS = 24, CHUNK_COL_BITS = 12;  // references below 2^24 rows (the seeded pass's limit), reads of up to 4,095 bases
__host__ __device__ inline unsigned long long chunk_key(uint32_t score, uint32_t row1, uint32_t col1) {  // row1 / col1: 1-based, 0 = none
    return ((unsigned long long)score << (CHUNK_ROW_BITS + CHUNK_COL_BITS)) | ((unsigned long long)(((1u << CHUNK_ROW_BITS) - 1u) - row1) << CHUNK_COL_BITS) |
           (unsigned long long)(((1u << CHUNK_COL_BITS) - 1u) - col1);
}

template <int G, int C, int MODE, bool WIDE = false, bool TILED = false>
__global__ __launch_bounds__(BLOCK, min_waves(C, MODE)) void score_kernel_v2(ScoreArgsV2 a) {
    __shared__ uint2 rp[WIDE ? 1 : CH + G];
    __shared__ uint16_t rpw[WIDE ? CH + G : 1];      // WIDE: byte offset of each staged row's table row
    __shared__ uint32_t wt32[WIDE ? 33 * 9 : 1];     // WIDE: the score table
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);

    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;
    const uint32_t itemA = 2 * group, itemB = 2 * group + 1;
    const uint32_t n_items = a.n_items_dev ? min(*a.n_items_dev, a.b.n_items) : a.b.n_items;
    if (2 * blockIdx.x * (BLOCK / G) >= n_items || n_items < a.gate_lo || n_items >= a.gate_hi) return;
    const bool validA = itemA < n_items, validB = itemB < n_items;
    const uint32_t idA = validA ? (a.b.items ? a.b.items[itemA] : itemA) : 0;
    const uint32_t idB = validB ? (a.b.items ? a.b.items[itemB] : itemB) : 0;

    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    if (tid < 9) swt[tid] = make_uint2(a.wtab[tid][0], a.wtab[tid][1]);
    if (WIDE) {
        for (int i = tid; i < 33 * 9; i += BLOCK) wt32[WIDE ? i : 0] = reinterpret_cast<const uint32_t*>(a.wide)[i];
    }
    __syncthreads();
    const int8_t* wt = reinterpret_cast<const int8_t*>(wt32);

    uint64_t offA = 0, offB = 0;
    uint32_t lenA = 0, lenB = 0;
    if (validA) {
        if (a.b.offsets) {
            offA = a.b.offsets[idA];
            lenA = (uint32_t)(a.b.offsets[idA + 1] - offA);
        } else {
            offA = (uint64_t)idA * a.b.fixed_len;
            lenA = a.b.fixed_len;
        }
    }
    if (validB) {
        if (a.b.offsets) {
            offB = a.b.offsets[idB];
            lenB = (uint32_t)(a.b.offsets[idB + 1] - offB);
        } else {
            offB = (uint64_t)idB * a.b.fixed_len;
            lenB = a.b.fixed_len;
        }
    }

    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t q = (TILED ? a.tile_q0 : 0u) + (uint32_t)(g * C + c);
        uint32_t kA = PAD_K, kB = PAD_K;
        if (q < lenA) kA = lut[a.b.bases[offA + q]];
        if (q < lenB) kB = lut[a.b.bases[offB + q]];
        // residue 0..3: signed byte 2k+1, sign-extended by selector 8+k; residue 4..6: byte 2(k-3), zero-extended; pad: byte 0
        if (WIDE) {  // the residues themselves: byte offsets into the row of the LDS table
            sel[c] = (kA == PAD_K ? (uint32_t)WIDE_PAD : kA) | ((kB == PAD_K ? (uint32_t)WIDE_PAD : kB) << 16);
        } else {
            const uint32_t sA = kA < 4 ? (2 * kA + 1) | ((8 + kA) << 8) : (kA == PAD_K ? 0x0c00u : (2 * (kA - 3)) | 0x0c00u);
            const uint32_t sB = kB < 4 ? (2 * kB + 1) | ((8 + kB) << 8) : (kB == PAD_K ? 0x0c00u : (2 * (kB - 3)) | 0x0c00u);
            sel[c] = sA | (sB << 16);
        }
    }

    const uint32_t ge2 = a.ge2, gd2 = a.gd2;
    const uint32_t ge1 = ge2 & 0xffffu;
    const uint32_t K = a.K;
    const uint32_t Kge2 = (K * ge1) * 0x00010001u;
    // row of this lane at step t is r = t - g; D_r = FLOOR + r*ge with FLOOR >= 1152 + (G-1)*ge, so D_{-g-1} stays a normal f16
    uint32_t Dr = (a.floor0 - (uint32_t)(g + 1) * ge1) * 0x00010001u;  // D of row r-1 (advanced to D_r at the top of each step)
    uint32_t H[C], E[C];
    uint32_t snap[MODE == 2 ? C : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        H[c] = Dr;                   // H~_{r-1} = true 0 of row r-1
        E[c] = pk_addu(Dr, ge2);     // E~_r = true 0 of row r
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = 0;
    }
    uint32_t snapD = 0;
    uint32_t best = 0;  // true scores (no offset)
    uint32_t Fout = Dr, Hlast = Dr, Hin_prev = Dr;
    int rA = 0, rB = 0;
    // a row-chunked launch sees rows [row_off, row_off + R) as its reference
    int row_off = 0, R = (int)a.ref_len;
    if (!TILED && a.chunk_rows) {
        const int lo = max(0, (int)(blockIdx.y * a.chunk_rows) - (int)a.chunk_overlap), hi = min(R, (int)((blockIdx.y + 1) * a.chunk_rows));
        row_off = lo;
        R = max(hi - lo, 0);
    }
    const uint8_t* const refp = a.ref + row_off;
    const int T = R + G - 1;
    uint2 bd = make_uint2(0u, 0u);  // TILED: the boundary of the row lane 0 reaches next
    if (TILED) {
        if (g == 0 && validA && a.tile_in != nullptr && R > 0) bd = a.tile_in[(size_t)group * (size_t)R];
    }

    for (int base = 0; base < T; base += CH) {
        __syncthreads();
        for (int j = tid; j < CH + G - 1; j += BLOCK) {
            const int row = base - (G - 1) + j;
            if (WIDE) {
                int idx = WIDE_NEUTRAL;
                if (row >= 0 && row < R) idx = lut[refp[row]];
                rpw[WIDE ? j : 0] = (uint16_t)(idx * WIDE_STRIDE);
            } else {
                int idx = NEUTRAL;
                if (row >= 0 && row < R) idx = lut[refp[row]];
                rp[WIDE ? 0 : j] = swt[idx];
            }
        }
        __syncthreads();
        const int tend = (T < base + CH) ? T : base + CH;
        const int joff = (G - 1 - g) - base;
        uint2 w = WIDE ? make_uint2(rpw[WIDE ? base + joff : 0], 0) : rp[WIDE ? 0 : base + joff];
        // column score of the lane's current row for both reads, as a packed pair of i16 (score + ge)
        auto lookup = [&](const uint2 ww, const uint32_t sl) -> uint32_t {
            if constexpr (WIDE) {
                const int sa = wt[ww.x + (sl & 0xffffu)], sb = wt[ww.x + (sl >> 16)];
                return __builtin_amdgcn_perm((uint32_t)sb, (uint32_t)sa, 0x05040100u);
            } else {
                return __builtin_amdgcn_perm(ww.y, ww.x, sl);
            }
        };
#pragma unroll 1
        for (int t = base; t < tend; ++t) {
            const uint2 wn = WIDE ? make_uint2(rpw[WIDE ? t + 1 + joff : 0], 0) : rp[WIDE ? 0 : t + 1 + joff];
            const int row = t - g;
            // re-base the lane's state when its row index reaches a multiple of K (uniform branch, rare)
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
            Dr = pk_addu(Dr, ge2);                 // D_r
            const uint32_t Dn = pk_addu(Dr, ge2);  // D_{r+1}
            uint32_t Fin = (uint32_t)__shfl_up((int)Fout, 1, G);
            uint32_t Hin = (uint32_t)__shfl_up((int)Hlast, 1, G);
            if (g == 0) {
                Fin = Dr;
                Hin = Dr;
                if (TILED) {  // the strip to the left belongs to the previous tile: its row-r boundary comes from HBM (same D_r)
                    if (validA && a.tile_in != nullptr && row >= 0 && row < R) {  // lane groups past the batch own no boundary rows
                        Hin = bd.x;
                        Fin = bd.y;
                    }
                    if (validA && a.tile_in != nullptr && row + 1 < R) bd = a.tile_in[(size_t)group * (size_t)R + (size_t)(row + 1)];
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
                // plain 32-bit subtractions: both halves of every stored value are >= 0x0400 > go - ge, ge, so no borrow crosses
                // the halves — and v_sub_u32 issues in about 0.6 of a packed instruction's time on gfx950
                const uint32_t hg = h - gd2;
                E[c] = pk_max3(E[c], hg, Dn);
                F = pk_max3(F, hg, Dn) - ge2;
                if (c & 1) rmax = pk_max3(rmax, H[c - (c & 1)], h);
                else if (c == C - 1) rmax = pk_max3(rmax, h, h);
                hd = hd_next;
            }
            Fout = F;
            Hlast = H[C - 1];
            if (TILED) {
                if (g == G - 1 && validA && a.tile_out != nullptr && row >= 0 && row < R)
                    a.tile_out[(size_t)group * (size_t)R + (size_t)row] = make_uint2(Hlast, Fout);
            }
            const uint32_t tmax = pk_subu(rmax, Dr);  // true row maximum (>= 0: every H~ >= D_r)
            const uint32_t nb = pk_maxu(best, tmax);
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
    }

    // ---- per-read reduction over the G lanes of the group (true scores) ----
    int bA = (int)(best & 0xffffu), bB = (int)(best >> 16);
    int gbA = bA, gbB = bB;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        gbA = max(gbA, __shfl_xor(gbA, d, G));
        gbB = max(gbB, __shfl_xor(gbB, d, G));
    }
    uint32_t reA = 0, reB = 0, qeA = 0, qeB = 0;
    if (MODE != 0) {
        int kA = (bA == gbA) ? rA : 0x7fffffff, kB = (bB == gbB) ? rB : 0x7fffffff;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) {
            kA = min(kA, __shfl_xor(kA, d, G));
            kB = min(kB, __shfl_xor(kB, d, G));
        }
        reA = (uint32_t)kA + 1;
        reB = (uint32_t)kB + 1;
        if (MODE == 2) {
            int cA = 0x7fffffff, cB = 0x7fffffff;
            const int dA = (int)(snapD & 0xffffu), dB = (int)(snapD >> 16);
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t sv = snap[MODE == 2 ? c : 0];
                if ((int)(sv & 0xffffu) - dA == gbA) cA = (TILED ? (int)a.tile_q0 : 0) + g * C + c;
                if ((int)(sv >> 16) - dB == gbB) cB = (TILED ? (int)a.tile_q0 : 0) + g * C + c;
            }
            if (!(bA == gbA && rA == kA)) cA = 0x7fffffff;
            if (!(bB == gbB && rB == kB)) cB = 0x7fffffff;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) {
                cA = min(cA, __shfl_xor(cA, d, G));
                cB = min(cB, __shfl_xor(cB, d, G));
            }
            qeA = (uint32_t)cA + 1;
            qeB = (uint32_t)cB + 1;
        }
    }

    const int lane = tid & 63;
    constexpr int RW = 2 * (64 / G);
    const int src = (lane >> 1) * G;
    const bool hi = lane & 1;
    auto pick = [&](int va, int vb) {
        const int xa = __shfl(va, src, 64), xb = __shfl(vb, src, 64);
        return hi ? xb : xa;
    };
    const uint32_t o_valid = (uint32_t)pick((int)validA, (int)validB);
    const uint32_t o_id = (uint32_t)pick((int)idA, (int)idB);
    const uint32_t o_len = (uint32_t)pick((int)lenA, (int)lenB);
    int o_true = pick(gbA, gbB);
    uint32_t o_re = (uint32_t)pick((int)reA, (int)reB);
    uint32_t o_qe = (uint32_t)pick((int)qeA, (int)qeB);
    if (TILED) {
        // fold this tile into the read's running result: larger score, then earlier row, then earlier column (= earlier tile)
        if (lane < RW && o_valid) {
            uint32_t best_u = (uint32_t)o_true >= a.limit ? 0xffffffffu : (uint32_t)o_true;
            if (a.tile_in != nullptr) {
                const uint4 prev = a.tile_state[o_id];
                const bool keep_prev = prev.x == 0xffffffffu || (best_u != 0xffffffffu && (prev.x > best_u || (prev.x == best_u && prev.y <= o_re)));
                if (keep_prev) {
                    best_u = prev.x;
                    o_re = prev.y;
                    o_qe = prev.z;
                }
            }
            if (a.tile_out != nullptr) a.tile_state[o_id] = make_uint4(best_u, o_re, o_qe, 0u);
            o_true = best_u == 0xffffffffu ? 0x7fffffff : (int)best_u;
        }
        if (a.tile_out != nullptr) return;  // results are written by the last tile
    }
    if (!TILED && a.chunk_rows) {  // one chunk of the read's rows: fold (score, first row, first column) into the read's key
        if (lane < RW && o_valid && o_len != 0 && o_true > 0)
            atomicMax(&a.chunk_keys[o_id], chunk_key((uint32_t)min(o_true, 0xffff), MODE != 0 ? o_re + (uint32_t)row_off : 0u, MODE == 2 ? o_qe : 0u));
        return;
    }
    if (lane < RW && o_valid) {
        if (o_len == 0) {
            a.out.score[o_id] = 0;
            a.out.status[o_id] = ZSW_STATUS_EMPTY;
            if (a.out.tier) a.out.tier[o_id] = 0;
            if (MODE != 0 && a.out.ref_end) a.out.ref_end[o_id] = 0;
            if (MODE == 2 && a.out.query_end) a.out.query_end[o_id] = 0;
        } else if ((uint32_t)o_true >= a.limit) {  // near the representable limit: recompute exactly in 32 bits
            const uint32_t k = atomicAdd(a.out.fb_count, 1u);
            a.out.fb_list[k] = o_id;
        } else {
            uint32_t score;
            uint8_t status, tier;
            apply_rule(a.rule, (uint64_t)o_true, &score, &status, &tier);
            a.out.score[o_id] = score;
            a.out.status[o_id] = status;
            if (a.out.tier) a.out.tier[o_id] = tier;
            const bool some = status == ZSW_STATUS_SOME;
            if (MODE != 0 && a.out.ref_end) a.out.ref_end[o_id] = some ? o_re : 0;
            if (MODE == 2 && a.out.query_end) a.out.query_end[o_id] = some ? o_qe : 0;
        }
    }
}

// zsw_score_wide.hip
hipError_t launch_table_cfg_v2_wide(const ScoreArgsV2& a, int G, int C, int mode, hipStream_t stream);
// one tile (columns [a.tile_q0, a.tile_q0 + TILE_COLS)) of reads longer than the widest strip configuration; `wide` picks the
// LDS-table form. The caller sets tile_q0 / tile_in / tile_out / tile_state.
constexpr int TILE_G = 64, TILE_C = 38, TILE_COLS = TILE_G * TILE_C;
hipError_t launch_tile_v2(const ScoreArgsV2& a, bool wide, int mode, hipStream_t stream);
// zsw_score_w32.hip: the same tile in 32-bit lanes over the reads list[0 .. a.b.n_items) (scores beyond the packed range)
hipError_t launch_tile_w32(const ScoreArgsV2& a, const uint32_t* list, int mode, hipStream_t stream);

}  // namespace zsw
