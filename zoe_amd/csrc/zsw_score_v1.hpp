// zsw_score_v1.hpp — the saturating packed score kernel (score_kernel) and its reverse-pass form, shared by zsw_score.hip
// and zsw_score_wide.hip (reverse pass for alphabets of 8..32 letters).
#pragma once
#include "zsw_internal.hpp"
#include "zsw_score_v2.hpp"

namespace zsw {

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s2 S2(uint32_t x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ uint32_t U(s2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return U(__builtin_elementwise_add_sat(S2(a), S2(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return U(__builtin_elementwise_sub_sat(S2(a), S2(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return U(__builtin_elementwise_max(S2(a), S2(b))); }

struct ScoreArgs {
    BatchDev b;
    const uint8_t* ref;
    uint32_t ref_len;
    const ScoringDev* sc;
    uint32_t wtab[9][2];  // per reference residue: the 8 table bytes v_perm selects from
    uint32_t go2, ge2, bias2;
    ResultRule rule;
    ScoreOut out;
    // reverse pass of sw_simd_score_ranges (REV kernels): per read, the forward ends and the per-row table in HBM
    const uint32_t* rev_ref_end;
    const uint32_t* rev_query_end;
    const uint32_t* rev_score;  // forward score: the reverse pass may stop once it has been reached (see the REV loop)
    const uint2* gtab;
    // WIDE reverse pass (alphabets of 8..32 letters): wide[r*WIDE_STRIDE + q] = (int8) score(r, q); column WIDE_PAD and
    // row WIDE_NEUTRAL score 0; gtab[i].x = byte offset of reference residue i's row
    int8_t wide[33 * 36];
};

// MODE 0: score; 1: score + ref_end; 2: score + ref_end + query_end
// REV (with MODE 2): the second pass of sw_simd_score_ranges (striped.rs:355-388) — sw_simd_score_ends_reverse on
// `reference[..ref_end]` with the profile of `reverse(read[..query_end])` (profile.rs:314-350). Every read has its own
// reference prefix: the two reads of a lane take their row tables separately from HBM/L2 (gtab[ref_end-1-row]; one v_perm per
// half and a merge instead of one v_perm per column) and each wave runs only as many steps as its longest prefix needs.
// Outputs: ref_end/query_end receive the STARTS.
template <int G, int C, bool FAST, int MODE, bool REV = false, bool WIDE = false>
__global__ __launch_bounds__(BLOCK, min_waves(C, MODE)) void score_kernel(ScoreArgs a) {
    static_assert(!WIDE || (REV && FAST), "the WIDE table form exists for the reverse pass only (signed scores, no bias)");
    __shared__ uint2 rp[REV ? 1 : CH + G];
    __shared__ uint32_t wt32[WIDE ? 33 * 9 : 1];
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);

    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;
    const uint32_t itemA = 2 * group, itemB = 2 * group + 1;
    const bool validA = itemA < a.b.n_items, validB = itemB < a.b.n_items;
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
    int rev_reA = 0, rev_reB = 0;  // REV: the reads' reference prefix lengths (forward ref_end); lenA/lenB become the query prefixes
    if (REV) {
        const uint32_t qeA = validA ? a.rev_query_end[idA] : 0, qeB = validB ? a.rev_query_end[idB] : 0;
        lenA = qeA <= lenA ? qeA : lenA;
        lenB = qeB <= lenB ? qeB : lenB;
        rev_reA = validA && lenA ? (int)a.rev_ref_end[idA] : 0;
        rev_reB = validB && lenB ? (int)a.rev_ref_end[idB] : 0;
    }

    // per-column selectors: which table bytes v_perm picks for read A (low half) and read B (high half)
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t q = (uint32_t)(g * C + c);
        uint32_t kA = PAD_K, kB = PAD_K;
        if (q < lenA) kA = lut[a.b.bases[REV ? offA + (lenA - 1 - q) : offA + q]];
        if (q < lenB) kB = lut[a.b.bases[REV ? offB + (lenB - 1 - q) : offB + q]];
        uint32_t sA, sB;
        if (WIDE) {  // the residue itself: byte offset into the LDS table row
            sA = kA == PAD_K ? (uint32_t)WIDE_PAD : kA;
            sB = kB == PAD_K ? (uint32_t)WIDE_PAD : kB;
        } else if (FAST) {  // table = W[r][0..3] as i16; residue >= 4 (an all-zero matrix column) and padding -> 0
            sA = kA < 4 ? 0x0100u + kA * 0x0202u : 0x0c0cu;
            sB = kB < 4 ? 0x0100u + kB * 0x0202u : 0x0c0cu;
        } else {  // table = biased u8 weights in bytes 0..6, byte 7 = bias (padding scores 0)
            sA = (kA == PAD_K ? 7u : kA) | 0x0c00u;
            sB = (kB == PAD_K ? 7u : kB) | 0x0c00u;
        }
        sel[c] = sA | (sB << 16);
    }

    uint32_t H[C], E[C];
    uint32_t snap[MODE == 2 ? C : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        H[c] = MIN2;
        E[c] = MIN2;
    }
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = MIN2;
    }
    uint32_t best = MIN2;  // running maximum (MODE 0) / best row maximum so far (MODE >= 1)
    uint32_t Fout = MIN2, Hlast = MIN2, Hin_prev = MIN2;
    int rA = 0, rB = 0;
    const uint32_t go2 = a.go2, ge2 = a.ge2, bias2 = a.bias2;
    const int R = (int)a.ref_len;
    const int T = R + G - 1;

    // one DP row of this lane's strip; `w` = the row's table entry, `row` = its index (for the end tracking)
    // `ww`: the row's table entry for read A (REV: `wb` the one for read B — the reads walk different reference prefixes)
    auto lookup = [&](const uint2 ww, const uint2 wb, const uint32_t sl) -> uint32_t {
        if constexpr (WIDE) {
            const int sa = wt[ww.x + (sl & 0xffffu)], sb = wt[(REV ? wb.x : ww.x) + (sl >> 16)];
            return __builtin_amdgcn_perm((uint32_t)sb, (uint32_t)sa, 0x05040100u);
        } else if constexpr (REV) {
            return __builtin_amdgcn_perm(__builtin_amdgcn_perm(wb.y, wb.x, sl), __builtin_amdgcn_perm(ww.y, ww.x, sl), 0x07060100u);
        } else {
            return __builtin_amdgcn_perm(ww.y, ww.x, sl);
        }
    };
    auto step = [&](const uint2 w, const uint2 wb, const int row) {
        uint32_t Fin = (uint32_t)__shfl_up((int)Fout, 1, G);
        uint32_t Hin = (uint32_t)__shfl_up((int)Hlast, 1, G);
        if (g == 0) {
            Fin = MIN2;
            Hin = MIN2;
        }
        // hd = H(r-1,c-1) + W(r,c) is formed one column ahead, so the previous row's H[c] is dead
        // before this row's H[c] is written (same register, no copy in the loop).
        uint32_t hd = pk_adds(Hin_prev, lookup(w, wb, sel[0]));
        if (!FAST) hd = pk_subs(hd, bias2);
        Hin_prev = Hin;
        uint32_t F = Fin;
        uint32_t rmax = MIN2;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t hd_next = 0;
            if (c + 1 < C) {
                hd_next = pk_adds(H[c], lookup(w, wb, sel[c + 1 < C ? c + 1 : c]));
                if (!FAST) hd_next = pk_subs(hd_next, bias2);
            }
            if (MODE == 0) best = pk_max(best, hd);  // E, F never exceed an H already folded into best
            uint32_t h = pk_max(hd, E[c]);
            h = pk_max(h, F);
            if (MODE != 0) rmax = pk_max(rmax, h);
            H[c] = h;
            const uint32_t hg = pk_subs(h, go2);
            E[c] = pk_max(pk_subs(E[c], ge2), hg);
            F = pk_max(pk_subs(F, ge2), hg);
            hd = hd_next;
        }
        Fout = F;
        Hlast = H[C - 1];
        if (MODE != 0) {
            const uint32_t nb = pk_max(best, rmax);
            const uint32_t ch = nb ^ best;  // a non-zero half = that read's maximum rose in this row
            best = nb;
            if (ch & 0xffffu) rA = row;
            if (ch >> 16) rB = row;
            if (MODE == 2) {
                const uint32_t m = ((ch & 0xffffu) ? 0xffffu : 0u) | ((ch >> 16) ? 0xffff0000u : 0u);
#pragma unroll
                for (int c = 0; c < C; ++c) snap[MODE == 2 ? c : 0] = (H[c] & m) | (snap[MODE == 2 ? c : 0] & ~m);
            }
        }
    };
    if (!REV) {
        for (int base = 0; base < T; base += CH) {
            __syncthreads();
            for (int j = tid; j < CH + G - 1; j += BLOCK) {
                const int row = base - (G - 1) + j;
                int idx = NEUTRAL;
                if (row >= 0 && row < R) idx = lut[a.ref[row]];
                rp[REV ? 0 : j] = swt[idx];
            }
            __syncthreads();
            const int tend = (T < base + CH) ? T : base + CH;
            const int joff = (G - 1 - g) - base;
            uint2 w = rp[REV ? 0 : base + joff];
#pragma unroll 1
            for (int t = base; t < tend; ++t) {
                const uint2 wn = rp[REV ? 0 : t + 1 + joff];
                step(w, w, t - g);
                w = wn;
            }
        }
    } else {
        int tw = max(rev_reA, rev_reB);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tw = max(tw, __shfl_xor(tw, d, 64));
        tw = tw ? tw + G - 1 : 0;  // steps of this wave: its longest reference prefix plus the strip skew
        const uint2 neutral = WIDE ? make_uint2(WIDE_NEUTRAL * WIDE_STRIDE, 0) : swt[NEUTRAL];
        auto row_entry = [&](int row, int rev_re) {
            const int rr = rev_re - 1 - row;
            return (row >= 0 && rr >= 0) ? a.gtab[rr] : neutral;
        };
        // The reverse problem's maximum equals the forward score (the same alignment read backwards), and its tie rule wants
        // the FIRST row that holds the maximum: once a read's running maximum has reached the forward score, and every lane
        // of its group has walked past that row (G more steps), nothing later can change its answer. Checked every 8 steps;
        // the wave leaves when all of its reads are finished — about the alignment's own span instead of the whole prefix.
        const int targetA = (validA && lenA) ? (int)a.rev_score[idA] - 32768 : 0x7fffffff;  // stored (offset) domain
        const int targetB = (validB && lenB) ? (int)a.rev_score[idB] - 32768 : 0x7fffffff;
        int t_doneA = (validA && lenA && rev_reA > 0) ? 0x3fffffff : -1000000;
        int t_doneB = (validB && lenB && rev_reB > 0) ? 0x3fffffff : -1000000;
        uint2 w = row_entry(-g, rev_reA), wb = row_entry(-g, rev_reB);
#pragma unroll 1
        for (int t = 0; t < tw; ++t) {
            const uint2 wn = row_entry(t + 1 - g, rev_reA), wbn = row_entry(t + 1 - g, rev_reB);
            step(w, wb, t - g);
            w = wn;
            wb = wbn;
            if ((t & 7) == 7) {
                int gmA = (int)(int16_t)(best & 0xffffu), gmB = (int)(int16_t)(best >> 16);
#pragma unroll
                for (int d = 1; d < G; d <<= 1) {
                    gmA = max(gmA, __shfl_xor(gmA, d, G));
                    gmB = max(gmB, __shfl_xor(gmB, d, G));
                }
                if (gmA >= targetA && t_doneA > t) t_doneA = t;
                if (gmB >= targetB && t_doneB > t) t_doneB = t;
                if (__ballot(t < max(t_doneA, t_doneB) + G) == 0) break;
            }
        }
    }

    // ---- per-read reduction over the G lanes of the group ----
    int bA = (int)(int16_t)(best & 0xffffu), bB = (int)(int16_t)(best >> 16);
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
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t sv = snap[MODE == 2 ? c : 0];
                if ((int)(int16_t)(sv & 0xffffu) == gbA) cA = g * C + c;
                if ((int)(int16_t)(sv >> 16) == gbB) cB = g * C + c;
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

    // ---- outputs: regroup so that lane l of the wave owns the wave's l-th read (coalesced stores) ----
    const int lane = tid & 63;
    constexpr int RW = 2 * (64 / G);  // reads per wave
    const int src = (lane >> 1) * G;  // first lane of the group that holds read `lane`
    const bool hi = lane & 1;
    // both shuffles run with every lane active (a lane that sits out a divergent branch cannot be a shuffle source)
    auto pick = [&](int va, int vb) {
        const int xa = __shfl(va, src, 64), xb = __shfl(vb, src, 64);
        return hi ? xb : xa;
    };
#define ZSW_PICK(A, B) pick((int)(A), (int)(B))
    const uint32_t o_valid = (uint32_t)ZSW_PICK(validA, validB);
    const uint32_t o_id = (uint32_t)ZSW_PICK(idA, idB);
    const uint32_t o_len = (uint32_t)ZSW_PICK(lenA, lenB);
    const int o_stored = ZSW_PICK(gbA, gbB);
    const uint32_t o_re = (uint32_t)ZSW_PICK(reA, reB);
    const uint32_t o_qe = (uint32_t)ZSW_PICK(qeA, qeB);
#undef ZSW_PICK
    if (lane < RW && o_valid) {
        if (o_len == 0) {  // StripedProfile::new -> Err(ProfileError::EmptySequence)
            a.out.score[o_id] = 0;
            a.out.status[o_id] = ZSW_STATUS_EMPTY;
            if (a.out.tier) a.out.tier[o_id] = 0;
            if (MODE != 0 && a.out.ref_end) a.out.ref_end[o_id] = 0;
            if (MODE == 2 && a.out.query_end) a.out.query_end[o_id] = 0;
        } else if (o_stored >= 32767 - 256) {  // at or near i16 saturation: recompute exactly in 32 bits
            const uint32_t k = atomicAdd(a.out.fb_count, 1u);
            a.out.fb_list[k] = o_id;
        } else {
            uint32_t score;
            uint8_t status, tier;
            apply_rule(a.rule, (uint64_t)(o_stored + 32768), &score, &status, &tier);
            a.out.score[o_id] = score;
            a.out.status[o_id] = status;
            if (a.out.tier) a.out.tier[o_id] = tier;
            const bool some = status == ZSW_STATUS_SOME;
            if (REV) {  // inclusive 0-based starts (striped.rs:326-328): prefix length minus the exclusive end found here
                a.out.ref_end[o_id] = some ? a.rev_ref_end[o_id] - o_re : 0;
                a.out.query_end[o_id] = some ? o_len - o_qe : 0;
            } else {
                if (MODE != 0 && a.out.ref_end) a.out.ref_end[o_id] = some ? o_re : 0;
                if (MODE == 2 && a.out.query_end) a.out.query_end[o_id] = some ? o_qe : 0;
            }
        }
    }
}

// zsw_score_wide.hip
hipError_t launch_cfg_rev_wide(const ScoreArgs& a, int G, int C, hipStream_t stream);

}  // namespace zsw
