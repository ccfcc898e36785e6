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
#include "zsw_score_v2.hpp"
#include "zsw_timer.hpp"

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
};

// MODE 0: score; 1: score + ref_end; 2: score + ref_end + query_end
// REV (with MODE 2): the second pass of sw_simd_score_ranges (striped.rs:355-388) — sw_simd_score_ends_reverse on
// `reference[..ref_end]` with the profile of `reverse(read[..query_end])` (profile.rs:314-350). Every read has its own
// reference prefix, so a lane carries ONE read (high half idle), takes its row table from HBM/L2 (gtab[ref_end-1-row])
// and each wave runs only as many steps as its longest prefix needs. Outputs: ref_end/query_end receive the STARTS.
template <int G, int C, bool FAST, int MODE, bool REV = false>
__global__ __launch_bounds__(BLOCK, min_waves(C, MODE)) void score_kernel(ScoreArgs a) {
    __shared__ uint2 rp[CH + G];
    __shared__ uint2 swt[9];
    __shared__ uint32_t lut32[64];
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(lut32);

    const int tid = threadIdx.x;
    const int g = tid & (G - 1);
    const uint32_t group = blockIdx.x * (BLOCK / G) + tid / G;
    const uint32_t itemA = REV ? group : 2 * group, itemB = 2 * group + 1;
    const bool validA = itemA < a.b.n_items, validB = !REV && itemB < a.b.n_items;
    const uint32_t idA = validA ? (a.b.items ? a.b.items[itemA] : itemA) : 0;
    const uint32_t idB = validB ? (a.b.items ? a.b.items[itemB] : itemB) : 0;

    if (tid < 64) lut32[tid] = reinterpret_cast<const uint32_t*>(a.sc->index_map)[tid];
    if (tid < 9) swt[tid] = make_uint2(a.wtab[tid][0], a.wtab[tid][1]);
    __syncthreads();

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
    int rev_re = 0;  // REV: this read's reference prefix length (forward ref_end) and query prefix length
    if (REV) {
        const uint32_t qe = validA ? a.rev_query_end[idA] : 0;
        lenA = qe <= lenA ? qe : lenA;
        rev_re = validA && lenA ? (int)a.rev_ref_end[idA] : 0;
    }

    // per-column selectors: which table bytes v_perm picks for read A (low half) and read B (high half)
    uint32_t sel[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t q = (uint32_t)(g * C + c);
        uint32_t kA = PAD_K, kB = PAD_K;
        if (q < lenA) kA = lut[a.b.bases[REV ? offA + (lenA - 1 - q) : offA + q]];
        if (q < lenB) kB = lut[a.b.bases[offB + q]];
        uint32_t sA, sB;
        if (FAST) {  // table = W[r][0..3] as i16; residue >= 4 (an all-zero matrix column) and padding -> 0
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
    auto step = [&](const uint2 w, const int row) {
        uint32_t Fin = (uint32_t)__shfl_up((int)Fout, 1, G);
        uint32_t Hin = (uint32_t)__shfl_up((int)Hlast, 1, G);
        if (g == 0) {
            Fin = MIN2;
            Hin = MIN2;
        }
        // hd = H(r-1,c-1) + W(r,c) is formed one column ahead, so the previous row's H[c] is dead
        // before this row's H[c] is written (same register, no copy in the loop).
        uint32_t hd = pk_adds(Hin_prev, __builtin_amdgcn_perm(w.y, w.x, sel[0]));
        if (!FAST) hd = pk_subs(hd, bias2);
        Hin_prev = Hin;
        uint32_t F = Fin;
        uint32_t rmax = MIN2;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t hd_next = 0;
            if (c + 1 < C) {
                hd_next = pk_adds(H[c], __builtin_amdgcn_perm(w.y, w.x, sel[c + 1 < C ? c + 1 : c]));
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
                rp[j] = swt[idx];
            }
            __syncthreads();
            const int tend = (T < base + CH) ? T : base + CH;
            const int joff = (G - 1 - g) - base;
            uint2 w = rp[base + joff];
#pragma unroll 1
            for (int t = base; t < tend; ++t) {
                const uint2 wn = rp[t + 1 + joff];
                step(w, t - g);
                w = wn;
            }
        }
    } else {
        int tw = rev_re;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tw = max(tw, __shfl_xor(tw, d, 64));
        tw = tw ? tw + G - 1 : 0;  // steps of this wave: its longest reference prefix plus the strip skew
        const uint2 neutral = swt[NEUTRAL];
        auto row_entry = [&](int row) {
            const int rr = rev_re - 1 - row;
            return (row >= 0 && rr >= 0) ? a.gtab[rr] : neutral;
        };
        // The reverse problem's maximum equals the forward score (the same alignment read backwards), and its tie rule wants
        // the FIRST row that holds the maximum: once a read's running maximum has reached the forward score, and every lane
        // of its group has walked past that row (G more steps), nothing later can change its answer. Checked every 8 steps;
        // the wave leaves when all of its reads are finished — about the alignment's own span instead of the whole prefix.
        const int target = (validA && lenA) ? (int)a.rev_score[idA] - 32768 : 0x7fffffff;  // stored (offset) domain
        int t_done = (validA && lenA && rev_re > 0) ? 0x3fffffff : -1000000;
        uint2 w = row_entry(-g);
#pragma unroll 1
        for (int t = 0; t < tw; ++t) {
            const uint2 wn = row_entry(t + 1 - g);
            step(w, t - g);
            w = wn;
            if ((t & 7) == 7) {
                int gm = (int)(int16_t)(best & 0xffffu);
#pragma unroll
                for (int d = 1; d < G; d <<= 1) gm = max(gm, __shfl_xor(gm, d, G));
                if (gm >= target && t_done > t) t_done = t;
                if (__ballot(t < t_done + G) == 0) break;
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
    constexpr int RW = REV ? 64 / G : 2 * (64 / G);  // reads per wave
    const int src = REV ? lane * G : (lane >> 1) * G;  // first lane of the group that holds read `lane`
    const bool hi = !REV && (lane & 1);
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
static const Cfg kCfgs[] = {{4, 19}, {4, 25}, {4, 32}, {4, 38}, {8, 19}, {8, 25}, {8, 32}, {8, 38},
                            {16, 25}, {16, 32}, {16, 38}, {64, 19}, {64, 38}};

bool score_config_for(uint32_t max_len, int* G, int* C) {
    for (const Cfg& c : kCfgs)
        if ((uint32_t)(c.G * c.C) >= max_len) {
            *G = c.G;
            *C = c.C;
            return true;
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
    if (mode == 0) hipLaunchKernelGGL((score_kernel_v2<G, C, 0>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else if (mode == 1) hipLaunchKernelGGL((score_kernel_v2<G, C, 1>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((score_kernel_v2<G, C, 2>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

static hipError_t launch_table_cfg_v2(const ScoreArgsV2& a, int G, int C, int mode, hipStream_t stream) {
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: return launch_cfg_v2<GV, CV>(a, mode, stream);
        ZSW_CASE(4, 19)
        ZSW_CASE(4, 25)
        ZSW_CASE(4, 32)
        ZSW_CASE(4, 38)
        ZSW_CASE(8, 19)
        ZSW_CASE(8, 25)
        ZSW_CASE(8, 32)
        ZSW_CASE(8, 38)
        ZSW_CASE(16, 25)
        ZSW_CASE(16, 32)
        ZSW_CASE(16, 38)
        ZSW_CASE(64, 19)
        ZSW_CASE(64, 38)
#undef ZSW_CASE
    }
    return hipErrorInvalidValue;
}

// v2 needs: S <= 7; for query residues 0..3: s + ge in [-128, 127]; for residues >= 4: s + ge in [0, 255].
static bool v2_ok(const ScoringDev& s) {
    if (getenv("ZSW_SCORE_V1")) return false;
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

// WIDE kernels: 8..32 letters, every score + ge must fit a signed byte.
static bool wide_ok(const ScoringDev& s) {
    if (getenv("ZSW_SCORE_NO_WIDE")) return false;
    if (s.S <= 7 || s.S > 32) return false;
    for (int i = 0; i < s.S * s.S; ++i) {
        const int t = s.w[i] + s.gap_extend;
        if (t < -128 || t > 127) return false;
    }
    return true;
}

static bool build_tables_wide(const ScoringDev& s, int G, ScoreArgsV2* a) {
    const int ge = s.gap_extend;
    for (int r = 0; r < 33; ++r)
        for (int q = 0; q < WIDE_STRIDE; ++q)
            a->wide[r * WIDE_STRIDE + q] = (int8_t)((r < s.S && q < s.S) ? s.w[r * s.S + q] + ge : ge);
    return v2_range_setup(s, G, a);
}

static hipError_t launch_table_cfg(const ScoreArgs& a, int G, int C, bool fast, int mode, hipStream_t stream) {
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: return launch_cfg<GV, CV>(a, fast, mode, stream);
        ZSW_CASE(4, 19)
        ZSW_CASE(4, 25)
        ZSW_CASE(4, 32)
        ZSW_CASE(4, 38)
        ZSW_CASE(8, 19)
        ZSW_CASE(8, 25)
        ZSW_CASE(8, 32)
        ZSW_CASE(8, 38)
        ZSW_CASE(16, 25)
        ZSW_CASE(16, 32)
        ZSW_CASE(16, 38)
        ZSW_CASE(64, 19)
        ZSW_CASE(64, 38)
#undef ZSW_CASE
    }
    return hipErrorInvalidValue;
}

// ---- ragged batches: group the reads by the smallest strip configuration that holds them -------------------
// class k < NCLS: kCfgs[kBucketCfg[k]]; class NCLS: longer than every table configuration (exact 32-bit kernel)
static const int kBucketCfg[] = {0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12};  // (8,19) duplicates the capacity of (4,38)
constexpr int NCLS = 12;

struct BucketCaps {
    uint32_t cap[NCLS];
};

__device__ __forceinline__ int bucket_of(const BucketCaps& caps, uint32_t len) {
    int k = 0;
    while (k < NCLS && len > caps.cap[k]) ++k;
    return k;
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
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int k = bucket_of(caps, (uint32_t)(offsets[i + 1] - offsets[i]));
        items[atomicAdd(&cursors[k], 1u)] = i;
    }
}

hipError_t launch_score(const ScoringDev* d_sc, const ScoringDev& h_sc, const BatchDev& b, uint32_t max_len,
                        const uint8_t* d_ref, uint32_t ref_len, const ResultRule& rule, const ScoreOut& out,
                        const ScoreWorkspace& ws, hipStream_t stream, KernelTimer* timer, int mode) {
    hipError_t e = hipMemsetAsync(out.fb_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    int G = 0, C = 0;
    const bool wide = wide_ok(h_sc);
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
    const bool use_v2 = table_ok && !wide && v2_ok(h_sc);
    ScoreArgsV2 a2;
    a2.b = b;
    a2.ref = d_ref;
    a2.ref_len = ref_len;
    a2.sc = d_sc;
    a2.rule = rule;
    a2.out = out;
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
    auto exact_all = [&](const BatchDev& bb) {
        hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, bb, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, d_ref, ref_len, d_sc, rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
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
        for (int k = 0; k <= NCLS; ++k) {
            if (!counts[k]) continue;
            BatchDev bk = b;
            bk.items = ws.bucket_items + starts[k];
            bk.n_items = counts[k];
            if (k == NCLS) {
                e = exact_all(bk);
            } else {
                e = launch_one(bk, kCfgs[kBucketCfg[k]].G, kCfgs[kBucketCfg[k]].C);
            }
            if (e != hipSuccess) return e;
        }
        if (timer) timer->end(stream);
    } else {
        if (!score_config_for_batch(max_len, b.n_items, &G, &C)) {  // longer than every strip configuration
            if (timer) timer->begin(stream);
            e = exact_all(b);
            if (timer) timer->end(stream);
            return e;
        }
        if (timer) timer->begin(stream);
        e = launch_one(b, G, C);
        if (timer) timer->end(stream);
        if (e != hipSuccess) return e;
    }
    // reads that saturated i16: exact pass over the device-side worklist (usually empty)
    hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, out.fb_list, out.fb_count, d_ref, ref_len, d_sc,
                       rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    return hipGetLastError();
}

// ---- reverse pass of sw_simd_score_ranges -------------------------------------------------------------------
__global__ void gtab_kernel(const uint8_t* ref, uint32_t n, const ScoringDev* sc, ScoreArgs a, uint2* gtab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int idx = sc->index_map[ref[i]];
        gtab[i] = make_uint2(a.wtab[idx][0], a.wtab[idx][1]);
    }
}

template <int G, int C>
static hipError_t launch_cfg_rev(const ScoreArgs& a, bool fast, hipStream_t stream) {
    const uint32_t reads_per_block = BLOCK / G;
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
    const bool table_ok = (h_sc.S <= 7 || fast_ok(h_sc)) && score_config_for(max_len, &G, &C);
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
    build_tables(h_sc, fast, &a);
    if (ref_len) hipLaunchKernelGGL(gtab_kernel, dim3((ref_len + 255) / 256), dim3(256), 0, stream, d_ref, ref_len, d_sc, a, d_gtab);
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: e = launch_cfg_rev<GV, CV>(a, fast, stream); break;
        ZSW_CASE(4, 19)
        ZSW_CASE(4, 25)
        ZSW_CASE(4, 32)
        ZSW_CASE(4, 38)
        ZSW_CASE(8, 19)
        ZSW_CASE(8, 25)
        ZSW_CASE(8, 32)
        ZSW_CASE(8, 38)
        ZSW_CASE(16, 25)
        ZSW_CASE(16, 32)
        ZSW_CASE(16, 38)
        ZSW_CASE(64, 19)
        ZSW_CASE(64, 38)
#undef ZSW_CASE
        default: e = hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exact32_kernel, dim3(exact_grid), dim3(64), 0, stream, b, out.fb_list, out.fb_count, d_ref, ref_len, d_sc,
                       rule, out, ws.scratch, (uint32_t)ws.slots, ws.scratch_len, d_fwd_ref_end, d_fwd_query_end);
    return hipGetLastError();
}

}  // namespace zsw
