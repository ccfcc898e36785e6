// zsw_align_pk_kernel.hpp — align_kernel_pk<N, NV>: pass 2 of sw_simd_align (striped.rs:449-598) for the common case,
// two reads per lane group in packed 16-bit halves, lazy-F in closed form (zsw_align_pk.hpp has the derivation and is also
// compiled for the host by tests/models/align_pk_twin.cpp). Included by zsw_align_pk{8,16,32,64}.hip, one lane count each.
//
// One wavefront = 2*64/N reads of one <N, nv> group. H, E, the row's flag codes and the profile scores of the row sit in
// VGPRs at static indices; the striped profile of the wave's reads (profile.rs:270-306, an i16 score per read and dword) is
// in LDS and the next row's dwords are fetched while the current row computes. Flags of the last W rows of each read go to
// the read's own ring region, one byte per cell as in backtrack.rs:98-130 ([row % W][lane][4*ceil(nv/4)] bytes), and one lane
// per read walks to_alignment (backtrack.rs:290-342) at the end. Rows before the first kept row only carry H and E.
#pragma once
#include "zsw_align_dev.hpp"
#include "zsw_align_pk.hpp"

namespace zsw {

struct DevOps {
    using V = uint32_t;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    typedef short ss2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ us2 u(V x) { return __builtin_bit_cast(us2, x); }
    static __device__ __forceinline__ ss2 s(V x) { return __builtin_bit_cast(ss2, x); }
    static __device__ __forceinline__ V b(us2 x) { return __builtin_bit_cast(V, x); }
    static __device__ __forceinline__ V b(ss2 x) { return __builtin_bit_cast(V, x); }
    static __device__ __forceinline__ V splat(uint32_t x) { return x; }
    static __device__ __forceinline__ V add(V a, V c) { return b(u(a) + u(c)); }                                  // v_pk_add_u16
    static __device__ __forceinline__ V sub(V a, V c) { return b(u(a) - u(c)); }                                  // v_pk_sub_u16
    static __device__ __forceinline__ V add_sat(V a, V c) { return b(__builtin_elementwise_add_sat(u(a), u(c))); }  // ... clamp
    static __device__ __forceinline__ V sub_sat(V a, V c) { return b(__builtin_elementwise_sub_sat(u(a), u(c))); }  // ... clamp
    static __device__ __forceinline__ V max_u(V a, V c) { return b(__builtin_elementwise_max(u(a), u(c))); }      // v_pk_max_u16
    static __device__ __forceinline__ V min_u(V a, V c) { return b(__builtin_elementwise_min(u(a), u(c))); }      // v_pk_min_u16
    static __device__ __forceinline__ V max_i(V a, V c) { return b(__builtin_elementwise_max(s(a), s(c))); }      // v_pk_max_i16
    static __device__ __forceinline__ V mul(V a, V c) { return b(u(a) * u(c)); }                                  // v_pk_mul_lo_u16
    static __device__ __forceinline__ V mad(V a, V c, V d) { return b(u(a) * u(c) + u(d)); }                      // v_pk_mad_u16
    static __device__ __forceinline__ V and_(V a, V c) { return a & c; }
    static __device__ __forceinline__ V xor_(V a, V c) { return a ^ c; }
    static __device__ __forceinline__ V and_or(V a, V c, V d) { return (a & c) | d; }                              // v_and_or_b32
    static __device__ __forceinline__ V bfi(V m, V a, V c) { return (m & a) | (~m & c); }                          // v_bfi_b32
    static __device__ __forceinline__ V lshl_or(V a, int n, V c) { return (a << n) | c; }                          // v_lshl_or_b32
    static __device__ __forceinline__ V shr(V a, int n) { return a >> n; }
    // shift_elements_right::<1>(T::MIN) inside groups of N lanes: row_shr:1 within rows of 16 lanes, wave_shr:1 across the
    // wavefront; lanes without a source get 0, the first lane of a narrower group is cleared with `keep`
    template <int N>
    static __device__ __forceinline__ V shr1(V x, V keep) {
        if constexpr (N <= 16) {
            const V y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
            return N < 16 ? (y & keep) : y;
        } else {
            const V y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);
            return N < 64 ? (y & keep) : y;
        }
    }
    template <int N>
    static __device__ __forceinline__ V shr_d(V x, int d) {  // d is a compile-time constant after unrolling
        const int li = (int)(threadIdx.x % N);
        if constexpr (N <= 16) {
            V y;
            switch (d) {
                case 1: y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); break;
                case 2: y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); break;
                case 4: y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); break;
                default: y = (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); break;
            }
            return (N < 16 && li < d) ? 0u : y;
        } else {
            const V y = (V)__shfl_up((int)x, d, N);
            return li < d ? 0u : y;
        }
    }
    template <int N>
    static __device__ __forceinline__ V group_or(V x) {
        x |= (V)__builtin_amdgcn_update_dpp(0, (int)x, 0xb1, 0xf, 0xf, false);                        // quad_perm:[1,0,3,2]
        if constexpr (N >= 4) x |= (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x4e, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
        if constexpr (N >= 8) x |= (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false);  // row_half_mirror
        if constexpr (N >= 16) x |= (V)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false);  // row_mirror
        if constexpr (N >= 32) x |= (V)__shfl_xor((int)x, 16, 64);
        if constexpr (N >= 64) x |= (V)__shfl_xor((int)x, 32, 64);
        return x;
    }
    static __device__ __forceinline__ bool any(V x) { return __builtin_amdgcn_ballot_w64(x != 0) != 0; }
    static __device__ __forceinline__ V lead_ones(V x, int nv) {
        const V y = (~x) << (32 - nv);
        const V z = y ? (V)__builtin_clz(y) : 32u;
        return z < (V)nv ? z : (V)nv;
    }
};

template <int N, int NV>
__global__ __launch_bounds__(64, NV <= 10 ? 3 : 1) void align_kernel_pk(AlignArgs a) {
    using O = DevOps;
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint8_t lut[256];
    constexpr int RPW = 2 * 64 / N;
    constexpr int NVQ = (NV + 3) / 4;
    constexpr bool PREFETCH = NV <= 16;  // the next row's profile dwords in a second register set
    const int lane = threadIdx.x;
    const int li = lane % N, grp = lane / N;
    const int S = a.sc->S;
    uint32_t* prof2 = reinterpret_cast<uint32_t*>(smem);  // [S][NV][64]: the two reads' scores of (residue, vector, lane)
    int8_t* wsh = reinterpret_cast<int8_t*>(prof2 + (size_t)S * NV * 64);  // [S][S] weights (i8, as in WeightMatrix<i8, S>)
    for (int i = lane; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = lane; i < S * S; i += 64) wsh[i] = (int8_t)a.sc->w[i];
    __syncthreads();
    zsw_pk::Consts<O, NV> c;
    c.ge = (uint32_t)a.sc->gap_extend;
    c.go2 = (uint32_t)a.sc->gap_open * zsw_pk::ONE2;
    c.ge2 = c.ge * zsw_pk::ONE2;
    c.nvge2 = (uint32_t)NV * c.ge * zsw_pk::ONE2;
    c.keep = li == 0 ? 0u : ~0u;
    asm volatile("v_mov_b32 %0, 0x10001" : "=v"(c.one));
    const long long warm = warmup_rows(a.sc->w, S, (int)c.ge, NV * N);
    const int maxw = max_weight(a.sc->w, S);
    const size_t row_bytes = (size_t)N * NVQ * 4;
    const int W = (int)a.W;
    const uint32_t* plane = prof2 + lane;

    // Reads are handed out through a counter: the rows a read costs vary with its r_end, and a CU holds a number of
    // wavefronts that need not be a multiple of its four SIMDs, so a fixed share per wavefront would leave SIMDs idle.
    for (;;) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(a.next_item, (uint32_t)RPW);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (first >= a.b.n_items) break;  // every wavefront gets here: the counter only grows
        // this lane group's two reads (halves 0 and 1 of every lane value)
        uint32_t id[2], len[2], item[2];
        uint64_t off[2];
        int rend[2], keep0[2], wrows[2];  // last row, first row whose flags the read keeps, rows kept
        long long start[2];               // row from which the read's state must be computed
        int32_t best[2];
        bool active[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            item[h] = first + (uint32_t)grp * 2 + (uint32_t)h;
            const bool valid = item[h] < a.b.n_items;
            id[h] = valid ? (a.b.items ? a.b.items[item[h]] : item[h]) : 0;
            off[h] = 0;
            len[h] = valid ? read_len(a.b, id[h], &off[h]) : 0;
            active[h] = valid && a.status[id[h]] == ZSW_STATUS_SOME && len[h] > 0 && (int)((len[h] + N - 1) / N) == NV;
            rend[h] = active[h] ? (int)a.ref_end[id[h]] - 1 : -1;
            best[h] = active[h] ? (int32_t)a.score[id[h]] : 0;
            wrows[h] = flag_rows_needed(W, (int)len[h], maxw, best[h], (int)a.sc->gap_open, (int)c.ge);
            keep0[h] = active[h] ? rend[h] - wrows[h] + 1 : 0x7fffffff;
            // late start: with the first pass's certificate (seed_safe_start) from that row, otherwise warmup_rows before the
            // first kept row
            const uint32_t safe = (active[h] && a.safe_row) ? a.safe_row[id[h]] : 0xffffffffu;
            start[h] = !active[h] ? 0x7fffffffll : safe != 0xffffffffu ? min((long long)keep0[h] - 1, (long long)safe) : (long long)keep0[h] - 1 - warm;
        }
        // StripedProfile::new_unchecked (profile.rs:270-306): position q = v + lane*nv, padding scores 0 (the bias)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint32_t q = (uint32_t)v + (uint32_t)li * (uint32_t)NV;
            const int k0 = (active[0] && q < len[0]) ? (int)lut[a.b.bases[off[0] + q]] : -1;
            const int k1 = (active[1] && q < len[1]) ? (int)lut[a.b.bases[off[1] + q]] : -1;
            for (int ri = 0; ri < S; ++ri) {
                const uint32_t w0 = k0 < 0 ? 0u : (uint32_t)(uint16_t)(int16_t)wsh[ri * S + k0];
                const uint32_t w1 = k1 < 0 ? 0u : (uint32_t)(uint16_t)(int16_t)wsh[ri * S + k1];
                prof2[((size_t)ri * NV + v) * 64 + lane] = w0 | (w1 << 16);
            }
        }
        zsw_pk::State<O, NV> st;
#pragma unroll
        for (int v = 0; v < NV; ++v) st.H[v] = st.E[v] = 0;
        int cend[2] = {0x7fffffff, 0x7fffffff};
        int rmax_v = max(rend[0], rend[1]);
        int rmin_v = min(keep0[0], keep0[1]);
        int rstart_v = (int)max(-1ll, min(min(start[0], start[1]), 0x7fffffffll));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            rmax_v = max(rmax_v, __shfl_xor(rmax_v, d, 64));
            rmin_v = min(rmin_v, __shfl_xor(rmin_v, d, 64));
            rstart_v = min(rstart_v, __shfl_xor(rstart_v, d, 64));
        }
        const int rmax = __builtin_amdgcn_readfirstlane(rmax_v);
        const int rmin = __builtin_amdgcn_readfirstlane(rmin_v);
        const int rflag = max(0, rmin);  // first row whose flags some read of the wave keeps (rmin: the smallest keep0)
        const int r0 = max(0, min(__builtin_amdgcn_readfirstlane(rstart_v), rmax));  // late start: the earliest row any read of the wave needs
        uint8_t* ring0 = a.ring + ((size_t)blockIdx.x * RPW + (size_t)grp * 2) * (size_t)W * row_bytes;
        uint8_t* ring1 = ring0 + (size_t)W * row_bytes;

        uint32_t p[NV];  // profile dwords of the row about to run

        auto do_row = [&](const int r, const uint32_t next_off, auto flags_tag) __attribute__((always_inline)) {
            constexpr bool FLAGS = decltype(flags_tag)::value;
            uint32_t pc[NV], flg[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) pc[v] = p[v];
            if constexpr (PREFETCH) {  // the next row's profile dwords travel while this row computes
                const uint32_t* nrow = plane + next_off;
#pragma unroll
                for (int v = 0; v < NV; ++v) p[v] = nrow[v * 64];
            }
            const uint32_t act2 = (r <= rend[0] ? 0xffffu : 0u) | (r <= rend[1] ? 0xffff0000u : 0u);
            zsw_pk::row<O, N, NV, FLAGS>(st, pc, act2, c, flg, [&]() __attribute__((always_inline)) {
                if constexpr (!PREFETCH) {  // more than 16 vectors: no second register set, the row's scores are loaded once they are dead
                    const uint32_t* nrow = plane + next_off;
#pragma unroll
                    for (int v = 0; v < NV; ++v) p[v] = nrow[v * 64];
                }
            });
            if constexpr (FLAGS) {
                // one byte per cell and read: bytes 4*vq .. 4*vq+3 of the lane's row slice
                uint32_t d0[NVQ], d1[NVQ], fp[NVQ * 4];
#pragma unroll
                for (int v = 0; v < NVQ * 4; ++v) fp[v] = v < NV ? flg[v < NV ? v : 0] : 0u;
#pragma unroll
                for (int vq = 0; vq < NVQ; ++vq) {
                    const uint32_t x01 = (fp[4 * vq + 1] << 8) | fp[4 * vq], x23 = (fp[4 * vq + 3] << 8) | fp[4 * vq + 2];
                    d0[vq] = __builtin_amdgcn_perm(x23, x01, 0x05040100u);  // read 0: the low halves
                    d1[vq] = __builtin_amdgcn_perm(x23, x01, 0x07060302u);  // read 1: the high halves
                }
                if (r <= rend[0] && r >= keep0[0]) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(ring0 + (size_t)(r % W) * row_bytes) + (size_t)li * NVQ;
#pragma unroll
                    for (int vq = 0; vq < NVQ; ++vq) dst[vq] = d0[vq];
                }
                if (r <= rend[1] && r >= keep0[1]) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(ring1 + (size_t)(r % W) * row_bytes) + (size_t)li * NVQ;
#pragma unroll
                    for (int vq = 0; vq < NVQ; ++vq) dst[vq] = d1[vq];
                }
                // c_end at a read's last row: first query position whose H equals the best score (striped.rs:571-583)
                if (__ballot(r == rend[0] || r == rend[1]) != 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (r == rend[h]) {
#pragma unroll
                            for (int v = NV - 1; v >= 0; --v) {
                                const uint32_t ci = (uint32_t)v + (uint32_t)li * (uint32_t)NV;
                                const int32_t hv = (int32_t)((st.H[v] >> (16 * h)) & 0xffffu);
                                if (ci < len[h] && hv == best[h]) cend[h] = (int)ci;
                            }
                        }
                }
            }
        };

#pragma unroll 1
        for (int rb = r0; rb <= rmax; rb += 64) {
            const int rr = rb + lane;
            const uint32_t offs = rr <= rmax ? (uint32_t)lut[a.ref[rr]] * (uint32_t)(NV * 64) : 0u;
            const int n = min(64, rmax - rb + 1);
            {
                const uint32_t* prow = plane + (uint32_t)__builtin_amdgcn_readlane((int)offs, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) p[v] = prow[v * 64];
            }
#pragma unroll 1
            for (int k = 0; k < n; ++k) {
                const int r = rb + k;
                const uint32_t noff = (uint32_t)__builtin_amdgcn_readlane((int)offs, min(k + 1, 63));
                if (r >= rflag) do_row(r, noff, std::true_type{});
                else do_row(r, noff, std::false_type{});
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 1; d < N; d <<= 1) cend[h] = min(cend[h], __shfl_xor(cend[h], d, N));
        __threadfence_block();  // this wave's ring stores are visible to its own traceback loads

        // lane 0 of the group walks read 0, lane 1 read 1
        // ---- traceback (backtrack.rs:290-342), one walker lane per read, helped by the other lanes of its group ----
        // Lanes 0 and 1 of a group walk reads 0 and 1. A walk is a chain of dependent loads from the flag ring (HBM / L2: most
        // of a microsecond each), and nearly all of its steps are diagonal, so the group's lanes fetch the walker's current
        // cell and the N/2 - 1 cells up-left of it in ONE round trip (lane 2k+h loads the cell k steps up the diagonal of
        // read h); the walker consumes them while its steps stay on that diagonal and asks for a new batch after a gap step
        // or after N/2 steps. Same decisions, same order, same ciglets as traceback_emit (zsw_align_dev.hpp).
        {
            constexpr int KB = N / 2;                 // cells per batch and read
            const int h = li & 1, kk = li >> 1;       // this lane helps read h with the cell kk steps up the diagonal
            const int grp0 = lane - li;               // first lane of the group
            const bool walker = li < 2 && active[h];  // (li < 2: h == li)
            const uint8_t* ringh = h ? ring1 : ring0;
            const uint32_t idw = id[h], lenw = len[h];
            const int rendw = rend[h], window = wrows[h];
            uint32_t* cig = a.cig + a.pool_base + (uint64_t)(a.by_item ? item[h] : idw) * a.maxc;
            uint32_t ncig = 0, cur_op = 0, cur_inc = 0, n_nons = 0, op = 0;
            bool overflow = walker && cend[h] == 0x7fffffff;
            auto push = [&](uint32_t inc, uint32_t o) {
                if (inc == 0) return;
                if (cur_inc && cur_op == o) {
                    cur_inc += inc;
                    return;
                }
                if (cur_inc) {
                    if (ncig < a.maxc) cig[ncig] = (cur_inc << 8) | cur_op;
                    else overflow = true;
                    ++ncig;
                }
                cur_op = o;
                cur_inc = inc;
            };
            int r = rendw + 1, c = (walker && !overflow) ? cend[h] + 1 : 0;
            bool walking = walker && !overflow;  // inside the while loop of to_alignment
            if (walking) push(lenw - (uint32_t)c, 'S');
            int cr = rendw, cc = c - 1;  // the cell the walker looks at next
            int idx = 0;                 // its position in the current batch
            bool need = true;            // a new batch is needed
            uint32_t pre = 0;            // this lane's cell of its read's current batch (kept while the walker still consumes it)
#pragma unroll 1
            while (__ballot(walking) != 0) {
                // anchors of the two walkers of the group, and one load per lane
                const int src = grp0 + h;
                const int ar = __shfl(cr, src, 64), ac = __shfl(cc, src, 64);
                const bool want = __shfl((int)(walking && need), src, 64) != 0;
                if (want && kk < KB && ar - kk >= 0 && ac - kk >= 0)
                    pre = __hip_atomic_load(ringh + (size_t)((ar - kk) % W) * row_bytes + (size_t)((ac - kk) / NV) * (size_t)NVQ * 4 + (size_t)((ac - kk) % NV),
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (walking && need) {
                    idx = 0;
                    need = false;
                }
                // consume
#pragma unroll 1
                for (int s = 0; s < KB; ++s) {
                    if (__ballot(walking && !need) == 0) break;  // every walker of the wave waits for a new batch (or is done)
                    const uint32_t fv = (uint32_t)__shfl((int)pre, grp0 + 2 * idx + h, 64);
                    if (walking && !need) {
                        const uint32_t f = fv;
                        if ((f & BT_STOP) || r <= 0 || c <= 0) {
                            walking = false;
                        } else {
                            bool diag = false;
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
                                diag = true;
                            }
                            if (!(cur_inc && cur_op == op)) ++n_nons;
                            push(1, op);
                            if (r > 0 && c > 0) {
                                if (r - 1 + window <= rendw) {  // the walk left the retained window
                                    overflow = true;
                                    walking = false;
                                } else {
                                    cr = r - 1;
                                    cc = c - 1;
                                    if (diag && idx + 1 < KB) ++idx;
                                    else need = true;
                                }
                            } else {
                                walking = false;
                            }
                        }
                    }
                }
            }
            if (walker) {
                if (!(cend[h] == 0x7fffffff)) {
                    push((uint32_t)c, 'S');
                    push(1, 0);  // flush the pending ciglet (the sentinel op 0 itself is never stored)
                }
                if (overflow || ncig > a.maxc) {
                    const uint32_t k = atomicAdd(a.fb_count, 1u);
                    a.fb_list[k] = idw;
                } else {
                    zsw_alignment out;
                    out.score = (uint32_t)best[h];
                    out.ref_start = (uint32_t)r;
                    out.ref_end = (uint32_t)(rendw + 1);
                    out.query_start = (uint32_t)c;
                    out.query_end = (uint32_t)(cend[h] + 1);
                    out.ref_len = a.ref_len;
                    out.query_len = lenw;
                    // forward count; inverted: clips re-derived from ref_range (output.rs:399-414)
                    out.n_ciglets = a.invert ? n_nons + (r > 0 ? 1u : 0u) + (a.ref_len > (uint32_t)(rendw + 1) ? 1u : 0u) : ncig;
                    out.ciglet_offset = 0;  // filled by write_ciglets_kernel
                    a.aln[idw] = out;
                    a.cig_start[idw] = (uint64_t)(uintptr_t)cig;
                    a.cig_raw[idw] = ncig;
                }
            }
        }
    }
}

// ---- host side of one lane count (ZSW_PK_N) ----
template <int N, int NV>
static hipError_t pk_blocks_per_cu(size_t lds, int* out) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&align_kernel_pk<N, NV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 9 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, align_kernel_pk<N, NV>, 64, lds);
}

template <int N, int NV>
static hipError_t pk_launch(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&align_kernel_pk<N, NV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 9 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((align_kernel_pk<N, NV>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

#define ZSW_PK_FOR_NV(X)                                                                                                  \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) \
    X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

template <int N>
static hipError_t pk_occupancy_n(uint32_t nv, size_t lds, int* out) {
    switch (nv) {
#define X(NV_) \
    case NV_: return pk_blocks_per_cu<N, NV_>(lds, out);
        ZSW_PK_FOR_NV(X)
#undef X
    }
    return hipErrorInvalidValue;
}

template <int N>
static hipError_t pk_launch_n(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream) {
    switch (a.nv) {
#define X(NV_) \
    case NV_: return pk_launch<N, NV_>(a, grid, lds, stream);
        ZSW_PK_FOR_NV(X)
#undef X
    }
    return hipErrorInvalidValue;
}

}  // namespace zsw
