// zsw_align.hip — full alignment: sw_simd_align (striped.rs:449-598) + traceback (backtrack.rs:290-342)
// + AlignmentStates (state.rs) + SeqSrc inversion (alignment/mod.rs:176-190, output.rs:396-425), batched.
//
// Zoe's traceback flags depend on the striped layout (which cells see F in the main pass and which only
// in the lazy-F pass, and where the vector-wide `any()` stops the lazy-F loop), so — unlike the score —
// they must be produced by the SAME <N lanes, nv = ceil(L/N) vectors> recurrence the CPU runs.  The kernel
// below therefore emulates one `Simd<T,N>` with N adjacent GPU lanes (64/N reads per wavefront):
//   shift_elements_right::<1>  -> __shfl_up within the N-lane group, lane 0 filled with T::MIN (true 0)
//   mask.any()                 -> wave ballot restricted to the group's N lanes (padding lanes included)
//   StripedProfile             -> per-wave LDS table prof[ref_idx][v][lane] (profile.rs:270-306)
//   load/store/e_scores        -> per-wave LDS rows H[v][lane], E[v][lane]
// Values are kept as true scores in i32 (offset T::MIN removed); every saturating_sub of the reference
// that floors at T::MIN is a max(0, .) here. Width (i8/i16/i32, signed/unsigned) only decides whether the
// read is Overflowed, which pass 1 (the packed score kernel) has already settled.
//
// Memory: the CPU materialises R*nv*N flag bytes per call. Here pass 1 supplies (score, ref_end); pass 2
// recomputes rows 0..=r_end and keeps only the last W rows of flags in a per-wave ring in HBM/L2, then one
// lane walks the traceback. A read whose walk leaves the window (or overflows the ciglet scratch) is
// re-run with a window covering every row.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "zsw_align_dev.hpp"
#include "zsw_shared.hpp"

namespace zsw {

// Generic form: any nv, DP rows in LDS — or, for profiles too long for one wavefront's LDS (nv above ~250), in a per-block
// region of HBM behind the flag ring (GLOBAL_ROWS).
// SHARED: the one-profile-many-sequences role (zsw_shared.hpp): the profile of every item is striped over a.prof_seq, the rows
// are the bases of read i, and every row's flags are kept (W >= the longest read).
template <int N, bool GLOBAL_ROWS = false, bool SHARED = false>
__global__ __launch_bounds__(64) void align_kernel(AlignArgs a) {
    extern __shared__ __align__(16) uint8_t smem_lds[];
    uint8_t* smem = GLOBAL_ROWS ? a.rows + (size_t)blockIdx.x * align_rows_bytes(a.nv) : smem_lds;
    __shared__ uint8_t lut[256];
    __shared__ int32_t wsh[MAX_S * MAX_S];
    __shared__ int32_t wpad[MAX_S * (MAX_S + 1)];  // weights with one more column: the padding residue S scores 0
    constexpr int RPW = 64 / N;
    const int lane = threadIdx.x;
    const int li = lane % N, grp = lane / N;
    const uint32_t nv = a.nv;
    const int S = a.sc->S;
    const int go = a.sc->gap_open, ge = a.sc->gap_extend;
    int32_t* Hs = reinterpret_cast<int32_t*>(smem);  // [nv][64]
    int32_t* Es = Hs + nv * 64;                      // [nv][64]
    // the striped profile as residue codes [nv][64] (S = padding, scoring 0); the score of a cell is a second LDS lookup in the
    // weight matrix, so the footprint does not grow with the alphabet and nv up to ~250 fits one wavefront's LDS
    uint8_t* kq = reinterpret_cast<uint8_t*>(Es + nv * 64);
    uint8_t* fl = kq + (size_t)nv * 64;  // [nv][64]
    for (int i = lane; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = lane; i < S * S; i += 64) wsh[i] = a.sc->w[i];
    for (int i = lane; i < S * (S + 1); i += 64) wpad[i] = (i % (S + 1)) < S ? a.sc->w[(i / (S + 1)) * S + i % (S + 1)] : 0;
    __syncthreads();
    const unsigned long long gmask = (N == 64) ? ~0ull : (((1ull << N) - 1ull) << (grp * N));
    const long long warm = warmup_rows(wsh, S, ge, (int)nv * N);
    const size_t row_bytes = (size_t)nv * N;
    uint8_t* ring = a.ring + ((size_t)blockIdx.x * RPW + grp) * (size_t)a.W * row_bytes;

    for (;;) {  // reads are handed out through a work counter (see align_kernel_pk)
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(a.next_item, (uint32_t)RPW);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (first >= a.b.n_items) break;
        const uint32_t item = first + grp;
        const bool valid = item < a.b.n_items;
        const uint32_t id = valid ? (a.b.items ? a.b.items[item] : item) : 0;
        uint64_t off = 0;
        const uint32_t rlen = valid ? read_len(a.b, id, &off) : 0;  // the read
        const uint32_t len = SHARED ? a.prof_len : rlen;            // the sequence the profile is built from
        const bool active = valid && a.status[id] == ZSW_STATUS_SOME && len > 0 && rlen > 0 && (len + N - 1) / N == nv;
        const int rend = active ? (int)a.ref_end[id] - 1 : -1;
        const int32_t best = active ? (int32_t)a.score[id] : 0;
        const uint8_t* pbase = SHARED ? a.prof_seq : a.b.bases + off;

        // StripedProfile::new_unchecked (profile.rs:270-306): position q = v + lane*nv; padding scores bias (= true 0)
        for (uint32_t v = 0; v < nv; ++v) {
            const uint32_t q = v + (uint32_t)li * nv;
            kq[v * 64 + lane] = (active && q < len) ? lut[pbase[q]] : (uint8_t)S;
            Hs[v * 64 + lane] = 0;
            Es[v * 64 + lane] = 0;
        }
        int rmax = rend;
        int rmin = active ? rend : 0x7fffffff;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            rmax = max(rmax, __shfl_xor(rmax, d, 64));
            rmin = min(rmin, __shfl_xor(rmin, d, 64));
        }
        const int r0 = max(0, (int)min((long long)rmin - (long long)a.W - warm, (long long)rmax));

        for (int r = r0; r <= rmax; ++r) {
            const bool act = r <= rend;
            const int ri = lut[SHARED ? (act ? a.b.bases[off + (uint32_t)r] : (uint8_t)0) : a.ref[r]];
            const int32_t* wrow = wpad + ri * (S + 1);
            // main pass (striped.rs:481-526)
            int32_t F = 0;
            int32_t H = __shfl_up(Hs[(nv - 1) * 64 + lane], 1, N);
            if (li == 0) H = 0;
            for (uint32_t v = 0; v < nv; ++v) {
                int32_t E = Es[v * 64 + lane];
                const int32_t hold = Hs[v * 64 + lane];
                H = max(H + wrow[kq[v * 64 + lane]], 0);
                H = max(H, max(E, F));
                uint32_t flags = (E == H ? BT_UP : 0) | (F == H ? BT_LEFT : 0);
                const bool stopped = H == 0;
                const int32_t Hn = H;
                H = max(H - go, 0);
                E = max(max(E - ge, 0), H);
                F = max(max(F - ge, 0), H);
                flags |= (E > H ? BT_UP_EXT : 0) | (F > H ? BT_LEFT_EXT : 0);
                if (stopped) flags = BT_STOP;
                if (act) {
                    Hs[v * 64 + lane] = Hn;
                    Es[v * 64 + lane] = E;
                    fl[v * 64 + lane] = (uint8_t)flags;
                }
                H = hold;
            }
            // lazy-F pass (striped.rs:528-553): per read, up to N rounds, stops at the first vector where no lane
            // of THAT read has F > H - gap_open
            bool done = !act;
            for (int it = 0; it < N; ++it) {
                F = __shfl_up(F, 1, N);
                if (li == 0) F = 0;
                bool all_done = false;
                for (uint32_t v = 0; v < nv; ++v) {
                    H = Hs[v * 64 + lane];
                    const bool cond = !done && F > max(H - go, 0);
                    const unsigned long long bal = __ballot(cond);
                    if ((bal & gmask) == 0) done = true;
                    if (bal == 0) {  // every read of the wave has left its lazy-F loop
                        all_done = true;
                        break;
                    }
                    if (!done) {
                        H = max(H, F);
                        uint32_t flags = fl[v * 64 + lane];
                        const bool stopped = H == 0;
                        if (F == H) flags = (flags & BT_UP_EXT) | BT_LEFT;  // simd_correct_and_set_left
                        Hs[v * 64 + lane] = H;
                        H = max(H - go, 0);
                        F = max(F - ge, 0);
                        if (F > H) flags |= BT_LEFT_EXT;
                        if (stopped) flags = BT_STOP;
                        fl[v * 64 + lane] = (uint8_t)flags;
                    }
                }
                if (all_done) break;
            }
            // keep the last W rows of flags of every read still running
            if (act && r + (int)a.W > rend) {
                uint8_t* dst = ring + (size_t)(r % (int)a.W) * row_bytes;
                for (uint32_t v = 0; v < nv; ++v) dst[(size_t)v * N + li] = fl[v * 64 + lane];
            }
        }

        // c_end: first query position of row r_end whose H equals the best score (striped.rs:571-583)
        int cend = 0x7fffffff;
        if (active) {
            for (int v = (int)nv - 1; v >= 0; --v) {
                const uint32_t ci = (uint32_t)v + (uint32_t)li * nv;
                if (ci < len && Hs[v * 64 + lane] == best) cend = (int)ci;
            }
        }
#pragma unroll
        for (int d = 1; d < N; d <<= 1) cend = min(cend, __shfl_xor(cend, d, N));
        __threadfence_block();  // this wave's ring stores are visible to its own traceback loads

        if (active && li == 0) {
            auto cell = [&](int rr, int cc) -> uint32_t {
                // agent-scope load: served by L2, never by a stale L1 line of an earlier item's window
                return __hip_atomic_load(ring + (size_t)(rr % (int)a.W) * row_bytes + (size_t)(cc % (int)nv) * N + (size_t)(cc / (int)nv),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            traceback_emit(a, id, item, len, rend, cend, best, cell, (int)a.W, SHARED ? rlen : a.ref_len);
        }
    }
}

// The one-profile-many-sequences role when every score of the launch fits 16 bits (the i8 / i16 instantiations) and the rows of the
// shared sequence's profile fit LDS with three wavefronts per CU: H and E of a vector in ONE dword per lane (two 16-bit halves), the
// profile's residue codes once per block ([nv][N] bytes: every read of the launch meets the same profile), the row's flag bytes; the
// next vector's dword and code are fetched before the current vector's are stored. Same recurrence, same lazy-F loop, same flags
// as align_kernel<N, ., true> — which kept these rows in L2 (80 KB per wavefront at nv = 125 left LDS with two wavefronts per CU)
// and ran at 29 cycles per instruction behind its own stores.
// a lane's flag bytes of one row: nv rounded up to whole dwords, an odd number of them (lanes one stride apart never share a bank)
__host__ __device__ inline uint32_t align_shared_row_stride(uint32_t nv) { return (((nv + 3) / 4) | 1u) * 4; }
__host__ __device__ inline size_t align_shared_lds_bytes(uint32_t nv, int N) {
    return (size_t)nv * 64 * 4 + (size_t)64 * align_shared_row_stride(nv) + (((size_t)nv * N + 15) & ~(size_t)15);
}
constexpr size_t ALIGN_SHARED_LDS_LIMIT = 52 * 1024;  // three blocks per CU

template <int N>
__global__ __launch_bounds__(64) void align_shared_kernel_h(AlignArgs a) {
    extern __shared__ __align__(16) uint8_t smem_lds[];
    __shared__ uint8_t lut[256];
    __shared__ int32_t wpad[MAX_S * (MAX_S + 1)];  // weights with one more column: the padding residue S scores 0
    constexpr int RPW = 64 / N;
    const int lane = threadIdx.x;
    const int li = lane % N, grp = lane / N;
    const uint32_t nv = a.nv;
    const uint32_t nvp = align_shared_row_stride(nv);  // a lane's flag bytes of one row: nv, padded so that 64 lanes fall into 32 banks twice
    const int S = a.sc->S;
    const int go = a.sc->gap_open, ge = a.sc->gap_extend;
    uint32_t* HE = reinterpret_cast<uint32_t*>(smem_lds);     // [nv][64]: H in the low half, E in the high half
    uint8_t* fl = smem_lds + (size_t)nv * 64 * 4;             // [64][nvp]: the row's flags, a lane's vectors side by side
    uint8_t* kq = fl + (size_t)64 * nvp;                      // [nv][N]: residue code of position v + lane * nv (S = padding)
    for (int i = lane; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = lane; i < S * (S + 1); i += 64) wpad[i] = (i % (S + 1)) < S ? a.sc->w[(i / (S + 1)) * S + i % (S + 1)] : 0;
    __syncthreads();
    const uint32_t plen = a.prof_len;
    for (uint32_t i = lane; i < nv * N; i += 64) {  // StripedProfile::new_unchecked (profile.rs:270-306), once per block
        const uint32_t v = i / N, l = i % N, q = v + l * nv;
        kq[i] = q < plen ? lut[a.prof_seq[q]] : (uint8_t)S;
    }
    for (uint32_t i = lane; i < 64 * nvp / 4; i += 64) reinterpret_cast<uint32_t*>(fl)[i] = 0;  // (the padding bytes travel to the ring)
    __syncthreads();
    const unsigned long long gmask = (N == 64) ? ~0ull : (((1ull << N) - 1ull) << (grp * N));
    // the ring: per read W rows of [N lanes][nvp] flag bytes — a lane's share of a row goes out as whole dwords
    const size_t row_bytes = (size_t)N * nvp;
    uint8_t* ring = a.ring + ((size_t)blockIdx.x * RPW + grp) * (size_t)a.W * row_bytes;
    uint8_t* const myfl = fl + (size_t)lane * nvp;

    for (;;) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(a.next_item, (uint32_t)RPW);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (first >= a.b.n_items) break;
        const uint32_t item = first + grp;
        const bool valid = item < a.b.n_items;
        const uint32_t id = valid ? (a.b.items ? a.b.items[item] : item) : 0;
        uint64_t off = 0;
        const uint32_t rlen = valid ? read_len(a.b, id, &off) : 0;
        const bool active = valid && a.status[id] == ZSW_STATUS_SOME && plen > 0 && rlen > 0;
        const int rend = active ? (int)a.ref_end[id] - 1 : -1;
        const int32_t best = active ? (int32_t)a.score[id] : 0;
        for (uint32_t v = 0; v < nv; ++v) HE[v * 64 + lane] = 0;
        int rmax = rend;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) rmax = max(rmax, __shfl_xor(rmax, d, 64));
        uint8_t base_next = (active && rend >= 0) ? a.b.bases[off] : (uint8_t)0;

        for (int r = 0; r <= rmax; ++r) {  // every row's flags are kept (W >= the longest read): the rows start at the read's first base
            const bool act = r <= rend;
            const int ri = lut[act ? base_next : (uint8_t)0];
            if (r + 1 <= rend) base_next = a.b.bases[off + (uint32_t)r + 1];  // (in flight under this row)
            const int32_t* wrow = wpad + ri * (S + 1);
            // main pass (striped.rs:481-526). The loads of a vector step are issued one step ahead: the dword of H and E and the
            // weight of vector v + 1 (whose residue code came in a step earlier still), the code of vector v + 2.
            int32_t F = 0;
            int32_t H = __shfl_up((int32_t)(HE[(nv - 1) * 64 + lane] & 0xffffu), 1, N);
            if (li == 0) H = 0;
            uint32_t he_cur = HE[lane];
            int32_t w_cur = wrow[kq[li]];
            uint32_t k_next = nv > 1 ? kq[N + li] : 0u;
            for (uint32_t v = 0; v < nv; ++v) {
                const uint32_t he = he_cur;
                const int32_t w = w_cur;
                if (v + 1 < nv) {
                    he_cur = HE[(v + 1) * 64 + lane];
                    w_cur = wrow[k_next];
                    if (v + 2 < nv) k_next = kq[(v + 2) * N + li];
                }
                int32_t E = (int32_t)(he >> 16);
                const int32_t hold = (int32_t)(he & 0xffffu);
                H = max(H + w, 0);
                H = max(H, max(E, F));
                uint32_t flags = (E == H ? BT_UP : 0) | (F == H ? BT_LEFT : 0);
                const bool stopped = H == 0;
                const int32_t Hn = H;
                H = max(H - go, 0);
                E = max(max(E - ge, 0), H);
                F = max(max(F - ge, 0), H);
                flags |= (E > H ? BT_UP_EXT : 0) | (F > H ? BT_LEFT_EXT : 0);
                if (stopped) flags = BT_STOP;
                if (act) {
                    HE[v * 64 + lane] = (uint32_t)Hn | ((uint32_t)E << 16);
                    myfl[v] = (uint8_t)flags;
                }
                H = hold;
            }
            // lazy-F pass (striped.rs:528-553): per read, up to N rounds, stops at the first vector where no lane of THAT read has
            // F > H - gap_open
            bool done = !act;
            for (int it = 0; it < N; ++it) {
                F = __shfl_up(F, 1, N);
                if (li == 0) F = 0;
                bool all_done = false;
                for (uint32_t v = 0; v < nv; ++v) {
                    const uint32_t he = HE[v * 64 + lane];
                    H = (int32_t)(he & 0xffffu);
                    const bool cond = !done && F > max(H - go, 0);
                    const unsigned long long bal = __ballot(cond);
                    if ((bal & gmask) == 0) done = true;
                    if (bal == 0) {  // every read of the wave has left its lazy-F loop
                        all_done = true;
                        break;
                    }
                    if (!done) {
                        H = max(H, F);
                        uint32_t flags = myfl[v];
                        const bool stopped = H == 0;
                        if (F == H) flags = (flags & BT_UP_EXT) | BT_LEFT;  // simd_correct_and_set_left
                        HE[v * 64 + lane] = (he & 0xffff0000u) | (uint32_t)H;
                        H = max(H - go, 0);
                        F = max(F - ge, 0);
                        if (F > H) flags |= BT_LEFT_EXT;
                        if (stopped) flags = BT_STOP;
                        myfl[v] = (uint8_t)flags;
                    }
                }
                if (all_done) break;
            }
            if (act) {  // the row's flags: this lane's nvp bytes as dwords
                uint32_t* dst = reinterpret_cast<uint32_t*>(ring + (size_t)(r % (int)a.W) * row_bytes + (size_t)li * nvp);
                const uint32_t* src = reinterpret_cast<const uint32_t*>(myfl);
                for (uint32_t q = 0; q < nvp / 4; ++q) dst[q] = src[q];
            }
        }

        // c_end: first position of the profile sequence in row r_end whose H equals the best score (striped.rs:571-583)
        int cend = 0x7fffffff;
        if (active) {
            for (int v = (int)nv - 1; v >= 0; --v) {
                const uint32_t ci = (uint32_t)v + (uint32_t)li * nv;
                if (ci < plen && (int32_t)(HE[v * 64 + lane] & 0xffffu) == best) cend = (int)ci;
            }
        }
#pragma unroll
        for (int d = 1; d < N; d <<= 1) cend = min(cend, __shfl_xor(cend, d, N));
        __threadfence_block();  // this wave's ring stores are visible to its own traceback loads

        if (active && li == 0) {
            auto cell = [&](int rr, int cc) -> uint32_t {
                return __hip_atomic_load(ring + (size_t)(rr % (int)a.W) * row_bytes + (size_t)(cc / (int)nv) * nvp + (size_t)(cc % (int)nv),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            traceback_emit(a, id, item, plen, rend, cend, best, cell, (int)a.W, rlen);
        }
    }
}

// shift_elements_right::<1>(T::MIN) inside groups of N lanes as one DPP move: row_shr:1 shifts within rows of 16 lanes and
// wave_shr:1 across the whole wavefront; lanes without a source keep `old` = 0, group-leading lanes are zeroed.
template <int N>
__device__ __forceinline__ int32_t shr1(int32_t x, bool li0) {
    if constexpr (N <= 16) {
        const int32_t y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
        return (N < 16 && li0) ? 0 : y;
    } else {
        const int32_t y = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false);
        return (N < 64 && li0) ? 0 : y;
    }
}

// Register-resident form specialised on NVQ = ceil(nv/4) (nv <= 32): every vector step except the last group's is
// compiled without a bounds test, the last group runs `last` = nv - 4*(NVQ-1) steps behind wave-uniform branches.
// H, E and the row's flags live in VGPRs at static indices; the profile is in LDS packed four scores per dword and the
// next row's dwords are fetched while the current row computes; the reference bytes of 64 rows are fetched by one load
// (lane k holds the profile offset of row rb+k, v_readlane hands it to the row). Rows before the first retained flag
// row only carry the DP state (FLAGS = false).
// Ring row layout: [lane in group][NVQ*4 bytes]; cell (r, c) is byte (c % nv) of lane c / nv.
template <int N, int NVQ>
__global__ __launch_bounds__(64) void align_kernel_x(AlignArgs a) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint8_t lut[256];
    __shared__ int32_t wsh[MAX_S * MAX_S];
    constexpr int RPW = 64 / N;
    constexpr int NV4 = 4 * NVQ;
    const int lane = threadIdx.x;
    const int li = lane % N, grp = lane / N;
    const bool li0 = li == 0;
    const int nv = (int)a.nv;
    const int last = nv - 4 * (NVQ - 1);  // 1..4 steps in the last group
    const int S = a.sc->S;
    const int go = a.sc->gap_open, ge = a.sc->gap_extend;
    uint32_t* prof4 = reinterpret_cast<uint32_t*>(smem);  // [S][NVQ][64]
    for (int i = lane; i < 256; i += 64) lut[i] = a.sc->index_map[i];
    for (int i = lane; i < S * S; i += 64) wsh[i] = a.sc->w[i];
    __syncthreads();
    // the ballot bits of this lane's group, split over the two halves of the wavefront
    uint32_t gsel_lo, gsel_hi;
    if constexpr (N == 64) {
        gsel_lo = gsel_hi = ~0u;
    } else if constexpr (N == 32) {
        gsel_lo = grp == 0 ? ~0u : 0u;
        gsel_hi = grp == 1 ? ~0u : 0u;
    } else {
        const uint32_t m = ((1u << N) - 1u) << ((grp * N) & 31);
        gsel_lo = grp * N < 32 ? m : 0u;
        gsel_hi = grp * N >= 32 ? m : 0u;
    }
    const long long warm = warmup_rows(wsh, S, ge, nv * N);
    const size_t row_bytes = (size_t)N * NVQ * 4;
    uint8_t* ring = a.ring + ((size_t)blockIdx.x * RPW + grp) * (size_t)a.W * row_bytes;
    const int W = (int)a.W;
    const uint32_t* plane = prof4 + lane;

    for (;;) {  // reads are handed out through a work counter (see align_kernel_pk)
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(a.next_item, (uint32_t)RPW);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        if (first >= a.b.n_items) break;
        const uint32_t item = first + grp;
        const bool valid = item < a.b.n_items;
        const uint32_t id = valid ? (a.b.items ? a.b.items[item] : item) : 0;
        uint64_t off = 0;
        const uint32_t len = valid ? read_len(a.b, id, &off) : 0;
        const bool active = valid && a.status[id] == ZSW_STATUS_SOME && len > 0 && (int)((len + N - 1) / N) == nv;
        const int rend = active ? (int)a.ref_end[id] - 1 : -1;
        const int32_t best = active ? (int32_t)a.score[id] : 0;

        // StripedProfile::new_unchecked (profile.rs:270-306), four consecutive vectors per dword
#pragma unroll
        for (int vq = 0; vq < NVQ; ++vq) {
            int k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = 4 * vq + j;
                const uint32_t q = (uint32_t)v + (uint32_t)li * (uint32_t)nv;
                k[j] = (active && v < nv && q < len) ? (int)lut[a.b.bases[off + q]] : -1;
            }
            for (int ri = 0; ri < S; ++ri) {
                uint32_t p = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) p |= (uint32_t)(uint8_t)(k[j] < 0 ? 0 : wsh[ri * S + k[j]]) << (8 * j);
                prof4[((size_t)ri * NVQ + vq) * 64 + lane] = p;
            }
        }
        int32_t H[NV4], E[NV4];
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            H[v] = 0;
            E[v] = 0;
        }
        int32_t Hlast = 0;  // previous row's H of vector nv-1
        int cend = 0x7fffffff;
        int rmax_v = rend, rmin_v = active ? rend : 0x7fffffff;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            rmax_v = max(rmax_v, __shfl_xor(rmax_v, d, 64));
            rmin_v = min(rmin_v, __shfl_xor(rmin_v, d, 64));
        }
        const int rmax = __builtin_amdgcn_readfirstlane(rmax_v);
        const int rmin = __builtin_amdgcn_readfirstlane(rmin_v);
        const int rflag = (int)max(0ll, (long long)rmin - W + 1);  // first row whose flags some read of the wave keeps
        const int r0 = (int)max(0ll, min((long long)rmin - (long long)W - warm, (long long)rmax));

        uint32_t pn[NVQ];  // profile dwords of the row about to run

        auto do_row = [&](const int r, const uint32_t next_off, auto flags_tag) __attribute__((always_inline)) {
            constexpr bool FLAGS = decltype(flags_tag)::value;
            const bool act = r <= rend;
            uint32_t p[NVQ];
#pragma unroll
            for (int vq = 0; vq < NVQ; ++vq) p[vq] = pn[vq];
            {
                const uint32_t* nrow = plane + next_off;
#pragma unroll
                for (int vq = 0; vq < NVQ; ++vq) pn[vq] = nrow[vq * 64];
            }
            // main pass (striped.rs:481-526)
            int32_t F = 0;
            int32_t Hd = shr1<N>(Hlast, li0);
            uint32_t flg[FLAGS ? NV4 : 1];
            if constexpr (FLAGS) {
#pragma unroll
                for (int v = NV4 - 3; v < NV4; ++v) flg[FLAGS ? v : 0] = 0;
            }
            auto step = [&](const int v, const uint32_t p4) __attribute__((always_inline)) {
                const int32_t s = __builtin_amdgcn_sbfe((int)p4, 8 * (v & 3), 8);
                const int32_t Eo = E[v];
                const int32_t hold = H[v];
                const int32_t h = max(Hd + s, max(Eo, F));  // Eo >= 0 supplies the floor at T::MIN
                const int32_t hg = subsat(h, go);
                const int32_t En = max(subsat(Eo, ge), hg);
                const int32_t Fn = max(subsat(F, ge), hg);
                if constexpr (FLAGS) {
                    uint32_t fl = (Eo == h ? (uint32_t)BT_UP : 0u) | (F == h ? (uint32_t)BT_LEFT : 0u) |
                                  (En > hg ? (uint32_t)BT_UP_EXT : 0u) | (Fn > hg ? (uint32_t)BT_LEFT_EXT : 0u);
                    if (h == 0) fl = BT_STOP;
                    flg[FLAGS ? v : 0] = fl;
                }
                H[v] = h;
                E[v] = En;
                F = Fn;
                Hd = hold;
                if (v >= NV4 - 4) Hlast = h;  // the last executed step of the last group wins
            };
#pragma unroll
            for (int vq = 0; vq < NVQ - 1; ++vq) {
#pragma unroll
                for (int j = 0; j < 4; ++j) step(4 * vq + j, p[vq]);
            }
            step(NV4 - 4, p[NVQ - 1]);
            if (last > 1) {
                step(NV4 - 3, p[NVQ - 1]);
                if (last > 2) {
                    step(NV4 - 2, p[NVQ - 1]);
                    if (last > 3) step(NV4 - 1, p[NVQ - 1]);
                }
            }
            // lazy-F pass (striped.rs:528-553). A read that has left its loop (or is past its last row) carries F = 0:
            // with F = 0 a step changes neither H nor the flags (H = max(H,0); F == H only where H == 0, which stays STOP),
            // so finished reads need no predication while other reads of the wave keep iterating.
            if (!act) F = 0;
            auto lazy_step = [&](const int v) __attribute__((always_inline)) -> bool {
                const int32_t h0 = H[v];
                const unsigned long long bal = __ballot(F > subsat(h0, go));
                if (bal == 0) return true;  // every read of the wave has left its lazy-F loop
                if constexpr (FLAGS) {
                    // this read breaks out here. (Rows without flags skip the test: H is already exact for a read that
                    // has left its loop, so further steps with its stale F cannot change H; only flags could differ.)
                    if ((((uint32_t)bal & gsel_lo) | ((uint32_t)(bal >> 32) & gsel_hi)) == 0) F = 0;
                }
                const int32_t h = max(h0, F);
                const int32_t Fn = subsat(F, ge);
                if constexpr (FLAGS) {
                    uint32_t fl = flg[FLAGS ? v : 0];
                    if (F == h) fl = (fl & BT_UP_EXT) | BT_LEFT;  // simd_correct_and_set_left
                    if (Fn > subsat(h, go)) fl |= BT_LEFT_EXT;
                    if (h == 0) fl = BT_STOP;
                    flg[FLAGS ? v : 0] = fl;
                }
                F = Fn;
                H[v] = h;
                if (v >= NV4 - 4) Hlast = (v - (NV4 - 4) == last - 1) ? h : Hlast;
                return false;
            };
            auto lazy_round = [&]() __attribute__((always_inline)) -> bool {
#pragma unroll
                for (int v = 0; v < NV4 - 3; ++v)
                    if (lazy_step(v)) return true;
                if (last > 1) {
                    if (lazy_step(NV4 - 3)) return true;
                    if (last > 2) {
                        if (lazy_step(NV4 - 2)) return true;
                        if (last > 3) {
                            if (lazy_step(NV4 - 1)) return true;
                        }
                    }
                }
                return false;
            };
#pragma unroll 1
            for (int it = 0; it < N; ++it) {
                F = shr1<N>(F, li0);
                if (lazy_round()) break;
            }
            if constexpr (FLAGS) {
                // keep the last W rows of flags of every read still running
                if (act && r + W > rend) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(ring + (size_t)(r % W) * row_bytes) + (size_t)li * NVQ;
#pragma unroll
                    for (int vq = 0; vq < NVQ; ++vq)
                        dst[vq] = flg[FLAGS ? 4 * vq : 0] | (flg[FLAGS ? 4 * vq + 1 : 0] << 8) | (flg[FLAGS ? 4 * vq + 2 : 0] << 16) |
                                  (flg[FLAGS ? 4 * vq + 3 : 0] << 24);
                }
                // c_end at the read's last row: first query position whose H equals the best score (striped.rs:571-583)
                if (__ballot(r == rend) != 0) {
                    if (r == rend) {
#pragma unroll
                        for (int v = NV4 - 1; v >= 0; --v) {
                            if (v < nv) {
                                const uint32_t ci = (uint32_t)v + (uint32_t)li * (uint32_t)nv;
                                if (ci < len && H[v] == best) cend = (int)ci;
                            }
                        }
                    }
                }
            }
        };

#pragma unroll 1
        for (int rb = r0; rb <= rmax; rb += 64) {
            const int rr = rb + lane;
            const uint32_t offs = rr <= rmax ? (uint32_t)lut[a.ref[rr]] * (uint32_t)(NVQ * 64) : 0u;
            const int n = min(64, rmax - rb + 1);
            {
                const uint32_t* row = plane + (uint32_t)__builtin_amdgcn_readlane((int)offs, 0);
#pragma unroll
                for (int vq = 0; vq < NVQ; ++vq) pn[vq] = row[vq * 64];
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
        for (int d = 1; d < N; d <<= 1) cend = min(cend, __shfl_xor(cend, d, N));
        __threadfence_block();  // this wave's ring stores are visible to its own traceback loads

        if (active && li == 0) {
            auto cell = [&](int rr, int cc) -> uint32_t {
                return __hip_atomic_load(ring + (size_t)(rr % W) * row_bytes + (size_t)(cc / nv) * (size_t)NVQ * 4 + (size_t)(cc % nv),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            traceback_emit(a, id, item, len, rend, cend, best, cell, (int)a.W, a.ref_len);
        }
    }
}

// ---- exclusive scan of n_ciglets and the final packed write -------------------------------------
constexpr int SCAN_BLOCK = 1024;

__global__ void count_block_kernel(const zsw_alignment* aln, const uint8_t* status, uint32_t n, uint64_t* block_sums) {
    __shared__ uint64_t sh[SCAN_BLOCK / 64];
    const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint64_t v = (i < n && status[i] == ZSW_STATUS_SOME) ? aln[i].n_ciglets : 0;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor((long long)v, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x / 64] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int k = 0; k < SCAN_BLOCK / 64; ++k) t += sh[k];
        block_sums[blockIdx.x] = t;
    }
}

__global__ void scan_sums_kernel(uint64_t* block_sums, uint32_t nblocks, uint64_t* total) {
    // one wavefront: 64 entries per step, an inclusive scan across the lanes, the running total carried on (10 M reads: 9,766 entries)
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    uint64_t run = 0;
    for (uint32_t k0 = 0; k0 < nblocks; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        const uint64_t t = k < nblocks ? block_sums[k] : 0;
        uint64_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = (uint64_t)__shfl_up((unsigned long long)inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (k < nblocks) block_sums[k] = run + inc - t;
        run += (uint64_t)__shfl((unsigned long long)inc, 63, 64);
    }
    if (lane == 0) *total = run;
}

__global__ void write_ciglets_kernel(zsw_alignment* aln, const uint8_t* status, uint32_t n, const uint64_t* block_sums,
                                     const uint64_t* cig_start, const uint32_t* cig_raw, int invert, uint32_t* out_inc,
                                     uint8_t* out_op, uint64_t cap) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const bool some = i < n && status[i] == ZSW_STATUS_SOME;
    const uint64_t mine = some ? aln[i].n_ciglets : 0;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {  // Hillis-Steele inclusive scan
        uint64_t t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (i >= n) return;
    if (!some) {
        zsw_alignment z = {};
        aln[i] = z;
        return;
    }
    const uint64_t off = block_sums[blockIdx.x] + sh[threadIdx.x] - mine;
    zsw_alignment rec = aln[i];
    const uint32_t raw = cig_raw[i];
    rec.ciglet_offset = off;
    const uint32_t* src = reinterpret_cast<const uint32_t*>((uintptr_t)cig_start[i]);
    if (off + mine <= cap) {
        uint64_t w = off;
        if (!invert) {
            for (uint32_t k = 0; k < raw; ++k) {  // make_reverse (state.rs:281-283)
                const uint32_t e = src[raw - 1 - k];
                out_inc[w] = e >> 8;
                out_op[w] = (uint8_t)(e & 0xff);
                ++w;
            }
        } else {  // Alignment::invert (output.rs:396-425)
            if (rec.ref_start > 0) {
                out_inc[w] = rec.ref_start;
                out_op[w] = 'S';
                ++w;
            }
            for (uint32_t k = 0; k < raw; ++k) {
                const uint32_t e = src[raw - 1 - k];
                uint8_t op = (uint8_t)(e & 0xff);
                if (op == 'S' || op == 'H') continue;
                if (op == 'D') op = 'I';
                else if (op == 'I') op = 'D';
                out_inc[w] = e >> 8;
                out_op[w] = op;
                ++w;
            }
            if (rec.ref_len > rec.ref_end) {
                out_inc[w] = rec.ref_len - rec.ref_end;
                out_op[w] = 'S';
                ++w;
            }
        }
    }
    if (invert) {
        const uint32_t rs = rec.ref_start, re = rec.ref_end, rl = rec.ref_len;
        rec.ref_start = rec.query_start;
        rec.ref_end = rec.query_end;
        rec.ref_len = rec.query_len;
        rec.query_start = rs;
        rec.query_end = re;
        rec.query_len = rl;
    }
    aln[i] = rec;
}

static size_t align_lds_bytes(uint32_t nv, int S) { (void)S; return align_rows_bytes(nv); }
static size_t align_lds_bytes_reg(uint32_t nv, int S) { return (size_t)S * ((nv + 3) / 4) * 64 * 4; }

template <typename K>
static hipError_t launch_with_lds(K kernel, const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream) {
    if (lds > 48 * 1024) {  // above the default dynamic-LDS limit: raise it (per device, so it is not cached in a process-wide flag)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 9 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_align_n(const AlignArgs& a, int S, uint32_t grid, hipStream_t stream) {
    const size_t lds = align_lds_bytes_reg(a.nv, S);
    if (a.nv <= 32) {
        switch ((a.nv + 3) / 4) {
            case 1: return launch_with_lds(&align_kernel_x<N, 1>, a, grid, lds, stream);
            case 2: return launch_with_lds(&align_kernel_x<N, 2>, a, grid, lds, stream);
            case 3: return launch_with_lds(&align_kernel_x<N, 3>, a, grid, lds, stream);
            case 4: return launch_with_lds(&align_kernel_x<N, 4>, a, grid, lds, stream);
            case 5: return launch_with_lds(&align_kernel_x<N, 5>, a, grid, lds, stream);
            case 6: return launch_with_lds(&align_kernel_x<N, 6>, a, grid, lds, stream);
            case 7: return launch_with_lds(&align_kernel_x<N, 7>, a, grid, lds, stream);
            case 8: return launch_with_lds(&align_kernel_x<N, 8>, a, grid, lds, stream);
        }
    }
    if (align_lds_bytes(a.nv, S) > ALIGN_LDS_LIMIT) {  // rows in HBM
        hipLaunchKernelGGL((align_kernel<N, true>), dim3(grid), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    return launch_with_lds(&align_kernel<N, false>, a, grid, align_lds_bytes(a.nv, S), stream);
}

}  // namespace zsw

using namespace zsw;

namespace {

hipError_t run_group(int N, AlignArgs a, int S, uint32_t grid, hipStream_t stream) {
    switch (N) {
        case 2: return launch_align_n<2>(a, S, grid, stream);
        case 4: return launch_align_n<4>(a, S, grid, stream);
        case 8: return launch_align_n<8>(a, S, grid, stream);
        case 16: return launch_align_n<16>(a, S, grid, stream);
        case 32: return launch_align_n<32>(a, S, grid, stream);
        case 64: return launch_align_n<64>(a, S, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// The host orchestration (zsw_align_batch / zsw_align_batch_from) lives in zsw_capi.hip because it needs the
// context internals; it calls these entry points.
namespace zsw {

static size_t align_ring_only_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid);

hipError_t align_pass2(int N, uint32_t nv, const BatchDev& b, const uint8_t* d_ref, uint32_t ref_len, const ScoringDev* d_sc,
                       int S, const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W,
                       uint32_t maxc, uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item,
                       uint64_t* d_cig_start, uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list,
                       uint32_t* d_fb_count, int invert, hipStream_t stream, const uint32_t* d_safe_row) {
    (void)d_safe_row;  // the 32-bit kernels keep the warm-up bound (they serve the reruns and the odd groups)
    AlignArgs a;
    a.safe_row = nullptr;
    a.prof_seq = nullptr;
    a.prof_len = 0;
    a.b = b;
    a.ref = d_ref;
    a.ref_len = ref_len;
    a.sc = d_sc;
    a.score = d_score;
    a.ref_end = d_ref_end;
    a.status = d_status;
    a.nv = nv;
    a.W = W;
    a.maxc = maxc;
    a.ring = d_ring;
    a.cig = d_cig;
    a.pool_base = pool_base;
    a.by_item = by_item;
    a.cig_start = d_cig_start;
    a.cig_raw = d_cig_raw;
    a.aln = d_aln;
    a.fb_list = d_fb_list;
    a.fb_count = d_fb_count;
    a.invert = invert;
    // profiles too long for LDS keep their rows behind the ring (align_ring_bytes reserves the space)
    a.rows = (nv > 32 && align_rows_bytes(nv) > ALIGN_LDS_LIMIT) ? d_ring + align_ring_only_bytes(N, nv, W, grid) : nullptr;
    a.next_item = d_fb_count + 1;  // the word after the fallback counter
    hipError_t ce = hipMemsetAsync(a.next_item, 0, 4, stream);
    if (ce != hipSuccess) return ce;
    return run_group(N, a, S, grid, stream);
}

static size_t align_ring_only_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid) {
    const size_t row = (size_t)N * (nv <= 32 ? ((nv + 3) / 4) * 4 : nv);
    return ((size_t)grid * (64 / N) * (size_t)W * row + 255) / 256 * 256;
}
size_t align_ring_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid) {
    size_t bytes = align_ring_only_bytes(N, nv, W, grid);
    if (nv > 32 && align_rows_bytes(nv) > ALIGN_LDS_LIMIT) bytes += (size_t)grid * align_rows_bytes(nv);
    return bytes;
}
// ---- the one-profile-many-sequences role (zsw_shared.hpp): generic kernel, rows in LDS or behind the ring ----
// The profile is striped over the whole shared sequence (nv = 125 at <i16,16> for 2 kb: 80 KB of DP rows per wavefront), so rows
// in LDS leave two wavefronts per CU — half a wavefront per SIMD, every instruction waiting for the one before. Beyond
// SHARED_ROWS_LDS_LIMIT the rows live behind the flag ring in global memory (L2 / HBM) and the CU holds as many wavefronts as
// its registers allow (100,000 reads vs 2 kb: 590 -> see DESIGN.md 4.5).
constexpr size_t SHARED_ROWS_LDS_LIMIT = 16 * 1024;
static bool shared_rows_global(uint32_t nv, int S) { return align_lds_bytes(nv, S) > SHARED_ROWS_LDS_LIMIT; }
// (rows of N * align_shared_row_stride(nv) bytes: what align_shared_kernel_h writes; the generic kernel's N * nv fit inside)
static size_t shared_ring_only_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid) {
    return ((size_t)grid * (64 / N) * (size_t)W * ((size_t)N * align_shared_row_stride(nv)) + 255) / 256 * 256;
}
size_t align_shared_ring_bytes(int N, uint32_t plen, uint32_t W, uint32_t grid, int S) {
    const uint32_t nv = (plen + (uint32_t)N - 1) / (uint32_t)N;
    size_t bytes = shared_ring_only_bytes(N, nv, W, grid);
    if (shared_rows_global(nv, S)) bytes += (size_t)grid * align_rows_bytes(nv);
    return bytes;
}

template <int N>
static hipError_t launch_align_shared_n(const AlignArgs& a, int S, uint32_t grid, hipStream_t stream, bool half_ok) {
    if (half_ok && align_shared_lds_bytes(a.nv, N) <= ALIGN_SHARED_LDS_LIMIT)
        return launch_with_lds(&align_shared_kernel_h<N>, a, grid, align_shared_lds_bytes(a.nv, N), stream);
    if (shared_rows_global(a.nv, S)) {
        hipLaunchKernelGGL((align_kernel<N, true, true>), dim3(grid), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    return launch_with_lds(&align_kernel<N, false, true>, a, grid, align_lds_bytes(a.nv, S), stream);
}

hipError_t align_pass2_shared(int N, const uint8_t* d_pseq, uint32_t plen, const BatchDev& b, const ScoringDev* d_sc, int S,
                              const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W, uint32_t maxc,
                              uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item, uint64_t* d_cig_start,
                              uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list, uint32_t* d_fb_count, int invert,
                              hipStream_t stream, bool half_ok) {
    const uint32_t nv = (plen + (uint32_t)N - 1) / (uint32_t)N;
    AlignArgs a;
    a.b = b;
    a.ref = nullptr;
    a.ref_len = 0;
    a.sc = d_sc;
    a.score = d_score;
    a.ref_end = d_ref_end;
    a.status = d_status;
    a.nv = nv;
    a.W = W;
    a.maxc = maxc;
    a.ring = d_ring;
    a.cig = d_cig;
    a.pool_base = pool_base;
    a.by_item = by_item;
    a.cig_start = d_cig_start;
    a.cig_raw = d_cig_raw;
    a.aln = d_aln;
    a.fb_list = d_fb_list;
    a.fb_count = d_fb_count;
    a.invert = invert;
    a.safe_row = nullptr;
    a.prof_seq = d_pseq;
    a.prof_len = plen;
    a.rows = shared_rows_global(nv, S) ? d_ring + shared_ring_only_bytes(N, nv, W, grid) : nullptr;
    a.next_item = d_fb_count + 1;
    hipError_t ce = hipMemsetAsync(a.next_item, 0, 4, stream);
    if (ce != hipSuccess) return ce;
    switch (N) {
        case 2: return launch_align_shared_n<2>(a, S, grid, stream, half_ok);
        case 4: return launch_align_shared_n<4>(a, S, grid, stream, half_ok);
        case 8: return launch_align_shared_n<8>(a, S, grid, stream, half_ok);
        case 16: return launch_align_shared_n<16>(a, S, grid, stream, half_ok);
        case 32: return launch_align_shared_n<32>(a, S, grid, stream, half_ok);
        case 64: return launch_align_shared_n<64>(a, S, grid, stream, half_ok);
    }
    return hipErrorInvalidValue;
}

// ---- packed kernel (zsw_align_pk{8,16,32,64}.hip) ----
hipError_t align_pk_occupancy_8(uint32_t nv, size_t lds, int* blocks_per_cu);
hipError_t align_pk_occupancy_16(uint32_t nv, size_t lds, int* blocks_per_cu);
hipError_t align_pk_occupancy_32(uint32_t nv, size_t lds, int* blocks_per_cu);
hipError_t align_pk_occupancy_64(uint32_t nv, size_t lds, int* blocks_per_cu);
hipError_t align_pk_launch_8(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream);
hipError_t align_pk_launch_16(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream);
hipError_t align_pk_launch_32(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream);
hipError_t align_pk_launch_64(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream);

static size_t align_pk_lds(uint32_t nv, int S) { return (size_t)S * nv * 64 * 4 + (((size_t)S * S + 15) / 16) * 16; }

bool align_pk_supported(int N, uint32_t nv, int S) {
    // at least two wavefronts' profiles per CU
    return (N == 8 || N == 16 || N == 32 || N == 64) && nv >= 1 && nv <= 32 && align_pk_lds(nv, S) <= 72 * 1024;
}

uint32_t align_pk_grid(int N, uint32_t nv, int S, uint32_t count, uint32_t cu_count) {
    const uint32_t rpw = 2 * 64 / (uint32_t)N;
    const uint32_t items = (count + rpw - 1) / rpw;
    int per_cu = 0;
    const size_t lds = align_pk_lds(nv, S);
    hipError_t e = N == 8 ? align_pk_occupancy_8(nv, lds, &per_cu) : N == 16 ? align_pk_occupancy_16(nv, lds, &per_cu)
                 : N == 32 ? align_pk_occupancy_32(nv, lds, &per_cu) : align_pk_occupancy_64(nv, lds, &per_cu);
    if (e != hipSuccess || per_cu < 1) per_cu = 4;
    return std::max<uint32_t>(1u, std::min<uint32_t>(items, cu_count * (uint32_t)per_cu));
}

size_t align_pk_ring_bytes(int N, uint32_t nv, uint32_t W, uint32_t grid) {
    const size_t row = (size_t)N * ((nv + 3) / 4) * 4;
    return ((size_t)grid * (2 * 64 / (size_t)N) * (size_t)W * row + 255) / 256 * 256;
}

hipError_t align_pass2_pk(int N, uint32_t nv, const BatchDev& b, const uint8_t* d_ref, uint32_t ref_len, const ScoringDev* d_sc,
                          int S, const uint32_t* d_score, const uint32_t* d_ref_end, const uint8_t* d_status, uint32_t W,
                          uint32_t maxc, uint8_t* d_ring, uint32_t grid, uint32_t* d_cig, uint64_t pool_base, int by_item,
                          uint64_t* d_cig_start, uint32_t* d_cig_raw, zsw_alignment* d_aln, uint32_t* d_fb_list,
                          uint32_t* d_fb_count, int invert, hipStream_t stream, const uint32_t* d_safe_row) {
    AlignArgs a;
    a.safe_row = d_safe_row;
    a.prof_seq = nullptr;
    a.prof_len = 0;
    a.b = b;
    a.ref = d_ref;
    a.ref_len = ref_len;
    a.sc = d_sc;
    a.score = d_score;
    a.ref_end = d_ref_end;
    a.status = d_status;
    a.nv = nv;
    a.W = W;
    a.maxc = maxc;
    a.ring = d_ring;
    a.cig = d_cig;
    a.pool_base = pool_base;
    a.by_item = by_item;
    a.cig_start = d_cig_start;
    a.cig_raw = d_cig_raw;
    a.aln = d_aln;
    a.fb_list = d_fb_list;
    a.fb_count = d_fb_count;
    a.invert = invert;
    a.rows = nullptr;
    a.next_item = d_fb_count + 1;  // the word after the fallback counter (the caller zeroes both)
    hipError_t ce = hipMemsetAsync(a.next_item, 0, 4, stream);
    if (ce != hipSuccess) return ce;
    const size_t lds = align_pk_lds(nv, S);
    switch (N) {
        case 8: return align_pk_launch_8(a, grid, lds, stream);
        case 16: return align_pk_launch_16(a, grid, lds, stream);
        case 32: return align_pk_launch_32(a, grid, lds, stream);
        case 64: return align_pk_launch_64(a, grid, lds, stream);
    }
    return hipErrorInvalidValue;
}

size_t align_lds_need(uint32_t nv, int S) {
    if (nv <= 32) return align_lds_bytes_reg(nv, S);
    return align_lds_bytes(nv, S) > ALIGN_LDS_LIMIT ? 0 : align_lds_bytes(nv, S);  // 0: the rows live in HBM
}

hipError_t align_finalize(zsw_alignment* d_aln, const uint8_t* d_status, uint32_t n, uint64_t* d_block_sums,
                          uint64_t* d_total, const uint64_t* d_cig_start, const uint32_t* d_cig_raw, int invert,
                          uint32_t* out_inc, uint8_t* out_op, uint64_t cap, bool count_only, hipStream_t stream) {
    const uint32_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nblocks == 0) return hipSuccess;
    if (count_only) {
        hipLaunchKernelGGL(count_block_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, stream, d_aln, d_status, n, d_block_sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, stream, d_block_sums, nblocks, d_total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(write_ciglets_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, stream, d_aln, d_status, n, d_block_sums,
                       d_cig_start, d_cig_raw, invert, out_inc, out_op, cap);
    return hipGetLastError();
}

}  // namespace zsw
