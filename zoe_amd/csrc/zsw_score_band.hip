// zsw_score_band.hip — the banded form of the seeded exact pass (sw_simd_score / sw_simd_score_ends, striped.rs:65-142, 153-336:
// MODE 0 score only, 1 with the reference end, 2 with both ends, 3 = 2 + whether the maximum sits in exactly one cell: then the
// tie rule does not matter and the shared-profile role, whose rule runs over the other sequence first, can use the result —
// zsw_capi_shared.hip). seed_window_kernel computes every query column for all
// ~len + 60 rows around the anchor; the alignment itself occupies a band of a few diagonals. Here a lane owns one read pair
// (16-bit halves, as everywhere) and walks it strip by strip: strip k = query columns [kC, (k+1)C) in registers (score_kernel_v2's
// packed column loop) against the reference rows [dt + kC - Wu, dt + (k+1)C + Wd) only, the strip's last column (H, outgoing F,
// true scores) handed to the next strip through a per-lane buffer in global memory (8 bytes per row, written and re-read once).
// No lane talks to another.
// What lies outside the band enters it as an UPPER BOUND, not as zero (zsw_seed.hpp, "banded pass"): every score is doubled, a
// value that comes from outside is odd, and the two bound programmes along the columns (SeedColDP: what a path above / below
// the band can hold after each column, k-mer by k-mer) run between the strips. An even maximum is the score of a real path
// inside the band that no path through an outside cell reaches; the reads whose maximum is odd, or below what a path ending
// outside may score, are flagged (`retry`: walked again in the launch's wider band, launch_score_seeded) or join the worklist
// of reads scored over all their cells. Host model, cell by cell against the full Gotoh matrix: tests/models/seed_band.cpp; the
// same model as a library checks this kernel's own values (maximum, oa, ob) on the GPU: tests/test_gpu_bounds.py.
// 150 bp: 10 strips x 30 rows x 16 columns (first tier) or 4 strips x 85 rows x 48 columns (second tier) per pair in one lane instead of
// 4 lanes x 211 steps x 38 columns.
#include <algorithm>
#include <type_traits>

#include "zsw_score_seed.hpp"
#include "zsw_score_v2.hpp"
#include "zsw_timer.hpp"

namespace zsw {

namespace {

// Columns per strip (a multiple of 8: a strip's residue codes are whole dwords of the packed reads; MODE 3 keeps one bit per column
// in a 32- or 64-bit mask). A strip computes a rectangle of C + Wu + Wd rows, the band itself is Wu + Wd + 1 diagonals: narrower strips
// waste fewer cells (16 columns: 10 strips of 30 rows for a 150-base read in the narrow band),
// but every strip boundary is a place where a bound that entered the strip's corner can leave it again inside a k-mer without paying
// for it (zsw_seed.hpp) — about lambda / 3 of slack per boundary and side. So the narrow first tier, which nearly every read
// within a few per cent of the reference passes, walks 16-column strips, and the reads that fail there are walked again in
// 48-column strips: two boundaries inside a 150-base read instead of nine (tests/models/seed_band.cpp `report`: of reads with 5 %
// substitutions 60 % prove their score in 16-column strips, 86 % in 32-column strips, 91 % in 48-column strips, 94 % without any
// boundary; at 8 %: 12 / 33 / 42 / 50 %). 48 columns are what the register file holds: H, E and the selectors are 144 VGPRs
// (score only: two wavefronts per SIMD; with the MODE 2 snapshot of the H row one, the AGPRs as spill space).
constexpr int BC_NARROW = 16, BC_WIDE = 48;

// What the kernel keeps per lane between strips, in LDS ([field][lane]: the registers of the row loop hold nothing but the strip):
// the k-mer layout and masks of the lane's two reads, and the two bound programmes of zsw_seed.hpp (SeedColDP) for BOTH reads at
// once — 16-bit halves like everything else, the tracks biased by PKB so that "no such path" (0) stays below every real value.
// The host model (tests/models/seed_band.cpp) runs the same programmes with plain integers through the header's functions;
// tests/test_gpu_bounds.py requires this kernel's values to equal the model's.
enum BandField {
    BF_M_A, BF_STRIDE_A, BF_C0_A, BF_MAGIC_A, BF_MASKS_A, BF_M_B, BF_STRIDE_B, BF_C0_B, BF_MAGIC_B, BF_MASKS_B, BF_LAM2,
    BF_UP_CH, BF_UP_FR, BF_LO_CH, BF_LO_FR, BF_OA, BF_OB, BF_YH, BF_YF, BF_N
};
constexpr uint32_t PKB = 0x2000u, PKB2 = PKB * 0x00010001u;

struct BandLayout {
    int m, stride, c0;
    uint32_t magic, fa, fb;
};
struct BandPk {
    uint32_t ch, fr;  // pays / does not pay for the next k-mer of the mask that ends (per half: value + PKB, 0 = none)
};

__device__ __forceinline__ void band_lane_init(const SeedParams& p, int* st, int lenA, uint32_t masksA, int lenB, uint32_t masksB) {
    int m, stride, c0;
    seed_layout(lenA, p.K, p.spacer, &m, &stride, &c0);
    st[BF_M_A * BLOCK] = m;
    st[BF_STRIDE_A * BLOCK] = stride;
    st[BF_C0_A * BLOCK] = c0;
    st[BF_MAGIC_A * BLOCK] = (int)seed_div_magic(stride);
    st[BF_MASKS_A * BLOCK] = (int)masksA;
    const uint32_t lamA = (uint32_t)seed_lambda(p, stride);
    seed_layout(lenB, p.K, p.spacer, &m, &stride, &c0);
    st[BF_M_B * BLOCK] = m;
    st[BF_STRIDE_B * BLOCK] = stride;
    st[BF_C0_B * BLOCK] = c0;
    st[BF_MAGIC_B * BLOCK] = (int)seed_div_magic(stride);
    st[BF_MASKS_B * BLOCK] = (int)masksB;
    st[BF_LAM2 * BLOCK] = (int)(lamA | ((uint32_t)seed_lambda(p, stride) << 16));
    st[BF_UP_CH * BLOCK] = st[BF_LO_CH * BLOCK] = (int)PKB2;
    st[BF_UP_FR * BLOCK] = st[BF_LO_FR * BLOCK] = 0;
    st[BF_OA * BLOCK] = st[BF_OB * BLOCK] = 0;
    st[BF_YH * BLOCK] = st[BF_YF * BLOCK] = 0;
}
__device__ __forceinline__ BandLayout band_layout(const int* st, bool second) {
    BandLayout y;
    const int o = second ? (BF_M_B - BF_M_A) * BLOCK : 0;
    y.m = st[BF_M_A * BLOCK + o];
    y.stride = st[BF_STRIDE_A * BLOCK + o];
    y.c0 = st[BF_C0_A * BLOCK + o];
    y.magic = (uint32_t)st[BF_MAGIC_A * BLOCK + o];
    const uint32_t masks = (uint32_t)st[BF_MASKS_A * BLOCK + o];
    y.fa = masks & 0xffffu;
    y.fb = masks >> 16;
    return y;
}

// 0xffff in the half whose read has bit c set
__device__ __forceinline__ uint32_t half_mask(uint32_t bitsA, uint32_t bitsB, int c) {
    const uint32_t mA = (uint32_t)__builtin_amdgcn_sbfe((int)bitsA, (uint32_t)c, 1u), mB = (uint32_t)__builtin_amdgcn_sbfe((int)bitsB, (uint32_t)c, 1u);
    return __builtin_amdgcn_perm(mB, mA, 0x05040100u);
}
// (strips of more than 32 columns keep their per-column bits in 64-bit masks; c is a constant after unrolling)
__device__ __forceinline__ uint32_t half_mask(uint64_t bitsA, uint64_t bitsB, int c) {
    return half_mask((uint32_t)(c < 32 ? bitsA : bitsA >> 32), (uint32_t)(c < 32 ? bitsB : bitsB >> 32), c & 31);
}
__device__ __forceinline__ int mask_popc(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int mask_popc(uint64_t x) { return __popcll(x); }
template <class M>
__device__ __forceinline__ M mask_first(int n) {  // bits [0, n)
    return n >= (int)sizeof(M) * 8 ? ~(M)0 : (((M)1 << n) - 1u);
}
__device__ __forceinline__ uint32_t pk_subu_sat(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_shr1(uint32_t a) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) >> (us2){1, 1}); }
// seed_tag / seed_untag per half: 2v + TAG, a bound of 0 stays the plain zero floor
template <int TAG>
__device__ __forceinline__ uint32_t pk_tag(uint32_t v) {
    if (TAG < 0) return pk_maxu(v << 1, 0x00010001u) - 0x00010001u;
    uint32_t nz;  // 1 in the halves that are not 0 (written as min(v, 1) the compiler turns it into four compares)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(v), "v"(0x00010001u));
    return (v << 1) | nz;
}
template <int TAG>
__device__ __forceinline__ uint32_t pk_untag(uint32_t t2) {
    return TAG > 0 ? pk_shr1(t2) : pk_shr1(pk_addu(t2, 0x00010001u));
}
// seed_col_step's events per half (sm / em: the halves in which a k-mer of the mask starts / ends in this column)
__device__ __forceinline__ void pk_events(BandPk* t, uint32_t sm, uint32_t em, uint32_t lam2) {
    t->fr = pk_maxu(t->fr, PKB2 & sm);
    t->ch = pk_subu(t->ch, lam2 & em);
    t->ch = pk_maxu(t->ch, t->fr & em);
    t->fr &= ~em;
}

template <int C, int MINW, int MODE>
__global__ __launch_bounds__(BLOCK, MINW) void seed_band_kernel(SeedBandArgs a) {
    constexpr int TAG = MODE == 0 ? -1 : 1;
    using M = std::conditional_t<(C > 32), uint64_t, uint32_t>;  // one bit per column of a strip
    static_assert(C % 8 == 0 && C <= 64, "a strip's residue codes are whole dwords; its column masks at most 64 bits");
    __shared__ uint16_t sel_lut[16];
    __shared__ int sst[BF_N * BLOCK];
    const int tid = threadIdx.x;
    int* const st = sst + tid;
    if (tid < 16) {
        const uint32_t k = (uint32_t)tid;
        sel_lut[tid] = (uint16_t)(k < 4 ? (2 * k + 1) | ((8 + k) << 8) : (k == 15 ? 0x0c00u : (2 * (k - 3)) | 0x0c00u));
    }
    __syncthreads();
    const int R = (int)a.ref_len;
    const uint32_t n_items = a.n_dev ? min(*a.n_dev, a.n) : a.n;
    const uint32_t n_pairs = (n_items + 1) / 2;
    const bool bail = a.bail_check && a.accepted && ((uint64_t)(*a.accepted) * SEED_BAIL_RATIO < (uint64_t)n_items || n_items < a.bail_below);
    uint32_t n_accepted = 0;  // (first tier)
    const uint32_t ge2 = a.ge2, gd2 = a.gd2;  // doubled gap penalties (the tables hold doubled scores)
    const uint32_t ge1 = ge2 & 0xffffu;
    const uint32_t maxw2 = (uint32_t)a.sp.maxw * 0x00010001u, go2p = (uint32_t)a.sp.go * 0x00010001u;  // plain (not doubled) scores of the bound programmes
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
        if (bail) {  // (launch-uniform)
            a.fail_list[atomicAdd(a.fail_count, 1u)] = idA;
            if (validB) a.fail_list[atomicAdd(a.fail_count, 1u)] = idB;
            continue;
        }
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
        // what the seed kernel left: the k-mer masks of the two sides; (potential and d_fa: seed_safe_start below)
        const uint32_t infoA = a.info[ridA], infoB = a.info[ridB];
        band_lane_init(a.sp, st, (int)lenA, a.band_masks[ridA], (int)lenB, validB ? a.band_masks[ridB] : 0u);

        const uint32_t* codeA = a.codes + (size_t)ridA * a.cs;
        const uint32_t* codeB = a.codes + (size_t)ridB * a.cs;
        uint32_t best = 0;          // doubled scores, odd: held by a path that touched a cell outside the band
        // ends (MODE 1, 2): first row holding the maximum, then the first column of that row (striped.rs:296-321). A strip walks its
        // rows in order; strips overlap in rows, so each strip keeps its own (maximum, first row, H row of that row) and the strips
        // merge by (higher maximum, then earlier row; the same row in two strips: the earlier strip holds the earlier column).
        int bestA = 0, bestB = 0, rowA = 0x7fffffff, rowB = 0x7fffffff, colA = 0x7fffffff, colB = 0x7fffffff;
        bool multA = false, multB = false;  // MODE 3: the maximum so far sits in more than one (real) cell
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
            const int nrA = max(0, min(C, (int)lenA - k * C)), nrB = max(0, min(C, (int)lenB - k * C));  // the strip's real columns per read
            uint32_t Dr = (a.floor0 - ge1) * 0x00010001u;                   // D of the row before the strip's first row
            // above the strip's columns: a(c) of either read; the strip's first row receives it (its E: less gap_open)
            uint32_t H[C], E[C];
            const bool all_full = __ballot(nrA != C || nrB != C) == 0;  // no lane of the wavefront has a padding column in this strip
            const uint32_t lam2 = (uint32_t)st[BF_LAM2 * BLOCK];
            {
                const BandLayout yA = band_layout(st, false), yB = band_layout(st, true);
                const SeedStripEventsT<M> eA = seed_strip_events_t<M>(k * C, C, yA.m, yA.c0, yA.stride, a.sp.K, yA.magic, yA.fa);
                const SeedStripEventsT<M> eB = seed_strip_events_t<M>(k * C, C, yB.m, yB.c0, yB.stride, a.sp.K, yB.magic, yB.fa);
                const M any = eA.start | eA.end | eB.start | eB.end;
                BandPk up;
                up.ch = (uint32_t)st[BF_UP_CH * BLOCK];
                up.fr = (uint32_t)st[BF_UP_FR * BLOCK];
                uint32_t oa2 = (uint32_t)st[BF_OA * BLOCK];
                const uint32_t above = top > 0 ? 0xffffffffu : 0u;
                const uint32_t DrE = pk_addu(Dr, ge2);
                auto sweep = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    const M realA = mask_first<M>(nrA), realB = mask_first<M>(nrB);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        up.ch += maxw2;
                        up.fr += maxw2;
                        if (__ballot((uint32_t)(any >> c) & 1u) != 0) pk_events(&up, half_mask(eA.start, eB.start, c), half_mask(eA.end, eB.end, c), lam2);
                        uint32_t v = pk_subu_sat(pk_maxu(up.ch, up.fr), PKB2) & above;
                        if (!FULL) v &= half_mask(realA, realB, c);
                        oa2 = pk_maxu(oa2, v);
                        H[c] = pk_addu(Dr, pk_tag<TAG>(v));
                        E[c] = pk_addu(DrE, pk_tag<TAG>(pk_subu_sat(v, go2p)));
                    }
                };
                if (all_full) sweep(std::true_type{});
                else sweep(std::false_type{});
                st[BF_UP_CH * BLOCK] = (int)up.ch;
                st[BF_UP_FR * BLOCK] = (int)up.fr;
                st[BF_OA * BLOCK] = (int)oa2;
            }
            // the previous strip's last column: rows [top - 1, prev_bot) wait in bnd[row - (top - 1)] as true scores; below them the
            // strip receives the bound of the cells below the previous strip
            const bool have_left = k > 0;
            const uint32_t yh2 = have_left ? pk_tag<TAG>((uint32_t)st[BF_YH * BLOCK]) : 0u;
            const uint32_t yf2 = have_left ? pk_tag<TAG>((uint32_t)st[BF_YF * BLOCK]) : 0u;
            uint32_t Hin_prev = Dr;
            if (have_left && top >= 1) Hin_prev = pk_addu(Dr, top - 1 < prev_bot ? bnd[0].x : yh2);
            uint2 w = a.gtab[SEED_GTAB_PAD + top];
            uint2 bl = make_uint2(0u, 0u);
            if (have_left && top < prev_bot) bl = bnd[(size_t)1 * BLOCK];
            uint32_t uk = 0;                // the largest H leaving through the right edge in the rows above the next strip
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
            const M realA = mask_first<M>(nrA), realB = mask_first<M>(nrB);
            bool smA = false, smB = false;
#pragma unroll 1
            for (int r = top; r < bot; ++r) {
                const uint2 wn = a.gtab[SEED_GTAB_PAD + r + 1];
                const bool left = have_left && r < prev_bot;
                uint2 bln = make_uint2(0u, 0u);
                if (have_left && r + 1 < prev_bot) bln = bnd[(size_t)(r + 1 - top + 1) * BLOCK];
                Dr = pk_addu(Dr, ge2);
                const uint32_t Dn = pk_addu(Dr, ge2);
                const uint32_t Hin = pk_addu(Dr, left ? bl.x : yh2);
                uint32_t F = pk_addu(Dr, left ? bl.y : yf2);
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
                const uint32_t tH = pk_subu(H[C - 1], Dr);  // true (doubled) score leaving the strip in this row
                if (has_next) {
                    if (r >= next_top - 1) bnd[(size_t)(r - (next_top - 1)) * BLOCK] = make_uint2(tH, pk_subu(F, Dr));
                    if (r < next_top) uk = pk_maxu(uk, tH);  // (the outgoing F of a cell never exceeds its H)
                }
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
                            M hitA = 0, hitB = 0;
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                const uint32_t x = H[c] ^ target;
                                if (!(x & 0xffffu)) hitA |= (M)1 << c;
                                if (!(x >> 16)) hitB |= (M)1 << c;
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
                    M hitA = 0, hitB = 0;
                    const int sdA = (int)(snapD & 0xffffu), sdB = (int)(snapD >> 16);
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        const uint32_t sv = snap[MODE >= 2 ? c : 0];
                        if ((int)(sv & 0xffffu) - sdA == sA) {
                            cA = k * C + c;
                            hitA |= (M)1 << c;
                        }
                        if ((int)(sv >> 16) - sdB == sB) {
                            cB = k * C + c;
                            hitB |= (M)1 << c;
                        }
                    }
                    if (upA) colA = cA;
                    if (upB) colB = cB;
                    if (MODE == 3) {
                        // cells of this strip holding its maximum: the real columns of the row of its latest rise, plus later rows
                        const int nA = mask_popc(hitA & realA) + (smA ? 1 : 0), nB = mask_popc(hitB & realB) + (smB ? 1 : 0);
                        if (sA > 0 && (upA && !eqA)) multA = nA > 1;   // a higher maximum: this strip's cells are all there are so far
                        else if (eqA && nA > 0) multA = true;          // the same value in two strips: two cells
                        if (sB > 0 && (upB && !eqB)) multB = nB > 1;
                        else if (eqB && nB > 0) multB = true;
                    }
                }
            }
            // what left the band through the right edge joins the paths above it after the strip's last column
            const int xl = (k + 1) * C - 1;
            const BandLayout yA = band_layout(st, false), yB = band_layout(st, true);
            if (has_next && min(bot, next_top) > top) {
                const uint32_t joinA = xl < (int)lenA - 1 ? 0xffffu : 0u, joinB = xl < (int)lenB - 1 ? 0xffff0000u : 0u;
                const uint32_t freeA = seed_exit_is_free(xl, yA.m, yA.c0, yA.stride, a.sp.K, a.sp.spacer, yA.magic, yA.fa) ? 0xffffu : 0u;
                const uint32_t freeB = seed_exit_is_free(xl, yB.m, yB.c0, yB.stride, a.sp.K, a.sp.spacer, yB.magic, yB.fa) ? 0xffff0000u : 0u;
                const uint32_t ux = pk_addu(pk_untag<TAG>(uk), PKB2) & (joinA | joinB);
                st[BF_UP_FR * BLOCK] = (int)pk_maxu((uint32_t)st[BF_UP_FR * BLOCK], ux & (freeA | freeB));
                st[BF_UP_CH * BLOCK] = (int)pk_maxu((uint32_t)st[BF_UP_CH * BLOCK], ux & ~(freeA | freeB));
            }
            // below the strip's columns: b(c) of either read, joined by the cells of the strip's last row (H holds them)
            {
                const uint32_t below = bot < R ? 0xffffffffu : 0u, exits = (bot < R && bot > top) ? 0xffffffffu : 0u;
                const SeedStripEventsT<M> eA = seed_strip_events_t<M>(k * C, C, yA.m, yA.c0, yA.stride, a.sp.K, yA.magic, yA.fb);
                const SeedStripEventsT<M> eB = seed_strip_events_t<M>(k * C, C, yB.m, yB.c0, yB.stride, a.sp.K, yB.magic, yB.fb);
                const M any = eA.start | eA.end | eB.start | eB.end;
                BandPk lo;
                lo.ch = (uint32_t)st[BF_LO_CH * BLOCK];
                lo.fr = (uint32_t)st[BF_LO_FR * BLOCK];
                uint32_t ob2 = (uint32_t)st[BF_OB * BLOCK], b2 = 0;
                auto sweep = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    const M realA = mask_first<M>(nrA), realB = mask_first<M>(nrB);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        lo.ch += maxw2;
                        lo.fr += maxw2;
                        if (__ballot((uint32_t)(any >> c) & 1u) != 0) pk_events(&lo, half_mask(eA.start, eB.start, c), half_mask(eA.end, eB.end, c), lam2);
                        uint32_t he = pk_addu(pk_untag<TAG>(pk_subu(H[c], Dr)), PKB2) & exits;
                        uint32_t rm = 0xffffffffu;
                        if (!FULL) {
                            rm = half_mask(realA, realB, c);
                            he &= rm;
                        }
                        const uint32_t im = half_mask(eA.inside, eB.inside, c);
                        lo.fr = pk_maxu(lo.fr, he & im);
                        lo.ch = pk_maxu(lo.ch, he & ~im);
                        b2 = pk_subu_sat(pk_maxu(lo.ch, lo.fr), PKB2) & rm;
                        ob2 = pk_maxu(ob2, b2 & below);
                    }
                };
                if (all_full) sweep(std::true_type{});
                else sweep(std::false_type{});
                st[BF_LO_CH * BLOCK] = (int)lo.ch;
                st[BF_LO_FR * BLOCK] = (int)lo.fr;
                st[BF_OB * BLOCK] = (int)ob2;
                // the next strip's first column, rows below this strip's last: H of the cell to the left; F entering (the bound may
                // have stood lambda - maxw higher just before a k-mer's last column; a horizontal run opens with gap_open)
                const uint32_t nm = ((nrA == C ? 0xffffu : 0u) | (nrB == C ? 0xffff0000u : 0u)) & below;
                st[BF_YH * BLOCK] = (int)(b2 & nm);
                st[BF_YF * BLOCK] = (int)(pk_subu_sat(pk_addu(b2, pk_subu_sat(lam2, maxw2)), go2p) & nm);
            }
            prev_bot = bot;
        }
        const uint32_t best2A = best & 0xffffu, best2B = best >> 16;
        const int tallA = (int)(infoA & 0xffffu), tallB = (int)(infoB & 0xffffu);
        const int dfaA = (int)((infoA >> 16) & 0xffu), dfaB = (int)((infoB >> 16) & 0xffu);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !validB) break;
            const uint32_t id = h ? idB : idA;
            const uint32_t b2 = h ? best2B : best2A;
            const int oa = (int)(((uint32_t)st[BF_OA * BLOCK] >> (h ? 16 : 0)) & 0xffffu), ob = (int)(((uint32_t)st[BF_OB * BLOCK] >> (h ? 16 : 0)) & 0xffffu);
            const int S = (int)(b2 >> 1), outside = max(oa, ob);
            // an odd maximum: a path through a cell outside the band may hold it. Score only: a path that ends outside matters if it
            // can score MORE than S; with ends also if it can score S (it could end in an earlier row or column)
            const bool redo = (b2 & 1u) != 0 || (MODE == 0 ? outside > S : outside >= S);
            if (a.dbg) {  // tests/test_gpu_bounds.py: the kernel's own values against the host model's
                int* rec = a.dbg + (size_t)id * 8;
                rec[0] = (int)b2;
                rec[1] = oa;
                rec[2] = ob;
                rec[3] = dtmin;
                rec[4] = dtmax;
                rec[5] = n_strips | (wu << 8) | (wd << 20);
                rec[6] = h ? dtB : dtA;
                rec[7] = (((h ? lenB : lenA) == 0 || redo) ? 0 : 1) | (C << 8);
            }
            if ((h ? lenB : lenA) == 0 || redo) {
                if (a.retry) a.retry[h ? itemB : itemA] = 1;
                else a.fail_list[atomicAdd(a.fail_count, 1u)] = id;
            } else {
                ++n_accepted;
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
    if (a.retry && a.accepted) {  // first tier: one atomic per wavefront
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n_accepted += (uint32_t)__shfl_xor((int)n_accepted, d, 64);
        if ((tid & 63) == 0 && n_accepted) atomicAdd(a.accepted, n_accepted);
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

constexpr int BC = BC_WIDE;  // (sizes and limits below: the wider of the two)
bool seed_band_applicable(const SeedParams& p, uint32_t max_len, uint32_t rebase_rows, uint32_t limit) {
    // a strip's rows must fit one drift period of the packed domain (no re-basing inside a strip); doubled scores (+ the mark)
    // must stay inside the packed range, the bound programmes' plain values below their bias; the masks are relative to Dn / Dm
    static_assert(SEED_NARROW_WU >= SEED_DN && SEED_NARROW_WD >= SEED_DM, "the band must hold the diagonals that count as near the anchor");
    if (p.M1 < p.Dn || p.Wd < p.Dm) return false;
    if (2ull * (uint64_t)p.maxw * max_len + 8 >= limit || (uint64_t)p.maxw * (max_len + BC) >= 0x2000u) return false;
    return (uint32_t)BC + seed_band_rows(p, max_len) + 2 <= rebase_rows;
}

hipError_t launch_seed_band(const SeedBandArgs& a, int mode, bool narrow_strips, hipStream_t stream) {
    // H, E and the selectors are 3 * C registers (4 * C with the MODE 2 snapshot of the H row); measured: the 48-column ends kernels at
    // one wavefront per SIMD (AGPRs as spill space) beat two with scratch by 16 %, the score-only kernel at two beats one by 4 %
    if (narrow_strips) {
        if (mode == 0) hipLaunchKernelGGL((seed_band_kernel<BC_NARROW, 4, 0>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else if (mode == 1) hipLaunchKernelGGL((seed_band_kernel<BC_NARROW, 4, 1>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else if (mode == 2) hipLaunchKernelGGL((seed_band_kernel<BC_NARROW, 3, 2>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((seed_band_kernel<BC_NARROW, 3, 3>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((seed_band_kernel<BC_WIDE, 2, 0>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else if (mode == 1) hipLaunchKernelGGL((seed_band_kernel<BC_WIDE, 1, 1>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else if (mode == 2) hipLaunchKernelGGL((seed_band_kernel<BC_WIDE, 1, 2>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((seed_band_kernel<BC_WIDE, 1, 3>), dim3(a.grid), dim3(BLOCK), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace zsw
