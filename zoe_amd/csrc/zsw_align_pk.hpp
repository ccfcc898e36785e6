// zsw_align_pk.hpp — the row update of the packed alignment kernel (align_kernel_pk, zsw_align_pk.hip), written once
// over a "wave value" type so that the same text is compiled for gfx950 (V = one 32-bit VGPR per lane, every op below is a
// single VALU/DPP instruction) and for the host (tests/models/align_pk_twin.cpp: V = 64 explicit lanes), where it is
// compared cell by cell with the oracle's literal restatement of sw_simd_align (striped.rs:449-598).
//
// Layout. N adjacent lanes are one Zoe vector `Simd<T, N>`; a lane owns the query positions q = v + lane*nv of its segment
// (profile.rs:285), v = 0..NV-1 at static register indices. Every 32-bit lane value carries TWO reads as packed unsigned
// 16-bit halves (true scores, offset T::MIN removed): 2*64/N reads per wavefront. Scores stay below 2^15, so signed and
// unsigned packed instructions agree on them; the host routes reads whose score could leave that range to the 32-bit kernel.
//
// Lazy-F in closed form (instead of Zoe's step-by-step loop, striped.rs:528-553). In round k the F vector of lane l is
// Fk[l] = Fend[l-1-k] - k*nv*ge (saturating), the value visiting cell (v, l) is Fk[l] - v*ge, and with
//   Y(v, l)   = sat_sub(Hmain(v, l), go) + v*ge          (fixed for the row)
//   P_{k-1}[l] = max_{k' < k} Fk'[l]                      (what earlier rounds brought to the lane)
// the loop's test `F > H - go` at step (k, v) in lane l is  Fk[l] > Y(v, l)  and  Fk[l] + go > P_{k-1}[l].
// So the break position T = (kb, vb) follows from one compare per cell and round (a bit string over v per lane, OR-reduced
// over the N lanes), with no update of H or the flags inside the loop. Afterwards, with M(v, l) = the largest F that visited
// the cell = sat_sub(v < vb ? P_kb[l] : P_{kb-1}[l], v*ge):
//   H = max(Hmain, M);  if M >= Hmain and M > 0: flags = (flags & UP_EXTENDING) | LEFT   (simd_correct_and_set_left)
//   if sat_sub(M, ge) > sat_sub(H, go): flags |= LEFT_EXTENDING
// which is what the sequence of visits leaves behind (the last visit that reaches the running maximum resets the flags, and
// LEFT_EXTENDING is monotone in the visiting F). Rows whose flags are not kept only need H, which does not depend on T
// (a visit after the break cannot raise H), so they take a log-step prefix scan over the lanes instead.
// tests/models/align_closed_form.cpp checks this derivation alone (plain i32 arithmetic), align_pk_twin.cpp this file.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZSW_PK_FN __host__ __device__ __forceinline__
#else
#define ZSW_PK_FN inline
#endif

namespace zsw_pk {

constexpr uint32_t ONE2 = 0x00010001u;
// backtrack.rs:18-34, one 5-bit code per read in each 16-bit half
constexpr uint32_t UP2 = 0x00010001u, UPX2 = 0x00020002u, LEFT2 = 0x00040004u, LEFTX2 = 0x00080008u, STOP2 = 0x00100010u;

template <class O, int NV>
struct Consts {
    typename O::V go2, ge2;  // gap_open, gap_extend (positive magnitudes) in both halves
    typename O::V nvge2;     // nv * gap_extend
    typename O::V keep;      // 0 in the first lane of every N-lane group, ~0 elsewhere
    typename O::V one;       // 1 in both halves. The device build hides the value from the optimiser, which otherwise rewrites
                             // min(x - y, 1) into per-half compares and selects (three instructions per half instead of one packed one)
    uint32_t ge;             // scalar gap_extend
};

template <class O, int NV>
struct State {
    typename O::V H[NV], E[NV];
};

template <class O>
static ZSW_PK_FN typename O::V nzmask(typename O::V d, typename O::V one) {  // per half: 0xffff where the half is non-zero
    return O::sub(O::splat(0), O::min_u(d, one));
}

// v*ge in both halves (v, ge wave-uniform)
template <class O>
static ZSW_PK_FN typename O::V vge2(uint32_t v, uint32_t ge) {
    const uint32_t x = v * ge;
    return O::splat(x | (x << 16));
}

// Main pass of one reference row (striped.rs:481-526). p[v]: the profile scores of the row (i16 per half).
// On return st.H holds Hmain, st.E the next row's E, flg[v] (FLAGS) the main-pass flag codes, Fend the F leaving each segment.
template <class O, int N, int NV, bool FLAGS>
static ZSW_PK_FN void main_pass(State<O, NV>& st, const typename O::V (&p)[NV], const Consts<O, NV>& c, typename O::V& Fend,
                             typename O::V (&flg)[NV]) {
    using V = typename O::V;
    const V one = c.one;
    V F = O::splat(0);
    V Hd = O::template shr1<N>(st.H[NV - 1], c.keep);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const V Eo = st.E[v];
        const V hold = st.H[v];
        const V h = O::max_i(O::max_i(O::add(Hd, p[v]), Eo), F);  // Eo >= 0 supplies the floor at T::MIN
        const V hg = O::sub_sat(h, c.go2);
        const V En = O::max_u(O::sub_sat(Eo, c.ge2), hg);
        const V Fn = O::max_u(O::sub_sat(F, c.ge2), hg);
        if constexpr (FLAGS) {
            const V n_up = O::min_u(O::sub(h, Eo), one);   // 0 where E == H
            const V n_lf = O::min_u(O::sub(h, F), one);    // 0 where F == H
            const V ux = O::min_u(O::sub(En, hg), one);    // 1 where E' > H - go
            const V lx = O::min_u(O::sub(Fn, hg), one);    // 1 where F' > H - go
            const V n_st = O::min_u(h, one);               // 0 where H == T::MIN
            V acc = O::lshl_or(n_lf, 2, n_up);
            acc = O::lshl_or(lx, 3, acc);
            acc = O::lshl_or(ux, 1, acc);
            acc = O::xor_(acc, O::splat(UP2 | LEFT2));
            // H == MIN: E = F = 0 = H, so the code so far is UP|LEFT (5); simd_stop overwrites it with STOP (16)
            flg[v] = O::mad(O::xor_(n_st, one), O::splat(0x000b000bu), acc);
        }
        st.H[v] = h;
        st.E[v] = En;
        F = Fn;
        Hd = hold;
    }
    Fend = F;
}

// Bit strings over the vectors, 16 per register half: chunk j holds vectors 16*j .. 16*j + bits(j) - 1, first vector in the
// highest used bit.
template <int NV>
struct Chunks {
    static constexpr int NM = (NV + 15) / 16;
    static constexpr int bits(int j) { return j + 1 < NM ? 16 : NV - 16 * (NM - 1); }
    static constexpr uint32_t all(int j) { return ((1u << bits(j)) - 1u) * ONE2; }
};

// Break position of the lazy-F loop. Fend must be 0 in the halves of reads that are past their last row.
// Returns Pa = P_kb, Pb = P_{kb-1} and, per half, the bit strings of the breaking round (set: vector visited).
template <class O, int N, int NV>
static ZSW_PK_FN void lazy_rounds(const State<O, NV>& st, typename O::V Fend, const Consts<O, NV>& c, typename O::V& Pa,
                               typename O::V& Pb, typename O::V (&mfin)[Chunks<NV>::NM]) {
    using V = typename O::V;
    using CH = Chunks<NV>;
    const V one = c.one;
    V Y[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) Y[v] = O::add_sat(O::sub_sat(st.H[v], c.go2), vge2<O>(v, c.ge));
    V Ymax = Y[0];  // a lane whose F exceeds it passes at every vector: the round is complete whatever the other lanes do
#pragma unroll
    for (int v = 1; v < NV; ++v) Ymax = O::max_u(Ymax, Y[v]);
    Pa = O::splat(0);
    Pb = O::splat(0);
#pragma unroll
    for (int j = 0; j < CH::NM; ++j) mfin[j] = O::splat(CH::all(j));  // never broke: every vector visited in every round
    V run = O::splat(0xffffffffu);
    V Fk = O::template shr1<N>(Fend, c.keep);
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
        if (k) Fk = O::sub_sat(O::template shr1<N>(Fk, c.keep), c.nvge2);
        // the test is Fk > Y(v) and Fk + go > P_{k-1}, i.e. Fk > max(Y(v), sat_sub(P_{k-1}, go)): a lane whose Fk does not exceed
        // the second bound takes no part (Fe = 0 passes nowhere, Y >= 0). Reads that have left their loop carry Fk = 0.
        const V Fe = O::mul(Fk, O::min_u(O::sub_sat(Fk, O::sub_sat(Pb, c.go2)), one));
        // Most rounds are carried by one lane whose F is far above everything in its segment (the F leaving the alignment's
        // diagonal sweeps the lanes to its right): when every read that still runs has such a lane, the bit strings are all ones.
        const V strong = nzmask<O>(O::template group_or<N>(O::sub_sat(Fe, Ymax)), one);
        V m[CH::NM];
#pragma unroll
        for (int j = 0; j < CH::NM; ++j) m[j] = O::and_(strong, O::splat(CH::all(j)));
        if (O::any(O::bfi(strong, O::splat(0), run))) {  // a running read without such a lane: one compare per cell
#pragma unroll
            for (int j = 0; j < CH::NM; ++j) {
                V x = O::splat(0);
#pragma unroll
                for (int v = 16 * j; v < 16 * j + CH::bits(j); ++v) x = O::lshl_or(x, 1, O::min_u(O::sub_sat(Fe, Y[v]), one));
                m[j] = O::template group_or<N>(x);
            }
        }
        V diff = O::xor_(m[0], O::splat(CH::all(0)));
#pragma unroll
        for (int j = 1; j < CH::NM; ++j) diff = O::and_or(O::xor_(m[j], O::splat(CH::all(j))), O::splat(0xffffffffu), diff);
        const V brk = nzmask<O>(diff, one);  // some vector without any lane passing
        const V Pn = O::max_u(Pb, Fk);
        const V fin = O::and_(run, brk);
        const V upd = O::xor_(run, fin);
        Pa = O::bfi(run, Pn, Pa);
        Pb = O::bfi(upd, Pn, Pb);
#pragma unroll
        for (int j = 0; j < CH::NM; ++j) mfin[j] = O::bfi(fin, m[j], mfin[j]);
        run = upd;
        Fk = O::and_(Fk, run);
        if (!O::any(run)) break;
    }
}

// per half: vectors visited in the breaking round = leading ones of the concatenated bit strings
template <class O, int NV>
static ZSW_PK_FN typename O::V visited(const typename O::V (&mfin)[Chunks<NV>::NM]) {
    using V = typename O::V;
    using CH = Chunks<NV>;
    V total = O::splat(0);
    V open = O::splat(ONE2);  // 1 while every earlier chunk was all ones
#pragma unroll
    for (int j = 0; j < CH::NM; ++j) {
        const V lo = O::lead_ones(O::and_(mfin[j], O::splat(0xffffu)), CH::bits(j));
        const V hi = O::lead_ones(O::shr(mfin[j], 16), CH::bits(j));
        const V n = O::lshl_or(hi, 16, lo);
        total = O::mad(open, n, total);
        if (j + 1 < CH::NM) open = O::mul(open, O::sub_sat(O::add(n, O::splat(ONE2)), O::splat((uint32_t)CH::bits(j) * ONE2)));  // stays 1 iff n == bits(j)
    }
    return total;
}

// H and the flags after the lazy-F loop (striped.rs:528-553 in closed form, see the header).
template <class O, int N, int NV, bool FLAGS>
static ZSW_PK_FN void fixup(State<O, NV>& st, typename O::V Pa, typename O::V Pb, typename O::V vb2, const Consts<O, NV>& c,
                         typename O::V (&flg)[NV]) {
    using V = typename O::V;
    const V one = c.one;
    const V Dp = O::sub(Pa, Pb);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const V t = O::min_u(O::sub_sat(vb2, O::splat((uint32_t)v * ONE2)), one);  // 1 where v < vb
        const V M = O::sub_sat(O::mad(t, Dp, Pb), vge2<O>(v, c.ge));
        const V hm = st.H[v];
        const V h = O::max_u(hm, M);
        st.H[v] = h;
        if constexpr (FLAGS) {
            const V rs = O::sub_sat(O::min_u(M, one), O::sub(h, M));  // 1 where M >= Hmain (h == M) and M > 0
            V f = flg[v];
            const V fr = O::and_or(f, O::splat(UPX2), O::splat(LEFT2));
            f = O::mad(rs, O::sub(fr, f), f);
            const V lx = O::min_u(O::sub_sat(O::sub_sat(M, c.ge2), O::sub_sat(h, c.go2)), one);
            flg[v] = O::lshl_or(lx, 3, f);
        }
    }
}

// Rows whose flags are not kept: H = max(Hmain, F carried in from every earlier segment), by a log-step scan over the lanes.
template <class O, int N, int NV>
static ZSW_PK_FN void lazy_scan(State<O, NV>& st, typename O::V Fend, const Consts<O, NV>& c) {
    using V = typename O::V;
    V X = O::template shr1<N>(Fend, c.keep);
    V dec = c.nvge2;
#pragma unroll
    for (int d = 1; d < N; d *= 2) {
        X = O::max_u(X, O::sub_sat(O::template shr_d<N>(X, d), dec));
        dec = O::add_sat(dec, dec);
    }
    if (!O::any(X)) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) st.H[v] = O::max_u(st.H[v], O::sub_sat(X, vge2<O>(v, c.ge)));
}

// One reference row. act2: 0xffff in the halves of reads that still run (r <= r_end). `next_row()` is called once the row's
// profile scores are dead (the kernel reloads p[] with the next row's there, so that they arrive under the fix-up).
template <class O, int N, int NV, bool FLAGS, class Next>
static ZSW_PK_FN void row(State<O, NV>& st, const typename O::V (&p)[NV], typename O::V act2, const Consts<O, NV>& c,
                          typename O::V (&flg)[NV], Next&& next_row) {
    using V = typename O::V;
    V Fend;
    main_pass<O, N, NV, FLAGS>(st, p, c, Fend, flg);
    Fend = O::and_(Fend, act2);
    if constexpr (FLAGS) {
        V Pa, Pb, mfin[Chunks<NV>::NM];
        lazy_rounds<O, N, NV>(st, Fend, c, Pa, Pb, mfin);
        next_row();
        if (O::any(Pa)) fixup<O, N, NV, true>(st, Pa, Pb, visited<O, NV>(mfin), c, flg);
    } else {
        next_row();
        lazy_scan<O, N, NV>(st, Fend, c);
    }
}

}  // namespace zsw_pk
