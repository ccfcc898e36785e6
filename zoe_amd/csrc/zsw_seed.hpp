// zsw_seed.hpp — arithmetic of the seeded exact score pass (zsw_score_seed.hip), shared with its host model
// (tests/models/seed_bounds.cpp compiles this header with g++ and checks every claim below against the full Gotoh matrix).
//
// sw_simd_score returns only the maximum of the DP matrix (striped.rs:65-142). The seeded pass computes a WINDOW of reference
// rows [a0, b1) of that matrix (all query columns, zero state above row a0) and proves, per read, that no alignment path with a
// cell outside the window can score more than the window's maximum S'. The proof uses k-mers sampled from the read:
//
//   * every query column c can add at most Wp[c] = max(0, max_x w[x][q_c]) to a path, so a path spanning columns [cs, ce)
//     scores at most the sum of Wp over the span (gaps only cost);
//   * "good" residues are those with w[q][q] == maxw; a sampled k-mer (K consecutive query columns, all good) that a path
//     spans completely is either aligned diagonally, without a gap, to an identical piece of the reference (an exact
//     occurrence), or costs the path at least lambda of that potential: a mismatch costs maxw - w[x][q] >= lambda1, a deletion
//     inside it gap_open, an inserted column maxw + gap_open (run starts in the k-mer's territory) or (s + 1) * (maxw +
//     gap_extend) (run started before the s spacer columns in front of the k-mer, which belong to no other k-mer) — provided
//     those columns have the potential maxw: seed_read sums what the columns in front of each k-mer really lose to a run
//     (their own potential + gap_extend each) and does not use a k-mer for which that stays below lambda;
//   * the reference index holds, per k-mer, the first and last position at which it occurs, so "no exact occurrence on a
//     diagonal left of the window" and "none below the window" are two comparisons per k-mer.
//
// Paths with a cell outside the window are of three kinds, bounded separately (seed_bounds):
//   above  the path starts in a row < a0. Its first cell lies on a diagonal (row - column) <= a0 - 1. Either it never reaches
//          diagonal dq = dt - Dn (dt: the read's anchor diagonal) — then every exact occurrence it uses has a diagonal < dq, and
//          it loses lambda per sampled k-mer without such an occurrence (maximised over the columns it may span: U_fa) — or it
//          does, which takes deletions of total length >= dq - a0 + 1: T_all - gap_open - (dq - a0) * gap_extend;
//   below  the path starts in a row >= b1: exact occurrences at positions >= b1 only (U_below);
//   exit   the path starts inside and leaves through the last computed rows: bounded from the kernel's final H / E / F state
//          plus the potential of the columns to the right (the V3 check of the column-pruned pass).
// If all three are <= S' (< S' when the ends are wanted), S' is the read's score; otherwise the read is scored over all its cells.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __HIPCC__
#define ZSW_SEED_HD __host__ __device__ __forceinline__
#else
#define ZSW_SEED_HD inline
#endif

namespace zsw {

constexpr int SEED_MAX_KMERS = 16;
constexpr int SEED_WILD_MAX = 3;      // reference windows with up to this many non-good residues are indexed under every spelling
constexpr uint32_t SEED_NONE = 0xffffffffu;

struct SeedParams {
    int K;            // k-mer length
    int maxw;         // largest weight of the matrix
    int lambda;       // least loss of a sampled k-mer that is spanned but not traversed exactly (before the spacer term)
    int ins_col;      // maxw + gap_extend: loss of one query column inside a running insertion
    int go, ge;       // positive magnitudes
    int spacer;       // columns left free between two sampled k-mers, at least (so that an insertion that runs into a k-mer from
                      // before its spacer costs lambda by the time it gets there)
    int M1, M1_per8;  // window rows above the first row of the anchor diagonal: M1 + len * M1_per8 / 8
    int M2;           // and below its last row
    int Dn;           // diagonals left of the anchor that still count as "near" (the read's own insertions)
    int Dm;           // and right of it (the read's own deletions) — banded pass
    int Wd, Wd_per16; // banded pass: diagonals kept below the anchor, Wd + len * Wd_per16 / 16 (above it: the rows of seed_rows_above)
    int tol;          // anchor vote: k-mers within this many diagonals support each other
    uint8_t code[32]; // residue index -> 2-bit code; 0xff: not a good residue
    uint8_t wp[32];   // residue index of the READ -> potential of a column holding it
};

// Sampled k-mers of a read of `len` bases: m of them at columns c0 + j * stride, stride >= K + spacer.
ZSW_SEED_HD void seed_layout(int len, int K, int spacer, int* m, int* stride, int* c0) {
    int mm = len / (K + spacer);
    if (mm > SEED_MAX_KMERS) mm = SEED_MAX_KMERS;
    if (len < K) mm = 0;
    else if (mm < 1) mm = 1;
    *m = mm;
    *stride = mm ? len / mm : 0;
    *c0 = mm ? (*stride - K) / 2 : 0;
}

// lambda of a read whose sampled k-mers are `stride` columns apart
ZSW_SEED_HD int seed_lambda(const SeedParams& p, int stride) {
    const int spacer = (stride - p.K + 1) * p.ins_col;
    return p.lambda < spacer ? p.lambda : spacer;
}

// Anchor diagonal: the candidate (first / last occurrence of a k-mer, as a diagonal) that most k-mers agree with. Candidates
// come from every k-mer when there are at most 8, from every other one above that (the anchor only decides which rows are
// computed, never whether a result is accepted). has[j]: k-mer j is usable and occurs; dlo / dhi: diagonals of its first / last
// occurrence. Returns the support. Loops of fixed length over arrays of SEED_MAX_KMERS: registers, not scratch memory.
ZSW_SEED_HD int seed_vote(int m, const bool* has, const int* dlo, const int* dhi, int tol, int* dt) {
    int best = 0, bd = 0;
    const int step = m > 8 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < 2 * SEED_MAX_KMERS; ++i) {
        const int j = i >> 1;
        if (j >= m || !has[j] || (j % step) != 0 || ((i & 1) && dhi[j] == dlo[j])) continue;
        const int d = (i & 1) ? dhi[j] : dlo[j];
        int sup = 0;
#pragma unroll
        for (int k = 0; k < SEED_MAX_KMERS; ++k) {
            if (k >= m || !has[k]) continue;
            const int a = dlo[k] - d, b = dhi[k] - d;
            if ((a <= tol && a >= -tol) || (b <= tol && b >= -tol)) ++sup;
        }
        if (sup > best) {
            best = sup;
            bd = d;
        }
    }
    *dt = bd;
    return best;
}

// max over column spans [L, Rr) of  potential(span) - lambda * #{j in `set` : k-mer j inside the span}.
ZSW_SEED_HD int seed_span_bound(int m, const int* pot_lo, const int* pot_hi, int t_all, const bool* set, int lambda) {
    // left cut a: 0 -> column 0 (k-mers a.. included); a = i + 1 -> column c_i + 1 (k-mer i excluded). pot_lo[i] = potential of
    // columns [0, c_i + 1). right cut b: m -> column len; b = i -> column c_i + K - 1 exclusive (k-mer i excluded). pot_hi[i] =
    // potential of columns [0, c_i + K - 1). With left cut a and right cut b the k-mers a .. b-1 are inside.
    // best = max_b ( hi(b) - lambda * cnt[b] - min_{a <= b} ( lo(a) - lambda * cnt[a] ) ), one sweep over b.
    int best = 0, cnt = 0;
    int low = 0;  // min over a <= b of lo(a) - lambda * cnt[a]; a = 0: lo = 0, cnt = 0
#pragma unroll
    for (int b = 0; b <= SEED_MAX_KMERS; ++b) {
        if (b > m) continue;
        // a = b enters: left cut just right of k-mer b-1's first column
        if (b > 0) {
            const int la = pot_lo[b - 1] - lambda * cnt;
            low = la < low ? la : low;
        }
        const int hi = b == m ? t_all : pot_hi[b < SEED_MAX_KMERS ? b : 0];
        const int v = hi - lambda * cnt - low;
        best = v > best ? v : best;
        if (b < m && set[b < SEED_MAX_KMERS ? b : 0]) ++cnt;
    }
    return best;
}

// Exit bound. A path that leaves the computed rows downwards continues in rows >= b1, where the k-mers of `set` (bit j: k-mer j
// has no occurrence down there) cost lambda each. q[i], i in [0, m]: the most such a path can add from column c_i on (c_m = len),
// every column worth maxw at most, with the right end of the path free: q[m] = 0, q[i] = max(stop before k-mer i ends, go on).
ZSW_SEED_HD void seed_suffix_q(int m, int K, int c0, int stride, int len, int maxw, uint32_t set, int lambda, int* q /* m + 1 */) {
    q[m] = 0;
    for (int i = m - 1; i >= 0; --i) {
        const int ci = c0 + i * stride, cn = i + 1 < m ? ci + stride : len;
        const int on = maxw * (cn - ci) - (((set >> i) & 1u) ? lambda : 0) + q[i + 1];
        const int stop = maxw * (K - 1);
        q[i] = on > stop ? on : stop;
    }
}

// v: value (H, or the E / F already inside a gap) of the last computed cell, in column x; the path consumes columns > x afterwards
ZSW_SEED_HD int seed_exit_bound(int v, int x, int m, int c0, int stride, int len, int maxw, const int* q) {
    int i = 0;  // first sampled k-mer that lies completely right of column x
    if (m > 0 && x >= c0) {
        i = (x - c0) / stride + 1;
        if (i > m) i = m;
    }
    const int ci = i < m ? c0 + i * stride : len;
    return v + maxw * (ci - 1 - x) + q[i];
}

// window rows kept above the first row of a read's anchor diagonal: what a path pays to come down to the anchor from above the
// window (gap_open + (rows - Dn) * gap_extend) should exceed what a read of this length loses to its own errors
ZSW_SEED_HD int seed_rows_above(const SeedParams& p, int len) { return p.M1 + len * p.M1_per8 / 8; }
// banded pass: diagonals kept below the anchor diagonal. Leaving the band downwards and coming back costs a deletion and an
// insertion of about this many positions; that, too, should exceed what a read of this length loses to its own errors
ZSW_SEED_HD int seed_rows_below(const SeedParams& p, int len) { return p.Wd + len * p.Wd_per16 / 16; }

struct SeedBounds {
    int above, below;  // -1: no such path exists (the window touches that end of the reference)
};

// a0 / b1: first computed row / first row below the computed rows (of the lane group the read ran in); dt: the read's anchor
// diagonal; d_fa = T_all - U_fa, d_bl = T_all - U_below as stored by the seed kernel.
ZSW_SEED_HD SeedBounds seed_bounds(const SeedParams& p, int t_all, int d_fa, int d_bl, int dt, int a0, int b1, int ref_len) {
    SeedBounds r;
    r.above = -1;
    r.below = -1;
    if (a0 > 0) {
        const int need = dt - p.Dn - a0 + 1;  // deleted reference rows it takes to reach diagonal dt - Dn from above the window
        const int gm = need >= 1 ? p.go + (need - 1) * p.ge : 0;
        const int cut = d_fa < gm ? d_fa : gm;
        r.above = t_all - cut;
    }
    if (b1 < ref_len) r.below = t_all - d_bl;
    return r;
}

// What the seed kernel derives from one read. res(c): residue index of query column c; look(code, &first1, &last1): the index
// entry of a k-mer (positions + 1; 0 = the k-mer does not occur).
struct SeedRead {
    int ok;     // an anchor was found
    int dt;     // anchor diagonal (reference row - query column)
    int t_all;  // potential of all columns
    int d_fa;   // t_all - (bound of the paths that stay left of diagonal dt - Dn), capped at 255
    int d_bl;   // t_all - (bound of the paths below row dt + len + M2), capped at 255
    uint32_t bl_mask;  // bit j: sampled k-mer j is usable and does not occur at or below row dt + len + M2 (the exit bound's set)
    // banded pass (seed_band_kernel): the same by diagonals — fa: no occurrence on a diagonal < dt - Dn (d_fa above), fb: none on a
    // diagonal > dt + Dm
    int d_fb;          // t_all - (bound of the paths that stay right of diagonal dt + Dm), capped at 255
    uint32_t fa_mask, fb_mask;
};

// cell(c): what the seed kernel needs of query column c: potential Wp in bits 0-7, 2-bit code in bits 8-15 (0xff: the residue is
// not a good one). look(code, &first1, &last1): the index entry of a k-mer.
ZSW_SEED_HD uint32_t seed_cell(const SeedParams& p, int residue) { return (uint32_t)p.wp[residue & 31] | ((uint32_t)p.code[residue & 31] << 8); }

template <class GetCell, class Lookup>
ZSW_SEED_HD SeedRead seed_read(const SeedParams& p, int len, GetCell cell, Lookup look) {
    SeedRead out;
    out.ok = 0;
    out.dt = 0;
    out.t_all = 0;
    out.d_fa = 0;
    out.d_bl = 0;
    out.bl_mask = 0;
    out.d_fb = 0;
    out.fa_mask = out.fb_mask = 0;
    int m, stride, c0;
    seed_layout(len, p.K, p.spacer, &m, &stride, &c0);
    // one sweep over the columns, k-mer by k-mer (the loops over the k-mers have a fixed trip count: their arrays stay in registers)
    int pot_lo[SEED_MAX_KMERS], pot_hi[SEED_MAX_KMERS], dlo[SEED_MAX_KMERS], dhi[SEED_MAX_KMERS], last[SEED_MAX_KMERS];
    bool usable[SEED_MAX_KMERS], has[SEED_MAX_KMERS];
    uint32_t codes[SEED_MAX_KMERS];
    int pot = 0, c = 0;
    const int lam = seed_lambda(p, stride);
#pragma unroll
    for (int j = 0; j < SEED_MAX_KMERS; ++j) {
        pot_lo[j] = pot_hi[j] = dlo[j] = dhi[j] = last[j] = 0;
        usable[j] = has[j] = false;
        codes[j] = 0;
        if (j >= m) continue;
        const int cj = c0 + j * stride;
        // An insertion run that opened inside the previous k-mer (which charged it lambda for the opening) reaches this k-mer by
        // swallowing the columns between the two: each of them loses its OWN potential + gap_extend — nothing but gap_extend if
        // its residue has no potential (N) — and this k-mer's first column loses maxw + gap_extend. Only if that comes to lambda
        // does a path that spans the k-mer without traversing it exactly pay lambda for it on top of the previous one's.
        int run = p.ins_col;
        for (; c < cj; ++c) {
            const int wpc = (int)(cell(c) & 0xffu);
            pot += wpc;
            run += wpc + p.ge;
        }
        uint32_t code = 0;
        bool ok = j == 0 || run >= lam;
        for (int k = 0; k < p.K; ++k, ++c) {
            const uint32_t x = cell(c);
            pot += (int)(x & 0xffu);
            ok = ok && (x >> 8) != 0xffu;
            code |= ((x >> 8) & 3u) << (2 * k);
            if (k == 0) pot_lo[j] = pot;
            if (k == p.K - 2) pot_hi[j] = pot;
        }
        usable[j] = ok;
        codes[j] = ok ? code : 0u;
    }
    for (; c < len; ++c) pot += (int)(cell(c) & 0xffu);
    // the index entries of all sampled k-mers: sixteen independent loads in flight together (on the GPU each is a trip to L2;
    // entry 0 stands in for the k-mers that are not looked up, and is not used)
    uint32_t f1s[SEED_MAX_KMERS], l1s[SEED_MAX_KMERS];
#pragma unroll
    for (int j = 0; j < SEED_MAX_KMERS; ++j) {
        f1s[j] = l1s[j] = 0;
        look(codes[j], &f1s[j], &l1s[j]);
    }
#pragma unroll
    for (int j = 0; j < SEED_MAX_KMERS; ++j) {
        if (j < m && usable[j] && f1s[j] != 0) {
            const int cj = c0 + j * stride;
            has[j] = true;
            dlo[j] = (int)(f1s[j] - 1) - cj;
            dhi[j] = (int)(l1s[j] - 1) - cj;
            last[j] = (int)(l1s[j] - 1);
        }
    }
    out.t_all = pot;
    if (m == 0) return out;
    int dt = 0;
    const int support = seed_vote(m, has, dlo, dhi, p.tol, &dt);
    if (support < (m >= 3 ? 2 : 1)) return out;
    bool fa[SEED_MAX_KMERS], bl[SEED_MAX_KMERS], fb[SEED_MAX_KMERS];
#pragma unroll
    for (int i = 0; i < SEED_MAX_KMERS; ++i) {
        fa[i] = i < m && usable[i] && (!has[i] || dlo[i] >= dt - p.Dn);
        bl[i] = i < m && usable[i] && (!has[i] || last[i] < dt + len + p.M2);
        fb[i] = i < m && usable[i] && (!has[i] || dhi[i] <= dt + p.Dm);
        out.bl_mask |= bl[i] ? 1u << i : 0u;
        out.fa_mask |= fa[i] ? 1u << i : 0u;
        out.fb_mask |= fb[i] ? 1u << i : 0u;
    }
    const int u_fa = seed_span_bound(m, pot_lo, pot_hi, pot, fa, lam);
    const int u_bl = seed_span_bound(m, pot_lo, pot_hi, pot, bl, lam);
    out.ok = 1;
    out.dt = dt;
    out.d_fa = pot - u_fa > 255 ? 255 : pot - u_fa;
    out.d_bl = pot - u_bl > 255 ? 255 : pot - u_bl;
    const int u_fb = seed_span_bound(m, pot_lo, pot_hi, pot, fb, lam);
    out.d_fb = pot - u_fb > 255 ? 255 : pot - u_fb;
    return out;
}

// ---- banded pass (seed_band_kernel): the computed cells are a band of diagonals around the anchor, strip by strip ----------
// Strip k holds query columns [kC, (k+1)C) and the reference rows [top_k, bot_k) = [dt + kC - Wu, dt + (k+1)C + Wd) (clamped to
// the reference; dt = the smaller / larger anchor of the lane's two reads). Cells of a strip's columns in rows < top_k lie ABOVE
// the band (on diagonals < dt - Wu <= dt - Dn), cells in rows >= bot_k BELOW it (diagonals > dt + Wd >= dt + Dm). A path changes
// region only in these ways: above -> band through a strip's first row (vertical or diagonal step), below -> band through a
// strip's first column in rows >= bot_{k-1} (horizontal or diagonal step), band -> above through the right edge of strip k in
// rows < top_{k+1}, band -> below through a strip's last row.
//
// The banded pass is ONE dynamic programme over the band whose inputs from outside are not zero but UPPER BOUNDS of what any
// path can hold in the outside cell it comes from ("injection"), every score doubled and an injected value made odd:
//   * with all inputs >= the truth the recurrence (max and + only) keeps every cell >= the truth: U(cell) >= H(cell);
//   * even values descend from the zero floor through even weights only: they are scores of real paths inside the band;
//   * so if the band's maximum M is even, M / 2 is the score of a real path and no path that ever touched an outside cell and
//     ends inside the band scores more (injected = 2 * bound - 1: score-only calls) or as much (2 * bound + 1: calls that want
//     the ends, where an equal score elsewhere could end in an earlier row or column);
//   * paths that END outside are bounded by the same outside bounds (oa / ob below) and compared with M / 2 the same way.
// The outside bounds are two small dynamic programmes along the query columns, one per side (SeedColDP), run strip by strip
// between the band's strips: a(c) / b(c) = the most any path can hold in a cell above / below the band after consuming column c.
//   * A path above the band stays left of diagonal dt - Dn: every sampled k-mer of fa_mask (no occurrence out there) that it
//     spans completely costs it lambda; below the band: fb_mask, right of dt + Dm. Each column adds at most maxw.
//   * Two tracks: `vch` will pay for the next k-mer of the mask that ends, `vfr` will not (paths that started, or arrived from the
//     band, inside that k-mer — or, above the band, within `spacer` columns before it in an open insertion run, which swallows the
//     k-mer's first column for less than lambda); at the k-mer's last column vch pays, the tracks merge.
//   * What leaves the band joins them: above, the largest value (H or outgoing F) on strip k's right edge in rows < top_{k+1},
//     after column xl; below, the H of every cell of strip k's last row, after its own column (into the free track if that
//     column lies inside a k-mer of the mask: the path has taken part of it in the band).
// Strip k + 1 receives a(c) above its columns (E: minus gap_open) and b(xl) left of its first column in rows >= bot_k (as F: the
// bound may have stood lambda - maxw higher just before a k-mer's last column, minus gap_open). The bounds are exact in the
// k-mer charges column by column — a value that enters a strip's corner far from the anchor and leaves it again through the
// right edge must not gain more than a path outside would, or the charges would be lost strip after strip.
// The band's width is no longer the read's error budget: an injected path must still PAY the gap back to the anchor's diagonals
// inside the band, and it competes there with the band's own paths, which have the read's real errors — not with the potential
// of all columns. Host model: tests/models/seed_band.cpp (every cell of the band and every bound against the full Gotoh matrix).

// floor(x / d) for 0 <= x < 4095 * d with x * d < 2^20 (columns of reads of up to 2,432 bases + a strip, strides of len / 16 at most)
ZSW_SEED_HD uint32_t seed_div_magic(int d) { return (1u << 20) / (uint32_t)(d > 0 ? d : 1) + 1u; }
ZSW_SEED_HD int seed_div(int x, uint32_t magic) { return (int)(((uint32_t)x * magic) >> 20); }
// sampled k-mers whose first column is <= x
ZSW_SEED_HD int seed_started(int x, int m, int c0, uint32_t magic) {
    if (x < c0) return 0;
    const int n = seed_div(x - c0, magic) + 1;
    return n < m ? n : m;
}

constexpr int SEED_COL_NONE = -(1 << 20);

struct SeedColDP {  // one side of one read: the most a path outside the band holds after the current column
    int vch, vfr;   // pays / does not pay for the next k-mer of the mask that ends (vfr < 0: no such path)
};
// fresh = false (host model only): no path starts outside the band — what is left is the class of paths that came from the band
ZSW_SEED_HD void seed_col_init(SeedColDP* s, bool fresh = true) {
    s->vch = fresh ? 0 : SEED_COL_NONE;
    s->vfr = SEED_COL_NONE;
}
// one column: `start` / `end` = a k-mer of the mask has its first / last column here. Returns the bound after the column.
ZSW_SEED_HD int seed_col_step(SeedColDP* s, int maxw, int lam, bool start, bool end, bool fresh = true) {
    s->vch += maxw;
    s->vfr += maxw;
    if (start && fresh) s->vfr = s->vfr > 0 ? s->vfr : 0;  // a path that starts behind the k-mer's first column does not span it
    if (end) {
        s->vch -= lam;
        s->vch = s->vfr > s->vch ? s->vfr : s->vch;
        s->vfr = SEED_COL_NONE;
    }
    return s->vch > s->vfr ? s->vch : s->vfr;
}
// a value leaving the band joins the paths outside
ZSW_SEED_HD void seed_col_join(SeedColDP* s, int v, bool free_track) {
    if (free_track) s->vfr = v > s->vfr ? v : s->vfr;
    else s->vch = v > s->vch ? v : s->vch;
}
// the events of a strip's columns [kC, kC + C): bit i = a k-mer of `mask` starts / ends in column kC + i. T = uint32_t for strips of
// up to 32 columns, uint64_t for up to 64
template <class T>
struct SeedStripEventsT {
    T start, end;
    T inside;  // bit i: column kC + i lies in a k-mer of the mask, but is not its last column
};
template <class T>
ZSW_SEED_HD SeedStripEventsT<T> seed_strip_events_t(int kC, int C, int m, int c0, int stride, int K, uint32_t magic, uint32_t mask) {
    constexpr int BITS = (int)sizeof(T) * 8;
    SeedStripEventsT<T> e;
    e.start = e.end = e.inside = 0;
    for (int j = seed_started(kC - 1, m, c0, magic); j < m; ++j) {
        const int cj = c0 + j * stride;
        if (cj >= kC + C) break;
        if ((mask >> j) & 1u) e.start |= (T)1 << (cj - kC);
    }
    for (int j = seed_started(kC - K, m, c0, magic); j < m; ++j) {  // the first k-mer whose last column is not left of the strip
        const int cj = c0 + j * stride, ej = cj + K - 1;
        if (cj >= kC + C) break;
        if (!((mask >> j) & 1u)) continue;
        if (ej < kC + C) e.end |= (T)1 << (ej - kC);
        const int lo = cj > kC ? cj - kC : 0, hi = ej < kC + C ? ej - kC : C;  // columns [lo, hi) of the strip
        if (hi > lo) e.inside |= (hi >= BITS ? ~(T)0 : (((T)1 << hi) - 1u)) & ~(((T)1 << lo) - 1u);
    }
    return e;
}
using SeedStripEvents = SeedStripEventsT<uint32_t>;
ZSW_SEED_HD SeedStripEvents seed_strip_events(int kC, int C, int m, int c0, int stride, int K, uint32_t magic, uint32_t mask) {
    return seed_strip_events_t<uint32_t>(kC, C, m, c0, stride, K, magic, mask);
}
// A value leaves the band above it after column x, possibly as an open insertion run: it does not pay for a k-mer of the mask
// whose territory (the `spacer` columns before it and all but its last column) holds column x.
ZSW_SEED_HD bool seed_exit_is_free(int x, int m, int c0, int stride, int K, int spacer, uint32_t magic, uint32_t mask) {
    const int j = seed_started(x - K + 1, m, c0, magic);  // the first k-mer whose last column lies behind x
    if (j >= m || !((mask >> j) & 1u)) return false;
    return c0 + j * stride - spacer <= x;
}

// Doubled scores with the mark of a path that touched a cell outside the band: TAG = -1 (score only: such a path must score
// MORE than the band's own to matter) or +1 (ends: as much). A bound of 0 stays the plain zero floor.
ZSW_SEED_HD uint32_t seed_tag(int v, int tag) { return v > 0 ? (uint32_t)(2 * v + tag) : 0u; }
// the plain bound a doubled value stands for
ZSW_SEED_HD int seed_untag(uint32_t v2, int tag) { return (int)((v2 + (tag < 0 ? 1u : 0u)) >> 1); }

// Certificate for the late start of the alignment's second pass (sw_simd_align's flags, striped.rs:449-598, recomputed for the
// rows the traceback can visit). Every quantity of the striped recurrence at a cell (H, the E entering it, the F of a lazy-F
// round) is a maximum over alignment paths ending there; starting the recompute at row r0 with a zero state drops the paths
// that start above r0. Such a path, continued by the remainder of the optimal path after a cell x the traceback visits, is a
// valid path that scores at least (its value at x) + S - (optimal path's value at x) - (gap_open - gap_extend) (the remainder
// may have to open a gap the optimal path was merely extending), and at most B_above(r0). A flag of x differs only if a
// dropped path reaches H(x) - gap_open + gap_extend + 1 or more there (the lowest of the thresholds E == H, F == H,
// E - ge > H - go, F - ge > H - go), hence not while B_above(r0) <= S - 2 * gap_open - 1; the same bound keeps every
// dropped path below S in the last row, where the walk finds its first column.
// Returns the largest such r0 (>= 0), or -1 if the read's k-mers do not allow one (the caller then uses the warm-up bound of
// zsw_align_dev.hpp). t_all, d_fa, dt: the read's SeedRead values; S: its score.
ZSW_SEED_HD int seed_safe_start(const SeedParams& p, int t_all, int d_fa, int dt, int S) {
    const int x = t_all - S + 2 * p.go + 1;  // what both terms of B_above must take off t_all
    if (d_fa < x) return -1;
    int rows = 0;                            // dt - Dn - r0 + 1 >= 1 deleted rows, so that gap_open + (rows - 1) * gap_extend >= x
    if (x > p.go) {
        if (p.ge <= 0) return -1;
        rows = (x - p.go + p.ge - 1) / p.ge;
    }
    const int r0 = dt - p.Dn - rows;
    return r0 < 0 ? 0 : r0;
}

// ---- host side: what the matrix allows, and the reference index ------------------------------------------------------------

// Fills p->maxw / lambda / ins_col / go / ge / code / wp from the S x S matrix (row = reference residue). `ref_has[x]`: residue
// index x occurs in the reference. Returns false when the seeded pass cannot prune anything for this matrix / reference (no two
// to four good residues, a mismatch that scores maxw, free gaps): the caller then scores every cell.
inline bool seed_analyze(int S, const int32_t* w, int go, int ge, const bool* ref_has, int K, SeedParams* p) {
    if (S < 2 || S > 32) return false;
    int maxw = 0;
    for (int i = 0; i < S * S; ++i) maxw = w[i] > maxw ? w[i] : maxw;
    if (maxw <= 0 || go <= 0) return false;
    int n_good = 0;
    for (int q = 0; q < 32; ++q) {
        p->code[q] = 0xff;
        p->wp[q] = 0;
    }
    for (int q = 0; q < S; ++q) {
        int col_max = 0;
        for (int x = 0; x < S; ++x) col_max = w[x * S + q] > col_max ? w[x * S + q] : col_max;
        p->wp[q] = (uint8_t)(col_max > 255 ? 255 : col_max);
        if (w[q * S + q] == maxw) {
            if (n_good == 4) return false;  // more good residues than two bits spell
            p->code[q] = (uint8_t)n_good++;
        }
    }
    if (n_good < 2) return false;
    int lambda = go < K * maxw ? go : K * maxw;  // a deletion inside the k-mer; and never more than the k-mer can add at all
    for (int q = 0; q < S; ++q) {
        if (p->code[q] == 0xff) continue;
        for (int x = 0; x < S; ++x) {
            if (x == q) continue;
            const int loss = maxw - w[x * S + q];
            if (p->code[x] != 0xff) {
                lambda = loss < lambda ? loss : lambda;  // a mismatch between good residues
            } else if (ref_has[x]) {
                // a reference residue that is not good (N, ...) under a good query residue: windows with up to SEED_WILD_MAX of
                // them are indexed under every spelling, more of them must cost at least lambda together
                const int many = (SEED_WILD_MAX + 1) * loss;
                lambda = many < lambda ? many : lambda;
            }
        }
    }
    if (lambda < 2) return false;
    p->K = K;
    p->maxw = maxw;
    p->lambda = lambda;
    p->ins_col = maxw + ge;
    p->go = go;
    p->ge = ge;
    p->spacer = 2;
    while ((p->spacer + 1) * p->ins_col < lambda) ++p->spacer;
    return true;
}

// k-mer length for a reference of R residues: a random k-mer should occur in it with a probability of a few per cent
inline int seed_k_for(uint64_t R) {
    int K = 8;
    while (K < 12 && (uint64_t(1) << (2 * K)) < 32 * R) ++K;
    return K;
}

// The index: entry[code] = (first position + 1, last position + 1) of the k-mer in the reference, (0, 0) if it does not occur.
// `res[i]` = residue index of reference position i. Windows holding 1..SEED_WILD_MAX non-good residues are entered under every
// spelling of those positions (a path may run through them at the loss of a wildcard, which is less than lambda).
inline void seed_index_build(const SeedParams& p, const uint8_t* res, uint64_t R, uint32_t* table /* 2 * 4^K entries, zeroed */) {
    const int K = p.K;
    if (R < (uint64_t)K) return;
    auto enter = [&](uint32_t code, uint32_t pos) {
        uint32_t* e = table + 2 * (size_t)code;
        if (e[0] == 0 || pos + 1 < e[0]) e[0] = pos + 1;
        if (pos + 1 > e[1]) e[1] = pos + 1;
    };
    for (uint64_t i = 0; i + K <= R; ++i) {
        uint32_t code = 0;
        int wild[SEED_WILD_MAX], nw = 0;
        bool skip = false;
        for (int k = 0; k < K; ++k) {
            const uint8_t c = p.code[res[i + k] & 31];
            if (c == 0xff) {
                if (nw == SEED_WILD_MAX) {
                    skip = true;
                    break;
                }
                wild[nw++] = k;
            } else {
                code |= (uint32_t)c << (2 * k);
            }
        }
        if (skip) continue;
        const int combos = 1 << (2 * nw);
        for (int v = 0; v < combos; ++v) {
            uint32_t cv = code;
            for (int t = 0; t < nw; ++t) cv |= (uint32_t)((v >> (2 * t)) & 3) << (2 * wild[t]);
            enter(cv, (uint32_t)i);
        }
    }
}

}  // namespace zsw
