// seed_bounds.cpp — host model of the seeded exact score pass (zoe_amd/csrc/zsw_score_seed.hip). It compiles the header the
// kernels use (zsw_seed.hpp: k-mer layout, anchor vote, span bounds, the reference index) and checks, with plain integers
// against the full Gotoh matrix, every claim the pass rests on:
//   A  no path that STARTS above the window (row < a0) scores more than seed_bounds().above;
//   B  no path that starts below it (row >= b1) scores more than seed_bounds().below;
//   C  no path that starts inside and leaves through the last row scores more than the exit bound (H / E of the last row plus
//      what the columns to the right can add below the window), unless it only trails a gap from the last column (< S');
//   D  hence: all three <= S' (the window's own maximum)  =>  S' is the read's score; all three < S'  =>  the window's first row
//      and first column of the maximum are the true ones (tie rule of striped.rs:296-321).
// A, B and C are checked for every read whether or not it passes (the true maxima of the three path classes come from DPs
// whose fresh starts are restricted to the class's rows). Reads: copies with substitutions / indels / N, chimeras, repeats and
// tandem repeats in the reference, reads hanging over its ends, random reads; small k so that chance occurrences are common.
// usage: seed_bounds <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../zoe_amd/csrc/zsw_seed.hpp"
#include "adversarial_reads.hpp"

namespace {

using zsw::SeedParams;

struct Scheme {
    int S;
    std::vector<int32_t> w;  // S x S, row = reference residue
    int go, ge;
};

Scheme dna(int match, int mismatch, int n_score, int go, int ge) {
    Scheme s;
    s.S = 5;
    s.w.assign(25, 0);
    for (int x = 0; x < 4; ++x)
        for (int q = 0; q < 4; ++q) s.w[x * 5 + q] = x == q ? match : mismatch;
    for (int x = 0; x < 5; ++x) s.w[x * 5 + 4] = s.w[4 * 5 + x] = n_score;
    s.go = go;
    s.ge = ge;
    return s;
}

constexpr int NEG = -(1 << 28);

struct Region {
    int start_lo, start_hi;  // rows in which a path may start: [start_lo, start_hi)
    int row_lo, row_hi;      // rows the DP covers
};

struct Best {
    int best = 0, row = -1, col = -1;
};

// Gotoh over rows [row_lo, row_hi) where a path may only start in rows [start_lo, start_hi); `count_from`: only cells in rows
// >= count_from enter the maximum. Returns the maximum (0 if no path) and the first row / column holding it.
Best gotoh(const Scheme& s, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const Region& g, int count_from,
           std::vector<int>* last_h = nullptr, std::vector<int>* next_e = nullptr) {
    const int L = (int)q.size();
    std::vector<int> Hprev(L + 1, NEG), Eprev(L + 1, NEG), H(L + 1), E(L + 1);
    Best b;
    for (int r = g.row_lo; r < g.row_hi; ++r) {
        const bool fresh = r >= g.start_lo && r < g.start_hi;
        int f = NEG;
        H[0] = NEG;
        for (int c = 1; c <= L; ++c) {
            E[c] = std::max(Eprev[c] - s.ge, Hprev[c] - s.go);
            f = std::max(f - s.ge, H[c - 1] - s.go);
            // a path may start at this cell (its first aligned pair) if the row allows it: the diagonal predecessor counts as 0
            int diag = Hprev[c - 1];
            if (fresh) diag = std::max(diag, 0);
            int h = diag > NEG / 2 ? diag + s.w[ref[r] * s.S + q[c - 1]] : NEG;
            h = std::max(h, std::max(E[c], f));
            if (h < NEG / 2) h = NEG;
            H[c] = h;
            if (r >= count_from && h > b.best) {
                b.best = h;
                b.row = r;
                b.col = c - 1;
            }
        }
        std::swap(H, Hprev);
        std::swap(E, Eprev);
    }
    if (last_h) *last_h = Hprev;
    if (next_e) {
        next_e->assign(L + 1, NEG);
        for (int c = 1; c <= L; ++c) (*next_e)[c] = std::max(Eprev[c] - s.ge, Hprev[c] - s.go);
    }
    return b;
}

struct Counters {
    long reads = 0, anchored = 0, pass_score = 0, pass_ends = 0, plain = 0, plain_pass = 0;
};

bool check_read(const Scheme& s, const SeedParams& p, const std::vector<uint32_t>& table, const std::vector<uint8_t>& ref,
                const std::vector<uint8_t>& q, bool plain, int extra_top, int extra_bottom, Counters* cnt) {
    const int R = (int)ref.size(), L = (int)q.size();
    ++cnt->reads;
    if (plain) ++cnt->plain;
    const Best truth = gotoh(s, ref, q, Region{0, R, 0, R}, 0);
    auto res = [&](int c) { return zsw::seed_cell(p, (int)q[c]); };
    auto look = [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
        *f1 = table[2 * (size_t)code];
        *l1 = table[2 * (size_t)code + 1];
    };
    const zsw::SeedRead sr = zsw::seed_read(p, L, res, look);
    if (!sr.ok) return true;  // no anchor: the read is scored over all its cells
    ++cnt->anchored;
    int a0 = std::max(0, sr.dt - zsw::seed_rows_above(p, L) - extra_top);
    int b1 = std::min(R, sr.dt + L + p.M2 + extra_bottom);
    if (b1 <= a0) return true;  // anchor outside the reference: handed back
    // the window: zero state above row a0
    std::vector<int> last_h, next_e;
    const Best win = gotoh(s, ref, q, Region{a0, b1, a0, b1}, a0, &last_h, &next_e);
    const zsw::SeedBounds sb = zsw::seed_bounds(p, sr.t_all, sr.d_fa, sr.d_bl, sr.dt, a0, b1, R);
    // exit bound: H / next-row E of the last row plus what the columns to the right can add below the window
    int v3 = -1;
    if (b1 < R) {
        int m, stride, c0, q[zsw::SEED_MAX_KMERS + 1];
        zsw::seed_layout(L, p.K, p.spacer, &m, &stride, &c0);
        zsw::seed_suffix_q(m, p.K, c0, stride, L, p.maxw, sr.bl_mask, zsw::seed_lambda(p, stride), q);
        for (int c = 1; c < L; ++c) {  // from the last column no path goes on below the window except by trailing a gap
            const int he = std::max(std::max(last_h[c], next_e[c]), 0);
            v3 = std::max(v3, zsw::seed_exit_bound(he, c - 1, m, c0, stride, L, p.maxw, q));
        }
    }
    // the three path classes, exactly
    const int true_above = a0 > 0 ? gotoh(s, ref, q, Region{0, a0, 0, R}, 0).best : 0;
    const int true_below = b1 < R ? gotoh(s, ref, q, Region{b1, R, b1, R}, b1).best : 0;
    const int true_exit = b1 < R ? gotoh(s, ref, q, Region{a0, b1, a0, R}, b1).best : 0;
    bool ok = true;
    if (a0 > 0 && true_above > sb.above) {
        printf("claim A violated: above %d > bound %d (dt %d a0 %d t_all %d d_fa %d)\n", true_above, sb.above, sr.dt, a0, sr.t_all, sr.d_fa);
        ok = false;
    }
    if (b1 < R && true_below > sb.below) {
        printf("claim B violated: below %d > bound %d (dt %d b1 %d t_all %d d_bl %d)\n", true_below, sb.below, sr.dt, b1, sr.t_all, sr.d_bl);
        ok = false;
    }
    if (b1 < R && true_exit > std::max(v3, win.best - 1)) {  // (a path that leaves from the last column only trails a gap: < window max)
        printf("claim C violated: exit %d > bound %d\n", true_exit, v3);
        ok = false;
    }
    if (truth.best != std::max(std::max(win.best, true_above), std::max(true_below, true_exit))) {
        printf("model inconsistency: truth %d vs classes %d %d %d %d\n", truth.best, win.best, true_above, true_below, true_exit);
        ok = false;
    }
    const int bound = std::max(sb.above, std::max(sb.below, v3));
    if (bound <= win.best) {
        ++cnt->pass_score;
        if (plain) ++cnt->plain_pass;
        if (win.best != truth.best) {
            printf("claim D violated (score): window %d, truth %d, bounds %d %d %d\n", win.best, truth.best, sb.above, sb.below, v3);
            ok = false;
        }
    }
    if (bound < win.best) {
        ++cnt->pass_ends;
        if (win.best != truth.best || win.row != truth.row || win.col != truth.col) {
            printf("claim D violated (ends): window %d (%d,%d), truth %d (%d,%d)\n", win.best, win.row, win.col, truth.best, truth.row, truth.col);
            ok = false;
        }
    }
    if (!ok) {
        printf("  ref (%d): ", R);
        for (uint8_t x : ref) putchar("ACGTN"[x]);
        printf("\n  read (%d): ", L);
        for (uint8_t x : q) putchar("ACGTN"[x]);
        printf("\n  K %d M1 %d M2 %d Dn %d tol %d lambda %d go %d ge %d maxw %d\n", p.K, p.M1, p.M2, p.Dn, p.tol, p.lambda, p.go, p.ge, p.maxw);
    }
    return ok;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    const Scheme schemes[] = {dna(2, -5, 0, 10, 1), dna(1, -1, 0, 2, 1), dna(3, -2, 0, 5, 0), dna(1, -3, 0, 5, 2), dna(5, -4, 0, 8, 0),
                              dna(2, -5, -1, 10, 1), dna(4, -6, 1, 12, 2), dna(2, -2, 0, 3, 3),
                              // mismatch loss >= gap_open: lambda = gap_open, an insertion run is the cheapest way through a k-mer
                              dna(2, -10, 0, 10, 1), dna(2, -5, 0, 5, 1), dna(3, -9, 0, 6, 1), dna(2, -10, 0, 10, 0)};
    Counters cnt;
    long adversarial_reads = 0;
    bool all_ok = true;
    for (int it = 0; it < iters && all_ok; ++it) {
        const Scheme& s = schemes[it % (sizeof(schemes) / sizeof(schemes[0]))];
        // reference: random, with a duplicated segment, a tandem repeat and a few N now and then
        const int R = rnd(60, 420);
        std::vector<uint8_t> ref(R);
        for (auto& x : ref) x = (uint8_t)rnd(0, 3);
        if (rnd(0, 2) == 0 && R > 120) {  // a second copy of a segment
            const int len = rnd(20, 50), from = rnd(0, R - len), to = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[to + i] = ref[from + i];
        }
        if (rnd(0, 3) == 0 && R > 100) {  // tandem repeat
            const int unit = rnd(1, 6), len = rnd(20, 60), at = rnd(0, R - len);
            for (int i = unit; i < len; ++i) ref[at + i] = ref[at + i - unit];
        }
        if (rnd(0, 3) == 0)
            for (int k = rnd(1, 6); k > 0; --k) ref[rnd(0, R - 1)] = 4;
        if (rnd(0, 7) == 0 && R > 80) {  // a run of N
            const int len = rnd(3, 20), at = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[at + i] = 4;
        }
        bool ref_has[32] = {false};
        for (uint8_t x : ref) ref_has[x] = true;
        SeedParams p;
        const int K = rnd(3, 6);
        if (!zsw::seed_analyze(s.S, s.w.data(), s.go, s.ge, ref_has, K, &p)) continue;
        p.M1 = rnd(2, 24);
        p.M1_per8 = rnd(0, 2);
        if (rnd(0, 2) == 0) p.spacer += rnd(0, 6);  // sparser sampling is valid too
        p.M2 = rnd(2, 14);
        p.Dn = rnd(0, 4);
        p.tol = rnd(0, 5);
        std::vector<uint32_t> table((size_t)2 << (2 * K), 0);
        zsw::seed_index_build(p, ref.data(), (uint64_t)R, table.data());
        const int n_reads = 60;
        for (int k = 0; k < n_reads && all_ok; ++k) {
            const int kind = rnd(0, 9);
            const int L = rnd(K, std::min(R, 90));
            std::vector<uint8_t> q;
            bool plain = false;
            auto copy_with_errors = [&](int start, int len, int sub_pct, int indel_pct) {
                int i = start;
                while ((int)q.size() < len) {
                    uint8_t b = (i >= 0 && i < R) ? ref[i] : (uint8_t)rnd(0, 3);
                    const int e = rnd(0, 999);
                    if (e < sub_pct * 10) b = (uint8_t)rnd(0, 3);
                    else if (e < sub_pct * 10 + indel_pct * 5) { ++i; continue; }       // deletion from the read
                    else if (e < sub_pct * 10 + indel_pct * 10) { q.push_back((uint8_t)rnd(0, 3)); continue; }  // insertion
                    q.push_back(b);
                    ++i;
                }
                q.resize(len);
            };
            if (kind <= 4) {  // plain copy, few errors
                copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(0, 3), rnd(0, 1));
                plain = true;
            } else if (kind == 5) {  // many errors
                copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(5, 20), rnd(1, 5));
            } else if (kind == 6) {  // chimera of two places
                const int l1 = rnd(K, std::max(K, L - 1));
                copy_with_errors(rnd(0, std::max(0, R - l1)), l1, 1, 0);
                copy_with_errors(rnd(0, std::max(0, R - L)), L, 1, 0);
            } else if (kind == 7) {  // hanging over an end
                copy_with_errors(rnd(0, 1) ? -rnd(1, L / 2 + 1) : R - rnd(1, L / 2 + 1) - L / 2, L, 1, 0);
            } else if (kind == 8) {  // a long deletion or insertion in the middle
                const int l1 = L / 2, st = rnd(0, std::max(0, R - L - 30));
                copy_with_errors(st, l1, 0, 0);
                if (rnd(0, 1)) {
                    const int skip = rnd(3, 28);
                    int i = st + l1 + skip;
                    while ((int)q.size() < L) q.push_back(i < R ? ref[i++] : (uint8_t)rnd(0, 3));
                } else {
                    for (int x = rnd(3, 20); x > 0 && (int)q.size() < L; --x) q.push_back((uint8_t)rnd(0, 3));
                    int i = st + l1;
                    while ((int)q.size() < L) q.push_back(i < R ? ref[i++] : (uint8_t)rnd(0, 3));
                }
            } else {  // random
                for (int i = 0; i < L; ++i) q.push_back((uint8_t)rnd(0, 3));
            }
            if (rnd(0, 4) == 0)
                for (int x = rnd(1, 3); x > 0; --x) q[rnd(0, L - 1)] = 4;
            all_ok = check_read(s, p, table, ref, q, plain && kind <= 4, rnd(0, 3) ? 0 : rnd(0, 12), rnd(0, 3) ? 0 : rnd(0, 12), &cnt);
        }
        // structured cases (adversarial_reads.hpp): residues without potential between the sampled k-mers, a far copy that skips
        // them and an anchor copy that loses about twice lambda; each with a reference (and index) of its own
        for (int k = 0; k < 24 && all_ok; ++k) {
            std::vector<uint8_t> aref, aq;
            SeedParams pa;
            bool has[32] = {false};
            has[0] = has[1] = has[2] = has[3] = true;
            if (!zsw::seed_analyze(s.S, s.w.data(), s.go, s.ge, has, K, &pa)) break;
            pa.M1 = p.M1;
            pa.M1_per8 = p.M1_per8;
            pa.M2 = p.M2;
            pa.Dn = p.Dn;
            pa.tol = p.tol;
            if (rnd(0, 3) == 0) pa.spacer += rnd(0, 3);
            if (!adversarial::spacer_case(rng, pa, rnd(2 * (K + pa.spacer), 96), &aref, &aq)) continue;
            std::vector<uint32_t> atable((size_t)2 << (2 * K), 0);
            zsw::seed_index_build(pa, aref.data(), (uint64_t)aref.size(), atable.data());
            Counters unused;
            all_ok = check_read(s, pa, atable, aref, aq, false, 0, 0, &unused);
            ++adversarial_reads;
        }
    }
    printf("reads %ld, anchored %ld, passed (score) %ld, passed (ends) %ld; plain reads %ld, of which passed %ld; structured cases %ld\n", cnt.reads,
           cnt.anchored, cnt.pass_score, cnt.pass_ends, cnt.plain, cnt.plain_pass, adversarial_reads);
    if (!all_ok) return 1;
    if (cnt.plain > 200 && cnt.plain_pass * 4 < cnt.plain) {
        printf("the checks are vacuous: fewer than a quarter of the plain reads pass\n");
        return 1;
    }
    printf("seed_bounds OK\n");
    return 0;
}
