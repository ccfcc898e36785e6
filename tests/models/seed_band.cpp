// seed_band.cpp — host model of the banded seeded pass (zoe_amd/csrc/zsw_score_band.hip): the computed cells are a band of
// diagonals around the read's anchor, strip by strip (strip k: query columns [kC, (k+1)C), reference rows [dt + kC - Wu,
// dt + (k+1)C + Wd)), every input from outside the band taken as zero. The model compiles zsw_seed.hpp (as the kernels do) and
// checks against a two-layer Gotoh DP — layer 0: paths wholly inside the band; layer 1: paths that have touched a cell outside
// it — that
//   * no path of layer 1 scores more than the largest of the bounds (fresh starts above / below the band, exits through the right
//     edge of a strip above the next strip's first row, exits through a strip's last row), for every read, passing or not;
//   * hence a read whose bounds are all <= the band's maximum S' has S' as its score (all < S': also the first row and column);
//   * mode 3 (the shared-profile role, zsw_capi_shared.hip): with all bounds < S' the cells of the whole matrix that hold the
//     maximum are exactly the band's cells that hold S' — so "S' sits in one cell of the band" means "the maximum sits in one cell",
//     and that cell is the answer under the other tie rule (first column, then first row) as well. The kernel's bookkeeping
//     for that flag (per strip: maximum, row of the latest rise, a later row reaching it again, the columns of the snapshot row;
//     strips merged) is restated here and must say exactly whether more than one cell of the band holds S' (dropping the
//     tie events or the cross-strip rule is caught within a hundred iterations).
// Reads, references and schemes as in seed_bounds.cpp. usage: seed_band <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../zoe_amd/csrc/zsw_seed.hpp"

namespace {

using zsw::SeedParams;

struct Scheme {
    int S;
    std::vector<int32_t> w;
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

struct Geometry {
    int C, K, wu, wd, dt, R, L;
    int top(int k) const { return std::max(0, std::min(R, dt + k * C - wu)); }
    int bot(int k) const { return std::max(0, std::min(R, dt + (k + 1) * C + wd)); }
    bool inside(int r, int c) const {  // c: 0-based column
        const int k = c / C;
        return r >= top(k) && r < bot(k);
    }
};

struct Counters {
    long reads = 0, anchored = 0, pass_score = 0, pass_ends = 0, plain = 0, plain_pass = 0, unique = 0;
};

bool check_read(const Scheme& s, const SeedParams& p, const std::vector<uint32_t>& table, const std::vector<uint8_t>& ref,
                const std::vector<uint8_t>& q, bool plain, int C, Counters* cnt) {
    const int R = (int)ref.size(), L = (int)q.size();
    ++cnt->reads;
    if (plain) ++cnt->plain;
    auto cell = [&](int c) { return zsw::seed_cell(p, (int)q[c]); };
    auto look = [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
        *f1 = table[2 * (size_t)code];
        *l1 = table[2 * (size_t)code + 1];
    };
    const zsw::SeedRead sr = zsw::seed_read(p, L, cell, look);
    if (!sr.ok) return true;
    ++cnt->anchored;
    Geometry g;
    g.C = C;
    g.K = (L + C - 1) / C;
    g.wu = zsw::seed_rows_above(p, L);
    g.wd = zsw::seed_rows_below(p, L);
    g.dt = sr.dt;
    g.R = R;
    g.L = L;
    // layered Gotoh. Layer 0: H / E / F of the best path ending here that never left the band. Layers 1-4: the best one that has,
    // by how it first did: FU / FD = it starts outside, above / below the band; XU / XL = it leaves the band from a computed cell
    // into a cell above / below it. Each class has its own bound.
    enum { IN = 0, FU, FD, XU, XL, NL };
    const int W = L + 1;
    std::vector<int> Hs[NL], Es[NL], Fs[NL];
    for (int j = 0; j < NL; ++j) {
        Hs[j].assign((size_t)(R + 1) * W, NEG);
        Es[j] = Hs[j];
        Fs[j] = Hs[j];
    }
    std::vector<int>&H0 = Hs[IN], &E0 = Es[IN], &F0 = Fs[IN];
    auto at = [&](std::vector<int>& v, int r, int c) -> int& { return v[(size_t)r * W + c]; };  // r, c 1-based; 0 = border
    auto clampneg = [](int x) { return x < NEG / 2 ? NEG : x; };
    int best0 = 0, row0 = -1, col0 = -1, best1 = 0, truth = 0, trow = -1, tcol = -1;
    int n_best0 = 0, n_truth = 0, trow2 = -1, tcol2 = -1;  // cells holding the two maxima; the truth under the transposed tie rule
    bool ok_mode3 = true;
    int best_cls[NL] = {0, 0, 0, 0, 0};
    for (int r = 1; r <= R; ++r)
        for (int c = 1; c <= L; ++c) {
            const bool in = g.inside(r - 1, c - 1);
            const bool above = !in && r - 1 < g.top((c - 1) / C);
            const int wgt = s.w[ref[r - 1] * s.S + q[c - 1]];
            int d[NL], e[NL], f[NL];
            for (int j = 0; j < NL; ++j) {
                d[j] = at(Hs[j], r - 1, c - 1);
                e[j] = std::max(at(Es[j], r - 1, c) - s.ge, at(Hs[j], r - 1, c) - s.go);
                f[j] = std::max(at(Fs[j], r, c - 1) - s.ge, at(Hs[j], r, c - 1) - s.go);
            }
            if (in) {
                for (int j = 0; j < NL; ++j) {
                    at(Es[j], r, c) = clampneg(e[j]);
                    at(Fs[j], r, c) = clampneg(f[j]);
                    const int dj = j == IN ? std::max(d[j], 0) + wgt : (d[j] > NEG / 2 ? d[j] + wgt : NEG);  // fresh starts: layer 0
                    at(Hs[j], r, c) = clampneg(std::max(std::max(dj, e[j]), f[j]));
                }
            } else {  // an outside cell: what arrives from layer 0 joins XU / XL here, a fresh start FU / FD
                const int xj = above ? XU : XL, fj = above ? FU : FD;
                e[xj] = std::max(e[xj], e[IN]);
                f[xj] = std::max(f[xj], f[IN]);
                d[xj] = std::max(d[xj], d[IN]);
                d[fj] = std::max(d[fj], 0);
                for (int j = 1; j < NL; ++j) {
                    at(Es[j], r, c) = clampneg(e[j]);
                    at(Fs[j], r, c) = clampneg(f[j]);
                    at(Hs[j], r, c) = clampneg(std::max(std::max(d[j] > NEG / 2 ? d[j] + wgt : NEG, e[j]), f[j]));
                }
            }
            if (at(H0, r, c) > best0) {
                best0 = at(H0, r, c);
                row0 = r - 1;
                col0 = c - 1;
                n_best0 = 1;
            } else if (at(H0, r, c) == best0 && best0 > 0) {
                ++n_best0;
            }
            int h = at(H0, r, c);
            for (int j = 1; j < NL; ++j) {
                best_cls[j] = std::max(best_cls[j], at(Hs[j], r, c));
                best1 = std::max(best1, at(Hs[j], r, c));
                h = std::max(h, at(Hs[j], r, c));
            }
            if (h > truth) {
                truth = h;
                trow = trow2 = r - 1;
                tcol = tcol2 = c - 1;
                n_truth = 1;
            } else if (h == truth && truth > 0) {
                ++n_truth;
                if (c - 1 < tcol2 || (c - 1 == tcol2 && r - 1 < trow2)) {
                    tcol2 = c - 1;
                    trow2 = r - 1;
                }
            }
        }
    // ---- mode 3's bookkeeping as seed_band_kernel<.., 3> keeps it: per strip the maximum, the row of its latest rise, whether a
    // later row reached it again (a tie event), the columns of the snapshot row that hold it; strips merged by (higher, or equal:
    // two cells). `mult` must say exactly whether more than one cell of the band holds S'.
    {
        int bbest = 0, brow = 0x7fffffff;
        bool mult = false;
        for (int k = 0; k < g.K; ++k) {
            const int c_lo = k * C, c_hi = std::min(L, (k + 1) * C);  // real columns (padding columns only ever hold copies)
            int sbest = 0, sr = 0x7fffffff, snaprow = -1;
            bool sm = false;
            for (int r = g.top(k); r < g.bot(k); ++r) {
                int tmax = 0;
                for (int c = c_lo; c < c_hi; ++c) tmax = std::max(tmax, at(H0, r + 1, c + 1));
                const int nsb = std::max(sbest, tmax);
                if (nsb > sbest) {
                    sr = snaprow = r;
                    sm = false;
                } else if (tmax == nsb && nsb > 0) {
                    sm = true;
                }
                sbest = nsb;
            }
            const bool eq = sbest == bbest && sbest > 0, up = sbest > bbest || (eq && sr < brow);
            if (up) {
                bbest = sbest;
                brow = sr;
            }
            if (up || eq) {
                int hits = 0;
                for (int c = c_lo; c < c_hi && snaprow >= 0; ++c) hits += at(H0, snaprow + 1, c + 1) == sbest;
                const int n = hits + (sm ? 1 : 0);
                if (sbest > 0 && up && !eq) mult = n > 1;
                else if (eq && n > 0) mult = true;
            }
        }
        if (bbest != best0 || (best0 > 0 && mult != (n_best0 > 1))) {
            printf("mode 3 bookkeeping: best %d (band %d), mult %d, cells of the band holding S' %d\n", bbest, best0, (int)mult, n_best0);
            ok_mode3 = false;
        }
    }
    // ---- the bounds, as the kernel assembles them ----
    int m, stride, c0;
    zsw::seed_layout(L, p.K, p.spacer, &m, &stride, &c0);
    const int lam = zsw::seed_lambda(p, stride);
    int qfa[zsw::SEED_MAX_KMERS + 1], qfb[zsw::SEED_MAX_KMERS + 1];
    zsw::seed_suffix_q(m, p.K, c0, stride, L, p.maxw, sr.fa_mask, lam, qfa);
    zsw::seed_suffix_q(m, p.K, c0, stride, L, p.maxw, sr.fb_mask, lam, qfb);
    const int gup = zsw::seed_gap_up(p, g.wu), gdn = zsw::seed_gap_down(p, g.wd, L, sr.t_all);
    int bound_cls[NL] = {-1, -1, -1, -1, -1};
    if (g.dt + (g.K - 1) * C - g.wu > 0) bound_cls[FU] = sr.t_all - std::min(sr.d_fa, gup);  // fresh start above the band
    if (g.dt + C + g.wd < R) bound_cls[FD] = sr.t_all - std::min(sr.d_fb, gdn);              // fresh start below it
    for (int k = 0; k < g.K; ++k) {
        const int clast = std::min(L, (k + 1) * C) - 1, xl = (k + 1) * C - 1;
        if (k + 1 < g.K) {  // right edge of strip k above strip k+1's first row: H (diagonal step) and the outgoing F
            const int next_top = g.dt + (k + 1) * C - g.wu;  // unclamped: e counts diagonals
            int uk = 0, ug = 0;
            bool any = false;
            for (int r = g.top(k); r < std::min(g.bot(k), g.top(k + 1)); ++r) {
                const int h = at(H0, r + 1, clast + 1);
                const int fout = std::max(at(F0, r + 1, clast + 1) - s.ge, h - s.go);
                const int v = std::max(std::max(h, fout), 0);
                const int e = next_top - 1 - r;
                uk = std::max(uk, v);
                ug = std::max(ug, std::max(v - s.ge * std::max(e - 1, 0), 0));
                any = true;
            }
            if (any) bound_cls[XU] = std::max(bound_cls[XU], zsw::seed_band_upper(p, uk, ug, xl, L, g.wu, m, c0, stride, qfa));
        }
        if (g.bot(k) < R && g.bot(k) > g.top(k)) {  // last row of strip k: H (diagonal step) and the E of the next row
            const int r = g.bot(k) - 1;
            int mk = 0, mg = 0;
            for (int c = k * C; c <= clast; ++c) {
                const int h = at(H0, r + 1, c + 1);
                const int en = std::max(at(E0, r + 1, c + 1) - s.ge, h - s.go);
                const int he = std::max(std::max(h, en), 0);
                const int e = xl - c;
                mk = std::max(mk, he + p.maxw * e);
                mg = std::max(mg, std::max(he - s.ge * std::max(e - 1, 0), 0) + (e >= 1 ? p.maxw : 0));
            }
            bound_cls[XL] = std::max(bound_cls[XL], zsw::seed_band_lower(p, mk, mg, xl, L, g.wd, sr.t_all, m, c0, stride, qfb));
        }
    }
    bool ok = ok_mode3;
    int bound = -1;
    static const char* const cls_name[NL] = {"", "fresh start above", "fresh start below", "upper exit", "lower exit"};
    for (int j = 1; j < NL; ++j) {
        bound = std::max(bound, bound_cls[j]);
        // every class against its own bound (not the largest of the four: a weak bound must not hide behind another)
        if (best_cls[j] > std::max(bound_cls[j], 0)) {
            printf("%s: the class's best path %d exceeds its bound %d (S' %d)\n", cls_name[j], best_cls[j], bound_cls[j], best0);
            ok = false;
        }
    }
    if (truth != std::max(best0, best1)) {
        printf("model inconsistency\n");
        ok = false;
    }
    if (bound <= best0) {
        ++cnt->pass_score;
        if (plain) ++cnt->plain_pass;
        if (best0 != truth) {
            printf("passing read with a wrong score: band %d, truth %d, bound %d\n", best0, truth, bound);
            ok = false;
        }
    }
    if (bound < best0) {
        ++cnt->pass_ends;
        if (best0 != truth || row0 != trow || col0 != tcol) {
            printf("passing read with wrong ends: band %d (%d,%d), truth %d (%d,%d)\n", best0, row0, col0, truth, trow, tcol);
            ok = false;
        }
        if (n_best0 != n_truth) {  // mode 3: every cell holding the maximum is a cell of the band holding S'
            printf("passing read (ends): %d cells of the band hold S' = %d, %d cells of the matrix hold the maximum\n", n_best0, best0, n_truth);
            ok = false;
        }
        if (n_best0 == 1) {
            ++cnt->unique;
            if (row0 != trow2 || col0 != tcol2) {
                printf("unique maximum, but the transposed tie rule picks (%d,%d), the band (%d,%d)\n", trow2, tcol2, row0, col0);
                ok = false;
            }
        }
    }
    if (!ok) {
        printf("  dt %d C %d wu %d wd %d t_all %d d_fa %d d_fb %d fa %x fb %x gup %d gdn %d lambda %d\n  ref (%d): ", sr.dt, C, g.wu, g.wd, sr.t_all, sr.d_fa,
               sr.d_fb, sr.fa_mask, sr.fb_mask, gup, gdn, lam, R);
        for (uint8_t x : ref) putchar("ACGTN"[x]);
        printf("\n  read (%d): ", L);
        for (uint8_t x : q) putchar("ACGTN"[x]);
        printf("\n  K %d Dn %d Dm %d go %d ge %d maxw %d\n", p.K, p.Dn, p.Dm, p.go, p.ge, p.maxw);
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
                              dna(2, -5, -1, 10, 1), dna(4, -6, 1, 12, 2), dna(2, -2, 0, 3, 3)};
    Counters cnt;
    bool all_ok = true;
    for (int it = 0; it < iters && all_ok; ++it) {
        const Scheme& s = schemes[it % (sizeof(schemes) / sizeof(schemes[0]))];
        const int R = rnd(60, 420);
        std::vector<uint8_t> ref(R);
        for (auto& x : ref) x = (uint8_t)rnd(0, 3);
        if (rnd(0, 2) == 0 && R > 120) {
            const int len = rnd(20, 50), from = rnd(0, R - len), to = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[to + i] = ref[from + i];
        }
        if (rnd(0, 3) == 0 && R > 100) {
            const int unit = rnd(1, 6), len = rnd(20, 60), at = rnd(0, R - len);
            for (int i = unit; i < len; ++i) ref[at + i] = ref[at + i - unit];
        }
        if (rnd(0, 3) == 0)
            for (int k = rnd(1, 6); k > 0; --k) ref[rnd(0, R - 1)] = 4;
        bool ref_has[32] = {false};
        for (uint8_t x : ref) ref_has[x] = true;
        SeedParams p;
        const int K = rnd(3, 6);
        if (!zsw::seed_analyze(s.S, s.w.data(), s.go, s.ge, ref_has, K, &p)) continue;
        p.M1 = rnd(2, 24);
        p.M1_per8 = rnd(0, 2);
        p.M2 = rnd(2, 14);
        p.Dn = rnd(0, 4);
        p.Dm = rnd(0, 4);
        p.Wd = rnd(2, 20);
        p.Wd_per16 = rnd(0, 2);
        p.tol = rnd(0, 5);
        if (rnd(0, 2) == 0) p.spacer += rnd(0, 6);
        std::vector<uint32_t> table((size_t)2 << (2 * K), 0);
        zsw::seed_index_build(p, ref.data(), (uint64_t)R, table.data());
        for (int k = 0; k < 60 && all_ok; ++k) {
            const int kind = rnd(0, 10);
            const int L = rnd(K, std::min(R, 90));
            std::vector<uint8_t> q;
            auto copy_with_errors = [&](int start, int len, int sub_pct, int indel_pct) {
                int i = start;
                while ((int)q.size() < len) {
                    uint8_t b = (i >= 0 && i < R) ? ref[i] : (uint8_t)rnd(0, 3);
                    const int e = rnd(0, 999);
                    if (e < sub_pct * 10) b = (uint8_t)rnd(0, 3);
                    else if (e < sub_pct * 10 + indel_pct * 5) { ++i; continue; }
                    else if (e < sub_pct * 10 + indel_pct * 10) { q.push_back((uint8_t)rnd(0, 3)); continue; }
                    q.push_back(b);
                    ++i;
                }
                q.resize(len);
            };
            if (kind <= 4) copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(0, 3), rnd(0, 1));
            else if (kind == 5) copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(5, 20), rnd(1, 5));
            else if (kind == 6) {
                const int l1 = rnd(K, std::max(K, L - 1));
                copy_with_errors(rnd(0, std::max(0, R - l1)), l1, 1, 0);
                copy_with_errors(rnd(0, std::max(0, R - L)), L, 1, 0);
            } else if (kind == 7) copy_with_errors(rnd(0, 1) ? -rnd(1, L / 2 + 1) : R - rnd(1, L / 2 + 1) - L / 2, L, 1, 0);
            else if (kind == 8) {
                const int l1 = L / 2, st = rnd(0, std::max(0, R - L - 30));
                copy_with_errors(st, l1, 0, 0);
                if (rnd(0, 1)) {
                    int i = st + l1 + rnd(3, 28);
                    while ((int)q.size() < L) q.push_back(i < R ? ref[i++] : (uint8_t)rnd(0, 3));
                } else {
                    for (int x = rnd(3, 20); x > 0 && (int)q.size() < L; --x) q.push_back((uint8_t)rnd(0, 3));
                    int i = st + l1;
                    while ((int)q.size() < L) q.push_back(i < R ? ref[i++] : (uint8_t)rnd(0, 3));
                }
            } else if (kind == 10) copy_with_errors(rnd(0, std::max(0, R - L)), L, 0, 0);  // an exact copy: every detour out of the band and back is open
            else
                for (int i = 0; i < L; ++i) q.push_back((uint8_t)rnd(0, 3));
            if (rnd(0, 4) == 0)
                for (int x = rnd(1, 3); x > 0; --x) q[rnd(0, L - 1)] = 4;
            all_ok = check_read(s, p, table, ref, q, kind <= 4, rnd(5, 30), &cnt);
        }
    }
    printf("reads %ld, anchored %ld, passed (score) %ld, passed (ends) %ld, of which with the maximum in one cell %ld; plain reads %ld, of which passed %ld\n",
           cnt.reads, cnt.anchored, cnt.pass_score, cnt.pass_ends, cnt.unique, cnt.plain, cnt.plain_pass);
    if (!all_ok) return 1;
    if (cnt.plain > 200 && cnt.plain_pass * 5 < cnt.plain) {
        printf("the checks are vacuous: fewer than a fifth of the plain reads pass\n");
        return 1;
    }
    printf("seed_band OK\n");
    return 0;
}
