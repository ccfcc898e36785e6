// seed_band.cpp — host model of the banded seeded pass (zoe_amd/csrc/zsw_score_band.hip). The computed cells are a band of
// diagonals around the read's anchor, strip by strip (strip k: query columns [kC, (k+1)C), reference rows [dtmin + kC - Wu,
// dtmax + (k+1)C + Wd)); what enters the band from outside is an UPPER BOUND of the outside cell's value, doubled and made odd
// ("injected", zsw_seed.hpp), so that one dynamic programme yields the band's own maximum and the proof that no path through
// an outside cell beats it. This file walks a read exactly as the kernel does — same strips, same rows, same order of
// operations, every bound through zsw_seed.hpp (seed_col_step / seed_col_join / seed_strip_events / seed_tag / seed_untag), plain
// integers instead of packed halves — and checks against the full Gotoh matrix of the read:
//   I1  every cell of the band: the walk's value stands for a bound >= the cell's true H;
//   O1  every cell above the band in strip k's columns: true H <= what the strip's first row received above that column;
//   O2  every cell below the band in a strip's last column: true H <= yh, and the F it sends right <= yf, of the next strip;
//   O3  every cell outside the band: true H <= the final oa (above) / ob (below);
//   A   a read the walk accepts has the true score (tag -1) / the true score, first row, first column and — mode 3 — the true
//       number of cells holding the maximum (tag +1); the mode-3 bookkeeping of the kernel is restated and checked as before.
// I1-O3 are checked for every read, accepted or not, so an unsound bound shows at the first cell it is too low for — not only
// when a read happens to exploit it. Also built as a library (-DZSW_MODEL_LIB): zsw_model_band() gives the same walk to
// tests/test_gpu_bounds.py, which requires the kernel's own values (maximum, oa, ob) to EQUAL the model's on every read.
// usage: seed_band <iterations> <seed> [report]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../zoe_amd/csrc/zsw_seed.hpp"
#include "adversarial_reads.hpp"

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

struct Geometry {
    int C, n_strips, wu, wd, dtmin, dtmax, R;
    int top(int k) const { return std::max(0, std::min(R, dtmin + k * C - wu)); }
    int bot(int k) const { return std::max(0, std::min(R, dtmax + (k + 1) * C + wd)); }
};

struct Walk {
    uint32_t best2 = 0;  // the band's maximum, doubled, odd if a path through an outside cell holds it
    int oa = -1, ob = -1;
    int row = 0x7fffffff, col = 0x7fffffff;
    bool mult = false;
    // for the checks
    std::vector<uint32_t> U;      // [r * L + c]: the walk's value of a band cell, 0xffffffff = not a band cell
    std::vector<int> a, b;        // [c]: the bound of the cells above / below the band after column c, as the walk used it
    std::vector<int> yh, yf;      // [k]: what strip k's first column received in rows below strip k - 1 (k >= 1)
    struct Exit {
        int pos;  // row (upper exits) / column (lower exits) of the band cell the path leaves from
        int h;    // plain value of that cell
    };
    std::vector<std::vector<Exit>> exits_u, exits_l;  // [k]
};

struct Layout {
    int m, stride, c0, lam;
    uint32_t magic;
};

Layout layout_of(const SeedParams& p, int L) {
    Layout y;
    zsw::seed_layout(L, p.K, p.spacer, &y.m, &y.stride, &y.c0);
    y.lam = zsw::seed_lambda(p, y.stride);
    y.magic = zsw::seed_div_magic(y.stride);
    return y;
}

// The kernel's walk over one read (zsw_score_band.hip), MODE 3 bookkeeping included.
Walk walk_band(const Scheme& s, const SeedParams& p, const zsw::SeedRead& sr, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q,
               const Geometry& g, int tag, bool trace) {
    const int R = g.R, L = (int)q.size(), C = g.C;
    Walk out;
    if (trace) out.U.assign((size_t)R * L, 0xffffffffu);
    out.a.assign(L, -1);
    out.b.assign(L, -1);
    out.yh.assign(g.n_strips, 0);
    out.yf.assign(g.n_strips, 0);
    out.exits_u.resize(g.n_strips);
    out.exits_l.resize(g.n_strips);
    const Layout y = layout_of(p, L);
    zsw::SeedColDP up, lo;
    zsw::seed_col_init(&up);
    zsw::seed_col_init(&lo);
    int oa = -1, ob = -1;
    const int go2 = 2 * s.go, ge2 = 2 * s.ge;
    auto wgt2 = [&](int r, int c) { return c < L ? 2 * s.w[ref[r] * s.S + q[c]] : 0; };  // padding columns score 0
    std::vector<uint32_t> bndH, bndF;  // the previous strip's last column, rows [bnd_first, prev_bot)
    int bnd_first = 0, prev_bot = 0;
    int yh = 0, yf = 0;
    int best = 0, brow = 0x7fffffff, bcol = 0x7fffffff;
    bool mult = false;
    for (int k = 0; k < g.n_strips; ++k) {
        const int top = g.top(k), bot = g.bot(k);
        const bool has_next = k + 1 < g.n_strips, have_left = k > 0;
        const int next_top = g.top(k + 1);
        const int nr = std::max(0, std::min(C, L - k * C));  // real columns of the strip
        const int xl = (k + 1) * C - 1;
        const zsw::SeedStripEventsT<uint64_t> ea = zsw::seed_strip_events_t<uint64_t>(k * C, C, y.m, y.c0, y.stride, p.K, y.magic, sr.fa_mask);
        const zsw::SeedStripEventsT<uint64_t> eb = zsw::seed_strip_events_t<uint64_t>(k * C, C, y.m, y.c0, y.stride, p.K, y.magic, sr.fb_mask);
        out.yh[k] = yh;
        out.yf[k] = yf;
        // above the strip's columns: a(c); its first row receives it (E: less gap_open)
        std::vector<uint32_t> H(C, 0), E(C, 0);
        for (int c = 0; c < nr; ++c) {
            const int a = zsw::seed_col_step(&up, p.maxw, y.lam, (ea.start >> c) & 1u, (ea.end >> c) & 1u);
            if (top > 0) {
                oa = std::max(oa, a);
                out.a[k * C + c] = a;
                H[c] = zsw::seed_tag(a, tag);
                E[c] = zsw::seed_tag(std::max(0, a - s.go), tag);
            }
        }
        const uint32_t yh2 = have_left ? zsw::seed_tag(yh, tag) : 0u, yf2 = have_left ? zsw::seed_tag(yf, tag) : 0u;
        auto leftH = [&](int r) -> uint32_t {  // H of (r, kC - 1) as the strip sees it
            if (!have_left || r < 0) return 0u;
            return r < prev_bot ? (r >= bnd_first ? bndH[r - bnd_first] : 0u) : yh2;
        };
        uint32_t Hin_prev = (have_left && top >= 1) ? leftH(top - 1) : 0u;
        uint32_t uk = 0;
        std::vector<uint32_t> nH, nF;
        int sbest = 0, srow = 0x7fffffff, snaprow = -1;
        bool sm = false;
        std::vector<uint32_t> snap(C, 0);
        for (int r = top; r < bot; ++r) {
            const bool left = have_left && r < prev_bot;
            const uint32_t Hin = left ? bndH[r - bnd_first] : yh2;
            uint32_t F = left ? bndF[r - bnd_first] : yf2;
            int64_t hd = (int64_t)Hin_prev + wgt2(r, k * C);
            Hin_prev = Hin;
            uint32_t rmax = 0;
            for (int c = 0; c < C; ++c) {
                const int64_t hd_next = c + 1 < C ? (int64_t)H[c] + wgt2(r, k * C + c + 1) : 0;
                const uint32_t h = (uint32_t)std::max<int64_t>(std::max<int64_t>(hd, E[c]), F);
                H[c] = h;
                E[c] = (uint32_t)std::max<int64_t>(std::max<int64_t>((int64_t)E[c] - ge2, (int64_t)h - go2), 0);
                F = (uint32_t)std::max<int64_t>(std::max<int64_t>((int64_t)F - ge2, (int64_t)h - go2), 0);
                rmax = std::max(rmax, h);
                hd = hd_next;
                if (trace && c < nr) out.U[(size_t)r * L + k * C + c] = h;
            }
            if (has_next) {
                if (r >= next_top - 1) {
                    nH.push_back(H[C - 1]);
                    nF.push_back(F);
                }
                if (r < next_top) {  // the outgoing F of a cell never exceeds its H
                    uk = std::max(uk, H[C - 1]);
                    if (trace) out.exits_u[k].push_back({r, zsw::seed_untag(H[C - 1], tag)});
                }
            }
            // mode bookkeeping: the strip's maximum, the row of its latest rise, a later row reaching it again in a real column
            if ((int)rmax > sbest) {
                sbest = (int)rmax;
                srow = snaprow = r;
                sm = false;
                snap = H;
            } else if ((int)rmax == sbest && sbest > 0) {
                bool real_hit = false;
                for (int c = 0; c < nr; ++c) real_hit = real_hit || (int)H[c] == sbest;
                if (real_hit) sm = true;
            }
        }
        // merge the strip's maximum into the read's (higher, or equal in an earlier row; the same value in two strips: two cells)
        {
            const bool eq = sbest == best && sbest > 0, up_ = sbest > best || (eq && srow < brow);
            if (up_) {
                best = sbest;
                brow = srow;
            }
            if (up_ || eq) {
                int hits = 0, first = 0x7fffffff;
                for (int c = C - 1; c >= 0; --c)
                    if (snaprow >= 0 && (int)snap[c] == sbest) {
                        first = k * C + c;
                        if (c < nr) ++hits;
                    }
                if (up_) bcol = first;
                const int n = hits + (sm ? 1 : 0);
                if (sbest > 0 && up_ && !eq) mult = n > 1;
                else if (eq && n > 0) mult = true;
            }
        }
        // what left the band through the right edge joins the paths above it after column xl
        if (has_next && std::min(bot, next_top) > top && xl < L - 1)
            zsw::seed_col_join(&up, zsw::seed_untag(uk, tag), zsw::seed_exit_is_free(xl, y.m, y.c0, y.stride, p.K, p.spacer, y.magic, sr.fa_mask));
        // below the strip's columns: b(c), joined by the strip's last row
        const bool rows_below = bot < R, exits_below = rows_below && bot > top;
        int b = 0;
        for (int c = 0; c < nr; ++c) {
            b = zsw::seed_col_step(&lo, p.maxw, y.lam, (eb.start >> c) & 1u, (eb.end >> c) & 1u);
            if (exits_below) {
                const int he = zsw::seed_untag(H[c], tag);
                zsw::seed_col_join(&lo, he, (eb.inside >> c) & 1u);
                b = std::max(b, he);
                if (trace) out.exits_l[k].push_back({k * C + c, he});
            }
            if (rows_below) {
                ob = std::max(ob, b);
                out.b[k * C + c] = b;
            }
        }
        yh = (rows_below && nr == C) ? b : 0;
        yf = rows_below && nr == C ? std::max(0, b + std::max(0, y.lam - p.maxw) - s.go) : 0;
        bndH.swap(nH);
        bndF.swap(nF);
        bnd_first = std::max(next_top - 1, top);  // the rows stored: [max(next_top - 1, top), bot)
        prev_bot = bot;
    }
    out.best2 = (uint32_t)best;
    out.oa = oa;
    out.ob = ob;
    out.row = brow;
    out.col = bcol;
    out.mult = mult;
    return out;
}

constexpr int NEG = -(1 << 28);

struct Truth {
    std::vector<int> H, F;  // [r * L + c]; F: the horizontal-gap value ENTERING the cell (r, c) from the left
    int best = 0, row = -1, col = -1, n_best = 0, row2 = -1, col2 = -1;
};

Truth gotoh_full(const Scheme& s, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q) {
    const int R = (int)ref.size(), L = (int)q.size();
    Truth t;
    t.H.assign((size_t)R * L, 0);
    t.F.assign((size_t)R * L, 0);
    std::vector<int> Hp(L, 0), Ep(L, 0);  // previous row's H, and E entering the current row
    for (int r = 0; r < R; ++r) {
        int f = 0, hleft = 0, hdiag = 0;
        for (int c = 0; c < L; ++c) {
            const int e = Ep[c];
            f = c == 0 ? 0 : std::max(std::max(f - s.ge, hleft - s.go), 0);
            const int h = std::max(std::max(hdiag + s.w[ref[r] * s.S + q[c]], e), std::max(f, 0));
            t.F[(size_t)r * L + c] = f;
            t.H[(size_t)r * L + c] = h;
            hdiag = Hp[c];
            Hp[c] = h;
            Ep[c] = std::max(std::max(e - s.ge, h - s.go), 0);
            hleft = h;
            if (h > t.best) {
                t.best = h;
                t.row = t.row2 = r;
                t.col = t.col2 = c;
                t.n_best = 1;
            } else if (h == t.best && h > 0) {
                ++t.n_best;
                if (c < t.col2 || (c == t.col2 && r < t.row2)) {
                    t.col2 = c;
                    t.row2 = r;
                }
            }
        }
    }
    return t;
}

// Per-class checks (so that a weak bound cannot hide behind a larger one): the best path of each class from a dynamic programme
// that is open in the outside cells of one side only, against the column DP of that class alone.
//   pure   paths wholly above (below) the band, fresh starts allowed: after column c <= the column DP without any join;
//   exits  paths that leave strip k through its right edge (last row) with the walk's values and stay outside: after column c <=
//          the column DP without fresh starts, joined by that strip's exits only.
bool check_classes(const Scheme& s, const SeedParams& p, const zsw::SeedRead& sr, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const Geometry& g,
                   const Walk& w) {
    const int R = g.R, L = (int)q.size(), C = g.C;
    const int n_real = (L + C - 1) / C;  // strips with real columns
    const Layout y = layout_of(p, L);
    auto strip_of = [&](int c) { return c / C; };
    bool ok = true;
    for (int side = 0; side < 2 && ok; ++side) {  // 0 above, 1 below
        auto open = [&](int r, int c) { return side == 0 ? r < g.top(strip_of(c)) : r >= g.bot(strip_of(c)); };
        const uint32_t mask = side == 0 ? sr.fa_mask : sr.fb_mask;
        // source -1: the pure class; source k: the exits of strip k
        for (int src = -1; src < n_real && ok; ++src) {
            const std::vector<Walk::Exit>* ex = src < 0 ? nullptr : (side == 0 ? &w.exits_u[src] : &w.exits_l[src]);
            if (ex && ex->empty()) continue;
            const int xl = src < 0 ? -1 : (src + 1) * C - 1;
            if (ex && side == 0 && xl >= L - 1) continue;
            // the class's own bound per column
            std::vector<int> bound(L, NEG);
            {
                zsw::SeedColDP dp;
                zsw::seed_col_init(&dp, src < 0);
                for (int k = 0; k < n_real; ++k) {
                    const zsw::SeedStripEventsT<uint64_t> e = zsw::seed_strip_events_t<uint64_t>(k * C, C, y.m, y.c0, y.stride, p.K, y.magic, mask);
                    int ux = -1;
                    for (int c = k * C; c < std::min(L, (k + 1) * C); ++c) {
                        int v = zsw::seed_col_step(&dp, p.maxw, y.lam, (e.start >> (c - k * C)) & 1u, (e.end >> (c - k * C)) & 1u, src < 0);
                        if (ex && side == 1 && k == src)
                            for (const auto& x : *ex)
                                if (x.pos == c) {
                                    zsw::seed_col_join(&dp, x.h, (e.inside >> (c - k * C)) & 1u);
                                    v = std::max(v, x.h);
                                }
                        bound[c] = v;
                    }
                    if (ex && side == 0 && k == src) {
                        for (const auto& x : *ex) ux = std::max(ux, x.h);
                        zsw::seed_col_join(&dp, ux, zsw::seed_exit_is_free(xl, y.m, y.c0, y.stride, p.K, p.spacer, y.magic, mask));
                    }
                }
            }
            std::vector<int> H((size_t)(R + 1) * (L + 1), NEG), E = H, F = H;  // 1-based, 0 = border
            auto at = [&](std::vector<int>& v, int r, int c) -> int& { return v[(size_t)(r + 1) * (L + 1) + (c + 1)]; };
            // values a band cell sends out: exit cells act as closed cells with given H (their gap values: H - gap_open)
            std::vector<int> exH;  // by row (above) or column (below)
            if (ex) {
                exH.assign(side == 0 ? R : L, NEG);
                for (const auto& e : *ex) exH[e.pos] = e.h;
            }
            const char* side_name = side == 0 ? "above" : "below";
            for (int r = 0; r < R && ok; ++r)
                for (int c = 0; c < L && ok; ++c) {
                    if (!open(r, c)) continue;
                    int d = NEG, e = NEG, f = NEG;
                    if (r >= 1 && c >= 1) {
                        if (open(r - 1, c - 1)) d = at(H, r - 1, c - 1);
                        if (ex && side == 0 && c - 1 == xl && exH[r - 1] > NEG / 2) d = std::max(d, exH[r - 1]);
                        if (ex && side == 1 && r - 1 == g.bot(src) - 1 && strip_of(c - 1) == src && exH[c - 1] > NEG / 2) d = std::max(d, exH[c - 1]);
                    }
                    if (!ex) d = std::max(d, 0);  // a fresh start
                    if (r >= 1) {
                        if (open(r - 1, c)) e = std::max(at(E, r - 1, c) - s.ge, at(H, r - 1, c) - s.go);
                        if (ex && side == 1 && r - 1 == g.bot(src) - 1 && strip_of(c) == src && exH[c] > NEG / 2) e = std::max(e, exH[c] - std::min(s.go, s.ge));
                    }
                    if (c >= 1) {
                        if (open(r, c - 1)) f = std::max(at(F, r, c - 1) - s.ge, at(H, r, c - 1) - s.go);
                        if (ex && side == 0 && c - 1 == xl && exH[r] > NEG / 2) f = std::max(f, exH[r] - std::min(s.go, s.ge));
                    }
                    int h = d > NEG / 2 ? d + s.w[ref[r] * s.S + q[c]] : NEG;
                    h = std::max(h, std::max(e, f));
                    if (h < NEG / 2) h = NEG;
                    at(H, r, c) = h;
                    at(E, r, c) = e < NEG / 2 ? NEG : e;
                    at(F, r, c) = f < NEG / 2 ? NEG : f;
                    if (h > std::max(bound[c], 0)) {
                        printf("class %s %s", src < 0 ? "pure" : "exit", side_name);
                        if (src >= 0) printf(" of strip %d", src);
                        printf(": cell (%d,%d) holds %d > the class's bound %d\n", r, c, h, bound[c]);
                        ok = false;
                    }
                }
        }
    }
    return ok;
}

struct Counters {
    long reads = 0, anchored = 0, pass_score = 0, pass_ends = 0, plain = 0, plain_pass = 0, unique = 0;
    long div_reads[4] = {0, 0, 0, 0}, div_pass[4] = {0, 0, 0, 0};  // reads with 3 / 5 / 8 / 12 % substitutions: how many the walk accepts
};

bool check_read(const Scheme& s, const SeedParams& p, const std::vector<uint32_t>& table, const std::vector<uint8_t>& ref,
                const std::vector<uint8_t>& q, bool plain, int div_class, int C, int wu, int wd, int slack_lo, int slack_hi, int extra_len, Counters* cnt) {
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
    if (div_class >= 0) ++cnt->div_reads[div_class];
    Geometry g;
    g.C = C;
    g.n_strips = (L + extra_len + C - 1) / C;  // a longer partner in the lane: strips of padding
    g.wu = wu;
    g.wd = wd;
    g.dtmin = sr.dt - slack_lo;  // a partner whose anchor lies up to SEED_BAND_SLACK diagonals away widens the band on one side
    g.dtmax = sr.dt + slack_hi;
    g.R = R;
    const Truth t = gotoh_full(s, ref, q);
    bool ok = true;
    for (int tag = -1; tag <= 1 && ok; tag += 2) {
        const Walk w = walk_band(s, p, sr, ref, q, g, tag, true);
        // I1, O1, O2, O3
        for (int k = 0; k < g.n_strips && ok; ++k) {
            const int top = g.top(k), bot = g.bot(k);
            for (int c = k * C; c < std::min(L, (k + 1) * C) && ok; ++c) {
                for (int r = 0; r < R && ok; ++r) {
                    const int h = t.H[(size_t)r * L + c];
                    if (r >= top && r < bot) {
                        const uint32_t u = w.U[(size_t)r * L + c];
                        if (u == 0xffffffffu || zsw::seed_untag(u, tag) < h) {
                            printf("I1: band cell (%d,%d) holds %u (tag %d), true H %d\n", r, c, u, tag, h);
                            ok = false;
                        }
                    } else if (r < top) {
                        if (h > w.a[c] || h > w.oa) {
                            printf("O1/O3: cell (%d,%d) above strip %d: true H %d > a(c) %d or oa %d\n", r, c, k, h, w.a[c], w.oa);
                            ok = false;
                        }
                    } else {
                        if (h > w.b[c] || h > w.ob) {
                            printf("O1/O3: cell (%d,%d) below strip %d: true H %d > b(c) %d or ob %d\n", r, c, k, h, w.b[c], w.ob);
                            ok = false;
                        }
                        if (c == (k + 1) * C - 1 && k + 1 < g.n_strips) {
                            const int fout = std::max(std::max(t.F[(size_t)r * L + c] - s.ge, h - s.go), 0);
                            if (h > w.yh[k + 1] || fout > w.yf[k + 1]) {
                                printf("O2: cell (%d,%d) below strip %d: true H %d / outgoing F %d > yh %d / yf %d\n", r, c, k, h, fout, w.yh[k + 1], w.yf[k + 1]);
                                ok = false;
                            }
                        }
                    }
                }
            }
        }
        if (ok) ok = check_classes(s, p, sr, ref, q, g, w);
        // A
        const bool even = (w.best2 & 1u) == 0;
        const int S = (int)(w.best2 >> 1), outside = std::max(w.oa, w.ob);
        if (tag < 0) {
            if (even && outside <= S) {
                ++cnt->pass_score;
                if (plain) ++cnt->plain_pass;
                if (div_class >= 0) ++cnt->div_pass[div_class];
                if (S != t.best) {
                    printf("A: accepted (score) with %d, truth %d\n", S, t.best);
                    ok = false;
                }
            }
        } else if (even && outside < S) {
            ++cnt->pass_ends;
            int n_band = 0;  // cells of the band holding S'
            for (int k = 0; k < g.n_strips; ++k)
                for (int c = k * C; c < std::min(L, (k + 1) * C); ++c)
                    for (int r = g.top(k); r < g.bot(k); ++r) n_band += w.U[(size_t)r * L + c] == w.best2;
            if (S != t.best || w.row != t.row || w.col != t.col) {
                printf("A: accepted (ends) with %d (%d,%d), truth %d (%d,%d)\n", S, w.row, w.col, t.best, t.row, t.col);
                ok = false;
            }
            if (S > 0 && (n_band != t.n_best || w.mult != (n_band > 1))) {
                printf("A: mode 3: %d cells of the band hold S', %d cells of the matrix the maximum, mult %d\n", n_band, t.n_best, (int)w.mult);
                ok = false;
            }
            if (t.n_best == 1) {
                ++cnt->unique;
                if (w.row != t.row2 || w.col != t.col2) {
                    printf("A: unique maximum, but the transposed tie rule picks (%d,%d), the band (%d,%d)\n", t.row2, t.col2, w.row, w.col);
                    ok = false;
                }
            }
        }
        if (!ok) printf("  tag %d best2 %u oa %d ob %d\n", tag, w.best2, w.oa, w.ob);
    }
    if (!ok) {
        printf("  dt %d C %d wu %d wd %d dtmin %d dtmax %d strips %d t_all %d d_fa %d d_fb %d fa %x fb %x\n  ref (%d): ", sr.dt, C, wu, wd, g.dtmin, g.dtmax, g.n_strips,
               sr.t_all, sr.d_fa, sr.d_fb, sr.fa_mask, sr.fb_mask, R);
        for (uint8_t x : ref) putchar("ACGTN"[x]);
        printf("\n  read (%d): ", L);
        for (uint8_t x : q) putchar("ACGTN"[x]);
        printf("\n  K %d spacer %d Dn %d Dm %d go %d ge %d maxw %d lambda %d\n", p.K, p.spacer, p.Dn, p.Dm, p.go, p.ge, p.maxw, p.lambda);
    }
    return ok;
}

}  // namespace

#ifdef ZSW_MODEL_LIB
// The walk for one read of a GPU batch, with the geometry the kernel reports (tests/test_gpu_bounds.py). w: S x S weights (row =
// reference residue), ref / read: residue indices. params: K, spacer override (0 = seed_analyze's), Dn, Dm, tol.
// out: [0] anchored, [1] anchor diagonal, [2] the band's maximum (doubled, tagged), [3] oa, [4] ob, [5] first row, [6] first column, [7] mult
extern "C" int zsw_model_band(const int32_t* w, int S, int go, int ge, const uint8_t* ref, int R, const uint8_t* read, int L, int K, int Dn, int Dm, int tol,
                              int C, int n_strips, int wu, int wd, int dtmin, int dtmax, int tag, int32_t* out) {
    static std::vector<uint32_t> table;
    static std::vector<uint8_t> table_ref;
    static std::vector<int32_t> table_w;
    static SeedParams p;
    static bool usable = false;
    Scheme s;
    s.S = S;
    s.w.assign(w, w + S * S);
    s.go = go;
    s.ge = ge;
    std::vector<uint8_t> vref(ref, ref + R), q(read, read + L);
    if (table_ref != vref || table_w != s.w || p.K != K || p.go != go || p.ge != ge) {  // one index per reference and scheme
        bool ref_has[32] = {false};
        for (uint8_t x : vref) ref_has[x & 31] = true;
        usable = zsw::seed_analyze(S, w, go, ge, ref_has, K, &p);
        p.Dn = Dn;
        p.Dm = Dm;
        p.tol = tol;
        p.M1 = p.M1_per8 = p.M2 = p.Wd = p.Wd_per16 = 0;
        table.assign((size_t)2 << (2 * K), 0u);
        if (usable) zsw::seed_index_build(p, vref.data(), (uint64_t)R, table.data());
        table_ref = vref;
        table_w = s.w;
    }
    std::memset(out, 0, 8 * sizeof(int32_t));
    if (!usable) return 1;
    auto cell = [&](int c) { return zsw::seed_cell(p, (int)q[c]); };
    auto look = [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
        *f1 = table[2 * (size_t)code];
        *l1 = table[2 * (size_t)code + 1];
    };
    const zsw::SeedRead sr = zsw::seed_read(p, L, cell, look);
    out[0] = sr.ok;
    out[1] = sr.dt;
    if (!sr.ok) return 0;
    Geometry g{C, n_strips, wu, wd, dtmin, dtmax, R};
    const Walk wk = walk_band(s, p, sr, vref, q, g, tag, false);
    out[2] = (int32_t)wk.best2;
    out[3] = wk.oa;
    out[4] = wk.ob;
    out[5] = wk.row;
    out[6] = wk.col;
    out[7] = wk.mult;
    return 0;
}
#else
// `report`: the production setting (150-base reads against a 2 kb reference, 2 / -5, -10 / -1, K = 8, strips of ZSW_REPORT_C columns, default 32) —
// how many reads of each divergence the walk accepts in a band of (wu, wd), for choosing the kernel's tiers. Every accepted read is
// compared with the full matrix.
int report(uint64_t seed, int n_reads) {
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    const Scheme s = dna(2, -5, 0, 10, 1);
    const int R = 2000, L = 150, K = 8, C = std::getenv("ZSW_REPORT_C") ? atoi(std::getenv("ZSW_REPORT_C")) : 32;
    std::vector<uint8_t> ref(R);
    for (auto& x : ref) x = (uint8_t)rnd(0, 3);
    bool ref_has[32] = {false};
    for (uint8_t x : ref) ref_has[x] = true;
    SeedParams p;
    if (!zsw::seed_analyze(s.S, s.w.data(), s.go, s.ge, ref_has, K, &p)) return 1;
    p.M1 = p.M1_per8 = p.M2 = p.Wd = p.Wd_per16 = 0;
    p.Dn = 4;
    p.Dm = 4;
    p.tol = 8;
    std::vector<uint32_t> table((size_t)2 << (2 * K), 0);
    zsw::seed_index_build(p, ref.data(), (uint64_t)R, table.data());
    static const int rates[] = {10, 20, 30, 50, 80, 120};
    static const int bands[][2] = {{6, 5}, {8, 6}, {10, 6}, {12, 8}, {16, 10}, {21, 10}, {24, 12}, {32, 16}, {42, 18}, {64, 32}};
    printf("accepted reads per 1000 (score-only tag / ends tag), 150 bases vs 2 kb, lambda %d, spacer %d\n   wu wd |", p.lambda, p.spacer);
    for (int rate : rates) printf("   %4.1f %%   |", rate / 10.0);
    printf("\n");
    for (const auto& band : bands) {
        printf("   %2d %2d |", band[0], band[1]);
        for (int rate : rates) {
            std::mt19937_64 rr(seed * 7919 + (uint64_t)rate);
            auto rn = [&](int lo, int hi) { return lo + (int)(rr() % (uint64_t)(hi - lo + 1)); };
            int acc[2] = {0, 0}, anchored = 0;
            for (int i = 0; i < n_reads; ++i) {
                std::vector<uint8_t> q;
                int pos = rn(0, R - L - 8);
                while ((int)q.size() < L) {
                    uint8_t b = pos < R ? ref[pos] : (uint8_t)rn(0, 3);
                    const int e = rn(0, 999);
                    if (e < rate) b = (uint8_t)((b + rn(1, 3)) & 3);
                    else if (e < rate + rate / 20) { ++pos; continue; }
                    else if (e < rate + rate / 10) { q.push_back((uint8_t)rn(0, 3)); continue; }
                    q.push_back(b);
                    ++pos;
                }
                auto cell = [&](int c) { return zsw::seed_cell(p, (int)q[c]); };
                auto look = [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
                    *f1 = table[2 * (size_t)code];
                    *l1 = table[2 * (size_t)code + 1];
                };
                const zsw::SeedRead sr = zsw::seed_read(p, L, cell, look);
                if (!sr.ok) continue;
                ++anchored;
                Geometry g{C, (L + C - 1) / C, band[0], band[1], sr.dt, sr.dt, R};
                for (int t = 0; t < 2; ++t) {
                    const int tag = t ? 1 : -1;
                    const Walk w = walk_band(s, p, sr, ref, q, g, tag, false);
                    const int S = (int)(w.best2 >> 1), outside = std::max(w.oa, w.ob);
                    if (t == 0 && band[0] == 21 && std::getenv("ZSW_REPORT_WHY")) {
                        static int shown = 0;
                        const bool odd = w.best2 & 1u;
                        if ((odd || outside > S) && rate >= 50 && shown++ < 40)
                            printf("\n   rate %d: S' %d%s oa %d ob %d t_all %d fa %04x fb %04x truth %d", rate, S, odd ? " (odd)" : "", w.oa, w.ob, sr.t_all, sr.fa_mask,
                                   sr.fb_mask, gotoh_full(s, ref, q).best);
                    }
                    if (!(w.best2 & 1u) && (t ? outside < S : outside <= S)) {
                        ++acc[t];
                        if (t == 0 && i % 16 == 0 && S != gotoh_full(s, ref, q).best) {
                            printf("accepted with a wrong score\n");
                            return 1;
                        }
                    }
                }
            }
            (void)anchored;
            printf(" %4d / %4d |", acc[0] * 1000 / n_reads, acc[1] * 1000 / n_reads);
        }
        printf("\n");
    }
    (void)rnd;
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    if (argc > 3 && std::strcmp(argv[3], "report") == 0) return report(seed, iters);
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    // seed_div against the division it replaces
    for (int d = 1; d <= 400; ++d) {
        const uint32_t magic = zsw::seed_div_magic(d);
        for (int x = 0; (long)x * d < (1 << 20) && x < 8192 && x < 4095 * d; ++x)
            if (zsw::seed_div(x, magic) != x / d) {
                printf("seed_div(%d, %d) = %d\n", x, d, zsw::seed_div(x, magic));
                return 1;
            }
    }
    const Scheme schemes[] = {dna(2, -5, 0, 10, 1), dna(1, -1, 0, 2, 1), dna(3, -2, 0, 5, 0), dna(1, -3, 0, 5, 2), dna(5, -4, 0, 8, 0),
                              dna(2, -5, -1, 10, 1), dna(4, -6, 1, 12, 2), dna(2, -2, 0, 3, 3), dna(2, -10, 0, 10, 1), dna(2, -5, 0, 5, 1),
                              dna(3, -9, 0, 6, 1)};
    Counters cnt;
    long structured = 0;
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
        p.M1 = p.M1_per8 = p.M2 = p.Wd = p.Wd_per16 = 0;
        p.Dn = rnd(0, 4);
        p.Dm = rnd(0, 4);
        p.tol = rnd(0, 5);
        if (rnd(0, 2) == 0) p.spacer += rnd(0, 6);
        std::vector<uint32_t> table((size_t)2 << (2 * K), 0);
        zsw::seed_index_build(p, ref.data(), (uint64_t)R, table.data());
        auto geometry = [&](int* C, int* wu, int* wd, int* slo, int* shi, int* extra) {
            *C = rnd(0, 3) ? rnd(5, 32) : rnd(33, 64);  // (the kernel: 16 and 48)
            *wu = p.Dn + rnd(0, 24);
            *wd = p.Dm + rnd(0, 16);
            *slo = rnd(0, 3) ? 0 : rnd(0, 32);
            *shi = rnd(0, 3) ? 0 : rnd(0, 32);
            *extra = rnd(0, 4) ? 0 : rnd(1, 40);
        };
        for (int k = 0; k < 60 && all_ok; ++k) {
            const int kind = rnd(0, 12);
            const int L = rnd(K, std::min(R, 90));
            std::vector<uint8_t> q;
            auto copy_with_errors = [&](int start, int len, int sub_pm, int indel_pm) {  // per mille
                int i = start;
                while ((int)q.size() < len) {
                    uint8_t b = (i >= 0 && i < R) ? ref[i] : (uint8_t)rnd(0, 3);
                    const int e = rnd(0, 999);
                    if (e < sub_pm) b = (uint8_t)((b + rnd(1, 3)) & 3);
                    else if (e < sub_pm + indel_pm / 2) { ++i; continue; }
                    else if (e < sub_pm + indel_pm) { q.push_back((uint8_t)rnd(0, 3)); continue; }
                    q.push_back(b);
                    ++i;
                }
                q.resize(len);
            };
            int div_class = -1;
            if (kind <= 3) copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(0, 30), rnd(0, 10));
            else if (kind == 4 || kind == 11 || kind == 12) {  // diverged reads: 3 / 5 / 8 / 12 % substitutions + a tenth of that in indels
                static const int rate[4] = {30, 50, 80, 120};
                div_class = rnd(0, 3);
                copy_with_errors(rnd(0, std::max(0, R - L)), L, rate[div_class], rate[div_class] / 10);
            } else if (kind == 5) copy_with_errors(rnd(0, std::max(0, R - L)), L, rnd(50, 200), rnd(10, 50));
            else if (kind == 6) {
                const int l1 = rnd(K, std::max(K, L - 1));
                copy_with_errors(rnd(0, std::max(0, R - l1)), l1, 10, 0);
                copy_with_errors(rnd(0, std::max(0, R - L)), L, 10, 0);
            } else if (kind == 7) copy_with_errors(rnd(0, 1) ? -rnd(1, L / 2 + 1) : R - rnd(1, L / 2 + 1) - L / 2, L, 10, 0);
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
            int C, wu, wd, slo, shi, extra;
            geometry(&C, &wu, &wd, &slo, &shi, &extra);
            all_ok = check_read(s, p, table, ref, q, kind <= 3, div_class, C, wu, wd, slo, shi, extra, &cnt);
        }
        for (int k = 0; k < 12 && all_ok; ++k) {  // structured cases (adversarial_reads.hpp), each with its own reference
            std::vector<uint8_t> aref, aq;
            SeedParams pa;
            bool has[32] = {false};
            has[0] = has[1] = has[2] = has[3] = true;
            if (!zsw::seed_analyze(s.S, s.w.data(), s.go, s.ge, has, K, &pa)) break;
            pa.M1 = pa.M1_per8 = pa.M2 = pa.Wd = pa.Wd_per16 = 0;
            pa.Dn = p.Dn;
            pa.Dm = p.Dm;
            pa.tol = p.tol;
            if (!adversarial::spacer_case(rng, pa, rnd(2 * (K + pa.spacer), 96), &aref, &aq)) continue;
            std::vector<uint32_t> atable((size_t)2 << (2 * K), 0);
            zsw::seed_index_build(pa, aref.data(), (uint64_t)aref.size(), atable.data());
            Counters unused;
            int C, wu, wd, slo, shi, extra;
            geometry(&C, &wu, &wd, &slo, &shi, &extra);
            all_ok = check_read(s, pa, atable, aref, aq, false, -1, C, wu, wd, slo, shi, extra, &unused);
            ++structured;
        }
    }
    printf("reads %ld, anchored %ld, accepted (score) %ld, accepted (ends) %ld, of which with the maximum in one cell %ld; plain reads %ld, of which accepted %ld; "
           "structured cases %ld\n",
           cnt.reads, cnt.anchored, cnt.pass_score, cnt.pass_ends, cnt.unique, cnt.plain, cnt.plain_pass, structured);
    printf("diverged reads accepted (3 / 5 / 8 / 12 %% substitutions): %ld/%ld %ld/%ld %ld/%ld %ld/%ld\n", cnt.div_pass[0], cnt.div_reads[0], cnt.div_pass[1],
           cnt.div_reads[1], cnt.div_pass[2], cnt.div_reads[2], cnt.div_pass[3], cnt.div_reads[3]);
    if (!all_ok) return 1;
    if (cnt.plain > 200 && cnt.plain_pass * 5 < cnt.plain) {
        printf("the checks are vacuous: fewer than a fifth of the plain reads are accepted\n");
        return 1;
    }
    printf("seed_band OK\n");
    return 0;
}
#endif
