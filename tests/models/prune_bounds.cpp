// prune_bounds.cpp — host model of the column-pruned first pass (zoe_amd/csrc/zsw_score_prune.hip): the strip, the window, the
// three bound checks, with plain integers. For every read it computes the full Gotoh matrix (the truth: maximum, first row
// holding it, first column of that row — the tie rule of striped.rs:296-321), then what the two kernels compute, and asserts the
// claim the pruned pass rests on:
//   * score only: if all bounds are <= S', then S' is the true maximum;
//   * with ends: if all bounds are < S' and the strip's maximum is < S', then the window's (row, column) of the first maximum is
//     the true one.
// A read that fails a check is counted (the GPU path rescans it over all cells); the model also checks that the checks are not
// vacuous (most plain reads pass). usage: prune_bounds <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace {

struct Scheme {
    int match, mismatch, go, ge;  // gaps as positive magnitudes
    // wide form (8-32 letter alphabets, prune_*_kernel<.., WIDE>): letters 'A' + k, scores from `w`, and the bounds add the
    // columns' own potentials max(0, max_x w[x][q_c]) instead of maxw per column
    int letters = 0;
    std::vector<int> w;
};

int code(uint8_t b) {
    switch (b | 32) {
        case 'a': return 0;
        case 'c': return 1;
        case 'g': return 2;
        case 't': return 3;
        default: return 4;
    }
}

struct Truth {
    int best = 0, row = -1, col = -1;  // first row holding the maximum, first column of that row
};

struct Dp {
    int R, L;
    std::vector<int> H, E, F;  // (R+1) x (L+1), row/column 0 = border
    int& h(int r, int c) { return H[(size_t)r * (L + 1) + c]; }
    int& e(int r, int c) { return E[(size_t)r * (L + 1) + c]; }
    int& f(int r, int c) { return F[(size_t)r * (L + 1) + c]; }
};

int weight(const Scheme& s, uint8_t a, uint8_t b) {
    if (s.letters) return s.w[(size_t)(a - 'A') * s.letters + (b - 'A')];
    const int x = code(a), y = code(b);
    if (x == 4 || y == 4) return 0;  // the ignored residue
    return x == y ? s.match : s.mismatch;
}

// E[r][c]: best score ending at (r, c) in a vertical gap; F: in a horizontal gap (both floored at 0, like the saturating lanes)
void full_dp(const Scheme& s, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, Dp* d, Truth* t) {
    const int R = (int)ref.size(), L = (int)q.size();
    d->R = R;
    d->L = L;
    d->H.assign((size_t)(R + 1) * (L + 1), 0);
    d->E = d->H;
    d->F = d->H;
    for (int r = 1; r <= R; ++r)
        for (int c = 1; c <= L; ++c) {
            d->e(r, c) = std::max(0, std::max(d->e(r - 1, c) - s.ge, d->h(r - 1, c) - s.go));
            d->f(r, c) = std::max(0, std::max(d->f(r, c - 1) - s.ge, d->h(r, c - 1) - s.go));
            int h = std::max(0, d->h(r - 1, c - 1) + weight(s, ref[r - 1], q[c - 1]));
            h = std::max(h, std::max(d->e(r, c), d->f(r, c)));
            d->h(r, c) = h;
            if (h > t->best) {
                t->best = h;
                t->row = r - 1;
                t->col = c - 1;
            }
        }
}

constexpr int BLK = 32, M1 = 8;

struct Outcome {
    bool pass_score = false, pass_ends = false;
    int S = 0, row = -1, col = -1;
};

// what prune_strip_kernel<CP> + prune_window_kernel compute for one read (window kernel: a rectangular cut at the last window row)
Outcome pruned(const Scheme& s, const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, Dp& d, int CP, int M2) {
    const int R = d.R, L = d.L, maxw = std::max(s.match, 0);
    Outcome o;
    // P[c]: the most the 0-based columns c.. can add to a path (gaps add nothing)
    std::vector<int> P(L + 2, 0);
    for (int c = L - 1; c >= 0; --c) {
        int pot = maxw;
        if (s.letters) {
            pot = 0;
            for (int x = 0; x < s.letters; ++x) pot = std::max(pot, s.w[(size_t)x * s.letters + (q[c] - 'A')]);
        }
        P[c] = P[c + 1] + pot;
    }
    // ---- strip: columns [0, CP) of every row are exact; what leaves it per row ----
    int strip = 0;
    std::vector<int> tH(R, 0), tF(R, 0);
    const int cp = std::min(CP, L);
    for (int r = 1; r <= R; ++r) {
        for (int c = 1; c <= cp; ++c) strip = std::max(strip, d.h(r, c));
        tH[r - 1] = CP <= L ? d.h(r, CP) : 0;                                                   // H[r][CP-1]
        tF[r - 1] = CP < L ? std::max(0, std::max(d.f(r, CP) - s.ge, d.h(r, CP) - s.go)) : 0;  // F[r][CP] = F entering column CP
    }
    const int nblk = (R + BLK - 1) / BLK;
    std::vector<int> mH(nblk, 0), mF(nblk, 0);
    int anchor = 0;
    for (int r = 0; r < R; ++r) {
        mH[r / BLK] = std::max(mH[r / BLK], tH[r]);
        mF[r / BLK] = std::max(mF[r / BLK], tF[r]);
        if (tH[r] > tH[anchor]) anchor = r;
    }
    // ---- window ----
    const int Rup = nblk * BLK;
    const int a0 = std::max(0, anchor - M1) / BLK * BLK;
    const int b1 = std::min(Rup, (anchor + 1 + std::max(0, L - CP) + M2 + BLK - 1) / BLK * BLK);
    const int rows_end = std::min(R, b1);  // rows [a0, rows_end) exist
    // right part from a zero state, exact boundary inputs; the diagonal into column CP of row a0 comes from outside: 0
    std::vector<int> Hp(L + 1, 0), Ep(L + 1, 0), Hc(L + 1, 0), Ec(L + 1, 0);
    int best2 = 0, row2 = -1, col2 = -1;
    int Fout_last = 0;
    for (int r = a0; r < rows_end; ++r) {
        int Hleft = tH[r];
        const int fin = tF[r];  // the F entering column CP
        for (int c = CP; c < L; ++c) {  // 0-based columns CP..L-1 (1-based c+1)
            const int e = std::max(0, std::max(Ep[c] - s.ge, Hp[c] - s.go));
            const int f = c == CP ? fin : std::max(0, std::max(Fout_last - s.ge, Hleft - s.go));
            const int hd = c == CP ? (r > a0 ? tH[r - 1] : 0) : Hp[c - 1];
            int h = std::max(0, hd + weight(s, ref[r], q[c]));
            h = std::max(h, std::max(e, f));
            Hc[c] = h;
            Ec[c] = e;
            Hleft = h;
            Fout_last = f;
            if (h > best2) {
                best2 = h;
                row2 = r;
                col2 = c;
            }
        }
        std::swap(Hp, Hc);
        std::swap(Ep, Ec);
    }
    o.S = std::max(strip, best2);
    o.row = row2;
    o.col = col2;
    const int rem = std::max(0, L - CP);
    int bound = P[std::min(CP, L)];  // V1
    if (rem > 0)
        for (int k = 0; k < nblk; ++k)  // V2: blocks outside the window
            if (k < a0 / BLK || k >= b1 / BLK) bound = std::max(bound, std::max(mH[k] + P[CP], mF[k] + P[CP + 1]));
    if (rows_end < R)  // V3: what can still leave the last window row (E of the next row from this row's state)
        for (int c = CP; c < L; ++c) {
            const int enext = std::max(0, std::max(Ep[c] - s.ge, Hp[c] - s.go));
            bound = std::max(bound, std::max(Hp[c], enext) + P[c + 1]);
        }
    o.pass_score = bound <= o.S;
    o.pass_ends = bound < o.S && strip < o.S;
    return o;
}

std::vector<uint8_t> random_seq(std::mt19937& g, int n, const char* alpha = "ACGT") {
    std::vector<uint8_t> v(n);
    int k = 0;
    while (alpha[k]) ++k;
    for (auto& x : v) x = (uint8_t)alpha[g() % k];
    return v;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40;
    std::mt19937 g(argc > 2 ? (unsigned)atoll(argv[2]) : 1u);
    const Scheme schemes[] = {{2, -5, 10, 1}, {1, -1, 2, 1}, {5, -4, 12, 2}, {3, -2, 4, 0}, {1, -3, 5, 2}, {2, 0, 0, 0}};
    long checked = 0, passed = 0, passed_ends = 0, plain = 0, plain_passed = 0;
    for (int it = 0; it < iters; ++it) {
        Scheme s = schemes[g() % 6];
        const bool wide = it % 3 == 2;
        char alpha[33] = "ACGT";
        if (wide) {  // a BLOSUM-shaped matrix: identities 4..11, substitutions -4..3 (some of them close to an identity), symmetric or not
            s.letters = 8 + (int)(g() % 18);
            s.w.assign((size_t)s.letters * s.letters, 0);
            for (int x = 0; x < s.letters; ++x)
                for (int y = 0; y <= x; ++y) {
                    const int v = x == y ? 4 + (int)(g() % 8) : -4 + (int)(g() % 8);
                    s.w[(size_t)x * s.letters + y] = v;
                    s.w[(size_t)y * s.letters + x] = g() % 8 == 0 ? v - 1 : v;
                }
            if (g() % 4 == 0)  // a letter that scores nothing against anything (X)
                for (int x = 0; x < s.letters; ++x) s.w[(size_t)x * s.letters + s.letters - 1] = s.w[(size_t)(s.letters - 1) * s.letters + x] = g() % 2 ? 0 : -1;
            s.go = 6 + (int)(g() % 8);
            s.ge = (int)(g() % 3);
            for (int x = 0; x < s.letters; ++x) alpha[x] = (char)('A' + x);
            alpha[s.letters] = 0;
        }
        static const int kR[4] = {40, 333, 700, 1500}, kCP[4] = {8, 16, 24, 48};
        const int R = kR[g() % 4];
        const int L = 65 + (int)(g() % 120);
        const int CP = kCP[g() % 4], M2 = 16;
        std::vector<uint8_t> ref = g() % 5 == 0 ? random_seq(g, R, "AC") : random_seq(g, R, alpha);
        if (g() % 4 == 0 && R >= 300)  // a second copy of a stretch elsewhere
            std::copy(ref.begin(), ref.begin() + R / 3, ref.begin() + R / 2);
        for (int k = 0; k < 60; ++k) {
            std::vector<uint8_t> q;
            const unsigned kind = g() % 8;
            bool is_plain = false;
            if (kind <= 2 && R > L) {  // sampled with a few edits
                const int p = (int)(g() % (R - L));
                q.assign(ref.begin() + p, ref.begin() + p + L);
                for (int e = (int)(g() % (wide ? 12 : 4)); e > 0; --e) q[g() % L] = wide ? (uint8_t)alpha[g() % s.letters] : (uint8_t)"ACGTN"[g() % 5];
                is_plain = true;
            } else if (kind == 3 && R > 2 * L) {  // long deletion / chimera
                const int p = (int)(g() % (R - 2 * L)), cut = 10 + (int)(g() % (L - 20)), gap = (int)(g() % 90);
                q.assign(ref.begin() + p, ref.begin() + p + cut);
                q.insert(q.end(), ref.begin() + p + cut + gap, ref.begin() + p + cut + gap + (L - cut));
            } else if (kind == 4 && R > L) {  // junk at one end
                const int p = (int)(g() % (R - L)), j = 1 + (int)(g() % (L - 1));
                q = random_seq(g, L, alpha);
                if (g() & 1) std::copy(ref.begin() + p + j, ref.begin() + p + L, q.begin() + j);
                else std::copy(ref.begin() + p, ref.begin() + p + L - j, q.begin());
            } else if (kind == 5) {  // hanging over an end of the reference
                const int kk = 1 + (int)(g() % std::min(L - 1, R));
                q = random_seq(g, L, alpha);
                if (g() & 1) std::copy(ref.end() - kk, ref.end(), q.begin());
                else std::copy(ref.begin(), ref.begin() + kk, q.end() - kk);
            } else if (kind == 6) {
                q = random_seq(g, L, g() & 1 ? "A" : "AC");
            } else {
                q = random_seq(g, L, alpha);
            }
            Dp d;
            Truth t;
            full_dp(s, ref, q, &d, &t);
            const Outcome o = pruned(s, ref, q, d, CP, M2);
            ++checked;
            plain += is_plain;
            if (o.S > t.best) {
                printf("FAIL: a computed score %d exceeds the true maximum %d (iteration %d read %d)\n", o.S, t.best, it, k);
                return 1;
            }
            if (o.pass_score) {
                ++passed;
                plain_passed += is_plain;
                if (o.S != t.best) {
                    printf("FAIL score: passed the checks with %d, truth %d (iteration %d read %d, CP %d, R %d, L %d)\n", o.S, t.best, it, k, CP, R, L);
                    return 1;
                }
            }
            if (o.pass_ends) {
                ++passed_ends;
                if (o.S != t.best || o.row != t.row || o.col != t.col) {
                    printf("FAIL ends: (%d, %d, %d) vs truth (%d, %d, %d) (iteration %d read %d, CP %d)\n", o.S, o.row, o.col, t.best, t.row, t.col, it, k, CP);
                    return 1;
                }
            }
        }
    }
    printf("prune_bounds OK: %ld reads, %ld passed the score checks, %ld the ends checks; plain reads: %ld of %ld passed\n", checked, passed,
           passed_ends, plain_passed, plain);
    if (passed * 10 < checked) {
        printf("FAIL: the checks are vacuous (almost nothing passes)\n");
        return 1;
    }
    return 0;
}
