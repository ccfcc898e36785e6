// reverse_unique.cpp — host model of the claim behind the shared role's seeded reverse pass (zoe_amd/csrc/zsw_capi_shared.hip,
// run_ranges_shared / settle_reverse_kernel).
//
// sw_simd_score_ranges (striped.rs:355-388) finds the starts of the alignment with a second score pass over the REVERSED PREFIXES
// reverse(a[..a_end]) x reverse(b[..b_end]). The GPU path runs that pass over the whole reversed sequences instead (a seeded,
// banded pass needs no prefix lengths) and accepts its answer for a read only if
//   (1) the forward maximum S sits in exactly one cell (a_end - 1, b_end - 1) of the matrix, and
//   (2) the maximum of the whole reversed matrix sits in exactly one cell too.
// Claim checked here against plain Gotoh matrices: under (1) the cells holding S in the reversed matrix of the prefixes and in the
// reversed matrix of the whole sequences are the same cells (as positions of a and b) — every alignment scoring S ends in the one
// forward cell, so it lies inside the prefixes — hence under (2) that cell is the restricted pass's answer under any tie rule.
// usage: reverse_unique <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <utility>
#include <vector>

namespace {

struct Scheme {
    int S;
    std::vector<int> w;  // w[x * S + y]: residue x of `a` against residue y of `b` (asymmetric matrices allowed)
    int go, ge;          // positive magnitudes
};

// local-alignment H of a (rows) x b (columns); returns the maximum and the 0-based cells holding it
int gotoh(const Scheme& s, const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, std::vector<std::pair<int, int>>* cells) {
    const int R = (int)a.size(), L = (int)b.size();
    std::vector<int> H((size_t)(R + 1) * (L + 1), 0), E = H, F = H;
    auto at = [&](std::vector<int>& v, int r, int c) -> int& { return v[(size_t)r * (L + 1) + c]; };
    int best = 0;
    cells->clear();
    for (int r = 1; r <= R; ++r)
        for (int c = 1; c <= L; ++c) {
            at(E, r, c) = std::max(0, std::max(at(E, r - 1, c) - s.ge, at(H, r - 1, c) - s.go));
            at(F, r, c) = std::max(0, std::max(at(F, r, c - 1) - s.ge, at(H, r, c - 1) - s.go));
            const int h = std::max(std::max(0, at(H, r - 1, c - 1) + s.w[a[r - 1] * s.S + b[c - 1]]), std::max(at(E, r, c), at(F, r, c)));
            at(H, r, c) = h;
            if (h > best) {
                best = h;
                cells->clear();
            }
            if (h == best && h > 0) cells->push_back({r - 1, c - 1});
        }
    return best;
}

std::vector<uint8_t> reversed(const std::vector<uint8_t>& v, size_t n) { return std::vector<uint8_t>(v.rend() - (long)n, v.rend()); }

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937 g(argc > 2 ? (unsigned)atoll(argv[2]) : 1u);
    auto rnd = [&](int lo, int hi) { return lo + (int)(g() % (unsigned)(hi - lo + 1)); };
    long checked = 0, forward_unique = 0, both_unique = 0;
    for (int it = 0; it < iters; ++it) {
        Scheme s;
        s.S = it % 4 == 3 ? rnd(6, 12) : 5;
        s.w.assign((size_t)s.S * s.S, 0);
        const int ma = rnd(1, 5), mi = -rnd(0, 5);
        for (int x = 0; x < s.S; ++x)
            for (int y = 0; y < s.S; ++y) s.w[x * s.S + y] = x == y ? ma + (s.S > 5 ? rnd(0, 3) : 0) : mi + (it % 5 == 0 ? rnd(-1, 1) : 0);
        if (s.S == 5)
            for (int x = 0; x < 5; ++x) s.w[x * 5 + 4] = s.w[4 * 5 + x] = rnd(-1, 1) * (it % 3 == 0);  // N
        s.go = rnd(0, 8);
        s.ge = rnd(0, std::max(s.go, 1));
        const int R = rnd(20, 90), L = rnd(8, 50), letters = it % 6 == 0 ? 2 : (s.S == 5 ? 4 : s.S);
        std::vector<uint8_t> a(R), b;
        for (auto& x : a) x = (uint8_t)rnd(0, letters - 1);
        if (rnd(0, 2) == 0 && R > 30) std::copy(a.begin(), a.begin() + R / 3, a.begin() + R / 2);  // a repeat
        if (rnd(0, 3) != 0 && R > L) {  // a piece of `a` with a few edits
            const int p = rnd(0, R - L);
            b.assign(a.begin() + p, a.begin() + p + L);
            for (int e = rnd(0, 4); e > 0; --e) b[rnd(0, L - 1)] = (uint8_t)rnd(0, letters - 1);
            if (rnd(0, 3) == 0) b.erase(b.begin() + rnd(1, L - 2));
        } else {
            b.resize(L);
            for (auto& x : b) x = (uint8_t)rnd(0, letters - 1);
        }
        std::vector<std::pair<int, int>> fwd, rr, rw;
        const int S = gotoh(s, a, b, &fwd);
        ++checked;
        if (S == 0 || fwd.size() != 1) continue;
        ++forward_unique;
        const int ae = fwd[0].first + 1, be = fwd[0].second + 1;  // exclusive ends
        // the reversed problem: matrix entries transposed with the roles kept (a stays the row sequence)
        const int Sr = gotoh(s, reversed(a, ae) , reversed(b, be), &rr);   // restricted to the prefixes (striped.rs:355-388)
        const int Sw = gotoh(s, reversed(a, a.size()), reversed(b, b.size()), &rw);  // whole sequences
        if (Sr != S || Sw != S) {
            printf("FAIL: reverse maxima %d (prefixes) / %d (whole) differ from the forward score %d (iteration %d)\n", Sr, Sw, S, it);
            return 1;
        }
        std::set<std::pair<int, int>> A, B;  // as 0-based start positions in a and b
        for (auto& c : rr) A.insert({ae - 1 - c.first, be - 1 - c.second});
        for (auto& c : rw) B.insert({(int)a.size() - 1 - c.first, (int)b.size() - 1 - c.second});
        if (A != B) {
            printf("FAIL: forward maximum in one cell, but the reversed matrices hold the score in different cells (%zu vs %zu; iteration %d)\n",
                   A.size(), B.size(), it);
            return 1;
        }
        if (B.size() == 1) ++both_unique;
    }
    printf("reverse_unique OK: %ld pairs, forward maximum in one cell %ld, reversed maximum in one cell as well %ld\n", checked, forward_unique, both_unique);
    if (forward_unique * 4 < checked) {
        printf("FAIL: the check is vacuous (few pairs have a unique forward maximum)\n");
        return 1;
    }
    return 0;
}
