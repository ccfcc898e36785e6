// align_onegap_cert.cpp — host model of the second certificate that lets sw_simd_align's second pass be skipped for a read: the
// read's only optimal alignment has exactly ONE gap run (zoe_amd/csrc/zsw_threepass.hip, the classify pass in certificate mode;
// the gapless case: align_gapless_cert.cpp).
//
// From the two seeded passes: (1) the maximum S of the matrix sits in exactly one cell (re - 1, qe - 1), (2) the maximum of the
// reversed matrix sits in exactly one cell, (rs, qs) turned round — every alignment that scores S runs from (rs, qs) to
// (re - 1, qe - 1). Let rlen = re - rs, qlen = qe - qs, g = |rlen - qlen| >= 1, m = min(rlen, qlen).
//   (3) Alignments between these corners with ONE gap run are: p pairs on the first diagonal, the run of g, the other m - p pairs
//       on the second diagonal, p = 1 .. m - 1. Their scores are P0(p) + (T1 - P1(p)) - gap_open - (g - 1) * gap_extend with the
//       prefix sums P0 / P1 of the two diagonals: one sweep. One p must reach S — or several ADJACENT ones (a gap inside a
//       homopolymer run or a short repeat: the placements are the same alignment shifted along the run). Then the alignments that
//       score S differ only in where the run sits, and the walk from the end (backtrack.rs:290-342) takes a gap as soon as one
//       ends an optimal path — E == H or F == H is tested before the diagonal — i.e. the LAST placement; the values along an
//       optimal path are exact at every striping (E opens from an H that a diagonal step produced; the lazy-F loop does not
//       stop while the lane that carries the run still has F > H - gap_open), so that choice does not depend on <T, N>.
//   (4) An alignment between these corners with three or more gap runs has at most m pairs and pays at least 3 * gap_open +
//       max(g - 3, 0) * gap_extend: S must lie beyond maxw * m minus that. One with TWO runs — signed lengths a and rlen - qlen - a,
//       i pairs on the first diagonal, the first run, j - i pairs on the diagonal a away, the second run, the rest on the last
//       diagonal — scores P0(i) + Pa(j) - Pa(i) + Pz(M) - Pz(j) - cost(a): for every a that the potential does not rule out,
//       the best (i, j) is one sweep over the prefix sums of the three diagonals, and it must stay below S. (Until the middle
//       of round 4: the potential bound alone, S > maxw * m - 2 * gap_open - max(g - 2, 0) * gap_extend — one substitution
//       beside the gap at 2 / -5, -10 / -1; now two, or one and a few N.)
// Then that alignment is the ONLY one scoring S and every exact algorithm returns it: the oracle's literal sw_simd_align
// (oracle/zoe_oracle.hpp, the restated striped.rs:449-598) must return [qs S][p M][g D|I][m - p M][len - qe S] at every lane count.
// Checked for N = 2 .. 64 in 16-bit lanes and N = 16, 32 in 8-bit lanes under ten schemes, on pairs with one indel and few other
// errors, gaps inside homopolymer runs and repeats (tied placements: 4 in 10 of the certified pairs). usage: align_onegap_cert <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/zoe_oracle.hpp"

using namespace zor;

namespace {

struct Cells {
    int best = 0, n = 0, r = -1, c = -1;
};

// plain Gotoh; the maximum, how many cells hold it, and one of them
Cells gotoh(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, const WeightMatrixI8& wm, const ByteIndexMap& map, int go, int ge) {
    const int R = (int)a.size(), L = (int)b.size();
    std::vector<int> H(L + 1, 0), E(L + 1, 0);
    Cells out;
    for (int r = 0; r < R; ++r) {
        int diag = 0, f = 0;
        for (int c = 1; c <= L; ++c) {
            const int e = std::max(std::max(E[c] - ge, H[c] - go), 0);
            f = std::max(std::max(f - ge, H[c - 1] - go), 0);
            const int h = std::max(std::max(diag + wm.w[map.to_index(a[r])][map.to_index(b[c - 1])], e), std::max(f, 0));
            diag = H[c];
            H[c] = h;
            E[c] = e;
            if (h > out.best) {
                out.best = h;
                out.n = 1;
                out.r = r;
                out.c = c - 1;
            } else if (h == out.best && h > 0) {
                ++out.n;
            }
        }
    }
    return out;
}

template <typename T, int N>
bool returns_onegap(const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge, int S,
                    int rs, int re, int qs, int qe, int p, int g, char op) {
    auto prof = StripedProfile<T, N>::make(q.data(), q.size(), pw, map, -go, -ge);
    const Maybe<Alignment> a = sw_simd_align<T, N>(ref.data(), ref.size(), prof);
    if (a.status != SOME) return a.status == OVERFLOWED;  // (an overflowing width answers at the next one)
    const int m = std::min(re - rs, qe - qs);
    AlignmentStates want;
    want.soft_clip((size_t)qs);
    want.add_ciglet({(size_t)p, 'M'});
    want.add_ciglet({(size_t)g, (uint8_t)op});
    want.add_ciglet({(size_t)(m - p), 'M'});
    want.soft_clip(q.size() - (size_t)qe);
    return (int)a.value.score == S && (int)a.value.ref_start == rs && (int)a.value.ref_end == re && (int)a.value.query_start == qs &&
           (int)a.value.query_end == qe && a.value.states == want;
}


// The same pair with the roles swapped, as the shared-profile role sees it (zsw_capi_shared.hip): the profile is striped over the
// reference-side sequence, the read supplies the rows. The matrix is the transpose; its only optimal alignment is the transpose of
// the certified one (insertions and deletions trade places, the clipped ends are those of the long sequence), and among adjacent
// tied placements the walk from the end meets the same one first.
template <typename T, int N>
bool returns_onegap_swapped(const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge, int S,
                            int rs, int re, int qs, int qe, int p, int g, char op) {
    auto prof = StripedProfile<T, N>::make(ref.data(), ref.size(), pw, map, -go, -ge);
    const Maybe<Alignment> a = sw_simd_align<T, N>(q.data(), q.size(), prof);
    if (a.status != SOME) return a.status == OVERFLOWED;
    const int m = std::min(re - rs, qe - qs);
    AlignmentStates want;
    want.soft_clip((size_t)rs);
    want.add_ciglet({(size_t)p, 'M'});
    want.add_ciglet({(size_t)g, (uint8_t)(op == 'D' ? 'I' : 'D')});
    want.add_ciglet({(size_t)(m - p), 'M'});
    want.soft_clip(ref.size() - (size_t)re);
    return (int)a.value.score == S && (int)a.value.ref_start == qs && (int)a.value.ref_end == qe && (int)a.value.query_start == rs &&
           (int)a.value.query_end == re && a.value.states == want;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    const uint8_t keys[5] = {'A', 'C', 'G', 'T', 'N'};
    const ByteIndexMap map = ByteIndexMap::make(keys, 5, 'N', true);
    struct Sch {
        int match, mismatch, go, ge;
    };
    const Sch schemes[] = {{2, -5, 10, 1}, {1, -1, 2, 1}, {3, -2, 5, 1}, {1, -3, 5, 2}, {5, -4, 8, 1}, {2, -2, 3, 3}, {4, -6, 12, 2}, {2, -10, 10, 1}, {1, -1, 1, 1}, {3, -1, 1, 0}};
    long pairs = 0, certified = 0, unique_both = 0, tied = 0, swept = 0;
    for (int it = 0; it < iters; ++it) {
        const Sch& sc = schemes[it % (sizeof(schemes) / sizeof(schemes[0]))];
        const WeightMatrixI8 wm = WeightMatrixI8::make(map, (int8_t)sc.match, (int8_t)sc.mismatch, 'N');
        const ProfileWeights pw = ProfileWeights::from(wm, true);
        const int R = rnd(60, 300);
        std::vector<uint8_t> ref(R);
        const int letters = it % 7 == 0 ? 2 : 4;
        for (auto& x : ref) x = keys[rnd(0, letters - 1)];
        if (rnd(0, 2) == 0) {  // a second copy of a stretch
            const int len = rnd(10, 40), from = rnd(0, R - len), to = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[to + i] = ref[from + i];
        }
        if (rnd(0, 2) == 0) {  // homopolymer / short tandem runs
            const int unit = rnd(1, 3), len = rnd(6, 20), at = rnd(0, R - len);
            for (int i = unit; i < len; ++i) ref[at + i] = ref[at + i - unit];
        }
        for (int k = 0; k < 12; ++k) {
            const int L = rnd(12, 70);
            std::vector<uint8_t> q;
            int p = rnd(0, std::max(0, R - L - 8));
            const int indel_at = rnd(3, L - 4), indel_len = (rnd(0, 1) ? 1 : -1) * (rnd(0, 3) ? 1 : rnd(2, 5));
            for (int i = 0; i < L; ++i) {
                const int e = rnd(0, 99);
                uint8_t b = ref[std::min(p, R - 1)];
                if (e < 4) b = keys[rnd(0, 3)];            // a substitution
                else if (e < 5) b = 'N';
                else if (i == indel_at && indel_len < 0) { p += -indel_len; }                                  // the deletion ...
                else if (i == indel_at && indel_len > 0) { for (int x = 0; x < indel_len; ++x) q.push_back(keys[rnd(0, 3)]); }  // ... or the insertion
                else if (e == 5 && k % 5 == 0) { ++p; }     // now and then a second gap (such reads must not be certified)
                q.push_back(b);
                ++p;
            }
            q.resize(L);
            if (k % 4 == 1) {  // junk ends: the alignment is clipped
                for (int i = 0; i < rnd(1, 6); ++i) q[i] = keys[rnd(0, 3)];
                for (int i = 0; i < rnd(1, 6); ++i) q[L - 1 - i] = keys[rnd(0, 3)];
            }
            ++pairs;
            // rows = reference, columns = query, as in striped.rs
            const Cells fwd = gotoh(ref, q, wm, map, sc.go, sc.ge);
            if (fwd.best == 0 || fwd.n != 1) continue;
            std::vector<uint8_t> rref(ref.rbegin(), ref.rend()), rq(q.rbegin(), q.rend());
            const Cells rev = gotoh(rref, rq, wm, map, sc.go, sc.ge);
            if (rev.best != fwd.best || rev.n != 1) continue;
            ++unique_both;
            const int S = fwd.best, re = fwd.r + 1, qe = fwd.c + 1, rs = R - 1 - rev.r, qs = L - 1 - rev.c;
            const int rlen = re - rs, qlen = qe - qs;
            if (rlen <= 0 || qlen <= 0 || rlen == qlen) continue;
            const int g = std::abs(rlen - qlen), m = std::min(rlen, qlen);
            const bool del = rlen > qlen;  // the run consumes reference rows
            auto wt = [&](int r, int c) { return (long)wm.w[map.to_index(ref[r])][map.to_index(q[c])]; };
            long t1 = 0;
            for (int i = 0; i < m; ++i) t1 += del ? wt(rs + g + i, qs + i) : wt(rs + i, qs + g + i);
            long p0 = 0, p1 = 0, best = -(1l << 40);
            int best_p = -1, n_best = 0, first_p = -1;
            for (int p = 1; p < m; ++p) {
                p0 += wt(rs + p - 1, qs + p - 1);
                p1 += del ? wt(rs + g + p - 1, qs + p - 1) : wt(rs + p - 1, qs + g + p - 1);
                const long s = p0 + (t1 - p1) - sc.go - (long)(g - 1) * sc.ge;
                if (s > best) {
                    best = s;
                    best_p = first_p = p;
                    n_best = 1;
                } else if (s == best) {
                    ++n_best;
                    best_p = p;  // the LAST placement that reaches the best
                }
            }
            if (best != S || best_p - first_p != n_best - 1) continue;  // one placement, or a run of adjacent ones
            // (4) nothing with more runs reaches S. Three or more runs: at most m pairs, 3 * gap_open and, if all of them make up g,
            // (g - 3) extensions. Two runs: signed lengths a and b = gs - a (gs = rlen - qlen: a deletion counts +, an insertion -),
            // i pairs on the first diagonal, the run a, j - i pairs on the diagonal a away, the run b, the rest on the last diagonal;
            // for every a that the potential does not rule out, the best (i, j) is one sweep over the prefix sums of three diagonals.
            if (!((long)S > (long)sc.match * m - 3l * sc.go - (long)std::max(g - 3, 0) * sc.ge)) continue;
            bool two_runs_below = true;
            int swept_here = 0;
            const int gs = rlen - qlen;
            for (int a = -(m + g); a <= m + g && two_runs_below; ++a) {
                const int b = gs - a;
                if (a == 0 || b == 0) continue;
                const int ap = std::max(a, 0), an = std::max(-a, 0), bp = std::max(b, 0), bn = std::max(-b, 0);
                const int M = rlen - ap - bp;  // pairs (= qlen - an - bn)
                if (M < 2) continue;
                const long cost = 2l * sc.go + (long)sc.ge * (std::abs(a) + std::abs(b) - 2);
                if ((long)sc.match * M - cost < (long)S) continue;  // the potential rules this pair of runs out
                ++swept_here;
                // score(i, j) = P0(i) + Pa(j) - Pa(i) + Pz(M) - Pz(j) - cost,  1 <= i <= j <= M - 1
                long p0 = 0, pa = 0, pz = 0, pzM = 0;
                for (int t = 0; t < M; ++t) pzM += wt(rs + t + ap + bp, qs + t + an + bn);
                long low = 1l << 40, best_alt = -(1l << 40);
                for (int j = 1; j <= M - 1; ++j) {
                    p0 += wt(rs + j - 1, qs + j - 1);
                    pa += wt(rs + j - 1 + ap, qs + j - 1 + an);
                    pz += wt(rs + j - 1 + ap + bp, qs + j - 1 + an + bn);
                    low = std::min(low, pa - p0);                       // i = j: no pairs between the runs
                    best_alt = std::max(best_alt, pa - pz - low);
                }
                if (pzM - cost + best_alt >= (long)S) two_runs_below = false;
            }
            if (!two_runs_below) continue;
            if (swept_here) ++swept;
            ++certified;
            if (n_best > 1) ++tied;
            const char op = del ? 'D' : 'I';
            const bool ok = returns_onegap<int16_t, 2>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int16_t, 4>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int16_t, 8>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int16_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int16_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int16_t, 64>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int8_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap<int8_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap_swapped<int16_t, 4>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap_swapped<int16_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap_swapped<int16_t, 64>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op) &&
                            returns_onegap_swapped<int8_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe, best_p, g, op);
            if (!ok) {
                printf("certified read whose striped alignment is not the one-gap alignment: S %d ref [%d,%d) query [%d,%d) p %d g %d %c scheme %d/%d/%d/%d\n  ref  ", S, rs, re,
                       qs, qe, best_p, g, op, sc.match, sc.mismatch, sc.go, sc.ge);
                for (uint8_t x : ref) putchar(x);
                printf("\n  read ");
                for (uint8_t x : q) putchar(x);
                printf("\n");
                return 1;
            }
        }
    }
    printf("pairs %ld, both maxima in one cell %ld, certified %ld, of which with tied placements %ld, with two-run sweeps %ld\n", pairs, unique_both, certified, tied, swept);
    if (certified * 12 < pairs) {
        printf("the certificate is vacuous: fewer than a twelfth of the pairs get one\n");
        return 1;
    }
    printf("align_onegap_cert OK\n");
    return 0;
}
