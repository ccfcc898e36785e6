// align_gapless_cert.cpp — host model of the certificate that lets sw_simd_align's second pass be skipped for a read
// (zoe_amd/csrc/zsw_capi.hip, run_align; zsw_threepass.hip, the classify pass in certificate mode).
//
// sw_simd_align's CIGAR depends on the <T, N> striping only where several optimal alignments exist (E from the
// pre-lazy-F H, simd_correct_and_set_left on F == H, the vector-wide any()). Certificate, from quantities the seeded passes leave:
//   (1) the maximum S of the matrix sits in exactly one cell (re - 1, qe - 1)               [forward seeded pass, mode 3];
//   (2) the maximum of the reversed matrix sits in exactly one cell, (rs, qs) turned round   [reverse seeded pass, mode 3]
//       — every alignment that scores S then starts in (rs, qs) and ends in (re - 1, qe - 1);
//   (3) re - rs == qe - qs =: n and the weights of the diagonal from (rs, qs) add up to S;
//   (4) no other path between these corners reaches S: with three or more gap runs it has at most n - 1 pairs and pays 3 * gap_open
//       (S > maxw * (n - 1) - 3 * gap_open rules them out); with two runs it has an insertion and a deletion of the same length k,
//       in either order, and the pairs between them lie on the diagonal k away — for every k that the potential alone does not
//       rule out (maxw * (n - k) - 2 * gap_open - 2 * (k - 1) * gap_extend >= S) the best placement of the two runs is one sweep
//       over prefix sums of the two diagonals, and it must stay below S. (Until the middle of round 4 the condition was the
//       potential bound for k = 1 alone, S > maxw * (n - 1) - 2 * gap_open: three substitutions at 2 / -5, -10 / -1; now four.)
// Then the diagonal is the ONLY alignment scoring S, and every exact algorithm returns it: the oracle's literal sw_simd_align
// (oracle/zoe_oracle.hpp, the restated striped.rs:449-598) must return [qs S][n M][len - qe S] with these ranges at every lane
// count. Checked for N = 2 .. 64, signed 16-bit and 8-bit lanes, on pairs built to be near the threshold (few mismatches, N's,
// homopolymer runs, repeats in the reference). usage: align_gapless_cert <iterations> <seed>
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
bool returns_diagonal(const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge, int S,
                      int rs, int re, int qs, int qe) {
    auto prof = StripedProfile<T, N>::make(q.data(), q.size(), pw, map, -go, -ge);
    const Maybe<Alignment> a = sw_simd_align<T, N>(ref.data(), ref.size(), prof);
    if (a.status != SOME) return a.status == OVERFLOWED;  // (an overflowing width answers at the next one)
    AlignmentStates want;
    want.soft_clip((size_t)qs);
    want.add_ciglet({(size_t)(qe - qs), 'M'});
    want.soft_clip(q.size() - (size_t)qe);
    return (int)a.value.score == S && (int)a.value.ref_start == rs && (int)a.value.ref_end == re && (int)a.value.query_start == qs &&
           (int)a.value.query_end == qe && a.value.states == want;
}


// The same pair with the roles swapped, as the shared-profile role sees it (zsw_capi_shared.hip): the profile is striped over the
// reference-side sequence, the read supplies the rows; the only optimal alignment is the same diagonal, the clipped ends those of
// the long sequence.
template <typename T, int N>
bool returns_diagonal_swapped(const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge,
                              int S, int rs, int re, int qs, int qe) {
    auto prof = StripedProfile<T, N>::make(ref.data(), ref.size(), pw, map, -go, -ge);
    const Maybe<Alignment> a = sw_simd_align<T, N>(q.data(), q.size(), prof);
    if (a.status != SOME) return a.status == OVERFLOWED;
    AlignmentStates want;
    want.soft_clip((size_t)rs);
    want.add_ciglet({(size_t)(re - rs), 'M'});
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
    long pairs = 0, certified = 0, unique_both = 0, swept = 0;
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
            const int L = rnd(6, 60);
            std::vector<uint8_t> q;
            int p = rnd(0, R - L);
            for (int i = 0; i < L; ++i) {
                const int e = rnd(0, 99);
                uint8_t b = ref[std::min(p, R - 1)];
                if (e < 4) b = keys[rnd(0, 3)];            // a substitution
                else if (e < 5) b = 'N';
                else if (e == 5 && k % 3 == 0) { ++p; }     // now and then a deletion ...
                else if (e == 6 && k % 3 == 0) { q.push_back(keys[rnd(0, 3)]); }  // ... or an insertion (such reads must not be certified)
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
            const int n = re - rs;
            if (n <= 0 || qe - qs != n) continue;
            long sum = 0;
            for (int i = 0; i < n; ++i) sum += wm.w[map.to_index(ref[rs + i])][map.to_index(q[qs + i])];
            if (sum != S) continue;
            // (4) no other path between the corners reaches S. Three or more gap runs: at most n - 1 pairs and 3 * gap_open. Two runs:
            // an insertion and a deletion of the same length k (anything else ends on another diagonal), in either order — the pairs
            // between the runs lie on the diagonal k rows below / above; for every k that the potential does not rule out, the best
            // placement (i, j) of the runs is one sweep over prefix sums.
            if (!((long)S > (long)sc.match * (n - 1) - 3l * sc.go)) continue;
            bool two_runs_below = true;
            int checked_k = 0;
            for (int k = 1; k < n && two_runs_below; ++k) {
                if ((long)sc.match * (n - k) - 2l * sc.go - 2l * sc.ge * (k - 1) < (long)S) break;
                ++checked_k;
                for (int dir = 0; dir < 2 && two_runs_below; ++dir) {
                    // dir 0: deletion first (rows shifted by +k between the runs), dir 1: insertion first (columns shifted by +k)
                    auto wsh = [&](int t) {  // pair t of the shifted stretch
                        const int r = rs + t + (dir == 0 ? k : 0), c = qs + t + (dir == 0 ? 0 : k);
                        return (r < (int)ref.size() && c < (int)q.size()) ? (long)wm.w[map.to_index(ref[r])][map.to_index(q[c])] : -(1l << 30);
                    };
                    auto w0 = [&](int t) { return (long)wm.w[map.to_index(ref[rs + t])][map.to_index(q[qs + t])]; };
                    // path: pairs 0 .. i-1 on the diagonal, run 1 (k), shifted pairs i .. j-1, run 2 (k), diagonal pairs j+k .. n-1
                    // score = S - 2go - 2ge(k-1) + [Q(j) - P0(j+k)] - [Q(i) - P0(i)],  1 <= i <= j <= n - k - 1 (a pair before run 1 and after run 2)
                    std::vector<long> P0(n + 1, 0), Q(n + 1, 0);
                    for (int t = 0; t < n; ++t) P0[t + 1] = P0[t] + w0(t);
                    for (int t = 0; t + k < n + k && t < n; ++t) Q[t + 1] = Q[t] + wsh(t);
                    long best_alt = -(1l << 40), low = 1l << 40;
                    for (int j = 1; j + k <= n - 1; ++j) {
                        low = std::min(low, Q[j] - P0[j]);  // i = j allowed: no pairs between the runs
                        best_alt = std::max(best_alt, Q[j] - P0[j + k] - low);
                    }
                    if (best_alt > -(1l << 39) && (long)S - 2l * sc.go - 2l * sc.ge * (k - 1) + best_alt >= (long)S) two_runs_below = false;
                }
            }
            if (!two_runs_below) continue;
            if (checked_k) ++swept;
            ++certified;
            const bool ok = returns_diagonal<int16_t, 2>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int16_t, 4>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int16_t, 8>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int16_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int16_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int16_t, 64>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int8_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal<int8_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal_swapped<int16_t, 4>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal_swapped<int16_t, 16>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal_swapped<int16_t, 64>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe) &&
                            returns_diagonal_swapped<int8_t, 32>(ref, q, pw, map, sc.go, sc.ge, S, rs, re, qs, qe);
            if (!ok) {
                printf("certified read whose striped alignment is not the diagonal: S %d ref [%d,%d) query [%d,%d) scheme %d/%d/%d/%d\n  ref  ", S, rs, re, qs, qe, sc.match,
                       sc.mismatch, sc.go, sc.ge);
                for (uint8_t x : ref) putchar(x);
                printf("\n  read ");
                for (uint8_t x : q) putchar(x);
                printf("\n");
                return 1;
            }
        }
    }
    printf("pairs %ld, both maxima in one cell %ld, certified %ld (two-run sweeps needed for %ld)\n", pairs, unique_both, certified, swept);
    if (certified * 6 < pairs) {
        printf("the certificate is vacuous: fewer than a sixth of the pairs get one\n");
        return 1;
    }
    printf("align_gapless_cert OK\n");
    return 0;
}
