// seed_warmup.cpp — host model of the late start of the alignment's second pass (zoe_amd/csrc/zsw_align_pk_kernel.hpp) under
// the certificate of zsw_seed.hpp::seed_safe_start. Claim: if r0 = seed_safe_start(...) >= 0, then Zoe's own striped alignment
// (oracle/zoe_oracle.hpp, sw_simd_align at <i16, N>) of the read against reference[r0..] — i.e. the striped recurrence started
// with a zero state at row r0 — is the alignment against the whole reference, shifted by r0: same score, same ranges, same
// CIGAR. Reads, references and scoring schemes as in seed_bounds.cpp; lane counts 4, 8, 16.
// usage: seed_warmup <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/zoe_oracle.hpp"
#include "../../zoe_amd/csrc/zsw_seed.hpp"
#include "adversarial_reads.hpp"

using namespace zor;

namespace {

template <int N>
bool same_alignment(const uint8_t* ref, size_t R, const uint8_t* q, size_t L, const ProfileWeights& pw, const ByteIndexMap& map, int go, int ge,
                    int r0, bool* mapped) {
    auto prof = StripedProfile<int16_t, N>::make(q, L, pw, map, -go, -ge);
    const Maybe<Alignment> full = sw_simd_align<int16_t, N>(ref, R, prof);
    const Maybe<Alignment> late = sw_simd_align<int16_t, N>(ref + r0, R - (size_t)r0, prof);
    *mapped = full.status == SOME;
    if (full.status != late.status) return false;
    if (full.status != SOME) return true;
    const Alignment &a = full.value, &b = late.value;
    return a.score == b.score && a.ref_start == b.ref_start + (size_t)r0 && a.ref_end == b.ref_end + (size_t)r0 && a.query_start == b.query_start &&
           a.query_end == b.query_end && a.states == b.states;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    const uint8_t keys[5] = {'A', 'C', 'G', 'T', 'N'};
    const ByteIndexMap map = ByteIndexMap::make(keys, 5, 'N', true);
    struct Sch {
        int match, mismatch, go, ge;
    };
    const Sch schemes[] = {{2, -5, 10, 1}, {1, -1, 2, 1}, {3, -2, 5, 1}, {1, -3, 5, 2}, {5, -4, 8, 1}, {2, -2, 3, 3}, {4, -6, 12, 2},
                           {2, -10, 10, 1}, {2, -5, 5, 1}};  // the last two: mismatch loss >= gap_open (lambda = gap_open)
    long reads = 0, certified = 0, late_rows = 0, structured = 0;
    for (int it = 0; it < iters; ++it) {
        const Sch& sc = schemes[it % (sizeof(schemes) / sizeof(schemes[0]))];
        const WeightMatrixI8 wm = WeightMatrixI8::make(map, (int8_t)sc.match, (int8_t)sc.mismatch, 'N');
        const ProfileWeights pw = ProfileWeights::from(wm, true);
        const int R = rnd(120, 500);
        std::vector<uint8_t> ref(R);
        for (auto& x : ref) x = keys[rnd(0, 3)];
        if (rnd(0, 2) == 0) {  // a second copy of a segment
            const int len = rnd(20, 60), from = rnd(0, R - len), to = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[to + i] = ref[from + i];
        }
        if (rnd(0, 3) == 0) {  // tandem repeat
            const int unit = rnd(1, 6), len = rnd(20, 60), at = rnd(0, R - len);
            for (int i = unit; i < len; ++i) ref[at + i] = ref[at + i - unit];
        }
        if (rnd(0, 3) == 0)
            for (int k = rnd(1, 5); k > 0; --k) ref[rnd(0, R - 1)] = 'N';
        std::vector<uint8_t> res(R);
        bool ref_has[32] = {false};
        for (int i = 0; i < R; ++i) {
            res[i] = map.to_index(ref[i]);
            ref_has[res[i]] = true;
        }
        int32_t w[25];
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) w[i * 5 + j] = wm.w[i][j];
        zsw::SeedParams p;
        const int K = rnd(4, 6);
        if (!zsw::seed_analyze(5, w, sc.go, sc.ge, ref_has, K, &p)) continue;
        p.M1 = rnd(2, 24);
        p.M1_per8 = rnd(0, 2);
        p.M2 = rnd(2, 14);
        p.Dn = rnd(0, 4);
        p.tol = rnd(0, 5);
        std::vector<uint32_t> table((size_t)2 << (2 * K), 0);
        zsw::seed_index_build(p, res.data(), (uint64_t)R, table.data());
        // one read against one reference (and its index): a certificate must not change the alignment. Returns false on a violation.
        auto check = [&](const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, const zsw::SeedParams& p, const std::vector<uint32_t>& table) -> bool {
            const int R = (int)ref.size(), L = (int)q.size();
            ++reads;
            auto cell = [&](int c) { return zsw::seed_cell(p, (int)map.to_index(q[c])); };
            auto look = [&](uint32_t code, uint32_t* f1, uint32_t* l1) {
                *f1 = table[2 * (size_t)code];
                *l1 = table[2 * (size_t)code + 1];
            };
            const zsw::SeedRead sr = zsw::seed_read(p, L, cell, look);
            if (!sr.ok) return true;
            // the read's score (any lane count: the score is layout-invariant)
            auto prof = StripedProfile<int16_t, 8>::make(q.data(), (size_t)L, pw, map, -sc.go, -sc.ge);
            uint32_t S = 0;
            if (sw_simd_score<int16_t, 8>(ref.data(), (size_t)R, prof, &S) != SOME) return true;
            const int r0 = zsw::seed_safe_start(p, sr.t_all, sr.d_fa, sr.dt, (int)S);
            if (r0 < 0) return true;
            if (r0 >= R) return true;
            ++certified;
            late_rows += r0;
            bool mapped = false;
            const bool ok = same_alignment<4>(ref.data(), (size_t)R, q.data(), (size_t)L, pw, map, sc.go, sc.ge, r0, &mapped) &&
                            same_alignment<8>(ref.data(), (size_t)R, q.data(), (size_t)L, pw, map, sc.go, sc.ge, r0, &mapped) &&
                            same_alignment<16>(ref.data(), (size_t)R, q.data(), (size_t)L, pw, map, sc.go, sc.ge, r0, &mapped);
            if (!ok) {
                printf("late start changes the alignment: r0 %d, S %u, t_all %d, d_fa %d, dt %d, go %d ge %d match %d mismatch %d K %d\n  ref  ", r0, S,
                       sr.t_all, sr.d_fa, sr.dt, sc.go, sc.ge, sc.match, sc.mismatch, K);
                for (uint8_t x : ref) putchar(x);
                printf("\n  read ");
                for (uint8_t x : q) putchar(x);
                printf("\n");
                return false;
            }
            return true;
        };
        for (int k = 0; k < 40; ++k) {
            const int L = rnd(K + 4, std::min(R, 100));
            std::vector<uint8_t> q;
            const int kind = rnd(0, 9);
            int i = rnd(0, std::max(0, R - L));
            const int sub = kind < 6 ? rnd(0, 4) : rnd(5, 15), indel = kind < 6 ? rnd(0, 1) : rnd(1, 4);
            while ((int)q.size() < L) {
                uint8_t b = i < R ? ref[i] : keys[rnd(0, 3)];
                const int e = rnd(0, 999);
                if (e < sub * 10) b = keys[rnd(0, 3)];
                else if (e < sub * 10 + indel * 5) { ++i; continue; }
                else if (e < sub * 10 + indel * 10) { q.push_back(keys[rnd(0, 3)]); continue; }
                q.push_back(b);
                ++i;
            }
            if (kind == 9) {  // junk prefix: the alignment starts late in the read
                for (int j = 0; j < L / 3; ++j) q[j] = keys[rnd(0, 3)];
            }
            if (kind == 8 && L > 30) {  // a long deletion in the middle
                const int cut = L / 2, skip = rnd(3, 25);
                int j = i - (L - cut) + skip;
                for (int c = cut; c < L; ++c, ++j) q[c] = (j >= 0 && j < R) ? ref[j] : keys[rnd(0, 3)];
            }
            if (!check(ref, q, p, table)) return 1;
        }
        for (int k = 0; k < 16; ++k) {  // structured cases (adversarial_reads.hpp), each with a reference and an index of its own
            zsw::SeedParams pa;
            bool has[32] = {false};
            has[0] = has[1] = has[2] = has[3] = true;
            if (!zsw::seed_analyze(5, w, sc.go, sc.ge, has, K, &pa)) break;
            pa.M1 = p.M1;
            pa.M1_per8 = p.M1_per8;
            pa.M2 = p.M2;
            pa.Dn = p.Dn;
            pa.tol = p.tol;
            std::vector<uint8_t> ares, aq;
            if (!adversarial::spacer_case(rng, pa, rnd(2 * (K + pa.spacer), 100), &ares, &aq)) continue;
            std::vector<uint32_t> atable((size_t)2 << (2 * K), 0);
            zsw::seed_index_build(pa, ares.data(), (uint64_t)ares.size(), atable.data());
            std::vector<uint8_t> aref(ares.size()), aread(aq.size());
            for (size_t i = 0; i < ares.size(); ++i) aref[i] = keys[ares[i]];
            for (size_t i = 0; i < aq.size(); ++i) aread[i] = keys[aq[i]];
            const long before = reads;
            if (!check(aref, aread, pa, atable)) return 1;
            reads = before;  // (not part of the vacuity count)
            ++structured;
        }
    }
    printf("reads %ld, certified %ld, mean late-start row %.1f; structured cases %ld\n", reads, certified, certified ? (double)late_rows / certified : 0.0, structured);
    if (certified * 5 < reads) {
        printf("the certificate is vacuous: fewer than a fifth of the reads get one\n");
        return 1;
    }
    printf("seed_warmup OK\n");
    return 0;
}
