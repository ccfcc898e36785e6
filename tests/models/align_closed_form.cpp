// Host-side twin of align_kernel_x's row update (zoe_amd/csrc/zsw_align.hip) — TEST INFRASTRUCTURE.
//
// The kernel does not run Zoe's lazy-F loop (striped.rs:528-553) step by step. It evaluates it in closed form:
//   * the loop's break position T = (round kb, vector vb) from per-round bitmasks over v (one compare per cell and round),
//   * the final H and the flag rewrite of every cell from M(v, lane) = the largest F that visited the cell.
// This program restates that arithmetic with plain loops (true scores in i32, lanes as array indices) and compares the
// complete striped backtrack matrix, row by row, with the oracle's literal restatement of sw_simd_align
// (oracle/zoe_oracle.hpp, `flags_out`) on random inputs: all lane counts, random scoring incl. gap_open = 0 / gap_extend = 0,
// low-complexity and indel-rich reads.  Exit code 0 = identical everywhere.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../oracle/zoe_oracle.hpp"

using namespace zor;

namespace {

constexpr uint8_t UP = 1, UPX = 2, LEFT = 4, LEFTX = 8, STOP = 16;

inline int subsat(int a, int b) { return a > b ? a - b : 0; }

// One call = all rows. Returns flags [R][nv][N] like the oracle.
void model_flags(const uint8_t* ref, size_t R, const uint8_t* read, size_t L, const ProfileWeights& pw, const ByteIndexMap& map,
                 int go, int ge, int N, std::vector<uint8_t>& out) {
    const int nv = (int)((L + (size_t)N - 1) / (size_t)N);
    std::vector<int> H(nv * N, 0), E(nv * N, 0), Hm(nv * N), Y(nv * N), Fend(N), Pa(N), Pb(N), Fk(N);
    std::vector<uint8_t> fl(nv * N);
    std::vector<int> kq(nv * N);
    for (int v = 0; v < nv; ++v)
        for (int l = 0; l < N; ++l) {
            size_t q = (size_t)v + (size_t)l * nv;
            kq[v * N + l] = q < L ? (int)map.to_index(read[q]) : -1;
        }
    out.assign(R * nv * N, 0);
    for (size_t r = 0; r < R; ++r) {
        const int ri = (int)map.to_index(ref[r]);
        // main pass (striped.rs:481-526), lane by lane
        for (int l = 0; l < N; ++l) {
            int F = 0;
            int Hd = l == 0 ? 0 : H[(nv - 1) * N + l - 1];
            // NOTE: H[] still holds the previous row; Hd for v > 0 is the previous row's H[v-1][l]
            int prev = Hd;
            for (int v = 0; v < nv; ++v) {
                const int Eo = E[v * N + l];
                const int hold = H[v * N + l];
                const int s = kq[v * N + l] < 0 ? 0 : pw.w[ri][kq[v * N + l]] - pw.bias;
                int h = prev + s;
                if (h < 0) h = 0;
                if (Eo > h) h = Eo;
                if (F > h) h = F;
                const int hg = subsat(h, go);
                const int En = std::max(subsat(Eo, ge), hg);
                const int Fn = std::max(subsat(F, ge), hg);
                uint8_t f = (Eo == h ? UP : 0) | (F == h ? LEFT : 0) | (En > hg ? UPX : 0) | (Fn > hg ? LEFTX : 0);
                if (h == 0) f = STOP;
                fl[v * N + l] = f;
                Hm[v * N + l] = h;
                E[v * N + l] = En;
                F = Fn;
                prev = hold;
            }
            Fend[l] = F;
        }
        // wait: lanes read H of lane l-1's last vector from the PREVIOUS row, so H must not be overwritten above (it is not: Hm)
        // ---- lazy-F in closed form ----
        for (int i = 0; i < nv * N; ++i) Y[i] = subsat(Hm[i], go) + (i / N) * ge;
        for (int l = 0; l < N; ++l) Pb[l] = 0;  // P_{k-1}
        int vb = nv;                             // vectors visited in the last (possibly partial) round
        bool broke = false;
        for (int k = 0; k < N && !broke; ++k) {
            uint64_t U = 0;
            for (int l = 0; l < N; ++l) {
                Fk[l] = (l - 1 - k >= 0) ? subsat(Fend[l - 1 - k], k * nv * ge) : 0;
                const bool a = Fk[l] + go > Pb[l];
                if (a)
                    for (int v = 0; v < nv; ++v)
                        if (Y[v * N + l] < Fk[l]) U |= 1ull << v;
            }
            int first = 0;
            while (first < nv && ((U >> first) & 1)) ++first;
            for (int l = 0; l < N; ++l) Pa[l] = std::max(Pb[l], Fk[l]);
            if (first < nv) {
                vb = first;
                broke = true;
            } else {
                for (int l = 0; l < N; ++l) Pb[l] = Pa[l];
            }
        }
        if (!broke) {
            vb = nv;  // all N rounds ran to the end: every cell has seen every round
            for (int l = 0; l < N; ++l) Pa[l] = Pb[l];
        }
        for (int v = 0; v < nv; ++v)
            for (int l = 0; l < N; ++l) {
                const int M = subsat(v < vb ? Pa[l] : Pb[l], v * ge);
                const int hm = Hm[v * N + l];
                const int h = std::max(hm, M);
                uint8_t f = fl[v * N + l];
                if (M >= hm) f = (uint8_t)((f & UPX) | LEFT);
                if (subsat(M, ge) > subsat(h, go)) f |= LEFTX;
                if (h == 0) f = STOP;
                H[v * N + l] = h;
                out[(r * nv + v) * N + l] = f;
            }
    }
}

template <int N>
bool check_one(std::mt19937_64& rng, int iter) {
    std::uniform_int_distribution<int> pick(0, 1 << 20);
    const ByteIndexMap map = ByteIndexMap::dna_profile_map();
    const int schemes[][5] = {{2, -5, -10, -1, 1}, {4, -2, -3, -1, 1}, {3, -1, -4, -1, 1}, {2, -3, 0, 0, 1},  {1, -1, -2, -2, 0},
                              {5, -4, -1, 0, 1},  {2, -5, -5, -5, 1}, {10, -10, -5, -5, 1}, {1, -3, -6, -2, 0}, {2, -2, -1, -1, 1}};
    const int* sc = schemes[pick(rng) % 10];
    WeightMatrixI8 wm = WeightMatrixI8::make(map, (int8_t)sc[0], (int8_t)sc[1], sc[4] ? 'N' : -1);
    const size_t R = 20 + pick(rng) % 180;
    const size_t L = 1 + pick(rng) % (N * 6 < 120 ? N * 6 : 120);
    std::vector<uint8_t> ref(R), read(L);
    const char* alpha = "ACGTN";
    const int mode = pick(rng) % 4;
    for (auto& c : ref) c = (uint8_t)alpha[mode == 1 ? pick(rng) % 2 : pick(rng) % 4];
    if (mode == 2) {
        for (auto& c : read) c = (uint8_t)alpha[pick(rng) % 5];
    } else {  // a mutated window of the reference
        size_t p = pick(rng) % R;
        for (size_t i = 0; i < L; ++i) {
            int d = pick(rng) % 100;
            if (d < 6) p += 1 + pick(rng) % 3;  // deletion
            if (d >= 6 && d < 12 && p > 0) --p;  // insertion-ish
            uint8_t c = ref[p % R];
            if (d >= 12 && d < 20) c = (uint8_t)alpha[pick(rng) % 5];
            read[i] = c;
            ++p;
        }
    }
    const bool is_signed = pick(rng) % 2;
    ProfileWeights pw = ProfileWeights::from(wm, is_signed);
    std::vector<uint8_t> want, got;
    if (is_signed) {
        auto prof = StripedProfile<int32_t, N>::make(read.data(), L, pw, map, sc[2], sc[3]);
        sw_simd_align<int32_t, N>(ref.data(), R, prof, &want);
    } else {
        auto prof = StripedProfile<uint32_t, N>::make(read.data(), L, pw, map, sc[2], sc[3]);
        sw_simd_align<uint32_t, N>(ref.data(), R, prof, &want);
    }
    model_flags(ref.data(), R, read.data(), L, pw, map, -sc[2], -sc[3], N, got);
    if (want.size() != got.size()) {
        fprintf(stderr, "size mismatch N=%d iter=%d\n", N, iter);
        return false;
    }
    for (size_t i = 0; i < want.size(); ++i)
        if (want[i] != got[i]) {
            const size_t nv = (L + N - 1) / N;
            fprintf(stderr, "N=%d iter=%d scheme=%d/%d/%d/%d R=%zu L=%zu: cell r=%zu v=%zu lane=%zu oracle=%u model=%u\n", N, iter, sc[0],
                    sc[1], sc[2], sc[3], R, L, i / (nv * N), (i / N) % nv, i % N, want[i], got[i]);
            return false;
        }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 20261004ull);
    long cells = 0;
    for (int it = 0; it < iters; ++it) {
        if (!check_one<2>(rng, it) || !check_one<4>(rng, it) || !check_one<8>(rng, it) || !check_one<16>(rng, it) ||
            !check_one<32>(rng, it) || !check_one<64>(rng, it))
            return 1;
        ++cells;
    }
    printf("closed-form lazy-F == oracle on %d x 6 random pairs (N = 2..64)\n", iters);
    return 0;
}
