// chunk_rows.cpp — host model of the row-chunked full pass (zoe_amd/csrc/zsw_score_v2.hpp, ScoreArgsV2::chunk_rows): a read that the
// seeded pass hands back is scored over all its cells, and against a long reference that is one work item walking every row. The
// chunked form scores it as independent items — rows [kB - overlap, (k + 1)B) from a zero state, k = 0, 1, ... — and takes the
// largest (score, then earliest row, then earliest column) over the items. Claim: with overlap >= L + L * maxw / gap_extend + 2
// (a path that spans more rows than that cannot be positive: zsw_align_dev.hpp, warmup_rows) the result is sw_simd_score_ends'
// (striped.rs:153-336): the maximum of the whole matrix, its first row, the first column of that row.
// Values a chunk computes in its overlap rows are lower bounds of the true ones (paths that start above the chunk are missing),
// never higher; values in the rows it owns are exact. usage: chunk_rows <iterations> <seed>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace {

struct Best {
    int score = 0, row = -1, col = -1;
    bool better_than(const Best& o) const { return score > o.score || (score == o.score && score > 0 && (row < o.row || (row == o.row && col < o.col))); }
};

// Gotoh over rows [lo, hi) from a zero state; only rows >= count_from enter the result (-1: all)
Best gotoh(const std::vector<uint8_t>& ref, const std::vector<uint8_t>& q, int match, int mismatch, int go, int ge, int lo, int hi) {
    const int L = (int)q.size();
    std::vector<int> H(L + 1, 0), E(L + 1, 0);
    Best b;
    for (int r = lo; r < hi; ++r) {
        int diag = 0, f = 0;
        for (int c = 1; c <= L; ++c) {
            const int e = std::max(std::max(E[c] - ge, H[c] - go), 0);  // H[c] still holds row r - 1
            f = std::max(std::max(f - ge, H[c - 1] - go), 0);          // H[c - 1] already holds row r
            const int w = (ref[r] == 4 || q[c - 1] == 4) ? 0 : (ref[r] == q[c - 1] ? match : mismatch);
            const int h = std::max(std::max(diag + w, e), std::max(f, 0));
            diag = H[c];
            H[c] = h;
            E[c] = e;
            if (h > b.score) {
                b.score = h;
                b.row = r;
                b.col = c - 1;
            }
        }
    }
    return b;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); };
    struct Sch {
        int match, mismatch, go, ge;
    };
    const Sch schemes[] = {{2, -5, 10, 1}, {1, -1, 2, 1}, {3, -2, 5, 2}, {1, -3, 5, 2}, {5, -4, 8, 3}, {2, -2, 3, 3}, {2, -5, 1, 1}};  // (validate_profile_args: gap_open is at least gap_extend in magnitude)
    long reads = 0, multi = 0;
    for (int it = 0; it < iters; ++it) {
        const Sch& s = schemes[it % (sizeof(schemes) / sizeof(schemes[0]))];
        const int R = rnd(200, 1500);
        std::vector<uint8_t> ref(R);
        for (auto& x : ref) x = (uint8_t)rnd(0, 3);
        if (rnd(0, 1)) {  // repeats: the same score in several places, the earliest must win
            const int len = rnd(10, 40), from = rnd(0, R - len), to = rnd(0, R - len);
            for (int i = 0; i < len; ++i) ref[to + i] = ref[from + i];
        }
        for (int k = 0; k < 6; ++k) {
            const int L = rnd(4, 40);
            std::vector<uint8_t> q(L);
            int p = rnd(0, R - 1);
            for (int i = 0; i < L; ++i) {
                const int e = rnd(0, 99);
                if (e < 4) p += rnd(1, 30);  // a long deletion: the path spans many rows
                q[i] = (e < 10 || p >= R) ? (uint8_t)rnd(0, 3) : ref[p];
                if (e >= 4) ++p;
                if (rnd(0, 60) == 0) q[i] = 4;
            }
            const int overlap = L + (L * s.match) / s.ge + 2;
            const int B = rnd(1, 3) * overlap + rnd(0, 50);
            const Best whole = gotoh(ref, q, s.match, s.mismatch, s.go, s.ge, 0, R);
            Best got;
            int n_chunks = 0;
            for (int lo = 0; lo < R; lo += B, ++n_chunks) {
                const Best c = gotoh(ref, q, s.match, s.mismatch, s.go, s.ge, std::max(0, lo - overlap), std::min(R, lo + B));
                if (c.better_than(got)) got = c;
            }
            ++reads;
            if (n_chunks > 1) ++multi;
            if (got.score != whole.score || (whole.score > 0 && (got.row != whole.row || got.col != whole.col))) {
                printf("chunked (%d,%d,%d) vs whole (%d,%d,%d): L %d R %d B %d overlap %d scheme %d/%d/%d/%d\n", got.score, got.row, got.col, whole.score, whole.row,
                       whole.col, L, R, B, overlap, s.match, s.mismatch, s.go, s.ge);
                return 1;
            }
        }
    }
    printf("reads %ld, of which in more than one chunk %ld\nchunk_rows OK\n", reads, multi);
    return multi * 2 > reads ? 0 : 1;
}
