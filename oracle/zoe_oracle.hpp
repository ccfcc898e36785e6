// zoe_oracle.hpp — TEST INFRASTRUCTURE ONLY (parity oracle).
//
// A CPU restatement of the striped Smith-Waterman hot path of CDCgov/zoe
// (v0.0.32-dev). Nothing in the product path (zoe_amd/, include/) may include,
// link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
//
// Parity status: the reference is nightly Rust and cannot be compiled or run
// in the build container (no rustc/cargo, no network), so this restatement is
// pinned by the reference's own known-answer tests and doctests
// (tests/test_oracle_golden.py lists every vector with its file:line).
//
// Every function cites the reference file:line it follows (paths relative to
// the reference checkout). Lane vectors (`Simd<T, N>`) are plain arrays.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace zor {

// ---- src/math/integer.rs:17-238 (AnyInt / AlignableIntWidth semantics) ----
template <typename T> struct Int {
    static constexpr T MIN = std::numeric_limits<T>::min();
    static constexpr T MAX = std::numeric_limits<T>::max();
    static constexpr bool SIGNED = std::is_signed<T>::value;
};

template <typename T> inline T sat_add(T a, T b) {
    // src/data/extension/simd.rs:9-87 (saturating_add lane-wise)
    using W = int64_t;
    W r = W(a) + W(b);
    if (r > W(Int<T>::MAX)) r = W(Int<T>::MAX);
    if (r < W(Int<T>::MIN)) r = W(Int<T>::MIN);
    return T(r);
}
template <typename T> inline T sat_sub(T a, T b) {
    using W = int64_t;
    W r = W(a) - W(b);
    if (r > W(Int<T>::MAX)) r = W(Int<T>::MAX);
    if (r < W(Int<T>::MIN)) r = W(Int<T>::MIN);
    return T(r);
}

// `Simd<T, N>` as a plain array.
template <typename T, int N> struct Vec {
    T v[N];
    static Vec splat(T x) {
        Vec r;
        for (int i = 0; i < N; ++i) r.v[i] = x;
        return r;
    }
    // std::simd shift_elements_right::<1>(pad): lane i -> lane i+1, lane 0 = pad
    Vec shr1(T pad) const {
        Vec r;
        r.v[0] = pad;
        for (int i = 1; i < N; ++i) r.v[i] = v[i - 1];
        return r;
    }
    Vec sadd(const Vec& o) const {
        Vec r;
        for (int i = 0; i < N; ++i) r.v[i] = sat_add<T>(v[i], o.v[i]);
        return r;
    }
    Vec ssub(const Vec& o) const {
        Vec r;
        for (int i = 0; i < N; ++i) r.v[i] = sat_sub<T>(v[i], o.v[i]);
        return r;
    }
    Vec max(const Vec& o) const {
        Vec r;
        for (int i = 0; i < N; ++i) r.v[i] = v[i] > o.v[i] ? v[i] : o.v[i];
        return r;
    }
    T reduce_max() const {
        T m = v[0];
        for (int i = 1; i < N; ++i) m = v[i] > m ? v[i] : m;
        return m;
    }
    bool any_gt(const Vec& o) const {
        for (int i = 0; i < N; ++i)
            if (v[i] > o.v[i]) return true;
        return false;
    }
};

// ---- tri-state result: src/alignment/types/output.rs:18-25 ----
enum Status : uint32_t { SOME = 0, OVERFLOWED = 1, UNMAPPED = 2 };

// ---- src/alignment/errors.rs:6-15 / src/alignment/profile.rs:32-44 ----
enum ProfileError : int {
    PROFILE_OK = 0,
    EMPTY_SEQUENCE = 1,
    GAP_OPEN_OUT_OF_RANGE = 2,
    GAP_EXTEND_OUT_OF_RANGE = 3,
    BAD_GAP_WEIGHTS = 4,
};
inline int validate_profile_args(size_t seq_len, int gap_open, int gap_extend) {
    if (seq_len == 0) return EMPTY_SEQUENCE;
    if (gap_open < -127 || gap_open > 0) return GAP_OPEN_OUT_OF_RANGE;
    if (gap_extend < -127 || gap_extend > 0) return GAP_EXTEND_OUT_OF_RANGE;
    if (gap_extend < gap_open) return BAD_GAP_WEIGHTS;
    return PROFILE_OK;
}

// ---- src/data/constants/mappings/byte_index.rs:250-287,331-333 ----
struct ByteIndexMap {
    uint8_t index_map[256];
    int S;
    // ByteIndexMap::new(byte_keys, catch_all)
    static ByteIndexMap make(const uint8_t* keys, int S, uint8_t catch_all, bool ignore_case) {
        ByteIndexMap m;
        m.S = S;
        auto up = [](uint8_t b) -> uint8_t { return (b >= 'a' && b <= 'z') ? uint8_t(b - 32) : b; };
        auto lo = [](uint8_t b) -> uint8_t { return (b >= 'A' && b <= 'Z') ? uint8_t(b + 32) : b; };
        int ca = -1;
        for (int i = 0; i < S; ++i) {
            uint8_t k = ignore_case ? up(keys[i]) : keys[i];
            uint8_t c = ignore_case ? up(catch_all) : catch_all;
            if (k == c) ca = i;
        }
        for (int b = 0; b < 256; ++b) m.index_map[b] = uint8_t(ca < 0 ? 0 : ca);
        for (int i = 0; i < S; ++i) {
            if (ignore_case) {
                m.index_map[up(keys[i])] = uint8_t(i);
                m.index_map[lo(keys[i])] = uint8_t(i);
            } else {
                m.index_map[keys[i]] = uint8_t(i);
            }
        }
        return m;
    }
    void add_synonym_ignore_case(uint8_t new_key, uint8_t prev) {
        uint8_t idx = index_map[prev];
        auto up = [](uint8_t b) -> uint8_t { return (b >= 'a' && b <= 'z') ? uint8_t(b - 32) : b; };
        auto lo = [](uint8_t b) -> uint8_t { return (b >= 'A' && b <= 'Z') ? uint8_t(b + 32) : b; };
        index_map[up(new_key)] = idx;
        index_map[lo(new_key)] = idx;
    }
    // src/data/constants/mappings/dna.rs:177-178 (DNA_PROFILE_MAP)
    static ByteIndexMap dna_profile_map() {
        const uint8_t keys[5] = {'A', 'C', 'G', 'T', 'N'};
        ByteIndexMap m = make(keys, 5, 'N', true);
        m.add_synonym_ignore_case('U', 'T');
        return m;
    }
    inline size_t to_index(uint8_t b) const { return index_map[b]; }
};

constexpr int MAX_S = 32;

// ---- src/data/matrices/mod.rs:230-235 (signed matrix) ----
struct WeightMatrixI8 {
    int8_t w[MAX_S][MAX_S];
    int S;
    // WeightMatrix::new (matrices/mod.rs:358-399)
    static WeightMatrixI8 make(const ByteIndexMap& map, int8_t matching, int8_t mismatch, int ignoring /* byte or -1 */) {
        WeightMatrixI8 m;
        m.S = map.S;
        int skip = ignoring >= 0 ? int(map.to_index(uint8_t(ignoring))) : -1;
        for (int i = 0; i < m.S; ++i)
            for (int j = 0; j < m.S; ++j) {
                if (skip >= 0 && (skip == i || skip == j)) {
                    m.w[i][j] = 0;
                    continue;
                }
                m.w[i][j] = (i == j) ? matching : mismatch;
            }
        return m;
    }
    // get_bias (matrices/mod.rs:452-466)
    int8_t get_bias() const {
        int8_t mn = 0;
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j)
                if (w[i][j] < mn) mn = w[i][j];
        return mn;
    }
};

// Weights as seen by a StripedProfile<T,..>: signed T takes the signed matrix
// (bias 0); unsigned T takes to_biased_matrix() (matrices/mod.rs:471-491).
struct ProfileWeights {
    int w[MAX_S][MAX_S];
    int bias;
    int S;
    static ProfileWeights from(const WeightMatrixI8& m, bool is_signed) {
        ProfileWeights p;
        p.S = m.S;
        if (is_signed) {
            p.bias = 0;
            for (int i = 0; i < m.S; ++i)
                for (int j = 0; j < m.S; ++j) p.w[i][j] = m.w[i][j];
        } else {
            int8_t b = m.get_bias();
            for (int i = 0; i < m.S; ++i)
                for (int j = 0; j < m.S; ++j) p.w[i][j] = int(uint8_t(int16_t(m.w[i][j]) - int16_t(b)));
            p.bias = b < 0 ? -int(b) : int(b);  // unsigned_abs
        }
        return p;
    }
};

// ---- src/alignment/profile.rs:198-207,270-306 ----
template <typename T, int N> struct StripedProfile {
    std::vector<Vec<T, N>> profile;  // [ref_index * nv + v]
    T gap_open, gap_extend, bias;    // gaps stored NEGATED (positive magnitudes)
    const ByteIndexMap* mapping;
    size_t seq_len;
    int S;
    size_t number_vectors() const { return profile.size() / size_t(S); }  // profile.rs:399-401

    // new_unchecked (profile.rs:270-306)
    static StripedProfile make(const uint8_t* seq, size_t len, const ProfileWeights& pw, const ByteIndexMap& map,
                               int gap_open, int gap_extend) {
        StripedProfile p;
        p.S = pw.S;
        p.mapping = &map;
        p.seq_len = len;
        size_t nv = (len + size_t(N) - 1) / size_t(N);
        size_t total_lanes = size_t(N) * nv;
        p.bias = T(pw.bias);
        Vec<T, N> biases = Vec<T, N>::splat(p.bias);
        p.profile.assign(size_t(p.S) * nv, biases);
        for (size_t v = 0; v < nv; ++v)
            for (int ref_index = 0; ref_index < p.S; ++ref_index) {
                Vec<T, N> vec = biases;
                size_t i = 0;
                for (size_t q = v; q < total_lanes; q += nv, ++i)
                    if (q < len) vec.v[i] = T(pw.w[ref_index][map.to_index(seq[q])]);
                p.profile[size_t(ref_index) * nv + v] = vec;
            }
        p.gap_open = T(-gap_open);      // from_literal(-gap_open), integer.rs:66-74
        p.gap_extend = T(-gap_extend);
        return p;
    }

    // reverse_from_forward (profile.rs:314-350); returns false for `None`
    bool reverse_from_forward(size_t seq_end, StripedProfile& out) const {
        if (seq_end == 0 || seq_end > seq_len) return false;
        size_t nv = (seq_end + size_t(N) - 1) / size_t(N);
        size_t nv_old = number_vectors();
        size_t total_lanes = size_t(N) * nv;
        Vec<T, N> biases = Vec<T, N>::splat(bias);
        out.profile.assign(size_t(S) * nv, biases);
        for (size_t v = 0; v < nv; ++v)
            for (int ref_index = 0; ref_index < S; ++ref_index) {
                Vec<T, N> vec = biases;
                size_t i = 0;
                for (size_t q = v; q < total_lanes; q += nv, ++i)
                    if (q < seq_end) {
                        size_t q_old = seq_end - 1 - q;
                        size_t v_old = q_old % nv_old;
                        size_t lane_old = q_old / nv_old;
                        vec.v[i] = profile[size_t(ref_index) * nv_old + v_old].v[lane_old];
                    }
                out.profile[size_t(ref_index) * nv + v] = vec;
            }
        out.gap_open = gap_open;
        out.gap_extend = gap_extend;
        out.bias = bias;
        out.mapping = mapping;
        out.seq_len = seq_end;
        out.S = S;
        return true;
    }
};

// ---- src/data/types/cigar/mod.rs:303-308 ; src/alignment/types/state.rs ----
struct Ciglet {
    size_t inc;
    uint8_t op;
    bool operator==(const Ciglet& o) const { return inc == o.inc && op == o.op; }
};
struct AlignmentStates {
    std::vector<Ciglet> c;
    // add_ciglet (state.rs:142-152)
    void add_ciglet(Ciglet g) {
        if (g.inc > 0) {
            if (!c.empty() && c.back().op == g.op)
                c.back().inc += g.inc;
            else
                c.push_back(g);
        }
    }
    void add_state(uint8_t op) { add_ciglet({1, op}); }          // state.rs:130-132
    void soft_clip(size_t inc) { add_ciglet({inc, 'S'}); }       // state.rs:232-234
    void make_reverse() { std::reverse(c.begin(), c.end()); }    // state.rs:281-283
    // Display (types/std_traits.rs:108-117): decimal inc + op char
    std::string to_string() const {
        std::string s;
        for (auto& g : c) {
            s += std::to_string(g.inc);
            s.push_back(char(g.op));
        }
        return s;
    }
    bool operator==(const AlignmentStates& o) const { return c == o.c; }
};

// ---- src/alignment/types/output.rs:264-279 ----
struct Alignment {
    uint32_t score = 0;
    size_t ref_start = 0, ref_end = 0;      // ref_range
    size_t query_start = 0, query_end = 0;  // query_range
    AlignmentStates states;
    size_t ref_len = 0, query_len = 0;
    bool operator==(const Alignment& o) const {
        return score == o.score && ref_start == o.ref_start && ref_end == o.ref_end && query_start == o.query_start &&
               query_end == o.query_end && states == o.states && ref_len == o.ref_len && query_len == o.query_len;
    }
    // invert (output.rs:396-425)
    Alignment invert() const {
        Alignment r;
        r.states.soft_clip(ref_start);
        bool first = true;
        for (auto g : states.c) {
            if (g.op == 'S' || g.op == 'H') continue;
            if (g.op == 'D')
                g.op = 'I';
            else if (g.op == 'I')
                g.op = 'D';
            // extend_from_ciglets: first goes through add_ciglet, the rest are pushed
            if (first) {
                r.states.add_ciglet(g);
                first = false;
            } else {
                r.states.c.push_back(g);
            }
        }
        r.states.soft_clip(ref_len - ref_end);
        r.score = score;
        r.ref_start = query_start;
        r.ref_end = query_end;
        r.query_start = ref_start;
        r.query_end = ref_end;
        r.ref_len = query_len;
        r.query_len = ref_len;
        return r;
    }
};

template <typename R> struct Maybe {
    Status status = UNMAPPED;
    R value{};
};

struct ScoreIndices {
    uint32_t score;
    size_t ref_idx, query_idx;
};
struct ScoreAndRanges {
    uint32_t score;
    size_t ref_start, ref_end, query_start, query_end;
};

// ---- flag bits: src/alignment/types/backtrack.rs:18-34 ----
constexpr uint8_t BT_UP = 1, BT_UP_EXTENDING = 2, BT_LEFT = 4, BT_LEFT_EXTENDING = 8, BT_STOP = 16;

// BackTrackable::to_alignment (backtrack.rs:290-342) over any cell accessor.
template <typename CellFn>
Alignment to_alignment(CellFn cell, uint32_t score, size_t r_end, size_t c_end, size_t ref_len, size_t query_len) {
    Alignment a;
    uint8_t op = 0;
    uint8_t f = cell(r_end, c_end);
    r_end += 1;
    c_end += 1;
    size_t r = r_end, c = c_end;
    a.states.soft_clip(query_len - c);
    while (!(f & BT_STOP) && r > 0 && c > 0) {
        if (op == 'D' && (f & BT_UP_EXTENDING)) {
            op = 'D';
            r -= 1;
        } else if (op == 'I' && (f & BT_LEFT_EXTENDING)) {
            op = 'I';
            c -= 1;
        } else if (f & BT_UP) {
            op = 'D';
            r -= 1;
        } else if (f & BT_LEFT) {
            op = 'I';
            c -= 1;
        } else {
            op = 'M';
            r -= 1;
            c -= 1;
        }
        a.states.add_state(op);
        f = cell(r > 0 ? r - 1 : 0, c > 0 ? c - 1 : 0);
    }
    a.states.soft_clip(c);
    a.states.make_reverse();
    a.score = score;
    a.ref_start = r;
    a.ref_end = r_end;
    a.query_start = c;
    a.query_end = c_end;
    a.ref_len = ref_len;
    a.query_len = query_len;
    return a;
}

// score_to_maybe_aligned (striped.rs:610-633). Returns status; *score set when SOME.
template <typename T> inline Status score_to_maybe_aligned(T best, T bias, uint32_t* score) {
    if (Int<T>::SIGNED) {
        if (!(best < Int<T>::MAX)) return OVERFLOWED;
        uint32_t s = uint32_t(uint32_t(Int<T>::MAX) + 1u) + uint32_t(int32_t(best));  // wrapping_add_signed
        if (s == 0) return UNMAPPED;
        *score = s;
        return SOME;
    } else {
        // best.checked_add(bias + 1)
        uint64_t sum = uint64_t(best) + uint64_t(bias) + 1u;
        if (sum > uint64_t(Int<T>::MAX)) return OVERFLOWED;
        uint32_t s = uint32_t(best);
        if (s == 0) return UNMAPPED;
        *score = s;
        return SOME;
    }
}

// ---- sw_simd_score (striped.rs:65-142) ----
template <typename T, int N>
Status sw_simd_score(const uint8_t* reference, size_t ref_len, const StripedProfile<T, N>& query, uint32_t* score) {
    using V = Vec<T, N>;
    const size_t num_vecs = query.number_vectors();
    const T min = Int<T>::MIN;
    const V gap_opens = V::splat(query.gap_open), gap_extends = V::splat(query.gap_extend);
    const V minimums = V::splat(min), biases = V::splat(query.bias);
    std::vector<V> load(num_vecs, minimums), store(num_vecs, minimums), e_scores(num_vecs, minimums);
    V max_scores = minimums;

    for (size_t r = 0; r < ref_len; ++r) {
        size_t ref_index = query.mapping->to_index(reference[r]);
        V F = minimums;
        V H = store[num_vecs - 1].shr1(min);
        std::swap(load, store);
        const V* scores_vec = &query.profile[ref_index * num_vecs];
        for (size_t j = 0; j < num_vecs; ++j) {
            V E = e_scores[j];
            H = H.sadd(scores_vec[j]);
            if (!Int<T>::SIGNED) H = H.ssub(biases);
            H = H.max(E).max(F);
            max_scores = max_scores.max(H);
            store[j] = H;
            H = H.ssub(gap_opens);
            E = E.ssub(gap_extends).max(H);
            F = F.ssub(gap_extends).max(H);
            e_scores[j] = E;
            H = load[j];
        }
        size_t j = 0;
        H = store[j];
        F = F.shr1(min);
        while (F.any_gt(H.ssub(gap_opens))) {
            H = H.max(F);
            store[j] = H;
            F = F.ssub(gap_extends);
            j += 1;
            if (j >= num_vecs) {
                j = 0;
                F = F.shr1(min);
            }
            H = store[j];
        }
    }
    T best = max_scores.reduce_max();
    return score_to_maybe_aligned<T>(best, query.bias, score);
}

// ---- sw_simd_score_ends_dir (striped.rs:213-336) ----
template <typename T, int N, bool FORWARD>
Maybe<ScoreIndices> sw_simd_score_ends_dir(const uint8_t* reference, size_t ref_len, const StripedProfile<T, N>& query) {
    using V = Vec<T, N>;
    Maybe<ScoreIndices> out;
    if (ref_len == 0) {
        out.status = UNMAPPED;
        return out;
    }
    const size_t num_vecs = query.number_vectors();
    const T min = Int<T>::MIN;
    const V gap_opens = V::splat(query.gap_open), gap_extends = V::splat(query.gap_extend);
    const V minimums = V::splat(min), biases = V::splat(query.bias);
    const T saturating_threshold = Int<T>::SIGNED ? Int<T>::MAX : T(Int<T>::MAX - query.bias);
    std::vector<V> load(num_vecs, minimums), store(num_vecs, minimums), e_scores(num_vecs, minimums),
        max_row(num_vecs, minimums);
    T best = min;
    size_t r_end = ref_len - 1;
    const size_t len = ref_len;
    for (size_t r = 0; r < len; ++r) {
        size_t ref_index = query.mapping->to_index(reference[FORWARD ? r : len - 1 - r]);
        V F = minimums;
        V H = store[num_vecs - 1].shr1(min);
        if (r > 1 && r_end == r - 2) std::swap(max_row, load);
        std::swap(load, store);
        const V* scores_vec = &query.profile[ref_index * num_vecs];
        V max_scores = minimums;
        for (size_t v = 0; v < num_vecs; ++v) {
            V E = e_scores[v];
            H = H.sadd(scores_vec[v]);
            if (!Int<T>::SIGNED) H = H.ssub(biases);
            H = H.max(E).max(F);
            max_scores = max_scores.max(H);
            store[v] = H;
            H = H.ssub(gap_opens);
            E = E.ssub(gap_extends).max(H);
            F = F.ssub(gap_extends).max(H);
            e_scores[v] = E;
            H = load[v];
        }
        for (int it = 0; it < N; ++it) {  // 'lazy_f
            F = F.shr1(min);
            bool brk = false;
            for (size_t v = 0; v < num_vecs; ++v) {
                H = store[v];
                if (!F.any_gt(H.ssub(gap_opens))) {
                    brk = true;
                    break;
                }
                H = H.max(F);
                store[v] = H;
                F = F.ssub(gap_extends);
            }
            if (brk) break;
        }
        T row_best = max_scores.reduce_max();
        if (row_best > best) {
            if (row_best >= saturating_threshold) {
                out.status = OVERFLOWED;
                return out;
            }
            best = row_best;
            r_end = r;
        }
    }
    if (r_end == ref_len - 1)
        max_row = store;
    else if (r_end == ref_len - 2)
        max_row = load;

    size_t c_end = query.seq_len - 1;
    for (size_t ci = 0; ci < query.seq_len; ++ci) {
        size_t v = ci % num_vecs, lane = ci / num_vecs;
        if (max_row[v].v[lane] == best) {
            c_end = ci;
            break;
        }
    }
    if (FORWARD) {
        r_end += 1;
        c_end += 1;
    } else {
        r_end = ref_len - 1 - r_end;
        c_end = query.seq_len - 1 - c_end;
    }
    uint32_t score = 0;
    out.status = score_to_maybe_aligned<T>(best, query.bias, &score);
    out.value = ScoreIndices{score, r_end, c_end};
    return out;
}

// ---- sw_simd_score_ranges (striped.rs:355-388) ----
template <typename T, int N>
Maybe<ScoreAndRanges> sw_simd_score_ranges(const uint8_t* reference, size_t ref_len, const StripedProfile<T, N>& query) {
    Maybe<ScoreAndRanges> out;
    auto fwd = sw_simd_score_ends_dir<T, N, true>(reference, ref_len, query);
    if (fwd.status != SOME) {
        out.status = fwd.status;
        return out;
    }
    StripedProfile<T, N> rev;
    if (!query.reverse_from_forward(fwd.value.query_idx, rev)) {
        out.status = UNMAPPED;
        return out;
    }
    auto bwd = sw_simd_score_ends_dir<T, N, false>(reference, fwd.value.ref_idx, rev);
    if (bwd.status != SOME) {
        out.status = bwd.status;
        return out;
    }
    out.status = SOME;
    out.value = ScoreAndRanges{fwd.value.score, bwd.value.ref_idx, fwd.value.ref_idx, bwd.value.query_idx,
                               fwd.value.query_idx};
    return out;
}

// ---- sw_simd_align (striped.rs:449-598) ----
// `flags_out` (optional) receives the raw striped backtrack matrix
// (R * num_vecs * N bytes, backtrack.rs:98-130) for kernel-level parity tests.
template <typename T, int N>
Maybe<Alignment> sw_simd_align(const uint8_t* reference, size_t ref_len, const StripedProfile<T, N>& query,
                               std::vector<uint8_t>* flags_out = nullptr) {
    using V = Vec<T, N>;
    using F8 = Vec<uint8_t, N>;
    Maybe<Alignment> out;
    if (ref_len == 0) {
        out.status = UNMAPPED;
        return out;
    }
    const size_t num_vecs = query.number_vectors();
    const T min = Int<T>::MIN;
    const V gap_opens = V::splat(query.gap_open), gap_extends = V::splat(query.gap_extend);
    const V minimums = V::splat(min), biases = V::splat(query.bias);
    const T saturating_threshold = Int<T>::SIGNED ? Int<T>::MAX : T(Int<T>::MAX - query.bias);
    std::vector<V> load(num_vecs, minimums), store(num_vecs, minimums), e_scores(num_vecs, minimums),
        max_row(num_vecs, minimums);
    T best = min;
    size_t r_end = ref_len - 1;
    std::vector<F8> backtrack(ref_len * num_vecs);

    for (size_t r = 0; r < ref_len; ++r) {
        size_t ref_index = query.mapping->to_index(reference[r]);
        V F = minimums;
        V H = store[num_vecs - 1].shr1(min);
        if (r > 1 && r_end == r - 2) std::swap(max_row, load);
        std::swap(load, store);
        const V* scores_vec = &query.profile[ref_index * num_vecs];
        F8* backtrack_row = &backtrack[r * num_vecs];
        V max_scores = minimums;
        for (size_t v = 0; v < num_vecs; ++v) {
            V E = e_scores[v];
            H = H.sadd(scores_vec[v]);
            if (!Int<T>::SIGNED) H = H.ssub(biases);
            H = H.max(E).max(F);
            F8 flags = F8::splat(0);
            max_scores = max_scores.max(H);
            bool stopped[N];
            for (int i = 0; i < N; ++i) {
                if (E.v[i] == H.v[i]) flags.v[i] |= BT_UP;
                if (F.v[i] == H.v[i]) flags.v[i] |= BT_LEFT;
                stopped[i] = H.v[i] == min;
            }
            store[v] = H;
            H = H.ssub(gap_opens);
            E = E.ssub(gap_extends).max(H);
            F = F.ssub(gap_extends).max(H);
            for (int i = 0; i < N; ++i) {
                if (E.v[i] > H.v[i]) flags.v[i] |= BT_UP_EXTENDING;
                if (F.v[i] > H.v[i]) flags.v[i] |= BT_LEFT_EXTENDING;
                if (stopped[i]) flags.v[i] = BT_STOP;
            }
            backtrack_row[v] = flags;
            e_scores[v] = E;
            H = load[v];
        }
        for (int it = 0; it < N; ++it) {  // 'lazy_f
            F = F.shr1(min);
            bool brk = false;
            for (size_t v = 0; v < num_vecs; ++v) {
                H = store[v];
                if (!F.any_gt(H.ssub(gap_opens))) {
                    brk = true;
                    break;
                }
                H = H.max(F);
                store[v] = H;
                F8 flags = backtrack_row[v];
                bool stopped[N];
                for (int i = 0; i < N; ++i) {
                    stopped[i] = H.v[i] == min;
                    // simd_correct_and_set_left (backtrack.rs:217-220)
                    if (F.v[i] == H.v[i]) flags.v[i] = uint8_t((flags.v[i] & BT_UP_EXTENDING) | BT_LEFT);
                }
                H = H.ssub(gap_opens);
                F = F.ssub(gap_extends);
                for (int i = 0; i < N; ++i) {
                    if (F.v[i] > H.v[i]) flags.v[i] |= BT_LEFT_EXTENDING;
                    if (stopped[i]) flags.v[i] = BT_STOP;
                }
                backtrack_row[v] = flags;
            }
            if (brk) break;
        }
        T row_best = max_scores.reduce_max();
        if (row_best > best) {
            if (row_best >= saturating_threshold) {
                out.status = OVERFLOWED;
                return out;
            }
            best = row_best;
            r_end = r;
        }
    }
    if (r_end == ref_len - 1)
        max_row = store;
    else if (r_end == ref_len - 2)
        max_row = load;

    size_t c_end = query.seq_len - 1;
    for (size_t ci = 0; ci < query.seq_len; ++ci) {
        size_t v = ci % num_vecs, lane = ci / num_vecs;
        if (max_row[v].v[lane] == best) {
            c_end = ci;
            break;
        }
    }
    if (flags_out) {
        flags_out->resize(ref_len * num_vecs * size_t(N));
        for (size_t i = 0; i < backtrack.size(); ++i)
            for (int l = 0; l < N; ++l) (*flags_out)[i * size_t(N) + size_t(l)] = backtrack[i].v[l];
    }
    uint32_t score = 0;
    out.status = score_to_maybe_aligned<T>(best, query.bias, &score);
    if (out.status == SOME) {
        // BacktrackMatrixStriped::move_to (backtrack.rs:473-477)
        auto cell = [&](size_t r, size_t c) -> uint8_t {
            size_t v = c % num_vecs, lane = (c - v) / num_vecs;
            return backtrack[num_vecs * r + v].v[lane];
        };
        out.value = to_alignment(cell, score, r_end, c_end, ref_len, query.seq_len);
    }
    return out;
}

// ---- ScalarProfile + sw_scalar_score / sw_scalar_align (profile.rs:56-116, scalar.rs:55-122,173-271) ----
struct ScalarProfile {
    const uint8_t* seq;
    size_t len;
    const WeightMatrixI8* matrix;
    const ByteIndexMap* mapping;
    int32_t gap_open, gap_extend;  // kept negative
    int32_t weight(uint8_t ref_residue, uint8_t query_residue) const {
        return matrix->w[mapping->to_index(ref_residue)][mapping->to_index(query_residue)];
    }
};

inline Status sw_scalar_score(const uint8_t* reference, size_t ref_len, const ScalarProfile& q, uint32_t* score) {
    int32_t best_score = 0;
    std::vector<int32_t> h_row(q.len, 0), e_row(q.len, q.gap_open);
    for (size_t r = 0; r < ref_len; ++r) {
        int32_t f = q.gap_open, h = 0;
        for (size_t c = 0; c < q.len; ++c) {
            h += q.weight(reference[r], q.seq[c]);
            int32_t e = e_row[c];
            h = std::max(std::max(std::max(h, e), f), 0);
            best_score = std::max(best_score, h);
            e = std::max(e + q.gap_extend, h + q.gap_open);
            f = std::max(f + q.gap_extend, h + q.gap_open);
            std::swap(h, h_row[c]);
            e_row[c] = e;
        }
    }
    if (best_score > 0) {
        *score = uint32_t(best_score);
        return SOME;
    }
    return UNMAPPED;
}

inline Maybe<Alignment> sw_scalar_align(const uint8_t* reference, size_t ref_len, const ScalarProfile& q) {
    Maybe<Alignment> out;
    if (ref_len == 0) {
        out.status = UNMAPPED;
        return out;
    }
    int32_t best_score = 0;
    size_t r_end = 0, c_end = 0;
    std::vector<int32_t> h_row(q.len, 0), e_row(q.len, q.gap_open);
    std::vector<uint8_t> bt(ref_len * q.len, 0);
    for (size_t r = 0; r < ref_len; ++r) {
        int32_t f = q.gap_open, h = 0;
        for (size_t c = 0; c < q.len; ++c) {
            uint8_t& cell = bt[q.len * r + c];
            h += q.weight(reference[r], q.seq[c]);
            int32_t e = e_row[c];
            h = std::max(std::max(std::max(h, e), f), 0);
            if (h > best_score) {
                best_score = h;
                r_end = r;
                c_end = c;
            }
            if (e == h) cell |= BT_UP;
            if (f == h) cell |= BT_LEFT;
            if (h == 0) cell = BT_STOP;
            int32_t next_diag = h_row[c];
            h_row[c] = h;
            h += q.gap_open;
            e = std::max(e + q.gap_extend, h);
            f = std::max(f + q.gap_extend, h);
            if (h != q.gap_open) {
                if (e > h) cell |= BT_UP_EXTENDING;
                if (f > h) cell |= BT_LEFT_EXTENDING;
            }
            h = next_diag;
            e_row[c] = e;
        }
    }
    if (best_score == 0) {
        out.status = UNMAPPED;
        return out;
    }
    out.status = SOME;
    auto cell = [&](size_t r, size_t c) -> uint8_t { return bt[q.len * r + c]; };
    out.value = to_alignment(cell, uint32_t(best_score), r_end, c_end, ref_len, q.len);
    return out;
}

// ---- AlignmentStates helpers used by the 3-pass path (state.rs:156-166, 201-208, 244-246) ----
inline void prepend_ciglet(AlignmentStates& st, Ciglet g) {
    if (g.inc > 0) {
        if (!st.c.empty() && st.c.front().op == g.op)
            st.c.front().inc += g.inc;
        else
            st.c.insert(st.c.begin(), g);
    }
}
inline AlignmentStates new_no_gaps(size_t start, size_t end, size_t query_len) {
    AlignmentStates st;
    st.soft_clip(start);
    st.add_ciglet({end - start, 'M'});
    st.soft_clip(query_len - end);
    return st;
}

// ---- sw_banded_align (sw/banded.rs:40-133) with BandedBacktrackMatrix (types/backtrack.rs:541-635) ----
// Returns UNMAPPED when the reference would panic on an out-of-band traceback index (never a result there).
inline Maybe<Alignment> sw_banded_align(const uint8_t* reference, size_t ref_len, const ScalarProfile& q, size_t band_width) {
    Maybe<Alignment> out;
    if (ref_len == 0) {
        out.status = UNMAPPED;
        return out;
    }
    int32_t best_score = 0;
    size_t r_end = 0, c_end = 0;
    const size_t q_len = q.len;
    std::vector<int32_t> h_row(q_len, 0), e_row(q_len, q.gap_open);
    const size_t full = 2 * band_width + 1;
    std::vector<uint8_t> bt(ref_len * full, 0);
    auto cursor_of = [&](size_t r, size_t c, bool* ok) -> size_t {  // BandedBacktrackMatrix::move_to (:629-634)
        const size_t skipped = r > band_width ? r - band_width : 0;
        if (c < skipped) {
            *ok = false;  // usize underflow: the reference panics
            return 0;
        }
        const size_t cur = r * full + (c - skipped);
        if (cur >= bt.size()) *ok = false;
        return cur;
    };
    int32_t h_store = 0;
    for (size_t r = 0; r < ref_len; ++r) {
        int32_t f = q.gap_open;
        int32_t h = h_store;
        const size_t start_col = r > band_width ? r - band_width : 0;
        const size_t end_col = std::min(r + band_width + 1, q_len);
        if (start_col >= end_col) break;
        if (start_col + band_width == r) {
            const int32_t match_score = q.weight(reference[r], q.seq[start_col]);
            const int32_t e = e_row[start_col];
            h_store = std::max(std::max(h + match_score, e), 0);
        }
        for (size_t c = start_col; c < end_col; ++c) {
            bool ok = true;
            uint8_t& cell = bt[cursor_of(r, c, &ok)];
            h += q.weight(reference[r], q.seq[c]);
            int32_t e = e_row[c];
            h = std::max(std::max(std::max(h, e), f), 0);
            if (h > best_score) {
                best_score = h;
                r_end = r;
                c_end = c;
            }
            if (e == h) cell |= BT_UP;
            if (f == h) cell |= BT_LEFT;
            if (h == 0) cell = BT_STOP;
            const int32_t next_diag = h_row[c];
            h_row[c] = h;
            h += q.gap_open;
            e = std::max(e + q.gap_extend, h);
            f = std::max(f + q.gap_extend, h);
            if (h != q.gap_open) {
                if (e > h) cell |= BT_UP_EXTENDING;
                if (f > h) cell |= BT_LEFT_EXTENDING;
            }
            h = next_diag;
            e_row[c] = e;
        }
    }
    if (best_score == 0) {
        out.status = UNMAPPED;
        return out;
    }
    bool ok = true;
    auto cell = [&](size_t r, size_t c) -> uint8_t {
        const size_t cur = cursor_of(r, c, &ok);
        return ok ? bt[cur] : BT_STOP;
    };
    out.value = to_alignment(cell, uint32_t(best_score), r_end, c_end, ref_len, q_len);
    out.status = ok ? SOME : UNMAPPED;
    return out;
}

// ---- sw_align_3pass (sw/three_pass.rs:21-104) ----
template <typename T, int N>
Maybe<Alignment> sw_align_3pass(const uint8_t* reference, size_t ref_len, const StripedProfile<T, N>& query_profile,
                                const uint8_t* query, size_t query_len, const WeightMatrixI8& matrix, const ByteIndexMap& map,
                                int gap_open, int gap_extend, int* how = nullptr) {
    Maybe<Alignment> out;
    auto sr = sw_simd_score_ranges<T, N>(reference, ref_len, query_profile);
    if (sr.status != SOME) {
        out.status = sr.status;
        return out;
    }
    const uint32_t score = sr.value.score;
    const size_t rs = sr.value.ref_start, re = sr.value.ref_end, qs = sr.value.query_start, qe = sr.value.query_end;
    if (qs >= qe) {
        out.status = UNMAPPED;
        return out;
    }
    ScalarProfile whole{query, query_len, &matrix, &map, gap_open, gap_extend};
    if (qe - qs == re - rs) {
        int64_t sum = 0;
        for (size_t k = 0; k < qe - qs; ++k) sum += whole.weight(reference[rs + k], query[qs + k]);
        const uint32_t usum = sum < 0 ? 0u : uint32_t(sum);  // try_into().unwrap_or(0)
        if (usum == score) {
            out.status = SOME;
            out.value.score = score;
            out.value.ref_start = rs;
            out.value.ref_end = re;
            out.value.query_start = qs;
            out.value.query_end = qe;
            out.value.states = new_no_gaps(qs, qe, query_len);
            out.value.ref_len = ref_len;
            out.value.query_len = query_len;
            if (how) *how = 0;
            return out;
        }
    }
    ScalarProfile query_new{query + qs, qe - qs, &matrix, &map, gap_open, gap_extend};
    const uint8_t* reference_new = reference + rs;
    const size_t rlen = re - rs, qlen = qe - qs;
    size_t band_width = (rlen > qlen ? rlen - qlen : qlen - rlen) + 1;
    const size_t max_bandwidth = (qlen - 1) / 2;
    Maybe<Alignment> inner;
    bool have = false;
    while (band_width <= max_bandwidth) {
        Maybe<Alignment> b = sw_banded_align(reference_new, rlen, query_new, band_width);
        if (b.status == SOME && b.value.score == score) {
            inner = b;
            have = true;
            if (how) *how = 1;
            break;
        }
        band_width *= 2;
    }
    if (!have) {
        inner = sw_scalar_align(reference_new, rlen, query_new);
        if (how) *how = 2;
    }
    Alignment a = inner.value;
    const size_t aq_start = a.query_start + qs, aq_end = a.query_end + qs;
    const size_t ar_start = a.ref_start + rs, ar_end = a.ref_end + rs;
    prepend_ciglet(a.states, {aq_start, 'S'});         // prepend_soft_clip (state.rs:244-246)
    a.states.soft_clip(query_len - aq_end);
    out.status = SOME;
    out.value.score = score;
    out.value.ref_start = ar_start;
    out.value.ref_end = ar_end;
    out.value.query_start = aq_start;
    out.value.query_end = aq_end;
    out.value.states = a.states;
    out.value.ref_len = ref_len;
    out.value.query_len = query_len;
    return out;
}

// ---- sw_score_from_path (sw/mod.rs:399-454); returns -1 on any ScoringError ----
inline int64_t sw_score_from_path(const std::vector<Ciglet>& ciglets, const uint8_t* ref_in_alignment, size_t ref_n,
                                  const ScalarProfile& q) {
    int64_t score = 0;
    size_t r = 0, c = 0;
    for (auto g : ciglets) {
        switch (g.op) {
            case 'M':
            case '=':
            case 'X':
                for (size_t i = 0; i < g.inc; ++i) {
                    if (r >= ref_n || c >= q.len) return -1;
                    score += q.weight(ref_in_alignment[r], q.seq[c]);
                    ++r;
                    ++c;
                }
                break;
            case 'I':
                score += q.gap_open + q.gap_extend * int64_t(g.inc - 1);
                c += g.inc;
                break;
            case 'D':
                score += q.gap_open + q.gap_extend * int64_t(g.inc - 1);
                r += g.inc;
                break;
            case 'S': c += g.inc; break;
            case 'N': r += g.inc; break;
            case 'H':
            case 'P': break;
            default: return -1;
        }
    }
    if (c != q.len || r != ref_n || score < 0) return -1;
    return score;
}

// ---- sneaky_snake (alignment/sneaky_snake.rs:78-131) ----
// Returns 0 = Some(false), 1 = Some(true), 2 = None. Written the way the reference walks it: diagonals ("rows") of the chip
// maze in order 0..window, each from the current checkpoint to its first obstacle.
inline int sneaky_snake(const uint8_t* reference, size_t ref_len, const uint8_t* query, size_t query_len, float threshold) {
    if (!(threshold >= 0.0f && threshold <= 1.0f)) return 2;                       // :79-81 (NaN is outside the range)
    const size_t edit_thresh = (size_t)std::floor((float)query_len * threshold);   // :83 (f32 product, floor, as usize)
    const size_t len_diff = ref_len > query_len ? ref_len - query_len : query_len - ref_len;
    if (len_diff > edit_thresh) return 2;                                          // :86-88
    if (edit_thresh == query_len) return 1;                                        // :89-91
    const uint8_t *s1 = reference, *s2 = query;                                    // :94-98 s1 = the shorter one
    size_t n1 = ref_len, n2 = query_len;
    if (ref_len > query_len) {
        s1 = query, n1 = query_len;
        s2 = reference, n2 = ref_len;
    }
    const size_t window = 2 * edit_thresh + 1, diffpad_len = len_diff / 2;
    size_t obstacles = 0, checkpoint = 0;
    while (checkpoint < n1 && obstacles <= edit_thresh && n1 - checkpoint > edit_thresh - obstacles) {  // :106
        size_t last_col = checkpoint;
        for (size_t row = 0; row < window; ++row) {
            for (size_t col = checkpoint; col < n1; ++col) {
                const size_t shifted = col + row + diffpad_len;                    // :113 checked_sub(edit_thresh)
                if (shifted >= edit_thresh && shifted - edit_thresh < n2 && s2[shifted - edit_thresh] == s1[col]) {
                    if (col == n1 - 1 || n1 - col - 1 <= edit_thresh - obstacles) return 1;  // :116-118
                } else {
                    last_col = std::max(last_col, col);                            // :120-121
                    break;
                }
            }
        }
        checkpoint = last_col + 1;                                                 // :126-127
        ++obstacles;
    }
    return obstacles <= edit_thresh ? 1 : 0;                                       // :130
}

}  // namespace zor
