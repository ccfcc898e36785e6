// zoe_cpu_fast.cpp — TEST INFRASTRUCTURE ONLY: the timed CPU baseline.
//
// The same algorithm as zoe_oracle.hpp's sw_simd_score (reference:
// src/alignment/sw/striped.rs:65-142), restated with 256-bit AVX2 intrinsics for
// the two instantiations `Nucleotides::into_local_profile` uses at its w256
// preset (src/data/types/nucleotides/mod.rs:262-266, profile_set.rs:451-466):
// StripedProfile<i8, 32, S> and StripedProfile<i16, 16, S>; the i32 tier falls
// back to the plain-array restatement. It is labelled "restated Zoe CPU path"
// in every report: Zoe itself (nightly Rust) cannot be built here.
//
// Role convention (sw/mod.rs:119-120): the read is the profile ("query"), a
// fresh profile is built per read, and the long sequence is `reference`.
#include <immintrin.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "zoe_oracle.hpp"

using namespace zor;

namespace {

struct I16 {
    using T = int16_t;
    static constexpr int N = 16;
    static __m256i splat(int x) { return _mm256_set1_epi16(short(x)); }
    static __m256i adds(__m256i a, __m256i b) { return _mm256_adds_epi16(a, b); }
    static __m256i subs(__m256i a, __m256i b) { return _mm256_subs_epi16(a, b); }
    static __m256i vmax(__m256i a, __m256i b) { return _mm256_max_epi16(a, b); }
    static bool any_gt(__m256i a, __m256i b) { return _mm256_movemask_epi8(_mm256_cmpgt_epi16(a, b)) != 0; }
    // shift_elements_right::<1>(MIN)
    static __m256i shr1(__m256i v) {
        __m256i t = _mm256_permute2x128_si256(v, v, 0x08);  // [0, v.lo]
        __m256i s = _mm256_alignr_epi8(v, t, 14);
        return _mm256_insert_epi16(s, short(-32768), 0);
    }
    static int reduce_max(__m256i v) {
        alignas(32) int16_t a[16];
        _mm256_store_si256((__m256i*)a, v);
        int m = a[0];
        for (int i = 1; i < 16; ++i) m = a[i] > m ? a[i] : m;
        return m;
    }
};

struct I8 {
    using T = int8_t;
    static constexpr int N = 32;
    static __m256i splat(int x) { return _mm256_set1_epi8(char(x)); }
    static __m256i adds(__m256i a, __m256i b) { return _mm256_adds_epi8(a, b); }
    static __m256i subs(__m256i a, __m256i b) { return _mm256_subs_epi8(a, b); }
    static __m256i vmax(__m256i a, __m256i b) { return _mm256_max_epi8(a, b); }
    static bool any_gt(__m256i a, __m256i b) { return _mm256_movemask_epi8(_mm256_cmpgt_epi8(a, b)) != 0; }
    static __m256i shr1(__m256i v) {
        __m256i t = _mm256_permute2x128_si256(v, v, 0x08);
        __m256i s = _mm256_alignr_epi8(v, t, 15);
        return _mm256_insert_epi8(s, char(-128), 0);
    }
    static int reduce_max(__m256i v) {
        alignas(32) int8_t a[32];
        _mm256_store_si256((__m256i*)a, v);
        int m = a[0];
        for (int i = 1; i < 32; ++i) m = a[i] > m ? a[i] : m;
        return m;
    }
};

struct Scoring {
    int S;
    int8_t w[MAX_S][MAX_S];
    uint8_t index_map[256];
    int gap_open, gap_extend;  // negative
};

struct Scratch {
    std::vector<__m256i> profile, load, store, e;
};

// StripedProfile::new_unchecked (profile.rs:270-306), signed T
template <typename K>
size_t build_profile(const Scoring& sc, const uint8_t* seq, size_t len, std::vector<__m256i>& profile) {
    using T = typename K::T;
    constexpr int N = K::N;
    const size_t nv = (len + N - 1) / N;
    profile.resize(size_t(sc.S) * nv);
    alignas(32) T lanes[N];
    for (size_t v = 0; v < nv; ++v)
        for (int ri = 0; ri < sc.S; ++ri) {
            for (int i = 0; i < N; ++i) {
                size_t q = v + size_t(i) * nv;
                lanes[i] = q < len ? T(sc.w[ri][sc.index_map[seq[q]]]) : T(0);
            }
            profile[size_t(ri) * nv + v] = _mm256_load_si256((const __m256i*)lanes);
        }
    return nv;
}

template <typename K>
Status run_profile(const Scoring& sc, const std::vector<__m256i>& profile, size_t nv, const uint8_t* reference, size_t ref_len,
                   Scratch& s, uint32_t* score);

// fresh profile per call + sw_simd_score (striped.rs:65-142)
template <typename K>
Status score_one(const Scoring& sc, const uint8_t* read, size_t len, const uint8_t* reference, size_t ref_len,
                 Scratch& s, uint32_t* score) {
    const size_t nv = build_profile<K>(sc, read, len, s.profile);
    return run_profile<K>(sc, s.profile, nv, reference, ref_len, s, score);
}

// sw_simd_score (striped.rs:65-142) against a prebuilt profile
template <typename K>
Status run_profile(const Scoring& sc, const std::vector<__m256i>& profile, size_t nv, const uint8_t* reference, size_t ref_len,
                   Scratch& s, uint32_t* score) {
    using T = typename K::T;
    const __m256i minimums = K::splat(Int<T>::MIN);
    const __m256i go = K::splat(-sc.gap_open), ge = K::splat(-sc.gap_extend);
    s.load.assign(nv, minimums);
    s.store.assign(nv, minimums);
    s.e.assign(nv, minimums);
    __m256i* load = s.load.data();
    __m256i* store = s.store.data();
    __m256i* es = s.e.data();
    __m256i max_scores = minimums;
    for (size_t r = 0; r < ref_len; ++r) {
        size_t ref_index = sc.index_map[reference[r]];
        __m256i F = minimums;
        __m256i H = K::shr1(store[nv - 1]);
        std::swap(load, store);
        const __m256i* scores_vec = &profile[ref_index * nv];
        for (size_t j = 0; j < nv; ++j) {
            __m256i E = es[j];
            H = K::adds(H, scores_vec[j]);
            H = K::vmax(K::vmax(H, E), F);
            max_scores = K::vmax(max_scores, H);
            store[j] = H;
            H = K::subs(H, go);
            E = K::vmax(K::subs(E, ge), H);
            F = K::vmax(K::subs(F, ge), H);
            es[j] = E;
            H = load[j];
        }
        size_t j = 0;
        H = store[j];
        F = K::shr1(F);
        while (K::any_gt(F, K::subs(H, go))) {
            H = K::vmax(H, F);
            store[j] = H;
            F = K::subs(F, ge);
            j += 1;
            if (j >= nv) {
                j = 0;
                F = K::shr1(F);
            }
            H = store[j];
        }
    }
    T best = T(K::reduce_max(max_scores));
    return score_to_maybe_aligned<T>(best, T(0), score);
}

Status score_i32(const Scoring& sc, const uint8_t* read, size_t len, const uint8_t* reference, size_t ref_len,
                 uint32_t* score) {
    ByteIndexMap map;
    map.S = sc.S;
    std::memcpy(map.index_map, sc.index_map, 256);
    WeightMatrixI8 wm;
    wm.S = sc.S;
    for (int i = 0; i < sc.S; ++i)
        for (int j = 0; j < sc.S; ++j) wm.w[i][j] = sc.w[i][j];
    ProfileWeights pw = ProfileWeights::from(wm, true);
    auto p = StripedProfile<int32_t, 8>::make(read, len, pw, map, sc.gap_open, sc.gap_extend);
    return sw_simd_score<int32_t, 8>(reference, ref_len, p, score);
}

}  // namespace

extern "C" {

// Batched `read.into_local_profile(..).sw_score_from_{i8,i16}(reference)` at the w256 preset.
//   from_width 8  : i8x32 → i16x16 → i32x8   (ProfileSets::sw_score_from_i8, profile_set.rs:71-78)
//   from_width 16 : i16x16 → i32x8           (sw_score_from_i16, :90-97)
// reads: concatenated bytes; offsets[n+1] (or NULL with fixed_len). Returns a ProfileError code of the
// first invalid read/argument (0 = ok). out_tier (optional) = width of the tier that answered.
int zor_batch_score_w256(int from_width, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                         int gap_extend, const uint8_t* reads, const uint64_t* offsets, size_t fixed_len, size_t n_reads,
                         const uint8_t* reference, size_t ref_len, int threads, uint32_t* out_score,
                         uint8_t* out_status, uint8_t* out_tier) {
    if (S > MAX_S) return -2;
    Scoring sc;
    sc.S = S;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) sc.w[i][j] = weights[i * S + j];
    std::memcpy(sc.index_map, index_map, 256);
    sc.gap_open = gap_open;
    sc.gap_extend = gap_extend;
    int e = validate_profile_args(1, gap_open, gap_extend);
    if (e) return e;
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    auto worker = [&]() {
        Scratch s;
        const size_t CH = 256;
        for (;;) {
            size_t b = next.fetch_add(CH);
            if (b >= n_reads) break;
            size_t eidx = std::min(n_reads, b + CH);
            for (size_t i = b; i < eidx; ++i) {
                size_t off = offsets ? size_t(offsets[i]) : i * fixed_len;
                size_t len = offsets ? size_t(offsets[i + 1] - offsets[i]) : fixed_len;
                if (len == 0) {
                    err.store(EMPTY_SEQUENCE);
                    out_status[i] = UNMAPPED;
                    out_score[i] = 0;
                    continue;
                }
                uint32_t score = 0;
                Status st = OVERFLOWED;
                int tier = 0;
                if (from_width <= 8) {
                    st = score_one<I8>(sc, reads + off, len, reference, ref_len, s, &score);
                    tier = 8;
                }
                if (st == OVERFLOWED && from_width <= 16) {
                    st = score_one<I16>(sc, reads + off, len, reference, ref_len, s, &score);
                    tier = 16;
                }
                if (st == OVERFLOWED) {
                    st = score_i32(sc, reads + off, len, reference, ref_len, &score);
                    tier = 32;
                }
                out_status[i] = uint8_t(st);
                out_score[i] = st == SOME ? score : 0;
                if (out_tier) out_tier[i] = uint8_t(tier);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return err.load();
}

// The other way Zoe is meant to be used for many reads vs one reference (SharedProfiles, sw/mod.rs:73-78): ONE profile
// built from the long sequence, every read passed as the `reference` argument of sw_simd_score. Scores are role-symmetric
// for symmetric matrices (profile.rs:443-445). i8x32 -> i16x16 -> i32 cascade, w256.
int zor_batch_score_shared_w256(int from_width, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                                int gap_extend, const uint8_t* reads, const uint64_t* offsets, size_t fixed_len, size_t n_reads,
                                const uint8_t* profile_seq, size_t profile_len, int threads, uint32_t* out_score,
                                uint8_t* out_status) {
    if (S > MAX_S) return -2;
    Scoring sc;
    sc.S = S;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) sc.w[i][j] = weights[j * S + i];  // the profile sequence plays the query role: transpose
    std::memcpy(sc.index_map, index_map, 256);
    sc.gap_open = gap_open;
    sc.gap_extend = gap_extend;
    int e = validate_profile_args(profile_len, gap_open, gap_extend);
    if (e) return e;
    std::vector<__m256i> p8, p16;
    const size_t nv8 = build_profile<I8>(sc, profile_seq, profile_len, p8);
    const size_t nv16 = build_profile<I16>(sc, profile_seq, profile_len, p16);
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Scratch s;
        for (;;) {
            size_t b = next.fetch_add(256);
            if (b >= n_reads) break;
            size_t eidx = std::min(n_reads, b + 256);
            for (size_t i = b; i < eidx; ++i) {
                size_t off = offsets ? size_t(offsets[i]) : i * fixed_len;
                size_t len = offsets ? size_t(offsets[i + 1] - offsets[i]) : fixed_len;
                uint32_t score = 0;
                Status st = OVERFLOWED;
                if (from_width <= 8) st = run_profile<I8>(sc, p8, nv8, reads + off, len, s, &score);
                if (st == OVERFLOWED) st = run_profile<I16>(sc, p16, nv16, reads + off, len, s, &score);
                if (st == OVERFLOWED) st = score_i32(sc, profile_seq, profile_len, reads + off, len, &score);
                out_status[i] = uint8_t(st);
                out_score[i] = st == SOME ? score : 0;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return 0;
}

int zor_hardware_threads() { return int(std::thread::hardware_concurrency()); }

}  // extern "C"
