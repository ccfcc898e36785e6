// zoe_oracle_capi.cpp — TEST INFRASTRUCTURE ONLY: C entry points (for ctypes)
// over the CPU restatement in zoe_oracle.hpp. Not part of the product path.
#include <cstring>
#include <string>

#include "zoe_oracle.hpp"

using namespace zor;

namespace {

struct Ctx {
    ByteIndexMap map;
    WeightMatrixI8 wm;
};

Ctx make_ctx(int S, const int8_t* weights, const uint8_t* index_map) {
    Ctx c;
    c.map.S = S;
    std::memcpy(c.map.index_map, index_map, 256);
    c.wm.S = S;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) c.wm.w[i][j] = weights[i * S + j];
    return c;
}

// type codes: 0 i8, 1 i16, 2 i32, 3 u8, 4 u16, 5 u32
template <typename Fn> int dispatch(int tcode, int lanes, Fn&& fn) {
#define ZOR_LANES(T)                                           \
    switch (lanes) {                                           \
        case 2: return fn(T{}, std::integral_constant<int, 2>{});   \
        case 4: return fn(T{}, std::integral_constant<int, 4>{});   \
        case 8: return fn(T{}, std::integral_constant<int, 8>{});   \
        case 16: return fn(T{}, std::integral_constant<int, 16>{}); \
        case 32: return fn(T{}, std::integral_constant<int, 32>{}); \
        case 64: return fn(T{}, std::integral_constant<int, 64>{}); \
        default: return -2;                                    \
    }
    switch (tcode) {
        case 0: ZOR_LANES(int8_t)
        case 1: ZOR_LANES(int16_t)
        case 2: ZOR_LANES(int32_t)
        case 3: ZOR_LANES(uint8_t)
        case 4: ZOR_LANES(uint16_t)
        case 5: ZOR_LANES(uint32_t)
        default: return -2;
    }
#undef ZOR_LANES
}

void put_alignment(const Maybe<Alignment>& m, bool invert, uint32_t* out_status, uint64_t* out_fields, char* cigar_buf,
                   size_t cigar_cap) {
    *out_status = m.status;
    if (m.status != SOME) return;
    Alignment a = invert ? m.value.invert() : m.value;
    out_fields[0] = a.score;
    out_fields[1] = a.ref_start;
    out_fields[2] = a.ref_end;
    out_fields[3] = a.query_start;
    out_fields[4] = a.query_end;
    out_fields[5] = a.ref_len;
    out_fields[6] = a.query_len;
    out_fields[7] = a.states.c.size();
    std::string s = a.states.to_string();
    if (cigar_buf && cigar_cap) {
        size_t n = std::min(s.size(), cigar_cap - 1);
        std::memcpy(cigar_buf, s.data(), n);
        cigar_buf[n] = 0;
    }
}

}  // namespace

extern "C" {

int zor_validate_profile_args(size_t seq_len, int gap_open, int gap_extend) {
    return validate_profile_args(seq_len, gap_open, gap_extend);
}

void zor_dna_profile_map(uint8_t* out256) {
    ByteIndexMap m = ByteIndexMap::dna_profile_map();
    std::memcpy(out256, m.index_map, 256);
}

void zor_byte_index_map(const uint8_t* keys, int S, uint8_t catch_all, int ignore_case, uint8_t* out256) {
    ByteIndexMap m = ByteIndexMap::make(keys, S, catch_all, ignore_case != 0);
    std::memcpy(out256, m.index_map, 256);
}

// WeightMatrix::new → S*S signed weights
void zor_weight_matrix_new(const uint8_t* index_map, int S, int matching, int mismatch, int ignoring, int8_t* out) {
    ByteIndexMap m;
    m.S = S;
    std::memcpy(m.index_map, index_map, 256);
    WeightMatrixI8 w = WeightMatrixI8::make(m, int8_t(matching), int8_t(mismatch), ignoring);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) out[i * S + j] = w.w[i][j];
}

// to_biased_matrix → S*S u8 weights + bias
int zor_to_biased_matrix(const int8_t* weights, int S, uint8_t* out) {
    WeightMatrixI8 w;
    w.S = S;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) w.w[i][j] = weights[i * S + j];
    ProfileWeights p = ProfileWeights::from(w, false);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) out[i * S + j] = uint8_t(p.w[i][j]);
    return p.bias;
}

// Dumps StripedProfile::new (or, if rev_end > 0, .reverse_from_forward(rev_end)) as int64 elements
// [S*nv][N]; returns nv (or <0 on error / None).
long zor_profile_dump(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                      int gap_extend, const uint8_t* seq, size_t len, size_t rev_end, int64_t* out, size_t out_cap) {
    int e = validate_profile_args(len, gap_open, gap_extend);
    if (e) return -long(e);
    Ctx c = make_ctx(S, weights, index_map);
    long nv_out = -100;
    int rc = dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(seq, len, pw, c.map, gap_open, gap_extend);
        StripedProfile<T, N> r;
        const StripedProfile<T, N>* use = &p;
        if (rev_end > 0) {
            if (!p.reverse_from_forward(rev_end, r)) return -3;
            use = &r;
        }
        size_t total = use->profile.size() * size_t(N);
        if (total > out_cap) return -4;
        for (size_t i = 0; i < use->profile.size(); ++i)
            for (int l = 0; l < N; ++l) out[i * size_t(N) + size_t(l)] = int64_t(use->profile[i].v[l]);
        nv_out = long(use->number_vectors());
        return 0;
    });
    return rc < 0 ? rc : nv_out;
}

// StripedProfile::<T,N,S>::new(prof_seq).sw_score(other)  (profile.rs:440-446 → striped.rs:65)
int zor_score(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
              const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len, uint32_t* out_status,
              uint32_t* out_score) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    return dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(prof_seq, prof_len, pw, c.map, gap_open, gap_extend);
        *out_score = 0;
        *out_status = sw_simd_score<T, N>(other, other_len, p, out_score);
        return 0;
    });
}

// sw_score_ends (forward != 0) or sw_score_ends_reverse (forward == 0; caller passes an already reversed profile seq)
int zor_score_ends(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                   int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                   int forward, uint32_t* out_status, uint64_t* out3) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    return dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(prof_seq, prof_len, pw, c.map, gap_open, gap_extend);
        Maybe<ScoreIndices> m = forward ? sw_simd_score_ends_dir<T, N, true>(other, other_len, p)
                                        : sw_simd_score_ends_dir<T, N, false>(other, other_len, p);
        *out_status = m.status;
        if (m.status == SOME) {
            out3[0] = m.value.score;
            out3[1] = m.value.ref_idx;
            out3[2] = m.value.query_idx;
        }
        return 0;
    });
}

int zor_score_ranges(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                     int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                     uint32_t* out_status, uint64_t* out5) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    return dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(prof_seq, prof_len, pw, c.map, gap_open, gap_extend);
        auto m = sw_simd_score_ranges<T, N>(other, other_len, p);
        *out_status = m.status;
        if (m.status == SOME) {
            out5[0] = m.value.score;
            out5[1] = m.value.ref_start;
            out5[2] = m.value.ref_end;
            out5[3] = m.value.query_start;
            out5[4] = m.value.query_end;
        }
        return 0;
    });
}

// profile.sw_align(SeqSrc::Reference(other)) (other_is_query == 0) or SeqSrc::Query(other) (inverted result).
// out_fields: score, ref_start, ref_end, query_start, query_end, ref_len, query_len, n_ciglets
// flags_out (optional): R*nv*N raw striped flag bytes.
int zor_align(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
              const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len, int other_is_query,
              uint32_t* out_status, uint64_t* out_fields, char* cigar_buf, size_t cigar_cap, uint8_t* flags_out,
              size_t flags_cap) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    return dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(prof_seq, prof_len, pw, c.map, gap_open, gap_extend);
        std::vector<uint8_t> flags;
        auto m = sw_simd_align<T, N>(other, other_len, p, flags_out ? &flags : nullptr);
        if (flags_out && m.status != OVERFLOWED && other_len > 0) {
            if (flags.size() > flags_cap) return -4;
            std::memcpy(flags_out, flags.data(), flags.size());
        }
        put_alignment(m, other_is_query != 0, out_status, out_fields, cigar_buf, cigar_cap);
        return 0;
    });
}

int zor_scalar_score(int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
                     const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                     uint32_t* out_status, uint32_t* out_score) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    ScalarProfile q{prof_seq, prof_len, &c.wm, &c.map, gap_open, gap_extend};
    *out_score = 0;
    *out_status = sw_scalar_score(other, other_len, q, out_score);
    return 0;
}

int zor_scalar_align(int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
                     const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len, int other_is_query,
                     uint32_t* out_status, uint64_t* out_fields, char* cigar_buf, size_t cigar_cap) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    ScalarProfile q{prof_seq, prof_len, &c.wm, &c.map, gap_open, gap_extend};
    auto m = sw_scalar_align(other, other_len, q);
    put_alignment(m, other_is_query != 0, out_status, out_fields, cigar_buf, cigar_cap);
    return 0;
}

// ProfileSets::sw_score_from_{i8,i16,i32} (profile_set.rs:71-107) with lane presets (:434-483)
// from_width: 8, 16, 32; preset: 128, 256, 512 (bits)
int zor_cascade_score(int from_width, int preset, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                      int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                      uint32_t* out_status, uint32_t* out_score, int* out_tier) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    const int widths[3] = {8, 16, 32};
    for (int k = 0; k < 3; ++k) {
        if (widths[k] < from_width) continue;
        int lanes = preset / widths[k];
        int rc = zor_score(k, lanes, S, weights, index_map, gap_open, gap_extend, prof_seq, prof_len, other, other_len,
                           out_status, out_score);
        if (rc) return rc;
        *out_tier = widths[k];
        if (*out_status != OVERFLOWED) break;  // or_else_overflowed (output.rs:81-83)
    }
    return 0;
}

// ProfileSets::sw_score_ranges_from_{i8,i16,i32} (profile_set.rs:313-362)
int zor_cascade_score_ranges(int from_width, int preset, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                             int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                             uint32_t* out_status, uint64_t* out5, int* out_tier) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    const int widths[3] = {8, 16, 32};
    for (int k = 0; k < 3; ++k) {
        if (widths[k] < from_width) continue;
        int rc = zor_score_ranges(k, preset / widths[k], S, weights, index_map, gap_open, gap_extend, prof_seq, prof_len, other,
                                  other_len, out_status, out5);
        if (rc) return rc;
        *out_tier = widths[k];
        if (*out_status != OVERFLOWED) break;  // or_else_overflowed (output.rs:81-83)
    }
    return 0;
}

// ProfileSets::sw_align_from_{i8,i16,i32} (profile_set.rs:124-179)
int zor_cascade_align(int from_width, int preset, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                      int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                      int other_is_query, uint32_t* out_status, uint64_t* out_fields, char* cigar_buf, size_t cigar_cap,
                      int* out_tier) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    const int widths[3] = {8, 16, 32};
    for (int k = 0; k < 3; ++k) {
        if (widths[k] < from_width) continue;
        int lanes = preset / widths[k];
        int rc = zor_align(k, lanes, S, weights, index_map, gap_open, gap_extend, prof_seq, prof_len, other, other_len,
                           other_is_query, out_status, out_fields, cigar_buf, cigar_cap, nullptr, 0);
        if (rc) return rc;
        *out_tier = widths[k];
        if (*out_status != OVERFLOWED) break;
    }
    return 0;
}

// sw_banded_align (banded.rs:40-133) with a ScalarProfile built from prof_seq
int zor_banded_align(int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend, const uint8_t* prof_seq,
                     size_t prof_len, const uint8_t* other, size_t other_len, size_t band_width, uint32_t* out_status,
                     uint64_t* out_fields, char* cigar_buf, size_t cigar_cap) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    ScalarProfile q{prof_seq, prof_len, &c.wm, &c.map, gap_open, gap_extend};
    auto m = sw_banded_align(other, other_len, q, band_width);
    put_alignment(m, false, out_status, out_fields, cigar_buf, cigar_cap);
    return 0;
}

// StripedProfile::<T,N,S>::new(prof_seq).sw_align_3pass(SeqSrc::Reference(other) | Query, prof_seq, matrix, go, ge)
// (profile.rs:546-552 → three_pass.rs:21-104). out_how: 0 no-gaps shortcut, 1 banded, 2 scalar fallback.
int zor_align_3pass(int tcode, int lanes, int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
                    const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len, int other_is_query,
                    uint32_t* out_status, uint64_t* out_fields, char* cigar_buf, size_t cigar_cap, int* out_how) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    Ctx c = make_ctx(S, weights, index_map);
    return dispatch(tcode, lanes, [&](auto t, auto n) -> int {
        using T = decltype(t);
        constexpr int N = decltype(n)::value;
        ProfileWeights pw = ProfileWeights::from(c.wm, Int<T>::SIGNED);
        auto p = StripedProfile<T, N>::make(prof_seq, prof_len, pw, c.map, gap_open, gap_extend);
        int how = -1;
        auto m = sw_align_3pass<T, N>(other, other_len, p, prof_seq, prof_len, c.wm, c.map, gap_open, gap_extend, &how);
        if (out_how) *out_how = how;
        put_alignment(m, other_is_query != 0, out_status, out_fields, cigar_buf, cigar_cap);
        return 0;
    });
}

// ProfileSets::sw_align_from_i{8,16,32}_3pass (profile_set.rs:212-283)
int zor_cascade_align_3pass(int from_width, int preset, int S, const int8_t* weights, const uint8_t* index_map, int gap_open,
                            int gap_extend, const uint8_t* prof_seq, size_t prof_len, const uint8_t* other, size_t other_len,
                            int other_is_query, uint32_t* out_status, uint64_t* out_fields, char* cigar_buf, size_t cigar_cap,
                            int* out_tier, int* out_how) {
    int e = validate_profile_args(prof_len, gap_open, gap_extend);
    if (e) return e;
    const int widths[3] = {8, 16, 32};
    for (int k = 0; k < 3; ++k) {
        if (widths[k] < from_width) continue;
        int rc = zor_align_3pass(k, preset / widths[k], S, weights, index_map, gap_open, gap_extend, prof_seq, prof_len, other,
                                 other_len, other_is_query, out_status, out_fields, cigar_buf, cigar_cap, out_how);
        if (rc) return rc;
        *out_tier = widths[k];
        if (*out_status != OVERFLOWED) break;
    }
    return 0;
}

// sneaky_snake (alignment/sneaky_snake.rs:78-131): 0 Some(false), 1 Some(true), 2 None
int zor_sneaky_snake(const uint8_t* reference, size_t ref_len, const uint8_t* query, size_t query_len, float threshold) {
    return sneaky_snake(reference, ref_len, query, query_len, threshold);
}

// sw_score_from_path over a CIGAR string; -1 on any ScoringError
long long zor_score_from_path(int S, const int8_t* weights, const uint8_t* index_map, int gap_open, int gap_extend,
                              const uint8_t* query, size_t query_len, const uint8_t* ref_in_alignment, size_t ref_n,
                              const char* cigar) {
    Ctx c = make_ctx(S, weights, index_map);
    ScalarProfile q{query, query_len, &c.wm, &c.map, gap_open, gap_extend};
    std::vector<Ciglet> cs;
    size_t inc = 0;
    bool have = false;
    for (const char* p = cigar; *p; ++p) {
        if (*p >= '0' && *p <= '9') {
            inc = inc * 10 + size_t(*p - '0');
            have = true;
        } else {
            if (!have) return -1;
            cs.push_back({inc, uint8_t(*p)});
            inc = 0;
            have = false;
        }
    }
    return sw_score_from_path(cs, ref_in_alignment, ref_n, q);
}

}  // extern "C"
