// zoe_sw.hpp — header-only C++ mirror of the reference's interface over the C ABI (zoe_sw.h).
//
// Names follow the reference (file:line in the reference checkout):
//   zoe::ByteIndexMap / DNA_PROFILE_MAP     src/data/constants/mappings/byte_index.rs:231-358, dna.rs:177-178
//   zoe::WeightMatrix                       src/data/matrices/mod.rs:230-546
//   zoe::ProfileError                       src/alignment/errors.rs:6-15
//   zoe::MaybeAligned / Alignment           src/alignment/types/output.rs:18-25, 264-279
//   zoe::StripedProfileBatch                src/alignment/profile.rs:198-552 (one StripedProfile<T,N,S> per read of a batch)
//   zoe::LocalProfilesBatch                 src/alignment/profile_set.rs:382-483 (one LocalProfiles per read of a batch)
// Host-memory batches; one zoe::GpuContext per GPU.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "zoe_sw.h"

namespace zoe {

struct ProfileError : std::runtime_error {
    int code;  // 1 EmptySequence, 2 GapOpenOutOfRange, 3 GapExtendOutOfRange, 4 BadGapWeights
    explicit ProfileError(int c)
        : std::runtime_error(c == 1 ? "EmptySequence" : c == 2 ? "GapOpenOutOfRange" : c == 3 ? "GapExtendOutOfRange" : "BadGapWeights"), code(c) {}
};
struct GpuError : std::runtime_error {
    int code;
    GpuError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct ByteIndexMap {
    std::array<uint8_t, 256> index_map{};
    std::string byte_keys;
    static ByteIndexMap make(const std::string& keys, char catch_all, bool ignore_case) {
        ByteIndexMap m;
        auto up = [](unsigned char c) -> unsigned char { return (c >= 'a' && c <= 'z') ? c - 32 : c; };
        auto lo = [](unsigned char c) -> unsigned char { return (c >= 'A' && c <= 'Z') ? c + 32 : c; };
        m.byte_keys = keys;
        int ca = -1;
        for (size_t i = 0; i < keys.size(); ++i)
            if ((ignore_case ? up(keys[i]) : (unsigned char)keys[i]) == (ignore_case ? up(catch_all) : (unsigned char)catch_all)) ca = (int)i;
        if (ca < 0) throw std::invalid_argument("The catch_all must be present in the byte_keys.");
        m.index_map.fill((uint8_t)ca);
        for (size_t i = 0; i < keys.size(); ++i) {
            if (ignore_case) {
                m.index_map[up(keys[i])] = (uint8_t)i;
                m.index_map[lo(keys[i])] = (uint8_t)i;
            } else {
                m.index_map[(unsigned char)keys[i]] = (uint8_t)i;
            }
        }
        return m;
    }
    ByteIndexMap add_synonym_ignore_case(char new_key, char previous_key) const {
        ByteIndexMap m = *this;
        const uint8_t idx = m.index_map[(unsigned char)previous_key];
        unsigned char c = (unsigned char)new_key;
        m.index_map[(c >= 'a' && c <= 'z') ? c - 32 : c] = idx;
        m.index_map[(c >= 'A' && c <= 'Z') ? c + 32 : c] = idx;
        return m;
    }
    size_t len() const { return byte_keys.size(); }
    size_t to_index(uint8_t b) const { return index_map[b]; }
};

inline const ByteIndexMap& DNA_PROFILE_MAP() {
    static const ByteIndexMap m = ByteIndexMap::make("ACGTN", 'N', true).add_synonym_ignore_case('U', 'T');
    return m;
}

struct WeightMatrix {
    std::vector<int8_t> weights;  // [ref residue * S + query residue]
    const ByteIndexMap* mapping;
    int S;
    // WeightMatrix::new (matrices/mod.rs:358-399); ignoring < 0: none
    static WeightMatrix make(const ByteIndexMap& map, int8_t matching, int8_t mismatch, int ignoring) {
        WeightMatrix w;
        w.mapping = &map;
        w.S = (int)map.len();
        w.weights.assign((size_t)w.S * w.S, 0);
        const int skip = ignoring >= 0 ? (int)map.to_index((uint8_t)ignoring) : -1;
        for (int i = 0; i < w.S; ++i)
            for (int j = 0; j < w.S; ++j)
                if (!(skip >= 0 && (skip == i || skip == j))) w.weights[(size_t)i * w.S + j] = i == j ? matching : mismatch;
        return w;
    }
    static WeightMatrix new_dna_matrix(int8_t matching, int8_t mismatch, int ignoring) {
        return make(DNA_PROFILE_MAP(), matching, mismatch, ignoring);
    }
};

enum class Status : uint8_t { Some = 0, Overflowed = 1, Unmapped = 2, Empty = 3 };

template <typename T> struct MaybeAligned {
    Status status = Status::Unmapped;
    T value{};
    bool is_some() const { return status == Status::Some; }
    const T& unwrap() const {
        if (status == Status::Overflowed) throw std::runtime_error("Alignment score overflowed!");
        if (status != Status::Some) throw std::runtime_error("Sequence could not be mapped!");
        return value;
    }
};

struct Ciglet {
    size_t inc;
    uint8_t op;
};
struct Alignment {
    uint32_t score = 0;
    size_t ref_start = 0, ref_end = 0, query_start = 0, query_end = 0, ref_len = 0, query_len = 0;
    std::vector<Ciglet> states;
    std::string cigar() const {
        std::string s;
        for (auto& c : states) {
            s += std::to_string(c.inc);
            s.push_back((char)c.op);
        }
        return s;
    }
};

class GpuContext {
  public:
    explicit GpuContext(int device = 0) {
        zsw_error e = zsw_create(device, &ctx_);
        if (e != ZSW_OK) throw GpuError(e, std::string("zsw_create: ") + zsw_last_error_string(nullptr));
    }
    ~GpuContext() { zsw_destroy(ctx_); }
    GpuContext(const GpuContext&) = delete;
    GpuContext& operator=(const GpuContext&) = delete;
    zsw_context* raw() const { return ctx_; }
    // zsw_set_option(ZSW_OPTION_EXACT_PRUNING): the exact column-pruned first pass (same results for every input)
    void set_exact_pruning(bool on) { check(zsw_set_option(ctx_, ZSW_OPTION_EXACT_PRUNING, on ? 1 : 0)); }
    void check(zsw_error e) const {
        if (e == ZSW_OK) return;
        if (e >= 1 && e <= 4) throw ProfileError(e);
        throw GpuError(e, zsw_last_error_string(ctx_));
    }

  private:
    zsw_context* ctx_ = nullptr;
};

// ScoreEnds / ScoreAndRanges (src/alignment/types/output.rs:201-219); ranges are 0-based half-open
struct ScoreEnds {
    uint32_t score = 0;
    size_t ref_end = 0, query_end = 0;
};
struct ScoreAndRanges {
    uint32_t score = 0;
    size_t ref_start = 0, ref_end = 0, query_start = 0, query_end = 0;
};

// Reads of a batch (host memory) bound to one scoring scheme; the common part of the two profile mirrors below.
class ProfileBatchBase {
  public:
    size_t size() const { return offsets_.size() - 1; }

  protected:
    ProfileBatchBase(GpuContext& ctx, const std::vector<std::string>& reads, const WeightMatrix& matrix, int8_t gap_open, int8_t gap_extend)
        : ctx_(ctx) {
        offsets_.reserve(reads.size() + 1);
        offsets_.push_back(0);
        for (auto& r : reads) {
            if (r.empty()) throw ProfileError(1);  // validate_profile_args (profile.rs:32-44)
            bases_.insert(bases_.end(), r.begin(), r.end());
            offsets_.push_back(bases_.size());
        }
        ctx_.check(zsw_set_scoring(ctx_.raw(), matrix.weights.data(), matrix.S, matrix.mapping->index_map.data(), gap_open, gap_extend));
    }
    zsw_batch batch() const {
        zsw_batch b;
        b.bases = bases_.data();
        b.offsets = offsets_.data();
        b.fixed_len = 0;
        b.n_reads = offsets_.size() - 1;
        b.mem = ZSW_MEM_HOST;
        b.encoding = ZSW_ENCODING_BYTES;
        return b;
    }
    void set_reference(const std::string& reference) {
        ctx_.check(zsw_set_reference(ctx_.raw(), (const uint8_t*)reference.data(), reference.size(), ZSW_MEM_HOST));
    }
    // Runs an align-type entry point (growing the ciglet buffers once if the call asks for it) and rebuilds the Alignments.
    template <typename Call>
    std::vector<MaybeAligned<Alignment>> collect(Call call) {
        const size_t n = size();
        std::vector<zsw_alignment> aln(n);
        std::vector<uint8_t> status(n);
        std::vector<uint32_t> inc(16 * n + 64);
        std::vector<uint8_t> op(inc.size());
        uint64_t total = 0;
        zsw_error e = call(aln.data(), status.data(), inc.data(), op.data(), (uint64_t)inc.size(), &total);
        if (e == ZSW_ERR_INVALID_ARGUMENT && total > inc.size()) {
            inc.resize(total);
            op.resize(total);
            e = call(aln.data(), status.data(), inc.data(), op.data(), (uint64_t)inc.size(), &total);
        }
        ctx_.check(e);
        std::vector<MaybeAligned<Alignment>> out(n);
        for (size_t i = 0; i < n; ++i) {
            out[i].status = (Status)status[i];
            if (status[i] != 0) continue;
            Alignment& a = out[i].value;
            a.score = aln[i].score;
            a.ref_start = aln[i].ref_start;
            a.ref_end = aln[i].ref_end;
            a.query_start = aln[i].query_start;
            a.query_end = aln[i].query_end;
            a.ref_len = aln[i].ref_len;
            a.query_len = aln[i].query_len;
            for (uint32_t k = 0; k < aln[i].n_ciglets; ++k)
                a.states.push_back({inc[aln[i].ciglet_offset + k], op[aln[i].ciglet_offset + k]});
        }
        return out;
    }
    GpuContext& ctx_;
    std::vector<uint8_t> bases_;
    std::vector<uint64_t> offsets_;
};

// `StripedProfile::<T, N, S>::new(read_i, &matrix, gap_open, gap_extend)` for every read of a batch (profile.rs:239-247).
// Unsigned T takes the signed matrix too: the library derives the bias the way to_biased_matrix does (matrices/mod.rs:452-491).
class StripedProfileBatch : public ProfileBatchBase {
  public:
    StripedProfileBatch(GpuContext& ctx, const std::vector<std::string>& reads, const WeightMatrix& matrix, int8_t gap_open,
                        int8_t gap_extend, zsw_int_type T, int N)
        : ProfileBatchBase(ctx, reads, matrix, gap_open, gap_extend), T_(T), N_(N) {}
    // profile.rs:440-446 -> sw_simd_score (striped.rs:65-142)
    std::vector<MaybeAligned<uint32_t>> sw_score(const std::string& reference) {
        set_reference(reference);
        const size_t n = size();
        zsw_batch b = batch();
        std::vector<uint32_t> score(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_batch(ctx_.raw(), &b, T_, N_, score.data(), status.data(), nullptr));
        std::vector<MaybeAligned<uint32_t>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], score[i]};
        return out;
    }
    // profile.rs:456-460 -> sw_simd_score_ends (striped.rs:153-162)
    std::vector<MaybeAligned<ScoreEnds>> sw_score_ends(const std::string& reference) {
        set_reference(reference);
        const size_t n = size();
        zsw_batch b = batch();
        std::vector<uint32_t> score(n), re(n), qe(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_ends_batch(ctx_.raw(), &b, T_, N_, score.data(), re.data(), qe.data(), status.data(), nullptr));
        std::vector<MaybeAligned<ScoreEnds>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], ScoreEnds{score[i], re[i], qe[i]}};
        return out;
    }
    // profile.rs:529-533 -> sw_simd_score_ranges (striped.rs:355-388)
    std::vector<MaybeAligned<ScoreAndRanges>> sw_score_ranges(const std::string& reference) {
        set_reference(reference);
        const size_t n = size();
        zsw_batch b = batch();
        std::vector<uint32_t> score(n), rs(n), re(n), qs(n), qe(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_ranges_batch(ctx_.raw(), &b, T_, N_, score.data(), rs.data(), re.data(), qs.data(), qe.data(), status.data(), nullptr));
        std::vector<MaybeAligned<ScoreAndRanges>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], ScoreAndRanges{score[i], rs[i], re[i], qs[i], qe[i]}};
        return out;
    }
    // profile.rs:515-519 -> sw_simd_align (striped.rs:449-598); seq_is_query = SeqSrc::Query (alignment/mod.rs:176-190)
    std::vector<MaybeAligned<Alignment>> sw_align(const std::string& seq, bool seq_is_query = false) {
        set_reference(seq);
        zsw_batch b = batch();
        return collect([&](zsw_alignment* aln, uint8_t* st, uint32_t* inc, uint8_t* op, uint64_t cap, uint64_t* total) {
            return zsw_align_batch(ctx_.raw(), &b, T_, N_, seq_is_query, aln, st, inc, op, cap, total, nullptr);
        });
    }
    // profile.rs:546-552 -> sw_align_3pass (three_pass.rs:21-104)
    std::vector<MaybeAligned<Alignment>> sw_align_3pass(const std::string& seq, bool seq_is_query = false) {
        set_reference(seq);
        zsw_batch b = batch();
        return collect([&](zsw_alignment* aln, uint8_t* st, uint32_t* inc, uint8_t* op, uint64_t cap, uint64_t* total) {
            return zsw_align_3pass_batch(ctx_.raw(), &b, T_, N_, seq_is_query, aln, st, inc, op, cap, total, nullptr);
        });
    }

  private:
    zsw_int_type T_;
    int N_;
};

// `reads[i].into_local_profile(&matrix, gap_open, gap_extend)` for a whole batch (nucleotides/mod.rs:262-266):
// LocalProfiles with the i8 -> i16 -> i32 cascade of ProfileSets (profile_set.rs:71-283), lane presets w128/w256/w512.
class LocalProfilesBatch : public ProfileBatchBase {
  public:
    LocalProfilesBatch(GpuContext& ctx, const std::vector<std::string>& reads, const WeightMatrix& matrix, int8_t gap_open,
                       int8_t gap_extend, int preset_bits = 256)
        : ProfileBatchBase(ctx, reads, matrix, gap_open, gap_extend), preset_(preset_bits) {}
    // ProfileSets::sw_score_from_i{8,16,32} (profile_set.rs:71-107)
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i8(const std::string& reference) { return score_from(reference, 8); }
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i16(const std::string& reference) { return score_from(reference, 16); }
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i32(const std::string& reference) { return score_from(reference, 32); }
    // ProfileSets::sw_score_ranges_from_i{8,16,32} (profile_set.rs:313-362)
    std::vector<MaybeAligned<ScoreAndRanges>> sw_score_ranges_from_i8(const std::string& reference) { return ranges_from(reference, 8); }
    std::vector<MaybeAligned<ScoreAndRanges>> sw_score_ranges_from_i16(const std::string& reference) { return ranges_from(reference, 16); }
    std::vector<MaybeAligned<ScoreAndRanges>> sw_score_ranges_from_i32(const std::string& reference) { return ranges_from(reference, 32); }
    // ProfileSets::sw_align_from_i{8,16,32} (profile_set.rs:124-179)
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 8, false); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i16(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 16, false); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i32(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 32, false); }
    // ProfileSets::sw_align_from_i{8,16,32}_3pass (profile_set.rs:212-283)
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8_3pass(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 8, true); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i16_3pass(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 16, true); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i32_3pass(const std::string& seq, bool seq_is_query = false) { return align_from(seq, seq_is_query, 32, true); }
    // alignment::sneaky_snake(&reference[ref_start[i]..][..ref_len[i]], read_i, threshold) (sneaky_snake.rs:78-131):
    // true / false per read, Status::Unmapped standing in for `None`.
    std::vector<MaybeAligned<bool>> sneaky_snake(const std::string& reference, const std::vector<uint32_t>& ref_start,
                                                 const std::vector<uint32_t>& ref_len, float threshold) {
        const size_t n = size();
        if (ref_start.size() != n || ref_len.size() != n) throw GpuError(ZSW_ERR_INVALID_ARGUMENT, "one window per read");
        set_reference(reference);
        zsw_batch b = batch();
        std::vector<uint8_t> pass(n);
        ctx_.check(zsw_sneaky_snake_batch(ctx_.raw(), &b, ref_start.data(), ref_len.data(), threshold, pass.data(), nullptr));
        std::vector<MaybeAligned<bool>> out(n);
        for (size_t i = 0; i < n; ++i) {
            out[i].status = pass[i] == ZSW_FILTER_NONE ? Status::Unmapped : Status::Some;
            out[i].value = pass[i] == ZSW_FILTER_PASS;
        }
        return out;
    }
    // the width that answered each read of the last cascade call (8, 16 or 32)
    const std::vector<uint8_t>& last_tiers() const { return tier_; }

  private:
    std::vector<MaybeAligned<Alignment>> align_from(const std::string& seq, bool seq_is_query, int width, bool three_pass) {
        set_reference(seq);
        zsw_batch b = batch();
        tier_.assign(size(), 0);
        return collect([&](zsw_alignment* aln, uint8_t* st, uint32_t* inc, uint8_t* op, uint64_t cap, uint64_t* total) {
            return (three_pass ? zsw_align_3pass_batch_from : zsw_align_batch_from)(ctx_.raw(), &b, width, preset_, seq_is_query, aln, st,
                                                                                   tier_.data(), inc, op, cap, total, nullptr);
        });
    }
    std::vector<MaybeAligned<ScoreAndRanges>> ranges_from(const std::string& reference, int width) {
        set_reference(reference);
        const size_t n = size();
        zsw_batch b = batch();
        std::vector<uint32_t> score(n), rs(n), re(n), qs(n), qe(n);
        std::vector<uint8_t> status(n);
        tier_.assign(n, 0);
        ctx_.check(zsw_score_ranges_batch_from(ctx_.raw(), &b, width, preset_, score.data(), rs.data(), re.data(), qs.data(), qe.data(),
                                               status.data(), tier_.data(), nullptr));
        std::vector<MaybeAligned<ScoreAndRanges>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], ScoreAndRanges{score[i], rs[i], re[i], qs[i], qe[i]}};
        return out;
    }
    std::vector<MaybeAligned<uint32_t>> score_from(const std::string& reference, int width) {
        set_reference(reference);
        const size_t n = size();
        zsw_batch b = batch();
        std::vector<uint32_t> score(n);
        std::vector<uint8_t> status(n);
        tier_.assign(n, 0);
        ctx_.check(zsw_score_batch_from(ctx_.raw(), &b, width, preset_, score.data(), status.data(), tier_.data(), nullptr));
        std::vector<MaybeAligned<uint32_t>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], score[i]};
        return out;
    }
    int preset_;
    std::vector<uint8_t> tier_;
};

// The other role of the striped functions (sw/mod.rs:63-67; SharedProfiles, profile_set.rs:552-560;
// Nucleotides::into_shared_profile, nucleotides/mod.rs:295-299): ONE profile built from `sequence`, used against every read of a
// batch — the reads are the sequences sw_simd_* walks row by row, so in the results `ref_*` is the read and `query_*` the profile
// sequence unless seq_is_query (SeqSrc::Query(read), alignment/mod.rs:176-190) hands the roles back.
class SharedProfileBase {
  protected:
    SharedProfileBase(GpuContext& ctx, const std::string& sequence, const WeightMatrix& matrix, int8_t gap_open, int8_t gap_extend) : ctx_(ctx) {
        if (sequence.empty()) throw ProfileError(1);  // StripedProfile::new -> ProfileError::EmptySequence (profile.rs:32-44)
        ctx_.check(zsw_set_scoring(ctx_.raw(), matrix.weights.data(), matrix.S, matrix.mapping->index_map.data(), gap_open, gap_extend));
        ctx_.check(zsw_set_profile_sequence(ctx_.raw(), (const uint8_t*)sequence.data(), sequence.size(), ZSW_MEM_HOST));
    }
    // the reads as one host batch (an empty read is an empty `reference` argument: Unmapped, striped.rs:219-221)
    struct HostReads {
        std::vector<uint8_t> bases;
        std::vector<uint64_t> offsets;
        zsw_batch batch() const {
            zsw_batch b;
            b.bases = bases.data();
            b.offsets = offsets.data();
            b.fixed_len = 0;
            b.n_reads = offsets.size() - 1;
            b.mem = ZSW_MEM_HOST;
            b.encoding = ZSW_ENCODING_BYTES;
            return b;
        }
    };
    static HostReads pack(const std::vector<std::string>& reads) {
        HostReads h;
        h.offsets.push_back(0);
        for (auto& r : reads) {
            h.bases.insert(h.bases.end(), r.begin(), r.end());
            h.offsets.push_back(h.bases.size());
        }
        return h;
    }
    template <typename Call>
    std::vector<MaybeAligned<Alignment>> collect(size_t n, Call call) {
        std::vector<zsw_alignment> aln(n);
        std::vector<uint8_t> status(n);
        std::vector<uint32_t> inc(16 * n + 64);
        std::vector<uint8_t> op(inc.size());
        uint64_t total = 0;
        zsw_error e = call(aln.data(), status.data(), inc.data(), op.data(), (uint64_t)inc.size(), &total);
        if (e == ZSW_ERR_INVALID_ARGUMENT && total > inc.size()) {
            inc.resize(total);
            op.resize(total);
            e = call(aln.data(), status.data(), inc.data(), op.data(), (uint64_t)inc.size(), &total);
        }
        ctx_.check(e);
        std::vector<MaybeAligned<Alignment>> out(n);
        for (size_t i = 0; i < n; ++i) {
            out[i].status = (Status)status[i];
            if (status[i] != 0) continue;
            Alignment& a = out[i].value;
            a.score = aln[i].score;
            a.ref_start = aln[i].ref_start;
            a.ref_end = aln[i].ref_end;
            a.query_start = aln[i].query_start;
            a.query_end = aln[i].query_end;
            a.ref_len = aln[i].ref_len;
            a.query_len = aln[i].query_len;
            for (uint32_t k = 0; k < aln[i].n_ciglets; ++k)
                a.states.push_back({inc[aln[i].ciglet_offset + k], op[aln[i].ciglet_offset + k]});
        }
        return out;
    }
    GpuContext& ctx_;
};

// `StripedProfile::<T, N, S>::new(sequence, ..)` once; sw_score / sw_score_ends / sw_score_ranges / sw_align against a batch of reads
class SharedStripedProfile : public SharedProfileBase {
  public:
    SharedStripedProfile(GpuContext& ctx, const std::string& sequence, const WeightMatrix& matrix, int8_t gap_open, int8_t gap_extend,
                         zsw_int_type T, int N)
        : SharedProfileBase(ctx, sequence, matrix, gap_open, gap_extend), T_(T), N_(N) {}
    std::vector<MaybeAligned<uint32_t>> sw_score(const std::vector<std::string>& reads) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        const size_t n = reads.size();
        std::vector<uint32_t> score(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_shared_batch(ctx_.raw(), &b, T_, N_, score.data(), status.data(), nullptr));
        std::vector<MaybeAligned<uint32_t>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], score[i]};
        return out;
    }
    // ref_end: exclusive end in the read, query_end: in the profile sequence (first read position holding the maximum, then the
    // first sequence position)
    std::vector<MaybeAligned<ScoreEnds>> sw_score_ends(const std::vector<std::string>& reads) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        const size_t n = reads.size();
        std::vector<uint32_t> score(n), re(n), qe(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_ends_shared_batch(ctx_.raw(), &b, T_, N_, score.data(), re.data(), qe.data(), status.data(), nullptr));
        std::vector<MaybeAligned<ScoreEnds>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], ScoreEnds{score[i], re[i], qe[i]}};
        return out;
    }
    std::vector<MaybeAligned<ScoreAndRanges>> sw_score_ranges(const std::vector<std::string>& reads) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        const size_t n = reads.size();
        std::vector<uint32_t> score(n), rs(n), re(n), qs(n), qe(n);
        std::vector<uint8_t> status(n);
        ctx_.check(zsw_score_ranges_shared_batch(ctx_.raw(), &b, T_, N_, score.data(), rs.data(), re.data(), qs.data(), qe.data(), status.data(), nullptr));
        std::vector<MaybeAligned<ScoreAndRanges>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], ScoreAndRanges{score[i], rs[i], re[i], qs[i], qe[i]}};
        return out;
    }
    // seq_is_query = true: SeqSrc::Query(read) — the usual call when the profile is the reference
    std::vector<MaybeAligned<Alignment>> sw_align(const std::vector<std::string>& reads, bool seq_is_query = true) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        return collect(reads.size(), [&](zsw_alignment* aln, uint8_t* st, uint32_t* inc, uint8_t* op, uint64_t cap, uint64_t* total) {
            return zsw_align_shared_batch(ctx_.raw(), &b, T_, N_, seq_is_query, aln, st, inc, op, cap, total, nullptr);
        });
    }

  private:
    zsw_int_type T_;
    int N_;
};

// `sequence.into_shared_profile(&matrix, gap_open, gap_extend)`: SharedProfiles with the i8 -> i16 -> i32 cascade
class SharedProfilesBatch : public SharedProfileBase {
  public:
    SharedProfilesBatch(GpuContext& ctx, const std::string& sequence, const WeightMatrix& matrix, int8_t gap_open, int8_t gap_extend,
                        int preset_bits = 256)
        : SharedProfileBase(ctx, sequence, matrix, gap_open, gap_extend), preset_(preset_bits) {}
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i8(const std::vector<std::string>& reads) { return score_from(reads, 8); }
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i16(const std::vector<std::string>& reads) { return score_from(reads, 16); }
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i32(const std::vector<std::string>& reads) { return score_from(reads, 32); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 8); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i16(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 16); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i32(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 32); }
    // ProfileSets::sw_align_from_i*_3pass (profile_set.rs:212-283) of SharedProfiles (:552-560)
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8_3pass(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 8, true); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i16_3pass(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 16, true); }
    std::vector<MaybeAligned<Alignment>> sw_align_from_i32_3pass(const std::vector<std::string>& reads, bool seq_is_query = true) { return align_from(reads, seq_is_query, 32, true); }
    const std::vector<uint8_t>& last_tiers() const { return tier_; }

  private:
    std::vector<MaybeAligned<uint32_t>> score_from(const std::vector<std::string>& reads, int width) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        const size_t n = reads.size();
        std::vector<uint32_t> score(n);
        std::vector<uint8_t> status(n);
        tier_.assign(n, 0);
        ctx_.check(zsw_score_shared_batch_from(ctx_.raw(), &b, width, preset_, score.data(), status.data(), tier_.data(), nullptr));
        std::vector<MaybeAligned<uint32_t>> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = {(Status)status[i], score[i]};
        return out;
    }
    std::vector<MaybeAligned<Alignment>> align_from(const std::vector<std::string>& reads, bool seq_is_query, int width, bool three_pass = false) {
        const HostReads h = pack(reads);
        const zsw_batch b = h.batch();
        tier_.assign(reads.size(), 0);
        return collect(reads.size(), [&](zsw_alignment* aln, uint8_t* st, uint32_t* inc, uint8_t* op, uint64_t cap, uint64_t* total) {
            return (three_pass ? zsw_align_3pass_shared_batch_from : zsw_align_shared_batch_from)(ctx_.raw(), &b, width, preset_, seq_is_query, aln, st,
                                                                                                 tier_.data(), inc, op, cap, total, nullptr);
        });
    }
    int preset_;
    std::vector<uint8_t> tier_;
};

}  // namespace zoe
