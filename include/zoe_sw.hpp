// zoe_sw.hpp — header-only C++ mirror of the reference's interface over the C ABI (zoe_sw.h).
//
// Names follow the reference (file:line in the reference checkout):
//   zoe::ByteIndexMap / DNA_PROFILE_MAP     src/data/constants/mappings/byte_index.rs:231-358, dna.rs:177-178
//   zoe::WeightMatrix                       src/data/matrices/mod.rs:230-546
//   zoe::ProfileError                       src/alignment/errors.rs:6-15
//   zoe::MaybeAligned / Alignment           src/alignment/types/output.rs:18-25, 264-279
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
    void check(zsw_error e) const {
        if (e == ZSW_OK) return;
        if (e >= 1 && e <= 4) throw ProfileError(e);
        throw GpuError(e, zsw_last_error_string(ctx_));
    }

  private:
    zsw_context* ctx_ = nullptr;
};

// `reads[i].into_local_profile(&matrix, gap_open, gap_extend)` for a whole batch (nucleotides/mod.rs:262-266).
class LocalProfilesBatch {
  public:
    LocalProfilesBatch(GpuContext& ctx, const std::vector<std::string>& reads, const WeightMatrix& matrix, int8_t gap_open,
                       int8_t gap_extend, int preset_bits = 256)
        : ctx_(ctx), preset_(preset_bits) {
        offsets_.reserve(reads.size() + 1);
        offsets_.push_back(0);
        for (auto& r : reads) {
            if (r.empty()) throw ProfileError(1);
            bases_.insert(bases_.end(), r.begin(), r.end());
            offsets_.push_back(bases_.size());
        }
        ctx_.check(zsw_set_scoring(ctx_.raw(), matrix.weights.data(), matrix.S, matrix.mapping->index_map.data(), gap_open, gap_extend));
    }
    // ProfileSets::sw_score_from_i8 (profile_set.rs:71-78)
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i8(const std::string& reference) { return score_from(reference, 8); }
    std::vector<MaybeAligned<uint32_t>> sw_score_from_i16(const std::string& reference) { return score_from(reference, 16); }
    // ProfileSets::sw_align_from_i8 with SeqSrc::Reference (profile_set.rs:124-135)
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8(const std::string& reference, bool seq_is_query = false) {
        return align_from(reference, seq_is_query, zsw_align_batch_from);
    }
    // ProfileSets::sw_align_from_i8_3pass (profile_set.rs:212-235)
    std::vector<MaybeAligned<Alignment>> sw_align_from_i8_3pass(const std::string& reference, bool seq_is_query = false) {
        return align_from(reference, seq_is_query, zsw_align_3pass_batch_from);
    }
    // alignment::sneaky_snake(&reference[ref_start[i]..][..ref_len[i]], read_i, threshold) (sneaky_snake.rs:78-131):
    // true / false per read, Status::Unmapped standing in for `None`.
    std::vector<MaybeAligned<bool>> sneaky_snake(const std::string& reference, const std::vector<uint32_t>& ref_start,
                                                 const std::vector<uint32_t>& ref_len, float threshold) {
        const size_t n = offsets_.size() - 1;
        if (ref_start.size() != n || ref_len.size() != n) throw GpuError(ZSW_ERR_INVALID_ARGUMENT, "one window per read");
        ctx_.check(zsw_set_reference(ctx_.raw(), (const uint8_t*)reference.data(), reference.size(), ZSW_MEM_HOST));
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

  private:
    using AlignFn = zsw_error (*)(zsw_context*, const zsw_batch*, int, int, int, zsw_alignment*, uint8_t*, uint8_t*, uint32_t*,
                                  uint8_t*, uint64_t, uint64_t*, void*);
    std::vector<MaybeAligned<Alignment>> align_from(const std::string& reference, bool seq_is_query, AlignFn fn) {
        const size_t n = offsets_.size() - 1;
        ctx_.check(zsw_set_reference(ctx_.raw(), (const uint8_t*)reference.data(), reference.size(), ZSW_MEM_HOST));
        zsw_batch b = batch();
        std::vector<zsw_alignment> aln(n);
        std::vector<uint8_t> status(n), tier(n);
        std::vector<uint32_t> inc(16 * n + 64);
        std::vector<uint8_t> op(inc.size());
        uint64_t total = 0;
        zsw_error e = fn(ctx_.raw(), &b, 8, preset_, seq_is_query, aln.data(), status.data(), tier.data(), inc.data(), op.data(),
                         inc.size(), &total, nullptr);
        if (e == ZSW_ERR_INVALID_ARGUMENT && total > inc.size()) {
            inc.resize(total);
            op.resize(total);
            e = fn(ctx_.raw(), &b, 8, preset_, seq_is_query, aln.data(), status.data(), tier.data(), inc.data(), op.data(),
                   inc.size(), &total, nullptr);
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
    zsw_batch batch() const {
        zsw_batch b;
        b.bases = bases_.data();
        b.offsets = offsets_.data();
        b.fixed_len = 0;
        b.n_reads = offsets_.size() - 1;
        b.mem = ZSW_MEM_HOST;
        return b;
    }
    std::vector<MaybeAligned<uint32_t>> score_from(const std::string& reference, int width) {
        const size_t n = offsets_.size() - 1;
        ctx_.check(zsw_set_reference(ctx_.raw(), (const uint8_t*)reference.data(), reference.size(), ZSW_MEM_HOST));
        zsw_batch b = batch();
        std::vector<uint32_t> score(n);
        std::vector<uint8_t> status(n), tier(n);
        ctx_.check(zsw_score_batch_from(ctx_.raw(), &b, width, preset_, score.data(), status.data(), tier.data(), nullptr));
        std::vector<MaybeAligned<uint32_t>> out(n);
        for (size_t i = 0; i < n; ++i) {
            out[i].status = (Status)status[i];
            out[i].value = score[i];
        }
        return out;
    }
    GpuContext& ctx_;
    int preset_;
    std::vector<uint8_t> bases_;
    std::vector<uint64_t> offsets_;
};

}  // namespace zoe
