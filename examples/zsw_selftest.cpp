// zsw_selftest — the reference's own known-answer tests for the striped Smith-Waterman path, run through the C++ mirror
// (include/zoe_sw.hpp) on the GPU. Each check cites the reference test it restates. Exit code 0 = all passed.
//   g++ -O2 -std=c++17 -Iinclude examples/zsw_selftest.cpp -o examples/zsw_selftest -Lzoe_amd -lzoe_sw_hip -Wl,-rpath,$PWD/zoe_amd
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "zoe_sw.hpp"

static int failures = 0;

#define CHECK(cond, what)                                              \
    do {                                                               \
        if (!(cond)) {                                                 \
            ++failures;                                                \
            std::printf("FAIL  %s  (%s:%d)\n", what, __FILE__, __LINE__); \
        } else {                                                       \
            std::printf("ok    %s\n", what);                           \
        }                                                              \
    } while (0)

int main() {
    try {
        zoe::GpuContext ctx(0);
        const std::string ref1 = "ATGCATCGATCGATCGATCGATCGATCGATGC", q1 = "CGTTCGCCATAAAGGGGG";
        const zoe::WeightMatrix m42 = zoe::WeightMatrix::new_dna_matrix(4, -2, 'N');
        {  // src/alignment/sw/striped.rs:45-55 (u8 x 32, biased matrix)
            zoe::StripedProfileBatch p(ctx, {q1}, m42, -3, -1, ZSW_U8, 32);
            auto s = p.sw_score(ref1);
            CHECK(s[0].is_some() && s[0].value == 26, "striped.rs:45 sw_simd_score u8x32 = 26");
        }
        {  // src/alignment/sw/striped.rs:418-441 (u8 x 8 alignment)
            zoe::StripedProfileBatch p(ctx, {q1}, m42, -3, -1, ZSW_U8, 8);
            auto a = p.sw_align(ref1);
            CHECK(a[0].is_some() && a[0].value.score == 26 && a[0].value.cigar() == "6M2D9M3S", "striped.rs:418 sw_simd_align u8x8 = 26, 6M2D9M3S");
        }
        {  // src/alignment/profile_set.rs:293-310 (score + ranges at the i8 tier of w256)
            zoe::StripedProfileBatch p(ctx, {q1}, m42, -3, -1, ZSW_I8, 32);
            auto r = p.sw_score_ranges(ref1);
            CHECK(r[0].is_some() && r[0].value.score == 26 && r[0].value.query_start == 0 && r[0].value.query_end == 15 &&
                      r[0].value.ref_start == 14 && r[0].value.ref_end == 31,
                  "profile_set.rs:293 sw_score_ranges = 26, query 0..15, ref 14..31");
            zoe::LocalProfilesBatch lp(ctx, {q1}, m42, -3, -1, 256);
            auto rf = lp.sw_score_ranges_from_i8(ref1);
            CHECK(rf[0].is_some() && rf[0].value.score == 26 && rf[0].value.query_end == 15 && rf[0].value.ref_start == 14 &&
                      rf[0].value.ref_end == 31 && lp.last_tiers()[0] == 8,
                  "profile_set.rs:293 new_with_w256(..).sw_score_ranges_from_i8 = 26, 0..15, 14..31");
            auto e = p.sw_score_ends(ref1);
            CHECK(e[0].is_some() && e[0].value.ref_end == 31 && e[0].value.query_end == 15, "striped.rs:153 sw_simd_score_ends = (31, 15)");
        }
        {  // the same pair with the roles of sw/mod.rs:63-67: ONE profile from the reference, the read as the other sequence
           // (SharedProfiles, profile_set.rs:552-560); SeqSrc::Query(read) hands the roles back (alignment/mod.rs:176-190)
            zoe::SharedStripedProfile sp(ctx, ref1, m42, -3, -1, ZSW_I8, 32);
            auto s = sp.sw_score({q1});
            CHECK(s[0].is_some() && s[0].value == 26, "shared profile: sw_score = 26");
            auto e = sp.sw_score_ends({q1});
            CHECK(e[0].is_some() && e[0].value.ref_end == 15 && e[0].value.query_end == 31, "shared profile: sw_score_ends = (15 in the read, 31 in the sequence)");
            auto r = sp.sw_score_ranges({q1});
            CHECK(r[0].is_some() && r[0].value.ref_start == 0 && r[0].value.ref_end == 15 && r[0].value.query_start == 14 && r[0].value.query_end == 31,
                  "shared profile: sw_score_ranges = read 0..15, sequence 14..31");
            auto a = sp.sw_align({q1});  // SeqSrc::Query
            CHECK(a[0].is_some() && a[0].value.score == 26 && a[0].value.cigar() == "6M2D9M3S" && a[0].value.ref_start == 14 && a[0].value.ref_end == 31,
                  "shared profile: sw_align(SeqSrc::Query(read)) = 26, 6M2D9M3S, ref 14..31");
            auto ar = sp.sw_align({q1}, false);  // SeqSrc::Reference: the read is the alignment's reference
            CHECK(ar[0].is_some() && ar[0].value.cigar() == "14S6M2I9M1S", "shared profile: sw_align(SeqSrc::Reference(read)) = 14S6M2I9M1S");
            zoe::SharedProfilesBatch sb(ctx, ref1, m42, -3, -1, 256);
            auto ac = sb.sw_align_from_i8({q1});
            CHECK(ac[0].is_some() && ac[0].value.cigar() == "6M2D9M3S" && sb.last_tiers()[0] == 8, "into_shared_profile(..).sw_align_from_i8(SeqSrc::Query(read)) = 6M2D9M3S at i8");
        }
        {  // src/alignment/sw/mod.rs:164-188 and scalar.rs:165-169
            zoe::StripedProfileBatch p(ctx, {"CTCAGATTG"}, m42, -3, -1, ZSW_I8, 32);
            auto a = p.sw_align("GGCCACAGGATTGAG");
            CHECK(a[0].is_some() && a[0].value.score == 27 && a[0].value.cigar() == "5M1D4M" && a[0].value.ref_start == 3,
                  "sw/mod.rs:164 sw_align i8x32 = 27, 5M1D4M, ref_range.start 3");
            zoe::LocalProfilesBatch lp(ctx, {"CTCAGATTG"}, m42, -3, -1);
            auto b = lp.sw_align_from_i8("GGCCACAGGATTGAG");
            CHECK(b[0].is_some() && b[0].value.cigar() == "5M1D4M" && lp.last_tiers()[0] == 8, "sw/mod.rs:224 sw_align_from_i8 answers at the i8 tier");
            auto c = lp.sw_align_from_i8_3pass("GGCCACAGGATTGAG");
            CHECK(c[0].is_some() && c[0].value.score == 27, "profile_set.rs:183 sw_align_from_i8_3pass score 27");
        }
        {  // src/alignment/sw/mod.rs:193-218 (custom alphabet)
            const zoe::ByteIndexMap abcd = zoe::ByteIndexMap::make("ABCD", 'A', false);
            const zoe::WeightMatrix m = zoe::WeightMatrix::make(abcd, 1, -1, -1);
            zoe::StripedProfileBatch p(ctx, {"AABDDAB"}, m, -4, -2, ZSW_I8, 32);
            auto a = p.sw_align("BDAACAABDDDB");
            CHECK(a[0].is_some() && a[0].value.score == 5 && a[0].value.cigar() == "5M2S", "sw/mod.rs:193 custom alphabet = 5, 5M2S");
        }
        const zoe::WeightMatrix m25 = zoe::WeightMatrix::new_dna_matrix(2, -5, 'N');
        {  // src/alignment/sw/test.rs:88-100 (U = T, case-insensitive, N scores 0)
            for (auto t : {ZSW_U16, ZSW_I16}) {
                zoe::StripedProfileBatch p(ctx, {"ACGTUNacgtun"}, m25, -10, -1, t, 16);
                auto s = p.sw_score("ACGTTNACGTTN");
                CHECK(s[0].is_some() && s[0].value == 20, "sw/test.rs:88 ACGTUNacgtun vs ACGTTNACGTTN = 20");
            }
        }
        {  // src/alignment/sw/test.rs:265-271
            const std::string polya(100, 'A');
            zoe::StripedProfileBatch p(ctx, {polya}, m25, -10, -1, ZSW_U16, 16);
            auto s = p.sw_score(polya);
            CHECK(s[0].is_some() && s[0].value == 200, "sw/test.rs:265 poly-A x 100 self = 200");
        }
        {  // src/alignment/sw/test.rs:283-290 (lazy-F regression)
            const zoe::WeightMatrix m = zoe::WeightMatrix::new_dna_matrix(10, -10, 'N');
            zoe::StripedProfileBatch p(ctx, {"AGA"}, m, -5, -5, ZSW_U16, 4);
            auto s = p.sw_score("AA");
            CHECK(s[0].is_some() && s[0].value == 15, "sw/test.rs:283 AGA vs AA = 15");
        }
        {  // src/alignment/sw/test.rs:293-301
            const zoe::WeightMatrix m = zoe::WeightMatrix::new_dna_matrix(127, 0, 'N');
            zoe::StripedProfileBatch p(ctx, {"AAAA"}, m, -10, -1, ZSW_U8, 8);
            auto s = p.sw_score("AAAA");
            CHECK(s[0].status == zoe::Status::Overflowed, "sw/test.rs:293 AAAA vs AAAA at 127 = Overflowed");
        }
        {  // src/alignment/sneaky_snake.rs:55-60
            zoe::LocalProfilesBatch lp(ctx, {"GGTGAGAGTTGT"}, m25, -10, -1);
            auto f = lp.sneaky_snake("GGTGCAGAGCTC", {0}, {12}, 0.25f);
            CHECK(f[0].is_some() && f[0].value, "sneaky_snake.rs:55 doc example = Some(true)");
        }
        {  // src/alignment/profile.rs:32-44 (ProfileError)
            int code = 0;
            try {
                zoe::StripedProfileBatch p(ctx, {"ACGT"}, m25, -1, -10, ZSW_I16, 16);
            } catch (const zoe::ProfileError& e) {
                code = e.code;
            }
            CHECK(code == 4, "profile.rs:32 gap_extend < gap_open = BadGapWeights");
            code = 0;
            try {
                zoe::StripedProfileBatch p(ctx, {""}, m25, -10, -1, ZSW_I16, 16);
            } catch (const zoe::ProfileError& e) {
                code = e.code;
            }
            CHECK(code == 1, "profile.rs:32 empty sequence = EmptySequence");
        }
    } catch (const std::exception& e) {
        std::cerr << "zsw_selftest: " << e.what() << '\n';
        return 2;
    }
    std::printf("%s\n", failures ? "FAILED" : "ALL PASSED");
    return failures ? 1 : 0;
}
