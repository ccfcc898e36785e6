// zsw_bench — the C ABI timed from C++ with host-resident reads and results (no Python in the loop): n synthetic 150 bp reads
// against a 2 kb reference through sw_score_from_i8, sw_score_ranges_from_i8, sw_align_from_i8 and sw_align_from_i8_3pass.
//   g++ -O2 -std=c++17 -Iinclude examples/zsw_bench.cpp -o examples/zsw_bench -Lzoe_amd -lzoe_sw_hip -Wl,-rpath,$PWD/zoe_amd
//   ./examples/zsw_bench [n_reads]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "zoe_sw.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
    const uint32_t L = 150, R = 2000;
    try {
        zoe::GpuContext ctx(0);
        std::vector<uint8_t> ref(R), bases(n * L);
        zsw_synth_reference_host(42, R, ref.data());
        zsw_synth_reads_host(1337, 0, n, L, ref.data(), R, bases.data());
        const zoe::WeightMatrix m = zoe::WeightMatrix::new_dna_matrix(2, -5, 'N');
        ctx.check(zsw_set_scoring(ctx.raw(), m.weights.data(), m.S, m.mapping->index_map.data(), -10, -1));
        ctx.check(zsw_set_reference(ctx.raw(), ref.data(), R, ZSW_MEM_HOST));
        zsw_batch b;
        b.bases = bases.data();
        b.offsets = nullptr;
        b.fixed_len = L;
        b.n_reads = n;
        b.mem = ZSW_MEM_HOST;
        b.encoding = ZSW_ENCODING_BYTES;
        std::vector<uint32_t> score(n), rs(n), re(n), qs(n), qe(n), inc(8 * n + 64);
        std::vector<uint8_t> status(n), tier(n), op(inc.size());
        std::vector<zsw_alignment> aln(n);
        uint64_t total = 0;
        auto timed = [&](const char* name, auto&& call) {
            call();  // warm-up (first-touch allocations)
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now();
                ctx.check(call());
                best = std::min(best, now() - t0);
            }
            std::printf("%-28s %8.1f ms  %7.2f M reads/s  (host in -> host out)\n", name, best * 1e3, n / best / 1e6);
        };
        timed("sw_score_from_i8", [&] { return zsw_score_batch_from(ctx.raw(), &b, 8, 256, score.data(), status.data(), tier.data(), nullptr); });
        timed("sw_score_ranges_from_i8", [&] {
            return zsw_score_ranges_batch_from(ctx.raw(), &b, 8, 256, score.data(), rs.data(), re.data(), qs.data(), qe.data(), status.data(),
                                               tier.data(), nullptr);
        });
        timed("sw_align_from_i8_3pass", [&] {
            return zsw_align_3pass_batch_from(ctx.raw(), &b, 8, 256, 0, aln.data(), status.data(), tier.data(), inc.data(), op.data(),
                                              inc.size(), &total, nullptr);
        });
        timed("sw_align_from_i8", [&] {
            return zsw_align_batch_from(ctx.raw(), &b, 8, 256, 0, aln.data(), status.data(), tier.data(), inc.data(), op.data(), inc.size(),
                                        &total, nullptr);
        });
        std::printf("%llu reads, %llu ciglets in the last call\n", (unsigned long long)n, (unsigned long long)total);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "zsw_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
