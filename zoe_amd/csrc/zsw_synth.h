// zsw_synth.h — counter-based synthetic read generator (bench/test utility).
// Read i depends only on (seed, i, reference): every rank regenerates its own shard.
// Shared by the device kernel (zsw_synth.hip) and its host twin (zsw_synth_reads_host).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZSW_HD __host__ __device__ inline
#else
#define ZSW_HD inline
#endif

ZSW_HD uint64_t zsw_sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
ZSW_HD uint64_t zsw_rnd(uint64_t seed, uint64_t i, uint64_t k) {
    return zsw_sm64(zsw_sm64(seed + 0x632BE59BD9B4E019ull * i) + k);
}
ZSW_HD uint8_t zsw_base(unsigned k) {
    const uint32_t acgt = 0x54474341u;  // 'A','C','G','T' little-endian
    return (uint8_t)(acgt >> (8 * (k & 3)));
}
ZSW_HD uint32_t zsw_synth_len(uint64_t seed, uint64_t i, uint32_t min_len, uint32_t max_len) {
    return min_len + (uint32_t)(zsw_rnd(seed, i, 0xFFFF0000ull) % (uint64_t)(max_len - min_len + 1));
}
// Uniform ACGT reference base j (seed 42 in the benchmark).
ZSW_HD uint8_t zsw_synth_ref_base(uint64_t seed, uint64_t j) { return zsw_base((unsigned)(zsw_rnd(seed, j, 0) & 3)); }

// Writes read i (length L) sampled from `ref[0..R)`: 1 % substitutions, 0.1 % insertions, 0.1 % deletions,
// 0.5 % of bases -> 'N', 2 % of reads fully random; forward strand only.
ZSW_HD void zsw_synth_read(uint64_t seed, uint64_t i, const uint8_t* ref, uint32_t R, uint32_t L, uint8_t* out) {
    uint64_t h = zsw_rnd(seed, i, 0);
    bool random_read = (h % 100) < 2 || R == 0;
    uint32_t start = (R > L) ? (uint32_t)((h >> 32) % (uint64_t)(R - L + 1)) : 0;
    uint32_t p = start, o = 0;
    uint64_t j = 0;
    while (o < L) {
        uint64_t u = zsw_rnd(seed, i, 1 + j);
        ++j;
        uint8_t b;
        if (random_read) {
            b = zsw_base((unsigned)(u & 3));
        } else {
            uint32_t ev = (uint32_t)(u % 1000);
            if (ev == 0) {
                b = zsw_base((unsigned)((u >> 10) & 3));  // insertion
            } else if (ev == 1 && j < 4ull * L) {
                ++p;  // deletion
                continue;
            } else {
                b = p < R ? ref[p] : zsw_base((unsigned)((u >> 12) & 3));
                ++p;
                if (((u >> 16) % 100) == 0) {  // substitution
                    unsigned k = b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u;
                    b = zsw_base(k + 1 + (unsigned)((u >> 24) % 3));
                }
            }
        }
        if (((u >> 32) % 200) == 0) b = 'N';
        out[o++] = b;
    }
}
