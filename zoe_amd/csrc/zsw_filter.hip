// Batched `sneaky_snake` pre-alignment filter (reference: src/alignment/sneaky_snake.rs:78-131).
//
// The reference walks the "chip maze" of 2E+1 diagonals one after the other and keeps, per checkpoint, the longest run of
// matches over all diagonals. Here G lanes of a wavefront share one (reference window, read) pair and each lane owns the
// diagonals l, l+G, ...: the runs are independent, the longest one is a group max, and the early `return Some(true)` is a
// group `any`. The result does not depend on the order in which diagonals are visited (the reference only keeps the max
// and returns true from whichever row gets there), so it is identical to the sequential walk.
#include <hip/hip_runtime.h>

#include "zsw_internal.hpp"

namespace zsw {
namespace {

template <int G>
__global__ __launch_bounds__(256) void sneaky_kernel(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
                                                      uint32_t fixed_len, uint32_t n_pairs, const uint8_t* __restrict__ ref,
                                                      uint32_t R, const uint32_t* __restrict__ ref_start,
                                                      const uint32_t* __restrict__ ref_len, float threshold,
                                                      uint8_t* __restrict__ out) {
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int l = threadIdx.x % G;
    if (gid >= n_pairs) return;  // whole groups leave together
    const uint64_t qoff = offsets ? offsets[gid] : (uint64_t)gid * fixed_len;
    const uint32_t qlen = offsets ? (uint32_t)(offsets[gid + 1] - qoff) : fixed_len;
    const uint32_t rs = ref_start[gid], rl = ref_len[gid];
    if ((uint64_t)rs + rl > R) {
        if (l == 0) out[gid] = ZSW_FILTER_BAD_WINDOW;
        return;
    }
    const uint8_t* q = bases + qoff;
    const uint8_t* r = ref + rs;

    // sneaky_snake.rs:79-91
    if (!(threshold >= 0.f && threshold <= 1.f)) {
        if (l == 0) out[gid] = ZSW_FILTER_NONE;
        return;
    }
    const uint32_t et = (uint32_t)floorf((float)qlen * threshold);
    const uint32_t len_diff = rl > qlen ? rl - qlen : qlen - rl;
    if (len_diff > et) {
        if (l == 0) out[gid] = ZSW_FILTER_NONE;
        return;
    }
    if (et == qlen) {
        if (l == 0) out[gid] = ZSW_FILTER_PASS;
        return;
    }
    // :94-98 the shorter sequence walks the columns
    const uint8_t* s1 = rl > qlen ? q : r;
    const uint8_t* s2 = rl > qlen ? r : q;
    const int n1 = (int)(rl > qlen ? qlen : rl);
    const int n2 = (int)(rl > qlen ? rl : qlen);
    const int window = 2 * (int)et + 1;
    const int diffpad = (int)(len_diff / 2);
    int obstacles = 0, checkpoint = 0;
    bool pass = false;
    while (checkpoint < n1 && obstacles <= (int)et && n1 - checkpoint > (int)et - obstacles) {  // :106
        int last = checkpoint;
        bool hit = false;
        const int finish = n1 - 1 - ((int)et - obstacles);  // a match at col >= finish (or at n1-1) ends the walk (:116)
        for (int row = l; row < window; row += G) {
            const int shift = row + diffpad - (int)et;
            // m = first column >= checkpoint of this diagonal that does not match (or leaves s2); four columns per step while
            // both words are inside the sequences, then byte by byte
            int m = checkpoint;
            const int stop = min(n1, n2 - shift);  // columns < stop have a partner inside s2 (if col + shift >= 0)
            if (m + shift >= 0) {
                const uint8_t* p1 = s1;
                const uint8_t* p2 = s2 + shift;
                while (m + 4 <= stop) {
                    uint32_t x, y;
                    __builtin_memcpy(&x, p1 + m, 4);
                    __builtin_memcpy(&y, p2 + m, 4);
                    const uint32_t d = x ^ y;
                    if (d) {
                        m += __builtin_ctz(d) >> 3;
                        goto scanned;
                    }
                    m += 4;
                }
                while (m < stop && p1[m] == p2[m]) ++m;
            }
        scanned:
            if (m > finish) {  // the match at column `finish` (or the last column) ends the walk (:116)
                hit = true;
                break;
            }
            last = max(last, m);
        }
        int h = hit ? 1 : 0;
#pragma unroll
        for (int d = G / 2; d >= 1; d >>= 1) {
            h |= __shfl_xor(h, d, G);
            last = max(last, __shfl_xor(last, d, G));
        }
        if (h) {
            pass = true;
            break;
        }
        checkpoint = last + 1;  // :126-127
        ++obstacles;
    }
    if (l == 0) out[gid] = (pass || obstacles <= (int)et) ? ZSW_FILTER_PASS : ZSW_FILTER_REJECT;  // :117, :130
}

}  // namespace

hipError_t launch_sneaky(const BatchDev& b, const uint8_t* d_ref, uint32_t R, const uint32_t* d_ref_start,
                         const uint32_t* d_ref_len, float threshold, uint8_t* d_out, hipStream_t stream) {
    if (b.n_reads == 0) return hipSuccess;
    constexpr int G = 16;
    const uint64_t threads = (uint64_t)b.n_reads * G;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(sneaky_kernel<G>, dim3(blocks), dim3(256), 0, stream, b.bases, b.offsets, b.fixed_len, b.n_reads, d_ref, R,
                       d_ref_start, d_ref_len, threshold, d_out);
    return hipGetLastError();
}

}  // namespace zsw
