// zsw_group.hip — device-side grouping of the reads of an alignment call.
//
// Pass 2 of the alignment path runs one launch per <N lanes, nv vectors> instantiation and wants the reads of a launch
// ordered by their reference end row and, among equal end rows, by score: the reads of a wavefront walk the reference rows
// together, so their last rows should coincide, and the rows of flags a read keeps follow from its score
// (flag_rows_needed, zsw_align_dev.hpp), so reads of similar score keep the same rows. Two stable radix sorts (hipCUB — a
// utility step, not the hot path): by descending score, then by the composite key (N, packed/wide, nv, certificate, ref_end), followed
// by a scan for the group boundaries. Only the small group table travels to the host.
#include <hipcub/hipcub.hpp>

#include "zsw_align.hpp"

namespace zsw {

__global__ void score_keys_kernel(uint32_t n, const uint8_t* status, const uint32_t* score, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = status[i] == ZSW_STATUS_SOME ? (uint64_t)(0xffffffffu - score[i]) : ~0ull;  // descending score
    vals[i] = i;
}

// keys[j] for the read at position j of the score order (ids[j]); vals are the ids themselves
__global__ void group_keys_kernel(BatchDev b, const uint32_t* ids, const uint8_t* status, const uint8_t* tier, const uint32_t* ref_end,
                                  const uint32_t* score, const uint32_t* safe_row, int lanes_w8, int lanes_w16, int lanes_w32, uint64_t* keys) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.n_reads) return;
    const uint32_t i = ids[j];
    uint64_t key = ~0ull;  // reads without an alignment sort to the end
    if (status[i] == ZSW_STATUS_SOME) {
        const uint32_t len = b.offsets ? (uint32_t)(b.offsets[i + 1] - b.offsets[i]) : b.fixed_len;
        const uint32_t N = tier[i] == 8 ? lanes_w8 : tier[i] == 16 ? lanes_w16 : lanes_w32;
        const uint32_t nv = (len + N - 1) / N;
        // bit 55: the score does not fit the packed kernel's 16-bit lanes (its reads form a group of their own)
        const uint64_t wide = score[i] > ALIGN_PK_MAX_SCORE ? 1u : 0u;
        // bit 31: no late-start certificate from the first pass (these reads need warmup_rows more rows: behind the others of the
        // group, so that they share wavefronts with each other and not with reads that start a few rows above their alignment)
        const uint64_t slow = (safe_row && safe_row[i] == 0xffffffffu) ? 1u : 0u;
        key = ((uint64_t)N << 56) | (wide << 55) | ((uint64_t)(nv & 0x7fffffu) << 32) | (slow << 31) | (ref_end[i] & 0x7fffffffu);
    }
    keys[j] = key;
}

__global__ void group_bounds_kernel(const uint64_t* sorted_keys, uint32_t n, uint32_t* table, uint32_t* table_count, uint32_t cap) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t k = sorted_keys[j];
    if (k == ~0ull) return;
    const uint32_t g = (uint32_t)(k >> 32);
    if (j == 0 || (uint32_t)(sorted_keys[j - 1] >> 32) != g) {
        const uint32_t slot = atomicAdd(table_count, 1u);
        if (slot < cap) {
            table[2 * slot] = g;      // N << 24 | wide << 23 | nv
            table[2 * slot + 1] = j;  // first position of the group in the sorted order
        }
    }
}

size_t group_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n, 0, 64, (hipStream_t)0);
    return bytes;
}

// keys_in/keys_out: n u64 each; vals_in: n u32; items_out: n u32 (sorted read ids); table: 2*cap u32; table_count: 1 u32.
hipError_t group_reads(const BatchDev& b, const uint8_t* d_status, const uint8_t* d_tier, const uint32_t* d_ref_end, const uint32_t* d_score,
                       const uint32_t* d_safe_row, int lanes_w8, int lanes_w16, int lanes_w32, uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* items_out,
                       void* temp, size_t temp_bytes, uint32_t* table, uint32_t* table_count, uint32_t cap, hipStream_t stream) {
    const uint32_t n = b.n_reads;
    if (n == 0) return hipSuccess;
    const uint32_t grid = (n + 255) / 256;
    // 1. by descending score (32 key bits): items_out = read ids in that order
    hipLaunchKernelGGL(score_keys_kernel, dim3(grid), dim3(256), 0, stream, n, d_status, d_score, keys_in, vals_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, items_out, (int)n, 0, 33, stream);
    if (e != hipSuccess) return e;
    // 2. stable sort of that order by (N, packed/wide, nv, ref_end): vals_in = final order, copied back to items_out
    hipLaunchKernelGGL(group_keys_kernel, dim3(grid), dim3(256), 0, stream, b, items_out, d_status, d_tier, d_ref_end, d_score, d_safe_row, lanes_w8,
                       lanes_w16, lanes_w32, keys_in);
    e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, items_out, vals_in, (int)n, 0, 64, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(items_out, vals_in, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(table_count, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(group_bounds_kernel, dim3(grid), dim3(256), 0, stream, keys_out, n, table, table_count, cap);
    return hipGetLastError();
}

}  // namespace zsw
