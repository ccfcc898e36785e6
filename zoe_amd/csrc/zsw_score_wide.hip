// zsw_score_wide.hip — score_kernel_v2 for alphabets of 8..32 letters (amino acids with the BLOSUM matrices of
// src/data/matrices/aa.rs, S = 25): the same packed drift-domain recurrence as zsw_score.hip, the substitution score of a
// cell read from an LDS copy of the weight matrix (two signed-byte loads per packed cell pair) instead of one v_perm_b32.
#include "zsw_score_v1.hpp"
#include "zsw_score_v2.hpp"

namespace zsw {

template <int G, int C>
static hipError_t launch_cfg_wide(const ScoreArgsV2& a, int mode, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
    // MODE 1 (score + ref_end) is served by MODE 2: the query end goes nowhere when out.query_end is null
    if (mode == 0) hipLaunchKernelGGL((score_kernel_v2<G, C, 0, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((score_kernel_v2<G, C, 2, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_table_cfg_v2_wide(const ScoreArgsV2& a, int G, int C, int mode, hipStream_t stream) {
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: return launch_cfg_wide<GV, CV>(a, mode, stream);
        ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CASE)
#undef ZSW_CASE
    }
    return hipErrorInvalidValue;
}

// One tile of the reads that are longer than the widest strip configuration (both table forms live here to keep this
// translation unit and zsw_score.hip of similar size).
hipError_t launch_tile_v2(const ScoreArgsV2& a, bool wide, int mode, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / TILE_G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
    if (wide) {
        if (mode == 0) hipLaunchKernelGGL((score_kernel_v2<TILE_G, TILE_C, 0, true, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((score_kernel_v2<TILE_G, TILE_C, 2, true, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((score_kernel_v2<TILE_G, TILE_C, 0, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((score_kernel_v2<TILE_G, TILE_C, 2, false, true>), dim3(grid), dim3(BLOCK), 0, stream, a);
    }
    return hipGetLastError();
}

// reverse pass of sw_simd_score_ranges for these alphabets (score_kernel<..., REV, WIDE>)
hipError_t launch_cfg_rev_wide(const ScoreArgs& a, int G, int C, hipStream_t stream) {
    const uint32_t reads_per_block = 2 * (BLOCK / G);
    const uint32_t grid = (a.b.n_items + reads_per_block - 1) / reads_per_block;
    if (grid == 0) return hipSuccess;
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV) \
    case GV * 100 + CV: hipLaunchKernelGGL((score_kernel<GV, CV, true, 2, true, true>), dim3(grid), dim3(BLOCK), 0, stream, a); break;
        ZSW_FOR_EACH_STRIP_CONFIG(ZSW_CASE)
#undef ZSW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace zsw
