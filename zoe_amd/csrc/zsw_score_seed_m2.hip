// zsw_score_seed_m2.hip — seed_window_kernel<G, C, 2> (zsw_score_seed_kernel.hpp) for every strip configuration.
#include "zsw_score_seed_kernel.hpp"

namespace zsw {

hipError_t launch_seed_window_m2(const SeedWindowArgs& a, int G, int C, hipStream_t stream) {
    const uint32_t groups = (a.n + 1) / 2;
    switch (G * 100 + C) {
#define ZSW_CASE(GV, CV)                                                                                                         \
    case GV * 100 + CV:                                                                                                          \
        hipLaunchKernelGGL((seed_window_kernel<GV, CV, 2>), dim3((groups + BLOCK / GV - 1) / (BLOCK / GV)), dim3(BLOCK), 0, stream, a); \
        break;
        ZSW_FOR_EACH_SEED_CONFIG(ZSW_CASE)
#undef ZSW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace zsw
