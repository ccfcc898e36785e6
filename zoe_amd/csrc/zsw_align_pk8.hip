// zsw_align_pk8.hip — align_kernel_pk<8, 1..16> (zsw_align_pk_kernel.hpp): Zoe vectors of 8 lanes.
#include <type_traits>

#include "zsw_align_pk_kernel.hpp"

namespace zsw {
hipError_t align_pk_occupancy_8(uint32_t nv, size_t lds, int* blocks_per_cu) { return pk_occupancy_n<8>(nv, lds, blocks_per_cu); }
hipError_t align_pk_launch_8(const AlignArgs& a, uint32_t grid, size_t lds, hipStream_t stream) { return pk_launch_n<8>(a, grid, lds, stream); }
}  // namespace zsw
