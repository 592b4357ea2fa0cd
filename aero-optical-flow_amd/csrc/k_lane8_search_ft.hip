// K2 k_search_lane8<false, true>: exhaustive scan with mean equalisation (C3 level 0 on images that do not prune, level-1 searches of the split coarse passes).
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_search_lane8_ft(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream)
{
    hipLaunchKernelGGL((k_search_lane8<false, true>), dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), a, items, wgs, 1);
    return (int)hipGetLastError();
}

}  // namespace aof
