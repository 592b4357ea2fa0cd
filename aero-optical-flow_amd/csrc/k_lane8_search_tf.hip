// K2 k_search_lane8<true, false>: exhaustive scan with the half-pixel refinement in the lane.
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_search_lane8_tf(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream)
{
    hipLaunchKernelGGL((k_search_lane8<true, false>), dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), a, items, wgs, 1);
    return (int)hipGetLastError();
}

}  // namespace aof
