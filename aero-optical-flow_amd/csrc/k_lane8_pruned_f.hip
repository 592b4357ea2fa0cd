// K2 k_search_lane8_pruned<false>: exact partial-distortion elimination, chunk walk (grids that are not dense).
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_search_lane8_pruned_f(const SearchArgs &a, uint32_t items, uint32_t wgs, int spw, const PruneReport &rep, void *stream)
{
    hipLaunchKernelGGL((k_search_lane8_pruned<false>), dim3(wgs), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a, items, wgs, spw, rep);
    return (int)hipGetLastError();
}

}  // namespace aof
