// K2 k_search_lane8_cols<false>: the pruned 8x8 search as a column walk on dense grids -- the C2 / C3 headline kernel (K3 follows).
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_cols8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_search_lane8_cols_f(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, uint32_t wgs, int threads, void *stream)
{
    hipLaunchKernelGGL((k_search_lane8_cols<false>), dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), a, plan, rep);
    return (int)hipGetLastError();
}

}  // namespace aof
