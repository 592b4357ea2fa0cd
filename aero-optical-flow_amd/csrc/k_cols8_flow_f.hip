// K2 k_flow_lane8_cols<false>: the column walk with the reduction in its launch (aof_set_reduce_fusion).
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_cols8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_flow_lane8_cols_f(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, const ColsVotes &cv, uint32_t grid, int threads,
                               void *stream)
{
    hipLaunchKernelGGL((k_flow_lane8_cols<false>), dim3(grid), dim3(threads), 0, static_cast<hipStream_t>(stream), a, plan, rep, cv);
    return (int)hipGetLastError();
}

}  // namespace aof
