// K2 k_flow_lane8_flat<false, true>: the same scan with the reduction in its launch (aof_set_reduce_fusion).
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_flow_lane8_flat_ft(const SearchArgs &a, uint32_t items, uint32_t search_wgs, uint32_t grid, int threads, const FlowTail &t,
                                const VoteMem &v, void *stream)
{
    hipLaunchKernelGGL((k_flow_lane8_flat<false, true>), dim3(grid), dim3(threads), 0, static_cast<hipStream_t>(stream), a, items, search_wgs, t, v);
    return (int)hipGetLastError();
}

}  // namespace aof
