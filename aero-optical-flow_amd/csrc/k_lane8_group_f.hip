// K2 k_flow_lane8<false>: grids of 8..256 blocks, a workgroup owns whole pairs and finalises their flow records.
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_flow_lane8_f(const SearchArgs &a, const FlowTail &t, int ppw, uint32_t wgs, size_t lds, void *stream)
{
    hipLaunchKernelGGL((k_flow_lane8<false>), dim3(wgs), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a, t, ppw);
    return (int)hipGetLastError();
}

}  // namespace aof
