// K2 k_search_lane8<false, false>: the C2 headline's exhaustive scan: no half-pixel step, no equalisation code.
// ONE kernel per translation unit: aof_lane8_kernels.hpp says why.
#include "aof_lane8_kernels.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

int launch_k_search_lane8_ff(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream)
{
    hipLaunchKernelGGL((k_search_lane8<false, false>), dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), a, items, wgs, 1);
    return (int)hipGetLastError();
}

}  // namespace aof
