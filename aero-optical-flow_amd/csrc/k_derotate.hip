// Gyro de-rotation of a batch of flow records (include/aof.h).
// Three float operations per axis with FMA contraction switched off for the block, so
// every operation rounds once and the device result is bit-identical to the host
// arithmetic.  (Plain operators on purpose: __fmul_rn/__fadd_rn are inline functions whose
// bodies keep the translation unit's default contraction and fuse after inlining.)
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

__global__ __launch_bounds__(256) void k_derotate(aof_derotate_params p, const aof_flow *flows,
                                                  const aof_gyro *gyro, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {
#pragma clang fp contract(off)
        const aof_flow f = flows[i];
        const aof_gyro g = gyro[i];
        const float lim = p.rate_threshold * g.dt_s;
        float x = f.flow_x, y = f.flow_y;
        if (fabsf(g.integ_y) > lim) {
            const float pix = g.integ_y * p.focal_x;
            x = f.flow_x + pix;
            x = x < -p.max_flow ? -p.max_flow : (x > p.max_flow ? p.max_flow : x);
        }
        if (fabsf(g.integ_x) > lim) {
            const float pix = g.integ_x * p.focal_y;
            y = f.flow_y - pix;
            y = y < -p.max_flow ? -p.max_flow : (y > p.max_flow ? p.max_flow : y);
        }
        out[2 * i + 0] = x;
        out[2 * i + 1] = y;
    }
}

}  // namespace

int launch_derotate(const aof_derotate_params &p, const aof_flow *flows, const aof_gyro *gyro, int64_t n,
                    float *out, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_derotate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p, flows, gyro, n, out);
    return (int)hipGetLastError();
}

}  // namespace aof
