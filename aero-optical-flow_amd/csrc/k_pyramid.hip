// K1 -- frame sums and 2x2 box pyramid (DESIGN.md "Spec": Mean, Pyramid).
//
// One pass over every level-0 pixel: each lane loads 16 B from two adjacent
// rows (coalesced 1 KiB per wave-instruction), emits 8 level-1 pixels
// (a+b+c+d+2)>>2 and accumulates the level-0 / level-1 byte sums with
// v_sad_u8 against zero; a wave reduction and one integer atomic per wave
// publish the sums (integer atomics: order-independent, bit-exact).
// HBM-bound: reads W*H, writes W*H/4 per frame.
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;
constexpr int kItemsPerBlock = 1024;  // 16-px x 2-row items per workgroup

// One wave's share of a frame's byte sums into the pairs' records [pair][prev, cur][level 0, level 1].  In a
// sequence frame f is prev of pair f and cur of pair f - 1: both records get it (a.n_pairs counts FRAMES then).
__device__ __forceinline__ void publish_sums(const PyramidArgs &a, int64_t unit, int frame, uint32_t sum0, uint32_t sum1)
{
    if (!a.sequence) {
        atomicAdd(&a.sums[unit * 4 + frame * 2 + 0], sum0);
        atomicAdd(&a.sums[unit * 4 + frame * 2 + 1], sum1);
        return;
    }
    if (unit < a.n_pairs - 1) {
        atomicAdd(&a.sums[unit * 4 + 0], sum0);
        atomicAdd(&a.sums[unit * 4 + 1], sum1);
    }
    if (unit > 0) {
        atomicAdd(&a.sums[(unit - 1) * 4 + 2], sum0);
        atomicAdd(&a.sums[(unit - 1) * 4 + 3], sum1);
    }
}

__global__ __launch_bounds__(kThreads) void k_pyramid_vec(PyramidArgs a, int rows_per_strip,
                                                          int nstrips)
{
    const int w1 = a.w / 2, h1 = a.h / 2;
    const int chunks = a.w / 16;
    uint32_t id = blockIdx.x;
    const int strip = (int)(id % (uint32_t)nstrips);
    id /= (uint32_t)nstrips;
    // a frame SEQUENCE (a.sequence): unit `id` is FRAME id -- prev of pair id and cur of pair id - 1 --, summed
    // and filtered once; otherwise frame (id & 1) of pair (id >> 1)
    const int frame = a.sequence ? 0 : (int)(id & 1u);
    const int64_t pair = a.sequence ? (int64_t)id : (int64_t)(id >> 1);
    const uint8_t *src = (frame ? a.cur : a.prev) + pair * a.pair_stride;
    uint8_t *dst = nullptr;
    if (a.l1_prev) dst = (frame ? a.l1_cur : a.l1_prev) + pair * (int64_t)w1 * h1;

    const int y_begin = strip * rows_per_strip;
    const int y_end = min(h1, y_begin + rows_per_strip);
    const int items = (y_end - y_begin) * chunks;
    uint32_t sum0 = 0, sum1 = 0;
    for (int it = threadIdx.x; it < items; it += kThreads) {
        const int y1 = y_begin + it / chunks, c = it % chunks;
        // streamed once: non-temporal loads keep the frames from pushing the level-1 frames this
        // kernel writes (and the next kernel reads) out of L2 / the memory-side cache (K1 -10 %)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (int64_t)(2 * y1) * a.w + c * 16));
        const u32x4 q1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + (int64_t)(2 * y1 + 1) * a.w + c * 16));
        const uint4 r0 = make_uint4(q0.x, q0.y, q0.z, q0.w), r1 = make_uint4(q1.x, q1.y, q1.z, q1.w);
        sum0 = byte_sum(r0.x, sum0); sum0 = byte_sum(r0.y, sum0);
        sum0 = byte_sum(r0.z, sum0); sum0 = byte_sum(r0.w, sum0);
        sum0 = byte_sum(r1.x, sum0); sum0 = byte_sum(r1.y, sum0);
        sum0 = byte_sum(r1.z, sum0); sum0 = byte_sum(r1.w, sum0);
        const uint32_t p0 = box2(r0.x, r1.x), p1 = box2(r0.y, r1.y);
        const uint32_t p2 = box2(r0.z, r1.z), p3 = box2(r0.w, r1.w);
        uint2 o;
        o.x = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
        o.y = __builtin_amdgcn_perm(p3, p2, 0x06040200u);
        sum1 = byte_sum(o.x, sum1);
        sum1 = byte_sum(o.y, sum1);
        if (dst) *reinterpret_cast<uint2 *>(dst + (int64_t)y1 * w1 + c * 8) = o;
    }
    if (a.sums) {
        sum0 = wave_sum_u32(sum0);
        sum1 = wave_sum_u32(sum1);
        if ((threadIdx.x & 63) == 0) publish_sums(a, pair, frame, sum0, sum1);
    }
}

// Any width/height/alignment: one 2x2 cell per lane, byte loads.
__global__ __launch_bounds__(kThreads) void k_pyramid_scalar(PyramidArgs a, int nstrips)
{
    const int w1c = (a.w + 1) / 2, h1c = (a.h + 1) / 2;
    const int w1 = a.w / 2, h1 = a.h / 2;
    uint32_t id = blockIdx.x;
    const int strip = (int)(id % (uint32_t)nstrips);
    id /= (uint32_t)nstrips;
    const int frame = a.sequence ? 0 : (int)(id & 1u);
    const int64_t pair = a.sequence ? (int64_t)id : (int64_t)(id >> 1);
    const uint8_t *src = (frame ? a.cur : a.prev) + pair * a.pair_stride;
    uint8_t *dst = nullptr;
    if (a.l1_prev) dst = (frame ? a.l1_cur : a.l1_prev) + pair * (int64_t)w1 * h1;
    const int64_t cells = (int64_t)w1c * h1c;
    uint32_t sum0 = 0, sum1 = 0;
    for (int64_t cell = (int64_t)strip * kThreads + threadIdx.x; cell < cells;
         cell += (int64_t)nstrips * kThreads) {
        const int y = (int)(cell / w1c), x = (int)(cell % w1c);
        const int x0 = 2 * x, y0 = 2 * y;
        const bool xin = x0 + 1 < a.w, yin = y0 + 1 < a.h;
        const uint32_t p00 = src[(int64_t)y0 * a.w + x0];
        const uint32_t p01 = xin ? src[(int64_t)y0 * a.w + x0 + 1] : 0;
        const uint32_t p10 = yin ? src[(int64_t)(y0 + 1) * a.w + x0] : 0;
        const uint32_t p11 = (xin && yin) ? src[(int64_t)(y0 + 1) * a.w + x0 + 1] : 0;
        sum0 += p00 + p01 + p10 + p11;
        if (xin && yin) {
            const uint32_t v = (p00 + p01 + p10 + p11 + 2) >> 2;
            sum1 += v;
            if (dst) dst[(int64_t)y * w1 + x] = (uint8_t)v;
        }
    }
    if (a.sums) {
        sum0 = wave_sum_u32(sum0);
        sum1 = wave_sum_u32(sum1);
        if ((threadIdx.x & 63) == 0) publish_sums(a, pair, frame, sum0, sum1);
    }
}

// Accumulators that integer atomics add to start from zero.  Zeroed by a kernel of our own, not by
// hipMemsetAsync: captured into a hipGraph, the memset node of ROCm 7.2 left garbage in the 16 bytes of
// pixel sums per pair when the graph was replayed (tests/test_gpu_parity.py::test_two_level_batch_
// replays_from_a_graph found it) -- a kernel node replays like every other kernel here.
__global__ __launch_bounds__(kThreads) void k_zero_words(uint32_t *words, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < count) words[i] = 0;
}

}  // namespace

int launch_zero_words(uint32_t *words, int64_t count, void *stream)
{
    if (count <= 0) return 0;
    const int64_t wgs = (count + kThreads - 1) / kThreads;
    if (wgs > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_zero_words, dim3((uint32_t)wgs), dim3(kThreads), 0, static_cast<hipStream_t>(stream), words, count);
    return (int)hipGetLastError();
}

int launch_pyramid(const PyramidArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (a.sequence: n_pairs counts the FRAMES of the sequence, one unit each; n_pairs - 1 pairs of sums)
    const int64_t units = a.sequence ? a.n_pairs : a.n_pairs * 2;
    if (a.sums) {
        const int rc = launch_zero_words(a.sums, (a.sequence ? a.n_pairs - 1 : a.n_pairs) * 4, stream);
        if (rc) return rc;
    }
    const bool vec = (a.w % 16 == 0) && (a.h % 2 == 0) && (a.pair_stride % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(a.prev) % 16 == 0) &&
                     (a.sequence || reinterpret_cast<uintptr_t>(a.cur) % 16 == 0);
    if (vec) {
        const int chunks = a.w / 16, h1 = a.h / 2;
        int rows = kItemsPerBlock / chunks;
        if (rows < 1) rows = 1;
        const int nstrips = (h1 + rows - 1) / rows;
        const int64_t total = units * nstrips;
        if (total > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_pyramid_vec, dim3((uint32_t)total), dim3(kThreads), 0, s, a, rows,
                           nstrips);
    } else {
        const int64_t cells = (int64_t)((a.w + 1) / 2) * ((a.h + 1) / 2);
        int nstrips = (int)((cells + kItemsPerBlock - 1) / kItemsPerBlock);
        if (nstrips < 1) nstrips = 1;
        const int64_t total = units * nstrips;
        if (total > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_pyramid_scalar, dim3((uint32_t)total), dim3(kThreads), 0, s, a,
                           nstrips);
    }
    return (int)hipGetLastError();
}

}  // namespace aof
