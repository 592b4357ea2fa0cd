// K0 -- frame ingest: centre crop + auto-exposure histogram.
//
// Restates on the device what the reference does on the host immediately before
// calcFlow: cv::Rect centre crop and contiguous copy
// (/root/reference/src/mainloop.cpp:295-298,317-319) and the 10-bin masked cv::calcHist
// over the centred 128x128 region of the cropped image (mainloop.cpp:203-214).
//
// HBM-bound byte mover: only the crop window of each sensor frame is read (16 B per lane,
// rows of the window are contiguous runs of crop_w bytes) and written once.  The bin of
// a grey value is floor(v*10/255) -- equal to cv::calcHist's double arithmetic for all
// 256 values (tests/test_ingest.py) -- with v = 255 outside the half-open range.  A lane
// counts the 16 pixels of its piece in two registers of packed 12-bit fields (5 bins
// each; a whole wave's 1024 pixels still fit a field), the wave adds them with a
// butterfly of shuffles, and one lane per wave adds the ten sums to the workgroup's LDS
// histogram; a single integer global atomic per bin and workgroup publishes it
// (order-independent).
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 128;
constexpr int kField = 12;  // bits per packed counter

__device__ __forceinline__ int exposure_bin(uint32_t v) { return (int)((v * 10u) / 255u); }  // 10 => dropped

// Wave-wide sum of the packed counters (callers flush after at most 3 pieces per lane, so
// a field holds <= 3 x 16 x 64 = 3072 < 2^12), then one lane adds the ten totals to the
// workgroup histogram.  Must be called by every lane of the wave.
__device__ __forceinline__ void flush_counts(uint32_t *s_hist, u64 lo, u64 hi)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo += ((u64)(uint32_t)__shfl_xor((int)(lo >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)lo, o, 64);
        hi += ((u64)(uint32_t)__shfl_xor((int)(hi >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)hi, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const uint32_t c0 = (uint32_t)(lo >> (kField * b)) & ((1u << kField) - 1);
            const uint32_t c1 = (uint32_t)(hi >> (kField * b)) & ((1u << kField) - 1);
            if (c0) atomicAdd(&s_hist[b], c0);
            if (c1) atomicAdd(&s_hist[b + 5], c1);
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_ingest(aof_ingest_params p, const uint8_t *camera,
                                                     int64_t camera_stride, uint8_t *cropped,
                                                     int64_t cropped_stride, uint32_t *hist, int nstrips,
                                                     int vec)
{
    __shared__ uint32_t s_hist[AOF_EXPOSURE_BINS];
    const int strip = blockIdx.x % nstrips;
    const int64_t frame = blockIdx.x / nstrips;
    const int tid = threadIdx.x;
    if (tid < AOF_EXPOSURE_BINS) s_hist[tid] = 0;
    __syncthreads();
    u64 cnt_lo = 0, cnt_hi = 0;  // bins 0..4 / 5..9, kField bits each

    const int x0 = p.camera_width / 2 - p.crop_width / 2, y0 = p.camera_height / 2 - p.crop_height / 2;
    int mx0 = p.crop_width / 2 - AOF_EXPOSURE_MASK_SIZE / 2, my0 = p.crop_height / 2 - AOF_EXPOSURE_MASK_SIZE / 2;
    int mx1 = mx0 + AOF_EXPOSURE_MASK_SIZE, my1 = my0 + AOF_EXPOSURE_MASK_SIZE;
    mx0 = max(mx0, 0); my0 = max(my0, 0); mx1 = min(mx1, p.crop_width); my1 = min(my1, p.crop_height);

    const uint8_t *src = camera + frame * camera_stride + (int64_t)y0 * p.camera_width + x0;
    uint8_t *dst = cropped ? cropped + frame * cropped_stride : nullptr;
    const int row_begin = strip * kRowsPerBlock, row_end = min(p.crop_height, row_begin + kRowsPerBlock);

    if (vec) {  // crop_w % 16 == 0: one 16-byte piece per lane
        const int pieces = p.crop_width / 16, items = (row_end - row_begin) * pieces;
        // four pieces per lane and trip, all four loads in flight before the first is used (the
        // kernel moves 32 KB per frame and lives on memory-level parallelism)
        constexpr int kUnroll = 4;
        int round = 0;
        for (int base = 0; base < items; base += kUnroll * kThreads) {  // uniform trip count
            uint4 v[kUnroll];
            int ys[kUnroll], xs[kUnroll];
            bool act[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const int it = base + u * kThreads + tid;
                act[u] = it < items;
                ys[u] = row_begin + (act[u] ? it / pieces : 0);
                xs[u] = act[u] ? (it % pieces) * 16 : 0;
                v[u] = make_uint4(0, 0, 0, 0);
                if (act[u]) __builtin_memcpy(&v[u], src + (int64_t)ys[u] * p.camera_width + xs[u], 16);  // window start may be unaligned
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++, round++) {
                const int y = ys[u], x = xs[u];
                const bool active = act[u];
                if (active && dst) *reinterpret_cast<uint4 *>(dst + (int64_t)y * p.crop_width + x) = v[u];
                if (active && hist && y >= my0 && y < my1 && x + 16 > mx0 && x < mx1) {
                    const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const bool in = x + k >= mx0 && x + k < mx1;
                        const int b = exposure_bin((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                        // b == 10 (v == 255) is outside cv::calcHist's range: shifted out of both words
                        cnt_lo += (in && b < 5) ? 1ull << (kField * b) : 0ull;
                        cnt_hi += (in && b >= 5 && b < 10) ? 1ull << (kField * (b - 5)) : 0ull;
                    }
                }
                // 3 pieces x 16 pixels x 64 lanes = 3072 < 4096: flush before a field can overflow
                if (hist && round % 3 == 2) { flush_counts(s_hist, cnt_lo, cnt_hi); cnt_lo = cnt_hi = 0; }
            }
        }
        if (hist) flush_counts(s_hist, cnt_lo, cnt_hi);
    } else {
        const int items = (row_end - row_begin) * p.crop_width;
        for (int it = tid; it < items; it += kThreads) {
            const int y = row_begin + it / p.crop_width, x = it % p.crop_width;
            const uint32_t v = src[(int64_t)y * p.camera_width + x];
            if (dst) dst[(int64_t)y * p.crop_width + x] = (uint8_t)v;
            if (hist && y >= my0 && y < my1 && x >= mx0 && x < mx1) {
                const int b = exposure_bin(v);
                if (b < AOF_EXPOSURE_BINS) atomicAdd(&s_hist[b], 1u);
            }
        }
    }
    if (!hist) return;
    __syncthreads();
    if (tid < AOF_EXPOSURE_BINS && s_hist[tid]) atomicAdd(&hist[frame * AOF_EXPOSURE_BINS + tid], s_hist[tid]);
}

}  // namespace

int launch_ingest(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride,
                  int64_t n_frames, uint8_t *cropped, int64_t cropped_stride, uint32_t *hist, void *stream)
{
    if (n_frames == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hist) {
        const int rc = launch_zero_words(hist, n_frames * AOF_EXPOSURE_BINS, stream);
        if (rc) return rc;
    }
    const int nstrips = (p.crop_height + kRowsPerBlock - 1) / kRowsPerBlock;
    const int vec = (p.crop_width % 16 == 0) && (!cropped || (reinterpret_cast<uintptr_t>(cropped) % 16 == 0 &&
                                                                cropped_stride % 16 == 0));
    hipLaunchKernelGGL(k_ingest, dim3((uint32_t)(n_frames * nstrips)), dim3(kThreads), 0, s, p, camera,
                       camera_stride, cropped, cropped_stride, hist, nstrips, vec);
    return (int)hipGetLastError();
}

}  // namespace aof
