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
// 256 values (tests/test_ingest.py) -- with v = 255 outside the half-open range.  Each
// wave votes into its own 10-bin LDS histogram, merged once per workgroup with integer
// global atomics (order-independent).
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 16;

__device__ __forceinline__ int exposure_bin(uint32_t v) { return (int)((v * 10u) / 255u); }  // 10 => dropped

__global__ __launch_bounds__(kThreads) void k_ingest(aof_ingest_params p, const uint8_t *camera,
                                                     int64_t camera_stride, uint8_t *cropped,
                                                     int64_t cropped_stride, uint32_t *hist, int nstrips,
                                                     int vec)
{
    __shared__ uint32_t s_hist[kThreads / 64][AOF_EXPOSURE_BINS + 1];
    const int strip = blockIdx.x % nstrips;
    const int64_t frame = blockIdx.x / nstrips;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid < (kThreads / 64) * (AOF_EXPOSURE_BINS + 1)) (&s_hist[0][0])[tid] = 0;
    __syncthreads();

    const int x0 = p.camera_width / 2 - p.crop_width / 2, y0 = p.camera_height / 2 - p.crop_height / 2;
    int mx0 = p.crop_width / 2 - AOF_EXPOSURE_MASK_SIZE / 2, my0 = p.crop_height / 2 - AOF_EXPOSURE_MASK_SIZE / 2;
    int mx1 = mx0 + AOF_EXPOSURE_MASK_SIZE, my1 = my0 + AOF_EXPOSURE_MASK_SIZE;
    mx0 = max(mx0, 0); my0 = max(my0, 0); mx1 = min(mx1, p.crop_width); my1 = min(my1, p.crop_height);

    const uint8_t *src = camera + frame * camera_stride + (int64_t)y0 * p.camera_width + x0;
    uint8_t *dst = cropped ? cropped + frame * cropped_stride : nullptr;
    const int row_begin = strip * kRowsPerBlock, row_end = min(p.crop_height, row_begin + kRowsPerBlock);

    if (vec) {  // crop_w % 16 == 0: one 16-byte piece per lane
        const int pieces = p.crop_width / 16, items = (row_end - row_begin) * pieces;
        for (int it = tid; it < items; it += kThreads) {
            const int y = row_begin + it / pieces, x = (it % pieces) * 16;
            uint4 v;
            __builtin_memcpy(&v, src + (int64_t)y * p.camera_width + x, 16);  // window start may be unaligned
            if (dst) *reinterpret_cast<uint4 *>(dst + (int64_t)y * p.crop_width + x) = v;
            if (hist && y >= my0 && y < my1 && x + 16 > mx0 && x < mx1) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const bool in = x + k >= mx0 && x + k < mx1;
                    const int b = exposure_bin((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                    if (in) atomicAdd(&s_hist[wave][b], 1u);  // bin 10 = dropped (v == 255)
                }
            }
        }
    } else {
        const int items = (row_end - row_begin) * p.crop_width;
        for (int it = tid; it < items; it += kThreads) {
            const int y = row_begin + it / p.crop_width, x = it % p.crop_width;
            const uint32_t v = src[(int64_t)y * p.camera_width + x];
            if (dst) dst[(int64_t)y * p.crop_width + x] = (uint8_t)v;
            if (hist && y >= my0 && y < my1 && x >= mx0 && x < mx1) atomicAdd(&s_hist[wave][exposure_bin(v)], 1u);
        }
    }
    if (!hist) return;
    __syncthreads();
    if (tid < AOF_EXPOSURE_BINS) {
        uint32_t s = 0;
        for (int w = 0; w < kThreads / 64; w++) s += s_hist[w][tid];
        if (s) atomicAdd(&hist[frame * AOF_EXPOSURE_BINS + tid], s);
    }
}

}  // namespace

int launch_ingest(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride,
                  int64_t n_frames, uint8_t *cropped, int64_t cropped_stride, uint32_t *hist, void *stream)
{
    if (n_frames == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hist) {
        hipError_t e = hipMemsetAsync(hist, 0, (size_t)n_frames * AOF_EXPOSURE_BINS * sizeof(uint32_t), s);
        if (e != hipSuccess) return (int)e;
    }
    const int nstrips = (p.crop_height + kRowsPerBlock - 1) / kRowsPerBlock;
    const int vec = (p.crop_width % 16 == 0) && (!cropped || (reinterpret_cast<uintptr_t>(cropped) % 16 == 0 &&
                                                                cropped_stride % 16 == 0));
    hipLaunchKernelGGL(k_ingest, dim3((uint32_t)(n_frames * nstrips)), dim3(kThreads), 0, s, p, camera,
                       camera_stride, cropped, cropped_stride, hist, nstrips, vec);
    return (int)hipGetLastError();
}

}  // namespace aof
