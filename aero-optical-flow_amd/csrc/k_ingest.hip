// K0 -- frame ingest: centre crop + auto-exposure histogram.
//
// Restates on the device what the reference does on the host immediately before
// calcFlow: cv::Rect centre crop and contiguous copy
// (/root/reference/src/mainloop.cpp:295-298,317-319) and the 10-bin masked cv::calcHist
// over the centred 128x128 region of the cropped image (mainloop.cpp:203-214).
//
// HBM-bound byte mover: only the crop window of each sensor frame is read (16 B per lane,
// rows of the window are contiguous runs of crop_w bytes) and written once -- a plain copy of exactly
// these 128-byte row pieces at a 640-byte stride runs at the device's dense copy rate (5.4-6.0 TB/s,
// tools/ubench_rowcopy.hip, profiles/r03_ubench_rowcopy.txt), so the access pattern is no limit; the
// histogram arithmetic is what the kernel has to hide under it.  The bin of a grey value is
// floor(v*10/255) -- equal to cv::calcHist's double arithmetic for all 256 values
// (tests/test_ingest.py) -- with v = 255 outside the half-open range.  Per pixel that is ONE LDS read:
// a 256-entry table per workgroup gives the pixel's one-hot increment, two words of five 6-bit counters
// (bins 0..4 / 5..9; v = 255 adds nothing; 16-byte entries for 8-bit counters measured 1.4x slower), which
// a lane adds up over three pieces and then widens into 16-bit counters; at the end the wave adds those
// with ONE butterfly of shuffles and lanes 0..9 add a bin each to the workgroup's LDS histogram.  A workgroup that owns its frame's whole crop
// (crops of up to 128 rows) stores the ten totals; taller crops take several workgroups per frame, which
// add to a zeroed histogram with one integer global atomic per bin (order-independent).
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

// PYRAMID (the sequence pipeline, aof_sequence_device): the lane's four pieces are two vertically adjacent PAIRS
// of rows, and what K1 would compute from the cropped frame in a second pass over it comes out of the same
// registers: the frame's 2x2-box level-1 image (one per FRAME: a sequence's level-1 frames form a sequence of their
// own) and its level-0 / level-1 byte sums, added to the pixel-sum records of the two pairs the frame belongs to
// (prev of pair f, cur of pair f - 1; [pair][prev, cur][level]).
struct IngestPyramid {
    uint8_t *l1;          // [n_frames][crop_h / 2][crop_w / 2], or nullptr (sums only)
    uint32_t *sums;       // [n_frames - 1][2][2] zeroed by the launcher, or nullptr
    int64_t n_frames;
};

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 128;   // (64 / 32 rows per workgroup: 80 / 115 us instead of 68 -- the table and the flush do not amortise)
constexpr int kLaneField = 6;      // bits per counter of a lane's table sums: <= 3 pieces x 16 pixels = 48 < 64
constexpr int kLanePieces = 3;     // pieces between two widenings
constexpr int kWavePieces = 63;    // pieces per lane between two wave sums: 63 x 16 x 64 lanes = 64 512 < 65 536

__device__ __forceinline__ int exposure_bin(uint32_t v) { return (int)((v * 10u) / 255u); }  // 10 => dropped

// A lane's ten counters as 16-bit fields, two per word (bins 2k and 2k+1 in word k).
struct LaneCounts { uint32_t w[5]; };

// Adds the two table-sum words (five 6-bit counters each: bins 0..4 / 5..9) to the lane's 16-bit counters.
__device__ __forceinline__ void widen_add(LaneCounts &n, uint32_t lo, uint32_t hi)
{
    constexpr uint32_t m = (1u << kLaneField) - 1;
    n.w[0] += (lo & m) | (((lo >> kLaneField) & m) << 16);
    n.w[1] += ((lo >> (2 * kLaneField)) & m) | (((lo >> (3 * kLaneField)) & m) << 16);
    n.w[2] += ((lo >> (4 * kLaneField)) & m) | ((hi & m) << 16);
    n.w[3] += ((hi >> kLaneField) & m) | (((hi >> (2 * kLaneField)) & m) << 16);
    n.w[4] += ((hi >> (3 * kLaneField)) & m) | (((hi >> (4 * kLaneField)) & m) << 16);
}

// Wave-wide sum of the lanes' counters (plain word adds: no field can overflow, see kWavePieces); every
// lane ends up with the totals, and lanes 0..9 add one bin each to the workgroup histogram.  Must be
// called by every lane of the wave.
__device__ __forceinline__ void flush_counts(uint32_t *s_hist, LaneCounts &n)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) n.w[k] += (uint32_t)__shfl_xor((int)n.w[k], o, 64);
    }
    const int lane = (int)(threadIdx.x & 63);
    if (lane < AOF_EXPOSURE_BINS) {
        uint32_t word = n.w[0];
#pragma unroll
        for (int k = 1; k < 5; k++) word = (lane >> 1) == k ? n.w[k] : word;
        const uint32_t c = (lane & 1) ? word >> 16 : word & 0xFFFFu;
        if (c) atomicAdd(&s_hist[lane], c);
    }
#pragma unroll
    for (int k = 0; k < 5; k++) n.w[k] = 0;
}

template <bool PYRAMID>
__global__ __launch_bounds__(kThreads) void k_ingest(aof_ingest_params p, const uint8_t *camera,
                                                     int64_t camera_stride, uint8_t *cropped,
                                                     int64_t cropped_stride, uint32_t *hist, int nstrips,
                                                     int vec, IngestPyramid pyr)
{
    __shared__ uint32_t s_hist[AOF_EXPOSURE_BINS];
    __shared__ uint2 s_onehot[256];   // grey value -> one-hot increment of a lane's table sums (bins 0..4, bins 5..9)
    const int strip = blockIdx.x % nstrips;
    const int64_t frame = blockIdx.x / nstrips;
    const int tid = threadIdx.x;
    if (tid < AOF_EXPOSURE_BINS) s_hist[tid] = 0;
    if (hist) {   // kThreads == 256: one table entry per lane
        const int b = exposure_bin((uint32_t)tid);   // 10 for v = 255: outside cv::calcHist's range, counts nowhere
        s_onehot[tid] = make_uint2(b < 5 ? 1u << (kLaneField * b) : 0u, (b >= 5 && b < 10) ? 1u << (kLaneField * (b - 5)) : 0u);
    }
    __syncthreads();
    LaneCounts cnt = {{0u, 0u, 0u, 0u, 0u}};
    uint32_t sum_lo = 0, sum_hi = 0;  // table sums since the last widening
    uint32_t pyr_sum0 = 0, pyr_sum1 = 0;   // PYRAMID: the lane's share of the frame's level-0 / level-1 byte sums
    (void)pyr_sum0; (void)pyr_sum1;

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
                int it = base + u * kThreads + tid;
                if (PYRAMID) {
                    // pieces 2q and 2q + 1 of a lane are rows 2r and 2r + 1 of ONE row pair (items counted in row
                    // pairs x pieces x 2: row_begin and the strip height are even)
                    const int pr = base / 2 + (u >> 1) * kThreads + tid;   // (row pair, piece) index
                    it = pr < items / 2 ? (2 * (pr / pieces) + (u & 1)) * pieces + pr % pieces : items;
                }
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
                    if (x >= mx0 && x + 16 <= mx1) {   // the whole piece lies inside the mask: one table read per pixel
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const uint2 e = s_onehot[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
                            sum_lo += e.x; sum_hi += e.y;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            if (x + k < mx0 || x + k >= mx1) continue;
                            const uint2 e = s_onehot[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
                            sum_lo += e.x; sum_hi += e.y;
                        }
                    }
                }
                // (uniform trip counts) the 6-bit table sums hold three pieces, the 16-bit counters of a
                // wave sum 63 pieces per lane
                if (hist && round % kLanePieces == kLanePieces - 1) { widen_add(cnt, sum_lo, sum_hi); sum_lo = sum_hi = 0; }
                if (hist && round % kWavePieces == kWavePieces - 1) flush_counts(s_hist, cnt);
            }
            if constexpr (PYRAMID) {
#pragma unroll
                for (int q = 0; q < kUnroll / 2; q++) {
                    if (!act[2 * q]) continue;   // (a row pair is active as a whole)
                    const uint4 r0 = v[2 * q], r1 = v[2 * q + 1];
                    pyr_sum0 = byte_sum(r0.x, pyr_sum0); pyr_sum0 = byte_sum(r0.y, pyr_sum0);
                    pyr_sum0 = byte_sum(r0.z, pyr_sum0); pyr_sum0 = byte_sum(r0.w, pyr_sum0);
                    pyr_sum0 = byte_sum(r1.x, pyr_sum0); pyr_sum0 = byte_sum(r1.y, pyr_sum0);
                    pyr_sum0 = byte_sum(r1.z, pyr_sum0); pyr_sum0 = byte_sum(r1.w, pyr_sum0);
                    const uint32_t p0 = box2(r0.x, r1.x), p1 = box2(r0.y, r1.y);
                    const uint32_t p2 = box2(r0.z, r1.z), p3 = box2(r0.w, r1.w);
                    uint2 o;
                    o.x = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
                    o.y = __builtin_amdgcn_perm(p3, p2, 0x06040200u);
                    pyr_sum1 = byte_sum(o.x, pyr_sum1);
                    pyr_sum1 = byte_sum(o.y, pyr_sum1);
                    if (pyr.l1)
                        *reinterpret_cast<uint2 *>(pyr.l1 + frame * (int64_t)(p.crop_width / 2) * (p.crop_height / 2) +
                                                   (int64_t)(ys[2 * q] / 2) * (p.crop_width / 2) + xs[2 * q] / 2) = o;
                }
            }
        }
        if (hist) { widen_add(cnt, sum_lo, sum_hi); flush_counts(s_hist, cnt); }
        if constexpr (PYRAMID) {
            if (pyr.sums) {
                pyr_sum0 = wave_sum_u32(pyr_sum0);
                pyr_sum1 = wave_sum_u32(pyr_sum1);
                if ((tid & 63) == 0) {
                    if (frame < pyr.n_frames - 1) { atomicAdd(&pyr.sums[frame * 4 + 0], pyr_sum0); atomicAdd(&pyr.sums[frame * 4 + 1], pyr_sum1); }
                    if (frame > 0) { atomicAdd(&pyr.sums[(frame - 1) * 4 + 2], pyr_sum0); atomicAdd(&pyr.sums[(frame - 1) * 4 + 3], pyr_sum1); }
                }
            }
        }
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
    if (tid < AOF_EXPOSURE_BINS) {
        if (nstrips == 1) hist[frame * AOF_EXPOSURE_BINS + tid] = s_hist[tid];   // the workgroup owns the frame: no zeroing pass, no atomic
        else if (s_hist[tid]) atomicAdd(&hist[frame * AOF_EXPOSURE_BINS + tid], s_hist[tid]);
    }
}

}  // namespace

int launch_ingest(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride,
                  int64_t n_frames, uint8_t *cropped, int64_t cropped_stride, uint32_t *hist, void *stream)
{
    if (n_frames == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nstrips = (p.crop_height + kRowsPerBlock - 1) / kRowsPerBlock;
    if (hist && nstrips > 1) {   // several workgroups add to a frame's histogram
        const int rc = launch_zero_words(hist, n_frames * AOF_EXPOSURE_BINS, stream);
        if (rc) return rc;
    }
    const int vec = (p.crop_width % 16 == 0) && (!cropped || (reinterpret_cast<uintptr_t>(cropped) % 16 == 0 &&
                                                                cropped_stride % 16 == 0));
    hipLaunchKernelGGL(k_ingest<false>, dim3((uint32_t)(n_frames * nstrips)), dim3(kThreads), 0, s, p, camera,
                       camera_stride, cropped, cropped_stride, hist, nstrips, vec, IngestPyramid{nullptr, nullptr, 0});
    return (int)hipGetLastError();
}

// The sequence pipeline's ingest: crop + exposure histogram + the frame's level-1 image and pixel sums in one pass.
bool ingest_pyramid_supported(const aof_ingest_params &p, const uint8_t *cropped, int64_t cropped_stride)
{
    return p.crop_width % 16 == 0 && p.crop_height % 2 == 0 && cropped && reinterpret_cast<uintptr_t>(cropped) % 16 == 0 &&
           cropped_stride % 16 == 0;   // (strips of kRowsPerBlock rows: even)
}

int launch_ingest_pyramid(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride, int64_t n_frames,
                          uint8_t *cropped, int64_t cropped_stride, uint32_t *hist, uint8_t *l1, uint32_t *sums, void *stream)
{
    if (n_frames == 0) return 0;
    if (!ingest_pyramid_supported(p, cropped, cropped_stride)) return (int)hipErrorInvalidValue;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nstrips = (p.crop_height + kRowsPerBlock - 1) / kRowsPerBlock;
    if (hist && nstrips > 1) {
        const int rc = launch_zero_words(hist, n_frames * AOF_EXPOSURE_BINS, stream);
        if (rc) return rc;
    }
    if (sums && n_frames > 1) {
        const int rc = launch_zero_words(sums, (n_frames - 1) * 4, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_ingest<true>, dim3((uint32_t)(n_frames * nstrips)), dim3(kThreads), 0, s, p, camera,
                       camera_stride, cropped, cropped_stride, hist, nstrips, 1, IngestPyramid{l1, n_frames > 1 ? sums : nullptr, n_frames});
    return (int)hipGetLastError();
}

}  // namespace aof
