// ubench_rowcopy -- what the access pattern of k_ingest can reach (not part of the product).
// k_ingest reads the 128x128 centre crop of 640x480 sensor frames: 128-byte row pieces at a 640-byte
// stride, and writes them contiguously.  This copies exactly those bytes with nothing else (no
// histogram), for 1, 2, 4 and 8 sixteen-byte pieces per lane in flight, plain and non-temporal loads,
// one workgroup per frame (256 lanes) or per half / quarter frame, next to a dense copy of the same
// number of bytes (the device's plain copy rate at this launch size).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rowcopy.hip -o tools/ubench_rowcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int CAM_W = 640, CAM_H = 480, CROP = 128, PIECES = CROP / 16;

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_crop(const uint8_t *cam, uint8_t *out, int rows_per_wg)
{
    const int wgs_per_frame = CROP / rows_per_wg;
    const int64_t frame = blockIdx.x / wgs_per_frame;
    const int row0 = (blockIdx.x % wgs_per_frame) * rows_per_wg;
    const uint8_t *src = cam + frame * (int64_t)(CAM_W * CAM_H) + (int64_t)(CAM_H / 2 - CROP / 2) * CAM_W + (CAM_W / 2 - CROP / 2);
    uint8_t *dst = out + frame * (int64_t)(CROP * CROP);
    const int items = rows_per_wg * PIECES;
    for (int base = 0; base < items; base += U * 256) {
        u32x4 v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int it = base + u * 256 + (int)threadIdx.x;
            const int y = row0 + it / PIECES, x = (it % PIECES) * 16;
            off[u] = it < items ? y * CROP + x : -1;
            if (it < items) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(src + (int64_t)y * CAM_W + x);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (off[u] >= 0) *reinterpret_cast<u32x4 *>(dst + off[u]) = v[u];
    }
}

__global__ __launch_bounds__(256) void k_dense(const uint8_t *in, uint8_t *out, int64_t bytes)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 64;   // four 16-byte pieces per lane in flight
    if (i + 64 > bytes) return;
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const u32x4 *>(in + i + 16 * u);
#pragma unroll
    for (int u = 0; u < 4; u++) *reinterpret_cast<u32x4 *>(out + i + 16 * u) = v[u];
}

template <typename F>
static double time_us(F &&launch, int reps = 30)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 300; i++) launch();   // clocks settle
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e3;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    uint8_t *cam, *out, *dense;
    CHECK(hipMalloc(&cam, (size_t)n * CAM_W * CAM_H));
    CHECK(hipMalloc(&out, (size_t)n * CROP * CROP));
    CHECK(hipMalloc(&dense, (size_t)n * CROP * CROP));
    CHECK(hipMemset(cam, 7, (size_t)n * CAM_W * CAM_H));
    CHECK(hipMemset(dense, 9, (size_t)n * CROP * CROP));
    const double bytes = 2.0 * n * CROP * CROP;   // read + write
    printf("%d sensor frames 640x480 -> 128x128 crops: %.1f MB read + written per launch\n", n, bytes / 1e6);
    const int64_t db = (int64_t)n * CROP * CROP;
    double t = time_us([&] { hipLaunchKernelGGL(k_dense, dim3((unsigned)(db / (256 * 64))), dim3(256), 0, 0, dense, out, db); });
    printf("dense copy of the same bytes                          %8.2f us  %7.1f GB/s\n", t, bytes / t / 1e3);
#define RUN(U, NT, ROWS)                                                                                              \
    t = time_us([&] { hipLaunchKernelGGL((k_crop<U, NT>), dim3((unsigned)(n * (CROP / ROWS))), dim3(256), 0, 0, cam, out, ROWS); }); \
    printf("crop: %d pieces per lane in flight, %-3s loads, %3d rows per workgroup  %8.2f us  %7.1f GB/s\n", U, NT ? "nt" : "", ROWS, t, bytes / t / 1e3);
    RUN(1, false, 128) RUN(2, false, 128) RUN(4, false, 128) RUN(8, false, 128)
    RUN(1, true, 128) RUN(2, true, 128) RUN(4, true, 128) RUN(8, true, 128)
    RUN(4, false, 64) RUN(4, false, 32) RUN(2, false, 64) RUN(1, false, 32) RUN(4, true, 64) RUN(4, true, 32)
    CHECK(hipDeviceSynchronize());
    return 0;
}
