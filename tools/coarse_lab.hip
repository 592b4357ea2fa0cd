// coarse_lab -- phase timing of the fused coarse kernel (not part of the product).
// Builds k_coarse.hip with the lab definitions of its instrumentation points (tools/coarse_lab_hooks.hpp:
// in-kernel s_memrealtime stamps at the phase boundaries) and prints the median duration of every phase
// over all workgroups, next to the kernel's wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAOF_LAB_HOOKS='"../../tools/coarse_lab_hooks.hpp"' \
//         -Iinclude -Iaero-optical-flow_amd/csrc \
//         tools/coarse_lab.hip aero-optical-flow_amd/csrc/aof_params.cpp -o tools/coarse_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../aero-optical-flow_amd/csrc/k_coarse.hip"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

using namespace aof;

int main(int argc, char **argv)
{
    const int W = 640, H = 480, n = argc > 1 ? atoi(argv[1]) : 1024;
    aof_params p; aof_params_default(&p, W, H);
    p.pyramid_levels = 2; p.mean_subtract = 1;
    Grid g1; grid_for_level(p, 1, &g1);
    const size_t frame = (size_t)W * H;
    uint8_t *d_prev, *d_cur; aof_block *d_blocks; aof_flow *d_flows; uint32_t *d_sums;
    unsigned long long *d_stamps;
    CHECK(hipMalloc(&d_prev, frame * n)); CHECK(hipMalloc(&d_cur, frame * n));
    CHECK(hipMalloc(&d_blocks, sizeof(aof_block) * (size_t)g1.blocks() * n));
    CHECK(hipMalloc(&d_flows, sizeof(aof_flow) * n)); CHECK(hipMalloc(&d_sums, 16 * (size_t)n));
    CHECK(hipMalloc(&d_stamps, 64 * (size_t)n)); CHECK(hipMemset(d_stamps, 0, 64 * (size_t)n));
    std::vector<uint8_t> h(frame * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((rand() & 0x7F) + 40 + ((i / 7) & 31));
    if (argc > 4 && atoi(argv[4]) != 0)   // smooth the lab frames a little, so that the level-1 tiles match their moved copies
        for (size_t i = 1; i + 1 < h.size(); i++) h[i] = (uint8_t)((h[i - 1] + 2 * h[i] + h[i + 1]) / 4);
    const bool match = argc > 4 && atoi(argv[4]) != 0;   // cur = prev moved by (+4, -2): real matches, uniform votes (the bench's case)
    std::vector<uint8_t> moved(frame);
    for (int i = 0; i < n; i++) {
        const uint8_t *pf = h.data() + frame * (i % 8);
        CHECK(hipMemcpy(d_prev + frame * i, pf, frame, hipMemcpyHostToDevice));
        if (match) {
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) moved[(size_t)y * W + x] = pf[(size_t)((y + 2) % H) * W + (x + W - 4) % W];
            CHECK(hipMemcpy(d_cur + frame * i, moved.data(), frame, hipMemcpyHostToDevice));
        } else {
            CHECK(hipMemcpy(d_cur + frame * i, h.data() + frame * ((i + 3) % 8), frame, hipMemcpyHostToDevice));
        }
    }
    CoarseArgs a{};
    a.prev = d_prev; a.cur = d_cur; a.pair_stride = (int64_t)frame; a.w = W; a.h = H;
    a.tile = 8; a.search = 4; a.subpixel = 0; a.grid = g1; a.feature_threshold = 30; a.value_threshold = 3000;
    a.sums = d_sums; a.blocks = d_blocks; a.n_pairs = n;
    a.tail.nblocks = g1.blocks(); a.tail.range = 4; a.tail.hist_filter = 1; a.tail.min_valid = 10;
    a.tail.flows = d_flows; a.tail.pred = nullptr; a.tail.emit_predictor = 1;
    if (!coarse_fused_supported(a)) { printf("not supported\n"); return 1; }
    a.first_generation = 256;
    a.stagger_groups = argc > 2 ? atoi(argv[2]) : 1;
    a.stagger_ticks = argc > 3 ? atoi(argv[3]) : 1200;
    printf("stagger: %d groups, %d ticks of 10 ns apart\n", a.stagger_groups, a.stagger_ticks);
    unsigned long long *null_stamps = nullptr;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_stamps), &null_stamps, sizeof(null_stamps)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch_coarse_fused(a, nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < 30; i++) {
        CHECK(hipEventRecord(e0)); launch_coarse_fused(a, nullptr); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("k_coarse, %d VGA pairs: median %.4f ms (min %.4f)  = %.2f TB/s of frame reads\n", n, t[t.size() / 2], t[0],
           2.0 * frame * n / (t[t.size() / 2] * 1e-3) / 1e12);
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_stamps), &d_stamps, sizeof(d_stamps)));
    launch_coarse_fused(a, nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(8 * (size_t)n);
    CHECK(hipMemcpy(st.data(), d_stamps, 64 * (size_t)n, hipMemcpyDeviceToHost));
    const char *names[5] = {"phase 1 stream+box", "phase 2 equalise+gate", "phase 3 search", "phase 4 records+votes", "finalise (1 lane)"};
    unsigned long long first = ~0ull, last = 0;
    for (int w = 0; w < n; w++) { first = std::min(first, st[8 * w]); last = std::max(last, st[8 * w + 5]); }
    printf("stamped launch: first start -> last end %.1f us (s_memrealtime, 100 MHz)\n", (last - first) / 100.0);
    for (int k = 0; k < 5; k++) {
        std::vector<double> d;
        for (int w = 0; w < n; w++) d.push_back((st[8 * w + k + 1] - st[8 * w + k]) / 100.0);
        std::sort(d.begin(), d.end());
        printf("  %-24s median %7.2f us   p10 %7.2f   p90 %7.2f\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    std::vector<double> tot;
    for (int w = 0; w < n; w++) tot.push_back((st[8 * w + 5] - st[8 * w]) / 100.0);
    std::sort(tot.begin(), tot.end());
    printf("  %-24s median %7.2f us   p10 %7.2f   p90 %7.2f\n", "whole workgroup", tot[tot.size() / 2], tot[tot.size() / 10], tot[tot.size() * 9 / 10]);
    return 0;
}
