// k2_lab -- experiment bench for the tile8 search kernel (not part of the product).
// Builds the kernel source with -DAOF_LAB hooks and times variants on random frames.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAOF_LAB -Iinclude -Iaero-optical-flow_amd/csrc \
//         tools/k2_lab.hip aero-optical-flow_amd/csrc/aof_params.cpp -o tools/k2_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace aof { extern int g_lab_rb, g_lab_dyg; }
#include "../aero-optical-flow_amd/csrc/k_search_tile8.hip"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

using namespace aof;

static float time_launch(const SearchArgs &a, int reps)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch_search_tile8(a, nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(e0));
        int rc = launch_search_tile8(a, nullptr);
        if (rc) { printf("launch failed %d\n", rc); exit(1); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    const int W = 640, H = 480, n = argc > 1 ? atoi(argv[1]) : 1024;
    aof_params p; aof_params_default(&p, W, H);
    Grid g; grid_for_level(p, 0, &g);
    const size_t frame = (size_t)W * H;
    uint8_t *d_prev, *d_cur; aof_block *d_blocks;
    CHECK(hipMalloc(&d_prev, frame * n)); CHECK(hipMalloc(&d_cur, frame * n));
    CHECK(hipMalloc(&d_blocks, sizeof(aof_block) * (size_t)g.blocks() * n));
    std::vector<uint8_t> h(frame * 8);
    srand(1);
    for (auto &v : h) v = rand() & 0xFF;
    for (int i = 0; i < n; i++) {
        CHECK(hipMemcpy(d_prev + frame * i, h.data() + frame * (i % 8), frame, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_cur + frame * i, h.data() + frame * ((i + 3) % 8), frame, hipMemcpyHostToDevice));
    }
    SearchArgs a{};
    a.prev = d_prev; a.cur = d_cur; a.pair_stride = (int64_t)frame; a.w = W; a.h = H;
    a.tile = 8; a.search = 4; a.grid = g; a.feature_threshold = 30; a.value_threshold = 3000;
    a.subpixel = 0; a.blocks = d_blocks; a.subdirs = nullptr; a.pred = nullptr; a.sums = nullptr;
    a.level = 0; a.n_pairs = n; a.hist_parts = nullptr; a.hist_range = 4;
    const double alg = (2.0 * frame + 4.0 * g.blocks() + 16) * n;
    printf("pairs %d  blocks/pair %d\n", n, g.blocks());
    printf("%4s %8s %6s %10s %10s %10s %8s\n", "rb", "threads", "lds_KB", "full_ms", "nostage_ms", "nosearch_ms", "roof%");
    const int only_rb = argc > 2 ? atoi(argv[2]) : 0;
    for (int stag : {0, 8, 16, 24, 32, 48})
    for (int dyg : {9})
    for (int rb = 1; rb <= 6; rb++) {
        if (only_rb && rb != only_rb) continue;
        g_lab_rb = rb; g_lab_dyg = dyg; g_lab_stagger = stag;
        Tile8Plan pl = plan_tile8(a.w, a.grid.nx, a.grid.ny);
        if (!pl.rb) continue;
        float t[3];
        for (int mode = 0; mode < 3; mode++) {
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_mode), &mode, sizeof(int)));
            t[mode] = time_launch(a, 9);
        }
        printf("stag%-3d dyg%d %4d %8d %6.1f %10.4f %10.4f %10.4f %8.2f\n", stag, dyg, rb, pl.threads, pl.lds / 1024.0, t[0], t[1], t[2],
               100.0 * alg / (t[0] * 1e-3) / 8e12);
    }
    return 0;
}
