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
    a.level = 0; a.n_pairs = n; a.hist_parts = nullptr; a.hist_range = 4; a.prune = 0;
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
    // pruned search: how many dy rows get the full eight row pairs?
    {
        g_lab_rb = 0; g_lab_dyg = 0; g_lab_stagger = -1;
        int mode = 0;
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_mode), &mode, sizeof(int)));
        for (int identical = 0; identical < 2; identical++) {
            SearchArgs b = a; b.prune = 1;
            if (identical & 1) b.cur = b.prev;  // every block matches itself at (0,0): row 4 first, all others prunable
            unsigned long long z[2] = {0, 0};
            int one = 1, zero = 0;
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_lab_rows), z, sizeof(z)));
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_count), &one, sizeof(int)));
            launch_search_tile8(b, nullptr);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpyFromSymbol(z, HIP_SYMBOL(d_lab_rows), sizeof(z)));
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_count), &zero, sizeof(int)));
            float tm[2];
            for (int mode2 = 0; mode2 < 2; mode2++) {
                CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_mode), &mode2, sizeof(int)));
                tm[mode2] = time_launch(b, 9);
            }
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lab_mode), &zero, sizeof(int)));
            float t = tm[0];
            printf("   (no staging: %.4f ms) ", tm[1]);
            printf("pruned %s: %.4f ms, rows visited %llu, fully evaluated %llu (%.1f %%)\n",
                   (identical & 1) ? "cur == prev" : "unrelated frames", t, z[0], z[1], 100.0 * z[1] / (z[0] ? z[0] : 1));
        }
    }
    return 0;
}
