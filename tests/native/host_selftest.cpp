// Sanitizer self-test of the product's HOST-ONLY logic (no HIP calls): parameter checks, grids,
// strip plans, workspace layout (aof_params.cpp) and the OPTICAL_FLOW_RAD packer.
// Built with -fsanitize=address,undefined by tests/test_host_asan.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aof.h"
#include "aof_internal.hpp"
#include "optical_flow_rad.hpp"

static unsigned rng_state = 777u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main()
{
    int bad = 0, valid = 0;
    for (int it = 0; it < 20000; it++) {
        aof_params p;
        aof_params_default(&p, (int)(rnd() % 2100) - 20, (int)(rnd() % 1300) - 20);
        p.tile = (rnd() % 5 == 0) ? (int)(rnd() % 20) : ((rnd() & 1) ? 8 : 16);
        p.search = (int)(rnd() % 11) - 1;
        p.grid_mode = (int)(rnd() % 3);
        p.num_blocks = (int)(rnd() % 12) - 1;
        p.subpixel = (int)(rnd() % 2);
        p.pyramid_levels = (int)(rnd() % 4);
        p.mean_subtract = (int)(rnd() % 2);
        p.value_threshold = (int)(rnd() % 100000) - 5;
        if (aof_params_check(&p) != 0) { bad++; continue; }
        valid++;
        for (int l = 0; l < p.pyramid_levels; l++) {
            int32_t g[6];
            if (aof_grid(&p, l, &g[0], &g[1], &g[2], &g[3], &g[4], &g[5])) return 1;
            if (g[4] < 1 || g[5] < 1) return 2;
            // every tile of the grid, with its search window, lies inside the level's frame
            const int w = p.width >> l, h = p.height >> l, m = p.subpixel ? 1 : 0;
            const int lx = g[0] + (g[4] - 1) * g[2], ly = g[1] + (g[5] - 1) * g[3];
            if (g[0] - p.search - m < 0 || lx + p.tile + p.search + m > w) return 3;
            if (g[1] - p.search - m < 0 || ly + p.tile + p.search + m > h) return 4;
            // the two-step reduction of large grids covers every block with its chunks
            const int chunks = aof::reduce_chunks(g[4] * g[5]);
            if (chunks < 0 || (chunks > 0 && (long long)chunks * 4096 < (long long)g[4] * g[5])) return 5;
            if (aof::hist_bytes_per_pair(p, l) != (size_t)chunks * 2 * (2 * (2 * (size_t)aof::level_range(p, l) + 1) + 1) * 4) return 6;
        }
        aof_ws_layout L;
        if (aof_workspace_layout(&p, (int64_t)(rnd() % 5000), &L)) return 7;
        if (L.total_bytes % 256 || L.l0_hist < L.l0_subdirs || L.total_bytes < L.l1_hist) return 8;
    }
    uint8_t wire[OPTICAL_FLOW_RAD_MAX_FRAME];
    for (int it = 0; it < 2000; it++) {
        OpticalFlowRad m;
        fillOpticalFlowRad(m, rnd(), rnd(), (int)rnd(), 0.001f * (float)(rnd() % 100), -0.002f * (float)(rnd() % 100),
                           0.1, 0.2, 0.3, (int)(rnd() % 256));
        size_t n = packOpticalFlowRad(m, (uint8_t)it, 1, 100, wire);
        if (n < 13 || n > OPTICAL_FLOW_RAD_MAX_FRAME || wire[0] != 0xFD || wire[1] != n - 12) return 9;
    }
    uint32_t hist[AOF_EXPOSURE_BINS] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10};
    if (aof_exposure_msv(hist) <= 0.0f || aof_exposure_bin(255) != -1 || aof_exposure_bin(-3) != -1) return 10;
    std::printf("host selftest: %d valid parameter sets, %d rejected\n", valid, bad);
    return valid > 1000 ? 0 : 11;
}
