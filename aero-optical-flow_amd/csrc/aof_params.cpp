// Host-only parameter logic of the C ABI (include/aof.h): defaults, validation,
// block grids and the workspace layout.  No HIP calls here, so these entry
// points work on a machine without a GPU.
#include <cerrno>
#include <cstring>

#include "aof_internal.hpp"

namespace aof {

int grid_for_level(const aof_params &p, int level, Grid *g)
{
    const int w = p.width >> level, h = p.height >> level;
    const int B = p.tile, S = p.search;
    if (p.grid_mode == AOF_GRID_DENSE) {
        const int M = S + (p.subpixel ? 1 : 0);
        g->x0 = g->y0 = M;
        g->step_x = g->step_y = B;
        g->nx = (w - 2 * M) / B;
        g->ny = (h - 2 * M) / B;
    } else {
        // Published PX4Flow grid: tiles spread between S+1 and dim-(S+1)-B.
        const int lo = S + 1;
        const int hix = w - (S + 1) - B, hiy = h - (S + 1) - B;
        if (hix <= lo || hiy <= lo) return -EINVAL;
        g->x0 = g->y0 = lo;
        g->step_x = (hix - lo) / p.num_blocks + 1;
        g->step_y = (hiy - lo) / p.num_blocks + 1;
        g->nx = (hix - lo + g->step_x - 1) / g->step_x;
        g->ny = (hiy - lo + g->step_y - 1) / g->step_y;
    }
    return (g->nx >= 1 && g->ny >= 1) ? 0 : -EINVAL;
}

int level_range(const aof_params &p, int level)
{
    return (p.pyramid_levels == 2 && level == 0) ? 3 * p.search + 1 : p.search;
}

int value_threshold_u16(const aof_params &p)
{
    return p.value_threshold > 0xFFFF ? 0xFFFF : p.value_threshold;
}

// K3 on grids beyond kReduceChunk*2 blocks runs in two steps: one workgroup per chunk of records
// votes into a partial histogram, then the usual per-pair workgroup sums the chunks.
int reduce_chunks(int nblocks)
{
    const int kReduceChunk = 4096;
    return nblocks > 2 * kReduceChunk ? (nblocks + kReduceChunk - 1) / kReduceChunk : 0;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace aof

using namespace aof;

namespace aof {

// Bytes of per-pair vote-histogram scratch of a level: the per-chunk histograms of the two-step
// reduction of large grids (0 for grids one reduction workgroup reads alone).
size_t hist_bytes_per_pair(const aof_params &p, int level)
{
    Grid g;
    if (grid_for_level(p, level, &g)) return 0;
    const size_t bins = 2 * (2 * (size_t)level_range(p, level) + 1) + 1;
    return (size_t)reduce_chunks(g.blocks()) * 2 * bins * sizeof(uint32_t);
}

}  // namespace aof

extern "C" {

int aof_version(void) { return AOF_VERSION; }

const char *aof_strerror(int code)
{
    switch (code) {
    case 0: return "ok";
    case -EINVAL: return "invalid argument";
    case -ENODEV: return "no usable gfx950 device";
    case -ENOMEM: return "out of memory";
    case -EIO: return "HIP runtime error";
    case -ENOSPC: return "workspace too small";
    default: return "unknown error";
    }
}

float aof_exposure_msv(const uint32_t hist[AOF_EXPOSURE_BINS])
{
    // /root/reference/src/mainloop.cpp:216-220, same float operation order
    float msv = 0.0f;
    for (int i = 0; i < AOF_EXPOSURE_BINS; i++) msv += (i + 1) * (float)hist[i] / 16384.0f;
    return msv;
}

int aof_exposure_bin(int grey)
{
    if (grey < 0 || grey > 255) return -1;
    const int b = (grey * 10) / 255;  // == cvFloor(grey * (10 / 255.0)) for every 8-bit value
    return b < AOF_EXPOSURE_BINS ? b : -1;
}

int aof_params_default(aof_params *p, int width, int height)
{
    if (!p) return -EINVAL;
    std::memset(p, 0, sizeof(*p));
    p->width = width;
    p->height = height;
    p->tile = 8;
    p->search = 4;
    p->grid_mode = AOF_GRID_DENSE;
    p->num_blocks = 5;
    p->feature_threshold = 30;
    p->value_threshold = 3000;
    p->subpixel = 0;
    p->hist_filter = 1;
    p->pyramid_levels = 1;
    p->mean_subtract = 0;
    p->min_valid = 10;
    return 0;
}

int aof_params_px4flow(aof_params *p, int width, int height, int search, int feature_threshold,
                       int value_threshold)
{
    int rc = aof_params_default(p, width, height);
    if (rc) return rc;
    p->search = search;
    p->grid_mode = AOF_GRID_PX4FLOW;
    p->num_blocks = 5;
    p->feature_threshold = feature_threshold;
    p->value_threshold = value_threshold;
    p->subpixel = 1;
    return 0;
}

int aof_params_check(const aof_params *p)
{
    if (!p) return -EINVAL;
    if (p->tile != 8 && p->tile != 16) return -EINVAL;
    if (p->search < 1 || p->search > 8) return -EINVAL;
    if (p->pyramid_levels != 1 && p->pyramid_levels != 2) return -EINVAL;
    if (p->grid_mode != AOF_GRID_DENSE && p->grid_mode != AOF_GRID_PX4FLOW) return -EINVAL;
    if (p->grid_mode == AOF_GRID_PX4FLOW && p->num_blocks < 1) return -EINVAL;
    if (p->width < 1 || p->height < 1) return -EINVAL;
    if ((int64_t)p->width * p->height > (int64_t)1 << 24) return -EINVAL;  // u32 pixel sums
    if (p->pyramid_levels == 2 && ((p->width | p->height) & 1)) return -EINVAL;
    if (p->feature_threshold < 0 || p->value_threshold < 0) return -EINVAL;
    for (int l = 0; l < p->pyramid_levels; l++) {
        Grid g;
        if (grid_for_level(*p, l, &g)) return -EINVAL;
    }
    return 0;
}

int aof_grid(const aof_params *p, int level, int32_t *x0, int32_t *y0, int32_t *step_x,
             int32_t *step_y, int32_t *nx, int32_t *ny)
{
    if (!p || level < 0 || level >= p->pyramid_levels) return -EINVAL;
    int rc = aof_params_check(p);
    if (rc) return rc;
    Grid g;
    grid_for_level(*p, level, &g);
    if (x0) *x0 = g.x0;
    if (y0) *y0 = g.y0;
    if (step_x) *step_x = g.step_x;
    if (step_y) *step_y = g.step_y;
    if (nx) *nx = g.nx;
    if (ny) *ny = g.ny;
    return 0;
}

int aof_workspace_layout(const aof_params *p, int64_t n_pairs, aof_ws_layout *out)
{
    if (!p || !out || n_pairs < 0) return -EINVAL;
    int rc = aof_params_check(p);
    if (rc) return rc;
    Grid g0, g1 = {0, 0, 0, 0, 0, 0};
    grid_for_level(*p, 0, &g0);
    const bool two = p->pyramid_levels == 2;
    if (two) grid_for_level(*p, 1, &g1);
    const size_t n = (size_t)n_pairs;
    const size_t l1_frame = two ? (size_t)(p->width / 2) * (size_t)(p->height / 2) : 0;
    size_t off = 0;
    std::memset(out, 0, sizeof(*out));
    out->sums = off;        off = align_up(off + n * 4 * sizeof(uint32_t), 256);
    out->l1_prev = off;     off = align_up(off + n * l1_frame, 256);
    out->l1_cur = off;      off = align_up(off + n * l1_frame, 256);
    out->l1_blocks = off;   off = align_up(off + n * (size_t)g1.blocks() * sizeof(aof_block), 256);
    out->l1_subdirs = off;  off = align_up(off + n * (size_t)g1.blocks(), 256);
    out->l1_flows = off;    off = align_up(off + (two ? n * sizeof(aof_flow) : 0), 256);
    out->l0_blocks = off;   off = align_up(off + n * (size_t)g0.blocks() * sizeof(aof_block), 256);
    out->l0_subdirs = off;  off = align_up(off + n * (size_t)g0.blocks(), 256);
    for (int level = 0; level < p->pyramid_levels; level++) {
        (level ? out->l1_hist : out->l0_hist) = off;
        off = align_up(off + n * hist_bytes_per_pair(*p, level), 256);
    }
    out->hints = off;       off = align_up(off + (p->tile == 16 ? n * sizeof(uint32_t) : 0), 256);
    out->total_bytes = off ? off : 256;
    return 0;
}

}  // extern "C"
