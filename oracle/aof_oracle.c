/*
 * aof_oracle.c -- scalar CPU restatement of the flow path (see aof_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the reference's engine source,
 * modules/OpticalFlow, is absent from /root/reference and the reference has no
 * tests or golden vectors).  Every function restates the published PX4Flow
 * algorithm (ICRA 2013) or a build-defined extension from DESIGN.md "Spec";
 * the reference anchors are the call sites in /root/reference/src/mainloop.cpp.
 * There is NO upstream commit this file was transcribed against: PX4/OpticalFlow (branch
 * static_lib, /root/reference/.gitmodules:1-4) was never readable here; what pins it instead
 * are analytic known answers and the hand-derived end-to-end vectors of
 * tests/test_hand_vectors.py, whose expectations are literals written down from the construction
 * of the frames (the oracle on CPU and the HIP path on the GPU are held to the same numbers).
 *
 * Written for clarity, not speed: plain loops, no intrinsics.
 */
#include "aof_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_SKIPPED 0xFFFFu
#define ORC_MAX_HIST 256

void orc_params_default(orc_params *p, int width, int height)
{
    /* Defaults = the dense configuration BASELINE.json quotes the metric on:
     * 8x8 tiles, +-4 search, gradient gate 30, SAD gate 3000 (PX4Flow's
     * published defaults), histogram filter on, integer shifts only. */
    memset(p, 0, sizeof(*p));
    p->width = width;
    p->height = height;
    p->tile = 8;
    p->search = 4;
    p->grid_mode = ORC_GRID_DENSE;
    p->num_blocks = 5;
    p->feature_threshold = 30;
    p->value_threshold = 3000;
    p->subpixel = 0;
    p->hist_filter = 1;
    p->pyramid_levels = 1;
    p->mean_subtract = 0;
    p->min_valid = 10;
}

int orc_params_check(const orc_params *p)
{
    if (!p) return -EINVAL;
    if (p->tile != 8 && p->tile != 16) return -EINVAL;
    if (p->search < 1 || p->search > 8) return -EINVAL;
    if (p->pyramid_levels != 1 && p->pyramid_levels != 2) return -EINVAL;
    if (p->grid_mode != ORC_GRID_DENSE && p->grid_mode != ORC_GRID_PX4FLOW) return -EINVAL;
    if (p->grid_mode == ORC_GRID_PX4FLOW && p->num_blocks < 1) return -EINVAL;
    if (p->width < 1 || p->height < 1) return -EINVAL;
    if (p->pyramid_levels == 2 && ((p->width | p->height) & 1)) return -EINVAL;
    if (p->feature_threshold < 0 || p->value_threshold < 0) return -EINVAL;
    for (int l = 0; l < p->pyramid_levels; l++) {
        orc_grid g;
        if (orc_grid_for_level(p, l, &g)) return -EINVAL;
    }
    return 0;
}

/* Spec "Grid".  DENSE: origin = margin M = S + (subpixel ? 1 : 0), step = B,
 * n = floor((dim - 2M) / B).  PX4FLOW: the published sparse grid
 * pixLo = S+1, pixHi = dim-(S+1)-B, step = (pixHi-pixLo)/num_blocks + 1,
 * positions pixLo, pixLo+step, ... < pixHi. */
int orc_grid_for_level(const orc_params *p, int level, orc_grid *g)
{
    int w = p->width >> level, h = p->height >> level;
    int B = p->tile, S = p->search;
    if (p->grid_mode == ORC_GRID_DENSE) {
        int M = S + (p->subpixel ? 1 : 0);
        g->x0 = g->y0 = M;
        g->step_x = g->step_y = B;
        g->nx = (w - 2 * M) / B;
        g->ny = (h - 2 * M) / B;
    } else {
        int lo = S + 1;
        int hix = w - (S + 1) - B, hiy = h - (S + 1) - B;
        if (hix <= lo || hiy <= lo) return -EINVAL;
        g->x0 = g->y0 = lo;
        g->step_x = (hix - lo) / p->num_blocks + 1;
        g->step_y = (hiy - lo) / p->num_blocks + 1;
        g->nx = (hix - lo + g->step_x - 1) / g->step_x;
        g->ny = (hiy - lo + g->step_y - 1) / g->step_y;
    }
    if (g->nx < 1 || g->ny < 1) return -EINVAL;
    return 0;
}

/* Half-range of the shift histogram in pixels of the level's own grid.
 * One level, or level 1 of two: R = S.  Level 0 of two: the predictor adds up
 * to 2S+1, so R = 3S+1. */
static int level_range(const orc_params *p, int level)
{
    if (p->pyramid_levels == 2 && level == 0) return 3 * p->search + 1;
    return p->search;
}

int orc_hist_size(const orc_params *p, int level)
{
    return 2 * (2 * level_range(p, level) + 1) + 1;
}

/* Published "compute_diff": gradient energy of the 4x4 patch in the middle of
 * the tile (offset B/2-2): sum of |vertical neighbour differences| over the
 * 3x4 row pairs plus |horizontal neighbour differences| over the 4x3 column
 * pairs. */
uint32_t orc_compute_diff(const uint8_t *img, int x, int y, int stride, int tile)
{
    int off = tile / 2 - 2;
    const uint8_t *p = img + (int64_t)(y + off) * stride + (x + off);
    uint32_t acc = 0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++)
            acc += (uint32_t)abs((int)p[r * stride + c] - (int)p[(r + 1) * stride + c]);
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 4; r++)
            acc += (uint32_t)abs((int)p[r * stride + c] - (int)p[r * stride + c + 1]);
    return acc;
}

/* Timing aid for bench.py's cpu_baseline leg: with orc_set_fast_sad(1) the 8x8 / 16x16 SAD
 * uses the host's own SAD instruction (SSE2 psadbw) so that the CPU is not handicapped by a
 * byte loop.  The checker (tests, smoke) never switches it on; tests/test_oracle.py pins it to
 * the scalar loop. */
static int g_fast_sad = 0;
void orc_set_fast_sad(int on) { g_fast_sad = on != 0; }
int orc_fast_sad_available(void)
{
#if defined(__SSE2__)
    return 1;
#else
    return 0;
#endif
}

#if defined(__SSE2__)
#include <emmintrin.h>
static uint32_t sad_sse2(const uint8_t *pa, const uint8_t *pb, int stride, int tile)
{
    __m128i acc = _mm_setzero_si128();
    if (tile == 8) {
        for (int r = 0; r < 8; r += 2) {
            const __m128i va = _mm_unpacklo_epi64(_mm_loadl_epi64((const __m128i *)(pa + (int64_t)r * stride)),
                                                  _mm_loadl_epi64((const __m128i *)(pa + (int64_t)(r + 1) * stride)));
            const __m128i vb = _mm_unpacklo_epi64(_mm_loadl_epi64((const __m128i *)(pb + (int64_t)r * stride)),
                                                  _mm_loadl_epi64((const __m128i *)(pb + (int64_t)(r + 1) * stride)));
            acc = _mm_add_epi64(acc, _mm_sad_epu8(va, vb));
        }
    } else {
        for (int r = 0; r < 16; r++)
            acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(pa + (int64_t)r * stride)),
                                                  _mm_loadu_si128((const __m128i *)(pb + (int64_t)r * stride))));
    }
    return (uint32_t)(_mm_cvtsi128_si32(acc) + _mm_cvtsi128_si32(_mm_srli_si128(acc, 8)));
}
#endif

/* Published "compute_sad_8x8", generalised to BxB. */
uint32_t orc_sad(const uint8_t *a, int ax, int ay, const uint8_t *b, int bx, int by,
                 int stride, int tile)
{
#if defined(__SSE2__)
    if (g_fast_sad && (tile == 8 || tile == 16))
        return sad_sse2(a + (int64_t)ay * stride + ax, b + (int64_t)by * stride + bx, stride, tile);
#endif
    uint32_t acc = 0;
    for (int r = 0; r < tile; r++) {
        const uint8_t *pa = a + (int64_t)(ay + r) * stride + ax;
        const uint8_t *pb = b + (int64_t)(by + r) * stride + bx;
        for (int c = 0; c < tile; c++) acc += (uint32_t)abs((int)pa[c] - (int)pb[c]);
    }
    return acc;
}

static inline int havg(int a, int b) { return (a + b) >> 1; } /* UHADD8: floor */

/* Published "compute_subpixel": SAD of the tile against the 8 half-pixel
 * neighbours of the best integer match in image b.  Direction numbering
 * (x right, y down):   5 6 7
 *                      4 X 0
 *                      3 2 1
 * Even directions average two pixels; odd (diagonal) ones average two of the
 * already-halved values, each step flooring. */
void orc_subpixel(const uint8_t *a, int ax, int ay, const uint8_t *b, int bx, int by,
                  int stride, int tile, uint32_t acc[8])
{
    for (int k = 0; k < 8; k++) acc[k] = 0;
    for (int r = 0; r < tile; r++) {
        for (int c = 0; c < tile; c++) {
            const uint8_t *q = b + (int64_t)(by + r) * stride + (bx + c);
            int p00 = q[0];
            int s0 = havg(p00, q[1]);
            int s1 = havg(q[stride], q[stride + 1]);
            int s2 = havg(p00, q[stride]);
            int s3 = havg(q[stride], q[stride - 1]);
            int s4 = havg(p00, q[-1]);
            int s5 = havg(q[-stride], q[-stride - 1]);
            int s6 = havg(p00, q[-stride]);
            int s7 = havg(q[-stride], q[-stride + 1]);
            int t1 = havg(s0, s1);
            int t3 = havg(s3, s4);
            int t5 = havg(s4, s5);
            int t7 = havg(s7, s0);
            int ref = a[(int64_t)(ay + r) * stride + (ax + c)];
            acc[0] += (uint32_t)abs(ref - s0);
            acc[1] += (uint32_t)abs(ref - t1);
            acc[2] += (uint32_t)abs(ref - s2);
            acc[3] += (uint32_t)abs(ref - t3);
            acc[4] += (uint32_t)abs(ref - s4);
            acc[5] += (uint32_t)abs(ref - t5);
            acc[6] += (uint32_t)abs(ref - s6);
            acc[7] += (uint32_t)abs(ref - t7);
        }
    }
}

/* Spec "Mean": round-half-up integer mean of all pixels. */
uint32_t orc_frame_mean(const uint8_t *img, int64_t n)
{
    uint64_t s = 0;
    for (int64_t i = 0; i < n; i++) s += img[i];
    return (uint32_t)((s + (uint64_t)n / 2) / (uint64_t)n);
}

/* Spec "Pyramid": 2x2 box, round-half-up. */
void orc_pyramid_down(const uint8_t *src, int w, int h, uint8_t *dst)
{
    int w1 = w / 2, h1 = h / 2;
    for (int y = 0; y < h1; y++)
        for (int x = 0; x < w1; x++) {
            const uint8_t *q = src + (int64_t)(2 * y) * w + 2 * x;
            dst[(int64_t)y * w1 + x] = (uint8_t)((q[0] + q[1] + q[w] + q[w + 1] + 2) >> 2);
        }
}

/* Spec "Equalise": cur' = clamp(cur + delta, 0, 255). */
void orc_equalise(const uint8_t *src, int64_t n, int delta, uint8_t *dst)
{
    for (int64_t i = 0; i < n; i++) {
        int v = (int)src[i] + delta;
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static void subdir_to_half(int sd, int *hx, int *hy)
{
    *hx = (sd == 0 || sd == 1 || sd == 7) ? 1 : ((sd == 3 || sd == 4 || sd == 5) ? -1 : 0);
    *hy = (sd == 1 || sd == 2 || sd == 3) ? 1 : ((sd == 5 || sd == 6 || sd == 7) ? -1 : 0);
}

/* Spec "Search": per block gradient gate, exhaustive SAD search in scan order
 * (dy outer, dx inner, strict '<' so the first minimum wins), optional
 * half-pixel refinement of accepted blocks. */
static void level_search(const orc_params *p, const uint8_t *prev, const uint8_t *cur, int w,
                         int h, const orc_grid *g, int pred_x, int pred_y, orc_block *blocks,
                         uint8_t *subdirs)
{
    const int B = p->tile, S = p->search, m = p->subpixel ? 1 : 0;
    const uint32_t vthr = p->value_threshold > 0xFFFF ? 0xFFFFu : (uint32_t)p->value_threshold;
    for (int by = 0; by < g->ny; by++) {
        for (int bx = 0; bx < g->nx; bx++) {
            const int i = g->x0 + bx * g->step_x, j = g->y0 + by * g->step_y;
            orc_block rec = {0, 0, ORC_SKIPPED};
            uint8_t sd = 8;
            const int lo_x = i + pred_x - S - m, hi_x = i + pred_x + S + m + B;
            const int lo_y = j + pred_y - S - m, hi_y = j + pred_y + S + m + B;
            if (lo_x >= 0 && lo_y >= 0 && hi_x <= w && hi_y <= h &&
                orc_compute_diff(prev, i, j, w, B) >= (uint32_t)p->feature_threshold) {
                uint32_t dist = 0xFFFFFFFFu;
                int sumx = 0, sumy = 0;
                for (int jj = -S; jj <= S; jj++)
                    for (int ii = -S; ii <= S; ii++) {
                        uint32_t t = orc_sad(prev, i, j, cur, i + pred_x + ii, j + pred_y + jj, w, B);
                        if (t < dist) {
                            dist = t;
                            sumx = ii;
                            sumy = jj;
                        }
                    }
                rec.dx = (int8_t)(pred_x + sumx);
                rec.dy = (int8_t)(pred_y + sumy);
                rec.sad = (uint16_t)dist;
                if (p->subpixel && dist < vthr) {
                    uint32_t acc[8];
                    orc_subpixel(prev, i, j, cur, i + rec.dx, j + rec.dy, w, B, acc);
                    uint32_t mind = dist;
                    for (int k = 0; k < 8; k++)
                        if (acc[k] < mind) {
                            mind = acc[k];
                            sd = (uint8_t)k;
                        }
                }
            }
            blocks[by * g->nx + bx] = rec;
            if (subdirs) subdirs[by * g->nx + bx] = sd;
        }
    }
}

static int64_t floor_div(int64_t a, int64_t b) /* b > 0 */
{
    int64_t q = a / b;
    if ((a % b) < 0) q--;
    return q;
}

/* Published histogram window around the peak bin. */
static void peak_window(int pos, int n, int *lo, int *hi)
{
    *lo = *hi = pos;
    if (pos > 1 && pos < n - 2) {
        *lo = pos - 2;
        *hi = pos + 2;
    } else if (pos == 0) {
        *hi = pos + 2;
    } else if (pos == n - 1) {
        *lo = pos - 2;
    } else if (pos == 1) {
        *lo = pos - 1;
        *hi = pos + 2;
    } else if (pos == n - 2) {
        *lo = pos - 2;
        *hi = pos + 1;
    }
}

/* Spec "Reduce": accepted = searched && sad < value_threshold.  Half-pixel
 * histograms (bin = 2*shift + (2R+1) +- 1), first-maximum peak, weighted mean
 * over the published +-2-bin window; or the plain average.  Quality =
 * count*255/total, zero when count <= min_valid.  Also yields the integer
 * level-1 -> level-0 predictor (round-half-up of the flow in half-pixels). */
void orc_reduce(const orc_params *p, const orc_block *blocks, const uint8_t *subdirs, int nblocks,
                int range, orc_flow *out, int32_t *pred_x, int32_t *pred_y)
{
    const int centre = 2 * range + 1, n = 2 * centre + 1;
    const uint32_t vthr = p->value_threshold > 0xFFFF ? 0xFFFFu : (uint32_t)p->value_threshold;
    uint32_t histx[ORC_MAX_HIST] = {0}, histy[ORC_MAX_HIST] = {0};
    uint32_t count = 0;
    int64_t sum2x = 0, sum2y = 0;
    for (int b = 0; b < nblocks; b++) {
        if (blocks[b].sad == ORC_SKIPPED || blocks[b].sad >= vthr) continue;
        int hx = 0, hy = 0;
        if (subdirs) subdir_to_half(subdirs[b], &hx, &hy);
        int ix = 2 * blocks[b].dx + centre + hx, iy = 2 * blocks[b].dy + centre + hy;
        histx[ix]++;
        histy[iy]++;
        sum2x += 2 * blocks[b].dx + hx;
        sum2y += 2 * blocks[b].dy + hy;
        count++;
    }
    out->flow_x = out->flow_y = 0.0f;
    out->count = count;
    out->quality = 0;
    out->flags = 0;
    if (pred_x) *pred_x = 0;
    if (pred_y) *pred_y = 0;
    if (!((int64_t)count > (int64_t)p->min_valid) || count == 0) return;

    if (p->hist_filter) {
        int posx = 0, posy = 0;
        uint32_t maxx = 0, maxy = 0;
        for (int k = 0; k < n; k++) {
            if (histx[k] > maxx) { maxx = histx[k]; posx = k; }
            if (histy[k] > maxy) { maxy = histy[k]; posy = k; }
        }
        int lo, hi;
        uint32_t vx = 0, wx = 0, vy = 0, wy = 0;
        peak_window(posx, n, &lo, &hi);
        for (int k = lo; k <= hi; k++) { vx += (uint32_t)k * histx[k]; wx += histx[k]; }
        peak_window(posy, n, &lo, &hi);
        for (int k = lo; k <= hi; k++) { vy += (uint32_t)k * histy[k]; wy += histy[k]; }
        out->flow_x = ((float)vx / (float)wx - (float)centre) / 2.0f;
        out->flow_y = ((float)vy / (float)wy - (float)centre) / 2.0f;
        if (pred_x) *pred_x = (int32_t)(floor_div(2 * (int64_t)vx + wx, 2 * (int64_t)wx) - centre);
        if (pred_y) *pred_y = (int32_t)(floor_div(2 * (int64_t)vy + wy, 2 * (int64_t)wy) - centre);
    } else {
        out->flow_x = ((float)sum2x * 0.5f) / (float)count;
        out->flow_y = ((float)sum2y * 0.5f) / (float)count;
        if (pred_x) *pred_x = (int32_t)floor_div(2 * sum2x + count, 2 * (int64_t)count);
        if (pred_y) *pred_y = (int32_t)floor_div(2 * sum2y + count, 2 * (int64_t)count);
    }
    out->quality = (uint8_t)((uint64_t)count * 255u / (uint64_t)nblocks);
    out->flags |= ORC_FLAG_FLOW_VALID;
}

int orc_flow_pair(const orc_params *p, const uint8_t *prev, const uint8_t *cur, orc_block *blocks,
                  uint8_t *subdirs, orc_block *blocks_l1, uint8_t *subdirs_l1, orc_flow *out)
{
    int rc = orc_params_check(p);
    if (rc) return rc;
    if (!prev || !cur || !out) return -EINVAL;
    const int W = p->width, H = p->height;
    const int64_t N0 = (int64_t)W * H;
    orc_grid g0, g1;
    orc_grid_for_level(p, 0, &g0);
    int32_t pred_x = 0, pred_y = 0;
    uint8_t flags = 0;
    uint8_t *tmp = NULL;      /* equalised cur, level 0 */
    orc_block *own_blocks = NULL;
    uint8_t *own_sub = NULL;

    if (p->pyramid_levels == 2) {
        const int w1 = W / 2, h1 = H / 2;
        const int64_t N1 = (int64_t)w1 * h1;
        orc_grid_for_level(p, 1, &g1);
        uint8_t *p1 = (uint8_t *)malloc((size_t)N1), *c1 = (uint8_t *)malloc((size_t)N1);
        orc_block *b1 = (orc_block *)malloc(sizeof(orc_block) * (size_t)(g1.nx * g1.ny));
        uint8_t *s1 = p->subpixel ? (uint8_t *)malloc((size_t)(g1.nx * g1.ny)) : NULL;
        orc_pyramid_down(prev, W, H, p1);
        orc_pyramid_down(cur, W, H, c1);
        if (p->mean_subtract) {
            int d1 = (int)orc_frame_mean(p1, N1) - (int)orc_frame_mean(c1, N1);
            orc_equalise(c1, N1, d1, c1);
        }
        level_search(p, p1, c1, w1, h1, &g1, 0, 0, b1, s1);
        orc_flow f1;
        orc_reduce(p, b1, s1, g1.nx * g1.ny, level_range(p, 1), &f1, &pred_x, &pred_y);
        if (f1.flags & ORC_FLAG_FLOW_VALID) flags |= ORC_FLAG_PRED_VALID;
        if (blocks_l1) memcpy(blocks_l1, b1, sizeof(orc_block) * (size_t)(g1.nx * g1.ny));
        if (subdirs_l1 && s1) memcpy(subdirs_l1, s1, (size_t)(g1.nx * g1.ny));
        free(p1); free(c1); free(b1); free(s1);
    }

    const uint8_t *cur0 = cur;
    if (p->mean_subtract) {
        int d0 = (int)orc_frame_mean(prev, N0) - (int)orc_frame_mean(cur, N0);
        tmp = (uint8_t *)malloc((size_t)N0);
        orc_equalise(cur, N0, d0, tmp);
        cur0 = tmp;
    }
    const int nb = g0.nx * g0.ny;
    if (!blocks) blocks = own_blocks = (orc_block *)malloc(sizeof(orc_block) * (size_t)nb);
    if (!subdirs && p->subpixel) subdirs = own_sub = (uint8_t *)malloc((size_t)nb);
    level_search(p, prev, cur0, W, H, &g0, pred_x, pred_y, blocks, p->subpixel ? subdirs : NULL);
    if (!p->subpixel && subdirs) memset(subdirs, 8, (size_t)nb);
    orc_reduce(p, blocks, p->subpixel ? subdirs : NULL, nb, level_range(p, 0), out, NULL, NULL);
    out->flags |= flags;
    out->pred_x = (int8_t)pred_x;
    out->pred_y = (int8_t)pred_y;
    free(tmp); free(own_blocks); free(own_sub);
    return 0;
}

int orc_flow_batch(const orc_params *p, const uint8_t *prev, const uint8_t *cur,
                   int64_t pair_stride, int64_t n_pairs, orc_block *blocks, orc_flow *flows,
                   int threads)
{
    orc_grid g0;
    if (orc_params_check(p)) return -EINVAL;
    orc_grid_for_level(p, 0, &g0);
    const int64_t nb = (int64_t)g0.nx * g0.ny;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < n_pairs; i++)
        orc_flow_pair(p, prev + i * pair_stride, cur + i * pair_stride,
                      blocks ? blocks + i * nb : NULL, NULL, NULL, NULL, &flows[i]);
    (void)threads;
    return used;
}

/* ---- frame ingest (reference code present: /root/reference/src/mainloop.cpp) ---- */
void orc_crop_rect(int cam_w, int cam_h, int crop_w, int crop_h, int *x0, int *y0)
{
    *x0 = cam_w / 2 - crop_w / 2; /* mainloop.cpp:295 */
    *y0 = cam_h / 2 - crop_h / 2; /* mainloop.cpp:296 */
}

int orc_exposure_bin(int v)
{
    /* cv::calcHist, uniform ranges: idx = cvFloor(v * (histSize / (high - low)) - low * ...)
     * with histSize = 10, low = 0, high = 255 (mainloop.cpp:210-214); out of range is dropped. */
    const double a = (double)ORC_EXPOSURE_BINS / (255.0 - 0.0);
    int idx = (int)floor((double)v * a);
    return (idx >= 0 && idx < ORC_EXPOSURE_BINS) ? idx : -1;
}

int orc_ingest(const uint8_t *cam, int cam_w, int cam_h, int crop_w, int crop_h, uint8_t *crop_out,
               uint32_t hist[ORC_EXPOSURE_BINS])
{
    if (!cam || crop_w < 1 || crop_h < 1 || crop_w > cam_w || crop_h > cam_h) return -EINVAL;
    int x0, y0;
    orc_crop_rect(cam_w, cam_h, crop_w, crop_h, &x0, &y0);
    /* mask rectangle inside the cropped image, mainloop.cpp:203-206 */
    int mx0 = crop_w / 2 - ORC_EXPOSURE_MASK_SIZE / 2, my0 = crop_h / 2 - ORC_EXPOSURE_MASK_SIZE / 2;
    int mx1 = mx0 + ORC_EXPOSURE_MASK_SIZE, my1 = my0 + ORC_EXPOSURE_MASK_SIZE;
    if (mx0 < 0) mx0 = 0;
    if (my0 < 0) my0 = 0;
    if (mx1 > crop_w) mx1 = crop_w;
    if (my1 > crop_h) my1 = crop_h;
    if (hist) memset(hist, 0, sizeof(uint32_t) * ORC_EXPOSURE_BINS);
    for (int y = 0; y < crop_h; y++)
        for (int x = 0; x < crop_w; x++) {
            const uint8_t v = cam[(int64_t)(y0 + y) * cam_w + (x0 + x)];
            if (crop_out) crop_out[(int64_t)y * crop_w + x] = v;
            if (hist && x >= mx0 && x < mx1 && y >= my0 && y < my1) {
                int b = orc_exposure_bin(v);
                if (b >= 0) hist[b]++;
            }
        }
    return 0;
}

float orc_exposure_msv(const uint32_t hist[ORC_EXPOSURE_BINS])
{
    float msv = 0.0f;
    for (int i = 0; i < ORC_EXPOSURE_BINS; i++)
        msv += (float)(i + 1) * (float)hist[i] / 16384.0f; /* mainloop.cpp:219 */
    return msv;
}

/* ---- gyro de-rotation ---- */
void orc_derotate(float flow_x, float flow_y, float gx, float gy, float dt_s, float focal_x,
                  float focal_y, float max_flow, float rate_threshold, float *out_x, float *out_y)
{
    volatile float lim = rate_threshold * dt_s; /* volatile: one rounding per operation */
    float x = flow_x, y = flow_y;
    if (fabsf(gy) > lim) {
        volatile float pix = gy * focal_x;
        volatile float c = flow_x + pix;
        x = c < -max_flow ? -max_flow : (c > max_flow ? max_flow : c);
    }
    if (fabsf(gx) > lim) {
        volatile float pix = gx * focal_y;
        volatile float c = flow_y - pix;
        y = c < -max_flow ? -max_flow : (c > max_flow ? max_flow : c);
    }
    *out_x = x;
    *out_y = y;
}

/* ---- facade semantics ---------------------------------------------------
 * The calcFlow contract visible at /root/reference/src/mainloop.cpp:322-331:
 * keep the previous frame, return a negative value until 1/output_rate has
 * elapsed, then hand out the flow integrated over that period as an angle
 * (rad) together with dt_us and a 0..255 quality. */
/* Pixel flow -> angular flow.  DESIGN.md "Facade semantics": atan2(flow_px, focal_px) evaluated as a FIXED
 * sequence of IEEE double operations (so that host and device agree to the bit; libm's atan2f is not
 * specified to the last bit): t = min(|x|,|y|) / max(|x|,|y|); above tan(pi/8) the argument is folded by
 * atan(t) = pi/4 + atan((t-1)/(t+1)); atan(u) = u (1 - z/3 + z^2/5 - ... + z^14/29), z = u^2, Horner from
 * the last term with every product and every difference rounded on its own (no fused multiply-add);
 * quadrant by the signs; rounded to float once at the end.  The oracle's own restatement of that spec. */
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("fp-contract=off")))
#endif
float orc_angle(float flow_px, float focal_px)
{
    const double PI = 3.14159265358979323846;
    double y = flow_px, x = focal_px;
    if (isnan(x) || isnan(y)) return flow_px + focal_px;
    double ax = fabs(x), ay = fabs(y);
    double small = ax < ay ? ax : ay, large = ax < ay ? ay : ax;
    double t = large == 0.0 ? 0.0 : (small == large ? 1.0 : small / large);
    double fold = 0.0, u = t;
    if (t > 0.41421356237309504880) {
        u = (t - 1.0) / (t + 1.0);
        fold = PI / 4.0;
    }
    double z = u * u, series = 1.0 / 29.0;
    for (int odd = 27; odd >= 1; odd -= 2) {
        double prod = series * z;
        series = 1.0 / (double)odd - prod;
    }
    double prod = u * series;
    double angle = fold + prod;
    if (ay > ax) angle = PI / 2.0 - angle;
    if (signbit(x)) angle = PI - angle;
    if (signbit(y)) angle = -angle;
    return (float)angle;
}

static void limit_rate_reset(orc_px4 *s)
{
    s->sum_flow_x = s->sum_flow_y = 0.0f;
    s->sum_flow_quality = 0;
    s->valid_frame_count = 0;
}

int orc_px4_init(orc_px4 *s, const orc_params *p, float fx, float fy, int output_rate)
{
    int rc = orc_params_check(p);
    if (rc) return rc;
    memset(s, 0, sizeof(*s));
    s->params = *p;
    s->focal_x = fx;
    s->focal_y = fy;
    s->output_rate = output_rate;
    s->img_old = (uint8_t *)malloc((size_t)p->width * p->height);
    limit_rate_reset(s);
    return s->img_old ? 0 : -ENOMEM;
}

void orc_px4_free(orc_px4 *s)
{
    free(s->img_old);
    s->img_old = NULL;
}

int orc_px4_calc_flow(orc_px4 *s, const uint8_t *img, uint32_t t_us, int *dt_us, float *flow_x,
                      float *flow_y)
{
    const size_t n = (size_t)s->params.width * s->params.height;
    if (!s->initialized) {
        memcpy(s->img_old, img, n);
        s->initialized = 1;
        return 0;
    }
    orc_flow f;
    orc_flow_pair(&s->params, s->img_old, img, NULL, NULL, NULL, NULL, &f);
    memcpy(s->img_old, img, n);
    int q = f.quality;
    float fx = f.flow_x, fy = f.flow_y;

    if (s->output_rate <= 0) { /* no rate limit */
        *dt_us = (int)(t_us - s->time_last_pub);
        s->time_last_pub = t_us;
    } else {
        if (q > 0) {
            s->sum_flow_x += fx;
            s->sum_flow_y += fy;
            s->sum_flow_quality += q;
            s->valid_frame_count++;
        }
        if ((float)(t_us - s->time_last_pub) > 1.0e6f / (float)s->output_rate) {
            int avg_q = 0;
            if (s->valid_frame_count > 0)
                avg_q = (int)floorf((float)s->sum_flow_quality / (float)s->valid_frame_count);
            fx = s->sum_flow_x;
            fy = s->sum_flow_y;
            limit_rate_reset(s);
            *dt_us = (int)(t_us - s->time_last_pub);
            s->time_last_pub = t_us;
            q = avg_q;
        } else {
            return -1; /* still integrating */
        }
    }
    *flow_x = orc_angle(fx, s->focal_x);
    *flow_y = orc_angle(fy, s->focal_y);
    return q;
}
