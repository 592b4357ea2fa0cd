/* Sanitizer self-test of the CPU oracle (TEST INFRASTRUCTURE).  Built by `make -C oracle asan`
 * with -fsanitize=address,undefined and run by tests/test_oracle_asan.py: GPU sanitizers are
 * not available on the pool, so memory safety is checked on the CPU restatement instead. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aof_oracle.h"

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static int run_case(orc_params p, int style)
{
    if (orc_params_check(&p)) return 0; /* geometry too small: fine */
    const size_t n = (size_t)p.width * p.height;
    uint8_t *a = malloc(n), *b = malloc(n);
    for (size_t i = 0; i < n; i++) { a[i] = (uint8_t)rnd(); b[i] = style == 1 ? a[i] : (uint8_t)rnd(); }
    if (style == 2) memset(b, 255, n);
    if (style == 3) { memset(a, 0, n); memset(b, 0, n); }
    orc_grid g0, g1 = {0, 0, 0, 0, 0, 0};
    orc_grid_for_level(&p, 0, &g0);
    if (p.pyramid_levels == 2) orc_grid_for_level(&p, 1, &g1);
    orc_block *blocks = malloc(sizeof(orc_block) * (size_t)(g0.nx * g0.ny));
    uint8_t *sub = malloc((size_t)(g0.nx * g0.ny));
    orc_block *b1 = malloc(sizeof(orc_block) * (size_t)(g1.nx * g1.ny + 1));
    uint8_t *s1 = malloc((size_t)(g1.nx * g1.ny + 1));
    orc_flow f;
    int rc = orc_flow_pair(&p, a, b, blocks, sub, b1, s1, &f);
    rc |= orc_flow_pair(&p, a, b, NULL, NULL, NULL, NULL, &f);
    free(a); free(b); free(blocks); free(sub); free(b1); free(s1);
    return rc;
}

int main(void)
{
    int fails = 0, cases = 0;
    for (int it = 0; it < 300; it++) {
        orc_params p;
        orc_params_default(&p, 24 + (int)(rnd() % 120), 24 + (int)(rnd() % 100));
        p.tile = (rnd() & 1) ? 8 : 16;
        p.search = 1 + (int)(rnd() % 8);
        p.grid_mode = (int)(rnd() % 2);
        p.num_blocks = 1 + (int)(rnd() % 8);
        p.subpixel = (int)(rnd() % 2);
        p.hist_filter = (int)(rnd() % 2);
        p.pyramid_levels = 1 + (int)(rnd() % 2);
        if (p.pyramid_levels == 2) { p.width += p.width & 1; p.height += p.height & 1; }
        p.mean_subtract = (int)(rnd() % 2);
        p.min_valid = (int)(rnd() % 3) * 5;
        p.feature_threshold = (int)(rnd() % 3) * 30;
        p.value_threshold = (int)(rnd() % 4) * 30000;
        fails += run_case(p, (int)(rnd() % 4)) != 0;
        cases++;
    }
    /* ingest + MSV + de-rotation + facade */
    for (int it = 0; it < 50; it++) {
        int cw = 16 + (int)(rnd() % 300), ch = 16 + (int)(rnd() % 200);
        int w = cw + (int)(rnd() % 100), h = ch + (int)(rnd() % 100);
        uint8_t *cam = malloc((size_t)w * h), *crop = malloc((size_t)cw * ch);
        for (int i = 0; i < w * h; i++) cam[i] = (uint8_t)rnd();
        uint32_t hist[ORC_EXPOSURE_BINS];
        fails += orc_ingest(cam, w, h, cw, ch, crop, hist) != 0;
        (void)orc_exposure_msv(hist);
        free(cam); free(crop);
    }
    float ox, oy;
    orc_derotate(1.0f, 2.0f, 0.01f, -0.02f, 0.013f, 216.f, 216.f, 4.5f, 0.01f, &ox, &oy);
    orc_params p;
    orc_params_default(&p, 64, 64);
    p.grid_mode = ORC_GRID_PX4FLOW; p.subpixel = 1;
    orc_px4 s;
    fails += orc_px4_init(&s, &p, 216.f, 216.f, 15) != 0;
    uint8_t *img = malloc(64 * 64);
    for (int k = 0; k < 20; k++) {
        for (int i = 0; i < 64 * 64; i++) img[i] = (uint8_t)rnd();
        int dt = 0; float fx = 0, fy = 0;
        (void)orc_px4_calc_flow(&s, img, (uint32_t)(k * 13333u + 0xFFFF0000u), &dt, &fx, &fy);
    }
    free(img);
    orc_px4_free(&s);
    printf("oracle selftest: %d cases, %d failures\n", cases, fails);
    return fails ? 1 : 0;
}
