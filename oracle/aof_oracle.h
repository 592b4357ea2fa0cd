/*
 * aof_oracle.h -- CPU oracle for the sparse SAD block-matching flow path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product (include/aof.h,
 * aero-optical-flow_amd/) never links or calls it.
 *
 * PARITY UNPINNED.  The reference (intel-aero/aero-optical-flow) keeps this
 * arithmetic in the git submodule modules/OpticalFlow (PX4/OpticalFlow,
 * branch static_lib, commit unknown: /root/reference/.gitmodules:1-4), which
 * is an empty directory in the mount, and the reference has no tests, golden
 * vectors or fixtures.  This file therefore restates the PUBLISHED PX4Flow
 * block-matching algorithm (Honegger, Meier, Tanskanen, Pollefeys, "An Open
 * Source and Open Hardware Embedded Metric Optical Flow CMOS Camera for
 * Indoor and Outdoor Applications", ICRA 2013: 8x8 SAD, +-4 px search,
 * 4x4 gradient gate, half-pixel refinement, histogram-filtered flow) plus
 * the build-defined extensions BASELINE.json names (dense grid, 2-level
 * pyramid, mean equalisation, 16x16 tiles).  The normative text is
 * DESIGN.md section "Spec"; the reference-side anchors are its call sites
 * only: /root/reference/src/mainloop.cpp:322 (calcFlow), :327-331 (negative
 * return gate), :359-371 (outputs), :423-424 (constructor).
 */
#ifndef AOF_ORACLE_H
#define AOF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_GRID_DENSE 0
#define ORC_GRID_PX4FLOW 1

typedef struct orc_params {
    int32_t width, height;      /* level-0 frame size, stride == width (mainloop.cpp:317-320) */
    int32_t tile;               /* B: SAD tile edge, 8 or 16 */
    int32_t search;             /* S: search radius in pixels */
    int32_t grid_mode;          /* ORC_GRID_DENSE | ORC_GRID_PX4FLOW */
    int32_t num_blocks;         /* PX4FLOW grid: tiles per axis */
    int32_t feature_threshold;  /* 4x4 gradient gate */
    int32_t value_threshold;    /* SAD acceptance threshold */
    int32_t subpixel;           /* half-pixel refinement on/off */
    int32_t hist_filter;        /* histogram peak filter (1) or plain average (0) */
    int32_t pyramid_levels;     /* 1 or 2 */
    int32_t mean_subtract;      /* equalise cur to prev frame mean, per level */
    int32_t min_valid;          /* flow is valid iff count > min_valid */
} orc_params;

/* 4 bytes per block: integer shift of the best match and its SAD.
 * sad == 0xFFFF: block skipped (gradient gate failed or window outside frame). */
typedef struct orc_block {
    int8_t dx, dy;
    uint16_t sad;
} orc_block;

#define ORC_FLAG_FLOW_VALID 1u
#define ORC_FLAG_PRED_VALID 2u

/* 16 bytes per frame pair. */
typedef struct orc_flow {
    float flow_x, flow_y;   /* level-0 pixels */
    uint32_t count;         /* accepted blocks */
    uint8_t quality;        /* count*255/total_blocks */
    uint8_t flags;
    int8_t pred_x, pred_y;  /* level-1 predictor in level-0 pixels (0 if 1 level) */
} orc_flow;

typedef struct orc_grid {
    int32_t x0, y0, step_x, step_y, nx, ny;
} orc_grid;

/* --- building blocks (exposed so tests can pin each one) ---------------- */
void orc_params_default(orc_params *p, int width, int height);
int orc_params_check(const orc_params *p);
int orc_grid_for_level(const orc_params *p, int level, orc_grid *g);
int orc_hist_size(const orc_params *p, int level);
uint32_t orc_compute_diff(const uint8_t *img, int x, int y, int stride, int tile);
uint32_t orc_sad(const uint8_t *a, int ax, int ay, const uint8_t *b, int bx, int by,
                 int stride, int tile);
void orc_subpixel(const uint8_t *a, int ax, int ay, const uint8_t *b, int bx, int by,
                  int stride, int tile, uint32_t acc[8]);
/* bench.py's cpu_baseline leg only: SAD through the host's SAD instruction (SSE2 psadbw)
 * instead of the byte loop; results are identical (tests/test_oracle.py). */
void orc_set_fast_sad(int on);
int orc_fast_sad_available(void);
uint32_t orc_frame_mean(const uint8_t *img, int64_t n);
void orc_pyramid_down(const uint8_t *src, int w, int h, uint8_t *dst);
void orc_equalise(const uint8_t *src, int64_t n, int delta, uint8_t *dst);
/* Flow reduction over block records: fills out->flow_x/flow_y/count/quality/flags.
 * range = histogram half-range R in pixels; scale = multiply of the result. */
void orc_reduce(const orc_params *p, const orc_block *blocks, const uint8_t *subdirs,
                int nblocks, int range, orc_flow *out, int32_t *pred_x, int32_t *pred_y);

/* --- the path ------------------------------------------------------------
 * prev/cur: width*height u8.  blocks: nx0*ny0 records (level 0), may be NULL.
 * subdirs: nx0*ny0 bytes (half-pixel direction 0..7, 8 = none), may be NULL.
 * blocks_l1/subdirs_l1: level-1 records when pyramid_levels == 2, may be NULL.
 * Returns 0, or a negative errno-style code for bad parameters. */
int orc_flow_pair(const orc_params *p, const uint8_t *prev, const uint8_t *cur,
                  orc_block *blocks, uint8_t *subdirs,
                  orc_block *blocks_l1, uint8_t *subdirs_l1, orc_flow *out);

/* n_pairs independent pairs, frame i at prev + i*pair_stride; OpenMP over pairs
 * when built with -fopenmp (threads = 0: library default). Returns threads used. */
int orc_flow_batch(const orc_params *p, const uint8_t *prev, const uint8_t *cur,
                   int64_t pair_stride, int64_t n_pairs, orc_block *blocks,
                   orc_flow *flows, int threads);

/* --- frame ingest: the caller-side steps the reference runs on the host just
 *     before calcFlow.  Unlike the engine these ARE
 *     present in the reference, so each function cites the lines it follows. --- */
#define ORC_EXPOSURE_MASK_SIZE 128 /* /root/reference/src/mainloop.cpp:52 */
#define ORC_EXPOSURE_BINS 10       /* mainloop.cpp:210 */
/* Centre crop rectangle, mainloop.cpp:295-297. */
void orc_crop_rect(int cam_w, int cam_h, int crop_w, int crop_h, int *x0, int *y0);
/* Histogram bin of a grey value as cv::calcHist computes it for 10 uniform bins over
 * the half-open range [0,255) (mainloop.cpp:208-214): floor(v * (10/255.0)) in double;
 * returns -1 for values outside the range (v == 255). */
int orc_exposure_bin(int v);
/* crop (mainloop.cpp:295-298,317-319: contiguous copy) + masked histogram of the centred
 * 128x128 region of the CROPPED image (mainloop.cpp:203-214).  The reference needs
 * crop >= 128 (cv::Rect outside the matrix asserts); smaller crops clip the mask here. */
int orc_ingest(const uint8_t *cam, int cam_w, int cam_h, int crop_w, int crop_h,
               uint8_t *crop_out, uint32_t hist[ORC_EXPOSURE_BINS]);
/* Mean sample value, mainloop.cpp:216-220: sum (i+1)*hist[i]/16384.0f in float, i ascending. */
float orc_exposure_msv(const uint32_t hist[ORC_EXPOSURE_BINS]);

/* --- gyro de-rotation (published PX4Flow compensation).
 *     gx, gy: gyro angles integrated over dt as the reference accumulates them
 *     (/root/reference/src/mainloop.cpp:393-395); "-y gives x flow, x gives y flow"
 *     (mainloop.cpp:364-365). --- */
void orc_derotate(float flow_x, float flow_y, float gx, float gy, float dt_s, float focal_x,
                  float focal_y, float max_flow, float rate_threshold, float *out_x, float *out_y);

/* --- facade semantics (calcFlow: previous-frame keeping, rate limiting,
 *     pixel->angle conversion), mainloop.cpp:322-331,359-363 ---------------- */
/* Pixel flow -> angular flow (rad): the spec's fixed-operation atan2 (DESIGN.md "Facade semantics"). */
float orc_angle(float flow_px, float focal_px);

typedef struct orc_px4 {
    orc_params params;
    float focal_x, focal_y;
    int output_rate;
    int initialized;
    uint8_t *img_old;
    /* rate limiter */
    uint32_t time_last_pub;
    float sum_flow_x, sum_flow_y;
    int sum_flow_quality;
    int valid_frame_count;
} orc_px4;

int orc_px4_init(orc_px4 *s, const orc_params *p, float fx, float fy, int output_rate);
void orc_px4_free(orc_px4 *s);
int orc_px4_calc_flow(orc_px4 *s, const uint8_t *img, uint32_t img_time_us, int *dt_us,
                      float *flow_x, float *flow_y);

#ifdef __cplusplus
}
#endif
#endif
