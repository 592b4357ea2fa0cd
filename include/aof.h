/*
 * aof.h -- C ABI of the MI355X-native sparse SAD block-matching flow engine.
 *
 * This is the drop-in boundary for the one hot path of
 * intel-aero/aero-optical-flow: the frame-to-frame flow estimate that
 * Mainloop::camera_callback() obtains from the PX4 OpticalFlow submodule via
 *     _optical_flow->calcFlow(img, t_us, dt_us, flow_x, flow_y)
 * (/root/reference/src/mainloop.cpp:322; constructor :423-424; getters
 * :295-297; negative-return gate :327-331).  The reference binds that path as
 * a C++ class (header <flow_opencv.hpp>, /root/reference/src/mainloop.h:36);
 * the C++ facade in aero-optical-flow_amd/facade re-exports those classes on
 * top of this ABI, and INTEGRATION.md shows the CMake hook that replaces
 * modules/OpticalFlow (/root/reference/CMakeLists.txt:10,17).
 *
 * Conventions: plain C types, caller owns every buffer, no exceptions cross
 * the boundary.  Functions return 0 on success or a negative errno-style code;
 * aof_last_error() gives the text.  A context is thread-compatible (one
 * caller at a time), which is what the reference guarantees: calcFlow is only
 * called under _mainloop_lock (/root/reference/src/mainloop.cpp:283).
 *
 * There is NO CPU fallback: every entry point that computes flow runs the HIP
 * kernels on a gfx950 device and fails (-ENODEV) when none is present.
 */
#ifndef AOF_H
#define AOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its exports */
#endif

#define AOF_VERSION 102 /* 0.1.2: ADAPTIVE is the default search mode of 8x8 contexts too, aof_search_stats */

#define AOF_GRID_DENSE 0   /* origin = margin, step = tile */
#define AOF_GRID_PX4FLOW 1 /* published sparse grid: num_blocks tiles per axis */

/* Algorithm parameters (DESIGN.md "Spec").  aof_params_default() gives the
 * configuration BASELINE.json quotes the metric on. */
typedef struct aof_params {
    int32_t width, height;     /* level-0 frame, 8-bit grey, row stride == width
                                  (the caller makes it contiguous, mainloop.cpp:317-320) */
    int32_t tile;              /* B: 8 or 16 */
    int32_t search;            /* S: 1..8 */
    int32_t grid_mode;         /* AOF_GRID_* */
    int32_t num_blocks;        /* AOF_GRID_PX4FLOW only */
    int32_t feature_threshold; /* 4x4 gradient gate */
    int32_t value_threshold;   /* SAD acceptance gate */
    int32_t subpixel;          /* half-pixel refinement */
    int32_t hist_filter;       /* 1: histogram peak filter, 0: plain average */
    int32_t pyramid_levels;    /* 1 or 2 */
    int32_t mean_subtract;     /* equalise cur to prev frame mean per level */
    int32_t min_valid;         /* flow valid iff accepted blocks > min_valid */
} aof_params;

/* Per-block record, 4 bytes: integer shift of the best match and its SAD.
 * sad == 0xFFFF marks a skipped block (gradient gate / window outside frame). */
typedef struct aof_block {
    int8_t dx, dy;
    uint16_t sad;
} aof_block;

#define AOF_SAD_SKIPPED 0xFFFFu
#define AOF_FLAG_FLOW_VALID 1u
#define AOF_FLAG_PRED_VALID 2u

/* Per-pair result, 16 bytes. */
typedef struct aof_flow {
    float flow_x, flow_y; /* level-0 pixels */
    uint32_t count;       /* accepted blocks */
    uint8_t quality;      /* count*255/blocks, 0 when flow invalid (mainloop.cpp:371) */
    uint8_t flags;        /* AOF_FLAG_* */
    int8_t pred_x, pred_y;/* level-1 predictor, level-0 pixels */
} aof_flow;

/* Byte offsets of the intermediates inside the caller's workspace (tests and
 * tools read them; the layout is fixed by aof_workspace_layout()). */
typedef struct aof_ws_layout {
    size_t total_bytes;
    size_t sums;      /* uint32 [n_pairs][2 frames: prev,cur][2 levels] pixel sums */
    size_t l1_prev;   /* u8 [n_pairs][h/2][w/2] */
    size_t l1_cur;    /* u8 [n_pairs][h/2][w/2] */
    size_t l1_blocks; /* aof_block [n_pairs][nb1] */
    size_t l1_subdirs;/* u8 [n_pairs][nb1] */
    size_t l1_flows;  /* aof_flow [n_pairs] (pred_x/pred_y = predictor) */
    size_t l0_blocks; /* aof_block [n_pairs][nb0] when the caller passes none */
    size_t l0_subdirs;/* u8 [n_pairs][nb0] when the caller passes none */
    size_t l0_hist;   /* u32 [n_pairs][chunks0][2][bins0]: per-chunk vote histograms (grids beyond 8 192 blocks) */
    size_t l1_hist;   /* u32 [n_pairs][chunks1][2][bins1] */
    size_t hints;     /* u32 [n_pairs], 16x16 tiles: the adaptive search's per-pair verdict low byte (0 = exhaustive scan; 1 / 2 / 3 / 4 = pruning pays, on two- / one- / four- / eight-row bounds), a diagnostic above it */
} aof_ws_layout;

typedef struct aof_ctx aof_ctx;

/* Kernel ids for aof_kernel_ms(). */
#define AOF_K_PYRAMID 0 /* K1: frame sums + 2x2 pyramid */
#define AOF_K_SEARCH_L1 1
#define AOF_K_REDUCE_L1 2
#define AOF_K_SEARCH 3  /* K2: SAD search, level 0 -- the dominant kernel */
#define AOF_K_REDUCE 4  /* K3: histogram-filtered flow reduction */
#define AOF_K_COUNT 5

int aof_version(void);
const char *aof_strerror(int code);

/* ---- parameters (host only, no GPU needed) ---- */
int aof_params_default(aof_params *p, int width, int height);
/* Published PX4Flow configuration: sparse grid, half-pixel refinement. */
int aof_params_px4flow(aof_params *p, int width, int height, int search,
                       int feature_threshold, int value_threshold);
int aof_params_check(const aof_params *p);
/* Block grid of a pyramid level: origin, step, counts. */
int aof_grid(const aof_params *p, int level, int32_t *x0, int32_t *y0, int32_t *step_x,
             int32_t *step_y, int32_t *nx, int32_t *ny);
int aof_workspace_layout(const aof_params *p, int64_t n_pairs, aof_ws_layout *out);

/* ---- context ---- */
/* device: HIP device ordinal.  Fails with -ENODEV when no gfx950 GPU is usable. */
int aof_create(const aof_params *p, int device, aof_ctx **out);
void aof_destroy(aof_ctx *ctx);
const char *aof_last_error(const aof_ctx *ctx);
int aof_get_params(const aof_ctx *ctx, aof_params *out);
/* Name of the search kernel variant the context selected ("lane8", "tile16_lds", "generic"). */
const char *aof_search_variant(const aof_ctx *ctx);
/* Force the generic search kernel (tests compare the two device paths). */
int aof_set_force_generic(aof_ctx *ctx, int on);
/* Search strategy.  All return bit-identical records.
 * EXHAUSTIVE: all candidates of every block are summed completely -- a data-independent rate.
 * PRUNED: exact partial-distortion elimination (8x8 tiles on grids of more than 256 blocks, and
 *   16x16 tiles).  The dy rows are visited outwards from dy = 0; after a few of a row's tile rows a
 *   wave drops the row when no lane's partial SAD can still beat its best (a partial sum only grows).
 *   The rate then depends on the images: fast when blocks have a clear match near the centre, slower
 *   than the exhaustive search on noise (16x16: up to 1.5x; the 8x8 kernel falls back per wave).
 * ADAPTIVE (the default): PRUNED where it pays.
 *   16x16 tiles: a small probe kernel in front of the search computes the two-row bounds of a sample of every
 *   pair's blocks (1.6 % of them) and the search runs a pair's block rows pruned when the bounds predict that few
 *   candidates survive, exhaustively otherwise (the verdicts live in the workspace, aof_ws_layout.hints).
 *   8x8 tiles (level-0 searches of at least 2 048 x 256 blocks per launch -- 112 VGA pairs --; everything else runs
 *   EXHAUSTIVE): a probe per launch would cost more than it saves, so the CONTEXT learns from its own launches.  The
 *   pruned kernel -- whose waves run a block exhaustively, judge from its SADs whether rows could have been dropped,
 *   and prune the next blocks where they could -- reports the share of blocks that pruned (plain stores into pinned
 *   host memory, read at the next enqueue, never waited for).  While that share is at least 40 % the context keeps
 *   launching it, and its waves prune from their FIRST block on (which votes for the dy row to start in; round 5) --
 *   1 024 VGA pairs per launch, same box (profiles/r05_final_c2_noise.txt): noise-free translations 1.6-1.85x EXHAUSTIVE, +-2 LSB
 *   of noise 1.5x, +-4 LSB 1.4x, +-8 LSB 1.2x --; otherwise it launches the exhaustive kernel, and the pruned one once in 16
 *   launches to look again (+-16 LSB and more: within 1 % of EXHAUSTIVE, where PRUNED alone loses 12 %).  A context's
 *   first launch and a graph captured from it use whatever is known at that moment (aof_set_search_belief tells a fresh
 *   context); half-pixel configurations prune too.  aof_get_search_stats tells what happened. */
#define AOF_SEARCH_EXHAUSTIVE 0
#define AOF_SEARCH_PRUNED 1
#define AOF_SEARCH_ADAPTIVE 2
int aof_set_search_mode(aof_ctx *ctx, int mode);
int aof_get_search_mode(const aof_ctx *ctx);
/* What the ADAPTIVE mode of an 8x8 context has done so far (diagnostics; tests assert on them). */
typedef struct aof_search_stats {
    uint64_t pruned_launches;     /* flat 8x8 searches run by the pruned kernel (it reports back) */
    uint64_t exhaustive_launches; /* ... by the exhaustive kernel because the reports said pruning does not pay */
    uint64_t reports_read;        /* launches' reports evaluated */
    int32_t belief;               /* -1 nothing known yet, 0 pruning does not pay on this context's images, 1 it does */
    int32_t paying_pct;           /* share of the last report's chunks that left with "pruning pays" */
} aof_search_stats;
int aof_get_search_stats(const aof_ctx *ctx, aof_search_stats *out);
/* What an ADAPTIVE 8x8 context believes about its images is learnt from its own launches, so a context that lives for
 * ONE batch never uses it: its first launch lets every wave judge its first block exhaustively (6-10 % slower than either
 * dedicated kernel).  Callers that know their footage -- from aof_get_search_stats of an earlier context over the same
 * camera, or from the sensor -- say so here: 1 = pruning pays (launches prune from the first block on), 0 = it does not
 * (the exhaustive kernel, with one pruned launch in 16 to look again), -1 = forget (judge again).  The context keeps
 * learning from its launches afterwards.  Speed only: the records are the same.  The better cure is to create a
 * context once and reuse it ("Batch callers"). */
int aof_set_search_belief(aof_ctx *ctx, int belief);

/* ---- the hot path, device-resident (batched) ----
 * d_prev/d_cur: device pointers, pair i at +i*pair_stride bytes, each frame
 * width*height bytes.  For a frame SEQUENCE pass d_cur = d_prev + width*height
 * and pair_stride = width*height: frame k is cur of pair k-1 and prev of pair k, and
 * the passes that work per frame (pixel sums, 2x2 pyramid) then run once per FRAME
 * (the workspace's level-1 frames, where the separate kernels write them, are then
 * n_pairs + 1 consecutive frames from offset l1_prev on).
 * d_blocks: [n_pairs][nb0] records (4-byte aligned) or NULL.  d_subdirs: [n_pairs][nb0] or NULL.
 * d_flows: [n_pairs], required.  d_workspace: >= aof_workspace_layout().total_bytes,
 * 256-byte aligned.  stream: hipStream_t (NULL = default stream).
 * Asynchronous: returns after enqueueing; no allocation, no host sync.
 * Calls of at most 128 small pairs (8x8 tiles, every grid 8..256 blocks, width a multiple of 16,
 * frames that fit LDS: the published sparse grid up to about 256x224) run as ONE kernel, a
 * workgroup per pair, whatever the number of levels; the workspace's level-1 frames then stay
 * untouched (aof_set_split_coarse(ctx, 1) selects the separate kernels instead). */
int aof_flow_batch_device(aof_ctx *ctx, const uint8_t *d_prev, const uint8_t *d_cur,
                          int64_t pair_stride, int64_t n_pairs, aof_block *d_blocks,
                          uint8_t *d_subdirs, aof_flow *d_flows, void *d_workspace,
                          size_t workspace_bytes, void *stream);

/* Two-level configurations of 8x8 tiles whose two level-1 frames fit one CU's LDS (VGA: 150 KB)
 * run their coarse passes -- pixel sums, 2x2 pyramid, level-1 search, level-1 reduction -- as
 * ONE kernel, a workgroup per pair, and never write the level-1 frames to memory (the
 * workspace regions l1_prev / l1_cur then stay untouched).  on = 1 runs them as the separate
 * kernels K1 / K2 / K3 instead, which also fills l1_prev / l1_cur (tests compare the two), and
 * keeps small batches (see aof_flow_batch_device) on the separate kernels as well. */
int aof_set_split_coarse(aof_ctx *ctx, int on);

/* 8x8 tiles on grids of more than 256 blocks (C2, C3): the search kernel also reduces -- every wave adds
 * its votes to the pair's record in the CONTEXT's vote memory with integer atomics and the last wave
 * of a pair writes its aof_flow -- so no K3 launch follows (at 128 VGA pairs per call, configs[3]'s
 * per-GPU share, K3 and its launch gap were a quarter of the step).  Integer adds commute: the records
 * are bit-identical to the separate K3's.  Because the vote memory belongs to the context, calls on
 * ONE context must not overlap on the device: eager calls on different streams are ordered behind each
 * other by the library; captured graphs that contain calls on a context must not be replayed
 * concurrently with each other or with eager calls on it (once such a graph has been captured, the
 * library's own eager calls on the context keep to the separate K3).  Both forms of the search have it: the exhaustive
 * scan (k_flow_lane8_flat) and, on dense grids, the pruned column walk (k_flow_lane8_cols, which adds the agreeing votes of
 * a walk once); launches that prune on other grids keep K3.  Launches of more than 2 048 pairs
 * keep K3 as well.
 * on = 0 (THE DEFAULT): K3 runs as a separate kernel behind the search.  on = 1: opt in.
 * A finaliser wave that does not see its pair's votes complete within the deadline (50 ms; a launch in which
 * that happens is broken) writes that pair's record as "nothing measured" -- flow 0, count 0, quality 0,
 * flags 0 -- and raises the context's fault word: every later aof_flow_batch_device / aof_flow_pair_host /
 * aof_stream_push_host on the context returns -EIO (aof_last_error names the pair).  The condition is
 * sticky, like a HIP fault: recover with a new context. */
int aof_set_reduce_fusion(aof_ctx *ctx, int on);
/* Diagnostic knob: the finaliser deadline above, in microseconds (default 50 000; at least 100, -EINVAL below:
 * a deadline no launch can meet would disable the context with one call). */
int aof_set_vote_deadline_us(aof_ctx *ctx, uint32_t microseconds);
/* Fault injection for the tests of that path: the deadline in ticks of the 100 MHz counter, unchecked (0 = every
 * finaliser gives up at once -> zero records, sticky -EIO). */
int aof_debug_vote_deadline_ticks(aof_ctx *ctx, uint32_t ticks);

/* ---- host-buffer conveniences (what the C++ facade calls) ----
 * Synchronous: copy in, run the kernels above, copy out.  blocks/subdirs may be NULL. */
int aof_flow_pair_host(aof_ctx *ctx, const uint8_t *prev, const uint8_t *cur, aof_block *blocks,
                       uint8_t *subdirs, aof_flow *flow);
/* Streaming: the context keeps the previous frame on the device.  Returns 1 for
 * the first frame after create/reset (nothing to compare, *flow zeroed), 0 afterwards. */
int aof_stream_push_host(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow);
int aof_stream_reset(aof_ctx *ctx);
/* Opt-in resident form of the streaming entry point for small frames (the one-workgroup kernel's class:
 * 8x8 tiles, grids of 8..256 blocks, frames of at most 64 KB).  Most of the 14 us a call takes through a
 * replayed hipGraph is the launch; with on = 1 ONE workgroup stays on the device
 * between calls, polls a request word in pinned host memory, computes the pair exactly as the one-launch
 * kernel does and posts the 16-byte record and a completion word the host polls -- no launch per frame.
 * The kernel always ends by itself: after 50 ms without a request, after 200 ms in total (so nothing that
 * waits for the device to drain, e.g. a hipFree elsewhere in the process, waits longer), or when the
 * library stops it (aof_destroy, aof_set_*, this call with on = 0); the next call starts it again.  It runs
 * on a stream of its own at the highest stream priority, i.e. on a hardware queue that no normal-priority
 * stream of the process shares.
 * Every host-side wait of this path is bounded.  A request that is not answered within 250 ms of the
 * launch call's return (or of the request, when the kernel was already there) switches the mode off for the
 * context: the kernel is asked to leave and the call, like all later ones, takes the graph path (one line on
 * stderr, aof_stream_stats.resident_fallbacks, .last_report).  A kernel that does not leave within a second
 * either keeps its pinned buffers for good (they are leaked, the context continues on fresh ones and the call
 * returns 1 like the first call of a sequence; aof_destroy then frees no device memory at all).
 * Results are bit-identical in both forms.  on < 0 queries whether the kernel is on the device now. */
int aof_set_stream_resident(aof_ctx *ctx, int on);
/* Counters of the streaming entry point since aof_create (diagnostics; tests assert on them). */
typedef struct aof_stream_stats {
    uint64_t calls;              /* aof_stream_push_host calls that had a previous frame to compare with */
    uint64_t resident_served;    /* ... of them answered by the resident kernel */
    uint32_t resident_launches;  /* resident kernel instances started */
    uint32_t resident_fallbacks; /* requests it did not answer within 250 ms (resident mode switched off) */
    uint32_t resident_lost;      /* instances that did not leave within 1 s of being asked (buffers abandoned) */
    uint32_t tagged_slow;        /* graph path: tagged record not there within 2 ms (the stream was drained instead) */
    float launch_call_us_max;    /* longest hipLaunchKernelGGL call of a resident start (the first loads the code object) */
    float start_latency_us_max;  /* longest launch-return -> first-poll latency of a resident kernel, the host only
                                    spinning on pinned memory in between (no HIP call) */
    char last_report[320];       /* text of the last fallback report, "" if none */
} aof_stream_stats;
int aof_stream_get_stats(const aof_ctx *ctx, aof_stream_stats *out);
/* Fault injection for tests of the path above: resident kernel instances started from now on ignore the request to
 * leave (deaf = 1; they still go on their own 50 ms / 200 ms deadlines), and the library waits stop_wait_us for a
 * kernel to leave instead of one second (0 = the default).  Exercises the "kernel lost" branch: buffers abandoned,
 * the context continues on fresh ones, aof_destroy leaks instead of freeing. */
int aof_debug_resident_fault(aof_ctx *ctx, int deaf, uint32_t stop_wait_us);
/* The streaming entry point replays a captured hipGraph per call (H2D frame, kernels, the result
 * written into pinned host memory; for frames of at most 64 KB served by the one-workgroup kernel:
 * that ONE kernel reading the pinned frames in place and publishing the record with a tag the host
 * polls for, without waiting for the stream); this switches the capture off (1 = on, the default:
 * eager launches and a stream wait otherwise).  Returns whether a graph is currently instantiated
 * for the next call when on < 0 (query). */
int aof_set_stream_graph(aof_ctx *ctx, int on);

/* ---- frame ingest: the caller-side steps the reference runs on
 * the host right before calcFlow, moved next to the data so a full sensor frame is
 * uploaded once: centre crop to the engine's frame size with a contiguous copy
 * (/root/reference/src/mainloop.cpp:295-298,317-319) and the 10-bin histogram of the
 * centred 128x128 region of the cropped image that feeds the auto-exposure loop
 * (mainloop.cpp:203-214; EXPOSURE_MASK_SIZE :52).  Stateless. ---- */
#define AOF_EXPOSURE_BINS 10
#define AOF_EXPOSURE_MASK_SIZE 128
typedef struct aof_ingest_params {
    int32_t camera_width, camera_height; /* Y plane of the sensor frame, stride == width */
    int32_t crop_width, crop_height;     /* engine frame size (getImageWidth/Height) */
} aof_ingest_params;
/* d_camera: frame i at +i*camera_stride bytes.  d_cropped: [n][crop_h][crop_w], frame i at
 * +i*cropped_stride (pass it on to aof_flow_batch_device).  d_hist: uint32 [n][10] or NULL.
 * Either output may be NULL.  Asynchronous on `stream`. */
int aof_ingest_batch_device(const aof_ingest_params *p, const uint8_t *d_camera,
                            int64_t camera_stride, int64_t n_frames, uint8_t *d_cropped,
                            int64_t cropped_stride, uint32_t *d_hist, void *stream);
/* Mean sample value of one histogram, exactly as mainloop.cpp:216-220 computes it. */
float aof_exposure_msv(const uint32_t hist[AOF_EXPOSURE_BINS]);
/* Histogram bin (0..9) of a grey value, -1 if cv::calcHist would drop it (v == 255). */
int aof_exposure_bin(int grey);

/* ---- gyro de-rotation: the output-side neighbour of the path.
 * The reference integrates the gyro between flow outputs and ships it next to the flow
 * for the autopilot to de-rotate (/root/reference/src/mainloop.cpp:383-405, axis swap
 * :364-365: "-y gives x flow, x gives y flow").  This applies the published PX4Flow
 * compensation to a batch of flow records on the device:
 *   if |gy| > threshold*dt:  x' = clamp(flow_x + gy*focal_x, +-max_flow)   else x' = flow_x
 *   if |gx| > threshold*dt:  y' = clamp(flow_y - gx*focal_y, +-max_flow)   else y' = flow_y
 * with gx, gy the gyro angles (rad) integrated over the pair's interval dt (s). ---- */
typedef struct aof_gyro {
    float integ_x, integ_y, integ_z; /* rad, body rates integrated over the frame interval */
    float dt_s;                      /* the interval */
} aof_gyro;
typedef struct aof_derotate_params {
    float focal_x, focal_y;  /* px (main.cpp:60-61) */
    float max_flow;          /* clamp: search radius + 0.5 px */
    float rate_threshold;    /* rad/s below which an axis is left alone */
} aof_derotate_params;
/* d_out: float [n][2] compensated pixel flow.  Asynchronous on `stream`. */
int aof_derotate_batch_device(const aof_derotate_params *p, const aof_flow *d_flows,
                              const aof_gyro *d_gyro, int64_t n, float *d_out, void *stream);

/* ---- a recorded frame SEQUENCE as one device pipeline (the reference's per-frame loop,
 * /root/reference/src/mainloop.cpp:295-373, for all frames of a recording at once) ----
 * n sensor frames, their time stamps and the gyro integrated between them, all resident in device memory, go
 * through, in ONE call that only enqueues (no allocation, no host synchronisation: it can be captured):
 *   1. ingest: centre crop + exposure histogram per frame (aof_ingest_batch_device; mainloop.cpp:295-298,203-214)
 *   2. flow in sequence mode: frame k is `cur` of pair k-1 and `prev` of pair k (aof_flow_batch_device)
 *   3. the rate limiter of calcFlow (flows of frames with quality > 0 summed until
 *      (float)(t - t_last) > 1e6f / output_rate, u32 wrap-around arithmetic; mainloop.cpp:322-331)
 *   4. gyro de-rotation of every pair's pixel flow (aof_derotate_batch_device; an extra output: the message
 *      itself carries the flow and the gyro side by side, as the reference sends them)
 *   5. pixel flow -> angular flow (aof_flow_angle), the OPTICAL_FLOW_RAD field mapping (mainloop.cpp:359-371)
 *      and the MAVLink 2 frame (mavlink_tcp.cpp:142-162) of every published flow.
 * The records and frames are byte-identical to driving the C++ facade frame by frame over the same frames
 * (OpticalFlowOpenCV::calcFlow + fillOpticalFlowRad + packOpticalFlowRad): the float operations are the
 * host's, in the host's order.  The context's parameters must describe the CROPPED frame. */
typedef struct aof_sequence_params {
    aof_ingest_params ingest;        /* sensor frame -> crop; crop size == the context's frame size */
    float focal_x, focal_y;          /* px (main.cpp:60-61) */
    int32_t output_rate;             /* Hz (main.cpp:59); <= 0 publishes every frame */
    uint64_t offset_timestamp_usec;  /* vehicle time of the first frame (mainloop.cpp:360); 0 = not known yet:
                                        records are written, no frame is sent (mainloop.cpp:353-357) */
    uint8_t system_id, component_id; /* MAVLink ids (mavlink_tcp.h:66-67: 1, MAV_COMP_ID_CAMERA = 100) */
    uint8_t first_seq;               /* MAVLink sequence number of the first frame sent */
    uint8_t derotate;                /* 1: also write the de-rotated pixel flow of every pair */
    aof_derotate_params derotate_params;
} aof_sequence_params;

/* What calcFlow returned for a frame it published (quality >= 0), plus the gyro taken with it. */
typedef struct aof_seq_record {
    uint32_t frame;          /* index of the frame */
    int32_t quality;         /* 0..255 */
    int32_t dt_us;           /* integration time */
    float flow_x, flow_y;    /* rad */
    float gyro_x, gyro_y, gyro_z; /* rad, integrated since the previous record (before the axis switch) */
} aof_seq_record;

#define AOF_SEQ_FRAME_BYTES 56     /* room per MAVLink 2 frame: 10 + 44 + 2 */
#define AOF_SEQ_STATUS_STALLED 1u  /* more than 4096 frames in a row without reaching the output period: the
                                      time stamps do not advance; nothing is published behind that point */

/* Byte offsets of the pipeline's outputs and intermediates inside the caller's workspace. */
typedef struct aof_seq_layout {
    size_t total_bytes;
    size_t cropped;    /* u8 [n][crop_h][crop_w]: the frame sequence the flow runs on */
    size_t exposure;   /* u32 [n][10]: exposure histograms */
    size_t flows;      /* aof_flow [n-1]: pair k = frames k, k+1 */
    size_t derotated;  /* float [n-1][2] (params.derotate) */
    size_t count;      /* u32 [4]: records written, frames sent, AOF_SEQ_STATUS_* flags, 0 */
    size_t records;    /* aof_seq_record [n], the first count[0] valid, in frame order */
    size_t frames;     /* u8 [n][AOF_SEQ_FRAME_BYTES]: frame of record m at + 56 m */
    size_t frame_len;  /* u8 [n]: length of frame m (0: not sent) */
    size_t scratch;    /* limiter state and the flow engine's workspace */
} aof_seq_layout;
int aof_sequence_layout(const aof_params *p, const aof_sequence_params *sp, int64_t n_frames, aof_seq_layout *out);
/* d_camera: frame i at + i*camera_stride.  d_time_us: [n] frame times in microseconds relative to the first
 * frame (mainloop.cpp:305-311; calcFlow sees them truncated to 32 bits).  d_gyro: [n], entry k = gyro integrated
 * over the interval that ends at frame k (dt_s: that interval, for the de-rotation), or NULL (zeros).
 * d_workspace: >= aof_sequence_layout().total_bytes, 256-byte aligned. */
int aof_sequence_device(aof_ctx *ctx, const aof_sequence_params *sp, const uint8_t *d_camera, int64_t camera_stride,
                        int64_t n_frames, const uint64_t *d_time_us, const aof_gyro *d_gyro, void *d_workspace,
                        size_t workspace_bytes, void *stream);
/* Pixel flow -> angular flow (rad) exactly as the facade and the device pipeline compute it:
 * atan2(flow_px, focal_px) as a fixed sequence of IEEE double operations (include/aof_math.h).  Host only. */
float aof_flow_angle(float flow_px, float focal_px);

/* ---- measurement ----
 * With profiling on, every launch is bracketed by HIP events on the stream it
 * is launched on; the last AOF_PROFILE_RING launches of each kernel are kept.
 * aof_kernel_ms() synchronises on the newest pair and returns its duration;
 * aof_profile_count()/aof_profile_ms() walk the ring (index 0 = oldest kept)
 * so a benchmark can average a kernel over its whole timed region.
 * Turning profiling on resets the ring. */
#define AOF_PROFILE_RING 256
int aof_set_profiling(aof_ctx *ctx, int on);
/* Same, for a subset: bit k of `mask` times kernel id k (events cost a few microseconds of
 * stream serialisation each, so a benchmark times only the kernel it prices). */
int aof_set_profiling_mask(aof_ctx *ctx, uint32_t mask);
int aof_kernel_ms(aof_ctx *ctx, int kernel_id, float *ms);
int aof_profile_count(const aof_ctx *ctx, int kernel_id);
int aof_profile_ms(aof_ctx *ctx, int kernel_id, int index, float *ms);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
