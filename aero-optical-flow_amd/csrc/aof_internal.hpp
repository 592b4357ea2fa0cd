// Internal declarations shared by the host side of the C ABI and the kernel
// launchers.  Not installed; the public surface is include/aof.h.
#pragma once

#include <cstddef>
#include <cstdint>

#include "aof.h"

#if defined(__HIPCC__)
#define AOF_HD __host__ __device__
#else
#define AOF_HD
#endif

namespace aof {

// n / d for n < 2^31 as (umulhi(n, mul) + n) >> shift (round-up method of Granlund & Montgomery;
// with n < 2^31 the sum cannot overflow 32 bits).  Filled on the host, used by the lane-per-block
// kernels, where a run-time integer division costs ~20 VALU instructions per lane.
struct FastDiv {
    uint32_t mul, shift;
};
inline FastDiv fastdiv_make(uint32_t d)
{
    FastDiv f = {0u, 0u};
    if (d == 0) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.shift = l;
    return f;
}

struct Grid {
    int32_t x0, y0, step_x, step_y, nx, ny;
    AOF_HD int32_t blocks() const { return nx * ny; }
};

// Host-only parameter logic (aof_params.cpp).
int grid_for_level(const aof_params &p, int level, Grid *g);
int level_range(const aof_params &p, int level);  // histogram half-range R
int value_threshold_u16(const aof_params &p);     // SAD gate clamped to the u16 record
int reduce_chunks(int nblocks);                   // 0: one reduction workgroup per pair reads all records
size_t hist_bytes_per_pair(const aof_params &p, int level);  // vote-histogram scratch of one pair

// What turns a pair's vote histograms into its aof_flow (K3).
struct FlowTail {
    int32_t nblocks;
    int32_t range;             // R
    int32_t hist_filter;
    int32_t min_valid;
    aof_flow *flows;           // [n_pairs]
    const aof_flow *pred;      // copy pred_x/pred_y + PRED_VALID from here (level 0 of two), or nullptr
    int32_t emit_predictor;    // level 1: write the integer predictor into pred_x/pred_y
};

// The context's vote memory (zero at rest): one record per pair of the running launch -- word 0 the
// number of blocks that have arrived, then the two vote histograms (aof_reduce.hpp).
struct VoteMem {
    uint32_t *base;     // [pairs][stride] words
    uint32_t stride;    // words per pair, a multiple of 64 (256 bytes: pairs never share a line)
    uint32_t *fault;    // pinned host word of the context: a finaliser that gives up stores pair + 1 here
    uint32_t deadline_ticks;   // finaliser waves give up after this many ticks of the 100 MHz counter
};

// What the pruned lane8 search tells the host about itself (the ADAPTIVE mode of 8x8 contexts, aof_capi.hip):
// one workgroup in `stride` stores how many of its first wave's chunks left with "pruning pays" into a word of
// pinned host memory: tag (low 16 bits of launch_no) << 16 | paying << 8 | chunks.  Plain stores, nobody waits
// for them: the host reads whatever has arrived when it enqueues the next launch.  Speed only, never results.
constexpr int kPruneSlots = 256;
struct PruneReport {
    uint32_t *slots;      // pinned, kPruneSlots words; nullptr = no report
    uint32_t launch_no;
    uint32_t stride;      // (filled by the launcher)
    uint32_t expected;    // slots this launch writes (filled by the launcher)
};

// Everything one search launch needs; passed to the kernels by value.
struct SearchArgs {
    const uint8_t *prev;       // level frames of the older image, pair i at +i*stride
    const uint8_t *cur;        // level frames of the newer image
    int64_t pair_stride;       // bytes between consecutive pairs (both arrays)
    int32_t w, h;              // level frame size, row stride == w
    int32_t tile, search;      // B, S
    Grid grid;
    int32_t feature_threshold;
    int32_t value_threshold;   // clamped to <= 0xFFFF
    int32_t subpixel;
    aof_block *blocks;         // [n_pairs][grid.blocks()]
    uint8_t *subdirs;          // [n_pairs][grid.blocks()] or nullptr
    const aof_flow *pred;      // [n_pairs] level-1 results carrying the predictor, or nullptr
    const uint32_t *sums;      // [n_pairs][2][2] pixel sums, or nullptr when not equalising
    int32_t level;             // which sums column to use
    int64_t n_pairs;
    int32_t hist_range;        // R
    int32_t prune;             // 0 exhaustive, 1 exact partial-distortion elimination (lane8, tile16), 2 adaptive (tile16: per pair, by hints; lane8: the first chunk of a wave runs exhaustively and judges)
    uint32_t *hints;           // tile16, adaptive: [n_pairs] written by its probe kernel (1 = pruning pays), workspace
    FastDiv div_nb, div_nx;    // lane8: filled by its launchers (item -> pair, block -> row)
};

struct ReduceArgs {
    const aof_block *blocks;
    const uint8_t *subdirs;    // nullptr when half-pixel refinement is off
    int32_t value_threshold;
    FlowTail tail;
    int64_t n_pairs;
    const uint32_t *parts;     // per-chunk histograms of the first step of a two-step reduction (then blocks are not read)
    int32_t nstrips;           // chunks per pair
    uint32_t *chunk_parts;     // large grids: scratch for per-chunk histograms ([n_pairs][chunks][2][bins]), or nullptr
};

// The fused coarse kernel (k_coarse): K1 + level-1 search + level-1 reduction of one pair per
// workgroup, level-1 frames in LDS.
struct CoarseArgs {
    const uint8_t *prev, *cur; // level-0 frames
    int64_t pair_stride;
    int32_t w, h;              // level-0 frame size
    int32_t tile, search, subpixel;
    Grid grid;                 // level-1 block grid
    int32_t feature_threshold;
    int32_t value_threshold;   // clamped to <= 0xFFFF
    uint32_t *sums;            // [n_pairs][2][2] written here (no memset, no atomics), or nullptr
    aof_block *blocks;         // level-1 records [n_pairs][grid.blocks()]
    FlowTail tail;             // level-1 flows: the predictor
    int64_t n_pairs;
    FastDiv div_nb, div_nx, div_chunks;   // filled by the launcher
    int32_t first_generation;  // workgroups of the first generation = CUs of the device (set by the caller)
    int32_t stagger_groups, stagger_ticks;   // start-up stagger; stagger_groups == 0: the launcher chooses
    int32_t rows_per_sweep;    // launcher: level-1 rows one sweep of the workgroup's lanes covers
};

// The one-workgroup kernel for small pairs (k_flow_small): one or two levels of a pair whose frames
// fit LDS and whose grids have at most 256 blocks each.  Reads l0.prev / l0.cur only; l1.prev,
// l1.cur, the pred and sums pointers inside l0 / l1 are not used (level-1 frames, predictor and
// deltas stay on chip).
struct SmallArgs {
    SearchArgs l0, l1;         // l1 only for levels == 2
    FlowTail t0, t1;
    uint32_t *sums;            // [n_pairs][2][2] written here, or nullptr when not equalising
    int32_t levels;
};

struct PyramidArgs {
    const uint8_t *prev, *cur;
    int64_t pair_stride;
    int32_t w, h;
    uint8_t *l1_prev, *l1_cur; // [n_pairs][h/2][w/2] or nullptr (sums only)
    uint32_t *sums;            // [n_pairs][2][2] or nullptr (pyramid only)
    int64_t n_pairs;
    // A frame SEQUENCE (cur = prev + one frame, pair_stride = one frame): every frame is summed and filtered ONCE.
    // n_pairs then counts the FRAMES (pairs + 1), `prev` is the first frame, `cur` / `l1_cur` are not used,
    // l1_prev receives one level-1 frame per frame, and frame f's sums go to pair f (prev) and pair f - 1 (cur).
    int32_t sequence;
};

// Kernel launchers (one per .hip file).  All enqueue on `stream` and return the
// hipError_t of the launch as int (0 = ok).
int launch_pyramid(const PyramidArgs &a, void *stream);   // zeroes a.sums first
int launch_zero_words(uint32_t *words, int64_t count, void *stream);   // (a kernel: replays correctly from a hipGraph)
bool coarse_fused_supported(const CoarseArgs &a);
int launch_coarse_fused(const CoarseArgs &a, void *stream);
int launch_search_generic(const SearchArgs &a, void *stream);
// K2b: half-pixel refinement of records written by an integer search (tile 8 or 16; today only
// the 16x16 kernel needs it); fills a.subdirs.
int launch_refine(const SearchArgs &a, void *stream);
// Lane-per-block kernel straight from global memory for B=8, S=4 on ANY grid / width /
// predictor (sparse PX4Flow grid, rows that are no multiple of 16 bytes), half-pixel refinement
// included.
bool lane8_supported(const SearchArgs &a);
// tail + votes: the reduction runs inside the same launch (votes through agent-scope atomics into the
// context's vote memory, the last wave of a pair writes its flow record); no K3 follows.
bool lane8_votes_supported(const SearchArgs &a, const VoteMem &votes, int64_t capacity_pairs);
int launch_search_lane8(const SearchArgs &a, void *stream, const FlowTail *tail = nullptr,
                        const VoteMem *votes = nullptr, PruneReport *report = nullptr);
// 256-block chunks of the launch: the pruned kernels carry their hints from block to block of a wave and need a
// launch large enough for walks of three blocks before that pays (kPruneMinChunks below).
int64_t lane8_chunks(const SearchArgs &a);
// The pruned search on dense grids as a column walk (k_search_cols8.hip): a lane keeps the lower half of its window for
// the block below.  Same modes (a.prune 1 / 2), same report.
bool lane8_cols_supported(const SearchArgs &a);
// tail + votes: the reduction in the same launch (k_flow_lane8_cols), as launch_search_lane8 does for the exhaustive scan
bool lane8_cols_votes_supported(const SearchArgs &a, const VoteMem &votes, int64_t capacity_pairs);
int launch_search_lane8_cols(const SearchArgs &a, void *stream, PruneReport *report = nullptr, const FlowTail *tail = nullptr,
                             const VoteMem *votes = nullptr);
// Smallest launch (in 256-block chunks) the ADAPTIVE mode lets prune.  Round 5 sweep, bench.py --pairs N as it chooses
// itself (two batches in flight, graph replay), pruned column walk against the exhaustive kernel with the reduction in
// its launch, us per step (profiles/r05_small_launch_prune_sweep.txt): 32 pairs 14.7 / 11.0, 64: 18.6 / 14.8,
// 96: 22.2 / 20.4, 128: 23.0-24.5 / 26.0, 192: 26.6 / 38.0, 256: 33.8 / 49.7 -- a walk of two blocks pays for its vote
// and its second round of row loads with too little; from three blocks on (112 VGA pairs) it wins.
constexpr int64_t kPruneMinChunks = 2048;
// Grids of 8..256 blocks: a workgroup owns whole pairs and also writes their flow records (no K3).
int lane8_group(const SearchArgs &a);  // pairs per workgroup, 0 = not applicable
int launch_flow_lane8(const SearchArgs &a, const FlowTail &tail, void *stream);
// LDS-tiled (block, dy)-per-lane kernel for B=16, S=8 on a dense grid (any predictor).
bool tile16_supported(const SearchArgs &a);
bool tile16_refines(const SearchArgs &a);   // the launch also writes the half-pixel directions (no K2b behind it)
int launch_search_tile16(const SearchArgs &a, void *stream);
// Small pairs (frames fit LDS, grids <= 256 blocks), one or two levels, in one launch: one workgroup per pair.
bool flow_small_supported(const SmallArgs &a);
int launch_flow_small(const SmallArgs &a, void *stream);
// One pair, the record published in pinned host memory by ONE tagged 16-byte store (top byte of `count` =
// low byte of *tag, which the host wrote before the launch): the per-call path polls for it.
int launch_flow_small_tagged(const SmallArgs &a, aof_flow *host_record, const uint32_t *tag, void *stream);
// The resident form of the same kernel for the per-call path: one workgroup that stays on the device and
// serves requests posted through a mailbox in pinned host memory (k_flow_small.hip).
struct ResidentBox {
    // host -> device, ONE 64-bit word (one PCIe read per poll): bits 0..31 the number of the newest request
    // (never 0), bit 32 which of the two pinned frames is the newest (the other is its predecessor), bits
    // 33..62 the low 30 bits of the request at which that OTHER frame was posted + 1 (0 = not through a
    // request), bit 63 = leave now
    unsigned long long word;
    uint32_t pad0[14];
    uint32_t done;      // device: number of the last request served (its record is in place)
    uint32_t running;   // host sets 1 before the launch, the device 0 when it leaves (the host never clears it)
    uint32_t exited;    // device: `done` at the moment it left
    uint32_t started;   // device: launch number of the instance that last reached its polling loop (diagnostics:
                        // tells "never got onto the device" from "on the device but not answering")
    uint32_t pad1[12];
};
static_assert(sizeof(ResidentBox) == 128, "two cache lines: host -> device, device -> host");
constexpr unsigned long long kResidentStopBit = 1ull << 63;
inline unsigned long long resident_word(uint32_t request, int slot, uint32_t prev_request)
{
    const unsigned long long prev = prev_request ? (unsigned long long)(prev_request & 0x3FFFFFFFu) + 1ull : 0ull;
    return (unsigned long long)request | ((unsigned long long)(slot & 1) << 32) | (prev << 33);
}
// host_record: the pinned 16-byte record the host polls (top byte of `count` = low byte of the request).
int launch_flow_resident(const SmallArgs &a, ResidentBox *box, aof_flow *host_record, const uint8_t *frame_a,
                         const uint8_t *frame_b, uint32_t served, uint32_t launch_no, uint64_t idle_ticks,
                         uint64_t life_ticks, bool deaf, void *stream);
// The output side of a frame sequence (k_sequence.hip): rate limiter, gyro sums, angles, OPTICAL_FLOW_RAD frames.
constexpr int64_t kLimitScanFrames = 4096;   // a publication must come within this many frames of the previous one
struct SequenceArgs {
    int64_t n_frames;
    const uint64_t *time_us;       // [n_frames] frame times relative to the first frame (mainloop.cpp:305-311)
    const aof_gyro *gyro;          // [n_frames] gyro integrated over the interval that ends at frame k, or nullptr
    const aof_flow *flows;         // [n_frames - 1]: pair k = frames k, k + 1
    int32_t output_rate;
    float period_us;               // 1e6f / output_rate, divided on the host as the facade divides
    float focal_x, focal_y;
    uint64_t offset_timestamp_usec;
    uint8_t system_id, component_id, first_seq;
    uint32_t *jump[2], *hops[2];   // [n_frames + 1] each: pointer doubling, ping-pong
    uint8_t *reached;              // [n_frames + 1]: 0, or the round in which the node was marked + 1
    uint32_t *rank;                // [n_frames + 1]: position of a marked node on the chain = its message index
    uint32_t *count;               // [2]: records written, frames sent
    uint32_t *status;              // [1]: AOF_SEQ_STATUS_* flags (zeroed by the caller of the launcher)
    aof_seq_record *records;       // [n_frames]
    uint8_t *frames;               // [n_frames][AOF_SEQ_FRAME_BYTES], message m at + 56 m; or nullptr
    uint8_t *frame_len;            // [n_frames]
};
inline int sequence_rounds(int64_t n_frames)   // smallest R with 4^R > n_frames (the chain has at most that many hops)
{
    int r = 0;
    while ((1ll << (2 * r)) <= n_frames) r++;
    return r;
}
int launch_sequence_output(const SequenceArgs &a, void *stream);
// (aof_capi.hip) sticky fault / wedged state and current-device check of a context, before anything is enqueued; and
// aof_last_error's text for the callers outside aof_capi.hip
int precheck(aof_ctx *ctx);
int ctx_fail(aof_ctx *ctx, int code, const char *what);
// (aof_capi.hip) the flow of a frame sequence for the pipeline: does the call run K1 as a pass of its own, and the
// call itself with K1's outputs already in the workspace
bool sequence_runs_k1(aof_ctx *ctx, const uint8_t *d_frames, int64_t n_pairs, void *d_workspace);
int flow_sequence(aof_ctx *ctx, const uint8_t *d_frames, int64_t n_pairs, aof_flow *d_flows, void *d_workspace,
                  size_t workspace_bytes, void *stream, bool k1_ready);
int launch_reduce(const ReduceArgs &a, void *stream);
int launch_derotate(const aof_derotate_params &p, const aof_flow *flows, const aof_gyro *gyro,
                    int64_t n, float *out, void *stream);
int launch_ingest(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride,
                  int64_t n_frames, uint8_t *cropped, int64_t cropped_stride, uint32_t *hist,
                  void *stream);
// The sequence pipeline's ingest: the same pass also leaves every frame's level-1 image (l1: [n_frames] frames, or
// nullptr) and adds its byte sums to the pixel-sum records of the pairs it belongs to (sums: [n_frames - 1][2][2],
// zeroed here; or nullptr) -- what K1 would otherwise compute in a second pass over the cropped frames.
bool ingest_pyramid_supported(const aof_ingest_params &p, const uint8_t *cropped, int64_t cropped_stride);
int launch_ingest_pyramid(const aof_ingest_params &p, const uint8_t *camera, int64_t camera_stride, int64_t n_frames,
                          uint8_t *cropped, int64_t cropped_stride, uint32_t *hist, uint8_t *l1, uint32_t *sums, void *stream);

}  // namespace aof
