// Host side of the C ABI (include/aof.h): context, launch sequencing, the
// host-buffer conveniences the C++ facade uses, and event-based kernel timing.
//
// Launch sequence of aof_flow_batch_device (DESIGN.md "Kernels"):
//   1 level :                      K2 search(L0) -> K3 reduce
//   + mean  : K1 (zeroes and fills the pixel sums) -> K2 search(L0) -> K3 reduce
//   2 levels: k_coarse (sums, pyramid, level-1 search and predictor of a pair in one workgroup)
//             or K1 -> K2 search(L1) -> K3 reduce(L1: predictor),
//             then K2 search(L0, shifted by predictor) -> K3 reduce
//   small pairs (<= 128 per call, frames that fit LDS): k_flow_small, everything in one launch
// Everything is enqueued on the caller's stream; nothing allocates or
// synchronises, so the sequence can be captured into a hipGraph.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

#include "aof_internal.hpp"

using namespace aof;

struct aof_ctx {
    aof_params params;
    Grid g0, g1;
    int device;
    int cus;                  // compute units of `device`
    bool force_generic;
    bool profiling;
    uint32_t profile_mask;
    int search_mode;
    char err[256];
    hipEvent_t (*ev)[AOF_PROFILE_RING][2];  // [AOF_K_COUNT][ring][start,stop], created on demand
    int64_t ev_count[AOF_K_COUNT];           // launches timed since profiling was switched on
    // host-convenience state (one pair)
    hipStream_t stream;
    uint8_t *d_frames[2];  // ping-pong: previous / current frame (streaming entry point)
    uint8_t *d_pair[2];    // scratch of the stateless two-frame entry point
    int cur_slot;          // slot holding the newest frame
    bool have_prev;
    bool host_ready;       // everything below exists (ensure_host_state)
    bool host_dirty;       // part of it was abandoned to a lost resident kernel: rebuild before the next use
    aof_block *d_blocks;
    uint8_t *d_subdirs;
    aof_flow *d_flow;
    void *d_ws;
    size_t ws_bytes;
    // streaming entry point as two captured hipGraphs (one per ping-pong slot):
    // H2D of the pinned frame -> kernels -> D2H of the 16-byte result, one launch per call
    uint8_t *h_frame;           // pinned staging copy of the caller's frame
    uint8_t *h_frames[2];       // small frames: pinned ping-pong frames the kernels read in place
    bool zero_copy;             // (no H2D copy: a 64x64 frame is 4 KB over PCIe)
    aof_flow *h_flow;           // pinned result
    uint32_t *h_tag;            // pinned (same allocation, its own cache line): the tag the next tagged record carries
    hipGraphExec_t push_graph[2];
    bool push_tagged[2];        // the slot's graph publishes a tagged record: the host polls for it, no stream wait
    bool graph_disabled;        // capture failed once: stay on the plain path
    bool capturing;
    bool split_coarse;          // run K1 / level-1 search / level-1 reduce as separate kernels
    bool k1_ready;              // (set around one call by the sequence pipeline) K1's outputs are in the workspace already
    // resident form of the per-call path (aof_set_stream_resident): one workgroup stays on the device
    // and serves aof_stream_push_host through a mailbox in pinned memory
    bool resident_on;
    bool resident_lost;         // a resident kernel did not leave when asked: nothing it may touch is ever freed
    bool wedged;                // a bounded wait for the device ran out: every later call fails, destroy frees nothing
    hipStream_t rstream;        // the resident kernel's own stream (highest priority: its own pool of hardware queues)
    ResidentBox *box;           // pinned, device-visible
    uint32_t rseq;              // number of the last request posted
    uint32_t rlaunches;         // resident kernel instances started on `box`
    bool rdeaf;                 // fault injection (aof_debug_resident_fault): the next instances ignore the stop bit
    double rstop_wait_s;        // how long resident_stop waits for the exit flag (1 s; the fault injection shortens it)
    uint32_t rframe_req[2];     // request at which pinned frame b was posted as the newest frame, 0 = written otherwise
    aof_stream_stats stats;     // aof_stream_get_stats
    // device -> host fault word (pinned, its own allocation): a kernel that gave up on a device-side wait
    // stores a non-zero code here; every entry point that enqueues work looks at it first
    uint32_t *h_fault;
    // ADAPTIVE search of 8x8 contexts (run_search): what the pruned kernel's last reporting launch said, in
    // pinned host words behind h_fault (same allocation), and what the context does with it
    uint32_t *h_prune_slots;    // kPruneSlots words
    uint32_t prune_launch_no;   // number of the last reporting launch (its low 16 bits tag the words)
    uint32_t prune_expected;    // words that launch writes, 0 = none yet
    int prune_belief;           // -1 nothing known yet, 0 pruning does not pay on this context's images, 1 it does
    int prune_since_probe;      // exhaustive launches since the last look
    aof_search_stats search_stats;
    uint32_t vote_deadline_ticks;   // finaliser waves of the in-launch reduction give up after this (100 MHz ticks)
    bool votes_captured;        // a captured graph holds an in-launch reduction: eager launches keep to K3
    // reduction inside the flat lane8 search (no K3 launch): the pairs' vote records, zero at rest
    uint32_t *d_votes;
    int64_t votes_pairs;        // records allocated
    bool separate_reduce;       // aof_set_reduce_fusion(ctx, 0): always launch K3
    hipEvent_t votes_done;      // recorded behind every launch that uses d_votes
    hipStream_t votes_stream;   // stream of that launch
    bool votes_used;
};

// Vote records per context (launches of more pairs keep K3).  2 048 finaliser waves are at most 256 per XCD --
// half of an XCD's wave slots at the search kernel's occupancy -- so the search workgroups of ANOTHER
// context's launch always find room beside them: two in-launch reductions in flight cannot wait for each
// other (they could from 4 096 pairs on, until the deadline).
constexpr int64_t kVotePairs = 2048;
constexpr uint32_t kVoteStride = 128;            // words per record: 1 + 2 * 55 bins at the most (R = 13)
constexpr uint32_t kVoteDeadlineTicks = 5000000; // 50 ms of the 100 MHz counter (aof_set_vote_deadline_us)

namespace {

// The C ABI must not leave the calling thread on another HIP device than it found it on.
struct DeviceGuard {
    int prev;
    explicit DeviceGuard(int device) : prev(-1)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void)hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int fail(aof_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Waits until everything enqueued on `s` has completed, for at most `seconds`: hipStreamQuery never blocks,
// hipStreamSynchronize has no time limit of its own.  hipErrorNotReady = the time ran out.
hipError_t drain_bounded(hipStream_t s, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        const double t = seconds_since(t0);
        if (t > seconds) return hipErrorNotReady;
        if (t > 200e-6) std::this_thread::sleep_for(std::chrono::microseconds(t > 5e-3 ? 500 : 20));
    }
}

hipError_t event_wait_bounded(hipEvent_t ev, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (seconds_since(t0) > seconds) return hipErrorNotReady;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

constexpr double kDrainS = 2.0;   // bounded waits for a stream of this library's kernels (each runs microseconds to milliseconds)

// The device did not finish within a bounded wait, or reported a fault: the context stays unusable
// (recovery = a new context) and aof_destroy frees nothing a kernel might still touch.
int wedge(aof_ctx *ctx, const char *what, hipError_t e)
{
    ctx->wedged = true;
    std::fprintf(stderr, "aof: %s: %s -- the context is disabled\n", what,
                 e == hipErrorNotReady ? "the device did not finish within the time limit" : hipGetErrorString(e));
    return fail(ctx, e == hipErrorNotReady ? -ETIMEDOUT : -EIO, "%s: %s", what,
                e == hipErrorNotReady ? "the device did not finish within the time limit" : hipGetErrorString(e));
}

// Sticky device-side condition of the context, checked by every entry point that enqueues work: a bounded
// wait that ran out earlier, or the fault word a kernel raised (a finaliser wave of the in-launch reduction
// that gave up on its pair: that pair's record says quality 0, flags 0).  Like a sticky HIP error, it stays.
int sticky_error(aof_ctx *ctx)
{
    if (ctx->wedged) return -EIO;   // (ctx->err still holds the text of the first report)
    if (ctx->h_fault) {
        const uint32_t code = __atomic_load_n(ctx->h_fault, __ATOMIC_ACQUIRE);
        if (code)
            return fail(ctx, -EIO, "a reduction inside a search launch gave up waiting for the votes of pair %u of its "
                                   "launch (device-side deadline): that record carries quality 0 and no valid flag; "
                                   "the context's vote memory is no longer trusted -- create a new context", code - 1);
    }
    return 0;
}

#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(ctx, -EIO, "%s: %s", #expr, hipGetErrorString(e_));             \
    } while (0)

struct Timed {
    aof_ctx *ctx; int id; hipStream_t s; int slot;
    Timed(aof_ctx *c, int k, hipStream_t st) : ctx(c), id(k), s(st), slot(0)
    {
        if (!ctx->profiling || ctx->capturing || !((ctx->profile_mask >> id) & 1u)) return;
        slot = (int)(ctx->ev_count[id] % AOF_PROFILE_RING);
        (void)hipEventRecord(ctx->ev[id][slot][0], s);
    }
    ~Timed()
    {
        if (!ctx->profiling || ctx->capturing || !((ctx->profile_mask >> id) & 1u)) return;
        (void)hipEventRecord(ctx->ev[id][slot][1], s);
        ctx->ev_count[id]++;
    }
};

SearchArgs search_args(const aof_ctx *ctx, int level, const uint8_t *prev, const uint8_t *cur,
                       int64_t stride, aof_block *blocks, uint8_t *subdirs, const aof_flow *pred,
                       const uint32_t *sums, int64_t n)
{
    const aof_params &p = ctx->params;
    SearchArgs a;
    a.prev = prev; a.cur = cur; a.pair_stride = stride;
    a.w = p.width >> level; a.h = p.height >> level;
    a.tile = p.tile; a.search = p.search;
    a.grid = level ? ctx->g1 : ctx->g0;
    a.feature_threshold = p.feature_threshold;
    a.value_threshold = value_threshold_u16(p);
    a.subpixel = p.subpixel;
    a.blocks = blocks; a.subdirs = p.subpixel ? subdirs : nullptr;
    a.pred = pred; a.sums = sums; a.level = level; a.n_pairs = n;
    a.hist_range = level_range(p, level);
    // 0 exhaustive, 1 pruned, 2 adaptive (16x16: a probe kernel judges every pair first; flat 8x8 launches:
    // run_search decides per launch from what the context's previous launches reported)
    a.prune = ctx->search_mode;
    a.hints = nullptr;   // (16x16 searches: set from the workspace by enqueue_coarse / enqueue_fine)
    return a;
}

constexpr int64_t kSmallMaxPairs = 128;   // one-launch path for small pairs: measured faster than the separate kernels up to here

enum SearchKind { SK_TILE16, SK_LANE8_GROUP, SK_LANE8, SK_GENERIC };

// Which search kernel serves these arguments (run_search, enqueue_level and aof_search_variant agree).
SearchKind search_kind(const aof_ctx *ctx, const SearchArgs &a)
{
    if (ctx->force_generic) return SK_GENERIC;
    // 8x8 tiles run lane-per-block straight from L2 (measured faster than LDS-staged strips on every
    // dense configuration: full lane use, no staging phases, no barriers)
    if (tile16_supported(a)) return SK_TILE16;
    if (a.prune) {   // the pruned steps' tables (68 B per block column) may not fit LDS where the exhaustive tile does: widths of ~3 000 px
        SearchArgs x = a;
        x.prune = 0;
        if (tile16_supported(x)) return SK_TILE16;
    }
    if (lane8_supported(a)) return lane8_group(a) > 0 ? SK_LANE8_GROUP : SK_LANE8;
    return SK_GENERIC;
}

// ADAPTIVE search of the flat 8x8 kernel: does THIS launch run the pruned kernel?  The exhaustive kernel is the
// faster one wherever nothing can be pruned (the pruned kernels' own exhaustive path costs 6-10 % more: their shape,
// blocks walked in sequence), and a probe in front of every launch would cost more than it saves at
// 0.2 us per pair -- so the context goes by what its PREVIOUS launches found: the pruned kernel reports how many
// of its chunks left with "pruning pays" (PruneReport, plain stores into pinned memory that nobody waits for), the
// context keeps using it while at least kPayingPct of them did, and otherwise runs the exhaustive kernel, with one
// pruned launch in kProbeEvery to look again.  Speed only: every kernel writes the same records.
constexpr uint32_t kPayingPct = 40;
constexpr int kProbeEvery = 16;

bool adaptive_lane8_prunes(aof_ctx *ctx, const SearchArgs &a)
{
    // level-1 searches and small launches: too few blocks per wave to carry a hint along
    if (a.level != 0 || !ctx->h_prune_slots || lane8_chunks(a) < kPruneMinChunks) return false;
    if (ctx->prune_expected) {
        const uint32_t tag = ctx->prune_launch_no & 0xFFFFu;
        uint32_t arrived = 0, paying = 0, seen = 0;
        for (uint32_t i = 0; i < ctx->prune_expected && i < (uint32_t)kPruneSlots; i++) {
            const uint32_t w = __atomic_load_n(ctx->h_prune_slots + i, __ATOMIC_RELAXED);
            if ((w >> 16) != tag) continue;
            arrived++;
            paying += (w >> 8) & 0xFFu;
            seen += w & 0xFFu;
        }
        if (arrived * 4 >= ctx->prune_expected && seen) {   // (a launch still running has told enough after a quarter)
            ctx->search_stats.paying_pct = (int32_t)(paying * 100u / seen);
            ctx->prune_belief = paying * 100u >= seen * kPayingPct ? 1 : 0;
            ctx->search_stats.belief = ctx->prune_belief;
            ctx->search_stats.reports_read++;
        }
    }
    if (ctx->prune_belief != 0) return true;
    if (++ctx->prune_since_probe >= kProbeEvery) {
        ctx->prune_since_probe = 0;
        return true;
    }
    return false;
}

// Runs the level's search.  *reduced: the search kernel also wrote the pairs' flow records (grouped
// lane8, flat lane8 with the reduction in its launch), no K3 follows.
int run_search(aof_ctx *ctx, SearchArgs a, const FlowTail &tail, bool *reduced, hipStream_t s)
{
    int rc;
    *reduced = false;
    switch (search_kind(ctx, a)) {
    case SK_TILE16:
        if (a.prune && !tile16_supported(a)) a.prune = 0;   // (search_kind: only the exhaustive tile fits LDS at this width)
        rc = launch_search_tile16(a, s);   // (refines out of its LDS tile when directions are wanted)
        if (!rc && a.subpixel && !tile16_refines(a)) rc = launch_refine(a, s);
        break;
    case SK_LANE8_GROUP:   // refines in the same lane and finalises the flow records
        rc = launch_flow_lane8(a, tail, s);
        *reduced = true;
        break;
    case SK_LANE8: {
        PruneReport rep = {nullptr, 0, 1, 0};
        if (a.prune) {
            if (ctx->search_mode == AOF_SEARCH_ADAPTIVE) {
                // (where the caller switched the in-launch reduction on, launches that do not prune -- too small, or
                //  images on which it does not pay -- still get it: that kernel is the exhaustive one.  256 VGA pairs,
                //  two batches in flight: pruned + K3 40.9 us, exhaustive with the reduction in the launch 49.9 us)
                if (!adaptive_lane8_prunes(ctx, a)) {
                    a.prune = 0;
                    ctx->search_stats.exhaustive_launches++;
                } else {
                    // a context that knows pruning pays starts every wave in the pruned code (optimistic, like
                    // PRUNED); one that does not lets the first block (chunk) of every wave run exhaustively and judge
                    a.prune = ctx->prune_belief == 1 ? 1 : 2;
                    if (++ctx->prune_launch_no % 0x10000u == 0) ctx->prune_launch_no++;   // (tag 0 = never written)
                    rep.slots = ctx->h_prune_slots;
                    rep.launch_no = ctx->prune_launch_no & 0xFFFFu;
                    ctx->search_stats.pruned_launches++;
                }
            }
        }
        // (dense grids prune as a column walk, whose lanes keep half of their window for the block below)
        const bool cols = a.prune && lane8_cols_supported(a);
        auto plain = [&]() -> int {   // K3 follows
            if (cols) return launch_search_lane8_cols(a, s, &rep);
            return a.prune ? launch_search_lane8(a, s, nullptr, nullptr, &rep) : launch_search_lane8(a, s);
        };
        // search + reduction in one launch when the context's vote memory can serve it (the exhaustive flat kernel and
        // the column walk have that form; the chunk-walking pruned kernel has not); launches on another stream than the
        // last one wait for that one first (the records are shared)
        const VoteMem vm = {ctx->d_votes, kVoteStride, ctx->h_fault, ctx->vote_deadline_ticks};
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool votes_fit = cols ? lane8_cols_votes_supported(a, vm, ctx->votes_pairs)
                                    : (!a.prune && lane8_votes_supported(a, vm, ctx->votes_pairs));
        const bool eager = hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
        // A captured graph that holds an in-launch reduction is replayed whenever its owner likes and records
        // no votes_done event: eager launches on this context keep to the separate K3 from then on, so that
        // the library never puts a second user on the vote records behind the graph's back.
        if (ctx->separate_reduce || ctx->force_generic || !votes_fit || (eager && ctx->votes_captured)) {
            rc = plain();
        } else {
            if (eager && ctx->votes_used && ctx->votes_stream != s &&
                hipStreamWaitEvent(s, ctx->votes_done, 0) != hipSuccess)
                return fail(ctx, -EIO, "cannot order the launch behind the context's previous one");
            rc = cols ? launch_search_lane8_cols(a, s, &rep, &tail, &vm) : launch_search_lane8(a, s, &tail, &vm);
            if (!rc && eager) {
                rc = (int)hipEventRecord(ctx->votes_done, s);
                ctx->votes_stream = s;
                ctx->votes_used = true;
            }
            if (!rc && !eager) ctx->votes_captured = true;   // (replays are the owner's to order: include/aof.h)
            *reduced = true;
        }
        if (a.prune && rep.slots) ctx->prune_expected = rep.expected;
        break;
    }
    default:
        rc = launch_search_generic(a, s);
    }
    if (rc) return fail(ctx, -EIO, "search launch: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

}  // namespace

namespace {

// Device views of one batch: frames, outputs and the workspace regions (all [n_pairs]-major).
struct BatchView {
    const uint8_t *prev, *cur;
    int64_t stride;
    uint32_t *sums;
    uint8_t *l1_prev, *l1_cur;
    aof_block *blocks1; uint8_t *subdirs1; aof_flow *flows1; uint8_t *hist1;
    aof_block *blocks0; uint8_t *subdirs0; aof_flow *flows; uint8_t *hist0;
    uint32_t *hints;   // 16x16 adaptive search: per-pair verdicts (both levels use it, one after the other)
};

FlowTail flow_tail(const aof_ctx *ctx, int level, aof_flow *flows, const aof_flow *pred)
{
    const aof_params &p = ctx->params;
    FlowTail t;
    t.nblocks = (level ? ctx->g1 : ctx->g0).blocks(); t.range = level_range(p, level);
    t.hist_filter = p.hist_filter; t.min_valid = p.min_valid;
    t.flows = flows; t.pred = pred; t.emit_predictor = level ? 1 : 0;
    return t;
}

// One level of pairs [first, first+n): search, then K3 unless the search kernel reduced itself.
int enqueue_level(aof_ctx *ctx, int level, SearchArgs a, const FlowTail &tail, uint8_t *hist, int kid_search,
                  int kid_reduce, hipStream_t s)
{
    int rc;
    bool reduced = false;
    {
        Timed t(ctx, kid_search, s);
        rc = run_search(ctx, a, tail, &reduced, s);
        if (rc) return rc;
    }
    if (reduced) return 0;
    ReduceArgs r;
    r.parts = nullptr; r.nstrips = 0;
    r.blocks = a.blocks; r.subdirs = a.subdirs;
    r.value_threshold = value_threshold_u16(ctx->params);
    r.tail = tail; r.n_pairs = a.n_pairs;
    r.chunk_parts = reinterpret_cast<uint32_t *>(hist);
    Timed t(ctx, kid_reduce, s);
    rc = launch_reduce(r, s);
    if (rc) return fail(ctx, -EIO, "reduce launch: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

// Arguments of the fused coarse kernel for pairs [first, first + n).
CoarseArgs coarse_args(const aof_ctx *ctx, const BatchView &v, int64_t first, int64_t n, uint32_t *sums)
{
    const aof_params &p = ctx->params;
    CoarseArgs c;
    c.prev = v.prev + first * v.stride; c.cur = v.cur + first * v.stride; c.pair_stride = v.stride;
    c.w = p.width; c.h = p.height; c.tile = p.tile; c.search = p.search; c.subpixel = p.subpixel;
    c.grid = ctx->g1;
    c.feature_threshold = p.feature_threshold;
    c.value_threshold = value_threshold_u16(p);
    c.sums = sums;
    c.blocks = v.blocks1 + first * ctx->g1.blocks();
    c.tail = flow_tail(ctx, 1, v.flows1 + first, nullptr);
    c.n_pairs = n;
    c.first_generation = ctx->cus; c.stagger_groups = 0; c.stagger_ticks = 0;   // stagger chosen by the launcher
    return c;
}

// Coarse passes of pairs [first, first+n): pixel sums, level-1 frames, level-1 search and its
// reduction (the predictor).  Nothing to do for one level without equalisation.
int enqueue_coarse(aof_ctx *ctx, const BatchView &v, int64_t first, int64_t n, hipStream_t s)
{
    const aof_params &p = ctx->params;
    const bool two = p.pyramid_levels == 2, eq = p.mean_subtract != 0;
    if (!two && !eq) return 0;
    const int64_t l1_frame = (int64_t)(p.width / 2) * (p.height / 2);
    uint32_t *sums = v.sums ? v.sums + first * 4 : nullptr;
    if (two && !ctx->force_generic && !ctx->split_coarse) {
        // K1C: sums, pyramid, level-1 search and predictor of a pair in one workgroup, the
        // level-1 frames never leave LDS (workspace regions l1_prev / l1_cur stay untouched)
        const CoarseArgs c = coarse_args(ctx, v, first, n, sums);
        if (coarse_fused_supported(c)) {
            Timed t(ctx, AOF_K_PYRAMID, s);
            const int rc = launch_coarse_fused(c, s);
            if (rc) return fail(ctx, -EIO, "coarse launch: %s", hipGetErrorString((hipError_t)rc));
            return 0;
        }
    }
    PyramidArgs a;
    a.prev = v.prev + first * v.stride; a.cur = v.cur + first * v.stride; a.pair_stride = v.stride;
    a.w = p.width; a.h = p.height;
    a.l1_prev = two ? v.l1_prev + first * l1_frame : nullptr;
    a.l1_cur = two ? v.l1_cur + first * l1_frame : nullptr;
    a.sums = sums; a.n_pairs = n;
    // A frame sequence (aof.h: d_cur = d_prev + one frame, pair_stride = one frame): frame k is cur of pair k-1
    // and prev of pair k -- K1 sums and filters every frame once instead of twice.  The level-1 frames then
    // form a sequence of their own (n + 1 frames from the start of the workspace's two level-1 regions, which
    // are adjacent: 2 n frames of room), which the level-1 search views twice the same way.
    const bool sequence = v.cur == v.prev + (int64_t)p.width * p.height && v.stride == (int64_t)p.width * p.height &&
                          first == 0 && v.l1_cur >= v.l1_prev;
    if (sequence) {
        a.sequence = 1;
        a.n_pairs = n + 1;
        a.cur = nullptr; a.l1_cur = nullptr;
    } else {
        a.sequence = 0;
    }
    if (!(sequence && ctx->k1_ready)) {   // (the sequence pipeline's ingest has left sums and level-1 frames already)
        Timed t(ctx, AOF_K_PYRAMID, s);
        const int rc = launch_pyramid(a, s);
        if (rc) return fail(ctx, -EIO, "pyramid launch: %s", hipGetErrorString((hipError_t)rc));
    }
    if (!two) return 0;
    const int64_t nb1 = ctx->g1.blocks();
    SearchArgs sa = search_args(ctx, 1, a.l1_prev, sequence ? a.l1_prev + l1_frame : a.l1_cur, l1_frame, v.blocks1 + first * nb1,
                                v.subdirs1 ? v.subdirs1 + first * nb1 : nullptr, nullptr, sums, n);
    sa.hints = v.hints ? v.hints + first : nullptr;
    return enqueue_level(ctx, 1, sa, flow_tail(ctx, 1, v.flows1 + first, nullptr),
                         v.hist1 + (size_t)first * hist_bytes_per_pair(p, 1), AOF_K_SEARCH_L1, AOF_K_REDUCE_L1, s);
}

// Level-0 search (under the level-1 predictor when there is one) and the final reduction.
int enqueue_fine(aof_ctx *ctx, const BatchView &v, int64_t first, int64_t n, hipStream_t s)
{
    const aof_params &p = ctx->params;
    const bool two = p.pyramid_levels == 2;
    const int64_t nb0 = ctx->g0.blocks();
    const aof_flow *pred = two ? v.flows1 + first : nullptr;
    SearchArgs sa = search_args(ctx, 0, v.prev + first * v.stride, v.cur + first * v.stride, v.stride,
                                v.blocks0 + first * nb0, v.subdirs0 ? v.subdirs0 + first * nb0 : nullptr, pred,
                                v.sums ? v.sums + first * 4 : nullptr, n);
    sa.hints = v.hints ? v.hints + first : nullptr;
    return enqueue_level(ctx, 0, sa, flow_tail(ctx, 0, v.flows + first, pred),
                         v.hist0 + (size_t)first * hist_bytes_per_pair(p, 0), AOF_K_SEARCH, AOF_K_REDUCE, s);
}

// Views of one batch inside the caller's buffers and workspace.
BatchView batch_view(const aof_ctx *ctx, const aof_ws_layout &L, const uint8_t *d_prev, const uint8_t *d_cur,
                     int64_t pair_stride, aof_block *d_blocks, uint8_t *d_subdirs, aof_flow *d_flows, void *d_workspace)
{
    const aof_params &p = ctx->params;
    BatchView v;
    uint8_t *ws = static_cast<uint8_t *>(d_workspace);
    const bool two = p.pyramid_levels == 2, eq = p.mean_subtract != 0;
    v.prev = d_prev; v.cur = d_cur; v.stride = pair_stride;
    v.sums = eq ? reinterpret_cast<uint32_t *>(ws + L.sums) : nullptr;
    v.l1_prev = two ? ws + L.l1_prev : nullptr;
    v.l1_cur = two ? ws + L.l1_cur : nullptr;
    v.blocks1 = reinterpret_cast<aof_block *>(ws + L.l1_blocks);
    v.subdirs1 = p.subpixel ? ws + L.l1_subdirs : nullptr;
    v.flows1 = reinterpret_cast<aof_flow *>(ws + L.l1_flows);
    v.hist1 = ws + L.l1_hist;
    v.blocks0 = d_blocks ? d_blocks : reinterpret_cast<aof_block *>(ws + L.l0_blocks);
    v.subdirs0 = nullptr;
    if (p.subpixel) v.subdirs0 = d_subdirs ? d_subdirs : ws + L.l0_subdirs;
    v.flows = d_flows;
    v.hist0 = ws + L.l0_hist;
    v.hints = p.tile == 16 ? reinterpret_cast<uint32_t *>(ws + L.hints) : nullptr;
    return v;
}

// Small pairs (sparse grids, frames that fit LDS -- the reference's call shape): sums, pyramid, searches
// and reductions of a pair in one launch, one workgroup per pair.  Large batches of such pairs keep the
// separate kernels, whose grouped searches pack several pairs into a workgroup.  Fills *sm and says
// whether the one-launch kernel serves the batch.
bool small_args(const aof_ctx *ctx, const BatchView &v, int64_t n_pairs, SmallArgs *sm)
{
    const aof_params &p = ctx->params;
    const bool two = p.pyramid_levels == 2;
    if (ctx->force_generic || ctx->split_coarse || n_pairs > kSmallMaxPairs) return false;
    sm->levels = two ? 2 : 1;
    sm->l0 = search_args(ctx, 0, v.prev, v.cur, v.stride, v.blocks0, v.subdirs0, nullptr, v.sums, n_pairs);
    sm->l1 = search_args(ctx, 1, v.l1_prev, v.l1_cur, (int64_t)(p.width / 2) * (p.height / 2), v.blocks1, v.subdirs1,
                         nullptr, v.sums, n_pairs);
    sm->t0 = flow_tail(ctx, 0, v.flows, two ? v.flows1 : nullptr);
    sm->t1 = flow_tail(ctx, 1, v.flows1, nullptr);
    sm->sums = v.sums;
    return search_kind(ctx, sm->l0) == SK_LANE8_GROUP && (!two || search_kind(ctx, sm->l1) == SK_LANE8_GROUP) &&
           flow_small_supported(*sm);
}

// ---- resident form of the per-call path ----
constexpr uint64_t kResidentIdleTicks = 5000000;    // 50 ms of the 100 MHz counter without a request: the kernel leaves
constexpr uint64_t kResidentLifeTicks = 20000000;   // 200 ms in total: nothing that waits for the device waits longer
constexpr double kTaggedRecordWaitS = 0.002;         // per-call graph: polling for the tagged record this long, then the stream decides
constexpr double kResidentHostTimeoutS = 0.25;      // the host gives up on a request and falls back to the graph path

// Asks the resident kernel to leave and waits for it (bounded by the kernel's own deadlines).  Must run
// before anything that frees or reallocates what the kernel reads, and before a change of kernel choice.
// The stop bit is only ever cleared after the DEVICE has cleared `running` (the kernel's last store): a
// launched instance that has not started yet still finds the bit on its first poll and leaves at once.
// false: it did not leave within a second (five lifetimes).  The box keeps its stop bit for good, the
// context forgets the box, the stream and every buffer the kernel may still read or write (leaked, never
// freed or reused), and aof_destroy frees no device memory at all (a hipFree waits for every kernel).
bool resident_stop(aof_ctx *ctx)
{
    if (!ctx->box || !ctx->rstream) return true;
    ResidentBox *box = ctx->box;
    if (!__atomic_load_n(&box->running, __ATOMIC_ACQUIRE)) return true;   // nothing launched since the last exit
    const unsigned long long word = __atomic_load_n(&box->word, __ATOMIC_ACQUIRE);
    __atomic_store_n(&box->word, word | kResidentStopBit, __ATOMIC_RELEASE);
    // The kernel clears `running` when it leaves -- at the latest on its 200 ms lifetime deadline.  Wait for
    // THAT, bounded.  No HIP call is needed for the launch to reach the device: hipLaunchKernelGGL has
    // written the AQL packet and rung the queue's doorbell before it returned (direct dispatch; the
    // launch-to-first-poll latency in aof_stream_stats is measured with the host spinning on pinned memory
    // and nothing else).
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(&box->running, __ATOMIC_ACQUIRE) && seconds_since(t0) < ctx->rstop_wait_s) {
    }
    hipError_t e = hipSuccess;
    if (!__atomic_load_n(&box->running, __ATOMIC_ACQUIRE)) {
        // it has left; the stream retires the launch within microseconds -- bounded all the same
        e = drain_bounded(ctx->rstream, kDrainS);
        if (e == hipSuccess) {
            __atomic_store_n(&box->word, word & ~kResidentStopBit, __ATOMIC_RELEASE);
            return true;
        }
    }
    std::fprintf(stderr, "aof: the resident kernel did not leave within %.0f ms of being asked to (launch %u, started %u, "
                         "served %u, exited at %u, on device %u, stream: %s): its buffers are abandoned\n",
                 ctx->rstop_wait_s * 1e3, ctx->rlaunches, (unsigned)box->started, (unsigned)box->done, (unsigned)box->exited, (unsigned)box->running,
                 e == hipSuccess ? hipGetErrorString(hipStreamQuery(ctx->rstream)) : hipGetErrorString(e));
    ctx->resident_lost = true;
    ctx->stats.resident_lost++;
    ctx->resident_on = false;
    // forget (leak) everything the kernel may still touch; the host-buffer state is rebuilt on the next call
    ctx->box = nullptr; ctx->rstream = nullptr;
    ctx->h_frames[0] = ctx->h_frames[1] = nullptr; ctx->h_flow = nullptr; ctx->h_tag = nullptr;
    ctx->d_blocks = nullptr; ctx->d_subdirs = nullptr; ctx->d_flow = nullptr; ctx->d_ws = nullptr;
    ctx->host_dirty = true;
    return false;
}

}  // namespace

extern "C" {

int aof_create(const aof_params *p, int device, aof_ctx **out)
{
    if (!out) return -EINVAL;
    *out = nullptr;
    int rc = aof_params_check(p);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return -ENODEV;  // kernels are gfx950 only

    aof_ctx *ctx = new (std::nothrow) aof_ctx();
    if (!ctx) return -ENOMEM;
    std::memset(ctx, 0, sizeof(*ctx));
    ctx->params = *p;
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    grid_for_level(*p, 0, &ctx->g0);
    if (p->pyramid_levels == 2) grid_for_level(*p, 1, &ctx->g1);
    std::snprintf(ctx->err, sizeof(ctx->err), "ok");
    if (p->tile == 8 && p->search == 4) {   // the flat lane8 search can reduce in its own launch
        DeviceGuard guard(device);
        const size_t bytes = (size_t)kVotePairs * kVoteStride * sizeof(uint32_t);
        if (hipMalloc((void **)&ctx->d_votes, bytes) != hipSuccess || hipMemset(ctx->d_votes, 0, bytes) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->votes_done, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void **)&ctx->h_fault, 64 + kPruneSlots * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            aof_destroy(ctx);
            return -EIO;
        }
        std::memset(ctx->h_fault, 0, 64 + kPruneSlots * sizeof(uint32_t));
        ctx->h_prune_slots = ctx->h_fault + 16;
        ctx->votes_pairs = kVotePairs;
    }
    ctx->vote_deadline_ticks = kVoteDeadlineTicks;
    ctx->rstop_wait_s = 1.0;
    ctx->separate_reduce = true;   // the in-launch reduction is opt-in (aof_set_reduce_fusion)
    // exact pruning wherever it pays (include/aof.h): 16x16 tiles by a probe per pair, 8x8 tiles by what the
    // context's previous launches reported
    ctx->search_mode = AOF_SEARCH_ADAPTIVE;
    ctx->prune_belief = -1;
    ctx->search_stats.belief = -1;

    *out = ctx;
    return 0;
}

void aof_destroy(aof_ctx *ctx)
{
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)resident_stop(ctx);   // before anything it reads is freed
    // Every wait here is bounded: a hipFree / hipStreamDestroy / hipStreamSynchronize waits for the device
    // without a time limit, so they only run once the context's own work is known to have drained.  If it
    // has not (a lost resident kernel, a wedged device, a fault), the context's device and pinned memory and
    // its streams are leaked -- the caller (calcFlow's owner holds _mainloop_lock, mainloop.cpp:283) gets
    // control back either way.
    bool leak = ctx->resident_lost || ctx->wedged;
    hipError_t e = hipSuccess;
    if (!leak && ctx->stream && (e = drain_bounded(ctx->stream, kDrainS)) != hipSuccess) leak = true;
    if (!leak && ctx->votes_done && ctx->votes_used && (e = event_wait_bounded(ctx->votes_done, kDrainS)) != hipSuccess)
        leak = true;
    if (ctx->ev) {
        for (int k = 0; k < AOF_K_COUNT; k++)
            for (int r = 0; r < AOF_PROFILE_RING; r++)
                for (int ev = 0; ev < 2; ev++)
                    if (ctx->ev[k][r][ev]) (void)hipEventDestroy(ctx->ev[k][r][ev]);
        delete[] ctx->ev;
    }
    if (leak) {
        std::fprintf(stderr, "aof: destroying a context whose device work has not drained (%s): its device memory, pinned "
                             "memory and streams are leaked, not freed\n",
                     ctx->resident_lost ? "resident kernel lost" : ctx->wedged ? ctx->err : hipGetErrorString(e));
        delete ctx;
        return;
    }
    if (ctx->rstream) (void)hipStreamDestroy(ctx->rstream);
    if (ctx->box) (void)hipHostFree(ctx->box);
    for (int i = 0; i < 2; i++) if (ctx->push_graph[i]) (void)hipGraphExecDestroy(ctx->push_graph[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->h_frame) (void)hipHostFree(ctx->h_frame);
    for (int i = 0; i < 2; i++) if (ctx->h_frames[i]) (void)hipHostFree(ctx->h_frames[i]);
    if (ctx->h_flow) (void)hipHostFree(ctx->h_flow);
    for (int i = 0; i < 2; i++) if (ctx->d_frames[i]) (void)hipFree(ctx->d_frames[i]);
    for (int i = 0; i < 2; i++) if (ctx->d_pair[i]) (void)hipFree(ctx->d_pair[i]);
    if (ctx->d_blocks) (void)hipFree(ctx->d_blocks);
    if (ctx->d_subdirs) (void)hipFree(ctx->d_subdirs);
    if (ctx->d_flow) (void)hipFree(ctx->d_flow);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->votes_done) (void)hipEventDestroy(ctx->votes_done);
    if (ctx->d_votes) (void)hipFree(ctx->d_votes);
    if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
    delete ctx;
}

const char *aof_last_error(const aof_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int aof_get_search_mode(const aof_ctx *ctx) { return ctx ? ctx->search_mode : -EINVAL; }

int aof_get_search_stats(const aof_ctx *ctx, aof_search_stats *out)
{
    if (!ctx || !out) return -EINVAL;
    *out = ctx->search_stats;
    return 0;
}

int aof_set_search_belief(aof_ctx *ctx, int belief)
{
    if (!ctx || belief < -1 || belief > 1) return -EINVAL;
    ctx->prune_belief = belief;
    ctx->search_stats.belief = belief;
    ctx->prune_since_probe = 0;
    ctx->prune_expected = 0;   // (reports of earlier launches no longer overrule the caller)
    return 0;
}

int aof_get_params(const aof_ctx *ctx, aof_params *out)
{
    if (!ctx || !out) return -EINVAL;
    *out = ctx->params;
    return 0;
}

const char *aof_search_variant(const aof_ctx *ctx)
{
    if (!ctx) return "";
    // which search kernel will level 0 use? (probe with aligned dummy pointers)
    const aof_params &p = ctx->params;
    SearchArgs probe = search_args(ctx, 0, nullptr, nullptr, (int64_t)p.width * p.height, nullptr, nullptr,
                                   nullptr, nullptr, 1);
    switch (search_kind(ctx, probe)) {
    case SK_TILE16: return "tile16_lds";
    case SK_GENERIC: return "generic";
    default: return "lane8";
    }
}

// Captured graphs hold the kernels chosen so far.  (The per-call path does not wait for the stream after a
// tagged record has arrived: drain it before a graph goes.)
static void drop_push_graphs(aof_ctx *ctx)
{
    if (!ctx->push_graph[0] && !ctx->push_graph[1]) return;
    DeviceGuard guard(ctx->device);
    if (ctx->stream && !ctx->wedged) {
        const hipError_t e = drain_bounded(ctx->stream, kDrainS);
        if (e != hipSuccess) (void)wedge(ctx, "draining the per-call stream before its graphs are dropped", e);
    }
    for (int i = 0; i < 2; i++)   // (a wedged context leaks the executables: a replay may still be running)
        if (ctx->push_graph[i]) { if (!ctx->wedged) (void)hipGraphExecDestroy(ctx->push_graph[i]); ctx->push_graph[i] = nullptr; }
}

int aof_set_force_generic(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    { DeviceGuard guard(ctx->device); (void)resident_stop(ctx); }   // it runs the kernels chosen so far
    if ((on != 0) != ctx->force_generic) drop_push_graphs(ctx);
    ctx->force_generic = on != 0;
    return 0;
}

int aof_set_search_mode(aof_ctx *ctx, int mode)
{
    if (!ctx || mode < AOF_SEARCH_EXHAUSTIVE || mode > AOF_SEARCH_ADAPTIVE) return -EINVAL;
    { DeviceGuard guard(ctx->device); (void)resident_stop(ctx); }   // it runs the kernels chosen so far
    if (mode != ctx->search_mode) drop_push_graphs(ctx);
    ctx->search_mode = mode;
    return 0;
}

int aof_set_profiling(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    DeviceGuard guard(ctx->device);
    if (on && !ctx->ev) {
        ctx->ev = new (std::nothrow) hipEvent_t[AOF_K_COUNT][AOF_PROFILE_RING][2]();
        if (!ctx->ev) return fail(ctx, -ENOMEM, "event ring");
        for (int k = 0; k < AOF_K_COUNT; k++)
            for (int r = 0; r < AOF_PROFILE_RING; r++)
                // timing only: without the system-scope fence a default event performs when it is
                // recorded (an L2 write-back and invalidation between the kernels it brackets)
                for (int e = 0; e < 2; e++)
                    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev[k][r][e], hipEventDisableSystemFence));
    }
    ctx->profiling = on != 0;
    ctx->profile_mask = 0xFFFFFFFFu;
    if (on) for (int k = 0; k < AOF_K_COUNT; k++) ctx->ev_count[k] = 0;
    return 0;
}

int aof_set_profiling_mask(aof_ctx *ctx, uint32_t mask)
{
    int rc = aof_set_profiling(ctx, mask != 0);
    if (rc) return rc;
    ctx->profile_mask = mask;
    return 0;
}

int aof_profile_count(const aof_ctx *ctx, int kernel_id)
{
    if (!ctx || kernel_id < 0 || kernel_id >= AOF_K_COUNT) return -EINVAL;
    const int64_t n = ctx->ev_count[kernel_id];
    return (int)(n < AOF_PROFILE_RING ? n : AOF_PROFILE_RING);
}

int aof_profile_ms(aof_ctx *ctx, int kernel_id, int index, float *ms)
{
    if (!ctx || !ms) return -EINVAL;
    const int kept = aof_profile_count(ctx, kernel_id);
    if (kept < 0 || index < 0 || index >= kept)
        return fail(ctx, -EINVAL, "kernel %d has no timed launch %d", kernel_id, index);
    const int64_t n = ctx->ev_count[kernel_id];
    const int slot = (int)((n - kept + index) % AOF_PROFILE_RING);
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[kernel_id][slot][1]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev[kernel_id][slot][0], ctx->ev[kernel_id][slot][1]));
    return 0;
}

int aof_kernel_ms(aof_ctx *ctx, int kernel_id, float *ms)
{
    const int kept = ctx ? aof_profile_count(ctx, kernel_id) : -EINVAL;
    if (kept < 0) return -EINVAL;
    if (kept == 0) return fail(ctx, -EINVAL, "kernel %d was not timed", kernel_id);
    return aof_profile_ms(ctx, kernel_id, kept - 1, ms);
}

int aof_flow_batch_device(aof_ctx *ctx, const uint8_t *d_prev, const uint8_t *d_cur,
                          int64_t pair_stride, int64_t n_pairs, aof_block *d_blocks,
                          uint8_t *d_subdirs, aof_flow *d_flows, void *d_workspace,
                          size_t workspace_bytes, void *stream)
{
    if (!ctx) return -EINVAL;
    if (n_pairs < 0 || (n_pairs > 0 && (!d_prev || !d_cur || !d_flows)))
        return fail(ctx, -EINVAL, "null frame or flow pointer");
    if (int sticky = sticky_error(ctx)) return sticky;
    if (n_pairs == 0) return 0;
    const aof_params &p = ctx->params;
    if (pair_stride < (int64_t)p.width * p.height && n_pairs > 1)
        return fail(ctx, -EINVAL, "pair_stride smaller than a frame");
    aof_ws_layout L;
    int rc = aof_workspace_layout(&p, n_pairs, &L);
    if (rc) return fail(ctx, rc, "bad workspace layout");
    if (!d_workspace || workspace_bytes < L.total_bytes)
        return fail(ctx, -ENOSPC, "workspace %zu B < required %zu B", workspace_bytes, L.total_bytes);
    if (reinterpret_cast<uintptr_t>(d_workspace) % 256)
        return fail(ctx, -EINVAL, "workspace must be 256-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_blocks) % 4 || reinterpret_cast<uintptr_t>(d_flows) % 4)
        return fail(ctx, -EINVAL, "block and flow records must be 4-byte aligned");
    int cur_dev = -1;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev != ctx->device)
        return fail(ctx, -EINVAL, "context was created for device %d but the calling thread's current "
                                  "device is %d", ctx->device, cur_dev);

    hipStream_t s = static_cast<hipStream_t>(stream);
    const BatchView v = batch_view(ctx, L, d_prev, d_cur, pair_stride, d_blocks, d_subdirs, d_flows, d_workspace);

    {
        SmallArgs sm;
        if (small_args(ctx, v, n_pairs, &sm)) {
            Timed t(ctx, AOF_K_SEARCH, s);
            rc = launch_flow_small(sm, s);
            if (rc) return fail(ctx, -EIO, "small-pair launch: %s", hipGetErrorString((hipError_t)rc));
            return 0;
        }
    }

    rc = enqueue_coarse(ctx, v, 0, n_pairs, s);
    if (!rc) rc = enqueue_fine(ctx, v, 0, n_pairs, s);
    return rc;
}

}  // extern "C"

namespace aof {

// What every entry point that enqueues work on a context checks before its first launch: the sticky device-side
// condition, and that the calling thread's current device is the context's.  Sets aof_last_error.
int precheck(aof_ctx *ctx)
{
    if (int sticky = sticky_error(ctx)) return sticky;
    int cur_dev = -1;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev != ctx->device)
        return fail(ctx, -EINVAL, "context was created for device %d but the calling thread's current "
                                  "device is %d", ctx->device, cur_dev);
    return 0;
}

int ctx_fail(aof_ctx *ctx, int code, const char *what) { return fail(ctx, code, "%s", what); }

// Would a sequence-view call (frames viewed twice, n_pairs = frames - 1) run K1 as a pass of its own?  Mirrors the
// choices of aof_flow_batch_device / enqueue_coarse.  The sequence pipeline asks, because its ingest kernel can
// leave K1's outputs (pixel sums at ws + L.sums, one level-1 frame per FRAME from ws + L.l1_prev on) itself.
bool sequence_runs_k1(aof_ctx *ctx, const uint8_t *d_frames, int64_t n_pairs, void *d_workspace)
{
    const aof_params &p = ctx->params;
    const bool two = p.pyramid_levels == 2, eq = p.mean_subtract != 0;
    if (n_pairs < 1 || (!two && !eq)) return false;
    aof_ws_layout L;
    if (aof_workspace_layout(&p, n_pairs, &L)) return false;
    const int64_t frame = (int64_t)p.width * p.height;
    const BatchView v = batch_view(ctx, L, d_frames, d_frames + frame, frame, nullptr, nullptr, nullptr, d_workspace);
    SmallArgs sm;
    if (small_args(ctx, v, n_pairs, &sm)) return false;
    if (two && !ctx->force_generic && !ctx->split_coarse &&
        coarse_fused_supported(coarse_args(ctx, v, 0, n_pairs, v.sums)))
        return false;
    return true;
}

// aof_flow_batch_device on the sequence view of `d_frames`; k1_ready: K1's outputs are in the workspace already.
int flow_sequence(aof_ctx *ctx, const uint8_t *d_frames, int64_t n_pairs, aof_flow *d_flows, void *d_workspace,
                  size_t workspace_bytes, void *stream, bool k1_ready)
{
    const int64_t frame = (int64_t)ctx->params.width * ctx->params.height;
    ctx->k1_ready = k1_ready;
    const int rc = aof_flow_batch_device(ctx, d_frames, d_frames + frame, frame, n_pairs, nullptr, nullptr, d_flows,
                                         d_workspace, workspace_bytes, stream);
    ctx->k1_ready = false;
    return rc;
}

}  // namespace aof

extern "C" {

int aof_set_split_coarse(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    { DeviceGuard guard(ctx->device); (void)resident_stop(ctx); }   // it runs the kernels chosen so far
    if ((on != 0) != ctx->split_coarse) drop_push_graphs(ctx);
    ctx->split_coarse = on != 0;
    return 0;
}

int aof_set_reduce_fusion(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    ctx->separate_reduce = on == 0;
    return 0;
}

int aof_ingest_batch_device(const aof_ingest_params *p, const uint8_t *d_camera,
                            int64_t camera_stride, int64_t n_frames, uint8_t *d_cropped,
                            int64_t cropped_stride, uint32_t *d_hist, void *stream)
{
    if (!p || n_frames < 0) return -EINVAL;
    if (p->crop_width < 1 || p->crop_height < 1 || p->crop_width > p->camera_width ||
        p->crop_height > p->camera_height)
        return -EINVAL;
    if (n_frames == 0) return 0;
    if (!d_camera || (!d_cropped && !d_hist)) return -EINVAL;
    if (camera_stride < (int64_t)p->camera_width * p->camera_height && n_frames > 1) return -EINVAL;
    if (d_cropped && cropped_stride < (int64_t)p->crop_width * p->crop_height && n_frames > 1) return -EINVAL;
    if (n_frames * ((p->crop_height + 15) / 16) > 0x7FFFFFFF) return -EINVAL;
    return launch_ingest(*p, d_camera, camera_stride, n_frames, d_cropped, cropped_stride, d_hist, stream)
               ? -EIO : 0;
}

int aof_derotate_batch_device(const aof_derotate_params *p, const aof_flow *d_flows,
                              const aof_gyro *d_gyro, int64_t n, float *d_out, void *stream)
{
    if (!p || n < 0) return -EINVAL;
    if (n == 0) return 0;
    if (!d_flows || !d_gyro || !d_out || (n + 255) / 256 > 0x7FFFFFFF) return -EINVAL;
    return launch_derotate(*p, d_flows, d_gyro, n, d_out, stream) ? -EIO : 0;
}

// ---- host-buffer conveniences ------------------------------------------------

// Forgets the host-buffer state without freeing it (part of it belongs to a resident kernel that did not
// leave, or the device did not drain: a hipFree would wait for that without a time limit).
static void forget_host_state(aof_ctx *ctx)
{
    ctx->stream = nullptr; ctx->h_frame = nullptr;
    ctx->h_frames[0] = ctx->h_frames[1] = nullptr; ctx->h_flow = nullptr; ctx->h_tag = nullptr;
    ctx->d_frames[0] = ctx->d_frames[1] = nullptr; ctx->d_pair[0] = ctx->d_pair[1] = nullptr;
    ctx->d_blocks = nullptr; ctx->d_subdirs = nullptr; ctx->d_flow = nullptr; ctx->d_ws = nullptr;
    ctx->push_graph[0] = ctx->push_graph[1] = nullptr;
    ctx->host_ready = false;
    ctx->host_dirty = false;
    ctx->have_prev = false;
}

static void free_host_state(aof_ctx *ctx)
{
    (void)resident_stop(ctx);
    if (ctx->resident_lost || ctx->wedged || ctx->host_dirty) { forget_host_state(ctx); return; }
    if (ctx->stream) {
        const hipError_t e = drain_bounded(ctx->stream, kDrainS);
        if (e != hipSuccess) { (void)wedge(ctx, "draining the per-call stream", e); forget_host_state(ctx); return; }
        for (int i = 0; i < 2; i++)
            if (ctx->push_graph[i]) { (void)hipGraphExecDestroy(ctx->push_graph[i]); ctx->push_graph[i] = nullptr; }
        (void)hipStreamDestroy(ctx->stream); ctx->stream = nullptr;
    }
    if (ctx->h_frame) { (void)hipHostFree(ctx->h_frame); ctx->h_frame = nullptr; }
    for (int i = 0; i < 2; i++) if (ctx->h_frames[i]) { (void)hipHostFree(ctx->h_frames[i]); ctx->h_frames[i] = nullptr; }
    if (ctx->h_flow) { (void)hipHostFree(ctx->h_flow); ctx->h_flow = nullptr; ctx->h_tag = nullptr; }
    for (int i = 0; i < 2; i++) if (ctx->d_frames[i]) { (void)hipFree(ctx->d_frames[i]); ctx->d_frames[i] = nullptr; }
    for (int i = 0; i < 2; i++) if (ctx->d_pair[i]) { (void)hipFree(ctx->d_pair[i]); ctx->d_pair[i] = nullptr; }
    if (ctx->d_blocks) { (void)hipFree(ctx->d_blocks); ctx->d_blocks = nullptr; }
    if (ctx->d_subdirs) { (void)hipFree(ctx->d_subdirs); ctx->d_subdirs = nullptr; }
    if (ctx->d_flow) { (void)hipFree(ctx->d_flow); ctx->d_flow = nullptr; }
    if (ctx->d_ws) { (void)hipFree(ctx->d_ws); ctx->d_ws = nullptr; }
    ctx->host_ready = false;
}

static int alloc_host_state(aof_ctx *ctx)
{
    const aof_params &p = ctx->params;
    const size_t frame = (size_t)p.width * p.height;
    aof_ws_layout L;
    aof_workspace_layout(&p, 1, &L);
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_frames[i], frame));
    for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_pair[i], frame));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_blocks, sizeof(aof_block) * (size_t)ctx->g0.blocks()));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_subdirs, (size_t)ctx->g0.blocks()));
    HIP_TRY(ctx, hipMalloc(&ctx->d_ws, L.total_bytes));
    ctx->ws_bytes = L.total_bytes;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_flow, sizeof(aof_flow)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_frame, frame, hipHostMallocDefault));
    // Frames of up to 64 KB (the reference's 64x64 .. 128x128 images) are not copied to the device
    // at all: the kernels read the pinned host copies over PCIe, which takes less time than the
    // copy node it replaces.  Larger frames keep the H2D copy and the device-resident previous frame.
    ctx->zero_copy = frame <= 64 * 1024;
    if (ctx->zero_copy)
        for (int i = 0; i < 2; i++)
            HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_frames[i], frame, hipHostMallocMapped | hipHostMallocCoherent));
    // (record in the first cache line, the tag of the next tagged record in the second)
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_flow, 128, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(ctx->h_flow, 0, 128);
    ctx->h_tag = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->h_flow) + 64);
    return 0;
}

// Stream, device frames and pinned buffers of the host-buffer entry points, made on first use.
// A failure half-way frees what was made, so that a later call starts from scratch instead of
// overwriting (leaking) live handles.
static int ensure_host_state(aof_ctx *ctx)
{
    if (ctx->host_dirty) forget_host_state(ctx);   // (abandoned to a lost resident kernel: start over with fresh buffers)
    if (ctx->host_ready) return 0;
    const int rc = alloc_host_state(ctx);
    if (rc) {
        free_host_state(ctx);
        return rc;
    }
    ctx->host_ready = true;
    return 0;
}

// Captures [H2D frame -> kernels (result written to pinned host memory)] for destination
// slot `slot` into a graph.
// Any failure leaves the context on the plain (un-captured) path; never an error.
static void build_push_graph(aof_ctx *ctx, int slot)
{
    const aof_params &p = ctx->params;
    const size_t bytes = (size_t)p.width * p.height;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        ctx->graph_disabled = true;
        return;
    }
    ctx->capturing = true;
    uint8_t *const *frames = ctx->zero_copy ? ctx->h_frames : ctx->d_frames;
    bool ok = false, tagged = false;
    if (ctx->zero_copy) {
        // small frames served by the one-workgroup kernel: the record comes tagged (stream_push_graph polls
        // for it); the kernel's own copy goes to device memory
        aof_ws_layout L;
        aof_workspace_layout(&p, 1, &L);
        const BatchView v = batch_view(ctx, L, frames[1 - slot], frames[slot], (int64_t)bytes, ctx->d_blocks, ctx->d_subdirs,
                                       ctx->d_flow, ctx->d_ws);
        SmallArgs sm;
        if (small_args(ctx, v, 1, &sm)) {
            tagged = true;
            ok = launch_flow_small_tagged(sm, ctx->h_flow, ctx->h_tag, ctx->stream) == 0;
        }
    }
    if (!tagged) {
        ok = ctx->zero_copy ||
             hipMemcpyAsync(ctx->d_frames[slot], ctx->h_frame, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        // K3 writes the 16-byte result straight into the pinned (device-visible, coherent) host
        // record: no D2H copy node; it is visible to the host once the stream has drained.
        ok = ok && aof_flow_batch_device(ctx, frames[1 - slot], frames[slot], (int64_t)bytes, 1,
                                         ctx->d_blocks, ctx->d_subdirs, ctx->h_flow, ctx->d_ws, ctx->ws_bytes,
                                         ctx->stream) == 0;
    }
    ctx->capturing = false;
    const bool ended = hipStreamEndCapture(ctx->stream, &graph) == hipSuccess && graph;
    if (ok && ended && hipGraphInstantiate(&ctx->push_graph[slot], graph, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(graph);
        ctx->push_tagged[slot] = tagged;
        return;
    }
    if (graph) (void)hipGraphDestroy(graph);
    ctx->push_graph[slot] = nullptr;
    ctx->graph_disabled = true;
    (void)hipGetLastError();
}

static int run_one(aof_ctx *ctx, const uint8_t *d_prev, const uint8_t *d_cur, aof_block *blocks,
                   uint8_t *subdirs, aof_flow *flow)
{
    const aof_params &p = ctx->params;
    int rc = aof_flow_batch_device(ctx, d_prev, d_cur, (int64_t)p.width * p.height, 1, ctx->d_blocks,
                                   ctx->d_subdirs, ctx->d_flow, ctx->d_ws, ctx->ws_bytes,
                                   ctx->stream);
    if (rc) return rc;
    const size_t nb = (size_t)ctx->g0.blocks();
    HIP_TRY(ctx, hipMemcpyAsync(flow, ctx->d_flow, sizeof(aof_flow), hipMemcpyDeviceToHost, ctx->stream));
    if (blocks)
        HIP_TRY(ctx, hipMemcpyAsync(blocks, ctx->d_blocks, nb * sizeof(aof_block),
                                    hipMemcpyDeviceToHost, ctx->stream));
    if (subdirs) {
        if (p.subpixel)
            HIP_TRY(ctx, hipMemcpyAsync(subdirs, ctx->d_subdirs, nb, hipMemcpyDeviceToHost, ctx->stream));
        else
            std::memset(subdirs, 8, nb);
    }
    if (const hipError_t e = drain_bounded(ctx->stream, kDrainS)) return wedge(ctx, "waiting for the pair's kernels and copies", e);
    return 0;
}

int aof_flow_pair_host(aof_ctx *ctx, const uint8_t *prev, const uint8_t *cur, aof_block *blocks,
                       uint8_t *subdirs, aof_flow *flow)
{
    if (!ctx) return -EINVAL;
    if (!prev || !cur || !flow) return fail(ctx, -EINVAL, "null frame or flow pointer");
    if (int sticky = sticky_error(ctx)) return sticky;
    DeviceGuard guard(ctx->device);
    int rc = ensure_host_state(ctx);
    if (rc) return rc;
    const size_t frame = (size_t)ctx->params.width * ctx->params.height;
    // own scratch frames: the streaming state (aof_stream_push_host) is left untouched
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pair[0], prev, frame, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pair[1], cur, frame, hipMemcpyHostToDevice, ctx->stream));
    return run_one(ctx, ctx->d_pair[0], ctx->d_pair[1], blocks, subdirs, flow);
}

static int stream_push_graph(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow, int slot);
static int stream_push_resident(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow, int slot, bool *served);

int aof_stream_push_host(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow)
{
    if (!ctx) return -EINVAL;
    if (!frame || !flow) return fail(ctx, -EINVAL, "null frame or flow pointer");
    if (int sticky = sticky_error(ctx)) return sticky;
    DeviceGuard guard(ctx->device);
    int rc = ensure_host_state(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)ctx->params.width * ctx->params.height;
    int slot = ctx->have_prev ? 1 - ctx->cur_slot : 0;
    if (ctx->have_prev) ctx->stats.calls++;
    if (ctx->have_prev && ctx->resident_on && ctx->zero_copy && !ctx->profiling) {
        bool served = false;
        rc = stream_push_resident(ctx, frame, flow, slot, &served);
        if (served || rc) return rc;   // (not served and no error: this configuration takes the paths below)
        if (ctx->host_dirty) {
            // The kernel neither answered nor left: the pinned frames it may still read are abandoned, the
            // older frame with them.  This frame starts a new sequence on fresh buffers (return 1, as after
            // aof_stream_reset) -- one flow sample is lost, nothing wrong is ever reported.
            rc = ensure_host_state(ctx);
            if (rc) return rc;
            slot = 0;
        }
    }
    if (ctx->have_prev && !ctx->graph_disabled && !ctx->profiling) {
        if (!ctx->push_graph[slot]) build_push_graph(ctx, slot);
        if (ctx->push_graph[slot]) return stream_push_graph(ctx, frame, flow, slot);
    }
    uint8_t *const *frames = ctx->zero_copy ? ctx->h_frames : ctx->d_frames;
    ctx->rframe_req[slot] = 0;   // (written outside a resident request)
    if (ctx->zero_copy) std::memcpy(ctx->h_frames[slot], frame, bytes);
    else HIP_TRY(ctx, hipMemcpyAsync(ctx->d_frames[slot], frame, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->have_prev) {
        // caller may free `frame` on return
        if (const hipError_t e = drain_bounded(ctx->stream, kDrainS)) return wedge(ctx, "waiting for the first frame's copy", e);
        ctx->cur_slot = slot;
        ctx->have_prev = true;
        std::memset(flow, 0, sizeof(*flow));
        return 1;
    }
    rc = run_one(ctx, frames[ctx->cur_slot], frames[slot], nullptr, nullptr, flow);
    if (rc) {   // the new frame may be incomplete on the device: do not compare the next one with it
        ctx->have_prev = false;
        return rc;
    }
    ctx->cur_slot = slot;
    return 0;
}

// The tagged 16-byte record of the per-call paths (k_flow_small_tagged, k_flow_resident): the device
// publishes it with ONE 16-byte store to a 16-byte aligned address in pinned, coherent host memory -- one
// PCIe write, which the root complex commits to its cache line as a whole -- and the top byte of `count`
// (word 2) carries the tag.  The host reads it with ONE 16-byte load (an aligned SSE load is a single
// access), checks the tag IN THAT COPY, and reads once more to see the same bytes again.
typedef uint32_t RecordWords __attribute__((vector_size(16)));
static inline bool read_tagged_record(const aof_flow *pinned, uint32_t tag, aof_flow *out)
{
    const volatile RecordWords *rec = reinterpret_cast<const volatile RecordWords *>(pinned);
    const RecordWords a = *rec;
    if ((a[2] & 0xFF000000u) != tag) return false;
    const RecordWords b = *rec;
    if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]) return false;
    std::memcpy(out, &a, sizeof(*out));
    out->count &= 0x00FFFFFFu;
    return true;
}

// Before a request is posted: whatever record is in place (first use, a record of the other path) must not
// carry the new request's tag.
static inline void retag_stale_record(aof_flow *pinned, uint32_t tag)
{
    volatile uint32_t *word = &reinterpret_cast<volatile uint32_t *>(pinned)[2];
    if ((*word & 0xFF000000u) == tag) *word ^= 0x80000000u;
}

// Same contract as the plain path above, one hipGraphLaunch per frame.
static int stream_push_graph(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow, int slot)
{
    ctx->rframe_req[slot] = 0;   // (written outside a resident request)
    std::memcpy(ctx->zero_copy ? ctx->h_frames[slot] : ctx->h_frame, frame,
                (size_t)ctx->params.width * ctx->params.height);
    const bool tagged = ctx->push_tagged[slot];
    uint32_t tag = 0;
    if (tagged) {
        tag = ++ctx->rseq << 24;
        retag_stale_record(ctx->h_flow, tag);
        __atomic_store_n(ctx->h_tag, ctx->rseq, __ATOMIC_RELEASE);
    }
    hipError_t e = hipGraphLaunch(ctx->push_graph[slot], ctx->stream);
    if (e == hipSuccess && tagged) {
        // The record arrives tagged: the kernel is through with both frames when it is there, and the
        // runtime's own completion path (longer than the kernel) is not waited for.  A record that stays
        // away for 2 ms is left to the stream -- bounded: the stream drains (and the record is there), or it
        // reports the fault, or the time runs out and the context is disabled.
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1; !read_tagged_record(ctx->h_flow, tag, flow); spins++) {
            if ((spins & 0x3FFu) == 0 && seconds_since(t0) > kTaggedRecordWaitS) {
                ctx->stats.tagged_slow++;
                e = drain_bounded(ctx->stream, kDrainS);
                if (e == hipSuccess && !read_tagged_record(ctx->h_flow, tag, flow)) e = hipErrorUnknown;
                break;
            }
        }
    } else if (e == hipSuccess) {
        e = drain_bounded(ctx->stream, kDrainS);
        if (e == hipSuccess) *flow = *ctx->h_flow;
    }
    if (e != hipSuccess) {
        ctx->have_prev = false;
        if (e == hipErrorNotReady) return wedge(ctx, "per-call graph replay", e);
        return fail(ctx, -EIO, "graph replay: %s", hipGetErrorString(e));
    }
    ctx->cur_slot = slot;
    return 0;
}

// The resident path: post the request, make sure the kernel is there, wait for its tagged record.
// *served = false (and 0) when the one-workgroup kernel does not serve this configuration, or when it did
// not answer (the caller's frame then takes the launch-per-call path).
static int stream_push_resident(aof_ctx *ctx, const uint8_t *frame, aof_flow *flow, int slot, bool *served)
{
    const aof_params &p = ctx->params;
    const size_t bytes = (size_t)p.width * p.height;
    *served = false;
    if (!ctx->box) {
        // The kernel's own stream at the HIGHEST priority: the runtime keeps a separate pool of hardware
        // queues per priority, so this stream never shares a hardware queue with the context's other stream,
        // the caller's or torch's (all normal priority) -- on a shared queue every packet carries the barrier
        // bit, and a parked resident kernel would hold up the other stream's work for up to its lifetime
        // (and be held up by it).  It can only meet other contexts' resident streams there, from the fifth on
        // (GPU_MAX_HW_QUEUES = 4): a resident kernel queued behind another one starts when that one leaves,
        // at most 200 ms later, which is inside the 250 ms a request waits.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipHostMalloc((void **)&ctx->box, sizeof(ResidentBox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            (hipStreamCreateWithPriority(&ctx->rstream, hipStreamNonBlocking, greatest) != hipSuccess &&
             hipStreamCreateWithFlags(&ctx->rstream, hipStreamNonBlocking) != hipSuccess)) {
            if (ctx->box) { (void)hipHostFree(ctx->box); ctx->box = nullptr; }
            ctx->rstream = nullptr;
            ctx->resident_on = false;
            (void)hipGetLastError();
            return 0;
        }
        std::memset(ctx->box, 0, sizeof(ResidentBox));
        ctx->rlaunches = 0;
    }
    aof_ws_layout L;
    aof_workspace_layout(&p, 1, &L);
    // (the frame pointers of the view are placeholders: the kernel picks the two pinned frames by slot)
    // (the kernel's own copy of the record goes to device memory; the host's comes tagged, below)
    const BatchView v = batch_view(ctx, L, ctx->h_frames[0], ctx->h_frames[1], (int64_t)bytes, ctx->d_blocks, ctx->d_subdirs,
                                   ctx->d_flow, ctx->d_ws);
    SmallArgs sm;
    if (!small_args(ctx, v, 1, &sm)) return 0;
    ResidentBox *box = ctx->box;
    std::memcpy(ctx->h_frames[slot], frame, bytes);
    uint32_t seq = ++ctx->rseq;
    if (seq == 0) seq = ++ctx->rseq;   // 0 means "no request" to the kernel
    const uint32_t tag = seq << 24;
    retag_stale_record(ctx->h_flow, tag);
    __atomic_store_n(&box->word, resident_word(seq, slot, ctx->rframe_req[1 - slot]), __ATOMIC_RELEASE);   // the frame bytes first
    ctx->rframe_req[slot] = seq;
    // The clock of the request: restarted whenever a launch returns -- the FIRST launch of the kernel in a
    // process loads its code object and creates the stream's hardware queue inside hipLaunchKernelGGL, which
    // takes longer than any answer (measured: aof_stream_stats.launch_call_us_max), and that is not the
    // kernel failing to answer.
    auto t0 = std::chrono::steady_clock::now();
    bool launched = false, start_seen = true;
    for (unsigned spins = 0;;) {
        if (read_tagged_record(ctx->h_flow, tag, flow)) break;
        if (!__atomic_load_n(&box->running, __ATOMIC_ACQUIRE)) {
            // not there (first call, or it left on its idle / lifetime deadline): start it behind its
            // predecessor, serving from the last request that one completed
            if (read_tagged_record(ctx->h_flow, tag, flow)) break;
            __atomic_store_n(&box->running, 1u, __ATOMIC_RELEASE);
            const auto l0 = std::chrono::steady_clock::now();
            const int lrc = launch_flow_resident(sm, box, ctx->h_flow, ctx->h_frames[0], ctx->h_frames[1],
                                                 __atomic_load_n(&box->done, __ATOMIC_ACQUIRE), ++ctx->rlaunches,
                                                 kResidentIdleTicks, kResidentLifeTicks, ctx->rdeaf, ctx->rstream);
            if (lrc) {
                // nothing was enqueued: the flag is the host's to take back
                __atomic_store_n(&box->running, 0u, __ATOMIC_RELEASE);
                ctx->resident_on = false;
                ctx->have_prev = false;
                return fail(ctx, -EIO, "resident kernel launch: %s", hipGetErrorString((hipError_t)lrc));
            }
            t0 = std::chrono::steady_clock::now();
            const float us = (float)(std::chrono::duration<double>(t0 - l0).count() * 1e6);
            if (us > ctx->stats.launch_call_us_max) ctx->stats.launch_call_us_max = us;
            ctx->stats.resident_launches++;
            launched = true;
            start_seen = false;
            continue;
        }
        if (!start_seen && __atomic_load_n(&box->started, __ATOMIC_ACQUIRE) == ctx->rlaunches) {
            // launch return -> the kernel's first instruction on the device, with no HIP call in between
            const float us = (float)(seconds_since(t0) * 1e6);
            if (us > ctx->stats.start_latency_us_max) ctx->stats.start_latency_us_max = us;
            start_seen = true;
        }
        if ((++spins & 0x3FFu) == 0 && seconds_since(t0) > kResidentHostTimeoutS) {
            // no answer: stop it, leave the resident mode and let the caller's frame take the graph path
            const hipError_t q = hipStreamQuery(ctx->rstream);
            std::snprintf(ctx->stats.last_report, sizeof(ctx->stats.last_report),
                          "request %u unanswered for %.0f ms%s: record word %08x, launch %u, started %u, served %u, "
                          "exited at %u, on device %u, stream %s, longest launch call %.0f us",
                          seq, seconds_since(t0) * 1e3, launched ? " after this call's launch returned" : "",
                          (unsigned)reinterpret_cast<volatile uint32_t *>(ctx->h_flow)[2], ctx->rlaunches,
                          (unsigned)box->started, (unsigned)box->done, (unsigned)box->exited, (unsigned)box->running,
                          q == hipSuccess ? "drained" : q == hipErrorNotReady ? "busy" : hipGetErrorString(q),
                          ctx->stats.launch_call_us_max);
            std::fprintf(stderr, "aof: the resident kernel did not answer (%s): falling back to one launch per call\n",
                         ctx->stats.last_report);
            ctx->stats.resident_fallbacks++;
            (void)resident_stop(ctx);   // (if it does not leave either, its buffers are abandoned: host_dirty)
            ctx->resident_on = false;
            (void)hipGetLastError();
            return 0;
        }
    }
    ctx->stats.resident_served++;
    ctx->cur_slot = slot;
    *served = true;
    return 0;
}

int aof_set_stream_resident(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    if (on < 0) return (ctx->box && __atomic_load_n(&ctx->box->running, __ATOMIC_ACQUIRE)) ? 1 : 0;
    if (!on) { DeviceGuard guard(ctx->device); (void)resident_stop(ctx); }
    ctx->resident_on = on != 0;
    return 0;
}

int aof_debug_resident_fault(aof_ctx *ctx, int deaf, uint32_t stop_wait_us)
{
    if (!ctx) return -EINVAL;
    ctx->rdeaf = deaf != 0;
    ctx->rstop_wait_s = stop_wait_us ? stop_wait_us * 1e-6 : 1.0;
    return 0;
}

int aof_stream_get_stats(const aof_ctx *ctx, aof_stream_stats *out)
{
    if (!ctx || !out) return -EINVAL;
    *out = ctx->stats;
    return 0;
}

int aof_set_vote_deadline_us(aof_ctx *ctx, uint32_t microseconds)
{
    if (!ctx) return -EINVAL;
    // below 100 us every finaliser wave would give up on its first polls, write a zero record and raise the sticky fault
    // word: one call would disable the context for good
    if (microseconds < 100u) return fail(ctx, -EINVAL, "vote deadline of %u us: at least 100 us", microseconds);
    ctx->vote_deadline_ticks = microseconds > 10000000u ? 1000000000u : microseconds * 100u;   // 100 MHz counter
    return 0;
}

int aof_debug_vote_deadline_ticks(aof_ctx *ctx, uint32_t ticks)
{
    if (!ctx) return -EINVAL;
    ctx->vote_deadline_ticks = ticks;   // (fault injection: 0 makes every finaliser wave give up at once)
    return 0;
}

int aof_set_stream_graph(aof_ctx *ctx, int on)
{
    if (!ctx) return -EINVAL;
    if (on < 0) return (ctx->push_graph[0] || ctx->push_graph[1]) ? 1 : 0;
    ctx->graph_disabled = on == 0;
    return 0;
}

int aof_stream_reset(aof_ctx *ctx)
{
    if (!ctx) return -EINVAL;
    ctx->have_prev = false;
    return 0;
}

}  // extern "C"
