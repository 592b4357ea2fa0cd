// K2 (lane8), host side: which of the lane-per-block kernels serves a launch, and the launch geometry.  The kernels
// themselves are templates in aof_lane8_kernels.hpp, each shipped instantiation compiled in its own translation unit
// (k_lane8_*.hip) together with the launch function declared below.
#include <hip/hip_runtime.h>

#include "aof_internal.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

namespace {
constexpr int kThreads = 256;   // block size of the pruned and grouped kernels (aof_lane8_kernels.hpp)
}

// One launch indexes its (pair, block) items with 31 bits; larger batches are cut into several.
constexpr int64_t kMaxItems = 0x7FFF0000ll;

bool lane8_supported(const SearchArgs &a)
{
    if (a.tile != 8 || a.search != 4) return false;
    const int64_t frame = (int64_t)a.w * a.h;
    const int nb = a.grid.blocks();
    if (frame > (1 << 24) || nb < 1 || nb >= (1 << 24)) return false;   // 24-bit multiplies in the kernels
    // a wave's lanes address their frames with 32-bit offsets from the wave's first pair
    const int64_t span = 64 / nb + 2;
    if (a.n_pairs > 1 && (a.pair_stride < 0 || span * a.pair_stride + frame > 0xFFFFFFFFll)) return false;
    return true;
}

namespace {

// Runs `launch(slice, pairs_done)` over slices of at most kMaxItems (pair, block) items.
template <typename F>
int for_slices(const SearchArgs &a, F &&launch)
{
    const int nb = a.grid.blocks();
    int64_t per = kMaxItems / nb;
    if (per >= (1 << 24)) per = (1 << 24) - 1;   // pair indices are multiplied in 24 bits
    for (int64_t done = 0; done < a.n_pairs; done += per) {
        SearchArgs s = a;
        s.n_pairs = a.n_pairs - done < per ? a.n_pairs - done : per;
        s.prev += done * a.pair_stride;
        s.cur += done * a.pair_stride;
        s.blocks += done * nb;
        if (s.subdirs) s.subdirs += done * nb;
        if (s.pred) s.pred += done;
        if (s.sums) s.sums += done * 4;
        s.div_nb = fastdiv_make((uint32_t)nb);
        s.div_nx = fastdiv_make((uint32_t)a.grid.nx);
        const int rc = launch(s, done);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

// Workgroup size of the flat kernels.  Large launches: 64, 128 and 256 threads measure the same (512 is
// 4 % slower).  A launch of only a few generations of waves (configs[3]'s per-GPU share: 128 VGA pairs
// are 2.3 generations) ends sooner with one-wave workgroups, whose slots free up wave by wave.
constexpr int kSmallLaunchThreads = 64;
static int flat_threads(int64_t items)
{
    return items < 6 * 256 * 1024 ? kSmallLaunchThreads : kThreads;   // fewer than six generations of 256 CUs x 16 waves
}

bool lane8_votes_supported(const SearchArgs &a, const VoteMem &votes, int64_t capacity_pairs)
{
    const int n = 2 * (2 * a.hist_range + 1) + 1;
    if (a.prune || !votes.base || !votes.fault || n > 62 || votes.stride < (uint32_t)(2 + 2 * n)) return false;
    if (a.grid.blocks() <= 64) return false;   // a wave covers at most two pairs
    const int64_t per = kMaxItems / a.grid.blocks();   // pairs per launch slice
    return (a.n_pairs < per ? a.n_pairs : per) <= capacity_pairs;
}

int64_t lane8_chunks(const SearchArgs &a)
{
    return (a.n_pairs * a.grid.blocks() + kThreads - 1) / kThreads;
}

int launch_search_lane8(const SearchArgs &a, void *stream, const FlowTail *tail, const VoteMem *votes, PruneReport *report)
{
    if (report) report->expected = 0;
    if (a.n_pairs == 0) return 0;
    if (a.subpixel && !a.subdirs) return (int)hipErrorInvalidValue;
    return for_slices(a, [&](const SearchArgs &s, int64_t done) {
        const int64_t items = s.n_pairs * s.grid.blocks();
        if (tail && votes) {   // search + reduction in one launch
            const int threads = flat_threads(items);
            const int64_t wgs = (items + threads - 1) / threads;
            const int64_t finalisers = (s.n_pairs + threads / 64 - 1) / (threads / 64);   // one wave per pair
            FlowTail t = *tail;
            t.flows += done;
            if (t.pred) t.pred += done;
            auto fn = s.sums ? (s.subpixel ? launch_k_flow_lane8_flat_tt : launch_k_flow_lane8_flat_ft)
                             : (s.subpixel ? launch_k_flow_lane8_flat_tf : launch_k_flow_lane8_flat_ff);
            return fn(s, (uint32_t)items, (uint32_t)wgs, (uint32_t)(wgs + finalisers), threads, t, *votes, stream);
        }
        const hipStream_t st = static_cast<hipStream_t>(stream);
        if (s.prune) {
            // consecutive chunks per workgroup, so that all but the first inherit start row and verdict -- as many as
            // leave some 1 500 workgroups to the launch (1 024 VGA pairs: 8 chunks 6.29, 4: 6.14, 1: 5.13 M pairs/s;
            // 256 pairs: 3 chunks 4.84, 8: 4.38; profiles/r04_p8_chunks_per_workgroup.txt)
            const int64_t chunks = (items + kThreads - 1) / kThreads;
            const int spw = (int)(chunks / 1536 < 1 ? 1 : (chunks / 1536 > 8 ? 8 : chunks / 1536));
            const int64_t wgs = (chunks + spw - 1) / spw;
            PruneReport rep = {nullptr, 0, 1, 0};
            if (report && report->slots && done == 0) {   // (launches of more than 2^31 items: the first slice reports)
                rep = *report;
                rep.stride = (uint32_t)((wgs + kPruneSlots - 1) / kPruneSlots);
                report->stride = rep.stride;
                report->expected = (uint32_t)((wgs + rep.stride - 1) / rep.stride);
            }
            return (s.subpixel ? launch_k_search_lane8_pruned_t : launch_k_search_lane8_pruned_f)(s, (uint32_t)items, (uint32_t)wgs, spw, rep, st);
        }
        const int threads = flat_threads(items);
        const int64_t wgs = (items + threads - 1) / threads;
        // (EQ = false: a launch without pixel sums runs a kernel without any equalisation code)
        auto fn = s.sums ? (s.subpixel ? launch_k_search_lane8_tt : launch_k_search_lane8_ft)
                         : (s.subpixel ? launch_k_search_lane8_tf : launch_k_search_lane8_ff);
        return fn(s, (uint32_t)items, (uint32_t)wgs, threads, st);
    });
}

// Pairs per workgroup of the grouped kernel; 0 = the grid does not qualify.
int lane8_group(const SearchArgs &a)
{
    const int nb = a.grid.blocks();
    if (!lane8_supported(a) || nb < 8 || nb > kThreads) return 0;  // 1 .. 32 pairs per workgroup
    return kThreads / nb;
}

int launch_flow_lane8(const SearchArgs &a, const FlowTail &tail, void *stream)
{
    if (a.n_pairs == 0) return 0;
    const int ppw = lane8_group(a);
    if (ppw == 0) return (int)hipErrorInvalidValue;
    const int n = 2 * (2 * a.hist_range + 1) + 1;
    if (a.subpixel && !a.subdirs) return (int)hipErrorInvalidValue;
    return for_slices(a, [&](const SearchArgs &s, int64_t done) {
        FlowTail t = tail;
        t.flows += done;
        if (t.pred) t.pred += done;
        const int64_t wgs = (s.n_pairs + ppw - 1) / ppw;
        return (s.subpixel ? launch_k_flow_lane8_t : launch_k_flow_lane8_f)(s, t, ppw, (uint32_t)wgs, (size_t)ppw * 2 * n * sizeof(uint32_t), stream);
    });
}

}  // namespace aof
