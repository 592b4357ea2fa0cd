// K2, pruned 8x8 search on DENSE grids, column walk: host side (segments, launch geometry).  The kernels are templates in
// aof_cols8_kernels.hpp, each shipped instantiation compiled in its own translation unit (k_cols8_*.hip).
#include <hip/hip_runtime.h>

#include "aof_cols8_plan.hpp"
#include "aof_internal.hpp"
#include "aof_lane8_launch.hpp"

namespace aof {

bool lane8_cols_supported(const SearchArgs &a)
{
    // a dense grid whose next block row is the same window eight rows further down; one unit per column and segment
    return lane8_supported(a) && a.grid.step_y == 8 && a.grid.nx >= 16 && a.grid.ny >= 2 && a.grid.blocks() > 256;
}

bool lane8_cols_votes_supported(const SearchArgs &a, const VoteMem &votes, int64_t capacity_pairs)
{
    const int n = 2 * (2 * a.hist_range + 1) + 1;
    if (!lane8_cols_supported(a) || !votes.base || !votes.fault || n > 62 || votes.stride < (uint32_t)(2 + 2 * n)) return false;
    return a.n_pairs <= capacity_pairs;   // (one launch: capacity is far below the 31-bit unit index)
}

int launch_search_lane8_cols(const SearchArgs &a, void *stream, PruneReport *report, const FlowTail *tail, const VoteMem *votes)
{
    if (report) report->expected = 0;
    if (a.n_pairs == 0) return 0;
    if (!lane8_cols_supported(a) || (a.subpixel && !a.subdirs)) return (int)hipErrorInvalidValue;
    auto segments = [&](int len) {
        ColsSegments g;
        g.len = len > a.grid.ny ? a.grid.ny : len;
        g.segs = (a.grid.ny + g.len - 1) / g.len;
        g.units_per_pair = (uint32_t)((g.segs * a.grid.nx + 63) / 64 * 64);
        g.div_units = fastdiv_make(g.units_per_pair);
        return g;
    };
    auto waves = [&](int len) { return a.n_pairs * (int64_t)(segments(len).units_per_pair / 64); };
    int len = kColsMaxRows;
    while (len > kColsMinRows && waves(len) < kColsWavesWanted) len--;
    ColsPlan plan;
    plan.head = segments(len);
    plan.tail = segments(len / 2 < kColsMinRows ? kColsMinRows : len / 2);
    plan.div_nx = fastdiv_make((uint32_t)a.grid.nx);
    // (columns eight bytes apart: every lane of a pair is misaligned alike and finds its last bytes in the next column's load)
    plan.aligned = (a.grid.step_x == 8 && a.w % 4 == 0 && a.pair_stride % 4 == 0 && reinterpret_cast<uintptr_t>(a.cur) % 4 == 0 &&
                    ((int64_t)a.w * a.h) % 4 == 0) ? 1u : 0u;
    const int64_t per = 0x7FFF0000ll / plan.tail.units_per_pair;   // pairs per launch: units are indexed with 31 bits
    for (int64_t done = 0; done < a.n_pairs; done += per) {
        SearchArgs s = a;
        s.n_pairs = a.n_pairs - done < per ? a.n_pairs - done : per;
        s.prev += done * a.pair_stride;
        s.cur += done * a.pair_stride;
        s.blocks += done * a.grid.blocks();
        if (s.subdirs) s.subdirs += done * a.grid.blocks();
        if (s.pred) s.pred += done;
        if (s.sums) s.sums += done * 4;
        // pairs beyond the last full generation of wave slots (256 CUs x 16 waves) get the short segments
        const int64_t wpp = plan.head.units_per_pair / 64, all = s.n_pairs * wpp, rest = all % kWaveSlots;
        int64_t tail_pairs = 0;
        if (rest != 0 && rest * 5 < kWaveSlots * 4 && plan.tail.len < plan.head.len && all > kWaveSlots) tail_pairs = (rest + wpp - 1) / wpp;
        plan.head_pairs = (uint32_t)(s.n_pairs - tail_pairs);
        plan.head_units = plan.head_pairs * plan.head.units_per_pair;
        const int64_t units = (int64_t)plan.head_units + tail_pairs * plan.tail.units_per_pair;
        // a launch of one or two generations of waves ends sooner with one-wave workgroups, whose slots free up wave by
        // wave for the other batch in flight (as flat_threads does for the exhaustive kernel, k_search_lane8.hip)
        // (one-wave workgroups for launches of one or two generations of waves, as the exhaustive kernel has them, measure
        //  the same within the spread of 200-step runs: profiles/r05_small_launch_shape.txt)
        const int threads = kColsThreads;
        const int64_t wgs = (units + threads - 1) / threads;
        PruneReport rep = {nullptr, 0, 1, 0};
        if (report && report->slots && done == 0) {
            rep = *report;
            rep.stride = (uint32_t)((wgs + kPruneSlots - 1) / kPruneSlots);
            report->stride = rep.stride;
            report->expected = (uint32_t)((wgs + rep.stride - 1) / rep.stride);
        }
        int rc;
        if (tail && votes) {
            if (done != 0 || s.n_pairs != a.n_pairs) return (int)hipErrorInvalidValue;   // (lane8_cols_votes_supported)
            ColsVotes cv;
            cv.search_wgs = (uint32_t)wgs;
            cv.tail = *tail;
            cv.votes = *votes;
            const int64_t finalisers = (s.n_pairs + (threads >> 6) - 1) / (threads >> 6);   // one wave per pair
            rc = (s.subpixel ? launch_k_flow_lane8_cols_t : launch_k_flow_lane8_cols_f)(s, plan, rep, cv, (uint32_t)(wgs + finalisers), threads, stream);
        } else {
            rc = (s.subpixel ? launch_k_search_lane8_cols_t : launch_k_search_lane8_cols_f)(s, plan, rep, (uint32_t)wgs, threads, stream);
        }
        if (rc) return rc;
    }
    return 0;
}

}  // namespace aof
