// Segment plan and launch constants of the column walk (k_search_cols8.hip fills the plan, aof_cols8_kernels.hpp walks it).
#pragma once

#include <cstdint>

#include "aof_internal.hpp"

namespace aof {

// (outside the anonymous namespace: profilers print kernel names with their parameter types, and the tools cut the
//  name at the first "(anonymous namespace)::")
struct ColsSegments {
    int32_t segs, len;          // segments per column, block rows per segment
    uint32_t units_per_pair;    // segs * nx padded to a multiple of 64
    FastDiv div_units;
};
// The launch's pairs [0, head_pairs) are cut into `head` segments; the pairs behind them -- the part of the launch that
// would fill the device's wave slots only partly, at the end -- into segments half as long (`tail`), so that the last
// waves to start are the short ones and the launch does not end on a third of the device (workgroups run in launch order).
struct ColsPlan {
    ColsSegments head, tail;
    uint32_t head_pairs, head_units;   // head_units = head_pairs * head.units_per_pair
    FastDiv div_nx;
    uint32_t aligned;                  // rows, pair strides and frame bases are multiples of four bytes (window_row)
};
// VOTE: the reduction in the same launch (aof_reduce.hpp: vote_and_arrive / await_votes_and_finalise, as in
// k_flow_lane8_flat).  Workgroups [0, search_wgs) search and vote, the ones behind them are finalisers, one wave per pair.
struct ColsVotes {
    uint32_t search_wgs;
    FlowTail tail;
    VoteMem votes;
};


constexpr int kColsThreads = 256;
// Block rows a lane walks: as many as leave the launch three quarters of a generation of waves (256 CUs x 16), between 2
// and 8.  Longer segments load less (a segment's first block loads a whole window) and carry their hints further; shorter
// ones fill the device on small launches (256 VGA pairs, two batches in flight: 6 rows 38.9 us per step, 8: 41.2, 2: 45.3;
// 1 024 pairs: 8 rows 148 us, 3: 160; profiles/r04_p8_column_walk.txt).
constexpr int kColsMaxRows = 8, kColsMinRows = 2;
constexpr int64_t kColsWavesWanted = 3072;
constexpr int64_t kWaveSlots = 4096;   // 256 CUs x 4 SIMDs x 4 waves of this kernel

}  // namespace aof
