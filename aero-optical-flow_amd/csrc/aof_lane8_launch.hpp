// Launch functions of the lane-per-block 8x8 kernels, one per translation unit (k_lane8_*.hip, k_cols8_*.hip: ONE kernel
// instantiation each, see aof_lane8_kernels.hpp).  All enqueue on `stream` and return the hipError_t of the launch.
#pragma once

#include <cstdint>

#include "aof_internal.hpp"

namespace aof {

struct ColsPlan;
struct ColsVotes;

// k_search_lane8<SUBPIXEL, EQ>: flat items, K3 follows
int launch_k_search_lane8_ff(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream);
int launch_k_search_lane8_ft(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream);
int launch_k_search_lane8_tf(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream);
int launch_k_search_lane8_tt(const SearchArgs &a, uint32_t items, uint32_t wgs, int threads, void *stream);
// k_flow_lane8_flat<SUBPIXEL, EQ>: flat items, the reduction in the launch (grid = search + finaliser workgroups)
int launch_k_flow_lane8_flat_ff(const SearchArgs &a, uint32_t items, uint32_t search_wgs, uint32_t grid, int threads, const FlowTail &t, const VoteMem &v, void *stream);
int launch_k_flow_lane8_flat_ft(const SearchArgs &a, uint32_t items, uint32_t search_wgs, uint32_t grid, int threads, const FlowTail &t, const VoteMem &v, void *stream);
int launch_k_flow_lane8_flat_tf(const SearchArgs &a, uint32_t items, uint32_t search_wgs, uint32_t grid, int threads, const FlowTail &t, const VoteMem &v, void *stream);
int launch_k_flow_lane8_flat_tt(const SearchArgs &a, uint32_t items, uint32_t search_wgs, uint32_t grid, int threads, const FlowTail &t, const VoteMem &v, void *stream);
// k_search_lane8_pruned<SUBPIXEL>: a workgroup walks spw chunks of 256 items
int launch_k_search_lane8_pruned_f(const SearchArgs &a, uint32_t items, uint32_t wgs, int spw, const PruneReport &rep, void *stream);
int launch_k_search_lane8_pruned_t(const SearchArgs &a, uint32_t items, uint32_t wgs, int spw, const PruneReport &rep, void *stream);
// k_flow_lane8<SUBPIXEL>: grids of 8..256 blocks, ppw whole pairs per workgroup, lds bytes of vote histograms
int launch_k_flow_lane8_f(const SearchArgs &a, const FlowTail &t, int ppw, uint32_t wgs, size_t lds, void *stream);
int launch_k_flow_lane8_t(const SearchArgs &a, const FlowTail &t, int ppw, uint32_t wgs, size_t lds, void *stream);
// k_search_lane8_cols<SUBPIXEL> / k_flow_lane8_cols<SUBPIXEL>: the column walk of the pruned search on dense grids
int launch_k_search_lane8_cols_f(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, uint32_t wgs, int threads, void *stream);
int launch_k_search_lane8_cols_t(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, uint32_t wgs, int threads, void *stream);
int launch_k_flow_lane8_cols_f(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, const ColsVotes &cv, uint32_t grid, int threads, void *stream);
int launch_k_flow_lane8_cols_t(const SearchArgs &a, const ColsPlan &plan, const PruneReport &rep, const ColsVotes &cv, uint32_t grid, int threads, void *stream);

}  // namespace aof
