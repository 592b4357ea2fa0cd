// The lane-per-block 8x8 kernels (DESIGN.md "Kernels": K2) as templates.  Every shipped instantiation is compiled in a
// translation unit of its OWN (k_lane8_*.hip, one kernel each, with its launch function): the code hipcc emits for a
// kernel depends on which other kernels of its translation unit inline the same instantiation of search_block
// (aof_lane8.hpp; LAB_LOG.md rounds 4 and 5: adding a kernel moved the ISA of untouched ones by 2.5 %), so kernels
// no longer share one.  tests/test_isa_hashes.py holds every kernel's ISA to the committed hash.
//
// K2 (lane8) -- the dominant kernels: 8x8 SAD search over +-4 px on ANY grid, frame width and
// predictor (DESIGN.md "Spec": Search; "Kernels": K2).  The dense grids of BASELINE configs[1..3]
// as well as the published sparse PX4Flow grid of configs[0].
//
// One LANE per block, 256 consecutive (pair, block) items per workgroup, no LDS, no barriers.  A
// lane reads its 8x8 reference tile (8 unaligned 8-byte loads) and its 16 search rows (16
// unaligned 16-byte loads) straight from global memory, all issued before the first use, and
// evaluates all 81 candidates with v_qsad_pk_u16_u8 -- four horizontally sliding 4-byte SADs
// per instruction, packed u16 accumulators (max 64*255 = 16320 fits) -- plus v_sad_hi_u8 for
// the ninth column, which accumulates straight into the high half of a register pre-loaded
// with the candidate index.  No cross-lane traffic: the arg-min is a per-lane v_min3_u32 tree
// over the packed keys (sad << 16 | idx), i.e. "first minimum in scan order wins".  Half-pixel
// refinement, when enabled, follows in the same lane from the ring of the best match
// (aof_refine.hpp).
//
// Why no LDS staging: on a dense grid neighbouring lanes read neighbouring 8-byte columns, so
// a wave's row load is one contiguous run; vertically adjacent blocks share half their rows
// through L1/L2 (workgroup ids are remapped so that consecutive block rows of a pair stay on one
// XCD); HBM sees every frame byte once (PMC: 624 MB read per 629 MB of frames).  Against an
// LDS-strip kernel (round 1's first dominant kernel, removed in round 3) all 64 lanes of every
// wave work, nothing waits at a barrier and there is no staging phase to hide: 8 % faster on
// C2, 16 % with half-pixel refinement (profiles/r01_p_lane8_vs_strips.txt).
//
// Variants: k_search_lane8 (flat items, K3 follows), k_flow_lane8_flat (flat items, the reduction in
// the same launch: aof_set_reduce_fusion), k_flow_lane8 (grids of 8..256 blocks: a workgroup owns
// whole pairs and finalises their flow records itself), k_search_lane8_pruned (AOF_SEARCH_PRUNED:
// exact partial-distortion elimination, see pruned_row in aof_lane8.hpp).
#pragma once

#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_lane8.hpp"
#include "aof_reduce.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;  // 64, 128 and 256 measure the same, 512 is 4 % slower

// Flat mapping: 256 consecutive (pair, block) items per workgroup; K3 follows.
// Four waves per SIMD (<= 128 VGPRs): a lane has all 24 of its row loads in flight at once and
// the other three waves of the SIMD cover that round trip.
// PRUNE: a workgroup walks `spw` consecutive 256-item chunks (block rows further down the same
// frame) and each wave carries the dy row where its previous chunk matched.
template <bool SUBPIXEL, bool PRUNE, bool EQ, bool VOTE = false>
__device__ __forceinline__ void search_chunks(const SearchArgs &a, uint32_t items, uint32_t total_wgs, int spw,
                                              const FlowTail *tail = nullptr, const VoteMem *votes = nullptr,
                                              const PruneReport *report = nullptr)
{
    // consecutive workgroups = consecutive block rows of one pair: keep them on one XCD, whose L2
    // then serves the search rows that vertically adjacent blocks share
    const uint32_t wg = xcd_remap(blockIdx.x, total_wgs);
    const uint32_t nb = (uint32_t)a.grid.blocks();
    // ADAPTIVE lets the first chunk run exhaustively and judge; PRUNED prunes from the first chunk on: in the centre row
    // under a predictor, otherwise in the row its first chunk votes for (start_row < 0: vote_start_row, aof_lane8.hpp)
    int start_row = !PRUNE || a.pred ? 4 : -1, prune_pays = PRUNE && a.prune == 2 ? 0 : 1;
    int chunks_seen = 0, chunks_paying = 0;   // (PRUNE: what this wave reports, below)
    for (int c = 0; c < spw; c++) {
        const uint32_t item0 = (wg * (uint32_t)spw + (uint32_t)c) * blockDim.x;   // < 2^31 (launcher)
        const uint32_t item = item0 + threadIdx.x;
        const bool live = item < items;
        if (!PRUNE && !VOTE && !live) return;
        if (PRUNE && item0 >= items) break;    // whole workgroup past the end (uniform)
        const uint32_t pair = live ? fast_div(item, a.div_nb) : 0u;
        aof_block rec;
        rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
        const int subdir = search_block<SUBPIXEL, PRUNE, EQ, VOTE && !SUBPIXEL && EQ>(a, pair, live ? item - __umul24(pair, nb) : 0u, item0, live,
                                                             rec, start_row, prune_pays);
        if constexpr (PRUNE) {
            chunks_seen++;
            chunks_paying += __builtin_amdgcn_readfirstlane(prune_pays);
        }
        if constexpr (VOTE) {
            // The reduction in the same launch: every lane stays until here and the wave adds its votes
            // to the records of the one or two pairs it covers (a pair has more than 64 blocks), without
            // waiting for anything (aof_reduce.hpp).  Item and pair are derived again from an
            // opaque copy of the lane id, so that none of it occupies a register during the search
            // (the kernel sits at the 128 VGPRs of four waves per SIMD).
            uint32_t lane_id = threadIdx.x;
            asm volatile("" : "+v"(lane_id));
            const uint32_t item2 = item0 + lane_id;
            const bool live2 = item2 < items;
            // the wave's first pair and where the next one begins, in scalar registers (a wave covers at
            // most two pairs; lane 0 is live or the whole wave is past the end)
            const uint32_t first_item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item2);
            const uint32_t lead = fast_div(first_item, a.div_nb);
            const uint32_t next_pair_at = (lead + 1) * nb;
            const uint32_t rec32 = __builtin_bit_cast(uint32_t, rec);
            const bool ok = live2 && (rec32 >> 16) < (uint32_t)a.value_threshold;   // skipped = 0xFFFF
            int hx = 0, hy = 0;
            if (SUBPIXEL) {
                hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
                hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
            }
            const int centre = 2 * a.hist_range + 1;
            const int bin_x = 2 * rec.dx + hx + centre, bin_y = 2 * rec.dy + hy + centre;
            vote_and_arrive(*votes, a.hist_range, lead, live2 && item2 < next_pair_at, ok, bin_x, bin_y);
            if (first_item + 63 >= next_pair_at)   // (scalar) the wave reaches into the next pair
                vote_and_arrive(*votes, a.hist_range, lead + 1, live2 && item2 >= next_pair_at, ok, bin_x, bin_y);
        }
    }
    if constexpr (PRUNE) {
        // one workgroup in report->stride tells the host how its first wave fared (aof_internal.hpp: PruneReport)
        if (report->slots && threadIdx.x == 0 && wg % report->stride == 0)
            __hip_atomic_store(report->slots + wg / report->stride,
                               (report->launch_no << 16) | ((uint32_t)chunks_paying << 8) | (uint32_t)chunks_seen,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool SUBPIXEL, bool EQ>
__global__ __launch_bounds__(kThreads, 4) void k_search_lane8(SearchArgs a, uint32_t items, uint32_t total_wgs, int spw)
{
    search_chunks<SUBPIXEL, false, EQ>(a, items, total_wgs, spw);
}

// The same search with the reduction in the launch: no K3 behind it.  Workgroups [0, search_wgs) search
// and vote through agent-scope atomics; the workgroups behind them are finalisers, one wave per pair,
// which wait for their pair's votes and write its flow record (aof_reduce.hpp).
template <bool SUBPIXEL, bool EQ>
__global__ __launch_bounds__(kThreads, 4) void k_flow_lane8_flat(SearchArgs a, uint32_t items, uint32_t search_wgs,
                                                                 FlowTail tail, VoteMem votes)
{
    if (blockIdx.x >= search_wgs) {   // (uniform in the workgroup)
        const uint32_t pair = (blockIdx.x - search_wgs) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (pair < (uint32_t)a.n_pairs) await_votes_and_finalise(votes, tail, pair);
        return;
    }
    // Every kernel argument the search reads, fetched HERE in one batch of scalar loads: with the
    // finaliser branch above, the compiler otherwise sinks them into the blocks that use them -- seven
    // dependent s_load / s_waitcnt round trips in front of the row loads of every wave (+7 % on the
    // 1 024-pair launch).
    asm volatile("" ::"s"(a.prev), "s"(a.cur), "s"(a.pair_stride), "s"(a.w), "s"(a.h), "s"(a.grid.x0), "s"(a.grid.y0),
                 "s"(a.grid.step_x), "s"(a.grid.step_y), "s"(a.grid.nx), "s"(a.grid.ny), "s"(a.feature_threshold),
                 "s"(a.value_threshold));
    asm volatile("" ::"s"(a.blocks), "s"(a.subdirs), "s"(a.pred), "s"(a.sums), "s"(a.level), "s"(a.n_pairs),
                 "s"(a.hist_range), "s"(a.div_nb.mul), "s"(a.div_nb.shift), "s"(a.div_nx.mul), "s"(a.div_nx.shift),
                 "s"(items), "s"(votes.base), "s"(votes.stride));
    search_chunks<SUBPIXEL, false, EQ, true>(a, items, search_wgs, 1, &tail, &votes);
}

// The pruned search holds both code paths (pruned rows and the exhaustive scan that judges) and keeps the whole
// window live across a data-dependent loop.  Both work dy row by dy row on five accumulator registers, so the
// kernel stays within the 128 VGPRs of four waves per SIMD (122; 148 and three waves while its exhaustive path
// carried the 45 accumulators of exhaustive_search).
template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads, 4) void k_search_lane8_pruned(SearchArgs a, uint32_t items, uint32_t total_wgs,
                                                                      int spw, PruneReport report)
{
    search_chunks<SUBPIXEL, true, true>(a, items, total_wgs, spw, nullptr, nullptr, &report);
}

// Grouped mapping for grids of a few dozen blocks (the published sparse grid): a workgroup owns
// `ppw` WHOLE pairs, so their votes meet in LDS and one lane per pair finalises the flow record
// -- no K3 launch, no second pass over the records.
template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads, 4) void k_flow_lane8(SearchArgs a, FlowTail tail, int ppw)
{
    extern __shared__ uint32_t s_votes[];  // [ppw][2][n]
    const int nb = a.grid.blocks(), tid = threadIdx.x;
    const int centre = 2 * a.hist_range + 1, n = 2 * centre + 1;
    const uint32_t pair0 = blockIdx.x * (uint32_t)ppw;   // pairs * blocks < 2^31 per launch (launcher)
    const int np = (int)min((int64_t)ppw, a.n_pairs - (int64_t)pair0);
    for (int k = tid; k < ppw * 2 * n; k += kThreads) s_votes[k] = 0;
    __syncthreads();
    const bool live = tid < np * nb;
    const int p = live ? (int)fast_div((uint32_t)tid, a.div_nb) : 0, blk = live ? tid - p * nb : 0;
    aof_block rec;
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    int subdir = 8;
    int start_row = 4, prune_pays = 1;  // (unused: the grouped kernel always searches exhaustively)
    // (a live lane's record index (pair0 + p) * nb + blk is pair0 * nb + tid)
    if (live) subdir = search_block<SUBPIXEL, false, true, !SUBPIXEL>(a, pair0 + (uint32_t)p, (uint32_t)blk, pair0 * (uint32_t)nb, true, rec,
                                                     start_row, prune_pays);
    // (the lane's pair is derived again from an opaque copy of its id: nothing of it holds a register
    //  during the search, which sits at the 128 VGPRs of four waves per SIMD)
    uint32_t tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const bool live2 = (int)tid2 < np * nb;
    const int p2 = live2 ? (int)fast_div(tid2, a.div_nb) : 0;
    const bool ok = live2 && (uint32_t)rec.sad < (uint32_t)a.value_threshold;  // skipped = 0xFFFF
    const int hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
    const int hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
    wave_vote2(s_votes, s_votes, p2 * 2 * n + 2 * rec.dx + hx + centre, p2 * 2 * n + n + 2 * rec.dy + hy + centre, ok);
    __syncthreads();
    if ((int)tid2 < np) {
        const int tid = (int)tid2;
        const uint32_t *hxp = s_votes + tid * 2 * n, *hyp = hxp + n;
        int sums[3] = {0, 0, 0};
        for (int k = 0; k < n; k++) {
            sums[0] += (k - centre) * (int)hxp[k];
            sums[1] += (k - centre) * (int)hyp[k];
            sums[2] += (int)hxp[k];
        }
        finalise_flow(tail, pair0 + tid, hxp, hyp, sums);
    }
}

}  // namespace

}  // namespace aof
