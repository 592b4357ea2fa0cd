// K2, pruned 8x8 search on DENSE grids, column walk (DESIGN.md "The 8x8 adaptive search").
//
// The chunk-walking pruned kernel (k_search_lane8.hip) loads the whole 16x16 window of every block (24 row loads per lane)
// and, with equalisation, shifts all of it.  On a dense grid (block step = tile = 8 rows) the window of the block BELOW
// a block is the same window moved down by eight rows.  So here a lane owns a COLUMN of blocks: it walks `len` vertically
// adjacent blocks, keeps rows 8..15 of its window in registers as rows 0..7 of the next one and loads -- and shifts --
// only the eight new rows (16 row loads per block instead of 24); consecutive lanes are consecutive columns, so every
// row load of a wave is one coalesced stretch of a frame row, and the wave's blocks of one step are neighbours -- the
// wave-wide row dropping of pruned_row, the start row and the verdict carried from block to block work as in the chunk
// walk (aof_lane8.hpp).  What it buys is instruction count where there is most of it: the half-pixel and the equalising
// configurations (c3's search kernel 216 -> 175 us, c2h's 267 -> 218 us per 1 024 VGA pairs; plain c2 -1..3 %).
//
// Units: a pair has `segs` segments of `len` block rows; unit = (pair, segment, column), `units_per_pair` of them
// padded to a multiple of 64, so that a wave never straddles two pairs (everything that depends on the pair is scalar).
//
// Templates; every shipped instantiation is compiled in a translation unit of its own (k_cols8_*.hip), like the other
// lane-per-block kernels (aof_lane8_kernels.hpp says why).
#pragma once

#include <type_traits>

#include "aof_cols8_plan.hpp"
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_lane8.hpp"
#include "aof_reduce.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

template <bool SUBPIXEL, bool VOTE>
__device__ __forceinline__ void cols_walk(const SearchArgs &a, const ColsPlan &plan, const PruneReport &report, const VoteMem *votes)
{
    // (workgroups in launch order, no XCD remap: the short segments must start last on every XCD, and the remap buys this
    //  kernel nothing -- 133.5 against 133.5 us per 1 024 VGA pairs.  Short segments for the launch's last pairs: c3's search
    //  kernel 180.8 -> 174.8 us, c2h's 227.7 -> 218.3, c2's 130.0 -> 128.7)
    const uint32_t wg = blockIdx.x;
    const uint32_t unit0 = wg * blockDim.x + threadIdx.x;            // < 2^31 (launcher)
    // the wave's segment class and pair, in scalar registers: units per pair are multiples of 64
    const bool in_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit0) >= plan.head_units;
    const ColsSegments sg = in_tail ? plan.tail : plan.head;
    const uint32_t unit = in_tail ? unit0 - plan.head_units : unit0;
    const uint32_t pair = (in_tail ? plan.head_pairs : 0u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)fast_div(unit, sg.div_units));
    if (pair >= (uint32_t)a.n_pairs) return;                           // (whole waves)
    const uint32_t local = unit - (pair - (in_tail ? plan.head_pairs : 0u)) * sg.units_per_pair;
    const uint32_t seg = fast_div(local, plan.div_nx), bx = local - __umul24(seg, (uint32_t)a.grid.nx);
    const bool live = local < (uint32_t)(sg.segs * a.grid.nx);
    const int by0 = (int)seg * sg.len;
    const int W = a.w;
    constexpr int m = SUBPIXEL ? 1 : 0;

    // per pair: frames (buffer resources: reads outside what is left of the arrays return zero), predictor, equalisation
    const uint64_t span = a.n_pairs > 1 ? (uint64_t)(a.n_pairs - pair) * (uint64_t)a.pair_stride : (uint64_t)a.w * (uint64_t)a.h;
    const uint32_t records = span > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)span;
    const int64_t base = (int64_t)pair * a.pair_stride;
    const __amdgpu_buffer_rsrc_t rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.prev) + base, 0, records, kRawBuffer);
    const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.cur) + base, 0, records, kRawBuffer);
    typedef const __attribute__((address_space(4))) uint32_t *const_u32;   // read-only during the kernel
    int px = 0, py = 0, delta = 0;
    if (a.pred) {
        const uint32_t w3 = ((const_u32)(a.pred + pair))[3];   // quality, flags, pred_x, pred_y
        px = (int8_t)(w3 >> 16); py = (int8_t)(w3 >> 24);
    }
    if (a.sums) {
        const uint32_t npix = (uint32_t)(a.w * a.h);
        const const_u32 sm = (const_u32)(a.sums + (size_t)pair * 4);
        delta = (int)((sm[a.level] + npix / 2) / npix) - (int)((sm[2 + a.level] + npix / 2) / npix);
    }
    const int i = a.grid.x0 + __mul24((int)bx, a.grid.step_x);
    const int wx0 = i + px - 4;
    const bool column_inside = live && wx0 - m >= 0 && wx0 + 16 + m <= a.w;
    uint32_t *const slots = reinterpret_cast<uint32_t *>(a.blocks) + (size_t)pair * (size_t)a.grid.blocks();
    uint8_t *const dirs = SUBPIXEL ? a.subdirs + (size_t)pair * (size_t)a.grid.blocks() : nullptr;

    // Window rows under a predictor start at any byte (wx0 = 8 bx + px), and a 16-byte buffer load that is not dword-aligned
    // costs the pruned search -- whose four waves per SIMD just about cover the aligned loads -- 30 % (C3, same box, every pair shifted
    // alike: px = 0, 4, 8: 142 us per 1 024 pairs, every other px: 184 us; tools/lab/align_probe.py).  The misalignment is
    // the same for every lane and row of a pair (rows are multiples of four bytes: plan.aligned, checked by the launcher).
    // A misaligned pair loads its 16 bytes from the dword below (aligned), which leaves the last `mis` bytes out: they are the
    // first bytes of dword 2 of the NEXT column's load (columns are eight bytes apart) -- one wave shift --, and the lanes
    // without a next column in the wave (the row's last column, lane 63) fetch that dword themselves.
    const uint32_t mis = plan.aligned ? (uint32_t)__builtin_amdgcn_readfirstlane(wx0 & 3) : 0u;   // (wave-uniform)
    const bool lonely = (threadIdx.x & 63u) == 63u || bx + 1u == (uint32_t)a.grid.nx;
    auto shifted_rows = [&](uint32_t off, uint4 *dst) {   // eight window rows from `off` on, through aligned loads
        u32x4 v[8];
        uint32_t e[8];
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off - mis, s * W, 0);
        if (lonely) {
#pragma unroll
            for (int s = 0; s < 8; s++) e[s] = __builtin_amdgcn_raw_buffer_load_b32(rs_cur, off - mis + 16u, s * W, 0);
        }
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[s].z, 0x130, 0xF, 0xF, false);   // wave_shl:1: lane + 1's dword 2
            const uint32_t tail = lonely ? e[s] : next;
            dst[s] = make_uint4(__builtin_amdgcn_alignbyte(v[s].y, v[s].x, mis), __builtin_amdgcn_alignbyte(v[s].z, v[s].y, mis),
                                __builtin_amdgcn_alignbyte(v[s].w, v[s].z, mis), __builtin_amdgcn_alignbyte(tail, v[s].w, mis));
        }
    };
    uint4 win[16];
    // the upper half of the segment's first window (every later block inherits its upper half from the block above).
    // Row bases: rows 0..7 are used only by a block that is searched (then wy0 >= 0), rows 8..15 also by the block below
    // (then wy0 + 8 >= 0), so each half has a base of its own that is a valid offset whenever its rows are used -- the range
    // check of a buffer load need not see the scalar row offset.  Bases of rows nobody uses may wrap: they read zeros.
    {
        const int wy0 = a.grid.y0 + by0 * a.grid.step_y + py - 4;
        const uint32_t off_cur = (uint32_t)(__mul24(wy0, W) + wx0);
        if (mis != 0) {   // (scalar; every lane of the wave: the wave shift reads its neighbours)
            shifted_rows(off_cur, win);
        } else if (live) {
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off_cur, s * W, 0);
                win[s] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
        if (live && delta != 0) {
#pragma unroll
            for (int s = 0; s < 8; s++) win[s] = sat_add_u8x16(win[s], delta);
        }
    }
    // ADAPTIVE lets the first block of every wave run exhaustively and judge; PRUNED (a caller, or a context whose launches
    // have reported that pruning pays) prunes from the first block on -- under a predictor it starts in the centre row,
    // without one the first block votes for the row (vote_start_row, aof_lane8.hpp)
    int start_row = a.pred ? 4 : -1, prune_pays = a.prune == 2 ? 0 : 1;   // (wave-uniform: scalar registers)
    int seen = 0, paying = 0;
    std::conditional_t<VOTE && !SUBPIXEL, WalkVotes, NoWalkVotes> pending;
    pending.init();
    for (int step = 0; step < sg.len; step++) {
        const int by = by0 + step;
        const bool act = live && by < a.grid.ny;
        if (__ballot(act) == 0) break;                                   // (segments end together; the last one is shorter)
        const int j = a.grid.y0 + by * a.grid.step_y;                    // step_y == 8 (launcher)
        const int wy0 = j + py - 4;
        const bool inside = act && column_inside && wy0 - m >= 0 && wy0 + 16 + m <= a.h;
        const uint32_t off_prev = (uint32_t)(__mul24(j, W) + i);
        const uint32_t off_cur8 = (uint32_t)(__mul24(wy0 + 8, W) + wx0);
        uint32_t ref[8][2];
        // the eight new window rows.  Blocks that are not searched load too: the block below inherits these rows.
        if (act) {
#pragma unroll
            for (int r = 0; r < 8; r++) {   // (every byte used once: non-temporal, as in the chunk walk)
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_prev, off_prev, r * W, 2);
                ref[r][0] = v.x; ref[r][1] = v.y;
            }
            if (mis == 0) {
#pragma unroll
                for (int s = 8; s < 16; s++) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off_cur8, (s - 8) * W, 0);
                    win[s] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
        }
        if (mis != 0 && __ballot(act) != 0) shifted_rows(off_cur8, win + 8);   // (every lane: the wave shift; rows of lanes that are not `act` are never used)
        if (act) {
            asm volatile("" : "+v"(win[15].w));   // (one wait for all of the block's rows, in front of the gate)
        }
        if (act && delta != 0) {
#pragma unroll
            for (int s = 8; s < 16; s++) win[s] = sat_add_u8x16(win[s], delta);
        }
        uint32_t gradient = 0;
        if (inside) gradient = gradient_gate(ref);
        const bool need = inside && gradient >= (uint32_t)a.feature_threshold;
        aof_block rec;
        rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
        uint32_t best = 0xFFFFFFFFu;
        const unsigned long long needing = __ballot(need);
        if (needing != 0) {   // (every lane of the wave goes through the search: its ballots see them all)
            const int src = __ffsll((long long)needing) - 1;
            if (prune_pays == 0) {
                int droppable = 0;
                best = exhaustive_search_judged(win, ref, needing, droppable);
                prune_pays = __builtin_amdgcn_readfirstlane(droppable >= kJudgedRowsToPrune ? 1 : 0);
            } else {
                if (start_row < 0) start_row = __builtin_amdgcn_readfirstlane(vote_start_row(win, ref, needing));
                const int dropped = pruned_search(win, ref, needing, start_row, best);
                prune_pays = __builtin_amdgcn_readfirstlane(dropped >= 2 ? 1 : 0);
            }
            start_row = (int)((uint32_t)__builtin_amdgcn_readlane((int)best, src) & 0xFFFFu) / 9;   // where the first live block matched
        }
        seen++;
        paying += prune_pays;
        int subdir = 8;
        if (need) {
            const int idx = (int)(best & 0xFFFFu);
            const int dyi = idx / 9, dxi = idx - 9 * dyi;
            rec.dx = (int8_t)(px + dxi - 4);
            rec.dy = (int8_t)(py + dyi - 4);
            rec.sad = (uint16_t)(best >> 16);
            if constexpr (SUBPIXEL) {
                if ((uint32_t)rec.sad < (uint32_t)a.value_threshold) {
                    // the ring of a match that is not on the window's rim is in the window registers (aof_lane8.hpp)
                    const unsigned long long refining = __ballot(true);
                    const int udy = __builtin_amdgcn_readlane(dyi, __ffsll((long long)refining) - 1);
                    const bool have_ring = dyi == udy && udy >= 1 && udy <= 7 && dxi >= 1 && dxi <= 7;
                    RefineState<2> st;
                    st.init();
                    if (have_ring) refine_from_window(win, udy, dxi - 1, ref, st, (uint32_t)rec.sad);
                    if (!have_ring && rec.sad != 0) {   // (nothing is below a SAD of zero)
                        const uint32_t ring = off_cur8 + (uint32_t)((dyi - 9) * W + (dxi - 1));   // (the match's row - 1: >= 0)
                        uint32_t rows[10][3];
                        load_ring(rs_cur, ring, W, records, rows);
                        for_rows<-1, 8>([&](auto yc) {
                            constexpr int Y = decltype(yc)::value;
                            uint32_t d[3] = {rows[Y + 1][0], rows[Y + 1][1], rows[Y + 1][2]};
                            if (delta != 0) {
#pragma unroll
                                for (int q = 0; q < 3; q++) d[q] = sat_add_u8x4(d[q], delta);
                            }
                            st.template row<Y>(d, ref);
                        });
                    }
                    subdir = st.direction(rec.sad);
                }
            }
        }
        if (act) {
            const uint32_t slot = (uint32_t)(by * a.grid.nx) + bx;
            slots[slot] = __builtin_bit_cast(uint32_t, rec);
            if constexpr (SUBPIXEL) dirs[slot] = (uint8_t)subdir;
        }
        if constexpr (VOTE) {
            // a wave lies inside one pair: the steps of its walk that agree on the motion vote and arrive together (WalkVotes)
            const bool ok = act && (uint32_t)rec.sad < (uint32_t)a.value_threshold;   // skipped = 0xFFFF
            int hx = 0, hy = 0;
            if constexpr (SUBPIXEL) {
                hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
                hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
            }
            const int centre = 2 * a.hist_range + 1;
            // (the half-pixel walk sits at 128 VGPRs with nothing to spare for the pending votes: it adds step by step)
            if constexpr (SUBPIXEL) vote_and_arrive(*votes, a.hist_range, pair, act, ok, 2 * rec.dx + hx + centre, 2 * rec.dy + hy + centre);
            else pending.step(*votes, a.hist_range, pair, act, ok, 2 * rec.dx + hx + centre, 2 * rec.dy + hy + centre);
        }
        // the lower half of this window is the upper half of the next block's
#pragma unroll
        for (int s = 0; s < 8; s++) win[s] = win[s + 8];
    }
    if constexpr (VOTE) pending.flush(*votes, a.hist_range, pair);
    // one workgroup in report.stride tells the host how its first wave fared (aof_internal.hpp: PruneReport)
    if (report.slots && threadIdx.x == 0 && wg % report.stride == 0)
        __hip_atomic_store(report.slots + wg / report.stride, (report.launch_no << 16) | ((uint32_t)paying << 8) | (uint32_t)seen,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool SUBPIXEL>
__global__ __launch_bounds__(kColsThreads, 4) void k_search_lane8_cols(SearchArgs a, ColsPlan plan, PruneReport report)
{
    cols_walk<SUBPIXEL, false>(a, plan, report, nullptr);
}

// The same walk with the reduction in the launch: no K3 behind it (launches of up to the context's vote records).
template <bool SUBPIXEL>
__global__ __launch_bounds__(kColsThreads, 4) void k_flow_lane8_cols(SearchArgs a, ColsPlan plan, PruneReport report, ColsVotes cv)
{
    if (blockIdx.x >= cv.search_wgs) {   // (uniform in the workgroup)
        const uint32_t pair = (blockIdx.x - cv.search_wgs) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (pair < (uint32_t)a.n_pairs) await_votes_and_finalise(cv.votes, cv.tail, pair);
        return;
    }
    cols_walk<SUBPIXEL, true>(a, plan, report, &cv.votes);
}

}  // namespace

}  // namespace aof
