// K2 (lane8) -- 8x8 SAD search over +-4 px for ANY grid, frame width and predictor: the
// published sparse PX4Flow grid (BASELINE configs[0]) and dense grids whose rows are not a
// multiple of 16 bytes, which the LDS-strip kernel cannot stage (DESIGN.md "Kernels").
//
// One LANE per block, 256 consecutive (pair, block) items per workgroup, no LDS: a sparse grid
// touches each pixel about once, so there is nothing for LDS to share.  A lane reads its 8x8
// reference tile (8 unaligned 8-byte loads) and its 16 search rows (16 unaligned 16-byte
// loads) straight from global memory -- small frames stay in L2, e.g. a 64x64 pair is 8 KB --
// and runs the same arithmetic as k_search_tile8: per (search row, reference row) four
// v_qsad_pk_u16_u8 and two v_sad_hi_u8, packed u16 accumulators, per-lane v_min3 arg-min over
// (sad << 16 | idx) = first minimum in scan order.  Half-pixel refinement, when enabled,
// follows in the same lane from the ring of the best match (aof_refine.hpp).
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_reduce.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;  // 64, 128 and 256 measure the same, 512 is 4 % slower

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

// One block: record (and direction) written to global memory and returned for the votes.
// Returns the half-pixel direction (8 = none).
template <bool SUBPIXEL>
__device__ __forceinline__ int search_block(const SearchArgs &a, int64_t pair, int blk, int64_t item,
                                            aof_block &rec)
{
    const int bx = blk % a.grid.nx, by = blk / a.grid.nx;
    const int i = a.grid.x0 + bx * a.grid.step_x, j = a.grid.y0 + by * a.grid.step_y;
    const int W = a.w;
    constexpr int m = SUBPIXEL ? 1 : 0;
    int px = 0, py = 0;
    if (a.pred) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(a.w * a.h));

    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    uint32_t *out = reinterpret_cast<uint32_t *>(a.blocks) + item;  // one dword store per record
    // the search window (plus the half-pixel ring) must lie inside the frame
    const int wx0 = i + px - 4, wy0 = j + py - 4;
    if (wx0 - m < 0 || wy0 - m < 0 || wx0 + 16 + m > a.w || wy0 + 16 + m > a.h) {
        *out = __builtin_bit_cast(uint32_t, rec);
        if (SUBPIXEL) a.subdirs[item] = 8;
        return 8;
    }
    const uint8_t *pr = a.prev + pair * a.pair_stride + (int64_t)j * W + i;
    const uint8_t *pc = a.cur + pair * a.pair_stride + (int64_t)wy0 * W + wx0;

    uint32_t ref[8][2];
    uint4 win[16];
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_memcpy(ref[r], pr + r * W, 8);
#pragma unroll
    for (int s = 0; s < 16; s++) __builtin_memcpy(&win[s], pc + s * W, 16);

    // 4x4 gradient gate on tile bytes [2..5] x rows [2..5]
    uint32_t diff = 0;
    {
        uint32_t mid[4];
#pragma unroll
        for (int r = 0; r < 4; r++) mid[r] = __builtin_amdgcn_alignbyte(ref[r + 2][1], ref[r + 2][0], 2);
#pragma unroll
        for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
        for (int r = 0; r < 4; r++)  // bytes (3,4,5,5) against (2,3,4,5): the doubled byte adds 0
            diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
    }
    if (diff < (uint32_t)a.feature_threshold) {
        *out = __builtin_bit_cast(uint32_t, rec);
        if (SUBPIXEL) a.subdirs[item] = 8;
        return 8;
    }

    // per dy: offsets 0..3 / 4..7 as packed u16, offset 8 as (sad << 16 | idx)
    u64 acc_lo[9], acc_hi[9];
    uint32_t acc_8[9];
#pragma unroll
    for (int d = 0; d < 9; d++) { acc_lo[d] = 0; acc_hi[d] = 0; acc_8[d] = (uint32_t)(d * 9 + 8); }
#pragma unroll
    for (int s = 0; s < 16; s++) {
        uint4 w = win[s];
        if (delta != 0) w = sat_add_u8x16(w, delta);
        const u64 p01 = pack64(w.x, w.y), p12 = pack64(w.y, w.z), p23 = pack64(w.z, w.w);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int d = s - r;  // dy index, dy = d - 4
            if (d < 0 || d >= 9) continue;
            acc_lo[d] = qsad(p01, ref[r][0], acc_lo[d]);
            acc_lo[d] = qsad(p12, ref[r][1], acc_lo[d]);
            acc_hi[d] = qsad(p12, ref[r][0], acc_hi[d]);
            acc_hi[d] = qsad(p23, ref[r][1], acc_hi[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], acc_8[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], acc_8[d]);
        }
    }
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const uint32_t base = (uint32_t)(d * 9);
        const uint32_t l0 = (uint32_t)acc_lo[d], l1 = (uint32_t)(acc_lo[d] >> 32);
        const uint32_t h0 = (uint32_t)acc_hi[d], h1 = (uint32_t)(acc_hi[d] >> 32);
        const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
        const uint32_t k2 = (l1 << 16) | (base + 2), k3 = (l1 & 0xFFFF0000u) | (base + 3);
        const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
        const uint32_t k6 = (h1 << 16) | (base + 6), k7 = (h1 & 0xFFFF0000u) | (base + 7);
        best = min(best, min(min(k0, k1), k2));
        best = min(best, min(min(k3, k4), k5));
        best = min(best, min(min(k6, k7), acc_8[d]));
    }
    const int idx = (int)(best & 0xFFFFu);
    rec.dx = (int8_t)(px + idx % 9 - 4);
    rec.dy = (int8_t)(py + idx / 9 - 4);
    rec.sad = (uint16_t)(best >> 16);
    *out = __builtin_bit_cast(uint32_t, rec);

    // Half-pixel refinement of accepted blocks: the ring of the best match, rows -1..8 and
    // bytes -1..8, again straight from global memory (the lines were touched a moment ago).
    int subdir = 8;
    if constexpr (SUBPIXEL) {
        if ((uint32_t)rec.sad < (uint32_t)a.value_threshold) {
            const uint8_t *ring = pc + (idx / 9 - 1) * W + (idx % 9 - 1);
            uint32_t rows[10][3];
#pragma unroll
            for (int y = 0; y < 10; y++) {
                uint16_t tail;
                __builtin_memcpy(rows[y], ring + y * W, 8);
                __builtin_memcpy(&tail, ring + y * W + 8, 2);
                rows[y][2] = tail;
            }
            RefineState<2> st;
            st.init();
            for_rows<-1, 8>([&](auto yc) {
                constexpr int Y = decltype(yc)::value;
                uint32_t d[3] = {rows[Y + 1][0], rows[Y + 1][1], rows[Y + 1][2]};
                if (delta != 0) {
#pragma unroll
                    for (int q = 0; q < 3; q++) d[q] = sat_add_u8x4(d[q], delta);
                }
                st.template row<Y>(d, ref);
            });
            subdir = st.direction(rec.sad);
        }
        a.subdirs[item] = (uint8_t)subdir;
    }
    return subdir;
}

// Flat mapping: 256 consecutive (pair, block) items per workgroup; K3 follows.
// Two to three waves per SIMD: the kernel trades occupancy for registers, so that a lane has
// all 24 of its row loads in flight at once (one memory round trip per block instead of 16).
template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads, 4) void k_search_lane8(SearchArgs a, int64_t items, uint32_t total_wgs)
{
    // consecutive workgroups = consecutive block rows of one pair: keep them on one XCD, whose L2
    // then serves the search rows that vertically adjacent blocks share
    const int64_t item = (int64_t)xcd_remap(blockIdx.x, total_wgs) * kThreads + threadIdx.x;
    if (item >= items) return;
    const int nb = a.grid.blocks();
    const int64_t pair = item / nb;
    aof_block rec;
    (void)search_block<SUBPIXEL>(a, pair, (int)(item - pair * nb), item, rec);
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
    const int64_t pair0 = (int64_t)blockIdx.x * ppw;
    const int np = (int)min((int64_t)ppw, a.n_pairs - pair0);
    for (int k = tid; k < ppw * 2 * n; k += kThreads) s_votes[k] = 0;
    __syncthreads();
    const bool live = tid < np * nb;
    const int p = live ? tid / nb : 0, blk = live ? tid - p * nb : 0;
    aof_block rec;
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    int subdir = 8;
    if (live) subdir = search_block<SUBPIXEL>(a, pair0 + p, blk, (pair0 + p) * nb + blk, rec);
    const bool ok = live && (uint32_t)rec.sad < (uint32_t)a.value_threshold;  // skipped = 0xFFFF
    const int hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
    const int hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
    wave_vote(s_votes, p * 2 * n + 2 * rec.dx + hx + centre, ok);
    wave_vote(s_votes, p * 2 * n + n + 2 * rec.dy + hy + centre, ok);
    __syncthreads();
    if (tid < np) {
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

bool lane8_supported(const SearchArgs &a)
{
    if (a.tile != 8 || a.search != 4) return false;
    if ((int64_t)a.w * a.h > 0x7FFFFFFF) return false;
    return a.n_pairs <= 0x7FFFFFFFll * kThreads / (a.grid.blocks() > 0 ? a.grid.blocks() : 1);
}

int launch_search_lane8(const SearchArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    const int64_t items = a.n_pairs * a.grid.blocks();
    const int64_t wgs = (items + kThreads - 1) / kThreads;
    if (wgs > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    if (a.subpixel && !a.subdirs) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(a.subpixel ? k_search_lane8<true> : k_search_lane8<false>, dim3((uint32_t)wgs),
                       dim3(kThreads), 0, static_cast<hipStream_t>(stream), a, items, (uint32_t)wgs);
    return (int)hipGetLastError();
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
    const int64_t wgs = (a.n_pairs + ppw - 1) / ppw;
    if (wgs > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    const int n = 2 * (2 * a.hist_range + 1) + 1;
    if (a.subpixel && !a.subdirs) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(a.subpixel ? k_flow_lane8<true> : k_flow_lane8<false>, dim3((uint32_t)wgs), dim3(kThreads),
                       (size_t)ppw * 2 * n * sizeof(uint32_t),
                       static_cast<hipStream_t>(stream), a, tail, ppw);
    return (int)hipGetLastError();
}

}  // namespace aof
