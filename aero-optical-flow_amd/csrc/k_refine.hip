// K2b -- half-pixel refinement as a separate pass (DESIGN.md "Spec": Half-pixel refinement).
//
// Behind the searches that do not refine themselves: the generic wave-per-block kernel, and the
// 16x16 kernel where its LDS tile with the two extra rows does not fit (k_search_tile16 refines
// out of LDS otherwise, the 8x8 kernels in the search lane).  One lane per block, consecutive
// lanes = consecutive (pair, block) items: the lane reads its reference tile and the (B+2)^2
// neighbourhood of its best match straight from global memory, rows as unaligned 8/16-byte
// loads, and feeds them to the shared v_lerp_u8 arithmetic of aof_refine.hpp.  Blocks the
// search skipped or rejected get direction 8 (none).
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

constexpr int kRefineThreads = 256;

template <int NW>  // dwords per tile row: 2 (8x8) or 4 (16x16)
__global__ __launch_bounds__(kRefineThreads) void k_refine(SearchArgs a, int64_t items)
{
    constexpr int B = 4 * NW;
    // consecutive lanes = consecutive (pair, block) items: full waves on sparse grids too
    const int64_t item = (int64_t)blockIdx.x * kRefineThreads + threadIdx.x;
    if (item >= items) return;
    const int nb = a.grid.blocks();
    const int64_t pair = item / nb;
    const int blk = (int)(item - pair * nb);
    // one dword load (the C ABI requires a 4-byte aligned record array)
    const aof_block rec = __builtin_bit_cast(
        aof_block, reinterpret_cast<const uint32_t *>(a.blocks)[pair * nb + blk]);
    uint8_t *out = a.subdirs + pair * nb + blk;
    if (rec.sad == AOF_SAD_SKIPPED || (uint32_t)rec.sad >= (uint32_t)a.value_threshold) {
        *out = 8;
        return;
    }
    const int bx = blk % a.grid.nx, by = blk / a.grid.nx;
    const int i = a.grid.x0 + bx * a.grid.step_x, j = a.grid.y0 + by * a.grid.step_y;
    const int W = a.w;
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(a.w * a.h));
    const uint8_t *pr = a.prev + pair * a.pair_stride + (int64_t)j * W + i;
    // window row y = -1 starts one pixel left of the best match (the search kept the whole
    // ring inside the frame, or the block would have been skipped)
    const uint8_t *pc = a.cur + pair * a.pair_stride + (int64_t)(j + rec.dy - 1) * W + (i + rec.dx - 1);

    uint32_t ref[B][NW];
#pragma unroll
    for (int r = 0; r < B; r++) __builtin_memcpy(ref[r], pr + r * W, 4 * NW);

    // 8x8: all ten window rows are requested before the first is used (one memory round trip
    // per block); 16x16 has no registers to spare for that and loads row by row
    constexpr bool kPreload = NW == 2;
    uint32_t rows[kPreload ? B + 2 : 1][NW + 1];
    auto load_row = [&](int y, uint32_t (&d)[NW + 1]) {
        // bytes -1 .. B+1 of the row: NW dwords + one 16-bit tail (never past the ring)
        uint16_t tail;
        __builtin_memcpy(d, pc + (y + 1) * W, 4 * NW);
        __builtin_memcpy(&tail, pc + (y + 1) * W + 4 * NW, 2);
        d[NW] = tail;
    };
    if constexpr (kPreload) {
#pragma unroll
        for (int y = -1; y <= B; y++) load_row(y, rows[y + 1]);
    }

    RefineState<NW> st;
    st.init();
    for_rows<-1, B>([&](auto yc) {
        constexpr int Y = decltype(yc)::value;
        uint32_t d[NW + 1];
        if constexpr (kPreload) {
#pragma unroll
            for (int q = 0; q <= NW; q++) d[q] = rows[Y + 1][q];
        } else {
            load_row(Y, d);
        }
        if (delta != 0) {
#pragma unroll
            for (int q = 0; q <= NW; q++) d[q] = sat_add_u8x4(d[q], delta);
        }
        st.template row<Y>(d, ref);
    });
    const int subdir = st.direction(rec.sad);
    *out = (uint8_t)subdir;
}

}  // namespace

int launch_refine(const SearchArgs &a, void *stream)
{
    if (a.n_pairs == 0 || !a.subdirs) return 0;
    const int64_t items = a.n_pairs * a.grid.blocks();
    const int64_t wgs = (items + kRefineThreads - 1) / kRefineThreads;
    if (wgs > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    const dim3 grid((uint32_t)wgs);
    if (a.tile == 8)
        hipLaunchKernelGGL(k_refine<2>, grid, dim3(kRefineThreads), 0, static_cast<hipStream_t>(stream), a, items);
    else if (a.tile == 16)
        hipLaunchKernelGGL(k_refine<4>, grid, dim3(kRefineThreads), 0, static_cast<hipStream_t>(stream), a, items);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

}  // namespace aof
