// Half-pixel refinement arithmetic of K2b (k_refine)
// (DESIGN.md "Spec": Half-pixel refinement).
//
// The eight half-pixel images are built four pixels at a time with v_lerp_u8 -- floor((a+b)/2)
// per byte, the UHADD8 of the published algorithm -- from three row-shifted copies of each
// window row:
//   H+(y) = lerp(P(y,x), P(y,x+1))   H-(y) = lerp(P(y,x), P(y,x-1))   C(y) = P(y,x)
//   dir 0 = H+(y)                dir 4 = H-(y)
//   dir 2 = lerp(C(y),C(y+1))    dir 6 = lerp(C(y),C(y-1))
//   dir 1 = lerp(H+(y),H+(y+1))  dir 7 = lerp(H+(y-1),H+(y))
//   dir 3 = lerp(H-(y),H-(y+1))  dir 5 = lerp(H-(y-1),H-(y))
// so one pass over window rows y = -1..B with the previous row kept in registers feeds all
// eight SADs.
#pragma once

#include <type_traits>

#include "aof_device.hpp"

namespace aof {

// Running state of one block's refinement: feed rows y = -1, 0, ..., B in order (a slice of ROWS
// tile rows: its rows -1 .. ROWS; the direction sums of the slices of a block add up).
template <int NW, int ROWS = 4 * NW>  // dwords per tile row: 2 (8x8) or 4 (16x16); tile rows fed (a slice when lanes share a block)
struct RefineState {
    uint32_t acc[8];
    uint32_t pc[NW], ph[NW], pl[NW];  // previous window row: C, H+, H-

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = 0;
    }

    // d: bytes -1 .. 4 NW + 1 of window row Y relative to the best match (d[NW]: only its two low
    // bytes are used); ref: the reference tile rows of the slice.  Y is a compile-time constant
    // at every call.
    template <int Y>
    __device__ __forceinline__ void row(const uint32_t (&d)[NW + 1], const uint32_t (&ref)[ROWS][NW])
    {
        constexpr int B = ROWS;
        uint32_t c[NW], hp[NW], hm[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            c[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
            const uint32_t right = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
            hp[q] = __builtin_amdgcn_lerp(c[q], right, 0u);
            hm[q] = __builtin_amdgcn_lerp(c[q], d[q], 0u);
        }
#pragma unroll
        for (int q = 0; q < NW; q++) {
            if constexpr (Y >= 0 && Y < B) {
                acc[0] = __builtin_amdgcn_sad_u8(hp[q], ref[Y][q], acc[0]);
                acc[4] = __builtin_amdgcn_sad_u8(hm[q], ref[Y][q], acc[4]);
            }
            if constexpr (Y >= 0) {
                const uint32_t v = __builtin_amdgcn_lerp(pc[q], c[q], 0u);
                const uint32_t dr = __builtin_amdgcn_lerp(ph[q], hp[q], 0u);
                const uint32_t dl = __builtin_amdgcn_lerp(pl[q], hm[q], 0u);
                if constexpr (Y >= 1) {  // tile row Y-1 looks down
                    acc[2] = __builtin_amdgcn_sad_u8(v, ref[Y - 1][q], acc[2]);
                    acc[1] = __builtin_amdgcn_sad_u8(dr, ref[Y - 1][q], acc[1]);
                    acc[3] = __builtin_amdgcn_sad_u8(dl, ref[Y - 1][q], acc[3]);
                }
                if constexpr (Y < B) {   // tile row Y looks up
                    acc[6] = __builtin_amdgcn_sad_u8(v, ref[Y][q], acc[6]);
                    acc[7] = __builtin_amdgcn_sad_u8(dr, ref[Y][q], acc[7]);
                    acc[5] = __builtin_amdgcn_sad_u8(dl, ref[Y][q], acc[5]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NW; q++) { pc[q] = c[q]; ph[q] = hp[q]; pl[q] = hm[q]; }
    }

    // The direction sums only grow, and a direction wins only with a sum BELOW the integer match's SAD: once no lane of the
    // wave that is still refining has such a sum, every one of them ends with "none" and the rows that are left change nothing.
    // (Exact; what it saves depends on the input: integer shifts stop after two of ten ring rows, a true half-pixel motion never.)
    __device__ __forceinline__ bool nobody_can_win(uint32_t integer_sad) const
    {
        const uint32_t m = min(min(min(acc[0], acc[1]), min(acc[2], acc[3])), min(min(acc[4], acc[5]), min(acc[6], acc[7])));
        return __ballot(m < integer_sad) == 0;   // (the lanes in this control flow)
    }

    // first direction whose SAD beats the integer match, 8 = none
    __device__ __forceinline__ int direction(uint32_t integer_sad) const
    {
        uint32_t mind = integer_sad;
        int subdir = 8;
#pragma unroll
        for (int dir = 0; dir < 8; dir++)
            if (acc[dir] < mind) { mind = acc[dir]; subdir = dir; }
        return subdir;
    }
};

// Compile-time loop over the window rows Y = FIRST .. LAST: f(std::integral_constant<int, Y>).
template <int Y, int LAST, typename F>
__device__ __forceinline__ void for_rows(F &&f)
{
    f(std::integral_constant<int, Y>{});
    if constexpr (Y < LAST) for_rows<Y + 1, LAST>(f);
}

}  // namespace aof
