// Device-side helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace aof {

typedef unsigned long long u64;
typedef short short2_t __attribute__((ext_vector_type(2)));

// Spec "Mean": round-half-up integer mean; delta = mean(prev) - mean(cur).
// sums layout: [pair][frame: 0 prev, 1 cur][level].
__device__ __forceinline__ int equalise_delta(const uint32_t *sums, int64_t pair, int level,
                                              uint32_t npix)
{
    if (!sums) return 0;
    const uint32_t sp = sums[pair * 4 + 0 * 2 + level];
    const uint32_t sc = sums[pair * 4 + 1 * 2 + level];
    const int mp = (int)((sp + npix / 2) / npix);
    const int mc = (int)((sc + npix / 2) / npix);
    return mp - mc;
}

__device__ __forceinline__ int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// clamp(b + delta, 0, 255) on four packed bytes: widen to two packed-i16 pairs
// (v_perm_b32), v_pk_add/max/min_i16, narrow again (v_perm_b32).
__device__ __forceinline__ uint32_t sat_add_u8x4(uint32_t v, int delta)
{
    const uint32_t lo = __builtin_amdgcn_perm(0u, v, 0x0c010c00u);  // b1:b0 as i16 pair
    const uint32_t hi = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);  // b3:b2
    const short2_t d = {(short)delta, (short)delta};
    const short2_t zero = {0, 0}, top = {255, 255};
    short2_t l = __builtin_bit_cast(short2_t, lo) + d;
    short2_t h = __builtin_bit_cast(short2_t, hi) + d;
    l = __builtin_elementwise_min(__builtin_elementwise_max(l, zero), top);
    h = __builtin_elementwise_min(__builtin_elementwise_max(h, zero), top);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, h), __builtin_bit_cast(uint32_t, l),
                                 0x06040200u);
}

// Sum of the four bytes of w added to acc (v_sad_u8 against zero).
__device__ __forceinline__ uint32_t byte_sum(uint32_t w, uint32_t acc)
{
    return __builtin_amdgcn_sad_u8(w, 0u, acc);
}

// 64-lane integer reductions (wave = 64 on gfx950).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)v, o, 64);
        v = other < v ? other : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

// Adds one vote per active lane to hist[bin] (LDS): neighbouring blocks mostly vote for
// the SAME bin, which would serialise a per-lane LDS atomic 64 ways, so equal bins are
// first aggregated across the wave (ballot) and one lane adds the count.  Must be called
// by every lane of the wave.
__device__ __forceinline__ void wave_vote(uint32_t *hist, int bin, bool active)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(active && bin == b) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// Bijective XCD-aware remap of a 1-D grid (workgroups b and b+8 share an XCD's
// L2 under round-robin placement): XCD k gets one contiguous chunk of logical
// ids, so consecutive strips of one frame pair are staged through the same L2.
// Placement only affects speed, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t total)
{
    const uint32_t q = total / 8, r = total % 8, xcd = b % 8, slot = b / 8;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace aof
