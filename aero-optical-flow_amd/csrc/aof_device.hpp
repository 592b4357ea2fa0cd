// Device-side helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "aof_internal.hpp"

namespace aof {

typedef unsigned long long u64;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Fourth dword of a raw (untyped, stride 0) buffer resource on gfx950: DATA_FORMAT = 32.
constexpr int kRawBuffer = 0x00020000;

// n / d for n < 2^31 with the host-made magic number of aof_internal.hpp.
__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shift; }

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

// clamp(b + delta, 0, 255) on four packed bytes in four instructions.  A byte sitting in the HIGH
// half of a u16 lane saturates exactly like a u8 under v_pk_add_u16 / v_pk_sub_u16 with clamp
// (the addend delta<<8 leaves the low half alone): the odd bytes already sit there, the even
// bytes after a shift by 8, and one v_perm_b32 gathers the four high halves.
// d8 = |delta| in the high byte of both u16 lanes (sat_delta_u16x2).
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_delta_u16x2(int delta)
{
    const uint32_t m = (uint32_t)(delta < 0 ? -delta : delta);
    return (m << 8) | (m << 24);
}

template <bool NEG>
__device__ __forceinline__ uint32_t sat_shift_u8x4(uint32_t v, uint32_t d8)
{
    const ushort2_t odd = __builtin_bit_cast(ushort2_t, v), even = __builtin_bit_cast(ushort2_t, v << 8);
    const ushort2_t d = __builtin_bit_cast(ushort2_t, d8);
    const ushort2_t o2 = NEG ? __builtin_elementwise_sub_sat(odd, d) : __builtin_elementwise_add_sat(odd, d);
    const ushort2_t e2 = NEG ? __builtin_elementwise_sub_sat(even, d) : __builtin_elementwise_add_sat(even, d);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, o2), __builtin_bit_cast(uint32_t, e2),
                                 0x07030501u);
}

// delta is wave-uniform at every call site: the branch is scalar.
__device__ __forceinline__ uint32_t sat_add_u8x4(uint32_t v, int delta)
{
    const uint32_t d8 = sat_delta_u16x2(delta);
    return delta < 0 ? sat_shift_u8x4<true>(v, d8) : sat_shift_u8x4<false>(v, d8);
}

__device__ __forceinline__ uint4 sat_add_u8x16(uint4 v, int delta)
{
    const uint32_t d8 = sat_delta_u16x2(delta);
    if (delta < 0) {
        v.x = sat_shift_u8x4<true>(v.x, d8); v.y = sat_shift_u8x4<true>(v.y, d8);
        v.z = sat_shift_u8x4<true>(v.z, d8); v.w = sat_shift_u8x4<true>(v.w, d8);
    } else {
        v.x = sat_shift_u8x4<false>(v.x, d8); v.y = sat_shift_u8x4<false>(v.y, d8);
        v.z = sat_shift_u8x4<false>(v.z, d8); v.w = sat_shift_u8x4<false>(v.w, d8);
    }
    return v;
}

// 2x2 box filter (a+b+c+d+2)>>2: two level-1 pixels from one dword of each of two rows, returned in
// bytes 0 and 2.
__device__ __forceinline__ uint32_t box2(uint32_t a, uint32_t b)
{
    const uint32_t m = 0x00FF00FFu;
    const uint32_t s = (a & m) + ((a >> 8) & m) + (b & m) + ((b >> 8) & m) + 0x00020002u;
    return (s >> 2) & m;
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

// Two histograms at once (the x and y votes of one block).  Under a global motion every lane of the
// wave votes for the same (bin_x, bin_y) pair: one ballot, one leader, two adds.  Otherwise the two
// axes are voted separately -- at most as many trips as there are distinct bins per axis (joint
// (x, y) keys would need up to their product on unrelated frames).
__device__ __forceinline__ void wave_vote2(uint32_t *hist_x, uint32_t *hist_y, int bin_x, int bin_y, bool active)
{
    const unsigned long long todo = __ballot(active);
    if (todo == 0) return;
    const int key = bin_x | (bin_y << 16);
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(key, leader, 64);
    if ((__ballot(active && key == k) & todo) == todo) {   // (wave-uniform)
        if ((int)(threadIdx.x & 63) == leader) {
            const uint32_t c = (uint32_t)__popcll(todo);
            atomicAdd(&hist_x[k & 0xFFFF], c);
            atomicAdd(&hist_y[k >> 16], c);
        }
        return;
    }
    wave_vote(hist_x, bin_x, active);
    wave_vote(hist_y, bin_y, active);
}

// The same with a small weight per lane (0 = no vote, at most 4: a lane that speaks for up to four consecutive records
// which agree).  The weights of the lanes that share a bin are summed through the ballots of "weight == w" (scalar
// popcounts), so a weighted vote costs what a plain one does.  Integer adds: the histograms do not depend on the grouping.
__device__ __forceinline__ void wave_vote_weighted(uint32_t *hist, int bin, int weight, unsigned long long m1, unsigned long long m2,
                                                   unsigned long long m3, unsigned long long m4)
{
    unsigned long long todo = m1 | m2 | m3 | m4;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(weight != 0 && bin == b) & todo;
        const uint32_t total = (uint32_t)(__popcll(same & m1) + 2 * __popcll(same & m2) + 3 * __popcll(same & m3) + 4 * __popcll(same & m4));
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b], total);
        todo &= ~same;
    }
}

// key = bin_x | bin_y << 16.  Must be called by every lane of the wave.
__device__ __forceinline__ void wave_vote2_weighted(uint32_t *hist_x, uint32_t *hist_y, int key, int weight)
{
    const unsigned long long m1 = __ballot(weight == 1), m2 = __ballot(weight == 2), m3 = __ballot(weight == 3), m4 = __ballot(weight == 4);
    const unsigned long long todo = m1 | m2 | m3 | m4;
    if (todo == 0) return;
    const int leader = __ffsll((long long)todo) - 1;
    const int k = __shfl(key, leader, 64);
    if ((__ballot(weight != 0 && key == k) & todo) == todo) {   // (wave-uniform) one global motion: one leader, two adds
        if ((int)(threadIdx.x & 63) == leader) {
            const uint32_t total = (uint32_t)(__popcll(m1) + 2 * __popcll(m2) + 3 * __popcll(m3) + 4 * __popcll(m4));
            atomicAdd(&hist_x[k & 0xFFFF], total);
            atomicAdd(&hist_y[k >> 16], total);
        }
        return;
    }
    wave_vote_weighted(hist_x, key & 0xFFFF, weight, m1, m2, m3, m4);
    wave_vote_weighted(hist_y, key >> 16, weight, m1, m2, m3, m4);
}

// Bijective XCD-aware remap of a 1-D grid (workgroups b and b+8 share an XCD's
// L2 under round-robin placement): XCD k gets one contiguous chunk of logical
// ids, so consecutive block rows of one frame pair are served by the same L2.
// Placement only affects speed, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t total)
{
    const uint32_t q = total / 8, r = total % 8, xcd = b % 8, slot = b / 8;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace aof
