// Flow finalisation kept apart from K3's vote collection (k_reduce): from the two
// half-pixel vote histograms of one pair to its 16-byte aof_flow (DESIGN.md "Spec": Reduce).
// Executed by ONE thread.  The float arithmetic is a handful of exactly-representable integers
// and correctly-rounded divisions (__fdiv_rn), bit-identical to the host arithmetic of the oracle.
#pragma once

#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

__device__ __forceinline__ void peak_window(int pos, int n, int *lo, int *hi)
{
    *lo = *hi = pos;
    if (pos > 1 && pos < n - 2) { *lo = pos - 2; *hi = pos + 2; }
    else if (pos == 0) { *hi = pos + 2; }
    else if (pos == n - 1) { *lo = pos - 2; }
    else if (pos == 1) { *lo = pos - 1; *hi = pos + 2; }
    else if (pos == n - 2) { *lo = pos - 2; *hi = pos + 1; }
}

template <typename T>
__device__ __forceinline__ T floor_div(T a, T b)
{
    T q = a / b;
    if ((a % b) < 0) q--;
    return q;
}

// Writes one pair's aof_flow from the peak windows' sums (vx, wx, vy, wy: sum of k*h[k] and of h[k]
// over the +-2-bin window of each axis) or, without the histogram filter, from the vote sums.
__device__ __forceinline__ void write_flow(const FlowTail &a, int64_t pair, int n, uint32_t vx, uint32_t wx, uint32_t vy,
                                           uint32_t wy, const int *sums)
{
#pragma clang fp contract(off)
    const int centre = 2 * a.range + 1;
    aof_flow out;
    out.flow_x = out.flow_y = 0.0f;
    out.count = (uint32_t)sums[2];
    out.quality = 0;
    out.flags = 0;
    out.pred_x = out.pred_y = 0;
    int px = 0, py = 0;
    const long long count = sums[2];
    // count <= nblocks, |sums| <= n * count, vx <= n * count: with n * nblocks < 2^28 nothing below
    // leaves 30 bits
    const bool small = (long long)n * a.nblocks < (1ll << 28);
    if (count > (long long)a.min_valid && count > 0) {
        if (a.hist_filter) {
            out.flow_x = (__fdiv_rn((float)vx, (float)wx) - (float)centre) / 2.0f;
            out.flow_y = (__fdiv_rn((float)vy, (float)wy) - (float)centre) / 2.0f;
            if (small) {   // every operand below 2^30: 32-bit divisions (a 64-bit one is ~150 instructions)
                px = floor_div<int>((int)(2 * vx + wx), (int)(2 * wx)) - centre;
                py = floor_div<int>((int)(2 * vy + wy), (int)(2 * wy)) - centre;
            } else {
                px = (int)(floor_div<long long>(2ll * vx + wx, 2ll * wx) - centre);
                py = (int)(floor_div<long long>(2ll * vy + wy, 2ll * wy) - centre);
            }
        } else {
            out.flow_x = __fdiv_rn((float)sums[0] * 0.5f, (float)count);
            out.flow_y = __fdiv_rn((float)sums[1] * 0.5f, (float)count);
            if (small) {
                px = floor_div<int>(2 * sums[0] + (int)count, 2 * (int)count);
                py = floor_div<int>(2 * sums[1] + (int)count, 2 * (int)count);
            } else {
                px = (int)floor_div<long long>(2ll * sums[0] + count, 2ll * count);
                py = (int)floor_div<long long>(2ll * sums[1] + count, 2ll * count);
            }
        }
        out.quality = small ? (uint8_t)((uint32_t)count * 255u / (uint32_t)a.nblocks)
                            : (uint8_t)((unsigned long long)count * 255ull / (unsigned long long)a.nblocks);
        out.flags |= AOF_FLAG_FLOW_VALID;
    }
    if (a.emit_predictor) {
        out.pred_x = (int8_t)px;
        out.pred_y = (int8_t)py;
    } else if (a.pred) {
        const aof_flow p = a.pred[pair];
        out.pred_x = p.pred_x;
        out.pred_y = p.pred_y;
        if (p.flags & AOF_FLAG_FLOW_VALID) out.flags |= AOF_FLAG_PRED_VALID;
    }
    a.flows[pair] = out;
}

// The same result as finalise_flow below, computed by one whole WAVE (all 64 lanes must call it):
// lane k holds bin k of both histograms, the first maximum is a wave maximum + ballot, the window
// sums are wave sums, lane 0 does the divisions.  One lane alone walks the bins with dependent LDS
// reads (2 us for 19 bins) while its workgroup -- in k_coarse the whole CU -- waits.  n <= 64.
// record_out (optional, LDS): lane 0 also leaves the record there; pred_rec (optional): the level-1
// record to copy the predictor fields from, instead of a.pred[pair] in global memory.
__device__ __forceinline__ void finalise_flow_wave(const FlowTail &a, int64_t pair, const uint32_t *hist_x,
                                                   const uint32_t *hist_y, const int *sums, aof_flow *record_out = nullptr,
                                                   const aof_flow *pred_rec = nullptr)
{
    const int centre = 2 * a.range + 1, n = 2 * centre + 1;
    const int k = (int)(threadIdx.x & 63);
    const uint32_t hx = k < n ? hist_x[k] : 0u, hy = k < n ? hist_y[k] : 0u;
    uint32_t mx = hx, my = hy;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t ox = (uint32_t)__shfl_xor((int)mx, o, 64), oy = (uint32_t)__shfl_xor((int)my, o, 64);
        mx = ox > mx ? ox : mx;
        my = oy > my ? oy : my;
    }
    const int posx = __ffsll((long long)__ballot(k < n && hx == mx)) - 1;   // first maximum (bin 0 when all are empty)
    const int posy = __ffsll((long long)__ballot(k < n && hy == my)) - 1;
    int lox, hix, loy, hiy;
    peak_window(posx, n, &lox, &hix);
    peak_window(posy, n, &loy, &hiy);
    const bool inx = k >= lox && k <= hix, iny = k >= loy && k <= hiy;
    const uint32_t vx = wave_sum_u32(inx ? (uint32_t)k * hx : 0u), wx = wave_sum_u32(inx ? hx : 0u);
    const uint32_t vy = wave_sum_u32(iny ? (uint32_t)k * hy : 0u), wy = wave_sum_u32(iny ? hy : 0u);
    // lane 0 divides for x, lane 1 for y (each axis is a chain of dependent division sequences)
    const bool axis_y = k == 1;
    const uint32_t v = axis_y ? vy : vx, w = axis_y ? wy : wx;
    const int s2 = axis_y ? sums[1] : sums[0];
    const long long count = sums[2];
    const bool valid = count > (long long)a.min_valid && count > 0;
    const bool small = (long long)n * a.nblocks < (1ll << 28);
    float flow = 0.0f;
    int pred = 0;
    if (k < 2 && valid) {
#pragma clang fp contract(off)
        if (a.hist_filter) {
            flow = (__fdiv_rn((float)v, (float)w) - (float)centre) / 2.0f;
            pred = small ? floor_div<int>((int)(2 * v + w), (int)(2 * w)) - centre
                         : (int)(floor_div<long long>(2ll * v + w, 2ll * w) - centre);
        } else {
            flow = __fdiv_rn((float)s2 * 0.5f, (float)count);
            pred = small ? floor_div<int>(2 * s2 + (int)count, 2 * (int)count)
                         : (int)floor_div<long long>(2ll * s2 + count, 2ll * count);
        }
    }
    const float flow_y = __shfl(flow, 1, 64);
    const int pred_y = __shfl(pred, 1, 64);
    if (k == 0) {
        aof_flow out;
        out.flow_x = flow; out.flow_y = flow_y;
        out.count = (uint32_t)sums[2];
        out.quality = 0;
        out.flags = 0;
        out.pred_x = out.pred_y = 0;
        if (valid) {
            out.quality = small ? (uint8_t)((uint32_t)count * 255u / (uint32_t)a.nblocks)
                                : (uint8_t)((unsigned long long)count * 255ull / (unsigned long long)a.nblocks);
            out.flags |= AOF_FLAG_FLOW_VALID;
        }
        if (a.emit_predictor) {
            out.pred_x = (int8_t)pred;
            out.pred_y = (int8_t)pred_y;
        } else if (a.pred) {
            const aof_flow p = pred_rec ? *pred_rec : a.pred[pair];
            out.pred_x = p.pred_x;
            out.pred_y = p.pred_y;
            if (p.flags & AOF_FLAG_FLOW_VALID) out.flags |= AOF_FLAG_PRED_VALID;
        }
        a.flows[pair] = out;
        if (record_out) *record_out = out;
    }
}

// hist_x/hist_y: n = 2*(2R+1)+1 bins each; sums = {sum of 2*dx votes, sum of 2*dy votes, count}.
// Executed by ONE thread.
__device__ __forceinline__ void finalise_flow(const FlowTail &a, int64_t pair, const uint32_t *hist_x,
                                              const uint32_t *hist_y, const int *sums)
{
    const int centre = 2 * a.range + 1, n = 2 * centre + 1;
    uint32_t vx = 0, wx = 0, vy = 0, wy = 0;
    if (a.hist_filter && sums[2] > a.min_valid && sums[2] > 0) {
        int posx = 0, posy = 0;
        uint32_t maxx = 0, maxy = 0;
        for (int k = 0; k < n; k++) {
            if (hist_x[k] > maxx) { maxx = hist_x[k]; posx = k; }
            if (hist_y[k] > maxy) { maxy = hist_y[k]; posy = k; }
        }
        int lo, hi;
        peak_window(posx, n, &lo, &hi);
        for (int k = lo; k <= hi; k++) { vx += (uint32_t)k * hist_x[k]; wx += hist_x[k]; }
        peak_window(posy, n, &lo, &hi);
        for (int k = lo; k <= hi; k++) { vy += (uint32_t)k * hist_y[k]; wy += hist_y[k]; }
    }
    write_flow(a, pair, n, vx, wx, vy, wy, sums);
}

}  // namespace aof
