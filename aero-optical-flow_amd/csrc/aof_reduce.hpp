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
__device__ __forceinline__ void finalise_flow_wave_bins(const FlowTail &a, int64_t pair, uint32_t hx, uint32_t hy,
                                                        int sum2x, int sum2y, int total, aof_flow *record_out = nullptr,
                                                        const aof_flow *pred_rec = nullptr);

__device__ __forceinline__ void finalise_flow_wave(const FlowTail &a, int64_t pair, const uint32_t *hist_x,
                                                   const uint32_t *hist_y, const int *sums, aof_flow *record_out = nullptr,
                                                   const aof_flow *pred_rec = nullptr)
{
    const int n = 2 * (2 * a.range + 1) + 1;
    const int k = (int)(threadIdx.x & 63);
    finalise_flow_wave_bins(a, pair, k < n ? hist_x[k] : 0u, k < n ? hist_y[k] : 0u, sums[0], sums[1], sums[2],
                            record_out, pred_rec);
}

// The same with the bins already in registers: lane k brings bin k of both histograms (0 for k >= n);
// sum2x / sum2y / total (wave-uniform): the sums of the 2*dx and 2*dy votes and their number.
__device__ __forceinline__ void finalise_flow_wave_bins(const FlowTail &a, int64_t pair, uint32_t hx, uint32_t hy,
                                                        int sum2x, int sum2y, int total, aof_flow *record_out,
                                                        const aof_flow *pred_rec)
{
    const int centre = 2 * a.range + 1, n = 2 * centre + 1;
    const int k = (int)(threadIdx.x & 63);
    const int sums[3] = {sum2x, sum2y, total};
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
            if (a.emit_predictor)   // (uniform; the integer division is as long as everything else here)
                pred = small ? floor_div<int>((int)(2 * v + w), (int)(2 * w)) - centre
                             : (int)(floor_div<long long>(2ll * v + w, 2ll * w) - centre);
        } else {
            flow = __fdiv_rn((float)s2 * 0.5f, (float)count);
            if (a.emit_predictor)
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

// ---- votes in global memory: the reduction inside the search kernel (no K3 launch) ----------------
// A pair's votes meet in a small record of the context's own vote memory (aof_ctx::d_votes, zero at
// rest): a 64-bit arrival word -- high half the number of blocks that have arrived, low half the number
// of votes they cast -- followed by the two histograms.  SEARCH waves add their votes (aggregated across
// the wave first: under a global motion two adds per wave) and then their arrival with agent-scope
// integer atomics WITHOUT waiting for any of them: nothing is returned, the wave ends at once.  The
// launch carries one extra FINALISER wave per pair behind its search workgroups; it reads arrival word
// and bins (agent-scope loads) until all blocks have arrived AND both histograms hold as many votes as
// the arrival word announces -- counts only grow, so equal sums mean every add has landed, in whatever
// order the memory system performed them -- then writes the pair's aof_flow and zeroes the record for
// the next launch.  Integer adds commute: the flow record does not depend on arrival order.
// No deadlock: a finaliser only waits for search waves, search waves wait for nothing, and a
// finaliser's workgroup id is higher than that of every search workgroup, so all search workgroups of
// its XCD have been dispatched before it occupies a slot.  Nothing here needs a cache write-back or
// invalidate: the record is only ever touched by agent-scope atomics, which are performed beyond the
// XCDs' L2s (an agent-scope RELEASE per workgroup, which records in ordinary memory would need, costs
// an L2 write-back each and made the kernel 4x slower: round 1; a last-arriver scheme, in which every
// search wave waits for its adds and an arrival counter, cost 14 % of the search: round 3).
__device__ __forceinline__ void vote_add_agent(uint32_t *p, uint32_t v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no return, no wait
}

// One axis: one add per distinct bin among the lanes in `active` (called by the whole wave).
__device__ __forceinline__ void wave_vote_agent(uint32_t *hist, int bin, bool active)
{
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(active && bin == b) & todo;
        if ((int)(threadIdx.x & 63) == leader) vote_add_agent(&hist[b], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// Search side, called by ALL 64 lanes of a wave.  `member`: the lane's block belongs to `pair`
// (wave-uniform pair); `ok`: it votes, for bins (bin_x, bin_y).  Written so that everything a wave
// under ONE motion needs stays in scalar registers (ballots, the leader's key through v_readlane, the
// counts through s_bcnt1) and one lane issues the three adds: the search kernel is VALU-issue-bound, and
// every vector instruction here is paid for at the full rate (a first version cost 120 VALU per wave,
// 6 % of the search).
__device__ __forceinline__ void vote_and_arrive(const VoteMem &vm, int range, uint32_t pair, bool member, bool ok,
                                                int bin_x, int bin_y)
{
    const unsigned long long members = __ballot(member);
    if (members == 0) return;   // (wave-uniform)
    const int n = 2 * (2 * range + 1) + 1, lane = (int)(threadIdx.x & 63);
    uint32_t *rec = vm.base + (size_t)pair * vm.stride, *hist_x = rec + 2, *hist_y = rec + 2 + n;
    ok = ok && member;
    const unsigned long long voters = __ballot(ok);
    const int head = __ffsll((long long)members) - 1;
    const unsigned long long arrival = ((unsigned long long)__popcll(members) << 32) | (unsigned long long)__popcll(voters);
    bool one_motion = true;
    int k = 0;
    if (voters) {
        const int key = bin_x | (bin_y << 8);
        k = __builtin_amdgcn_readlane(key, __ffsll((long long)voters) - 1);   // scalar
        one_motion = (__ballot(ok && key == k) & voters) == voters;
    }
    if (one_motion) {   // (wave-uniform) two adds for the votes, one for the arrival, all from one lane
        if (lane == head) {
            if (voters) {
                const uint32_t c = (uint32_t)__popcll(voters);
                vote_add_agent(&hist_x[k & 0xFF], c);
                vote_add_agent(&hist_y[k >> 8], c);
            }
            (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(rec), arrival, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    wave_vote_agent(hist_x, bin_x, ok);
    wave_vote_agent(hist_y, bin_y, ok);
    if (lane == head)
        (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(rec), arrival, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
}

// The same for a wave that WALKS several blocks per lane inside one pair (the column walk, aof_cols8_kernels.hpp): under a
// global motion every step of the walk votes for the same pair of bins, so the wave keeps votes and arrivals of
// consecutive steps that agree in scalar registers and adds them ONCE -- three atomics per walk instead of three per
// step (eight steps per walk at 1 024 VGA pairs; with an add per step the kernel that reduces in its launch was 3 %
// slower than search + K3, round 5).  A step whose motion differs flushes what is pending first; a step whose lanes
// disagree among themselves votes at once, lane by lane, as vote_and_arrive does.  flush() behind the walk.
struct NoWalkVotes {   // (the same walk without votes: nothing, so that it compiles to what it did)
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void flush(const VoteMem &, int, uint32_t) {}
    __device__ __forceinline__ void step(const VoteMem &, int, uint32_t, bool, bool, int, int) {}
};
struct WalkVotes {
    int key;              // bins (x | y << 8) of the pending votes; only meaningful while votes != 0
    uint32_t votes, arrivals;

    __device__ __forceinline__ void init() { key = 0; votes = 0; arrivals = 0; }

    __device__ __forceinline__ void flush(const VoteMem &vm, int range, uint32_t pair)
    {
        if (arrivals == 0) return;   // (wave-uniform; votes without arrivals do not exist)
        const int n = 2 * (2 * range + 1) + 1;
        uint32_t *rec = vm.base + (size_t)pair * vm.stride, *hist_x = rec + 2, *hist_y = rec + 2 + n;
        if ((threadIdx.x & 63) == 0) {
            if (votes) {
                vote_add_agent(&hist_x[key & 0xFF], votes);
                vote_add_agent(&hist_y[key >> 8], votes);
            }
            (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(rec), ((unsigned long long)arrivals << 32) | votes,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        votes = 0; arrivals = 0;
    }

    // one step of the walk, called by ALL 64 lanes; member / ok / bins as in vote_and_arrive
    __device__ __forceinline__ void step(const VoteMem &vm, int range, uint32_t pair, bool member, bool ok, int bin_x, int bin_y)
    {
        const unsigned long long members = __ballot(member);
        if (members == 0) return;   // (wave-uniform)
        ok = ok && member;
        const unsigned long long voters = __ballot(ok);
        const uint32_t nm = (uint32_t)__popcll(members), nv = (uint32_t)__popcll(voters);
        if (voters == 0) { arrivals = (uint32_t)__builtin_amdgcn_readfirstlane((int)(arrivals + nm)); return; }
        const int mine = bin_x | (bin_y << 8);
        const int k = __builtin_amdgcn_readlane(mine, __ffsll((long long)voters) - 1);   // scalar
        if ((__ballot(ok && mine == k) & voters) == voters) {   // (wave-uniform) one motion in this step
            if (votes != 0 && key != k) flush(vm, range, pair);
            // (scalar registers: the walk's lanes sit at the 128 VGPRs of four waves per SIMD)
            key = k;
            votes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(votes + nv));
            arrivals = (uint32_t)__builtin_amdgcn_readfirstlane((int)(arrivals + nm));
            return;
        }
        flush(vm, range, pair);
        const int n = 2 * (2 * range + 1) + 1;
        uint32_t *rec = vm.base + (size_t)pair * vm.stride;
        wave_vote_agent(rec + 2, bin_x, ok);
        wave_vote_agent(rec + 2 + n, bin_y, ok);
        if ((threadIdx.x & 63) == 0)
            (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(rec), ((unsigned long long)nm << 32) | nv, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    }
};

// Finaliser side: ONE WAVE per pair (all 64 lanes), n <= 62.  Gives up after vm.deadline_ticks of the
// 100 MHz real-time counter (a search wave that never arrives means the launch is broken anyway): the
// pair's flow record then says "nothing measured" -- flow 0, count 0, quality 0, NO valid flag -- and the
// wave raises the context's fault word in pinned host memory (pair + 1), which makes every later call on the
// context fail (sticky_error in aof_capi.hip): voters that come late add into a record this wave has
// already zeroed, so the context's vote memory cannot be trusted again.
__device__ __forceinline__ void await_votes_and_finalise(const VoteMem &vm, const FlowTail &tail, uint32_t pair)
{
    const uint64_t deadline_ticks = vm.deadline_ticks;
    const int centre = 2 * tail.range + 1, n = 2 * centre + 1, lane = (int)(threadIdx.x & 63);
    uint32_t *rec = vm.base + (size_t)pair * vm.stride, *hist_x = rec + 2, *hist_y = rec + 2 + n;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t hx = 0, hy = 0;
    int total = 0;
    bool complete = false;
    for (;;) {
        uint32_t arrived = 0, announced = 0;
        if (lane < n) {
            hx = __hip_atomic_load(&hist_x[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hy = __hip_atomic_load(&hist_y[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 63) {
            const unsigned long long a = __hip_atomic_load(reinterpret_cast<unsigned long long *>(rec), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            arrived = (uint32_t)(a >> 32);
            announced = (uint32_t)a;
        }
        arrived = (uint32_t)__shfl((int)arrived, 63, 64);
        announced = (uint32_t)__shfl((int)announced, 63, 64);
        total = (int)wave_sum_u32(hx);
        complete = arrived == (uint32_t)tail.nblocks && (uint32_t)total == announced && wave_sum_u32(hy) == announced;
        if (complete || __builtin_amdgcn_s_memrealtime() - t0 > deadline_ticks) break;   // (wave-uniform)
        __builtin_amdgcn_s_sleep(8);
    }
    if (lane < n) {   // zero at rest
        __hip_atomic_store(&hist_x[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&hist_y[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane == 63) {
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(rec), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!complete) {
        if (lane == 0) {
            aof_flow none;
            __builtin_memset(&none, 0, sizeof(none));
            tail.flows[pair] = none;
            __hip_atomic_store(vm.fault, pair + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int sum2x = (int)wave_sum_u32((uint32_t)((lane - centre) * (int)hx));
    const int sum2y = (int)wave_sum_u32((uint32_t)((lane - centre) * (int)hy));
    finalise_flow_wave_bins(tail, (int64_t)pair, hx, hy, sum2x, sum2y, total);
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
