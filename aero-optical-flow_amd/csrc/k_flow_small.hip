// KS -- a whole small pair, one or two levels, in ONE workgroup, ONE launch and out of LDS
// (DESIGN.md "Kernels": KS).
//
// The call shape of the reference -- one small frame per calcFlow(), mainloop.cpp:322 -- is bound by
// launches and memory round trips, not by arithmetic.  As separate kernels two levels are a memset,
// K1, the level-1 search and the level-0 search (four graph nodes, 45 us per call), and each search
// is a few dozen lanes walking all 81 candidates of their block alone, every row from memory.  Here
// one workgroup does the lot for a pair whose frames fit LDS and whose grids have at most 256 blocks
// (the published sparse grid: 25):
//   A  both level-0 frames: global (or pinned host) memory -> LDS, every load issued before the
//      first use -- one round trip --, pixel sums on the way;
//   B  two levels: 2x2 box pyramid LDS -> LDS, level-1 sums;
//   C  per level: one lane per (block, dy) -- nine lanes share a block, 25 blocks fill the workgroup --
//      reads tile and window rows from LDS at any byte offset (aligned dwords + v_alignbyte), sums its
//      nine dx candidates with v_qsad_pk_u16_u8 / v_sad_hi_u8 and joins its block through one LDS
//      atomicMin on the packed key (sad << 16 | idx): first minimum in scan order;
//   D  one lane per block: record, half-pixel refinement from LDS, votes; wave 0 finalises the
//      level's flow record; the level-1 predictor reaches level 0 through LDS.
// Nothing but the block records, the flow records and the sums leaves the chip (the level-1 frames of
// the workspace stay untouched, as under the fused coarse kernel).  Results are those of the separate
// kernels bit for bit.
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_reduce.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBins = 64;   // n = 2(2R+1)+1: 19 at level 1 (and for one level), 55 at level 0 of two (S = 4)
constexpr int kLoads = 8;      // 16-byte chunks a thread has in flight in pass A
constexpr int kPad = 16;       // bytes after each LDS frame: the dword reads of the last row may run past it

__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

// Unaligned reads from an LDS frame: aligned dwords, funnel-shifted by the byte offset's low bits.
__device__ __forceinline__ uint4 lds_bytes16(const uint8_t *frame, int off)
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(frame) + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    return make_uint4(__builtin_amdgcn_alignbyte(q1, q0, sh), __builtin_amdgcn_alignbyte(q2, q1, sh),
                      __builtin_amdgcn_alignbyte(q3, q2, sh), __builtin_amdgcn_alignbyte(q4, q3, sh));
}
__device__ __forceinline__ void lds_bytes8(const uint8_t *frame, int off, uint32_t (&out)[2])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(frame) + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
    out[0] = __builtin_amdgcn_alignbyte(q1, q0, sh);
    out[1] = __builtin_amdgcn_alignbyte(q2, q1, sh);
}
__device__ __forceinline__ void lds_bytes12(const uint8_t *frame, int off, uint32_t (&out)[3])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(frame) + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    out[0] = __builtin_amdgcn_alignbyte(q1, q0, sh);
    out[1] = __builtin_amdgcn_alignbyte(q2, q1, sh);
    out[2] = __builtin_amdgcn_alignbyte(q3, q2, sh);
}

// 4x4 gradient gate on tile bytes [2..5] x rows [2..5] (same arithmetic as aof_lane8.hpp)
__device__ __forceinline__ uint32_t gate_4x4(const uint32_t (&ref)[8][2])
{
    uint32_t mid[4], diff = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) mid[r] = __builtin_amdgcn_alignbyte(ref[r + 2][1], ref[r + 2][0], 2);
#pragma unroll
    for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
    for (int r = 0; r < 4; r++)
        diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
    return diff;
}

struct LevelMeta { int px, py, delta; };

// One level of one pair out of LDS: search, records, refinement, votes, flow record.
// `keys`: one packed key per block; `record_out`: LDS copy of the flow record (level 1: it carries the
// predictor); `pred_rec`: the level-1 record whose predictor fields level 0 copies into its own.
template <bool SUBPIXEL>
__device__ __forceinline__ void run_level(const SearchArgs &a, const FlowTail &tail, uint32_t pair, const uint8_t *fp,
                                          const uint8_t *fc, const LevelMeta &m, uint32_t *keys,
                                          uint32_t (*votes)[kMaxBins], int *tot, aof_flow *record_out,
                                          const aof_flow *pred_rec)
{
    const int tid = threadIdx.x, W = a.w;
    const int nb = a.grid.blocks(), nx = a.grid.nx;
    const int centre = 2 * tail.range + 1;
    constexpr int ring = SUBPIXEL ? 1 : 0;
    for (int k = tid; k < nb; k += kThreads) keys[k] = 0xFFFFFFFFu;
    if (tid < kMaxBins) { votes[0][tid] = 0; votes[1][tid] = 0; }
    if (tid < 3) tot[tid] = 0;
    __syncthreads();

    // ---- C: one lane per (block, dy) ----
    for (int it = tid; it < nb * 9; it += kThreads) {
        const int blk = it / 9, d = it - blk * 9;
        const int by = blk / nx, bx = blk - by * nx;
        const int i = a.grid.x0 + bx * a.grid.step_x, j = a.grid.y0 + by * a.grid.step_y;
        const int wx0 = i + m.px - 4, wy0 = j + m.py - 4;
        // the search window (plus the half-pixel ring) must lie inside the frame
        if (wx0 - ring < 0 || wy0 - ring < 0 || wx0 + 16 + ring > a.w || wy0 + 16 + ring > a.h) continue;
        uint32_t ref[8][2];
#pragma unroll
        for (int r = 0; r < 8; r++) lds_bytes8(fp, (j + r) * W + i, ref[r]);
        if (gate_4x4(ref) < (uint32_t)a.feature_threshold) continue;
        u64 lo = 0, hi = 0;
        uint32_t k8 = (uint32_t)(d * 9 + 8);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 w = lds_bytes16(fc, (wy0 + d + r) * W + wx0);
            if (m.delta != 0) w = sat_add_u8x16(w, m.delta);
            const u64 p01 = pack64(w.x, w.y), p12 = pack64(w.y, w.z), p23 = pack64(w.z, w.w);
            lo = __builtin_amdgcn_qsad_pk_u16_u8(p01, ref[r][0], lo);
            lo = __builtin_amdgcn_qsad_pk_u16_u8(p12, ref[r][1], lo);
            hi = __builtin_amdgcn_qsad_pk_u16_u8(p12, ref[r][0], hi);
            hi = __builtin_amdgcn_qsad_pk_u16_u8(p23, ref[r][1], hi);
            k8 = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], k8);
            k8 = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], k8);
        }
        const uint32_t base = (uint32_t)(d * 9);
        const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32), h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
        uint32_t best = min(min((l0 << 16) | (base + 0), (l0 & 0xFFFF0000u) | (base + 1)),
                            min((l1 << 16) | (base + 2), (l1 & 0xFFFF0000u) | (base + 3)));
        best = min(best, min(min((h0 << 16) | (base + 4), (h0 & 0xFFFF0000u) | (base + 5)),
                             min((h1 << 16) | (base + 6), (h1 & 0xFFFF0000u) | (base + 7))));
        best = min(best, k8);
        atomicMin(&keys[blk], best);
    }
    __syncthreads();

    // ---- D1 (half-pixel refinement): four lanes per accepted block, two tile rows each with their window
    // rows -1 .. 2 relative to the best match (aof_refine.hpp); the eight direction sums of the four slices
    // add up across the quad (integer sums: any order) and the direction rides back in bits 8..11 of the
    // block's key (idx < 81 needs seven).  A quarter of the dependent row steps of one lane per block: this
    // is the latency path.  Whole waves: shuffles.
    if constexpr (SUBPIXEL) {
        constexpr int kParts = 4, kRows = 8 / kParts;
        for (int q = tid; q < (kParts * nb + 63) / 64 * 64; q += kThreads) {
            const int blk = q / kParts, part = q % kParts;
            const uint32_t key = blk < nb ? keys[blk] : 0xFFFFFFFFu;
            const bool refine = key != 0xFFFFFFFFu && (key >> 16) < (uint32_t)a.value_threshold;
            RefineState<2, kRows> st;
            st.init();
            if (refine && (key >> 16) != 0) {   // (nothing is below a SAD of zero: such a match keeps "none" from the empty sums)
                const int idx = (int)(key & 0xFFFFu);
                const int by = blk / nx, bx = blk - by * nx;
                const int i = a.grid.x0 + bx * a.grid.step_x, j = a.grid.y0 + by * a.grid.step_y + kRows * part;
                const int rx = i + m.px - 4 + idx % 9 - 1, ry = j + m.py - 4 + idx / 9 - 1;
                uint32_t ref[kRows][2];
#pragma unroll
                for (int r = 0; r < kRows; r++) lds_bytes8(fp, (j + r) * W + i, ref[r]);
                for_rows<-1, kRows>([&](auto yc) {
                    constexpr int Y = decltype(yc)::value;
                    uint32_t dd[3];
                    lds_bytes12(fc, (ry + Y + 1) * W + rx, dd);
                    if (m.delta != 0) {
#pragma unroll
                        for (int k = 0; k < 3; k++) dd[k] = sat_add_u8x4(dd[k], m.delta);
                    }
                    st.template row<Y>(dd, ref);
                });
            }
#pragma unroll
            for (int o = 1; o < kParts; o <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; k++) st.acc[k] += (uint32_t)__shfl_xor((int)st.acc[k], o, 64);
            }
            // (the quad's four lanes read the key above, in the same wave, before this write)
            if (refine && part == 0) keys[blk] = key | ((uint32_t)st.direction(key >> 16) << 8);
        }
        __syncthreads();
    }

    // ---- D: one lane per block ----
    const bool live = tid < nb;   // nb <= kThreads
    aof_block rec;
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    int subdir = 8;
    if (live) {
        const uint32_t key = keys[tid];
        const size_t item = (size_t)pair * (size_t)nb + (size_t)tid;
        if (key != 0xFFFFFFFFu) {
            const int idx = (int)(key & 0xFFu);
            rec.dx = (int8_t)(m.px + idx % 9 - 4);
            rec.dy = (int8_t)(m.py + idx / 9 - 4);
            rec.sad = (uint16_t)(key >> 16);
            if constexpr (SUBPIXEL) {
                if ((uint32_t)rec.sad < (uint32_t)a.value_threshold) subdir = (int)((key >> 8) & 0xFu);
            }
        }
        reinterpret_cast<uint32_t *>(a.blocks)[item] = __builtin_bit_cast(uint32_t, rec);
        if (SUBPIXEL) a.subdirs[item] = (uint8_t)subdir;
    }
    const bool ok = live && (uint32_t)rec.sad < (uint32_t)a.value_threshold;  // skipped = 0xFFFF
    const int hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
    const int hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
    const int vx = 2 * rec.dx + hx, vy = 2 * rec.dy + hy;
    wave_vote2(votes[0], votes[1], vx + centre, vy + centre, ok);
    const int s2x = (int)wave_sum_u32((uint32_t)(ok ? vx : 0)), s2y = (int)wave_sum_u32((uint32_t)(ok ? vy : 0));
    const int cnt = (int)wave_sum_u32(ok ? 1u : 0u);
    if ((tid & 63) == 0 && cnt) {
        atomicAdd(&tot[0], s2x);
        atomicAdd(&tot[1], s2y);
        atomicAdd(&tot[2], cnt);
    }
    __syncthreads();
    if (tid < 64) finalise_flow_wave(tail, pair, votes[0], votes[1], tot, record_out, pred_rec);   // bins <= 64 (launcher)
}

// One pair: passes A..D.  The workgroup keeps two frame buffers in LDS (each with its level-1 image and
// its pixel sums); `src[b]` is the frame that belongs in buffer b (device or pinned host memory),
// `cur_buf` says which buffer holds the newer frame of the pair, and bit b of `load` says whether buffer
// b has to be fetched (and its level-1 image and sums made) -- the resident kernel keeps the frame of its
// previous call in LDS and fetches only the new one.
template <bool SUBPIXEL>
__device__ __forceinline__ void flow_small_pair(const SmallArgs &a, uint32_t pair, const uint8_t *src0, const uint8_t *src1,
                                                int cur_buf, uint32_t load, aof_flow *final_copy = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    __shared__ uint32_t s_keys[kThreads];
    __shared__ uint32_t s_votes[2][kMaxBins];
    __shared__ int s_tot[3];
    __shared__ aof_flow s_flow1;     // the level-1 flow record: predictor of level 0
    __shared__ uint32_t s_sums[4];   // [buffer][level]
    const int tid = threadIdx.x;
    const int w = a.l0.w, h = a.l0.h, w1 = w / 2, h1 = h / 2;
    const int frame0 = w * h, frame1 = w1 * h1;
    const bool two = a.levels == 2;
    uint8_t *f0[2] = {s_mem, s_mem + frame0 + kPad};
    uint8_t *f1[2] = {s_mem + 2 * (frame0 + kPad), s_mem + 2 * (frame0 + kPad) + ((frame1 + kPad + 15) & ~15)};
    const int first = (load & 1u) ? 0 : 1, count = (load == 3u) ? 2 : (load ? 1 : 0);   // buffers to fetch: first, first + 1, ...

    if (tid < 4 && ((load >> (tid >> 1)) & 1u)) s_sums[tid] = 0;
    __syncthreads();

    // ---- A: level-0 frames -> LDS (a frame is contiguous: row stride == w) ----
    {
        const int per_frame = frame0 / 16, items = count * per_frame;
        uint32_t sum[2] = {0, 0};
        for (int base = 0; base < items; base += kLoads * kThreads) {
            uint4 v[kLoads];
#pragma unroll
            for (int k = 0; k < kLoads; k++) {
                const int it = base + k * kThreads + tid;
                v[k] = make_uint4(0, 0, 0, 0);
                if (it < items) {
                    const int buf = first + (it >= per_frame), c = it - (it >= per_frame) * per_frame;
                    v[k] = *reinterpret_cast<const uint4 *>((buf ? src1 : src0) + c * 16);
                }
            }
#pragma unroll
            for (int k = 0; k < kLoads; k++) {
                const int it = base + k * kThreads + tid;
                if (it >= items) continue;
                const int buf = first + (it >= per_frame), c = it - (it >= per_frame) * per_frame;
                uint32_t s = 0;
                s = byte_sum(v[k].x, s); s = byte_sum(v[k].y, s);
                s = byte_sum(v[k].z, s); s = byte_sum(v[k].w, s);
                sum[buf] += s;
                *reinterpret_cast<uint4 *>(f0[buf] + c * 16) = v[k];
            }
        }
        if (a.sums) {
#pragma unroll
            for (int buf = 0; buf < 2; buf++) {
                const uint32_t t = wave_sum_u32(sum[buf]);
                if ((tid & 63) == 0 && ((load >> buf) & 1u)) atomicAdd(&s_sums[buf * 2], t);
            }
        }
    }
    __syncthreads();

    // ---- B: level-1 frames, LDS -> LDS ----
    if (two) {
        const int chunks = w / 16, per_frame = h1 * chunks;
        uint32_t sum[2] = {0, 0};
        for (int it = tid; it < count * per_frame; it += kThreads) {
            const int buf = first + (it >= per_frame), rest = it - (it >= per_frame) * per_frame;
            const int y1 = rest / chunks, c = rest - y1 * chunks;
            const uint4 r0 = *reinterpret_cast<const uint4 *>(f0[buf] + (2 * y1) * w + c * 16);
            const uint4 r1 = *reinterpret_cast<const uint4 *>(f0[buf] + (2 * y1 + 1) * w + c * 16);
            const uint32_t p0 = box2(r0.x, r1.x), p1 = box2(r0.y, r1.y);
            const uint32_t p2 = box2(r0.z, r1.z), p3 = box2(r0.w, r1.w);
            uint2 o;
            o.x = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
            o.y = __builtin_amdgcn_perm(p3, p2, 0x06040200u);
            sum[buf] = byte_sum(o.x, sum[buf]);
            sum[buf] = byte_sum(o.y, sum[buf]);
            *reinterpret_cast<uint2 *>(f1[buf] + y1 * w1 + c * 8) = o;
        }
        if (a.sums) {
#pragma unroll
            for (int buf = 0; buf < 2; buf++) {
                const uint32_t t = wave_sum_u32(sum[buf]);
                if ((tid & 63) == 0 && ((load >> buf) & 1u)) atomicAdd(&s_sums[buf * 2 + 1], t);
            }
        }
        __syncthreads();
    }

    const int pb = 1 - cur_buf;   // the buffer of the older frame
    LevelMeta m0 = {0, 0, 0}, m1 = {0, 0, 0};
    if (a.sums) {
        const uint32_t n0 = (uint32_t)frame0;
        m0.delta = (int)((s_sums[pb * 2] + n0 / 2) / n0) - (int)((s_sums[cur_buf * 2] + n0 / 2) / n0);
        if (two) {
            const uint32_t n1 = (uint32_t)frame1;
            m1.delta = (int)((s_sums[pb * 2 + 1] + n1 / 2) / n1) - (int)((s_sums[cur_buf * 2 + 1] + n1 / 2) / n1);
        }
        // workspace layout: [frame: 0 prev, 1 cur][level]
        if (tid < 4) a.sums[(size_t)pair * 4 + tid] = s_sums[((tid >> 1) ? cur_buf : pb) * 2 + (tid & 1)];
    }

    if (two) {
        run_level<SUBPIXEL>(a.l1, a.t1, pair, f1[pb], f1[cur_buf], m1, s_keys, s_votes, s_tot, &s_flow1, nullptr);
        __syncthreads();
        m0.px = s_flow1.pred_x;
        m0.py = s_flow1.pred_y;
    }
    // (final_copy: an LDS copy of the pair's flow record for the caller; visible after its next barrier)
    run_level<SUBPIXEL>(a.l0, a.t0, pair, f0[pb], f0[cur_buf], m0, s_keys, s_votes, s_tot, final_copy, two ? &s_flow1 : nullptr);
}

template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads) void k_flow_small(SmallArgs a)   // (latency path: occupancy does not matter)
{
    const uint32_t pair = blockIdx.x;
    flow_small_pair<SUBPIXEL>(a, pair, a.l0.prev + (int64_t)pair * a.l0.pair_stride, a.l0.cur + (int64_t)pair * a.l0.pair_stride,
                              1, 3u);
}

// The per-call path (one pair per launch, record in pinned host memory): the record leaves with ONE 16-byte
// store whose `count` carries the low byte of *tag in its top byte (block counts stay below 2^24) -- the
// host, which wrote the tag before the launch, polls the record for it instead of waiting for the stream
// to drain (the runtime's completion path costs more than the kernel).
template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads) void k_flow_small_tagged(SmallArgs a, aof_flow *host_record, const uint32_t *tag)
{
    __shared__ aof_flow s_record;
    uint32_t t = 0;
    if (threadIdx.x == 0) t = __hip_atomic_load(tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (in flight beside the frames)
    flow_small_pair<SUBPIXEL>(a, 0, a.l0.prev, a.l0.cur, 1, 3u, &s_record);
    __syncthreads();
    if (threadIdx.x == 0) {
        aof_flow r = s_record;
        r.count = (r.count & 0x00FFFFFFu) | (t << 24);
        *reinterpret_cast<uint4 *>(host_record) = __builtin_bit_cast(uint4, r);
    }
}

// ---- the resident form of the per-call path (aof_set_stream_resident) --------------------------------
// calcFlow() hands over ONE small frame per call (mainloop.cpp:322), and 20 of the 25 us such a call takes
// through a replayed hipGraph are the runtime's launch and completion, not the 4 us kernel.  Here ONE
// workgroup stays on the device between calls: the host writes the frame into its pinned ping-pong slot and
// bumps a request word in pinned memory; lane 0 polls that word, the workgroup computes the pair exactly
// as k_flow_small does (same function) and publishes the 16-byte record in pinned memory with ONE store,
// the request's low byte riding in the top byte of `count`, which the host polls for.  The kernel ALWAYS ends by itself: after `idle_ticks` of the 100 MHz
// real-time counter without a request, after `life_ticks` in total (so that nothing that waits for the
// device to drain -- a hipFree anywhere in the process -- waits longer than that), or when the host sets
// the stop word; the next call starts it again.
template <bool SUBPIXEL>
__global__ __launch_bounds__(kThreads) void k_flow_resident(SmallArgs a, ResidentBox *box, aof_flow *host_record,
                                                            const uint8_t *frame_a, const uint8_t *frame_b, uint32_t served,
                                                            uint32_t launch_no, uint64_t idle_ticks, uint64_t life_ticks, int deaf)
{
    __shared__ uint32_t s_req[3];   // request number (0 = leave), slot of the newest frame, buffers to fetch
    __shared__ aof_flow s_record;   // the call's flow record (the kernel's own copy goes to device memory)
    const uint64_t born = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) __hip_atomic_store(&box->started, launch_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint64_t idle_since = born;
    uint32_t held[2] = {0u, 0u};    // (thread 0) tag of the request at which LDS buffer b received pinned frame b, 0 = never
    for (;;) {
        if (threadIdx.x == 0) {
            uint32_t req = 0, slot = 0, load = 3u;
            unsigned long long word = 0;
            for (;;) {
                word = __hip_atomic_load(&box->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // one PCIe read
                // (deaf: fault injection of the tests -- the stop bit is ignored, the kernel neither serves nor leaves
                //  when asked and goes on its idle / lifetime deadline only: aof_debug_resident_fault)
                const bool stop = (word & kResidentStopBit) != 0;
                if (!stop && (uint32_t)word != served) { req = (uint32_t)word; break; }
                const uint64_t now = __builtin_amdgcn_s_memrealtime();
                if ((stop && !deaf) || now - idle_since > idle_ticks || now - born > life_ticks) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (req) {
                slot = (uint32_t)(word >> 32) & 1u;
                // the older frame of this pair sits in pinned frame 1 - slot; the host says at which request
                // it was posted (0: not through a request) -- if that is when this workgroup fetched its
                // LDS buffer 1 - slot, the copy is still good and only the new frame crosses PCIe
                const uint32_t prev_tag = (uint32_t)(word >> 33) & 0x7FFFFFFFu;   // (low 30 bits of the request) + 1
                load = (prev_tag != 0u && held[1u - slot] == prev_tag) ? (1u << slot) : 3u;
                held[slot] = (req & 0x3FFFFFFFu) + 1u;
                if (load == 3u) held[1u - slot] = prev_tag;
            }
            s_req[0] = req;
            s_req[1] = slot;
            s_req[2] = load;
        }
        __syncthreads();
        const uint32_t req = s_req[0], slot = s_req[1], load = s_req[2];
        if (req == 0) break;   // (uniform)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // the host's frame bytes: nothing stale out of L1 / L2
        flow_small_pair<SUBPIXEL>(a, 0, frame_a, frame_b, (int)slot, load, &s_record);
        __syncthreads();
        if (threadIdx.x == 0) {
            // ONE 16-byte store publishes the record: the low byte of the request number rides in the top
            // byte of `count` (block counts stay below 2^24), the host polls for it and masks it out -- the host
            // does not wait for a second PCIe write behind a fence.  `done` follows for the next kernel instance
            // (read only after this one has left).
            aof_flow r = s_record;
            r.count = (r.count & 0x00FFFFFFu) | (req << 24);
            *reinterpret_cast<uint4 *>(host_record) = __builtin_bit_cast(uint4, r);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // pushes the record out (the host is polling for it already)
            __hip_atomic_store(&box->done, req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        served = req;
        idle_since = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&box->exited, served, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&box->running, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

size_t small_lds_bytes(const SmallArgs &a)
{
    const size_t frame0 = (size_t)a.l0.w * a.l0.h, frame1 = (size_t)(a.l0.w / 2) * (a.l0.h / 2);
    size_t bytes = 2 * (frame0 + kPad);
    if (a.levels == 2) bytes += 2 * ((frame1 + kPad + 15) & ~(size_t)15);
    return bytes;
}

}  // namespace

bool flow_small_supported(const SmallArgs &a)
{
    const SearchArgs &l0 = a.l0, &l1 = a.l1;
    if (a.levels != 1 && a.levels != 2) return false;
    if (l0.tile != 8 || l0.search != 4) return false;
    if (l0.n_pairs < 1 || l0.n_pairs > 0x7FFFFFFF / kThreads) return false;
    if (l0.grid.blocks() < 1 || l0.grid.blocks() > kThreads) return false;
    if (2 * (2 * a.t0.range + 1) + 1 > kMaxBins) return false;
    // 16-byte chunks of a contiguous frame
    if (l0.w % 16 || (l0.n_pairs > 1 && l0.pair_stride % 16)) return false;
    if (reinterpret_cast<uintptr_t>(l0.prev) % 16 || reinterpret_cast<uintptr_t>(l0.cur) % 16) return false;
    if (l0.subpixel && !l0.subdirs) return false;
    if (a.levels == 2) {
        if (l0.h % 2 || l1.w != l0.w / 2 || l1.h != l0.h / 2) return false;
        if (l1.grid.blocks() < 1 || l1.grid.blocks() > kThreads) return false;
        if (2 * (2 * a.t1.range + 1) + 1 > kMaxBins) return false;
        if (l0.subpixel && !l1.subdirs) return false;
    }
    return small_lds_bytes(a) + 4096 <= 160 * 1024;   // frames + the kernel's static arrays
}

int launch_flow_resident(const SmallArgs &a, ResidentBox *box, aof_flow *host_record, const uint8_t *frame_a,
                         const uint8_t *frame_b, uint32_t served, uint32_t launch_no, uint64_t idle_ticks,
                         uint64_t life_ticks, bool deaf, void *stream)
{
    if (a.l0.n_pairs != 1 || !flow_small_supported(a)) return (int)hipErrorInvalidValue;
    void (*fn)(SmallArgs, ResidentBox *, aof_flow *, const uint8_t *, const uint8_t *, uint32_t, uint32_t, uint64_t, uint64_t, int) =
        a.l0.subpixel ? k_flow_resident<true> : k_flow_resident<false>;
    const size_t lds = small_lds_bytes(a);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3(1), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a, box, host_record, frame_a, frame_b,
                       served, launch_no, idle_ticks, life_ticks, deaf ? 1 : 0);
    return (int)hipGetLastError();
}

int launch_flow_small_tagged(const SmallArgs &a, aof_flow *host_record, const uint32_t *tag, void *stream)
{
    if (a.l0.n_pairs != 1 || !flow_small_supported(a) || !host_record || !tag) return (int)hipErrorInvalidValue;
    void (*fn)(SmallArgs, aof_flow *, const uint32_t *) = a.l0.subpixel ? k_flow_small_tagged<true> : k_flow_small_tagged<false>;
    const size_t lds = small_lds_bytes(a);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3(1), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a, host_record, tag);
    return (int)hipGetLastError();
}

int launch_flow_small(const SmallArgs &a, void *stream)
{
    if (a.l0.n_pairs == 0) return 0;
    if (!flow_small_supported(a)) return (int)hipErrorInvalidValue;
    void (*fn)(SmallArgs) = a.l0.subpixel ? k_flow_small<true> : k_flow_small<false>;
    const size_t lds = small_lds_bytes(a);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3((uint32_t)a.l0.n_pairs), dim3(kThreads), lds, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // namespace aof
