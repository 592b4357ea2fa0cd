// K1C -- the coarse passes of a two-level pair in ONE kernel (DESIGN.md "Kernels": K1C):
// pixel sums + 2x2 box pyramid (K1), the level-1 8x8 SAD search over +-4 (K2 at level 1) and
// its histogram-filtered reduction to the level-0 predictor (K3 at level 1).
//
// One workgroup of 512 threads owns one frame pair and keeps BOTH level-1 frames in LDS
// (2 * W/2 * H/2 bytes: 150 KB at VGA -- the reason CDNA4's 160 KB LDS per CU matters here), so
// the level-1 frames are never written to HBM and never read back:
//   phase 1  streams the pair's two frames once (16 B per lane from two adjacent rows, many loads
//            in flight), box-filters them into LDS and sums the level-0 / level-1 bytes;
//   phase 2  equalises the level-1 `cur` frame in place (each pixel once, not once per window);
//   phase 3  gates every block (4x4 gradient) and searches: one lane per (block, three dy rows) item,
//            dy-major, so a wave reads 64 neighbouring windows of one row -- conflict-free
//            ds_read_b64 --; three neighbouring dy rows share ten window rows and the tile (28 reads
//            where three one-row items made 72; round 5: 168 -> 157 us per 1 024 VGA pairs), 3 * nb
//            items fill 512 lanes to 95 %; the three items of a block meet in an LDS atomicMin on
//            the packed key (sad << 16 | idx) = first minimum wins;
//   phase 4  writes the level-1 records, votes, and wave 0 finalises the predictor -- while the other
//            waves already stream the workgroup's next pair (two sets of histograms take turns).
// HBM traffic: the frames once, 4.5 KB of records and 16 B of predictor per pair.
//
// Bound: phase 1 by HBM (2 * W * H bytes per pair), phase 3 by the SAD issue rate; workgroups
// on different CUs drift out of phase, so the chip overlaps the two.
#include <climits>
#include <cstdint>

#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_lab_hooks.hpp"
#include "aof_reduce.hpp"

namespace aof {

namespace {

// 512 lanes = two waves per SIMD (148 VGPRs since the search sums three dy rows per lane): alone the kernel is as fast
// as with 1 024 lanes (171 us against 174 us per 1 024 VGA pairs, round 3), and it leaves 216 of a SIMD's 512 registers
// and two wave slots to the level-0 search waves of ANOTHER batch in flight (bench.py --streams 2)
constexpr int kThreads = 512;
constexpr int kUnroll = 4;            // sweeps per batch; two batches = 16 loads of 16 B in flight per lane
constexpr uint32_t kGated = 0xFFFFFFFEu;   // key of a block the gradient gate rejected
constexpr uint32_t kOpen = 0xFFFFFFFFu;    // key of a block still waiting for its first candidate
constexpr int kMaxBins = 64;
constexpr int kNonTemporal = 2;                // buffer-load cache policy: nt (the frames are streamed once)
constexpr int kScratch = 8;                    // words behind the histograms: pixel sums, vote sums
constexpr int kStaggerGroups = 3;              // lab sweep (tools/coarse_lab.hip): 1: 0.215, 2: 0.206, 3: 0.203, 4: 0.212 ms
constexpr int64_t kStaggerBytesPerUs = 75000;  // start-up spacing of the groups: ~8 us at VGA

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() carries a workgroup
// fence, for which the compiler drains EVERY outstanding vector-memory operation (s_waitcnt
// vmcnt(0)) -- here that would wait for the next pair's prefetched frame rows at every phase
// boundary.  The hardware barrier itself does not drain loads; LDS traffic is ordered by lgkmcnt.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(kThreads) void k_coarse(CoarseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int w1 = a.w / 2, h1 = a.h / 2;
    const int l1_frame = w1 * h1;
    uint8_t *l1[2] = {lds, lds + l1_frame};                 // prev, cur
    uint32_t *keys = reinterpret_cast<uint32_t *>(lds + 2 * l1_frame);
    const int nb = a.grid.blocks();
    // Vote histograms [2][kMaxBins] + [4] pixel sums + [3] vote sums, TWICE: consecutive pairs of a
    // workgroup alternate between the two sets, so that wave 0 can still finalise one pair's flow
    // record (2 us of dependent divisions) while the other waves already stream the next pair.
    constexpr int kVoteWords = 2 * kMaxBins + kScratch;
    uint32_t *votes0 = keys + nb;
    const int tid = threadIdx.x;

    // Every workgroup takes the same time, so left alone the whole chip streams (HBM saturated,
    // VALU idle) and then searches (HBM idle) in lockstep.  The first generation of workgroups
    // (one per CU) therefore starts in `stagger_groups` groups, each `stagger_ticks` (10 ns units)
    // after the one before; from then on one group streams while the others compute.
    if (a.stagger_groups > 1 && blockIdx.x < (uint32_t)a.first_generation) {
        const uint32_t grp = blockIdx.x % (uint32_t)a.stagger_groups;
        if (grp) {
            const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)grp * a.stagger_ticks;
            while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
        }
    }

    // ---- phase 1 machinery: stream the two frames, box-filter into LDS, byte sums ----
    // A lane keeps its 16-byte column and walks down the 2*h1 level-1 rows of (prev, cur) in
    // sweeps of `rpi` rows: no index arithmetic in the loop.  Two batches of kUnroll sweeps (two
    // 16-byte loads each) are always in flight per lane; the pipeline runs ACROSS pairs: while the
    // last sweeps of a pair are filtered, the first 2*kUnroll sweeps of the workgroup's next pair
    // are already requested, and they land in registers during the search of the current pair --
    // a CU streams from HBM at ~25 GB/s whatever it does, so this is the share of the next pair's
    // stream phase that hides under the search.
    const int chunks = a.w / 16;
    const int rpi = a.rows_per_sweep;                        // level-1 rows per sweep of the workgroup (launcher)
    int yoff = (int)fast_div((uint32_t)tid, a.div_chunks), col = tid - yoff * chunks;
    const bool active = yoff < rpi;
    if (!active) { yoff = 0; col = 0; }   // idle lanes (32 of 512 at VGA: 12 rows of 40 chunks) load valid bytes and drop them
    // sweeps never straddle the two frames: nkf sweeps per frame, prev first (k < nkf), then cur
    const int nkf = (h1 + rpi - 1) / rpi, nk = 2 * nkf;
    const int nk_pad = (nk + 2 * kUnroll - 1) / (2 * kUnroll) * (2 * kUnroll);   // whole rounds of two batches
    // Loads through buffer resources: frame base in scalar registers, ONE 32-bit byte offset per
    // lane that advances by a scalar step per sweep -- a 64-bit address pair per load in flight
    // would cost 40 VGPRs here.  Rows past the frame's end re-read its last row pair (dropped in
    // filter_batch): straight-line code, a branch would drain the loads in flight at every join.
    const uint32_t frame_bytes = (uint32_t)(a.w * a.h);
    const uint32_t voff0 = (uint32_t)(2 * yoff * a.w + col * 16), vstep = (uint32_t)(2 * rpi * a.w);
    const uint32_t vlast = (uint32_t)((a.h - 2) * a.w + col * 16);
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    struct Batch { u32x4_t q[kUnroll][2]; };
    // sweeps k0 .. k0+kUnroll-1 of the pair at `pair`; k0 >= nk_pad continues in the pair at `next`
    auto load_batch = [&](int64_t pair, int64_t next, int k0, Batch &b) {
        const int64_t pr = k0 >= nk_pad ? next : pair;
        const int kb = k0 >= nk_pad ? k0 - nk_pad : k0;
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int k = kb + u;                                   // (scalar)
            const bool in_cur = k >= nkf;
            const uint8_t *base = (in_cur ? a.cur : a.prev) + pr * a.pair_stride;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, frame_bytes, kRawBuffer);
            const uint32_t kk = (uint32_t)(in_cur ? k - nkf : k);
            const uint32_t off = min(voff0 + kk * vstep, vlast);
            b.q[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kNonTemporal);
            b.q[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, a.w, kNonTemporal);
        }
    };
    uint32_t sum_p0 = 0, sum_p1 = 0, sum_c0 = 0, sum_c1 = 0;   // level-0 / level-1 byte sums of prev, cur
    // (a+b+c+d+2)>>2 of a 2x2 cell = byte 1 of 64*(a+b+c+d) + 128: two v_dot4_u32_u8 per cell
    // (weights 64 on the cell's two bytes of each row), three v_perm_b32 gather four cells.
    auto box4 = [](uint32_t r0a, uint32_t r1a, uint32_t r0b, uint32_t r1b) -> uint32_t {
        const uint32_t wl = 0x00004040u, wh = 0x40400000u;
        uint32_t t0 = __builtin_amdgcn_udot4(r0a, wl, 128u, false); t0 = __builtin_amdgcn_udot4(r1a, wl, t0, false);
        uint32_t t1 = __builtin_amdgcn_udot4(r0a, wh, 128u, false); t1 = __builtin_amdgcn_udot4(r1a, wh, t1, false);
        uint32_t t2 = __builtin_amdgcn_udot4(r0b, wl, 128u, false); t2 = __builtin_amdgcn_udot4(r1b, wl, t2, false);
        uint32_t t3 = __builtin_amdgcn_udot4(r0b, wh, 128u, false); t3 = __builtin_amdgcn_udot4(r1b, wh, t3, false);
        const uint32_t h01 = __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u), h23 = __builtin_amdgcn_perm(t3, t2, 0x0c0c0501u);
        return __builtin_amdgcn_perm(h23, h01, 0x05040100u);
    };
    auto filter_batch = [&](int k0, const Batch &b) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int k = k0 + u;                                      // (scalar)
            const bool in_cur = k >= nkf;
            const int y1 = yoff + (in_cur ? k - nkf : k) * rpi;        // level-1 row inside the frame
            const int y = in_cur ? h1 + y1 : y1;                       // row of the (prev, cur) LDS image
            if (active && y1 < h1 && k < nk) {
                const u32x4_t r0 = b.q[u][0], r1 = b.q[u][1];
                uint32_t s0 = 0;
                s0 = byte_sum(r0.x, s0); s0 = byte_sum(r0.y, s0); s0 = byte_sum(r0.z, s0); s0 = byte_sum(r0.w, s0);
                s0 = byte_sum(r1.x, s0); s0 = byte_sum(r1.y, s0); s0 = byte_sum(r1.z, s0); s0 = byte_sum(r1.w, s0);
                uint2 o;
                o.x = box4(r0.x, r1.x, r0.y, r1.y);
                o.y = box4(r0.z, r1.z, r0.w, r1.w);
                const uint32_t s1 = byte_sum(o.y, byte_sum(o.x, 0u));
                if (!in_cur) { sum_p0 += s0; sum_p1 += s1; } else { sum_c0 += s0; sum_c1 += s1; }
                *reinterpret_cast<uint2 *>(lds + y * w1 + col * 8) = o;   // l1[0] and l1[1] are contiguous
            }
        }
    };

    // Persistent workgroups: pair, pair + gridDim.x, ...  (one workgroup per CU: LDS)
    Batch A, B;
    {
        const int64_t p0 = blockIdx.x;
        const int64_t p1 = p0 + gridDim.x < a.n_pairs ? p0 + gridDim.x : p0;
        load_batch(p0, p1, 0, A);
        load_batch(p0, p1, kUnroll, B);
    }
    for (int k = tid; k < 2 * kVoteWords; k += kThreads) votes0[k] = 0;
    lds_barrier();
    int par = 0;
#pragma unroll 1
    for (int64_t pair = blockIdx.x; pair < a.n_pairs; pair += gridDim.x, par ^= 1) {
    const int64_t next = pair + gridDim.x < a.n_pairs ? pair + gridDim.x : pair;   // (last pair: harmless re-reads)
    uint32_t *hist = votes0 + par * kVoteWords;              // [2][kMaxBins]
    uint32_t *sums = hist + 2 * kMaxBins;                    // [4] pixel sums, [3] vote sums
    AOF_LAB_STAMP(pair, 0);
    sum_p0 = sum_p1 = sum_c0 = sum_c1 = 0;
#pragma unroll 1
    for (int k0 = 0; k0 < nk_pad; k0 += 2 * kUnroll) {
        filter_batch(k0, A);
        load_batch(pair, next, k0 + 2 * kUnroll, A);
        filter_batch(k0 + kUnroll, B);
        load_batch(pair, next, k0 + 3 * kUnroll, B);
    }
    sum_p0 = wave_sum_u32(sum_p0); sum_p1 = wave_sum_u32(sum_p1);
    sum_c0 = wave_sum_u32(sum_c0); sum_c1 = wave_sum_u32(sum_c1);
    if ((tid & 63) == 0) {
        atomicAdd(&sums[0], sum_p0); atomicAdd(&sums[1], sum_p1);
        atomicAdd(&sums[2], sum_c0); atomicAdd(&sums[3], sum_c1);
    }
    lds_barrier();
    AOF_LAB_STAMP(pair, 1);
    if (tid == 0) {   // (wave-uniform addresses: nothing for the compiler to keep per lane across the pair loop)
        if (a.sums) *reinterpret_cast<uint4 *>(a.sums + pair * 4) = make_uint4(sums[0], sums[1], sums[2], sums[3]);
    }

    // ---- phase 2: equalise the level-1 cur frame in place; gate the blocks ----
    int delta = 0;
    if (a.sums) {
        const uint32_t npix = (uint32_t)l1_frame;
        delta = (int)((sums[1] + npix / 2) / npix) - (int)((sums[3] + npix / 2) / npix);
    }
    if (delta != 0) {
        uint32_t *c32 = reinterpret_cast<uint32_t *>(l1[1]);
        for (int k = tid; k < l1_frame / 4; k += kThreads) c32[k] = sat_add_u8x4(c32[k], delta);
    }
    // the other set of histograms and sums is idle by now (its last reader, wave 0 finalising the
    // previous pair, has passed the barrier above): cleared here for the next pair
    for (int k = tid; k < kVoteWords; k += kThreads) votes0[(par ^ 1) * kVoteWords + k] = 0;
    const int x0 = a.grid.x0, y0 = a.grid.y0;   // dense grid, step 8: windows start at 8-byte columns
    for (int blk = tid; blk < nb; blk += kThreads) {
        const int by = (int)fast_div((uint32_t)blk, a.div_nx), bx = blk - by * a.grid.nx;
        const uint8_t *t = l1[0] + (y0 + 8 * by + 2) * w1 + x0 + 8 * bx;   // tile rows 2..5
        uint32_t mid[4], diff = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(t + r * w1);   // 4-byte aligned (x0 = 4)
            mid[r] = __builtin_amdgcn_alignbyte(q[1], q[0], 2);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
        for (int r = 0; r < 4; r++)
            diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
        keys[blk] = diff >= (uint32_t)a.feature_threshold ? kOpen : kGated;
    }
    lds_barrier();
    AOF_LAB_STAMP(pair, 2);

    // ---- phase 3: one lane per (three dy rows, block) item ----
    // Three neighbouring dy rows of a block share ten window rows and the tile: 28 eight-byte LDS reads for 144 SAD
    // instructions where three single-row items read 72 -- the phase is as much LDS reads, index arithmetic and keys as it is
    // SAD issue.  3 * nb items fill 512 lanes to 95 % (9 * nb single rows: 99 %).
    const int items = 3 * nb;
    for (int item = tid; item < items; item += kThreads) {
        // (everything below 2^24: full-rate 24-bit multiplies instead of quarter-rate 32-bit ones)
        const int t = (int)fast_div((uint32_t)item, a.div_nb), blk = item - __mul24(t, nb);
        if (keys[blk] == kGated) continue;
        const int d0 = 3 * t;
        const int by = (int)fast_div((uint32_t)blk, a.div_nx), bx = blk - __mul24(by, a.grid.nx);
        const uint8_t *ref = l1[0] + __mul24(y0 + 8 * by, w1) + x0 + 8 * bx;
        const uint8_t *win = l1[1] + __mul24(y0 + 8 * by - 4 + d0, w1) + (x0 - 4) + 8 * bx;   // 8-byte aligned
        uint32_t tile[8][2];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(ref + r * w1);   // (x0 = 4: 4-byte aligned)
            tile[r][0] = q[0]; tile[r][1] = q[1];
        }
        u64 lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        uint32_t a8[3];
#pragma unroll
        for (int dd = 0; dd < 3; dd++) a8[dd] = (uint32_t)((d0 + dd) * 9 + 8);
#pragma unroll
        for (int j = 0; j < 10; j++) {   // window row j of the triple: tile row j - dd of dy row d0 + dd
            const uint2 wa = *reinterpret_cast<const uint2 *>(win + j * w1);
            const uint2 wb = *reinterpret_cast<const uint2 *>(win + j * w1 + 8);
            const u64 p01 = pack64(wa.x, wa.y), p12 = pack64(wa.y, wb.x), p23 = pack64(wb.x, wb.y);
#pragma unroll
            for (int dd = 0; dd < 3; dd++) {
                const int r = j - dd;
                if (r < 0 || r >= 8) continue;
                lo[dd] = qsad(p01, tile[r][0], lo[dd]);
                lo[dd] = qsad(p12, tile[r][1], lo[dd]);
                hi[dd] = qsad(p12, tile[r][0], hi[dd]);
                hi[dd] = qsad(p23, tile[r][1], hi[dd]);
                a8[dd] = __builtin_amdgcn_sad_hi_u8(wb.x, tile[r][0], a8[dd]);
                a8[dd] = __builtin_amdgcn_sad_hi_u8(wb.y, tile[r][1], a8[dd]);
            }
        }
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int dd = 0; dd < 3; dd++) {
            const uint32_t base = (uint32_t)((d0 + dd) * 9);
            const uint32_t l0 = (uint32_t)lo[dd], l1w = (uint32_t)(lo[dd] >> 32), h0 = (uint32_t)hi[dd], h1w = (uint32_t)(hi[dd] >> 32);
            const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
            const uint32_t k2 = (l1w << 16) | (base + 2), k3 = (l1w & 0xFFFF0000u) | (base + 3);
            const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
            const uint32_t k6 = (h1w << 16) | (base + 6), k7 = (h1w & 0xFFFF0000u) | (base + 7);
            best = min(best, min(min(k0, k1), k2));
            best = min(best, min(min(k3, k4), k5));
            best = min(best, min(min(k6, k7), a8[dd]));
        }
        atomicMin(&keys[blk], best);
    }
    lds_barrier();
    AOF_LAB_STAMP(pair, 3);

    // ---- phase 4: records, votes, predictor ----
    const int centre = 2 * a.tail.range + 1, n = 2 * centre + 1;
    uint32_t *out = reinterpret_cast<uint32_t *>(a.blocks) + pair * nb;
    int s2x = 0, s2y = 0, cnt = 0;
    const int rounds = (nb + kThreads - 1) / kThreads;   // uniform trip count: the ballots need every lane
    for (int rd = 0; rd < rounds; rd++) {
        const int blk = rd * kThreads + tid;
        aof_block rec;
        rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
        bool ok = false;
        if (blk < nb) {
            const uint32_t key = keys[blk];
            if (key != kGated) {
                const int idx = (int)(key & 0xFFFFu);
                rec.dx = (int8_t)(idx % 9 - 4);
                rec.dy = (int8_t)(idx / 9 - 4);
                rec.sad = (uint16_t)(key >> 16);
                ok = (int)rec.sad < a.value_threshold;
            }
            out[blk] = __builtin_bit_cast(uint32_t, rec);
        }
        wave_vote2(hist, hist + kMaxBins, 2 * rec.dx + centre, 2 * rec.dy + centre, ok);
        if (ok) { s2x += 2 * rec.dx; s2y += 2 * rec.dy; cnt++; }
    }
    s2x = (int)wave_sum_u32((uint32_t)s2x);
    s2y = (int)wave_sum_u32((uint32_t)s2y);
    cnt = (int)wave_sum_u32((uint32_t)cnt);
    int *vs = reinterpret_cast<int *>(sums + 4);
    if ((tid & 63) == 0) {
        atomicAdd(&vs[0], s2x);
        atomicAdd(&vs[1], s2y);
        atomicAdd(&vs[2], cnt);
    }
    lds_barrier();
    AOF_LAB_STAMP(pair, 4);
    if (tid < 64) finalise_flow_wave(a.tail, pair, hist, hist + kMaxBins, vs);   // (kMaxBins = 64 bins at most)
    AOF_LAB_STAMP(pair, 5);
    (void)n;
    // (no barrier: the next pair votes into the other set; this one is cleared two barriers from now)
    }   // next pair of this workgroup
}


#ifdef AOF_LAB_COARSE_WS
#include AOF_LAB_COARSE_WS   // tools/coarse_ws_lab.hpp: the wave-specialised lab form of this kernel (measured, not kept)
#endif

}  // namespace

size_t coarse_lds_bytes(const CoarseArgs &a)
{
    return (size_t)2 * (a.w / 2) * (a.h / 2) + (size_t)a.grid.blocks() * 4 + 2 * (2 * kMaxBins + kScratch) * 4;
}

// One workgroup must hold both level-1 frames, and the windows must sit on 8-byte columns.
bool coarse_fused_supported(const CoarseArgs &a)
{
    if (a.tile != 8 || a.search != 4 || a.subpixel) return false;
    if (a.w % 16 || a.h % 2 || a.pair_stride % 16 || a.w / 16 > kThreads) return false;
    if (reinterpret_cast<uintptr_t>(a.prev) % 16 || reinterpret_cast<uintptr_t>(a.cur) % 16) return false;
    const Grid &g = a.grid;
    if (g.x0 != 4 || g.y0 != 4 || g.step_x != 8 || g.step_y != 8 || g.nx < 1 || g.ny < 1) return false;
    if ((a.w / 2) % 8) return false;   // level-1 rows a multiple of 8 bytes: aligned 8-byte LDS reads
    if (2 * (2 * a.tail.range + 1) + 1 > kMaxBins) return false;
    if (a.n_pairs > 0x7FFFFFFF) return false;
    return coarse_lds_bytes(a) <= 160 * 1024;
}

int launch_coarse_fused(const CoarseArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    CoarseArgs k = a;
    k.div_nb = fastdiv_make((uint32_t)a.grid.blocks());
    k.div_nx = fastdiv_make((uint32_t)a.grid.nx);
    k.div_chunks = fastdiv_make((uint32_t)(a.w / 16));
    {   // Rows per sweep: as many as the workgroup has lanes for, but such that the two frames take a
        // whole number of rounds of two batches -- padding sweeps would be loaded for nothing
        // (VGA: 20 rows = 800 lanes, 24 sweeps, no padding; 25 rows would pad 20 sweeps to 24).
        const int chunks = a.w / 16, h1 = a.h / 2, rmax = kThreads / chunks;
        int best = rmax;
        int64_t best_rows = LLONG_MAX;
        for (int r = rmax; r >= 1 && r >= rmax / 2; r--) {
            const int nk = 2 * ((h1 + r - 1) / r);
            const int nk_pad = (nk + 2 * kUnroll - 1) / (2 * kUnroll) * (2 * kUnroll);
            const int64_t loaded = (int64_t)nk_pad * r;
            if (loaded < best_rows) { best_rows = loaded; best = r; }
        }
        k.rows_per_sweep = best;
    }
    // first_generation: the CU count of the context's device, set by the caller
    if (k.first_generation <= 0) return (int)hipErrorInvalidValue;
    if (k.stagger_groups == 0) {   // (the lab tool sets its own)
        k.stagger_groups = a.n_pairs >= 2 * (int64_t)k.first_generation ? kStaggerGroups : 1;
        // one group's share of the stream phase: the pair's bytes at the rate a CU reaches when
        // only 1/groups of the chip streams
        k.stagger_ticks = (int32_t)((int64_t)2 * a.w * a.h * 100 / kStaggerBytesPerUs);
    }
    const size_t lds = coarse_lds_bytes(a);
    // one workgroup per CU (LDS); each walks pairs blockIdx.x, blockIdx.x + gridDim.x, ...
    const int64_t wgs = a.n_pairs < k.first_generation ? a.n_pairs : k.first_generation;
#ifdef AOF_LAB_COARSE_WS
    {
        int rc = 0;
        if (lab_launch_coarse_ws(k, wgs, lds, stream, &rc)) return rc;
    }
#endif
    {   // (per launch, like the 16x16 kernel: the attribute belongs to the current device's code object)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_coarse),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_coarse, dim3((uint32_t)wgs), dim3(kThreads), lds, static_cast<hipStream_t>(stream), k);
    return (int)hipGetLastError();
}

}  // namespace aof
