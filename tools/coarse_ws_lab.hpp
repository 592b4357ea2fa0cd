// Lab only (not part of the product): the WAVE-SPECIALISED form of the fused coarse kernel, measured in round 4
// and not kept (DESIGN.md "C3's coarse kernel", LAB_LOG.md round 4, profiles/r04_coarse_wave_specialised.txt).
// Built into a lab library by tools/coarse_ws_lab.sh: k_coarse.hip is compiled with
//   -DAOF_LAB_COARSE_WS='"../../tools/coarse_ws_lab.hpp"' -DAOF_WS_SEARCH=512 -DAOF_WS_NK=40 -DAOF_WS_S1=2 -DAOF_WS_S2=34 -DAOF_WS_DEPTH=4
// and includes this file inside namespace aof's anonymous namespace, behind k_coarse; the launcher hook at
// the end takes over launches that give every workgroup two pairs or more and whose geometry matches.

// ---- the wave-specialised form (DESIGN.md "C3's coarse kernel", round 4) -------------------------------
// The kernel above streams a pair and THEN searches it: the CU's HBM share idles during the search and its
// VALUs and LDS during the stream, and 150 KB of level-1 frames admit no second workgroup to fill the gaps.
// Here ONE 1 024-lane workgroup splits into two roles.  Waves 0..7 (SEARCH) run phases 2..4 of pair i out of
// LDS exactly as above.  Waves 8..15 (STREAM) meanwhile stream pair i+1, box-filter it and PARK the level-1
// pixels in registers -- NK sweeps x 8 bytes per lane, 80 VGPRs at VGA: the register file is the one place
// on the CU with room for a second pair -- and write them to LDS once the search has read its last window
// (behind the barrier that ends phase 3; phase 4 only touches the keys).  Barriers are workgroup-wide, so the
// STREAM role passes the SEARCH role's three barriers per pair inside its sweep loop (loads stay in flight
// across them: lds_barrier does not drain vmcnt).
constexpr int kWsThreads = 1024;
#ifndef AOF_WS_SEARCH
#define AOF_WS_SEARCH 512
#define AOF_WS_NK 40
#define AOF_WS_S1 2
#define AOF_WS_S2 34
#define AOF_WS_DEPTH 4
#endif

__device__ __forceinline__ uint32_t box4_cells(uint32_t r0a, uint32_t r1a, uint32_t r0b, uint32_t r1b)
{
    const uint32_t wl = 0x00004040u, wh = 0x40400000u;
    uint32_t t0 = __builtin_amdgcn_udot4(r0a, wl, 128u, false); t0 = __builtin_amdgcn_udot4(r1a, wl, t0, false);
    uint32_t t1 = __builtin_amdgcn_udot4(r0a, wh, 128u, false); t1 = __builtin_amdgcn_udot4(r1a, wh, t1, false);
    uint32_t t2 = __builtin_amdgcn_udot4(r0b, wl, 128u, false); t2 = __builtin_amdgcn_udot4(r1b, wl, t2, false);
    uint32_t t3 = __builtin_amdgcn_udot4(r0b, wh, 128u, false); t3 = __builtin_amdgcn_udot4(r1b, wh, t3, false);
    const uint32_t h01 = __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u), h23 = __builtin_amdgcn_perm(t3, t2, 0x0c0c0501u);
    return __builtin_amdgcn_perm(h23, h01, 0x05040100u);
}

// NK: sweeps per pair and STREAM lane (both frames), S1 / S2: sweeps in front of the SEARCH role's first and
// second barrier (phase 2 is ~1.5 us, phase 3 ~18 us, phase 4 ~5 us of a pair's ~26 us)
// SEARCH_LANES: lanes of the SEARCH role (the other 1 024 - SEARCH_LANES stream); DEPTH: sweeps in flight per STREAM
// lane (two loads of 16 B each); EXACT: the level-1 rows divide into whole sweeps (no sweep straddles the frame's end)
template <int SEARCH_LANES, int NK, int S1, int S2, int DEPTH, bool EXACT>
__global__ __launch_bounds__(kWsThreads) void k_coarse_ws(CoarseArgs a)
{
    constexpr int kWsRole = SEARCH_LANES, kWsDepth = DEPTH;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int w1 = a.w / 2, h1 = a.h / 2;
    const int l1_frame = w1 * h1;
    uint8_t *l1[2] = {lds, lds + l1_frame};                 // prev, cur
    uint32_t *keys = reinterpret_cast<uint32_t *>(lds + 2 * l1_frame);
    const int nb = a.grid.blocks();
    constexpr int kVoteWords = 2 * kMaxBins + kScratch;
    uint32_t *votes0 = keys + nb;
    const int tid = threadIdx.x;
    const bool streamer = tid >= kWsRole;                   // (wave-uniform)
    const int rt = streamer ? tid - kWsRole : tid;          // lane of the role
    constexpr int kStreamLanes = kWsThreads - SEARCH_LANES;
    (void)kStreamLanes;

    if (a.stagger_groups > 1 && blockIdx.x < (uint32_t)a.first_generation) {
        const uint32_t grp = blockIdx.x % (uint32_t)a.stagger_groups;
        if (grp) {
            const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)grp * a.stagger_ticks;
            while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
        }
    }
    for (int k = tid; k < 2 * kVoteWords; k += kWsThreads) votes0[k] = 0;
    lds_barrier();

    // pairs of this workgroup: blockIdx.x + k * gridDim.x, k = 0 .. npw - 1; iteration `it` searches pair it - 1
    // and streams pair it
    const int64_t npw = (a.n_pairs - (int64_t)blockIdx.x + gridDim.x - 1) / gridDim.x;
    // STREAM lane geometry (as in k_coarse, for kWsRole lanes)
    const int chunks = a.w / 16, rpi = a.rows_per_sweep, nkf = NK / 2;
    int yoff = (int)fast_div((uint32_t)rt, a.div_chunks), col = rt - yoff * chunks;
    const bool active = yoff < rpi;
    if (!active) { yoff = 0; col = 0; }
    const uint32_t frame_bytes = (uint32_t)(a.w * a.h);
    const uint32_t voff0 = (uint32_t)(2 * yoff * a.w + col * 16), vstep = (uint32_t)(2 * rpi * a.w);
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#pragma unroll 1
    for (int64_t it = 0; it <= npw; it++) {
        const int64_t pair = (int64_t)blockIdx.x + (it - 1) * (int64_t)gridDim.x;   // searched (it >= 1)
        const int64_t next = (int64_t)blockIdx.x + it * (int64_t)gridDim.x;         // streamed (it < npw)
        const bool searching = it >= 1, streaming = it < npw;
        if (streamer) {
            uint32_t *nsums = votes0 + (int)(it & 1) * kVoteWords + 2 * kMaxBins;    // pixel sums of the pair being streamed
            if (rt < 4) nsums[rt] = 0;                                               // (added to behind the second barrier)
            uint2 park[NK];
            uint32_t sum_p0 = 0, sum_p1 = 0, sum_c0 = 0, sum_c1 = 0;
            u32x4_t q0[kWsDepth], q1[kWsDepth];
            // one buffer resource per frame, the lane's byte offset in ONE register for all sweeps, the sweep's
            // own offset k * vstep in the instruction's scalar offset: no per-sweep vector registers (the
            // sweeps are unrolled, and forty per-lane offsets would be hoisted out of the pair loop and kept)
            const __amdgpu_buffer_rsrc_t rs_prev = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(a.prev) + next * a.pair_stride, 0, frame_bytes, kRawBuffer);
            const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t *>(a.cur) + next * a.pair_stride, 0, frame_bytes, kRawBuffer);
            auto load = [&](int k, int slot) {
                const bool in_cur = k >= nkf;
                const uint32_t soff = (uint32_t)(in_cur ? k - nkf : k) * vstep;     // (scalar; rows past the frame read as zero)
                q0[slot] = __builtin_amdgcn_raw_buffer_load_b128(in_cur ? rs_cur : rs_prev, voff0, soff, kNonTemporal);
                q1[slot] = __builtin_amdgcn_raw_buffer_load_b128(in_cur ? rs_cur : rs_prev, voff0 + (uint32_t)a.w, soff, kNonTemporal);
            };
            if (streaming) {
#pragma unroll
                for (int k = 0; k < kWsDepth; k++) load(k, k);
            }
#pragma unroll
            for (int k = 0; k < NK; k++) {
                if (k == S1 || k == S2) lds_barrier();
                if (streaming) {
                    const u32x4_t r0 = q0[k % kWsDepth], r1 = q1[k % kWsDepth];
                    uint32_t s0 = 0;
                    s0 = byte_sum(r0.x, s0); s0 = byte_sum(r0.y, s0); s0 = byte_sum(r0.z, s0); s0 = byte_sum(r0.w, s0);
                    s0 = byte_sum(r1.x, s0); s0 = byte_sum(r1.y, s0); s0 = byte_sum(r1.z, s0); s0 = byte_sum(r1.w, s0);
                    uint2 o;
                    o.x = box4_cells(r0.x, r1.x, r0.y, r1.y);
                    o.y = box4_cells(r0.z, r1.z, r0.w, r1.w);
                    const uint32_t s1 = byte_sum(o.y, byte_sum(o.x, 0u));
                    // (EXACT: the level-1 rows divide into whole sweeps; otherwise only a frame's LAST sweep can
                    //  reach past its end -- rows that read as zero and are dropped here and in the dump)
                    const bool last_of_frame = (k == nkf - 1 || k == NK - 1);
                    const bool inside = EXACT || !last_of_frame || yoff + (nkf - 1) * rpi < h1;
                    if (inside) {
                        if (k < nkf) { sum_p0 += s0; sum_p1 += s1; } else { sum_c0 += s0; sum_c1 += s1; }
                    }
                    park[k] = o;
                    if (k + kWsDepth < NK) load(k + kWsDepth, k % kWsDepth);
                }
            }
            // (behind the second barrier: nobody reads the level-1 frames of the searched pair any more)
            if (streaming) {
                // (the lane's LDS offset is made opaque per pair: forty loop-invariant store addresses would
                //  otherwise be hoisted out of the pair loop and held in registers beside the parked pixels)
                uint32_t lane_off = (uint32_t)(yoff * w1 + col * 8);
                asm volatile("" : "+v"(lane_off));
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t sweep_off = (uint32_t)(((k >= nkf ? h1 : 0) + (k >= nkf ? k - nkf : k) * rpi) * w1);   // (scalar)
                    const bool last_of_frame = (k == nkf - 1 || k == NK - 1);
                    const bool inside = EXACT || !last_of_frame || yoff + (nkf - 1) * rpi < h1;
                    if (active && inside) *reinterpret_cast<uint2 *>(lds + lane_off + sweep_off) = park[k];
                }
                if (!active) sum_p0 = sum_p1 = sum_c0 = sum_c1 = 0;   // (idle lanes loaded valid bytes: dropped)
                sum_p0 = wave_sum_u32(sum_p0); sum_p1 = wave_sum_u32(sum_p1);
                sum_c0 = wave_sum_u32(sum_c0); sum_c1 = wave_sum_u32(sum_c1);
                if ((rt & 63) == 0) {
                    atomicAdd(&nsums[0], sum_p0); atomicAdd(&nsums[1], sum_p1);
                    atomicAdd(&nsums[2], sum_c0); atomicAdd(&nsums[3], sum_c1);
                }
            }
            lds_barrier();
            continue;
        }

        // ---- SEARCH role: phases 2 .. 4 of `pair`, whose level-1 frames and pixel sums the STREAM role left in LDS ----
        uint32_t *hist = votes0 + (int)((it - 1) & 1) * kVoteWords;   // [2][kMaxBins]
        uint32_t *sums = hist + 2 * kMaxBins;                         // [4] pixel sums, [3] vote sums
        if (searching) {
            if (rt == 0 && a.sums) *reinterpret_cast<uint4 *>(a.sums + pair * 4) = make_uint4(sums[0], sums[1], sums[2], sums[3]);
            int delta = 0;
            if (a.sums) {
                const uint32_t npix = (uint32_t)l1_frame;
                delta = (int)((sums[1] + npix / 2) / npix) - (int)((sums[3] + npix / 2) / npix);
            }
            if (delta != 0) {
                uint32_t *c32 = reinterpret_cast<uint32_t *>(l1[1]);
                for (int k = rt; k < l1_frame / 4; k += kWsRole) c32[k] = sat_add_u8x4(c32[k], delta);
            }
            // this pair's histograms and vote sums: last used two pairs ago, finalised long since
            for (int k = rt; k < 2 * kMaxBins; k += kWsRole) hist[k] = 0;
            if (rt < 3) sums[4 + rt] = 0;
            const int x0 = a.grid.x0, y0 = a.grid.y0;
            for (int blk = rt; blk < nb; blk += kWsRole) {
                const int by = (int)fast_div((uint32_t)blk, a.div_nx), bx = blk - by * a.grid.nx;
                const uint8_t *t = l1[0] + (y0 + 8 * by + 2) * w1 + x0 + 8 * bx;   // tile rows 2..5
                uint32_t mid[4], diff = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(t + r * w1);
                    mid[r] = __builtin_amdgcn_alignbyte(q[1], q[0], 2);
                }
#pragma unroll
                for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
                for (int r = 0; r < 4; r++)
                    diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
                keys[blk] = diff >= (uint32_t)a.feature_threshold ? kOpen : kGated;
            }
        }
        lds_barrier();
        if (searching) {
            const int x0 = a.grid.x0, y0 = a.grid.y0;
            const int items = 9 * nb;
            for (int item = rt; item < items; item += kWsRole) {
                const int d = (int)fast_div((uint32_t)item, a.div_nb), blk = item - __mul24(d, nb);
                if (keys[blk] == kGated) continue;
                const int by = (int)fast_div((uint32_t)blk, a.div_nx), bx = blk - __mul24(by, a.grid.nx);
                const uint8_t *ref = l1[0] + __mul24(y0 + 8 * by, w1) + x0 + 8 * bx;
                const uint8_t *win = l1[1] + __mul24(y0 + 8 * by - 4 + d, w1) + (x0 - 4) + 8 * bx;   // 8-byte aligned
                u64 lo = 0, hi = 0;
                uint32_t a8 = (uint32_t)(d * 9 + 8);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(ref + r * w1);
                    const uint32_t r0 = q[0], r1 = q[1];
                    const uint2 wa = *reinterpret_cast<const uint2 *>(win + r * w1);
                    const uint2 wb = *reinterpret_cast<const uint2 *>(win + r * w1 + 8);
                    const u64 p01 = pack64(wa.x, wa.y), p12 = pack64(wa.y, wb.x), p23 = pack64(wb.x, wb.y);
                    lo = qsad(p01, r0, lo);
                    lo = qsad(p12, r1, lo);
                    hi = qsad(p12, r0, hi);
                    hi = qsad(p23, r1, hi);
                    a8 = __builtin_amdgcn_sad_hi_u8(wb.x, r0, a8);
                    a8 = __builtin_amdgcn_sad_hi_u8(wb.y, r1, a8);
                }
                const uint32_t base = (uint32_t)(d * 9);
                const uint32_t l0 = (uint32_t)lo, l1w = (uint32_t)(lo >> 32), h0 = (uint32_t)hi, h1w = (uint32_t)(hi >> 32);
                const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
                const uint32_t k2 = (l1w << 16) | (base + 2), k3 = (l1w & 0xFFFF0000u) | (base + 3);
                const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
                const uint32_t k6 = (h1w << 16) | (base + 6), k7 = (h1w & 0xFFFF0000u) | (base + 7);
                uint32_t best = min(min(k0, k1), k2);
                best = min(best, min(min(k3, k4), k5));
                best = min(best, min(min(k6, k7), a8));
                atomicMin(&keys[blk], best);
            }
        }
        lds_barrier();
        int *vs = reinterpret_cast<int *>(sums + 4);
        if (searching) {
            const int centre = 2 * a.tail.range + 1;
            uint32_t *out = reinterpret_cast<uint32_t *>(a.blocks) + pair * nb;
            int s2x = 0, s2y = 0, cnt = 0;
            const int rounds = (nb + kWsRole - 1) / kWsRole;   // uniform trip count: the ballots need every lane
            for (int rd = 0; rd < rounds; rd++) {
                const int blk = rd * kWsRole + rt;
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
            if ((rt & 63) == 0) {
                atomicAdd(&vs[0], s2x);
                atomicAdd(&vs[1], s2y);
                atomicAdd(&vs[2], cnt);
            }
        }
        lds_barrier();
        if (searching && rt < 64) finalise_flow_wave(a.tail, pair, hist, hist + kMaxBins, vs);
        // (no barrier: the next pair votes into the other set; this one is cleared two pairs from now)
    }
}


// Launcher hook (called from launch_coarse_fused in a lab build): true = launched (or failed) here.
static bool lab_launch_coarse_ws(CoarseArgs &k, int64_t wgs, size_t lds, void *stream, int *rc)
{
    constexpr int kWsSearchLanes = AOF_WS_SEARCH, kWsSweeps = AOF_WS_NK;
    const int ws_rpi = (kWsThreads - kWsSearchLanes) / (k.w / 16), ws_nkf = ws_rpi > 0 ? ((k.h / 2) + ws_rpi - 1) / ws_rpi : 0;
    if (k.n_pairs >= 2 * wgs && ws_rpi >= 1 && 2 * ws_nkf == kWsSweeps) {
        k.rows_per_sweep = ws_rpi;
        const bool exact = ws_nkf * ws_rpi == k.h / 2;
        void (*fn)(CoarseArgs) = exact ? k_coarse_ws<kWsSearchLanes, kWsSweeps, AOF_WS_S1, AOF_WS_S2, AOF_WS_DEPTH, true>
                                       : k_coarse_ws<kWsSearchLanes, kWsSweeps, AOF_WS_S1, AOF_WS_S2, AOF_WS_DEPTH, false>;
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { *rc = (int)e; return true; }
        hipLaunchKernelGGL(fn, dim3((uint32_t)wgs), dim3(kWsThreads), lds, static_cast<hipStream_t>(stream), k);
        *rc = (int)hipGetLastError();
        return true;
    }
    return false;
}
