// K2 (tile8) -- 8x8 SAD search over +-4 px on a dense grid from LDS-staged strips
// (DESIGN.md "Spec": Search; "Kernels": K2).  This round's first dominant kernel; since
// k_search_lane8 (no LDS, rows straight from L2) measured 8-16 % faster on the same
// configurations it serves the comparison modes AOF_SEARCH_EXHAUSTIVE_STRIPS and
// AOF_SEARCH_PRUNED_STRIPS only.
//
// Mapping.  A workgroup owns a strip of `rb` block rows of one frame pair and
// stages the strip's pixels ONCE into LDS as two flat, fully coalesced 16-byte
// copies (rows are contiguous in HBM because stride == width), issued as LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip;
// __syncthreads() drains it with s_waitcnt vmcnt(0) before the barrier):
//     cur  rows [8*by0 + py, +8*rows+8)   -> smem[0 ..)
//     prev rows [8*by0 + 4,  +8*rows)     -> smem[cur_bytes ..)
// Each LANE then owns one whole block: it keeps the 8x8 reference tile in 16
// VGPRs, streams the 16 search rows out of LDS and evaluates all 81 candidates
// with v_qsad_pk_u16_u8 -- four horizontally sliding 4-byte SADs per
// instruction, packed u16 accumulators (max 64*255 = 16320 fits) -- plus
// v_sad_hi_u8 for the ninth column, which accumulates straight into the high
// half of a register pre-loaded with the candidate index.  No cross-lane
// traffic at all: the arg-min is a per-lane v_min3_u32 tree over the packed
// keys (sad << 16 | idx), i.e. "first minimum in scan order wins".
//
// With the dense grid origin at S = 4 and step 8 every search window starts on
// an 8-byte boundary, so LDS is read with ds_read_b64 and needs no byte
// realignment.  The SHIFTED variant (level 0 of the 2-level pyramid, window
// displaced by the per-pair predictor (px, py)) keeps that property by staging
// the cur rows PRE-SHIFTED: the flat copy simply starts (px & 7) bytes later in
// HBM, so LDS column c holds frame column c + (px & 7) and a block's window sits
// at the 8-aligned LDS column 8*(bx + (px >> 3)).  The search loop is identical.
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

constexpr int kMaxThreads = 512;

#ifdef AOF_LAB  // experiment knobs of tools/k2_lab.hip; never defined in the product build
int g_lab_stagger = -1;     // first-generation stagger units (-1 = product default)
__constant__ int c_lab_mode;  // 1: skip staging loads, 2: skip the search
__constant__ int c_lab_count;  // 1: count rows in the pruned search (slow)
__device__ unsigned long long d_lab_rows[2];  // pruned search: rows visited / rows fully evaluated
#define LAB_MODE c_lab_mode
#else
#define LAB_MODE 0
#endif

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

// DYG = vertical candidate offsets (dy) evaluated per lane: 9 = the whole block in one
// lane; 3 = three lanes per block, each doing three of the nine dy rows (less state per
// lane, smaller strips, more resident waves); the partial minima of a block then meet in
// an LDS atomicMin on the packed key -- integer min, so still exactly first-minimum-wins.
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, x),
                                                                  __builtin_bit_cast(ushort2_t, y)));
}

// PRUNE = exact partial-distortion elimination (include/aof.h: AOF_SEARCH_PRUNED).
template <int DYG, bool SHIFTED, bool PRUNE>
__global__ __launch_bounds__(kMaxThreads) void k_search_tile8(SearchArgs a, int rb, int nstrips, int spw,
                                                              uint32_t total_wgs, uint32_t first_gen,
                                                              uint32_t stagger_units)
{
    constexpr int NG = 9 / DYG;  // lanes per block
    // De-phase the workgroups that share a CU.  Every workgroup costs the same, so the
    // first generation (all slots filled at launch) would stage, search and retire in
    // lock-step for the whole kernel, and the HBM latency of staging would never hide
    // behind another workgroup's search (measured: +13 % kernel time).  A one-off
    // pseudo-random delay of the first generation spreads the phases; later workgroups
    // inherit the spread because they start whenever a slot frees.  Speed only.
    if (blockIdx.x < first_gen && stagger_units) {
        const uint32_t h = (blockIdx.x * 2654435761u) >> 16;
        const uint32_t n = h % stagger_units;  // x ~1000 cycles
        for (uint32_t i = 0; i < n; i++) __builtin_amdgcn_s_sleep(16);
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // A workgroup walks `spw` consecutive strips of ONE pair (spw = 1 for the exhaustive search).
    // The pruned search uses that to carry, per wave and in a register, the dy row where the
    // previous strip's blocks matched: under a global translation the next strip then meets
    // its row first and prunes the other eight.
    const uint32_t logical = xcd_remap(blockIdx.x, total_wgs);
    const int wg_per_pair = (nstrips + spw - 1) / spw;
    const int first_strip = (int)(logical % (uint32_t)wg_per_pair) * spw;
    const int64_t pair = (int64_t)(logical / (uint32_t)wg_per_pair);
    int start_row = 4;                                // dy index visited first (wave-uniform)
    int prune_pays = 1;                               // previous strip of this wave dropped rows
    for (int strip = first_strip; strip < min(nstrips, first_strip + spw); strip++) {
    // With half-pixel refinement the grid origin is S+1 = 5 and every search window carries a
    // one-pixel ring.  The strip is then staged one byte later per row (LDS column c = frame
    // column c + 1, so windows stay 8-byte aligned) with one extra row above and below: window
    // rows start at LDS row 8*brow + 1, the ring at 8*brow.  The ring column left of the first
    // block is the last byte of the previous LDS row -- the flat copy puts the right pixel there
    // -- except for the first staged row, whose lead byte is fetched separately.
    const int org = a.grid.x0 - 4;                    // 0, or 1 with the half-pixel margin
    const int W = a.w, H = a.h, Wb = a.w - 2 * org, nx = a.grid.nx, ny = a.grid.ny;
    const int by0 = strip * rb;
    const int rows = min(rb, ny - by0);
    const int tid = threadIdx.x, nthreads = blockDim.x;

    int px = 0, py = 0;
    if (SHIFTED) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }

    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(a.w * a.h));

    // ---- stage the strip into LDS (flat 16-byte copies) ----
    const int n_cur_rows = 8 * rows + 8 + 2 * org;
    const int yc0 = 8 * by0 + py;                 // frame row of LDS cur row 0
    const int r_lo = max(0, -yc0);                // valid LDS cur rows [r_lo, r_hi)
    const int r_hi = min(n_cur_rows, H - yc0);
    constexpr uint32_t kLead = 16;                // room for the lead byte in front of the cur rows
    const uint32_t prev_off = kLead + (uint32_t)((8 * rb + 8 + 2 * org) * W);
    uint8_t *s_cur = smem + kLead;
    uint8_t *s_prev = smem + prev_off;
    uint32_t *s_best = reinterpret_cast<uint32_t *>(smem + prev_off + (uint32_t)(8 * rb * W) + 16);
    uint32_t *s_hist = s_best + rb * nx;            // [2][bins] votes of this strip
    const int centre = 2 * a.hist_range + 1, bins = 2 * centre + 1;
    const int sh7 = SHIFTED ? (px & 7) : 0;        // floor-mod: px = 8*(px >> 3) + sh7
    const uint8_t *g_cur = a.cur + pair * a.pair_stride + org + (int64_t)(yc0 + r_lo) * W + sh7;
    const uint8_t *g_prev = a.prev + pair * a.pair_stride + (int64_t)org * (W + 1) + (int64_t)(8 * by0 + 4) * W;
    int cur_chunks = r_hi > r_lo ? (r_hi - r_lo) * (W / 16) : 0;
    const int prev_chunks = 8 * rows * (W / 16);
    // The displaced copy runs sh7 + org bytes past its last row; when that row is the last
    // row of the frame the final chunk is copied bytewise with a bounds check instead.
    const int over = sh7 + org;
    const bool tail_guard = over != 0 && cur_chunks > 0 && yc0 + r_hi == H;
    if (tail_guard) cur_chunks -= 1;
    if (LAB_MODE == 1) {
        // lab: no global traffic
    } else if (!SHIFTED && delta == 0 && org == 0) {
        uint8_t *dst = s_cur + r_lo * W;
        // LDS-DMA: each wave-instruction moves 1 KiB global -> LDS without touching VGPRs
        // (destination = wave-uniform base + lane*16, so the flat copy maps 1:1)
        // address = wave-uniform base (SGPRs) + zero-extended 32-bit lane offset (one VGPR)
        const uint32_t wbase = (uint32_t)(tid & ~63) * 16u, lane16 = (uint32_t)(tid & 63) * 16u;
        const uint32_t total = (uint32_t)cur_chunks * 16u, step = (uint32_t)nthreads * 16u;
        uint32_t o = wbase;
        for (; o + 1024u <= total; o += step)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(g_cur + (size_t)(o + lane16)),
                (__attribute__((address_space(3))) void *)(dst + o), 16, 0, 0);
        if (o < total && o + lane16 < total)  // ragged last KiB: through VGPRs
            *reinterpret_cast<uint4 *>(dst + o + lane16) =
                *reinterpret_cast<const uint4 *>(g_cur + (size_t)(o + lane16));
    } else {
        for (int c = tid; c < cur_chunks; c += nthreads) {
            uint4 v;  // 16-B load from a byte-aligned address (gfx950 handles unaligned global loads)
            __builtin_memcpy(&v, g_cur + c * 16, 16);
            if (delta != 0) {
                v = sat_add_u8x16(v, delta);
            }
            *reinterpret_cast<uint4 *>(s_cur + r_lo * W + c * 16) = v;
        }
        if (tail_guard && tid < 16 - over)
            s_cur[r_lo * W + cur_chunks * 16 + tid] =
                (uint8_t)clamp_u8((int)g_cur[cur_chunks * 16 + tid] + delta);
        // lead byte: ring column left of block column 0 in the first staged row
        if (org != 0 && cur_chunks > 0 && tid == 0)
            (s_cur + r_lo * W)[-1] = (uint8_t)clamp_u8((int)g_cur[-1] + delta);
    }
    if (LAB_MODE != 1 && org != 0) {  // moved origin: byte-aligned source, through registers
        for (int c = tid; c < prev_chunks; c += nthreads) {
            uint4 v;
            __builtin_memcpy(&v, g_prev + c * 16, 16);
            *reinterpret_cast<uint4 *>(s_prev + c * 16) = v;
        }
    } else if (LAB_MODE != 1) {
        const uint32_t wbase = (uint32_t)(tid & ~63) * 16u, lane16 = (uint32_t)(tid & 63) * 16u;
        const uint32_t total = (uint32_t)prev_chunks * 16u, step = (uint32_t)nthreads * 16u;
        uint32_t o = wbase;
        for (; o + 1024u <= total; o += step)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(g_prev + (size_t)(o + lane16)),
                (__attribute__((address_space(3))) void *)(s_prev + o), 16, 0, 0);
        if (o < total && o + lane16 < total)
            *reinterpret_cast<uint4 *>(s_prev + o + lane16) =
                *reinterpret_cast<const uint4 *>(g_prev + (size_t)(o + lane16));
    }
    if (NG > 1)
        for (int b = tid; b < rows * nx; b += nthreads) s_best[b] = 0xFFFFFFFFu;
    if (a.hist_parts)
        for (int k = tid; k < 2 * bins; k += nthreads) s_hist[k] = 0;
    __syncthreads();
    if (LAB_MODE == 2) {
        if (tid < rows * nx) {
            aof_block z; z.dx = (int8_t)s_cur[tid]; z.dy = (int8_t)s_prev[tid]; z.sad = 0;
            a.blocks[pair * (int64_t)(nx * ny) + (int64_t)by0 * nx + tid] = z;
        }
        return;
    }

    // ---- one (block, dy group) per lane; group-major so that group 0 = lanes [0, rows*nx) ----
    const int nblk = rows * nx;
    const bool live = tid < nblk * NG;
    const int grp = (NG > 1 && live) ? tid / nblk : 0, blk = live ? tid - grp * nblk : 0;
    const int brow = blk / nx, bx = blk % nx;
    constexpr int kRows = DYG + 7;                // search rows this lane streams
    const int s0 = grp * DYG;                     // first dy index of the group
    bool inside = true;
    int xs = 8 * bx;                              // LDS byte column of the window start
    if (SHIFTED) {
        const int xf = 8 * bx + px;               // frame column of the window start
        inside = xf >= 0 && xf + 16 <= Wb && 8 * brow >= r_lo && 8 * brow + 16 + 2 * org <= r_hi;
        xs = inside ? xf - sh7 : 0;               // 8-aligned; ignored reads stay in range
    }

    uint32_t diff = 0, best = 0xFFFFFFFFu;
    uint32_t ref[8][2];  // reference tile: 8 rows x 2 dwords (also feeds the half-pixel refinement)
    // whole waves beyond the strip's blocks skip the search (wave-uniform) but still
    // reach the barriers and the vote below
    if ((tid & ~63) < nblk * NG) {
    // reference tile at frame column 8*bx + 4 (+ org)
    // LDS offsets as 32-bit integers from the one shared array (no 64-bit pointer maths)
    const uint32_t ref_off = prev_off + (uint32_t)(8 * brow * W + 8 * bx + 4);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(smem + (ref_off + (uint32_t)(r * W)));
        ref[r][0] = p[0];
        ref[r][1] = p[1];
    }

    // 4x4 gradient gate on tile bytes [2..5] x rows [2..5]
    {
        uint32_t mid[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            mid[r] = __builtin_amdgcn_alignbyte(ref[r + 2][1], ref[r + 2][0], 2);  // bytes 2..5
#pragma unroll
        for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // bytes (3,4,5,5) against (2,3,4,5): the doubled last byte adds |p5-p5| = 0
            const uint32_t sh = __builtin_amdgcn_perm(0u, mid[r], 0x03030201u);
            diff = __builtin_amdgcn_sad_u8(mid[r], sh, diff);
        }
    }

    bool exhaustive = !PRUNE;  // pruned mode may still choose it per wave and strip (below)
    if constexpr (PRUNE) {
        static_assert(!PRUNE || DYG == 9, "the pruned search keeps one block per lane");
        // Exact pruning.  A partial SAD only grows, so once a row's partial sums all exceed a
        // lane's best SAD the row cannot win (nor tie) for that lane; the wave drops the row
        // when that holds for every lane that still needs a result.  Rows are visited outwards
        // from dy = 0 (the predictor's row in the 2-level search), so small motions meet their
        // row early and the remaining rows cost two row pairs each.  (A per-pair hint word
        // shared between workgroups was tried: its agent-scope load/store cost more than the
        // better order saved, and workgroups of one pair run concurrently anyway.)
        const bool need = live && inside && diff >= (uint32_t)a.feature_threshold;
        const int start = __builtin_amdgcn_readfirstlane(start_row);
        const uint32_t win_base = kLead + (uint32_t)((8 * brow + org) * W + xs);
        // Only the two rows of the first test (r = 0, 4) are fetched ahead, one dy row early
        // (ping-pong registers), so pruned rows cost two LDS reads; the other six rows are read
        // only when the row survives.  (Fetching all eight ahead made the kernel LDS-bound:
        // 72 reads per block against 16 in the exhaustive kernel.)  Two accumulator sets (even /
        // odd row pairs) keep consecutive v_qsad independent; they are added as packed u16.
        uint2 head_a[2][2], head_b[2][2];
        int rows_dropped = 0;
        auto fetch = [&](uint2 (&buf)[2][2], int d) {
            const uint32_t row_off = win_base + (uint32_t)(d * W);
            const uint2 *p0 = reinterpret_cast<const uint2 *>(smem + row_off);
            const uint2 *p4 = reinterpret_cast<const uint2 *>(smem + (row_off + (uint32_t)(4 * W)));
            buf[0][0] = p0[0]; buf[0][1] = p0[1];
            buf[1][0] = p4[0]; buf[1][1] = p4[1];
        };
        auto evaluate = [&](const uint2 (&head)[2][2], int d) {
            u64 alo[2] = {0, 0}, ahi[2] = {0, 0};
            uint32_t a8[2] = {(uint32_t)(d * 9 + 8), 0u};
            auto row_pair = [&](int r, int set, uint2 a0, uint2 a1) {
                const u64 p01 = pack64(a0.x, a0.y), p12 = pack64(a0.y, a1.x), p23 = pack64(a1.x, a1.y);
                alo[set] = qsad(p01, ref[r][0], alo[set]);
                ahi[set] = qsad(p12, ref[r][0], ahi[set]);
                alo[set] = qsad(p12, ref[r][1], alo[set]);
                ahi[set] = qsad(p23, ref[r][1], ahi[set]);
                a8[set] = __builtin_amdgcn_sad_hi_u8(a1.x, ref[r][0], a8[set]);
                a8[set] = __builtin_amdgcn_sad_hi_u8(a1.y, ref[r][1], a8[set]);
            };
            row_pair(0, 0, head[0][0], head[0][1]);
            row_pair(4, 1, head[1][0], head[1][1]);
            // partial sums of two row pairs: 16 pixels <= 4080 per field, no u16 carry when added
            const uint32_t s0 = (uint32_t)alo[0] + (uint32_t)alo[1], s1 = (uint32_t)(alo[0] >> 32) + (uint32_t)(alo[1] >> 32);
            const uint32_t s2 = (uint32_t)ahi[0] + (uint32_t)ahi[1], s3 = (uint32_t)(ahi[0] >> 32) + (uint32_t)(ahi[1] >> 32);
            const uint32_t m = pk_min_u16(pk_min_u16(s0, s1), pk_min_u16(s2, s3));
            const uint32_t pmin = min(min(m & 0xFFFFu, m >> 16), (a8[0] + a8[1]) >> 16);
#ifdef AOF_LAB
            if (c_lab_count && (tid & 63) == 0) atomicAdd(&d_lab_rows[0], 1ull);
#endif
            if (__ballot(need && pmin <= (best >> 16)) == 0) { rows_dropped++; return; }  // nobody can improve
#ifdef AOF_LAB
            if (c_lab_count && (tid & 63) == 0) atomicAdd(&d_lab_rows[1], 1ull);
#endif
            const uint32_t row_off = win_base + (uint32_t)(d * W);
            // the other six rows are requested together (one LDS round trip) ...
            uint2 rest[6][2];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int r = i < 2 ? 2 + 4 * i : 2 * (i - 2) + 1;  // 2, 6, then 1, 3, 5, 7
                const uint2 *p = reinterpret_cast<const uint2 *>(smem + (row_off + (uint32_t)(r * W)));
                rest[i][0] = p[0];
                rest[i][1] = p[1];
            }
            // ... and a second, tighter test follows on half of the tile (row pairs 0, 2, 4, 6):
            // under moderate noise 16 pixels are too few to separate the candidates, 32 usually are
            row_pair(2, 0, rest[0][0], rest[0][1]);
            row_pair(6, 1, rest[1][0], rest[1][1]);
            {
                const uint32_t t0 = (uint32_t)alo[0] + (uint32_t)alo[1], t1 = (uint32_t)(alo[0] >> 32) + (uint32_t)(alo[1] >> 32);
                const uint32_t t2 = (uint32_t)ahi[0] + (uint32_t)ahi[1], t3 = (uint32_t)(ahi[0] >> 32) + (uint32_t)(ahi[1] >> 32);
                const uint32_t mm = pk_min_u16(pk_min_u16(t0, t1), pk_min_u16(t2, t3));
                const uint32_t pmin2 = min(min(mm & 0xFFFFu, mm >> 16), (a8[0] + a8[1]) >> 16);
                if (__ballot(need && pmin2 <= (best >> 16)) == 0) { rows_dropped++; return; }
            }
#pragma unroll
            for (int i = 2; i < 6; i++) row_pair(2 * (i - 2) + 1, i & 1, rest[i][0], rest[i][1]);
            // whole-row sums: 64 pixels <= 16320 per field, still no carry between the u16 fields
            const uint32_t l0 = (uint32_t)alo[0] + (uint32_t)alo[1], l1 = (uint32_t)(alo[0] >> 32) + (uint32_t)(alo[1] >> 32);
            const uint32_t h0 = (uint32_t)ahi[0] + (uint32_t)ahi[1], h1 = (uint32_t)(ahi[0] >> 32) + (uint32_t)(ahi[1] >> 32);
            const uint32_t base = (uint32_t)(d * 9);
            const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
            const uint32_t k2 = (l1 << 16) | (base + 2), k3 = (l1 & 0xFFFF0000u) | (base + 3);
            const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
            const uint32_t k6 = (h1 << 16) | (base + 6), k7 = (h1 & 0xFFFF0000u) | (base + 7);
            best = min(best, min(min(k0, k1), k2));
            best = min(best, min(min(k3, k4), k5));
            best = min(best, min(min(k6, k7), a8[0] + a8[1]));
        };
        // visiting order: start, then alternately below / above it, whichever is still in range
        // (start, start-1, start+1, start-2, ...): the k-th row in closed form, k = 0..8
        auto order = [start](int k) -> int {
            if (k == 0) return start;
            const int below = start, above = 8 - start;      // rows available on either side
            const int pairs = below < above ? below : above;  // alternating part: 2*pairs rows
            if (k <= 2 * pairs) return (k & 1) ? start - (k + 1) / 2 : start + k / 2;
            const int rest = k - 2 * pairs;                   // one-sided tail
            return below > above ? start - pairs - rest : start + pairs + rest;
        };
        // When the previous strip of this wave could drop (almost) nothing -- noise-dominated
        // images -- the exhaustive code is the faster way to evaluate everything: use it for
        // the remaining strips of this workgroup.
        const bool fall_back = __builtin_amdgcn_readfirstlane(prune_pays) == 0;
        if (fall_back) {
            exhaustive = true;  // and stays so for the rest of this workgroup's strips
        } else if (__ballot(need) != 0) {
            rows_dropped = 0;
            fetch(head_a, order(0));
            for (int k = 0; k < 9; k += 2) {
                if (k + 1 < 9) fetch(head_b, order(k + 1));
                evaluate(head_a, order(k));
                if (k + 1 >= 9) break;
                if (k + 2 < 9) fetch(head_a, order(k + 2));
                evaluate(head_b, order(k + 1));
            }
            // next strip of this workgroup starts where this wave's first live block matched
            const unsigned long long needing = __ballot(need);
            const int src = __ffsll((long long)needing) - 1;
            start_row = (int)((uint32_t)__shfl((int)best, src, 64) & 0xFFFFu) / 9;
            prune_pays = rows_dropped >= 2;
        }
    }
    // The data-independent search: all nine dy rows, all eight row pairs (the whole kernel in
    // exhaustive mode; the per-wave fallback of the pruned mode when nothing can be pruned).
    if (exhaustive) {
    // accumulators: per dy, offsets 0..3 / 4..7 packed u16, offset 8 as (sad<<16 | idx)
    u64 acc_lo[DYG], acc_hi[DYG];
    uint32_t acc_8[DYG];
#pragma unroll
    for (int d = 0; d < DYG; d++) { acc_lo[d] = 0; acc_hi[d] = 0; acc_8[d] = (uint32_t)((s0 + d) * 9 + 8); }

    const uint32_t win_off = kLead + (uint32_t)((8 * brow + org + s0) * W + xs);
#pragma unroll
    for (int s = 0; s < kRows; s++) {
        const uint2 *p = reinterpret_cast<const uint2 *>(smem + (win_off + (uint32_t)(s * W)));
        const uint2 a0 = p[0], a1 = p[1];
        const uint32_t w0 = a0.x, w1 = a0.y, w2 = a1.x, w3 = a1.y;
        const u64 p01 = pack64(w0, w1), p12 = pack64(w1, w2), p23 = pack64(w2, w3);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int d = s - r;  // dy index within the group (dy = s0 + d - 4)
            if (d < 0 || d >= DYG) continue;
            acc_lo[d] = qsad(p01, ref[r][0], acc_lo[d]);
            acc_lo[d] = qsad(p12, ref[r][1], acc_lo[d]);
            acc_hi[d] = qsad(p12, ref[r][0], acc_hi[d]);
            acc_hi[d] = qsad(p23, ref[r][1], acc_hi[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w2, ref[r][0], acc_8[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w3, ref[r][1], acc_8[d]);
        }
    }

    // arg-min over the 81 packed keys, scan order = key order
#pragma unroll
    for (int d = 0; d < DYG; d++) {
        const uint32_t base = (uint32_t)((s0 + d) * 9);
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
    }  // exhaustive
    }  // wave has live lanes

    if (NG > 1) {
        if (live) atomicMin(&s_best[blk], best);
        __syncthreads();
        best = s_best[blk];
    }
    const bool writer = live && grp == 0;        // one lane per block
    aof_block rec;
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    if (inside && diff >= (uint32_t)a.feature_threshold) {
        const int idx = (int)(best & 0xFFFFu);
        rec.dx = (int8_t)(px + idx % 9 - 4);
        rec.dy = (int8_t)(py + idx / 9 - 4);
        rec.sad = (uint16_t)(best >> 16);
    }
    // the strip's records are contiguous: (by0 + brow)*nx + bx == by0*nx + blk; keep the
    // 64-bit part of the address wave-uniform (SALU) and the lane part 32-bit
    aof_block *strip_out = a.blocks + (pair * (int64_t)(nx * ny) + (int64_t)by0 * nx);
    // one dword store per record (aof_block alone is only 2-byte aligned; tile8_supported()
    // checks the array)
    if (writer) reinterpret_cast<uint32_t *>(strip_out)[(uint32_t)blk] = __builtin_bit_cast(uint32_t, rec);
    const bool ok = writer && rec.sad != AOF_SAD_SKIPPED && (int)rec.sad < a.value_threshold;

    // Half-pixel refinement of accepted blocks, from the ring around the best match that the
    // strip already holds in LDS (equalised): aligned dwords + v_alignbyte by the lane's phase.
    int hx = 0, hy = 0;
    if (a.subpixel) {  // uniform
        int subdir = 8;
        if (ok) {
            const int idx = (int)(best & 0xFFFFu);
            // ring top-left: LDS row 8*brow + dy index, column xs + dx index - 1
            const int at = (8 * brow + idx / 9) * W + xs + idx % 9 - 1;
            const uint32_t sh = (uint32_t)(at & 3);
            const uint32_t base = kLead + (uint32_t)(at & ~3);   // at >= -1: the lead pad covers it
            RefineState<2> st;
            st.init();
            for_rows<-1, 8>([&](auto yc) {
                constexpr int Y = decltype(yc)::value;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + (base + (uint32_t)((Y + 1) * W)));
                uint32_t d[3];
                d[0] = __builtin_amdgcn_alignbyte(q[1], q[0], sh);
                d[1] = __builtin_amdgcn_alignbyte(q[2], q[1], sh);
                d[2] = __builtin_amdgcn_alignbyte(q[3], q[2], sh);
                st.template row<Y>(d, ref);
            });
            subdir = st.direction(rec.sad);
            hx = (subdir == 0 || subdir == 1 || subdir == 7) ? 1 : ((subdir == 3 || subdir == 4 || subdir == 5) ? -1 : 0);
            hy = (subdir == 1 || subdir == 2 || subdir == 3) ? 1 : ((subdir == 5 || subdir == 6 || subdir == 7) ? -1 : 0);
        }
        if (writer && a.subdirs)
            a.subdirs[pair * (int64_t)(nx * ny) + (int64_t)by0 * nx + blk] = (uint8_t)subdir;
    }

    // Votes of this strip's accepted blocks, so that K3 sums nstrips small histograms per
    // pair instead of re-reading every record (DESIGN.md "Kernels": K3).
    if (a.hist_parts) {
        wave_vote(s_hist, 2 * rec.dx + hx + centre, ok);
        wave_vote(s_hist + bins, 2 * rec.dy + hy + centre, ok);
        __syncthreads();
        uint32_t *out = a.hist_parts + ((size_t)pair * nstrips + strip) * (size_t)(2 * bins);
        for (int k = tid; k < 2 * bins; k += nthreads) out[k] = s_hist[k];
    }
    if (spw > 1) __syncthreads();  // LDS (tiles, votes) is rewritten by the next strip
    }  // strips of this workgroup
}

}  // namespace

bool tile8_supported(const SearchArgs &a)
{
    if (a.tile != 8 || a.search != 4) return false;
    // dense grid at origin S (integer search) or S+1 (a half-pixel refinement pass follows)
    const int org = a.subpixel ? 1 : 0;
    if (a.grid.x0 != 4 + org || a.grid.y0 != 4 + org || a.grid.step_x != 8 || a.grid.step_y != 8) return false;
    if (a.w % 16 || a.pair_stride % 16) return false;
    if (reinterpret_cast<uintptr_t>(a.prev) % 16 || reinterpret_cast<uintptr_t>(a.cur) % 16) return false;
    if (reinterpret_cast<uintptr_t>(a.blocks) % 4) return false;
    if ((int64_t)a.w * a.h > 0x7FFFFFFF) return false;
    return plan_tile8(a.w, a.grid.nx, a.grid.ny).rb > 0;
}

int launch_search_tile8(const SearchArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    const Tile8Plan p = plan_tile8(a.w, a.grid.nx, a.grid.ny);
    if (p.rb == 0) return (int)hipErrorInvalidValue;
    // pruned search: consecutive strips per workgroup, so all but the first inherit a start row;
    // fewer on short frames, which need the workgroups for parallelism
    int spw = 1;
    if (a.prune && p.dyg == 9) spw = p.nstrips >= 12 ? 4 : (p.nstrips >= 8 ? 2 : 1);
    const int64_t total = a.n_pairs * ((p.nstrips + spw - 1) / spw);
    if (total > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    hipStream_t s = static_cast<hipStream_t>(stream);
    void (*fn)(SearchArgs, int, int, int, uint32_t, uint32_t, uint32_t);
    if (p.dyg == 9 && a.prune) fn = a.pred ? k_search_tile8<9, true, true> : k_search_tile8<9, false, true>;
    else if (p.dyg == 9) fn = a.pred ? k_search_tile8<9, true, false> : k_search_tile8<9, false, false>;
    else fn = a.pred ? k_search_tile8<3, true, false> : k_search_tile8<3, false, false>;
    if (p.lds > 64 * 1024) {  // beyond the default dynamic-LDS window
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return (int)e;
    }
    // first generation = workgroups resident at launch (LDS- or register-limited), 256 CUs
    int per_cu = (int)((160 * 1024) / p.lds);
    const int by_regs = 2048 / p.threads;  // 128 VGPRs -> 4 waves per SIMD = 16 waves per CU
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu < 1) per_cu = 1;
    // the spread costs up to `stagger` x ~1000 cycles once per launch: scale it down for
    // launches of only a few generations of workgroups
    uint32_t first_gen = 256u * (uint32_t)per_cu;
    uint32_t stagger = (uint32_t)(4 * total / first_gen);
    if (stagger > 24) stagger = 24;
#ifdef AOF_LAB
    if (g_lab_stagger >= 0) stagger = (uint32_t)g_lab_stagger;
#endif
    hipLaunchKernelGGL(fn, dim3((uint32_t)total), dim3(p.threads), p.lds, s, a, p.rb, p.nstrips, spw,
                       (uint32_t)total, first_gen, stagger);
    return (int)hipGetLastError();
}

}  // namespace aof
