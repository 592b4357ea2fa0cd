// K2 (tile16) -- 16x16 SAD search over +-8 px on a dense grid (BASELINE configs[4],
// the LDS-tile stress case; DESIGN.md "Kernels").
//
// 289 candidates x 256 pixels per block is too much state for one lane (64 VGPRs of
// reference tile + 17 x 9 accumulator registers), so the work item here is
// (block, dy): a workgroup stages ONE block row -- 32 cur rows + 16 prev rows, two
// flat coalesced copies, 60 KB at 1280 px -- and its lanes walk the 17*nx items
// dy-major, so the lanes of a wave share dy and read neighbouring 32-byte windows
// (16-byte aligned: origin S = 8, step 16 -> ds_read_b128).  Per (ref row, search
// row) pair an item issues 16 v_qsad_pk_u16_u8 (offsets 0..15, four per
// instruction) and 4 v_sad_hi_u8 (offset 16, accumulating into sad<<16 | idx).
// The 17 dy-items of a block meet in an LDS atomicMin on the packed key
// (sad << 16 | idx): integer min is order-independent and reproduces "first
// minimum in scan order wins" exactly (16x16 max SAD 65 280 fits 16 bits, idx < 289).
// Half-pixel configurations stage one more cur row either side and refine the accepted blocks
// behind the search, out of the same tile (<.., REFINE>, below).
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_refine.hpp"

namespace aof {

bool tile16_refines(const SearchArgs &a);

namespace {

constexpr int kThreads = 512;
constexpr int kSide = 17;  // 2S+1
constexpr int kBoundRows = 2;  // tile rows summed for the lower bound of the pruned search
// ADAPTIVE mode (the default of 16x16 contexts): the probe kernel below decides per pair.
constexpr int kProbeThreads = 1024;        // a pair's ~1 200 probe items in two rounds
constexpr int kProbeStride = 8;            // every eighth block in x and in y is probed (VGA-like grids: ~1.6 % of the blocks)
constexpr int kProbeMaxBlocks = 128;       // sample blocks per pair at the most (the strides grow beyond that)
// Blocks stride/2, stride/2 + stride, ... of an axis of n blocks.
__host__ __device__ constexpr int probe_samples(int n, int stride) { return (n + stride - 1 - stride / 2) / stride; }
// (thresholds from a same-box sweep over sensor noise 0 .. 40 LSB, profiles/r04_c5_adaptive_thresholds.txt: with
//  6 x / 35 % the mode follows the faster of the two fixed modes within 5 % at every noise level but one, 8 %)
constexpr uint32_t kFullOverBound = 6;     // a row can survive when its two-row bound <= this x the block's smallest bound
constexpr uint32_t kMaxSurvivorsPct = 35;  // more predicted survivors than this: the exhaustive scan is faster
// ONE-row bounds in step A (half its SAD instructions) where the probe finds that one tile row separates the candidates already
// -- noise-free pairs: c5 310 -> 260 us per 256 pairs; from +-2 LSB on the two-row bounds are the cheaper way (1-row bounds for
// every pair: +-2 LSB 326 -> 391 us) --: hints[pair] = 2.  A row would survive its one-row bound when that does not exceed
// kFullOverOneRow times the block's smallest one; at most kMaxOneRowSurvivorsPct of them may.
constexpr uint32_t kFullOverOneRow = 12;
constexpr uint32_t kMaxOneRowSurvivorsPct = 4;
// DEEPER bounds where two rows do not separate the candidates any more (sensor noise): a partial sum over four or eight of
// the sixteen tile rows is a lower bound like any other, costs a quarter or half of the exhaustive scan in step A, and still
// drops most rows where the two-row bound drops none -- hints[pair] = 3 (four rows) or 4 (eight rows).  The probe sums them only
// for pairs it would otherwise send to the exhaustive scan, and it does not guess there: it evaluates the sample blocks' best
// row completely (what step B1 will do) and counts the rows whose deeper bound does not exceed THAT -- the survivors the search
// will really have.  A verdict needs the expected work -- step A + the survivors' sums -- well under the exhaustive scan's.
constexpr uint32_t kMaxFourRowSurvivorsPct = 40, kMaxEightRowSurvivorsPct = 10;
// The deeper look costs the probe 22 us per 256 pairs (its loads are scattered): it is spared where the best row does not stand
// out of its block's two-row bounds at all -- its bound in per mille of their mean: +-8 LSB 210, +-16 LSB 390 (eight rows still
// pay there), +-24 LSB 543, +-40 LSB 738, the realistic input 683 (no depth pays from there on).
constexpr uint32_t kMaxSeparationForDeeperLook = 450;
constexpr int kRefineParts = 4;  // lanes per block in the half-pixel refinement (1: 3.56, 2: 3.31, 4: 3.28 ms per 1 024 c5h pairs)

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

// One (dy row, block) item: the SADs of all 17 dx over the NR tile rows first, first + STEP, ...
// (16, 1 from row 0: the whole tile; 4, 4: a quarter of it -- the lower bound of the pruned search
// or one lane's share when four lanes split an item).  Leaves the sums in acc (offsets 4g..4g+3,
// packed u16) and acc16 (offset 16 in the high half, on top of what the caller put there).
template <int NR, int STEP>
__device__ __forceinline__ void sum_item(const uint8_t *s_prev, const uint8_t *s_cur, int W, int dyi, int bx, int xs,
                                         int first, u64 (&acc)[4], uint32_t &acc16, int delta = 0)
{
    // reference tile rows: 4 dwords at frame column 16*bx + 8.  The prev rows are staged from column 8 on, so that a
    // tile starts on a 16-byte boundary of the LDS image: ONE ds_read_b128 per row, neighbouring lanes on neighbouring
    // 16 bytes (two 8-byte reads at a 16-byte lane stride hit every bank pair twice per pass)
    uint32_t ref[NR][4];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        uint4 q;
        __builtin_memcpy(&q, s_prev + (size_t)(first + i * STEP) * W + 16 * bx, 16);   // (LDS: aligned; the probe reads global memory here)
        ref[i][0] = q.x; ref[i][1] = q.y; ref[i][2] = q.z; ref[i][3] = q.w;
    }
    const uint8_t *win = s_cur + (size_t)(dyi + first) * W + xs;
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const uint4 *p = reinterpret_cast<const uint4 *>(win + (size_t)(i * STEP) * W);
        uint4 q0 = p[0], q1 = p[1];
        if (delta != 0) { q0 = sat_add_u8x16(q0, delta); q1 = sat_add_u8x16(q1, delta); }   // (the probe: raw frames)
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        u64 pr[7];
#pragma unroll
        for (int j = 0; j < 7; j++) pr[j] = pack64(w[j], w[j + 1]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g] = qsad(pr[g + k], ref[i][k], acc[g]);
            acc16 = __builtin_amdgcn_sad_hi_u8(w[4 + k], ref[i][k], acc16);
        }
    }
}

// Smallest packed key (sad << 16 | idx) of a dy row's 17 sums = its first minimum in scan order.
__device__ __forceinline__ uint32_t row_key(const u64 (&acc)[4], uint32_t acc16, int dyi)
{
    uint32_t best = acc16;
    const uint32_t base = (uint32_t)(dyi * kSide);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t l = (uint32_t)acc[g], h = (uint32_t)(acc[g] >> 32);
        const uint32_t k0 = (l << 16) | (base + 4 * g + 0), k1 = (l & 0xFFFF0000u) | (base + 4 * g + 1);
        const uint32_t k2 = (h << 16) | (base + 4 * g + 2), k3 = (h & 0xFFFF0000u) | (base + 4 * g + 3);
        best = min(best, min(min(k0, k1), min(k2, k3)));
    }
    return best;
}

// The smallest of a dy row's 17 partial sums alone (the lower bounds of the pruned search need no candidate index): packed
// minima instead of 17 keys -- 11 instructions against row_key's 25, on every (dy, block) item of every block row.
typedef unsigned short ushort2_t16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t16, x), __builtin_bit_cast(ushort2_t16, y)));
}
template <int NR, int STEP>
__device__ __forceinline__ uint32_t bound_item(const uint8_t *s_prev, const uint8_t *s_cur, int W, int dyi, int bx, int xs,
                                               int first, int delta = 0)
{
    u64 acc[4] = {0, 0, 0, 0};
    uint32_t acc16 = 0;   // offset 16 in the high half
    sum_item<NR, STEP>(s_prev, s_cur, W, dyi, bx, xs, first, acc, acc16, delta);
    const uint32_t m01 = pk_min_u16(pk_min_u16((uint32_t)acc[0], (uint32_t)(acc[0] >> 32)), pk_min_u16((uint32_t)acc[1], (uint32_t)(acc[1] >> 32)));
    const uint32_t m23 = pk_min_u16(pk_min_u16((uint32_t)acc[2], (uint32_t)(acc[2] >> 32)), pk_min_u16((uint32_t)acc[3], (uint32_t)(acc[3] >> 32)));
    const uint32_t m = pk_min_u16(pk_min_u16(m01, m23), acc16 | 0xFFFFu);
    return min(m & 0xFFFFu, m >> 16);
}

template <int NR, int STEP>
__device__ __forceinline__ uint32_t eval_item(const uint8_t *s_prev, const uint8_t *s_cur, int W, int dyi, int bx, int xs,
                                              int first, int delta = 0)
{
    u64 acc[4] = {0, 0, 0, 0};
    uint32_t acc16 = (uint32_t)(dyi * kSide + 16);  // offset 16 as sad<<16 | idx
    sum_item<NR, STEP>(s_prev, s_cur, W, dyi, bx, xs, first, acc, acc16, delta);
    return row_key(acc, acc16, dyi);
}

// Staging of a block row: `total` 16-byte chunks, global memory -> LDS, `map(c, src, dst, is_cur)` names chunk c's addresses.
// ALL of a lane's loads are issued before its first LDS write (eight in flight: the whole 60 KB tile of a 1280-pixel row is
// on its way at once).  A plain copy loop compiles to load / s_waitcnt vmcnt(0) / ds_write per chunk -- one memory round
// trip per 8 KB of the tile, seven or eight in a row, which is what the pruned search's workgroups spent most of their
// time on (round 5: profiles/r05_tile16_staging.txt).  Lanes past the end load the last chunk again and drop it.
// Only where the workgroup is short of work behind the staging: the pruned steps and the half-pixel refinement (c5h
// 492 -> 435 us per 256 pairs, c5 365 -> 352, c5p 875 -> 833).  The exhaustive scan is bound by its SAD stream, its two
// workgroups per CU overlap one's staging with the other's scan, and a tile that arrives all at once only puts them in
// step: +3 % (706 -> 730 us) -- it keeps one chunk per lane in flight (kStageUnroll = 1).
template <int kStageUnroll, typename Map>
__device__ __forceinline__ void stage_chunks(int total, int tid, int delta, Map &&map)
{
    for (int c0 = tid; c0 < total; c0 += kStageUnroll * kThreads) {
        uint4 v[kStageUnroll];
        uint8_t *to[kStageUnroll];       // (LDS: one register each; kept from the address pass, not computed twice)
        bool shift[kStageUnroll];
#pragma unroll
        for (int u = 0; u < kStageUnroll; u++) {
            const int c = min(c0 + u * kThreads, total - 1);
            const uint8_t *src;
            bool is_cur;
            map(c, src, to[u], is_cur);
            shift[u] = is_cur && delta != 0;
            __builtin_memcpy(&v[u], src, 16);
        }
#pragma unroll
        for (int u = 0; u < kStageUnroll; u++) {
            if (c0 + u * kThreads < total) {
                uint4 w = v[u];
                if (shift[u]) w = sat_add_u8x16(w, delta);
                *reinterpret_cast<uint4 *>(to[u]) = w;
            }
        }
    }
}

// Appends the items of the wave's lanes that `keep` to a list in LDS, in lane order, with one
// atomic per wave: neighbouring list entries stay neighbouring (dy, block) items, so the lanes
// that later walk the list read neighbouring LDS windows like the exhaustive scan does (a list
// filled by per-lane atomics comes out shuffled, and its window reads collide in the banks).
// Must be reached by whole waves.
__device__ __forceinline__ void wave_append(uint16_t *list, uint32_t *count, bool keep, int item)
{
    const unsigned long long m = __ballot(keep);
    if (m == 0) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (keep) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)item;
}

// REFINE (half-pixel refinement, origin S+1): the ring of every best match lies in the staged block row
// once one more cur row above and below it is staged (rows -1 and 32), so the eight direction SADs are
// summed out of LDS behind the search -- four lanes per block, four tile rows each, joined by shuffles --
// instead of by a second pass over global memory (K2b, bound by the L1 rate of its per-lane row loads).
template <bool PRUNE, bool REFINE>
__global__ __launch_bounds__(kThreads) void k_search_tile16(SearchArgs a, uint32_t total_wgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int kLead = REFINE ? 1 : 0;            // cur rows staged above row 0 (and below row 31)
    constexpr int kCurRows = 32 + 2 * kLead;
    constexpr int kPadBytes = REFINE ? 16 : 0;       // the byte left of row -1 (ring column -1 of block 0) lives here

    const uint32_t logical = xcd_remap(blockIdx.x, total_wgs);
    const int W = a.w, nx = a.grid.nx, ny = a.grid.ny;
    const int by = (int)(logical % (uint32_t)ny);
    const int64_t pair = (int64_t)(logical / (uint32_t)ny);
    const int tid = threadIdx.x;
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(W * a.h));
    // ADAPTIVE (a.prune == 2): the probe kernel in front of this launch has judged the pair (a.hints); PRUNED: always
    typedef const __attribute__((address_space(4))) uint32_t *const_u32;   // (scalar load: uniform in the workgroup)
    const uint32_t hint = PRUNE ? (a.prune != 2 ? 1u : ((const_u32)a.hints)[pair] & 0xFFu) : 0u;   // 0: exhaustive scan; 1, 2, 3, 4: step A on two-, one-, four-, eight-row bounds
    const bool pays = hint != 0u;

    // Level 0 under a predictor (px, py): the block row's windows move by py rows and px columns.
    // The cur rows are staged PRE-SHIFTED by px mod 16, so that LDS column c
    // holds frame column c + sh and every window is again 16-byte aligned at LDS column
    // 16*(bx + (px >> 4)); rows pushed outside the frame skip the whole block row.
    int px = 0, py = 0;
    if (a.pred) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
    const int sh = px & 15;                                       // floor-mod
    uint8_t *s_cur = smem + kPadBytes + (size_t)kLead * W;                // frame rows [16*by + py, +32) (REFINE: one more either side)
    uint8_t *s_prev = smem + kPadBytes + (size_t)kCurRows * W;            // frame rows [16*by + 8, +16) from column 8 on (sum_item)
    uint32_t *s_best = reinterpret_cast<uint32_t *>(smem + kPadBytes + (size_t)(kCurRows + 16) * W);
    // Half-pixel refinement moves the grid origin to S+1 = 9: the same geometry on a frame whose
    // origin is moved by (1, 1) -- the flat copies start W+1 bytes later (byte-aligned loads);
    // the grid keeps every window inside the smaller frame.  K2b adds the directions.
    const int org = a.grid.x0 - 8;
    const int H = a.h - 2 * org, Wb = W - 2 * org;               // the moved frame
    const int64_t org_off = (int64_t)org * (W + 1);
    const int yc0 = 16 * by + py;                                 // moved-frame row of LDS cur row 0
    const bool rows_ok = yc0 >= 0 && yc0 + 32 <= H;               // wave-uniform: the whole block row
    const uint8_t *g_cur = a.cur + pair * a.pair_stride + org_off + (int64_t)yc0 * W + sh;
    const uint8_t *g_prev = a.prev + pair * a.pair_stride + org_off + (int64_t)(16 * by + 8) * W + 8;   // (8-byte aligned at best)
    int cur_chunks = rows_ok ? 32 * (W / 16) : 0;
    const int prev_chunks = 16 * (W / 16);
    // the displaced copy ends sh bytes past its last row: bytewise when that is the frame's end
    // (with the moved origin the frame's own last row and column absorb the over-read)
    const bool tail_guard = sh != 0 && org == 0 && rows_ok && yc0 + 32 == H;
    if (tail_guard) {
        cur_chunks -= 1;
        if (tid < 16 - sh)
            s_cur[(size_t)cur_chunks * 16 + tid] = (uint8_t)clamp_u8((int)g_cur[(size_t)cur_chunks * 16 + tid] + delta);
    }
    if constexpr (REFINE) {
        // (origin 1: byte-aligned sources.)  Rows -1 .. 31 as one flat copy -- it ends 1 + sh bytes into
        // frame row yc0 + 33, which exists --, row 32 only as far as the rings reach (the frame's last
        // column: its flat copy would end past the frame's last row), and the one byte left of row -1.
        const int flat_chunks = rows_ok ? 33 * (W / 16) : 0;
        const int last_bytes = rows_ok ? W - 1 - sh : 0;
        const int last_chunks = last_bytes / 16;
        const uint8_t *g_flat = g_cur - W, *g_last = g_cur + (int64_t)32 * W;
        uint8_t *s_flat = s_cur - W, *s_last = s_cur + (size_t)32 * W;
        stage_chunks<8>(flat_chunks + last_chunks + prev_chunks, tid, delta, [&](int c, const uint8_t *&src, uint8_t *&dst, bool &is_cur) {
            is_cur = true;
            if (c < flat_chunks) { src = g_flat + (size_t)c * 16; dst = s_flat + (size_t)c * 16; }
            else if (c < flat_chunks + last_chunks) { src = g_last + (size_t)(c - flat_chunks) * 16; dst = s_last + (size_t)(c - flat_chunks) * 16; }
            else { is_cur = false; src = g_prev + (size_t)(c - flat_chunks - last_chunks) * 16; dst = s_prev + (size_t)(c - flat_chunks - last_chunks) * 16; }
        });
        const int odd = last_bytes - 16 * last_chunks;   // < 16 bytes of row 32, and the byte left of row -1
        if (tid < odd) s_last[16 * last_chunks + tid] = (uint8_t)clamp_u8((int)g_last[16 * last_chunks + tid] + delta);
        if (tid == 32 && rows_ok) s_flat[-1] = (uint8_t)clamp_u8((int)g_flat[-1] + delta);
    } else {   // (a moved origin or a displaced row: byte-aligned sources; unaligned 16-byte loads all the same)
        auto where = [&](int c, const uint8_t *&src, uint8_t *&dst, bool &is_cur) {
            is_cur = c < cur_chunks;
            const int cc = is_cur ? c : c - cur_chunks;
            src = (is_cur ? g_cur : g_prev) + (size_t)cc * 16;
            dst = (is_cur ? s_cur : s_prev) + (size_t)cc * 16;
        };
        if (PRUNE && pays) stage_chunks<8>(cur_chunks + prev_chunks, tid, delta, where);   // (uniform in the workgroup)
        else stage_chunks<1>(cur_chunks + prev_chunks, tid, delta, where);
    }
    for (int b = tid; b < nx; b += kThreads) s_best[b] = 0xFFFFFFFFu;
    __syncthreads();

    const int items = rows_ok ? kSide * nx : 0;
    if constexpr (!PRUNE) {
        for (int item = tid; item < items; item += kThreads) {
            const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
            const int xf = 16 * bx + px;                  // moved-frame column of the window start
            if (xf < 0 || xf + 32 > Wb) continue;         // window leaves the frame: block skipped below
            atomicMin(&s_best[bx], eval_item<16, 1>(s_prev, s_cur, W, dyi, bx, xf - sh, 0));
        }
    } else {
        // Exact pruning (AOF_SEARCH_PRUNED), records bit-identical to the exhaustive scan:
        //  A  every (dy, block) item sums TWO of its sixteen tile rows for all 17 dx (an eighth of the
        //     work) and leaves the smallest partial SAD in s_pmin -- a lower bound of every full SAD
        //     of that dy row;
        //  B1 four lanes per block evaluate the dy row with the smallest bound completely: under a
        //     clean match that is the true row, and the block's best key is final;
        //  B2 every other item whose bound does not exceed the block's best SAD so far may still win
        //     or TIE (ties go to the earlier scan index): it gets a four-row bound, and if that does
        //     not exceed the best either it is evaluated completely.  Items with bound > best can
        //     neither win nor tie (a partial sum only grows) and are dropped.  The items left after
        //     each step are compacted into a list so that all lanes share them.
        uint16_t *s_pmin = reinterpret_cast<uint16_t *>(s_best + nx);          // [17][nx]
        uint16_t *s_list = s_pmin + kSide * nx;                                 // [17 * nx] item ids
        uint32_t *s_count = reinterpret_cast<uint32_t *>(s_list + kSide * nx);   // (2 * 34 * nx bytes: dword-aligned)
        if (tid == 0) *s_count = 0;
        // ADAPTIVE (a.prune == 2): the probe kernel in front of this launch has judged the pair (a.hints):
        // 0 = its candidates look alike (sensor noise: nothing could be dropped and the pruned steps would
        // cost 1.5x the exhaustive scan), so the workgroup runs the exhaustive scan; the records are the same.
        if (!pays) {
            for (int item = tid; item < items; item += kThreads) {
                const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
                const int xf = 16 * bx + px;
                if (xf < 0 || xf + 32 > Wb) continue;
                atomicMin(&s_best[bx], eval_item<16, 1>(s_prev, s_cur, W, dyi, bx, xf - sh, 0));
            }
        }
        const int items_all = items;
        const int items = pays ? items_all : 0;   // (the pruned steps below then have nothing to do)
        if (hint >= 3u) {   // (sensor noise: deeper bounds; the pair's verdict is uniform in the workgroup)
            for (int item = tid; item < items; item += kThreads) {
                const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
                const int xf = 16 * bx + px;
                uint32_t bound = 0xFFFFu;
                if (!(xf < 0 || xf + 32 > Wb))
                    bound = hint == 3u ? bound_item<4, 4>(s_prev, s_cur, W, dyi, bx, xf - sh, 2)    // rows 2, 6, 10, 14
                                       : bound_item<8, 2>(s_prev, s_cur, W, dyi, bx, xf - sh, 0);   // rows 0, 2, .., 14
                s_pmin[item] = (uint16_t)bound;
            }
        } else {
        const bool one_row = hint == 2u;
        for (int item = tid; item < items; item += kThreads) {
            const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
            const int xf = 16 * bx + px;
            uint32_t bound = 0xFFFFu;
            if (!(xf < 0 || xf + 32 > Wb))   // (uniform in the workgroup: the pair's hint)
                bound = one_row ? bound_item<1, 16>(s_prev, s_cur, W, dyi, bx, xf - sh, 8 / kBoundRows)
                                : bound_item<kBoundRows, 16 / kBoundRows>(s_prev, s_cur, W, dyi, bx, xf - sh, 8 / kBoundRows);
            s_pmin[item] = (uint16_t)bound;
        }
        }
        __syncthreads();
        // (four lanes per block, four interleaved tile rows each, sums joined across the quad: the
        //  step is one item's latency long whatever the lane count, so it is kept a quarter item)
        for (int q = tid; q < (rows_ok && pays ? 4 * ((nx + 15) / 16 * 16) : 0); q += kThreads) {   // whole waves: shuffles
            const int bx = q >> 2, part = q & 3;
            const int xf = 16 * bx + px;
            const bool in = bx < nx && !(xf < 0 || xf + 32 > Wb);
            uint32_t m = 0xFFFFFFFFu;                      // (bound << 8 | dy): first smallest bound
            if (in) {
#pragma unroll
                for (int d = 0; d < kSide; d++) m = min(m, ((uint32_t)s_pmin[d * nx + bx] << 8) | (uint32_t)d);
            }
            const int dyi = (int)(m & 0xFFu);
            u64 acc[4] = {0, 0, 0, 0};
            uint32_t acc16 = part == 0 ? (uint32_t)(dyi * kSide + 16) : 0u;   // (the index once per quad)
            if (in) sum_item<4, 4>(s_prev, s_cur, W, dyi, bx, xf - sh, part, acc, acc16);
            // u16 lanes cannot carry: a whole tile's SAD is at most 65 280
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = (uint32_t)acc[g] + (uint32_t)__shfl_xor((int)(uint32_t)acc[g], o, 64);
                    const uint32_t hi = (uint32_t)(acc[g] >> 32) + (uint32_t)__shfl_xor((int)(uint32_t)(acc[g] >> 32), o, 64);
                    acc[g] = pack64(lo, hi);
                }
                acc16 += (uint32_t)__shfl_xor((int)acc16, o, 64);
            }
            if (in && part == 0) {
                s_pmin[dyi * nx + bx] = 0xFFFFu;           // done: step B2 leaves it alone
                atomicMin(&s_best[bx], row_key(acc, acc16, dyi));
            }
        }
        __syncthreads();
        for (int item0 = tid - (tid & 63); item0 < items; item0 += kThreads) {   // whole waves
            const int item = item0 + (tid & 63);
            bool keep = false;
            if (item < items) {
                const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
                const uint32_t bound = s_pmin[item];
                keep = bound != 0xFFFFu && bound <= (s_best[bx] >> 16);
            }
            wave_append(s_list, s_count, keep, item);
        }
        __syncthreads();
        // B2: the survivors of the two-row bound get a four-row bound first (sensor noise lets half
        // of the two-row bounds through, the four-row bound stops most of those); what survives
        // that as well is evaluated completely.
        const int listed = (int)*s_count;
        __syncthreads();
        if (tid == 0) *s_count = 0;
        __syncthreads();
        uint16_t *s_list2 = s_pmin;   // (the two-row bounds are not read any more: list 2 takes their place)
        // The second bound is probed on the first round of the list: when it stops less than a
        // quarter of those items (noise-dominated frames), the rest of the list skips it and is
        // summed completely at once -- the row then costs the exhaustive scan plus the probes.
        bool bound_pays = true;
        for (int k0 = tid - (tid & 63); k0 < listed; k0 += kThreads) {   // whole waves
            const int k = k0 + (tid & 63);
            bool keep = false;
            int item = 0;
            if (k < listed) {
                item = s_list[k];
                keep = true;
                if (bound_pays) {
                    const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
                    const uint32_t bound = bound_item<4, 4>(s_prev, s_cur, W, dyi, bx, 16 * bx + px - sh, 1);
                    keep = bound <= (s_best[bx] >> 16);
                }
            }
            wave_append(s_list2, s_count, keep, item);
            if (k0 < kThreads && listed > kThreads) {   // after the first round (uniform)
                __syncthreads();
                bound_pays = 4 * (int)*s_count <= 3 * kThreads;
            }
        }
        __syncthreads();
        const int listed2 = (int)*s_count;
        for (int k = tid; k < listed2; k += kThreads) {
            const int item = s_list2[k];
            const int dyi = (int)fast_div((uint32_t)item, a.div_nx), bx = item - dyi * nx;
            atomicMin(&s_best[bx], eval_item<16, 1>(s_prev, s_cur, W, dyi, bx, 16 * bx + px - sh, 0));
        }
    }
    __syncthreads();

    // one lane per block: 4x4 gradient gate (tile bytes 6..9, rows 6..9) and the record
    for (int bx = tid; bx < nx; bx += kThreads) {
        uint32_t mid[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t *p =
                reinterpret_cast<const uint32_t *>(s_prev + (size_t)(6 + r) * W + 16 * bx + 4);
            mid[r] = __builtin_amdgcn_alignbyte(p[1], p[0], 2);  // tile bytes 6..9
        }
        uint32_t diff = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
        for (int r = 0; r < 4; r++)
            diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
        aof_block rec;
        rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
        const int xf = 16 * bx + px;
        uint32_t refine_key = 0xFFFFFFFFu;   // REFINE: the best key of an accepted block
        if (rows_ok && xf >= 0 && xf + 32 <= Wb && diff >= (uint32_t)a.feature_threshold) {
            const uint32_t best = s_best[bx];
            const int idx = (int)(best & 0xFFFFu);
            rec.dx = (int8_t)(px + idx % kSide - 8);
            rec.dy = (int8_t)(py + idx / kSide - 8);
            rec.sad = (uint16_t)(best >> 16);
            if (rec.sad != AOF_SAD_SKIPPED && (uint32_t)rec.sad < (uint32_t)a.value_threshold) refine_key = best;
        }
        a.blocks[pair * (int64_t)(nx * ny) + (int64_t)by * nx + bx] = rec;
        if constexpr (REFINE) s_best[bx] = refine_key;   // (this lane alone reads and writes entry bx here)
    }
    if constexpr (REFINE) {
        __syncthreads();
        // Four lanes per block, lane `part` feeds tile rows 4 part .. 4 part + 3 with their window rows
        // 4 part - 1 .. 4 part + 4 relative to the best match (aof_refine.hpp); the eight direction sums of
        // the four slices add up across the quad (integer sums: any order).  Whole waves: shuffles.
        uint8_t *subdirs = a.subdirs + pair * (int64_t)(nx * ny) + (int64_t)by * nx;
        constexpr int kParts = kRefineParts, kRows = 16 / kParts;
        for (int q = tid; q < (kParts * nx + 63) / 64 * 64; q += kThreads) {
            const int bx = q / kParts, part = q % kParts;
            const uint32_t key = bx < nx ? s_best[bx] : 0xFFFFFFFFu;
            const bool live = key != 0xFFFFFFFFu;
            RefineState<4, kRows> st;
            st.init();
            // (a match without a single count of difference has nothing below it; and a slice's direction sums are lower
            //  bounds of the block's: once no lane of the wave has one below its block's SAD, the rows that are left change nothing)
            if (live && (key >> 16) != 0) {
                bool over = false;   // (uniform among the lanes in here)
                const int idx = (int)(key & 0xFFFFu), dxi = idx % kSide, dyi = idx / kSide;
                uint32_t ref[kRows][4];
#pragma unroll
                for (int r = 0; r < kRows; r++) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(s_prev + (size_t)(kRows * part + r) * W + 16 * bx);
                    ref[r][0] = q.x; ref[r][1] = q.y; ref[r][2] = q.z; ref[r][3] = q.w;
                }
                // ring row y of the slice: LDS row dyi + kRows part + y, bytes [xs + dxi - 1, + 18)
                const int off0 = (dyi + kRows * part - 1) * W + (16 * bx + px - sh) + dxi - 1;
                for_rows<-1, kRows>([&](auto yc) {
                    constexpr int Y = decltype(yc)::value;
                    if (over) return;
                    const int off = off0 + (Y + 1) * W;
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(s_cur + (off & ~3));   // (off >= -W - 1: inside the pad)
                    const uint32_t shb = (uint32_t)off & 3u;
                    const uint32_t q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3], q4 = w[4], q5 = w[5];
                    const uint32_t d[5] = {__builtin_amdgcn_alignbyte(q1, q0, shb), __builtin_amdgcn_alignbyte(q2, q1, shb),
                                           __builtin_amdgcn_alignbyte(q3, q2, shb), __builtin_amdgcn_alignbyte(q4, q3, shb),
                                           __builtin_amdgcn_alignbyte(q5, q4, shb)};
                    st.template row<Y>(d, ref);
                    if constexpr (Y == 1) over = st.nobody_can_win(key >> 16);
                });
            }
#pragma unroll
            for (int o = 1; o < kParts; o <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; k++) st.acc[k] += (uint32_t)__shfl_xor((int)st.acc[k], o, 64);
            }
            if (bx < nx && part == 0) subdirs[bx] = (uint8_t)(live ? st.direction(key >> 16) : 8);
        }
    }
}

// The probe of the ADAPTIVE mode: ONE workgroup per pair takes every kProbeStride-th block in x and y,
// computes the two-row lower bounds of its 17 dy rows straight from global memory (what step A of the pruned
// search computes for every block; here for 1.6 % of them), and estimates how many (dy, block) items
// would survive the bounds: a row survives when its bound does not exceed kFullOverBound times the block's
// smallest bound -- the complete SAD of the best row is several times its two-row share (up to 8x when the
// residual is noise), and a row whose bound already exceeds the best complete SAD can be dropped.  More than
// kMaxSurvivorsPct of them: the pair would run the exhaustive scan, else hints[pair] = 1 -- or 2 where ONE row already separates
// the candidates (kFullOverOneRow, kMaxOneRowSurvivorsPct).  A pair that would run the exhaustive scan gets a deeper look at
// every other sample block: the row with the smallest two-row bound is evaluated completely (step B1's result), the four- and
// the eight-row bounds of the block's items are summed in one pass, and the rows at or below the evaluated SAD are counted --
// few enough of them: hints[pair] = 3 (step A on four-row bounds) or 4 (eight rows), else 0.  A verdict about SPEED only:
// every branch of the search kernel writes the same records.
__global__ __launch_bounds__(kProbeThreads) void k_tile16_probe(SearchArgs a, uint32_t *hints, int stride_x, int stride_y)
{
    __shared__ uint16_t s_bound[kProbeMaxBlocks][kSide + 1], s_bound1[kProbeMaxBlocks][kSide + 1];   // two-row, one-row bounds
    __shared__ uint32_t s_tot[5];   // survivors (two-row), rows, survivors (one-row); sum of the blocks' smallest two-row bounds, of all of them
    const int64_t pair = blockIdx.x;
    const int tid = threadIdx.x, W = a.w, nx = a.grid.nx, ny = a.grid.ny;
    const int sx = probe_samples(nx, stride_x), sy = probe_samples(ny, stride_y);   // blocks stride/2, + stride, ... per axis
    const int nsamp = sx * sy;   // 1 .. kProbeMaxBlocks (launcher)
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(W * a.h));
    int px = 0, py = 0;
    if (a.pred) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
    const int org = a.grid.x0 - 8;
    const int H = a.h - 2 * org, Wb = W - 2 * org;
    const uint8_t *prev = a.prev + pair * a.pair_stride + (int64_t)org * (W + 1);
    const uint8_t *cur = a.cur + pair * a.pair_stride + (int64_t)org * (W + 1);
    if (tid < 5) s_tot[tid] = 0;
    for (int s = tid; s < kSide * nsamp; s += kProbeThreads) {
        const int blk = s / kSide, dyi = s - blk * kSide;
        const int by = (blk / sx) * stride_y + stride_y / 2, bx = (blk % sx) * stride_x + stride_x / 2;
        const int xf = 16 * bx + px, yc0 = 16 * by + py;
        uint32_t bound = 0xFFFFu, bound1 = 0xFFFFu;
        if (xf >= 0 && xf + 32 <= Wb && yc0 >= 0 && yc0 + 32 <= H) {
            bound = bound_item<kBoundRows, 16 / kBoundRows>(prev + (int64_t)(16 * by + 8) * W + 8, cur + (int64_t)yc0 * W, W, dyi, bx, xf,
                                                            8 / kBoundRows, delta);
            bound1 = bound_item<1, 16>(prev + (int64_t)(16 * by + 8) * W + 8, cur + (int64_t)yc0 * W, W, dyi, bx, xf, 8 / kBoundRows, delta);
        }
        s_bound[blk][dyi] = (uint16_t)bound;
        s_bound1[blk][dyi] = (uint16_t)bound1;
    }
    __syncthreads();
    uint32_t would_survive = 0, would_survive1 = 0, rows = 0, smallest2 = 0, all2 = 0;
    for (int blk = tid; blk < nsamp; blk += kProbeThreads) {
        uint32_t m = 0xFFFFu, m1 = 0xFFFFu;
#pragma unroll
        for (int d = 0; d < kSide; d++) { m = min(m, (uint32_t)s_bound[blk][d]); m1 = min(m1, (uint32_t)s_bound1[blk][d]); }
        if (m == 0xFFFFu) continue;   // window outside the frame: the block is skipped anyway
        const uint32_t limit = kFullOverBound * m, limit1 = kFullOverOneRow * m1;
#pragma unroll
        for (int d = 0; d < kSide; d++) {
            would_survive += (uint32_t)s_bound[blk][d] <= limit ? 1u : 0u;
            would_survive1 += (uint32_t)s_bound1[blk][d] <= limit1 ? 1u : 0u;
            all2 += (uint32_t)s_bound[blk][d];
        }
        smallest2 += m;
        would_survive -= 1;            // (the best row itself is evaluated completely either way)
        would_survive1 -= 1;
        rows += kSide - 1;
    }
    would_survive = wave_sum_u32(would_survive);
    would_survive1 = wave_sum_u32(would_survive1);
    rows = wave_sum_u32(rows);
    smallest2 = wave_sum_u32(smallest2);
    all2 = wave_sum_u32(all2);
    if ((tid & 63) == 0) {
        atomicAdd(&s_tot[0], would_survive); atomicAdd(&s_tot[1], rows); atomicAdd(&s_tot[2], would_survive1);
        atomicAdd(&s_tot[3], smallest2); atomicAdd(&s_tot[4], all2);
    }
    __syncthreads();
    // how far the best row's two-row bound lies under the mean of its block's, in per mille (0: perfect matches; 1000: nothing
    // stands out) -- a diagnostic in the verdict word's upper bits
    const uint32_t separation = s_tot[4] != 0 ? (uint32_t)(((uint64_t)s_tot[3] * kSide * 1000u) / s_tot[4]) : 0u;
    uint32_t hint = (s_tot[1] != 0 && 100u * s_tot[0] > kMaxSurvivorsPct * s_tot[1]) ? 0u : 1u;   // (uniform: LDS totals)
    if (hint == 1u && s_tot[1] != 0 && 100u * s_tot[2] <= kMaxOneRowSurvivorsPct * s_tot[1]) hint = 2u;
    // two rows do not separate the candidates: do four, do eight?  (Only these pairs pay for the deeper sums -- their search
    // takes twice as long as a clean pair's anyway.)
    if (hint == 0u && s_tot[1] != 0 && separation <= kMaxSeparationForDeeperLook) {   // (uniform)
        __shared__ uint16_t s_full[kProbeMaxBlocks];   // the sample blocks' best SAD after step B1: the row with the smallest two-row bound, complete
        __syncthreads();
        if (tid < 3) s_tot[tid] = 0;
        // (the probe is bound by its scattered loads: the deeper look takes every OTHER sample block)
        // (a) what step B1 will find: four lanes per block, four interleaved tile rows each, joined across the quad
        for (int q = tid; q < (4 * ((nsamp + 1) / 2) + 63) / 64 * 64; q += kProbeThreads) {   // whole waves: shuffles
            const int blk = 2 * (q >> 2), part = q & 3;
            uint32_t m = 0xFFFFFFFFu;
            if (blk < nsamp) {
#pragma unroll
                for (int d = 0; d < kSide; d++) m = min(m, ((uint32_t)s_bound[blk][d] << 8) | (uint32_t)d);
            }
            const bool in = blk < nsamp && (m >> 8) != 0xFFFFu;
            u64 acc[4] = {0, 0, 0, 0};
            uint32_t acc16 = 0;
            if (in) {
                const int by = (blk / sx) * stride_y + stride_y / 2, bx = (blk % sx) * stride_x + stride_x / 2;
                sum_item<4, 4>(prev + (int64_t)(16 * by + 8) * W + 8, cur + (int64_t)(16 * by + py) * W, W, (int)(m & 0xFFu), bx, 16 * bx + px, part, acc, acc16, delta);
            }
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = (uint32_t)acc[g] + (uint32_t)__shfl_xor((int)(uint32_t)acc[g], o, 64);
                    const uint32_t hi = (uint32_t)(acc[g] >> 32) + (uint32_t)__shfl_xor((int)(uint32_t)(acc[g] >> 32), o, 64);
                    acc[g] = pack64(lo, hi);
                }
                acc16 += (uint32_t)__shfl_xor((int)acc16, o, 64);
            }
            if (blk < nsamp && part == 0) s_full[blk] = in ? (uint16_t)(row_key(acc, acc16, 0) >> 16) : (uint16_t)0xFFFFu;
        }
        // (b) the four-row bounds (tile rows 2, 6, 10, 14) and the eight-row bounds (+ rows 0, 4, 8, 12) of those blocks' items in
        //     one pass (24 loads per item)
        const int ndeep = (nsamp + 1) / 2;
        for (int s = tid; s < kSide * ndeep; s += kProbeThreads) {
            const int blk = 2 * (s / kSide), dyi = s - (s / kSide) * kSide;
            const int by = (blk / sx) * stride_y + stride_y / 2, bx = (blk % sx) * stride_x + stride_x / 2;
            const int xf = 16 * bx + px, yc0 = 16 * by + py;
            uint32_t b4 = 0xFFFFu, b8 = 0xFFFFu;
            if (xf >= 0 && xf + 32 <= Wb && yc0 >= 0 && yc0 + 32 <= H) {
                u64 acc[4] = {0, 0, 0, 0};
                uint32_t acc16 = 0;
                auto smallest = [](const u64 (&acc)[4], uint32_t acc16) -> uint32_t {
                    const uint32_t m01 = pk_min_u16(pk_min_u16((uint32_t)acc[0], (uint32_t)(acc[0] >> 32)), pk_min_u16((uint32_t)acc[1], (uint32_t)(acc[1] >> 32)));
                    const uint32_t m23 = pk_min_u16(pk_min_u16((uint32_t)acc[2], (uint32_t)(acc[2] >> 32)), pk_min_u16((uint32_t)acc[3], (uint32_t)(acc[3] >> 32)));
                    const uint32_t m = pk_min_u16(pk_min_u16(m01, m23), acc16 | 0xFFFFu);
                    return min(m & 0xFFFFu, m >> 16);
                };
                sum_item<4, 4>(prev + (int64_t)(16 * by + 8) * W + 8, cur + (int64_t)yc0 * W, W, dyi, bx, xf, 2, acc, acc16, delta);
                b4 = smallest(acc, acc16);
                // (the same sums go on: u16 lanes, eight rows of 16 pixels stay below 65 536)
                sum_item<4, 4>(prev + (int64_t)(16 * by + 8) * W + 8, cur + (int64_t)yc0 * W, W, dyi, bx, xf, 0, acc, acc16, delta);
                b8 = smallest(acc, acc16);
            }
            s_bound[blk][dyi] = (uint16_t)b4;
            s_bound1[blk][dyi] = (uint16_t)b8;
        }
        __syncthreads();
        uint32_t surv4 = 0, surv8 = 0, n = 0;
        for (int blk = 2 * tid; blk < nsamp; blk += 2 * kProbeThreads) {
            const uint32_t full = s_full[blk];
            if (full == 0xFFFFu) continue;
#pragma unroll
            for (int d = 0; d < kSide; d++) {
                surv4 += (uint32_t)s_bound[blk][d] <= full ? 1u : 0u;
                surv8 += (uint32_t)s_bound1[blk][d] <= full ? 1u : 0u;
            }
            surv4 -= 1; surv8 -= 1;   // (the evaluated row itself)
            n += kSide - 1;
        }
        surv4 = wave_sum_u32(surv4);
        surv8 = wave_sum_u32(surv8);
        n = wave_sum_u32(n);
        if ((tid & 63) == 0) { atomicAdd(&s_tot[0], surv4); atomicAdd(&s_tot[1], n); atomicAdd(&s_tot[2], surv8); }
        __syncthreads();
        if (s_tot[1] != 0 && 100u * s_tot[0] <= kMaxFourRowSurvivorsPct * s_tot[1]) hint = 3u;
        else if (s_tot[1] != 0 && 100u * s_tot[2] <= kMaxEightRowSurvivorsPct * s_tot[1]) hint = 4u;
    }
    if (tid == 0) hints[pair] = hint | (separation << 8);   // (low byte: the verdict)
}

size_t tile16_lds(const SearchArgs &a)
{
    size_t bytes = (size_t)48 * a.w + 4 * (size_t)a.grid.nx + 16;
    if (tile16_refines(a)) bytes += 2 * (size_t)a.w + 16 + 16;   // cur rows -1 and 32, the pad in front, dword reads past the last ring
    if (a.prune) bytes += 4 * (size_t)kSide * a.grid.nx + 16;   // lower bounds + item list + counter
    return bytes;
}

}  // namespace

// Half-pixel refinement runs inside the search launch (k_search_tile16<.., true>) whenever directions are wanted.
bool tile16_refines(const SearchArgs &a) { return a.subpixel && a.subdirs; }

bool tile16_supported(const SearchArgs &a)
{
    if (a.tile != 16 || a.search != 8) return false;
    const int org = a.subpixel ? 1 : 0;  // origin S+1: K2b follows
    if (a.grid.x0 != 8 + org || a.grid.y0 != 8 + org || a.grid.step_x != 16 || a.grid.step_y != 16) return false;
    if (a.w % 16 || a.pair_stride % 16) return false;
    if (reinterpret_cast<uintptr_t>(a.prev) % 16 || reinterpret_cast<uintptr_t>(a.cur) % 16) return false;
    if ((int64_t)a.w * a.h > 0x7FFFFFFF) return false;
    return tile16_lds(a) <= 156 * 1024;
}

int launch_search_tile16(const SearchArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    const int64_t total = a.n_pairs * a.grid.ny;
    if (total > 0x7FFFFFFF) return (int)hipErrorInvalidValue;
    if (a.prune == 2) {   // ADAPTIVE: judge every pair first (hints in the workspace), then the search decides per pair
        if (!a.hints) return (int)hipErrorInvalidValue;
        // every kProbeStride-th block per axis; an axis with fewer blocks than that is sampled more densely (at least one
        // block), and the strides grow -- the longer axis first -- until the sample fits the kernel's table
        int stx = kProbeStride, sty = kProbeStride;
        while (stx > 1 && probe_samples(a.grid.nx, stx) == 0) stx /= 2;
        while (sty > 1 && probe_samples(a.grid.ny, sty) == 0) sty /= 2;
        while (probe_samples(a.grid.nx, stx) * probe_samples(a.grid.ny, sty) > kProbeMaxBlocks) {
            if (probe_samples(a.grid.nx, stx) >= probe_samples(a.grid.ny, sty)) stx *= 2; else sty *= 2;
        }
        hipLaunchKernelGGL(k_tile16_probe, dim3((uint32_t)a.n_pairs), dim3(kProbeThreads), 0, static_cast<hipStream_t>(stream), a,
                           a.hints, stx, sty);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    SearchArgs k = a;
    k.div_nx = fastdiv_make((uint32_t)a.grid.nx);   // item -> (dy row, block): a run-time division costs ~20 VALU per lane and item
    const size_t lds = tile16_lds(a);
    void (*fn)(SearchArgs, uint32_t) =
        tile16_refines(a) ? (a.prune ? k_search_tile16<true, true> : k_search_tile16<false, true>)
                          : (a.prune ? k_search_tile16<true, false> : k_search_tile16<false, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3((uint32_t)total), dim3(kThreads), lds, static_cast<hipStream_t>(stream), k,
                       (uint32_t)total);
    return (int)hipGetLastError();
}

}  // namespace aof
