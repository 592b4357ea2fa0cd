// The lane-per-block 8x8 search shared by the flat and grouped kernels (k_search_lane8.hip) and by
// the one-workgroup two-level kernel (k_flow_small.hip): all 81 candidates of one block in one lane
// (DESIGN.md "Kernels": K2).
#pragma once

#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_refine.hpp"

namespace aof {

namespace {

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }
// Dwords 1 and 2 of a window row whose dwords (0, 1) and (2, 3) sit in two aligned register pairs: ONE v_pk_mov_b32
// (pack64(w.y, w.z) compiles to it in the exhaustive kernels, but to two v_mov_b32 in the pruned rows).
__device__ __forceinline__ u64 middle64(u64 p01, u64 p23)
{
    u64 r;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(p01), "v"(p23));
    return r;
}

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, x),
                                                                  __builtin_bit_cast(ushort2_t, y)));
}

// 4x4 gradient gate on tile bytes [2..5] x rows [2..5]
__device__ __forceinline__ uint32_t gradient_gate(const uint32_t (&ref)[8][2])
{
    uint32_t mid[4], diff = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) mid[r] = __builtin_amdgcn_alignbyte(ref[r + 2][1], ref[r + 2][0], 2);
#pragma unroll
    for (int r = 0; r < 3; r++) diff = __builtin_amdgcn_sad_u8(mid[r], mid[r + 1], diff);
#pragma unroll
    for (int r = 0; r < 4; r++)  // bytes (3,4,5,5) against (2,3,4,5): the doubled byte adds 0
        diff = __builtin_amdgcn_sad_u8(mid[r], __builtin_amdgcn_perm(0u, mid[r], 0x03030201u), diff);
    return diff;
}

// All 81 candidates: per dy, offsets 0..3 / 4..7 as packed u16, offset 8 as (sad << 16 | idx);
// returns the smallest packed key = first minimum in scan order.
template <bool EQUALISE>
__device__ __forceinline__ uint32_t exhaustive_search(const uint4 (&win)[16], const uint32_t (&ref)[8][2], int delta)
{
    u64 acc_lo[9], acc_hi[9];
    uint32_t acc_8[9];
#pragma unroll
    for (int d = 0; d < 9; d++) { acc_lo[d] = 0; acc_hi[d] = 0; acc_8[d] = (uint32_t)(d * 9 + 8); }
#pragma unroll
    for (int s = 0; s < 16; s++) {
        uint4 w = win[s];
        if (EQUALISE && delta != 0) w = sat_add_u8x16(w, delta);
        const u64 p01 = pack64(w.x, w.y), p12 = pack64(w.y, w.z), p23 = pack64(w.z, w.w);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int d = s - r;  // dy index, dy = d - 4
            if (d < 0 || d >= 9) continue;
            acc_lo[d] = qsad(p01, ref[r][0], acc_lo[d]);
            acc_lo[d] = qsad(p12, ref[r][1], acc_lo[d]);
            acc_hi[d] = qsad(p12, ref[r][0], acc_hi[d]);
            acc_hi[d] = qsad(p23, ref[r][1], acc_hi[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], acc_8[d]);
            acc_8[d] = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], acc_8[d]);
        }
    }
    // Keys and minimum strictly BEHIND the SAD stream: that is the order hipcc finds by itself in the plain
    // search kernel; in the kernel that also reduces it interleaves the two, which costs 3.5 % (round 3).
    __builtin_amdgcn_sched_barrier(0);
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const uint32_t base = (uint32_t)(d * 9);
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
    return best;
}

// The exhaustive scan of the PRUNED kernel, dy row by dy row: five accumulator registers live instead of the 45 of
// exhaustive_search, so that this path does not set the kernel's register allocation (122 VGPRs = four waves per SIMD,
// which the kernel needs to cover its row loads on images that prune: LAB_LOG.md, round 5).  Besides the best key
// it says how many of the nine dy rows the pruned code would PROBABLY have dropped for the whole wave: a row whose
// smallest SAD, scaled to what pruned_row's tests see of it, lies above every needing lane's best.  A guess about
// speed only (it picks the code that evaluates the wave's NEXT chunk); the keys are the exhaustive ones.
constexpr int kJudgedRowsToPrune = 3;       // the next chunk prunes when at least this many rows look droppable
constexpr uint32_t kPartialScaleQ10 = 410;  // 0.4: a row's smallest SAD over 64 pixels against what pruned_row's tests (16, then 32 pixels) see of it

template <int D>
__device__ __forceinline__ uint32_t full_row(const uint4 (&win)[16], const uint32_t (&ref)[8][2])
{
    u64 alo = 0, ahi = 0;
    uint32_t a8 = (uint32_t)(D * 9 + 8);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 w = win[D + r];
        const u64 p01 = pack64(w.x, w.y), p23 = pack64(w.z, w.w), p12 = middle64(p01, p23);
        alo = qsad(p01, ref[r][0], alo);
        ahi = qsad(p12, ref[r][0], ahi);
        alo = qsad(p12, ref[r][1], alo);
        ahi = qsad(p23, ref[r][1], ahi);
        a8 = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], a8);
        a8 = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], a8);
    }
    const uint32_t l0 = (uint32_t)alo, l1 = (uint32_t)(alo >> 32), h0 = (uint32_t)ahi, h1 = (uint32_t)(ahi >> 32);
    const uint32_t base = (uint32_t)(D * 9);
    const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
    const uint32_t k2 = (l1 << 16) | (base + 2), k3 = (l1 & 0xFFFF0000u) | (base + 3);
    const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
    const uint32_t k6 = (h1 << 16) | (base + 6), k7 = (h1 & 0xFFFF0000u) | (base + 7);
    return min(min(min(k0, k1), k2), min(min(min(k3, k4), k5), min(min(k6, k7), a8)));
}

__device__ __forceinline__ uint32_t exhaustive_search_judged(const uint4 (&win)[16], const uint32_t (&ref)[8][2],
                                                             unsigned long long needing, int &droppable)
{
    uint32_t row_key[9];
    for_rows<0, 8>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        row_key[D] = full_row<D>(win, ref);
    });
    uint32_t best = row_key[0];
#pragma unroll
    for (int d = 1; d < 9; d++) best = min(best, row_key[d]);
    droppable = 0;
#pragma unroll
    for (int d = 0; d < 9; d++)
        droppable += (__ballot((((row_key[d] >> 16) * kPartialScaleQ10) >> 10) <= (best >> 16)) & needing) == 0 ? 1 : 0;
    return best;
}

// One dy row (compile-time index D, so the window rows are plain registers) of the exact
// pruned search: a partial SAD only grows, so when after two (then four) of the eight row pairs
// no lane of the wave that still needs a result can beat or tie its best, the row is dropped for
// the whole wave.  Returns false when the row was dropped.  `needing` = ballot of the lanes that
// need a result (the others carry whatever their registers hold and are masked out of the test).
// A dropped row is the common case on images that prune at all (eight of nine rows under a global
// translation), so the test is kept short: ONE accumulator set (64 pixels x 255 fit a u16 field),
// the minimum over the nine offsets as four packed minima and one min, the ballot and its mask in
// scalar registers -- about 9 instructions behind the row's first 12 SAD instructions.
// `zero` (wave-uniform, set behind the block's FIRST row): every needing lane's best match has SAD 0.  Nothing is below that,
// and a row whose index lies BEHIND a lane's match cannot take a tie from it either (the key carries the index): best < 9 D as
// a key says both, and such a row goes without a SAD instruction.  (On BASELINE's noise-free translations: every row behind
// the matched one.  Where the matches have differences at all the flag is false and costs a scalar branch per row.)
template <int D, bool FIRST = false>
__device__ __forceinline__ bool pruned_row(const uint4 (&win)[16], const uint32_t (&ref)[8][2], unsigned long long needing,
                                           uint32_t &best, bool &zero)
{
    if constexpr (D > 0 && !FIRST) {
        if (zero) {
            if ((__ballot(best >= (uint32_t)(9 * D)) & needing) == 0) return false;
        }
    }
    u64 alo = 0, ahi = 0;
    uint32_t a8 = (uint32_t)(D * 9 + 8);
    auto row_pair = [&](int r) {
        const uint4 w = win[D + r];
        const u64 p01 = pack64(w.x, w.y), p23 = pack64(w.z, w.w), p12 = middle64(p01, p23);
        alo = qsad(p01, ref[r][0], alo);
        ahi = qsad(p12, ref[r][0], ahi);
        alo = qsad(p12, ref[r][1], alo);
        ahi = qsad(p23, ref[r][1], ahi);
        a8 = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], a8);
        a8 = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], a8);
    };
    auto nobody_can_win = [&]() -> bool {
        const uint32_t m = pk_min_u16(pk_min_u16((uint32_t)alo, (uint32_t)(alo >> 32)), pk_min_u16((uint32_t)ahi, (uint32_t)(ahi >> 32)));
        const uint32_t m9 = pk_min_u16(m, a8 | 0xFFFFu);   // offset 8's sum rides in a8's upper half
        const uint32_t smallest = min(m9 & 0xFFFFu, m9 >> 16);
        return (__ballot(smallest <= (best >> 16)) & needing) == 0;
    };
    row_pair(0);
    row_pair(4);
    if (nobody_can_win()) return false;
    row_pair(2);
    row_pair(6);
    if (nobody_can_win()) return false;
    row_pair(1);
    row_pair(3);
    row_pair(5);
    row_pair(7);
    const uint32_t l0 = (uint32_t)alo, l1 = (uint32_t)(alo >> 32), h0 = (uint32_t)ahi, h1 = (uint32_t)(ahi >> 32);
    const uint32_t base = (uint32_t)(D * 9);
    const uint32_t k0 = (l0 << 16) | (base + 0), k1 = (l0 & 0xFFFF0000u) | (base + 1);
    const uint32_t k2 = (l1 << 16) | (base + 2), k3 = (l1 & 0xFFFF0000u) | (base + 3);
    const uint32_t k4 = (h0 << 16) | (base + 4), k5 = (h0 & 0xFFFF0000u) | (base + 5);
    const uint32_t k6 = (h1 << 16) | (base + 6), k7 = (h1 & 0xFFFF0000u) | (base + 7);
    best = min(best, min(min(k0, k1), k2));
    best = min(best, min(min(k3, k4), k5));
    best = min(best, min(min(k6, k7), a8));
    if constexpr (FIRST) zero = (__ballot((best >> 16) != 0) & needing) == 0;
    return true;
}

// The k-th row of the visiting order start, start-1, start+1, start-2, ... (whichever side is
// still in range), closed form.
constexpr int visit_order(int start, int k)
{
    if (k == 0) return start;
    const int below = start, above = 8 - start;
    const int pairs = below < above ? below : above;
    if (k <= 2 * pairs) return (k & 1) ? start - (k + 1) / 2 : start + k / 2;
    return below > above ? start - pairs - (k - 2 * pairs) : start + pairs + (k - 2 * pairs);
}

// All nine rows from a compile-time start row: straight-line code, every window row a plain
// register (a run-time row index would push the 64-register window into scratch).
template <int START, int K = 0>
__device__ __forceinline__ int pruned_from(const uint4 (&win)[16], const uint32_t (&ref)[8][2], unsigned long long need,
                                           uint32_t &best, bool &zero)
{
    const int dropped = pruned_row<visit_order(START, K), K == 0>(win, ref, need, best, zero) ? 0 : 1;
    if constexpr (K < 8) return dropped + pruned_from<START, K + 1>(win, ref, need, best, zero);
    else return dropped;
}

// The nine dy rows in the order start, start-1, start+1, ... (start is wave-uniform): one
// specialised copy of the row sequence per start row; returns how many rows were dropped.
__device__ __forceinline__ int pruned_search(const uint4 (&win)[16], const uint32_t (&ref)[8][2], unsigned long long need,
                                             int start, uint32_t &best)
{
    bool zero = false;
    switch (start) {
    case 0: return pruned_from<0>(win, ref, need, best, zero);
    case 1: return pruned_from<1>(win, ref, need, best, zero);
    case 2: return pruned_from<2>(win, ref, need, best, zero);
    case 3: return pruned_from<3>(win, ref, need, best, zero);
    case 4: return pruned_from<4>(win, ref, need, best, zero);
    case 5: return pruned_from<5>(win, ref, need, best, zero);
    case 6: return pruned_from<6>(win, ref, need, best, zero);
    case 7: return pruned_from<7>(win, ref, need, best, zero);
    default: return pruned_from<8>(win, ref, need, best, zero);
    }
}

// Where a wave that knows nothing yet starts its pruned search (round 5).  A pruned block that starts in the wrong dy row
// costs 1.4x an exhaustive one (its first row sets no useful bound, and neither do the rows between it and the right
// one), and on a walk of two or three blocks -- a launch of 128 VGA pairs -- the first block is a third to a half of
// the work.  So the first block VOTES: of every dy row it sums what pruned_row's first test sees (tile rows 0 and 4:
// 16 of 64 pixels per candidate, 12 SAD instructions), every lane names the row with its smallest partial SAD, and the
// wave starts in the row most needing lanes name -- 108 SAD instructions against the 432 of a block, in front of a
// pruned search that then drops eight of nine rows under a global motion.  A guess about speed only.
template <int D>
__device__ __forceinline__ uint32_t partial_row_key(const uint4 (&win)[16], const uint32_t (&ref)[8][2])
{
    u64 alo = 0, ahi = 0;
    uint32_t a8 = 0;
#pragma unroll
    for (int r = 0; r < 8; r += 4) {
        const uint4 w = win[D + r];
        const u64 p01 = pack64(w.x, w.y), p23 = pack64(w.z, w.w), p12 = middle64(p01, p23);
        alo = qsad(p01, ref[r][0], alo);
        ahi = qsad(p12, ref[r][0], ahi);
        alo = qsad(p12, ref[r][1], alo);
        ahi = qsad(p23, ref[r][1], ahi);
        a8 = __builtin_amdgcn_sad_hi_u8(w.z, ref[r][0], a8);
        a8 = __builtin_amdgcn_sad_hi_u8(w.w, ref[r][1], a8);
    }
    const uint32_t m = pk_min_u16(pk_min_u16((uint32_t)alo, (uint32_t)(alo >> 32)), pk_min_u16((uint32_t)ahi, (uint32_t)(ahi >> 32)));
    const uint32_t m9 = pk_min_u16(m, a8 | 0xFFFFu);
    return (min(m9 & 0xFFFFu, m9 >> 16) << 4) | (uint32_t)D;
}

__device__ __forceinline__ int vote_start_row(const uint4 (&win)[16], const uint32_t (&ref)[8][2], unsigned long long needing)
{
    uint32_t key = 0xFFFFFFFFu;
    for_rows<0, 8>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        key = min(key, partial_row_key<D>(win, ref));
    });
    const int mine = (int)(key & 15u);
    int row = 4, most = 0;   // (scalar: nine ballots and population counts)
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const int n = __popcll(__ballot(mine == d) & needing);
        if (n > most) { most = n; row = d; }
    }
    return row;
}

// Half-pixel refinement with the ring taken out of the 16x16 window in registers: rows udy-1 .. udy+8 (udy wave-uniform,
// 1..7: compile-time inside the switch), bytes o .. o+9 (o per lane, 0..6: a v_alignbyte per dword).  Row by row, so
// that nothing but the window and the refinement's own state is live (the kernel stays at four waves per SIMD).
// `sad`: the integer match's SAD of the lane; the walk over the ring rows ends as soon as no lane in this control flow can
// still find a direction below it (RefineState::nobody_can_win).
template <int UDY>
__device__ __forceinline__ void refine_rows_from_window(const uint4 (&win)[16], int o, const uint32_t (&ref)[8][2], RefineState<2> &st,
                                                        uint32_t sad)
{
    const bool upper = o >= 4;   // (v_alignbyte shifts by o & 3)
    bool over = false;           // (wave-uniform)
    for_rows<-1, 8>([&](auto yc) {
        constexpr int Y = decltype(yc)::value;
        if (over) return;
        const uint4 w = win[UDY + Y];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(w.y, w.x, (uint32_t)o), a1 = __builtin_amdgcn_alignbyte(w.z, w.y, (uint32_t)o);
        const uint32_t a2 = __builtin_amdgcn_alignbyte(w.w, w.z, (uint32_t)o), a3 = __builtin_amdgcn_alignbyte(0u, w.w, (uint32_t)o);
        const uint32_t d[3] = {upper ? a1 : a0, upper ? a2 : a1, (upper ? a3 : a2) & 0xFFFFu};
        st.template row<Y>(d, ref);
        if constexpr (Y == 1 || Y == 3 || Y == 5) over = st.nobody_can_win(sad);   // (every direction has a tile row from Y = 1 on)
    });
}

__device__ __forceinline__ void refine_from_window(const uint4 (&win)[16], int udy, int o, const uint32_t (&ref)[8][2], RefineState<2> &st,
                                                   uint32_t sad)
{
    if (__ballot(sad != 0) == 0) return;   // (matches without a single count of difference: nothing can be below them)
    switch (udy) {
    case 1: refine_rows_from_window<1>(win, o, ref, st, sad); break;
    case 2: refine_rows_from_window<2>(win, o, ref, st, sad); break;
    case 3: refine_rows_from_window<3>(win, o, ref, st, sad); break;
    case 4: refine_rows_from_window<4>(win, o, ref, st, sad); break;
    case 5: refine_rows_from_window<5>(win, o, ref, st, sad); break;
    case 6: refine_rows_from_window<6>(win, o, ref, st, sad); break;
    default: refine_rows_from_window<7>(win, o, ref, st, sad); break;
    }
}

// The ring of a best match -- rows -1..8, bytes -1..8 -- straight from global memory (the lines were touched a moment ago):
// ONE 16-byte load per row (ten bytes of it are used), half the load instructions of an 8-byte + a 2-byte load.  A load
// that reaches past the end of the frame arrays does not return the bytes in front of the end either (the match in the
// last pair's bottom-right corner: tools/fuzz_gpu.py, seed 11 517 551), so a lane whose last row would do that loads
// byte-exactly; `records` = size of what the buffer resource covers, `ring` = offset of the ring's first byte in it.
__device__ __forceinline__ void load_ring(__amdgpu_buffer_rsrc_t rs_cur, uint32_t ring, int W, uint32_t records, uint32_t (&rows)[10][3])
{
    if ((uint64_t)ring + (uint64_t)(9 * W) + 16u <= (uint64_t)records) {
#pragma unroll
        for (int y = 0; y < 10; y++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, ring, y * W, 0);
            rows[y][0] = v.x; rows[y][1] = v.y; rows[y][2] = v.z & 0xFFFFu;
        }
    } else {
#pragma unroll
        for (int y = 0; y < 10; y++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_cur, ring, y * W, 0);
            rows[y][0] = v.x; rows[y][1] = v.y;
            rows[y][2] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_cur, ring + 8u, y * W, 0);
        }
    }
}

// One block: record (and direction) written to global memory and returned for the votes.
// Returns the half-pixel direction (8 = none).  PRUNE: the wave-uniform exact pruned search;
// start_row / prune_pays are the wave's hints carried from its previous chunk of blocks (every
// lane of the wave stays in the control flow until the search is over, so ballots see them all).
// EQ = false: the launch has no pixel sums (a.sums == nullptr): no equalisation code at all in the
// kernel (the exhaustive flat kernel is instantiated both ways; its code size and register
// allocation are what the headline configuration runs on).
// `given`: predictor and equalisation delta handed over by the caller (a kernel that has just
// computed them itself and must not read them back through the constant cache); a.pred and a.sums
// are then not read.
struct PairMeta { int px, py, delta; };

// PACKED: the predictor rides through the SAD stream as two bytes of ONE register.  The kernels that keep
// their lanes for the votes sit exactly at the 128 VGPRs of four waves per SIMD; without it two of their
// instantiations (no half-pixel step, equalising) spill one accumulator into scratch in the middle of the
// stream.  Only those two use it: the code hipcc emits for a kernel here depends on WHICH other kernels of
// the translation unit inline the same instantiation of this function (measured round 4: taking the flat
// vote kernel off search_block<false, false, false> changes the record epilogue of the plain C2 kernel,
// which shares it, and costs that kernel 2.5 %), so the headline kernel's instantiation keeps its callers.
template <bool SUBPIXEL, bool PRUNE, bool EQ = true, bool PACKED = false>
__device__ __forceinline__ int search_block(const SearchArgs &a, uint32_t pair, uint32_t blk, uint32_t item0, bool live,
                                            aof_block &rec, int &start_row, int &prune_pays,
                                            const PairMeta *given = nullptr)
{
    // item = item0 + threadIdx.x (item0 uniform in the workgroup): index of the block's record.  It is
    // derived again from the lane id wherever it is needed, so it holds no register during the search.
    auto record_slot = [&]() -> uint32_t {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return item0 + t;
    };
    // (blocks, grid coordinates and pixel offsets stay below 2^24 -- aof_params_check limits a frame to
    //  2^24 pixels --, so the products are full-rate 24-bit multiplies, not quarter-rate 32-bit ones)
    const uint32_t by = fast_div(blk, a.div_nx), bx = blk - __umul24(by, (uint32_t)a.grid.nx);
    const int i = a.grid.x0 + __mul24((int)bx, a.grid.step_x), j = a.grid.y0 + __mul24((int)by, a.grid.step_y);
    const int W = a.w;
    constexpr int m = SUBPIXEL ? 1 : 0;
    int px = 0, py = 0, delta = 0, delta_first = 0;
    // Almost every wave lies inside one pair.  Everything that depends on the pair alone is
    // computed for the wave's FIRST pair in scalar registers -- frame base addresses, predictor,
    // equalisation delta (scalar loads, served by the constant cache) -- and only the lanes of a
    // following pair correct it: a per-lane load here would be a whole memory round trip in front
    // of the 24 row loads, and a per-lane 64-bit base address costs two VALU per row.
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)pair);
    // Row loads through buffer resources: the descriptor (base of the wave's first pair) and the
    // row offset r*W live in scalar registers, the lane contributes ONE 32-bit byte offset -- no
    // VALU per row.  Reads past the last pair's frame return zero instead of faulting.
    const uint64_t span = a.n_pairs > 1 ? (uint64_t)(a.n_pairs - first) * (uint64_t)a.pair_stride
                                        : (uint64_t)a.w * (uint64_t)a.h;
    const uint32_t records = span > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)span;
    const int64_t base = (int64_t)first * a.pair_stride;
    const __amdgpu_buffer_rsrc_t rs_prev = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.prev) + base, 0, records, kRawBuffer);
    const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.cur) + base, 0, records, kRawBuffer);
    const uint32_t dp = (pair - first) * (uint32_t)a.pair_stride;   // lane8_supported: fits 32 bits
    const uint32_t off_prev = dp + (uint32_t)(__mul24(j, W) + i);
    // The reference tile does not depend on the predictor: its eight rows are requested before the
    // scalar loads of predictor and pixel sums have come back (the tile of a grid block always lies
    // inside the frame).
    uint32_t ref[8][2];
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            // (the pruned kernel is bound by memory, not by SAD issue: its reference tiles -- every byte used once -- are
            //  loaded non-temporally, so that they do not push the window rows the next chunk shares out of L2: -3.5 %)
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_prev, off_prev, r * W, PRUNE ? 2 : 0);
            ref[r][0] = v.x; ref[r][1] = v.y;
        }
    }
    typedef const __attribute__((address_space(4))) uint32_t *const_u32;  // read-only during the kernel
    const uint32_t npix = (uint32_t)(a.w * a.h);
    if (EQ && a.sums && !given) {   // (outside the lane-dependent branch: stays in scalar registers)
        const const_u32 sm = (const_u32)(a.sums + (size_t)first * 4);
        delta_first = (int)((sm[a.level] + npix / 2) / npix) - (int)((sm[2 + a.level] + npix / 2) / npix);
    }
    if (given) {
        if (live) { px = given->px; py = given->py; delta = given->delta; }
    } else if (live) {
        if (a.pred) {
            const uint32_t w3 = ((const_u32)(a.pred + first))[3];  // quality, flags, pred_x, pred_y
            px = (int8_t)(w3 >> 16); py = (int8_t)(w3 >> 24);
            if (pair != first) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
        }
        if (EQ && a.sums) {
            delta = delta_first;
            if (pair != first) delta = equalise_delta(a.sums, pair, a.level, npix);
        }
    }
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;
    uint32_t *const slots = reinterpret_cast<uint32_t *>(a.blocks);  // one dword store per record
    // the search window (plus the half-pixel ring) must lie inside the frame
    const int wx0 = i + px - 4, wy0 = j + py - 4;
    const bool inside = live && !(wx0 - m < 0 || wy0 - m < 0 || wx0 + 16 + m > a.w || wy0 + 16 + m > a.h);
    if (!PRUNE && !inside) {
        if (live) {
            const uint32_t item = record_slot();
            slots[item] = __builtin_bit_cast(uint32_t, rec);
            if (SUBPIXEL) a.subdirs[item] = 8;
        }
        return 8;
    }
    const uint32_t off_cur = dp + (uint32_t)(__mul24(wy0, W) + wx0);

    uint4 win[16];
    if (inside) {
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, off_cur, s * W, 0);
            win[s] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
    // ONE memory round trip: left alone, the compiler sinks the window loads (and half of the tile
    // loads) below the gate's branch, so a wave first waits for four tile rows, gates, and only
    // then requests the other twenty rows.  Naming the last window row here keeps all 24 loads
    // in front of the gate (they return in order: one wait for everything).
    if (inside) asm volatile("" : "+v"(win[15].w));
    uint32_t gradient = 0;
    if (inside) gradient = gradient_gate(ref);
    const bool need = inside && gradient >= (uint32_t)a.feature_threshold;
    if (!PRUNE && !need) {
        const uint32_t item = record_slot();
        slots[item] = __builtin_bit_cast(uint32_t, rec);
        if (SUBPIXEL) a.subdirs[item] = 8;
        return 8;
    }
    if (PRUNE && need && delta != 0) {  // the pruned rows read a window row several times
#pragma unroll
        for (int s = 0; s < 16; s++) win[s] = sat_add_u8x16(win[s], delta);
    }
    if constexpr (PACKED) {
        px = (int)(((uint32_t)px & 0xFFu) | (((uint32_t)py & 0xFFu) << 8));
        asm volatile("" : "+v"(px));   // (opaque: one register from here to the end of the SAD stream)
    }
    uint32_t best = 0xFFFFFFFFu;
    if constexpr (PRUNE) {
        const unsigned long long needing = __ballot(need);
        if (needing != 0) {
            if (__builtin_amdgcn_readfirstlane(prune_pays) == 0) {
                // noise-dominated images: the previous chunk of this wave could drop (almost)
                // nothing, and the exhaustive code is the faster way to evaluate everything
                // ADAPTIVE: the wave's FIRST chunk comes here too (a pruned chunk that starts in the wrong row or drops
                // nothing costs 1.4x the exhaustive one).  Every exhaustive chunk judges from its own SADs whether
                // the next one should prune, and where it should start.
                int droppable = 0;
                best = exhaustive_search_judged(win, ref, needing, droppable);
                const int src = __ffsll((long long)needing) - 1;
                start_row = (int)((uint32_t)__shfl((int)best, src, 64) & 0xFFFFu) / 9;
                prune_pays = droppable >= kJudgedRowsToPrune;
            } else {
                int start = __builtin_amdgcn_readfirstlane(start_row);
                if (start < 0) start = __builtin_amdgcn_readfirstlane(vote_start_row(win, ref, needing));   // (nothing known yet)
                const int dropped = pruned_search(win, ref, needing, start, best);
                // the wave's next chunk starts where its first live block matched
                const int src = __ffsll((long long)needing) - 1;
                start_row = (int)((uint32_t)__shfl((int)best, src, 64) & 0xFFFFu) / 9;
                prune_pays = dropped >= 2;
            }
        }
        if (!need) {
            if (live) {
                const uint32_t item = record_slot();
                slots[item] = __builtin_bit_cast(uint32_t, rec);
                if (SUBPIXEL) a.subdirs[item] = 8;
            }
            return 8;
        }
    } else {
        best = exhaustive_search<EQ>(win, ref, delta);
    }
    const int idx = (int)(best & 0xFFFFu);
    if constexpr (PACKED) {
        py = (int8_t)((uint32_t)px >> 8);
        px = (int8_t)((uint32_t)px & 0xFFu);
    }
    rec.dx = (int8_t)(px + idx % 9 - 4);
    rec.dy = (int8_t)(py + idx / 9 - 4);
    rec.sad = (uint16_t)(best >> 16);
    const uint32_t item = record_slot();
    slots[item] = __builtin_bit_cast(uint32_t, rec);

    // Half-pixel refinement of accepted blocks: the ring of the best match, rows -1..8 and
    // bytes -1..8, again straight from global memory (the lines were touched a moment ago).
    int subdir = 8;
    if constexpr (SUBPIXEL) {
        if ((uint32_t)rec.sad < (uint32_t)a.value_threshold) {
            const uint32_t ring = off_cur + (uint32_t)((idx / 9 - 1) * W + (idx % 9 - 1));
            uint32_t rows[10][3];
            if constexpr (PRUNE) {
                // The pruned kernel is latency-bound where it prunes, and a second round of row loads per chunk doubles
                // what the refinement costs it (135 against 66 us per 1 024 VGA pairs).  A best match that is not on the
                // window's rim has its whole ring IN the window registers: the wave's refining lanes that share the first
                // one's dy (under a global motion: all of them) cut their ten rows out of them -- the row index is then
                // wave-uniform, i.e. compile-time inside a seven-way switch, and the byte offset is a per-lane
                // v_alignbyte --; the window is equalised already.  The other lanes load their ring.
                const int dyi = idx / 9, dxi = idx - 9 * dyi;
                const unsigned long long refining = __ballot(true);
                const int udy = __builtin_amdgcn_readlane(dyi, __ffsll((long long)refining) - 1);
                const bool have_ring = dyi == udy && udy >= 1 && udy <= 7 && dxi >= 1 && dxi <= 7;
                RefineState<2> st;
                st.init();
                if (have_ring) refine_from_window(win, udy, dxi - 1, ref, st, (uint32_t)rec.sad);
                if (!have_ring && rec.sad != 0) {   // (nothing is below a SAD of zero)
                    load_ring(rs_cur, ring, W, records, rows);
                    for_rows<-1, 8>([&](auto yc) {
                        constexpr int Y = decltype(yc)::value;
                        uint32_t d[3] = {rows[Y + 1][0], rows[Y + 1][1], rows[Y + 1][2]};
                        if (delta != 0) {
#pragma unroll
                            for (int q = 0; q < 3; q++) d[q] = sat_add_u8x4(d[q], delta);
                        }
                        st.template row<Y>(d, ref);
                    });
                }
                subdir = st.direction(rec.sad);
            } else if (rec.sad != 0) {   // (nothing is below a SAD of zero: no ring is loaded for such a match)
            load_ring(rs_cur, ring, W, records, rows);
            RefineState<2> st;
            st.init();
            bool over = false;   // (uniform among the lanes in here: RefineState::nobody_can_win)
            for_rows<-1, 8>([&](auto yc) {
                constexpr int Y = decltype(yc)::value;
                if (over) return;
                uint32_t d[3] = {rows[Y + 1][0], rows[Y + 1][1], rows[Y + 1][2]};
                if (delta != 0) {
#pragma unroll
                    for (int q = 0; q < 3; q++) d[q] = sat_add_u8x4(d[q], delta);
                }
                st.template row<Y>(d, ref);
                if constexpr (Y == 1 || Y == 3 || Y == 5) over = st.nobody_can_win((uint32_t)rec.sad);
            });
            subdir = st.direction(rec.sad);
            }
        }
        a.subdirs[item] = (uint8_t)subdir;
    }
    return subdir;
}

}  // namespace

}  // namespace aof
