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
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

constexpr int kThreads = 512;
constexpr int kSide = 17;  // 2S+1

__device__ __forceinline__ u64 qsad(u64 window, uint32_t ref, u64 acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(window, ref, acc);
}
__device__ __forceinline__ u64 pack64(uint32_t lo, uint32_t hi) { return ((u64)hi << 32) | lo; }

__global__ __launch_bounds__(kThreads) void k_search_tile16(SearchArgs a, uint32_t total_wgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t logical = xcd_remap(blockIdx.x, total_wgs);
    const int W = a.w, nx = a.grid.nx, ny = a.grid.ny;
    const int by = (int)(logical % (uint32_t)ny);
    const int64_t pair = (int64_t)(logical / (uint32_t)ny);
    const int tid = threadIdx.x;
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(W * a.h));

    // Level 0 under a predictor (px, py): the block row's windows move by py rows and px columns.
    // As in k_search_tile8 the cur rows are staged PRE-SHIFTED by px mod 16, so that LDS column c
    // holds frame column c + sh and every window is again 16-byte aligned at LDS column
    // 16*(bx + (px >> 4)); rows pushed outside the frame skip the whole block row.
    int px = 0, py = 0;
    if (a.pred) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
    const int sh = px & 15;                                       // floor-mod
    uint8_t *s_cur = smem;                           // frame rows [16*by + py, +32)
    uint8_t *s_prev = smem + (size_t)32 * W;         // frame rows [16*by + 8, +16)
    uint32_t *s_best = reinterpret_cast<uint32_t *>(smem + (size_t)48 * W);
    // Half-pixel refinement moves the grid origin to S+1 = 9: the same geometry on a frame whose
    // origin is moved by (1, 1) -- the flat copies start W+1 bytes later (byte-aligned loads);
    // the grid keeps every window inside the smaller frame.  K2b adds the directions.
    const int org = a.grid.x0 - 8;
    const int H = a.h - 2 * org, Wb = W - 2 * org;               // the moved frame
    const int64_t org_off = (int64_t)org * (W + 1);
    const int yc0 = 16 * by + py;                                 // moved-frame row of LDS cur row 0
    const bool rows_ok = yc0 >= 0 && yc0 + 32 <= H;               // wave-uniform: the whole block row
    const uint8_t *g_cur = a.cur + pair * a.pair_stride + org_off + (int64_t)yc0 * W + sh;
    const uint8_t *g_prev = a.prev + pair * a.pair_stride + org_off + (int64_t)(16 * by + 8) * W;
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
    if (org != 0 || sh != 0) {  // byte-aligned source: through registers
        for (int c = tid; c < cur_chunks + prev_chunks; c += kThreads) {
            const bool is_cur = c < cur_chunks;
            const int cc = is_cur ? c : c - cur_chunks;
            uint4 v;
            __builtin_memcpy(&v, (is_cur ? g_cur : g_prev) + (size_t)cc * 16, 16);
            if (is_cur && delta != 0) {
                v = sat_add_u8x16(v, delta);
            }
            *reinterpret_cast<uint4 *>((is_cur ? s_cur : s_prev) + (size_t)cc * 16) = v;
        }
    } else if (delta == 0) {
        for (int c = tid; c < cur_chunks; c += kThreads)
            *reinterpret_cast<uint4 *>(s_cur + (size_t)c * 16) =
                *reinterpret_cast<const uint4 *>(g_cur + (size_t)c * 16);
    } else {
        for (int c = tid; c < cur_chunks; c += kThreads) {
            uint4 v = *reinterpret_cast<const uint4 *>(g_cur + (size_t)c * 16);
            v = sat_add_u8x16(v, delta);
            *reinterpret_cast<uint4 *>(s_cur + (size_t)c * 16) = v;
        }
    }
    if (org == 0 && sh == 0)
        for (int c = tid; c < prev_chunks; c += kThreads)
            *reinterpret_cast<uint4 *>(s_prev + (size_t)c * 16) =
                *reinterpret_cast<const uint4 *>(g_prev + (size_t)c * 16);
    for (int b = tid; b < nx; b += kThreads) s_best[b] = 0xFFFFFFFFu;
    __syncthreads();

    const int items = rows_ok ? kSide * nx : 0;
    for (int item = tid; item < items; item += kThreads) {
        const int dyi = item / nx, bx = item - dyi * nx;
        const int xf = 16 * bx + px;                  // moved-frame column of the window start
        if (xf < 0 || xf + 32 > Wb) continue;         // window leaves the frame: block skipped below
        const int xs = xf - sh;                       // its 16-aligned LDS column
        // reference tile: 16 rows x 4 dwords at frame column 16*bx + 8
        uint32_t ref[16][4];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint2 *p = reinterpret_cast<const uint2 *>(s_prev + (size_t)r * W + 16 * bx + 8);
            const uint2 lo = p[0], hi = p[1];
            ref[r][0] = lo.x; ref[r][1] = lo.y; ref[r][2] = hi.x; ref[r][3] = hi.y;
        }
        u64 acc[4] = {0, 0, 0, 0};                    // offsets 4g .. 4g+3, packed u16
        uint32_t acc16 = (uint32_t)(dyi * kSide + 16);  // offset 16 as sad<<16 | idx
        const uint8_t *win = s_cur + (size_t)dyi * W + xs;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint4 *p = reinterpret_cast<const uint4 *>(win + (size_t)r * W);
            const uint4 q0 = p[0], q1 = p[1];
            const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            u64 pr[7];
#pragma unroll
            for (int j = 0; j < 7; j++) pr[j] = pack64(w[j], w[j + 1]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int g = 0; g < 4; g++) acc[g] = qsad(pr[g + k], ref[r][k], acc[g]);
                acc16 = __builtin_amdgcn_sad_hi_u8(w[4 + k], ref[r][k], acc16);
            }
        }
        uint32_t best = acc16;
        const uint32_t base = (uint32_t)(dyi * kSide);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t l = (uint32_t)acc[g], h = (uint32_t)(acc[g] >> 32);
            const uint32_t k0 = (l << 16) | (base + 4 * g + 0), k1 = (l & 0xFFFF0000u) | (base + 4 * g + 1);
            const uint32_t k2 = (h << 16) | (base + 4 * g + 2), k3 = (h & 0xFFFF0000u) | (base + 4 * g + 3);
            best = min(best, min(min(k0, k1), min(k2, k3)));
        }
        atomicMin(&s_best[bx], best);
    }
    __syncthreads();

    // one lane per block: 4x4 gradient gate (tile bytes 6..9, rows 6..9) and the record
    for (int bx = tid; bx < nx; bx += kThreads) {
        uint32_t mid[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t *p =
                reinterpret_cast<const uint32_t *>(s_prev + (size_t)(6 + r) * W + 16 * bx + 8 + 4);
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
        if (rows_ok && xf >= 0 && xf + 32 <= Wb && diff >= (uint32_t)a.feature_threshold) {
            const uint32_t best = s_best[bx];
            const int idx = (int)(best & 0xFFFFu);
            rec.dx = (int8_t)(px + idx % kSide - 8);
            rec.dy = (int8_t)(py + idx / kSide - 8);
            rec.sad = (uint16_t)(best >> 16);
        }
        a.blocks[pair * (int64_t)(nx * ny) + (int64_t)by * nx + bx] = rec;
    }
}

size_t tile16_lds(const SearchArgs &a) { return (size_t)48 * a.w + 4 * (size_t)a.grid.nx + 16; }

}  // namespace

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
    const size_t lds = tile16_lds(a);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_search_tile16),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_search_tile16, dim3((uint32_t)total), dim3(kThreads), lds,
                       static_cast<hipStream_t>(stream), a, (uint32_t)total);
    return (int)hipGetLastError();
}

}  // namespace aof
