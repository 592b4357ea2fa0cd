// K2 (generic) -- SAD block search for any tile/search/grid/predictor setting,
// including the published sparse PX4Flow grid with half-pixel refinement
// (DESIGN.md "Spec": Search).
//
// One 64-lane wavefront per block.  The wave stages the BxB reference tile and
// the (B+2S+2m)^2 search window in LDS, then every lane owns candidates
// idx = lane, lane+64, ... of the (2S+1)^2 scan; the winner is the wave-wide
// minimum of the packed key (sad << 16 | idx), which reproduces "first minimum
// in scan order wins" exactly.  A candidate row is read as aligned LDS dwords,
// realigned with v_alignbyte_b32 by the lane's own byte phase and summed with
// v_sad_u8 (4 pixels per instruction).  The half-pixel refinement spreads its
// 8 directions x B rows over the wave and adds the rows with three shuffles.
// This kernel favours generality; the 8x8/+-4 configurations the metric is quoted on
// run k_search_lane8 instead, 16x16/+-8 runs k_search_tile16.
#include "aof_device.hpp"
#include "aof_internal.hpp"

namespace aof {

namespace {

constexpr int kMaxTile = 16, kMaxSearch = 8;
constexpr int kMaxWin = kMaxTile + 2 * kMaxSearch + 2;  // with the half-pixel margin
constexpr int kMaxPitch = (kMaxWin + 3) / 4 * 4;          // window rows start on a dword

__global__ __launch_bounds__(64) void k_search_generic(SearchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ref[kMaxTile * kMaxTile];
    __shared__ __attribute__((aligned(16))) uint8_t s_win[kMaxPitch * kMaxWin + 16];  // + dword over-read

    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    const int64_t pair = blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
    if (pair >= a.n_pairs) return;
    const int B = a.tile, S = a.search, m = a.subpixel ? 1 : 0;
    const int bx = blk % a.grid.nx, by = blk / a.grid.nx;
    const int i = a.grid.x0 + bx * a.grid.step_x, j = a.grid.y0 + by * a.grid.step_y;
    int px = 0, py = 0;
    if (a.pred) { px = a.pred[pair].pred_x; py = a.pred[pair].pred_y; }
    const int delta = equalise_delta(a.sums, pair, a.level, (uint32_t)(a.w * a.h));

    aof_block *out = a.blocks + pair * a.grid.blocks() + blk;
    uint8_t *out_sd = a.subdirs ? a.subdirs + pair * a.grid.blocks() + blk : nullptr;
    aof_block rec;
    rec.dx = 0; rec.dy = 0; rec.sad = AOF_SAD_SKIPPED;

    // Window of the search (plus the half-pixel ring) must lie inside the frame.
    const int wx0 = i + px - S - m, wy0 = j + py - S - m;
    const int win = B + 2 * S + 2 * m;
    const int pitch = (win + 3) & ~3;             // LDS row pitch of the window
    const bool inside = wx0 >= 0 && wy0 >= 0 && wx0 + win <= a.w && wy0 + win <= a.h;
    if (!inside) {  // wave-uniform
        if (lane == 0) { *out = rec; if (out_sd) *out_sd = 8; }
        return;
    }

    const uint8_t *prev = a.prev + pair * a.pair_stride;
    const uint8_t *cur = a.cur + pair * a.pair_stride;
    for (int t = lane; t < B * B; t += 64)
        s_ref[t] = prev[(int64_t)(j + t / B) * a.w + i + t % B];
    for (int t = lane; t < win * win; t += 64) {
        const int y = t / win, x = t - y * win;
        const int v = cur[(int64_t)(wy0 + y) * a.w + wx0 + x];
        s_win[y * pitch + x] = (uint8_t)clamp_u8(v + delta);
    }
    __syncthreads();

    // 4x4 gradient gate: 12 vertical + 12 horizontal neighbour differences.
    {
        const int off = B / 2 - 2;
        uint32_t d = 0;
        if (lane < 12) {
            const int r = lane / 4, c = lane % 4;
            d = (uint32_t)abs((int)s_ref[(off + r) * B + off + c] - (int)s_ref[(off + r + 1) * B + off + c]);
        } else if (lane < 24) {
            const int c = (lane - 12) / 4, r = (lane - 12) % 4;
            d = (uint32_t)abs((int)s_ref[(off + r) * B + off + c] - (int)s_ref[(off + r) * B + off + c + 1]);
        }
        d = wave_sum_u32(d);
        if (d < (uint32_t)a.feature_threshold) {
            if (lane == 0) { *out = rec; if (out_sd) *out_sd = 8; }
            return;
        }
    }

    // Exhaustive search; candidate idx = (jj+S)*(2S+1) + (ii+S).
    const int side = 2 * S + 1, ncand = side * side;
    uint32_t best = 0xFFFFFFFFu;
    for (int idx = lane; idx < ncand; idx += 64) {
        const int cy = idx / side + m, cx = idx % side + m;  // window coords of the candidate
        uint32_t sad = 0;
        const int shift = cx & 3;                 // this candidate's byte phase (rows share it)
        for (int r = 0; r < B; r++) {
            const uint32_t *wrow = reinterpret_cast<const uint32_t *>(s_win + (cy + r) * pitch + (cx & ~3));
            const uint32_t *rrow = reinterpret_cast<const uint32_t *>(s_ref + r * B);
            uint32_t lo = wrow[0];
            for (int q = 0; q < B / 4; q++) {
                const uint32_t hi = wrow[q + 1];
                sad = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, (uint32_t)shift), rrow[q], sad);
                lo = hi;
            }
        }
        const uint32_t key = (sad << 16) | (uint32_t)idx;
        best = key < best ? key : best;
    }
    best = wave_min_u32(best);
    const uint32_t dist = best >> 16;
    const int bidx = (int)(best & 0xFFFFu);
    const int sumx = bidx % side - S, sumy = bidx / side - S;
    rec.dx = (int8_t)(px + sumx);
    rec.dy = (int8_t)(py + sumy);
    rec.sad = (uint16_t)dist;

    // Half-pixel refinement of accepted blocks.
    int subdir = 8;
    if (a.subpixel && dist < (uint32_t)a.value_threshold) {
        // lane -> (direction k = lane & 7, row group lane >> 3): each lane sums its rows of its
        // direction, then the eight row groups of a direction are added by xor-shuffles 8/16/32
        const int k = lane & 7;
        const int ox = sumx + S + m, oy = sumy + S + m;  // best match, window coords
        uint32_t acc = 0;
        for (int r = lane >> 3; r < B; r += 8)
            for (int c = 0; c < B; c++) {
                const uint8_t *q = &s_win[(oy + r) * pitch + ox + c];
                const int p00 = q[0];
                int v;
                switch (k) {
                case 0: v = (p00 + q[1]) >> 1; break;
                case 2: v = (p00 + q[pitch]) >> 1; break;
                case 4: v = (p00 + q[-1]) >> 1; break;
                case 6: v = (p00 + q[-pitch]) >> 1; break;
                case 1: v = (((p00 + q[1]) >> 1) + ((q[pitch] + q[pitch + 1]) >> 1)) >> 1; break;
                case 3: v = (((q[pitch] + q[pitch - 1]) >> 1) + ((p00 + q[-1]) >> 1)) >> 1; break;
                case 5: v = (((p00 + q[-1]) >> 1) + ((q[-pitch] + q[-pitch - 1]) >> 1)) >> 1; break;
                default: v = (((q[-pitch] + q[-pitch + 1]) >> 1) + ((p00 + q[1]) >> 1)) >> 1; break;
                }
                acc += (uint32_t)abs((int)s_ref[r * B + c] - v);
            }
        acc += (uint32_t)__shfl_xor((int)acc, 8, 64);
        acc += (uint32_t)__shfl_xor((int)acc, 16, 64);
        acc += (uint32_t)__shfl_xor((int)acc, 32, 64);
        uint32_t mind = dist;
        for (int dir = 0; dir < 8; dir++) {
            const uint32_t v = (uint32_t)__shfl((int)acc, dir, 64);
            if (v < mind) { mind = v; subdir = dir; }
        }
    }
    if (lane == 0) {
        *out = rec;
        if (out_sd) *out_sd = (uint8_t)subdir;
    }
}

}  // namespace

int launch_search_generic(const SearchArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    // pairs split over grid.y/z to stay inside the 65535 limit of those dimensions
    const int64_t gy = a.n_pairs < 32768 ? a.n_pairs : 32768;
    const int64_t gz = (a.n_pairs + gy - 1) / gy;
    hipLaunchKernelGGL(k_search_generic, dim3((uint32_t)a.grid.blocks(), (uint32_t)gy, (uint32_t)gz),
                       dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // namespace aof
