// K3 -- histogram-filtered flow reduction (DESIGN.md "Spec": Reduce).
//
// One workgroup per frame pair.  The block records (4 B each, just written by
// K2 and still in L2) are read coalesced; accepted blocks vote into two
// half-pixel shift histograms held in LDS (integer LDS atomics, so the result
// does not depend on arrival order).  Neighbouring blocks mostly vote for the
// SAME bin, which would serialise a plain per-lane LDS atomic 64 ways, so votes
// are first aggregated across the wave (ballot of equal bins, one atomic per
// distinct bin).  Lane 0 then applies the published
// first-maximum peak search and the +-2-bin weighted mean, or the plain
// average, and writes the 16-byte aof_flow.  The float arithmetic is a handful
// of exactly-representable integers and two correctly-rounded divisions
// (__fdiv_rn), so it is bit-identical to the host arithmetic of the oracle.
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_reduce.hpp"

namespace aof {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxHist = 256;
constexpr int64_t kMidPairs = 1024;   // up to here a pair gets 512 lanes
constexpr int64_t kWidePairs = 512;   // up to here a pair gets 1 024 lanes instead of 256 (1 024 pairs: 18.3 against 13.4 us)

// GROUP = threads per pair: 256 (one workgroup per pair) or 64 (one wave per pair, four pairs
// per workgroup: sparse grids with a few dozen blocks per pair, where a whole workgroup per
// pair is mostly launch overhead).  A wave-sized group needs no workgroup barrier: its LDS
// operations retire in program order.
template <int GROUP, int kThreads = 256>
__global__ __launch_bounds__(kThreads) void k_reduce(ReduceArgs a)
{
    constexpr int kGroups = kThreads / GROUP;
    __shared__ uint32_t s_hist[kGroups][2][kMaxHist];
    __shared__ int s_sums[kGroups][3];  // sum2x, sum2y, count
    const int grp = (int)threadIdx.x / GROUP, tid = (int)threadIdx.x % GROUP;
    const int64_t pair = (int64_t)blockIdx.x * kGroups + grp;
    if (GROUP < kThreads && pair >= a.n_pairs) return;  // whole waves only: no barrier is skipped
    auto sync = [] { if (GROUP == kThreads) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    uint32_t (*hist)[kMaxHist] = s_hist[grp];
    int *sums = s_sums[grp];
    const int centre = 2 * a.tail.range + 1, n = 2 * centre + 1;
    for (int k = tid; k < n; k += GROUP) { hist[0][k] = 0; hist[1][k] = 0; }
    if (tid < 3) sums[tid] = 0;
    sync();

    if (a.parts) {
        // A first step (k_reduce_chunk) already voted per chunk of records: sum the chunks' histograms.  Count and
        // shift sums follow from the histogram: count = sum h[k], sum2 = sum (k - centre) h[k].
        const uint32_t *parts = a.parts + (size_t)pair * a.nstrips * (size_t)(2 * n);
        for (int k = tid; k < 2 * n; k += GROUP) {
            uint32_t v = 0;
            for (int st = 0; st < a.nstrips; st++) v += parts[(size_t)st * (2 * n) + k];
            const int axis = k >= n, bin = k - axis * n;
            hist[axis][bin] = v;
            if (v) {
                atomicAdd(&sums[axis], (bin - centre) * (int)v);
                if (!axis) atomicAdd(&sums[2], (int)v);
            }
        }
        sync();
    } else {
    const aof_block *blocks = a.blocks + pair * a.tail.nblocks;
    const uint8_t *subdirs = a.subdirs ? a.subdirs + pair * a.tail.nblocks : nullptr;
    int s2x = 0, s2y = 0, cnt = 0;
    auto vote = [&](bool ok, aof_block r, int sd) {
        ok = ok && !(r.sad == AOF_SAD_SKIPPED || (int)r.sad >= a.value_threshold);
        int hx = 0, hy = 0;
        if (ok && subdirs) {
            hx = (sd == 0 || sd == 1 || sd == 7) ? 1 : ((sd == 3 || sd == 4 || sd == 5) ? -1 : 0);
            hy = (sd == 1 || sd == 2 || sd == 3) ? 1 : ((sd == 5 || sd == 6 || sd == 7) ? -1 : 0);
        }
        const int vx = 2 * r.dx + hx, vy = 2 * r.dy + hy;
        wave_vote2(hist[0], hist[1], vx + centre, vy + centre, ok);
        if (ok) { s2x += vx; s2y += vy; cnt++; }
    };
    // Records four at a time: ONE 16-byte load per lane and quad (dword-aligned: a pair's records start at any
    // multiple of four bytes), and where the valid records of a quad agree -- under a global motion they do almost
    // everywhere -- ONE weighted vote for all of them.  A wave in which some quad disagrees votes that round record
    // by record.  (1 024 VGA pairs: a lane of the 512 casts 3 votes instead of 10; the kernel is issue-bound there.)
    typedef uint32_t u32_bytes __attribute__((aligned(1)));
    const uint32_t *words = reinterpret_cast<const uint32_t *>(blocks);
    const int quads = a.tail.nblocks / 4;
    const int qrounds = (quads + GROUP - 1) / GROUP;   // uniform trip count: ballots need every lane
    auto vote_quad = [&](bool have, const uint32_t (&w)[4], uint32_t sd4) {
        int key[4], weight = 0, first = 0;
        bool ok[4], messy = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t sad = w[j] >> 16;
            ok[j] = have && !(sad == AOF_SAD_SKIPPED || (int)sad >= a.value_threshold);
            const int sd = (int)((sd4 >> (8 * j)) & 0xFFu);
            int hx = 0, hy = 0;
            if (subdirs) {
                hx = (sd == 0 || sd == 1 || sd == 7) ? 1 : ((sd == 3 || sd == 4 || sd == 5) ? -1 : 0);
                hy = (sd == 1 || sd == 2 || sd == 3) ? 1 : ((sd == 5 || sd == 6 || sd == 7) ? -1 : 0);
            }
            const int vx = 2 * (int)(int8_t)(w[j] & 0xFFu) + hx, vy = 2 * (int)(int8_t)((w[j] >> 8) & 0xFFu) + hy;
            key[j] = (vx + centre) | ((vy + centre) << 16);
            if (ok[j]) {
                s2x += vx; s2y += vy; cnt++;
                if (weight == 0) first = key[j];
                else if (key[j] != first) messy = true;
                weight++;
            }
        }
        if (__ballot(messy) == 0) {   // (wave-uniform)
            wave_vote2_weighted(hist[0], hist[1], first, weight);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) wave_vote2(hist[0], hist[1], key[j] & 0xFFFF, key[j] >> 16, ok[j]);
        }
    };
    constexpr int kBatchQ = 3;   // quad loads in flight per lane
    for (int it0 = 0; it0 < qrounds; it0 += kBatchQ) {
        uint32_t rec[kBatchQ][4], sdir[kBatchQ];
#pragma unroll
        for (int k = 0; k < kBatchQ; k++) {
            const int q = (it0 + k) * GROUP + tid;
#pragma unroll
            for (int j = 0; j < 4; j++) rec[k][j] = 0xFFFF0000u;
            sdir[k] = 0x08080808u;
            if (it0 + k < qrounds && q < quads) {
                __builtin_memcpy(rec[k], words + 4 * q, 16);
                if (subdirs) sdir[k] = *reinterpret_cast<const u32_bytes *>(subdirs + 4 * q);
            }
        }
#pragma unroll
        for (int k = 0; k < kBatchQ; k++) {
            if (it0 + k >= qrounds) break;  // uniform
            const int q = (it0 + k) * GROUP + tid;
            vote_quad(q < quads, rec[k], sdir[k]);
        }
    }
    {   // the up to three records behind the last whole quad
        const int b = 4 * quads + tid;
        const bool in = b < a.tail.nblocks;
        aof_block r;
        r.dx = 0; r.dy = 0; r.sad = AOF_SAD_SKIPPED;
        if (in) r = blocks[b];
        vote(in, r, in && subdirs ? subdirs[b] : 8);
    }
    s2x = (int)wave_sum_u32((uint32_t)s2x);
    s2y = (int)wave_sum_u32((uint32_t)s2y);
    cnt = (int)wave_sum_u32((uint32_t)cnt);
    if ((tid & 63) == 0) {
        atomicAdd(&sums[0], s2x);
        atomicAdd(&sums[1], s2y);
        atomicAdd(&sums[2], cnt);
    }
    sync();
    }
    if (n <= 64) {   // (uniform) one wave of the group: the bins side by side
        if (tid < 64) finalise_flow_wave(a.tail, pair, hist[0], hist[1], sums);
    } else if (tid == 0) {
        finalise_flow(a.tail, pair, hist[0], hist[1], sums);
    }
}

// Large grids (a 4K frame has 128 000 blocks): one workgroup per chunk of 4 096 records votes
// into its own histogram; k_reduce then sums the chunks' histograms.
__global__ __launch_bounds__(kThreads) void k_reduce_chunk(ReduceArgs a, int chunks)
{
    __shared__ uint32_t hist[2][kMaxHist];
    const int chunk = (int)(blockIdx.x % (uint32_t)chunks);
    const int64_t pair = (int64_t)(blockIdx.x / (uint32_t)chunks);
    const int tid = threadIdx.x;
    const int centre = 2 * a.tail.range + 1, n = 2 * centre + 1;
    for (int k = tid; k < n; k += kThreads) { hist[0][k] = 0; hist[1][k] = 0; }
    __syncthreads();
    const int per = (a.tail.nblocks + chunks - 1) / chunks;
    const int b0 = chunk * per, b1 = min(a.tail.nblocks, b0 + per);
    const uint32_t *blocks = reinterpret_cast<const uint32_t *>(a.blocks) + pair * a.tail.nblocks;
    const uint8_t *subdirs = a.subdirs ? a.subdirs + pair * a.tail.nblocks : nullptr;
    for (int base = b0; base < b1; base += kThreads) {  // uniform trip count: ballots need every lane
        const int b = base + tid;
        const bool in = b < b1;
        const aof_block r = __builtin_bit_cast(aof_block, in ? blocks[b] : 0xFFFF0000u);
        const int sd = in && subdirs ? subdirs[b] : 8;
        const bool ok = in && !(r.sad == AOF_SAD_SKIPPED || (int)r.sad >= a.value_threshold);
        int hx = 0, hy = 0;
        if (ok && subdirs) {
            hx = (sd == 0 || sd == 1 || sd == 7) ? 1 : ((sd == 3 || sd == 4 || sd == 5) ? -1 : 0);
            hy = (sd == 1 || sd == 2 || sd == 3) ? 1 : ((sd == 5 || sd == 6 || sd == 7) ? -1 : 0);
        }
        wave_vote(hist[0], 2 * r.dx + hx + centre, ok);
        wave_vote(hist[1], 2 * r.dy + hy + centre, ok);
    }
    __syncthreads();
    uint32_t *out = a.chunk_parts + ((size_t)pair * chunks + chunk) * (size_t)(2 * n);
    for (int k = tid; k < 2 * n; k += kThreads) out[k] = k < n ? hist[0][k] : hist[1][k - n];
}

}  // namespace

int launch_reduce(const ReduceArgs &a, void *stream)
{
    if (a.n_pairs == 0) return 0;
    if (a.n_pairs > 0x7FFFFFFF) return (int)hipErrorInvalidValue;   // one workgroup (or wave) per pair
    const int chunks = a.parts ? 0 : reduce_chunks(a.tail.nblocks);
    if (chunks > 0 && a.chunk_parts && a.n_pairs * chunks <= 0x7FFFFFFF) {
        hipLaunchKernelGGL(k_reduce_chunk, dim3((uint32_t)(a.n_pairs * chunks)), dim3(kThreads), 0,
                           static_cast<hipStream_t>(stream), a, chunks);
        ReduceArgs b = a;
        b.parts = a.chunk_parts;
        b.nstrips = chunks;
        hipLaunchKernelGGL(k_reduce<kThreads>, dim3((uint32_t)a.n_pairs), dim3(kThreads), 0,
                           static_cast<hipStream_t>(stream), b);
        return (int)hipGetLastError();
    }
    if (a.tail.nblocks <= 256 && !a.parts)  // sparse grids: one wave per pair
        hipLaunchKernelGGL(k_reduce<64>, dim3((uint32_t)((a.n_pairs + 3) / 4)), dim3(kThreads), 0,
                           static_cast<hipStream_t>(stream), a);
    else if (a.n_pairs <= kWidePairs && a.tail.nblocks >= 2048 && !a.parts)
        // few pairs of many blocks (configs[3]'s per-GPU share: 128 VGA pairs on 256 CUs): the launch is
        // latency-bound -- sixteen waves per pair fetch its records in ONE round trip
        hipLaunchKernelGGL((k_reduce<1024, 1024>), dim3((uint32_t)a.n_pairs), dim3(1024), 0,
                           static_cast<hipStream_t>(stream), a);
    else if (a.n_pairs <= kMidPairs && a.tail.nblocks >= 2048 && !a.parts)
        // up to 1 024 such pairs eight waves per pair still fit the device at once (8 192 wave slots):
        // half the rounds of record loads and votes per lane (1 024 pairs: 10.9 against 12.0 us)
        hipLaunchKernelGGL((k_reduce<512, 512>), dim3((uint32_t)a.n_pairs), dim3(512), 0,
                           static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(k_reduce<kThreads>, dim3((uint32_t)a.n_pairs), dim3(kThreads), 0,
                           static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // namespace aof
