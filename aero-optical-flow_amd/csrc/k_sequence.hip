// The output side of a frame SEQUENCE on the device (aof_sequence_device, include/aof.h): what the
// reference's per-frame loop does with calcFlow's pixel flows, for all frames of a recorded sequence at once:
//   * the rate limiter of the facade (facade/src/optical_flow.cpp limitRate, the calcFlow contract of
//     /root/reference/src/mainloop.cpp:322-331): flows of frames with quality > 0 are summed until
//     (float)(t - t_last) > 1e6f / output_rate (u32 wrap-around arithmetic, mainloop.cpp:305-315), then
//     published with the mean quality;
//   * the gyro taken along with every published flow (mainloop.cpp:333-334: integrated since the last message);
//   * pixel flow -> angular flow (include/aof_math.h), the OPTICAL_FLOW_RAD field mapping of
//     mainloop.cpp:359-371 and the MAVLink 2 frame of mavlink_tcp.cpp:142-162 (facade/src/optical_flow_rad.cpp).
//
// WHICH frames publish depends on the time stamps alone: frame j publishes iff it is the first frame behind
// the previous publication i with (float)(t_j - t_i) > period.  That is a linked list through the frames
// (next[i]), entered at a virtual start node with time 0 (time_last_pub's initial value): k_limit_next builds
// the list (one lane per frame, a short forward scan), k_limit_double marks the nodes on the chain from the
// start by pointer jumping (log4(n) rounds: marked nodes mark their d-th, 2d-th and 3d-th successors, every node
// learns its 4d-th) and numbers them along the way (rank = position on the chain = index of the node's message),
// and k_sequence_emit -- one lane per published frame -- sums its segment IN FRAME ORDER (float adds are not
// associative: the host sums in that order), converts, fills and packs.  Every float operation is the host's,
// in the host's order: the frames are byte-identical to driving the C++ facade frame by frame.
#include "aof_device.hpp"
#include "aof_internal.hpp"
#include "aof_math.h"

namespace aof {

namespace {

constexpr int kThreads = 256;

// next[i] for frames i = 0 .. n-1 (node 0 is the START: time 0, not frame 0's time -- frame 0 never reaches
// the limiter, optical_flow.cpp integrate()), node n = END.  Also initialises the doubling state.
__global__ __launch_bounds__(kThreads) void k_limit_next(SequenceArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t n = a.n_frames;
    if (i > n) return;
    if (i == n) {   // END: its own successor, no hops
        a.jump[0][n] = (uint32_t)n;
        a.hops[0][n] = 0;
        a.reached[n] = 0;
        a.rank[n] = 0;
        return;
    }
    uint32_t nxt = (uint32_t)n;
    if (a.output_rate <= 0) {
        nxt = (uint32_t)(i + 1);   // no limit: every frame publishes
    } else {
        const uint32_t base = i == 0 ? 0u : (uint32_t)a.time_us[i];
        const int64_t stop = i + 1 + kLimitScanFrames < n ? i + 1 + kLimitScanFrames : n;
        int64_t j = i + 1;
        for (; j < stop; j++)
            if ((float)((uint32_t)a.time_us[j] - base) > a.period_us) break;
        if (j < stop) nxt = (uint32_t)j;
        else if (stop < n) atomicOr(a.status, 1u);   // the time stamps do not advance: this chain ends here
    }
    a.jump[0][i] = nxt;
    a.hops[0][i] = 1;
    a.reached[i] = i == 0 ? 1 : 0;   // (value = round in which the node was marked, + 1)
    a.rank[i] = 0;
}

// One round of pointer QUADRUPLING: in -> out.  Entering round r every node knows its d-th successor (d = 4^(r-1))
// and the nodes at fewer than d hops from the start are marked.  A node marked in an EARLIER round marks and
// numbers its successors at d, 2d and 3d hops (the marked set grows to 4d; every position has exactly one
// writer), and every node learns its 4d-th successor: half the launches of plain doubling.
__global__ __launch_bounds__(kThreads) void k_limit_double(SequenceArgs a, int round)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i > a.n_frames) return;
    const int in = (round + 1) & 1, out = round & 1;   // round 1 reads buffer 0
    const uint32_t j1 = a.jump[in][i], h1 = a.hops[in][i];
    const uint32_t j2 = a.jump[in][j1], h2 = a.hops[in][j1];
    const uint32_t j3 = a.jump[in][j2], h3 = a.hops[in][j2];
    a.jump[out][i] = a.jump[in][j3];
    a.hops[out][i] = h1 + h2 + h3 + a.hops[in][j3];
    const uint8_t mark = a.reached[i];
    if (mark != 0 && mark <= round) {
        const uint32_t r0 = a.rank[i];
        // (the END node absorbs every jump past the last publication: all of its writers carry the same rank)
        if (a.reached[j1] == 0) { a.rank[j1] = r0 + h1; a.reached[j1] = (uint8_t)(round + 1); }
        if (j2 != j1 && a.reached[j2] == 0) { a.rank[j2] = r0 + h1 + h2; a.reached[j2] = (uint8_t)(round + 1); }
        if (j3 != j2 && a.reached[j3] == 0) { a.rank[j3] = r0 + h1 + h2 + h3; a.reached[j3] = (uint8_t)(round + 1); }
    }
}

__device__ __forceinline__ uint16_t crc_accumulate(uint8_t byte, uint16_t crc)
{
    uint8_t tmp = (uint8_t)(byte ^ (uint8_t)(crc & 0xFF));
    tmp = (uint8_t)(tmp ^ (uint8_t)(tmp << 4));
    return (uint16_t)((crc >> 8) ^ ((uint16_t)tmp << 8) ^ ((uint16_t)tmp << 3) ^ (tmp >> 4));
}

template <typename T> __device__ __forceinline__ void put(uint8_t *&p, T v)
{
    __builtin_memcpy(p, &v, sizeof(T));   // little-endian wire order = the device's own
    p += sizeof(T);
}

// One lane per frame; the lanes of published frames (frame 0, and the frames on the chain) write their message.
__global__ __launch_bounds__(kThreads) void k_sequence_emit(SequenceArgs a)
{
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t n = a.n_frames;
    if (k == n) {                                              // (the END node's lane)
        a.count[0] = a.rank[n];                                // records: frame 0's + one per publication
        a.count[1] = a.offset_timestamp_usec ? a.rank[n] : 0u; // frames sent (mainloop.cpp:353-357)
    }
    if (k >= n) return;
    int quality = 0, dt_us = 0;
    float px = 0.0f, py = 0.0f;
    double gx = 0.0, gy = 0.0, gz = 0.0;
    uint32_t m = 0;
    if (k == 0) {
        // the first frame: calcFlow returns 0 with its outputs untouched (integrate(): nothing to compare it
        // with), and the caller sends what its zero-initialised locals hold (mainloop.cpp:280-281,322-373)
        if (a.gyro) { gx = a.gyro[0].integ_x; gy = a.gyro[0].integ_y; gz = a.gyro[0].integ_z; }
    } else {
        if (a.reached[k] == 0) return;
        m = a.rank[k];
        int64_t s = k - 1;                       // the previous publication (0 = the start)
        while (s > 0 && a.reached[s] == 0) s--;
        const uint32_t last = s == 0 ? 0u : (uint32_t)a.time_us[s];
        dt_us = (int)((uint32_t)a.time_us[k] - last);
        if (a.output_rate <= 0) {                // limitRate: no limit, the frame's own flow and quality
            const aof_flow f = a.flows[k - 1];
            quality = f.quality; px = f.flow_x; py = f.flow_y;
        } else {
            float sum_x = 0.0f, sum_y = 0.0f;
            int sum_q = 0, valid = 0;
            for (int64_t j = s + 1; j <= k; j++) {   // in frame order, as the host sums
                const aof_flow f = a.flows[j - 1];
                if (f.quality > 0) {
                    sum_x += f.flow_x;
                    sum_y += f.flow_y;
                    sum_q += f.quality;
                    valid++;
                }
            }
            if (valid > 0) quality = (int)floorf((float)sum_q / (float)valid);
            px = sum_x; py = sum_y;
        }
        if (a.gyro)
            for (int64_t j = s + 1; j <= k; j++) { gx += a.gyro[j].integ_x; gy += a.gyro[j].integ_y; gz += a.gyro[j].integ_z; }
    }
    float ang_x = 0.0f, ang_y = 0.0f;
    if (k != 0) { ang_x = aof_atan2f(px, a.focal_x); ang_y = aof_atan2f(py, a.focal_y); }
    aof_seq_record rec;
    rec.frame = (uint32_t)k; rec.quality = quality; rec.dt_us = dt_us;
    rec.flow_x = ang_x; rec.flow_y = ang_y;
    rec.gyro_x = (float)gx; rec.gyro_y = (float)gy; rec.gyro_z = (float)gz;
    a.records[m] = rec;
    if (!a.frames) return;
    uint8_t *out = a.frames + (size_t)m * AOF_SEQ_FRAME_BYTES;
    if (a.offset_timestamp_usec == 0) {          // vehicle time not known: nothing is sent (mainloop.cpp:353-357)
        a.frame_len[m] = 0;
        return;
    }
    // field mapping of mainloop.cpp:359-371 (gyro axes switched to match pixel directions), wire order of
    // OPTICAL_FLOW_RAD (message 106): by field size, then declaration
    uint8_t payload[44];
    uint8_t *p = payload;
    put(p, (uint64_t)(a.offset_timestamp_usec + a.time_us[k]));
    put(p, (uint32_t)dt_us);
    put(p, ang_x);
    put(p, ang_y);
    put(p, (float)(-gy));
    put(p, (float)gx);
    put(p, (float)gz);
    put(p, (uint32_t)0);        // time_delta_distance_us
    put(p, -1.0f);              // distance
    put(p, (int16_t)0);         // temperature
    put(p, (uint8_t)0);         // sensor_id
    put(p, (uint8_t)quality);
    int len = 44;
    while (len > 1 && payload[len - 1] == 0) len--;   // MAVLink 2 payload truncation
    uint8_t head[10] = {0xFD, (uint8_t)len, 0, 0, (uint8_t)(a.first_seq + m), a.system_id, a.component_id, 106, 0, 0};
    uint16_t crc = 0xFFFF;
#pragma unroll
    for (int b = 0; b < 10; b++) {
        out[b] = head[b];
        if (b) crc = crc_accumulate(head[b], crc);
    }
    for (int b = 0; b < len; b++) {
        out[10 + b] = payload[b];
        crc = crc_accumulate(payload[b], crc);
    }
    crc = crc_accumulate(138, crc);   // CRC_EXTRA of OPTICAL_FLOW_RAD
    out[10 + len] = (uint8_t)(crc & 0xFF);
    out[11 + len] = (uint8_t)(crc >> 8);
    a.frame_len[m] = (uint8_t)(12 + len);
}

}  // namespace

int launch_sequence_output(const SequenceArgs &a, void *stream)
{
    if (a.n_frames <= 0) return 0;
    if (a.n_frames >= 0x7FFFFFF0ll) return (int)hipErrorInvalidValue;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t wgs = (uint32_t)((a.n_frames + 1 + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_limit_next, dim3(wgs), dim3(kThreads), 0, s, a);
    for (int round = 1; round <= sequence_rounds(a.n_frames); round++)
        hipLaunchKernelGGL(k_limit_double, dim3(wgs), dim3(kThreads), 0, s, a, round);
    hipLaunchKernelGGL(k_sequence_emit, dim3(wgs), dim3(kThreads), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace aof
