// aof_sequence_device (include/aof.h): a recorded frame sequence through ingest, sequence-mode flow, the rate
// limiter, de-rotation and the OPTICAL_FLOW_RAD packer in one enqueue -- the reference's per-frame loop
// (/root/reference/src/mainloop.cpp:295-373) for all frames of a recording at once.  Host side only: it
// composes the batched entry points of the C ABI and the output kernels of k_sequence.hip on ONE stream;
// nothing here allocates or synchronises.
#include <cerrno>
#include <cstring>

#include "aof_internal.hpp"
#include "aof_math.h"

using namespace aof;

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Scratch {   // offsets inside aof_seq_layout.scratch
    size_t jump[2], hops[2], rank, reached, flow_ws, flow_ws_bytes, total;
};

int scratch_layout(const aof_params *p, int64_t n_frames, Scratch *s)
{
    const size_t nodes = (size_t)n_frames + 1;
    size_t off = 0;
    for (int b = 0; b < 2; b++) { s->jump[b] = off; off = align_up(off + nodes * 4, 256); }
    for (int b = 0; b < 2; b++) { s->hops[b] = off; off = align_up(off + nodes * 4, 256); }
    s->rank = off;    off = align_up(off + nodes * 4, 256);
    s->reached = off; off = align_up(off + nodes, 256);
    aof_ws_layout L;
    const int rc = aof_workspace_layout(p, n_frames > 1 ? n_frames - 1 : 0, &L);
    if (rc) return rc;
    s->flow_ws = off;
    s->flow_ws_bytes = L.total_bytes;
    s->total = align_up(off + L.total_bytes, 256);
    return 0;
}

int check(const aof_params *p, const aof_sequence_params *sp, int64_t n_frames)
{
    if (!p || !sp || n_frames < 0 || n_frames >= 0x7FFFFFF0ll) return -EINVAL;
    const int rc = aof_params_check(p);
    if (rc) return rc;
    const aof_ingest_params &g = sp->ingest;
    if (g.crop_width != p->width || g.crop_height != p->height) return -EINVAL;   // the flow runs on the crop
    if (g.crop_width < 1 || g.crop_height < 1 || g.crop_width > g.camera_width || g.crop_height > g.camera_height)
        return -EINVAL;
    if (!(sp->focal_x > 0.0f) || !(sp->focal_y > 0.0f)) return -EINVAL;
    return 0;
}

}  // namespace

extern "C" {

float aof_flow_angle(float flow_px, float focal_px) { return aof_atan2f(flow_px, focal_px); }

int aof_sequence_layout(const aof_params *p, const aof_sequence_params *sp, int64_t n_frames, aof_seq_layout *out)
{
    if (!out) return -EINVAL;
    int rc = check(p, sp, n_frames);
    if (rc) return rc;
    const size_t n = (size_t)n_frames, pairs = n > 1 ? n - 1 : 0;
    const size_t frame = (size_t)p->width * (size_t)p->height;
    Scratch s;
    rc = scratch_layout(p, n_frames, &s);
    if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    size_t off = 0;
    out->cropped = off;    off = align_up(off + n * frame, 256);
    out->exposure = off;   off = align_up(off + n * AOF_EXPOSURE_BINS * sizeof(uint32_t), 256);
    out->flows = off;      off = align_up(off + pairs * sizeof(aof_flow), 256);
    out->derotated = off;  off = align_up(off + (sp->derotate ? pairs * 2 * sizeof(float) : 0), 256);
    out->count = off;      off = align_up(off + 4 * sizeof(uint32_t), 256);
    out->records = off;    off = align_up(off + n * sizeof(aof_seq_record), 256);
    out->frames = off;     off = align_up(off + n * AOF_SEQ_FRAME_BYTES, 256);
    out->frame_len = off;  off = align_up(off + n, 256);
    out->scratch = off;    off += s.total;
    out->total_bytes = off ? off : 256;
    return 0;
}

int aof_sequence_device(aof_ctx *ctx, const aof_sequence_params *sp, const uint8_t *d_camera, int64_t camera_stride,
                        int64_t n_frames, const uint64_t *d_time_us, const aof_gyro *d_gyro, void *d_workspace,
                        size_t workspace_bytes, void *stream)
{
    if (!ctx) return -EINVAL;
    aof_params p;
    int rc = aof_get_params(ctx, &p);
    if (rc) return rc;
    aof_seq_layout L;
    rc = aof_sequence_layout(&p, sp, n_frames, &L);
    if (rc) return ctx_fail(ctx, rc, "sequence parameters do not match the context (crop size, frame count)");
    // before the first launch (the ingest kernel below): a faulted or wedged context, or a thread on another
    // device, must not put work into the caller's workspace
    rc = precheck(ctx);
    if (rc) return rc;
    if (n_frames == 0) return 0;
    if (!d_camera || !d_time_us || !d_workspace) return ctx_fail(ctx, -EINVAL, "null camera, time stamp or workspace pointer");
    if (workspace_bytes < L.total_bytes) return ctx_fail(ctx, -ENOSPC, "sequence workspace smaller than aof_sequence_layout().total_bytes");
    if (reinterpret_cast<uintptr_t>(d_workspace) % 256) return ctx_fail(ctx, -EINVAL, "workspace must be 256-byte aligned");
    if (sp->derotate && !d_gyro) return ctx_fail(ctx, -EINVAL, "de-rotation needs the gyro samples");
    Scratch s;
    scratch_layout(&p, n_frames, &s);
    uint8_t *ws = static_cast<uint8_t *>(d_workspace);
    const int64_t frame = (int64_t)p.width * p.height;
    uint8_t *cropped = ws + L.cropped;
    aof_flow *flows = reinterpret_cast<aof_flow *>(ws + L.flows);

    // 1. sensor frames -> the cropped sequence + exposure histograms (mainloop.cpp:295-298,203-214).  Where the flow
    //    would run K1 (pixel sums, 2x2 pyramid) as a pass of its own over the cropped frames, the ingest kernel
    //    leaves K1's outputs itself -- once per FRAME, out of the registers the crop passes through.
    uint8_t *flow_ws = ws + L.scratch + s.flow_ws;
    uint32_t *exposure = reinterpret_cast<uint32_t *>(ws + L.exposure);
    bool k1_ready = false;
    if (n_frames > 1 && ingest_pyramid_supported(sp->ingest, cropped, frame) && sequence_runs_k1(ctx, cropped, n_frames - 1, flow_ws)) {
        aof_ws_layout FL;
        rc = aof_workspace_layout(&p, n_frames - 1, &FL);
        if (rc) return rc;
        rc = launch_ingest_pyramid(sp->ingest, d_camera, camera_stride, n_frames, cropped, frame, exposure,
                                   p.pyramid_levels == 2 ? flow_ws + FL.l1_prev : nullptr,
                                   p.mean_subtract ? reinterpret_cast<uint32_t *>(flow_ws + FL.sums) : nullptr, stream);
        if (rc) return -EIO;
        k1_ready = true;
    } else {
        rc = aof_ingest_batch_device(&sp->ingest, d_camera, camera_stride, n_frames, cropped, frame, exposure, stream);
        if (rc) return rc;
    }
    // 2. flow of consecutive frames: the same buffer viewed twice (frame k is cur of pair k-1, prev of pair k)
    if (n_frames > 1) {
        rc = flow_sequence(ctx, cropped, n_frames - 1, flows, flow_ws, s.flow_ws_bytes, stream, k1_ready);
        if (rc) return rc;
    }
    // 3. + 5. limiter, gyro sums, angles, records and MAVLink frames
    uint32_t *count = reinterpret_cast<uint32_t *>(ws + L.count);
    rc = launch_zero_words(count, 4, stream);
    if (rc) return -EIO;
    SequenceArgs a;
    a.n_frames = n_frames;
    a.time_us = d_time_us;
    a.gyro = d_gyro;
    a.flows = flows;
    a.output_rate = sp->output_rate;
    a.period_us = sp->output_rate > 0 ? 1.0e6f / (float)sp->output_rate : 0.0f;   // (the facade's own division)
    a.focal_x = sp->focal_x; a.focal_y = sp->focal_y;
    a.offset_timestamp_usec = sp->offset_timestamp_usec;
    a.system_id = sp->system_id; a.component_id = sp->component_id; a.first_seq = sp->first_seq;
    uint8_t *scratch = ws + L.scratch;
    for (int b = 0; b < 2; b++) {
        a.jump[b] = reinterpret_cast<uint32_t *>(scratch + s.jump[b]);
        a.hops[b] = reinterpret_cast<uint32_t *>(scratch + s.hops[b]);
    }
    a.rank = reinterpret_cast<uint32_t *>(scratch + s.rank);
    a.reached = scratch + s.reached;
    a.count = count;
    a.status = count + 2;
    a.records = reinterpret_cast<aof_seq_record *>(ws + L.records);
    a.frames = ws + L.frames;
    a.frame_len = ws + L.frame_len;
    rc = launch_sequence_output(a, stream);
    if (rc) return -EIO;
    // 4. the de-rotated pixel flow of every pair (gyro of the interval that ends at the pair's newer frame)
    if (sp->derotate && n_frames > 1) {
        rc = aof_derotate_batch_device(&sp->derotate_params, flows, d_gyro + 1, n_frames - 1,
                                       reinterpret_cast<float *>(ws + L.derotated), stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
