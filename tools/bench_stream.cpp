// Times aof_stream_push_host (the call under OpticalFlowPX4::calcFlow) with and without the
// captured hipGraph.   g++ -O2 -Iinclude tools/bench_stream.cpp -Laero-optical-flow_amd/csrc -laof
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "aof.h"
int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 64, h = argc > 2 ? atoi(argv[2]) : 64;
    const int levels = argc > 3 ? atoi(argv[3]) : 1;   // 2: what OpticalFlowOpenCV ships (two levels + mean equalisation)
    const int only = argc > 4 ? atoi(argv[4]) : -1;    // run one mode only (1 = lane8 without the graph: rocprofv3 cannot trace graph launches)
    const int calls = argc > 5 ? atoi(argv[5]) : 5000;   // (a long run shows whether the per-call cost creeps: the tagged path never waits for the stream)
    aof_params p;
    aof_params_px4flow(&p, w, h, 4, 30, 3000);
    if (levels == 2) { p.pyramid_levels = 2; p.mean_subtract = 1; }
    std::vector<uint8_t> f[2];
    // a blurred random canvas cropped twice, 2 px / 1 px apart: every block has a clear match
    std::vector<int> canvas((size_t)(w + 8) * (h + 8));
    for (auto &v : canvas) v = rand() & 255;
    for (int k = 0; k < 2; k++) {
        f[k].resize((size_t)w * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int acc = 0;
                for (int dy = 0; dy < 3; dy++)
                    for (int dx = 0; dx < 3; dx++) acc += canvas[(size_t)(y + dy + k) * (w + 8) + x + dx + 2 * k];
                f[k][(size_t)y * w + x] = (uint8_t)(acc / 9);
            }
    }
    for (int mode = 0; mode < 5; mode++) {   // 4: the resident kernel (aof_set_stream_resident)
        if (only >= 0 && mode != only) continue;
        const int graph = !(mode & 1), generic = (mode >> 1) & 1, resident = mode == 4;
        // one line on stderr per phase, flushed: a run that is cut off by `timeout` leaves its position behind
        auto phase = [&](const char *what) { fprintf(stderr, "[bench_stream %dx%d levels=%d mode=%d] %s\n", w, h, levels, mode, what); fflush(stderr); };
        aof_ctx *ctx;
        phase("create");
        if (aof_create(&p, 0, &ctx)) { printf("no device\n"); return 1; }
        if (aof_set_force_generic(ctx, generic) || aof_set_stream_graph(ctx, graph) < 0 || aof_set_stream_resident(ctx, resident) < 0) {
            printf("mode %d: cannot select the path: %s\n", mode, aof_last_error(ctx));
            return 1;
        }
        aof_flow out;
        int bad = 0;
        phase("warm-up");
        for (int i = 0; i < 50; i++) bad += aof_stream_push_host(ctx, f[i & 1].data(), &out) < 0;
        phase("timed calls");
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < calls; i++) bad += aof_stream_push_host(ctx, f[i & 1].data(), &out) < 0;
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
        aof_stream_stats st;
        aof_stream_get_stats(ctx, &st);
        printf("%dx%d levels=%d %s graph=%d instantiated=%d resident=%d on_device=%d: %.2f us per call (quality %d flow %.3f %.3f)"
               " [failed calls %d, resident: served %llu of %llu, launches %u, fallbacks %u, lost %u, longest launch call %.0f us,"
               " longest launch->first poll %.1f us; tagged records later than 2 ms: %u]\n",
               w, h, levels, aof_search_variant(ctx), graph, aof_set_stream_graph(ctx, -1), resident,
               aof_set_stream_resident(ctx, -1), us, out.quality, out.flow_x, out.flow_y, bad,
               (unsigned long long)st.resident_served, (unsigned long long)st.calls, st.resident_launches, st.resident_fallbacks,
               st.resident_lost, st.launch_call_us_max, st.start_latency_us_max, st.tagged_slow);
        if (st.last_report[0]) printf("  last fallback report: %s\n", st.last_report);
        fflush(stdout);
        phase("destroy");
        aof_destroy(ctx);
        phase("done");
        if (bad || (resident && (st.resident_fallbacks || st.resident_lost))) return 2;   // a failed or unanswered call is an error, not a timing
    }
    return 0;
}
