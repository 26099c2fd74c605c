// vpp_latency -- single-frame latency of VideoProcessor::Convert through the C++ class (the only quantity the reference publishes is a latency:
// reference tests/src/WrapperTests.cpp:303-309 accepts getFrame() at 3 +- 3 ms, decoder wait included).  One synthetic NV12 frame
// resident in device memory, `iters` conversions, each timed on the host from the call to the end of hipStreamSynchronize on the consumer's stream:
//   convert       VideoProcessor::Convert as the reference's contract has it: the result is a fresh device buffer the caller frees (hipMalloc + hipFree
//                 per frame, like the reference's cudaMalloc / cudaFree)
//   convert_release  the same Convert, the result handed back with VideoProcessor::Release instead of hipFree: no allocator call in the steady state
//   convert_into  VideoProcessor::ConvertInto: caller-owned output, no allocation (what a torch-backed getFrame passes: tensor.data_ptr())
//   graph         the same launch captured once into a hipGraph on the consumer's stream and replayed
//   eager8 / graph8   eight conversions (eight ring slots) issued back to back + ONE synchronisation, directly and as one eight-node graph: hipGraphLaunch has a fixed
//                 host cost (~10-16 us on this runtime, /opt/skills/guides/MI355X_MICROARCH.md "graph-replay-floor") where a direct launch costs ~3.5 us, so a
//                 ONE-node graph is slower than the launch it holds and a graph only pays once it holds several launches
//   back_to_back  device time per frame of `iters` launches in a row, in order; *_inputs_ready: the same under TSVPP_OPT_INPUTS_READY (two streams, no barrier bit)
// Inputs and outputs ROTATE through a ring whose bytes exceed 768 MiB (round 6: rounds 3-5 reused one 14 MB pair, which the 256 MiB Infinity Cache served).
// Prints ONE JSON line.  usage: vpp_latency [W H dstW dstH resizeType fourcc planes norm iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "VideoProcessor.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void stats(std::vector<double> &v, double &p50, double &p99, double &mean, double &mn) {
    std::sort(v.begin(), v.end());
    p50 = v[v.size() / 2];
    p99 = v[std::min(v.size() - 1, (size_t)(v.size() * 0.99))];
    mn = v.front();
    mean = 0;
    for (double x : v) mean += x;
    mean /= (double)v.size();
}

int main(int argc, char **argv) {
    int W = 1920, H = 1080, dw = 1280, dh = 720, rt = 1, fcc = 2, planes = 0, norm = 1, iters = 2000;
    if (argc >= 9) {
        W = atoi(argv[1]); H = atoi(argv[2]); dw = atoi(argv[3]); dh = atoi(argv[4]); rt = atoi(argv[5]); fcc = atoi(argv[6]); planes = atoi(argv[7]); norm = atoi(argv[8]);
    }
    if (argc >= 10) iters = atoi(argv[9]);
    const int pitch = (W + 255) / 256 * 256;
    std::vector<uint8_t> host((size_t)pitch * H * 3 / 2);
    uint32_t s = 12345u;
    for (auto &b : host) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    const int ow0 = dw ? dw : W, oh0 = dh ? dh : H;
    const size_t out_bytes0 = (size_t)(channelsByFourCC((FourCC)fcc) * ow0 * oh0) * ((norm || fcc == 6) ? 4 : 1);
    const size_t in_bytes = (size_t)pitch * H * 3 / 2, out_stride = (out_bytes0 + 255) & ~(size_t)255;
    const int R = std::max<int>(8, (int)(((size_t)768 << 20) / (in_bytes + out_stride)) + 1); // ring slots
    uint8_t *ringIn = nullptr, *ringOut = nullptr;
    if (hipMalloc(&ringIn, in_bytes * R) != hipSuccess || hipMalloc(&ringOut, out_stride * R) != hipSuccess) return 3;
    for (int r = 0; r < R; r++) {
        host[(size_t)r % host.size()] ^= (uint8_t)(r + 1); // (slots differ)
        (void)hipMemcpy(ringIn + (size_t)r * in_bytes, host.data(), in_bytes, hipMemcpyHostToDevice);
    }
    int slot = 0;

    VideoProcessor vpp;
    if (vpp.Init(std::make_shared<Logger>(), 2) != 0) return 4;
    ColorOptions color((FourCC)fcc);
    color.planesPos = (Planes)planes;
    color.normalization = norm != 0;
    ResizeOptions resize(dw, dh);
    resize.type = (ResizeType)rt;
    FrameParameters params(resize, color, CropOptions({ 0, 0 }, { 0, 0 }));
    void *stream = nullptr;
    if (tsvpp_consumer_stream(vpp.context(), "latency", &stream) != 0) return 5;
    auto fill = [&](AVFrame *f) { // the next ring slot's input
        slot = (slot + 1) % R;
        *f = AVFrame();
        f->data[0] = ringIn + (size_t)slot * in_bytes; f->data[1] = f->data[0] + (size_t)pitch * H; f->linesize[0] = f->linesize[1] = pitch; f->width = W; f->height = H;
    };
    auto out_of = [&](int sl) { return (void *)(ringOut + (size_t)sl * out_stride); };
    AVFrame *in = av_frame_alloc(), *out = av_frame_alloc();
    const int ow = ow0, oh = oh0;

    std::vector<double> t_alloc, t_into, t_graph, t_release;
    const int warm = 200;
    for (int i = 0; i < warm + iters; i++) { // the reference's contract: fresh result buffer, freed by the caller
        fill(in);
        const double t0 = now_us();
        if (vpp.Convert(in, out, params, "latency") != 0) return 10;
        (void)hipStreamSynchronize((hipStream_t)stream);
        const double t1 = now_us();
        (void)hipFree(out->opaque);
        if (i >= warm) t_alloc.push_back(t1 - t0);
    }
    for (int i = 0; i < warm + iters; i++) { // ... and with the result handed back instead of freed
        fill(in);
        const double t0 = now_us();
        if (vpp.Convert(in, out, params, "latency") != 0) return 10;
        (void)hipStreamSynchronize((hipStream_t)stream);
        const double t1 = now_us();
        if (vpp.Release(out->opaque, (hipStream_t)stream) != 0) return 13;
        if (i >= warm) t_release.push_back(t1 - t0);
    }
    for (int i = 0; i < warm + iters; i++) {
        fill(in);
        const double t0 = now_us();
        if (vpp.ConvertInto(in, out_of(slot), params, "latency") != 0) return 11;
        (void)hipStreamSynchronize((hipStream_t)stream);
        if (i >= warm) t_into.push_back(now_us() - t0);
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    fill(in);
    bool graph_ok = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess && vpp.ConvertInto(in, out_of(slot), params, "latency") == 0 &&
                    hipStreamEndCapture((hipStream_t)stream, &graph) == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph_ok) {
        for (int i = 0; i < warm + iters; i++) {
            const double t0 = now_us();
            if (hipGraphLaunch(exec, (hipStream_t)stream) != hipSuccess) { graph_ok = false; break; }
            (void)hipStreamSynchronize((hipStream_t)stream);
            if (i >= warm) t_graph.push_back(now_us() - t0);
        }
    }
    // eight conversions + one synchronisation: directly, and as ONE eight-node graph (slots 0..7)
    std::vector<double> t_eager8, t_graph8;
    for (int i = 0; i < warm / 4 + iters / 8; i++) {
        const double t0 = now_us();
        for (int k = 0; k < 8; k++) { fill(in); if (vpp.ConvertInto(in, out_of(slot), params, "latency") != 0) return 12; }
        (void)hipStreamSynchronize((hipStream_t)stream);
        if (i >= warm / 4) t_eager8.push_back((now_us() - t0) / 8);
    }
    hipGraph_t graph8 = nullptr;
    hipGraphExec_t exec8 = nullptr;
    bool graph8_ok = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (graph8_ok) {
        slot = R - 1;
        for (int k = 0; k < 8; k++) { fill(in); graph8_ok = graph8_ok && vpp.ConvertInto(in, out_of(slot), params, "latency") == 0; }
        graph8_ok = hipStreamEndCapture((hipStream_t)stream, &graph8) == hipSuccess && graph8_ok && hipGraphInstantiate(&exec8, graph8, nullptr, nullptr, 0) == hipSuccess;
    }
    if (graph8_ok) {
        for (int i = 0; i < warm / 4 + iters / 8; i++) {
            const double t0 = now_us();
            if (hipGraphLaunch(exec8, (hipStream_t)stream) != hipSuccess) { graph8_ok = false; break; }
            (void)hipStreamSynchronize((hipStream_t)stream);
            if (i >= warm / 4) t_graph8.push_back((now_us() - t0) / 8);
        }
    }
    // device time of the launch alone (events around `iters` back-to-back launches, in order on the consumer's stream)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 2000; i++) { fill(in); (void)vpp.ConvertInto(in, out_of(slot), params, "latency"); } // clock ramp
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters; i++) { fill(in); (void)vpp.ConvertInto(in, out_of(slot), params, "latency"); }
    (void)hipEventRecord(e1, (hipStream_t)stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // ... and under TSVPP_OPT_INPUTS_READY: the consumer alternates between two streams, no barrier bit (wall clock: the work spans two streams)
    double ready_us = 0;
    if (tsvpp_set_option(vpp.context(), TSVPP_OPT_INPUTS_READY, 1) == 0) {
        for (int i = 0; i < 2000; i++) { fill(in); (void)vpp.ConvertInto(in, out_of(slot), params, "latency"); }
        (void)tsvpp_consumer_synchronize(vpp.context(), "latency");
        const double t0 = now_us();
        for (int i = 0; i < iters; i++) { fill(in); (void)vpp.ConvertInto(in, out_of(slot), params, "latency"); }
        (void)tsvpp_consumer_synchronize(vpp.context(), "latency");
        ready_us = (now_us() - t0) / iters;
        (void)tsvpp_set_option(vpp.context(), TSVPP_OPT_INPUTS_READY, 0);
    }

    double p50, p99, mean, mn;
    printf("{\"entry\": \"VideoProcessor::Convert (C++ class, one frame per call)\", \"workload\": \"%dx%d -> %dx%d rt=%d fourcc=%d planes=%d norm=%d\", \"iters\": %d", W, H, ow, oh, rt, fcc,
           planes, norm, iters);
    stats(t_alloc, p50, p99, mean, mn);
    printf(", \"convert_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f, \"note\": \"hipMalloc of the result inside, hipFree by the caller, like the reference\"}", p50, p99, mean, mn);
    stats(t_release, p50, p99, mean, mn);
    printf(", \"convert_release_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f, \"note\": \"the same Convert, the result handed back with VideoProcessor::Release instead of hipFree\"}", p50, p99, mean, mn);
    stats(t_into, p50, p99, mean, mn);
    printf(", \"convert_into_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f}", p50, p99, mean, mn);
    if (graph_ok && !t_graph.empty()) {
        stats(t_graph, p50, p99, mean, mn);
        printf(", \"graph_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f}", p50, p99, mean, mn);
    } else {
        printf(", \"graph_us\": null");
    }
    if (!t_eager8.empty()) {
        stats(t_eager8, p50, p99, mean, mn);
        printf(", \"eager8_us_per_frame\": {\"p50\": %.2f, \"p99\": %.2f}", p50, p99);
    }
    if (graph8_ok && !t_graph8.empty()) {
        stats(t_graph8, p50, p99, mean, mn);
        printf(", \"graph8_us_per_frame\": {\"p50\": %.2f, \"p99\": %.2f, \"note\": \"one hipGraphLaunch of eight kernel nodes + one synchronisation, per frame: the fixed cost of a replay is shared by eight launches\"}", p50, p99);
    }
    printf(", \"ring_slots\": %d, \"ring_MiB\": %.0f, \"back_to_back_inputs_ready_us_per_frame\": %.2f", R, (double)R * (in_bytes + out_stride) / 1048576.0, ready_us);
    printf(", \"back_to_back_us_per_frame\": %.2f, \"reference_published\": \"getFrame 3 +- 3 ms (decoder wait included), tests/src/WrapperTests.cpp:303-309\"}\n", ms * 1e3 / iters);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (exec8) (void)hipGraphExecDestroy(exec8);
    if (graph8) (void)hipGraphDestroy(graph8);
    vpp.Close();
    (void)hipFree(ringIn);
    (void)hipFree(ringOut);
    return 0;
}
