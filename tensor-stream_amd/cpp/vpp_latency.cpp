// vpp_latency -- single-frame latency of VideoProcessor::Convert through the C++ class (the only quantity the reference publishes is a latency:
// reference tests/src/WrapperTests.cpp:303-309 accepts getFrame() at 3 +- 3 ms, decoder wait included).  One synthetic NV12 frame
// resident in device memory, `iters` conversions, each timed on the host from the call to the end of hipStreamSynchronize on the consumer's stream:
//   convert       VideoProcessor::Convert as the reference's contract has it: the result is a fresh device buffer the caller frees (hipMalloc + hipFree
//                 per frame, like the reference's cudaMalloc / cudaFree)
//   convert_into  VideoProcessor::ConvertInto: caller-owned output, no allocation (what a torch-backed getFrame passes: tensor.data_ptr())
//   graph         the same launch captured once into a hipGraph on the consumer's stream and replayed
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
    uint8_t *dY = nullptr, *dUV = nullptr;
    if (hipMalloc(&dY, (size_t)pitch * H) != hipSuccess || hipMalloc(&dUV, (size_t)pitch * H / 2) != hipSuccess) return 3;
    (void)hipMemcpy(dY, host.data(), (size_t)pitch * H, hipMemcpyHostToDevice);
    (void)hipMemcpy(dUV, host.data() + (size_t)pitch * H, (size_t)pitch * H / 2, hipMemcpyHostToDevice);

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
    auto fill = [&](AVFrame *f) {
        *f = AVFrame();
        f->data[0] = dY; f->data[1] = dUV; f->linesize[0] = f->linesize[1] = pitch; f->width = W; f->height = H;
    };
    AVFrame *in = av_frame_alloc(), *out = av_frame_alloc();
    const int ow = dw ? dw : W, oh = dh ? dh : H;
    const size_t out_bytes = (size_t)(channelsByFourCC((FourCC)fcc) * ow * oh) * ((norm || fcc == 6) ? 4 : 1);
    void *dOut = nullptr;
    if (hipMalloc(&dOut, out_bytes) != hipSuccess) return 3;

    std::vector<double> t_alloc, t_into, t_graph;
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
    for (int i = 0; i < warm + iters; i++) {
        fill(in);
        const double t0 = now_us();
        if (vpp.ConvertInto(in, dOut, params, "latency") != 0) return 11;
        (void)hipStreamSynchronize((hipStream_t)stream);
        if (i >= warm) t_into.push_back(now_us() - t0);
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    fill(in);
    bool graph_ok = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess && vpp.ConvertInto(in, dOut, params, "latency") == 0 &&
                    hipStreamEndCapture((hipStream_t)stream, &graph) == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph_ok) {
        for (int i = 0; i < warm + iters; i++) {
            const double t0 = now_us();
            if (hipGraphLaunch(exec, (hipStream_t)stream) != hipSuccess) { graph_ok = false; break; }
            (void)hipStreamSynchronize((hipStream_t)stream);
            if (i >= warm) t_graph.push_back(now_us() - t0);
        }
    }
    // device time of the launch alone (events around `iters` back-to-back launches)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters; i++) { fill(in); (void)vpp.ConvertInto(in, dOut, params, "latency"); }
    (void)hipEventRecord(e1, (hipStream_t)stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);

    double p50, p99, mean, mn;
    printf("{\"entry\": \"VideoProcessor::Convert (C++ class, one frame per call)\", \"workload\": \"%dx%d -> %dx%d rt=%d fourcc=%d planes=%d norm=%d\", \"iters\": %d", W, H, ow, oh, rt, fcc,
           planes, norm, iters);
    stats(t_alloc, p50, p99, mean, mn);
    printf(", \"convert_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f, \"note\": \"hipMalloc of the result inside, hipFree by the caller, like the reference\"}", p50, p99, mean, mn);
    stats(t_into, p50, p99, mean, mn);
    printf(", \"convert_into_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f}", p50, p99, mean, mn);
    if (graph_ok && !t_graph.empty()) {
        stats(t_graph, p50, p99, mean, mn);
        printf(", \"graph_us\": {\"p50\": %.1f, \"p99\": %.1f, \"mean\": %.1f, \"min\": %.1f}", p50, p99, mean, mn);
    } else {
        printf(", \"graph_us\": null");
    }
    printf(", \"back_to_back_us_per_frame\": %.2f, \"reference_published\": \"getFrame 3 +- 3 ms (decoder wait included), tests/src/WrapperTests.cpp:303-309\"}\n", ms * 1e3 / iters);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipFree(dOut);
    vpp.Close();
    (void)hipFree(dY);
    (void)hipFree(dUV);
    return 0;
}
