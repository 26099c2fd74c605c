// vpp_cli -- drives the C++ VideoProcessor the way the reference's VPP tests do (tests/src/VPPTests.cpp:566-590):
// fill an AVFrame with device pointers, Convert(), copy the result back.  Used by tests/test_cpp_host_gpu.py.
//   vpp_cli in.nv12 W H pitch  cropL cropT cropR cropB  dstW dstH resizeType fourcc planes norm  out.bin [consumers]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "VideoProcessor.h"

int main(int argc, char **argv) {
    if (argc < 16) { fprintf(stderr, "usage: see source\n"); return 2; }
    const char *inPath = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), pitch = atoi(argv[4]);
    const int cl = atoi(argv[5]), ct = atoi(argv[6]), cr = atoi(argv[7]), cb = atoi(argv[8]);
    const int dw = atoi(argv[9]), dh = atoi(argv[10]), rt = atoi(argv[11]), fcc = atoi(argv[12]), planes = atoi(argv[13]), norm = atoi(argv[14]);
    const char *outPath = argv[15];
    std::vector<uint8_t> host((size_t)pitch * H * 3 / 2);
    FILE *f = fopen(inPath, "rb");
    if (!f || fread(host.data(), 1, host.size(), f) != host.size()) { fprintf(stderr, "cannot read %s\n", inPath); return 2; }
    fclose(f);
    uint8_t *dY = nullptr, *dUV = nullptr;
    if (hipMalloc(&dY, (size_t)pitch * H) != hipSuccess || hipMalloc(&dUV, (size_t)pitch * H / 2) != hipSuccess) return 3;
    (void)hipMemcpy(dY, host.data(), (size_t)pitch * H, hipMemcpyHostToDevice);
    (void)hipMemcpy(dUV, host.data() + (size_t)pitch * H, (size_t)pitch * H / 2, hipMemcpyHostToDevice);

    VideoProcessor vpp;
    if (vpp.Init(std::make_shared<Logger>(), 2) != 0) return 4;
    AVFrame *in = av_frame_alloc(), *out = av_frame_alloc();
    in->data[0] = dY;
    in->data[1] = dUV;
    in->linesize[0] = in->linesize[1] = pitch;
    in->width = W;
    in->height = H;
    ColorOptions color((FourCC)fcc);
    color.planesPos = (Planes)planes;
    color.normalization = norm != 0;
    ResizeOptions resize(dw, dh);
    resize.type = (ResizeType)rt;
    CropOptions crop({ cl, ct }, { cr, cb });
    FrameParameters params(resize, color, crop);
    int sts = vpp.Convert(in, out, params, "cli");
    if (sts != 0) { printf("status %d\n", sts); return 10; }
    // a third consumer on a 2-slot pool must be refused with VREADER_ERROR (findFree semantics)
    AVFrame *in2 = av_frame_alloc(), *out2 = av_frame_alloc();
    *in2 = AVFrame();
    in2->data[0] = dY; in2->data[1] = dUV; in2->linesize[0] = in2->linesize[1] = pitch; in2->width = W; in2->height = H;
    FrameParameters p2 = params;
    int s2 = vpp.Convert(in2, out2, p2, "second");
    if (s2 == 0) (void)hipFree(out2->opaque);
    *in2 = AVFrame();
    in2->data[0] = dY; in2->data[1] = dUV; in2->linesize[0] = in2->linesize[1] = pitch; in2->width = W; in2->height = H;
    int s3 = vpp.Convert(in2, out2, p2, "third");
    (void)hipDeviceSynchronize();
    const size_t n = (size_t)(channelsByFourCC((FourCC)fcc) * out->width * out->height) * (norm ? 4 : 1);
    std::vector<uint8_t> res(n);
    (void)hipMemcpy(res.data(), out->opaque, n, hipMemcpyDeviceToHost);
    FILE *o = fopen(outPath, "wb");
    fwrite(res.data(), 1, n, o);
    fclose(o);
    // Release(): the result handed back instead of freed -> the next Convert of the same size gets the same buffer (no allocator call) and the same bytes;
    // a pointer Convert never handed out, or one released twice, is VREADER_ERROR; hipFree of a result stays legal (reference contract)
    void *first = out->opaque;
    const int r1 = vpp.Release(first);
    *in2 = AVFrame();
    in2->data[0] = dY; in2->data[1] = dUV; in2->linesize[0] = in2->linesize[1] = pitch; in2->width = W; in2->height = H;
    FrameParameters p3 = params;
    const int s4 = vpp.Convert(in2, out, p3, "cli");
    (void)hipDeviceSynchronize();
    std::vector<uint8_t> res2(n);
    if (s4 == 0) (void)hipMemcpy(res2.data(), out->opaque, n, hipMemcpyDeviceToHost);
    const int reuse = (s4 == 0 && out->opaque == first) ? 1 : 0, same = (s4 == 0 && res2 == res) ? 1 : 0;
    const int r_foreign = vpp.Release((void *)dY), r_twice = (vpp.Release(out->opaque) == 0) ? vpp.Release(out->opaque) : 99;
    *in2 = AVFrame();
    in2->data[0] = dY; in2->data[1] = dUV; in2->linesize[0] = in2->linesize[1] = pitch; in2->width = W; in2->height = H;
    const int s5 = vpp.Convert(in2, out, p3, "cli"); // (takes the pooled buffer again; freed below the reference's way)
    printf("ok %d %d second=%d third=%d input_unref=%d release=%d reuse=%d same=%d foreign=%d twice=%d again=%d\n", out->width, out->height, s2, s3, in->data[0] == nullptr ? 1 : 0, r1,
           reuse, same, r_foreign, r_twice, s5);
    (void)hipDeviceSynchronize();
    (void)hipFree(out->opaque);
    vpp.Close();
    (void)hipFree(dY);
    (void)hipFree(dUV);
    return 0;
}
