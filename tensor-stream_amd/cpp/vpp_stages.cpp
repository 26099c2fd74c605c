// vpp_stages -- drives the three public stage launchers the way the reference's VideoProcessor::Convert chains them
// (src/VideoProcessor.cpp:94-166): cropHost -> sizes := box; resizeKernel(crop = true) -> sizes := resize; colorConversionKernel<T>,
// and writes every stage's result to a file so tests/test_cpp_stages_gpu.py can compare each with the oracle's crop-only / crop + resize /
// full conversions.
//   vpp_stages frame.nv12 W H PITCH  L T R B  DW DH TYPE  FOURCC PLANES NORM  prefix
// L..B = 0 0 0 0 skips the crop stage, DW DH = 0 0 skips the resize stage (both as Convert does).  Exit code 0 = all stages ran.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "VideoProcessor.h"

static bool dump(const std::string &name, const void *dev, size_t bytes) {
    std::vector<uint8_t> host(bytes);
    if (hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    FILE *f = fopen(name.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(host.data(), 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

int main(int argc, char **argv) {
    if (argc < 16) { fprintf(stderr, "usage: vpp_stages frame.nv12 W H PITCH L T R B DW DH TYPE FOURCC PLANES NORM prefix\n"); return 200; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), P = atoi(argv[4]);
    const int L = atoi(argv[5]), T = atoi(argv[6]), R = atoi(argv[7]), B = atoi(argv[8]);
    const int DW = atoi(argv[9]), DH = atoi(argv[10]), type = atoi(argv[11]);
    const int fcc = atoi(argv[12]), planes = atoi(argv[13]), norm = atoi(argv[14]);
    const std::string prefix = argv[15];
    std::vector<uint8_t> host((size_t)P * H * 3 / 2); // the file holds H rows of P bytes of luma, then H / 2 rows of P bytes of chroma
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(host.data(), 1, host.size(), f) != host.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 201; }
    fclose(f);
    uint8_t *dY = nullptr, *dUV = nullptr;
    if (hipMalloc(&dY, (size_t)P * H) != hipSuccess || hipMalloc(&dUV, (size_t)P * H / 2) != hipSuccess) return 202;
    (void)hipMemcpy(dY, host.data(), (size_t)P * H, hipMemcpyHostToDevice);
    (void)hipMemcpy(dUV, host.data() + (size_t)P * H, (size_t)P * H / 2, hipMemcpyHostToDevice);
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) return 203;

    AVFrame *input = av_frame_alloc(), *output = av_frame_alloc();
    input->data[0] = dY;
    input->data[1] = dUV;
    input->linesize[0] = input->linesize[1] = P;
    input->width = W;
    input->height = H;
    output->width = W;
    output->height = H;
    AVFrame *cur = input;
    bool cropped = false;
    const int cw = R - L, ch = B - T;
    if (cw > 0 && ch > 0) {
        CropOptions crop({ L, T }, { R, B });
        if (cropHost(input, output, crop, 1024, &stream) != 0) return 210;
        output->width = cw; // (Convert sets the sizes after the stage, src/VideoProcessor.cpp:118-119)
        output->height = ch;
        (void)hipStreamSynchronize(stream);
        if (!dump(prefix + ".crop.y", output->data[0], (size_t)cw * ch) || !dump(prefix + ".crop.uv", output->data[1], (size_t)cw * ch / 2)) return 211;
        cur = output;
        cropped = true;
    }
    if (DW > 0 && DH > 0) {
        ResizeOptions resize(DW, DH);
        resize.type = (ResizeType)type;
        if (resizeKernel(cur, output, cropped, resize, 1024, &stream) != 0) return 220; // frees cropHost's pair
        output->width = DW;
        output->height = DH;
        (void)hipStreamSynchronize(stream);
        if (!dump(prefix + ".resize.y", output->data[0], (size_t)DW * DH) || !dump(prefix + ".resize.uv", output->data[1], (size_t)DW * DH / 2)) return 221;
        cur = output;
    }
    ColorOptions color((FourCC)fcc);
    color.planesPos = (Planes)planes;
    color.normalization = norm != 0;
    const bool f32 = norm != 0 || fcc == (int)FourCC::HSV;
    // the wrong instantiation is refused like every unsupported request
    const int wrong = f32 ? colorConversionKernel<unsigned char>(cur, output, color, 1024, &stream) : colorConversionKernel<float>(cur, output, color, 1024, &stream);
    if (wrong != VREADER_UNSUPPORTED) return 229;
    const int sts = f32 ? colorConversionKernel<float>(cur, output, color, 1024, &stream) : colorConversionKernel<unsigned char>(cur, output, color, 1024, &stream);
    if (sts != 0) return 230;
    (void)hipStreamSynchronize(stream);
    const size_t n = (size_t)(output->width * output->height * channelsByFourCC((FourCC)fcc)) * (f32 ? 4 : 1);
    if (!dump(prefix + ".color", output->opaque, n)) return 231;
    printf("%d %d %zu\n", output->width, output->height, n);
    if (cur == output) { // the intermediates are the caller's to free, one by one (src/VideoProcessor.cpp:158-163)
        if (hipFree(output->data[0]) != hipSuccess || hipFree(output->data[1]) != hipSuccess) return 240;
    }
    if (hipFree(output->opaque) != hipSuccess) return 241;
    (void)hipFree(dY);
    (void)hipFree(dUV);
    (void)hipStreamDestroy(stream);
    av_frame_free(&input);
    av_frame_free(&output);
    return 0;
}
