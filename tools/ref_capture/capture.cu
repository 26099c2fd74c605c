// capture.cu -- five minutes on an NVIDIA box close the two parity questions this repository cannot decide: which fused-multiply-add
// pattern nvcc gave the reference's colour conversion (reference src/ColorConversion.cu:23-36: six distinguishable variants, 155 of the 2^24 (Y, U, V) triples tell
// them apart), and what its pow() makes of a non-dyadic BICUBIC weight (src/Resize.cu:45-48; the reference's own test accepts two CRCs, tests/src/PythonTests.cpp:208).
// It INCLUDES the reference's kernels, unmodified, from a checkout of osai-ai/tensor-stream v0.4.6 and runs them on two small inputs:
//   triples.bin                  (this directory)          -> g_triples_rgb.bin       RGB24 merged uint8 of a 2 x 2 block per triple
//   bbb_1080x608_frame0.nv12     (tests/golden/)           -> bicubic_480x360_nv12.bin  the resized NV12 (Y plane, then UV plane)
// Build inside the reference's own environment (its Dockerfile: CUDA 11.8 + FFmpeg 6.0), with the reference's flags (nvcc defaults: -fmad=true):
//   nvcc -O3 -std=c++14 -I$REF/include -I$REF/src $(pkg-config --cflags libavformat libavutil) -o capture capture.cu $REF/src/Common.cpp $(pkg-config --libs libavutil) -lnvToolsExt
//   ./capture triples.bin ../../tests/golden/bbb_1080x608_frame0.nv12 && cp g_triples_rgb.bin bicubic_480x360_nv12.bin ../../tests/golden/ref_capture/
// tests/test_ref_capture.py then names the variant and checks the oracle's default against it.  No line of the reference is copied here.
#include <cstdio>
#include <vector>

#include "ColorConversion.cu" // the reference's, via -I$REF/src
#include "Resize.cu"

float channelsByFourCC(FourCC fourCC) { // (lives in the reference's VideoProcessor.cpp, which drags the whole pipeline in: restated for the two formats used)
    return fourCC == Y800 ? 1.f : fourCC == UYVY ? 2.f : fourCC == NV12 ? 1.5f : 3.f;
}

static std::vector<unsigned char> slurp(const char *path) {
    std::vector<unsigned char> v;
    if (FILE *f = fopen(path, "rb")) {
        fseek(f, 0, SEEK_END);
        v.resize((size_t)ftell(f));
        fseek(f, 0, SEEK_SET);
        if (fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
        fclose(f);
    }
    return v;
}
static void dump(const char *path, const void *dev, size_t n) {
    std::vector<unsigned char> h(n);
    cudaMemcpy(h.data(), dev, n, cudaMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    fwrite(h.data(), 1, n, f);
    fclose(f);
}
static AVFrame device_frame(const std::vector<unsigned char> &nv12, int w, int h) { // tight NV12 on the device
    AVFrame f = {};
    unsigned char *d = nullptr;
    cudaMalloc(&d, nv12.size());
    cudaMemcpy(d, nv12.data(), nv12.size(), cudaMemcpyHostToDevice);
    f.data[0] = d;
    f.data[1] = d + (size_t)w * h;
    f.linesize[0] = f.linesize[1] = w;
    f.width = w;
    f.height = h;
    return f;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    cudaStream_t stream;
    cudaStreamCreate(&stream);
    const std::vector<unsigned char> trip = slurp(argv[1]), bbb = slurp(argv[2]);
    const int n = (int)trip.size() / 3, w = 2 * n;
    if (n == 0 || bbb.size() != 1080u * 608u * 3u / 2u) return 3;
    std::vector<unsigned char> nv12((size_t)w * 3); // two luma rows + one chroma row: triple k owns columns 2 k, 2 k + 1
    for (int k = 0; k < n; k++) {
        nv12[2 * k] = nv12[2 * k + 1] = nv12[w + 2 * k] = nv12[w + 2 * k + 1] = trip[3 * k];
        nv12[2 * w + 2 * k] = trip[3 * k + 1];
        nv12[2 * w + 2 * k + 1] = trip[3 * k + 2];
    }
    AVFrame src = device_frame(nv12, w, 2), dst = {};
    dst.width = w;
    dst.height = 2;
    ColorOptions color(RGB24);
    color.planesPos = Planes::MERGED;
    color.normalization = false;
    colorConversionKernel<unsigned char>(&src, &dst, color, 1024, &stream);
    cudaStreamSynchronize(stream);
    dump("g_triples_rgb.bin", dst.opaque, (size_t)w * 2 * 3);
    AVFrame big = device_frame(bbb, 1080, 608), small = {};
    ResizeOptions resize(480, 360);
    resize.type = ResizeType::BICUBIC;
    resizeKernel(&big, &small, false, resize, 1024, &stream);
    cudaStreamSynchronize(stream);
    dump("bicubic_480x360_nv12.bin", small.data[0], 480u * 360u);
    FILE *f = fopen("bicubic_480x360_nv12.bin", "ab");
    std::vector<unsigned char> uv(480u * 180u);
    cudaMemcpy(uv.data(), small.data[1], uv.size(), cudaMemcpyDeviceToHost);
    fwrite(uv.data(), 1, uv.size(), f);
    fclose(f);
    printf("wrote g_triples_rgb.bin (%d triples) and bicubic_480x360_nv12.bin\n", n);
    return 0;
}
