// vpp_goldens -- replays CRC-32 known answers through the C++ class exactly the way the reference's VPP test harness drives it
// (tests/src/VPPTests.cpp:101-132, fourCCTest): a fresh VideoProcessor per case, Init(), Convert(input, converted, FrameParameters,
// "visualize"), CRC of the device result (converted->opaque), DumpFrame() into a file, CRC of the file.  Driven by
// tests/test_cpp_goldens_gpu.py with the reference's 38 literals (tests/golden/reference_crcs.py) on frame 0 of its own clip.
//   vpp_goldens frame.nv12 W H table.txt      table line: fourcc planes dstW dstH resize cropL cropT cropR cropB crc [crc2]
// Prints "<line> ok|FAIL <crc of opaque> <crc of the dump>" per case; exit code = number of failures.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "VideoProcessor.h"

// av_crc(av_crc_get_table(AV_CRC_32_IEEE), -1, buf, n) as the reference's tests call it: libavutil's table is the MSB-first
// polynomial 0x04C11DB7 run on a byte-swapped state, no final xor (validated by the decoder test's plane CRCs, which this
// program checks first: tests/src/DecoderTests.cpp:63-65).
static uint32_t crc32_av(const uint8_t *buf, size_t n) {
    uint32_t c = __builtin_bswap32(0xFFFFFFFFu);
    for (size_t i = 0; i < n; i++) {
        c ^= (uint32_t)buf[i] << 24;
        for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
    }
    return __builtin_bswap32(c);
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: vpp_goldens frame.nv12 W H table.txt\n"); return 200; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<uint8_t> host((size_t)W * H * 3 / 2);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(host.data(), 1, host.size(), f) != host.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 201; }
    fclose(f);
    printf("input Y %u UV %u\n", crc32_av(host.data(), (size_t)W * H), crc32_av(host.data() + (size_t)W * H, (size_t)W * H / 2));
    uint8_t *dY = nullptr, *dUV = nullptr;
    if (hipMalloc(&dY, (size_t)W * H) != hipSuccess || hipMalloc(&dUV, (size_t)W * H / 2) != hipSuccess) return 202;
    (void)hipMemcpy(dY, host.data(), (size_t)W * H, hipMemcpyHostToDevice);
    (void)hipMemcpy(dUV, host.data() + (size_t)W * H, (size_t)W * H / 2, hipMemcpyHostToDevice);

    FILE *tab = fopen(argv[4], "r");
    if (!tab) return 203;
    char line[256];
    int failures = 0, index = 0;
    while (fgets(line, sizeof(line), tab)) {
        int fcc, planes, dw, dh, rt, cl, ct, cr, cb;
        unsigned long c1 = 0, c2 = 0;
        const int got = sscanf(line, "%d %d %d %d %d %d %d %d %d %lu %lu", &fcc, &planes, &dw, &dh, &rt, &cl, &ct, &cr, &cb, &c1, &c2);
        if (got < 10) continue;
        if (got < 11) c2 = c1;
        VideoProcessor vpp;
        if (vpp.Init(std::make_shared<Logger>()) != 0) return 204;
        AVFrame *input = av_frame_alloc(), *converted = av_frame_alloc();
        input->data[0] = dY;
        input->data[1] = dUV;
        input->linesize[0] = input->linesize[1] = W; // the decoder test copies the planes tight
        input->width = W;
        input->height = H;
        ColorOptions color((FourCC)fcc);
        color.planesPos = (Planes)planes;
        ResizeOptions resize(dw, dh);
        resize.type = (ResizeType)rt;
        CropOptions crop({ cl, ct }, { cr, cb });
        FrameParameters args(resize, color, crop);
        const int sts = vpp.Convert(input, converted, args, "visualize"); // (consumes the input reference)
        uint32_t crc_dev = 0, crc_file = 0;
        if (sts == 0) {
            const float channels = channelsByFourCC((FourCC)fcc);
            const size_t n = (size_t)(converted->width * converted->height * channels);
            std::vector<uint8_t> out(n);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(out.data(), converted->opaque, n, hipMemcpyDeviceToHost);
            crc_dev = crc32_av(out.data(), n);
            const char *dumpName = "DumpFrame_goldens.yuv";
            {
                std::shared_ptr<FILE> w(fopen(dumpName, "wb"), fclose);
                FrameParameters dumpArgs = args; // DumpFrame sizes the dump by the resize size, else the crop size: pass what Convert produced
                dumpArgs.resize.width = (unsigned)converted->width;
                dumpArgs.resize.height = (unsigned)converted->height;
                if (vpp.DumpFrame(static_cast<uint8_t *>(converted->opaque), dumpArgs, w) != 0) crc_file = 1;
            }
            {
                std::shared_ptr<FILE> r(fopen(dumpName, "rb"), fclose);
                std::vector<uint8_t> back(n);
                if (r && fread(back.data(), 1, n, r.get()) == n) crc_file = crc32_av(back.data(), n);
            }
            remove(dumpName);
            (void)hipFree(converted->opaque);
        }
        const bool ok = sts == 0 && (crc_dev == c1 || crc_dev == c2) && crc_file == crc_dev;
        printf("%d %s %u %u status=%d size=%dx%d\n", index, ok ? "ok" : "FAIL", crc_dev, crc_file, sts, converted->width, converted->height);
        if (!ok) failures++;
        index++;
        vpp.Close();
        av_frame_free(&input);
        av_frame_free(&converted);
    }
    fclose(tab);
    (void)hipFree(dY);
    (void)hipFree(dUV);
    return failures;
}
