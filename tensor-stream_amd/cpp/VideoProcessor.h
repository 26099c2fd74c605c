// VideoProcessor.h -- the reference's VPP class, re-hosted on the MI355X C ABI (include/tsvpp.h).
//
// Same public surface as reference include/VideoProcessor.h:20-149 (enum values, option structs,
// class methods and argument meaning) for `TensorStream::getFrame` (reference src/Wrappers/WrapperPython.cpp:312,
// src/Wrappers/WrapperC.cpp:291); cudaStream_t becomes hipStream_t.  No gtest and no FFmpeg exist in this image, so the reference's
// VPP tests themselves are not compiled: vpp_goldens.cpp replays their harness (tests/src/VPPTests.cpp:101-132 -- Init, Convert,
// CRC of opaque, DumpFrame, CRC of the file) with all 38 CRC literals of the reference (tests/test_cpp_goldens_gpu.py).
// Everything device-side lives behind tsvpp_*; this file is a thin adapter.
//
// AVFrame: when FFmpeg's <libavutil/frame.h> is on the include path it is used; otherwise (this
// image has no FFmpeg) a stand-in with the few fields Convert() touches keeps the class buildable
// and testable -- data[0], data[1], linesize[0], linesize[1], width, height, opaque.
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#if defined(__has_include)
#if __has_include(<libavutil/frame.h>)
extern "C" {
#include <libavutil/frame.h>
}
#define TSVPP_HAVE_LIBAV 1
#endif
#endif
#ifndef TSVPP_HAVE_LIBAV
struct AVFrame { // stand-in, NOT layout compatible with FFmpeg's
    uint8_t *data[8] = {};
    int linesize[8] = {};
    int width = 0, height = 0;
    void *opaque = nullptr;
};
inline AVFrame *av_frame_alloc() { return new AVFrame(); }
inline void av_frame_unref(AVFrame *f) { if (f) *f = AVFrame(); }
inline void av_frame_free(AVFrame **f) { if (f && *f) { delete *f; *f = nullptr; } }
#endif

#include <hip/hip_runtime.h>

#include "tsvpp.h"

// ---- reference include/Common.h:19-24, 29-34, 62-70 -------------------------------------------
enum Internal { VREADER_ERROR = -3, VREADER_UNSUPPORTED = -2, VREADER_REPEAT = -1, VREADER_OK = 0 };
enum LogsLevel { NONE, LOW, MEDIUM, HIGH };
class Logger {
public:
    void initialize(LogsLevel level, std::string logName = "logs.txt") { logsLevel = level; logFileName = logName; }
    std::string logFileName;
    LogsLevel logsLevel = LogsLevel::NONE;
    bool enableNVTX = false; // honoured as roctx ranges when the tracer is linked in
};

// ---- reference include/VideoProcessor.h:20-105 (same enumerator values, same defaults) ---------
enum FourCC { Y800 = 0, RGB24, BGR24, NV12, UYVY, YUV444, HSV };
enum Planes { PLANAR = 0, MERGED };
enum ResizeType { NEAREST = 0, BILINEAR, BICUBIC, AREA };

struct ColorOptions {
    ColorOptions(FourCC fourCC = FourCC::RGB24) : normalization(fourCC == FourCC::HSV), planesPos(Planes::MERGED), dstFourCC(fourCC) {}
    bool normalization;
    Planes planesPos;
    FourCC dstFourCC;
};
struct ResizeOptions {
    ResizeOptions(int w = 0, int h = 0) : width((unsigned)w), height((unsigned)h), type(ResizeType::NEAREST) {}
    unsigned int width, height; // 0 = no resize
    ResizeType type;
};
struct CropOptions {
    CropOptions(std::tuple<int, int> lt = { 0, 0 }, std::tuple<int, int> rb = { 0, 0 }) : leftTopCorner(lt), rightBottomCorner(rb) {}
    std::tuple<int, int> leftTopCorner, rightBottomCorner; // (x, y); empty box = no crop
};
struct FrameParameters {
    FrameParameters(ResizeOptions r = ResizeOptions(), ColorOptions c = ColorOptions(), CropOptions k = CropOptions()) : resize(r), color(c), crop(k) {}
    ResizeOptions resize;
    ColorOptions color;
    CropOptions crop;
};

float channelsByFourCC(FourCC fourCC);
float channelsByFourCC(std::string fourCC);

// ---- the reference's public stage launchers (include/VideoProcessor.h:110-115), source-compatible ----------------------
// The fused kernels have no stages; these thin functions run ONE stage each through the same C ABI (NV12 in, NV12 out for
// crop / resize) with the reference's buffer contract, so code that called the stages directly keeps compiling and the stages
// can be compared one by one (tests/test_cpp_stages_gpu.py).  `maxThreadsPerBlock` is accepted and ignored; the work is
// enqueued on *stream of the calling thread's current device (a process-wide context per device, created on first use).
//   cropHost:      dst->data[0] / data[1] = fresh device buffers (cw * ch, cw * ch / 2) holding the box; dst sizes / linesize untouched
//                  (reference src/Crop.cu:23-48; the caller sets dst->width / height and later hipFree()s both buffers)
//   resizeKernel:  src->width / height / linesize (0 = width) describe the input; dst->data[0] / data[1] = fresh buffers of
//                  resize.width x resize.height; `crop` = the input buffers came from cropHost and are freed here (src/Resize.cu:408-473)
//   colorConversionKernel<T>: input = src planes of dst->width x dst->height (pitch src->linesize[0] or that width);
//                  dst->opaque = fresh buffer of channels * w * h elements of T (src/ColorConversion.cu:280-382); T = float iff
//                  color.normalization (or HSV), as VideoProcessor::Convert calls it
int cropHost(AVFrame *src, AVFrame *dst, CropOptions crop, int maxThreadsPerBlock, hipStream_t *stream);
int resizeKernel(AVFrame *src, AVFrame *dst, bool crop, ResizeOptions resize, int maxThreadsPerBlock, hipStream_t *stream);
template <class T> int colorConversionKernel(AVFrame *src, AVFrame *dst, ColorOptions color, int maxThreadsPerBlock, hipStream_t *stream);

class VideoProcessor {
public:
    // Init(logger, maxConsumers, enableDumps): reference src/VideoProcessor.cpp:79-92.  `device` is an
    // addition (the reference always queries device 0): -1 = the calling thread's current HIP device.
    int Init(std::shared_ptr<Logger> logger, uint8_t maxConsumers = 5, bool enableDumps = false, int device = -1);
    // Convert: reference src/VideoProcessor.cpp:94-166.  Reads input->data[0..1], linesize[0..1], width,
    // height; writes output->opaque (device memory the CALLER frees with hipFree, exactly like the
    // reference's cudaMalloc'ed result), output->width/height; consumes `input` (av_frame_unref).
    // Work is enqueued on the stream owned by `consumerName`; nothing is freed or synchronised here.
    int Convert(AVFrame *input, AVFrame *output, FrameParameters &options, std::string consumerName);
    // Same conversion into caller-owned memory (no allocation at all) -- what a torch-backed
    // getFrame should call with tensor.data_ptr().
    int ConvertInto(AVFrame *input, void *deviceOut, FrameParameters &options, std::string consumerName, int *outWidth = nullptr, int *outHeight = nullptr);
    // Hands a result of Convert (output->opaque) BACK to the processor instead of hipFree()ing it (round 6).  hipFree stays legal -- it is the reference's
    // contract (c_examples/src/Sample.cpp:27,36) -- but it costs a device-wide synchronisation and the next Convert a hipMalloc: 120-180 us a frame with millisecond
    // outliers, where the conversion itself takes ~6.  A released buffer is reused by the next Convert that needs the same number of bytes: no allocator call in
    // the steady state.  `stream`: the stream on which the caller's LAST use of the buffer was enqueued (nullptr = the legacy default stream); the reuse is ordered
    // behind an event recorded there, nothing is synchronised.  Buffers still pooled at Close() are freed.  VREADER_ERROR for a pointer Convert did not hand out.
    int Release(void *opaque, hipStream_t stream = nullptr);
    template <class T> int DumpFrame(T *output, FrameParameters options, std::shared_ptr<FILE> dumpFile);
    void Close();
    ~VideoProcessor() { Close(); }
    tsvpp_ctx *context() const { return ctx; }

private:
    bool enableDumps = false;
    tsvpp_ctx *ctx = nullptr;
    std::mutex dumpSync;
    // result buffers: what Convert handed out (pointer -> bytes) and what Release took back (bytes -> buffers, each with the event of its release)
    struct Pooled { void *ptr; hipEvent_t released; };
    std::mutex poolSync;
    std::unordered_map<void *, size_t> handedOut;
    std::unordered_map<size_t, std::vector<Pooled>> pool;
    std::vector<hipEvent_t> spareEvents;
    bool isClosed = true;
    std::shared_ptr<Logger> logger;
};
