// VideoProcessor.cpp -- adapter from the reference's C++ class to the C ABI.  See VideoProcessor.h.
#include "VideoProcessor.h"

#include <hip/hip_runtime.h>

#include <iostream>
#include <mutex>
#include <thread>

namespace {
// reference include/Common.h:107-114 (CHECK_STATUS): report where, return the code
int report(int status, const char *func, int line) {
    if (status != 0) {
        std::cout << "TID: " << std::this_thread::get_id() << " Error status != 0, status: " << status << " ("
                  << tsvpp_strerror(status) << ")\n"
                  << "TID: " << std::this_thread::get_id() << " " << __FILE__ << " " << func << " " << line << "\n"
                  << std::flush;
    }
    return status;
}
#define CHECK_STATUS(s)                                              \
    do {                                                             \
        int s_ = (s);                                                \
        if (s_ != 0) return report(s_, __FUNCTION__, __LINE__);      \
    } while (0)

tsvpp_params flatten(const FrameParameters &o) {
    tsvpp_params p{};
    p.crop_left = std::get<0>(o.crop.leftTopCorner);
    p.crop_top = std::get<1>(o.crop.leftTopCorner);
    p.crop_right = std::get<0>(o.crop.rightBottomCorner);
    p.crop_bottom = std::get<1>(o.crop.rightBottomCorner);
    p.dst_width = (int)o.resize.width;
    p.dst_height = (int)o.resize.height;
    p.resize_type = (int)o.resize.type;
    p.fourcc = (int)o.color.dstFourCC;
    p.planes = (int)o.color.planesPos;
    p.normalization = o.color.normalization ? 1 : 0;
    return p;
}
} // namespace

float channelsByFourCC(FourCC fourCC) { return tsvpp_channels((int)fourCC); }
float channelsByFourCC(std::string fourCC) {
    if (fourCC == "Y800") return 1;
    if (fourCC == "UYVY") return 2;
    if (fourCC == "NV12") return 1.5f;
    return 3;
}

// ---- stage launchers ------------------------------------------------------------------------------------------------------
namespace {
// one context per device for the free functions (the class owns its own): created on first use, lives as long as the process
tsvpp_ctx *stage_ctx() {
    static std::mutex mu;
    static tsvpp_ctx *ctxs[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!ctxs[dev] && tsvpp_create(dev, 1, &ctxs[dev]) != 0) ctxs[dev] = nullptr;
    return ctxs[dev];
}
// NV12 stage: `in` -> tight NV12 of the request's output size, handed out as TWO device buffers the caller frees one by one
// (the reference's cudaMalloc pair): the conversion writes Y and UV into one buffer, the UV part moves to its own
int nv12_stage(const tsvpp_nv12 &in, tsvpp_params p, AVFrame *dst, hipStream_t stream) {
    tsvpp_ctx *ctx = stage_ctx();
    if (!ctx) CHECK_STATUS(VREADER_ERROR);
    p.fourcc = TSVPP_NV12;
    p.planes = TSVPP_PLANAR;
    p.normalization = 0;
    int w = 0, h = 0;
    CHECK_STATUS(tsvpp_out_dims(&p, in.width, in.height, &w, &h));
    uint8_t *y = nullptr, *uv = nullptr;
    CHECK_STATUS((int)hipMalloc((void **)&y, (size_t)w * h * 3 / 2));
    if (hipMalloc((void **)&uv, (size_t)w * h / 2) != hipSuccess) {
        (void)hipFree(y);
        CHECK_STATUS(VREADER_ERROR);
    }
    int sts = tsvpp_convert(ctx, &in, &p, y, stream);
    if (sts == 0) sts = (int)hipMemcpyAsync(uv, y + (size_t)w * h, (size_t)w * h / 2, hipMemcpyDeviceToDevice, stream);
    if (sts != 0) {
        (void)hipFree(y);
        (void)hipFree(uv);
        CHECK_STATUS(sts);
    }
    dst->data[0] = y;
    dst->data[1] = uv;
    return VREADER_OK;
}
} // namespace

int cropHost(AVFrame *src, AVFrame *dst, CropOptions crop, int, hipStream_t *stream) {
    if (!src || !dst || !stream) CHECK_STATUS(VREADER_ERROR);
    const int left = std::get<0>(crop.leftTopCorner), top = std::get<1>(crop.leftTopCorner);
    const int cw = std::get<0>(crop.rightBottomCorner) - left, ch = std::get<1>(crop.rightBottomCorner) - top;
    const int py = src->linesize[0] ? src->linesize[0] : src->width, puv = src->linesize[1] ? src->linesize[1] : src->width;
    if (cw <= 0 || ch <= 0 || left < 0 || top < 0) CHECK_STATUS(VREADER_ERROR);
    // the reference's cropKernel IS pointer arithmetic: luma (left + j, top + i), chroma row top / 2 + i / 2, chroma byte (j & ~1) + left
    // (an odd `left` swaps U and V there too, src/Crop.cu:10-18) -- so the box is described as a frame of its own, whatever its size
    const tsvpp_nv12 in{ src->data[0] + (size_t)top * py + left, src->data[1] + (size_t)(top / 2) * puv + left, py, puv, cw, ch };
    tsvpp_params p{}; // no crop, no resize: the planes themselves
    return nv12_stage(in, p, dst, *stream);
}

int resizeKernel(AVFrame *src, AVFrame *dst, bool crop, ResizeOptions resize, int, hipStream_t *stream) {
    if (!src || !dst || !stream) CHECK_STATUS(VREADER_ERROR);
    // reference src/Resize.cu:465-471: with crop == true the buffers that `dst` held on entry (cropHost's pair -- Convert passes src == dst) are
    // freed, then dst->data[0..1] are replaced; src's own buffers are never touched.  Both pairs are captured before anything is overwritten.
    uint8_t *in_y = src->data[0], *in_uv = src->data[1];
    uint8_t *dst_old_y = dst->data[0], *dst_old_uv = dst->data[1];
    const tsvpp_nv12 in{ in_y, in_uv, src->linesize[0] ? src->linesize[0] : src->width, src->linesize[1] ? src->linesize[1] : src->width, src->width, src->height };
    tsvpp_params p{};
    p.dst_width = (int)resize.width;
    p.dst_height = (int)resize.height;
    p.resize_type = (int)resize.type;
    CHECK_STATUS(nv12_stage(in, p, dst, *stream));
    if (crop) { // (hipFree waits for the work above, which may still read the pair when src == dst)
        CHECK_STATUS((int)hipFree(dst_old_y));
        CHECK_STATUS((int)hipFree(dst_old_uv));
    }
    return VREADER_OK;
}

template <class T> int colorConversionKernel(AVFrame *src, AVFrame *dst, ColorOptions color, int, hipStream_t *stream) {
    if (!src || !dst || !stream) CHECK_STATUS(VREADER_ERROR);
    tsvpp_ctx *ctx = stage_ctx();
    if (!ctx) CHECK_STATUS(VREADER_ERROR);
    const bool f32 = color.normalization || color.dstFourCC == FourCC::HSV;
    if (f32 != (sizeof(T) == sizeof(float))) CHECK_STATUS(VREADER_UNSUPPORTED); // Convert() picks <float> iff normalization
    const int w = dst->width, h = dst->height;
    const tsvpp_nv12 in{ src->data[0], src->data[1], src->linesize[0] ? src->linesize[0] : w, src->linesize[1] ? src->linesize[1] : w, w, h };
    tsvpp_params p{};
    p.fourcc = (int)color.dstFourCC;
    p.planes = (int)color.planesPos;
    p.normalization = color.normalization ? 1 : 0;
    const size_t bytes = tsvpp_out_bytes(&p, w, h);
    if (bytes == 0) CHECK_STATUS(VREADER_UNSUPPORTED);
    void *out = nullptr;
    CHECK_STATUS((int)hipMalloc(&out, bytes));
    const int sts = tsvpp_convert(ctx, &in, &p, out, *stream);
    if (sts != 0) {
        (void)hipFree(out);
        CHECK_STATUS(sts);
    }
    dst->opaque = out;
    return VREADER_OK;
}
template int colorConversionKernel<float>(AVFrame *, AVFrame *, ColorOptions, int, hipStream_t *);
template int colorConversionKernel<unsigned char>(AVFrame *, AVFrame *, ColorOptions, int, hipStream_t *);

int VideoProcessor::Init(std::shared_ptr<Logger> log, uint8_t maxConsumers, bool dumps, int device) {
    if (!isClosed) Close();
    enableDumps = dumps;
    logger = log;
    if (device < 0) CHECK_STATUS((int)hipGetDevice(&device));
    CHECK_STATUS(tsvpp_create(device, maxConsumers, &ctx));
    if (logger && logger->enableNVTX) (void)tsvpp_enable_markers(ctx, 1); // roctx ranges; silently absent without a tracer library
    isClosed = false;
    return VREADER_OK;
}

int VideoProcessor::ConvertInto(AVFrame *input, void *deviceOut, FrameParameters &options, std::string consumerName, int *outW, int *outH) {
    if (isClosed || !input || !deviceOut) CHECK_STATUS(VREADER_ERROR);
    void *stream = nullptr;
    // the consumer's stream -- or, under TSVPP_OPT_INPUTS_READY, the one of its two streams whose turn it is (include/tsvpp.h; one frame: always "small")
    CHECK_STATUS(tsvpp_consumer_next_stream(ctx, consumerName.c_str(), 0, &stream)); // pool exhausted -> VREADER_ERROR
    const tsvpp_nv12 in{ input->data[0], input->data[1], input->linesize[0], input->linesize[1], input->width, input->height };
    const tsvpp_params p = flatten(options);
    int w = 0, h = 0;
    CHECK_STATUS(tsvpp_out_dims(&p, in.width, in.height, &w, &h));
    CHECK_STATUS(tsvpp_convert(ctx, &in, &p, deviceOut, stream));
    if (outW) *outW = w;
    if (outH) *outH = h;
    return VREADER_OK;
}

int VideoProcessor::Convert(AVFrame *input, AVFrame *output, FrameParameters &options, std::string consumerName) {
    if (isClosed || !input || !output) CHECK_STATUS(VREADER_ERROR);
    const tsvpp_params p = flatten(options);
    const size_t bytes = tsvpp_out_bytes(&p, input->width, input->height);
    int w = 0, h = 0;
    CHECK_STATUS(tsvpp_out_dims(&p, input->width, input->height, &w, &h)); // also the error path of out_bytes == 0
    // the result buffer: one that Release() took back (same size), else a fresh allocation -- reference ownership either way: the caller frees output->opaque
    // (hipFree) or hands it back (Release)
    void *dst = nullptr;
    hipEvent_t released = nullptr;
    {
        std::lock_guard<std::mutex> lk(poolSync);
        auto it = pool.find(bytes);
        if (it != pool.end() && !it->second.empty()) {
            dst = it->second.back().ptr;
            released = it->second.back().released;
            it->second.pop_back();
        }
    }
    if (!dst) CHECK_STATUS((int)hipMalloc(&dst, bytes));
    if (released) { // the consumer's streams wait for the releasing stream's last use of the buffer (no host synchronisation)
        void *s0 = nullptr;
        if (tsvpp_consumer_stream(ctx, consumerName.c_str(), &s0) == 0) (void)hipStreamWaitEvent((hipStream_t)s0, released, 0);
        int ready = 0;
        if (tsvpp_get_option(ctx, TSVPP_OPT_INPUTS_READY, &ready) == 0 && ready) (void)hipEventSynchronize(released); // (two streams, barrier-free launches: the promise is "nobody uses the output")
        std::lock_guard<std::mutex> lk(poolSync);
        spareEvents.push_back(released);
    }
    int sts = ConvertInto(input, dst, options, consumerName);
    if (sts != VREADER_OK) {
        (void)hipFree(dst);
        return sts;
    }
    {
        std::lock_guard<std::mutex> lk(poolSync);
        if (handedOut.size() > 65536) handedOut.clear(); // a caller that only ever hipFree()s leaves its entries behind: bounded (Release then answers VREADER_ERROR for the forgotten ones)
        handedOut[dst] = bytes;
    }
    output->opaque = dst;
    output->width = w;
    output->height = h;
    // reference src/VideoProcessor.cpp:132-135: with neither crop nor resize the options receive the input size
    if (options.resize.width == 0 || options.resize.height == 0) {
        if (w == input->width && h == input->height) {
            options.resize.width = (unsigned)w;
            options.resize.height = (unsigned)h;
        }
    }
    if (enableDumps) {
        std::string fileName = std::string("Processed_") + consumerName + std::string(".yuv");
        std::shared_ptr<FILE> dumpFile(fopen(fileName.c_str(), "ab"), std::fclose);
        std::unique_lock<std::mutex> locker(dumpSync);
        FrameParameters dumpOpts = options;
        dumpOpts.resize.width = (unsigned)w;
        dumpOpts.resize.height = (unsigned)h;
        if (options.color.normalization) DumpFrame(static_cast<float *>(output->opaque), dumpOpts, dumpFile);
        else DumpFrame(static_cast<unsigned char *>(output->opaque), dumpOpts, dumpFile);
    }
    av_frame_unref(input);
    return VREADER_OK;
}

// reference src/VideoProcessor.cpp:28-72: device -> host copy of the tight output, appended to the file.
// Size rule as the reference: resize size if given, else crop size.
template <class T> int VideoProcessor::DumpFrame(T *output, FrameParameters options, std::shared_ptr<FILE> dumpFile) {
    const float channels = channelsByFourCC(options.color.dstFourCC);
    int w = 0, h = 0;
    const int cw = std::get<0>(options.crop.rightBottomCorner) - std::get<0>(options.crop.leftTopCorner);
    const int ch = std::get<1>(options.crop.rightBottomCorner) - std::get<1>(options.crop.leftTopCorner);
    if (cw > 0 && ch > 0) { w = cw; h = ch; }
    if (options.resize.width > 0 && options.resize.height > 0) { w = (int)options.resize.width; h = (int)options.resize.height; }
    const size_t n = (size_t)(channels * w * h);
    std::vector<T> host(n);
    CHECK_STATUS((int)hipDeviceSynchronize()); // the conversion is asynchronous
    CHECK_STATUS((int)hipMemcpy(host.data(), output, n * sizeof(T), hipMemcpyDeviceToHost));
    if (!dumpFile) CHECK_STATUS(VREADER_ERROR);
    fwrite(host.data(), n, sizeof(T), dumpFile.get());
    fflush(dumpFile.get());
    return VREADER_OK;
}
template int VideoProcessor::DumpFrame(float *, FrameParameters, std::shared_ptr<FILE>);
template int VideoProcessor::DumpFrame(uint8_t *, FrameParameters, std::shared_ptr<FILE>);

int VideoProcessor::Release(void *opaque, hipStream_t stream) {
    if (isClosed || !opaque) CHECK_STATUS(VREADER_ERROR);
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(poolSync);
        auto it = handedOut.find(opaque);
        if (it == handedOut.end()) CHECK_STATUS(VREADER_ERROR); // not a result of this processor's Convert (or released twice)
        bytes = it->second;
        handedOut.erase(it);
        if (!spareEvents.empty()) {
            ev = spareEvents.back();
            spareEvents.pop_back();
        }
    }
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (!ev || hipEventRecord(ev, stream) != hipSuccess) { // cannot order the reuse: fall back to the reference's contract
        if (ev) (void)hipEventDestroy(ev);
        CHECK_STATUS((int)hipFree(opaque));
        return VREADER_OK;
    }
    std::lock_guard<std::mutex> lk(poolSync);
    pool[bytes].push_back(Pooled{ opaque, ev });
    return VREADER_OK;
}

void VideoProcessor::Close() {
    if (isClosed) return;
    {
        std::lock_guard<std::mutex> lk(poolSync);
        for (auto &kv : pool)
            for (Pooled &b : kv.second) {
                (void)hipFree(b.ptr);
                (void)hipEventDestroy(b.released);
            }
        pool.clear();
        for (hipEvent_t e : spareEvents) (void)hipEventDestroy(e);
        spareEvents.clear();
        handedOut.clear(); // (buffers the caller still holds are the caller's to hipFree)
    }
    tsvpp_destroy(ctx);
    ctx = nullptr;
    isClosed = true;
}
