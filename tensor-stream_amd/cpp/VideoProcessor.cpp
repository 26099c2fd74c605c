// VideoProcessor.cpp -- adapter from the reference's C++ class to the C ABI.  See VideoProcessor.h.
#include "VideoProcessor.h"

#include <hip/hip_runtime.h>

#include <iostream>
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
    CHECK_STATUS(tsvpp_consumer_stream(ctx, consumerName.c_str(), &stream)); // pool exhausted -> VREADER_ERROR
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
    void *dst = nullptr;
    CHECK_STATUS((int)hipMalloc(&dst, bytes)); // reference ownership: the caller frees output->opaque
    int sts = ConvertInto(input, dst, options, consumerName);
    if (sts != VREADER_OK) {
        (void)hipFree(dst);
        return sts;
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

void VideoProcessor::Close() {
    if (isClosed) return;
    tsvpp_destroy(ctx);
    ctx = nullptr;
    isClosed = true;
}
