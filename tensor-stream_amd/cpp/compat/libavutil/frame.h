/* compat/libavutil/frame.h -- COMPILE-CHECK stand-in for FFmpeg's <libavutil/frame.h> (round 6).
 *
 * This image has no FFmpeg, so until round 6 the TSVPP_HAVE_LIBAV branch of VideoProcessor.h (the one a real integration takes: AVFrame from libavutil instead
 * of the class's own stand-in) had never been through a compiler.  `make -C tensor-stream_amd/cpp libav-check` compiles VideoProcessor.cpp and vpp_goldens.cpp with
 * this directory on the include path, objects only -- nothing is linked against it and nothing ships it.
 *
 * NOT FFmpeg's header and NOT layout compatible with any FFmpeg release: it declares, in FFmpeg's spelling, the subset of the public API the adapter uses
 * (reference include/VideoProcessor.h:3-5, src/VideoProcessor.cpp:94-166: data[], linesize[], width, height, opaque; av_frame_alloc / av_frame_unref /
 * av_frame_free) so that a build against the real header meets no surprise in OUR code.  Written from the API's documentation, no FFmpeg text copied. */
#ifndef TSVPP_COMPAT_AVUTIL_FRAME_H
#define TSVPP_COMPAT_AVUTIL_FRAME_H
#include <stdint.h>

#define AV_NUM_DATA_POINTERS 8

typedef struct AVFrame {
    uint8_t *data[AV_NUM_DATA_POINTERS]; /* plane pointers (NV12 on the device: data[0] = Y, data[1] = interleaved UV) */
    int linesize[AV_NUM_DATA_POINTERS];  /* bytes per row of each plane */
    uint8_t **extended_data;
    int width, height;
    int nb_samples;
    int format;
    int64_t pts;
    void *opaque; /* "for some private data of the user": the reference returns its result here */
} AVFrame;

AVFrame *av_frame_alloc(void);
void av_frame_free(AVFrame **frame);
void av_frame_unref(AVFrame *frame);
int av_frame_ref(AVFrame *dst, const AVFrame *src);

#endif
