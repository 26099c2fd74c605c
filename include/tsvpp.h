/*
 * tsvpp.h -- C ABI of the MI355X-native Video Post Processing (VPP) path.
 *
 * Drop-in boundary for the hot path of osai-ai/tensor-stream:
 *     NV12 crop -> nearest/bilinear/bicubic/area resize -> YUV->RGB24/BGR24
 *     (+ fp32 normalise, planar/merged)
 * i.e. everything that sits behind `VideoProcessor::Convert()`
 * (reference src/VideoProcessor.cpp:94-166) and its three CUDA launchers
 * `cropHost` (src/Crop.cu:23-48), `resizeKernel` (src/Resize.cu:408-473) and
 * `colorConversionKernel<T>` (src/ColorConversion.cu:280-382).
 *
 * The reference has no C ABI (its boundary is the C++ class); these entry points
 * are what a `VideoProcessor` built for ROCm binds instead of the CUDA launchers.
 * INTEGRATION.md shows the adapter; tensor-stream_amd/cpp/VideoProcessor.{h,cpp}
 * is that adapter, written out.
 *
 * Conventions
 *   - plain C types only: device pointers are `void*` / `const uint8_t*`, the HIP
 *     stream is passed as `void*` (a `hipStream_t`; NULL = the null stream);
 *   - every function returning `int` uses the reference's status convention
 *     (include/Common.h:19-24): 0 = OK, negative = VREADER_* code, positive = a
 *     `hipError_t` value (the reference returns `cudaError_t` the same way);
 *   - all work is enqueued on the given stream and is asynchronous; nothing in the
 *     per-frame path frees or synchronises, and nothing allocates once the request
 *     has been seen by tsvpp_prepare_batch (the reference does 1-5 cudaMalloc + 0-4
 *     cudaFree per frame, src/VideoProcessor.cpp:94-166).  Without a prepare call the
 *     FIRST conversion of a new AREA scale builds its weight table, and the first
 *     UYVY / YUV444 conversion behind a resize on a stream sizes that stream's NV12
 *     scratch buffer (one hipMalloc each; an outgrown buffer is kept until
 *     tsvpp_destroy, never freed under running work);
 *   - every entry point leaves the calling thread's current HIP device as it found it;
 *   - output memory is CALLER-allocated and tight: channels*W*H elements of
 *     uint8 (normalization == 0) or float (normalization != 0), layout as the
 *     reference's kernels write it (src/ColorConversion.cu:41-93).
 */
#ifndef TSVPP_H
#define TSVPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: reference include/Common.h:19-24 (enum Internal) ---- */
#define TSVPP_OK 0
#define TSVPP_REPEAT (-1)
#define TSVPP_UNSUPPORTED (-2)
#define TSVPP_ERROR (-3)

/* ---- enums: byte-compatible with reference include/VideoProcessor.h:20-28,32-35,57-62 ---- */
enum tsvpp_fourcc { TSVPP_Y800 = 0, TSVPP_RGB24 = 1, TSVPP_BGR24 = 2, TSVPP_NV12 = 3, TSVPP_UYVY = 4, TSVPP_YUV444 = 5, TSVPP_HSV = 6 };
enum tsvpp_planes { TSVPP_PLANAR = 0, TSVPP_MERGED = 1 };
enum tsvpp_resize { TSVPP_NEAREST = 0, TSVPP_BILINEAR = 1, TSVPP_BICUBIC = 2, TSVPP_AREA = 3 };

#define TSVPP_MAX_BATCH 128 /* frames per launch; larger batches are split (round 4: 64 -> 128; the pointer table travels in the kernarg segment: 3 KiB) */

/* One NV12 frame in device memory.  Mirrors the AVFrame fields Convert() reads
 * (reference src/Crop.cu:37-38, src/Resize.cu:420-421, src/ColorConversion.cu:301):
 * data[0], data[1], linesize[0], linesize[1], width, height.
 * pitch == 0 means "pitch = width", exactly as the reference's fallback. */
typedef struct tsvpp_nv12 {
    const uint8_t *y;  /* AVFrame::data[0] */
    const uint8_t *uv; /* AVFrame::data[1], interleaved U,V */
    int32_t pitch_y;   /* AVFrame::linesize[0] */
    int32_t pitch_uv;  /* AVFrame::linesize[1] */
    int32_t width;
    int32_t height;
} tsvpp_nv12;

/* Flat mirror of FrameParameters {ResizeOptions, ColorOptions, CropOptions}
 * (reference include/VideoProcessor.h:39-105).  Zero-initialised == the
 * reference defaults except fourcc/planes (reference: RGB24, MERGED). */
typedef struct tsvpp_params {
    int32_t crop_left, crop_top;     /* CropOptions::leftTopCorner  (x, y) */
    int32_t crop_right, crop_bottom; /* CropOptions::rightBottomCorner (x, y) */
    int32_t dst_width, dst_height;   /* ResizeOptions::width/height, 0 = no resize */
    int32_t resize_type;             /* enum tsvpp_resize */
    int32_t fourcc;                  /* enum tsvpp_fourcc */
    int32_t planes;                  /* enum tsvpp_planes */
    int32_t normalization;           /* ColorOptions::normalization */
} tsvpp_params;

/* The eight colour constants of reference src/ColorConversion.cu:23,25,30,35:
 * {1.163999557, 1.5959997177, 2.017999649, -0.812999725, -0.390999794, 0.5, 16, 128}.
 * They are literals in the reference; here they live in the context so that a
 * multi-GPU job can broadcast one block from rank 0 (RCCL) and every rank can
 * verify it against the compiled-in defaults. */
typedef struct tsvpp_coeffs {
    float y_scale, v_to_r, u_to_b, v_to_g, u_to_g, round_bias, y_offset, c_offset;
} tsvpp_coeffs;

typedef struct tsvpp_ctx tsvpp_ctx;

/* VideoProcessor::Init (reference src/VideoProcessor.cpp:79-92): selects `device`,
 * creates `max_consumers` streams for the named-consumer pool.  Unlike the
 * reference it queries the properties of `device`, not of device 0. */
int tsvpp_create(int device, int max_consumers, tsvpp_ctx **out_ctx);
/* VideoProcessor::Close (src/VideoProcessor.cpp:168-178) + release of cached tables. */
void tsvpp_destroy(tsvpp_ctx *ctx);

/* findFree<cudaStream_t>(consumerName, streamArr) (reference include/Common.h:225-237,
 * src/VideoProcessor.cpp:98-104): the stream bound to `name`, claiming a free slot
 * for a new name; TSVPP_ERROR when all `max_consumers` slots are taken. */
int tsvpp_consumer_stream(tsvpp_ctx *ctx, const char *name, void **out_stream);
/* The stream the consumer's NEXT conversion should be enqueued on (round 6; VideoProcessor::Convert / ConvertInto use it).  By default that is
 * tsvpp_consumer_stream's: one stream per consumer, every conversion ordered behind the previous one -- the reference's model.  Under
 * TSVPP_OPT_INPUTS_READY (below) a consumer owns TWO streams and this call alternates between them: conversion k + 1 is launched while conversion k still
 * drains (a single 1080p -> 720p frame is ~2.3 us of HBM time behind a ~1.5-1.9 us dependent-launch boundary; measured: one consumer, one frame per launch,
 * 0.27 -> 0.41 of the HBM roofline, two frames per launch 0.44 -> 0.58, four 0.47 -> 0.68, eight 0.60 -> 0.72; profiles/r06_curve_values.txt).  `launch_bytes` = the bytes the conversion about to be
 * enqueued moves (source + output bytes of all its frames; 0 = unknown, taken as one frame): only launches of at most TSVPP_OVERLAP_MAX_BYTES alternate -- two
 * LARGE launches side by side lose (64-frame launches of the headline: 0.78 -> 0.67 of the roofline), so those stay on the consumer's first stream.  The second
 * stream is created on the consumer's first call with the option set (one hipStreamCreate).  tsvpp_consumer_synchronize waits (on the host) for everything
 * enqueued on the consumer's streams. */
#define TSVPP_OVERLAP_MAX_BYTES ((size_t)256 << 20)
#define TSVPP_BARRIER_FREE_MAX_BYTES ((size_t)64 << 20) /* ... and only launches up to this size go out without the barrier bit (profiles/r06_curve_values.txt) */
int tsvpp_consumer_next_stream(tsvpp_ctx *ctx, const char *name, size_t launch_bytes, void **out_stream);
int tsvpp_consumer_synchronize(tsvpp_ctx *ctx, const char *name);

/* Stage selection of Convert() (reference src/VideoProcessor.cpp:106-135) without
 * running anything: final width/height and the tight output size in bytes. */
int tsvpp_out_dims(const tsvpp_params *p, int in_width, int in_height, int *out_width, int *out_height);
size_t tsvpp_out_bytes(const tsvpp_params *p, int in_width, int in_height);
/* channelsByFourCC (reference src/VideoProcessor.cpp:4-14). */
float tsvpp_channels(int fourcc);

/* VideoProcessor::Convert for one frame, as ONE fused kernel launch on `stream`.
 * `out` must hold tsvpp_out_bytes() bytes. */
int tsvpp_convert(tsvpp_ctx *ctx, const tsvpp_nv12 *in, const tsvpp_params *p, void *out, void *stream);

/* The same conversion for `n` independent frames of identical geometry, in
 * ceil(n / TSVPP_MAX_BATCH) launches.  `in` and `outs` are HOST arrays of n entries.
 * (Not in the reference: one 1080p frame is ~2.5 us of HBM time, below a launch.) */
int tsvpp_convert_batch(tsvpp_ctx *ctx, int n, const tsvpp_nv12 *in, const tsvpp_params *p, void *const *outs, void *stream);

/* ---- persistent frame tables (round 5; not in the reference) ------------------------------------------------------------------------------------------
 * tsvpp_convert_batch passes its (y, uv, out) pointer triples by value in the kernarg segment: no copy, no device-side descriptor -- and at most
 * TSVPP_MAX_BATCH frames per launch.  A pipeline whose decoder surfaces and output buffers are POOLS (the usual case: the same surfaces and tensors come
 * round again) can instead register them once in a device-resident table and convert any run of its entries with launches of up to
 * TSVPP_MAX_TABLE_LAUNCH frames: the small-output configurations (BASELINE C3: 1.8 MB per frame) move 0.64 of the HBM roofline in 64-frame launches and
 * 0.70+ from 256 frames on (profiles/r05_table_ab.txt); nothing is copied per call.
 *   tsvpp_table_create    `capacity` entries; every entry shares one geometry (width, height, pitches: fixed by the first tsvpp_table_set).
 *   tsvpp_table_set       entries [first, first + n) <- in[i] / outs[i] (HOST arrays); uploaded on `stream` out of pinned staging the table owns (one slot per
 *                         upload in flight, four slots: a fifth concurrent upload waits for the oldest): the arrays may be freed on return, conversions enqueued
 *                         on the same stream afterwards see the new entries, conversions enqueued BEFORE it see the old ones.
 *   tsvpp_convert_table   == tsvpp_convert_batch over entries [first, first + n), same results, same status codes.
 *   tsvpp_table_destroy   frees the table (the caller has waited for conversions that use it).  Destroying the CONTEXT first is legal: tsvpp_destroy releases the
 *                         device memory of its live tables, their handles remain valid arguments of tsvpp_table_destroy only.
 * A table belongs to the context that created it (its device).  tsvpp_table_set calls are serialised against each other; a tsvpp_convert_table that runs
 * concurrently with a tsvpp_table_set of the SAME entries from another thread is the caller's race (as two writers of one AVFrame would be), and a captured
 * hipGraph replays the table as the device holds it at replay time (entries are read by the kernels, not baked into the graph). */
#define TSVPP_MAX_TABLE_LAUNCH 1024
typedef struct tsvpp_table tsvpp_table;
int tsvpp_table_create(tsvpp_ctx *ctx, int capacity, tsvpp_table **out_table);
void tsvpp_table_destroy(tsvpp_table *table);
int tsvpp_table_set(tsvpp_table *table, int first, int n, const tsvpp_nv12 *in, void *const *outs, void *stream);
int tsvpp_convert_table(tsvpp_ctx *ctx, const tsvpp_table *table, int first, int n, const tsvpp_params *p, void *stream);

/* Pre-build everything a (params, input size) pair needs so that later tsvpp_convert* calls for it touch no
 * allocator -- e.g. before hipGraph capture: the AREA weight tables (the reference mallocs, copies and leaks them
 * per frame, src/Resize.cu:389-406,436-452) and, for UYVY / YUV444 behind a resize, the resized-NV12 scratch of
 * `stream` sized for calls of up to `n_frames` frames (the reference cudaMallocs that intermediate per frame,
 * src/Resize.cu:411-416).  BILINEAR / AREA up-scale requests also get the geometry tables of the 2x2-tap kernel (tile
 * footprints, per-column and per-row coordinates and weights, evaluated once on the host) for the tile shape a batch of
 * `n_frames` frames runs with; a conversion that was not prepared builds them on first use -- except while its stream
 * is being captured into a graph, where it never allocates and runs the kernel that computes coordinates itself (same
 * bits).  The tables are prepared for frames as the decoder delivers them: pitches that are multiples of 16 and plane
 * pointers that keep the crop origin dword-aligned; a conversion whose real pitches / alignment select another tile shape
 * builds its own set on first use (or, while capturing, runs the self-computing kernel).  A context keeps the table sets of
 * its 1024 most recently used geometries.  tsvpp_prepare == tsvpp_prepare_batch(..., 0, NULL): tables only (single-frame
 * tile shape). */
int tsvpp_prepare(tsvpp_ctx *ctx, const tsvpp_params *p, int in_width, int in_height);
int tsvpp_prepare_batch(tsvpp_ctx *ctx, const tsvpp_params *p, int in_width, int in_height, int n_frames, void *stream);

/* Memory a context keeps although it no longer uses it.  Nothing is ever freed under running work: a geometry-table set pushed out of the cache (the 1024 most recently
 * used geometries stay) and an outgrown NV12 scratch buffer are RETIRED -- a launch already enqueued, or a captured hipGraph, may still hold their addresses, and a
 * hipFree would synchronise the device.  A context retires at most 256 MiB of table sets (a table set is tens of KiB: several thousand distinct geometries); past that,
 * new geometries run the kernels that compute their own coordinates (same bits, slower) and the library says so once on stderr.  tsvpp_trim releases the retired
 * memory: call it at a QUIESCENT point -- every conversion enqueued through this context has finished and no hipGraph captured from it will be replayed again.
 * `released_bytes` (may be NULL) receives the bytes of table sets released. */
int tsvpp_trim(tsvpp_ctx *ctx, size_t *released_bytes);

/* roctx ranges around every conversion ("tsvpp_convert n=.. WxH->WxH ..."), the counterpart of the reference's NVTX
 * ranges (include/Common.h:72-105 PUSH_RANGE/POP_RANGE, src/VideoProcessor.cpp:95; switched on by
 * Logger::enableNVTX / TensorStreamConverter.enable_nvtx()).  The tracer library (rocprofiler-sdk-roctx or
 * libroctx64) is looked up at run time; TSVPP_UNSUPPORTED if neither is installed. */
int tsvpp_enable_markers(tsvpp_ctx *ctx, int on);

/* ---- context options (round 6; not in the reference) -----------------------------------------------------------------------------------------------
 * TSVPP_OPT_INPUTS_READY (default 0).  The reference's getFrame hands a consumer a frame the decoder has FINISHED (Decoder::GetFrame blocks on a condition
 * variable until the decode thread publishes it, reference src/Decoder.cpp:97-131) and converts it into a buffer nobody else uses; its consumer streams
 * carry nothing but conversions (src/VideoProcessor.cpp:98-104).  A pipeline with that shape may set this option: every fused launch of the context then goes
 * out with the AQL barrier bit cleared (hipExtAnyOrderLaunch) -- it does not wait for work enqueued EARLIER on its stream, so back-to-back single-frame
 * conversions of one consumer overlap instead of paying a dependent-launch boundary each (~1.5-1.9 us against ~2.3 us of HBM time for a 1080p -> 720p frame).
 * Consumers served through tsvpp_consumer_next_stream additionally alternate between two streams (see there).
 * The caller promises, for every conversion while the option is set:
 *   (1) the input planes are complete in device memory when the call is made (not merely enqueued earlier on `stream`);
 *   (2) nothing enqueued earlier on `stream` still reads or writes the output buffer.
 * What stays ordered: anything enqueued LATER on the stream (events, copies, other kernels, synchronisation) still waits for the conversion; the two-pass
 * formats (UYVY / YUV444 where no single-pass kernel applies) and launches out of a tsvpp_table never use the option (their scratch buffer / table upload
 * are ordered by the stream).
 * Results are identical either way.
 * Value: 0 = off; 1 = on; (2 / 3: A-B values -- 2 = barrier-free launches without the second stream, 3 = the second stream without barrier-free launches). */
#define TSVPP_OPT_INPUTS_READY 1
/* TSVPP_OPT_COLOR_G_TERM (default 0).  The colour conversion's green chroma term `-0.813 (V-128) - 0.391 (U-128)` (reference src/ColorConversion.cu:30-35) is
 * the one place of the path where the reference's result depends on how nvcc contracted the expression AND no golden of the reference decides it (DESIGN.md
 * section 2; 36 of the 2^24 (Y, U, V) triples differ by one in G between the variants, R and B never):
 *   0  fma(-0.813, V-128, -(0.391 (U-128)))   the LLVM fadd -> fma rule that the resize goldens pin, applied here (the default since 0.3.0)
 *   1  (-0.813 (V-128)) - (0.391 (U-128))     plain IEEE, both products rounded (the library's output until 0.2.x)
 *   2  fma(-0.391, U-128, -0.813 (V-128))     the right-hand product fused
 * for whoever holds goldens of the real reference binary (tools/ref_capture/ produces them on an NVIDIA box).  Same speed; every kernel honours it. */
#define TSVPP_OPT_COLOR_G_TERM 2
/* TSVPP_OPT_UNSAFE_COEFFS (default 0): tsvpp_set_coeffs refuses (TSVPP_UNSUPPORTED) a block that differs from tsvpp_default_coeffs by a single bit unless this is
 * set -- the library's parity statements are about the reference's literals. */
#define TSVPP_OPT_UNSAFE_COEFFS 3
#define TSVPP_HAVE_OPTIONS 1
int tsvpp_set_option(tsvpp_ctx *ctx, int option, int value);
int tsvpp_get_option(const tsvpp_ctx *ctx, int option, int *value);

/* Colour constants: read the active block, replace it (e.g. with the block received
 * from rank 0), restore the defaults.  tsvpp_set_coeffs accepts only the default block (bit for bit) unless
 * TSVPP_OPT_UNSAFE_COEFFS is set: TSVPP_UNSUPPORTED otherwise. */
int tsvpp_get_coeffs(const tsvpp_ctx *ctx, tsvpp_coeffs *out);
int tsvpp_set_coeffs(tsvpp_ctx *ctx, const tsvpp_coeffs *in);
void tsvpp_default_coeffs(tsvpp_coeffs *out);

/* AREA (down-scale) weight table of generateResizePattern (reference
 * src/Resize.cu:359-386) as this library builds it: writes rows*taps floats
 * (taps = ceil(scale)) to `out` if it fits `max_floats`; returns rows or <0. */
int tsvpp_area_pattern(float scale, float *out, int max_floats, int *taps);

/* What tsvpp_convert_batch WOULD launch for this request -- stage selection (reference
 * src/VideoProcessor.cpp:106-151) plus this library's kernel / workgroup-shape / LDS choice -- as one line of
 * key=value text, e.g. "mode=bilinear out=f32_planar ... kernel=vpp_bilinear_kernel<...> shape=32x8 rpt=2 ...".
 * Needs no context and no GPU (host logic only; TSVPP_* knobs are honoured).  `aligned_outputs`: outputs are 16-byte
 * aligned.  tail= says how an output 4 k + 2 columns wide ends its rows: 2 = the launch's last tile column is shifted to the
 * frame's right edge (one launch), 1 = a second, element-wise launch converts the two-column row tail, 0 = neither is needed.
 * Returns TSVPP_OK or the status tsvpp_convert would return for the request. */
int tsvpp_describe(const tsvpp_params *p, int in_width, int in_height, int pitch_y, int pitch_uv, int n_frames, int aligned_outputs, char *buf,
                   size_t buf_len);

/* Human-readable text for a status returned by this library. */
const char *tsvpp_strerror(int status);
/* "tsvpp <version> gfx950" */
const char *tsvpp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TSVPP_H */
