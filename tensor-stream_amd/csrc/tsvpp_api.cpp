// tsvpp_api.cpp -- host side of the C ABI declared in include/tsvpp.h.
//
// Holds what the reference keeps in `class VideoProcessor` (include/VideoProcessor.h:120-149):
// the per-consumer stream pool, plus what the reference rebuilds on every frame and this
// library builds once: the AREA weight tables (reference src/Resize.cu:436-452 mallocs, copies
// and leaks them per frame).  No per-frame allocation, free or synchronisation.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "tsvpp.h"
#include "vpp_kernels.h"
#include "vpp_axis.h"

using namespace tsvpp;

namespace {

struct AreaTable {
    float *dev = nullptr;
    int rows = 0, taps = 0;
    AreaQRow *qdev = nullptr; // integer form, only if every weight is k / 2^shift and taps <= 8
    int shift = -1;
    int uniform_sum = 0;      // > 0: every row of the integer table has this weight sum
    float *dev4 = nullptr;    // the float rows zero-padded to a multiple of four taps
    int nk = 0;               // taps4 = 4 * nk
    int ones_end = 0;         // taps 1 .. ones_end - 1 weigh exactly 1.0f in every row (the streaming kernel multiplies nothing there)
    std::shared_ptr<std::vector<float>> host4; // host copy of dev4 (divisor tables of the float AREA kernels)
};

// Largest E such that taps 1 .. E - 1 weigh exactly 1.0f in every row of a weight table (the reference's rows are [rest] 1 ... 1 [last
// fraction]: E >= taps - 1 or taps - 2).
int area_ones_end(const std::vector<float> &tab, int rows, int taps) {
    int end = taps;
    for (int r = 0; r < rows; r++) {
        int k = 1;
        while (k < end && tab[(size_t)r * taps + k] == 1.0f) k++;
        end = k;
    }
    return end;
}

// Integer form of a weight table if all weights are dyadic: w * 2^shift integral, shift <= 6.
bool quantise_area_rows(const std::vector<float> &tab, int rows, int taps, std::vector<AreaQRow> &q, int &shift) {
    if (taps > 8) return false;
    for (shift = 0; shift <= 6; shift++) {
        bool ok = true;
        for (float w : tab) {
            const float s = w * (float)(1 << shift);
            if (s != std::floor(s) || s > 255.0f) { ok = false; break; }
        }
        if (ok) break;
    }
    if (shift > 6) return false;
    q.assign((size_t)rows, AreaQRow{});
    for (int r = 0; r < rows; r++) {
        AreaQRow &e = q[(size_t)r];
        for (int k = 0; k < taps; k++) {
            const uint32_t wi = (uint32_t)(tab[(size_t)r * taps + k] * (float)(1 << shift));
            e.sum += (int32_t)wi;
            e.w[k >> 2] |= wi << (8 * (k & 3));
            e.wu[k >> 1] |= wi << (16 * (k & 1));
        }
    }
    return true;
}

constexpr int kMaxPatternRows = 65536; // the reference's generator has no bound (can spin forever)

// Weight rows of the AREA down-scale, semantics of generateResizePattern (reference
// src/Resize.cu:359-386) in plain float arithmetic.  Each row is emitted with exactly
// taps = ceil(scale) entries: the device code never reads further (src/Resize.cu:162-169),
// so a row's optional (taps+1)-th entry is dropped here instead of being uploaded.
bool build_area_rows(float scale, std::vector<float> &tab, int &rows, int &taps) {
    taps = (int)std::ceil((double)scale);
    rows = 0;
    tab.clear();
    if (!(scale > 1.0f) || taps < 1) return false;
    float carry = 0.0f; // part of the next source pixel already consumed by the previous row
    for (int k = 0;; k++) {
        const float pos = (float)k * scale;
        const bool more = (pos == 0.0f) || (pos - (float)(int)pos > FLT_EPSILON);
        if (!more) break;
        if (rows >= kMaxPatternRows) return false;
        std::vector<float> row;
        float left = scale;
        if (carry != 0.0f) {
            row.push_back(carry);
            left = left - carry;
        }
        while (left - 1.0f > 0.0f) {
            row.push_back(1.0f);
            left = left - 1.0f;
        }
        if (left > FLT_EPSILON) {
            row.push_back(left);
            carry = 1.0f - left;
        }
        row.resize((size_t)taps, 0.0f); // pads short rows with 0, truncates long ones
        tab.insert(tab.end(), row.begin(), row.end());
        rows++;
    }
    return rows > 0;
}

// Is every interpolation weight of this request zero?  (odd integer ratios; then BILINEAR and
// BICUBIC reduce exactly to their centre tap, src/Resize.cu:17-23, 45-50 with w = 0)
// Per-thread memo of a geometry predicate: a consumer thread converts thousands of frames with a handful of geometries (a
// full scan of a dyadic request costs dst_w + dst_h coordinate evaluations, ~20 us -- several single-frame conversions).
struct GeomMemo {
    struct Entry { int cls, dw, dh, sw, sh; bool res; };
    Entry e[16];
    int n = 0, next = 0;
    const bool *find(int cls, int dw, int dh, int sw, int sh) const {
        for (int i = 0; i < n; i++)
            if (e[i].cls == cls && e[i].dw == dw && e[i].dh == dh && e[i].sw == sw && e[i].sh == sh) return &e[i].res;
        return nullptr;
    }
    bool put(int cls, int dw, int dh, int sw, int sh, bool res) {
        e[next] = Entry{ cls, dw, dh, sw, sh, res };
        next = (next + 1) % 16;
        if (n < 16) n++;
        return res;
    }
};

bool all_weights_zero(Mode m, int dst_w, int dst_h, float xr, float yr, int src_w, int src_h) {
    static thread_local GeomMemo memo;
    if (const bool *hit = memo.find((int)m, dst_w, dst_h, src_w, src_h)) return *hit;
    auto remember = [&](bool r) { return memo.put((int)m, dst_w, dst_h, src_w, src_h, r); };
    for (int axis = 0; axis < 2; axis++) {
        const int n = axis ? dst_h : dst_w, lim = axis ? src_h : src_w;
        const float r = axis ? yr : xr;
        for (int o = 0; o < n; o++) { // the chroma grid uses indices 0 .. n/2-1, a subset
            int p;
            if (m == M_BILINEAR) {
                float w;
                bilinear_axis(o, r, lim, p, w);
                if (w != 0.0f) return remember(false);
            } else {
                double w;
                bicubic_axis(o, r, lim, p, w);
                if (w != 0.0) return remember(false);
            }
        }
    }
    return remember(true);
}

// BILINEAR: is every weight of ONE axis zero (an odd integer ratio on that axis only -- BASELINE config C3: 1280 -> 256 columns, ratio 5, rows at
// 2.8125)?  The taps that a zero weight multiplies need not be fetched (sample_luma / sample_chroma: LaunchDesc::wx_zero / wy_zero).
bool axis_weights_zero(int axis, int n, int lim, float r) {
    static thread_local GeomMemo memo;
    if (const bool *hit = memo.find(axis, n, 0, lim, 0)) return *hit;
    for (int o = 0; o < n; o++) {
        int p;
        float w;
        bilinear_axis(o, r, lim, p, w);
        if (w != 0.0f) return memo.put(axis, n, 0, lim, 0, false);
    }
    return memo.put(axis, n, 0, lim, 0, true);
}

// Is every interpolation weight of this request a multiple of 1/16?  (ratios 1.5, 2, 2.5, 4, 0.5, 1.25, 2.25 ...: then the
// reference's float / double evaluation is exact and the integer kernels -- vpp_bicubic_int.hip, the integer thread tile of
// the 2x2-tap kernel -- reproduce it bit for bit.)  BILINEAR and BICUBIC share one coordinate formula (src/Resize.cu:276-303,
// 321-347); the AREA up-scale variant has its own (:221-234).
bool weights_dyadic(Mode m, int dst_w, int dst_h, float xr, float yr, int src_w, int src_h) {
    static thread_local GeomMemo memo;
    const int cls = (m == M_AREA_UP) ? 1 : 0;
    if (const bool *hit = memo.find(cls, dst_w, dst_h, src_w, src_h)) return *hit;
    auto remember = [&](bool r) { return memo.put(cls, dst_w, dst_h, src_w, src_h, r); };
    for (int axis = 0; axis < 2; axis++) {
        const int n = axis ? dst_h : dst_w, lim = axis ? src_h : src_w;
        const float r = axis ? yr : xr;
        for (int o = 0; o < n; o++) { // the chroma grid uses indices 0 .. n/2-1, a subset
            int p;
            float w;
            if (m == M_AREA_UP) areaup_axis(o, r, p, w);
            else bilinear_axis(o, r, lim, p, w); // bicubic_axis: the same fp32 coordinate, widened afterwards
            const float s = w * 16.0f;
            if (s != std::floor(s)) return remember(false);
        }
    }
    return remember(true);
}

struct Plan {
    Mode mode = M_NONE;
    OutKind out = O_U8_MERGED;
    int off_x = 0, off_y = 0; // crop origin
    int src_w = 0, src_h = 0; // logical source after crop
    int dst_w = 0, dst_h = 0;
    float xr = 1.f, yr = 1.f;
    int swap_rb = 0;
    size_t out_bytes = 0;
    int point_kind = PK_NONE;
    int wx_zero = 0, wy_zero = 0; // BILINEAR: every weight of that axis is zero (and not of both: that is point_kind)
    int w_dyadic = 0;
    int fourcc = TSVPP_RGB24;
    bool f32 = false;
};

} // namespace

struct tsvpp_ctx {
    int device = 0;
    std::vector<std::pair<std::string, hipStream_t>> streams;
    std::vector<hipStream_t> streams2; // TSVPP_OPT_INPUTS_READY: a consumer's second stream (created on first use), same index as `streams`
    std::vector<uint8_t> turn;         // ... and which of the two its next conversion takes
    std::mutex stream_mu;
    tsvpp_coeffs coeffs;
    std::map<uint32_t, AreaTable> area; // keyed by the bit pattern of the float scale
    std::map<uint64_t, float *> area_div; // divisor tables, keyed by both scales' bit patterns (null: too large, not built)
    int force_gather = 0;               // TSVPP_FORCE_GATHER=1: always use the global-gather kernel (A/B, tests)
    int nt_stores = -1, tile_order = 0, shape_tx = 0, shape_ty = 0; // TSVPP_NT, TSVPP_TILE_ORDER, TSVPP_SHAPE=tx,ty
    int num_cus = 256;
    // dyadic AREA reads straight from global memory from this ratio on (both axes); below it the LDS kernel wins
    // (measured after its VGPR fix: 2x 501 k vs 368 k fps, 3x 743 k vs 621 k, 4x 162 k vs 251 k)
    float area_direct_min = 3.5f;   // TSVPP_AREA_DIRECT_MIN
    float area_direct_fmin = 2.0f;  // the same for non-dyadic weights (a constant since round 3)
    int bicubic_int = 1;            // TSVPP_BICUBIC_INT: integer kernels for dyadic weights (1: + the streaming kernel at 3 : 2 / 2 : 1, 2: the LDS kernel only, 0: none)
    int bilinear_int = 1;           // TSVPP_BILINEAR_INT: integer thread tile of the 2x2-tap kernel for dyadic weights
    int area_box = 1;               // TSVPP_AREA_BOX: contiguous-run box kernel for integer ratios >= 4
    int area2 = 1;                  // float-weight AREA at 2 x 2 .. 3 x 3 taps on its own LDS kernel
    int lds_kb = 40;                // LDS bytes a staged workgroup may use (four workgroups per CU)
    int bilinear_win = 1;           // window form of the float 2x2-tap thread tile where it measured faster (ratios 1 .. 1.45)
    int u8_xchg = 1;                // 16-byte stores for uint8 merged outputs through an in-wave LDS exchange
    int area_divtab = 1;            // TSVPP_AREA_DIVTAB: host-built divisor table for the float AREA kernels
    int area_cols_rows = 0;         // tile height of the column-per-lane AREA kernel: by tap count
    int tail_shift = 1;             // TSVPP_TAIL_SHIFT: outputs 4 k + 2 columns wide end in a shifted tile column instead of a row tail and its second launch
    int area_cols = 1;              // TSVPP_AREA_COLS
    int rpt = 0;                    // TSVPP_RPT: row pairs per thread, 0 = per kernel (launch_fused)
    int dma = 1;                    // TSVPP_DMA=0 selects the register-staged path
    int r32 = 1;                    // TSVPP_R32: streaming kernel for BILINEAR at exactly 3 : 2 with uint8 outputs
    int bicubic_cols = 1;           // TSVPP_BICUBIC_COLS: wave-per-tile BICUBIC kernel (1: what the integer kernel does not take, 2: every BICUBIC request, 0: off)
    int area_stream = 1;            // TSVPP_AREA_STREAM: float-weight AREA with the source rows streamed through a wave-private LDS ring (vpp_area_stream.hip)
    int area_stream_rows = 0;       // its tile height: automatic (4)
    int area_stream_min_taps = 40;  // ... from this many taps (rx * ry) per value on (measured cross-over)
    int bicubic_rows = 0;           // TSVPP_BICUBIC_ROWS: its tile height (8 / 16 / 24 / 32; 0 = automatic)
    int bicubic_dma = 1;            // TSVPP_BICUBIC_DMA: its source rows through a wave-private LDS-DMA ring (0: per-lane loads)
    int bilinear_rows = 1;          // TSVPP_BILINEAR_ROWS: BILINEAR at sparse ratios with the tapped rows as LDS-DMA row segments (vpp_bilinear_rows.hip; 1: ratio product >= 12, 2: wherever it applies, 0: byte gathers)
    int bicubic_u8x = 1;            // TSVPP_BICUBIC_U8X: uint8 outputs of the wave-per-tile BICUBIC kernel through the 8 x 4 output side of the streaming kernels
    int point_rn = 1;               // TSVPP_POINT_RN: streaming point sampler at exact integer ratios 3 / 4 / 5 (vpp_point_rn.hip; 0: the LDS point kernel)
    int bilinear_rows_waves = 0;    // TSVPP_BILINEAR_ROWS_WAVES: its waves per workgroup (1 / 2 / 4; 0 = automatic)
    int geo_pref = 1;               // TSVPP_GEO: host-built geometry tables for the 2x2-tap kernel's window tiles (1: where measured to win, 2: wherever they apply)
    GeoCache *geo = nullptr;        // ... their device copies, one set per (request geometry, tile shape)
    std::mutex area_mu;
    // NV12 intermediates of the two-pass formats (UYVY / YUV444 with a resize): one grow-only slot per stream.  A slot's
    // mutex is held while BOTH passes of a call are enqueued, so calls that share a stream (typically NULL) cannot
    // interleave their passes on the shared buffer; a buffer that is outgrown is retired, not freed -- work already
    // enqueued may still use it and hipFree would synchronise the device -- and released in tsvpp_destroy.
    struct ScratchSlot {
        std::mutex mu;
        uint8_t *buf = nullptr;
        size_t bytes = 0;
    };
    std::map<void *, std::unique_ptr<ScratchSlot>> scratch;
    std::vector<uint8_t *> retired;
    std::mutex scratch_mu;
    int markers = 0; // tsvpp_enable_markers: roctx ranges around every conversion (the reference's NVTX ranges)
    int inputs_ready = 0; // TSVPP_OPT_INPUTS_READY: fused launches do not wait for earlier work on their stream (include/tsvpp.h)
    // Replay cache (convert_impl): a thread keeps the finished launches of its last requests, keyed by this context's id and epoch.  The epoch moves whenever
    // something a finished launch descriptor depends on changes or may be freed: tsvpp_set_coeffs, tsvpp_set_option, tsvpp_enable_markers, tsvpp_trim.
    uint64_t id = 0;
    std::atomic<uint64_t> epoch{ 1 };
    int replay = 1; // TSVPP_REPLAY=0 (debug knob): every call runs the full selection
    int color_g = 0; // TSVPP_OPT_COLOR_G_TERM
    int unsafe_coeffs = 0; // TSVPP_OPT_UNSAFE_COEFFS: tsvpp_set_coeffs accepts a block that differs from the defaults
    std::set<struct tsvpp_table *> tables; // live frame tables: released with the context if the caller destroys it first
    std::mutex tables_mu;
};

namespace tsvpp {
thread_local LaunchRecord *g_launch_rec = nullptr;
}

namespace {

// Stage selection of VideoProcessor::Convert (reference src/VideoProcessor.cpp:106-142).
int make_plan(const tsvpp_params *p, int in_w, int in_h, Plan &pl) {
    if (!p || in_w <= 0 || in_h <= 0) return TSVPP_ERROR;
    if ((in_w | in_h) & 1) return TSVPP_UNSUPPORTED; // NV12 needs even sizes (reference: undefined)
    const int cw = p->crop_right - p->crop_left, ch = p->crop_bottom - p->crop_top;
    // crop only if the box is strictly smaller in BOTH dimensions (src/VideoProcessor.cpp:109)
    const bool crop = cw > 0 && ch > 0 && cw < in_w && ch < in_h;
    pl.src_w = in_w;
    pl.src_h = in_h;
    pl.off_x = pl.off_y = 0;
    if (crop) {
        if (p->crop_left < 0 || p->crop_top < 0 || p->crop_right > in_w || p->crop_bottom > in_h) return TSVPP_ERROR;
        if ((cw | ch) & 1) return TSVPP_UNSUPPORTED; // reference writes chroma out of bounds here
        pl.src_w = cw;
        pl.src_h = ch;
        pl.off_x = p->crop_left;
        pl.off_y = p->crop_top;
    }
    pl.dst_w = pl.src_w;
    pl.dst_h = pl.src_h;
    pl.mode = M_NONE;
    pl.xr = pl.yr = 1.0f;
    if (p->dst_width < 0 || p->dst_height < 0) return TSVPP_ERROR;
    if (p->dst_width && p->dst_height && (p->dst_width != pl.src_w || p->dst_height != pl.src_h)) {
        if ((p->dst_width | p->dst_height) & 1) return TSVPP_UNSUPPORTED; // reference leaves chroma unwritten
        pl.dst_w = p->dst_width;
        pl.dst_h = p->dst_height;
        pl.xr = (float)pl.src_w / (float)pl.dst_w; // src/Resize.cu:418-419
        pl.yr = (float)pl.src_h / (float)pl.dst_h;
        switch (p->resize_type) {
        case TSVPP_NEAREST: pl.mode = M_NEAREST; break;
        case TSVPP_BILINEAR: pl.mode = M_BILINEAR; break;
        case TSVPP_BICUBIC: pl.mode = M_BICUBIC; break;
        case TSVPP_AREA: pl.mode = (pl.xr > 1.0f && pl.yr > 1.0f) ? M_AREA_DOWN : M_AREA_UP; break; // src/Resize.cu:435
        default: return TSVPP_UNSUPPORTED; // reference launches nothing and returns garbage
        }
    }
    pl.point_kind = PK_NONE;
    if (pl.mode == M_NEAREST) pl.point_kind = PK_NEAREST;
    else if ((pl.mode == M_BILINEAR || pl.mode == M_BICUBIC) && all_weights_zero(pl.mode, pl.dst_w, pl.dst_h, pl.xr, pl.yr, pl.src_w, pl.src_h))
        pl.point_kind = pl.mode == M_BILINEAR ? PK_BILINEAR0 : PK_BICUBIC0;
    pl.wx_zero = pl.wy_zero = 0;
    if (pl.mode == M_BILINEAR && pl.point_kind == PK_NONE) {
        pl.wx_zero = axis_weights_zero(0, pl.dst_w, pl.src_w, pl.xr) ? 1 : 0;
        pl.wy_zero = (!pl.wx_zero && axis_weights_zero(1, pl.dst_h, pl.src_h, pl.yr)) ? 1 : 0;
    }
    pl.w_dyadic = ((pl.mode == M_BICUBIC || pl.mode == M_BILINEAR || pl.mode == M_AREA_UP) && pl.point_kind == PK_NONE &&
                   weights_dyadic(pl.mode, pl.dst_w, pl.dst_h, pl.xr, pl.yr, pl.src_w, pl.src_h)) ? 1 : 0;
    pl.fourcc = p->fourcc;
    switch (p->fourcc) {
    case TSVPP_RGB24: pl.swap_rb = 0; break;
    case TSVPP_BGR24: pl.swap_rb = 1; break;
    case TSVPP_Y800: case TSVPP_NV12: case TSVPP_HSV: break;  // output flavours of the fused kernels
    case TSVPP_UYVY: case TSVPP_YUV444: break;                // second pass over the (resized) NV12
    default: return TSVPP_UNSUPPORTED;
    }
    if (p->planes != TSVPP_PLANAR && p->planes != TSVPP_MERGED) return TSVPP_UNSUPPORTED;
    // element type: src/VideoProcessor.cpp:139-142; HSV always runs the float kernels (src/ColorConversion.cu:357-370)
    const bool f32 = p->normalization != 0 || p->fourcc == TSVPP_HSV;
    pl.f32 = f32;
    pl.out = f32 ? (p->planes == TSVPP_PLANAR ? O_F32_PLANAR : O_F32_MERGED)
                 : (p->planes == TSVPP_PLANAR ? O_U8_PLANAR : O_U8_MERGED);
    if (p->fourcc == TSVPP_Y800) pl.out = f32 ? O_Y800_F32 : O_Y800_U8;
    else if (p->fourcc == TSVPP_NV12) pl.out = f32 ? O_NV12_F32 : O_NV12_U8;
    else if (p->fourcc == TSVPP_HSV) pl.out = O_HSV_F32;
    else if (p->fourcc == TSVPP_UYVY || p->fourcc == TSVPP_YUV444) pl.out = O_NV12_U8; // pass 1
    // channelsByFourCC: 1.5 for NV12 (src/VideoProcessor.cpp:4-14)
    const size_t elems = p->fourcc == TSVPP_NV12 ? (size_t)pl.dst_w * pl.dst_h * 3 / 2
                                                 : (size_t)(tsvpp_channels(p->fourcc) * (float)pl.dst_w) * (size_t)pl.dst_h;
    pl.out_bytes = elems * (f32 ? sizeof(float) : 1);
    if (pl.out_bytes >= ((size_t)1 << 32)) return TSVPP_UNSUPPORTED; // kernels use 32-bit offsets inside a frame
    return TSVPP_OK;
}

// Tuning knobs (profiling / A-B only), read once per context -- and ONLY when TSVPP_DEBUG_KNOBS=1 (round 6, VERDICT r05 #8: until then a release library silently
// obeyed 25 environment variables).  A context created under TSVPP_DEBUG_KNOBS=1 names every knob that differs from its default on stderr, once; a TSVPP_* knob that is
// set WITHOUT the gate is reported once per process and ignored.
struct KnobRow {
    const char *name;
    int tsvpp_ctx::*field;
};
const KnobRow kKnobs[] = {
    { "TSVPP_FORCE_GATHER", &tsvpp_ctx::force_gather }, { "TSVPP_NT", &tsvpp_ctx::nt_stores }, // NT: 0 plain, 1 nt, 2 sc1; default -1 = per kernel
    { "TSVPP_DMA", &tsvpp_ctx::dma }, { "TSVPP_GEO", &tsvpp_ctx::geo_pref }, { "TSVPP_R32", &tsvpp_ctx::r32 }, { "TSVPP_RPT", &tsvpp_ctx::rpt },
    { "TSVPP_BICUBIC_INT", &tsvpp_ctx::bicubic_int }, { "TSVPP_BICUBIC_COLS", &tsvpp_ctx::bicubic_cols }, { "TSVPP_BILINEAR_ROWS", &tsvpp_ctx::bilinear_rows },
    { "TSVPP_BICUBIC_U8X", &tsvpp_ctx::bicubic_u8x }, { "TSVPP_POINT_RN", &tsvpp_ctx::point_rn },
    { "TSVPP_LDS_KB", &tsvpp_ctx::lds_kb }, // LDS budget of the staged kernels in KiB (tests: a budget nothing fits)
    { "TSVPP_BILINEAR_ROWS_WAVES", &tsvpp_ctx::bilinear_rows_waves }, { "TSVPP_BICUBIC_ROWS", &tsvpp_ctx::bicubic_rows }, { "TSVPP_AREA_STREAM", &tsvpp_ctx::area_stream },
    { "TSVPP_BICUBIC_DMA", &tsvpp_ctx::bicubic_dma }, { "TSVPP_AREA_BOX", &tsvpp_ctx::area_box }, { "TSVPP_BILINEAR_INT", &tsvpp_ctx::bilinear_int },
    { "TSVPP_AREA_DIVTAB", &tsvpp_ctx::area_divtab }, { "TSVPP_AREA_COLS", &tsvpp_ctx::area_cols },
    { "TSVPP_TAIL_SHIFT", &tsvpp_ctx::tail_shift },         // 0: the two-column tail launch of rounds 1-3 (A/B)
    { "TSVPP_AREA_COLS_ROWS", &tsvpp_ctx::area_cols_rows }, // 8 | 32: tile height of the column-per-lane AREA kernel (default: by tap count and launch size)
    { "TSVPP_TILE_ORDER", &tsvpp_ctx::tile_order }, { "TSVPP_REPLAY", &tsvpp_ctx::replay },
};
// `report`: name the knobs that took effect on stderr (tsvpp_create; tsvpp_describe's dry runs stay silent)
void read_env_knobs(tsvpp_ctx *ctx, bool report = false) {
    const char *gate = std::getenv("TSVPP_DEBUG_KNOBS");
    const bool on = gate && std::atoi(gate) != 0;
    std::string seen;
    auto note = [&](const char *name, const char *val) { seen += std::string(seen.empty() ? "" : " ") + name + "=" + val; };
    for (const KnobRow &k : kKnobs)
        if (const char *e = std::getenv(k.name)) {
            note(k.name, e);
            if (on) ctx->*(k.field) = std::atoi(e);
        }
    if (const char *e = std::getenv("TSVPP_AREA_DIRECT_MIN")) {
        note("TSVPP_AREA_DIRECT_MIN", e);
        if (on) ctx->area_direct_min = (float)std::atof(e);
    }
    if (const char *e = std::getenv("TSVPP_SHAPE")) {
        note("TSVPP_SHAPE", e);
        if (on) std::sscanf(e, "%d,%d", &ctx->shape_tx, &ctx->shape_ty);
    }
    if (seen.empty()) return;
    if (on) {
        if (report) std::fprintf(stderr, "tsvpp: TSVPP_DEBUG_KNOBS=1, this context runs with %s (profiling / A-B settings: parity statements assume the defaults)\n", seen.c_str());
    } else {
        static std::atomic<bool> told{ false };
        if (!told.exchange(true)) std::fprintf(stderr, "tsvpp: ignoring %s (debug knobs are honoured only under TSVPP_DEBUG_KNOBS=1)\n", seen.c_str());
    }
}

// The part of the launch descriptor that depends only on the request and the knobs.
void fill_desc(const tsvpp_ctx *ctx, const Plan &pl, int pitch_y, int pitch_uv, LaunchDesc &d) {
    std::memset(&d, 0, sizeof(d));
    d.src_w = pl.src_w;
    d.src_h = pl.src_h;
    d.pitch_y = pitch_y;
    d.pitch_uv = pitch_uv;
    d.dst_w = pl.dst_w;
    d.dst_h = pl.dst_h;
    d.xr = pl.xr;
    d.yr = pl.yr;
    d.swap_rb = pl.swap_rb;
    d.point_kind = pl.point_kind;
    d.wx_zero = pl.wx_zero;
    d.wy_zero = pl.wy_zero;
    d.k = ctx->coeffs;
    d.color_g = ctx->color_g;
    d.force_gather = ctx->force_gather;
    d.nt_stores = ctx->nt_stores;
    d.tile_order = ctx->tile_order;
    d.shape_tx = ctx->shape_tx;
    d.shape_ty = ctx->shape_ty;
    d.dma = ctx->dma;
    d.rpt_pref = ctx->rpt;
    d.area_direct_min = ctx->area_direct_min;
    d.area_direct_fmin = ctx->area_direct_fmin;
    d.bicubic_int_pref = ctx->bicubic_int;
    d.bicubic_cols_pref = ctx->bicubic_cols;
    d.bc_rows = ctx->bicubic_rows;
    d.area_stream_pref = ctx->area_stream;
    d.as_rows = ctx->area_stream_rows;
    d.as_min_taps = ctx->area_stream_min_taps;
    d.bc_dma_pref = ctx->bicubic_dma;
    d.area_box_pref = ctx->area_box;
    d.w_dyadic = pl.w_dyadic;
    d.bil_int_pref = ctx->bilinear_int;
    d.u8_xchg = ctx->u8_xchg;
    d.bil_win_pref = ctx->bilinear_win;
    d.area2_pref = ctx->area2;
    d.lds_budget_kb = ctx->lds_kb;
    d.area_cols_pref = ctx->area_cols;
    d.area_cols_rows = ctx->area_cols_rows;
    d.last_col0 = ctx->tail_shift ? 0 : -1; // (launch_fused decides; < 0 = not allowed)
    d.num_cus = ctx->num_cus;
    d.geo_pref = ctx->geo_pref;
    d.r32_pref = ctx->r32;
    d.bil_rows_pref = ctx->bilinear_rows;
    d.point_rn_pref = ctx->point_rn;
    d.bc_u8x_pref = ctx->bicubic_u8x;
    d.br_waves = ctx->bilinear_rows_waves; // (forced; launch_fused chooses)
    d.geo_cache = ctx->geo;
}

// Integer box sums are exact (and equal to the reference's float accumulation) while 255 * sum(wx) * sum(wy) stays
// below 2^24; one divisor for the whole frame allows an exact integer division by a constant in the kernel.
bool dyadic_usable(float xr, float yr, int shift_x, int shift_y) {
    return (double)255 * ((double)xr * (1 << shift_x) + 1) * ((double)yr * (1 << shift_y) + 1) < 16777216.0;
}

int get_area_table(tsvpp_ctx *ctx, float scale, AreaTable &out) {
    uint32_t key;
    std::memcpy(&key, &scale, 4);
    std::lock_guard<std::mutex> lk(ctx->area_mu);
    auto it = ctx->area.find(key);
    if (it != ctx->area.end()) {
        out = it->second;
        return TSVPP_OK;
    }
    std::vector<float> tab;
    AreaTable t;
    if (!build_area_rows(scale, tab, t.rows, t.taps)) return TSVPP_UNSUPPORTED;
    hipError_t e = hipMalloc((void **)&t.dev, tab.size() * sizeof(float));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(t.dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t.dev);
        return (int)e;
    }
    {
        t.nk = (t.taps + 3) / 4;
        t.ones_end = area_ones_end(tab, t.rows, t.taps);
        std::vector<float> pad((size_t)t.rows * 4 * t.nk, 0.0f);
        for (int r = 0; r < t.rows; r++)
            for (int k = 0; k < t.taps; k++) pad[(size_t)r * 4 * t.nk + k] = tab[(size_t)r * t.taps + k];
        t.host4 = std::make_shared<std::vector<float>>(pad);
        if (hipMalloc((void **)&t.dev4, pad.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(t.dev4, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(t.dev);
            if (t.dev4) (void)hipFree(t.dev4);
            return TSVPP_ERROR;
        }
    }
    std::vector<AreaQRow> q;
    if (quantise_area_rows(tab, t.rows, t.taps, q, t.shift)) {
        t.uniform_sum = q[0].sum;
        for (const AreaQRow &row : q)
            if (row.sum != q[0].sum) t.uniform_sum = 0;
        if (hipMalloc((void **)&t.qdev, q.size() * sizeof(AreaQRow)) == hipSuccess &&
            hipMemcpy(t.qdev, q.data(), q.size() * sizeof(AreaQRow), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(t.qdev);
            t.qdev = nullptr;
        }
    }
    ctx->area[key] = t;
    out = t;
    return TSVPP_OK;
}

// Divisor table of the float AREA kernels that keep one output column per lane (vpp_area_cols.hip): the reference
// accumulates `divide += weight` next to `colorSum = fma(data, weight, colorSum)` (src/Resize.cu:160-178), and the sum depends
// only on the column's and the row's weight patterns -- nx * ny distinct values per geometry.  Built here with the kernel's
// own fp32 operations in the kernel's order (product rounded to fp32, then added; rows outer, the 4 * nk zero-padded taps
// inner), so the kernel can load the divisor instead of spending one add per tap and lane on it.  Tables above 2^18 entries
// are not built (the kernel then sums the weights itself).
#pragma clang fp contract(off)
// The table is an optimisation, never a requirement (the kernels sum the weights themselves when it is null): while `stream`
// is capturing, or when the allocation fails, nothing is built or cached and the call still succeeds -- a first conversion
// inside a hipGraph capture without tsvpp_prepare_batch runs the self-summing kernel instead of failing.
int get_area_div(tsvpp_ctx *ctx, float xr, float yr, const AreaTable &tx, const AreaTable &ty, const float *&out, hipStream_t stream) {
    out = nullptr;
    uint32_t kx, ky;
    std::memcpy(&kx, &xr, 4);
    std::memcpy(&ky, &yr, 4);
    const uint64_t key = ((uint64_t)kx << 32) | ky;
    std::lock_guard<std::mutex> lk(ctx->area_mu);
    auto it = ctx->area_div.find(key);
    if (it != ctx->area_div.end()) {
        out = it->second;
        return TSVPP_OK;
    }
    float *dev = nullptr;
    if (tx.host4 && ty.host4 && (long)tx.rows * ty.rows <= (1L << 18)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        // (the NULL stream is never queried: asking the legacy stream while another stream captures in global mode invalidates that capture)
        if (stream && hipStreamIsCapturing(stream, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusNone) return TSVPP_OK; // no allocation, no synchronous copy during capture
        const int tx4 = 4 * tx.nk, ty4 = 4 * ty.nk;
        std::vector<float> tab((size_t)tx.rows * ty.rows);
        for (int jx = 0; jx < tx.rows; jx++) {
            const float *wx = tx.host4->data() + (size_t)jx * tx4;
            for (int iy = 0; iy < ty.rows; iy++) {
                const float *wy = ty.host4->data() + (size_t)iy * ty4;
                volatile float div = 0.0f; // volatile: every product and every partial sum is rounded to fp32, as on the device
                for (int a = 0; a < ty.taps; a++)
                    for (int k = 0; k < tx4; k++) {
                        volatile float wgt = wx[k] * wy[a];
                        div = div + wgt;
                    }
                tab[(size_t)jx * ty.rows + iy] = div;
            }
        }
        if (hipMalloc((void **)&dev, tab.size() * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return TSVPP_OK; // out stays null
        }
        if (hipMemcpy(dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(dev);
            return TSVPP_OK;
        }
    }
    ctx->area_div[key] = dev;
    out = dev;
    return TSVPP_OK;
}

// Selects the context's device for the duration of one API call and restores the caller's on exit (as torch's
// CUDAGuard does): a VideoProcessor bound to device k must not change the calling thread's current device.
struct DeviceGuard {
    int prev = -1, status = TSVPP_OK;
    bool switched = false;
    explicit DeviceGuard(const tsvpp_ctx *ctx) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) { status = (int)e; return; }
        if (prev != ctx->device) {
            e = hipSetDevice(ctx->device);
            if (e != hipSuccess) { status = (int)e; return; }
            switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// roctx ranges (the reference brackets Convert with NVTX ranges: include/Common.h:72-105, src/VideoProcessor.cpp:95).
// The tracer library is looked up at run time -- rocprofv3's SDK flavour first, the legacy roctracer one second --
// so libtsvpp.so carries no link-time dependency on either.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
            void *h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx &roctx() {
    static const Roctx r;
    return r;
}
struct RangeGuard {
    bool on;
    RangeGuard(bool enabled, const char *label) : on(enabled && roctx().push) {
        if (on) roctx().push(label);
    }
    ~RangeGuard() {
        if (on) roctx().pop();
    }
};

// The scratch slot of a stream, created on first use; `need` bytes are guaranteed after grow() (caller holds slot->mu).
tsvpp_ctx::ScratchSlot *scratch_slot(tsvpp_ctx *ctx, void *stream) {
    std::lock_guard<std::mutex> lk(ctx->scratch_mu);
    auto &slot = ctx->scratch[stream];
    if (!slot) slot.reset(new tsvpp_ctx::ScratchSlot());
    return slot.get();
}
int scratch_grow(tsvpp_ctx *ctx, tsvpp_ctx::ScratchSlot *slot, size_t need) {
    if (slot->bytes >= need) return TSVPP_OK;
    uint8_t *fresh = nullptr;
    hipError_t e = hipMalloc((void **)&fresh, need);
    if (e != hipSuccess) return (int)e;
    if (slot->buf) {
        std::lock_guard<std::mutex> lk(ctx->scratch_mu);
        ctx->retired.push_back(slot->buf);
    }
    slot->buf = fresh;
    slot->bytes = need;
    return TSVPP_OK;
}
size_t scratch_frame_bytes(const Plan &pl) { return (((size_t)pl.dst_w * pl.dst_h * 3 / 2) + 255) & ~(size_t)255; }
// Bytes one frame of this request moves, roughly: the output + the source rows its sampler taps (a point sampler at vertical ratio r reads every r-th row,
// the 2-tap kernels two of every r, BICUBIC four: C4's 4K -> 720p point sample 6.9 MB where the ROI formula says 15.2; PMC: 6.91 MB).  Only used to decide whether a
// launch is small enough to overlap its predecessor (TSVPP_BARRIER_FREE_MAX_BYTES).
size_t plan_moved_bytes(const Plan &pl) {
    double rows = 1.0;
    if (pl.yr > 1.0f) {
        if (pl.mode == M_NEAREST || pl.point_kind != PK_NONE) rows = 1.0 / pl.yr;
        else if (pl.mode == M_BILINEAR || pl.mode == M_AREA_UP) rows = 2.0 / pl.yr;
        else if (pl.mode == M_BICUBIC) rows = 4.0 / pl.yr;
    }
    if (rows > 1.0) rows = 1.0;
    return pl.out_bytes + (size_t)((double)pl.src_w * pl.src_h * 1.5 * rows);
}

bool needs_scratch(const Plan &pl) { return (pl.fourcc == TSVPP_UYVY || pl.fourcc == TSVPP_YUV444) && pl.mode != M_NONE; }

// ---- replay of a thread's recent launches (round 6) ---------------------------------------------------------------------------------------------------
// A conversion of ONE frame is host-bound: ~3.3 us of runtime per launch (tools/launch_cost.hip: 3.5 us with this library's 3.7 KiB kernarg segment, 2.65 with a
// 16-byte one) plus what this file adds -- make_plan, fill_desc, launch_fused's selection, the table caches' mutexes and map lookups.  The latter is the same work for
// every frame a consumer converts, so a thread keeps the FINISHED launches (host function, grid, final LaunchDesc) of its last eight requests and replays one when
// the request, the frame geometry, the alignment class of the pointers, the context and its epoch all match.  Only single-launch requests are kept (no second
// pass, no row-tail launch, no frame table, at most TSVPP_MAX_BATCH frames); TSVPP_REPLAY=0 disables it (the GPU suite is replayed both ways).
struct ReplayEntry {
    uint64_t ctx_id = 0, epoch = 0, stamp = 0;
    tsvpp_params p = {};
    int w = 0, h = 0, py = 0, puv = 0, n = 0, aligned4 = 0, vec = 0;
    size_t y_off = 0, uv_off = 0;
    LaunchRecord rec;
};
constexpr int kReplayEntries = 8;
constexpr int kReplayMiss = -1000;
thread_local ReplayEntry g_replay[kReplayEntries];
thread_local uint64_t g_replay_clock = 0;
struct RecordArm { // points launch_fused's launches at `rec` for the lifetime of this object
    explicit RecordArm(LaunchRecord *rec) { g_launch_rec = rec; }
    ~RecordArm() { g_launch_rec = nullptr; }
};

int try_replay(tsvpp_ctx *ctx, int n, const tsvpp_nv12 *in, const tsvpp_params *p, void *const *outs, void *stream) {
    const uint64_t epoch = ctx->epoch.load(std::memory_order_relaxed);
    const int py = in[0].pitch_y ? in[0].pitch_y : in[0].width, puv = in[0].pitch_uv ? in[0].pitch_uv : in[0].width;
    int flags_for = -1, aligned4 = 0, vec = 0; // alignment class of THIS call, computed once (every entry of one request has the same crop offsets)
    for (int k = 0; k < kReplayEntries; k++) {
        ReplayEntry &e = g_replay[k];
        if (e.ctx_id != ctx->id || e.epoch != epoch || e.n != n || e.w != in[0].width || e.h != in[0].height || e.py != py || e.puv != puv ||
            std::memcmp(&e.p, p, sizeof(*p)) != 0)
            continue;
        if (flags_for < 0) { // the slow path's per-frame checks, in its order
            bool a4 = (py % 4 == 0) && (puv % 4 == 0), v = true;
            for (int f = 0; f < n; f++) {
                if (!in[f].y || !in[f].uv || !outs[f]) return TSVPP_ERROR;
                if (in[f].width != in[0].width || in[f].height != in[0].height) return TSVPP_UNSUPPORTED;
                if ((in[f].pitch_y ? in[f].pitch_y : in[f].width) != py) return TSVPP_UNSUPPORTED;
                if ((in[f].pitch_uv ? in[f].pitch_uv : in[f].width) != puv) return TSVPP_UNSUPPORTED;
                a4 = a4 && (((uintptr_t)(in[f].y + e.y_off)) % 4 == 0) && (((uintptr_t)(in[f].uv + e.uv_off)) % 4 == 0);
                v = v && (((uintptr_t)outs[f] & 15) == 0);
            }
            aligned4 = a4 ? 1 : 0;
            vec = v ? 1 : 0;
            flags_for = k;
        }
        if (e.aligned4 != aligned4 || e.vec != vec) continue;
        DeviceGuard guard(ctx);
        if (guard.status != TSVPP_OK) return guard.status;
        FrameTable t; // (only the first n triples and the three ext / off pairs are read)
        t.y.ext = nullptr; t.y.off = 0;
        t.uv.ext = nullptr; t.uv.off = 0;
        t.out.ext = nullptr; t.out.off = 0;
        for (int f = 0; f < n; f++) {
            t.y[f] = in[f].y + e.y_off;
            t.uv[f] = in[f].uv + e.uv_off;
            t.out[f] = outs[f];
        }
        void *args[2] = { (void *)&e.rec.d, (void *)&t };
        const hipError_t err = e.rec.d.any_order
                                   ? hipExtLaunchKernel(e.rec.fn, e.rec.grid, e.rec.block, args, e.rec.lds, (hipStream_t)stream, nullptr, nullptr, hipExtAnyOrderLaunch)
                                   : hipLaunchKernel(e.rec.fn, e.rec.grid, e.rec.block, args, e.rec.lds, (hipStream_t)stream);
        e.stamp = ++g_replay_clock;
        return (int)err;
    }
    return kReplayMiss;
}

void remember_launch(const tsvpp_ctx *ctx, uint64_t epoch, int n, const tsvpp_nv12 *in, const tsvpp_params *p, int py, int puv, bool aligned4, bool vec, size_t y_off,
                     size_t uv_off, const LaunchRecord &rec) {
    ReplayEntry *slot = &g_replay[0];
    for (ReplayEntry &e : g_replay)
        if (e.stamp < slot->stamp) slot = &e;
    slot->ctx_id = ctx->id;
    slot->epoch = epoch;
    slot->stamp = ++g_replay_clock;
    slot->p = *p;
    slot->w = in[0].width;
    slot->h = in[0].height;
    slot->py = py;
    slot->puv = puv;
    slot->n = n;
    slot->aligned4 = aligned4 ? 1 : 0;
    slot->vec = vec ? 1 : 0;
    slot->y_off = y_off;
    slot->uv_off = uv_off;
    slot->rec = rec;
}

} // namespace

extern "C" {

void tsvpp_default_coeffs(tsvpp_coeffs *k) {
    if (!k) return;
    k->y_scale = 1.163999557f;
    k->v_to_r = 1.5959997177f;
    k->u_to_b = 2.017999649f;
    k->v_to_g = -0.812999725f;
    k->u_to_g = -0.390999794f;
    k->round_bias = 0.5f;
    k->y_offset = 16.0f;
    k->c_offset = 128.0f;
}

int tsvpp_create(int device, int max_consumers, tsvpp_ctx **out_ctx) {
    if (!out_ctx || max_consumers < 0) return TSVPP_ERROR;
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return (int)e;
    if (device < 0 || device >= count) return (int)hipErrorInvalidDevice;
    tsvpp_ctx *ctx = new tsvpp_ctx();
    static std::atomic<uint64_t> next_id{ 1 };
    ctx->id = next_id.fetch_add(1);
    ctx->device = device;
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) {
        delete ctx;
        return guard.status;
    }
    tsvpp_default_coeffs(&ctx->coeffs);
    read_env_knobs(ctx, true);
    ctx->geo = geo_cache_create();
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
    }
    for (int i = 0; i < max_consumers; i++) {
        hipStream_t s = nullptr;
        e = hipStreamCreate(&s); // blocking stream, as the reference (src/VideoProcessor.cpp:86)
        if (e != hipSuccess) {
            for (auto &st : ctx->streams)
                if (st.second) (void)hipStreamDestroy(st.second);
            geo_cache_destroy(ctx->geo);
            delete ctx;
            return (int)e;
        }
        ctx->streams.emplace_back(std::string("empty"), s);
    }
    ctx->streams2.assign(ctx->streams.size(), nullptr);
    ctx->turn.assign(ctx->streams.size(), 0);
    *out_ctx = ctx;
    return TSVPP_OK;
}

static void ctx_release_tables(tsvpp_ctx *ctx); // (defined with struct tsvpp_table below)

void tsvpp_destroy(tsvpp_ctx *ctx) {
    if (!ctx) return;
    {
        DeviceGuard guard(ctx);
        ctx_release_tables(ctx); // frame tables the caller has not destroyed yet: their device memory goes with the context, their handles stay valid for tsvpp_table_destroy
        for (auto &s : ctx->streams)
            if (s.second) (void)hipStreamDestroy(s.second);
        for (hipStream_t s : ctx->streams2)
            if (s) (void)hipStreamDestroy(s);
        for (auto &sc : ctx->scratch)
            if (sc.second && sc.second->buf) (void)hipFree(sc.second->buf);
        for (uint8_t *b : ctx->retired) (void)hipFree(b);
    for (auto &a : ctx->area) {
        if (a.second.dev) (void)hipFree(a.second.dev);
        if (a.second.qdev) (void)hipFree(a.second.qdev);
        if (a.second.dev4) (void)hipFree(a.second.dev4);
    }
        for (auto &a : ctx->area_div)
            if (a.second) (void)hipFree(a.second);
        geo_cache_destroy(ctx->geo);
        ctx->geo = nullptr;
    }
    delete ctx;
}

int tsvpp_consumer_stream(tsvpp_ctx *ctx, const char *name, void **out_stream) {
    if (!ctx || !name || !out_stream) return TSVPP_ERROR;
    std::lock_guard<std::mutex> lk(ctx->stream_mu);
    for (auto &s : ctx->streams) {
        if (s.first == name) {
            *out_stream = (void *)s.second;
            return TSVPP_OK;
        }
        if (s.first == "empty") {
            s.first = name;
            *out_stream = (void *)s.second;
            return TSVPP_OK;
        }
    }
    *out_stream = nullptr;
    return TSVPP_ERROR; // pool exhausted (reference src/VideoProcessor.cpp:100-103)
}

// Index of the consumer's pool slot (claiming a free one for a new name), or -1 (caller holds stream_mu).
static int consumer_slot(tsvpp_ctx *ctx, const char *name) {
    for (size_t i = 0; i < ctx->streams.size(); i++) {
        auto &s = ctx->streams[i];
        if (s.first == name) return (int)i;
        if (s.first == "empty") {
            s.first = name;
            return (int)i;
        }
    }
    return -1;
}

int tsvpp_consumer_next_stream(tsvpp_ctx *ctx, const char *name, size_t launch_bytes, void **out_stream) {
    if (!ctx || !name || !out_stream) return TSVPP_ERROR;
    *out_stream = nullptr;
    std::lock_guard<std::mutex> lk(ctx->stream_mu);
    const int i = consumer_slot(ctx, name);
    if (i < 0) return TSVPP_ERROR; // pool exhausted (reference src/VideoProcessor.cpp:100-103)
    if (!(ctx->inputs_ready == 1 || ctx->inputs_ready == 3) || launch_bytes > TSVPP_OVERLAP_MAX_BYTES) { // (large launches do not overlap: see the header)
        *out_stream = (void *)ctx->streams[(size_t)i].second;
        return TSVPP_OK;
    }
    if (!ctx->streams2[(size_t)i]) {
        DeviceGuard guard(ctx);
        if (guard.status != TSVPP_OK) return guard.status;
        hipStream_t s = nullptr;
        const hipError_t e = hipStreamCreate(&s); // blocking, as the pool's first stream
        if (e != hipSuccess) return (int)e;
        ctx->streams2[(size_t)i] = s;
    }
    const uint8_t t = ctx->turn[(size_t)i];
    ctx->turn[(size_t)i] = t ^ 1;
    *out_stream = (void *)(t ? ctx->streams2[(size_t)i] : ctx->streams[(size_t)i].second);
    return TSVPP_OK;
}

int tsvpp_consumer_synchronize(tsvpp_ctx *ctx, const char *name) {
    if (!ctx || !name) return TSVPP_ERROR;
    hipStream_t a = nullptr, b = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->stream_mu);
        int i = -1;
        for (size_t k = 0; k < ctx->streams.size(); k++)
            if (ctx->streams[k].first == name) i = (int)k;
        if (i < 0) return TSVPP_ERROR; // no such consumer
        a = ctx->streams[(size_t)i].second;
        b = ctx->streams2[(size_t)i];
    }
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    hipError_t e = hipStreamSynchronize(a);
    if (e == hipSuccess && b) e = hipStreamSynchronize(b);
    return (int)e;
}

float tsvpp_channels(int fourcc) {
    if (fourcc == TSVPP_Y800) return 1.0f;
    if (fourcc == TSVPP_UYVY) return 2.0f;
    if (fourcc == TSVPP_NV12) return 1.5f;
    return 3.0f;
}

int tsvpp_out_dims(const tsvpp_params *p, int in_width, int in_height, int *out_width, int *out_height) {
    Plan pl;
    int sts = make_plan(p, in_width, in_height, pl);
    if (sts != TSVPP_OK) return sts;
    if (out_width) *out_width = pl.dst_w;
    if (out_height) *out_height = pl.dst_h;
    return TSVPP_OK;
}

size_t tsvpp_out_bytes(const tsvpp_params *p, int in_width, int in_height) {
    Plan pl;
    if (make_plan(p, in_width, in_height, pl) != TSVPP_OK) return 0;
    return pl.out_bytes;
}

int tsvpp_prepare_batch(tsvpp_ctx *ctx, const tsvpp_params *p, int in_width, int in_height, int n_frames, void *stream) {
    if (!ctx || n_frames < 0) return TSVPP_ERROR;
    Plan pl;
    int sts = make_plan(p, in_width, in_height, pl);
    if (sts != TSVPP_OK) return sts;
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    if (pl.mode == M_AREA_DOWN) {
        AreaTable tx, ty;
        sts = get_area_table(ctx, pl.xr, tx);
        if (sts != TSVPP_OK) return sts;
        sts = get_area_table(ctx, pl.yr, ty);
        if (sts != TSVPP_OK) return sts;
        if (!(tx.qdev && ty.qdev) && ctx->area_divtab) { // the float-weight kernels' divisor table: the same condition as in convert
            const float *div = nullptr;
            sts = get_area_div(ctx, pl.xr, pl.yr, tx, ty, div, (hipStream_t)stream);
            if (sts != TSVPP_OK) return sts;
        }
    }
    if (needs_scratch(pl) && n_frames > 0) { // the NV12 intermediate of UYVY / YUV444 behind a resize
        tsvpp_ctx::ScratchSlot *slot = scratch_slot(ctx, stream);
        std::lock_guard<std::mutex> lk(slot->mu);
        sts = scratch_grow(ctx, slot, scratch_frame_bytes(pl) * (size_t)n_frames);
        if (sts != TSVPP_OK) return sts;
    }
    if (((pl.mode == M_BILINEAR || pl.mode == M_AREA_UP) && ctx->geo_pref) || (pl.mode == M_BICUBIC && ctx->bicubic_cols)) {
        // geometry tables of the 2x2-tap kernel: a dry run of the launch chooses the tile shape and builds them.  They do not
        // depend on the pitches (only their use does: multiples of 16) nor on the batch size beyond the tile shape -- a
        // conversion that ends up with another shape builds its own set on first use.
        LaunchDesc d;
        const int pitch = (in_width + 255) & ~255;
        fill_desc(ctx, pl, pitch, pitch, d);
        d.in_aligned4 = 1;
        d.geo_build = 1;
        FrameTable t = {};
        const int total = n_frames > 0 ? n_frames : 1;
        // the launch groups tsvpp_convert_batch forms: full groups of TSVPP_MAX_BATCH frames, then the remainder; and, for more frames than that, the groups
        // tsvpp_convert_table forms (up to TSVPP_MAX_TABLE_LAUNCH per launch; a launch the grid limit caps lower builds its tables on first use)
        const int groups[4] = { total >= TSVPP_MAX_BATCH ? TSVPP_MAX_BATCH : 0, total % TSVPP_MAX_BATCH,
                                total > TSVPP_MAX_BATCH ? (total < TSVPP_MAX_TABLE_LAUNCH ? total : TSVPP_MAX_TABLE_LAUNCH) : 0,
                                total > TSVPP_MAX_TABLE_LAUNCH ? total % TSVPP_MAX_TABLE_LAUNCH : 0 };
        for (int cnt : groups) {
            if (cnt == 0) continue;
            d.n_frames = cnt;
            LaunchInfo info = {};
            hipError_t e = launch_fused(pl.mode, pl.out, true, d, t, (hipStream_t)stream, &info);
            if (e != hipSuccess) return (int)e;
        }
    }
    return sts;
}

int tsvpp_prepare(tsvpp_ctx *ctx, const tsvpp_params *p, int in_width, int in_height) {
    return tsvpp_prepare_batch(ctx, p, in_width, in_height, 0, nullptr);
}

int tsvpp_enable_markers(tsvpp_ctx *ctx, int on) {
    if (!ctx) return TSVPP_ERROR;
    if (on && !roctx().push) return TSVPP_UNSUPPORTED; // no roctx library on this machine
    ctx->markers = on ? 1 : 0;
    ctx->epoch.fetch_add(1);
    return TSVPP_OK;
}

static OutKind single_pass_kind(const Plan &pl) { return pl.fourcc == TSVPP_UYVY ? (pl.f32 ? O_UYVY_F32 : O_UYVY_U8) : O_YUV444_U8; }
// UYVY / YUV444 (uint8; UYVY also fp32) behind a resize: does the streaming kernel take it (exact 3 : 2 or 2 : 1, BILINEAR / AREA / NEAREST, dword-aligned planes,
// width % 8 == 0, height % 4 == 0, 16-byte aligned outputs)?  Asked of launch_fused itself, as a dry run: one set of conditions.
static bool single_pass_format(const Plan &pl, const LaunchDesc &d, int n, void *const *outs, hipStream_t stream) {
    if ((pl.fourcc != TSVPP_UYVY && pl.fourcc != TSVPP_YUV444) || (pl.f32 && pl.fourcc != TSVPP_UYVY) || pl.mode == M_NONE) return false; // (fp32: UYVY only, round 6)
    if (outs)
        for (int f = 0; f < n; f++)
            if (((uintptr_t)outs[f] & 15) != 0) return false;
    LaunchDesc dd = d;
    dd.n_frames = n < TSVPP_MAX_BATCH ? (n > 0 ? n : 1) : TSVPP_MAX_BATCH;
    FrameTable t = {};
    LaunchInfo info = {};
    return launch_fused(pl.mode, single_pass_kind(pl), true, dd, t, stream, &info) == hipSuccess;
}

// Device-resident columns of a tsvpp_table, already advanced to the run's first entry (null: the pointer triples travel in the kernarg segment).
struct TableCols {
    const uint64_t *y = nullptr, *uv = nullptr, *out = nullptr;
};

// tsvpp_convert_batch and tsvpp_convert_table: `in` / `outs` are HOST arrays of the n frames (a table's pinned mirror for the latter).
static int convert_impl(tsvpp_ctx *ctx, int n, const tsvpp_nv12 *in, const tsvpp_params *p, void *const *outs, void *stream, const TableCols &tab) {
    if (!ctx || !in || !p || !outs || n < 0) return TSVPP_ERROR;
    if (n == 0) return TSVPP_OK;
    const bool replayable = ctx->replay && tab.y == nullptr && n <= TSVPP_MAX_BATCH && !ctx->markers;
    const uint64_t epoch0 = ctx->epoch.load(std::memory_order_relaxed);
    if (replayable) { // the same request as one of this thread's recent ones: its finished launch again, with this call's frame pointers
        const int r = try_replay(ctx, n, in, p, outs, stream);
        if (r != kReplayMiss) return r;
    }
    Plan pl;
    int sts = make_plan(p, in[0].width, in[0].height, pl);
    if (sts != TSVPP_OK) return sts;
    const int pitch_y = in[0].pitch_y ? in[0].pitch_y : in[0].width; // reference fallback
    const int pitch_uv = in[0].pitch_uv ? in[0].pitch_uv : in[0].width;
    if (pitch_y < in[0].width || pitch_uv < in[0].width) return TSVPP_ERROR;
    // dst_w is even; when it is 4 k + 2 launch_fused shifts the launch's last tile column to the frame's right edge (or, where it cannot, the
    // vector-store kernels skip the two-column tail of every row and a tiny element-wise launch converts it); rows then start 8 bytes / 2 bytes
    // off the vector alignment, which global stores tolerate.
    // (alignment is decided per launch group of TSVPP_MAX_BATCH frames below: one odd pointer does not push the
    // whole batch onto the element-wise kernel)
    for (int f = 0; f < n; f++) {
        if (!in[f].y || !in[f].uv || !outs[f]) return TSVPP_ERROR;
        if (in[f].width != in[0].width || in[f].height != in[0].height) return TSVPP_UNSUPPORTED;
        if ((in[f].pitch_y ? in[f].pitch_y : in[f].width) != pitch_y) return TSVPP_UNSUPPORTED;
        if ((in[f].pitch_uv ? in[f].pitch_uv : in[f].width) != pitch_uv) return TSVPP_UNSUPPORTED;
    }
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    char label[96] = "";
    const bool markers = ctx->markers != 0; // read once: a concurrent tsvpp_enable_markers cannot push an unformatted label
    if (markers)
        std::snprintf(label, sizeof(label), "tsvpp_convert n=%d %dx%d->%dx%d mode=%d fourcc=%d stream=%p", n, pl.src_w, pl.src_h, pl.dst_w, pl.dst_h,
                      (int)pl.mode, pl.fourcc, stream);
    RangeGuard range(markers, label);

    LaunchDesc d;
    fill_desc(ctx, pl, pitch_y, pitch_uv, d);
    if (pl.mode == M_AREA_DOWN) {
        AreaTable tx, ty;
        sts = get_area_table(ctx, pl.xr, tx);
        if (sts != TSVPP_OK) return sts;
        sts = get_area_table(ctx, pl.yr, ty);
        if (sts != TSVPP_OK) return sts;
        d.patx = tx.dev;
        d.nx = tx.rows;
        d.rx = tx.taps;
        d.paty = ty.dev;
        d.ny = ty.rows;
        d.ry = ty.taps;
        d.patx4 = tx.dev4;
        d.nkx = tx.nk;
        d.paty4 = ty.dev4;
        d.nky = ty.nk;
        d.as_ones_x = tx.ones_end;
        if (!(tx.qdev && ty.qdev) && ctx->area_divtab) { // float-weight kernels only
            sts = get_area_div(ctx, pl.xr, pl.yr, tx, ty, d.area_div, (hipStream_t)stream);
            if (sts != TSVPP_OK) return sts;
        }
        if (tx.qdev && ty.qdev && dyadic_usable(pl.xr, pl.yr, tx.shift, ty.shift)) {
            d.qx = tx.qdev;
            d.qy = ty.qdev;
            d.box_rx = (tx.rows == 1 && tx.shift == 0 && tx.uniform_sum == tx.taps) ? tx.taps : 0; // one row of all ones
            d.box_ry = (ty.rows == 1 && ty.shift == 0 && ty.uniform_sum == ty.taps) ? ty.taps : 0;
            // one divisor for the whole frame -> exact integer division by a constant in the kernel
            if (tx.uniform_sum > 0 && ty.uniform_sum > 0 && (long)tx.uniform_sum * ty.uniform_sum < 4096)
                d.area_rcp = 1.0f / (float)(tx.uniform_sum * ty.uniform_sum);
        }
    }
    // crop = pointer arithmetic (reference src/Crop.cu:10-18: luma at (left + j, top + i), chroma
    // row top/2 + i/2, chroma byte (j & ~1) + left -- an odd `left` therefore swaps U and V,
    // exactly as it does in the reference)
    const size_t y_off = (size_t)pl.off_y * (size_t)pitch_y + (size_t)pl.off_x;
    const size_t uv_off = (size_t)(pl.off_y / 2) * (size_t)pitch_uv + (size_t)pl.off_x;
    bool aligned4 = (pitch_y % 4 == 0) && (pitch_uv % 4 == 0);
    for (int f = 0; f < n && aligned4; f++)
        aligned4 = (((uintptr_t)(in[f].y + y_off)) % 4 == 0) && (((uintptr_t)(in[f].uv + uv_off)) % 4 == 0);
    d.in_aligned4 = aligned4 ? 1 : 0;
    // Formats other than RGB24/BGR24: the reference feeds its other colour kernels with
    // the resized NV12; here pass 1 (only if there is a resize) writes that intermediate with the
    // fused kernel, pass 2 converts it.  Crop alone needs no pass 1: it is pointer arithmetic.
    // ... unless the streaming 3 : 2 / 2 : 1 kernel takes the request: it writes UYVY / YUV444 (uint8) itself, in one pass (vpp_bilinear_r32.hip)
    const bool single = single_pass_format(pl, d, n, outs, (hipStream_t)stream);
    const bool two_pass = (pl.fourcc == TSVPP_UYVY || pl.fourcc == TSVPP_YUV444) && !single;
    uint8_t *scratch = nullptr;
    size_t frame_scratch = 0;
    std::unique_lock<std::mutex> scratch_lock; // held until both passes of this call are enqueued
    if (needs_scratch(pl) && !single) {
        frame_scratch = scratch_frame_bytes(pl);
        tsvpp_ctx::ScratchSlot *slot = scratch_slot(ctx, stream);
        scratch_lock = std::unique_lock<std::mutex>(slot->mu);
        // sized by tsvpp_prepare_batch: then this is a no-op; otherwise the first call for a (size, n) pays one hipMalloc
        sts = scratch_grow(ctx, slot, frame_scratch * (size_t)n);
        if (sts != TSVPP_OK) return sts;
        scratch = slot->buf;
    }
    const OutKind out_kind = single ? single_pass_kind(pl) : pl.out;
    if (two_pass && d.nt_stores < 0) d.nt_stores = 0; // the intermediate is read back at once: keep it in L2 / MALL
    // Frames per launch: TSVPP_MAX_BATCH with the pointer triples in the kernarg segment; out of a device-resident table (single-pass requests) up to
    // TSVPP_MAX_TABLE_LAUNCH, less where the grid would outgrow 2^31 threads (the smallest tile any kernel uses is 64 x 4 output pixels per workgroup of 256).
    const bool from_table = tab.y != nullptr && !two_pass;
    // TSVPP_OPT_INPUTS_READY: never out of a table (tsvpp_table_set's upload is enqueued on this stream: the launch must wait for it), never for the two-pass
    // formats (pass 1 writes the stream's scratch buffer, which the previous call's pass 2 may still be reading)
    // ... and only for launches small enough to gain from starting beside their predecessor: measured on the headline (profiles/r06_curve_values.txt, one consumer on two
    // streams) 1 / 2 / 4 frames per launch 0.367 / 0.573 / 0.661 -> 0.410 / 0.583 / 0.682 of the roofline, but 8 frames (113 MB) 0.719 -> 0.682
    d.any_order = ((ctx->inputs_ready == 1 || ctx->inputs_ready == 2) && tab.y == nullptr && !two_pass &&
                   (size_t)n * plan_moved_bytes(pl) <= TSVPP_BARRIER_FREE_MAX_BYTES) ? 1 : 0;
    int max_launch = TSVPP_MAX_BATCH;
    if (from_table) {
        const long wg_per_frame = (long)((pl.dst_w + 63) / 64) * ((pl.dst_h + 3) / 4);
        long cap = ((1L << 31) / 256) / (wg_per_frame > 0 ? wg_per_frame : 1);
        if (cap > TSVPP_MAX_TABLE_LAUNCH) cap = TSVPP_MAX_TABLE_LAUNCH;
        if (cap > TSVPP_MAX_BATCH) max_launch = (int)cap;
    }
    for (int base = 0; base < n; base += max_launch) {
        const int cnt = (n - base < max_launch) ? n - base : max_launch;
        FrameTable t = {};
        if (from_table) {
            t.y.ext = (const uint8_t *const *)(tab.y + base);
            t.y.off = (int64_t)y_off;
            t.uv.ext = (const uint8_t *const *)(tab.uv + base);
            t.uv.off = (int64_t)uv_off;
            t.out.ext = (void *const *)(tab.out + base);
            t.out.off = 0;
        } else {
            for (int f = 0; f < cnt; f++) {
                t.y[f] = in[base + f].y + y_off;
                t.uv[f] = in[base + f].uv + uv_off;
                t.out[f] = (two_pass && pl.mode != M_NONE) ? (void *)(scratch + (size_t)(base + f) * frame_scratch) : outs[base + f];
            }
        }
        d.n_frames = cnt;
        // Vector-store kernels need 16-byte aligned outputs (else: the element-wise gather kernel); scratch frames are
        // 256-byte aligned.
        bool vec = true;
        if (!two_pass)
            for (int f = 0; f < cnt; f++)
                if (((uintptr_t)outs[base + f] & 15) != 0) vec = false;
        if (!(two_pass && pl.mode == M_NONE)) {
            LaunchRecord rec;
            const bool keep = replayable && !two_pass && n <= max_launch; // one launch group, one pass
            hipError_t e;
            {
                RecordArm arm(keep ? &rec : nullptr);
                e = launch_fused(pl.mode, out_kind, vec, d, t, (hipStream_t)stream);
            }
            if (e != hipSuccess) return (int)e;
            if (keep && rec.count == 1 && rec.fn) remember_launch(ctx, epoch0, n, in, p, pitch_y, pitch_uv, aligned4, vec, y_off, uv_off, rec);
        }
        if (two_pass) {
            int fpy = pitch_y, fpuv = pitch_uv;
            if (pl.mode != M_NONE) { // resized intermediate: tight, pitch = width (as the reference's)
                for (int f = 0; f < cnt; f++) {
                    t.y[f] = scratch + (size_t)(base + f) * frame_scratch;
                    t.uv[f] = t.y[f] + (size_t)pl.dst_w * pl.dst_h;
                    t.out[f] = outs[base + f];
                }
                fpy = fpuv = pl.dst_w;
            }
            hipError_t e = launch_format(pl.fourcc, pl.f32, t, cnt, fpy, fpuv, pl.dst_w, pl.dst_h, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    return TSVPP_OK;
}

int tsvpp_convert_batch(tsvpp_ctx *ctx, int n, const tsvpp_nv12 *in, const tsvpp_params *p, void *const *outs, void *stream) {
    return convert_impl(ctx, n, in, p, outs, stream, TableCols());
}

// ---- persistent frame tables (include/tsvpp.h) --------------------------------------------------------------------------------------------------------
struct tsvpp_table {
    tsvpp_ctx *ctx = nullptr;   // null once the context has been destroyed (tsvpp_destroy releases the table's device memory; tsvpp_table_destroy then only frees the host side)
    int capacity = 0;
    uint64_t *dev = nullptr;    // three columns of `capacity` entries: y | uv | out
    std::vector<tsvpp_nv12> in; // host mirror (geometry checks, alignment decisions, the two-pass formats): updated only after an upload has been enqueued in full
    std::vector<void *> outs;
    std::vector<uint8_t> set;   // entry has been set
    bool have_geom = false;
    tsvpp_nv12 geom = {};
    std::mutex mu;
    // Upload staging: until round 5 every upload was copied out of ONE pinned mirror that the next tsvpp_table_set rewrote at once -- a second set of the
    // same entries before the first upload had run made the first copy pick up the second call's pointers, against the stream ordering the header promises.  Each upload now
    // has its own pinned slot, reused only after the event recorded behind its copies has completed; a slot that is too small is replaced (the old one freed: its event is done).
    struct Slot {
        uint64_t *buf = nullptr;
        size_t entries = 0; // capacity in table entries (3 columns each)
        hipEvent_t done = nullptr;
        bool busy = false;
    };
    static constexpr int kSlots = 4;
    Slot slots[kSlots];
    int next_slot = 0;
};

static void table_release_device(tsvpp_table *t) { // caller has selected the device
    if (t->dev) (void)hipFree(t->dev);
    t->dev = nullptr;
    for (auto &sl : t->slots) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.buf) (void)hipHostFree(sl.buf);
        sl = tsvpp_table::Slot();
    }
}

static void ctx_release_tables(tsvpp_ctx *ctx) { // caller has selected the device
    std::lock_guard<std::mutex> lk(ctx->tables_mu);
    for (tsvpp_table *t : ctx->tables) {
        std::lock_guard<std::mutex> tl(t->mu);
        table_release_device(t);
        t->ctx = nullptr;
    }
    ctx->tables.clear();
}

int tsvpp_table_create(tsvpp_ctx *ctx, int capacity, tsvpp_table **out_table) {
    if (!ctx || !out_table || capacity < 1 || capacity > (1 << 24)) return TSVPP_ERROR;
    *out_table = nullptr;
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    tsvpp_table *t = new tsvpp_table;
    t->ctx = ctx;
    t->capacity = capacity;
    const size_t bytes = (size_t)3 * capacity * sizeof(uint64_t);
    hipError_t e = hipMalloc((void **)&t->dev, bytes);
    if (e == hipSuccess) e = hipMemset(t->dev, 0, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (t->dev) (void)hipFree(t->dev);
        delete t;
        return (int)e;
    }
    t->in.resize((size_t)capacity);
    t->outs.assign((size_t)capacity, nullptr);
    t->set.assign((size_t)capacity, 0);
    {
        std::lock_guard<std::mutex> lk(ctx->tables_mu);
        ctx->tables.insert(t);
    }
    *out_table = t;
    return TSVPP_OK;
}

void tsvpp_table_destroy(tsvpp_table *table) {
    if (!table) return;
    if (tsvpp_ctx *ctx = table->ctx) { // (a table that outlived its context was already released by tsvpp_destroy)
        {
            std::lock_guard<std::mutex> lk(ctx->tables_mu);
            ctx->tables.erase(table);
        }
        DeviceGuard guard(ctx);
        table_release_device(table);
    }
    delete table;
}

int tsvpp_table_set(tsvpp_table *table, int first, int n, const tsvpp_nv12 *in, void *const *outs, void *stream) {
    if (!table || !table->ctx || !in || !outs || first < 0 || n < 0 || (long)first + n > table->capacity) return TSVPP_ERROR;
    if (n == 0) return TSVPP_OK;
    std::lock_guard<std::mutex> lk(table->mu);
    const tsvpp_nv12 g = table->have_geom ? table->geom : in[0];
    const int gpy = g.pitch_y ? g.pitch_y : g.width, gpuv = g.pitch_uv ? g.pitch_uv : g.width;
    for (int f = 0; f < n; f++) { // one geometry per table, as one per batch (tsvpp_convert_batch)
        if (!in[f].y || !in[f].uv || !outs[f]) return TSVPP_ERROR;
        if (in[f].width != g.width || in[f].height != g.height) return TSVPP_UNSUPPORTED;
        if ((in[f].pitch_y ? in[f].pitch_y : in[f].width) != gpy || (in[f].pitch_uv ? in[f].pitch_uv : in[f].width) != gpuv) return TSVPP_UNSUPPORTED;
    }
    DeviceGuard guard(table->ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    // this upload's staging slot: wait for the copies that last read it, grow it if needed
    tsvpp_table::Slot &sl = table->slots[table->next_slot];
    if (sl.busy) {
        const hipError_t e = hipEventSynchronize(sl.done);
        if (e != hipSuccess) return (int)e;
        sl.busy = false;
    }
    if (sl.entries < (size_t)n) {
        if (sl.buf) (void)hipHostFree(sl.buf);
        sl.buf = nullptr;
        sl.entries = 0;
        size_t want = 64;
        while (want < (size_t)n) want *= 2;
        const hipError_t e = hipHostMalloc((void **)&sl.buf, 3 * want * sizeof(uint64_t), hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
        sl.entries = want;
    }
    if (!sl.done) {
        const hipError_t e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
    }
    for (int f = 0; f < n; f++) {
        sl.buf[f] = (uint64_t)(uintptr_t)in[f].y;
        sl.buf[(size_t)n + f] = (uint64_t)(uintptr_t)in[f].uv;
        sl.buf[2 * (size_t)n + f] = (uint64_t)(uintptr_t)outs[f];
    }
    const size_t cap = (size_t)table->capacity;
    for (int col = 0; col < 3; col++) { // the three column ranges: asynchronous, ordered on `stream`
        const hipError_t e = hipMemcpyAsync(table->dev + col * cap + first, sl.buf + (size_t)col * n, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e != hipSuccess) {
            // a column may already be on its way: the entries are in an unknown state on the device -> they count as never set (tsvpp_convert_table refuses them)
            for (int f = 0; f < n; f++) table->set[(size_t)first + f] = 0;
            (void)hipEventRecord(sl.done, (hipStream_t)stream);
            sl.busy = true;
            return (int)e;
        }
    }
    const hipError_t er = hipEventRecord(sl.done, (hipStream_t)stream);
    if (er != hipSuccess) { // cannot tell when the slot is free again: wait for the stream now
        (void)hipStreamSynchronize((hipStream_t)stream);
    } else {
        sl.busy = true;
    }
    table->next_slot = (table->next_slot + 1) % tsvpp_table::kSlots;
    for (int f = 0; f < n; f++) { // the host mirror follows the device: only now
        const size_t k = (size_t)first + f;
        table->in[k] = in[f];
        table->outs[k] = outs[f];
        table->set[k] = 1;
    }
    table->geom = g;
    table->have_geom = true;
    return TSVPP_OK;
}

int tsvpp_convert_table(tsvpp_ctx *ctx, const tsvpp_table *table, int first, int n, const tsvpp_params *p, void *stream) {
    if (!ctx || !table || table->ctx != ctx || !p || first < 0 || n < 0 || (long)first + n > table->capacity) return TSVPP_ERROR;
    if (n == 0) return TSVPP_OK;
    for (int f = 0; f < n; f++)
        if (!table->set[(size_t)first + f]) return TSVPP_ERROR; // an entry that was never set
    const size_t cap = (size_t)table->capacity;
    TableCols cols;
    cols.y = table->dev + first;
    cols.uv = table->dev + cap + first;
    cols.out = table->dev + 2 * cap + first;
    return convert_impl(ctx, n, table->in.data() + first, p, table->outs.data() + first, stream, cols);
}

int tsvpp_convert(tsvpp_ctx *ctx, const tsvpp_nv12 *in, const tsvpp_params *p, void *out, void *stream) {
    void *outs[1] = { out };
    return tsvpp_convert_batch(ctx, 1, in, p, outs, stream);
}

int tsvpp_trim(tsvpp_ctx *ctx, size_t *released_bytes) {
    if (released_bytes) *released_bytes = 0;
    if (!ctx) return TSVPP_ERROR;
    DeviceGuard guard(ctx);
    if (guard.status != TSVPP_OK) return guard.status;
    ctx->epoch.fetch_add(1); // replay records may hold the addresses of retired table sets
    size_t n = geo_cache_trim(ctx->geo);
    {
        std::lock_guard<std::mutex> lk(ctx->scratch_mu);
        for (uint8_t *b : ctx->retired) (void)hipFree(b); // outgrown NV12 scratch buffers of the two-pass formats (their sizes are not tracked: not counted)
        ctx->retired.clear();
    }
    if (released_bytes) *released_bytes = n;
    return TSVPP_OK;
}

int tsvpp_set_option(tsvpp_ctx *ctx, int option, int value) {
    if (!ctx) return TSVPP_ERROR;
    switch (option) {
    case TSVPP_OPT_UNSAFE_COEFFS:
        ctx->unsafe_coeffs = value ? 1 : 0;
        return TSVPP_OK;
    case TSVPP_OPT_COLOR_G_TERM:
        if (value < 0 || value > 2) return TSVPP_ERROR;
        ctx->color_g = value;
        ctx->epoch.fetch_add(1);
        return TSVPP_OK;
    case TSVPP_OPT_INPUTS_READY:
        if (value < 0 || value > 3) return TSVPP_ERROR;
        ctx->inputs_ready = value;
        ctx->epoch.fetch_add(1);
        return TSVPP_OK;
    default: return TSVPP_UNSUPPORTED;
    }
}

int tsvpp_get_option(const tsvpp_ctx *ctx, int option, int *value) {
    if (!ctx || !value) return TSVPP_ERROR;
    switch (option) {
    case TSVPP_OPT_INPUTS_READY: *value = ctx->inputs_ready; return TSVPP_OK;
    case TSVPP_OPT_COLOR_G_TERM: *value = ctx->color_g; return TSVPP_OK;
    case TSVPP_OPT_UNSAFE_COEFFS: *value = ctx->unsafe_coeffs; return TSVPP_OK;
    default: return TSVPP_UNSUPPORTED;
    }
}

int tsvpp_get_coeffs(const tsvpp_ctx *ctx, tsvpp_coeffs *out) {
    if (!ctx || !out) return TSVPP_ERROR;
    *out = ctx->coeffs;
    return TSVPP_OK;
}

int tsvpp_set_coeffs(tsvpp_ctx *ctx, const tsvpp_coeffs *in) {
    if (!ctx || !in) return TSVPP_ERROR;
    // Every parity statement of this library is about the reference's literals (reference src/ColorConversion.cu:23-36).  A block that differs from them -- by a bit --
    // is refused unless the caller has said it wants one; the multi-GPU broadcast sets the block it verified against the defaults, which passes.
    tsvpp_coeffs def;
    tsvpp_default_coeffs(&def);
    if (std::memcmp(&def, in, sizeof(def)) != 0 && !ctx->unsafe_coeffs) return TSVPP_UNSUPPORTED;
    ctx->coeffs = *in;
    ctx->epoch.fetch_add(1);
    return TSVPP_OK;
}

int tsvpp_area_pattern(float scale, float *out, int max_floats, int *taps) {
    std::vector<float> tab;
    int rows = 0, t = 0;
    if (!build_area_rows(scale, tab, rows, t)) return TSVPP_UNSUPPORTED;
    if (taps) *taps = t;
    if (out && (long)tab.size() <= (long)max_floats) std::memcpy(out, tab.data(), tab.size() * sizeof(float));
    return rows;
}

int tsvpp_describe(const tsvpp_params *p, int in_width, int in_height, int pitch_y, int pitch_uv, int n_frames, int aligned_outputs, char *buf,
                   size_t buf_len) {
    if (!p || !buf || buf_len == 0) return TSVPP_ERROR;
    buf[0] = 0;
    Plan pl;
    int sts = make_plan(p, in_width, in_height, pl);
    if (sts != TSVPP_OK) return sts;
    if (pitch_y == 0) pitch_y = in_width;
    if (pitch_uv == 0) pitch_uv = in_width;
    if (pitch_y < in_width || pitch_uv < in_width || n_frames < 1) return TSVPP_ERROR;
    tsvpp_ctx tmp; // knobs only: no device, no streams
    tsvpp_default_coeffs(&tmp.coeffs);
    read_env_knobs(&tmp);
    LaunchDesc d;
    fill_desc(&tmp, pl, pitch_y, pitch_uv, d);
    // frame pointers are assumed 256-byte aligned; the crop origin decides the rest
    d.in_aligned4 = (pitch_y % 4 == 0 && pitch_uv % 4 == 0 && ((size_t)pl.off_y * pitch_y + pl.off_x) % 4 == 0 &&
                     ((size_t)(pl.off_y / 2) * pitch_uv + pl.off_x) % 4 == 0)
                        ? 1
                        : 0;
    static const float dummy_f[4] = { 0, 0, 0, 0 };
    static const AreaQRow dummy_q = {};
    if (pl.mode == M_AREA_DOWN) { // the same table properties get_area_table derives, without touching a device
        int shift[2] = { -1, -1 }, uniform[2] = { 0, 0 };
        bool dyadic[2] = { false, false };
        for (int axis = 0; axis < 2; axis++) {
            std::vector<float> tab;
            int rows = 0, taps = 0;
            if (!build_area_rows(axis ? pl.yr : pl.xr, tab, rows, taps)) return TSVPP_UNSUPPORTED;
            std::vector<AreaQRow> q;
            dyadic[axis] = quantise_area_rows(tab, rows, taps, q, shift[axis]);
            if (dyadic[axis]) {
                uniform[axis] = q[0].sum;
                for (const AreaQRow &e : q)
                    if (e.sum != q[0].sum) uniform[axis] = 0;
            }
            (axis ? d.ny : d.nx) = rows;
            (axis ? d.ry : d.rx) = taps;
            (axis ? d.nky : d.nkx) = (taps + 3) / 4;
            if (axis == 0) d.as_ones_x = area_ones_end(tab, rows, taps);
        }
        d.patx = d.paty = d.patx4 = d.paty4 = dummy_f;
        if (!(dyadic[0] && dyadic[1]) && tmp.area_divtab && (long)d.nx * d.ny <= (1L << 18)) d.area_div = dummy_f; // as get_area_div would
        if (dyadic[0] && dyadic[1] && dyadic_usable(pl.xr, pl.yr, shift[0], shift[1])) {
            d.qx = d.qy = &dummy_q;
            d.box_rx = (d.nx == 1 && shift[0] == 0 && uniform[0] == d.rx) ? d.rx : 0;
            d.box_ry = (d.ny == 1 && shift[1] == 0 && uniform[1] == d.ry) ? d.ry : 0;
            if (uniform[0] > 0 && uniform[1] > 0 && (long)uniform[0] * uniform[1] < 4096) d.area_rcp = 1.0f / (float)(uniform[0] * uniform[1]);
        }
    }
    const bool single = aligned_outputs != 0 && single_pass_format(pl, d, n_frames, nullptr, nullptr);
    const OutKind out_kind = single ? single_pass_kind(pl) : pl.out;
    const bool two_pass = (pl.fourcc == TSVPP_UYVY || pl.fourcc == TSVPP_YUV444) && !single;
    d.n_frames = n_frames < TSVPP_MAX_BATCH ? n_frames : TSVPP_MAX_BATCH;
    static const char *const out_names[O_COUNT_ALL] = { "u8_planar", "u8_merged", "f32_planar", "f32_merged", "nv12_u8", "nv12_f32", "y800_u8", "y800_f32", "hsv_f32", "uyvy_u8", "yuv444_u8", "uyvy_f32" };
    static const char *const mode_names[M_COUNT] = { "none", "nearest", "bilinear", "bicubic", "area_down", "area_up" };
    LaunchInfo info = {};
    info.kernel = "(none)";
    if (!(two_pass && pl.mode == M_NONE)) {
        FrameTable t = {};
        hipError_t e = launch_fused(pl.mode, out_kind, aligned_outputs != 0 || two_pass, d, t, nullptr, &info);
        if (e != hipSuccess) return (int)e;
    }
    char kname[128]; // the launcher's spelling without blanks: one token per key=value pair
    size_t kn = 0;
    for (const char *c = info.kernel; *c && kn + 1 < sizeof(kname); c++)
        if (*c != ' ') kname[kn++] = *c;
    kname[kn] = 0;
    std::snprintf(buf, buf_len, "mode=%s out=%s src=%dx%d dst=%dx%d kernel=%s shape=%dx%d rpt=%d dma=%d lds=%d grid=%d tiles=%dx%d frames=%d tail=%d geo=%d%s",
                  mode_names[pl.mode], out_names[out_kind], pl.src_w, pl.src_h, pl.dst_w, pl.dst_h, kname, info.tx, info.ty, info.rpt, info.dma,
                  info.lds_bytes, info.grid, info.tiles_x, info.tiles_y, d.n_frames, info.tail, info.geo,
                  two_pass ? (pl.fourcc == TSVPP_UYVY ? " pass2=fmt_uyvy" : " pass2=fmt_yuv444") : "");
    return TSVPP_OK;
}

const char *tsvpp_strerror(int status) {
    switch (status) {
    case TSVPP_OK: return "ok";
    case TSVPP_REPEAT: return "VREADER_REPEAT: repeat the request";
    case TSVPP_UNSUPPORTED: return "VREADER_UNSUPPORTED: unsupported parameters (odd size, unknown FourCC/resize type, mixed batch geometry)";
    case TSVPP_ERROR: return "VREADER_ERROR: invalid argument or consumer pool exhausted";
    default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

const char *tsvpp_version(void) { return "tsvpp 0.3.0 gfx950"; }

} // extern "C"
