// Launch descriptors shared by the host API (tsvpp_api.cpp) and the gfx950 kernels
// (vpp_kernels.hip).  Product code -- never includes anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>

#include "tsvpp.h"

#ifndef AS_RING_ROWS
#define AS_RING_ROWS 4 // source rows in a wave's ring of the streaming AREA kernel (a power of two)
#endif

namespace tsvpp {

// Resize mode of the fused kernel.  AREA splits in two exactly as the reference's host
// dispatch does (src/Resize.cu:435): weighted box if both ratios > 1, else the bilinear variant.
enum Mode : int { M_NONE = 0, M_NEAREST, M_BILINEAR, M_BICUBIC, M_AREA_DOWN, M_AREA_UP, M_COUNT };

// Per-launch pointer table.  Launches of up to TSVPP_MAX_BATCH frames carry it BY VALUE in the kernarg segment (3 KiB): the kernel reads its frame's three
// pointers with scalar loads, no device-side descriptor buffer and no host->device copy per batch.  Launches out of a persistent, device-resident table
// (tsvpp_table, round 5: up to TSVPP_MAX_TABLE_LAUNCH frames per launch) set `ext` instead: the same scalar loads, through the constant address space, off
// the table's column + `off` (the crop origin, which belongs to the request, not to the table).  Kernels index both the same way: t.y[frame].
template <class T> struct PtrCol {
    T v[TSVPP_MAX_BATCH];
    const T *ext; // device memory, or null
    int64_t off;  // bytes added to every `ext` entry
    __host__ __device__ __forceinline__ T operator[](int f) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (ext) {
            typedef const __attribute__((address_space(4))) uint64_t *CP; // written by the host before the launch, never by a kernel: scalar loads
            typedef __attribute__((address_space(1))) uint8_t *GP;        // ... and what they hold are GLOBAL addresses: without this the kernels' loads / stores become flat_*
            return (T)(GP)(uintptr_t)(((CP)(uintptr_t)ext)[f] + (uint64_t)off);
        }
        {
            typedef __attribute__((address_space(1))) uint8_t *GP;
            return (T)(GP)(uintptr_t)v[f];
        }
#endif
        return v[f]; // (host: only tables without `ext` are read back, vpp_formats.hip)
    }
    __host__ __forceinline__ T &operator[](int f) { return v[f]; }
};
struct FrameTable {
    PtrCol<const uint8_t *> y, uv;
    PtrCol<void *> out;
};

// One row of an AREA weight table quantised to integers (weights * 2^shift), for requests whose
// weights are all dyadic (integer ratios, 1.5, 2.25, ...): then every partial sum of the
// reference's float accumulation is exact, the summation order is irrelevant, and the weighted box
// sum can be taken with v_dot4_u32_u8 on packed bytes.
struct AreaQRow {
    int32_t sum;     // sum of the integer weights of the row
    uint32_t w[2];   // weights 0..7 packed as bytes (luma taps; also the vertical weights)
    uint32_t wu[4];  // the same weights spread over even bytes: (w0,0,w1,0) ... for interleaved U,V
    int32_t pad;
};

struct GeoCache; // host side: device-resident, host-built tables per request geometry (vpp_bilinear.hip, vpp_bicubic_cols.hip)

// One output column / row of a BICUBIC request (vpp_bicubic_cols.hip), evaluated once per request on the host: first sample of the
// 4-sample window, tap selector (byte k = offset of tap k from the window start: the reference's edge rule), largest tap offset,
// the weight, and the Keys coefficients quantised to 2^-22 as three planes of signed base-256 digits (digit of tap k in byte k).
struct BcEntry {
    int ws;
    uint32_t sel;
    int l0, l1, l2, bias;
    float w;
    int maxoff;
};

struct LaunchDesc {
    // logical source = the crop box if crop is active, else the whole frame; the frame
    // pointers in FrameTable are already advanced to its top-left corner
    int src_w, src_h;
    int pitch_y, pitch_uv;
    int dst_w, dst_h;
    float xr, yr; // (float)src_w / dst_w, (float)src_h / dst_h   (src/Resize.cu:418-419)
    int swap_rb;  // BGR24
    int color_g;  // TSVPP_OPT_COLOR_G_TERM: 0 left product of the green chroma term fused (default), 1 plain IEEE, 2 right product fused (chroma_terms, vpp_device.h)
    tsvpp_coeffs k;
    // AREA-down weight tables: nx rows of rx floats, ny rows of ry floats (rx = ceil(xr))
    const float *patx, *paty;
    int nx, ny, rx, ry;
    const AreaQRow *qx, *qy; // non-null: dyadic AREA tables (integer box sums)
    const float *patx4, *paty4; // float weight rows zero-padded to 4 * nkx / 4 * nky entries (direct float AREA kernel)
    const float *area_div;      // nx * ny divisors of the float AREA kernels (row = column pattern), or null: the kernel sums the weights itself
    int nkx, nky;
    float area_rcp;          // != 0: every (column, row) pattern pair has the same divisor S = sum(wx) * sum(wy); this is 1 / S
    // grid decomposition (filled by launch_fused)
    int tiles_x, tiles_y, n_frames;
    int blocks_per_xcd; // ceil(total_tiles / 8)
    int tx, ty, tx_shift; // workgroup = tx x ty thread tiles of 4 x 2 output pixels
    int rpt;              // row pairs per thread (1; 2 in the 2x2-tap kernel: thread tile 4 x 4)
    // LDS staging bounds (staged kernel): max source bytes per row, rows, 16-byte chunks per row
    int lds_span_y, lds_rows_y, lds_cpr_y;
    int lds_span_uv, lds_rows_uv, lds_cpr_uv;
    int lds_slot_y, lds_slot_uv; // log2 of the lanes serving one staged row (register-staged path)
    uint32_t lds_magic_y, lds_magic_uv; // 2^32 / chunks-per-row + 1 (LDS-DMA path: slot -> row by multiply-high)
    int wx_zero, wy_zero; // BILINEAR: every weight of that axis is zero (host): the samplers do not fetch the taps it multiplies
    int point_kind;   // PointKind: >= 0 when the request is a pure point sampler (host decides, see vpp_axis.h)
    int in_aligned4;  // every frame's (crop-adjusted) plane pointers and both pitches are multiples of 4
    int force_gather; // debugging / A-B: 1 = always use the global-gather kernel
    // tuning knobs (ctx options / TSVPP_* environment, see tsvpp_api.cpp)
    int nt_stores;    // 1 = non-temporal output stores
    int tile_order;   // 0 = tile row per XCD (default), 1 = raster, 2 = XCD-contiguous runs
    int shape_tx, shape_ty; // != 0: force the workgroup shape
    int area_direct;        // (launch_fused) dyadic AREA straight from global memory
    float area_direct_min;  // use it when both ratios are >= this (0 = never)
    float area_direct_fmin; // the same for the float-weight direct kernel
    int rpt_pref;           // preferred row pairs per thread for the 2x2-tap kernel (TSVPP_RPT)
    int dma;                // 1 = stage with LDS-DMA (global_load_lds_dwordx4) where the kernel supports it
    int num_cus;            // compute units of the device (how many workgroups a launch should have)
    int box_rx, box_ry;     // host: the AREA weight table of that axis is one row of all ones (integer ratio) -> its tap count, else 0
    int area_box_pref, area_box; // contiguous-run box kernel allowed (TSVPP_AREA_BOX) / chosen by launch_fused
    int w_dyadic;           // host: every interpolation weight of this BILINEAR / BICUBIC / AREA-up request is a multiple of 1/16
    int bil_int_pref, bil_int; // integer 2x2-tap thread tile allowed (TSVPP_BILINEAR_INT) / chosen by launch_fused
    int bil_win_pref, bil_win;   // window form of the float 2x2-tap thread tile allowed (where it measured faster) / chosen by launch_fused
    int u8_xchg;                 // uint8 merged outputs: in-wave LDS exchange -> 16-byte stores (TSVPP_U8_XCHG)
    int bicubic_int_pref, bicubic_int; // integer BICUBIC kernel allowed (TSVPP_BICUBIC_INT) / chosen by launch_fused
    int hcs_y, hcs_uv;      // integer BICUBIC kernel: byte stride of one column of the column-major H planes
    int area2_pref, area2;  // 2x2 float AREA kernel allowed (TSVPP_AREA2) / chosen by launch_fused
    int area_cols_pref, area_cols; // column-per-lane float AREA kernel allowed (TSVPP_AREA_COLS) / chosen by launch_fused
    int area_cols_rows;     // its tile height: 8 (one row pair per wave) or 32 (TSVPP_AREA_COLS_ROWS)
    int col0;               // first output column of this launch (0; dst_w & ~3 in the row-tail launch of widths 4 k + 2)
    int lds_budget_kb;      // LDS bytes a workgroup may use for staging + tables (default 40 KiB: four workgroups per CU)
    int luma_only;          // Y800 outputs: the chroma plane is neither staged nor sampled
    // Host-built geometry tables of the 2x2-tap kernel's window tiles (vpp_bilinear.hip, "geometry tables"): tile footprints
    // (scalar loads), one record per output column quad and per output row pair.  geo_pref: allowed (TSVPP_GEO); geo: chosen
    // by launch_bilinear; geo_build: a dry run (info != nullptr) still builds / uploads the tables (tsvpp_prepare_batch);
    // geo_cache: the owning context's cache (host pointer, never dereferenced on the device; null in tsvpp_describe).
    const int4 *geo_tx, *geo_ty;
    const uint4 *geo_col, *geo_row;
    int geo_pref, geo, geo_build;
    // BICUBIC with one wave per tile and one lane per output column (vpp_bicubic_cols.hip): allowed (TSVPP_BICUBIC_COLS: 1 = what the
    // integer kernel does not take, 2 = every BICUBIC request, 0 = never: generic gathers) / chosen by launch_fused (1 with the tie
    // test, 2 exact coefficients); bc_rows: forced tile height
    // (TSVPP_BICUBIC_ROWS, 0 = automatic); bc_sparse: the H plane holds the four taps of each output row (vertical ratio >= 4);
    // bc_dma_pref / bc_dma: source rows through a wave-private LDS-DMA ring (TSVPP_BICUBIC_DMA; horizontal ratios below 3.7) instead of
    // per-lane loads: the 16-byte chunks (lanes) a row segment takes, 0 = direct loads; bc_ring_bytes: that ring; bc_wave_bytes: LDS bytes of one wave (ring + H plane with column stride hcs_y + result tiles)
    int bicubic_cols_pref, bicubic_cols, bc_rows, bc_sparse, bc_wave_bytes, bc_dma_pref, bc_dma, bc_ring_bytes;
    int bc_u8x_pref, bc_u8x; // uint8 outputs leave through the 8 x 4 output side of the streaming kernels (dst_w % 8 == 0, dst_h % 4 == 0; TSVPP_BICUBIC_U8X=0: the 4 x 2 thread tiles)
    // luma columns [dst_w] | chroma pair columns [dst_w / 2] as BcEntry records, then the rows in blocks of four (32 ints: ws x4 | sel x4 |
    // l0 x4 | l1 x4 | l2 x4 | bias x4 | w x4 | pad: a wave fetches four rows' parameters with scalar loads off ONE address): luma row
    // blocks, chroma row blocks; bc_npy / bc_npc = their numbers (rows / 4 rounded up, + 1)
    const BcEntry *bc_tab;
    int bc_npy, bc_npc;
    // float-weight AREA down-scale, one wave per tile with the source rows streamed through a wave-private LDS ring (vpp_area_stream.hip):
    // allowed (TSVPP_AREA_STREAM) / chosen by launch_fused; as_nk: its instantiated tap count / 4 (>= nkx); as_rows / as_min_taps: its tile height (4) and
    // cross-over (40 taps per value), constants since the A/B runs of round 3.  Its LDS sizes travel in bc_wave_bytes / bc_ring_bytes.
    int area_stream_pref, area_stream, as_nk, as_rows, as_min_taps, as_ones_x, as_two; // as_ones_x: column taps 1 .. as_ones_x - 1 weigh 1.0f in every table row; as_two: 1 = a wave's tile is 128 columns wide (two per lane), 0 = 64 (large ratios)
    int r32_pref, r32; // streaming 3 : 2 / 2 : 1 kernels: BILINEAR / AREA / NEAREST for uint8 outputs allowed (TSVPP_R32) / chosen by launch_fused (1..6: vpp_bilinear_r32.hip; 7, 8: BICUBIC, vpp_bicubic_r32.hip)
    // The 2x2-tap kernel's integer window tile with another weight pattern (chosen by launch_fused; vpp_bilinear.hip): at exactly 3 : 2 / 2 : 1 the AREA
    // down-scale taps the SAME two samples per axis as BILINEAR -- only the integer weights and the final division differ.
    // 0 = off; 1 = 3 : 2 (weights (2,1) / (1,2) by index parity, sum / 9); 2 = 2 : 1 ((1,1), sum / 4).  The divisor travels as area_rcp.
    int tap22;
    int tap22_off;     // launch_fused's second attempt: the 2x2-tap integer tile could not be staged, the request runs as the AREA down-scale it is
    int last_col0;     // dst_w = 4 k + 2: first column of the launch's LAST tile column, shifted left so that it ends at the frame's right edge (tile_col0, vpp_device.h); 0 = no shift
    int copy16;        // no resize, Y800 / NV12 uint8 outputs: the planes are copied 16 bytes per lane (vpp_copy16_kernel), chosen by launch_fused
    // BILINEAR at sparse ratios, one wave per 64-column tile, the TAPPED rows staged as LDS-DMA row segments (vpp_bilinear_rows.hip): allowed (TSVPP_BILINEAR_ROWS: 1 = where the
    // byte-gather kernel ran until round 4 -- ratio product >= 12 --, 2 = wherever a segment fits one DMA instruction, 0 = never) / chosen by launch_fused: the 16-byte chunks of one
    // row segment (<= 64), 0 = off; br_waves: waves (64-column tiles) per workgroup; br_rpi: segments per DMA instruction (64 / bil_rows).  A wave's LDS bytes travel in bc_wave_bytes.
    int bil_rows_pref, bil_rows, br_waves, br_rpi;
    int point_rn_pref; // TSVPP_POINT_RN: streaming point sampler at exact integer ratios (vpp_point_rn.hip) allowed; chosen: r32 >= 100
    // TSVPP_OPT_INPUTS_READY (include/tsvpp.h): the launch does not wait for work enqueued earlier on its stream -- its AQL packet goes out with the barrier bit
    // cleared (hipExtAnyOrderLaunch), so the dependent-launch boundary (~1.5-1.9 us between streaming kernels) overlaps the predecessor's drain.  Host-side only.
    int any_order;
    GeoCache *geo_cache;
};

// What the last fused launch of this thread was: tsvpp_api.cpp points `g_launch_rec` at a record while it runs launch_fused and, when exactly ONE kernel was
// launched, keeps the record -- the host function, the grid and the FINAL descriptor -- so that the next call with the same request replays the launch without
// re-running the selection (round 6: a single-frame launch is host-bound; the selection, its cache lookups and mutexes are ~0.5 us of ~4).
struct LaunchRecord {
    const void *fn = nullptr;
    dim3 grid, block;
    uint32_t lds = 0;
    int count = 0; // launches seen since the record was armed
    LaunchDesc d;
};
extern thread_local LaunchRecord *g_launch_rec;
inline void record_launch(const void *fn, dim3 grid, dim3 block, size_t lds, const LaunchDesc &d) {
    LaunchRecord *r = g_launch_rec;
    if (!r) return;
    if (r->count++ == 0) {
        r->fn = fn;
        r->grid = grid;
        r->block = block;
        r->lds = (uint32_t)lds;
        r->d = d;
    }
}

// Every fused-kernel launch goes through this: an ordinary in-order launch, or -- LaunchDesc::any_order -- one whose packet does not wait for its predecessors.
#define TSVPP_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, D, T)                                                                         \
    do {                                                                                                                             \
        tsvpp::record_launch(reinterpret_cast<const void *>(KERNEL), GRID, BLOCK, LDS, D);                                           \
        if ((D).any_order) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (uint32_t)(LDS), STREAM, nullptr, nullptr, hipExtAnyOrderLaunch, D, T); \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, D, T);                                                             \
    } while (0)

//  everything a kernel receives travels in the kernarg segment: 4 KiB, of which HIP's hidden arguments take up to 256 bytes
static_assert(sizeof(LaunchDesc) + sizeof(FrameTable) + 256 <= 4096, "LaunchDesc + FrameTable no longer fit the kernarg segment");

// Device-resident geometry tables, one set per (kind, request geometry, tile shape); owned by a context, freed with it.
struct GeoKey {
    int v[16];
    bool operator<(const GeoKey &o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
struct GeoEntry {
    uint8_t *dev = nullptr; // one allocation (null: the request is not eligible)
    size_t bytes = 0;       // its size
    size_t off_ty = 0, off_col = 0, off_row = 0;
    uint64_t stamp = 0;     // last use (GeoCache::clock)
};
struct GeoCache {
    std::mutex mu; // held while a missing table set is built and uploaded (~0.1 ms, once per geometry): callers of other geometries wait that long
    std::map<GeoKey, GeoEntry> map;
    uint64_t clock = 0;
    // Table sets pushed out of the map.  Their device memory stays VALID until the context is destroyed: a lookup hands out raw device pointers
    // and the caller launches after the lock is released, a captured hipGraph keeps the pointers it was captured with, and a hipFree issued while
    // another thread captures in global mode invalidates that capture -- so nothing is ever freed under a live context.
    std::vector<uint8_t *> retired;
    size_t retired_bytes = 0;
    bool warned = false;
};
constexpr size_t kGeoMaxEntries = 1024;
constexpr size_t kGeoMaxRetiredBytes = (size_t)256 << 20;
// With the lock held, before an insertion.  A full map retires its least recently used table set (the set leaves the map, its memory stays
// allocated: see GeoCache::retired) so that later geometries still get tables; once the retired sets hold kGeoMaxRetiredBytes the cache stops
// growing for good and the caller takes the kernels that compute their own coordinates (false).  Table sets are tens of KiB: a context reaches
// that bound after several thousand distinct geometries.
inline bool geo_cache_make_room(GeoCache *c) {
    if (c->map.size() < kGeoMaxEntries) return true;
    if (c->retired_bytes >= kGeoMaxRetiredBytes) {
        //  from here on every NEW geometry runs the kernels that compute their own coordinates (same bits, 5-20 % slower for the uint8 2x2-tap and the
        // non-dyadic BICUBIC requests) until tsvpp_trim() / tsvpp_destroy() releases the retired sets: say so, once per context
        if (!c->warned) {
            c->warned = true;
            fprintf(stderr, "tsvpp: this context has retired %zu MiB of geometry tables (limit %zu MiB): new geometries get no tables until tsvpp_trim() is called at a quiescent point\n",
                    c->retired_bytes >> 20, kGeoMaxRetiredBytes >> 20);
        }
        return false;
    }
    auto victim = c->map.begin();
    for (auto it = c->map.begin(); it != c->map.end(); ++it)
        if (it->second.stamp < victim->second.stamp) victim = it;
    if (victim->second.dev) {
        c->retired.push_back(victim->second.dev);
        c->retired_bytes += victim->second.bytes;
    }
    c->map.erase(victim);
    return true;
}
GeoCache *geo_cache_create();
size_t geo_cache_trim(GeoCache *c);  // hipFree()s the RETIRED sets (tsvpp_trim: the caller guarantees that nothing in flight or captured still uses them); bytes released
void geo_cache_destroy(GeoCache *c); // hipFree()s the tables: the caller has selected the device

// Output flavour: element type x layout.
// O_NV12_*: the resized NV12 itself (Y plane then interleaved UV plane, tight) -- FourCC NV12 of the
// reference (NV12MergeBuffers, src/ColorConversion.cu:211-233), and as uint8 also the intermediate
// the two-pass formats (UYVY, YUV444; vpp_formats.hip) read.  O_Y800_*: the luma plane alone
// (:95-105).  O_HSV_F32: merged HSV of the normalised RGB (:235-278).  The fp32 flavours are /255.
enum OutKind : int { O_U8_PLANAR = 0, O_U8_MERGED, O_F32_PLANAR, O_F32_MERGED, O_NV12_U8, O_NV12_F32, O_Y800_U8, O_Y800_F32, O_HSV_F32, O_COUNT,
                     // flavours of the streaming 3 : 2 / 2 : 1 kernel only (vpp_bilinear_r32.hip): UYVY / YUV444 (uint8) behind such a resize in ONE pass -- everywhere else UYVY /
                     // YUV444 are a second pass over O_NV12_U8 (vpp_formats.hip).  launch_fused answers hipErrorNotSupported when the request is not that kernel's.
                     O_UYVY_U8 = O_COUNT, O_YUV444_U8, O_UYVY_F32 /* round 6: the same, every value / 255 as a float */, O_COUNT_ALL };

// What launch_fused chose (dry run, see tsvpp_describe).
struct LaunchInfo {
    const char *kernel;
    int tx, ty, rpt, dma, staged, lds_bytes, grid, tiles_x, tiles_y;
    int tail; // a second, element-wise launch covers the two-column row tail (dst_w = 4 k + 2)
    int geo;  // the 2x2-tap kernel reads host-built geometry tables instead of computing coordinates
};

// Launches the fused crop+resize+colour kernel.  `vec` selects the 16-byte/4-byte vector
// store path (needs 16-byte aligned outputs; when dst_w = 4 k + 2 a row's last thread tile takes the generic path).
// Returns hipError_t.
// `info` != nullptr: a dry run -- the selection is recorded there and nothing is launched.
hipError_t launch_fused(Mode mode, OutKind out, bool vec, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info = nullptr);
// (vpp_kernels.hip, for vpp_select.hip) the kernel of a (mode, flavour) pair as chosen; LDS bytes of the AREA kernels' coordinate tables
hipError_t launch_mode(Mode mode, OutKind out, bool vec, bool staged, LaunchDesc &d, const FrameTable &t, size_t lds, hipStream_t stream, LaunchInfo *info);
size_t area_dyadic_table_bytes(size_t cols, size_t rows);
size_t areaf_table_bytes(size_t cols, size_t rows);

// Integer BICUBIC kernel for dyadic weights (vpp_bicubic_int.hip): LDS bytes it needs beyond the staged planes, launch.
size_t bicubic_int_table_bytes(int tw, int th, int rows_y, int rows_uv, int hcs_y, int hcs_uv);
hipError_t launch_bicubic_int(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info);

// BICUBIC, one wave per tile, one lane per output column, no staging (vpp_bicubic_cols.hip).
hipError_t launch_bicubic_cols(OutKind out, bool exact, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info);
// its per-request tables: looked up in / built into the context's cache (never while `stream` is capturing); null if unavailable
const BcEntry *bicubic_cols_tables(const LaunchDesc &d, hipStream_t stream, bool may_build);
int bicubic_cols_rows_padded(int n); // row blocks for n rows

// AREA down-scale with float weights, rows streamed through a wave-private ring (vpp_area_stream.hip).
hipError_t launch_area_stream(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info);

// AREA down-scale at integer horizontal ratios 4..8 from contiguous dword runs (vpp_area_box.hip).
hipError_t launch_area_box(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info);

// BICUBIC at exactly 3 : 2 / 2 : 1 on both axes, every output flavour, straight from global memory (vpp_bicubic_r32.hip; d.r32 = 7 / 8).
hipError_t launch_bicubic_r32(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info);

// Pure point samplers (NEAREST; BILINEAR / BICUBIC with all-zero weights) at an exact integer ratio 3 / 4 / 5 on both axes, streaming (vpp_point_rn.hip; d.r32 = 100 + 10 N + OFF).
hipError_t launch_point_rn(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info);

// BILINEAR at sparse ratios: the tapped rows as LDS-DMA row segments, one wave per tile (vpp_bilinear_rows.hip; d.bil_rows).
hipError_t launch_bilinear_rows(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info);

// BILINEAR at exactly 3 : 2 on both axes, uint8 outputs, straight from global memory (vpp_bilinear_r32.hip).
hipError_t launch_bilinear_r32(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info);

// The 2x2-tap kernel family (vpp_bilinear.hip): BILINEAR / AREA up-scale.
hipError_t launch_bilinear(bool areaup, OutKind out, const LaunchDesc &d, const FrameTable &t, unsigned grid_x, size_t lds_bytes,
                           hipStream_t stream, LaunchInfo *info);

// UYVY / YUV444 from n (<= TSVPP_MAX_BATCH) NV12 frames of one geometry in one launch (vpp_formats.hip);
// t.y / t.uv are the (resized or cropped) NV12 planes, t.out the outputs.
hipError_t launch_format(int fourcc, bool f32, const FrameTable &t, int n, int py, int puv, int w, int h, hipStream_t stream);

} // namespace tsvpp
