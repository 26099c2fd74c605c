// vpp_select.hip -- the HOST side of a launch: which kernel, workgroup shape, rows per thread, staging layout and LDS budget a request gets
// (launch_fused and its sel_* steps).  No kernel lives here: the choice ends in launch_mode (vpp_kernels.hip) or in one of the per-kernel launchers,
// so a changed threshold does not touch a kernel's translation unit (bench.py stamps PMC traffic entries with the hash of the dispatched kernel's
// sources).  tsvpp_describe runs the same code as a dry run; tests/golden/describe_snapshot.json pins its answers for 417 requests.
#include "vpp_device.h"
#include "vpp_up2.h"

namespace tsvpp {

// tile geometry helpers
static int span_bound(Mode m, int n_out, float ratio, int taps) {
    // max over tile positions of (last tap - first tap + 1) for n_out consecutive outputs, +1 spare
    int ext = 0;
    switch (m) {
    case M_BILINEAR: case M_AREA_UP: ext = 1; break;
    case M_BICUBIC: ext = 3; break;
    case M_AREA_DOWN: ext = taps - 1; break;
    default: ext = 0; break;
    }
    return (int)((double)ratio * (double)(n_out - 1)) + ext + 3;
}
static int slot_shift_for(int cpr) {
    int s = 0;
    while ((1 << s) < cpr) s++;
    return s;
}

// ---- streaming kernels at the exact ratios 3 : 2 / 2 : 1 (vpp_bilinear_r32.hip, vpp_bicubic_r32.hip) ----------------------------------------------
// One row per (resize mode, ratio): the kernel instance (LaunchDesc::r32) and the same-box A/B that put the row here.  What a row needs beyond the
// exact ratio is in stream_select below.
struct StreamRow {
    Mode mode;
    int p2; // twice the ratio: 3 = 3 : 2, 4 = 2 : 1, 1 = 1 : 2 (the up-scale)
    int r32;
    const char *evidence;
};
static const StreamRow kStreamRows[] = {
    { M_BILINEAR, 3, 1, "profiles/r02_r32_ab.txt: uint8 1080p -> 720p planar 0.563 -> 0.687, merged 0.509 -> 0.673, NV12 0.578 -> 0.684, Y800 0.490 -> 0.686" },
    { M_AREA_DOWN, 3, 2, "profiles/r02_r32_ab.txt: AREA planar 0.442 -> 0.679, merged 0.418 -> 0.660 (weight rows {1, 1/2}, {1/2, 1})" },
    { M_NEAREST, 3, 3, "profiles/r02_r32_ab.txt: NEAREST 0.624 -> 0.800" },
    { M_BILINEAR, 4, 4, "profiles/r02_r32_ab.txt: 1080p -> 540p 0.579 -> 0.630, 4K -> 1080p merged 0.681 vs 0.684; NOT planar >= 1.5 Mpixel (0.721 on the LDS kernel vs 0.688)" },
    { M_AREA_DOWN, 4, 5, "profiles/r02_r32_ab.txt: 4K -> 1080p AREA 0.558 -> 0.684, merged 0.430 -> 0.623 (one weight row {1, 1})" },
    { M_NEAREST, 4, 6, "profiles/r02_r32_ab.txt: NEAREST merged 0.83 -> 0.99" },
    { M_BICUBIC, 3, 7, "profiles/r04_bicubic_r32_ab.txt: 1080p -> 720p fp32 planar 0.670 -> 0.711, uint8 merged 0.362 -> 0.548, fp32 merged 0.573 -> 0.702" },
    { M_BICUBIC, 4, 8, "profiles/r04_bicubic_r32_ab.txt: 4K -> 1080p fp32 planar 0.652 -> 0.730, uint8 merged 0.368 -> 0.631; 1080p -> 540p 0.619 -> 0.691" },
    { M_BILINEAR, 1, 10, "profiles/r04_up2_ab.txt (vpp_bilinear_up2.hip): uint8 outputs of the 1 : 2 up-scale, 0.34 on the LDS kernel (r04_upscale_u8_probe.txt)" },
    { M_NEAREST, 1, 20, "profiles/r05_rep2_ab.txt (vpp_point_rn.hip, vpp_rep2_kernel): NEAREST at 1 : 2 is 2 x 2 pixel replication" },
    { M_AREA_UP, 1, 20, "profiles/r05_rep2_ab.txt: the AREA up-scale at 1 : 2 has all-zero weights: the same replication" },
};
// The streaming kernel instance of this request, or 0.  `mode` is the mode the launch runs as (an AREA request that took the 2x2-tap integer tile,
// LaunchDesc::tap22, arrives here as BILINEAR and is not eligible: fp32 RGB).
// `point_kind`: the request's PointKind as the host decided it (launch_fused's copy may have been cleared by sel_point).
static int stream_select(Mode mode, OutKind out, bool vec, const LaunchDesc &d, int point_kind) {
    if (!vec || d.force_gather || !d.in_aligned4 || (d.dst_w & 7) != 0 || (d.dst_h & 3) != 0) return 0;
    // pure point samplers at an exact integer ratio 3 / 4 / 5 on both axes (vpp_point_rn.hip): r32 = 100 + 10 N + OFF, OFF = the tap's offset inside its N samples --
    // 0 for NEAREST, (N - 1) / 2 for the BILINEAR / BICUBIC requests whose weights are all zero (N odd).  Every flavour of the colour back end.
    // profiles/r05_point_rn_ab.txt: C4 (4K -> 720p BICUBIC -> BGR24 merged uint8) against vpp_point_kernel
    // uint8 flavours only (TSVPP_POINT_RN=2: fp32 too): fp32 outputs lose 5..12 % against the LDS point kernel, whose 4 x 2 thread tiles store whole lines without an exchange
    const bool prn_u8 = (out == O_U8_PLANAR || out == O_U8_MERGED || out == O_NV12_U8 || out == O_Y800_U8);
    if (point_kind != PK_NONE && d.point_rn_pref && out < O_COUNT && (prn_u8 || d.point_rn_pref == 2) && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC)) {
        for (int n = 3; n <= 5; n++)
            if ((long)d.src_w == (long)n * d.dst_w && (long)d.src_h == (long)n * d.dst_h) {
                if (point_kind == PK_NEAREST) return 100 + 10 * n;
                if (n & 1) return 100 + 10 * n + (n - 1) / 2;
            }
    }
    const int p2 = (2L * d.src_w == 3L * d.dst_w && 2L * d.src_h == 3L * d.dst_h) ? 3 : ((d.src_w == 2 * d.dst_w && d.src_h == 2 * d.dst_h) ? 4 :
                   ((2 * d.src_w == d.dst_w && 2 * d.src_h == d.dst_h) ? 1 : 0));
    if (!p2) return 0;
    const bool f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    const bool u8_flavour = (out == O_U8_PLANAR || out == O_U8_MERGED || out == O_NV12_U8 || out == O_Y800_U8 || out == O_UYVY_U8 || out == O_YUV444_U8 ||
                             out == O_UYVY_F32); // (fp32, but a flavour of this kernel alone: nothing else to lose against)
    int r32 = 0;
    for (const StreamRow &row : kStreamRows)
        if (row.mode == mode && row.p2 == p2) r32 = row.r32;
    if (!r32) return 0;
    if (r32 == 7 || r32 == 8) // BICUBIC: every flavour of the colour back end (TSVPP_BICUBIC_INT=2 keeps the LDS integer kernel, TSVPP_BICUBIC_COLS=2 the column kernel)
        return (d.w_dyadic && d.bicubic_int_pref == 1 && d.bicubic_cols_pref != 2 && out < O_COUNT) ? r32 : 0;
    // 2x2-tap kinds: uint8 flavours; fp32 flavours (round 4, through the shared output side vpp_r32_store.h) tie or lose against the LDS kernels for RGB / BGR
    // (profiles/r04_r32_f32_ab.txt: AREA 1080p -> 720p 0.681 vs 0.683, 4K -> 1080p 0.65 vs 0.72, BILINEAR 0.69 vs 0.77) and win for HSV, whose three
    // divisions per pixel make the launch VALU-bound (AREA 0.56 -> 0.65, BILINEAR 0.59 -> 0.67): HSV takes them, the rest only under TSVPP_R32=2
    if (!d.r32_pref) return 0;
    if (r32 == 10 || r32 == 20) // the 1 : 2 up-scales: uint8 RGB / BGR / NV12 / Y800 (fp32 outputs are output-bound on the LDS kernel already: 0.81; TSVPP_R32=2 takes them too)
        return (out < O_COUNT && (u8_flavour || (f32_out && d.r32_pref == 2))) ? r32 : 0;
    if (!(u8_flavour || (f32_out && out < O_COUNT && (out == O_HSV_F32 || d.r32_pref == 2)))) return 0;
    if (mode == M_AREA_DOWN) { // the weight pattern the kernel instance has compiled in
        const bool pat = d.qx && d.qy && d.rx == 2 && d.ry == 2 && d.area_rcp != 0.0f && (p2 == 3 ? (d.nx == 2 && d.ny == 2) : (d.nx == 1 && d.ny == 1));
        if (!pat) return 0;
    }
    if (r32 == 4 && out == O_U8_PLANAR && (long)d.dst_w * d.dst_h >= 1500000L && d.r32_pref != 2) return 0; // (see its row)
    return r32;
}
// Workgroup shape of a streaming launch (thread tile = 8 columns x 4 rows).  Measured: 64 x 4 threads (512 x 16 pixels) +0.4..3 % over 32 x 8 and 16 x 16 for
// the uint8 flavours of the 2x2-tap kinds (profiles/r02_r32_ab.txt); the exceptions below each carry their file.
static void stream_shape(int r32, OutKind out, const LaunchDesc &d, int &tx, int &ty) {
    const bool f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    const int n = d.dst_w / 8; // threads per output row
    auto waste = [&](int w) { return (double)((n + w - 1) / w * w) / (double)n - 1.0; };
    tx = 64;
    ty = 4;
    if (r32 >= 100) { // streaming point samplers: as the 2x2-tap kinds below
        if (f32_out) ty = out == O_HSV_F32 ? 4 : 2;
        return;
    }
    if (r32 == 10 || r32 == 20) return; // the 1 : 2 up-scales: 64 x 4 threads (neighbour dwords by wave shuffle; not swept yet)
    if (r32 < 7) {
        if (f32_out) ty = out == O_HSV_F32 ? 4 : 2; // fp32 outputs want short tiles (as the BICUBIC kernel below)
        // YUV444, the one VALU-bound flavour of the 2x2-tap kinds (125-137 VGPRs): lanes past the right edge cost what they idle -- 1280 columns = 2.5
        // rows of 64 threads: 0.465 -> 0.567 on 32 x 4 (profiles/r04_r32_shape_1280.txt); the other flavours do not care
        if (out == O_YUV444_U8 && waste(64) > 0.08 && waste(32) <= 0.08) {
            tx = 32;
            ty = 4;
        }
        return;
    }
    // BICUBIC.  uint8 outputs are VALU-bound (77 % busy, profiles/r04_bicubic_r32_pmc.txt), so idle lanes cost what they idle: 1280 columns = 160
    // threads -- 32-wide workgroups +9 % there; 1920 and 960 columns (240 / 120 threads) lose 6.7 % of a 64-wide row and still prefer it (longer store
    // runs).  fp32 outputs are bound by the issue of memory instructions: always 64 wide (neighbour dwords by wave shuffle instead of two loads per
    // row) and SHORT tiles, as the 2x2-tap kernel: two thread rows (profiles/r04_bicubic_r32_shapes.txt, r04_bicubic_r32_ab.txt); HSV -- three
    // divisions per pixel, VALU-bound again -- four (0.60 vs 0.53).
    tx = (f32_out || waste(64) <= 0.08) ? 64 : (waste(32) <= 0.08 ? 32 : (waste(16) < waste(32) ? 16 : 32));
    ty = f32_out ? (out == O_HSV_F32 ? 4 : 2) : 256 / tx;
    if (ty > 8) ty = 8;
}

// ---- launch_fused, step by step -------------------------------------------------------------------------------------------------------------------
// What launch_fused decides besides the LaunchDesc fields.  Each sel_* function below is one decision with the measurements that set its thresholds;
// launch_fused calls them in order (tests/golden/describe_snapshot.json pins the outcome for 417 requests).
struct FusedSel {
    int shapes[5][2] = { { 32, 8 }, { 32, 4 }, { 16, 4 }, { 0, 0 }, { 0, 0 } }; // candidate workgroup shapes, largest first
    size_t lds_budget = 40 * 1024, lds_bytes = 0, as_lds = 0, br_lds = 0;
    bool staged = false, f32_out = false, two_tap = false;
};
static long fused_workgroups(const LaunchDesc &d, const int *sh, int rpt) {
    return (long)((d.dst_w + sh[0] * PXW - 1) / (sh[0] * PXW)) * ((d.dst_h + sh[1] * PXH * rpt - 1) / (sh[1] * PXH * rpt)) * d.n_frames;
}

// 1. AREA on the 2x2-tap integer tile (may turn `mode` into M_BILINEAR)
static void sel_tap22(Mode &mode, OutKind out, bool vec, LaunchDesc &d) {
    // AREA down-scale at exactly 3 : 2 / 2 : 1 with fp32 RGB / BGR / NV12 outputs: it taps the SAME two samples per axis as BILINEAR at that ratio
    // ((int)(r j) == floor((j + 0.5) r - 0.5) for r = 1.5 and 2), so it runs on the 2x2-tap kernel's integer window tile with its own integer weights and a
    // division instead of the shift (LaunchDesc::tap22, vpp_bilinear.hip).  Same-box A/B (profiles/r04_tap22_ab.txt): 1080p -> 720p planar 0.692 -> 0.766
    // (vpp_area_dyadic_kernel before), merged 0.699 -> 0.715, NV12 0.692 -> 0.722, 4K -> 1080p 0.717 -> 0.736 (vpp_area_box_kernel<2> before), merged
    // 0.731 -> 0.782, 1080p -> 540p 0.718 -> 0.750, 4K -> 1440p 0.700 -> 0.762.  Not taken: Y800 (0.623 -> 0.593), NEAREST (measured with weights (1, 0):
    // 0.689 -> 0.678, merged 0.702 -> 0.630 -- the point sampler reads fewer rows), uint8 outputs and HSV (the streaming kernel below).
    // TSVPP_BILINEAR_INT=0 / 2 switch it off.
    d.tap22 = 0;
    {
        const bool f32 = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32); // (Y800: round 6, with two row pairs per thread -- see sel_staged)
        const bool r32x = 2L * d.src_w == 3L * d.dst_w && 2L * d.src_h == 3L * d.dst_h, r21 = d.src_w == 2 * d.dst_w && d.src_h == 2 * d.dst_h;
        // (4 k + 2 columns: the tail launch samples by MODE)
        if (!d.tap22_off && f32 && vec && !d.force_gather && d.bil_int_pref == 1 && d.r32_pref != 2 && (r32x || r21) && (d.dst_w & 3) == 0 && mode == M_AREA_DOWN && d.qx && d.qy &&
            d.rx == 2 && d.ry == 2 && d.area_rcp != 0.0f && ((r32x && d.nx == 2 && d.ny == 2) || (r21 && d.nx == 1 && d.ny == 1)))
            d.tap22 = r32x ? 1 : 2;
        if (d.tap22) {
            mode = M_BILINEAR;
            d.w_dyadic = 1;
            d.point_kind = PK_NONE;
            d.geo_pref = 0; // (the host-built geometry tables carry BILINEAR's weights)
        }
    }
}

// 2. thread-tile forms of the 2x2-tap kernel
static void sel_two_tap_forms(Mode mode, LaunchDesc &d) {
    d.bil_int = ((mode == M_BILINEAR || mode == M_AREA_UP) && d.w_dyadic && d.bil_int_pref) ? 1 : 0;
    // window form (one aligned 12-byte read per source row instead of byte reads): the four columns of a thread must span <= 8
    // bytes, i.e. horizontal ratio <= 2 (vpp_bilinear.hip); TSVPP_BILINEAR_INT=2 keeps the byte form
    if (d.bil_int && d.bil_int_pref != 2 && d.xr <= 2.0f) d.bil_int = 2;
    // float weights, same windows: measured +4..6 % at ratios 1.2 / 1.4 (1080p -> 1600x900, 1366x768), -2..5 % at 1.5 x 1.27 and 1.92,
    // even below 1 (profiles/r02_bilinear_winf_ab.txt) -- used between 1 and 1.45
    d.bil_win = (!d.bil_int && (mode == M_BILINEAR || mode == M_AREA_UP) && d.xr <= 2.0f &&
                 (d.bil_win_pref == 2 || (d.bil_win_pref == 1 && d.xr > 1.0f && d.xr <= 1.45f))) ? 1 : 0;
}

// 3. store policy
static void sel_store_policy(Mode mode, OutKind out, bool vec, LaunchDesc &d) {
    if (d.nt_stores < 0) { // per-kernel default
        // fp32 outputs: every store instruction of a wave covers whole 128-byte lines (planar: 16 contiguous
        // bytes per lane; merged: after the in-wave exchange of MergedRun) and nothing re-reads them.  The
        // scalar fallback of merged outputs (!vec) interleaves partial lines: plain stores, L2 combines them.
        const bool f32_lines = (out == O_F32_PLANAR || out == O_NV12_F32 || out == O_Y800_F32) || (vec && (out == O_F32_MERGED || out == O_HSV_F32));
        const bool f32_partial = !vec && (out == O_F32_MERGED || out == O_HSV_F32);
        d.nt_stores = f32_partial ? 0 : ((mode == M_NONE && f32_lines) ? 2 : 1);
        // The colour-only kernel's variant 2 was chosen on C2 (1080p planar fp32: 0.71 against 0.66 on variant 1) -- and holds for three-plane / merged outputs 1920 and 2048
        // columns wide only (round 6, profiles/r06_color_shapes.txt part 5): Y800 fp32 gains 6-22 % on variant 1 at every size (1080p 0.74 -> 0.79, 720p 0.70 -> 0.85, 4K 0.74 ->
        // 0.81), NV12 fp32 2-22 % (2048 columns: -4 %), planar / merged below 1536 columns 1-15 % (640 x 360 0.65 -> 0.75, 720p merged 0.70 -> 0.76).
        if (mode == M_NONE && f32_lines && d.nt_stores == 2 && (out == O_Y800_F32 || out == O_NV12_F32 || d.dst_w < 1536)) d.nt_stores = 1;
        // bit 2: the 4-byte uint8 stores non-temporal too -- only the colour-only kernel's planar output gains (st1o, vpp_device.h)
        // (round 6, profiles/r06_color_shapes.txt part 7: ... where the output rows are multiples of 64 bytes -- 720p 0.73 -> 0.755, 1080p 0.73 -> 0.77, 960 x 540 0.58 -> 0.68;
        // at 1360 / 1366 / 1918 / 854 columns the same bit LOSES 8-20 %: 1360 x 768 0.63 -> 0.51)
        if (mode == M_NONE && out == O_U8_PLANAR && vec && d.dst_w % 64 == 0) d.nt_stores = 1 | 4;
        // Planar fp32 rows that are no multiple of 64 bytes (round 6, profiles/r06_row_alignment.txt): a wave's 1 KiB row segment then starts and ends inside a line it shares
        // with the neighbouring workgroup's segment, and non-temporal stores send both halves of that line to memory on their own -- three planes at once.  Output width
        // 1360 (rows of 85 x 64 bytes) 0.71, 1376 (43 x 128) 0.74-0.75, but 1362 / 1364 / 1366 / 1370 0.58-0.60 and 1368 (171 x 32) 0.65.  Plain stores let L2 put the lines
        // together: 1366 columns BILINEAR 0.58 -> 0.68, AREA 0.46 -> 0.60, BICUBIC 0.43 -> 0.50, colour-only 0.43 -> 0.52, 854 columns 0.585 -> 0.67, 1368 +4...7 %.  They
        // LOSE 3-8 % on rows of whole lines, on one- / two-plane outputs (Y800, NV12) and on merged ones, and do nothing for small frames (300 / 600 columns): those stay.
        if (out == O_F32_PLANAR && vec && ((size_t)d.dst_w * 4) % 64 != 0 && d.dst_w >= 800) d.nt_stores = 0;
        // ... and the colour-only kernel with EVERY fp32 flavour (profiles/r06_row_alignment.txt, part 6): 1366 x 768 merged 0.44 -> 0.55, HSV 0.43 -> 0.53, NV12 0.46 -> 0.57,
        // Y800 0.45 -> 0.56 (behind a resize those flavours tie or lose with plain stores: the LDS kernels keep their non-temporal ones)
        if (mode == M_NONE && vec && (out == O_F32_MERGED || out == O_HSV_F32 || out == O_NV12_F32 || out == O_Y800_F32) && d.dst_w >= 800 &&
            ((size_t)d.dst_w * 4 * ((out == O_F32_MERGED || out == O_HSV_F32) ? 3 : 1)) % 64 != 0)
            d.nt_stores = 0;
    }
}

// 4. candidate workgroup shapes
static void sel_shapes(Mode mode, OutKind out, LaunchDesc &d, FusedSel &S) {
    int (&shapes)[5][2] = S.shapes;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    // Candidate workgroup shapes, largest first; the staged kernels take the first whose source
    // footprint fits the LDS budget (several workgroups per CU must stay resident to overlap one
    // group's loads with another's arithmetic).
    // The 2x2-tap kernel with fp32 outputs runs AT the HBM floor of its tile pattern, and that floor depends on the tile
    // shape (tools/membench2.hip, the headline's 22 % read / 78 % write mix with no arithmetic, 64 frames, rotating buffers):
    // 128 x 32 tiles 165 us, 128 x 16 159 us, 256 x 8 153 us (1 KiB row segments per plane, half the in-flight footprint),
    // whole rows 170 us.  The kernel follows: 1080p -> 720p 164.7 us on 128 x 32 tiles (0.688 of 8 TB/s), 151.6 us on
    // 256 x 8 (0.748) -- shape sweep in profiles/r02_shape_sweep.txt.  So: 256-wide, 8-row tiles wherever the output width
    // fills them (a half-empty last tile column costs more than the pattern gains: 1920-wide outputs stay on 128).
    // (the point samplers write the same pattern: NEAREST 1080p -> 720p 0.770 -> 0.795 on 64 x 4, profiles/r04_shape_sweep_misc.txt)
    if ((two_tap || (d.point_kind != PK_NONE && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC))) && f32_out && d.dst_w % 256 == 0) {
        shapes[3][0] = shapes[2][0]; shapes[3][1] = shapes[2][1];
        shapes[2][0] = shapes[1][0]; shapes[2][1] = shapes[1][1];
        shapes[1][0] = shapes[0][0]; shapes[1][1] = shapes[0][1];
        shapes[0][0] = 64; shapes[0][1] = 4;
    }
    // The colour-only kernel (no resize) with planar fp32 outputs writes the same three-plane pattern (round 6, profiles/r06_color_shapes.txt): at output widths that are
    // multiples of 256 its 32 x 8 workgroups (128 x 16 pixels) reach 0.62-0.64 of the roofline at 1280 / 2560 / 3840 columns against 0.73-0.74 at 1920 -- and 0.72-0.74 on
    // 64 x 4 (256 x 8 pixels: 1 KiB row segments per plane); 2048 columns 0.73 -> 0.76.  1024 and 640 columns do not care (-1.5 % / 0), uint8 planar loses at 1280 (-6 %).
    if (mode == M_NONE && out == O_F32_PLANAR && d.dst_w % 256 == 0 && d.dst_w >= 1280) {
        shapes[3][0] = shapes[2][0]; shapes[3][1] = shapes[2][1];
        shapes[2][0] = shapes[1][0]; shapes[2][1] = shapes[1][1];
        shapes[1][0] = shapes[0][0]; shapes[1][1] = shapes[0][1];
        shapes[0][0] = 64; shapes[0][1] = 4;
    }
    // uint8 outputs on the integer window tile read host-built geometry tables (vpp_bilinear.hip): with workgroups 64 thread tiles
    // wide a wave's lanes share their output rows and the row records are scalar loads -- measured (profiles/r02_geo_ab.txt)
    // 1080p -> 720p planar 0.545 -> 0.570, merged 0.481 -> 0.506, 4K -> 1080p 0.706 -> 0.717 against 32 x 8
    if (two_tap && !f32_out && d.bil_int == 2 && d.geo_pref && d.dma && (d.pitch_y & 15) == 0 && (d.pitch_uv & 15) == 0 && d.dst_w >= 256) {
        shapes[3][0] = shapes[2][0]; shapes[3][1] = shapes[2][1];
        shapes[2][0] = shapes[1][0]; shapes[2][1] = shapes[1][1];
        shapes[1][0] = shapes[0][0]; shapes[1][1] = shapes[0][1];
        shapes[0][0] = 64; shapes[0][1] = 4;
    }
    if (d.shape_tx > 0 && d.shape_ty > 0 && (d.shape_tx & (d.shape_tx - 1)) == 0 && d.shape_tx * d.shape_ty <= MAX_THREADS &&
        d.shape_tx * d.shape_ty >= 64) {
        shapes[0][0] = d.shape_tx;
        shapes[0][1] = d.shape_ty;
    }
}

// 5. AREA down-scale: which of the un-staged samplers (direct / box / streaming / column-per-lane), if any
static void sel_area(Mode mode, bool vec, LaunchDesc &d, FusedSel &S) {
    size_t &as_lds = S.as_lds;
    if (mode == M_AREA_DOWN && d.qx && d.qy && vec && !d.force_gather && d.area_direct_min > 0.0f && d.xr >= d.area_direct_min &&
        d.yr >= d.area_direct_min)
        d.area_direct = 1;
    else if (mode == M_AREA_DOWN && !(d.qx && d.qy) && vec && !d.force_gather && d.area_direct_fmin > 0.0f && d.xr >= d.area_direct_fmin &&
             d.yr >= d.area_direct_fmin && d.nkx >= 1 && d.nkx <= 8 && d.patx4 && d.paty4)
        d.area_direct = 2; // float weights
    else
        d.area_direct = 0;
    // integer horizontal ratio (one all-ones weight row), 4-byte aligned planes and pitches: the box kernel's contiguous runs
    d.area_box = (d.area_direct == 1 && d.area_box_pref && d.box_rx >= 4 && d.box_rx <= 8 && d.box_rx == d.rx && d.in_aligned4 && d.ry <= 8) ? 1 : 0;
    // integer horizontal ratios 2 and 3 (1080p -> 960x540, 1080p -> 640x360, 4K -> 720p): below area_direct_min these went to the LDS
    // kernel (measured against the round-1 direct kernel); the box kernel wins there too (profiles/r02_area_box23_ab.txt: 1080p -> 640x360
    // fp32 0.602 -> 0.715, uint8 0.492 -> 0.666, 4K -> 720p uint8 merged 0.536 -> 0.711, 1080p -> 960x540 fp32 0.673 -> 0.707).
    // TSVPP_AREA_BOX=4 keeps it to ratios >= 4.
    if (!d.area_box && d.area_box_pref && d.area_box_pref != 4 && mode == M_AREA_DOWN && d.qx && d.qy && vec && !d.force_gather && (d.box_rx == 2 || d.box_rx == 3) &&
        d.box_rx == d.rx && d.in_aligned4 && d.ry <= 8 && d.yr >= 2.0f) {
        d.area_direct = 1;
        d.area_box = 1;
    }
    // Float-weight AREA: one wave per 128-column tile, the source rows streamed through a wave-private ring (vpp_area_stream.hip).  Needs the
    // host-built divisor table, pitches that are multiples of 16 (LDS-DMA chunks), a row segment of at most 128 chunks (ratio <= ~15).
    d.area_stream = 0;
    // TSVPP_AREA_STREAM: 1 = from `as_min_taps` taps per value on (measured cross-over, profiles/r03_area_stream_ab*.txt), 2 = wherever it applies
    if (mode == M_AREA_DOWN && !(d.qx && d.qy) && vec && !d.force_gather && (d.area_stream_pref == 2 ||
         (d.area_stream_pref == 1 && (d.rx * d.ry >= d.as_min_taps || d.nkx > 3 || (d.area_direct != 2 && !(d.rx <= 3 && d.ry <= 3 && d.area2_pref))))) && d.area_div && d.patx4 && d.paty4 && d.nkx >= 1 && d.nkx <= 8 &&
        (d.pitch_y & 15) == 0 && (d.pitch_uv & 15) == 0) {
        const int nk = d.nkx <= 4 ? d.nkx : (d.nkx <= 6 ? 6 : 8);
        auto rowb_of = [&](int cols) {
            const int seg_y = (int)((double)d.xr * (cols - 1)) + 1 + 4 * nk, seg_uv = 2 * ((int)((double)d.xr * (cols / 2 - 1)) + 1) + 8 * nk;
            return ((seg_y > seg_uv ? seg_y : seg_uv) + 15 + 15) & ~15;
        };
        const int two = rowb_of(128) <= 2048 ? 1 : 0; // a row segment = at most two DMA instructions (128 chunks)
        const int rowb = rowb_of(two ? 128 : 64);
        const int nkmin = nk == 6 ? 5 : (nk == 8 ? 7 : nk);
        // (the kernel skips the multiply of column taps 1 .. 4 * nkmin - 5: they weigh 1.0f for every ratio -- checked against the table itself)
        if (rowb <= 2048 && d.as_ones_x >= 4 * nkmin - 4) {
            // tile height 4 (measured: 8 rows never win -- 4K -> 608x342 0.618 against 0.572, 1080p -> 160^2 0.59 against 0.47; TSVPP_AREA_STREAM_ROWS=8)
            int r = 4;
            if (d.as_rows == 4 || d.as_rows == 8) r = d.as_rows;
            if (!two) r = 8;
            d.area_stream = 1;
            d.as_nk = nk;
            d.as_two = two;
            d.bc_ring_bytes = AS_RING_ROWS * rowb + 16;
            d.bc_wave_bytes = d.bc_ring_bytes + 128 * (r + r / 2);
            as_lds = 4 * (size_t)d.bc_wave_bytes;
            d.area_direct = 0;
            d.tx = two ? 32 : 16; // colour phase: a wave = 32 x 2 thread tiles per 4-row slab, or 16 x 4 per 8-row slab (MergedRun: runs of 32 / 16 lanes)
            d.ty = two ? 2 : 4;
            d.rpt = two ? r / 4 : 1;
        }
    }
    // Below the streaming kernel's cross-over (fewer than 40 taps per value; <= 12 horizontal taps): measured in round 2
    // (profiles/r02_area_cols_ab.txt; TSVPP_AREA_COLS=0/1/2, TSVPP_AREA_COLS_ROWS=8/32) the column-per-lane kernel wins from 5 horizontal
    // taps on -- 1080p -> 300^2 +18 %, -> 416^2 +21 % -- and is even at 2-4 taps
    if (d.area_direct == 2 && d.nkx > 3) d.area_direct = 0; // (13+ taps without the streaming kernel: generic path)
    d.area_cols = (!d.area_stream && d.area_direct == 2 && (d.area_cols_pref == 2 || (d.area_cols_pref == 1 && d.nkx >= 2))) ? 1 : 0;
    if (d.area_cols_rows != 8 && d.area_cols_rows != 32) {
        d.area_cols_rows = d.nkx >= 3 ? 8 : 32;
        // Small outputs: 64 x 32-pixel tiles leave too few workgroups -- 1080p -> 300 x 300 is 5 x 10 tiles a frame, 3200 workgroups per 64-frame launch =
        // 1.6 rounds of the GPU, 12 % of the tile area past the frame's edges: 8-row tiles (12160 workgroups) from 16 workgroups per CU down
        // (profiles/r04_area_cols_rows_ab.txt; TSVPP_AREA_COLS_ROWS=8|32 forces)
        if (d.area_cols_rows == 32 && (long)((d.dst_w + 63) / 64) * ((d.dst_h + 31) / 32) * d.n_frames < 16L * d.num_cus) d.area_cols_rows = 8;
    }
    if (d.area_cols) { // fixed workgroup of 256 threads; tile = 16 x (rows / 2) thread tiles = 64 columns x 32 or 8 rows
        d.tx = 16;
        d.ty = d.area_cols_rows / 2;
    }
}

// 6. point samplers (NEAREST; BILINEAR / BICUBIC whose weights are all zero): one LDS row per output row
static void sel_point(Mode mode, bool vec, LaunchDesc &d, FusedSel &S) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool point = d.point_kind != PK_NONE && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC);
    if (!point) d.point_kind = PK_NONE;
    if (point && vec && !d.force_gather) {
        for (auto &sh : shapes) {
            if (sh[0] == 0) break;
            const int tw = sh[0] * PXW, th = sh[1] * PXH, nthreads = sh[0] * sh[1];
            const int span_y = (int)((double)d.xr * (tw - 1)) + 3, span_uv = 2 * ((int)((double)d.xr * (tw / 2 - 1)) + 3);
            const int cpr_y = (span_y + 15 + 15) / 16, cpr_uv = (span_uv + 15 + 15) / 16;
            if (cpr_y > nthreads || cpr_uv > nthreads) continue;
            const size_t need = (size_t)16 * ((size_t)th * cpr_y + (size_t)(th / 2) * cpr_uv) + sizeof(int) * (size_t)(tw + tw / 2 + th + th / 2);
            if (need <= kLdsBudget) {
                staged = true;
                lds_bytes = need;
                d.tx = sh[0];
                d.ty = sh[1];
                d.lds_span_y = span_y;
                d.lds_cpr_y = cpr_y;
                d.lds_slot_y = slot_shift_for(cpr_y);
                d.lds_span_uv = span_uv;
                d.lds_cpr_uv = cpr_uv;
                d.lds_slot_uv = slot_shift_for(cpr_uv);
                break;
            }
        }
        if (!staged) d.point_kind = PK_NONE; // footprint too large: generic paths below
    } else {
        d.point_kind = PK_NONE;
    }
}

// 6b. BILINEAR at sparse ratios -- and, round 6, the pure point samplers at sparse ratios: the tapped rows as LDS-DMA row segments, one wave per 64-column tile
// (vpp_bilinear_rows.hip).  `point`: the request is a point sampler (NEAREST, or BILINEAR / BICUBIC with all-zero weights; the caller asks BEFORE sel_point).
static void sel_bilinear_rows(Mode mode, bool vec, bool sparse_gather, int stream_r32, bool point, LaunchDesc &d, FusedSel &S) {
    // Until round 4 every BILINEAR request with a ratio product >= 12 ran on the byte-gather kernel (BASELINE config C3: 5.0 x 2.8125).  That kernel is bound by the
    // issue of its gathers -- 64 lanes = 64 cache lines per load instruction -- not by the launch ramp: 64 -> 128 frames took C3 from 23.2 to 50.1 us
    // (profiles/r04_batch128_ab.txt).  Here the tapped rows arrive as contiguous 16-byte chunks.  Needs pitches that are multiples of 16 (every row of a plane then has
    // the same misalignment) and a row segment of at most 64 chunks = ONE DMA instruction (horizontal ratios up to ~15.7).  TSVPP_BILINEAR_ROWS=2 takes every
    // BILINEAR / point request that satisfies those two (tests: ratios the LDS-staged kernels serve by default), 0 keeps the gathers / the LDS point kernel, 3 = 1 for BILINEAR
    // only (the point samplers stay on vpp_point_kernel: A/B).
    if (d.bil_rows) return; // (already chosen for a point request)
    d.bil_rows = 0;
    if (point) {
        if (!(mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC) || d.bil_rows_pref == 3) return;
    } else if (mode != M_BILINEAR || d.tap22 || S.staged) {
        return;
    }
    if (!vec || d.force_gather || !d.bil_rows_pref || stream_r32) return;
    if (!(sparse_gather || d.bil_rows_pref == 2)) return;
    if ((d.pitch_y & 15) != 0 || (d.pitch_uv & 15) != 0) return;
    // bytes a tile's 64 luma columns / 32 chroma pair columns span (+ the right-hand tap, + 1 for the float coordinate), + up to 15 bytes of misalignment, in chunks
    const int span_y = (int)(63.0 * (double)d.xr) + 3, span_c = 2 * ((int)(31.0 * (double)d.xr) + 3);
    const int L = ((span_y > span_c ? span_y : span_c) + 30) / 16;
    // (round 6: up to 128 chunks -- two instructions per segment, horizontal ratios up to ~31: 4K -> 224 x 224 -- instead of the byte-gather kernel; TSVPP_BILINEAR_ROWS_WAVES=9 keeps the
    // one-instruction limit for an A/B)
    if (L > 128 || (L > 64 && d.br_waves == 9)) return;
    const int rpt = (d.rpt_pref >= 1 && d.rpt_pref <= 4) ? d.rpt_pref : 1; // tile height 8 rows: the most waves in flight (TSVPP_RPT: 16 / 24 / 32)
    const int wave_bytes = (point ? 12 : 24) * rpt * 16 * L + 64;          // 2 (point: 1) segments per luma row, and per chroma row
    // waves (64-column tiles) per workgroup: the largest of 4 / 2 / 1 that launches no more waves than the narrowest choice (300 columns: five single-wave workgroups
    // instead of two four-wave ones, three of whose waves would exit at once) within 48 KiB of LDS
    int nw = 0;
    long best = 0;
    for (int w = 4; w >= 1; w >>= 1) {
        if ((long)w * wave_bytes > 48 * 1024) continue;
        if (d.br_waves && d.br_waves != 9 && d.br_waves != w) continue; // TSVPP_BILINEAR_ROWS_WAVES (A/B)
        const long waves = (long)((d.dst_w + 64 * w - 1) / (64 * w)) * w;
        if (!nw || waves < best) {
            nw = w;
            best = waves;
        }
    }
    if (!nw) return;
    d.bil_rows = L;
    d.br_rpi = 64 / L; // (0: two instructions per segment)
    d.br_waves = nw;
    d.bc_wave_bytes = wave_bytes;
    S.br_lds = (size_t)nw * wave_bytes;
    d.tx = 16; // colour phase: a wave = 16 x 4 thread tiles per 8-row slab (MergedRun: runs of 16 lanes)
    d.ty = 4;
    d.rpt = rpt;
}

// 7. the LDS-staged kernels (2x2-tap, integer BICUBIC, dyadic / small float AREA): first workgroup shape, rows per thread and staging layout that fit
static void sel_staged(Mode mode, bool vec, bool bicubic_staged, bool sparse_gather, int stream_r32, LaunchDesc &d, FusedSel &S) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    auto workgroups = [&](const int *sh, int rpt) { return fused_workgroups(d, sh, rpt); };
    // (a streaming kernel has been selected: nothing is staged -- ADVICE r05)
    if (!staged && !stream_r32 && mode != M_NONE && vec && !d.force_gather && !d.area_direct && !sparse_gather && !d.area_stream && !d.bil_rows) {
        const int want_dma = d.dma;
        for (auto &sh : shapes) {
            if (sh[0] == 0 || staged) break;
            const bool bint = bicubic_staged; // dyadic weights: integer kernel
            const bool area2 = mode == M_AREA_DOWN && !(d.qx && d.qy) && d.rx >= 2 && d.rx <= 3 && d.ry >= 2 && d.ry <= 3 && d.area2_pref;
            const bool dyadic = mode == M_AREA_DOWN && d.qx && d.qy;
            if ((mode == M_AREA_DOWN && !dyadic && !area2) || mode == M_NEAREST) break; // (no staged kernel: streaming kernel above, or gathers)
            // Row pairs per thread (TSVPP_RPT; 0 = per kernel).  Taller thread tiles amortise the tile decode, staging set-up and
            // table build over more pixels -- that pays where the kernel is VALU-bound (uint8 outputs, separable BICUBIC, the
            // AREA kernels: two row pairs) -- but the fp32 2x2-tap kernel is bound by the HBM write pattern, which prefers
            // SHORT tiles (round 2 sweep: one row pair wins by 2..9 % on 1080p -> 720p, 4K -> 1080p and 720p -> 1080p).
            // ... except its luma-only flavour (Y800 fp32, round 6): a third of the output bytes per tile for the same staging set-up -- two row pairs per thread 0.565 -> 0.740 of the
            // roofline (1080p -> 720p, profiles/r06_fmt_ab.txt; NV12 fp32 is indifferent: 0.731 / 0.734)
            const int rpt_auto = (two_tap && f32_out && !d.luma_only) ? 1 : 2;
            const int rpt_want = d.rpt_pref >= 1 && d.rpt_pref <= 8 ? d.rpt_pref : rpt_auto;
            int rpt_max = (two_tap || bint || area2 || dyadic) ? rpt_want : 1;
            // taller thread tiles only while the launch still has at least two full rounds of workgroups
            // (8 per CU): small outputs (C3: 256x256) need the parallelism more than the amortisation
            // (the 2x2-tap kernel wants six rounds)
            const long rounds = two_tap ? 48L : 16L;
            while (rpt_max > 1 && workgroups(sh, rpt_max) < rounds * d.num_cus) rpt_max--;
            // One (rows per thread, staging layout) candidate of this shape: its LDS need and descriptor fields.
            struct Cand { bool ok; size_t need; int rpt, dma, span_y, rows_y, cpr_y, span_uv, rows_uv, cpr_uv, hcs_y, hcs_uv; };
            auto candidate = [&](int rpt, int layout) {
                Cand c = {};
                c.rpt = rpt;
                c.span_y = span_bound(mode, sh[0] * PXW, d.xr, d.rx);
                const int rows_y = span_bound(mode, sh[1] * PXH * rpt, d.yr, d.ry);
                c.span_uv = 2 * span_bound(mode, sh[0] * PXW / 2, d.xr, d.rx);
                const int rows_uv = span_bound(mode, sh[1] * PXH * rpt / 2, d.yr, d.ry);
                const int nthreads = sh[0] * sh[1];
                c.cpr_y = (c.span_y + 15 + 15) / 16;
                c.cpr_uv = (c.span_uv + 15 + 15) / 16;
                if (c.cpr_y > nthreads || c.cpr_uv > nthreads) return c;
                c.rows_y = rows_y;
                c.rows_uv = rows_uv;
                c.dma = (layout == 1 && nthreads >= 64) ? 1 : 0;
                if (layout == 1 && !c.dma) return c;
                if (c.dma) { // a wave instruction fills 64 consecutive chunk slots: the plane is allocated up to a multiple of 64 slots
                    c.rows_y = ((rows_y * c.cpr_y + 63) / 64 * 64 + c.cpr_y - 1) / c.cpr_y;
                    c.rows_uv = ((rows_uv * c.cpr_uv + 63) / 64 * 64 + c.cpr_uv - 1) / c.cpr_uv;
                }
                const size_t cols = (size_t)sh[0] * PXW, rows = (size_t)sh[1] * PXH * rpt;
                size_t need = (size_t)16 * ((size_t)c.rows_y * c.cpr_y + (size_t)c.rows_uv * c.cpr_uv);
                if (dyadic) // tables + row bases + slack for the dword over-read
                    need += area_dyadic_table_bytes(cols, rows) + sizeof(int) * (size_t)(c.rows_y + c.rows_uv) + 32;
                if (area2) // column / row tables and row bases of the float AREA kernel
                    need += areaf_table_bytes(cols, rows) + sizeof(int) * (size_t)(c.rows_y + c.rows_uv);
                if (mode == M_BILINEAR || mode == M_AREA_UP) // coordinate tables
                    need += (cols + cols / 2) * sizeof(XEntry) + (rows + rows / 2) * sizeof(YEntry);
                if (bint) { // column-major H planes (column stride: an odd number of dwords >= rows + 8 bytes), tables, row bases
                    c.hcs_y = 4 * (((rows_y + 3) / 4 + 2) | 1);
                    c.hcs_uv = 4 * (((rows_uv + 3) / 4 + 2) | 1);
                    need += bicubic_int_table_bytes((int)cols, (int)rows, c.rows_y, c.rows_uv, c.hcs_y, c.hcs_uv);
                }
                c.need = need;
                c.ok = need <= kLdsBudget;
                return c;
            };
            Cand best = {};
            if (dyadic) {
                // the integer AREA kernel is sensitive to how many workgroups a CU holds (160 KiB of LDS): 1080p -> 960x540
                // runs 560 k frames/s on the compact two-row layout (19.7 KiB, 8 workgroups), 523 k on the compact
                // four-row one (34 KiB) and 494 k on the two-row LDS-DMA one (37.5 KiB).  Most resident workgroups
                // (up to five) wins; ties go to the taller tile, then to LDS-DMA.
                long best_key = -1;
                for (int rpt = rpt_max; rpt >= 1; rpt--)
                    for (int layout = want_dma ? 1 : 0; layout >= 0; layout--) {
                        const Cand c = candidate(rpt, layout);
                        if (!c.ok) continue;
                        long occ = (long)(160 * 1024 / c.need);
                        if (occ > 5) occ = 5; // beyond five resident workgroups the layout matters more (1080p -> 1536x864:
                                              // four-row LDS-DMA tiles at 6 per CU, 283 k, vs compact ones at 8, 268 k)
                        const long key = occ * 100 + rpt * 10 + c.dma;
                        if (key > best_key) {
                            best_key = key;
                            best = c;
                        }
                    }
            } else {
                // first fit: the LDS-DMA layout, then the compact register-staged one; the separable BICUBIC kernel tries
                // a shorter tile of the SAME workgroup shape before a smaller workgroup (1080p -> 640x640: 375 k vs 288 k)
                for (int rpt = rpt_max; rpt >= 1 && !best.ok; rpt = bint ? rpt - 1 : 0)
                    for (int layout = want_dma ? 1 : 0; layout >= 0 && !best.ok; layout--) best = candidate(rpt, layout);
            }
            if (best.ok) {
                staged = true;
                lds_bytes = best.need;
                d.tx = sh[0];
                d.ty = sh[1];
                d.rpt = best.rpt;
                d.lds_span_y = best.span_y;
                d.lds_rows_y = best.rows_y;
                d.lds_cpr_y = best.cpr_y;
                d.lds_slot_y = slot_shift_for(best.cpr_y);
                d.lds_span_uv = best.span_uv;
                d.lds_rows_uv = best.rows_uv;
                d.lds_cpr_uv = best.cpr_uv;
                d.lds_slot_uv = slot_shift_for(best.cpr_uv);
                d.lds_magic_y = 0xFFFFFFFFu / (uint32_t)best.cpr_y + 1u;
                d.lds_magic_uv = 0xFFFFFFFFu / (uint32_t)best.cpr_uv + 1u;
                d.dma = best.dma;
                d.bicubic_int = bint ? 1 : 0;
                d.hcs_y = best.hcs_y;
                d.hcs_uv = best.hcs_uv;
                d.area2 = area2 ? 1 : 0;
            }
        }
    }
}

// 8. the wave-per-tile BICUBIC kernel
static void sel_bicubic_cols(Mode mode, OutKind out, bool vec, int stream_r32, LaunchDesc &d, FusedSel &S, hipStream_t stream, LaunchInfo *info) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    // BICUBIC that the integer kernel above did not take (non-dyadic weights -- or TSVPP_BICUBIC_COLS=2: every request): one wave per
    // 64-column tile, one lane per output column, H sums in a wave-private column-major LDS plane (vpp_bicubic_cols.hip).  A taller
    // tile re-evaluates fewer H rows at its seams (3 / (R yr) of them), a shorter one keeps more waves in flight.
    // (not when ANY streaming kernel took the request -- the BICUBIC ones at 3 : 2 / 2 : 1, or a point sampler at an integer ratio: its tables would be built,
    // uploaded and cached for nothing)
    if (mode == M_BICUBIC && !staged && vec && !d.force_gather && d.bicubic_cols_pref && !stream_r32 && !d.bil_rows) { // (bil_rows: a zero-weight request on the row-segment kernel)
        const bool sparse = d.yr >= 4.0f;
        const bool exact = d.w_dyadic != 0; // every weight a multiple of 1/16: the quantised coefficients are exact, no tie test
        // LDS-DMA ring: a row segment is (64 columns at ratio xr + window + a misalignment of up to 15 bytes) rounded up to 16-byte chunks,
        // at most 16 of them (horizontal ratios up to 3.68); bc_dma = chunks (lanes) per row; the kernel takes that path when the pitch is
        // a multiple of 16
        const int seg_bytes = (int)((double)d.xr * 63.0) + 2 + 7 + 15 + 1;
        // ... round 5: dense rows up to 32 chunks (ratios up to 7.4: 1080p -> 300 x 300, 4K -> 640 x 360) as two instructions per group of four rows ("wide", vpp_bicubic_cols.hip);
        // TSVPP_BICUBIC_DMA=3 keeps the 16-chunk limit (per-lane loads beyond it: rounds 3-4)
        // fp32 outputs only: 1080p -> 300 x 300 0.441 -> 0.471, -> 416 x 416 0.515 -> 0.558; uint8 merged LOSES 13 % (0.434 -> 0.378: the larger ring costs the occupancy its fewer
        // bytes need; profiles/r05_bicubic_wide_ab.txt)
        const bool dma = d.bc_dma_pref && (seg_bytes <= 256 || (!sparse && S.f32_out && seg_bytes <= 512 && d.bc_dma_pref != 3 && d.bc_dma_pref != 2));
        const int dma_lanes = (seg_bytes + 15) / 16 < 2 ? 2 : (seg_bytes + 15) / 16;
        const int ring_bytes = dma ? (dma_lanes > 16 ? 3 * 64 * dma_lanes + 16 : 3 * 1024 + 16) : 0;
        auto col_stride = [&](int nout) { // bytes of one H-plane column: its dwords + one (phase 2 reads dword pairs), an odd number of them
            const int rows = sparse ? 4 * nout : (int)((double)d.yr * (nout - 1)) + 6;
            return 4 * ((((rows + 3) >> 2) + 1) | 1);
        };
        auto wave_bytes_of = [&](int r) { return ring_bytes + 64 * (col_stride(r) + r + r / 2); };
        // the uint8 merged flavour's output side (r32_store_tile, vpp_r32_store.h) carries a STATIC exchange slab of MAX_THREADS x 24 bytes on top of the dynamic LDS,
        // whether or not bc_u8x is chosen at run time: part of the workgroup's LDS for both budgets below and for info->lds_bytes
        const int static_lds = out == O_U8_MERGED ? MAX_THREADS * 24 : 0;
        // Tile height, measured (profiles/r03_bicubic_cols_ab*.txt): 32 rows for up-scales (few source rows per tile: the seams cost
        // most there; 720p -> 1080p 0.519 against 0.502 at 16), 16 rows while the launch still has 8 waves per SIMD (1080p -> 640^2 0.620
        // against 0.609 at 8, 4K -> 1080p 0.651 against 0.624), else 8 (1080p -> 224^2 0.748 against 0.642, -> 300^2 0.404 against 0.367)
        int best_r = 0;
        {
            auto waves_of = [&](int r) { return (long)((d.dst_w + 63) / 64) * ((d.dst_h + r - 1) / r) * d.n_frames; };
            int r = d.yr <= 1.0f ? 32 : 16;
            while (r > 8 && (waves_of(r) < 32L * d.num_cus || 4 * wave_bytes_of(r) + static_lds > 40 * 1024)) r -= 8;
            if (d.bc_rows >= 8 && d.bc_rows <= 32 && (d.bc_rows & 7) == 0) r = d.bc_rows;
            if (4 * wave_bytes_of(r) + static_lds <= 64 * 1024) best_r = r;
        }
        // the request's column / row tables (host-built, cached in the context): a real launch -- and the dry run of
        // tsvpp_prepare_batch -- looks them up or builds them; while the stream is capturing and they do not exist yet the
        // request takes the generic path below
        if (best_r && (!info || d.geo_build)) {
            d.bc_tab = bicubic_cols_tables(d, stream, true);
            if (!d.bc_tab) best_r = 0;
        }
        if (best_r) {
            d.bicubic_cols = exact ? 2 : 1;
            // uint8 flavours: the 8 x 4 output side of the streaming kernels -- only where its 64 lanes have work (32-row tiles: up-scales, +2..5 %); 8- / 16-row
            // tiles leave 48 / 32 of them idle and lose 3..5 % (profiles/r05_bicubic_cols_u8_ab.txt).  TSVPP_BICUBIC_U8X=2: every tile height, 0: never.
            d.bc_u8x = (d.bc_u8x_pref && (d.dst_w & 7) == 0 && (d.dst_h & 3) == 0 && (best_r == 32 || d.bc_u8x_pref == 2)) ? 1 : 0;
            d.bc_sparse = sparse ? 1 : 0;
            d.bc_dma = dma ? (d.bc_dma_pref == 2 ? 16 : dma_lanes) : 0; // TSVPP_BICUBIC_DMA=2: 256-byte segments whatever the ratio (A/B)
            d.bc_ring_bytes = ring_bytes;
            d.bc_npy = bicubic_cols_rows_padded(d.dst_h);
            d.bc_npc = bicubic_cols_rows_padded(d.dst_h >> 1);
            d.hcs_y = col_stride(best_r);
            d.hcs_uv = d.hcs_y;
            d.bc_wave_bytes = wave_bytes_of(best_r);
            lds_bytes = 4 * (size_t)d.bc_wave_bytes;
            d.tx = 16; // colour phase: a wave = 16 x 4 thread tiles per 8-row slab (MergedRun: runs of 16 lanes)
            d.ty = 4;
            d.rpt = best_r / 8;
            d.dma = 0;
        }
    }
}

hipError_t launch_fused(Mode mode, OutKind out, bool vec, const LaunchDesc &din, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    LaunchDesc d = din;
    d.rpt = 1;
    d.bicubic_int = 0;
    FusedSel S;
    const Mode mode_in = mode;
    sel_tap22(mode, out, vec, d);
    sel_two_tap_forms(mode, d);
    d.luma_only = (out == O_Y800_U8 || out == O_Y800_F32) ? 1 : 0;
    sel_store_policy(mode, out, vec, d);
    S.f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    S.two_tap = (mode == M_BILINEAR || mode == M_AREA_UP);
    S.lds_budget = (size_t)(d.lds_budget_kb > 0 ? d.lds_budget_kb : 40) * 1024; // TSVPP_LDS_KB
    sel_shapes(mode, out, d, S);
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    d.tx = S.shapes[0][0];
    d.ty = S.shapes[0][1];
    sel_area(mode, vec, d, S);
    // the streaming kernels at the exact ratios 3 : 2 / 2 : 1 and the streaming point samplers (stream_select above); a request that takes one needs neither staging nor tables
    const int stream_r32 = stream_select(mode, out, vec, d, din.point_kind);
    // Point samplers at sparse ratios: the row-segment kernel (round 6) instead of vpp_point_kernel's one LDS row per output row behind a workgroup barrier.  Same-box A/B
    // (profiles/r06_point_rows_ab.txt, NEAREST -> RGB24 planar fp32, launch time at 512 / 256 frames per launch): 1080p -> 224^2 (ratio product 41) 158 -> 112 us,
    // 4K -> 256^2 137 -> 104, 4K -> 300^2 190 -> 162, 4K -> 416^2 (48) 282 -> 198; even at 1080p -> 256^2 (32) and 4K -> 640^2 (20); 1080p -> 300^2 (23) LOSES 23 %
    // (206 -> 253 us), 1080p -> 640^2 6 %: the LDS kernel keeps everything below kPointRowsMin.
    d.bil_rows = 0;
    {
        const bool point_req = din.point_kind != PK_NONE && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC);
        constexpr float kPointRowsMin = 36.0f;
        if (point_req) sel_bilinear_rows(mode, vec, d.xr * d.yr >= kPointRowsMin, stream_r32, true, d, S);
    }
    if (!d.bil_rows) sel_point(mode, vec, d, S);
    d.bicubic_cols = 0;
    d.bc_sparse = 0;
    d.bc_dma = 0;
    // Interpolating kernels at large down-scale ratios tap only a few bytes of each source line: staging the whole
    // footprint through LDS then moves (and waits for) mostly unused bytes with few waves in flight, while plain
    // gathers touch each needed line once with full occupancy.  Measured cross-over (tools/matrix.sh, 1080p ->
    // 224^2 / 300^2 / 640^2, 4K -> 640x360; C3: 720p crop -> 256^2 = 14, gathers +8 %): BILINEAR gathers win from
    // xr*yr ~ 12 (3.5x at 41), BICUBIC from ~ 30.
    const float ratio_area = d.xr * d.yr;
    const int bc_r32 = (stream_r32 == 7 || stream_r32 == 8) ? stream_r32 : 0;
    // (BICUBIC: only the integer kernel for dyadic weights stages; everything else is vpp_bicubic_cols.hip)
    // (round 6, profiles/r06_bicubic_int_vs_cols.txt: the cross-over is a ratio product of ~6.5, not 30 -- the integer kernel stages and H-filters every source row of its
    // footprint, the column kernel's EXACT instance walks the same dyadic weights without the tie test: 4K -> 960 x 540 (4 x 4) 385 / 472 us per 64 frames (planar fp32 /
    // merged uint8) against 241 / 214, 4K -> 1024 x 576 287 / 342 against 219 / 222, 1080p -> 640 x 480 (3 x 2.25) 97 / 103 against 89 / 78; a tie at 2.5 x 2.5; below, and for
    // every up-scale, the integer kernel wins by 2-40 %)
    const bool bicubic_staged = mode == M_BICUBIC && d.w_dyadic && d.bicubic_int_pref && d.bicubic_cols_pref != 2 && ratio_area < 6.5f && !bc_r32;
    // (round 6: BILINEAR leaves the LDS-staged kernel for the row-segment kernel already from a ratio product of 7.5 when the vertical ratio is at least 2.1 -- the staged kernel
    // fetches every source row of its footprint, the row-segment kernel two per output row: 1080p -> 416^2 (4.6 x 2.6) 561 -> 373 us per 512 frames, -> 480^2 (4 x 2.25) 507 -> 469,
    // 4K -> 1024^2 (3.75 x 2.1) 285 -> 269; at a vertical ratio of 2.0 and below the staged kernel wins: 1080p -> 540^2 731 vs 756, -> 800^2 912 vs 1079; profiles/r06_point_rows_ab.txt)
    const bool sparse_gather = (mode == M_BILINEAR && (ratio_area >= 12.0f || (ratio_area >= 7.5f && d.yr >= 2.1f))) || (mode == M_BICUBIC && !bicubic_staged);
    sel_bilinear_rows(mode, vec, sparse_gather, stream_r32, false, d, S);
    sel_staged(mode, vec, bicubic_staged, sparse_gather, stream_r32, d, S);
    if (d.tap22 && !staged) {
        //  sel_tap22 turned this AREA request into the 2x2-tap kernel's integer tile BEFORE its staging was known to fit; a request that does not fit
        // (not reachable with the default 40 KiB budget at 3 : 2 / 2 : 1, but a smaller TSVPP_LDS_KB gets there) would fall through to the gather kernel with
        // BILINEAR's weights.  Start over as the AREA down-scale it is.
        LaunchDesc again = din;
        again.tap22_off = 1;
        return launch_fused(mode_in, out, vec, again, t, stream, info);
    }
    if (!staged) d.dma = 0;
    sel_bicubic_cols(mode, out, vec, stream_r32, d, S, stream, info);
    const size_t bc_lds = lds_bytes;
    if (mode == M_NONE && vec && d.in_aligned4 && !d.force_gather) staged = true; // colour-only fast path
    // ... and for the outputs that are the planes themselves (uint8 Y800 / NV12) a copy of 16 bytes per lane (vpp_copy16_kernel)
    d.copy16 = (mode == M_NONE && staged && (out == O_Y800_U8 || out == O_NV12_U8) && (d.dst_w & 15) == 0 && (d.dst_h & 3) == 0) ? 1 : 0;
    if (d.copy16 && !(d.shape_tx > 0 && d.shape_ty > 0)) {
        d.tx = 64;
        d.ty = 4;
    }
    d.r32 = stream_r32;
    if (d.r32) {
        d.point_kind = PK_NONE;
        d.area_direct = 0;
        staged = false;
        lds_bytes = 0;
        d.dma = 0;
        d.rpt = 1;
        d.geo = 0;
        if (!(d.shape_tx > 0 && d.shape_ty > 0)) stream_shape(d.r32, out, d, d.tx, d.ty); // (TSVPP_SHAPE overrides: shapes[0] above)
    }
    d.tx_shift = slot_shift_for(d.tx);
    if (d.r32) d.bicubic_cols = 0;
    if (d.bicubic_cols) lds_bytes = bc_lds;
    if (d.r32) d.area_stream = 0;
    if (d.area_stream) lds_bytes = S.as_lds;
    if (d.bil_rows) lds_bytes = S.br_lds;
    const int tile_w = d.area_stream ? (d.as_two ? 256 : 128) : d.bicubic_cols ? 256 : d.bil_rows ? 64 * d.br_waves : d.tx * (d.copy16 ? 16 : d.r32 ? 8 : PXW); // bicubic_cols: four waves side by side; area_stream: 2 x 2 waves; bil_rows: br_waves side by side
    const int tile_h = d.area_stream ? (d.as_two ? 8 * d.rpt : 16) : (d.bicubic_cols || d.bil_rows) ? 8 * d.rpt : (d.r32 || d.copy16) ? d.ty * 4 : d.ty * PXH * d.rpt;
    // dst_w = 4 k + 2: the last tile column shifted left to end at the right edge instead of a row tail and its second launch (tile_col0, vpp_device.h).
    // Not for the colour-only kernel and the box kernel (aligned dword reads at 4-column granularity), the streaming / copy kernels (never 4 k + 2), outputs
    // narrower than one tile; TSVPP_TAIL_SHIFT=0 (last_col0 < 0 on entry) keeps the tail launch of rounds 1-3.
    {
        const bool allowed = d.last_col0 >= 0;
        d.last_col0 = 0;
        if (allowed && vec && (d.dst_w & 3) != 0 && out < O_COUNT && mode != M_NONE && !d.area_box && !d.r32 && !d.copy16 && d.dst_w >= tile_w) d.last_col0 = d.dst_w - tile_w;
    }
    d.tiles_x = (d.dst_w + tile_w - 1) / tile_w;
    d.tiles_y = (d.dst_h + tile_h - 1) / tile_h;
    const long total = (long)d.tiles_x * d.tiles_y * d.n_frames;
    d.blocks_per_xcd = (int)((total + NUM_XCD - 1) / NUM_XCD);
    if (d.tile_order == 0) { // whole tile rows per XCD: rows padded to a multiple of 8
        const long rows = (long)d.tiles_y * d.n_frames;
        d.blocks_per_xcd = (int)(((rows + NUM_XCD - 1) / NUM_XCD) * d.tiles_x);
    }
    if (d.tile_order >= 3) { // G = 2 / 4 / 8 consecutive tile rows per XCD: rows padded to a multiple of 8 G
        const long rows = (long)d.tiles_y * d.n_frames, G = 1L << (d.tile_order - 2);
        d.blocks_per_xcd = (int)(((rows + G * NUM_XCD - 1) / (G * NUM_XCD)) * G * d.tiles_x);
    }
    if (info) {
        info->tail = d.last_col0 > 0 ? 2 : 0; // (1: the tail launch below)
        info->tx = d.tx;
        info->ty = d.ty;
        info->rpt = d.rpt;
        info->dma = d.dma;
        info->staged = staged ? 1 : 0;
        info->tiles_x = d.tiles_x;
        info->tiles_y = d.tiles_y;
    }
    auto dispatch = [&](bool v, bool st, LaunchDesc &dd, size_t lds, LaunchInfo *inf) { return launch_mode(mode, out, v, st, dd, t, lds, stream, inf); };
    if (out >= O_COUNT) { // flavours of the streaming kernel alone (O_UYVY_U8, O_YUV444_U8): the caller falls back to two passes
        if (!d.r32) return hipErrorNotSupported;
        return launch_bilinear_r32(out, d, t, stream, info);
    }
    if (d.r32 == 10) return launch_bilinear_up2(out, d, t, stream, info); // (its launcher is not vpp_kernels.hip's business: vpp_up2.h)
    hipError_t e = dispatch(vec, staged, d, lds_bytes, info);
    if (e != hipSuccess || !vec || (d.dst_w & 3) == 0 || d.last_col0 > 0) return e;
    // dst_w = 4 k + 2 without the shifted tile column: the vector-store kernels left the last two columns of every row alone (is_row_tail); one
    // more, tiny launch of the element-wise gather kernel converts them: workgroup = 1 x 64 thread tiles of 2 columns x 2 rows
    if (info) {
        info->tail = 1;
        return e;
    }
    LaunchDesc td = d;
    td.last_col0 = 0;
    td.col0 = d.dst_w & ~3;
    td.tx = 1;
    td.ty = 64; // one wave per workgroup: the few thousand tail threads spread over all CUs
    td.tx_shift = 0;
    td.rpt = 1;
    td.dma = 0;
    td.point_kind = PK_NONE; // the generic samplers give the point samplers' values (all weights are zero)
    td.area_direct = 0;
    td.bicubic_cols = 0;
    td.bil_rows = 0;
    td.area_stream = 0;
    td.tiles_x = 1;
    td.tiles_y = (d.dst_h + td.ty * PXH - 1) / (td.ty * PXH);
    const long rows = (long)td.tiles_y * td.n_frames;
    td.blocks_per_xcd = (int)((rows + NUM_XCD - 1) / NUM_XCD); // tiles_x == 1: the same for tile orders 0 .. 2
    if (td.tile_order >= 3) {
        const long G = 1L << (td.tile_order - 2);
        td.blocks_per_xcd = (int)(((rows + G * NUM_XCD - 1) / (G * NUM_XCD)) * G);
    }
    return dispatch(false, false, td, 0, nullptr);
}

} // namespace tsvpp
