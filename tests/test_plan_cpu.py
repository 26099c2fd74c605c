"""CPU: the host-side selection logic of the HIP path -- which kernel, workgroup shape, row pairs per thread and
LDS budget a request gets -- through tsvpp_describe (a dry run of launch_fused: nothing is launched, no GPU needed).
Guards the measured heuristics against silent regressions."""
import pytest

import tensor_stream as ts

N, B, C, A = 0, 1, 2, 3
Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = range(7)


def plan(src, dst=(0, 0), rt=N, fourcc=BGR24, planes=0, norm=True, crop=(0, 0, 0, 0), pitch=0, **kw):
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes,
                            normalization=norm)
    return ts.describe(fp, src[0], src[1], pitch=pitch, **kw)


def test_headline_runs_the_dma_staged_bilinear_kernel_on_256x8_tiles():
    """Round 2: the fp32 2x2-tap kernel sits at the HBM floor of its tile pattern and 256 x 8 tiles have the lowest one
    (tools/membench2.hip; profiles/r02_shape_sweep.txt): 64 x 4 thread tiles, one row pair per thread."""
    p = plan((1920, 1080), (1280, 720), B, pitch=2048)
    assert p["mode"] == "bilinear" and p["out"] == "f32_planar"
    assert p["kernel"] == "vpp_bilinear_kernel<bilinear,OUT>" and (p["shape"], p["rpt"], p["dma"]) == ("64x4", 1, 1)
    assert p["tiles"] == "5x90" and p["frames"] == 64 and p["lds"] <= 40 * 1024
    # whole tile rows per XCD: rows padded to a multiple of 8
    assert p["grid"] == (90 * 64 + 7) // 8 * 8 * 5
    # widths that do not fill 256-wide tiles stay on 128; uint8 outputs (VALU-bound) keep the tall thread tiles
    assert plan((3840, 2160), (1920, 1080), B)["shape"] == "32x8" and plan((3840, 2160), (1920, 1080), B)["geo"] == 0
    # ... and, with dyadic weights and 16-byte pitches, read host-built geometry tables on 64-wide workgroups (scalar row records)
    p = plan((1920, 1080), (1536, 864), B, norm=False)                  # 5 : 4
    assert (p["kernel"], p["shape"], p["geo"]) == ("vpp_bilinear_kernel<bilinear,OUT>", "64x4", 1)
    p = plan((1920, 1080), (1536, 864), B, norm=False, pitch=1928)      # pitch % 16 != 0: rows of differing misalignment
    assert (p["shape"], p["geo"]) == ("32x8", 0)
    # uint8 outputs at exactly 3 : 2 (1080p -> 720p) or 2 : 1 (4K -> 1080p): the streaming kernel without LDS -- BILINEAR, AREA and
    # NEAREST; fp32 and BICUBIC not
    for rt, kind in ((B, "bilinear"), (A, "area"), (N, "nearest")):
        p = plan((1920, 1080), (1280, 720), rt, norm=False, pitch=2048)
        assert (p["kernel"], p["shape"], p["tiles"], p["geo"]) == ("vpp_bilinear_r32_kernel<OUT,%s,3:2>" % kind, "64x4", "3x45", 0)
        p = plan((3840, 2160), (1920, 1080), rt, norm=False, planes=1)
        assert (p["kernel"], p["shape"], p["tiles"]) == ("vpp_bilinear_r32_kernel<OUT,%s,2:1>" % kind, "64x4", "4x68")
    # ... except BILINEAR 2 : 1 with large planar outputs, where the LDS kernel with geometry tables measured faster
    assert plan((3840, 2160), (1920, 1080), B, norm=False)["kernel"] == "vpp_bilinear_kernel<bilinear,OUT>"
    assert plan((1920, 1080), (960, 540), B, norm=False)["kernel"] == "vpp_bilinear_r32_kernel<OUT,bilinear,2:1>"
    assert plan((3840, 2160), (1920, 1080), B)["kernel"] == "vpp_bilinear_kernel<bilinear,OUT>"
    # BICUBIC at those two ratios (round 4): its own streaming kernel, every output flavour (fp32 too); other dyadic ratios keep the LDS integer kernel
    for norm, planes in ((False, 0), (True, 0), (False, 1), (True, 1)):
        p = plan((1920, 1080), (1280, 720), C, norm=norm, planes=planes, pitch=2048)
        # 1280 columns = 160 threads per row: for uint8 outputs (VALU-bound) 32-wide workgroups leave no lane idle (64-wide: 20 %); fp32 outputs (bound by
        # the issue of memory instructions): 64 wide -- neighbour dwords by wave shuffle instead of loads --, two thread rows (short tiles)
        assert (p["kernel"], p["shape"], p["tiles"]) == ("vpp_bicubic_r32_kernel<OUT,3:2>", "64x2" if norm else "32x8", "3x90" if norm else "5x23")
        assert plan((3840, 2160), (1920, 1080), C, norm=norm, planes=planes)["kernel"] == "vpp_bicubic_r32_kernel<OUT,2:1>"
    assert plan((1920, 1080), (1280, 720), C, fourcc=HSV)["kernel"] == "vpp_bicubic_r32_kernel<OUT,3:2>"
    assert plan((1920, 1080), (1536, 864), C)["kernel"] == "vpp_bicubic_int_kernel<OUT>"                                  # 5 : 4
    assert plan((1920, 1080), (1280, 720), C, pitch=1922)["kernel"] == "vpp_bicubic_int_kernel<OUT>"                      # planes not dword-aligned
    assert plan((1932, 1080), (1288, 720), C)["kernel"] == "vpp_bicubic_r32_kernel<OUT,3:2>"                              # width 8 k
    assert plan((1926, 1080), (1284, 720), C)["kernel"] == "vpp_bicubic_int_kernel<OUT>"                                  # width 8 k + 4
    assert plan((1920, 1086), (1280, 724), C)["kernel"] == "vpp_bicubic_r32_kernel<OUT,3:2>"                              # height 4 k
    assert plan((1920, 1084), (960, 542), C)["kernel"] == "vpp_bicubic_int_kernel<OUT>"                                   # 2 : 1, height 4 k + 2
    assert plan((1920, 1080), (1280, 720), B, norm=False, pitch=1922)["kernel"] == "vpp_bilinear_kernel<bilinear,OUT>"   # planes not dword-aligned
    assert plan((1926, 1080), (1284, 720), B, norm=False)["kernel"] == "vpp_bilinear_kernel<bilinear,OUT>"               # width 8 k + 4
    p = plan((1920, 1080), (1366, 768), B, norm=False)                  # float weights: the tables were measured to lose
    assert (p["shape"], p["geo"]) == ("32x8", 0)


@pytest.mark.parametrize("src,dst,rt,kernel", [
    ((1920, 1080), (0, 0), N, "vpp_color_kernel"),                       # C2: no resize
    ((3840, 2160), (1280, 720), C, "vpp_point_kernel<PK_BICUBIC0"),         # C4's geometry, fp32 output: every cubic weight is zero -> LDS point kernel
    ((3840, 2160), (1280, 720), B, "vpp_point_kernel<PK_BILINEAR0"),
    ((1960, 1120), (280, 160), B, "vpp_point_kernel<PK_BILINEAR0"),         # 7 : 1: all weights zero, no streaming instance -> LDS point kernel
    ((3840, 2160), (480, 270), N, "vpp_bilinear_rows_kernel<OUT,nearest>"),  # 8 : 1 -- ratio product >= 36: the tapped rows as LDS-DMA row segments (round 6; vpp_point_kernel before)
    ((1920, 1080), (300, 300), N, "vpp_point_kernel<PK_NEAREST"),            # 6.4 x 3.6 = 23: the LDS point kernel keeps it (measured 23 % faster there)
    ((1920, 1080), (384, 216), B, "vpp_point_kernel<PK_BILINEAR0"),          # 5 x 5, all weights zero: a point sampler, below the threshold
    ((3840, 2160), (256, 144), C, "vpp_bilinear_rows_kernel<OUT,point>"),    # 15 x 15, all weights zero
    ((1920, 1080), (1280, 720), N, "vpp_point_kernel<PK_NEAREST"),
    ((3840, 2160), (640, 360), A, "vpp_area_box_kernel<6,1"),            # C5: 6 x 6 box from contiguous dword runs
    ((3840, 2160), (960, 540), A, "vpp_area_box_kernel<4,1"),            # 4 x 4
    ((3840, 2160), (640, 480), A, "vpp_area_box_kernel<6,0"),            # 6 x 4.5: integer horizontally, dyadic rows
    ((3840, 2160), (768, 432), A, "vpp_area_box_kernel<5,1"),            # 5 x 5
    ((1920, 1080), (480, 360), A, "vpp_area_dyadic_kernel<1,3"),        # 4 x 3: the vertical ratio is below the direct threshold
    ((1920, 1080), (1280, 720), A, "vpp_bilinear_kernel<bilinear,OUT>[area-weights]"),  # fp32 at 3 : 2 / 2 : 1: the 2x2-tap kernel with AREA's weights (round 4)
    ((1920, 1080), (1536, 864), A, "vpp_area_dyadic_kernel<1,2"),       # 1.25: dyadic weights, LDS
    ((1920, 1080), (960, 540), A, "vpp_bilinear_kernel<bilinear,OUT>[area-weights]"),
    ((1924, 1084), (962, 542), A, "vpp_area_box_kernel<2,1"),           # 2 x 2 (4 k + 2 columns) and 3 x 3: the box kernel as well (round 2: measured faster than the LDS kernel)
    ((1920, 1080), (640, 360), A, "vpp_area_box_kernel<3,1"),
    ((1920, 1080), (960, 360), A, "vpp_area_box_kernel<2,0"),           # 2 x 3
    ((2560, 1440), (1920, 1080), A, "vpp_areaf_kernel<2,2"),            # 4/3: float weights, 2 x 2 taps
    ((1080, 608), (480, 360), A, "vpp_areaf_kernel<3,2"),               # 2.25 x 1.69
    ((1920, 1080), (800, 450), A, "vpp_area_direct_float_kernel<1"),     # 2.4: float weights from global memory
    ((1920, 1080), (224, 224), A, "vpp_area_stream_kernel<3,OUT>"),     # 8.57 x 4.82 = 45 taps: rows streamed through a wave-private ring, two columns per lane
    ((3840, 2160), (224, 224), A, "vpp_area_stream_kernel<6,OUT>"),     # 17.1 x 9.6: 18 horizontal taps (the next instantiated count: 24), 64-column tiles
    ((3840, 2160), (128, 72), A, "vpp_area_stream_kernel<8,OUT>"),      # 30 x 30
    ((3840, 2160), (96, 54), A, "vpp_fused_gather_kernel"),              # 40 x 40: beyond 32 taps
    ((1920, 1080), (300, 300), A, "vpp_area_cols_kernel<2,8"),           # 6.4 x 3.6 = 28 taps: below the streaming kernel's cross-over -> one output column per lane, taps from global
                                                                         # memory; 8-row tiles: 32-row ones would be 3200 workgroups a launch (round 4)
    ((1920, 1080), (416, 416), A, "vpp_area_cols_kernel<2,32"),
    ((1280, 720), (1920, 1080), A, "vpp_bilinear_kernel<areaup"),  # AREA up-scale = the bilinear variant
    ((1920, 1080), (1280, 720), C, "vpp_bicubic_r32_kernel<OUT,3:2>"),  # 1.5: the streaming kernel (round 4)
    ((3840, 2160), (1920, 1080), C, "vpp_bicubic_r32_kernel<OUT,2:1>"), # 2
    ((1280, 720), (512, 288), C, "vpp_bicubic_int_kernel"),             # 2.5: weights in quarters -> the LDS integer kernel
    ((960, 540), (1920, 1080), C, "vpp_bicubic_int_kernel"),            # 0.5
    ((1280, 720), (1920, 1080), C, "vpp_bicubic_cols_kernel<OUT,tie,dense>"),   # 2/3: not dyadic -> wave-per-tile kernel with the tie test
    ((1080, 608), (480, 360), C, "vpp_bicubic_cols_kernel<OUT,tie,dense>"),
    ((1920, 1080), (224, 224), C, "vpp_bicubic_cols_kernel<OUT,tie,sparse>"),    # vertical ratio >= 4: only the tapped rows are evaluated
    ((3840, 2160), (640, 360), C, "vpp_bicubic_cols_kernel<OUT,exact,sparse>"),  # 6: dyadic, but too sparse for the staged integer kernel
    ((1920, 1080), (224, 224), B, "vpp_bilinear_rows_kernel<OUT,2x2>"),  # very sparse sampling: the tapped rows as LDS-DMA row segments (round 5; byte gathers before)
    ((1920, 1080), (416, 416), B, "vpp_bilinear_rows_kernel<OUT,2x2>"),  # 4.6 x 2.6 = 11.98: from a product of 7.5 on when the vertical ratio is >= 2.1 (round 6: 561 -> 373 us per 512 frames)
    ((1920, 1080), (540, 540), B, "vpp_bilinear_kernel<bilinear"),       # 3.56 x 2.0: the LDS-staged kernel keeps vertical ratios up to 2
])
def test_kernel_families(src, dst, rt, kernel):
    p = plan(src, dst, rt)
    assert p["kernel"].startswith(kernel), p


def test_c3_crop_folds_into_pointers_and_the_sparse_bilinear_streams_its_rows():
    p = plan((1920, 1080), (256, 256), B, fourcc=RGB24, crop=(0, 0, 1280, 720), pitch=2048)
    assert (p["src"], p["dst"]) == ("1280x720", "256x256") and p["kernel"] == "vpp_bilinear_rows_kernel<OUT,wx0>", p  # 5.0 x 2.8125: every horizontal weight is zero
    assert (p["tiles"], p["shape"], p["lds"]) == ("1x32", "16x4", 4 * (3 * 8 * 16 * 21 + 64)), p                       # four waves side by side, 21 chunks per row segment
    # a pitch that is no multiple of 16 (rows with different misalignments), unaligned outputs: byte gathers
    assert plan((1920, 1080), (256, 256), B, fourcc=RGB24, crop=(0, 0, 1280, 720), pitch=1924)["kernel"].startswith("vpp_fused_gather_kernel")
    assert plan((1920, 1080), (256, 256), B, fourcc=RGB24, crop=(0, 0, 1280, 720), pitch=2048, aligned_outputs=False)["kernel"] == "vpp_fused_gather_kernel<MODE,OUT,false>"


def test_c4_runs_the_streaming_point_sampler():
    """BASELINE config C4: 4K -> 720p BICUBIC -> BGR24 MERGED uint8: exact 3 : 1, every cubic weight zero -> vpp_point_rn.hip (uint8 flavours; fp32 stays on the LDS kernel)."""
    p = plan((3840, 2160), (1280, 720), C, fourcc=BGR24, planes=1, norm=False, pitch=3840)
    assert p["kernel"] == "vpp_point_rn_kernel<OUT,3:1,centre>" and p["shape"] == "64x4" and p["tiles"] == "3x45", p
    assert plan((3840, 2160), (1280, 720), N, fourcc=RGB24, planes=0, norm=False, pitch=3840)["kernel"] == "vpp_point_rn_kernel<OUT,3:1,nearest>"
    assert plan((3840, 2160), (960, 540), N, fourcc=RGB24, planes=1, norm=False, pitch=3840)["kernel"] == "vpp_point_rn_kernel<OUT,4:1,nearest>"
    assert plan((1920, 1080), (384, 216), B, fourcc=RGB24, planes=1, norm=False, pitch=2048)["kernel"] == "vpp_point_rn_kernel<OUT,5:1,centre>"
    assert plan((3840, 2160), (1280, 720), C, fourcc=BGR24, planes=1, norm=False, pitch=3842)["kernel"].startswith("vpp_point_kernel")  # planes not dword-aligned


def test_small_outputs_keep_two_row_thread_tiles():
    # fp32 2x2-tap kernel: always one row pair (the HBM write pattern prefers short tiles); the VALU-bound flavours take
    # the 4-row thread tile once the launch has 48 * num_cus workgroups (16 in the other kernels)
    for n in (8, 64):
        assert plan((1920, 1080), (1280, 720), B, n_frames=n)["rpt"] == 1
    assert plan((3840, 2160), (1920, 1080), B, n_frames=64)["rpt"] == 1
    # (pitch 1922: planes not dword-aligned, so neither the streaming 3 : 2 kernel nor the geometry tables apply -- the plain LDS kernel)
    assert plan((1920, 1080), (1280, 720), B, n_frames=8, norm=False, pitch=1922)["rpt"] == 1    # 10 x 23 x 8 tiles of 128 x 32 = 1840
    assert plan((1920, 1080), (1280, 720), B, n_frames=64, norm=False, pitch=1922)["rpt"] == 2   # 14720 >= 12288
    assert plan((1920, 1080), (960, 540), B, n_frames=64, norm=False, pitch=1922)["rpt"] == 1    # 8704
    # dyadic AREA: most resident workgroups first; since round 2 the LDS-DMA layout is as compact as the register-staged one
    # (any number of chunks per row), so it wins the tie: 1080p -> 960x540 two-row tiles, 21 KiB, LDS-DMA
    p = plan((1920, 1080), (960, 540), A, n_frames=64, pitch=1922)       # (pitch 1922: not dword-aligned, so not the box kernel)
    assert (p["rpt"], p["dma"]) == (1, 1) and p["lds"] < 22 * 1024
    p = plan((1920, 1080), (1536, 864), A, n_frames=64)                  # >= 5 per CU either way: taller tile, LDS-DMA
    assert (p["rpt"], p["dma"]) == (2, 1)
    # the compact layout: the uint8 headline tile needs 20 KiB (round 1: 26.9 KiB with power-of-two row pitches)
    assert plan((1920, 1080), (1280, 720), B, norm=False, pitch=1922)["lds"] <= 20 * 1024


def test_colour_only_planar_fp32_takes_256x8_tiles_at_widths_that_fill_them():
    """Round 6 (profiles/r06_color_shapes.txt): no resize, planar fp32, 1280 / 2560 / 3840 columns: 0.62-0.64 of the roofline on 32 x 8 workgroups, 0.72-0.74 on 64 x 4."""
    for src in ((1280, 720), (2560, 1440), (3840, 2160), (2048, 1152)):
        p = plan(src, (0, 0), N)
        assert p["kernel"] == "vpp_color_kernel<OUT>" and p["shape"] == "64x4", (src, p)
    assert plan((1920, 1080), (0, 0), N)["shape"] == "32x8"              # C2: 7.5 tiles of 256 columns
    assert plan((1024, 576), (0, 0), N)["shape"] == "32x8"               # no gain below 1280 columns
    assert plan((1280, 720), (0, 0), N, norm=False)["shape"] == "32x8"   # uint8 planar loses 6 % on the wide tiles
    assert plan((1280, 720), (0, 0), N, planes=1)["shape"] == "32x8"     # merged: the exchange pattern, not this one


def test_output_flavours_share_the_sampling_kernels():
    for fourcc, out in [(Y800, "y800_f32"), (NV12, "nv12_f32"), (HSV, "hsv_f32"), (RGB24, "f32_planar")]:
        p = plan((1920, 1080), (1280, 720), B, fourcc=fourcc)
        # (HSV at exactly 3 : 2 / 2 : 1: the streaming kernel -- three divisions per pixel make that flavour VALU-bound, round 4)
        assert p["out"] == out and p["kernel"].startswith("vpp_bilinear_r32_kernel<" if fourcc == HSV else "vpp_bilinear_kernel<")
    assert plan((1920, 1080), (1280, 720), B, fourcc=Y800, norm=False)["out"] == "y800_u8"
    p = plan((1920, 1080), (1366, 768), B, fourcc=UYVY)
    assert p["out"] == "nv12_u8" and p["pass2"] == "fmt_uyvy"            # two passes: resized NV12, then the format kernel
    p = plan((1920, 1080), (0, 0), N, fourcc=YUV444)
    assert p["kernel"] == "(none)" and p["pass2"] == "fmt_yuv444"        # no resize: the format kernel reads the input itself
    # uint8 UYVY / YUV444 (round 3) and fp32 UYVY (round 6) behind an exact 3 : 2 / 2 : 1 resize: ONE pass, an output of the streaming kernel -- for its three
    # tap kinds, not for fp32 YUV444, BICUBIC, other ratios, misaligned crops or misaligned outputs
    for rt, kind in ((B, "bilinear"), (A, "area"), (N, "nearest")):
        p = plan((1920, 1080), (1280, 720), rt, fourcc=UYVY, norm=False)
        assert p["out"] == "uyvy_u8" and p["kernel"] == "vpp_bilinear_r32_kernel<OUT,%s,3:2>" % kind and "pass2" not in p
        p = plan((3840, 2160), (1920, 1080), rt, fourcc=UYVY, norm=False)
        assert p["out"] == "uyvy_u8" and p["kernel"] == "vpp_bilinear_r32_kernel<OUT,%s,2:1>" % kind and "pass2" not in p
        p = plan((1920, 1080), (1280, 720), rt, fourcc=YUV444, norm=False)
        assert p["out"] == "yuv444_u8" and p["kernel"] == "vpp_bilinear_r32_kernel<OUT,%s,3:2>" % kind and "pass2" not in p
        for src, dst, ratio in (((1920, 1080), (1280, 720), "3:2"), ((3840, 2160), (1920, 1080), "2:1")):
            p = plan(src, dst, rt, fourcc=UYVY, norm=True)
            assert p["out"] == "uyvy_f32" and p["kernel"] == "vpp_bilinear_r32_kernel<OUT,%s,%s>" % (kind, ratio) and "pass2" not in p
    for kw in (dict(fourcc=YUV444, norm=True), dict(fourcc=UYVY, norm=True, rt=C), dict(fourcc=UYVY, norm=True, dst=(1000, 720)), dict(fourcc=UYVY, norm=False, rt=C), dict(fourcc=UYVY, norm=False, dst=(1000, 720)),
               dict(fourcc=UYVY, norm=False, dst=(640, 360), crop=(6, 2, 966, 542)), dict(fourcc=UYVY, norm=False, aligned_outputs=False)):
        args = dict(src=(1920, 1080), dst=(1280, 720), rt=B)
        args.update(kw)
        p = plan(**args)
        assert p["out"] == "nv12_u8" and p["pass2"].startswith("fmt_"), (kw, p)


def test_widths_4k_plus_2_stay_on_the_fast_kernels_and_unaligned_outputs_gather():
    assert plan((1920, 1080), (854, 480), A)["kernel"].startswith("vpp_area_direct_float_kernel<1")
    assert plan((1920, 1080), (1366, 768), A)["kernel"].startswith("vpp_areaf_kernel<2,2")
    p = plan((1920, 1080), (854, 480), B)
    assert p["kernel"].startswith("vpp_bilinear_kernel<") and p["tail"] == 2   # the last tile column shifted to the right edge: no row tail, no second launch
    assert plan((1920, 1080), (54, 30), B)["tail"] == 1                        # narrower than one tile: + the two-column row-tail launch
    assert plan((1920, 1080), (1280, 720), B)["tail"] == 0
    assert plan((1920, 1080), (1280, 720), B, aligned_outputs=False)["kernel"] == "vpp_fused_gather_kernel<MODE,OUT,false>"


def test_large_footprints_fall_back_to_smaller_workgroups_or_gathers():
    p = plan((7680, 4320), (2560, 1440), C)            # 8K bicubic at ratio 3 (all weights zero -> point kernel; uint8: the streaming one, no footprint at all)
    assert p["kernel"].startswith("vpp_point_kernel") and p["lds"] <= 40 * 1024
    assert plan((7680, 4320), (2560, 1440), C, norm=False)["kernel"].startswith("vpp_point_rn_kernel")
    p = plan((7680, 4320), (2800, 1576), C)            # 2.74: staged bicubic must fit the 40 KiB LDS budget
    assert p["lds"] <= 40 * 1024 and p["kernel"].startswith("vpp_bicubic_cols_kernel")


def test_status_codes_match_convert():
    fp = ts.FrameParameters(width=1281, height=720, resize_type=B)
    with pytest.raises(RuntimeError, match="-2"):
        ts.describe(fp, 1920, 1080)
    fp = ts.FrameParameters(crop_coords=(100, 0, 2000, 720))  # smaller than the frame in both dimensions, but sticking out of it
    with pytest.raises(RuntimeError, match="-3"):
        ts.describe(fp, 1920, 1080)


def test_area_weight_rows_have_unit_interior_taps():
    """The streaming AREA kernel's lean arithmetic (ratios > 16) skips the multiply of column taps 1 .. 4 * nkx - 5 because they weigh
    exactly 1.0f in EVERY row of the weight table (the reference's rows are [rest] 1 ... 1 [last fraction], truncated to ceil(ratio)
    entries); the host checks it per table (AreaTable::ones_end) and this pins the property itself on the table the C ABI exports."""
    import ctypes
    import numpy as np
    from tensor_stream import _native as N
    lib = N.lib()
    for num, den in [(3840, 224), (3840, 128), (1920, 224), (1920, 300), (4096, 240), (7680, 300), (3840, 608), (2160, 72), (1920, 104)]:
        scale = np.float32(num) / np.float32(den)
        buf = (ctypes.c_float * (1 << 18))()
        taps = ctypes.c_int(0)
        rows = lib.tsvpp_area_pattern(ctypes.c_float(float(scale)), buf, len(buf), ctypes.byref(taps))
        assert rows > 0 and taps.value == int(np.ceil(scale))
        tab = np.frombuffer(buf, np.float32, rows * taps.value).reshape(rows, taps.value)
        nkx = (taps.value + 3) // 4
        assert (tab[:, 1: max(1, 4 * nkx - 4)] == 1.0).all(), (num, den)        # what the kernel relies on
        assert (tab[:, 1: taps.value - 2] == 1.0).all()                        # what the table actually guarantees: taps 1 .. taps - 3
        assert (tab >= 0).all() and (tab[:, 0] > 0).all()


def test_the_baseline_requests_select_the_kernels_the_profiles_name():
    """the selection of the BASELINE configurations is pinned against what was PROFILED -- profiles/traffic_latest.json names the kernel each
    PMC entry was taken on (tools/traffic_json.py), bench.py refuses an entry whose kernel is no longer dispatched; here the plan itself must agree, so a threshold
    that slips in vpp_select.hip fails the CPU suite, not a later round's bench line."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    want = {"headline": "vpp_bilinear_kernel<bilinear,OUT>", "c1": "vpp_color_kernel<OUT>", "c2": "vpp_color_kernel<OUT>", "c3": "vpp_bilinear_rows_kernel<OUT,wx0>",
            "c4": "vpp_point_rn_kernel<OUT,3:1,centre>", "c5": "vpp_area_box_kernel<6,1,OUT>"}
    got = {}
    for name, (sw, sh, pitch, crop, dst, rt, fcc, planes, norm) in bench.WORKLOADS.items():
        p = plan((sw, sh), dst, bench.RESIZE[rt], fourcc=bench.FOURCC[fcc], planes=bench.PLANES[planes], norm=norm, crop=crop, pitch=pitch)
        got[name] = p["kernel"]
    assert got == want, got
    traffic = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))
    for name, entry in traffic.items():
        if name in want and entry.get("kernel"):
            assert entry["kernel"].split("::")[-1] == want[name], (name, entry["kernel"], entry.get("round"))
