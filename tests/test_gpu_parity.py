"""GPU: the fused HIP path (through the C ABI) against the CPU oracle and the reference's goldens.

Bar: bit-exact for uint8 output; 0 ULP for the fp32-normalised output (north_star allows 1).
"""
import numpy as np
import pytest
import torch

from util import coverage_frame, synth_nv12, ulp_diff

pytestmark = pytest.mark.gpu

RGB24, BGR24 = 1, 2
PLANAR, MERGED = 0, 1
NEAREST, BILINEAR, BICUBIC, AREA = 0, 1, 2, 3


def run_hip(vpp, y, uv, width=None, **kw):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=kw.get("dst", (0, 0))[0], height=kw.get("dst", (0, 0))[1],
                            crop_coords=kw.get("crop", (0, 0, 0, 0)), resize_type=kw.get("resize_type", 0),
                            pixel_format=kw.get("fourcc", RGB24), planes_pos=kw.get("planes", MERGED),
                            normalization=kw.get("normalization", False))
    ty = torch.from_numpy(y).cuda()
    tuv = torch.from_numpy(uv).cuda()
    out = vpp.Convert(ty, tuv, fp, width=width)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def check(vpp, oracle, y, uv, width=None, **kw):
    got = run_hip(vpp, y, uv, width=width, **kw)
    ref, ow, oh = oracle.convert(y, uv, crop=kw.get("crop", (0, 0, 0, 0)), dst=kw.get("dst", (0, 0)),
                                 resize_type=kw.get("resize_type", 0), fourcc=kw.get("fourcc", RGB24),
                                 planes=kw.get("planes", MERGED), normalization=kw.get("normalization", False),
                                 nthreads=8, width=width)
    assert got.shape == oracle.shape_for(kw.get("fourcc", RGB24), kw.get("planes", MERGED), ow, oh)
    got = got.ravel()
    assert got.dtype == ref.dtype and got.size == ref.size
    if got.dtype == np.uint8:
        bad = int((got != ref).sum())
        assert bad == 0, f"{bad} of {got.size} bytes differ from the oracle ({kw})"
    else:
        assert ulp_diff(got, ref) == 0, f"fp32 output differs from the oracle by {ulp_diff(got, ref)} ulp ({kw})"
    return got


@pytest.mark.parametrize("fourcc,name", [(RGB24, "RGB24"), (BGR24, "BGR24")])
def test_reference_golden_files_bit_exact(vpp, golden, fourcc, name):
    """NV12 -> RGB24/BGR24 MERGED fp32 of the reference's own 320x240 frame == its golden dump
    (reference tests/src/VPPTests.cpp:338-384)."""
    got = run_hip(vpp, golden["Y"], golden["UVp"], fourcc=fourcc, planes=MERGED, normalization=True)
    assert got.shape == (240, 320, 3) and got.dtype == np.float32
    assert np.array_equal(got.ravel().view(np.uint32), golden[name])


@pytest.mark.parametrize("fourcc", [RGB24, BGR24])
@pytest.mark.parametrize("planes", [PLANAR, MERGED])
@pytest.mark.parametrize("norm", [False, True])
def test_colour_only_all_layouts(vpp, oracle, golden, fourcc, planes, norm):
    check(vpp, oracle, golden["Y"], golden["UVp"], fourcc=fourcc, planes=planes, normalization=norm)


def test_all_2pow24_yuv_triples(vpp, oracle):
    """Every (Y,U,V) input through the colour stage, u8 and fp32 (covers clamps, sub-16 luma, x/255)."""
    y, uv = coverage_frame()
    check(vpp, oracle, y, uv, fourcc=RGB24, planes=PLANAR, normalization=False)
    got = check(vpp, oracle, y, uv, fourcc=BGR24, planes=MERGED, normalization=True)
    # x/255 sequence hits all 256 quotients
    assert np.unique(got).size == 256


def test_all_2pow24_yuv_triples_hsv_and_formats(vpp, oracle):
    """Every (Y,U,V) input through HSV -- whose three divisions per pixel run as an in-range refinement chain instead of
    the compiler's full IEEE sequence: this is the exhaustive check over everything reachable -- and through the
    Y800 / NV12 / UYVY / YUV444 outputs (x/255 on integers and on multiples of 1/16)."""
    y, uv = coverage_frame()
    check(vpp, oracle, y, uv, fourcc=6, planes=MERGED, normalization=True)
    for fourcc in (0, 3, 4, 5):
        check(vpp, oracle, y, uv, fourcc=fourcc, planes=MERGED, normalization=True)


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
@pytest.mark.parametrize("src,dst", [((1920, 1080), (1280, 720)),   # headline, ratio 1.5
                                     ((1080, 608), (480, 360)),     # reference test size, non-dyadic
                                     ((1080, 608), (540, 304)),     # exact 2x
                                     ((640, 360), (1280, 720)),     # 2x up-scale (AREA -> bilinear variant)
                                     ((320, 240), (358, 202)),      # odd-ish ratios, dst_w % 4 != 0
                                     ((642, 362), (214, 182))])     # ratio 3 x 1.989, src_w % 4 != 0
def test_resize_types_bit_exact(vpp, oracle, rt, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=rt * 100 + src[0])
    check(vpp, oracle, y, uv, dst=dst, resize_type=rt, fourcc=RGB24, planes=MERGED, normalization=False)


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_resize_fp32_planar_bgr(vpp, oracle, rt):
    y, uv = synth_nv12(1080, 608, seed=7 + rt)
    check(vpp, oracle, y, uv, dst=(720, 404), resize_type=rt, fourcc=BGR24, planes=PLANAR, normalization=True)


@pytest.mark.parametrize("crop", [(0, 0, 1280, 720), (120, 64, 600, 400), (121, 65, 601, 401), (1, 0, 321, 200),
                                  (480, 340, 1080, 608)])
def test_crop_only_including_odd_origin(vpp, oracle, crop):
    """Crop is pointer arithmetic in the fused kernel; an odd left/top reproduces the reference's
    chroma quirk (src/Crop.cu:10-13)."""
    y, uv = synth_nv12(1920 if crop[2] > 1080 else 1080, 1080 if crop[2] > 1080 else 608, seed=crop[0] + 3)
    check(vpp, oracle, y, uv, crop=crop, fourcc=RGB24, planes=MERGED)
    check(vpp, oracle, y, uv, crop=crop, fourcc=BGR24, planes=PLANAR, normalization=True)


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_crop_plus_resize(vpp, oracle, rt):
    # reference tests/src/VPPTests.cpp:266-298 shapes: crop then resize
    y, uv = synth_nv12(1080, 608, seed=11 + rt)
    check(vpp, oracle, y, uv, crop=(480, 340, 1080, 608), dst=(480, 320), resize_type=rt, planes=PLANAR)
    check(vpp, oracle, y, uv, crop=(121, 65, 601, 401), dst=(300, 200), resize_type=rt, planes=MERGED, normalization=True)


def test_crop_ignored_unless_strictly_smaller_in_both_dims(vpp, oracle):
    y, uv = synth_nv12(640, 360, seed=5)
    got = check(vpp, oracle, y, uv, crop=(0, 0, 640, 200))  # same width -> no crop (src/VideoProcessor.cpp:109)
    assert got.size == 640 * 360 * 3


def test_resize_to_same_size_is_no_resize(vpp, oracle):
    y, uv = synth_nv12(640, 360, seed=6)
    check(vpp, oracle, y, uv, dst=(640, 360), resize_type=BICUBIC)


@pytest.mark.parametrize("pitch", [2048, 1984])
def test_padded_pitch(vpp, oracle, pitch):
    """Decoder surfaces have pitch > width (linesize aligned to 256/512 B)."""
    y, uv = synth_nv12(1920, 1080, seed=9, pitch=pitch)
    check(vpp, oracle, y, uv, width=1920, fourcc=BGR24, planes=PLANAR, normalization=True)
    check(vpp, oracle, y, uv, width=1920, dst=(1280, 720), resize_type=BILINEAR, fourcc=BGR24, planes=PLANAR, normalization=True)


# ---- BASELINE.json configs (C2..C5 + headline), full size, against the oracle ----
def test_config_C2_1080p_bgr_planar_fp32(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=2)
    check(vpp, oracle, y, uv, fourcc=BGR24, planes=PLANAR, normalization=True)


def test_colour_only_planar_fp32_on_256x8_tiles(vpp, oracle):
    """Round 6: the colour-only kernel on 64 x 4 workgroups (widths that are multiples of 256, from 1280 columns): every pixel of 720p and 1440p frames, a padded pitch."""
    for (w, h, pitch) in ((1280, 720, 1280), (2560, 1440, 2816)):
        y, uv = synth_nv12(w, h, seed=w, pitch=pitch)
        check(vpp, oracle, y, uv, width=w, fourcc=BGR24, planes=PLANAR, normalization=True)


def test_config_C3_crop_bilinear_256(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=3)
    check(vpp, oracle, y, uv, crop=(0, 0, 1280, 720), dst=(256, 256), resize_type=BILINEAR, fourcc=RGB24, planes=PLANAR, normalization=True)


def test_config_C4_4k_bicubic_merged_u8(vpp, oracle):
    y, uv = synth_nv12(3840, 2160, seed=4)
    check(vpp, oracle, y, uv, dst=(1280, 720), resize_type=BICUBIC, fourcc=BGR24, planes=MERGED, normalization=False)


def test_config_C5_4k_area_planar_fp32(vpp, oracle):
    y, uv = synth_nv12(3840, 2160, seed=5)
    check(vpp, oracle, y, uv, dst=(640, 360), resize_type=AREA, fourcc=BGR24, planes=PLANAR, normalization=True)


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_headline_1080p_to_720p_bgr_planar_fp32(vpp, oracle, rt):
    y, uv = synth_nv12(1920, 1080, seed=20 + rt)
    check(vpp, oracle, y, uv, dst=(1280, 720), resize_type=rt, fourcc=BGR24, planes=PLANAR, normalization=True)


def test_batch_equals_single_frames(vpp, oracle):
    """tsvpp_convert_batch (one launch, 70 frames -> two launches) == per-frame conversions."""
    import tensor_stream as ts
    n = 70
    frames = [synth_nv12(640, 360, seed=100 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=180, resize_type=BILINEAR, pixel_format=BGR24, planes_pos=PLANAR, normalization=True)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == (n, 3, 180, 320)
    for i in (0, 1, 63, 64, 69):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(320, 180), resize_type=BILINEAR, fourcc=BGR24,
                                   planes=PLANAR, normalization=True)
        assert ulp_diff(out[i].ravel(), ref) == 0


def test_errors(vpp):
    import tensor_stream as ts
    y = torch.zeros((360, 640), dtype=torch.uint8, device="cuda")
    uv = torch.zeros((180, 640), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="-2"):
        vpp.Convert(y, uv, ts.FrameParameters(width=321, height=181))  # odd output size
    with pytest.raises(RuntimeError, match="-2"):
        vpp.Convert(y, uv, ts.FrameParameters(width=320, height=180, resize_type=7))
    with pytest.raises(RuntimeError, match="-3"):
        vpp.Convert(y, uv, ts.FrameParameters(crop_coords=(600, 0, 700, 100)))  # crop box outside the frame


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_gather_fallback_kernel_matches_too(oracle, rt, monkeypatch):
    """The global-gather kernel (used for huge footprints / dst_w % 4 != 0) stays covered by forcing it."""
    import tensor_stream as ts
    monkeypatch.setenv("TSVPP_FORCE_GATHER", "1")
    v = ts.VideoProcessor(device=0)
    monkeypatch.delenv("TSVPP_FORCE_GATHER")
    y, uv = synth_nv12(1080, 608, seed=31 + rt)
    check(v, oracle, y, uv, dst=(480, 360), resize_type=rt, fourcc=BGR24, planes=PLANAR, normalization=True)
    check(v, oracle, y, uv, crop=(121, 65, 601, 401), dst=(300, 200), resize_type=rt, planes=MERGED)
    v.Close()


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
@pytest.mark.parametrize("src,dst", [((3840, 2160), (224, 224)),     # ratio 17 x 9.6: footprint too big for LDS -> gather
                                     ((3840, 2160), (640, 360)),     # ratio 6: smaller workgroup shape
                                     ((1920, 1080), (3840, 2160)),   # 2x up-scale
                                     ((1280, 720), (1920, 1080)),    # 1.5x up-scale
                                     ((1922, 1082), (1284, 724))])   # unaligned pitch: per-row LDS shift
def test_staging_shapes(vpp, oracle, rt, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=rt * 7 + dst[0])
    check(vpp, oracle, y, uv, dst=dst, resize_type=rt, fourcc=BGR24, planes=PLANAR, normalization=False)


@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC])
@pytest.mark.parametrize("src,dst", [((1920, 1080), (640, 360)),     # ratio 3: all weights zero -> point-sampling kernel
                                     ((3840, 2160), (768, 432)),     # ratio 5
                                     ((1920, 1080), (384, 360)),     # 5 x 3
                                     ((1282, 722), (428, 242))])     # ~2.995: NOT zero-weight -> interpolating kernels
def test_point_sampling_kernel(vpp, oracle, rt, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=rt * 13 + dst[0], pitch=src[0] + 6)
    check(vpp, oracle, y, uv, width=src[0], dst=dst, resize_type=rt, fourcc=BGR24, planes=MERGED, normalization=False)
    check(vpp, oracle, y, uv, width=src[0], crop=(2, 2, src[0] - 2, src[1] - 2), dst=dst, resize_type=rt, fourcc=RGB24, planes=PLANAR,
          normalization=True)


@pytest.mark.parametrize("src,dst", [((1440, 810), (640, 360)),    # 2.25: quarter weights
                                     ((1600, 900), (640, 360)),    # 2.5
                                     ((2560, 1440), (640, 360)),   # 4: all ones, one weight dword
                                     ((960, 540), (128, 72)),      # 7.5: eight taps
                                     ((1024, 576), (128, 72)),     # 8
                                     ((800, 450), (640, 360)),     # 1.25
                                     ((1920, 1080), (1280, 360)),  # 1.5 x 3
                                     ((1922, 1082), (1281 + 1, 722))])  # non-dyadic neighbour of 1.5 -> generic float path
def test_area_dyadic_integer_kernel(vpp, oracle, src, dst):
    """AREA down-scales whose weights are all k/2^s run on integer box sums (v_dot4_u32_u8); must equal the
    reference's float accumulation bit for bit."""
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0], pitch=src[0] + 10)
    check(vpp, oracle, y, uv, width=src[0], dst=dst, resize_type=AREA, fourcc=BGR24, planes=PLANAR, normalization=True)
    check(vpp, oracle, y, uv, width=src[0], crop=(3, 2, src[0] - 1, src[1] - 2) if (src[0] - 4) % 2 == 0 else (0, 0, 0, 0), dst=dst,
          resize_type=AREA, fourcc=RGB24, planes=MERGED, normalization=False)


def test_64_consumers_each_on_its_own_stream(oracle):
    """C5's shape: 64 concurrent consumers of one GPU -- 64 pooled streams, conversions issued
    back to back without any host synchronisation in between, results identical to the oracle."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0, max_consumers=64)
    frames = [synth_nv12(640, 360, seed=300 + i) for i in range(4)]
    dev = [(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()) for y, uv in frames]
    fp = ts.FrameParameters(width=320, height=180, resize_type=AREA, pixel_format=BGR24, planes_pos=PLANAR, normalization=True)
    outs = [v.Convert(*dev[c % 4], fp, consumer=f"consumer{c}") for c in range(64)]
    streams = {v.consumer_stream(f"consumer{c}") for c in range(64)}
    assert len(streams) == 64
    torch.cuda.synchronize()
    refs = [oracle.convert(y, uv, dst=(320, 180), resize_type=AREA, fourcc=BGR24, planes=PLANAR, normalization=True)[0] for y, uv in frames]
    for c, o in enumerate(outs):
        assert ulp_diff(o.cpu().numpy().ravel(), refs[c % 4]) == 0
    with pytest.raises(RuntimeError, match="-3"):
        v.consumer_stream("consumer64")
    v.Close()


@pytest.mark.parametrize("src,dst", [((1920, 1080), (224, 224)),   # 8.57 x 4.82: 9 taps -> three weight quads
                                     ((1920, 1080), (300, 300)),   # 6.4 x 3.6
                                     ((1920, 1080), (640, 400)),   # 3 x 2.7: one quad
                                     ((1282, 722), (224, 224)),    # unaligned pitch, 5.72 x 3.22
                                     ((1920, 1080), (160, 90))])   # 12 x 12 (dyadic -> integer kernel; 12 taps)
def test_area_large_ratio_direct_kernels(vpp, oracle, src, dst):
    """Large-ratio AREA reads its boxes straight from global memory (float weights when they are not dyadic);
    the accumulation order is the reference's, so results stay bit-identical."""
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + 1, pitch=src[0] + 2)
    check(vpp, oracle, y, uv, width=src[0], dst=dst, resize_type=AREA, fourcc=RGB24, planes=PLANAR, normalization=True)
    check(vpp, oracle, y, uv, width=src[0], crop=(6, 4, src[0] - 4, src[1] - 2), dst=dst, resize_type=AREA, fourcc=BGR24, planes=MERGED)


def test_area_direct_kernels_with_misaligned_crop_origin(vpp, oracle):
    """Regression: the direct kernels address aligned dwords from a dword-aligned base; an odd crop origin makes the
    plane pointer itself misaligned (first byte of the first row sits BEFORE the first aligned dword of the offset)."""
    y, uv = synth_nv12(1920, 1080, seed=77)
    for crop, dst in [((1, 2, 1601, 902), (160, 90)),      # ratio 10: dyadic integer kernel
                      ((3, 1, 1603, 901), (320, 180)),     # ratio 5
                      ((1, 1, 1501, 901), (224, 224)),     # non-dyadic float kernel
                      ((2, 0, 1502, 900), (224, 224))]:
        check(vpp, oracle, y, uv, crop=crop, dst=dst, resize_type=AREA, fourcc=BGR24, planes=PLANAR, normalization=True)


def test_hip_graph_capture_and_replay(vpp, oracle):
    """The conversion allocates, frees and synchronises nothing (after tsvpp_prepare), so a whole batch can be captured
    in a HIP graph once and replayed on new frame contents -- the reference's Convert (cudaMalloc/cudaFree per frame)
    cannot be captured at all."""
    import tensor_stream as ts
    n = 6
    frames_a = [synth_nv12(640, 360, seed=400 + i) for i in range(n)]
    frames_b = [synth_nv12(640, 360, seed=500 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames_a])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames_a])).cuda()
    fp = ts.FrameParameters(width=426, height=240, resize_type=AREA, pixel_format=BGR24, planes_pos=PLANAR, normalization=True)
    vpp.prepare(fp, 640, 360)                       # AREA tables built outside the capture
    out = vpp._alloc(fp.parameters, 640, 360, n)
    batch = vpp.make_batch(ys, uvs, fp, out=out)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        vpp.run_batch(batch, s.cuda_stream)         # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        vpp.run_batch(batch, torch.cuda.current_stream().cuda_stream)
    ys.copy_(torch.from_numpy(np.stack([f[0] for f in frames_b])).cuda())
    uvs.copy_(torch.from_numpy(np.stack([f[1] for f in frames_b])).cuda())
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in range(n):
        ref, _, _ = oracle.convert(frames_b[i][0], frames_b[i][1], dst=(426, 240), resize_type=AREA, fourcc=BGR24, planes=PLANAR,
                                   normalization=True)
        assert ulp_diff(o[i].ravel(), ref) == 0


def test_consumer_pool_semantics(vpp):
    """findFree: a name keeps its stream; a 6th name on a 5-slot pool is VREADER_ERROR
    (reference include/Common.h:225-237, src/VideoProcessor.cpp:100-103)."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0, max_consumers=2)
    a = v.consumer_stream("a")
    b = v.consumer_stream("b")
    assert a != b and v.consumer_stream("a") == a
    with pytest.raises(RuntimeError, match="-3"):
        v.consumer_stream("c")
    v.Close()


@pytest.mark.parametrize("src,dst", [((1280, 720), (256, 256)),    # BASELINE C3's ROI: ratio 5 (every x weight is zero) x 2.8125
                                     ((1280, 720), (256, 240)),    # 5 x 3: both axes (the point sampler)
                                     ((720, 1280), (256, 256)),    # 2.8125 x 5: every y weight is zero
                                     ((960, 540), (320, 300)),     # 3 x 1.8: zero x weights on the LDS 2x2-tap kernel
                                     ((540, 960), (300, 320)),     # 1.8 x 3
                                     ((1280, 360), (256, 240))])   # 5 x 1.5
@pytest.mark.parametrize("planes,norm", [(PLANAR, True), (MERGED, False)])
def test_bilinear_with_one_axis_of_zero_weights(vpp, oracle, src, dst, planes, norm):
    """Odd integer ratio on ONE axis: the samplers do not fetch the taps a zero weight multiplies (LaunchDesc::wx_zero / wy_zero) -- same bits."""
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[1])
    check(vpp, oracle, y, uv, dst=dst, resize_type=BILINEAR, fourcc=RGB24, planes=planes, normalization=norm)


@pytest.mark.parametrize("g_term,bits", [(1, 0), (2, 2048)])
def test_colour_g_term_variants_over_all_2pow24_triples(oracle, g_term, bits):
    """TSVPP_OPT_COLOR_G_TERM: the other two operation trees of the green chroma term, for holders of goldens of the real reference binary -- every
    (Y, U, V) triple against the oracle's matching contraction variant, through the LDS colour kernel (planar uint8) and the streaming output side (merged)."""
    import tensor_stream as ts
    from tensor_stream import vpp as V
    CT_RESIZE, CT_INNER = 1 | 2 | 8 | 16 | 64, 256
    v = ts.VideoProcessor(device=0)
    v.set_option(V.OPT_COLOR_G_TERM, g_term)
    assert v.get_option(V.OPT_COLOR_G_TERM) == g_term
    y, uv = coverage_frame()
    oracle.set_contract(CT_RESIZE | CT_INNER | bits)
    try:
        check(v, oracle, y, uv, fourcc=RGB24, planes=PLANAR, normalization=False)
        # ... and behind a resize whose streaming kernel ends in vpp_r32_store.h's colour stage
        ys, uvs = synth_nv12(1920, 1080, seed=31 + g_term)
        check(v, oracle, ys, uvs, dst=(1280, 720), resize_type=BILINEAR, fourcc=BGR24, planes=MERGED, normalization=False)
    finally:
        oracle.set_contract(-1)
    v.set_option(V.OPT_COLOR_G_TERM, 0)
    check(v, oracle, ys, uvs, dst=(1280, 720), resize_type=BILINEAR, fourcc=BGR24, planes=MERGED, normalization=False)
    v.Close()
