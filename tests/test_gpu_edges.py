"""GPU: edge cases -- tiny and huge frames, tile-boundary sizes, ragged pitches, unaligned outputs, odd crop boxes
at the frame border, every resize type at each."""
import numpy as np
import pytest
import torch

from util import synth_nv12, ulp_diff

pytestmark = pytest.mark.gpu
RT = [0, 1, 2, 3]


def conv(vpp, oracle, y, uv, width=None, out=None, **kw):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=kw.get("dst", (0, 0))[0], height=kw.get("dst", (0, 0))[1], crop_coords=kw.get("crop", (0, 0, 0, 0)),
                            resize_type=kw.get("rt", 0), pixel_format=kw.get("fourcc", 2), planes_pos=kw.get("planes", 0),
                            normalization=kw.get("norm", False))
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=width, out=out)
    torch.cuda.synchronize()
    ref, ow, oh = oracle.convert(y, uv, crop=kw.get("crop", (0, 0, 0, 0)), dst=kw.get("dst", (0, 0)), resize_type=kw.get("rt", 0),
                                 fourcc=kw.get("fourcc", 2), planes=kw.get("planes", 0), normalization=kw.get("norm", False), nthreads=8, width=width)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    assert np.array_equal(g.view(np.uint8), ref.view(np.uint8)), kw
    return got


@pytest.mark.parametrize("rt", RT)
@pytest.mark.parametrize("src,dst", [((2, 2), (0, 0)), ((2, 2), (4, 4)), ((4, 4), (2, 2)), ((16, 2), (4, 2)), ((2, 16), (2, 4)),
                                     ((6, 4), (8, 6)), ((128, 16), (132, 18)), ((130, 18), (128, 16)), ((256, 32), (124, 14))])
def test_tiny_and_tile_boundary_sizes(vpp, oracle, rt, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] * 7 + src[1] + rt)
    conv(vpp, oracle, y, uv, dst=dst, rt=rt, planes=0, norm=True)
    conv(vpp, oracle, y, uv, dst=dst, rt=rt, planes=1, norm=False)


@pytest.mark.parametrize("rt", RT)
def test_ragged_pitches_and_sub_pitch_width(vpp, oracle, rt):
    # pitch not a multiple of 16 / 4, width smaller than pitch, different for tests of the per-row LDS shift
    for pitch, w in [(1083, 1080), (1000, 998), (649, 640)]:
        h = 120
        rng = np.random.default_rng(pitch + rt)
        y = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
        uv = rng.integers(0, 256, (h // 2, pitch), dtype=np.uint8)
        conv(vpp, oracle, y, uv, width=w, dst=(w // 2 + (w // 2) % 2, 90), rt=rt, planes=0, norm=True)
        conv(vpp, oracle, y, uv, width=w, crop=(5, 3, w - 3, h - 1) if (w - 8) % 2 == 0 else (4, 2, w - 4, h - 2), dst=(320, 64), rt=rt, planes=1)


@pytest.mark.parametrize("rt", RT)
def test_unaligned_output_pointer_takes_scalar_path(vpp, oracle, rt):
    y, uv = synth_nv12(640, 360, seed=77 + rt)
    base = torch.empty(3 * 320 * 180 * 4 + 64, dtype=torch.uint8, device="cuda")
    for off, norm in [(1, False), (4, True)]:
        n = 3 * 320 * 180 * (4 if norm else 1)
        view = base[off: off + n]
        out = view.view(torch.float32 if norm else torch.uint8).view(3, 180, 320)
        conv(vpp, oracle, y, uv, dst=(320, 180), rt=rt, planes=0, norm=norm, out=out)


@pytest.mark.parametrize("rt", RT)
def test_crop_boxes_touching_every_border(vpp, oracle, rt):
    y, uv = synth_nv12(1080, 608, seed=91 + rt)
    for crop in [(0, 0, 2, 2), (1078, 606, 1080, 608), (0, 2, 1078, 608), (2, 0, 1080, 606), (1, 1, 1079, 607), (539, 303, 541, 305)]:
        conv(vpp, oracle, y, uv, crop=crop, planes=1)
        conv(vpp, oracle, y, uv, crop=crop, dst=(64, 48), rt=rt, planes=0, norm=True)


@pytest.mark.parametrize("rt", RT)
def test_8k_frame(vpp, oracle, rt):
    """7680x4320 (33 Mpx > 2^24 samples).  NEAREST / BICUBIC / AREA-down index with integers in the reference too.  The
    reference's BILINEAR helper forms its start index in float (src/Resize.cu:6) and lands on neighbouring samples up
    here; the kernels index with integers -- a stated deviation whose expected result is the oracle's
    exact-index mode, compared bit for bit, for BILINEAR and for the AREA up-scale variant that shares the helper."""
    y, uv = synth_nv12(7680, 4320, seed=5)
    if rt in (0, 2):
        conv(vpp, oracle, y, uv, dst=(1920, 1080), rt=rt, planes=1)
        return
    if rt == 3:
        conv(vpp, oracle, y, uv, dst=(3840, 2160), rt=3, planes=0, norm=True)  # AREA down-scale: integer indices
    oracle.set_exact_index(True)
    try:
        if rt == 1:
            conv(vpp, oracle, y, uv, dst=(1920, 1080), rt=1, planes=0, norm=True)
            conv(vpp, oracle, y, uv, dst=(2880, 1620), rt=1, planes=1)
        else:
            conv(vpp, oracle, y, uv, dst=(7684, 4322), rt=3, planes=1)  # AREA up-scale (ratio <= 1 on both axes)
    finally:
        oracle.set_exact_index(False)


def test_8k_bilinear_float_index_of_the_reference_is_not_copied(vpp, oracle):
    """The faithful oracle (float start index) and the kernel DO differ at 8K -- the deviation is real, documented, and
    confined to frames with pitch * height > 2^24."""
    import tensor_stream as ts
    y, uv = synth_nv12(7680, 4320, seed=6)
    fp = ts.FrameParameters(width=1920, height=1080, resize_type=1, pixel_format=2, planes_pos=1)
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp).cpu().numpy().ravel()
    ref, _, _ = oracle.convert(y, uv, dst=(1920, 1080), resize_type=1, fourcc=2, planes=1, nthreads=8)
    assert (got != ref).any()
    # ... and only in rows whose start index exceeds 2^24 (output rows below 2^24 / 7680 / 4 = 546)
    bad_rows = np.unique(np.nonzero((got != ref).reshape(1080, -1))[0])
    assert bad_rows.min() >= 540


def test_no_resize_8k_and_max_batch_split(vpp, oracle):
    import tensor_stream as ts
    y, uv = synth_nv12(7680, 4320, seed=6)
    conv(vpp, oracle, y, uv, planes=0, norm=True)
    # 130 frames -> 3 launches (64 + 64 + 2), frames at unrelated addresses
    frames = [synth_nv12(64, 36, seed=i) for i in range(130)]
    ys = [torch.from_numpy(f[0]).cuda() for f in frames]
    uvs = [torch.from_numpy(f[1]).cuda() for f in frames]
    fp = ts.FrameParameters(width=96, height=54, resize_type=1, pixel_format=1, planes_pos=1, normalization=False)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in (0, 63, 64, 127, 128, 129):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(96, 54), resize_type=1, fourcc=1, planes=1)
        assert np.array_equal(o[i].ravel(), ref)


def test_empty_batch_and_argument_errors(vpp):
    import ctypes
    import tensor_stream as ts
    from tensor_stream import _native as N
    L = N.lib()
    fp = ts.FrameParameters().parameters
    assert L.tsvpp_convert_batch(vpp._ctx, 0, None, ctypes.byref(fp), None, None) == -3      # null arrays
    frames = (N.NV12 * 1)(N.NV12(0, 0, 0, 0, 64, 36))
    outs = (ctypes.c_void_p * 1)(0)
    assert L.tsvpp_convert_batch(vpp._ctx, 0, frames, ctypes.byref(fp), outs, None) == 0     # n = 0: nothing to do
    assert L.tsvpp_convert_batch(vpp._ctx, 1, frames, ctypes.byref(fp), outs, None) == -3    # null planes
    y = torch.zeros((36, 64), dtype=torch.uint8, device="cuda")
    bad = (N.NV12 * 1)(N.NV12(y.data_ptr(), y.data_ptr(), 32, 32, 64, 36))                   # pitch < width
    outs = (ctypes.c_void_p * 1)(y.data_ptr())
    assert L.tsvpp_convert_batch(vpp._ctx, 1, bad, ctypes.byref(fp), outs, None) == -3


@pytest.mark.parametrize("shape", [None, "16,4", "64,4", "128,2"])
def test_merged_fp32_store_exchange(oracle, monkeypatch, shape):
    """Merged fp32 outputs swap pixel quads between the lanes of a run before storing: every run length
    (workgroup widths 16 / 32 / 64 / 128), ragged right edges (last run shorter than the others, down to one
    lane), every kernel family, RGB triples and HSV."""
    import tensor_stream as ts
    if shape:
        monkeypatch.setenv("TSVPP_SHAPE", shape)
    v = ts.VideoProcessor(device=0, max_consumers=2)
    try:
        y, uv = synth_nv12(1080, 608, seed=4242, pitch=1088)
        cases = [((0, 0), 0), ((132, 76), 0), ((300, 200), 1), ((1284, 724), 1), ((516, 290), 2), ((540, 304), 3), ((360, 152), 3),
                 ((432, 244), 3), ((772, 436), 3), ((4, 2), 1), ((68, 40), 1)]
        for dst, rt in cases:
            for fourcc in (1, 6):
                conv(v, oracle, y, uv, width=1080, dst=dst, rt=rt, planes=1, norm=True, fourcc=fourcc)
        conv(v, oracle, y, uv, width=1080, crop=(121, 65, 601, 401), planes=1, norm=True, fourcc=1)
    finally:
        v.Close()


@pytest.mark.parametrize("rt,fourcc", [(1, 2), (2, 2), (3, 2), (1, 0), (3, 5)])
def test_convert_is_hip_graph_capturable(vpp, oracle, rt, fourcc):
    """After prepare() the whole call is kernel launches on the caller's stream -- no allocation, no
    synchronisation, no host->device copy -- so a caller can capture it in a hipGraph (the remedy for
    launch-bound single-frame conversion) and replay it on new frame contents."""
    import tensor_stream as ts
    frames = [synth_nv12(640, 360, seed=300 + i) for i in range(4)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=424, height=240, resize_type=rt, pixel_format=fourcc, normalization=True)
    vpp.prepare(fp, 640, 360)
    batch = vpp.make_batch(ys, uvs, fp)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        vpp.run_batch(batch)  # warm-up on the capture stream: two-pass formats grow their per-stream scratch buffer here
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        vpp.run_batch(batch)
    # new contents in the same buffers, then replay
    frames2 = [synth_nv12(640, 360, seed=400 + i) for i in range(4)]
    ys.copy_(torch.from_numpy(np.stack([f[0] for f in frames2])))
    uvs.copy_(torch.from_numpy(np.stack([f[1] for f in frames2])))
    batch["out"].zero_()
    g.replay()
    torch.cuda.synchronize()
    out = batch["out"].cpu().numpy()
    for i in range(4):
        ref, _, _ = oracle.convert(frames2[i][0], frames2[i][1], dst=(424, 240), resize_type=rt, fourcc=fourcc, normalization=True)
        assert np.array_equal(out[i].ravel().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("src,dst", [((2560, 1440), (1920, 1080)), ((1080, 608), (1068, 600)), ((1080, 608), (640, 320)),
                                     ((1000, 562), (588, 500)), ((640, 360), (636, 182))])
def test_area_float_2x2_kernel(vpp, oracle, src, dst):
    """AREA down-scales below 2x with non-dyadic weights (4/3, 1.01, 1.69 x 1.9, ...): the 2x2 float kernel; the last
    case mixes it with a ratio >= 2 on the other axis (generic staged kernel)."""
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0], pitch=(src[0] + 63) // 64 * 64 + 3)
    conv(vpp, oracle, y, uv, width=src[0], dst=dst, rt=3, planes=0, norm=True)
    conv(vpp, oracle, y, uv, width=src[0], dst=dst, rt=3, planes=1, norm=False)
    conv(vpp, oracle, y, uv, width=src[0], dst=dst, rt=3, fourcc=0, norm=True)
    conv(vpp, oracle, y, uv, width=src[0], crop=(3, 2, src[0] - 5, src[1] - 2), dst=(dst[0] - 4, dst[1] - 2), rt=3, planes=0, norm=False)


@pytest.mark.parametrize("src,dst", [((1920, 1080), (960, 540)), ((1080, 600), (360, 200)), ((1080, 600), (432, 240)),
                                     ((1280, 720), (400, 240)), ((1920, 1080), (640, 540)), ((1600, 900), (400, 300))])
def test_area_dyadic_lds_kernel_ratios_2_to_3p5(vpp, oracle, src, dst):
    """Dyadic AREA ratios from 2 to 3.5 (2, 3, 2.5, 3.2 x 3, 3 x 2, 4 x 3) run on the LDS kernel with 2, 3 or 4 taps
    per axis (v_dot4 over one weight dword); from 3.5 in both axes the direct kernel takes over."""
    y, uv = synth_nv12(src[0], src[1], seed=src[0] * 3 + dst[1], pitch=src[0] + 6)
    conv(vpp, oracle, y, uv, width=src[0], dst=dst, rt=3, planes=0, norm=True)
    conv(vpp, oracle, y, uv, width=src[0], crop=(2, 2, src[0] - 2, src[1] - 2), dst=dst, rt=3, planes=1, norm=False)


@pytest.mark.parametrize("stream", [None, "2"])
def test_area_float_kernel_up_to_3x3_taps(oracle, monkeypatch, stream):
    """Non-dyadic AREA with 2 or 3 taps per axis in every combination (3x2, 2x3, 3x3): the LDS float kernel / the direct float
    kernel; with TSVPP_AREA_STREAM=2 the streaming kernel takes them all (pitches that are multiples of 16)."""
    import tensor_stream as ts
    if stream:
        monkeypatch.setenv("TSVPP_AREA_STREAM", stream)
    v = ts.VideoProcessor(device=0, max_consumers=2)
    try:
        y, uv = synth_nv12(1080, 608, seed=99, pitch=1104 if stream else 1091)
        for dst in [(480, 360), (640, 224), (452, 256), (400, 240), (364, 380)]:
            conv(v, oracle, y, uv, width=1080, dst=dst, rt=3, planes=0, norm=True)
            conv(v, oracle, y, uv, width=1080, crop=(1, 2, 1079, 606), dst=dst, rt=3, planes=1, norm=False)
            conv(v, oracle, y, uv, width=1080, dst=dst, rt=3, fourcc=6, norm=True)
    finally:
        v.Close()


@pytest.mark.parametrize("rt", RT)
def test_widths_of_the_form_4k_plus_2_stay_on_the_fast_kernels(vpp, oracle, rt):
    """854x480, 1366x768, ... : the last thread tile of every row has two columns.  It stores on the scalar path and
    must neither read past the source rows nor disturb the merged-fp32 exchange of its wave."""
    y, uv = synth_nv12(1366, 768, seed=31 + rt, pitch=1376)
    for dst in [(854, 480), (682, 384), (1366, 768) if rt == 0 else (1370, 770), (342, 192), (170, 96), (2050, 1154)]:
        conv(vpp, oracle, y, uv, width=1366, dst=dst, rt=rt, planes=0, norm=True)
        conv(vpp, oracle, y, uv, width=1366, dst=dst, rt=rt, planes=1, norm=True)
        conv(vpp, oracle, y, uv, width=1366, dst=dst, rt=rt, planes=1, norm=False)
        conv(vpp, oracle, y, uv, width=1366, dst=dst, rt=rt, fourcc=3, norm=False)
    # colour-only and crop-only with a 4 k + 2 wide source
    conv(vpp, oracle, y, uv, width=1366, planes=0, norm=True)
    conv(vpp, oracle, y, uv, width=1366, crop=(4, 2, 1362, 766), planes=0, norm=False)
    conv(vpp, oracle, y, uv, width=1366, crop=(0, 0, 1362, 760), dst=(0, 0), fourcc=5, norm=True)


@pytest.mark.parametrize("cols", ["0", "2"])
def test_large_ratio_float_area_both_samplers(oracle, monkeypatch, cols):
    """Non-dyadic AREA at ratios >= 2 has two samplers -- a 4-column thread tile per lane and one output column per lane
    (the default picks by tap count); both must give the reference's bits for 2-4, 5-8 and 9-12 horizontal taps, every
    output flavour, partial 64 x 32 tiles, 4 k + 2 widths and crops with misaligned origins."""
    import tensor_stream as ts
    monkeypatch.setenv("TSVPP_AREA_COLS", cols)
    v = ts.VideoProcessor(device=0, max_consumers=2)
    try:
        y, uv = synth_nv12(1920, 1080, seed=515, pitch=1936)
        for dst in [(800, 450), (300, 300), (416, 234), (224, 224), (174, 98), (854, 480)]:
            conv(v, oracle, y, uv, width=1920, dst=dst, rt=3, planes=0, norm=True)
            conv(v, oracle, y, uv, width=1920, dst=dst, rt=3, planes=1, norm=True)
            conv(v, oracle, y, uv, width=1920, dst=dst, rt=3, planes=1, norm=False)
            conv(v, oracle, y, uv, width=1920, dst=dst, rt=3, fourcc=0, norm=False)
        conv(v, oracle, y, uv, width=1920, crop=(1, 1, 1501, 901), dst=(224, 224), rt=3, planes=0, norm=True)
        conv(v, oracle, y, uv, width=1920, crop=(3, 2, 1603, 902), dst=(300, 170), rt=3, planes=0, norm=False)
    finally:
        v.Close()
