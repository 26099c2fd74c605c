"""GPU: AREA down-scales with 9-32 horizontal taps (ratios 8.5 .. 31: 1080p -> 224 x 224, 4K -> 224 x 224, thumbnails) against the
oracle, bit for bit: the streaming kernel (vpp_area_stream.hip: one wave per tile, source rows through a wave-private LDS ring, two
columns per lane up to ratio ~8 and one beyond), every instantiated tap count, pitches that are no multiple of 16 (the column-per-lane
kernel / the generic path), crops, row tails, every output flavour, forced for small ratios too."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
AREA = 3


def run(vpp, oracle, y, uv, w, dst, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0)):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=AREA, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=AREA, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst,kernel", [
    ((1920, 1080), (224, 224), "vpp_area_stream_kernel<3,OUT>"),   # 8.57 x 4.82
    ((3840, 2160), (384, 384), "vpp_area_stream_kernel<3,OUT>"),   # 10 x 5.6
    ((3840, 2160), (224, 224), "vpp_area_stream_kernel<6,OUT>"),   # 17.1 x 9.6: 64-column tiles
    ((1920, 1080), (128, 72), "vpp_area_stream_kernel<4,OUT>"),     # 15 x 15: integer ratio, weights all 1
    ((3840, 2160), (240, 136), "vpp_area_stream_kernel<4,OUT>"),         # 16 x 15.9
    ((3840, 2160), (150, 84), "vpp_area_stream_kernel<8,OUT>"),          # 25.6 x 25.7
    ((1920, 1080), (96, 96), "vpp_area_stream_kernel<6,OUT>"),           # 20 x 11.25
    ((3840, 2160), (160, 90), "vpp_area_stream_kernel<6,OUT>"),          # 24 x 24
    ((3840, 2160), (128, 72), "vpp_area_stream_kernel<8,OUT>"),     # 30 x 30
    ((3840, 2160), (96, 54), "vpp_fused_gather_kernel"),            # 40 x 40: beyond 32 taps -> generic path
])
def test_wide_ratios(vpp, oracle, src, dst, kernel):
    import tensor_stream as ts
    if not knob_run():   # (tools/knob_matrix*.sh replay the suite under knobs that change the selection)
        p = ts.describe(ts.FrameParameters(width=dst[0], height=dst[1], resize_type=AREA, normalization=True, planes_pos=0), src[0], src[1])
        assert p["kernel"].startswith(kernel), p
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False)


def test_wide_ratios_pitch_crop_flavours(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=77, pitch=1933)          # pitch % 16 != 0: per-row misalignment of the LDS-DMA rows
    run(vpp, oracle, y, uv, 1920, (224, 224), norm=True)
    run(vpp, oracle, y, uv, 1920, (128, 72), planes=1)
    run(vpp, oracle, y, uv, 1920, (150, 100), crop=(101, 53, 1801, 1003), norm=True)   # odd origin; 11.3 x 9.5
    run(vpp, oracle, y, uv, 1920, (222, 126), planes=1)          # width 4 k + 2: row tail launch
    for fourcc, planes, norm in [(1, 1, False), (0, 1, True), (3, 1, False), (4, 1, False), (6, 1, True)]:   # RGB24, Y800, NV12, UYVY (second pass), HSV
        run(vpp, oracle, y, uv, 1920, (224, 224), fourcc=fourcc, planes=planes, norm=norm)
    for val in (0, 255):
        yy = np.full((1080, 1920), val, np.uint8)
        uu = np.full((540, 1920), 255 - val, np.uint8)
        run(vpp, oracle, yy, uu, 1920, (224, 224), planes=1)
        run(vpp, oracle, yy, uu, 1920, (96, 96), norm=True)


def test_streaming_kernel_pitch_crop_flavours(vpp, oracle):
    """The same on pitches that ARE multiples of 16 (the streaming kernel): crops whose origins misalign the plane pointers, the
    planes' last rows, widths 4 k + 2, one-tile frames, every output flavour."""
    y, uv = synth_nv12(1920, 1080, seed=78, pitch=1936)
    run(vpp, oracle, y, uv, 1920, (224, 224), norm=True)
    run(vpp, oracle, y, uv, 1920, (150, 100), crop=(101, 53, 1801, 1003), norm=True)   # odd origin; 11.3 x 9.5
    run(vpp, oracle, y, uv, 1920, (222, 126), planes=1)                                 # width 4 k + 2: row tail launch
    run(vpp, oracle, y, uv, 1920, (100, 60), crop=(1119, 479, 1919, 1079), norm=True)   # bottom-right corner: last rows / last bytes of both planes
    run(vpp, oracle, y, uv, 1920, (30, 18), planes=1)                                   # 64 x 60: one partial tile, 64-column mode
    for fourcc, planes, norm in [(1, 1, False), (1, 0, False), (2, 1, True), (0, 1, True), (3, 1, False), (4, 1, False), (5, 1, True), (6, 1, True)]:
        run(vpp, oracle, y, uv, 1920, (224, 224), fourcc=fourcc, planes=planes, norm=norm)
        run(vpp, oracle, y, uv, 1920, (100, 96), fourcc=fourcc, planes=planes, norm=norm)   # 19.2 x 11.25: 64-column tiles


@pytest.mark.parametrize("env", [{"TSVPP_AREA_STREAM": "2"}, {"TSVPP_AREA_STREAM": "0"}])
def test_streaming_kernel_forced_everywhere_and_off(oracle, env, monkeypatch):
    """TSVPP_AREA_STREAM=2: every float-weight AREA down-scale (2 x 2 taps upward) on the streaming kernel; =0: none (round-2 kernels)."""
    import tensor_stream as ts
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    v = ts.VideoProcessor(device=0)  # the knobs are read when the context is created
    try:
        for src, dst in [((1080, 608), (480, 360)), ((1920, 1080), (1366, 768)), ((1920, 1080), (300, 300)), ((1920, 1080), (224, 224)), ((1280, 720), (854, 480)),
                         ((3840, 2160), (224, 224)), ((640, 360), (50, 30))]:
            y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[1])
            run(v, oracle, y, uv, src[0], dst, norm=True)
            run(v, oracle, y, uv, src[0], dst, planes=1)
    finally:
        v.Close()


@pytest.mark.parametrize("src,dst,kernel", [
    ((1920, 1080), (300, 300), "vpp_area_cols_kernel<2,"),    # 6.4 x 3.6: NK = 2
    ((1920, 1080), (416, 416), "vpp_area_cols_kernel<2,"),    # 4.6 x 2.6
    ((1920, 1080), (600, 400), "vpp_area_direct_float_kernel<1"),  # 3.2 x 2.7: up to four horizontal taps stay on the direct kernel (its neighbour in the dispatch)
    ((3840, 2160), (416, 720), "vpp_area_cols_kernel<3,"),    # 9.2 x 3: NK = 3 (below the streaming kernel's cross-over: 30 taps)
    ((1920, 1080), (270, 270), "vpp_area_cols_kernel<2,"),    # 7.1 x 4: 4 k + 2 columns
])
def test_column_per_lane_kernel_with_prefetch(vpp, oracle, src, dst, kernel):
    """Round 6: vpp_area_cols_kernel requests tap row a + 1 before it accumulates tap row a (unconditional loads above the plane's last row, the edge loads only on it):
    every instance, full frames (the bottom rows' last tap rows), crops -- and with the streaming kernel's divisor table switched off by a second context."""
    import tensor_stream as ts
    if not knob_run():
        p = ts.describe(ts.FrameParameters(width=dst[0], height=dst[1], resize_type=AREA, normalization=True, planes_pos=0), src[0], src[1])
        assert p["kernel"].startswith(kernel), p
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + 3 * dst[0])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False)
    run(vpp, oracle, y, uv, src[0], dst, fourcc=0, planes=1, norm=True)   # Y800: no chroma pass
    w, h = src
    run(vpp, oracle, y, uv, w, dst, crop=(w // 8 + 1, h // 8 + 1, w - w // 8 + 1, h - h // 8 + 1), norm=True)   # odd origin, other ratios' tables
    old = os.environ.get("TSVPP_AREA_DIVTAB")
    os.environ["TSVPP_AREA_DIVTAB"] = "0"   # (tests/conftest.py sets the debug gate): the kernel sums its weights itself
    try:
        v2 = ts.VideoProcessor(device=0)
        run(v2, oracle, y, uv, src[0], dst, planes=0, norm=True)
        v2.Close()
    finally:
        if old is None:
            del os.environ["TSVPP_AREA_DIVTAB"]
        else:
            os.environ["TSVPP_AREA_DIVTAB"] = old
