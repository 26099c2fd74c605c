"""GPU: the streaming BICUBIC kernel (vpp_bicubic_r32.hip: exact ratios 3 : 2 and 2 : 1, byte coefficients through v_dot4 on the source dwords)
against the oracle's fp64 evaluation, bit for bit: every output flavour, both ratios, frame edges (the reference's tap-collapse rule at the first /
last output row and column), crops that keep / break the dword alignment, ragged pitches, narrow and wide frames (runs of an odd number of lanes),
batches, and the plan check that it IS the kernel that ran.  The thread-tile code itself is also checked on the CPU (tests/test_bicubic_r32_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu

BICUBIC = 2
KNOBS = knob_run()  # knob runs (tools/knob_matrix.sh) pick other kernels


def run(vpp, oracle, y, uv, w, dst, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0), expect=True):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=BICUBIC, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if expect is not None and not KNOBS:
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=1)["kernel"]
        assert k.startswith("vpp_bicubic_r32_kernel") == expect, k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=BICUBIC, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


FLAVOURS = [(1, 0, False), (2, 1, False), (2, 0, True), (1, 1, True), (0, 1, False), (0, 1, True), (3, 1, False), (3, 1, True), (6, 1, True)]


@pytest.mark.parametrize("fourcc,planes,norm", FLAVOURS)
@pytest.mark.parametrize("src,dst", [((960, 540), (640, 360)), ((1280, 720), (640, 360))], ids=["3:2", "2:1"])
def test_every_flavour_at_both_ratios(vpp, oracle, src, dst, fourcc, planes, norm):
    y, uv = synth_nv12(src[0], src[1], seed=31 * fourcc + planes + src[0])
    run(vpp, oracle, y, uv, src[0], dst, fourcc=fourcc, planes=planes, norm=norm)


@pytest.mark.parametrize("src,dst", [((1920, 1080), (1280, 720)), ((3840, 2160), (1920, 1080)), ((1920, 1080), (960, 540)), ((3840, 2160), (2560, 1440))])
def test_full_sizes(vpp, oracle, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False)


def test_8k_frame(vpp, oracle):
    """7680 x 4320: 32-bit plane offsets and output offsets far from their limits, 960 / 640 threads per row."""
    y, uv = synth_nv12(7680, 4320, seed=88)
    run(vpp, oracle, y, uv, 7680, (5120, 2880), planes=1, norm=False)
    run(vpp, oracle, y, uv, 7680, (3840, 2160), planes=0, norm=True)


def test_uyvy_and_yuv444_behind_it(vpp, oracle):
    """Two-pass formats: the streaming kernel writes the NV12 intermediate."""
    y, uv = synth_nv12(960, 540, seed=8)
    for fourcc, norm in ((4, False), (5, False), (4, True), (5, True)):
        run(vpp, oracle, y, uv, 960, (640, 360), fourcc=fourcc, planes=1, norm=norm, expect=None)


def test_widths_heights_and_pitches(vpp, oracle):
    # smallest frames, one thread per row, runs of an odd number of lanes (uint8 merged takes its 8-byte store path), 65 threads per row
    for (sw, sh), (dw, dh) in [((12, 6), (8, 4)), ((24, 12), (16, 8)), ((36, 18), (24, 12)), ((780, 66), (520, 44)), ((16, 8), (8, 4)), ((48, 24), (24, 12)),
                               ((1040, 40), (520, 20)), ((804, 36), (536, 24))]:
        y, uv = synth_nv12(sw, sh, seed=sw + sh)
        for fourcc, planes, norm in ((2, 0, False), (2, 1, False), (1, 0, True), (1, 1, True)):
            run(vpp, oracle, y, uv, sw, (dw, dh), fourcc=fourcc, planes=planes, norm=norm)
    y, uv = synth_nv12(960, 540, seed=5, pitch=1012)     # pitch a multiple of 4, not of 16
    run(vpp, oracle, y, uv, 960, (640, 360), norm=True)
    run(vpp, oracle, y, uv, 960, (480, 270), planes=1, expect=False)  # 270 = 4 k + 2 rows: the LDS integer kernel
    y, uv = synth_nv12(960, 540, seed=6, pitch=1013)     # planes not dword-aligned: the LDS integer kernel takes it
    run(vpp, oracle, y, uv, 960, (640, 360), norm=True, expect=False)


def test_crops(vpp, oracle):
    y, uv = synth_nv12(1280, 720, seed=9)
    run(vpp, oracle, y, uv, 1280, (640, 360), crop=(64, 32, 1024, 572), planes=1)                 # 960x540 -> 3 : 2, dword-aligned origin
    run(vpp, oracle, y, uv, 1280, (480, 270 - 2), crop=(4, 6, 964, 542), norm=True, expect=None)  # 960x536 -> 2 : 1
    run(vpp, oracle, y, uv, 1280, (640, 360), crop=(66, 32, 1026, 572), planes=1, expect=False)   # origin not dword-aligned: another kernel, same bits
    run(vpp, oracle, y, uv, 1280, (640, 360), crop=(63, 33, 1023, 573), norm=True, expect=False)  # odd origin (U / V swap quirk)


def test_constant_and_extreme_frames(vpp, oracle):
    """Negative lobes at full contrast: sums below 0 and above 255 exercise the clamp of both passes; the borders carry the extremes."""
    for val in (0, 255):
        y = np.full((360, 960), val, np.uint8)
        uv = np.full((180, 960), 255 - val, np.uint8)
        run(vpp, oracle, y, uv, 960, (640, 240), planes=1)
        run(vpp, oracle, y, uv, 960, (480, 180), planes=0, norm=True)
    y = (np.indices((360, 960)).sum(0) % 2 * 255).astype(np.uint8)          # checkerboard
    uv = (np.indices((180, 960))[1] // 2 % 2 * 255).astype(np.uint8)
    y[:, :2], y[:, -2:], y[:2], y[-2:] = 255, 0, 0, 255
    for dst in [(640, 240), (480, 180)]:
        run(vpp, oracle, y, uv, 960, dst, planes=1)
        run(vpp, oracle, y, uv, 960, dst, planes=0, norm=True)
        run(vpp, oracle, y, uv, 960, dst, fourcc=3, planes=1)


def test_batch_of_70_frames_and_graph_capture(vpp, oracle):
    import tensor_stream as ts
    n = 70
    frames = [synth_nv12(480, 270 + 6, seed=300 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=184, resize_type=BICUBIC, pixel_format=2, planes_pos=1, normalization=False)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    refs = {i: oracle.convert(frames[i][0], frames[i][1], dst=(320, 184), resize_type=BICUBIC, fourcc=2, planes=1, normalization=False)[0] for i in (0, 1, 63, 64, 69)}
    for i, ref in refs.items():
        assert np.array_equal(o[i].ravel(), ref), i
    # the same batch captured into a graph on a side stream (no tables, no allocations: capture needs no prepare) and replayed
    out2 = torch.zeros_like(out)
    b = vpp.make_batch(ys, uvs, fp, out=out2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            vpp.run_batch(b, side.cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    o2 = out2.cpu().numpy()
    for i, ref in refs.items():
        assert np.array_equal(o2[i].ravel(), ref), i
