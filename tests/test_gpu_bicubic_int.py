"""GPU: the integer BICUBIC kernel (dyadic weights) against the oracle's fp64 evaluation, bit for bit: every ratio class it
takes (quarters, halves, eighths, sixteenths; down and up), every output flavour, crops with odd origins, ragged
pitches, widths 4 k + 2, frame edges (the reference's tap-collapse rule), and the plan check that it IS the kernel that ran."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu

BICUBIC = 2


def run(vpp, oracle, y, uv, w, dst, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0), expect_int=True):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=BICUBIC, pixel_format=fourcc, planes_pos=planes,
                            normalization=norm)
    if expect_int is not None and not knob_run():  # knob runs (tools/knob_matrix.sh) pick other kernels
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1])["kernel"]
        # (exactly 3 : 2 / 2 : 1 on dword-aligned planes: the streaming integer kernel of vpp_bicubic_r32.hip, tests/test_gpu_bicubic_r32.py)
        assert (k.startswith("vpp_bicubic_int_kernel") or k.startswith("vpp_bicubic_r32_kernel")) == expect_int, k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=BICUBIC, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst", [
    ((1920, 1080), (1280, 720)),   # 1.5: quarters (the headline geometry)
    ((1920, 1080), (960, 540)),    # 2: halves
    ((1280, 720), (512, 288)),     # 2.5
    ((1280, 720), (320, 180)),     # 4
    ((640, 360), (1280, 720)),     # 0.5: up-scale, clamped first column / row (w = 0)
    ((1280, 720), (1024, 576)),    # 1.25: eighths
    ((960, 540), (1280, 720)),     # 0.75: up-scale in eighths
    ((1440, 648), (640, 288)),     # 2.25: eighths
    ((1152, 648), (1024, 576)),    # 1.125: sixteenths
    ((1920, 1080), (1280, 540)),   # 1.5 x 2: mixed
])
def test_ratio_classes(vpp, oracle, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0])
    # (round 6: from a ratio product of 6.5 the column kernel's EXACT instance takes the dyadic requests too -- profiles/r06_bicubic_int_vs_cols.txt)
    staged = (src[0] / dst[0]) * (src[1] / dst[1]) < 6.5
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True, expect_int=staged)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False, expect_int=staged)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (2, 1, True), (0, 1, False), (0, 1, True), (3, 1, False),
                                                 (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(960, 540, seed=77 + fourcc)
    run(vpp, oracle, y, uv, 960, (640, 360), fourcc=fourcc, planes=planes, norm=norm)


def test_ragged_pitch_odd_crops_and_narrow_tails(vpp, oracle):
    y, uv = synth_nv12(1000, 600, seed=5, pitch=1037)  # pitch not a multiple of 16: every staged row has its own misalignment
    run(vpp, oracle, y, uv, 1000, (500, 300), norm=True)
    run(vpp, oracle, y, uv, 1000, (400, 300), planes=1)                          # 2.5 x 2
    run(vpp, oracle, y, uv, 1000, (322, 150), crop=(3, 5, 647, 305), norm=True)  # odd origin (U/V swap quirk), width 4 k + 2
    run(vpp, oracle, y, uv, 1000, (214, 100), crop=(101, 50, 529, 250), planes=1)
    run(vpp, oracle, y, uv, 1000, (666, 400), planes=1, expect_int=False)        # 1.5015...: NOT dyadic -> float kernel
    y, uv = synth_nv12(64, 32, seed=6)
    for dst in [(32, 16), (128, 64), (16, 8), (42, 16)]:
        run(vpp, oracle, y, uv, 64, dst, norm=True, expect_int=None)


def test_constant_and_extreme_frames(vpp, oracle):
    """Negative lobes at full contrast: sums below 0 and above 255 exercise the clamp and the sign of the shift."""
    for val in (0, 255):
        y = np.full((360, 640), val, np.uint8)
        uv = np.full((180, 640), 255 - val, np.uint8)
        run(vpp, oracle, y, uv, 640, (320, 180), planes=1)
    y = (np.indices((360, 640)).sum(0) % 2 * 255).astype(np.uint8)          # checkerboard
    uv = (np.indices((180, 640))[1] // 2 % 2 * 255).astype(np.uint8)
    for dst in [(320, 180), (1280, 720), (256, 144)]:
        run(vpp, oracle, y, uv, 640, dst, planes=1)
        run(vpp, oracle, y, uv, 640, dst, planes=0, norm=True)


def test_batch_of_70_frames(vpp, oracle):
    import tensor_stream as ts
    n = 70
    frames = [synth_nv12(480, 270, seed=300 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=180, resize_type=BICUBIC, pixel_format=2, planes_pos=0, normalization=True)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in (0, 1, 63, 64, 69):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(320, 180), resize_type=BICUBIC, fourcc=2, planes=0, normalization=True)
        assert np.array_equal(o[i].ravel().view(np.uint8), ref.view(np.uint8)), i
