"""GPU: the contiguous-run AREA box kernel (integer horizontal ratios 4..8; vpp_area_box.hip) against the oracle, bit for bit:
square and non-square boxes, every output flavour, padded pitches, 4-aligned crops, widths 4 k + 2, and the fall-back to the
general direct kernel when a crop origin breaks the 4-byte alignment."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
AREA = 3


def run(vpp, oracle, y, uv, w, dst, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0), expect=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=AREA, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if expect is not None and not knob_run():
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1])["kernel"]
        assert k.startswith(expect), k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=AREA, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst,kernel", [
    ((3840, 2160), (640, 360), "vpp_area_box_kernel<6,1"),   # C5
    ((3840, 2160), (960, 540), "vpp_area_box_kernel<4,1"),
    ((1920, 1080), (384, 216), "vpp_area_box_kernel<5,1"),
    ((1792, 1008), (256, 144), "vpp_area_box_kernel<7,1"),
    ((2048, 1152), (256, 144), "vpp_area_box_kernel<8,1"),
    ((3840, 2160), (640, 480), "vpp_area_box_kernel<6,0"),   # 6 x 4.5
    ((1920, 1080), (480, 240), "vpp_area_box_kernel<4,0"),   # 4 x 4.5
    ((2560, 1440), (512, 192), "vpp_area_box_kernel<5,0"),   # 5 x 7.5
    ((1920, 1080), (240, 270), "vpp_area_box_kernel<8,0"),   # 8 x 4
    ((1924, 1084), (962, 542), "vpp_area_box_kernel<2,1"),   # 2 x 2 and 3 x 3: below the direct threshold, the box kernel all the same (4 k + 2 columns: 4 k
                                                             # columns at exactly 2 : 1 with fp32 outputs are the 2x2-tap integer tile's since round 4)
    ((1920, 1080), (640, 360), "vpp_area_box_kernel<3,1"),
    ((3840, 2160), (1280, 720), "vpp_area_box_kernel<3,1"),
    ((1920, 1080), (960, 360), "vpp_area_box_kernel<2,0"),   # 2 x 3
    ((1920, 1080), (640, 540), "vpp_area_box_kernel<3,0"),   # 3 x 2
    ((1920, 1080), (960, 432), "vpp_area_box_kernel<2,0"),   # 2 x 2.5
])
def test_box_ratios(vpp, oracle, src, dst, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[1])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True, expect=kernel)
    # (uint8 at exactly 2 : 1 on both axes is the streaming kernel's, vpp_bilinear_r32.hip)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False, expect=None if kernel.startswith("vpp_area_box_kernel<2,1") else kernel)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (0, 1, False), (0, 1, True), (3, 1, False), (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(1536, 864, seed=40 + fourcc)
    run(vpp, oracle, y, uv, 1536, (256, 144), fourcc=fourcc, planes=planes, norm=norm)


def test_pitches_crops_and_tails(vpp, oracle):
    y, uv = synth_nv12(1284, 600, seed=9, pitch=1536)
    run(vpp, oracle, y, uv, 1284, (214, 100), norm=True, expect="vpp_area_box_kernel<6,1")        # width 4 k + 2: row tail launch
    y, uv = synth_nv12(2000, 1200, seed=10, pitch=2052)                                           # pitch % 16 != 0, % 4 == 0
    run(vpp, oracle, y, uv, 2000, (300, 180), crop=(100, 60, 1900, 1140), planes=1, expect="vpp_area_box_kernel<6,1")
    run(vpp, oracle, y, uv, 2000, (300, 180), crop=(101, 60, 1901, 1140), planes=1, expect="vpp_area_direct_kernel")  # odd origin: no 4-byte alignment
    run(vpp, oracle, y, uv, 2000, (400, 200), crop=(4, 2, 1604, 1002), norm=True, expect="vpp_area_box_kernel<4,0")
    y, uv = synth_nv12(2000, 1200, seed=11, pitch=2002)                                           # pitch % 4 != 0 -> general kernel
    run(vpp, oracle, y, uv, 2000, (400, 240), norm=True, expect="vpp_area_direct_kernel")


def test_extreme_frames(vpp, oracle):
    for val in (0, 255):
        y = np.full((720, 1280), val, np.uint8)
        uv = np.full((360, 1280), 255 - val, np.uint8)
        run(vpp, oracle, y, uv, 1280, (160, 90), planes=1)       # 8 x 8 boxes of 255: the largest sums
        run(vpp, oracle, y, uv, 1280, (320, 120), norm=True)     # 4 x 6
