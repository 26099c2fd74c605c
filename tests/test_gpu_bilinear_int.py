"""GPU: the integer thread tile of the 2x2-tap kernel (BILINEAR / AREA up-scale with dyadic weights) against the oracle, bit
for bit, next to the float tile (TSVPP_BILINEAR_INT=0 is part of tools/knob_matrix.sh): ratio classes, flavours, edges."""
import numpy as np
import pytest
import torch

from util import synth_nv12

pytestmark = pytest.mark.gpu
BILINEAR, AREA = 1, 3


def run(vpp, oracle, y, uv, w, dst, rt=BILINEAR, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0)):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, rt, fourcc, planes, norm, crop, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst", [
    ((1920, 1080), (1280, 720)),   # 1.5: quarters (the headline)
    ((1920, 1080), (960, 540)),    # 2
    ((1280, 720), (512, 288)),     # 2.5
    ((640, 360), (1280, 720)),     # 0.5: up-scale, clamped first column / row
    ((1280, 720), (1024, 576)),    # 1.25: eighths
    ((960, 540), (1280, 720)),     # 0.75
    ((1152, 648), (1024, 576)),    # 1.125: sixteenths on both axes (largest combined weights)
    ((1920, 1080), (1280, 540)),   # 1.5 x 2
    ((1280, 720), (1920, 1080)),   # 2/3: NOT dyadic -> float tile
])
def test_ratio_classes(vpp, oracle, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + 3 * dst[0])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False)


@pytest.mark.parametrize("src,dst", [((640, 360), (1280, 720)), ((960, 540), (1280, 720)), ((640, 360), (1280, 360)), ((800, 448), (1000, 560))])
def test_area_upscale_variant(vpp, oracle, src, dst):
    y, uv = synth_nv12(src[0], src[1], seed=dst[0])
    run(vpp, oracle, y, uv, src[0], dst, rt=AREA, planes=1)
    run(vpp, oracle, y, uv, src[0], dst, rt=AREA, planes=0, norm=True)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (0, 1, False), (0, 1, True), (3, 1, False), (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(960, 540, seed=70 + fourcc)
    run(vpp, oracle, y, uv, 960, (640, 360), fourcc=fourcc, planes=planes, norm=norm)


def test_edges_pitches_crops(vpp, oracle):
    y, uv = synth_nv12(1000, 600, seed=15, pitch=1037)
    run(vpp, oracle, y, uv, 1000, (500, 300), norm=True)
    run(vpp, oracle, y, uv, 1000, (322, 150), crop=(3, 5, 647, 305), norm=True)   # odd origin, width 4 k + 2
    run(vpp, oracle, y, uv, 1000, (2000, 1200), planes=1)
    for val in (0, 255):
        yy = np.full((360, 640), val, np.uint8)
        uu = np.full((180, 640), 255 - val, np.uint8)
        run(vpp, oracle, yy, uu, 640, (320, 180), planes=1)
        run(vpp, oracle, yy, uu, 640, (1280, 720), planes=1)
    y, uv = synth_nv12(64, 32, seed=16)
    for dst in [(32, 16), (128, 64), (16, 8), (42, 16), (2, 2)]:
        run(vpp, oracle, y, uv, 64, dst, norm=True)
