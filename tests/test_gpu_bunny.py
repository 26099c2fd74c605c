"""GPU: BASELINE.json configs[0] (C1) on real content -- the picture of the reference's demo clip (tests/golden/bunny_idr129_1280x720.npz, see
tests/golden/make_bunny_idr.py): NV12 -> RGB24 MERGED uint8 at native size through the HIP path against the oracle, bit for bit, plus every resize type
and the other flavours on the same picture (the synthetic frames of the other tests are white noise: this one has flat sky, edges and texture -- the
BICUBIC tie test, the clamps of the colour stage and the AREA sums see realistic statistics)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def bunny():
    z = np.load(os.path.join(HERE, "golden", "bunny_idr129_1280x720.npz"))
    return np.ascontiguousarray(z["y"]), np.ascontiguousarray(z["uv"])


def check(vpp, oracle, y, uv, dst=(0, 0), rt=0, fourcc=1, planes=1, norm=False, crop=(0, 0, 0, 0)):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=1280)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=1280)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, rt, fourcc, planes, norm, crop, bad[:8], bad.size)


def test_c1_native_size_rgb24_merged_uint8(vpp, oracle, bunny):
    check(vpp, oracle, *bunny)


@pytest.mark.parametrize("rt", [0, 1, 2, 3])
@pytest.mark.parametrize("dst", [(640, 360), (854, 480), (1920, 1080), (256, 256), (426, 240), (300, 200)])
def test_every_resize_type_on_real_content(vpp, oracle, bunny, rt, dst):
    check(vpp, oracle, *bunny, dst=dst, rt=rt)
    check(vpp, oracle, *bunny, dst=dst, rt=rt, fourcc=2, planes=0, norm=True)


@pytest.mark.parametrize("fourcc,planes,norm", [(0, 1, False), (3, 1, False), (4, 1, False), (5, 1, False), (6, 1, True), (1, 0, True)])
def test_other_flavours_and_a_crop(vpp, oracle, bunny, fourcc, planes, norm):
    check(vpp, oracle, *bunny, fourcc=fourcc, planes=planes, norm=norm)
    check(vpp, oracle, *bunny, dst=(480, 270), rt=2, fourcc=fourcc, planes=planes, norm=norm, crop=(161, 91, 1121, 631))
