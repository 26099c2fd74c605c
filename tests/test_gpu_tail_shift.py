"""GPU: outputs 4 k + 2 columns wide (854x480, 1366x768, 270x270).  The vector-store kernels used to leave the last two columns of every row to a second,
element-wise launch; now the launch's last tile column is shifted left so that it ends at the frame's right edge (LaunchDesc::last_col0, tile_col0 in
vpp_device.h): every kernel that takes such a request, every output flavour, batches and crops against the oracle, bit for bit -- and what keeps the
tail launch (outputs narrower than one tile, the box kernel, colour-only requests) still does."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
NEAREST, BILINEAR, BICUBIC, AREA = 0, 1, 2, 3
KNOBS = knob_run()


def check(vpp, oracle, y, uv, w, dst, rt, fourcc=2, planes=0, norm=True, crop=(0, 0, 0, 0), n=1, tail=2, kernel=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if not KNOBS:
        d = ts.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=n)
        assert d["tail"] == tail, (d, w, y.shape, dst, rt, fourcc, crop)
        if kernel:
            assert kernel in d["kernel"], (d, kernel)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    got = vpp.Convert(ty, tuv, fp, width=w) if n == 1 else vpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    for g in ([got] if n == 1 else [got[0], got[n - 1]]):
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (w, y.shape, dst, rt, fourcc, planes, norm, crop, n, bad[:8], bad.size)


FLAVOURS = [(2, 0, True), (1, 1, False), (1, 0, False), (2, 1, True)]  # fp32 planar, uint8 merged, uint8 planar, fp32 merged


@pytest.fixture(scope="module")
def frame():
    return synth_nv12(1920, 1080, seed=854, pitch=2048)


@pytest.mark.parametrize("fourcc,planes,norm", FLAVOURS)
@pytest.mark.parametrize("dst,rt,kernel", [
    ((854, 480), NEAREST, "point"), ((1366, 768), NEAREST, "point"), ((270, 270), NEAREST, "point"),
    ((854, 480), BILINEAR, "vpp_bilinear_kernel"), ((1366, 768), BILINEAR, "vpp_bilinear_kernel"), ((270, 270), BILINEAR, "bilinear_rows"),
    ((854, 480), BICUBIC, "bicubic_cols"), ((1366, 768), BICUBIC, "bicubic_cols"), ((270, 270), BICUBIC, "sparse"),
    ((854, 480), AREA, "area_direct_float"), ((1366, 768), AREA, "areaf"), ((270, 270), AREA, "area_cols"), ((266, 266), AREA, "area_stream"),
    ((418, 418), AREA, "area_cols"), ((642, 362), AREA, None), ((1082, 608), AREA, None), ((1082, 608), BILINEAR, None), ((1082, 608), BICUBIC, None),
])
def test_every_kernel_every_flavour(vpp, oracle, frame, dst, rt, kernel, fourcc, planes, norm):
    y, uv = frame
    check(vpp, oracle, y, uv, 1920, dst, rt, fourcc=fourcc, planes=planes, norm=norm, kernel=kernel if (fourcc, planes, norm) == (2, 0, True) else None)


def test_other_outputs_batches_crops(vpp, oracle, frame):
    y, uv = frame
    for rt in (NEAREST, BILINEAR, BICUBIC, AREA):
        check(vpp, oracle, y, uv, 1920, (854, 480), rt, fourcc=3, planes=1, norm=False)       # NV12 uint8 (the intermediate of the two-pass formats)
        check(vpp, oracle, y, uv, 1920, (854, 480), rt, fourcc=0, planes=1, norm=False)       # Y800
        check(vpp, oracle, y, uv, 1920, (854, 480), rt, fourcc=6, planes=1, norm=True)        # HSV
        check(vpp, oracle, y, uv, 1920, (854, 480), rt, n=3)
        check(vpp, oracle, y, uv, 1920, (854, 480), rt, n=64, planes=1, fourcc=1, norm=False)
        check(vpp, oracle, y, uv, 1920, (598, 338), rt, crop=(13, 7, 1293, 727))               # odd origin (the U / V swap quirk)
    for fourcc in (4, 5):  # UYVY / YUV444 (two passes: the first one writes NV12 with the shifted tile column)
        import tensor_stream as ts
        fp = ts.FrameParameters(width=854, height=480, resize_type=BILINEAR, pixel_format=fourcc, planes_pos=1, normalization=False)
        ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
        got = vpp.Convert(ty, tuv, fp, width=1920).cpu().numpy().ravel()
        ref, _, _ = oracle.convert(y, uv, dst=(854, 480), resize_type=BILINEAR, fourcc=fourcc, planes=1, normalization=False, nthreads=8, width=1920)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), fourcc


def test_upscales_and_dyadic_weights(vpp, oracle):
    y, uv = synth_nv12(640, 360, seed=3, pitch=640)
    for rt in (NEAREST, BILINEAR, BICUBIC, AREA):  # AREA up-scale = the 2x2-tap kernel's areaup flavour
        check(vpp, oracle, y, uv, 640, (854, 480), rt)
        check(vpp, oracle, y, uv, 640, (854, 480), rt, fourcc=1, planes=1, norm=False)
    y, uv = synth_nv12(1708, 960, seed=4, pitch=1712)
    check(vpp, oracle, y, uv, 1708, (854, 480), BILINEAR)                                      # weights 1/2: integer window tile
    check(vpp, oracle, y, uv, 1708, (854, 480), BILINEAR, fourcc=1, planes=1, norm=False)     # ... uint8: no geometry tables with a shifted column
    check(vpp, oracle, y, uv, 1708, (854, 480), BICUBIC, kernel="bicubic_int")
    check(vpp, oracle, y, uv, 1708, (854, 480), BICUBIC, fourcc=1, planes=1, norm=False)
    check(vpp, oracle, y, uv, 1708, (854, 480), AREA, tail=1, kernel="area_box")              # the box kernel keeps the tail launch
    y, uv = synth_nv12(2430, 1080, seed=5, pitch=2432)
    check(vpp, oracle, y, uv, 2430, (270, 120), AREA)                                          # ratio 9: dyadic AREA straight from global memory


def test_what_keeps_the_tail_launch(vpp, oracle, frame):
    y, uv = frame
    for rt in (NEAREST, BILINEAR, BICUBIC, AREA):
        check(vpp, oracle, y, uv, 1920, (54, 30), rt, tail=1)                                  # narrower than one tile
        check(vpp, oracle, y, uv, 1920, (54, 30), rt, tail=1, fourcc=1, planes=1, norm=False)
    y2, uv2 = synth_nv12(854, 480, seed=6, pitch=854)
    check(vpp, oracle, y2, uv2, 854, (0, 0), NEAREST, tail=1)                                  # colour only
    check(vpp, oracle, y2, uv2, 854, (0, 0), NEAREST, tail=1, fourcc=1, planes=1, norm=False)
