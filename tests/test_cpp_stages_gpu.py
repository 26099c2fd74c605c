"""GPU: the reference's public stage launchers (include/VideoProcessor.h:110-115: cropHost, resizeKernel, colorConversionKernel<T>),
re-exported by tensor-stream_amd/cpp/VideoProcessor.h as one-stage calls through the C ABI, chained by tensor-stream_amd/cpp/vpp_stages.cpp
the way the reference's Convert chains them -- every stage's buffers against the oracle's crop-only / crop + resize / full
conversion, bit for bit; the buffer contract (two device buffers per NV12 stage that the caller frees one by one, `crop = true`
frees the crop stage's pair, dst->opaque for the colour stage, the wrong T refused) is checked by the driver's exit code."""
import os
import subprocess

import numpy as np
import pytest

from util import synth_nv12

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tensor-stream_amd", "lib", "vpp_stages")
NV12 = 3


def nv12_planes(flat, w, h):
    flat = np.asarray(flat).ravel().view(np.uint8)
    return flat[: w * h], flat[w * h: w * h * 3 // 2]


@pytest.mark.parametrize("size,pitch,crop,dst,rtype,fourcc,planes,norm", [
    ((1080, 608), 1080, (0, 0, 0, 0), (480, 360), 1, 1, 0, True),            # the reference's own test shape: resize + colour
    ((1080, 608), 1088, (100, 60, 900, 500), (320, 240), 2, 2, 1, False),    # all three stages, BICUBIC, pitched input
    ((1080, 608), 1080, (121, 65, 921, 465), (0, 0), 0, 1, 1, False),        # crop (odd origin: the U / V swap quirk) + colour
    ((1920, 1080), 1920, (0, 0, 1920, 540), (0, 0), 0, 0, 1, False),         # a box as wide as the frame: Convert would skip it, the stage does not
    ((1920, 1080), 1920, (320, 180, 1600, 900), (224, 224), 3, 6, 1, True),  # AREA, HSV
    ((1280, 720), 1280, (0, 0, 0, 0), (1920, 1080), 0, 5, 0, False),         # NEAREST up-scale, YUV444
    ((642, 362), 642, (0, 0, 0, 0), (0, 0), 0, 4, 1, False),                 # colour stage alone, UYVY
])
def test_stage_by_stage(oracle, tmp_path, size, pitch, crop, dst, rtype, fourcc, planes, norm):
    assert os.path.exists(EXE), "vpp_stages not built (python -c 'import __graft_entry__ as g; g.build()')"
    w, h = size
    y, uv = synth_nv12(w, h, seed=w + dst[0] + fourcc, pitch=pitch)
    src = tmp_path / "in.nv12"
    with open(src, "wb") as f:
        f.write(y.tobytes())
        f.write(uv.tobytes())
    prefix = str(tmp_path / "o")
    args = [EXE, str(src), w, h, pitch, *crop, *dst, rtype, fourcc, planes, int(norm), prefix]
    r = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    ow, oh, nbytes = (int(v) for v in r.stdout.splitlines()[-1].split())   # (the refused wrong-T call reports itself above, as CHECK_STATUS does)

    cw, ch = crop[2] - crop[0], crop[3] - crop[1]
    cur_y, cur_uv, cur_w, cur_h = y, uv, w, h
    if cw > 0 and ch > 0:
        # the stage copies the box whatever its size: the oracle's crop (the reference's Convert applies it only when strictly smaller)
        # is pointer arithmetic on the planes, so for the full-width box the expected planes are the slices themselves
        ey = y[crop[1]: crop[3], crop[0]: crop[2]]
        euv = uv[crop[1] // 2: crop[1] // 2 + ch // 2, crop[0]: crop[0] + cw]  # chroma rows top / 2 + i / 2, bytes left + j (odd left: V first)
        gy = np.fromfile(prefix + ".crop.y", np.uint8).reshape(ch, cw)
        guv = np.fromfile(prefix + ".crop.uv", np.uint8).reshape(ch // 2, cw)
        assert np.array_equal(gy, ey)
        assert np.array_equal(guv, euv)
        if cw < w and ch < h:                                                 # and the oracle agrees where Convert would crop
            ref = oracle.convert(y, uv, crop=crop, dst=(0, 0), resize_type=0, fourcc=NV12, planes=0, normalization=False, nthreads=4, width=w)[0]
            ry, ruv = nv12_planes(ref, cw, ch)
            assert np.array_equal(gy.ravel(), ry) and np.array_equal(guv.ravel(), ruv)
        cur_y, cur_uv, cur_w, cur_h = np.ascontiguousarray(gy), np.ascontiguousarray(guv), cw, ch
    if dst[0] > 0:
        ref = oracle.convert(cur_y, cur_uv, dst=dst, resize_type=rtype, fourcc=NV12, planes=0, normalization=False, nthreads=4, width=cur_w)[0]
        ry, ruv = nv12_planes(ref, dst[0], dst[1])
        gy = np.fromfile(prefix + ".resize.y", np.uint8)
        guv = np.fromfile(prefix + ".resize.uv", np.uint8)
        assert np.array_equal(gy, ry) and np.array_equal(guv, ruv)
        cur_y, cur_uv = gy.reshape(dst[1], dst[0]), guv.reshape(dst[1] // 2, dst[0])
        cur_w, cur_h = dst
    assert (ow, oh) == (cur_w, cur_h)
    ref = oracle.convert(cur_y, cur_uv, dst=(0, 0), resize_type=0, fourcc=fourcc, planes=planes, normalization=norm, nthreads=4, width=cur_w)[0]
    got = np.fromfile(prefix + ".color", np.uint8)
    assert got.size == nbytes == ref.view(np.uint8).size
    assert np.array_equal(got, ref.ravel().view(np.uint8))
