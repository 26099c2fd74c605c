"""GPU: the C++ `VideoProcessor` adapter (tensor-stream_amd/cpp) driven like the reference's own VPP tests
drive the class (reference tests/src/VPPTests.cpp:566-590): AVFrame in, Convert(), opaque out."""
import os
import subprocess

import numpy as np
import pytest

from util import synth_nv12

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tensor-stream_amd", "lib", "vpp_cli")

CASES = [  # (w, h, pitch, crop, dst, rt, fourcc, planes, norm)
    (1080, 608, 1088, (0, 0, 0, 0), (0, 0), 0, 1, 1, 0),             # NV12ToRGB24 merged u8, native size
    (1080, 608, 1088, (0, 0, 0, 0), (540, 304), 0, 2, 0, 0),         # BGR24 planar, NEAREST 540x304
    (1080, 608, 1088, (480, 340, 1080, 608), (480, 320), 1, 1, 0, 0),  # crop + BILINEAR (VPPTests.cpp:292-298 shape)
    (1920, 1080, 2048, (0, 0, 0, 0), (1280, 720), 3, 2, 0, 1),       # headline with AREA, fp32
    (640, 360, 640, (0, 0, 0, 0), (320, 240), 2, 1, 1, 1),           # BICUBIC merged fp32
]


@pytest.mark.parametrize("case", CASES)
def test_cpp_videoprocessor_convert(tmp_path, oracle, case):
    assert os.path.exists(CLI), "vpp_cli not built (python -c 'import __graft_entry__ as g; g.build()')"
    w, h, pitch, crop, dst, rt, fcc, planes, norm = case
    y, uv = synth_nv12(w, h, seed=w + rt, pitch=pitch)
    src = tmp_path / "in.nv12"
    with open(src, "wb") as f:
        f.write(y.tobytes())
        f.write(uv.tobytes())
    out = tmp_path / "out.bin"
    args = [CLI, str(src), w, h, pitch, *crop, *dst, rt, fcc, planes, norm, str(out)]
    r = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    ref, ow, oh = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fcc, planes=planes,
                                 normalization=bool(norm), nthreads=8, width=w)
    # the refused third consumer makes CHECK_STATUS print its two "TID: ..." lines first (reference behaviour)
    assert "Error status != 0, status: -3" in r.stdout
    tok = [l for l in r.stdout.splitlines() if l.startswith("ok ")][-1].split()
    assert tok[0] == "ok" and (int(tok[1]), int(tok[2])) == (ow, oh)
    # consumer pool of 2: second name accepted, third refused with VREADER_ERROR; input frame was unref'ed
    assert "second=0" in r.stdout and "third=-3" in r.stdout and "input_unref=1" in r.stdout
    # VideoProcessor::Release (round 6): the result handed back is reused by the next Convert of that size, same bytes; foreign / double releases are refused
    assert all(t in r.stdout for t in ("release=0", "reuse=1", "same=1", "foreign=-3", "twice=-3", "again=0")), r.stdout
    got = np.fromfile(out, dtype=np.uint8)
    assert np.array_equal(got, ref.view(np.uint8))
