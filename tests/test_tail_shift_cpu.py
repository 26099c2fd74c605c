"""Host logic (no GPU): which requests end in a shifted tile column instead of a row tail and its second launch (tsvpp_describe: tail = 2 / 1 / 0)."""
import os
import subprocess
import sys

import tensor_stream as ts

NEAREST, BILINEAR, BICUBIC, AREA = 0, 1, 2, 3


def tail(dst, rt, src=(1920, 1080), pitch=2048, fourcc=2, planes=0, norm=True, n=64):
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    return ts.describe(fp, src[0], src[1], pitch=pitch, n_frames=n)["tail"]


def test_which_requests_shift():
    for rt in (NEAREST, BILINEAR, BICUBIC, AREA):
        for dst in ((854, 480), (1366, 768), (270, 270)):
            assert tail(dst, rt) == 2, (dst, rt)
            assert tail(dst, rt, fourcc=1, planes=1, norm=False) == 2, (dst, rt)
        assert tail((852, 480), rt) == 0                       # 4 k columns: nothing to do
        assert tail((54, 30), rt) == 1                         # narrower than one tile: the tail launch
    assert tail((0, 0), NEAREST, src=(854, 480), pitch=854) == 1        # colour only: aligned dword reads of the planes
    assert tail((854, 480), AREA, src=(1708, 960), pitch=1712) == 1     # the box kernel: aligned runs at 4-column granularity


def test_knob_keeps_the_tail_launch():
    code = ("import tensor_stream as ts; fp = ts.FrameParameters(width=854, height=480, resize_type=1, pixel_format=2, planes_pos=0, normalization=True); "
            "print(ts.describe(fp, 1920, 1080, pitch=2048, n_frames=64)['tail'])")
    env = dict(os.environ, TSVPP_TAIL_SHIFT="0", PYTHONPATH=os.pathsep.join(sys.path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert out == "1", out
