"""CPU: the thread-tile code of the streaming BILINEAR 1 : 2 kernel (tensor-stream_amd/csrc/vpp_bilinear_up2_core.h), compiled for the host with its
hardware operations emulated (tests/host/bilinear_up2_host.cpp), against the oracle on whole frames -- every byte mask, row pairing and edge rule of the
kernel is checked here before a GPU runs it; the GPU suite then checks the same code on the device."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = tmp_path_factory.mktemp("up2") / "libup2_host.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(so), os.path.join(ROOT, "tests", "host", "bilinear_up2_host.cpp")])
    L = ctypes.CDLL(str(so))
    L.bilinear_up2_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    return L


def test_sixteenths_are_exact():
    """w in {1/4, 3/4} on both axes: the reference's float sum of the four weighted taps is exact, so (int) of it is floor(S / 16) with integer S."""
    rng = np.random.default_rng(1)
    p = rng.integers(0, 256, (200000, 4)).astype(np.float32)
    p[:3] = [[255] * 4, [0] * 4, [255, 0, 0, 255]]
    for wx in (0.25, 0.75):
        for wy in (0.25, 0.75):
            wx_, wy_ = np.float32(wx), np.float32(wy)
            omx, omy = np.float32(1) - wx_, np.float32(1) - wy_
            s = (p[:, 0] * omx) * omy + (p[:, 1] * wx_) * omy + (p[:, 2] * wy_) * omx + p[:, 3] * (wx_ * wy_)   # every product and sum exact in fp32
            kx, ky = int(wx * 4), int(wy * 4)
            S = (p[:, 0] * (4 - kx) + p[:, 1] * kx) * (4 - ky) + (p[:, 2] * (4 - kx) + p[:, 3] * kx) * ky
            assert np.array_equal(s * 16, S) and np.array_equal(s.astype(np.int32), S.astype(np.int64) >> 4)


CASES = [(4, 2), (8, 2), (4, 4), (12, 6), (16, 8), (48, 24), (100, 36), (960, 540), (1920, 1080), (1280, 720)]


@pytest.mark.parametrize("w,h", CASES)
@pytest.mark.parametrize("pitch_pad", [0, 20])
def test_host_build_of_the_thread_tile_equals_the_oracle(host, oracle, w, h, pitch_pad):
    rng = np.random.default_rng(w * 31 + h)
    pitch = w + pitch_pad
    y = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    uv = rng.integers(0, 256, (h // 2, pitch), dtype=np.uint8)
    if w >= 16 and h >= 8:  # extremes at the borders and in the interior
        y[:, :3], y[:, w - 3:w], y[:2], y[h - 2:] = 255, 0, 0, 255
        y[h // 2, ::2], y[h // 2, 1::2] = 0, 255
        uv[:, :4], uv[:, w - 4:w] = 0, 255
    dw, dh = 2 * w, 2 * h
    out = np.zeros(dw * dh * 3 // 2, dtype=np.uint8)
    assert host.bilinear_up2_host(y.ctypes.data, uv.ctypes.data, pitch, pitch, w, h, out.ctypes.data) == 0
    ref, ow, oh = oracle.convert(y, uv, dst=(dw, dh), resize_type=oracle.BILINEAR, fourcc=oracle.NV12, planes=oracle.MERGED, normalization=False, nthreads=8, width=w)
    assert (ow, oh) == (dw, dh)
    bad = np.flatnonzero(out != ref)
    assert bad.size == 0, f"{bad.size} bytes differ, first at {bad[:8]} (luma plane has {dw * dh} bytes, width {dw})"
