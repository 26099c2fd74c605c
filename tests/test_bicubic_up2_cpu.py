"""CPU: the thread-tile code of the streaming BICUBIC 1 : 2 kernel (tensor-stream_amd/csrc/vpp_bicubic_up2_core.h), compiled for the host with its
hardware operations emulated (tests/host/bicubic_up2_host.cpp), against the oracle on whole frames -- every byte mask, window position and edge rule of the
kernel is checked here before a GPU runs it; the GPU suite then checks the same code on the device."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = tmp_path_factory.mktemp("bup2") / "libbup2_host.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(so), os.path.join(ROOT, "tests", "host", "bicubic_up2_host.cpp")])
    L = ctypes.CDLL(str(so))
    L.bicubic_up2_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    return L


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
    assert host.bicubic_up2_host(y.ctypes.data, uv.ctypes.data, pitch, pitch, w, h, out.ctypes.data) == 0
    ref, ow, oh = oracle.convert(y, uv, dst=(dw, dh), resize_type=oracle.BICUBIC, fourcc=oracle.NV12, planes=oracle.MERGED, normalization=False, nthreads=8, width=w)
    assert (ow, oh) == (dw, dh)
    bad = np.flatnonzero(out != ref)
    assert bad.size == 0, f"{bad.size} bytes differ, first at {bad[:8]} (luma plane has {dw * dh} bytes, width {dw})"
