"""CPU: the thread-tile code of the streaming BICUBIC kernel (tensor-stream_amd/csrc/vpp_bicubic_r32_core.h), compiled for the host with its four
hardware operations emulated (tests/host/bicubic_r32_host.cpp), against the oracle on whole frames -- every byte mask, window position and edge rule
of the kernel (ratios 3 : 2 and 2 : 1) is checked here before a GPU runs it; the GPU suite then checks the same code on the device."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = tmp_path_factory.mktemp("bcr32") / "libbcr32_host.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(so), os.path.join(ROOT, "tests", "host", "bicubic_r32_host.cpp")])
    L = ctypes.CDLL(str(so))
    L.bicubic_r32_host.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    return L


def coeffs(w):
    a = -0.75
    w2, w3 = w * w, w * w * w
    return np.array([(a * w - 2 * a * w2) + a * w3, (1 - (a + 3) * w2) + (a + 2) * w3, ((-a) * w + (2 * a + 3) * w2) - (a + 2) * w3, a * w2 - a * w3])


def test_byte_coefficients_and_the_complement_form():
    """w = 1/4, 3/4, 1/2: 256 c_t are integers whose magnitudes fit a byte, and  128 - 255 sum|C_neg| + sum_pos C p + sum_neg |C| (255 - p)  ==  S + 128."""
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, (100000, 4)).astype(np.int64)
    p[:4] = [[255, 0, 0, 255], [0, 255, 255, 0], [255] * 4, [0] * 4]
    for w, want in ((0.25, (-27, 225, 67, -9)), (0.75, (-9, 67, 225, -27)), (0.5, (-24, 152, 152, -24))):
        c = coeffs(w)
        C = np.rint(c * 256).astype(np.int64)
        assert tuple(C) == want and np.array_equal(C / 256.0, c) and np.abs(C).max() <= 255
        neg = np.where(C < 0, -C, 0)
        pos = np.where(C > 0, C, 0)
        acc = 128 - 255 * neg.sum() + (pos[None] * p).sum(1) + (neg[None] * (255 - p)).sum(1)
        assert np.array_equal(acc, (C[None] * p).sum(1) + 128)
        s = (c[None] * p).sum(1)
        ref = np.clip(np.where(s >= 0, np.floor(s + 0.5), np.ceil(s - 0.5)), 0, 255)  # round(): half away from zero
        assert np.array_equal(np.clip(acc >> 8, 0, 255), ref)


CASES = [(3, 12, 6), (3, 24, 12), (4, 16, 8), (3, 48, 24), (3, 96, 36), (3, 1080 // 2 * 2 * 12 // 12 + 12, 60), (3, 1920, 1080), (3, 1248, 480), (4, 32, 16), (4, 64, 40), (4, 1920, 1080), (4, 3840 // 2, 2160 // 2)]


@pytest.mark.parametrize("p2,w,h", CASES)
@pytest.mark.parametrize("pitch_pad", [0, 20])
def test_host_build_of_the_thread_tile_equals_the_oracle(host, oracle, p2, w, h, pitch_pad):
    if (w * 2 // p2) % 8 or (h * 2 // p2) % 4 or (w * 2) % p2 or (h * 2) % p2:
        pytest.skip("not a geometry of this kernel")
    rng = np.random.default_rng(w * 31 + h + p2)
    pitch = w + pitch_pad
    y = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    uv = rng.integers(0, 256, (h // 2, pitch), dtype=np.uint8)
    # extremes at the borders and in the interior: the clamps of every 4-tap sum fire in both directions
    y[:, :3], y[:, w - 3:w], y[:2], y[h - 2:] = 255, 0, 0, 255
    y[h // 2, ::2], y[h // 2, 1::2] = 0, 255
    uv[:, :4], uv[:, w - 4:w] = 0, 255
    dw, dh = w * 2 // p2, h * 2 // p2
    out = np.zeros(dw * dh * 3 // 2, dtype=np.uint8)
    assert host.bicubic_r32_host(p2, y.ctypes.data, uv.ctypes.data, pitch, pitch, w, h, out.ctypes.data) == 0
    ref, ow, oh = oracle.convert(y, uv, dst=(dw, dh), resize_type=oracle.BICUBIC, fourcc=oracle.NV12, planes=oracle.MERGED, normalization=False, nthreads=8, width=w)
    assert (ow, oh) == (dw, dh)
    bad = np.flatnonzero(out != ref)
    assert bad.size == 0, f"{bad.size} bytes differ, first at {bad[:8]} (luma plane has {dw * dh} bytes, width {dw})"
