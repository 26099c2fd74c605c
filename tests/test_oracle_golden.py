"""CPU: pin the oracle on every golden vector the reference's tests hold for this path
(reference tests/src/VPPTests.cpp:301-512; files re-packed by tests/golden/make_golden.py)."""
import numpy as np
import pytest


@pytest.mark.parametrize("name,fcc", [("RGB24", 1), ("BGR24", 2), ("Y800", 0), ("UYVY", 4), ("YUV444", 5), ("NV12", 3), ("HSV", 6)])
def test_oracle_reproduces_reference_golden_bit_exact(golden, oracle, name, fcc):
    out, ow, oh = oracle.convert(golden["Y"], golden["UVp"], fourcc=fcc, planes=oracle.MERGED, normalization=True)
    assert (ow, oh) == (320, 240)
    ref = golden[name]  # uint32 bit patterns of the reference's fp32 dump
    assert out.dtype == np.float32 and out.size == ref.size
    assert np.array_equal(out.view(np.uint32), ref)


def test_golden_input_is_exact_k_over_255(golden):
    nv = golden["NV12"].view(np.float32)
    k = golden["input_nv12_u8"].astype(np.float32)
    assert np.array_equal((k / np.float32(255)).view(np.uint32), nv.view(np.uint32))
    assert golden["Y"].min() >= 16 and golden["Y"].max() <= 240  # limited-range content


def test_bgr_is_rgb_with_channels_reversed_and_planar_is_transposed(golden, oracle):
    rgb, _, _ = oracle.convert(golden["Y"], golden["UVp"], fourcc=oracle.RGB24, planes=oracle.MERGED)
    bgr, _, _ = oracle.convert(golden["Y"], golden["UVp"], fourcc=oracle.BGR24, planes=oracle.MERGED)
    pl, _, _ = oracle.convert(golden["Y"], golden["UVp"], fourcc=oracle.RGB24, planes=oracle.PLANAR)
    rgb = rgb.reshape(240, 320, 3)
    assert np.array_equal(bgr.reshape(240, 320, 3), rgb[:, :, ::-1])
    assert np.array_equal(pl.reshape(3, 240, 320), rgb.transpose(2, 0, 1))
    # uint8 output is exactly 255 x the normalised golden
    ref = np.rint(golden["RGB24"].view(np.float32) * 255).astype(np.uint8).reshape(240, 320, 3)
    assert np.array_equal(rgb, ref)
