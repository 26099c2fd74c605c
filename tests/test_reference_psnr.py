"""The reference's PSNR known-answers (tests/src/VPPTests.cpp:673-911) replayed through the oracle -- the only
numbers the reference's own test-suite holds for BILINEAR / BICUBIC / AREA (and NEAREST) resizing that can be
evaluated without its H.264 decoder: two JPEGs, NEAREST to 720x480, then down to 480x360 (or up to 1920x1080) and
back with the resize type under test, PSNR against the 720x480 RGB24 image, 16 literals with EXPECT_NEAR(.., 0.01).

Inputs: tests/golden/psnr_inputs.npz, produced by tests/golden/make_psnr_inputs.py from the reference's JPEGs with a
plain baseline decoder (exact IDCT).  The reference decodes with NVDEC, whose integer IDCT differs by one LSB in a few
per cent of the samples, hence the tolerance of 0.015 dB here instead of 0.01 (measured deviations: 0.002 .. 0.010).
The GPU variant replays the same chain through libtsvpp.so and must reproduce the oracle's bytes."""
import os

import numpy as np
import pytest

NEAREST, BILINEAR, BICUBIC, AREA = range(4)
RGB24, NV12 = 1, 3
EXPECTED = {  # reference tests/src/VPPTests.cpp:690-911
    "forest": {"down": [14.15, 19.51, 20.81, 19.95], "up": [14.15, 28.00, 43.08, 30.14]},
    "tv_template": {"down": [19.14, 26.07, 25.80, 25.89], "up": [19.14, 39.27, 30.45, 39.34]},
}
SIZES = {"down": (480, 360), "up": (1920, 1080)}
DW, DH = 720, 480


@pytest.fixture(scope="module")
def inputs():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psnr_inputs.npz"))
    return {k: z[k] for k in z.files}


def check_psnr(ref, inp, w, h):
    """checkPSNR of the reference (tests/src/VPPTests.cpp:515-540), including its row stride of `width` bytes."""
    idx = np.arange(h)[:, None] * w + np.arange(0, 3 * w, 3)[None, :]
    mse = 0.0
    for c in range(3):
        d = ref[idx + c].astype(np.float64) - inp[idx + c].astype(np.float64)
        mse += (d * d).sum() / (h * w)
    return 10 * np.log10(255.0 ** 2 / (mse / 3))


def split_nv12(flat, w, h):
    return flat[: w * h].reshape(h, w), flat[w * h:].reshape(h // 2, w)


def chain(convert, y, uv, rt, rw, rh):
    """calculatePSNR of the reference (tests/src/VPPTests.cpp:588-671): four Convert calls."""
    source = convert(y, uv, (DW, DH), NEAREST, NV12)
    converted = convert(y, uv, (DW, DH), NEAREST, RGB24)
    scaled = convert(*split_nv12(source, DW, DH), (rw, rh), rt, NV12)
    rescaled = convert(*split_nv12(scaled, rw, rh), (DW, DH), rt, RGB24)
    return converted, rescaled


@pytest.mark.parametrize("name", ["forest", "tv_template"])
@pytest.mark.parametrize("kind", ["down", "up"])
@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_oracle_reproduces_reference_psnr_literals(oracle, inputs, name, kind, rt):
    def convert(y, uv, dst, resize, fourcc):
        return oracle.convert(y, uv, dst=dst, resize_type=resize, fourcc=fourcc, nthreads=8)[0]

    converted, rescaled = chain(convert, inputs[name + "_y"], inputs[name + "_uv"], rt, *SIZES[kind])
    psnr = check_psnr(converted, rescaled, DW, DH)
    assert abs(psnr - EXPECTED[name][kind][rt]) <= 0.015, (name, kind, rt, psnr)


def test_reference_orderings_hold(oracle, inputs):
    """ASSERT_GT(psnrBilinear, psnrNearest) of PSNRTVTemplateRGBDownscaledComparison (tests/src/VPPTests.cpp:673-688)."""
    def convert(y, uv, dst, resize, fourcc):
        return oracle.convert(y, uv, dst=dst, resize_type=resize, fourcc=fourcc, nthreads=8)[0]

    y, uv = inputs["tv_template_y"], inputs["tv_template_uv"]
    p = [check_psnr(*chain(convert, y, uv, rt, 480, 360), DW, DH) for rt in (NEAREST, BILINEAR)]
    assert p[1] > p[0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["forest", "tv_template"])
@pytest.mark.parametrize("kind", ["down", "up"])
@pytest.mark.parametrize("rt", [NEAREST, BILINEAR, BICUBIC, AREA])
def test_hip_chain_is_byte_identical_and_hits_the_literals(vpp, oracle, inputs, name, kind, rt):
    import torch
    import tensor_stream as ts

    def hip(y, uv, dst, resize, fourcc):
        fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=resize, pixel_format=fourcc, planes_pos=ts.Planes.MERGED)
        out = vpp.Convert(torch.from_numpy(np.ascontiguousarray(y)).cuda(), torch.from_numpy(np.ascontiguousarray(uv)).cuda(), fp)
        torch.cuda.synchronize()
        return out.cpu().numpy().ravel()

    def ref(y, uv, dst, resize, fourcc):
        return oracle.convert(y, uv, dst=dst, resize_type=resize, fourcc=fourcc, nthreads=8)[0]

    y, uv = inputs[name + "_y"], inputs[name + "_uv"]
    got = chain(hip, y, uv, rt, *SIZES[kind])
    want = chain(ref, y, uv, rt, *SIZES[kind])
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert abs(check_psnr(got[0], got[1], DW, DH) - EXPECTED[name][kind][rt]) <= 0.015
