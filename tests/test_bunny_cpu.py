"""CPU: BASELINE.json configs[0] (C1) -- a picture of the reference's own demo clip tests/resources/bunny.mp4 (the clip's second IDR, sample 129: real 1280x720
content; tests/golden/make_bunny_idr.py, tests/golden/bunny_idr129_1280x720.npz) through the oracle: NV12 -> RGB24 MERGED uint8 at native size, the conversion
C1 names.  The reference holds no literal for this picture (its tests use the other clip), so these are consistency properties on real content, not pins:
the pins of the colour path are the seven golden files and the 14 colour CRCs (tests/test_oracle_golden.py, tests/test_reference_crcs.py)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def bunny():
    z = np.load(os.path.join(HERE, "golden", "bunny_idr129_1280x720.npz"))
    return z["y"], z["uv"]


def test_fixture_is_a_picture_of_the_clip(bunny):
    y, uv = bunny
    assert y.shape == (720, 1280) and uv.shape == (360, 1280) and y.dtype == np.uint8
    assert y.std() > 30 and 100 < uv[:, 0::2].mean() < 140 and 100 < uv[:, 1::2].mean() < 150   # sky, hills and trees -- not the blank first IDR
    assert y.min() >= 16 - 16 and y.max() <= 255


def test_c1_conversion_properties(bunny, oracle):
    y, uv = bunny
    rgb, w, h = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.MERGED, normalization=False, nthreads=8)
    assert (w, h) == (1280, 720) and rgb.dtype == np.uint8 and rgb.size == 1280 * 720 * 3
    rgb = rgb.reshape(720, 1280, 3)
    bgr, _, _ = oracle.convert(y, uv, fourcc=oracle.BGR24, planes=oracle.MERGED, normalization=False, nthreads=8)
    assert np.array_equal(bgr.reshape(720, 1280, 3)[..., ::-1], rgb)
    pl, _, _ = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.PLANAR, normalization=False, nthreads=8)
    assert np.array_equal(pl.reshape(3, 720, 1280).transpose(1, 2, 0), rgb)
    f32, _, _ = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.MERGED, normalization=True, nthreads=8)
    assert np.array_equal(f32.reshape(720, 1280, 3), rgb.astype(np.float32) / np.float32(255))
    # the sky is bright and nearly grey, the trees are green: the matrix points the right way on real content
    assert rgb[:200, :400].mean() > 170 and rgb[500:, 900:, 1].mean() > rgb[500:, 900:, 2].mean()
    # the picture never reaches one of the 36 triples on which the colour stage's contraction matters (tests/test_oracle_contract.py)
    oracle.set_contract(1 | 2 | 8 | 16 | 64)
    try:
        plain, _, _ = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.MERGED, normalization=False, nthreads=8)
    finally:
        oracle.set_contract(-1)
    assert np.array_equal(plain.reshape(720, 1280, 3), rgb)
