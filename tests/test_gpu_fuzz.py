"""GPU: seeded differential fuzzing of the whole request space against the oracle -- random source sizes, pitches, crops
(odd origins included), output sizes, resize types, FourCCs, layouts and element types.  Small frames (the oracle finishes
in milliseconds) so that every kernel family, every edge rule and every launch-geometry corner (partial tiles, widths 4 k + 2,
one-row tiles, up- and down-scales, dyadic and non-dyadic ratios) is hit many times; bit-exact or fail."""
import numpy as np
import pytest
import torch

from util import synth_nv12

pytestmark = pytest.mark.gpu


def random_case(rng):
    w = int(rng.integers(1, 161)) * 2
    h = int(rng.integers(1, 91)) * 2
    pitch = w + int(rng.choice([0, 0, 1, 2, 6, 16, 30]))
    crop = (0, 0, 0, 0)
    sw, sh = w, h
    if rng.random() < 0.35 and w >= 8 and h >= 8:
        cw = int(rng.integers(1, w // 2)) * 2
        ch = int(rng.integers(1, h // 2)) * 2
        l = int(rng.integers(0, w - cw + 1))
        t = int(rng.integers(0, h - ch + 1))
        if cw < w and ch < h:
            crop, sw, sh = (l, t, l + cw, t + ch), cw, ch
    kind = rng.random()
    if kind < 0.15:
        dst = (0, 0)
    elif kind < 0.55:  # dyadic-friendly ratios: the integer fast paths
        num, den = [(3, 2), (2, 1), (5, 2), (4, 1), (1, 2), (5, 4), (3, 4), (6, 1), (3, 1), (8, 1), (5, 1), (9, 4)][int(rng.integers(0, 12))]
        dst = (max(2, sw * den // num // 2 * 2), max(2, sh * den // num // 2 * 2))
    else:
        dst = (int(rng.integers(1, 200)) * 2, int(rng.integers(1, 120)) * 2)
    rt = int(rng.integers(0, 4))
    fourcc = int(rng.choice([1, 1, 2, 2, 0, 3, 4, 5, 6]))
    planes = int(rng.integers(0, 2))
    norm = bool(rng.integers(0, 2)) or fourcc == 6
    return w, h, pitch, crop, dst, rt, fourcc, planes, norm


@pytest.mark.parametrize("chunk", range(8))
def test_random_requests_match_the_oracle(vpp, oracle, chunk):
    import tensor_stream as ts
    rng = np.random.default_rng(20260925 + chunk)
    for k in range(40):
        w, h, pitch, crop, dst, rt, fourcc, planes, norm = random_case(rng)
        y, uv = synth_nv12(w, h, seed=1000 * chunk + k, pitch=pitch)
        fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
        try:
            ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=4, width=w)
        except RuntimeError:
            continue  # a request the reference leaves undefined (e.g. an AREA pattern that never terminates): not compared
        got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
        torch.cuda.synchronize()
        g = got.cpu().numpy().ravel()
        assert g.size == ref.size, (w, h, pitch, crop, dst, rt, fourcc, planes, norm)
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, ((w, h, pitch, crop, dst, rt, fourcc, planes, norm), ts.describe(fp, w, h, pitch=pitch, n_frames=1)["kernel"], bad[:6], bad.size)
