"""CPU: the arithmetic claim behind the streaming kernel for the exact ratios 3 : 2 and 2 : 1 (vpp_bilinear_r32.hip): at these ratios
the taps of output index j sit at fixed positions (3 : 2 -> 3 (j / 2) + (j & 1), 2 : 1 -> 2 j), the weights are the constants of the
kernel's r32_wfirst / r32_wsecond, and the reference's value -- BILINEAR (float blend, truncated), AREA (float sums / weight sum,
truncated), NEAREST -- equals the integer expression the kernel evaluates with v_dot4_u32_u8:
    BILINEAR ( sum tap (16 wx)(16 wy) ) >> 8      AREA floor( sum tap wx wy / S )      NEAREST the first tap.
A numpy model of exactly that expression is compared with the oracle's resized NV12 (luma and interleaved chroma), bit for bit."""
import numpy as np
import pytest

from util import synth_nv12

NEAREST, BILINEAR, AREA = 0, 1, 3
NV12 = 3


def weights(kind, p2, odd):
    """(first, second) integer tap weights of an output index along one axis: the constants of r32_wfirst / r32_wsecond."""
    if p2 == 4:
        return {BILINEAR: (8, 8), AREA: (1, 1), NEAREST: (1, 0)}[kind]
    return {BILINEAR: ((4, 12) if odd else (12, 4)), AREA: ((1, 2) if odd else (2, 1)), NEAREST: (1, 0)}[kind]


def first_tap(p2, c):
    return 2 * c if p2 == 4 else 3 * (c >> 1) + (c & 1)


def model_plane(src, n_out_rows, n_out_cols, kind, p2, step):
    """src: 2-D uint8 plane (chroma: interleaved bytes, step 2 and n_out_cols counts PAIRS); returns the resized plane."""
    out = np.zeros((n_out_rows, n_out_cols * step), np.uint8)
    s = src.astype(np.int64)
    for i in range(n_out_rows):
        wy = weights(kind, p2, i & 1)
        r0 = first_tap(p2, i)
        for j in range(n_out_cols):
            wx = weights(kind, p2, j & 1)
            c0 = first_tap(p2, j)
            for comp in range(step):
                b0 = step * c0 + comp
                acc = 0
                for a, wya in enumerate(wy):
                    for b, wxb in enumerate(wx):
                        if wya * wxb:
                            acc += int(s[r0 + a, b0 + step * b]) * wya * wxb
                            assert wya * wxb < 256          # fits a byte weight of v_dot4_u32_u8
                if kind == BILINEAR:
                    v = acc >> 8
                elif kind == AREA:
                    S = sum(wy) * sum(wx)
                    v = int(np.trunc((np.float32(acc) + np.float32(0.5)) * np.float32(1.0 / S)))   # area_quot
                    assert v == acc // S
                else:
                    v = acc
                assert 0 <= v <= 255
                out[i, step * j + comp] = v
    return out


@pytest.mark.parametrize("p2", [3, 4])
@pytest.mark.parametrize("kind", [BILINEAR, AREA, NEAREST])
def test_streaming_expression_equals_the_reference(oracle, p2, kind):
    dw, dh = 48, 24
    w, h = dw * p2 // 2, dh * p2 // 2
    for seed in range(3):
        y, uv = synth_nv12(w, h, seed=100 * p2 + 10 * kind + seed)
        if seed == 2:   # extremes
            y[:] = np.where(np.random.default_rng(seed).random(y.shape) < 0.5, 0, 255).astype(np.uint8)
        ref, ow, oh = oracle.convert(y, uv, dst=(dw, dh), resize_type=kind, fourcc=NV12, planes=1, normalization=False, nthreads=1)
        assert (ow, oh) == (dw, dh)
        ref = ref.reshape(dh * 3 // 2, dw)
        got_y = model_plane(y, dh, dw, kind, p2, 1)
        got_uv = model_plane(uv, dh // 2, dw // 2, kind, p2, 2)
        assert np.array_equal(got_y, ref[:dh]), (p2, kind, seed, "luma")
        assert np.array_equal(got_uv, ref[dh:]), (p2, kind, seed, "chroma")
