"""CPU: the arithmetic claim behind the integer BICUBIC kernel (tensor-stream_amd/csrc/vpp_bicubic_int.hip).
For weights w = k/16 the reference's fp64 4-tap sum (src/Resize.cu:45-62: Keys a = -0.75, round half away from zero,
clamp to a byte) equals clamp((sum_t C_t p_t + 8192) >> 14, 0, 255) with C_t = 16384 c_t -- exactly, for every tap
combination, including the edge rule expressed by adding collapsed taps' coefficients to the centre's."""
import numpy as np


def coeffs(w):
    a = -0.75
    w2, w3 = w * w, w * w * w
    return np.array([(a * w - 2 * a * w2) + a * w3, (1 - (a + 3) * w2) + (a + 2) * w3, ((-a) * w + (2 * a + 3) * w2) - (a + 2) * w3, a * w2 - a * w3])


def ref_cubic(c, p):
    s = ((c[0] * p[:, 0] + c[1] * p[:, 1]) + c[2] * p[:, 2]) + c[3] * p[:, 3]
    r = np.where(s >= 0, np.floor(s + 0.5), np.ceil(s - 0.5))  # round(): half away from zero
    return np.clip(r, 0, 255).astype(np.int64)


def int_cubic(C, p):
    s = (C[None, :] * p).sum(axis=1) + 8192
    return np.clip(s >> 14, 0, 255)


def test_integer_form_equals_fp64_form_for_all_sixteenth_weights():
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, (200000, 4)).astype(np.int64)
    p[:64] = np.array([[255, 0, 0, 255], [0, 255, 255, 0], [255, 255, 255, 255], [0, 0, 0, 0]] * 16)
    for k in range(16):
        c = coeffs(k / 16.0)
        C = (c * 16384.0).astype(np.int64)
        assert np.array_equal(C / 16384.0, c), k            # the coefficients ARE multiples of 2^-14
        assert C.sum() == 16384 and np.abs(C).max() <= 16384
        assert np.array_equal(int_cubic(C, p), ref_cubic(c, p.astype(np.float64))), k
        # fp32 evaluation of the polynomials (what the kernel's table build does) is exact as well
        w = np.float32(k / 16.0)
        a = np.float32(-0.75)
        w2 = w * w
        w3 = w2 * w
        c32 = np.array([(a * w - (2 * a) * w2) + a * w3, (1 - (a + 3) * w2) + (a + 2) * w3, ((-a) * w + (2 * a + 3) * w2) - (a + 2) * w3, a * w2 - a * w3],
                       dtype=np.float32)
        assert np.array_equal(c32.astype(np.float64), c), k


def test_edge_rule_as_weight_folding():
    """Taps (p - lo, p, p + hi, p + 2 hi) with lo, hi in {0, 1} == the window [p - lo, p - lo + 3] with folded weights."""
    rng = np.random.default_rng(1)
    row = rng.integers(0, 256, 64).astype(np.int64)
    for k in (0, 3, 4, 8, 12, 15):
        c = coeffs(k / 16.0)
        C = (c * 16384.0).astype(np.int64)
        for lo in (0, 1):
            for hi in (0, 1):
                p = 10
                taps = np.array([[row[p - lo], row[p], row[p + hi], row[p + 2 * hi]]])
                want = ref_cubic(c, taps.astype(np.float64))[0]
                m1, m2, m3 = (C[1], C[2], C[3]) if hi else (C[1] + C[2] + C[3], 0, 0)
                wg = np.array([C[0], m1, m2, m3] if lo else [C[0] + m1, m2, m3, 0])
                ws = p - 1 if lo else p
                assert np.abs(wg).max() < 32768                       # int16 operands of v_dot2_i32_i16
                got = int_cubic(wg, row[None, ws:ws + 4])[0]
                assert got == want, (k, lo, hi)
