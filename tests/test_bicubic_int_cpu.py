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


def test_column_kernel_scheme_quantised_sums_tie_zone_and_exact_indices():
    """The arithmetic claims behind vpp_bicubic_cols.hip, for ANY weight: coefficients quantised to C = rint(c 2^22); the integer sum
    differs from 2^22 x the fp64 sum by at most 510 units, so outside a zone of 560 units around a rounding tie (S + 2^21) >> 22 IS the
    reference's value; and an index whose coefficients are exact in 2^-22 units (weights that are multiples of 1/16) needs no zone at
    all -- including true ties, which are COMMON there (w = 1/2: one sum in 32), the reason such indices are exempt (round 3)."""
    rng = np.random.default_rng(5)
    p = rng.integers(0, 256, (300000, 4)).astype(np.int64)
    p[:8] = [[255, 0, 0, 255], [0, 255, 255, 0], [255, 255, 255, 255], [0, 0, 0, 0], [3, 77, 200, 19], [1, 2, 3, 4], [255, 254, 1, 0], [128, 127, 129, 126]]
    weights = [np.float32(x) for x in (1 / 6, 5 / 6, 0.5, 0.25, 0.3125, 0.4296875, 0.6889, 0.0123, 0.999, 1 / 3, 0.625)]
    ties_at_half = 0
    for w in weights:
        c = coeffs(np.float64(w))
        C = np.rint(c * 4194304.0).astype(np.int64)
        exact = bool(np.all(c * 4194304.0 == np.rint(c * 4194304.0)))
        assert exact == (float(w) * 16 == np.floor(float(w) * 16)), w       # exact <=> the weight is a multiple of 1/16
        assert C[0] <= 0 <= C[1] and C[3] <= 0 <= C[2]                      # fixed signs: taps 0 and 3 are applied complemented
        S = (C[None, :] * p).sum(axis=1)
        val = np.clip((S + (1 << 21)) >> 22, 0, 255)
        ref = ref_cubic(c, p.astype(np.float64))
        exact_scaled = (c[None, :] * p).sum(axis=1) * 4194304.0
        assert np.abs(S - exact_scaled).max() <= 510                        # the error bound the zone is sized for
        key = (S + (1 << 21) + 560) % (1 << 22)                             # distance-to-tie key of the kernel (the bias holds the 2^21)
        in_zone = key < 1120
        assert np.array_equal(val[~in_zone], ref[~in_zone]), w              # outside the zone the integer value is the reference's
        if exact:
            assert np.array_equal(val, ref), w                              # exact coefficients: everywhere, true ties included
            if float(w) == 0.5:
                ties_at_half = int(((S + (1 << 21)) % (1 << 22) == 0).sum())
    assert ties_at_half > p.shape[0] // 64                                  # ~1 in 32 sums at w = 1/2 is a true tie
