"""CPU: properties of the oracle that follow from the reference's source text, independent of any GPU.
They guard the restatement itself."""
import numpy as np
import pytest

from util import coverage_frame, synth_nv12


def test_crop_pixel_equality_like_reference_test(oracle):
    """reference tests/src/VPPTests.cpp:46-99 (checkCropCorrectness): cropped NV12 == window of the input,
    chroma columns starting at (left & ~1) ... for EVEN left; odd left shifts the chroma bytes by one."""
    y, uv = synth_nv12(1080, 608, seed=1)
    for (l, t, r, b) in [(0, 0, 320, 240), (120, 64, 600, 400), (480, 340, 1080, 608)]:
        oy, ouv = oracle.crop_stage(y, uv, l, t, r, b)
        assert np.array_equal(oy, y[t:b, l:r])
        assert np.array_equal(ouv, uv[t // 2: t // 2 + (b - t) // 2, l:r])
    oy, ouv = oracle.crop_stage(y, uv, 121, 65, 601, 401)   # odd origin: chroma row floor(65/2), byte offset 121
    assert np.array_equal(oy, y[65:401, 121:601])
    assert np.array_equal(ouv, uv[32:32 + 168, 121:601])     # U slot now holds a V byte: the reference's quirk


def test_nearest_is_index_math(oracle):
    y, uv = synth_nv12(1080, 608, seed=2)
    dw, dh = 480, 360
    oy, ouv = oracle.resize_stage(y, uv, dw, dh, oracle.NEAREST)
    xr, yr = np.float32(1080) / np.float32(dw), np.float32(608) / np.float32(dh)
    xs = (xr * np.arange(dw, dtype=np.float32)).astype(np.int32)
    ys = (yr * np.arange(dh, dtype=np.float32)).astype(np.int32)
    assert np.array_equal(oy, y[np.ix_(ys, xs)])
    cx, cy = xs[: dw // 2], ys[: dh // 2]       # chroma uses the SAME formula on chroma-grid indices
    assert np.array_equal(ouv[:, 0::2], uv[np.ix_(cy, 2 * cx)])
    assert np.array_equal(ouv[:, 1::2], uv[np.ix_(cy, 2 * cx + 1)])


def test_bicubic_ratio3_is_point_sample_at_offset_1(oracle):
    """C4: (j+0.5)*3-0.5 = 3j+1 -> weights 0 -> coefficients (0,1,0,0)."""
    y, uv = synth_nv12(960, 540, seed=3)
    oy, ouv = oracle.resize_stage(y, uv, 320, 180, oracle.BICUBIC)
    assert np.array_equal(oy, y[1::3, 1::3])
    assert np.array_equal(ouv[:, 0::2], uv[1::3, 2::6][:90, :160])
    assert np.array_equal(ouv[:, 1::2], uv[1::3, 3::6][:90, :160])


def test_area_ratio6_is_floor_of_box_mean(oracle):
    """C5: integer ratio -> plain 6x6 box, floor(sum/36)."""
    y, uv = synth_nv12(768, 432, seed=4)
    oy, ouv = oracle.resize_stage(y, uv, 128, 72, oracle.AREA)
    box = y.astype(np.int64).reshape(72, 6, 128, 6).sum(axis=(1, 3))
    assert np.array_equal(oy, (box // 36).astype(np.uint8))
    u = uv[:, 0::2].astype(np.int64)   # chroma plane is 216 x 384 pairs; output chroma 36 x 64 pairs
    ubox = u[: 36 * 6, : 64 * 6].reshape(36, 6, 64, 6).sum(axis=(1, 3))
    assert np.array_equal(ouv[:, 0::2], (ubox // 36).astype(np.uint8))


def test_area_ratio_1p5_is_floor_k_over_9(oracle):
    """Headline AREA: rows [1,.5],[.5,1]; quotient k/9 truncated."""
    y, uv = synth_nv12(192, 108, seed=5)
    oy, _ = oracle.resize_stage(y, uv, 128, 72, oracle.AREA)
    Y = y.astype(np.int64)
    want = np.zeros((72, 128), np.int64)
    for i in range(72):
        wy = (1.0, 0.5) if i % 2 == 0 else (0.5, 1.0)
        y0 = int(np.float32(1.5) * np.float32(i))
        for j in range(128):
            wx = (1.0, 0.5) if j % 2 == 0 else (0.5, 1.0)
            x0 = int(np.float32(1.5) * np.float32(j))
            k4 = sum(int(4 * wy[a] * wx[b]) * Y[y0 + a, x0 + b] for a in range(2) for b in range(2))
            want[i, j] = k4 // 9
    assert np.array_equal(oy, want.astype(np.uint8))


def test_bilinear_exact_2x_downscale_is_floor_of_quarter_sum(oracle):
    y, uv = synth_nv12(640, 360, seed=6)
    oy, _ = oracle.resize_stage(y, uv, 320, 180, oracle.BILINEAR)
    s = y.astype(np.int64).reshape(180, 2, 320, 2).sum(axis=(1, 3))
    assert np.array_equal(oy, (s // 4).astype(np.uint8))     # weights .5/.5: exact in fp32, truncation
    ay, _ = oracle.resize_stage(y, uv, 320, 180, oracle.AREA)  # reference CRCs agree too (PythonTests.cpp:200,224)
    assert np.array_equal(ay, oy)


def test_upscale_area_runs_bilinear_variant(oracle):
    y, uv = synth_nv12(320, 180, seed=7)
    a, _ = oracle.resize_stage(y, uv, 640, 360, oracle.AREA)
    assert a.shape == (360, 640)
    assert np.array_equal(a[0::2, 0::2], y)   # fx = (j+1) - (x+1)/0.5 <= 0 at even j -> weight 0 -> source pixel


def _fma32(a, b, c):
    """fma(a, b, c) of float32 operands with ONE rounding, in numpy: every product here is a 24-bit constant times an integer below 2^8 and every addend has
    at most 24 significant bits within 2^-26 .. 2^9 of it, so the float64 product and sum are EXACT and the single conversion to float32 is the fma's rounding."""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def test_colour_matches_independent_numpy_restatement_on_all_triples(oracle):
    """Second, independent restatement of src/ColorConversion.cu:23-38 in numpy float32 ops over all 2^24 (Y,U,V) triples, with the operation tree of the
    reference AS COMPILED (oracle/vpp_oracle.c, CT_NVCC): the chroma terms' single-use products fused, the luma product and the final sums plain.  numpy
    never fuses: the two fused operations are spelt out (_fma32)."""
    y, uv = coverage_frame()
    out, _, _ = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.PLANAR, nthreads=8)
    out = out.reshape(3, 4096, 4096)
    f = np.float32
    Y = y.astype(f)
    U = np.repeat(np.repeat(uv[:, 0::2], 2, 0), 2, 1).astype(f) - f(128)
    V = np.repeat(np.repeat(uv[:, 1::2], 2, 0), 2, 1).astype(f) - f(128)
    half = np.full_like(Y, f(0.5))
    yv = np.maximum(f(0), Y - f(16)) * f(1.163999557)
    R = (yv + _fma32(V, f(1.5959997177), half)).astype(np.int32).clip(0, 255)
    B = (yv + _fma32(U, f(2.017999649), half)).astype(np.int32).clip(0, 255)
    G = (yv + (_fma32(V, f(-0.812999725), -(f(0.390999794) * U)) + f(0.5))).astype(np.int32).clip(0, 255)
    assert np.array_equal(out[0], R.astype(np.uint8))
    assert np.array_equal(out[1], G.astype(np.uint8))
    assert np.array_equal(out[2], B.astype(np.uint8))


def test_libm_pow_and_exact_products_agree_on_uint8_results(oracle):
    """oracle/pow_pin.c: libm pow(w,3) is not correctly rounded; the uint8 bicubic result does not care."""
    L = oracle.lib()
    y, uv = synth_nv12(1080, 608, seed=8)
    a = oracle.resize_stage(y, uv, 480, 360, oracle.BICUBIC, nthreads=8)
    L.vpp_oracle_set_libm_pow(1)
    try:
        b = oracle.resize_stage(y, uv, 480, 360, oracle.BICUBIC, nthreads=8)
    finally:
        L.vpp_oracle_set_libm_pow(0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_threads_do_not_change_results(oracle):
    y, uv = synth_nv12(642, 362, seed=9)
    for rt in range(4):
        a, _, _ = oracle.convert(y, uv, dst=(214, 182), resize_type=rt, nthreads=1)
        b, _, _ = oracle.convert(y, uv, dst=(214, 182), resize_type=rt, nthreads=8)
        assert np.array_equal(a, b)


def test_av_crc_helper_known_answer(oracle):
    """CRC-32/MPEG-2 of "123456789" is 0x0376E6E7; av_crc keeps the state byte-swapped [ext, unverified]."""
    v = oracle.av_crc32_ieee(np.frombuffer(b"123456789", np.uint8))
    assert v == int.from_bytes((0x0376E6E7).to_bytes(4, "big"), "little")


def test_two_op_division_by_255_is_exact_for_sixteenths():
    """The HIP kernels replace the IEEE `x / 255` by e = x*lo; q = fma(x, hi, e) with 1/255 = hi + lo
    (vpp_kernels.hip norm255, vpp_formats.hip div255).  Every value they normalise is an integer or a multiple of
    1/16 in [0, 255]; for all 4081 of them the two-op form equals the correctly rounded quotient (exact rational
    arithmetic, round-to-nearest-even to 24 bits)."""
    from fractions import Fraction as Fr

    def fl32(q):
        if q == 0:
            return Fr(0)
        sign, q, e = (1 if q > 0 else -1), abs(q), 0
        while q / Fr(2) ** e >= 2 ** 24:
            e += 1
        while q / Fr(2) ** e < 2 ** 23:
            e -= 1
        m = q / Fr(2) ** e
        n = m.numerator // m.denominator
        r = m - n
        if r > Fr(1, 2) or (r == Fr(1, 2) and n % 2 == 1):
            n += 1
        return sign * n * Fr(2) ** e

    hi, lo = Fr(float.fromhex("0x1.010102p-8")), Fr(float.fromhex("-0x1.fdfdfep-33"))
    for k in range(255 * 16 + 1):
        x = Fr(k, 16)
        assert fl32(x * hi + fl32(x * lo)) == fl32(x / 255), k


def test_exact_index_mode_equals_the_float_index_below_2pow24(oracle):
    """vpp_oracle_set_exact_index: the integer start index equals the reference's float expression (src/Resize.cu:6)
    wherever that is exact, i.e. for every frame with pitch * height <= 2^24 -- all BASELINE sizes (4K: 3840 * 2160 = 8.3 M)."""
    for (w, h, dst, rt) in [(1920, 1080, (1280, 720), 1), (3840, 2160, (1280, 720), 1), (640, 360, (1280, 720), 3), (1080, 608, (480, 360), 1)]:
        y, uv = synth_nv12(w, h, seed=w + rt)
        a, _, _ = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=oracle.NV12, nthreads=8)
        oracle.set_exact_index(True)
        try:
            b, _, _ = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=oracle.NV12, nthreads=8)
        finally:
            oracle.set_exact_index(False)
        assert np.array_equal(a, b), (w, h, dst, rt)
