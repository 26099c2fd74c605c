"""CPU: the arithmetic claim behind the integer thread tile of the 2x2-tap kernel (vpp_kernels.hip,
bilinear_int_thread_tile): for weights wx = kx/16, wy = ky/16 the reference's fp32 blend (src/Resize.cu:17-23), evaluated
operation by operation and truncated, equals ((A wx0 + B wx1) wy0 + (C wx0 + D wx1) wy1) >> 8 with wx0 = 16 - kx, wx1 = kx."""
import numpy as np


def test_integer_blend_equals_fp32_blend_for_all_sixteenth_weights():
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, (50000, 4)).astype(np.int64)
    p[:4] = [[255, 255, 255, 255], [0, 0, 0, 0], [255, 0, 0, 255], [0, 255, 255, 0]]
    A, B, C, D = (p[:, i].astype(np.float32) for i in range(4))
    f = np.float32
    for kx in range(17):       # 16/16 never occurs as a weight (w < 1) but the identity holds there too
        for ky in range(17):
            wx, wy = f(kx / 16.0), f(ky / 16.0)
            omx, omy = f(1) - wx, f(1) - wy
            s = ((A * omx) * omy + (B * wx) * omy)
            s = s + (C * wy) * omx
            s = s + D * (wx * wy)
            ref = s.astype(np.int64)  # (int): truncation
            top = p[:, 0] * (16 - kx) + p[:, 1] * kx
            bot = p[:, 2] * (16 - kx) + p[:, 3] * kx
            assert top.max() < 4096 and bot.max() < 4096      # operands of v_dot2_u32_u16
            got = (top * (16 - ky) + bot * ky) >> 8
            assert got.max() <= 255
            assert np.array_equal(got, ref), (kx, ky)
