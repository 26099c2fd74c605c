"""GPU: the streaming kernel for the exact ratios 3 : 2 and 2 : 1 with uint8 outputs (vpp_bilinear_r32_kernel: the whole 2x2 blend as
v_dot4 on source dwords with compile-time byte weights, no LDS) against the oracle, bit for bit: every flavour it takes, partial tile
columns / rows, odd and even partial runs of the merged-output exchange, crops, batches; requests it cannot take fall back."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
NEAREST, BILINEAR, AREA = 0, 1, 3
Y800, RGB24, BGR24, NV12, UYVY, YUV444 = 0, 1, 2, 3, 4, 5


def check(vpp, oracle, y, uv, w, dst, fourcc=RGB24, planes=0, crop=(0, 0, 0, 0), n=1, r32=True, norm=False, rt=BILINEAR):
    import tensor_stream as ts
    from tensor_stream import vpp as V
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if not knob_run():
        k = V.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=n)["kernel"]
        assert k.startswith("vpp_bilinear_r32_kernel") == r32, (k, w, y.shape, dst, crop)
        if r32:
            assert k.endswith("2:1>" if 2 * dst[0] == (crop[2] - crop[0] or w) else "3:2>"), k
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    got = vpp.Convert(ty, tuv, fp, width=w) if n == 1 else vpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    for g in ([got] if n == 1 else [got[0], got[n - 1]]):
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (w, y.shape, dst, rt, fourcc, planes, crop, n, bad[:8], bad.size)


@pytest.mark.parametrize("src,pitch", [((1920, 1080), 2048), ((3840, 2160), 3840), ((960, 540), 960), ((48, 24), 48), ((1944, 1092), 1952),
                                       ((444, 66), 444),     # 37 threads per row: a partial run of 5 lanes (odd: direct stores)
                                       ((456, 66), 460)])    # 38 threads per row: a partial run of 6 lanes (even: exchanged)
@pytest.mark.parametrize("fourcc,planes", [(RGB24, 0), (BGR24, 1), (RGB24, 1), (NV12, 1), (Y800, 1)])
@pytest.mark.parametrize("rt", [BILINEAR, AREA, NEAREST])
def test_r32_sizes_and_flavours(vpp, oracle, src, pitch, fourcc, planes, rt):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + fourcc + planes + rt, pitch=pitch)
    check(vpp, oracle, y, uv, src[0], (src[0] * 2 // 3, src[1] * 2 // 3), fourcc=fourcc, planes=planes, rt=rt)


def test_r32_batches_crops_two_pass_fallbacks(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=31, pitch=2048)
    check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=BGR24, planes=1, n=64)
    check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=RGB24, planes=1, n=64, rt=AREA)
    check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=RGB24, planes=0, n=64, rt=NEAREST)
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(12, 6, 972, 546), rt=AREA, planes=1)
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(12, 6, 972, 546), rt=NEAREST)
    check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=RGB24, planes=0, n=3)
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(4, 2, 964, 542))                       # origin a multiple of 4: pointers stay dword-aligned
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(8, 7, 968, 547), planes=1)
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(6, 2, 966, 542), r32=False)            # misaligned origin: the LDS kernel
    check(vpp, oracle, y, uv, 1920, (640, 360), crop=(5, 3, 965, 543), planes=1, r32=False)  # odd origin (U / V swapped)
    check(vpp, oracle, y, uv, 1920, (1280, 720), norm=True, r32=False)                       # fp32 outputs stay on vpp_bilinear_kernel
    for fcc in (UYVY, YUV444):                                                               # pass 1 of the two-pass formats
        check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=fcc, planes=1)
        check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=fcc, planes=1, norm=True)
    y, uv = synth_nv12(966, 546, seed=32, pitch=976)
    check(vpp, oracle, y, uv, 966, (644, 364), r32=False)                                    # 644 = 8 k + 4
    y, uv = synth_nv12(960, 543 * 2, seed=33)
    check(vpp, oracle, y, uv, 960, (640, 724))                                               # 181 row quads
    y, uv = synth_nv12(960, 546, seed=34)
    check(vpp, oracle, y, uv, 960, (640, 364))                                               # 91 row quads: a partial last tile row
    for val in (0, 255):
        yy = np.full((72, 96), val, np.uint8)
        uu = np.full((36, 96), 255 - val, np.uint8)
        for rt in (BILINEAR, AREA, NEAREST):
            check(vpp, oracle, yy, uu, 96, (64, 48), planes=1, rt=rt)
            check(vpp, oracle, yy, uu, 96, (64, 48), planes=0, rt=rt)


@pytest.mark.parametrize("src,pitch", [((1920, 1080), 2048), ((3840, 2160), 3840), ((48, 24), 48), ((1944, 1092), 1952), ((456, 66), 460), ((96, 12), 96),
                                       ((24, 12), 24), ((48, 48), 64)])   # 8 / 16 output columns: a row's first thread is (next to) its last
@pytest.mark.parametrize("rt", [BILINEAR, AREA, NEAREST])
@pytest.mark.parametrize("ratio", [(3, 2), (2, 1)])
@pytest.mark.parametrize("fmt", [UYVY, YUV444])
def test_r32_uyvy_yuv444_in_one_pass(vpp, oracle, src, pitch, rt, ratio, fmt):
    """UYVY / YUV444 (uint8) behind a 3 : 2 / 2 : 1 resize are outputs of the streaming kernel itself -- no NV12 intermediate, no second pass: the
    vertical (-1, 9, 9, -1) chroma filter of the odd chroma rows (clamped at the last row: sizes with one, two and many tile rows), both
    clamps of the filter (full-range random chroma), every tap kind; YUV444's horizontal filter in the reference's FLAT pair order (a row's first
    pair follows the previous row's last; zeros outside the image; the frame's first and last two odd pixels), its C division and byte wrap."""
    import tensor_stream as ts
    from tensor_stream import vpp as V
    w, h = src
    dst = (w * ratio[1] // ratio[0], h * ratio[1] // ratio[0])
    if dst[0] % 8 or dst[1] % 4 or dst[0] * ratio[0] != w * ratio[1] or dst[1] * ratio[0] != h * ratio[1]:
        pytest.skip("not a size of the streaming kernel")
    y, uv = synth_nv12(w, h, seed=w + rt + ratio[0], pitch=pitch)
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=rt, pixel_format=fmt, planes_pos=1, normalization=False)
    if not knob_run():
        dsc = V.describe(fp, w, h, pitch=pitch)
        assert dsc["out"] == ("uyvy_u8" if fmt == UYVY else "yuv444_u8") and dsc["kernel"].startswith("vpp_bilinear_r32_kernel") and "pass2" not in dsc, dsc
    check(vpp, oracle, y, uv, w, dst, fourcc=fmt, planes=1, rt=rt)
    check(vpp, oracle, y, uv, w, dst, fourcc=fmt, planes=1, rt=rt, n=5)
    if fmt == UYVY:   # round 6: fp32 UYVY too (the lanes of a run trade their packed dwords through LDS: whole-line float stores; partial runs, one-thread rows)
        fpn = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=rt, pixel_format=fmt, planes_pos=1, normalization=True)
        if not knob_run():
            dsc = V.describe(fpn, w, h, pitch=pitch)
            assert dsc["out"] == "uyvy_f32" and dsc["kernel"].startswith("vpp_bilinear_r32_kernel") and "pass2" not in dsc, dsc
        check(vpp, oracle, y, uv, w, dst, fourcc=fmt, planes=1, rt=rt, norm=True)
        check(vpp, oracle, y, uv, w, dst, fourcc=fmt, planes=1, rt=rt, norm=True, n=3)


def test_r32_uyvy_fallbacks(vpp, oracle):
    """Requests next to the single-pass domain keep the two passes: fp32 YUV444, BICUBIC, a misaligned crop origin, a misaligned output."""
    import tensor_stream as ts
    from tensor_stream import vpp as V
    y, uv = synth_nv12(1920, 1080, seed=35, pitch=2048)
    for kw in (dict(pixel_format=UYVY, normalization=True, resize_type=2), dict(pixel_format=YUV444, normalization=True), dict(pixel_format=UYVY, normalization=False, crop_coords=(6, 2, 966, 542), width=640, height=360)):
        args = dict(width=1280, height=720, resize_type=BILINEAR, planes_pos=1)
        args.update(kw)
        dsc = V.describe(ts.FrameParameters(**args), 1920, 1080, pitch=2048)
        assert dsc["out"] == "nv12_u8" and dsc["pass2"].startswith("fmt_"), dsc
    check(vpp, oracle, y, uv, 1920, (640, 360), fourcc=UYVY, planes=1, crop=(6, 2, 966, 542), r32=False)
    check(vpp, oracle, y, uv, 1920, (640, 360), fourcc=UYVY, planes=1, crop=(4, 2, 964, 542))      # aligned crop: one pass
    check(vpp, oracle, y, uv, 1920, (640, 360), fourcc=YUV444, planes=1, crop=(4, 2, 964, 542))
    check(vpp, oracle, y, uv, 1920, (640, 360), fourcc=YUV444, planes=1, crop=(6, 2, 966, 542), r32=False)
    # an output pointer that is not 16-byte aligned: two passes (the element-wise format kernel), same bytes
    fp = ts.FrameParameters(width=1280, height=720, resize_type=BILINEAR, pixel_format=UYVY, planes_pos=1, normalization=False)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    buf = torch.zeros(1280 * 720 * 2 + 64, dtype=torch.uint8, device="cuda")
    out = buf[4: 4 + 1280 * 720 * 2]
    vpp.Convert(ty, tuv, fp, width=1920, out=out)
    torch.cuda.synchronize()
    ref = oracle.convert(y, uv, dst=(1280, 720), resize_type=BILINEAR, fourcc=UYVY, planes=1, normalization=False, nthreads=8, width=1920)[0]
    assert np.array_equal(out.cpu().numpy(), ref.view(np.uint8).ravel())


@pytest.mark.parametrize("chunk", range(3))
def test_r32_fuzz(vpp, oracle, chunk):
    rng = np.random.default_rng(3200 + chunk)
    for k in range(24):
        dw, dh = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 40)) * 4
        w, h = dw * 3 // 2, dh * 3 // 2
        pitch = (w + 3) // 4 * 4 + 4 * int(rng.integers(0, 4))
        fourcc, planes = [(RGB24, 0), (RGB24, 1), (BGR24, 0), (BGR24, 1), (NV12, 1), (Y800, 1), (UYVY, 1), (YUV444, 1)][int(rng.integers(0, 8))]
        y, uv = synth_nv12(w, h, seed=8000 + 100 * chunk + k, pitch=pitch)
        check(vpp, oracle, y, uv, w, (dw, dh), fourcc=fourcc, planes=planes, n=int(rng.choice([1, 1, 2])), rt=int(rng.choice([NEAREST, BILINEAR, AREA])))


@pytest.mark.parametrize("src,pitch", [((3840, 2160), 3840), ((1920, 1080), 2048), ((64, 32), 64), ((1936, 1096), 1952),
                                       ((592, 88), 592),     # 37 threads per row: a partial run of 5 lanes
                                       ((608, 88), 612)])    # 38 threads per row: a partial run of 6 lanes
@pytest.mark.parametrize("fourcc,planes", [(RGB24, 0), (BGR24, 1), (NV12, 1), (Y800, 1)])
@pytest.mark.parametrize("rt", [BILINEAR, AREA, NEAREST])
def test_r21_sizes_and_flavours(vpp, oracle, src, pitch, fourcc, planes, rt):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + fourcc + planes + rt + 1, pitch=pitch)
    big_planar_bilinear = rt == BILINEAR and fourcc == RGB24 and planes == 0 and src[0] * src[1] // 4 >= 1500000   # stays on the LDS kernel
    check(vpp, oracle, y, uv, src[0], (src[0] // 2, src[1] // 2), fourcc=fourcc, planes=planes, rt=rt, r32=not big_planar_bilinear)


def test_r21_batches_crops_two_pass(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=41, pitch=2048)
    for rt in (BILINEAR, AREA, NEAREST):
        check(vpp, oracle, y, uv, 1920, (960, 540), fourcc=RGB24, planes=1, n=64, rt=rt)
        check(vpp, oracle, y, uv, 1920, (640, 360), crop=(8, 6, 1288, 726), rt=rt)
        check(vpp, oracle, y, uv, 1920, (640, 360), crop=(10, 6, 1290, 726), rt=rt, r32=False)   # misaligned origin
    for fcc in (UYVY, YUV444):
        check(vpp, oracle, y, uv, 1920, (960, 540), fourcc=fcc, planes=1)
    check(vpp, oracle, y, uv, 1920, (960, 540), norm=True, r32=False)


@pytest.mark.parametrize("chunk", range(2))
def test_r21_fuzz(vpp, oracle, chunk):
    rng = np.random.default_rng(2100 + chunk)
    for k in range(24):
        dw, dh = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 40)) * 4
        w, h = dw * 2, dh * 2
        pitch = w + 4 * int(rng.integers(0, 4))
        fourcc, planes = [(RGB24, 0), (RGB24, 1), (BGR24, 0), (BGR24, 1), (NV12, 1), (Y800, 1), (UYVY, 1), (YUV444, 1)][int(rng.integers(0, 8))]
        y, uv = synth_nv12(w, h, seed=9000 + 100 * chunk + k, pitch=pitch)
        check(vpp, oracle, y, uv, w, (dw, dh), fourcc=fourcc, planes=planes, n=int(rng.choice([1, 1, 2])), rt=int(rng.choice([NEAREST, BILINEAR, AREA])))


F32_FLAVOURS = [(BGR24, 0), (RGB24, 1), (NV12, 1), (Y800, 1), (6, 1)]  # fp32 planar / merged, NV12, Y800, HSV


@pytest.mark.parametrize("src,pitch,two", [((960, 540), 960, False), ((1280, 720), 1280, True), ((444, 66), 444, False), ((456, 66), 460, False), ((48, 24), 48, True)])
@pytest.mark.parametrize("fourcc,planes", F32_FLAVOURS)
@pytest.mark.parametrize("rt", [AREA, NEAREST, BILINEAR])
def test_r32_fp32_flavours(oracle, src, pitch, two, fourcc, planes, rt, monkeypatch):
    """Round 4: the streaming kernel has fp32 flavours too (the resized values as packed bytes through the shared output side, vpp_r32_store.h).  By default
    only HSV takes them (VALU-bound: measured faster); TSVPP_R32=2 routes every fp32 flavour there -- all of them are checked under that setting."""
    import tensor_stream as ts
    monkeypatch.setenv("TSVPP_R32", "2")
    v = ts.VideoProcessor(device=0)
    try:
        y, uv = synth_nv12(src[0], src[1], seed=src[0] + fourcc + planes + rt, pitch=pitch)
        dst = (src[0] // 2, src[1] // 2) if two else (src[0] * 2 // 3, src[1] * 2 // 3)
        check(v, oracle, y, uv, src[0], dst, fourcc=fourcc, planes=planes, rt=rt, norm=True)
    finally:
        v.Close()


def test_r32_fp32_default_routing_and_batches(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=32, pitch=2048)
    for rt in (AREA, NEAREST, BILINEAR):
        check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=6, planes=1, rt=rt, norm=True, n=64)                # HSV: the streaming kernel
        check(vpp, oracle, y, uv, 1920, (960, 540), fourcc=6, planes=1, rt=rt, norm=True, n=3)
        check(vpp, oracle, y, uv, 1920, (1280, 720), fourcc=BGR24, planes=0, rt=rt, norm=True, r32=False)       # RGB / BGR fp32: the LDS kernels
