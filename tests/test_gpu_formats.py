"""GPU: the other FourCC outputs -- Y800, NV12, UYVY, YUV444, HSV -- against the
reference's own golden dumps and against the oracle, with and without crop / resize in front."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, ulp_diff, knob_run

pytestmark = pytest.mark.gpu
Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = range(7)


def run(vpp, y, uv, fourcc, norm, crop=(0, 0, 0, 0), dst=(0, 0), rt=0, width=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, normalization=norm)
    out = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=width)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name,fcc", [("Y800", Y800), ("NV12", NV12), ("UYVY", UYVY), ("YUV444", YUV444), ("HSV", HSV)])
def test_reference_golden_files(vpp, golden, oracle, name, fcc):
    """reference tests/src/VPPTests.cpp:387-512: normalised fp32 dumps of its 320x240 frame."""
    got = run(vpp, golden["Y"], golden["UVp"], fcc, True)
    assert got.dtype == np.float32 and got.shape == oracle.shape_for(fcc, 1, 320, 240)
    assert np.array_equal(got.ravel().view(np.uint32), golden[name])


@pytest.mark.parametrize("fcc", [Y800, NV12, UYVY, YUV444, HSV])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("crop,dst,rt", [((0, 0, 0, 0), (0, 0), 0),
                                         ((0, 0, 0, 0), (480, 360), 1),
                                         ((121, 65, 601, 401), (0, 0), 0),
                                         ((120, 64, 600, 400), (300, 200), 3),
                                         ((0, 0, 0, 0), (362, 202), 2)])
def test_formats_vs_oracle(vpp, oracle, fcc, norm, crop, dst, rt):
    y, uv = synth_nv12(1080, 608, seed=fcc * 10 + rt, pitch=1088)
    got = run(vpp, y, uv, fcc, norm, crop, dst, rt, width=1080)
    ref, ow, oh = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=8, width=1080)
    assert got.shape == oracle.shape_for(fcc, 1, ow, oh)
    got = got.ravel()
    assert got.dtype == ref.dtype and got.size == ref.size
    if got.dtype == np.uint8:
        assert np.array_equal(got, ref)
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max ulp {ulp_diff(np.abs(got), np.abs(ref))}"


def test_batch_of_two_pass_format(vpp, oracle):
    import tensor_stream as ts
    frames = [synth_nv12(640, 360, seed=50 + i) for i in range(3)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=YUV444, normalization=True)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    for i in range(3):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(320, 180), resize_type=1, fourcc=YUV444, normalization=True)
        assert np.array_equal(out[i].cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("fcc", [Y800, NV12, HSV])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("dst,rt", [((540, 304), 0),      # NEAREST: point kernel
                                    ((540, 304), 1),      # BILINEAR with zero weights: point kernel
                                    ((1280, 720), 1),     # BILINEAR up: 2x2-tap kernel
                                    ((720, 404), 2),      # BICUBIC: table kernel
                                    ((540, 304), 3),      # AREA 2x: dyadic kernels
                                    ((216, 152), 3),      # AREA 5x / 4x: direct dyadic kernel
                                    ((360, 304), 3),      # AREA 3x / 2x: dyadic LDS kernel, 3 and 2 vertical taps
                                    ((432, 244), 3),      # AREA 2.5x / 2.49x: direct float kernel
                                    ((1440, 808), 3),     # AREA up: bilinear variant
                                    ((120, 76), 1)])      # BILINEAR 9x / 8x: sparse gather
def test_fused_formats_every_kernel_family(vpp, oracle, fcc, norm, dst, rt):
    """Y800 / NV12 / HSV are output flavours of the fused kernels: one case per kernel family."""
    y, uv = synth_nv12(1080, 608, seed=fcc * 7 + rt + dst[0], pitch=1152)
    got = run(vpp, y, uv, fcc, norm, (0, 0, 0, 0), dst, rt, width=1080).ravel()
    ref, ow, oh = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=8, width=1080)
    assert (ow, oh) == dst and got.dtype == ref.dtype and got.size == ref.size
    if got.dtype == np.uint8:
        assert np.array_equal(got, ref)
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max ulp {ulp_diff(np.abs(got), np.abs(ref))}"


@pytest.mark.parametrize("fcc", [Y800, NV12, UYVY, YUV444, HSV])
def test_format_batch_larger_than_one_launch(vpp, oracle, fcc):
    """70 frames = two launches (64 + 6) of each pass; every frame must land in its own output."""
    import tensor_stream as ts
    n = 70
    frames = [synth_nv12(320, 240, seed=900 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=160, height=120, resize_type=1, pixel_format=fcc, normalization=False)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in (0, 1, 63, 64, 69):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(160, 120), resize_type=1, fourcc=fcc, normalization=False)
        got = out[i].ravel()
        assert got.dtype == ref.dtype
        assert np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got, ref.view(np.uint32) if ref.dtype == np.float32 else ref)


@pytest.mark.parametrize("fcc", [UYVY, YUV444])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("src,pitch,dst,rt", [((1280, 720), 1280, (0, 0), 0),      # rows of 80 / 160 / 320 / 640 threads: wave boundaries inside a row
                                              ((2048, 6), 2048, (0, 0), 0),        # three chroma rows: the filter's clamps at both ends
                                              ((32, 4), 32, (0, 0), 0),            # two threads per row (uint8 YUV444): both take the row-end path
                                              ((64, 2), 64, (0, 0), 0),            # one chroma row
                                              ((1040, 250), 1056, (0, 0), 0),      # padded pitch, width 16 k + 0 but not 64 k: partial last wave
                                              ((1920, 1080), 1920, (1280, 720), 1),  # behind a resize: the tight NV12 intermediate
                                              ((1920, 1080), 1920, (1024, 578), 3)])
def test_row_pair_kernels(vpp, oracle, fcc, norm, src, pitch, dst, rt):
    """The row-pair UYVY / YUV444 kernels (two output rows per thread, neighbours' chroma by wave shuffles) at the geometries
    that exercise their wave-boundary, row-end and frame-end paths."""
    y, uv = synth_nv12(src[0], src[1], seed=fcc * 13 + src[0] + rt, pitch=pitch)
    got = run(vpp, y, uv, fcc, norm, (0, 0, 0, 0), dst, rt, width=src[0]).ravel()
    ref, ow, oh = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=8, width=src[0])
    assert got.dtype == ref.dtype and got.size == ref.size
    if got.dtype == np.uint8:
        bad = np.flatnonzero(got != ref)
    else:
        bad = np.flatnonzero(got.view(np.uint32) != ref.view(np.uint32))
    assert bad.size == 0, (bad[:10], bad.size, ow, oh)


@pytest.mark.parametrize("fcc", [UYVY, YUV444])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("src,pitch,dst,rt", [((1080, 608), 1088, (0, 0), 0),       # width 8 k, not 16 k: the 8-pixel uint8 instances
                                              ((600, 338), 608, (0, 0), 0),         # the same, several partial waves
                                              ((300, 200), 304, (0, 0), 0),         # width 4 k: the 4-pixel instances
                                              ((1100, 64), 1104, (0, 0), 0),
                                              ((12, 4), 16, (0, 0), 0),             # three / one thread per row
                                              ((1366, 768), 1376, (0, 0), 0),       # width 4 k + 2: UYVY one pair per thread, YUV444 four pixels and a half thread at the row's end
                                              ((854, 480), 854, (0, 0), 0),         # ... on a pitch that is no multiple of 4 either
                                              ((10, 4), 10, (0, 0), 0),             # ... three threads per row, the last one half
                                              ((14, 6), 16, (0, 0), 0),
                                              ((258, 4), 258, (0, 0), 0),           # ... the half thread is lane 0 of a second workgroup
                                              ((1026, 6), 1040, (0, 0), 0),
                                              ((6, 4), 6, (0, 0), 0),               # widths below 8: neighbours that wrap more than one row
                                              ((2, 2), 2, (0, 0), 0),
                                              ((1920, 1080), 1920, (1080, 608), 1),   # behind a resize
                                              ((1920, 1080), 1920, (1366, 768), 1),
                                              ((1920, 1080), 1920, (300, 300), 3)])
def test_row_pair_kernels_narrow_instances_and_one_pair_kernels(vpp, oracle, fcc, norm, src, pitch, dst, rt):
    """Round 6: uint8 row-pair instances of 8 / 4 pixels per thread (YUV444) and 4 (UYVY) for widths that are no multiple of 16 / 8, the row ends' wrapped neighbours
    loaded in one round trip, no alignment condition on planes / pitches / outputs (vector accesses at any address), YUV444 at widths 4 k + 2 (fmt_yuv444_rp_tail),
    and what is left for the one-row kernels (frames narrower than two threads)."""
    y, uv = synth_nv12(src[0], src[1], seed=fcc * 17 + src[0] + rt, pitch=pitch)
    got = run(vpp, y, uv, fcc, norm, (0, 0, 0, 0), dst, rt, width=src[0]).ravel()
    ref, ow, oh = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=8, width=src[0])
    assert got.dtype == ref.dtype and got.size == ref.size
    bad = np.flatnonzero(got.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (bad[:10], bad.size, ow, oh)


def test_one_pair_kernels_on_misaligned_outputs_and_crops(vpp, oracle):
    """Crops whose origin misaligns the planes (byte loads), every format kernel family."""
    y, uv = synth_nv12(1366, 768, seed=91, pitch=1371)
    for fcc in (UYVY, YUV444):
        for norm in (False, True):
            for crop in ((0, 0, 0, 0), (2, 2, 1364, 766), (102, 50, 1182, 658), (100, 50, 400, 250)):   # widths 1362 (4 k + 2), 1080 (8 k), 300 (4 k)
                got = run(vpp, y, uv, fcc, norm, crop, (0, 0), 0, width=1366).ravel()
                ref, ow, oh = oracle.convert(y, uv, crop=crop, resize_type=0, fourcc=fcc, normalization=norm, nthreads=8, width=1366)
                bad = np.flatnonzero(got.view(np.uint8) != ref.view(np.uint8))
                assert got.size == ref.size and bad.size == 0, (fcc, norm, crop, bad[:10], bad.size)


@pytest.mark.parametrize("chunk", range(3))
def test_row_pair_kernels_fuzz(vpp, oracle, chunk):
    """Seeded fuzzing of UYVY / YUV444 at geometries the row-pair kernels take (widths that are multiples of 16, 16-byte pitches):
    random sizes from one thread per row to several waves per row, with and without a resize in front."""
    rng = np.random.default_rng(4242 + chunk)
    for k in range(24):
        w = int(rng.integers(2, 90)) * 16
        h = int(rng.integers(1, 60)) * 2
        pitch = w + 16 * int(rng.integers(0, 3))
        fcc = int(rng.choice([UYVY, YUV444]))
        norm = bool(rng.integers(0, 2))
        dst, rt = (0, 0), 0
        if rng.random() < 0.4:
            dst, rt = (int(rng.integers(2, 40)) * 16, int(rng.integers(1, 40)) * 2), int(rng.integers(0, 4))
        y, uv = synth_nv12(w, h, seed=5000 + 100 * chunk + k, pitch=pitch)
        try:
            ref, ow, oh = oracle.convert(y, uv, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=4, width=w)
        except RuntimeError:
            continue
        got = run(vpp, y, uv, fcc, norm, (0, 0, 0, 0), dst, rt, width=w).ravel()
        assert got.dtype == ref.dtype and got.size == ref.size
        bad = np.flatnonzero(got.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, ((w, h, pitch, fcc, norm, dst, rt), bad[:8], bad.size)


@pytest.mark.parametrize("chunk", range(4))
def test_formats_fuzz_any_even_width_pitch_and_crop(vpp, oracle, chunk):
    """Round 6: UYVY / YUV444 at ANY even width and height, any pitch, odd crop origins (misaligned plane pointers), with and without a resize in front: every
    width class (16 k, 8 k, 4 k, 4 k + 2, below two threads), the row-pair kernels' partial waves, the half thread of fmt_yuv444_rp_tail in every lane position."""
    rng = np.random.default_rng(9100 + chunk)
    for k in range(40):
        w = 2 * int(rng.integers(1, 700))
        h = 2 * int(rng.integers(1, 40))
        pitch = w + int(rng.integers(0, 20))
        fcc = int(rng.choice([UYVY, YUV444]))
        norm = bool(rng.integers(0, 2))
        crop = (0, 0, 0, 0)
        if rng.random() < 0.4 and w >= 8 and h >= 8:
            l, t = int(rng.integers(0, w // 4)), int(rng.integers(0, h // 4))
            cw, ch = 2 * int(rng.integers(1, (w - l) // 2 + 1)), 2 * int(rng.integers(1, (h - t) // 2 + 1))
            crop = (l, t, l + cw, t + ch)
        dst, rt = (0, 0), 0
        if rng.random() < 0.3:
            dst, rt = (2 * int(rng.integers(1, 400)), 2 * int(rng.integers(1, 40))), int(rng.integers(0, 4))
        y, uv = synth_nv12(w, h, seed=9000 + 100 * chunk + k, pitch=pitch)
        try:
            ref, ow, oh = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=4, width=w)
        except RuntimeError:
            continue
        got = run(vpp, y, uv, fcc, norm, crop, dst, rt, width=w).ravel()
        assert got.dtype == ref.dtype and got.size == ref.size, (w, h, pitch, fcc, norm, crop, dst, rt)
        bad = np.flatnonzero(got.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, ((w, h, pitch, fcc, norm, crop, dst, rt), bad[:8], bad.size)


@pytest.mark.parametrize("fcc", [Y800, NV12])
def test_plane_copies_16_bytes_per_lane(vpp, oracle, fcc):
    """No resize, uint8 Y800 / NV12: the output is the (cropped) planes made tight -- vpp_copy16_kernel where widths are multiples of 16 and heights of 4,
    the colour-only kernel elsewhere; both against the oracle, with ragged pitches, crops that keep the planes dword-aligned and batches."""
    import tensor_stream as ts
    for (w, h, pitch), crop in [((1920, 1080, 2048), (0, 0, 0, 0)), ((1920, 1080, 1924), (0, 0, 0, 0)), ((1920, 1080, 2048), (64, 32, 1344, 752)),
                                ((1920, 1080, 2048), (4, 2, 1028, 770)), ((1936, 1088, 1936), (0, 0, 0, 0)), ((1920, 1080, 2048), (8, 8, 1000, 500)),
                                ((64, 8, 64), (0, 0, 0, 0)), ((16, 4, 20), (0, 0, 0, 0)), ((1920, 1082, 1920), (0, 0, 0, 0))]:
        y, uv = synth_nv12(w, h, seed=w + h + fcc, pitch=pitch)
        fp = ts.FrameParameters(crop_coords=crop, pixel_format=fcc, normalization=False)
        cw, ch = (crop[2] - crop[0], crop[3] - crop[1]) if crop[2] else (w, h)
        want16 = cw % 16 == 0 and ch % 4 == 0 and pitch % 4 == 0 and crop[0] % 4 == 0
        if not knob_run():  # (knob runs, tools/knob_matrix.sh, select other kernels on purpose)
            k = ts.describe(fp, w, h, pitch=pitch, n_frames=1)["kernel"]
            assert k.startswith("vpp_copy16_kernel") == want16, (k, w, h, pitch, crop)
        got = run(vpp, y, uv, fcc, False, crop=crop, width=w)
        ref, ow, oh = oracle.convert(y, uv, crop=crop, fourcc=fcc, normalization=False, nthreads=8, width=w)
        assert got.shape == oracle.shape_for(fcc, 1, ow, oh) and np.array_equal(got.ravel(), ref), (w, h, pitch, crop)
    ys = torch.randint(0, 256, (70, 360, 640), dtype=torch.uint8, device="cuda")
    uvs = torch.randint(0, 256, (70, 180, 640), dtype=torch.uint8, device="cuda")
    fp = ts.FrameParameters(pixel_format=fcc, normalization=False)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    for i in (0, 63, 64, 69):
        ref, _, _ = oracle.convert(ys[i].cpu().numpy(), uvs[i].cpu().numpy(), fourcc=fcc, normalization=False)
        assert np.array_equal(out[i].cpu().numpy().ravel(), ref), i
