"""GPU: BILINEAR at sparse ratios on the row-segment kernel (vpp_bilinear_rows.hip: one wave per 64-column tile, the TAPPED source rows staged as
LDS-DMA row segments, taps read from LDS) against the oracle, bit for bit -- BASELINE config C3 (crop 1280x720 -> 256x256, all horizontal weights
zero: the `wx0` instance), the 2x2-tap instance at ratios 3.6 .. 15, every output flavour, crops with odd origins (misaligned plane pointers, the
U / V swap quirk), widths 4 k + 2 and widths narrower than a tile, partial bottom tiles, the frame's last rows / columns (tap clamps), every
workgroup width (1 / 2 / 4 waves), forced tile heights, forced use at small ratios (TSVPP_BILINEAR_ROWS=2 semantics through a second context is not
needed: the ratios below select it by default), and the plan check that it IS the kernel that ran.  Pitches that are no multiple of 16 keep the
byte-gather kernel (checked)."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu

BILINEAR = 1
WX0, TWO = "vpp_bilinear_rows_kernel<OUT,wx0>", "vpp_bilinear_rows_kernel<OUT,2x2>"
KNOBS = knob_run()  # knob runs (tools/knob_matrix.sh) pick other kernels on purpose


def run(vpp, oracle, y, uv, w, dst, fourcc=1, planes=0, norm=True, crop=(0, 0, 0, 0), expect=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=BILINEAR, pixel_format=fourcc, planes_pos=planes,
                            normalization=norm)
    if expect is not None and not KNOBS:
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1])["kernel"]
        assert k.startswith(expect), k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=BILINEAR, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


def test_baseline_c3(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=3, pitch=2048)
    run(vpp, oracle, y, uv, 1920, (256, 256), crop=(0, 0, 1280, 720), expect=WX0)
    run(vpp, oracle, y, uv, 1920, (256, 256), crop=(0, 0, 1280, 720), planes=1, norm=False, expect=WX0)


@pytest.mark.parametrize("src,dst,kernel", [
    ((1920, 1080), (300, 300), TWO),     # 6.4 x 3.6: 27 chunks per segment, two segments per instruction, single-wave workgroups (300 = 4.7 tiles)
    ((1920, 1080), (224, 224), TWO),     # 8.57 x 4.82: 35 chunks, one segment per instruction, two waves per workgroup
    ((3840, 2160), (640, 360), TWO),     # 6 x 6: weights 0.5
    ((3840, 2160), (256, 256), WX0),     # 15 x 8.4375: 61 chunks -- the widest segment one instruction takes
    ((1920, 1080), (384, 216), WX0),     # 5 x 5: both weights zero -> a point sampler takes it; stays a plan check below
    ((1920, 1080), (128, 60), WX0),      # 15 x 18: vertical ratio beyond any dense staging, horizontal at the limit
    ((1280, 720), (256, 180), WX0),      # 5 x 4 (wy = 0.5)
    ((1080, 1920), (216, 160), WX0),     # 5 x 12, portrait
])
def test_ratio_classes(vpp, oracle, src, dst, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0], pitch=(src[0] + 15) // 16 * 16)
    point = dst == (384, 216)  # 5 x 5: both axes' weights are zero -> a point sampler (fp32: the LDS one, uint8: the streaming one at an exact integer ratio)
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True, expect="vpp_point_kernel" if point else kernel)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False, expect="vpp_point_rn_kernel" if point else kernel)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (2, 0, True), (2, 1, False), (0, 1, False), (0, 1, True), (3, 1, False),
                                                 (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(1920, 1080, seed=77 + fourcc, pitch=1920)
    run(vpp, oracle, y, uv, 1920, (256, 256), fourcc=fourcc, planes=planes, norm=norm, crop=(0, 0, 1280, 720))  # wx0
    run(vpp, oracle, y, uv, 1920, (300, 200), fourcc=fourcc, planes=planes, norm=norm)                           # 2x2: 6.4 x 5.4


@pytest.mark.parametrize("pitch", [2048, 1936, 1925, 1924])
def test_crops_tails_and_pitches(vpp, oracle, pitch):
    """pitch % 16 == 0 keeps the kernel whatever the crop origin does to the plane pointers; other pitches fall back to byte gathers (same bits)."""
    y, uv = synth_nv12(1920, 1080, seed=11 + pitch, pitch=pitch)
    k = (TWO if pitch % 16 == 0 else "vpp_fused_gather_kernel")
    run(vpp, oracle, y, uv, 1920, (310, 178), crop=(121, 65, 1721, 865), expect=k)                    # odd origin: U / V swap quirk, unaligned planes; width 4 k + 2
    run(vpp, oracle, y, uv, 1920, (310, 178), crop=(121, 65, 1721, 865), planes=1, norm=False, expect=k)
    run(vpp, oracle, y, uv, 1920, (250, 100), crop=(920, 580, 1920, 1080), expect=k)                  # the bottom-right corner: the planes' last rows and bytes; 4 k + 2
    run(vpp, oracle, y, uv, 1920, (62, 30), crop=(6, 2, 1006, 482))                                   # narrower than a tile, 4 k + 2: row-tail launch
    run(vpp, oracle, y, uv, 1920, (258, 132), planes=1, norm=False)                                   # 7.44 x 8.18, shifted last tile column (258 = 256 + 2)
    run(vpp, oracle, y, uv, 1920, (132, 34))                                                          # 14.5 x 31.8: partial bottom tile (34 = 4 x 8 + 2)


def test_full_frame_edges(vpp, oracle):
    """Outputs whose last column / row tap the source's last sample (clamped right / bottom taps) and whose first taps sit at 0."""
    for (w, h), dst in [((640, 368), (128, 64)), ((1024, 512), (64, 32)), ((2048, 64), (256, 8)), ((80, 1080), (8, 200)), ((1296, 730), (86, 72))]:
        y, uv = synth_nv12(w, h, seed=w + h, pitch=(w + 15) // 16 * 16)
        run(vpp, oracle, y, uv, w, dst)
        run(vpp, oracle, y, uv, w, dst, planes=1, norm=False)


def test_batch_and_frames_far_apart(vpp, oracle):
    """A 64-frame batch: frame 0, a middle one and the last (the tile decode over frames)."""
    import tensor_stream as ts
    fp = ts.FrameParameters(width=256, height=256, crop_coords=(0, 0, 1280, 720), resize_type=BILINEAR, pixel_format=1, planes_pos=0, normalization=True)
    n = 64
    rng = np.random.default_rng(5)
    ys = rng.integers(0, 256, (n, 1080, 2048), dtype=np.uint8)
    uvs = rng.integers(0, 256, (n, 540, 2048), dtype=np.uint8)
    out = vpp.convert_batch(torch.from_numpy(ys).cuda(), torch.from_numpy(uvs).cuda(), fp, width=1920)
    torch.cuda.synchronize()
    for f in (0, 29, 63):
        ref, _, _ = oracle.convert(ys[f], uvs[f], crop=(0, 0, 1280, 720), dst=(256, 256), resize_type=BILINEAR, fourcc=1, planes=0, normalization=True,
                                   nthreads=8, width=1920)
        assert np.array_equal(out[f].cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8)), f


# ---- round 6: the pure point samplers on the same front end (one segment per output row) ------------------------------------------------------------
NEAR, POINT = "vpp_bilinear_rows_kernel<OUT,nearest>", "vpp_bilinear_rows_kernel<OUT,point>"


def run_rt(vpp, oracle, y, uv, w, dst, rt, fourcc=1, planes=0, norm=True, crop=(0, 0, 0, 0), expect=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if expect is not None and not KNOBS:
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1])["kernel"]
        assert k.startswith(expect), k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, rt, fourcc, planes, norm, crop, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst,rt,kernel", [
    ((1920, 1080), (224, 224), 0, NEAR),     # NEAREST 8.57 x 4.82
    ((3840, 2160), (480, 270), 0, NEAR),     # NEAREST 8 x 8 (fp32; uint8 at an exact integer ratio 3 / 4 / 5 only goes to the streaming point sampler)
    ((3840, 2160), (300, 300), 0, NEAR),     # 12.8 x 7.2: 4 k + ... 300 = 4.7 tiles
    ((3840, 2160), (256, 144), 1, POINT),    # BILINEAR 15 x 15: every weight zero
    ((3840, 2160), (256, 144), 2, POINT),    # BICUBIC  15 x 15: every weight zero -> the centre tap
    ((1920, 1080), (128, 72), 1, POINT),     # BILINEAR 15 x 15 from 1080p
    ((1920, 1080), (174, 98), 0, NEAR),      # 11.03 x 11.02: 4 k + 2 columns, partial bottom tile
])
def test_point_samplers_on_row_segments(vpp, oracle, src, dst, rt, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + rt, pitch=(src[0] + 15) // 16 * 16)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, planes=0, norm=True, expect=kernel)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, planes=1, norm=False)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, fourcc=3, planes=1, norm=False)   # NV12
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, fourcc=0, planes=1, norm=True)    # Y800: the chroma plane is neither staged nor sampled
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, fourcc=6, planes=1, norm=True)    # HSV


def test_point_samplers_crops_edges_and_pitches(vpp, oracle):
    y, uv = synth_nv12(3840, 2160, seed=4242, pitch=3840)
    run_rt(vpp, oracle, y, uv, 3840, (310, 178), 0, crop=(121, 65, 3321, 1865), expect=NEAR)            # odd origin: U / V swap quirk, unaligned planes; width 4 k + 2
    run_rt(vpp, oracle, y, uv, 3840, (310, 178), 0, crop=(121, 65, 3321, 1865), planes=1, norm=False)
    run_rt(vpp, oracle, y, uv, 3840, (250, 100), 0, crop=(1840, 1160, 3840, 2160), expect=NEAR)         # the bottom-right corner: the planes' last rows and bytes
    run_rt(vpp, oracle, y, uv, 3840, (62, 30), 0, crop=(12, 4, 2012, 964))                              # narrower than a tile, 4 k + 2
    y2, uv2 = synth_nv12(1920, 1080, seed=4343, pitch=1925)                                             # pitch % 16 != 0: not this kernel (same bits)
    run_rt(vpp, oracle, y2, uv2, 1920, (224, 224), 0, expect="vpp_point_kernel" if not KNOBS else None)


def test_point_rows_batch(vpp, oracle):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=224, height=224, resize_type=0, pixel_format=2, planes_pos=0, normalization=True)
    n = 40
    rng = np.random.default_rng(6)
    ys = rng.integers(0, 256, (n, 1080, 1920), dtype=np.uint8)
    uvs = rng.integers(0, 256, (n, 540, 1920), dtype=np.uint8)
    out = vpp.convert_batch(torch.from_numpy(ys).cuda(), torch.from_numpy(uvs).cuda(), fp)
    torch.cuda.synchronize()
    for f in (0, 17, 39):
        ref, _, _ = oracle.convert(ys[f], uvs[f], dst=(224, 224), resize_type=0, fourcc=2, planes=0, normalization=True, nthreads=8)
        assert np.array_equal(out[f].cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8)), f


@pytest.mark.parametrize("src,dst,rt,kernel", [
    ((3840, 2160), (224, 224), 0, NEAR),     # NEAREST 17.1 x 9.6: 70 chunks per segment -- two DMA instructions per segment (round 6; byte gathers before)
    ((3840, 2160), (224, 224), 1, TWO),      # BILINEAR, the same geometry
    ((3840, 2160), (128, 72), 1, TWO),       # 30 x 30 (weights 0.5): 120 chunks, the widest segment
    ((3840, 2160), (160, 90), 0, NEAR),      # 24 x 24
    ((3840, 1080), (150, 300), 1, TWO),      # 25.6 x 3.6: 4 k + 2 columns, partial tiles
])
def test_segments_wider_than_one_instruction(vpp, oracle, src, dst, rt, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + 7 * rt, pitch=(src[0] + 15) // 16 * 16)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, planes=0, norm=True, expect=kernel)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, planes=1, norm=False)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, fourcc=3, planes=1, norm=False)
    run_rt(vpp, oracle, y, uv, src[0], dst, rt, crop=(101, 51, 3701, 2051) if src[1] == 2160 else (101, 51, 3701, 1051))  # odd origin: misaligned planes, U / V swap quirk
