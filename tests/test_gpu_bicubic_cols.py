"""GPU: the wave-per-tile BICUBIC kernel (vpp_bicubic_cols.hip: one lane per output column, host-built coefficient tables, integer
dot-product sums with the reference's fp64 evaluation near rounding ties) against the oracle, bit for bit -- non-dyadic ratios
(its default domain), every staging mode (LDS-DMA ring with 2..16 chunks per row, per-lane loads, pitches that are no multiple of
16 or 4), dense and sparse vertical sampling, every tile height, every output flavour, crops with odd origins, widths 4 k + 2,
frame edges (the reference's tap-collapse rule, the planes' last rows), the dyadic instance without the tie test (forced), graph
capture with and without prepared tables, and the plan check that it IS the kernel that ran."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu

BICUBIC = 2


def run(vpp, oracle, y, uv, w, dst, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0), expect=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=BICUBIC, pixel_format=fourcc, planes_pos=planes,
                            normalization=norm)
    if expect is not None and not knob_run():  # knob runs (tools/knob_matrix.sh) pick other kernels
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1])["kernel"]
        assert k.startswith(expect), k
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=BICUBIC, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    g = got.cpu().numpy().ravel()
    assert g.size == ref.size
    bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
    assert bad.size == 0, (dst, fourcc, planes, norm, crop, bad[:8], bad.size)


TIE_DENSE, TIE_SPARSE = "vpp_bicubic_cols_kernel<OUT,tie,dense>", "vpp_bicubic_cols_kernel<OUT,tie,sparse>"


@pytest.mark.parametrize("src,dst,kernel", [
    ((1080, 608), (480, 360), TIE_DENSE),     # 2.25 x 1.689: the reference's own test size; LDS-DMA, 11 chunks per row
    ((1920, 1080), (1440, 810), TIE_DENSE),   # 1.333: 7 chunks per row, eight rows per DMA instruction
    ((1280, 720), (1920, 1080), TIE_DENSE),   # 0.667 up-scale: 5 chunks, twelve rows per instruction, 32-row tiles
    ((640, 360), (1600, 900), TIE_DENSE),     # 0.4 up-scale: 3 chunks, sixteen rows per instruction
    ((1920, 1080), (640, 640), TIE_DENSE),    # 3.0 (all weights 0: the point kernel) x 1.6875 -> mixed: stays bicubic
    ((1920, 1080), (300, 300), TIE_DENSE),    # 6.4 x 3.6: 27 chunks per segment -- fp32 outputs: LDS-DMA, two instructions per group of four rows (round 5); uint8: per-lane loads
    ((1920, 1080), (224, 224), TIE_SPARSE),   # 8.57 x 4.82: per-lane loads, only the tapped rows
    ((1080, 1920), (480, 224), TIE_SPARSE),   # 2.25 x 8.57: LDS-DMA with one output row's four taps per instruction
    ((3840, 2160), (854, 480), TIE_SPARSE),   # 4.5 x 4.5: dyadic? no: 3840 / 854 is not -- and dst_w = 4 k + 2 (row-tail launch)
])
def test_ratio_classes(vpp, oracle, src, dst, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0])
    run(vpp, oracle, y, uv, src[0], dst, planes=0, norm=True, expect=kernel)
    run(vpp, oracle, y, uv, src[0], dst, planes=1, norm=False, expect=kernel)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (2, 1, True), (0, 1, False), (0, 1, True), (3, 1, False),
                                                 (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(1080, 608, seed=177 + fourcc)
    run(vpp, oracle, y, uv, 1080, (480, 360), fourcc=fourcc, planes=planes, norm=norm)
    run(vpp, oracle, y, uv, 1080, (200, 100), fourcc=fourcc, planes=planes, norm=norm)  # 5.4 x 6.08: sparse, per-lane loads


@pytest.mark.parametrize("pitch", [1037, 1036, 1040, 1056])
def test_ragged_pitches_odd_crops_and_narrow_tails(vpp, oracle, pitch):
    """pitch % 16 != 0 -> per-lane loads; % 4 != 0 -> every row has its own misalignment; 1040 / 1056 keep the LDS-DMA ring with
    crop origins that misalign the plane pointers."""
    y, uv = synth_nv12(1000, 600, seed=5 + pitch, pitch=pitch)
    run(vpp, oracle, y, uv, 1000, (430, 310), norm=True)
    run(vpp, oracle, y, uv, 1000, (430, 310), planes=1)
    run(vpp, oracle, y, uv, 1000, (310, 178), crop=(121, 65, 921, 465), norm=True)      # odd origin: U / V swap quirk, unaligned planes
    run(vpp, oracle, y, uv, 1000, (182, 94), crop=(7, 3, 507, 303), planes=1)           # width 4 k + 2
    run(vpp, oracle, y, uv, 1000, (1302, 780), crop=(500, 300, 1000, 600), norm=True)   # up-scale of the bottom-right corner: the planes' last rows / bytes


def test_tiny_and_one_tile_frames(vpp, oracle):
    for (w, h), dst in [((34, 18), (20, 14)), ((16, 16), (30, 22)), ((258, 130), (66, 34)), ((64, 64), (6, 6)), ((1920, 8), (500, 6)), ((8, 1080), (6, 400))]:
        y, uv = synth_nv12(w, h, seed=w + h)
        run(vpp, oracle, y, uv, w, dst, norm=True)
        run(vpp, oracle, y, uv, w, dst, planes=1)


def test_constant_and_extreme_frames(vpp, oracle):
    """Flat 0 / 255 frames (clamps at both ends: the Keys kernel overshoots), a checkerboard (maximal overshoot) and a frame of
    values that put many sums near x.5 (the tie zone: the fp64 redo decides)."""
    w, h = 642, 362
    for fill in (0, 255):
        y = np.full((h, w), fill, np.uint8)
        uv = np.full((h // 2, w), fill, np.uint8)
        run(vpp, oracle, y, uv, w, (300, 170), norm=True)
    yy, xx = np.mgrid[0:h, 0:w]
    y = (((yy + xx) & 1) * 255).astype(np.uint8)
    uv = (((yy[: h // 2] + xx[: h // 2] // 2) & 1) * 255).astype(np.uint8)
    run(vpp, oracle, y, uv, w, (300, 170), planes=1)
    run(vpp, oracle, y, uv, w, (900, 500), norm=True)
    rng = np.random.default_rng(3)
    y = rng.choice(np.array([0, 1, 2, 127, 128, 253, 254, 255], np.uint8), size=(h, w))
    uv = rng.choice(np.array([0, 1, 128, 254, 255], np.uint8), size=(h // 2, w))
    run(vpp, oracle, y, uv, w, (300, 170), norm=True)
    run(vpp, oracle, y, uv, w, (514, 290), planes=1)


@pytest.mark.parametrize("env", [{"TSVPP_BICUBIC_COLS": "2"}, {"TSVPP_BICUBIC_COLS": "2", "TSVPP_BICUBIC_DMA": "0"}, {"TSVPP_BICUBIC_DMA": "2"},
                                 {"TSVPP_BICUBIC_ROWS": "8"}, {"TSVPP_BICUBIC_ROWS": "24"}, {"TSVPP_BICUBIC_ROWS": "32"}, {"TSVPP_BICUBIC_COLS": "0"}])
def test_forced_variants(oracle, env, monkeypatch):
    """The dyadic instance (no tie test) that the integer kernel normally pre-empts, per-lane loads where the ring would run,
    256-byte segments, every tile height, and the generic gathers that remain when the kernel is switched off."""
    import tensor_stream as ts
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    v = ts.VideoProcessor(device=0)  # the knobs are read when the context is created
    try:
        for src, dst in [((1920, 1080), (1280, 720)), ((960, 540), (1280, 720)), ((1080, 608), (480, 360)), ((1280, 720), (160, 90)), ((1280, 720), (400, 260))]:
            y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[1])
            run(v, oracle, y, uv, src[0], dst, norm=True)
            run(v, oracle, y, uv, src[0], dst, planes=1)
    finally:
        v.Close()


@pytest.mark.parametrize("prepared", [True, False])
def test_first_call_graph_capture(oracle, prepared):
    """The coefficient tables are built by tsvpp_prepare_batch (or on the first conversion); a FIRST conversion inside a graph
    capture without them must not allocate: it runs on the generic gathers instead, same results."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    n = 3
    frames = [synth_nv12(1080, 608, seed=700 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=480, height=360, resize_type=BICUBIC, pixel_format=1, planes_pos=0, normalization=True)
    s = torch.cuda.Stream()
    if prepared:
        v.prepare(fp, 1080, 608, n_frames=n, stream=s.cuda_stream)
    out = v._alloc(fp.parameters, 1080, 608, n)
    batch = v.make_batch(ys, uvs, fp, out=out)
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):  # no warm-up call
        v.run_batch(batch, torch.cuda.current_stream().cuda_stream)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in range(n):
        ref = oracle.convert(frames[i][0], frames[i][1], dst=(480, 360), resize_type=BICUBIC, fourcc=1, planes=0, normalization=True, nthreads=4)[0]
        assert np.array_equal(o[i].ravel().view(np.uint8), ref.view(np.uint8))
    v.Close()


def test_batch_of_64_frames_and_consumer_streams(vpp, oracle):
    import tensor_stream as ts
    n = 64
    frames = [synth_nv12(640, 360, seed=3000 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=300, height=200, resize_type=BICUBIC, pixel_format=2, planes_pos=0, normalization=True)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in (0, 1, 31, 63):
        ref = oracle.convert(frames[i][0], frames[i][1], dst=(300, 200), resize_type=BICUBIC, fourcc=2, planes=0, normalization=True, nthreads=4)[0]
        assert np.array_equal(o[i].ravel().view(np.uint8), ref.view(np.uint8))


def test_table_cache_evicts_least_recently_used(oracle):
    """A context keeps the host-built tables of its 1024 most recently used geometries: the 1025th releases the oldest set (round 2
    stopped caching at 256 and left every later geometry on the slower kernels); an evicted geometry is rebuilt on its next use."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    y, uv = synth_nv12(96, 64, seed=77)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    sizes = [(40 + 2 * (k % 40), 20 + 2 * (k // 40)) for k in range(1040)]   # 1040 distinct geometries of one source
    assert len(set(sizes)) == len(sizes)
    keep = {}
    for k, dst in enumerate(sizes):
        fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=BICUBIC, pixel_format=1, planes_pos=1, normalization=False)
        out = v.Convert(ty, tuv, fp, width=96)
        if k in (0, 1, 519, 1039):
            keep[k] = out
    torch.cuda.synchronize()
    for k in (0, 1, 519, 1039, 0):   # geometry 0 was evicted meanwhile: converted again at the end
        dst = sizes[k]
        fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=BICUBIC, pixel_format=1, planes_pos=1, normalization=False)
        got = v.Convert(ty, tuv, fp, width=96)
        torch.cuda.synchronize()
        ref = oracle.convert(y, uv, dst=dst, resize_type=BICUBIC, fourcc=1, planes=1, normalization=False, nthreads=2, width=96)[0]
        assert np.array_equal(got.cpu().numpy().ravel(), ref.view(np.uint8).ravel())
        assert np.array_equal(keep[k].cpu().numpy().ravel(), ref.view(np.uint8).ravel())
    v.Close()
