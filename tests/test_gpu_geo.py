"""GPU: the geometry-table variant of the 2x2-tap kernel (vpp_bilinear_geo_kernel: host-built footprints, column and row-pair
records) against the oracle, bit for bit -- single frames and 64-frame batches (different tile shapes), misaligned crop
origins, right / bottom edges, the two-column row tail, taller thread tiles, graph capture after prepare()."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
BILINEAR, AREA = 1, 3


class _Env:
    """TSVPP_* knobs are read when a context is created / a request is described: set them for exactly that long."""
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="module")
def gvpp():
    """A context with TSVPP_GEO=2: the geometry tables wherever they apply (the default uses them for uint8 outputs with
    dyadic weights only, where they were measured to win)."""
    import tensor_stream
    with _Env(TSVPP_GEO="2", TSVPP_R32="0"):   # (the streaming 3 : 2 kernel would take the uint8 1080p -> 720p cases)
        v = tensor_stream.VideoProcessor(device=0, max_consumers=2)
    yield v
    v.Close()


def params(dst, rt=BILINEAR, fourcc=2, planes=0, norm=False, crop=(0, 0, 0, 0)):
    import tensor_stream as ts
    return ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)


def check(v, oracle, y, uv, w, dst, n=1, geo=1, **kw):
    from tensor_stream import vpp as V
    fp = params(dst, **kw)
    h = y.shape[0]
    crop = kw.get("crop", (0, 0, 0, 0))
    # the request takes the geometry tables (host logic; the crop must not change the pitch)
    if not knob_run():   # (tools/knob_matrix*.sh replay the suite under knobs that change the selection)
        with _Env(TSVPP_GEO="2", TSVPP_R32="0"):
            d = V.describe(fp, w, h, pitch=y.shape[1], n_frames=n)
            # (a width 4 k + 2 ends in a shifted tile column -- tail == 2 -- and the tables' column records are per aligned quad: no tables there)
            assert d["geo"] == (0 if d["tail"] == 2 else geo), (w, h, dst, kw, d)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    if n == 1:
        got = v.Convert(ty, tuv, fp, width=w)
    else:
        got = v.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=kw.get("rt", BILINEAR), fourcc=kw.get("fourcc", 2), planes=kw.get("planes", 0),
                               normalization=kw.get("norm", False), nthreads=8, width=w)
    frames = [got] if n == 1 else [got[0], got[n // 2], got[n - 1]]
    for g in frames:
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (dst, kw, n, bad[:8], bad.size)


@pytest.mark.parametrize("src,dst,rt", [
    ((1920, 1080), (1280, 720), BILINEAR),   # integer window tile (quarters): the headline
    ((1920, 1080), (1600, 900), BILINEAR),   # float window tile (ratio 1.2)
    ((1920, 1080), (1366, 768), BILINEAR),   # 4 k + 2 columns: the shifted tile column, without tables (with them and the row tail under TSVPP_TAIL_SHIFT=0)
    ((1280, 720), (2560, 1440), BILINEAR),   # up-scale x2: clamped first column / row, repeated taps
    ((960, 540), (1280, 720), BILINEAR),     # up-scale x4/3 (dyadic: quarters)
    ((1280, 720), (1920, 1080), AREA),       # the AREA up-scale variant, float weights
    ((640, 360), (1280, 720), AREA),         # ... dyadic weights
    ((1920, 1088), (960, 544), BILINEAR),    # ratio exactly 2: the widest windows
])
@pytest.mark.parametrize("n", [1, 64])
def test_geo_ratio_classes(gvpp, oracle, src, dst, rt, n):
    vpp = gvpp
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + n)
    check(vpp, oracle, y, uv, src[0], dst, n=n, rt=rt, planes=0, norm=True)
    check(vpp, oracle, y, uv, src[0], dst, n=n, rt=rt, planes=1, norm=False)
    check(vpp, oracle, y, uv, src[0], dst, n=n, rt=rt, planes=0, norm=False)


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 1, True), (0, 1, False), (0, 1, True), (3, 1, False), (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_geo_output_flavours(gvpp, oracle, fourcc, planes, norm):
    vpp = gvpp
    y, uv = synth_nv12(960, 528, seed=170 + fourcc)
    check(vpp, oracle, y, uv, 960, (640, 352), fourcc=fourcc, planes=planes, norm=norm)          # 1.5: integer window tile
    check(vpp, oracle, y, uv, 960, (768, 400), fourcc=fourcc, planes=planes, norm=norm, n=64)    # 1.25 x 1.32: float window tile


def test_geo_crops_pitches_edges(gvpp, oracle):
    vpp = gvpp
    y, uv = synth_nv12(1000, 600, seed=115, pitch=1024)   # padded pitch, width not a multiple of 16
    check(vpp, oracle, y, uv, 1000, (750, 450), norm=True)
    check(vpp, oracle, y, uv, 1000, (500, 300), n=64)
    check(vpp, oracle, y, uv, 1000, (400, 300), crop=(3, 5, 603, 455), norm=True)    # odd origin: U / V swapped, misaligned rows
    check(vpp, oracle, y, uv, 1000, (322, 150), crop=(38, 6, 682, 306), planes=1)    # width 4 k + 2
    check(vpp, oracle, y, uv, 1000, (1500, 900), planes=1, n=64, geo=0)   # 2/3: not dyadic, outside the float window tile's range
    check(vpp, oracle, y, uv, 1000, (2000, 1200), planes=1, n=64)         # 1/2
    check(vpp, oracle, y, uv, 1000, (1200, 300), rt=AREA, crop=(200, 100, 1000, 300), norm=True)   # mixed: up in x, AREA -> bilinear variant
    for val in (0, 255):
        yy = np.full((360, 640), val, np.uint8)
        uu = np.full((180, 640), 255 - val, np.uint8)
        check(vpp, oracle, yy, uu, 640, (480, 270), planes=1)
        check(vpp, oracle, yy, uu, 640, (1280, 720), planes=1, n=64)
    y, uv = synth_nv12(64, 32, seed=116)
    for dst in [(48, 24), (128, 64), (56, 28), (32, 16)]:
        check(vpp, oracle, y, uv, 64, dst, norm=True)


def test_geo_tall_thread_tiles_and_knob_off(oracle):
    """TSVPP_RPT=4 (two further row-pair records per thread) and TSVPP_GEO=0 (vpp_bilinear_kernel) give the same bits."""
    import tensor_stream
    y, uv = synth_nv12(1920, 1080, seed=117)
    for env in ({"TSVPP_RPT": "4"}, {"TSVPP_RPT": "3"}, {"TSVPP_GEO": "0"}, {}):   # {}: the default selection (uint8 + dyadic weights: tables)
        with _Env(TSVPP_R32="0", **env):
            v = tensor_stream.VideoProcessor(device=0)
        fp = params((1280, 720))
        got = v.convert_batch([torch.from_numpy(y).cuda()] * 64, [torch.from_numpy(uv).cuda()] * 64, fp, width=1920)
        torch.cuda.synchronize()
        ref, _, _ = oracle.convert(y, uv, dst=(1280, 720), resize_type=BILINEAR, fourcc=2, planes=0, normalization=False, nthreads=8, width=1920)
        for f in (0, 63):
            assert np.array_equal(got[f].cpu().numpy().ravel(), ref), env
        v.Close()


@pytest.mark.parametrize("prepared", [True, False])
def test_geo_graph_capture(gvpp, oracle, prepared):
    vpp = gvpp
    """prepare() builds the geometry tables, so the FIRST conversion of a geometry can be captured into a graph; without
    prepare() a capturing stream never allocates: the launch keeps vpp_bilinear_kernel (same bits)."""
    src, dst = ((1280, 736), (1024, 576)) if prepared else ((1280, 736), (960, 552))
    y, uv = synth_nv12(src[0], src[1], seed=118)
    fp = params(dst, norm=True)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    if prepared:
        vpp.prepare(fp, src[0], src[1], n_frames=1)
    out = vpp._alloc(fp.parameters, src[0], src[1])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        vpp.Convert(ty, tuv, fp, out=out, width=src[0])
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, dst=dst, resize_type=BILINEAR, fourcc=2, planes=0, normalization=True, nthreads=8, width=src[0])
    assert np.array_equal(out.cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8))


@pytest.mark.parametrize("chunk", range(4))
def test_geo_fuzz(gvpp, oracle, chunk):
    """Seeded differential fuzzing of the 2x2-tap requests with 16-byte pitches (the geometry-table kernel wherever the window
    tiles apply, TSVPP_GEO=2): random sizes up to several tile columns / rows, crops with arbitrary origins, up- and down-scales,
    dyadic and non-dyadic ratios, every output flavour, single frames and small batches."""
    import tensor_stream as ts
    rng = np.random.default_rng(977 + chunk)
    for k in range(30):
        w = int(rng.integers(8, 400)) * 2
        h = int(rng.integers(4, 150)) * 2
        pitch = (w + 15) // 16 * 16 + 16 * int(rng.integers(0, 3))
        crop, sw, sh = (0, 0, 0, 0), w, h
        if rng.random() < 0.3:
            cw, ch = int(rng.integers(4, w // 2)) * 2, int(rng.integers(2, h // 2)) * 2
            l, t = int(rng.integers(0, w - cw + 1)), int(rng.integers(0, h - ch + 1))
            if cw < w and ch < h:
                crop, sw, sh = (l, t, l + cw, t + ch), cw, ch
        if rng.random() < 0.6:   # dyadic ratios: the integer window tile
            num, den = [(3, 2), (2, 1), (1, 2), (5, 4), (3, 4), (9, 8), (7, 4), (1, 1)][int(rng.integers(0, 8))]
            dst = (max(2, sw * den // num // 2 * 2), max(2, sh * den // num // 2 * 2))
        else:                    # anything from x0.5 to x2 down: float window tile between 1 and 1.45, float byte tile elsewhere
            dst = (max(2, int(sw / rng.uniform(0.5, 2.0)) // 2 * 2), max(2, int(sh / rng.uniform(0.5, 2.0)) // 2 * 2))
        if dst == (sw, sh):
            continue
        rt = int(rng.choice([BILINEAR, AREA]))
        fourcc = int(rng.choice([1, 2, 0, 3, 4, 5, 6]))
        planes = int(rng.integers(0, 2))
        norm = bool(rng.integers(0, 2)) or fourcc == 6
        n = int(rng.choice([1, 1, 3]))
        y, uv = synth_nv12(w, h, seed=7000 + 100 * chunk + k, pitch=pitch)
        fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
        ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=4, width=w)
        ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
        got = gvpp.Convert(ty, tuv, fp, width=w) if n == 1 else gvpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)[n - 1]
        torch.cuda.synchronize()
        g = got.cpu().numpy().ravel()
        assert g.size == ref.size, (w, h, pitch, crop, dst, rt, fourcc, planes, norm, n)
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, ((w, h, pitch, crop, dst, rt, fourcc, planes, norm, n), bad[:6], bad.size)
