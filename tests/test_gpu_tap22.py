"""GPU: the AREA down-scale at exactly 3 : 2 / 2 : 1 with fp32 outputs on the 2x2-tap kernel's integer window tile (LaunchDesc::tap22: the same two taps
per axis as BILINEAR, its own integer weights, a division instead of the shift) against the oracle, bit for bit: every fp32 flavour it takes,
ragged pitches and crops (staging with per-row misalignment), frame edges, batches; requests it must not take keep their kernels."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
NEAREST, BILINEAR, AREA = 0, 1, 3
KNOBS = knob_run()


def check(vpp, oracle, y, uv, w, dst, rt, fourcc=2, planes=0, norm=True, crop=(0, 0, 0, 0), n=1, expect=True):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if not KNOBS and expect is not None:
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=n)["kernel"]
        assert ("-weights]" in k) == expect, (k, w, y.shape, dst, rt, fourcc, crop)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    got = vpp.Convert(ty, tuv, fp, width=w) if n == 1 else vpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    for g in ([got] if n == 1 else [got[0], got[n - 1]]):
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (w, y.shape, dst, rt, fourcc, planes, crop, n, bad[:8], bad.size)


@pytest.mark.parametrize("fourcc,planes", [(2, 0), (1, 1), (3, 1), (0, 1)])  # fp32 planar / merged, NV12, Y800 (round 6)
@pytest.mark.parametrize("src,pitch,dst,rt", [((960, 540), 960, (640, 360), AREA), ((960, 540), 1001, (640, 360), AREA),                                               ((1280, 720), 1280, (640, 360), AREA), ((1288, 724), 1290, (644, 362), AREA),
                                              ((48, 24), 48, (32, 16), AREA), ((16, 8), 16, (8, 4), AREA)])
def test_flavours_pitches_sizes(vpp, oracle, src, pitch, dst, rt, fourcc, planes):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + rt + fourcc, pitch=pitch)
    check(vpp, oracle, y, uv, src[0], dst, rt, fourcc=fourcc, planes=planes)


def test_full_sizes_batches_crops_and_what_it_leaves_alone(vpp, oracle):
    y, uv = synth_nv12(1920, 1080, seed=41, pitch=2048)
    check(vpp, oracle, y, uv, 1920, (1280, 720), AREA, n=64)
    check(vpp, oracle, y, uv, 1920, (1280, 720), NEAREST, planes=1, n=3, expect=False)                     # NEAREST: the point sampler reads fewer rows
    check(vpp, oracle, y, uv, 1920, (960, 540), AREA, planes=1, n=3)
    check(vpp, oracle, y, uv, 1920, (640, 360), AREA, crop=(13, 7, 973, 547))                  # odd origin (U / V swap quirk), 3 : 2
    check(vpp, oracle, y, uv, 1920, (640, 360), AREA, crop=(64, 32, 1024, 572), planes=1)
    check(vpp, oracle, y, uv, 1920, (480, 270), AREA, crop=(2, 2, 962, 542))                   # 2 : 1, 270 rows
    check(vpp, oracle, y, uv, 1920, (1280, 720), AREA, fourcc=0, planes=1, expect=True)        # Y800 fp32 (round 6, two row pairs per thread: 0.62 -> 0.74; slower before that)
    check(vpp, oracle, y, uv, 1920, (1280, 720), AREA, norm=False, expect=False)               # uint8: the streaming kernel
    check(vpp, oracle, y, uv, 1920, (1280, 720), AREA, fourcc=6, planes=1, expect=False)       # HSV: the streaming kernel
    check(vpp, oracle, y, uv, 1920, (1280, 540), AREA, expect=False)                           # 1.5 x 2: not one of the two ratios
    y, uv = synth_nv12(1924, 1080, seed=42)
    check(vpp, oracle, y, uv, 1924, (962, 540), AREA, expect=False)                            # 4 k + 2 columns: the tail launch samples by mode
    y4, uv4 = synth_nv12(3840, 2160, seed=43)
    check(vpp, oracle, y4, uv4, 3840, (1920, 1080), AREA)
    check(vpp, oracle, y4, uv4, 3840, (2560, 1440), AREA, planes=1)


def test_extreme_frames(vpp, oracle):
    y = (np.indices((360, 960)).sum(0) % 2 * 255).astype(np.uint8)
    uv = (np.indices((180, 960))[1] // 2 % 2 * 255).astype(np.uint8)
    y[:, :2], y[:, -2:], y[:2], y[-2:] = 255, 0, 0, 255
    for dst, rt in [((640, 240), AREA), ((480, 180), AREA)]:
        check(vpp, oracle, y, uv, 960, dst, rt)
        check(vpp, oracle, y, uv, 960, dst, rt, planes=1, fourcc=1)


def test_a_tile_that_cannot_be_staged_runs_as_the_area_request_it_is(oracle, monkeypatch):
    """sel_tap22 used to turn the AREA request into BILINEAR before its staging was known to fit; with an LDS budget nothing fits (TSVPP_LDS_KB=2, read
    when a context is created) the launch fell through to the gather kernel -- with BILINEAR's weights.  It now starts over as AREA: same bits as the oracle."""
    import tensor_stream as ts
    monkeypatch.setenv("TSVPP_LDS_KB", "2")
    v = ts.VideoProcessor(device=0, max_consumers=1)
    try:
        y, uv = synth_nv12(1920, 1080, seed=44, pitch=2048)
        fp = ts.FrameParameters(width=1280, height=720, resize_type=AREA, pixel_format=2, planes_pos=0, normalization=True)
        if not knob_run(("TSVPP_LDS_KB",)):  # (knob runs route the request elsewhere on purpose)
            assert ts.describe(fp, 1920, 1080, pitch=2048)["kernel"].startswith("vpp_fused_gather_kernel")  # (describe reads the same knob)
        got = v.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=1920)
        torch.cuda.synchronize()
        ref, _, _ = oracle.convert(y, uv, dst=(1280, 720), resize_type=AREA, fourcc=2, planes=0, normalization=True, nthreads=8, width=1920)
        assert np.array_equal(got.cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8))
    finally:
        v.Close()
