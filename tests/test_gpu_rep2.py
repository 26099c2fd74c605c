"""GPU: 2 x 2 pixel replication as a streaming kernel (vpp_rep2_kernel, vpp_point_rn.hip) -- NEAREST and the AREA up-scale at exactly 1 : 2 (the AREA
up-scale's weights are all zero at that ratio: the reference's blend returns the tap itself) -- against the oracle, bit for bit: every flavour, one-lane
and partial runs, crops, batches, the two-pass formats' first pass, fp32 under TSVPP_R32=2; requests it cannot take keep their kernels."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
NEAREST, BILINEAR, BICUBIC, AREA = 0, 1, 2, 3
Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = 0, 1, 2, 3, 4, 5, 6
KNOBS = knob_run()


def check(vpp, oracle, y, uv, w, rt, fourcc=RGB24, planes=0, crop=(0, 0, 0, 0), n=1, rep2=True, norm=False, knob_ctx=False):
    import tensor_stream as ts
    sw, sh = (crop[2] - crop[0] or w), (crop[3] - crop[1] or y.shape[0])
    dst = (2 * sw, 2 * sh)
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if not KNOBS:  # (KNOBS: the OUTER environment at import time -- tools/knob_matrix.sh; knobs a test sets itself are honoured by describe)
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=n)["kernel"]
        assert k.startswith("vpp_rep2_kernel") == rep2, (k, w, y.shape, dst, crop, fourcc, norm)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    got = vpp.Convert(ty, tuv, fp, width=w) if n == 1 else vpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    for g in ([got] if n == 1 else [got[0], got[n - 1]]):
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (w, y.shape, dst, rt, fourcc, planes, norm, crop, n, bad[:8], bad.size)


@pytest.mark.parametrize("rt", [NEAREST, AREA])
@pytest.mark.parametrize("src,pitch", [((960, 540), 960), ((1920, 1080), 2048), ((4, 2), 4), ((12, 6), 16), ((100, 36), 100), ((104, 36), 112), ((260, 20), 260), ((516, 10), 516)])
@pytest.mark.parametrize("fourcc,planes", [(RGB24, 0), (BGR24, 1), (NV12, 1), (Y800, 1)])
def test_sizes_and_flavours(vpp, oracle, rt, src, pitch, fourcc, planes):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + fourcc + planes + rt, pitch=pitch)
    check(vpp, oracle, y, uv, src[0], rt, fourcc=fourcc, planes=planes)


def test_batches_crops_two_pass_fallbacks(vpp, oracle):
    y, uv = synth_nv12(960, 540, seed=31, pitch=1024)
    for rt in (NEAREST, AREA):
        check(vpp, oracle, y, uv, 960, rt, fourcc=BGR24, planes=1, n=64)
        check(vpp, oracle, y, uv, 960, rt, crop=(4, 2, 484, 272))                        # origin a multiple of 4: pointers stay dword-aligned
        check(vpp, oracle, y, uv, 960, rt, crop=(5, 3, 485, 273), planes=1, rep2=False)   # odd origin (U / V swapped): the LDS kernels
        check(vpp, oracle, y, uv, 960, rt, norm=True, rep2=False)                         # fp32 outputs stay where they were
        for fcc in (UYVY, YUV444):                                                        # pass 1 of the two-pass formats writes NV12 with this kernel
            check(vpp, oracle, y, uv, 960, rt, fourcc=fcc, planes=1)
    y, uv = synth_nv12(962, 540, seed=32, pitch=964)
    check(vpp, oracle, y, uv, 962, NEAREST, rep2=False)                                   # 1924 columns = 8 k + 4
    y, uv = synth_nv12(960, 542, seed=33)
    check(vpp, oracle, y, uv, 960, AREA, planes=1)                                        # 271 row quads: a partial last tile row


@pytest.mark.parametrize("knobs", [{"TSVPP_R32": "2"}, {"TSVPP_SHAPE": "16,4"}])
def test_fp32_flavours_and_other_workgroup_shapes(oracle, knobs, monkeypatch):
    import tensor_stream as ts
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    v = ts.VideoProcessor(device=0)
    try:
        for src, pitch in (((960, 540), 960), ((260, 20), 260)):
            y, uv = synth_nv12(src[0], src[1], seed=src[0] + len(knobs), pitch=pitch)
            for rt in (NEAREST, AREA):
                check(v, oracle, y, uv, src[0], rt, fourcc=RGB24, planes=1, knob_ctx=True)
                if knobs.get("TSVPP_R32") == "2":
                    for fourcc, planes in ((BGR24, 0), (RGB24, 1), (NV12, 1), (Y800, 1), (HSV, 1)):
                        check(v, oracle, y, uv, src[0], rt, fourcc=fourcc, planes=planes, norm=True, knob_ctx=True)
    finally:
        v.Close()
