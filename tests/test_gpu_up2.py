"""GPU: the streaming kernel for the BILINEAR up-scale at exactly 1 : 2 with uint8 outputs (vpp_bilinear_up2_kernel: horizontal pair sums as v_dot4 on
the source dwords, neighbour dwords by wave shuffle, the 8 x 4-pixel output side of the streaming kernels) against the oracle, bit for bit: every
flavour it takes, one-lane and partial runs, the narrow (three loads per row) and the wide (shuffles) workgroups, frame edges, crops, batches, the
two-pass formats' first pass; requests it cannot take keep their kernels.  The thread-tile arithmetic itself is also checked on the CPU
(tests/test_bilinear_up2_cpu.py)."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
NEAREST, BILINEAR, BICUBIC, AREA = 0, 1, 2, 3
Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = 0, 1, 2, 3, 4, 5, 6
KNOBS = knob_run()


def check(vpp, oracle, y, uv, w, fourcc=RGB24, planes=0, crop=(0, 0, 0, 0), n=1, up2=True, norm=False, rt=BILINEAR, knob_ctx=False):
    import tensor_stream as ts
    sw, sh = (crop[2] - crop[0] or w), (crop[3] - crop[1] or y.shape[0])
    dst = (2 * sw, 2 * sh)
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    if not KNOBS:  # (KNOBS: the OUTER environment at import time -- tools/knob_matrix.sh; knobs a test sets itself are honoured by describe)
        k = ts.describe(fp, w, y.shape[0], pitch=y.shape[1], n_frames=n)["kernel"]
        assert k.startswith("vpp_bilinear_up2_kernel") == up2, (k, w, y.shape, dst, crop, fourcc, norm)
    ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
    got = vpp.Convert(ty, tuv, fp, width=w) if n == 1 else vpp.convert_batch([ty] * n, [tuv] * n, fp, width=w)
    torch.cuda.synchronize()
    ref, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8, width=w)
    for g in ([got] if n == 1 else [got[0], got[n - 1]]):
        g = g.cpu().numpy().ravel()
        assert g.size == ref.size
        bad = np.flatnonzero(g.view(np.uint8) != ref.view(np.uint8))
        assert bad.size == 0, (w, y.shape, dst, rt, fourcc, planes, norm, crop, n, bad[:8], bad.size)


@pytest.mark.parametrize("src,pitch", [((960, 540), 960), ((1920, 1080), 2048), ((640, 360), 640), ((4, 2), 4), ((8, 4), 8), ((12, 6), 16), ((24, 12), 24),
                                       ((100, 36), 100),    # 25 threads per row: one partial run (odd: direct stores of merged rows)
                                       ((104, 36), 112),    # 26 threads: an even partial run (exchanged)
                                       ((260, 20), 260),    # 65 threads: a full wave + a run of ONE lane (its own first and last)
                                       ((516, 10), 516)])   # 129 threads: two full waves + one lane
@pytest.mark.parametrize("fourcc,planes", [(RGB24, 0), (BGR24, 1), (RGB24, 1), (NV12, 1), (Y800, 1)])
def test_sizes_and_flavours(vpp, oracle, src, pitch, fourcc, planes):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + fourcc + planes, pitch=pitch)
    check(vpp, oracle, y, uv, src[0], fourcc=fourcc, planes=planes)


def test_batches_crops_two_pass_fallbacks(vpp, oracle):
    y, uv = synth_nv12(960, 540, seed=31, pitch=1024)
    check(vpp, oracle, y, uv, 960, fourcc=BGR24, planes=1, n=64)
    check(vpp, oracle, y, uv, 960, fourcc=RGB24, planes=0, n=3)
    check(vpp, oracle, y, uv, 960, crop=(4, 2, 484, 272))                        # origin a multiple of 4: pointers stay dword-aligned
    check(vpp, oracle, y, uv, 960, crop=(8, 7, 488, 277), planes=1)
    check(vpp, oracle, y, uv, 960, crop=(6, 2, 486, 272), up2=False)             # misaligned origin: the LDS kernel
    check(vpp, oracle, y, uv, 960, crop=(5, 3, 485, 273), planes=1, up2=False)   # odd origin (U / V swapped)
    check(vpp, oracle, y, uv, 960, norm=True, up2=False)                         # fp32 outputs stay on vpp_bilinear_kernel (output-bound there already)
    check(vpp, oracle, y, uv, 960, fourcc=HSV, planes=1, norm=True, up2=False)
    for rt in (NEAREST, BICUBIC, AREA):                                          # the other interpolations keep their kernels
        check(vpp, oracle, y, uv, 960, rt=rt, up2=False)
    for fcc in (UYVY, YUV444):                                                   # pass 1 of the two-pass formats writes NV12 with this kernel
        check(vpp, oracle, y, uv, 960, fourcc=fcc, planes=1)
        check(vpp, oracle, y, uv, 960, fourcc=fcc, planes=1, norm=True)
    y, uv = synth_nv12(962, 540, seed=32, pitch=964)
    check(vpp, oracle, y, uv, 962, up2=False)                                    # 1924 columns = 8 k + 4
    y, uv = synth_nv12(960, 542, seed=33)
    check(vpp, oracle, y, uv, 960)                                               # 271 row quads: a partial last tile row
    y, uv = synth_nv12(960, 540, seed=34, pitch=962)
    check(vpp, oracle, y, uv, 960, up2=False)                                    # pitch not a multiple of 4
    for val in (0, 255):                                                         # saturated planes: every sum at its extreme
        yy = np.full((36, 96), val, np.uint8)
        uu = np.full((18, 96), 255 - val, np.uint8)
        check(vpp, oracle, yy, uu, 96, planes=1)
        check(vpp, oracle, yy, uu, 96, planes=0)
    yy = (np.indices((36, 96)).sum(0) % 2 * 255).astype(np.uint8)                # checkerboard + hard frame edges
    uu = (np.indices((18, 96))[1] // 2 % 2 * 255).astype(np.uint8)
    yy[:, :1], yy[:, -1:], yy[:1], yy[-1:] = 255, 0, 0, 255
    check(vpp, oracle, yy, uu, 96, planes=1)
    check(vpp, oracle, yy, uu, 96, fourcc=NV12, planes=1)


@pytest.mark.parametrize("knobs", [{"TSVPP_R32": "2"}, {"TSVPP_R32": "2", "TSVPP_SHAPE": "32,8"}, {"TSVPP_SHAPE": "16,4"}, {"TSVPP_SHAPE": "128,2"}])
def test_fp32_flavours_and_other_workgroup_shapes(oracle, knobs, monkeypatch):
    """TSVPP_R32=2 routes the fp32 flavours here as well (through the shared output side); narrower workgroups take the three-loads-per-row path."""
    import tensor_stream as ts
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    v = ts.VideoProcessor(device=0)
    try:
        for src, pitch in (((960, 540), 960), ((260, 20), 260), ((24, 12), 24)):
            y, uv = synth_nv12(src[0], src[1], seed=src[0] + len(knobs), pitch=pitch)
            check(v, oracle, y, uv, src[0], fourcc=RGB24, planes=1, knob_ctx=True)
            check(v, oracle, y, uv, src[0], fourcc=BGR24, planes=0, knob_ctx=True)
            if knobs.get("TSVPP_R32") == "2":
                for fourcc, planes in ((BGR24, 0), (RGB24, 1), (NV12, 1), (Y800, 1), (HSV, 1)):
                    check(v, oracle, y, uv, src[0], fourcc=fourcc, planes=planes, norm=True, knob_ctx=True)
    finally:
        v.Close()


def test_r32_off_keeps_the_lds_kernel(oracle, monkeypatch):
    import tensor_stream as ts
    monkeypatch.setenv("TSVPP_R32", "0")
    v = ts.VideoProcessor(device=0)
    try:
        y, uv = synth_nv12(960, 540, seed=5)
        check(v, oracle, y, uv, 960, planes=1, up2=False, knob_ctx=True)
    finally:
        v.Close()
