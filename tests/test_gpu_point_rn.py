"""GPU: the streaming point sampler at exact integer ratios (vpp_point_rn.hip: NEAREST at 3 / 4 / 5 : 1 and the BILINEAR / BICUBIC requests whose weights are
all zero at 3 / 5 : 1 -- BASELINE config C4 is 4K -> 720p BICUBIC -> BGR24 merged uint8) against the oracle, bit for bit: every instance, every output
flavour, crops (aligned origins keep the kernel, odd ones fall back to the LDS point kernel: same bits), pitches, batches, widths that are no multiple of 8
(LDS kernel), and the plan check that it IS the kernel that ran."""
import os

import numpy as np
import pytest
import torch

from util import synth_nv12, knob_run

pytestmark = pytest.mark.gpu
KNOBS = knob_run()
NEAREST, BILINEAR, BICUBIC = 0, 1, 2


def run(vpp, oracle, y, uv, w, dst, rt, fourcc=2, planes=1, norm=False, crop=(0, 0, 0, 0), expect=None):
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
    ((3840, 2160), (1280, 720), BICUBIC, "vpp_point_rn_kernel<OUT,3:1,centre>"),   # BASELINE config C4's geometry
    ((3840, 2160), (1280, 720), BILINEAR, "vpp_point_rn_kernel<OUT,3:1,centre>"),
    ((3840, 2160), (1280, 720), NEAREST, "vpp_point_rn_kernel<OUT,3:1,nearest>"),
    ((1920, 1080), (640, 360), BICUBIC, "vpp_point_rn_kernel<OUT,3:1,centre>"),
    ((1920, 1088), (480, 272), NEAREST, "vpp_point_rn_kernel<OUT,4:1,nearest>"),
    ((1920, 1080), (384, 216), BILINEAR, "vpp_point_rn_kernel<OUT,5:1,centre>"),
    ((1920, 1080), (384, 216), BICUBIC, "vpp_point_rn_kernel<OUT,5:1,centre>"),
    ((1920, 1080), (384, 216), NEAREST, "vpp_point_rn_kernel<OUT,5:1,nearest>"),
    ((1920, 1080), (320, 180), NEAREST, "vpp_bilinear_rows_kernel<OUT,nearest>"),   # 6 : 1 -- no streaming instance: ratio product 36, the row-segment kernel (round 6; the LDS point kernel before)
    ((1920, 1080), (480, 216), NEAREST, "vpp_point_kernel"),                        # 4 x 5 = 20: the LDS point kernel
    ((1920, 1080), (1280, 360), BILINEAR, None),                                    # 1.5 x 3: not a point sampler
])
def test_instances(vpp, oracle, src, dst, rt, kernel):
    y, uv = synth_nv12(src[0], src[1], seed=src[0] + dst[0] + rt, pitch=(src[0] + 255) // 256 * 256)
    run(vpp, oracle, y, uv, src[0], dst, rt, planes=1, norm=False, expect=kernel)
    run(vpp, oracle, y, uv, src[0], dst, rt, planes=0, norm=True, expect=("vpp_point_kernel" if kernel and "point_rn" in kernel else kernel))  # fp32 outputs stay on the LDS point kernel


@pytest.mark.parametrize("fourcc,planes,norm", [(1, 0, False), (1, 1, True), (2, 0, True), (2, 1, False), (0, 1, False), (0, 1, True), (3, 1, False),
                                                 (3, 1, True), (6, 1, True), (4, 1, False), (5, 1, True)])
def test_output_flavours(vpp, oracle, fourcc, planes, norm):
    y, uv = synth_nv12(1920, 1080, seed=31 + fourcc, pitch=1920)
    run(vpp, oracle, y, uv, 1920, (640, 360), BICUBIC, fourcc=fourcc, planes=planes, norm=norm)   # 3 : 1 centre tap
    run(vpp, oracle, y, uv, 1920, (384, 216), NEAREST, fourcc=fourcc, planes=planes, norm=norm)   # 5 : 1


@pytest.mark.parametrize("pitch", [2048, 1924, 1922])
def test_crops_pitches_and_widths(vpp, oracle, pitch):
    y, uv = synth_nv12(1920, 1080, seed=7 + pitch, pitch=pitch)
    k = "vpp_point_rn_kernel" if pitch % 4 == 0 else "vpp_point_kernel"
    run(vpp, oracle, y, uv, 1920, (400, 240), BICUBIC, crop=(120, 60, 1320, 780), expect=k)              # aligned origin: 1200 x 720 -> 400 x 240
    run(vpp, oracle, y, uv, 1920, (400, 240), BICUBIC, crop=(121, 61, 1321, 781), planes=0, norm=True)   # odd origin (U / V swap quirk): planes not dword-aligned -> LDS kernel
    run(vpp, oracle, y, uv, 1920, (404, 240), NEAREST, crop=(0, 0, 1212, 720))                           # width no multiple of 8 -> LDS kernel
    run(vpp, oracle, y, uv, 1920, (400, 242), NEAREST, crop=(0, 0, 1200, 726), planes=0)                 # height no multiple of 4 -> LDS kernel
    run(vpp, oracle, y, uv, 1920, (8, 4), NEAREST, crop=(1880, 1060, 1920, 1080), fourcc=1)              # one thread tile in the bottom-right corner (5 : 1)
    run(vpp, oracle, y, uv, 1920, (640, 360), BILINEAR, fourcc=1, planes=1)


def test_batch(vpp, oracle):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=640, height=360, resize_type=BICUBIC, pixel_format=2, planes_pos=1, normalization=False)
    n = 70
    g = torch.Generator(device="cuda").manual_seed(4)
    ys = torch.randint(0, 256, (n, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
    uvs = torch.randint(0, 256, (n, 540, 1920), dtype=torch.uint8, device="cuda", generator=g)
    out = vpp.convert_batch(ys, uvs, fp, width=1920)
    torch.cuda.synchronize()
    for f in (0, 33, 69):
        ref, _, _ = oracle.convert(ys[f].cpu().numpy(), uvs[f].cpu().numpy(), dst=(640, 360), resize_type=BICUBIC, fourcc=2, planes=1, normalization=False, nthreads=8, width=1920)
        assert np.array_equal(out[f].cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8)), f
