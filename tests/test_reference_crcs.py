"""The reference's 38 CRC-32 goldens (tests/golden/reference_crcs.py), replayed on frame 0 of its own test clip -- the only
byte-exact pins the reference holds for the interpolating resize kernels (BILINEAR / BICUBIC / AREA at non-dyadic ratios,
crop + resize, UYVY, YUV444, NV12).  The input frame `tests/golden/bbb_1080x608_frame0.nv12` is decoded from
tests/resources/bbb_1080x608_420_10.h264 by the intra decoder of tests/golden/h264_intra.py (make_bbb_frame0.py) and is
itself pinned by the decoder test's plane CRCs (tests/src/DecoderTests.cpp:63-65), which also validates the av_crc
restatement.  The oracle passes all 38 with exactly one contraction pattern (oracle/vpp_oracle.c, CT_NVCC): that is what pins
the reference's arithmetic AS COMPILED by nvcc; the HIP kernels reproduce the same 38 literals on the GPU."""
import os

import numpy as np
import pytest

from golden.reference_crcs import GOLDENS, INPUT_PLANE_CRCS

FRAME = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bbb_1080x608_frame0.nv12")
needs_frame = pytest.mark.skipif(not os.path.exists(FRAME), reason="decoded frame 0 of bbb_1080x608_420_10.h264 not available (no decoder in the image)")


def load():
    a = np.fromfile(FRAME, dtype=np.uint8)
    assert a.size == 1080 * 608 * 3 // 2
    return a[: 1080 * 608].reshape(608, 1080), a[1080 * 608:].reshape(304, 1080)


def test_table_is_complete():
    assert len(GOLDENS) == 38 and all(len(g) == 7 for g in GOLDENS)
    assert {g[4] for g in GOLDENS} == {0, 1, 2, 3}          # every resize type has a CRC pin in the reference


@needs_frame
def test_input_frame_and_crc_definition(oracle):
    y, uv = load()
    assert oracle.av_crc32_ieee(y) == INPUT_PLANE_CRCS["Y"]
    assert oracle.av_crc32_ieee(uv) == INPUT_PLANE_CRCS["UV"]


@needs_frame
@pytest.mark.parametrize("g", GOLDENS, ids=[g[0] for g in GOLDENS])
def test_oracle_reproduces_reference_crc(oracle, g):
    _, fcc, planes, dst, rt, crop, crcs = g
    y, uv = load()
    out, _, _ = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fcc, planes=planes, normalization=False, nthreads=8)
    assert oracle.av_crc32_ieee(out) in crcs


@needs_frame
@pytest.mark.gpu
@pytest.mark.parametrize("g", GOLDENS, ids=[g[0] for g in GOLDENS])
def test_hip_reproduces_reference_crc(vpp, oracle, g):
    import torch
    import tensor_stream as ts
    _, fcc, planes, dst, rt, crop, crcs = g
    y, uv = load()
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fcc, planes_pos=planes, normalization=False)
    out = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp)
    torch.cuda.synchronize()
    assert oracle.av_crc32_ieee(out.cpu().numpy()) in crcs
