"""GPU: the launch configurations bench.py TIMES, checked at full size.

The dispatcher's choice of kernel, tile shape and rows per thread depends on the number of frames in a launch, so a 2-frame parity
batch does not prove the 64-frame launch.  Here every BASELINE configuration (headline, C2, C3, C4, C5) and the headline's other
resize types run as ONE 64-frame tsvpp_convert_batch on random frames drawn on the device (as bench.py draws them, pitch 256-aligned),
and frames 0 / 31 / 63 are compared with the oracle bit for bit (uint8) / 0 ULP (fp32).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WORKLOADS: the single definition of the BASELINE configurations)

CASES = [("headline", None), ("headline", "NEAREST"), ("headline", "BICUBIC"), ("headline", "AREA"), ("c2", None), ("c3", None), ("c4", None), ("c5", None)]


@pytest.mark.parametrize("name,resize", CASES, ids=[n + ("-" + r if r else "") for n, r in CASES])
def test_full_size_64_frame_launch(vpp, oracle, name, resize):
    import tensor_stream as ts
    spec = list(bench.WORKLOADS[name])
    if resize:
        spec[5] = resize
    src_w, src_h, pitch, crop, dst, rt, fcc, planes, norm = spec
    B = 64
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=bench.RESIZE[rt], pixel_format=bench.FOURCC[fcc],
                            planes_pos=bench.PLANES[planes], normalization=norm)
    g = torch.Generator(device="cuda").manual_seed(77)
    ys = torch.randint(0, 256, (B, src_h, pitch), dtype=torch.uint8, device="cuda", generator=g)
    uvs = torch.randint(0, 256, (B, src_h // 2, pitch), dtype=torch.uint8, device="cuda", generator=g)
    vpp.prepare(fp, src_w, src_h, n_frames=B)
    out = vpp.convert_batch(ys, uvs, fp, width=src_w)
    torch.cuda.synchronize()
    for k in (0, 31, 63):
        ref, _, _ = oracle.convert(ys[k].cpu().numpy(), uvs[k].cpu().numpy(), crop=crop, dst=dst, resize_type=bench.RESIZE[rt], fourcc=bench.FOURCC[fcc],
                                   planes=bench.PLANES[planes], normalization=norm, nthreads=8, width=src_w)
        got = out[k].cpu().numpy().ravel()
        assert got.dtype == ref.dtype and got.size == ref.size
        bad = int((got.view(np.uint8) != ref.view(np.uint8)).sum())
        assert bad == 0, f"{name}/{rt}: frame {k} of the 64-frame launch differs from the oracle in {bad} bytes"
    del ys, uvs, out
    torch.cuda.empty_cache()
