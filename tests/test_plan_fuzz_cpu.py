"""CPU: seeded fuzzing of the host-side planner (tsvpp_describe = launch_fused's dry run, including the geometry-table evaluation and
the streaming-kernel / box-kernel eligibility rules): every request either is refused with the reference's status codes or gets a
launchable plan -- a kernel name, a workgroup of at most 256 threads, a grid, LDS within the hardware's 64 KiB per workgroup."""
import numpy as np
import pytest

import tensor_stream as ts
from tensor_stream import _native as N


def random_request(rng):
    w = int(rng.integers(1, 2049)) * 2
    h = int(rng.integers(1, 1201)) * 2
    pitch = w + int(rng.choice([0, 0, 0, 2, 4, 6, 16, 30, 64]))
    crop = (0, 0, 0, 0)
    sw, sh = w, h
    if rng.random() < 0.3 and w >= 8 and h >= 8:
        cw, ch = int(rng.integers(1, w // 2)) * 2, int(rng.integers(1, h // 2)) * 2
        l, t = int(rng.integers(0, w - cw + 1)), int(rng.integers(0, h - ch + 1))
        if cw < w and ch < h:
            crop, sw, sh = (l, t, l + cw, t + ch), cw, ch
    kind = rng.random()
    if kind < 0.1:
        dst = (0, 0)
    elif kind < 0.55:  # exact ratios: the integer / streaming / box fast paths
        num, den = [(3, 2), (2, 1), (5, 2), (4, 1), (1, 2), (5, 4), (3, 4), (6, 1), (3, 1), (8, 1), (5, 1), (9, 4), (7, 1), (2, 3)][int(rng.integers(0, 14))]
        dst = (max(2, sw * den // num // 2 * 2), max(2, sh * den // num // 2 * 2))
        if rng.random() < 0.5 and crop == (0, 0, 0, 0):   # an exactly divisible source (what the streaming kernel needs), pitch a multiple of 4
            dst = (int(rng.integers(1, 161)) * 8, int(rng.integers(1, 121)) * 4)
            if (dst[0] * num) % (2 * den) == 0 and (dst[1] * num) % (2 * den) == 0:
                w, h = dst[0] * num // den, dst[1] * num // den
                pitch = (w + 3) // 4 * 4
    else:
        dst = (int(rng.integers(1, 1200)) * 2, int(rng.integers(1, 700)) * 2)
    return dict(w=w, h=h, pitch=pitch, crop=crop, dst=dst, rt=int(rng.integers(0, 4)), fourcc=int(rng.integers(0, 7)), planes=int(rng.integers(0, 2)),
                norm=bool(rng.integers(0, 2)), n=int(rng.choice([1, 2, 8, 64])), aligned=bool(rng.random() < 0.9))


@pytest.mark.parametrize("chunk", range(4))
def test_every_request_gets_a_launchable_plan_or_a_reference_status(chunk):
    rng = np.random.default_rng(555 + chunk)
    kernels = set()
    tails = {1: 0, 2: 0}
    for _ in range(400):
        r = random_request(rng)
        fp = ts.FrameParameters(width=r["dst"][0], height=r["dst"][1], crop_coords=r["crop"], resize_type=r["rt"], pixel_format=r["fourcc"],
                                planes_pos=r["planes"], normalization=r["norm"])
        try:
            p = ts.describe(fp, r["w"], r["h"], pitch=r["pitch"], n_frames=r["n"], aligned_outputs=r["aligned"])
        except RuntimeError as e:   # refused: must be one of the reference's status codes, never a HIP error or a crash
            assert "VREADER_UNSUPPORTED" in str(e) or "VREADER_ERROR" in str(e), (r, str(e))
            continue
        if p["kernel"] == "(none)":   # UYVY / YUV444 without a resize: the format pass alone
            assert p.get("pass2") in ("fmt_uyvy", "fmt_yuv444"), (r, p)
            continue
        tx, ty = (int(v) for v in p["shape"].split("x"))
        assert p["kernel"] and tx >= 1 and ty >= 1 and tx * ty <= 256, (r, p)
        assert 0 <= p["lds"] <= 64 * 1024 and p["grid"] >= 1 and p["rpt"] >= 1, (r, p)
        tw, th = (int(v) for v in p["tiles"].split("x"))
        assert tw >= 1 and th >= 1, (r, p)
        kernels.add(p["kernel"].split("<")[0])
        # how a row ends (include/tsvpp.h): only outputs 4 k + 2 columns wide on vector-store kernels need anything; a shifted tile column (2) only where a
        # kernel with its own resize takes the request and the output is at least two tiles wide; never for the box / colour / streaming / copy kernels
        dw = r["dst"][0] or (r["crop"][2] - r["crop"][0] or r["w"])
        vec = r["aligned"] or r["fourcc"] in (4, 5)   # (UYVY / YUV444: the first pass writes the library's own, aligned scratch frames)
        if dw % 4 == 0 or not vec:
            assert p["tail"] == 0, (r, p)
        else:
            assert p["tail"] in (1, 2), (r, p)
            if p["tail"] == 2:
                assert tw >= 2 and not any(k in p["kernel"] for k in ("area_box", "color", "r32", "copy16")), (r, p)
            tails[p["tail"]] += 1
    assert tails[1] > 0 and tails[2] > 0, tails
    # the sweep reaches every kernel family of the fused launch
    assert {"vpp_bilinear_kernel", "vpp_bilinear_r32_kernel", "vpp_area_box_kernel", "vpp_point_kernel", "vpp_color_kernel"} <= kernels, kernels
