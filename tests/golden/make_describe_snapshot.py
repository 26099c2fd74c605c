"""Writes tests/golden/describe_snapshot.json: what tsvpp_describe (the dry run of launch_fused: kernel, workgroup shape, rows per thread, tiles)
answers for ~240 requests -- every resize type x common geometries x the output flavours that steer the selection.  The CPU suite compares the live
answers with this file (tests/test_describe_snapshot_cpu.py), so a change to a selection heuristic shows up in review as a diff of this file:
    python tests/golden/make_describe_snapshot.py        # regenerate after an intended change
Host logic only (no GPU)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tensor-stream_amd"))

import tensor_stream as ts  # noqa: E402

GEOMETRIES = [  # (src, dst, pitch)
    ((1920, 1080), (1280, 720), 2048), ((1920, 1080), (960, 540), 2048), ((1920, 1080), (640, 360), 2048), ((1920, 1080), (224, 224), 2048),
    ((1920, 1080), (300, 300), 2048), ((1920, 1080), (640, 640), 2048), ((1920, 1080), (1440, 810), 2048), ((1920, 1080), (1536, 864), 2048),
    ((1920, 1080), (1366, 768), 2048), ((1920, 1080), (0, 0), 2048), ((3840, 2160), (1920, 1080), 3840), ((3840, 2160), (1280, 720), 3840),
    ((3840, 2160), (640, 360), 3840), ((3840, 2160), (2560, 1440), 3840), ((3840, 2160), (608, 342), 3840), ((3840, 2160), (300, 300), 3840),
    ((1280, 720), (1920, 1080), 1280), ((1280, 720), (256, 256), 1280), ((1280, 720), (854, 480), 1280), ((1080, 608), (480, 360), 1088),
    ((960, 540), (1920, 1080), 960), ((640, 360), (1280, 720), 640), ((1920, 1080), (1280, 720), 1922), ((1926, 1080), (1284, 720), 1926),
]
FLAVOURS = [("BGR24", "PLANAR", True), ("RGB24", "MERGED", False), ("RGB24", "PLANAR", False), ("RGB24", "MERGED", True)]
EXTRA = [("Y800", "MERGED", False), ("NV12", "MERGED", False), ("UYVY", "MERGED", False), ("YUV444", "MERGED", False), ("HSV", "MERGED", True)]
RT = {"NEAREST": 0, "BILINEAR": 1, "BICUBIC": 2, "AREA": 3}
FCC = {"Y800": 0, "RGB24": 1, "BGR24": 2, "NV12": 3, "UYVY": 4, "YUV444": 5, "HSV": 6}
KEYS = ("mode", "out", "kernel", "shape", "rpt", "dma", "tiles", "tail", "geo")


def requests():
    for src, dst, pitch in GEOMETRIES:
        rts = ["NEAREST"] if dst == (0, 0) else list(RT)
        for rt in rts:
            flavours = FLAVOURS + (EXTRA if (src, dst) in (((1920, 1080), (1280, 720)), ((1920, 1080), (0, 0)), ((3840, 2160), (1920, 1080))) and pitch % 16 == 0 else [])
            for fcc, planes, norm in flavours:
                yield src, dst, pitch, rt, fcc, planes, norm


def describe(src, dst, pitch, rt, fcc, planes, norm):
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=RT[rt], pixel_format=FCC[fcc], planes_pos=0 if planes == "PLANAR" else 1, normalization=norm)
    try:
        d = ts.describe(fp, src[0], src[1], pitch=pitch, n_frames=64)
        return {k: d.get(k) for k in KEYS}
    except RuntimeError as e:  # (two-pass formats answer for their first pass or refuse: recorded as they are)
        return {"error": str(e).split(":")[0]}


def key(src, dst, pitch, rt, fcc, planes, norm):
    return f"{src[0]}x{src[1]}/{pitch}->{dst[0]}x{dst[1]} {rt} {fcc} {planes} {'f32' if norm else 'u8'}"


def snapshot():
    return {key(*r): describe(*r) for r in requests()}


if __name__ == "__main__":
    snap = snapshot()
    with open(os.path.join(HERE, "describe_snapshot.json"), "w") as f:
        json.dump(snap, f, indent=0, sort_keys=True)
        f.write("\n")
    print(len(snap), "requests")
