"""Regenerates tests/golden/ref_320x240.npz from the reference's own golden files.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py

Source: /root/reference/tests/resources/test_references/*_320x240.yuv -- the fp32 dumps the
reference's VPP tests compare against (reference tests/src/VPPTests.cpp:301-512).  They are stored
here unmodified (raw float32 bit patterns, npz-compressed).  The NV12 file doubles as the INPUT: every
value is k/255 in fp32 and rint(x*255) inverts it exactly, giving the 320x240 NV12 frame the other six
files were produced from.
"""
import os
import sys

import numpy as np

REF = "/root/reference/tests/resources/test_references"
NAMES = ["NV12Normalization", "RGB24Normalization", "BGR24Normalization", "Y800Normalization",
         "UYVYNormalization", "YUV444Normalization", "HSV"]


def main():
    out = {}
    for n in NAMES:
        a = np.fromfile(os.path.join(REF, n + "_320x240.yuv"), dtype=np.float32)
        out[n.replace("Normalization", "")] = a.view(np.uint32)  # keep exact bit patterns
    nv = out["NV12"].view(np.float32)
    q = np.rint(nv * 255)
    assert ((q.astype(np.float32) / np.float32(255)) == nv).all(), "NV12 golden is not exactly k/255"
    out["input_nv12_u8"] = q.astype(np.uint8)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_320x240.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    sys.exit(main())
