#!/usr/bin/env python3
"""Writes tools/ref_capture/triples.bin: every (Y, U, V) triple on which two contraction variants of the reference's colour conversion
(reference src/ColorConversion.cu:23-36 as nvcc may have compiled it; oracle/vpp_oracle.c CT_COLOR_*) give different RGB -- the inputs that let ONE run of the real
reference binary decide which variant it is (tools/ref_capture/capture.cu runs them; tests/test_ref_capture.py reads the answer).  Test infrastructure."""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from util import coverage_frame  # noqa: E402

RESIZE = 1 | 2 | 8 | 16 | 64
INNER, OUTER, G_LEFT, G_RIGHT = 256, 512, 1024, 2048
VARIANTS = {f"inner{i}_g{g}_outer{o}": RESIZE | (INNER if i else 0) | {"plain": 0, "left": G_LEFT, "right": G_RIGHT}[g] | (OUTER if o else 0)
            for i, g, o in itertools.product((0, 1), ("plain", "left", "right"), (0, 1))}


def rgb_of(bits, y, uv):
    O.set_contract(bits)
    try:
        out, _, _ = O.convert(y, uv, fourcc=O.RGB24, planes=O.PLANAR, nthreads=O.host_cores())
    finally:
        O.set_contract(-1)
    return out.reshape(3, y.shape[0], y.shape[1]).copy()


def main():
    y, uv = coverage_frame()
    outs = {k: rgb_of(b, y, uv) for k, b in VARIANTS.items()}
    names = sorted(outs)
    diff = np.zeros(y.shape, bool)
    for a, b in itertools.combinations(names, 2):
        diff |= (outs[a] != outs[b]).any(axis=0)
    pos = np.argwhere(diff)
    trip = sorted({(int(y[i, j]), int(uv[i // 2, (j // 2) * 2]), int(uv[i // 2, (j // 2) * 2 + 1])) for i, j in pos})
    arr = np.array(trip, np.uint8)
    arr.tofile(os.path.join(HERE, "triples.bin"))
    classes = {}
    for k in names:
        classes.setdefault(outs[k][:, diff].tobytes(), []).append(k)
    print(f"{len(trip)} discriminating triples; {len(classes)} distinguishable classes of the {len(names)} variants:")
    for v in classes.values():
        print("  ", v)


if __name__ == "__main__":
    main()
