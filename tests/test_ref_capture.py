"""CPU: the two parity questions only the real reference binary can answer.

tools/ref_capture/capture.cu, built on an NVIDIA box inside the reference's own environment, runs the reference's unmodified kernels on (i) the 155 (Y, U, V)
triples on which the six distinguishable fused-multiply-add variants of its colour conversion differ (reference src/ColorConversion.cu:23-36) and (ii) frame 0 of the
reference's test clip through its BICUBIC resize to 480 x 360 (src/Resize.cu:27-91, 314-357: non-dyadic weights, pow() in fp64).  Its two output files go to
tests/golden/ref_capture/; until they exist the decisive tests below are SKIPPED and the oracle keeps the variant its resize goldens imply (CT_NVCC) and the
correctly rounded pow.  The remaining tests keep the kit itself honest: the triple list is complete and the reader recognises every variant."""
import itertools
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "ref_capture")
CAP = os.path.join(ROOT, "tests", "golden", "ref_capture")
RESIZE = 1 | 2 | 8 | 16 | 64
INNER, OUTER, G_LEFT, G_RIGHT = 256, 512, 1024, 2048
NVCC = RESIZE | INNER | G_LEFT
CLASSES = {f"g_{g}{'_outer' if o else ''}": RESIZE | INNER | {"plain": 0, "left": G_LEFT, "right": G_RIGHT}[g] | (OUTER if o else 0)
           for g, o in itertools.product(("plain", "left", "right"), (0, 1))}


def triples():
    return np.fromfile(os.path.join(KIT, "triples.bin"), np.uint8).reshape(-1, 3)


def frame_of(trip):
    """The NV12 frame capture.cu builds: triple k owns the 2 x 2 block at columns 2 k, 2 k + 1."""
    n = len(trip)
    y = np.repeat(trip[:, 0], 2)[None, :].repeat(2, axis=0).copy()
    uv = trip[:, 1:3].reshape(1, 2 * n).copy()
    return y, uv


def rgb_under(oracle, bits, y, uv):
    oracle.set_contract(bits)
    try:
        return oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.MERGED)[0].copy()
    finally:
        oracle.set_contract(-1)


def classify(oracle, capture):
    """Names of the variant classes whose output equals the captured bytes."""
    y, uv = frame_of(triples())
    return [k for k, bits in CLASSES.items() if np.array_equal(rgb_under(oracle, bits, y, uv), capture)]


def test_the_triple_list_separates_all_six_variant_classes(oracle):
    y, uv = frame_of(triples())
    outs = {k: rgb_under(oracle, b, y, uv) for k, b in CLASSES.items()}
    for a, b in itertools.combinations(outs, 2):
        assert not np.array_equal(outs[a], outs[b]), (a, b)
    # ... and the reader recognises each of them (a stand-in capture made by the oracle itself)
    for k, o in outs.items():
        assert classify(oracle, o) == [k]
    assert len(triples()) == 155 and NVCC == CLASSES["g_left"]


def test_the_harness_only_includes_the_reference():
    """capture.cu must stay a harness: it #includes the reference's .cu files from a checkout and holds none of their text."""
    src = open(os.path.join(KIT, "capture.cu")).read()
    assert '#include "ColorConversion.cu"' in src and '#include "Resize.cu"' in src
    assert "__global__" not in src and "1.163999557" not in src and len(src.splitlines()) <= 100


needs_capture = pytest.mark.skipif(not (os.path.isfile(os.path.join(CAP, "g_triples_rgb.bin")) and os.path.isfile(os.path.join(CAP, "bicubic_480x360_nv12.bin"))),
                                   reason="no capture of the real reference binary yet: run tools/ref_capture/capture.cu on an NVIDIA box (see its header)")


@needs_capture
def test_colour_contraction_of_the_real_reference_binary(oracle):
    cap = np.fromfile(os.path.join(CAP, "g_triples_rgb.bin"), np.uint8)
    hit = classify(oracle, cap)
    assert hit, "the capture matches none of the six variants: not the reference's colour kernel, or not these triples"
    assert hit == ["g_left"], f"the reference binary contracts its green term as {hit}: set TSVPP_OPT_COLOR_G_TERM / the oracle's CT_NVCC accordingly"


@needs_capture
def test_bicubic_pow_of_the_real_reference_binary(oracle):
    nv = np.fromfile(os.path.join(ROOT, "tests", "golden", "bbb_1080x608_frame0.nv12"), np.uint8)
    y, uv = nv[: 1080 * 608].reshape(608, 1080), nv[1080 * 608:].reshape(304, 1080)
    oy, ouv = oracle.resize_stage(y, uv, 480, 360, oracle.BICUBIC, nthreads=oracle.host_cores())
    cap = np.fromfile(os.path.join(CAP, "bicubic_480x360_nv12.bin"), np.uint8)
    mine = np.concatenate([oy.ravel(), ouv.ravel()])
    assert cap.size == mine.size
    bad = int((cap != mine).sum())
    assert bad == 0, f"{bad} of {mine.size} resized samples differ from the reference binary's: its pow(w, 2) / pow(w, 3) are not the exact square / correctly rounded cube (oracle/pow_pin.c)"
