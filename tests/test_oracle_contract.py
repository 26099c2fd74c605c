"""CPU: the arithmetic contract "the reference as nvcc compiled it" (oracle/vpp_oracle.c, CT_NVCC), as ONE rule for the whole path.

The reference is CUDA built with nvcc's default -fmad=true: which multiply-add pairs became fused multiply-adds is not in its source text.  For the resize
kernels its own CRC goldens decide (tests/test_reference_crcs.py: exactly one of the candidate patterns reproduces all 38); that pattern is what the
non-aggressive fadd / fsub -> fma combine of an LLVM-based compiler produces (left operand first, single-use products only -- AREA's `divide += weight` whose
product has a second use is NOT fused, and the goldens confirm it).  No golden discriminates the colour conversion's variants, so since round 5 the same rule
is applied to it.  This file states what that changes and what it cannot change:
  * the default contract = the resize pattern + fused chroma products + the G subtraction's left product fused; the luma product (three uses) plain;
  * R and B are identical under EVERY colour variant for all 2^24 (Y, U, V) triples; G moves by one on exactly 36 triples against plain IEEE;
  * every reference-held literal (seven golden files, 38 CRCs) is reproduced under the new default (tests/test_oracle_golden.py, tests/test_reference_crcs.py
    run with it) -- and also under plain IEEE colour arithmetic: the goldens cannot decide, the rule does."""
import numpy as np

from util import coverage_frame

CT = dict(COORD=1, SUM_LEFT=2, SUM_RIGHT=4, SUM3=8, SUM4=16, AREA_DIV=32, AREA_SUM=64, AREAUP_COORD=128, COLOR_INNER=256, COLOR_OUTER=512, COLOR_G_LEFT=1024,
          COLOR_G_RIGHT=2048)
RESIZE = CT["COORD"] | CT["SUM_LEFT"] | CT["SUM3"] | CT["SUM4"] | CT["AREA_SUM"]
NVCC = RESIZE | CT["COLOR_INNER"] | CT["COLOR_G_LEFT"]


def _rgb(oracle, bits):
    y, uv = coverage_frame()
    oracle.set_contract(bits)
    try:
        out, _, _ = oracle.convert(y, uv, fourcc=oracle.RGB24, planes=oracle.PLANAR, nthreads=16)
    finally:
        oracle.set_contract(-1)
    return out.reshape(3, 4096, 4096).copy()


def test_default_contract_is_the_rule_applied_to_the_whole_path(oracle):
    assert np.array_equal(_rgb(oracle, -1), _rgb(oracle, NVCC))


def test_g_channel_differs_from_plain_ieee_on_36_triples_and_r_b_never(oracle):
    y, uv = coverage_frame()
    plain, new = _rgb(oracle, RESIZE), _rgb(oracle, NVCC)
    assert np.array_equal(plain[0], new[0]) and np.array_equal(plain[2], new[2])
    d = np.argwhere(plain[1] != new[1])
    assert len(d) == 36
    assert int(np.abs(plain[1].astype(int) - new[1].astype(int)).max()) == 1
    # each (Y, U, V) triple occurs exactly once in the coverage frame: 36 positions = 36 triples; list a few for the record
    trip = sorted((int(y[i, j]), int(uv[i // 2, (j // 2) * 2]), int(uv[i // 2, (j // 2) * 2 + 1])) for i, j in d)
    assert len(set(trip)) == 36
    # the fused chroma products alone change nothing: fma(c, v, 0.5) and round(c v) + 0.5 truncate alike once Y' is added, for every triple
    assert np.array_equal(_rgb(oracle, RESIZE | CT["COLOR_INNER"]), plain)


def test_r_and_b_are_invariant_under_every_colour_variant(oracle):
    plain = _rgb(oracle, RESIZE)
    worst = 0
    for inner in (0, CT["COLOR_INNER"]):
        for outer in (0, CT["COLOR_OUTER"]):
            for g in (0, CT["COLOR_G_LEFT"], CT["COLOR_G_RIGHT"]):
                v = _rgb(oracle, RESIZE | inner | outer | g)
                assert np.array_equal(v[0], plain[0]) and np.array_equal(v[2], plain[2]), (inner, outer, g)
                n = int((v[1] != plain[1]).sum())
                assert n <= 124 and int(np.abs(v[1].astype(int) - plain[1].astype(int)).max()) <= 1
                worst = max(worst, n)
    assert worst > 36  # other variants move more triples than the adopted one: the choice is not vacuous


def test_goldens_do_not_discriminate_the_colour_variants(golden, oracle):
    """The reference's RGB24 / BGR24 / HSV golden files under plain-IEEE colour arithmetic and under the other candidate variants too (under the default:
    tests/test_oracle_golden.py): limited-range content never reaches one of the few dozen triples."""
    try:
        for bits in (RESIZE, RESIZE | CT["COLOR_INNER"] | CT["COLOR_OUTER"] | CT["COLOR_G_RIGHT"], RESIZE | CT["COLOR_OUTER"]):
            oracle.set_contract(bits)
            for name, fcc in (("RGB24", 1), ("BGR24", 2), ("HSV", 6)):
                out, _, _ = oracle.convert(golden["Y"], golden["UVp"], fourcc=fcc, planes=oracle.MERGED, normalization=True, nthreads=4)
                assert np.array_equal(out.view(np.uint32), golden[name]), (bits, name)
    finally:
        oracle.set_contract(-1)
