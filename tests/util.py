"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np


def knob_run(allowed=()):
    """Is the suite being replayed under A/B knobs (tools/knob_matrix*.sh)?  Then tests that name the kernel a request must take stand down.  The gate itself
    (TSVPP_DEBUG_KNOBS, set by tests/conftest.py so that tests CAN use knobs) and the bench.py test switches are not knobs."""
    skip = {"TSVPP_DEBUG_KNOBS"} | set(allowed)
    return any(k.startswith("TSVPP_") and not k.startswith("TSVPP_BENCH_") and k not in skip for k in os.environ)


def synth_nv12(w, h, seed, pitch=None):
    """Full-range random NV12 frame (exercises clamps and sub-16 luma), optional row padding."""
    pitch = w if pitch is None else pitch
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 256, size=(h, pitch), dtype=np.uint8)
    uv = rng.integers(0, 256, size=(h // 2, pitch), dtype=np.uint8)
    return y, uv


def coverage_frame():
    """4096x4096 NV12 frame that enumerates all 2^24 (Y,U,V) triples: each of the 65536 (U,V) pairs
    owns 64 consecutive 2x2 blocks whose 256 luma samples are 0..255."""
    w = h = 4096
    blk = np.arange(2048 * 2048, dtype=np.int64).reshape(2048, 2048)  # chroma-block index
    pair = blk // 64
    uv = np.empty((2048, 4096), np.uint8)
    uv[:, 0::2] = (pair & 0xFF).astype(np.uint8)
    uv[:, 1::2] = (pair >> 8).astype(np.uint8)
    k = (blk % 64) * 4
    y = np.empty((h, w), np.uint8)
    y[0::2, 0::2] = k
    y[0::2, 1::2] = k + 1
    y[1::2, 0::2] = k + 2
    y[1::2, 1::2] = k + 3
    return y, uv


def ulp_diff(a, b):
    """Max distance in units-in-the-last-place between two float32 arrays of non-negative values."""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    return int(np.abs(ia - ib).max()) if ia.size else 0
