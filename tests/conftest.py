import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensor-stream_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The tests drive the library's A/B knobs (TSVPP_GEO, TSVPP_R32, ...): those are honoured only under this gate (tsvpp_api.cpp: read_env_knobs)
os.environ.setdefault("TSVPP_DEBUG_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A kernel that never returns must fail its test, not eat the box: every GPU test gets a time limit (pytest-timeout, when installed)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def golden():
    """The reference's own fp32 golden files (tests/golden/make_golden.py) + the recovered NV12 input."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_320x240.npz"))
    g = {k: z[k] for k in z.files}
    nv = g["input_nv12_u8"]
    g["Y"] = nv[: 320 * 240].reshape(240, 320).copy()
    g["UVp"] = nv[320 * 240:].reshape(120, 320).copy()
    return g


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def vpp():
    """HIP VideoProcessor on cuda:0 (gpu tests only).  No fallback: a missing library is an error."""
    import torch
    assert torch.cuda.is_available(), "gpu test on a box without a GPU"
    import tensor_stream
    v = tensor_stream.VideoProcessor(device=0, max_consumers=5)
    yield v
    v.Close()
