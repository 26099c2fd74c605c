"""CPU: the kernel selection of ~500 requests against the committed snapshot (tests/golden/describe_snapshot.json, regenerate with
tests/golden/make_describe_snapshot.py after an INTENDED change): launch_fused is several hundred lines of measured thresholds, and a change to one
of them must show up in review as a diff of that file, not as a silent re-routing of some other geometry."""
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _maker():
    spec = importlib.util.spec_from_file_location("make_describe_snapshot", os.path.join(HERE, "golden", "make_describe_snapshot.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(any(k.startswith("TSVPP_") and k not in ("TSVPP_DEBUG_KNOBS", "TSVPP_BENCH_STUB") for k in os.environ), reason="knob runs select other kernels on purpose")
def test_selection_matches_the_committed_snapshot():
    m = _maker()
    want = json.load(open(os.path.join(HERE, "golden", "describe_snapshot.json")))
    got = m.snapshot()
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:10]
    diff = {k: (want[k], got[k]) for k in want if want[k] != got[k]}
    assert not diff, f"{len(diff)} requests are routed differently, e.g. {list(diff.items())[:3]} -- intended? then regenerate the snapshot"
    assert len(want) >= 200
    kernels = {v.get("kernel", "").split("<")[0] for v in want.values()}
    assert len(kernels) >= 12  # the snapshot spans the kernel families, not one corner of the dispatcher
