"""GPU: the reference's 38 CRC-32 goldens replayed through the C++ class `VideoProcessor` in the shape of the reference's own VPP
test harness (reference tests/src/VPPTests.cpp:101-132: Init, Convert(input, converted, FrameParameters, "visualize"), CRC of
converted->opaque, DumpFrame, CRC of the dumped file) by tensor-stream_amd/cpp/vpp_goldens.cpp -- the boundary proof in C++, next
to the ctypes replay of tests/test_reference_crcs.py."""
import os
import subprocess

import pytest

from golden.reference_crcs import GOLDENS, INPUT_PLANE_CRCS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tensor-stream_amd", "lib", "vpp_goldens")
FRAME = os.path.join(ROOT, "tests", "golden", "bbb_1080x608_frame0.nv12")


def test_cpp_class_reproduces_the_38_reference_crcs(tmp_path):
    assert os.path.exists(EXE), "vpp_goldens not built (python -c 'import __graft_entry__ as g; g.build()')"
    assert os.path.exists(FRAME)
    table = tmp_path / "goldens.txt"
    with open(table, "w") as f:
        for _, fcc, planes, dst, rt, crop, crcs in GOLDENS:
            f.write(" ".join(str(v) for v in (fcc, planes, dst[0], dst[1], rt, *crop, *crcs)) + "\n")
    r = subprocess.run([EXE, FRAME, "1080", "608", str(table)], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    lines = r.stdout.splitlines()
    assert lines and lines[0] == f"input Y {INPUT_PLANE_CRCS['Y']} UV {INPUT_PLANE_CRCS['UV']}", lines[:1]   # reference tests/src/DecoderTests.cpp:63-65
    results = [l.split() for l in lines[1:] if l and l[0].isdigit()]
    assert len(results) == len(GOLDENS) == 38, r.stdout[-2000:] + r.stderr[-2000:]
    bad = [(GOLDENS[int(t[0])][0], t) for t in results if t[1] != "ok"]
    assert not bad and r.returncode == 0, bad
    for t, g in zip(results, GOLDENS):
        assert int(t[2]) in g[6] and t[2] == t[3]     # CRC of the device result == a reference literal == CRC of the DumpFrame file
