"""GPU: TensorStreamConverter end to end on a synthetic source, following the reference's Python tests
(reference tests/python_tests/CommonTests.py) and checking the frames against the oracle."""
import os
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
URL = "synthetic://1920x1080?seed=5&frames=40&fps=200&pool=4"


def make(url=URL, **kw):
    import tensor_stream as ts
    r = ts.TensorStreamConverter(url, **kw)
    r.initialize()
    return r


def test_initialize_sets_size_and_fps():
    r = make()
    assert r.frame_size == (1920, 1080) and r.fps == 200
    r.stop()


def test_start_read_close_shapes_and_dtypes():
    import tensor_stream as ts
    r = make()
    r.start()
    t = r.read()
    assert t.shape == (1080, 1920, 3) and t.dtype == torch.uint8 and t.is_cuda
    t, idx = r.read(return_index=True)
    assert 0 < idx <= 40
    t = r.read(normalization=True, planes_pos=ts.Planes.PLANAR, pixel_format=ts.FourCC.BGR24, width=1280, height=720,
               resize_type=ts.ResizeType.BILINEAR)
    assert t.shape == (3, 720, 1280) and t.dtype == torch.float32
    t = r.read(pixel_format=ts.FourCC.NV12)
    assert t.shape == (1, 1620, 1920)
    r.stop()
    with pytest.raises(RuntimeError):
        r.read()                                  # reference test_read_after_stop


def test_frames_match_oracle_and_indices_increase(oracle):
    import tensor_stream as ts
    from tensor_stream.sources import open_source
    r = make("synthetic://640x360?seed=11&frames=12&fps=1000&pool=3", framerate_mode=ts.FrameRate.BLOCKING)
    src = open_source("synthetic://640x360?seed=11&frames=12&fps=1000&pool=3")
    pool = [src.next_frame() for _ in range(3)]
    r.start()
    seen = []
    try:
        while True:
            t, idx = r.read(width=320, height=180, resize_type=ts.ResizeType.AREA, normalization=True, return_index=True)
            torch.cuda.synchronize()
            y, uv = pool[(idx - 1) % 3]
            ref, _, _ = oracle.convert(y, uv, dst=(320, 180), resize_type=3, fourcc=1, planes=1, normalization=True)
            assert np.array_equal(t.cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32))
            seen.append(idx)
    except RuntimeError as e:
        assert "Decoding finished" in str(e)
    r.stop()
    assert seen == list(range(1, 13))             # BLOCKING mode: frame by frame, none skipped


def test_many_consumers_in_threads():
    import tensor_stream as ts
    r = make("synthetic://640x360?seed=2&frames=0&fps=500", max_consumers=4)
    r.start()
    out, errs = {}, []

    def work(name):
        try:
            idx = [r.read(name=name, width=256, height=144, return_index=True)[1] for _ in range(10)]
            out[name] = idx
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(f"c{i}",)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not errs and len(out) == 4
    for idx in out.values():
        assert idx == sorted(idx) and len(set(idx)) == len(idx)   # each consumer: strictly increasing frame numbers
    with pytest.raises(RuntimeError, match="-3"):
        r.read(name="one_too_many")                              # 5th consumer on a pool of 4
    r.stop()


def test_dump_sizes(tmp_path):
    r = make()
    r.start()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for _ in range(3):
            r.dump(r.read())
        r.dump(r.read(), name="named")
        assert os.stat("default.yuv").st_size == 1920 * 1080 * 3 * 3 and os.path.isfile("named.yuv")
    finally:
        os.chdir(cwd)
        r.stop()


def test_multiple_init_stop_cycles():
    r = make()
    for _ in range(5):
        r.stop()
        r.initialize()
    r.start()
    time.sleep(0.05)
    assert r.read().shape == (1080, 1920, 3)
    r.stop()
