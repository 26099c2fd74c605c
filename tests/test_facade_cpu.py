"""CPU: frame sources and the decoder-ring hand-off of the TensorStreamConverter facade (no GPU needed)."""
import threading
import time

import numpy as np
import pytest


def test_synthetic_source_is_deterministic_and_bounded():
    from tensor_stream.sources import open_source
    a = open_source("synthetic://640x360?seed=7&frames=3&fps=30&pool=2")
    b = open_source("synthetic://640x360?seed=7&frames=3&fps=30&pool=2")
    assert (a.width, a.height, a.fps_num) == (640, 360, 30)
    fa = [a.next_frame() for _ in range(4)]
    fb = [b.next_frame() for _ in range(4)]
    assert fa[3] is None and fb[3] is None
    for x, y in zip(fa[:3], fb[:3]):
        assert x[0].shape == (360, 640) and x[1].shape == (180, 640) and x[0].dtype == np.uint8
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    assert np.array_equal(fa[0][0], fa[2][0]) and not np.array_equal(fa[0][0], fa[1][0])  # pool of 2 cycles


def test_raw_nv12_source(tmp_path):
    from tensor_stream.sources import open_source
    w, h = 64, 32
    data = np.arange(w * h * 3 // 2 * 2, dtype=np.uint32).astype(np.uint8)
    p = tmp_path / "clip.nv12"
    p.write_bytes(data.tobytes())
    s = open_source(f"{p}?w={w}&h={h}&fps=24")
    y0, uv0 = s.next_frame()
    y1, _ = s.next_frame()
    assert s.next_frame() is None
    assert np.array_equal(y0.ravel(), data[: w * h]) and np.array_equal(uv0.ravel(), data[w * h: w * h * 3 // 2])
    assert np.array_equal(y1.ravel(), data[w * h * 3 // 2: w * h * 3 // 2 + w * h])


@pytest.mark.parametrize("url", ["wrong.h264", "rtmp://host/app/stream", "synthetic://640x361", "missing.nv12?w=64&h=32"])
def test_unopenable_sources_raise_runtimeerror(url):
    from tensor_stream.sources import open_source
    with pytest.raises(RuntimeError):
        open_source(url)


def test_ring_each_consumer_sees_each_frame_once_and_delay():
    """Semantics of Decoder::GetFrame (reference src/Decoder.cpp:97-131)."""
    from tensor_stream import FrameRing
    ring = FrameRing(4)
    got = []

    def consumer():
        try:
            while True:
                got.append(ring.get("c")[1])
        except RuntimeError as e:
            got.append(str(e))

    t = threading.Thread(target=consumer)
    t.start()
    for k in range(5):
        ring.publish(f"frame{k}")
        time.sleep(0.02)
    # a late joiner takes the latest frame immediately; delay -1 gives the one before; too-old -> REPEAT (None)
    assert ring.get("late") == ("frame4", 5)
    ring.publish("frame5")
    assert ring.get("late", -1) == ("frame4", 6)
    ring.publish("frame6")
    assert ring.get("late", 3) == ("frame6", 7)        # positive delay is forced to 0
    ring.publish("frame7")
    frame, idx = ring.get("late", -3)                  # slot (8-1)%4 - 3 = 0 -> frame4 was overwritten by frame 8 % 4 ... holds frame4? no: ring slot 0
    assert frame in ("frame4",) and idx == 8
    ring.finish()
    t.join(timeout=5)
    assert got[-1] == "Decoding finished"
    nums = [g for g in got if isinstance(g, int)]
    assert nums == sorted(set(nums)) and len(nums) >= 5    # never the same frame twice
    with pytest.raises(RuntimeError, match="Decoding finished"):
        ring.get("late")


def test_facade_constructor_and_errors_without_gpu():
    import tensor_stream as ts
    r = ts.TensorStreamConverter("synthetic://640x360", max_consumers=3, cuda_device=0, buffer_size=10)
    assert (r.max_consumers, r.cuda_device, r.buffer_size, r.stream_url) == (3, 0, 10, "synthetic://640x360")
    r.enable_logs(ts.LogsLevel.LOW, ts.LogsType.CONSOLE)
    r.enable_nvtx()
    r.set_timeout(1.5)
    r.skip_analyze()
    r.stop()                                   # stop without init: no crash (reference test_stop_without_init)
    with pytest.raises(RuntimeError):
        r.read()                               # read without init/start (reference test_read_without_init_start)
    bad = ts.TensorStreamConverter("wrong.h264")
    with pytest.raises(RuntimeError, match="Can't initialize TensorStream"):
        bad.initialize(repeat_number=5)


def test_ring_get_many_hands_every_consumer_the_same_new_frame_once():
    from tensor_stream.tensor_stream import FrameRing
    ring = FrameRing(3)
    names = [f"c{i}" for i in range(8)]
    got = []

    def consumer():
        try:
            while True:
                frames, idx = ring.get_many(names, 0, timeout=5)
                got.append((idx, frames))
        except RuntimeError as e:
            got.append(str(e))

    t = threading.Thread(target=consumer)
    t.start()
    for k in range(4):
        ring.publish(("frame", k))
        while not ring.all_consumed():
            time.sleep(0.001)
    ring.finish()
    t.join(timeout=5)
    assert got[-1] == "Decoding finished"
    assert [g[0] for g in got[:-1]] == [1, 2, 3, 4]
    for idx, frames in got[:-1]:
        assert len(frames) == 8 and all(f == ("frame", idx - 1) for f in frames)
    # a single-name get() after get_many() of the same frame sees nothing new: one hand-off per consumer and frame
    ring2 = FrameRing(2)
    ring2.publish("a")
    ring2.get_many(["x", "y"])
    with pytest.raises(RuntimeError, match="Timeout"):
        ring2.get("x", 0, timeout=0.05)


def test_coalescer_groups_concurrent_requests_and_returns_each_its_slice():
    from tensor_stream.tensor_stream import _Coalescer
    calls = []

    def convert_group(frames, fp):
        calls.append(len(frames))
        time.sleep(0.002)
        return [("out", f, fp) for f in frames]

    co = _Coalescer(convert_group, window=0.05, max_batch=16)
    res, errs = {}, []

    def work(i):
        try:
            res[i] = co.convert(("k", i % 2), ("frame", i), f"c{i}", "fp%d" % (i % 2))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(32)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=10)
    assert not errs and len(res) == 32
    for i in range(32):
        assert res[i] == ("out", ("frame", i), "fp%d" % (i % 2))   # every caller got ITS frame converted with ITS parameters
    assert sum(calls) == 32 and len(calls) <= 6 and max(calls) <= 16  # two keys x 16 = full groups wake the leader early
    assert co.requests == 32 and co.launches == len(calls)
    # a full group is closed by the follower that filled it: no group ever exceeds max_batch, whatever the scheduling
    sizes = []
    co3 = _Coalescer(lambda frames, fp: (sizes.append(len(frames)), time.sleep(0.001), list(frames))[2], window=0.2, max_batch=4)
    got = {}
    th = [threading.Thread(target=lambda i=i: got.__setitem__(i, co3.convert("k", i, f"c{i}", None))) for i in range(203)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=20)
    assert got == {i: i for i in range(203)} and sum(sizes) == 203 and max(sizes) <= 4 and co3.launches == len(sizes)
    # max_batch = 1: the leader fills its own group and must not wait the window out, nor may a follower join it
    sizes1 = []
    co4 = _Coalescer(lambda frames, fp: (sizes1.append(len(frames)), list(frames))[1], window=5.0, max_batch=1)
    t0 = time.monotonic()
    th = [threading.Thread(target=lambda i=i: got.__setitem__(1000 + i, co4.convert("k", i, f"c{i}", None))) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=20)
    assert time.monotonic() - t0 < 2.0 and sizes1 == [1] * 8 and all(got[1000 + i] == i for i in range(8))
    # the production coalescer's groups return a _Converted (batch tensor + event): convert() indexes it like a list
    from tensor_stream.tensor_stream import _Converted
    co5 = _Coalescer(lambda frames, fp: _Converted([f * 10 for f in frames], None), window=0.01)
    assert co5.convert("k", 7, "c", None) == 70
    # an error in the batched conversion reaches every member of the group
    co2 = _Coalescer(lambda frames, fp: (_ for _ in ()).throw(RuntimeError("boom")), window=0.02)
    out = []

    def bad():
        try:
            co2.convert("k", 1, "c", None)
        except RuntimeError as e:
            out.append(str(e))

    th = [threading.Thread(target=bad) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=5)
    assert out == ["boom"] * 4


def test_ring_get_batch_returns_the_temporal_window_oldest_first():
    """read(batch=k): the consumer's next frame with the k - 1 frames before it (absolute frame numbers: frame f lives in slot f mod depth)."""
    from tensor_stream.tensor_stream import FrameRing
    ring = FrameRing(5)
    ring.publish("f0")
    assert ring.get_batch("c", 1) == (["f0"], 1)
    ring.publish("f1")
    assert ring.get_batch("c", 3) == (None, -1)               # only two frames exist: VREADER_REPEAT ...
    ring.publish("f2")
    assert ring.get_batch("c", 3) == (["f0", "f1", "f2"], 3)  # ... until the ring holds three
    for k in range(3, 9):
        ring.publish(f"f{k}")
    assert ring.get_batch("c", 5) == (["f4", "f5", "f6", "f7", "f8"], 9)   # wrapped around the ring, still oldest first
    ring.publish("f9")
    assert ring.get_batch("c", 2, -2) == (["f6", "f7"], 10)    # delay moves the window's newest frame back
    ring.publish("f10")
    assert ring.get_batch("c", 4, -1) == (["f6", "f7", "f8", "f9"], 11)   # 4 + 1 = 5 frames back: just inside the ring
    with pytest.raises(ValueError):
        ring.get_batch("c", 6)                                 # more than the ring holds
    with pytest.raises(ValueError):
        ring.get_batch("c", 4, -2)                             # window reaches past the ring
    with pytest.raises(RuntimeError, match="Timeout"):
        ring.get_batch("c", 1, 0, timeout=0.05)                # every published frame has been handed to "c"
    ring.finish()
    with pytest.raises(RuntimeError, match="Decoding finished"):
        ring.get_batch("c", 1)
