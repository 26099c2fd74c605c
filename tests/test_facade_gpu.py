"""GPU: TensorStreamConverter end to end on a synthetic source, following the reference's Python tests
(reference tests/python_tests/CommonTests.py) and checking the frames against the oracle."""
import os
import threading
import time

import numpy as np
import pytest
import torch
from util import knob_run

pytestmark = pytest.mark.gpu
URL = "synthetic://1920x1080?seed=5&frames=4000&fps=200&pool=4"


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
    assert 0 < idx <= 4000
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


C5 = dict(width=640, height=360, resize_type=3, pixel_format=2, planes_pos=0, normalization=True)  # 4K -> 640x360 AREA BGR24 planar fp32


def test_read_many_64_consumers_c5_shape_matches_oracle_and_reports_its_rate(oracle):
    """BASELINE config C5's shape through the production entry: 64 consumers of ONE TensorStreamConverter, served by read_many
    (one hand-off, one batched launch per published frame).  Every tensor equals the oracle; the aggregate rate is printed and
    must be well above what one launch per read() reaches (~0.3 of the roofline)."""
    import tensor_stream as ts
    from tensor_stream.sources import open_source
    url = "synthetic://3840x2160?seed=9&frames=0&fps=100000&pool=3"
    r = make(url, max_consumers=64, framerate_mode=ts.FrameRate.FAST)
    pool = open_source(url).pool
    names = [f"consumer{i}" for i in range(64)]
    r.start()
    try:
        tensors, idx = r.read_many(names, return_index=True, **C5)
        torch.cuda.synchronize()
        assert len(tensors) == 64 and tensors[0].shape == (3, 360, 640) and tensors[0].dtype == torch.float32
        y, uv = pool[(idx - 1) % 3]
        ref, _, _ = oracle.convert(y, uv, dst=(640, 360), resize_type=3, fourcc=2, planes=0, normalization=True, nthreads=8)
        for k in (0, 1, 31, 63):
            assert np.array_equal(tensors[k].cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32))
        assert all(tensors[k].data_ptr() != tensors[0].data_ptr() for k in range(1, 64))  # every consumer owns its tensor
        # The rate is host-side wall clock through the interpreter: in some processes the first ~100 calls run at a quarter of the steady
        # rate (the caching allocator settling on its 177 MB batch tensors), and a full Python garbage collection (~40 ms) lands in one
        # window or another -- 22 runs of the single-window version of this test failed 5 times at 0.22-0.27 with every steady window at
        # 0.93.  So: a long warm-up, five short windows, the MEDIAN window judged.
        for _ in range(150):
            r.read_many(names, **C5)
        torch.cuda.synchronize()
        n, dts = 60, []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(n):
                r.read_many(names, **C5)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
    finally:
        r.stop()
    fracs = [64 * n / dt * 15206400 / 8e12 for dt in dts]
    frac = sorted(fracs)[2]
    print(f"\\nfacade read_many: {64 * n / sorted(dts)[2]:.0f} conversions/s, median window {frac:.3f} of the 8 TB/s roofline (C5 bytes per conversion); windows {[round(f, 3) for f in fracs]}")
    if not knob_run():  # (knob runs dispatch other -- slower -- kernels)
        assert frac > 0.30, fracs  # one launch per read() reaches ~0.30; measured here: 0.90-0.94


def test_64_consumer_threads_coalesce_into_few_launches(oracle):
    """64 consumer THREADS calling read() on one converter with coalesce_window_us: same tensors as the oracle, a fraction of
    the launches (the rendezvous itself is bound by the interpreter lock: the rate is printed, not asserted)."""
    import tensor_stream as ts
    from tensor_stream.sources import open_source
    url = "synthetic://1920x1080?seed=4&frames=0&fps=400&pool=2"
    r = make(url, max_consumers=64, coalesce_window_us=2000)
    pool = open_source(url).pool
    r.start()
    results, errs = {}, []

    def work(name):
        try:
            out = []
            for _ in range(6):
                t, idx = r.read(name=name, width=480, height=270, resize_type=ts.ResizeType.BILINEAR, normalization=True, planes_pos=ts.Planes.PLANAR,
                                pixel_format=ts.FourCC.BGR24, return_index=True)
                out.append((t, idx))
            results[name] = out
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(f"c{i}",)) for i in range(64)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    co = r._coalescer
    launches, requests = co.launches, co.requests
    r.stop()
    assert not errs and len(results) == 64 and requests == 64 * 6
    refs = {}
    for name in ("c0", "c17", "c63"):
        for t, idx in results[name]:
            k = (idx - 1) % 2
            if k not in refs:
                refs[k] = oracle.convert(pool[k][0], pool[k][1], dst=(480, 270), resize_type=1, fourcc=2, planes=0, normalization=True, nthreads=8)[0]
            assert np.array_equal(t.cpu().numpy().ravel().view(np.uint32), refs[k].view(np.uint32))
    print(f"\\nfacade 64 threads: {requests} reads in {launches} launches, {requests / dt:.0f} reads/s")
    assert launches <= requests // 2   # (measured: ~15 launches for the 384 reads; the bound leaves room for a box whose host side crawls)


def test_raw_nv12_file_source_streams_through_pinned_staging(tmp_path, oracle):
    """A source without a fixed frame pool: every frame goes pinned buffer -> copy stream -> device; the consumer waits on the
    frame's event, not on the host."""
    import tensor_stream as ts
    w, h, n = 320, 180, 9
    rng = np.random.default_rng(6)
    data = rng.integers(0, 256, size=(n, w * h * 3 // 2), dtype=np.uint8)
    p = tmp_path / "clip.nv12"
    p.write_bytes(data.tobytes())
    r = make(f"{p}?w={w}&h={h}&fps=2000", framerate_mode=ts.FrameRate.BLOCKING)
    r.start()
    seen = 0
    try:
        while True:
            t, idx = r.read(normalization=True, return_index=True)
            y = data[idx - 1][: w * h].reshape(h, w)
            uv = data[idx - 1][w * h:].reshape(h // 2, w)
            ref = oracle.convert(y, uv, fourcc=1, planes=1, normalization=True)[0]
            assert np.array_equal(t.cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32)), idx
            seen += 1
    except RuntimeError as e:
        assert "Decoding finished" in str(e)
    r.stop()
    assert seen == n


def test_coalesced_reads_on_the_callers_own_streams(oracle):
    """with coalesce_window_us the group is converted on the LEADER's stream; a consumer thread that runs under its own
    torch stream must still see a finished tensor (its stream waits for the group's event) and reads it on that stream right away."""
    import tensor_stream as ts
    from tensor_stream.sources import open_source
    url = "synthetic://3840x2160?seed=9&frames=0&fps=400&pool=2"
    r = make(url, max_consumers=16, coalesce_window_us=3000)
    pool = open_source(url).pool
    r.start()
    results, errs = {}, []

    def work(name):
        try:
            s = torch.cuda.Stream()
            out = []
            with torch.cuda.stream(s):
                for _ in range(5):
                    t, idx = r.read(name=name, width=1920, height=1080, resize_type=ts.ResizeType.BICUBIC, pixel_format=ts.FourCC.BGR24, planes_pos=ts.Planes.MERGED,
                                    return_index=True)
                    out.append((t.clone(), idx))  # consumed on the caller's stream at once, no synchronisation of the caller's own
            s.synchronize()
            results[name] = out
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(f"c{i}",)) for i in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    r.stop()
    assert not errs and len(results) == 16
    refs = {}
    for name, out in results.items():
        for t, idx in out:
            k = (idx - 1) % 2
            if k not in refs:
                refs[k] = oracle.convert(pool[k][0], pool[k][1], dst=(1920, 1080), resize_type=2, fourcc=2, planes=1, normalization=False, nthreads=8)[0]
            assert np.array_equal(t.cpu().numpy().ravel(), refs[k]), (name, idx)


def test_dump_fills_a_missing_dimension_from_the_tensor(tmp_path):
    """reference WrapperPython.cpp:425-445: width / height of 0 are taken from the tensor, one given dimension is not an error."""
    import tensor_stream as ts
    r = make()
    r.start()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        t = r.read(width=640, height=360, resize_type=ts.ResizeType.BILINEAR)
        r.dump(t, name="a", width=640)
        r.dump(t, name="a", height=360)
        r.dump(t, name="a", width=640, height=360)
        assert os.stat("a.yuv").st_size == 3 * 640 * 360 * 3
        p = r.read(width=640, height=360, resize_type=ts.ResizeType.BILINEAR, planes_pos=ts.Planes.PLANAR, normalization=True)
        r.dump(p, name="p", planes_pos=ts.Planes.PLANAR, normalization=True)
        assert os.stat("p.yuv").st_size == 640 * 360 * 3 * 4
        y = r.read(pixel_format=ts.FourCC.NV12)
        r.dump(y, name="y", pixel_format=ts.FourCC.NV12, width=1920)
        assert os.stat("y.yuv").st_size == 1920 * 1080 * 3 // 2
        with pytest.raises(RuntimeError, match="-3"):
            r.dump(t, name="bad", width=320, height=360)
    finally:
        os.chdir(cwd)
        r.stop()


@pytest.mark.parametrize("k", [1, 3, 5])
def test_read_batch_returns_the_last_k_frames_in_one_launch_bit_exact(oracle, k):
    """VERDICT r05 next #1c: read(batch=k) -> (k, ...) out of the decoder ring through ONE tsvpp_convert_batch, oldest frame first; every slice equals the oracle's
    conversion of the frame with that index."""
    import tensor_stream as ts
    from tensor_stream.sources import open_source
    url = "synthetic://640x360?seed=21&frames=14&fps=1000&pool=7"
    r = make(url, framerate_mode=ts.FrameRate.BLOCKING, buffer_size=5)
    src = open_source(url)
    pool = [src.next_frame() for _ in range(7)]
    r.start()
    seen = []
    try:
        while True:
            t, idx = r.read(width=320, height=180, resize_type=ts.ResizeType.BILINEAR, pixel_format=ts.FourCC.BGR24, planes_pos=ts.Planes.PLANAR,
                            normalization=True, return_index=True, batch=k)
            torch.cuda.synchronize()
            assert t.shape == (k, 3, 180, 320) and t.dtype == torch.float32
            for j in range(k):  # slice j = frame number idx - k + j (0-based: idx - 1 is the newest)
                y, uv = pool[(idx - k + j) % 7]
                ref, _, _ = oracle.convert(y, uv, dst=(320, 180), resize_type=1, fourcc=2, planes=0, normalization=True)
                assert np.array_equal(t[j].cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32)), (idx, j)
            seen.append(idx)
    except RuntimeError as e:
        assert "Decoding finished" in str(e)
    r.stop()
    assert seen == list(range(k, 15))  # BLOCKING: one window per published frame, the first once k frames exist


def test_read_batch_larger_than_the_ring_is_an_error():
    r = make("synthetic://640x360?seed=1&frames=0&fps=500", buffer_size=4)
    r.start()
    with pytest.raises(RuntimeError):
        r.read(batch=5)
    with pytest.raises(RuntimeError):
        r.read(batch=0)
    assert r.read(batch=2).shape == (2, 360, 640, 3)
    r.stop()
