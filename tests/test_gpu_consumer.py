"""GPU: the one-frame-per-call regime (round 6) -- the replay of a thread's recent launches, TSVPP_OPT_INPUTS_READY (barrier-free launches, two streams per
consumer) and the consumer pool's stream hand-out.  The reference's calling pattern: ONE frame per VideoProcessor::Convert on the consumer's stream
(reference src/Wrappers/WrapperPython.cpp:265-363, src/VideoProcessor.cpp:98-104)."""
import ctypes

import numpy as np
import pytest
import torch

from util import synth_nv12

pytestmark = pytest.mark.gpu

REQUESTS = [  # (src, dst, resize, fourcc, planes, norm): one per kernel family a single frame can take
    ((1920, 1080), (1280, 720), 1, 2, 0, True),    # headline: 2x2-tap LDS kernel
    ((1920, 1080), (1280, 720), 2, 1, 1, False),   # streaming BICUBIC 3:2, uint8 merged
    ((1920, 1080), (0, 0), 0, 2, 0, True),         # colour only
    ((1280, 720), (640, 360), 3, 1, 0, False),     # AREA 2:1
    ((1080, 608), (480, 360), 2, 1, 1, False),     # BICUBIC, non-dyadic: column kernel with its cached tables
    ((1080, 608), (480, 360), 3, 1, 1, True),      # AREA, float weights: divisor table
    ((640, 360), (854, 480), 1, 1, 1, False),      # 4 k + 2 columns: shifted tile column
    ((640, 360), (320, 180), 1, 4, 1, False),      # UYVY behind a resize (two passes or the streaming kernel)
]


def _ref(oracle, f, req):
    _src, dst, rt, fourcc, planes, norm = req
    return oracle.convert(f[0], f[1], dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=8)[0]


def _fp(ts, req):
    _src, dst, rt, fourcc, planes, norm = req
    return ts.FrameParameters(width=dst[0], height=dst[1], resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)


@pytest.mark.parametrize("req", REQUESTS, ids=lambda r: f"{r[0][0]}x{r[0][1]}-{r[1][0]}x{r[1][1]}-rt{r[2]}-f{r[3]}")
def test_repeated_single_frame_calls_replay_bit_exact(oracle, req):
    """The same request again and again with DIFFERENT frames and output buffers (what a consumer does): calls 2.. take the replay path of convert_impl
    and must give what the first (full selection) call gives -- the oracle's bytes."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    src = req[0]
    fp = _fp(ts, req)
    outs = []
    frames = [synth_nv12(src[0], src[1], seed=7100 + i) for i in range(4)]
    for f in frames:
        y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
        outs.append(v.Convert(y, uv, fp))
    torch.cuda.synchronize()
    for f, o in zip(frames, outs):
        assert np.array_equal(o.cpu().numpy().ravel().view(np.uint8), _ref(oracle, f, req).view(np.uint8))
    v.Close()


def test_replay_follows_the_alignment_class_and_survives_many_requests(oracle):
    """A replayed launch was selected for one alignment class (16-byte aligned output -> vector stores; dword-aligned planes): a call whose pointers fall in
    another class must not replay it.  And more distinct requests than the cache holds (8) still convert correctly, in any interleaving."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    req = ((640, 360), (320, 180), 1, 1, 1, False)
    fp = _fp(ts, req)
    f = synth_nv12(640, 360, seed=7200)
    y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
    ref = _ref(oracle, f, req)
    nbytes = ref.size
    a = v.Convert(y, uv, fp)  # aligned: full selection, remembered
    raw = torch.empty(nbytes + 64, dtype=torch.uint8, device="cuda")
    b = raw[3:3 + nbytes].view(180, 320, 3)  # misaligned output: element-wise kernel
    v.Convert(y, uv, fp, out=b)
    c = v.Convert(y, uv, fp)  # aligned again: replay of the first
    # an input plane that is not dword-aligned
    rawy = torch.empty(y.numel() + 64, dtype=torch.uint8, device="cuda")
    y1 = rawy[1:1 + y.numel()].view(y.shape)
    y1.copy_(y)
    d = v.Convert(y1, uv, fp)
    torch.cuda.synchronize()
    for o in (a, b, c, d):
        assert np.array_equal(o.contiguous().cpu().numpy().ravel(), ref)
    # 12 distinct requests, three rounds
    reqs = [((640, 360), (320 + 16 * k, 180 + 8 * k), 1 + (k % 3), 1 + (k & 1), k & 1, bool(k & 2)) for k in range(12)]
    refs = [_ref(oracle, f, r) for r in reqs]
    for _round in range(3):
        outs = [v.Convert(y, uv, _fp(ts, r)) for r in reqs]
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            assert np.array_equal(o.cpu().numpy().ravel().view(np.uint8), r.view(np.uint8))
    v.Close()


def test_replay_is_invalidated_by_coefficients_and_options(oracle):
    """tsvpp_set_coeffs / tsvpp_set_option move the context's epoch: a launch finished under the old state is not replayed."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    req = ((640, 360), (0, 0), 0, 1, 1, False)
    fp = _fp(ts, req)
    f = synth_nv12(640, 360, seed=7300)
    y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
    ref = _ref(oracle, f, req)
    a = v.Convert(y, uv, fp).cpu().numpy().ravel()
    k = v.get_coeffs()
    k2 = list(k)
    k2[0] = 1.0  # y_scale
    from tensor_stream import vpp as V
    with pytest.raises(RuntimeError):
        v.set_coeffs(k2)  # a block that differs from the reference's literals needs TSVPP_OPT_UNSAFE_COEFFS
    v.set_option(V.OPT_UNSAFE_COEFFS, 1)
    v.set_coeffs(k2)
    b = v.Convert(y, uv, fp).cpu().numpy().ravel()
    v.set_coeffs(k)
    c = v.Convert(y, uv, fp).cpu().numpy().ravel()
    assert np.array_equal(a, ref) and np.array_equal(c, ref) and not np.array_equal(b, ref)
    v.Close()


def test_consumer_streams_alternate_only_under_inputs_ready_and_only_for_small_launches():
    import tensor_stream as ts
    from tensor_stream import vpp as V
    v = ts.VideoProcessor(device=0, max_consumers=2)
    s0 = v.consumer_stream("a")
    assert [v.consumer_next_stream("a") for _ in range(4)] == [s0] * 4  # default: the consumer's one stream
    assert v.get_option(V.OPT_INPUTS_READY) == 0
    v.set_option(V.OPT_INPUTS_READY, 1)
    seq = [v.consumer_next_stream("a") for _ in range(6)]
    assert seq[0] == s0 and seq[1] != s0 and seq[0::2] == [s0] * 3 and len(set(seq[1::2])) == 1
    assert [v.consumer_next_stream("a", launch_bytes=1 << 30) for _ in range(3)] == [s0] * 3  # large launches stay on the first stream
    sb = v.consumer_stream("b")
    assert sb not in seq and v.consumer_next_stream("b") == sb
    with pytest.raises(RuntimeError):
        v.consumer_next_stream("c")  # pool of two exhausted (reference src/VideoProcessor.cpp:100-103)
    with pytest.raises(RuntimeError):
        v.consumer_synchronize("nobody")
    v.consumer_synchronize("a")
    v.set_option(V.OPT_INPUTS_READY, 0)
    assert [v.consumer_next_stream("a") for _ in range(3)] == [s0] * 3
    with pytest.raises(RuntimeError):
        v.set_option(77, 1)
    v.Close()


@pytest.mark.parametrize("value", [1, 2, 3])
def test_inputs_ready_conversions_are_bit_exact(oracle, value):
    """A ring of distinct frames and distinct outputs converted back to back through the consumer pool under TSVPP_OPT_INPUTS_READY (launches overlap, no barrier
    bit): every output equals the oracle's; tsvpp_consumer_synchronize covers both streams."""
    import tensor_stream as ts
    from tensor_stream import _native as N
    from tensor_stream import vpp as V
    v = ts.VideoProcessor(device=0, max_consumers=1)
    v.set_option(V.OPT_INPUTS_READY, value)
    req = ((1920, 1080), (1280, 720), 1, 2, 0, True)
    fp = _fp(ts, req)
    ring = 6
    frames = [synth_nv12(1920, 1080, seed=7400 + i) for i in range(ring)]
    ys = [torch.from_numpy(f[0]).cuda() for f in frames]
    uvs = [torch.from_numpy(f[1]).cuda() for f in frames]
    outs = [v._alloc(fp.parameters, 1920, 1080) for _ in range(ring)]
    torch.cuda.synchronize()  # the option's promise: inputs complete before the calls
    lib = N.lib()
    for rnd in range(20):
        for i in range(ring):
            fr = v._frame(ys[i], uvs[i], None, None)
            s = v.consumer_next_stream("ring")
            N.check(lib.tsvpp_convert(v._ctx, ctypes.byref(fr), ctypes.byref(fp.parameters), outs[i].data_ptr(), s))
        v.consumer_synchronize("ring")  # (an output is rewritten in the next round: wait first -- promise (2))
    for f, o in zip(frames, outs):
        assert np.array_equal(o.cpu().numpy().ravel().view(np.uint8), _ref(oracle, f, req).view(np.uint8))
    v.Close()


def test_later_work_on_the_stream_still_waits_for_a_barrier_free_launch(oracle):
    """What stays ordered under the option: work enqueued LATER on the stream.  A device-to-device copy of the output enqueued right behind each conversion (same
    stream) must see the finished frame."""
    import tensor_stream as ts
    from tensor_stream import _native as N
    from tensor_stream import vpp as V
    v = ts.VideoProcessor(device=0, max_consumers=1)
    v.set_option(V.OPT_INPUTS_READY, 1)
    req = ((1920, 1080), (1280, 720), 1, 2, 0, True)
    fp = _fp(ts, req)
    f = synth_nv12(1920, 1080, seed=7500)
    y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
    ref = _ref(oracle, f, req)
    torch.cuda.synchronize()
    lib = N.lib()
    fr = v._frame(y, uv, None, None)
    for _ in range(50):
        out = v._alloc(fp.parameters, 1920, 1080)
        out.fill_(-1.0)
        copy = torch.empty_like(out)
        torch.cuda.synchronize()
        s = v.consumer_next_stream("c")
        ext = torch.cuda.ExternalStream(s, device=0)
        N.check(lib.tsvpp_convert(v._ctx, ctypes.byref(fr), ctypes.byref(fp.parameters), out.data_ptr(), s))
        with torch.cuda.stream(ext):
            copy.copy_(out, non_blocking=True)
        ext.synchronize()
        assert np.array_equal(copy.cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8))
    v.Close()
