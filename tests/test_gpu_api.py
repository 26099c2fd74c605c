"""GPU: behaviour of the C ABI around the kernels: scratch pre-sizing and first-call graph capture of the
two-pass formats, concurrent callers sharing one stream, 16-byte aligned batch frames, device restoration, roctx ranges."""
import threading

import numpy as np
import pytest
import torch

from util import synth_nv12

pytestmark = pytest.mark.gpu

UYVY, YUV444, RGB24 = 4, 5, 1


def _ref(oracle, f, dst, rt, fourcc, norm, planes=1):
    return oracle.convert(f[0], f[1], dst=dst, resize_type=rt, fourcc=fourcc, planes=planes, normalization=norm, nthreads=4)[0]


@pytest.mark.parametrize("fourcc,norm", [(UYVY, False), (YUV444, True)])
def test_first_call_graph_capture_of_two_pass_formats(oracle, fourcc, norm):
    """tsvpp_prepare_batch sizes the resized-NV12 scratch of the capture stream, so the FIRST conversion of a
    UYVY / YUV444 + resize request can already be a captured one (no hipMalloc inside the capture)."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    n = 3
    frames = [synth_nv12(640, 360, seed=900 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=fourcc, normalization=norm)
    s = torch.cuda.Stream()
    v.prepare(fp, 640, 360, n_frames=n, stream=s.cuda_stream)
    out = v._alloc(fp.parameters, 640, 360, n)
    batch = v.make_batch(ys, uvs, fp, out=out)
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):  # no warm-up call: this is the first conversion of the context
        v.run_batch(batch, torch.cuda.current_stream().cuda_stream)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in range(n):
        ref = _ref(oracle, frames[i], (320, 180), 1, fourcc, norm)
        assert np.array_equal(o[i].ravel().view(np.uint8), ref.view(np.uint8))
    v.Close()


@pytest.mark.parametrize("prepared", [True, False])
@pytest.mark.parametrize("src,dst", [((1080, 608), (480, 360)), ((1920, 1080), (224, 224))])
def test_first_call_graph_capture_of_float_weight_area(oracle, src, dst, prepared):
    """a non-dyadic AREA down-scale uses a host-built divisor table (one entry per column pattern x row
    pattern).  prepared: tsvpp_prepare_batch builds it, so the FIRST conversion may already be a captured one.  Not prepared (the
    two weight tables exist -- built by two other requests that share one scale each -- but this request's divisor table does
    not): nothing is allocated inside the capture and the conversion still succeeds, on the kernel that sums the weights itself."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    n = 2
    frames = [synth_nv12(src[0], src[1], seed=950 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=3, pixel_format=RGB24, normalization=True, planes_pos=0)
    s = torch.cuda.Stream()
    if prepared:
        v.prepare(fp, src[0], src[1], n_frames=n, stream=s.cuda_stream)
    else:
        # the weight rows of both scales exist (transposed request: x and y scales swapped -> another divisor-table key),
        # the divisor table of THIS request does not
        fpt = ts.FrameParameters(width=dst[1] * src[0] // src[1] // 2 * 2, height=dst[1], resize_type=3, pixel_format=RGB24, normalization=True, planes_pos=0)
        v.prepare(fpt, src[0], src[1], n_frames=n, stream=s.cuda_stream)
        v.prepare(ts.FrameParameters(width=dst[0], height=dst[0] * src[1] // src[0] // 2 * 2, resize_type=3, pixel_format=RGB24, normalization=True, planes_pos=0),
                  src[0], src[1], n_frames=n, stream=s.cuda_stream)
    out = v._alloc(fp.parameters, src[0], src[1], n)
    batch = v.make_batch(ys, uvs, fp, out=out)
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):  # no warm-up call
        v.run_batch(batch, torch.cuda.current_stream().cuda_stream)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in range(n):
        ref = oracle.convert(frames[i][0], frames[i][1], dst=dst, resize_type=3, fourcc=RGB24, planes=0, normalization=True, nthreads=4)[0]
        assert np.array_equal(o[i].ravel().view(np.uint8), ref.view(np.uint8))
    v.Close()


def test_threads_sharing_the_null_stream_do_not_corrupt_the_scratch(oracle):
    """Two-pass formats keep their NV12 intermediate in a per-stream scratch: callers that share a stream (here the
    null stream) must neither interleave their passes on it nor free it under each other while it grows."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0)
    lib, ctx = v._lib, v._ctx
    from tensor_stream import _native as N
    import ctypes
    errors = []

    def worker(k):
        try:
            w, h = (640, 360) if k % 2 == 0 else (960, 540)
            f = synth_nv12(w, h, seed=1000 + k)
            y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
            for it in range(12):
                dst = (160 + 32 * ((it + k) % 5), 90 + 18 * ((it + k) % 5))  # growing and shrinking needs
                fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=3, pixel_format=UYVY)
                out = v._alloc(fp.parameters, w, h)
                fr = N.NV12(y.data_ptr(), uv.data_ptr(), y.stride(0), uv.stride(0), w, h)
                N.check(lib.tsvpp_convert(ctx, ctypes.byref(fr), ctypes.byref(fp.parameters), out.data_ptr(), None))  # NULL stream
                torch.cuda.synchronize()
                ref = _ref(oracle, f, dst, 3, UYVY, False)
                if not np.array_equal(out.cpu().numpy().ravel(), ref):
                    errors.append((k, it, dst))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts_ = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts_:
        t.start()
    for t in ts_:
        t.join()
    v.Close()
    assert not errors, errors[:5]


def test_batch_frames_are_16_byte_aligned_and_stay_on_the_vector_kernels(vpp, oracle):
    """uint8 frames of 250 x 250 x 3 bytes are not a multiple of 16: the batch tensor pads its frame stride, so every
    frame of the batch keeps the vector-store kernels (a silent fall-back to the element-wise kernel)."""
    import tensor_stream as ts
    n = 5
    frames = [synth_nv12(500, 500, seed=1100 + i) for i in range(n)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=250, height=250, resize_type=1, pixel_format=RGB24, planes_pos=1)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    assert out.shape == (n, 250, 250, 3)
    assert all(out[i].data_ptr() % 16 == 0 and out[i].is_contiguous() for i in range(n))
    for i in range(n):
        ref = _ref(oracle, frames[i], (250, 250), 1, RGB24, False)
        assert np.array_equal(out[i].cpu().numpy().ravel(), ref)
    # one misaligned frame only moves ITS launch group to the element-wise kernel; results are the same either way
    flat = torch.empty(n * 250 * 250 * 3 + 1, dtype=torch.uint8, device="cuda")
    odd = flat[1:].view(n, 250, 250, 3)
    vpp.convert_batch(ys, uvs, fp, out=odd)
    torch.cuda.synchronize()
    assert torch.equal(odd, out)


def test_calls_leave_the_current_device_alone(oracle):
    """Every entry point restores the calling thread's device (hipSetDevice leaked into torch)."""
    import tensor_stream as ts
    n_dev = torch.cuda.device_count()
    target = n_dev - 1  # another device than the current one when the box has several
    torch.cuda.set_device(0)
    v = ts.VideoProcessor(device=target)
    assert torch.cuda.current_device() == 0
    f = synth_nv12(320, 240, seed=7)
    with torch.cuda.device(target):
        y, uv = torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()
    fp = ts.FrameParameters(width=160, height=120, resize_type=3, pixel_format=RGB24)
    v.prepare(fp, 320, 240)
    with torch.cuda.device(target):
        out = torch.empty((120, 160, 3), dtype=torch.uint8, device=f"cuda:{target}")
        stream = torch.cuda.current_stream(target).cuda_stream
    from tensor_stream import _native as N
    import ctypes
    fr = N.NV12(y.data_ptr(), uv.data_ptr(), y.stride(0), uv.stride(0), 320, 240)
    N.check(v._lib.tsvpp_convert(v._ctx, ctypes.byref(fr), ctypes.byref(fp.parameters), out.data_ptr(), stream))
    assert torch.cuda.current_device() == 0
    torch.cuda.synchronize(target)
    assert np.array_equal(out.cpu().numpy().ravel(), _ref(oracle, f, (160, 120), 3, RGB24, False))
    v.Close()
    assert torch.cuda.current_device() == 0


def test_roctx_ranges_can_be_switched_on(vpp, oracle):
    """enable_nvtx() of the reference -> roctx ranges around every conversion; results unchanged."""
    import tensor_stream as ts
    f = synth_nv12(320, 240, seed=8)
    fp = ts.FrameParameters(width=160, height=120, resize_type=2, pixel_format=RGB24)
    try:
        vpp.enable_markers(True)
    except RuntimeError as e:
        pytest.skip(f"no roctx library: {e}")
    try:
        out = vpp.Convert(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda(), fp)
        torch.cuda.synchronize()
    finally:
        vpp.enable_markers(False)
    assert np.array_equal(out.cpu().numpy().ravel(), _ref(oracle, f, (160, 120), 2, RGB24, False))


def test_trim_releases_retired_memory_and_the_context_keeps_working(oracle):
    """tsvpp_trim: nothing is freed under running work -- geometry-table sets pushed out of the cache and outgrown scratch buffers are retired;
    trim releases them at a quiescent point.  Here: outgrow the two-pass scratch twice, trim, convert again (same bits)."""
    import tensor_stream as ts
    v = ts.VideoProcessor(device=0, max_consumers=1)
    try:
        fp = ts.FrameParameters(width=300, height=200, resize_type=1, pixel_format=4, planes_pos=1, normalization=False)  # UYVY behind a resize: two passes
        y, uv = synth_nv12(640, 360, seed=12)
        ty, tuv = torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda()
        ref, _, _ = oracle.convert(y, uv, dst=(300, 200), resize_type=1, fourcc=4, planes=1, normalization=False, nthreads=4, width=640)
        for n in (1, 3, 9):  # the scratch is outgrown twice: two retired buffers
            out = v.convert_batch([ty] * n, [tuv] * n, fp, width=640)
        torch.cuda.synchronize()
        assert np.array_equal(out[8].cpu().numpy().ravel(), ref)
        assert v.trim() == 0          # (bytes of TABLE sets released: none were retired here; the scratch buffers are not counted)
        out = v.convert_batch([ty] * 9, [tuv] * 9, fp, width=640)
        torch.cuda.synchronize()
        assert np.array_equal(out[0].cpu().numpy().ravel(), ref) and np.array_equal(out[8].cpu().numpy().ravel(), ref)
        assert v.trim() == 0
    finally:
        v.Close()
