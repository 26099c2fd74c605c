"""GPU: persistent device-resident frame tables (tsvpp_table_create / _set / tsvpp_convert_table, include/tsvpp.h) -- launches of more than
TSVPP_MAX_BATCH frames whose pointer triples are read from device memory instead of the kernarg segment.  Every kernel family must produce the same
bits through the table as through tsvpp_convert_batch (and the oracle): the indirection lives in ONE accessor (vpp_kernels.h: PtrCol), the tests
sweep the kernels that use it.  Also: runs that start inside the table, entries replaced after creation, crops (the origin travels with the request,
not with the table), the two-pass formats (which fall back to kernarg launches over the table's host mirror) and the error paths."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_FRAMES = 300  # > 2 x TSVPP_MAX_BATCH: one table launch where convert_batch needs three


def _pool(w, h, pitch, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    ys = torch.randint(0, 256, (n, h, pitch), dtype=torch.uint8, device="cuda", generator=g)
    uvs = torch.randint(0, 256, (n, h // 2, pitch), dtype=torch.uint8, device="cuda", generator=g)
    return ys, uvs


def _check(oracle, ys, uvs, out, f, w, crop, dst, rt, fourcc, planes, norm):
    ref, _, _ = oracle.convert(ys[f].cpu().numpy(), uvs[f].cpu().numpy(), crop=crop, dst=dst, resize_type=rt, fourcc=fourcc, planes=planes,
                               normalization=norm, nthreads=8, width=w)
    got = out[f].cpu().numpy().ravel()
    assert got.size == ref.size and np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (f, dst, rt, fourcc, planes, norm)


CASES = [
    # src, dst, resize, fourcc, planes, norm, crop                       (the kernel family each one lands on)
    ((640, 360), (0, 0), 0, 2, 0, True, (0, 0, 0, 0)),          # colour only
    ((640, 360), (0, 0), 0, 0, 1, False, (0, 0, 0, 0)),         # Y800 copy
    ((1280, 720), (256, 256), 1, 1, 0, True, (0, 0, 0, 0)),     # BILINEAR 5 x 2.8125: row segments (C3's geometry)
    ((960, 540), (640, 360), 1, 2, 0, True, (0, 0, 0, 0)),      # BILINEAR 3 : 2, LDS window tile
    ((960, 540), (640, 360), 1, 1, 1, False, (0, 0, 0, 0)),     # ... uint8: streaming kernel
    ((960, 540), (640, 360), 2, 2, 1, False, (0, 0, 0, 0)),     # BICUBIC 3 : 2 streaming
    ((960, 544), (320, 136), 2, 2, 1, False, (0, 0, 0, 0)),     # BICUBIC 3 x 4: both odd / even mix
    ((960, 540), (400, 300), 2, 1, 0, True, (0, 0, 0, 0)),      # BICUBIC non-dyadic: wave-per-tile kernel with host tables
    ((1280, 720), (214, 120), 3, 2, 0, True, (0, 0, 0, 0)),     # AREA ~6 x 6 float weights
    ((1280, 720), (320, 180), 3, 2, 0, True, (0, 0, 0, 0)),     # AREA 4 x 4: box kernel
    ((640, 360), (1280, 720), 3, 1, 1, True, (0, 0, 0, 0)),     # AREA up-scale
    ((640, 360), (1280, 720), 1, 1, 1, False, (0, 0, 0, 0)),    # BILINEAR 1 : 2 streaming
    ((960, 540), (320, 180), 0, 1, 0, False, (0, 0, 0, 0)),     # NEAREST point kernel
    ((960, 540), (300, 200), 1, 3, 1, False, (101, 51, 901, 451)),  # crop with an odd origin, NV12 output
    ((960, 540), (0, 0), 0, 6, 1, True, (100, 50, 500, 350)),   # crop only, HSV
]


@pytest.mark.parametrize("src,dst,rt,fourcc,planes,norm,crop", CASES)
def test_table_matches_batch_and_oracle(vpp, oracle, src, dst, rt, fourcc, planes, norm, crop):
    import tensor_stream as ts
    w, h = src
    pitch = (w + 63) // 64 * 64
    n = N_FRAMES
    ys, uvs = _pool(w, h, pitch, n, seed=w + dst[0] + rt)
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, planes_pos=planes, normalization=norm)
    want = vpp.convert_batch(ys, uvs, fp, width=w)
    tab = vpp.make_table(ys, uvs, fp, width=w)
    got = vpp.run_table(tab)
    torch.cuda.synchronize()
    assert got.data_ptr() != want.data_ptr()
    for f in range(n):
        assert torch.equal(got[f].view(torch.uint8), want[f].view(torch.uint8)), f
    for f in (0, 127, 128, n - 1):
        _check(oracle, ys, uvs, got, f, w, crop, dst, rt, fourcc, planes, norm)
    vpp.free_table(tab)


def test_runs_inside_the_table_and_replaced_entries(vpp, oracle):
    import tensor_stream as ts
    from tensor_stream import _native as N
    w, h, pitch, n = 640, 360, 640, 200
    ys, uvs = _pool(w, h, pitch, n, seed=9)
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=2, planes_pos=0, normalization=True)
    tab = vpp.make_table(ys, uvs, fp, width=w)
    out = tab["out"]
    out.zero_()
    vpp.run_table(tab, first=37, n=150)  # a run that starts and ends inside the table: one 150-frame launch
    torch.cuda.synchronize()
    assert not out[36].any() and not out[187].any()
    for f in (37, 100, 186):
        _check(oracle, ys, uvs, out, f, w, (0, 0, 0, 0), (320, 180), 1, 2, 0, True)
    # replace entries 10 .. 12 by frames 150 .. 152 writing into outputs 0 .. 2, on the stream the conversion then runs on
    stream = torch.cuda.current_stream().cuda_stream
    fr = (N.NV12 * 3)(*[vpp._frame(ys[150 + i], uvs[150 + i], w, None) for i in range(3)])
    outs = (ctypes.c_void_p * 3)(*[out[i].data_ptr() for i in range(3)])
    N.check(vpp._lib.tsvpp_table_set(tab["handle"], 10, 3, fr, outs, stream))
    vpp.run_table(tab, first=10, n=3)
    torch.cuda.synchronize()
    for i in range(3):
        ref, _, _ = oracle.convert(ys[150 + i].cpu().numpy(), uvs[150 + i].cpu().numpy(), dst=(320, 180), resize_type=1, fourcc=2, planes=0, normalization=True,
                                   nthreads=8, width=w)
        assert np.array_equal(out[i].cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8)), i
    # another request over the same table: the parameters belong to the call
    fp2 = ts.FrameParameters(width=0, height=0, crop_coords=(64, 32, 576, 328), resize_type=0, pixel_format=1, planes_pos=1, normalization=False)
    out2 = vpp._alloc(fp2.parameters, w, h, n)
    tab2 = vpp.make_table(ys, uvs, fp2, out=out2, width=w)
    vpp.run_table(tab2)
    torch.cuda.synchronize()
    for f in (0, 199):
        _check(oracle, ys, uvs, out2, f, w, (64, 32, 576, 328), (0, 0), 0, 1, 1, False)
    vpp.free_table(tab)
    vpp.free_table(tab2)


@pytest.mark.parametrize("fourcc,dst", [(4, (320, 180)), (5, (300, 200)), (4, (0, 0)), (5, (426, 240))])
def test_two_pass_formats_over_a_table(vpp, oracle, fourcc, dst):
    """UYVY / YUV444: single-pass where the streaming kernel takes them (640x360 -> 426x240 is 3 : 2 ... not exactly: two passes), two passes otherwise --
    through the table's host mirror in launches of <= 128 frames."""
    import tensor_stream as ts
    w, h, pitch, n = 640, 360, 640, 140
    ys, uvs = _pool(w, h, pitch, n, seed=fourcc)
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=1, pixel_format=fourcc, planes_pos=1, normalization=False)
    tab = vpp.make_table(ys, uvs, fp, width=w)
    got = vpp.run_table(tab)
    torch.cuda.synchronize()
    for f in (0, 127, 128, n - 1):
        _check(oracle, ys, uvs, got, f, w, (0, 0, 0, 0), dst, 1, fourcc, 1, False)
    vpp.free_table(tab)


def test_error_paths(vpp):
    import tensor_stream as ts
    from tensor_stream import _native as N
    L = vpp._lib
    w, h, n = 64, 32, 4
    ys, uvs = _pool(w, h, w, n, seed=1)
    fp = ts.FrameParameters(width=32, height=16, resize_type=1, pixel_format=1, planes_pos=0, normalization=False)
    out = vpp._alloc(fp.parameters, w, h, n)
    h_tab = ctypes.c_void_p()
    assert L.tsvpp_table_create(vpp._ctx, 0, ctypes.byref(h_tab)) == -3            # capacity < 1
    assert L.tsvpp_table_create(vpp._ctx, 8, ctypes.byref(h_tab)) == 0
    stream = torch.cuda.current_stream().cuda_stream
    fr = (N.NV12 * n)(*[vpp._frame(ys[i], uvs[i], w, None) for i in range(n)])
    outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
    assert L.tsvpp_table_set(h_tab, 6, n, fr, outs, stream) == -3                  # past the capacity
    assert L.tsvpp_table_set(h_tab, 0, n, fr, outs, stream) == 0
    assert L.tsvpp_convert_table(vpp._ctx, h_tab, 0, 5, ctypes.byref(fp.parameters), stream) == -3   # entry 4 was never set
    assert L.tsvpp_convert_table(vpp._ctx, h_tab, 2, 7, ctypes.byref(fp.parameters), stream) == -3   # past the capacity
    assert L.tsvpp_convert_table(vpp._ctx, h_tab, 0, 0, ctypes.byref(fp.parameters), stream) == 0    # nothing to do
    ys2, uvs2 = _pool(128, 32, 128, 1, seed=2)                                                       # another geometry: one per table
    fr2 = (N.NV12 * 1)(vpp._frame(ys2[0], uvs2[0], 128, None))
    assert L.tsvpp_table_set(h_tab, 4, 1, fr2, outs, stream) == -2
    bad = ts.FrameParameters(width=31, height=16, resize_type=1, pixel_format=1, planes_pos=0, normalization=False)
    assert L.tsvpp_convert_table(vpp._ctx, h_tab, 0, n, ctypes.byref(bad.parameters), stream) == -2  # odd output width: as tsvpp_convert_batch
    assert L.tsvpp_convert_table(vpp._ctx, h_tab, 0, n, ctypes.byref(fp.parameters), stream) == 0
    torch.cuda.synchronize()
    L.tsvpp_table_destroy(h_tab)


def test_a_table_conversion_is_hip_graph_capturable_and_replays_the_table_as_it_is_then(vpp, oracle):
    """tsvpp_convert_table inside a hipGraph: after prepare() it is kernel launches only.  The kernels READ the table at run time, so a replay sees entries that
    were replaced (stream-ordered) after the capture -- the pointer triples are not baked into the graph, unlike the kernarg table of tsvpp_convert_batch."""
    import tensor_stream as ts
    from tensor_stream import _native as N
    w, h, n = 640, 360, 160
    ys, uvs = _pool(w, h, w, n, seed=21)
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=1, planes_pos=0, normalization=True)
    vpp.prepare(fp, w, h, n_frames=n)
    tab = vpp.make_table(ys, uvs, fp, width=w)
    out = tab["out"]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        vpp.run_table(tab, stream=side.cuda_stream)  # warm-up on the capture stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        vpp.run_table(tab, stream=side.cuda_stream)
    # entry 5 now reads frame 150 (into output 5), set on the current stream; then replay
    fr = (N.NV12 * 1)(vpp._frame(ys[150], uvs[150], w, None))
    outs = (ctypes.c_void_p * 1)(out[5].data_ptr())
    N.check(vpp._lib.tsvpp_table_set(tab["handle"], 5, 1, fr, outs, torch.cuda.current_stream().cuda_stream))
    out.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    for f, src in ((0, 0), (5, 150), (159, 159)):
        ref, _, _ = oracle.convert(ys[src].cpu().numpy(), uvs[src].cpu().numpy(), dst=(320, 180), resize_type=1, fourcc=1, planes=0, normalization=True, nthreads=8, width=w)
        assert np.array_equal(out[f].cpu().numpy().ravel().view(np.uint8), ref.view(np.uint8)), (f, src)
    del g
    vpp.free_table(tab)


def test_table_set_twice_before_the_first_upload_ran_keeps_stream_order(vpp, oracle):
    """every upload is staged in its own pinned slot.  The stream is kept busy (a long device-side sleep) while the SAME entries are set twice
    with a conversion behind each: the first conversion must run with the first set's pointers (until round 5 both uploads were copied out of one pinned mirror
    that the second call had already rewritten)."""
    import tensor_stream as ts
    w, h, n = 640, 360, 8
    ysA, uvsA = _pool(w, h, w, n, seed=91)
    ysB, uvsB = _pool(w, h, w, n, seed=92)
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=1, planes_pos=1, normalization=False)
    outA = vpp._alloc(fp.parameters, w, h, n)
    outB = vpp._alloc(fp.parameters, w, h, n)
    outA.zero_()
    outB.zero_()
    tab = vpp.make_table(ysA, uvsA, fp, out=outA, width=w)
    torch.cuda.synchronize()
    lib, N = vpp._lib, __import__("tensor_stream")._native
    s = torch.cuda.Stream()

    def arrays(ys, uvs, out):
        fr = (N.NV12 * n)(*[vpp._frame(ys[i], uvs[i], w, None) for i in range(n)])
        outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        return fr, outs

    with torch.cuda.stream(s):
        torch.cuda._sleep(200_000_000)  # ~0.1 s of device time: everything below queues up behind it
        for rnd in range(3):            # more uploads in flight than a naive double buffer would hold
            for ys, uvs, out in ((ysA, uvsA, outA), (ysB, uvsB, outB)):
                fr, outs = arrays(ys, uvs, out)
                N.check(lib.tsvpp_table_set(tab["handle"], 0, n, fr, outs, s.cuda_stream))
                N.check(lib.tsvpp_convert_table(vpp._ctx, tab["handle"], 0, n, ctypes.byref(fp.parameters), s.cuda_stream))
    s.synchronize()
    for f in (0, n - 1):
        _check(oracle, ysA, uvsA, outA, f, w, (0, 0, 0, 0), (320, 180), 1, 1, 1, False)
        _check(oracle, ysB, uvsB, outB, f, w, (0, 0, 0, 0), (320, 180), 1, 1, 1, False)
    vpp.free_table(tab)


def test_run_table_with_other_parameters_checks_the_registered_outputs(vpp):
    """the outputs a table registered were sized for the parameters it was made with."""
    import tensor_stream as ts
    w, h, n = 640, 360, 4
    ys, uvs = _pool(w, h, w, n, seed=93)
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=1, planes_pos=1, normalization=False)
    tab = vpp.make_table(ys, uvs, fp, width=w)
    vpp.run_table(tab)
    vpp.run_table(tab, params=ts.FrameParameters(width=320, height=180, resize_type=0, pixel_format=2, planes_pos=0, normalization=False))  # same size: fine
    vpp.run_table(tab, params=ts.FrameParameters(width=160, height=90, resize_type=1, pixel_format=1, planes_pos=1, normalization=False))   # smaller: fine
    for bad in (ts.FrameParameters(width=640, height=360, pixel_format=1), ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=1, normalization=True)):
        with pytest.raises(RuntimeError):
            vpp.run_table(tab, params=bad)
    torch.cuda.synchronize()
    vpp.free_table(tab)


def test_context_destroyed_before_its_table():
    """tsvpp_destroy releases the device memory of live tables; their handles stay valid for tsvpp_table_destroy (and for nothing else)."""
    import tensor_stream as ts
    from tensor_stream import _native as N
    v = ts.VideoProcessor(device=0)
    w, h, n = 320, 180, 2
    ys, uvs = _pool(w, h, w, n, seed=94)
    fp = ts.FrameParameters(pixel_format=1)
    tab = v.make_table(ys, uvs, fp, width=w)
    v.run_table(tab)
    torch.cuda.synchronize()
    handle, ctx = tab["handle"], v._ctx
    v._tables = []          # (Close would destroy the table first: take it away)
    lib = N.lib()
    lib.tsvpp_destroy(ctx)
    v._ctx = ctypes.c_void_p()
    fr = (N.NV12 * n)(*[v._frame(ys[i], uvs[i], w, None) for i in range(n)])
    outs = (ctypes.c_void_p * n)(*[tab["out"][i].data_ptr() for i in range(n)])
    assert lib.tsvpp_table_set(handle, 0, n, fr, outs, None) == -3   # orphaned: VREADER_ERROR, no crash
    lib.tsvpp_table_destroy(handle)
