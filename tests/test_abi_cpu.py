"""CPU: the C-ABI library loads and exports every symbol include/tsvpp.h declares; host-side logic
(stage selection, sizes, AREA weight tables, status strings) -- no compute calls without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from tensor_stream import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _native


def test_library_exports_every_declared_symbol(native):
    hdr = open(os.path.join(ROOT, "include", "tsvpp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(tsvpp_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/tsvpp.h but not exported"
    assert sorted(native.SYMBOLS) == declared


def test_version_and_strerror(native):
    L = native.lib()
    assert b"gfx950" in L.tsvpp_version()
    assert L.tsvpp_strerror(0) == b"ok"
    assert b"UNSUPPORTED" in L.tsvpp_strerror(-2)
    assert b"ERROR" in L.tsvpp_strerror(-3)


def test_struct_layouts_match_header(native):
    assert ctypes.sizeof(native.NV12) == 32       # 2 pointers + 4 int32
    assert ctypes.sizeof(native.Params) == 40     # 10 int32
    assert ctypes.sizeof(native.Coeffs) == 32     # 8 float


def P(native, **kw):
    p = native.Params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def dims(native, p, w, h):
    ow, oh = ctypes.c_int(-1), ctypes.c_int(-1)
    sts = native.lib().tsvpp_out_dims(ctypes.byref(p), w, h, ctypes.byref(ow), ctypes.byref(oh))
    return sts, ow.value, oh.value


def test_stage_selection_mirrors_convert(native, oracle):
    """reference src/VideoProcessor.cpp:106-135"""
    L = native.lib()
    cases = [
        (dict(fourcc=1), (1920, 1080), (1920, 1080)),
        (dict(fourcc=1, dst_width=1280, dst_height=720), (1920, 1080), (1280, 720)),
        (dict(fourcc=1, dst_width=1280), (1920, 1080), (1920, 1080)),                      # needs both dims
        (dict(fourcc=2, crop_right=1280, crop_bottom=720), (1920, 1080), (1280, 720)),
        (dict(fourcc=2, crop_right=1920, crop_bottom=720), (1920, 1080), (1920, 1080)),    # not smaller in both
        (dict(fourcc=2, crop_right=1280, crop_bottom=720, dst_width=256, dst_height=256), (1920, 1080), (256, 256)),
        (dict(fourcc=2, crop_left=100, crop_top=50, crop_right=90, crop_bottom=40), (640, 360), (640, 360)),  # negative box
    ]
    for kw, (w, h), want in cases:
        p = P(native, **kw)
        assert dims(native, p, w, h) == (0,) + want, kw
        crop = (p.crop_left, p.crop_top, p.crop_right, p.crop_bottom)
        assert oracle.out_dims(w, h, crop, (p.dst_width, p.dst_height) if p.dst_width and p.dst_height else (0, 0)) == want
        assert L.tsvpp_out_bytes(ctypes.byref(p), w, h) == 3 * want[0] * want[1]
        p.normalization = 1
        assert L.tsvpp_out_bytes(ctypes.byref(p), w, h) == 12 * want[0] * want[1]


def test_unsupported_and_error_statuses(native):
    assert dims(native, P(native, fourcc=1, dst_width=321, dst_height=180), 640, 360)[0] == -2   # odd output
    assert dims(native, P(native, fourcc=1), 641, 360)[0] == -2                                    # odd source
    assert dims(native, P(native, fourcc=1, crop_right=101, crop_bottom=100), 640, 360)[0] == -2   # odd crop
    assert dims(native, P(native, fourcc=1, dst_width=320, dst_height=180, resize_type=9), 640, 360)[0] == -2
    assert dims(native, P(native, fourcc=9), 640, 360)[0] == -2
    assert dims(native, P(native, fourcc=1, crop_left=600, crop_right=700, crop_bottom=100), 640, 360)[0] == -3
    assert dims(native, P(native, fourcc=1), 0, 0)[0] == -3
    assert native.lib().tsvpp_out_bytes(ctypes.byref(P(native, fourcc=9)), 640, 360) == 0


def test_table_and_trim_entry_points_reject_null_arguments_without_a_gpu(native):
    """The round-5 entry points (persistent frame tables, tsvpp_trim) check their arguments before they touch a device: callable here, status VREADER_ERROR."""
    L = native.lib()
    h = ctypes.c_void_p()
    p = P(native, fourcc=1)
    assert L.tsvpp_table_create(None, 8, ctypes.byref(h)) == -3 and not h.value
    assert L.tsvpp_table_set(None, 0, 1, None, None, None) == -3
    assert L.tsvpp_convert_table(None, None, 0, 1, ctypes.byref(p), None) == -3
    assert L.tsvpp_trim(None, None) == -3
    L.tsvpp_table_destroy(None)  # a no-op, like free(NULL)


def test_channels(native, oracle):
    for f in range(7):
        assert native.lib().tsvpp_channels(f) == oracle.channels(f)


def test_default_coeffs_are_the_reference_literals(native):
    c = native.Coeffs()
    native.lib().tsvpp_default_coeffs(ctypes.byref(c))
    got = np.array([getattr(c, f[0]) for f in native.Coeffs._fields_], np.float32)
    # reference src/ColorConversion.cu:23,25,30,35
    want = np.array([1.163999557, 1.5959997177, 2.017999649, -0.812999725, -0.390999794, 0.5, 16, 128], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def area_rows(native, scale):
    buf = np.zeros(1 << 18, np.float32)
    taps = ctypes.c_int(0)
    n = native.lib().tsvpp_area_pattern(ctypes.c_float(scale), buf.ctypes.data, buf.size, ctypes.byref(taps))
    assert n > 0
    return buf[: n * taps.value].reshape(n, taps.value)


@pytest.mark.parametrize("src,dst", [(1920, 1280), (3840, 640), (3840, 1280), (1080, 480), (608, 360), (1920, 224),
                                     (1080, 224), (1920, 1000), (2880, 1280), (1920, 1001), (720, 256), (4320, 1024)])
def test_area_weight_table_matches_oracle_restatement(native, oracle, src, dst):
    """The product builds its own table (tsvpp_api.cpp); it must equal the oracle's restatement of
    generateResizePattern (reference src/Resize.cu:359-386) in the entries the kernel reads."""
    scale = np.float32(src) / np.float32(dst)
    mine = area_rows(native, float(scale))
    ref = oracle.area_pattern(float(scale))
    taps = int(np.ceil(float(scale)))
    assert mine.shape == (ref.shape[0], taps)
    assert np.array_equal(mine.view(np.uint32), ref[:, :taps].view(np.uint32))


def test_area_known_tables(native):
    """SURVEY.md 8-P.4 probe values."""
    assert area_rows(native, 1.5).tolist() == [[1.0, 0.5], [0.5, 1.0]]
    assert area_rows(native, 3.0).tolist() == [[1.0, 1.0, 1.0]]
    assert area_rows(native, 6.0).tolist() == [[1.0] * 6]
    assert area_rows(native, 2.25).tolist() == [[1, 1, .25], [.75, 1, .5], [.5, 1, .75], [.25, 1, 1]]
    assert area_rows(native, 7.5).shape == (2, 8)
    assert area_rows(native, 2.8125).shape == (16, 3)


def test_python_frame_parameters_defaults(native):
    import tensor_stream as ts
    p = ts.FrameParameters().parameters          # reference defaults: RGB24, MERGED, NEAREST, no crop, norm off
    assert (p.fourcc, p.planes, p.resize_type, p.normalization, p.dst_width, p.dst_height) == (1, 1, 0, 0, 0, 0)
    assert ts.FrameParameters(pixel_format=ts.FourCC.HSV).parameters.normalization == 1   # include/VideoProcessor.h:45-46
    assert ts.output_shape(ts.FrameParameters(planes_pos=ts.Planes.PLANAR).parameters, 320, 240) == (3, 240, 320)
    assert ts.output_shape(ts.FrameParameters().parameters, 320, 240) == (240, 320, 3)


def test_cpp_host_library_links_against_the_abi(native):
    """libtsvpp_host.so (class VideoProcessor) is built and resolves every tsvpp_* it uses from libtsvpp.so."""
    import subprocess
    host = os.path.join(os.path.dirname(native.LIB_PATH), "libtsvpp_host.so")
    if not os.path.exists(host):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--undefined-only", host], capture_output=True, text=True).stdout
    used = set(re.findall(r"\b(tsvpp_[a-z0-9_]+)", out))
    assert {"tsvpp_create", "tsvpp_convert", "tsvpp_consumer_next_stream", "tsvpp_out_dims", "tsvpp_destroy"} <= used
    assert used <= set(native.SYMBOLS)
    defined = subprocess.run(["nm", "-D", "--defined-only", host], capture_output=True, text=True).stdout
    assert "VideoProcessor7Convert" in defined and "VideoProcessor4Init" in defined and "channelsByFourCC" in defined


def test_no_product_code_touches_the_oracle():
    """The product path must never import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "tensor-stream_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                code = "\n".join(l for l in txt.splitlines() if not l.strip().startswith(("//", "#", "*", "/*")))
                assert "vpp_oracle" not in code and "from oracle" not in code and "import oracle" not in code, os.path.join(dp, f)


def test_product_fails_loudly_without_the_hip_library_or_a_gpu(monkeypatch):
    """No CPU fallback anywhere on the product path: a missing libtsvpp.so is a RuntimeError naming the build command, and
    on a box without a GPU the context cannot be created (tsvpp_create returns the HIP error, surfaced as RuntimeError)."""
    import torch
    from tensor_stream import _native as N
    import tensor_stream as ts
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libtsvpp.so")
    with pytest.raises(RuntimeError, match="HIP-only"):
        N.lib()
    monkeypatch.undo()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ts.VideoProcessor(device=0)


def test_cpp_mirror_exports_the_reference_class_and_stage_launchers():
    """libtsvpp_host.so (tensor-stream_amd/cpp): the reference's public C++ surface for this path -- class VideoProcessor (Init, Convert,
    DumpFrame, Close; include/VideoProcessor.h:117-147), channelsByFourCC and the three stage launchers (:110-115) -- by their mangled names,
    with the reference's signatures (hipStream_t in place of cudaStream_t)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tensor-stream_amd", "lib", "libtsvpp_host.so")
    assert os.path.exists(so), "python -c 'import __graft_entry__ as g; g.build()'"
    syms = subprocess.run(["nm", "-D", "--defined-only", "-C", so], capture_output=True, text=True, check=True).stdout
    for needle in ("cropHost(AVFrame*, AVFrame*, CropOptions, int, ihipStream_t**)",
                   "resizeKernel(AVFrame*, AVFrame*, bool, ResizeOptions, int, ihipStream_t**)",
                   "int colorConversionKernel<float>(AVFrame*, AVFrame*, ColorOptions, int, ihipStream_t**)",
                   "int colorConversionKernel<unsigned char>(AVFrame*, AVFrame*, ColorOptions, int, ihipStream_t**)",
                   "VideoProcessor::Init(", "VideoProcessor::Convert(AVFrame*, AVFrame*, FrameParameters&, std::", "VideoProcessor::DumpFrame<",
                   "VideoProcessor::Close()", "channelsByFourCC(FourCC)"):
        assert needle in syms, needle


def test_debug_knobs_are_honoured_only_under_the_gate():
    """the 25 TSVPP_* A/B knobs act only under TSVPP_DEBUG_KNOBS=1; without the gate a set knob is reported once on stderr and ignored.
    tsvpp_describe runs the same read_env_knobs as tsvpp_create: observable without a GPU."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import tensor_stream as ts; "
            "print(ts.describe(ts.FrameParameters(width=1280, height=720, resize_type=1, pixel_format=2, planes_pos=0, normalization=True), 1920, 1080, pitch=2048)['kernel'])"
            % os.path.join(ROOT, "tensor-stream_amd"))
    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("TSVPP_")}
        env.update(env_extra)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
        assert p.returncode == 0, p.stderr[-400:]
        return p.stdout.strip(), p.stderr
    base, err = run({})
    assert "vpp_bilinear_kernel" in base and "tsvpp:" not in err
    ignored, err = run({"TSVPP_FORCE_GATHER": "1"})
    assert ignored == base and "ignoring TSVPP_FORCE_GATHER=1" in err
    forced, err = run({"TSVPP_FORCE_GATHER": "1", "TSVPP_DEBUG_KNOBS": "1"})
    assert forced != base and "gather" in forced


def test_adapter_compiles_against_a_libavutil_frame_header():
    """the TSVPP_HAVE_LIBAV branch of cpp/VideoProcessor.h (AVFrame from <libavutil/frame.h>) goes through hipcc -- against a vendored
    compile-check declaration (cpp/compat/libavutil/frame.h), objects only."""
    import subprocess
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "tensor-stream_amd", "cpp"), "libav-check"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "libav-check ok" in p.stdout, (p.stdout[-300:], p.stderr[-600:])
