"""ctypes front-end of the CPU oracle (oracle/vpp_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

# libgomp workers must sleep, not spin, between parallel regions: spinning threads starve the HIP
# runtime's completion thread of the process that also drives the GPU (bench.py, GPU tests).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvpp_oracle.so")

Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = range(7)
PLANAR, MERGED = 0, 1
NEAREST, BILINEAR, BICUBIC, AREA = range(4)


def build(force=False):
    src = os.path.join(_HERE, "vpp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvpp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, vp, ip = ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)
        L.vpp_oracle_convert.argtypes = [u8p, u8p] + [ctypes.c_int] * 14 + [vp, ip, ip, ctypes.c_int]
        L.vpp_oracle_convert.restype = ctypes.c_int
        L.vpp_oracle_crop.argtypes = [u8p, u8p] + [ctypes.c_int] * 8 + [u8p, u8p]
        L.vpp_oracle_resize.argtypes = [u8p, u8p] + [ctypes.c_int] * 7 + [u8p, u8p, ctypes.c_int]
        L.vpp_oracle_area_pattern.argtypes = [ctypes.c_float, vp, ctypes.c_int, ip]
        L.vpp_oracle_channels.argtypes = [ctypes.c_int]
        L.vpp_oracle_channels.restype = ctypes.c_float
        L.vpp_oracle_av_crc32_ieee.argtypes = [ctypes.c_uint32, u8p, ctypes.c_long]
        L.vpp_oracle_av_crc32_ieee.restype = ctypes.c_uint32
        _lib = L
    return _lib


def set_exact_index(on):
    """BILINEAR / AREA-up start index in integers (the HIP kernels' stated behaviour) instead of the reference's float
    expression, which is exact only while pitch * height <= 2^24 (see bilinear_tap in vpp_oracle.c)."""
    lib().vpp_oracle_set_exact_index(1 if on else 0)


def set_contract(bits):
    """Which a * b + c pairs are evaluated as fused multiply-adds (see vpp_oracle.c: CT_* bits); 0 = none."""
    lib().vpp_oracle_set_contract(int(bits))


def channels(fourcc):
    return float(lib().vpp_oracle_channels(int(fourcc)))


def out_dims(w, h, crop=(0, 0, 0, 0), dst=(0, 0)):
    """Stage selection of VideoProcessor::Convert (reference src/VideoProcessor.cpp:106-135)."""
    cw, ch = crop[2] - crop[0], crop[3] - crop[1]
    ow, oh = w, h
    if cw > 0 and ch > 0 and cw < w and ch < h:
        ow, oh = cw, ch
    if dst[0] and dst[1]:
        ow, oh = dst
    return ow, oh


def host_cores():
    """Cores this process may actually use: CPU affinity, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def convert(y, uv, crop=(0, 0, 0, 0), dst=(0, 0), resize_type=NEAREST, fourcc=RGB24,
            planes=MERGED, normalization=False, nthreads=1, width=None):
    """y: (H, pitch_y) uint8, uv: (H/2, pitch_uv) uint8; width defaults to y.shape[1]."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    uv = np.ascontiguousarray(uv, dtype=np.uint8)
    h, pitch_y = y.shape
    pitch_uv = uv.shape[1]
    w = pitch_y if width is None else width
    ow, oh = out_dims(w, h, crop, dst)
    is_float = bool(normalization) or fourcc == HSV
    n = int(round(channels(fourcc) * ow * oh))
    out = np.zeros(n, dtype=np.float32 if is_float else np.uint8)
    rw, rh = ctypes.c_int(0), ctypes.c_int(0)
    sts = lib().vpp_oracle_convert(y.ctypes.data, uv.ctypes.data, pitch_y, pitch_uv, w, h,
                                   crop[0], crop[1], crop[2], crop[3], dst[0], dst[1], int(resize_type),
                                   int(fourcc), int(planes), int(bool(normalization)),
                                   out.ctypes.data, ctypes.byref(rw), ctypes.byref(rh), int(nthreads))
    if sts != 0:
        raise RuntimeError(f"oracle status {sts}")
    assert (rw.value, rh.value) == (ow, oh), ((rw.value, rh.value), (ow, oh))
    return out, ow, oh


def shape_for(fourcc, planes, ow, oh):
    """Tensor shapes of TensorStream::getFrame (reference src/Wrappers/WrapperPython.cpp:317-341)."""
    if fourcc in (RGB24, BGR24):
        return (oh, ow, 3) if planes == MERGED else (3, oh, ow)
    if fourcc in (YUV444, HSV):
        return (oh, ow, 3)
    return (1, int(oh * channels(fourcc)), ow)


def crop_stage(y, uv, l, t, r, b, width=None):
    y = np.ascontiguousarray(y, dtype=np.uint8); uv = np.ascontiguousarray(uv, dtype=np.uint8)
    h, py = y.shape; w = py if width is None else width
    cw, ch = r - l, b - t
    oy = np.zeros((ch, cw), np.uint8); ouv = np.zeros((ch // 2, cw), np.uint8)
    lib().vpp_oracle_crop(y.ctypes.data, uv.ctypes.data, py, uv.shape[1], w, h, l, t, r, b, oy.ctypes.data, ouv.ctypes.data)
    return oy, ouv


def resize_stage(y, uv, dw, dh, rtype, width=None, nthreads=1):
    y = np.ascontiguousarray(y, dtype=np.uint8); uv = np.ascontiguousarray(uv, dtype=np.uint8)
    h, py = y.shape; w = py if width is None else width
    oy = np.zeros((dh, dw), np.uint8); ouv = np.zeros((dh // 2, dw), np.uint8)
    sts = lib().vpp_oracle_resize(y.ctypes.data, uv.ctypes.data, py, uv.shape[1], w, h, dw, dh, int(rtype),
                                  oy.ctypes.data, ouv.ctypes.data, nthreads)
    if sts != 0:
        raise RuntimeError(f"oracle status {sts}")
    return oy, ouv


def area_pattern(scale):
    buf = np.zeros(1 << 20, np.float32)
    st = ctypes.c_int(0)
    n = lib().vpp_oracle_area_pattern(ctypes.c_float(scale), buf.ctypes.data, buf.size, ctypes.byref(st))
    if n <= 0:
        raise RuntimeError("pattern failed")
    return buf[: n * st.value].reshape(n, st.value).copy()


def av_crc32_ieee(data, crc=0xFFFFFFFF):
    data = np.ascontiguousarray(data).view(np.uint8).ravel()
    return int(lib().vpp_oracle_av_crc32_ieee(ctypes.c_uint32(crc), data.ctypes.data, data.size))
