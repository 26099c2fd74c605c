"""ctypes binding of libtsvpp.so (the C ABI in include/tsvpp.h).

Fails loudly when the HIP library is missing: there is no CPU or PyTorch fallback for the VPP path.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_PKG, "..", "lib", "libtsvpp.so"))

TSVPP_MAX_BATCH = 128
TSVPP_OPT_INPUTS_READY = 1
TSVPP_OPT_COLOR_G_TERM = 2
TSVPP_OPT_UNSAFE_COEFFS = 3


class NV12(ctypes.Structure):
    """struct tsvpp_nv12: the AVFrame fields VideoProcessor::Convert reads."""
    _fields_ = [("y", ctypes.c_void_p), ("uv", ctypes.c_void_p),
                ("pitch_y", ctypes.c_int32), ("pitch_uv", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32)]


class Params(ctypes.Structure):
    """struct tsvpp_params: flat FrameParameters {resize, color, crop}."""
    _fields_ = [("crop_left", ctypes.c_int32), ("crop_top", ctypes.c_int32),
                ("crop_right", ctypes.c_int32), ("crop_bottom", ctypes.c_int32),
                ("dst_width", ctypes.c_int32), ("dst_height", ctypes.c_int32),
                ("resize_type", ctypes.c_int32), ("fourcc", ctypes.c_int32),
                ("planes", ctypes.c_int32), ("normalization", ctypes.c_int32)]


class Coeffs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in
                ("y_scale", "v_to_r", "u_to_b", "v_to_g", "u_to_g", "round_bias", "y_offset", "c_offset")]


# every symbol include/tsvpp.h declares (tests check that the library exports all of them)
SYMBOLS = ["tsvpp_create", "tsvpp_destroy", "tsvpp_consumer_stream", "tsvpp_out_dims", "tsvpp_out_bytes",
           "tsvpp_channels", "tsvpp_convert", "tsvpp_convert_batch", "tsvpp_prepare", "tsvpp_prepare_batch", "tsvpp_enable_markers", "tsvpp_get_coeffs",
           "tsvpp_set_coeffs", "tsvpp_default_coeffs", "tsvpp_area_pattern", "tsvpp_describe", "tsvpp_strerror", "tsvpp_version",
           "tsvpp_table_create", "tsvpp_table_destroy", "tsvpp_table_set", "tsvpp_convert_table", "tsvpp_trim", "tsvpp_set_option", "tsvpp_get_option", "tsvpp_consumer_next_stream", "tsvpp_consumer_synchronize"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the VPP path is HIP-only (gfx950). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tensor-stream_amd/csrc`.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    pp, pn = ctypes.POINTER(Params), ctypes.POINTER(NV12)
    L.tsvpp_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    L.tsvpp_destroy.argtypes = [vp]
    L.tsvpp_destroy.restype = None
    L.tsvpp_consumer_stream.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp)]
    L.tsvpp_consumer_next_stream.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.tsvpp_consumer_next_stream.restype = i32
    L.tsvpp_consumer_synchronize.argtypes = [vp, ctypes.c_char_p]
    L.tsvpp_consumer_synchronize.restype = i32
    L.tsvpp_out_dims.argtypes = [pp, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.tsvpp_out_bytes.argtypes = [pp, i32, i32]
    L.tsvpp_out_bytes.restype = ctypes.c_size_t
    L.tsvpp_channels.argtypes = [i32]
    L.tsvpp_channels.restype = ctypes.c_float
    L.tsvpp_convert.argtypes = [vp, pn, pp, vp, vp]
    L.tsvpp_convert_batch.argtypes = [vp, i32, pn, pp, ctypes.POINTER(vp), vp]
    L.tsvpp_table_create.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.tsvpp_table_destroy.argtypes = [vp]
    L.tsvpp_table_destroy.restype = None
    L.tsvpp_table_set.argtypes = [vp, i32, i32, pn, ctypes.POINTER(vp), vp]
    L.tsvpp_convert_table.argtypes = [vp, vp, i32, i32, pp, vp]
    L.tsvpp_trim.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    L.tsvpp_trim.restype = i32
    L.tsvpp_set_option.argtypes = [vp, i32, i32]
    L.tsvpp_set_option.restype = i32
    L.tsvpp_get_option.argtypes = [vp, i32, ctypes.POINTER(i32)]
    L.tsvpp_get_option.restype = i32
    L.tsvpp_prepare.argtypes = [vp, pp, i32, i32]
    L.tsvpp_prepare_batch.argtypes = [vp, pp, i32, i32, i32, vp]
    L.tsvpp_enable_markers.argtypes = [vp, i32]
    L.tsvpp_get_coeffs.argtypes = [vp, ctypes.POINTER(Coeffs)]
    L.tsvpp_set_coeffs.argtypes = [vp, ctypes.POINTER(Coeffs)]
    L.tsvpp_default_coeffs.argtypes = [ctypes.POINTER(Coeffs)]
    L.tsvpp_default_coeffs.restype = None
    L.tsvpp_area_pattern.argtypes = [ctypes.c_float, vp, i32, ctypes.POINTER(i32)]
    L.tsvpp_describe.argtypes = [pp, i32, i32, i32, i32, i32, i32, ctypes.c_char_p, ctypes.c_size_t]
    L.tsvpp_strerror.argtypes = [i32]
    L.tsvpp_strerror.restype = ctypes.c_char_p
    L.tsvpp_version.argtypes = []
    L.tsvpp_version.restype = ctypes.c_char_p
    for f in ("tsvpp_create", "tsvpp_consumer_stream", "tsvpp_out_dims", "tsvpp_convert", "tsvpp_convert_batch",
              "tsvpp_prepare", "tsvpp_prepare_batch", "tsvpp_enable_markers", "tsvpp_get_coeffs", "tsvpp_set_coeffs", "tsvpp_area_pattern", "tsvpp_describe",
              "tsvpp_table_create", "tsvpp_table_set", "tsvpp_convert_table"):
        getattr(L, f).restype = i32
    _lib = L
    return L


def check(status):
    """The reference turns a non-zero status into std::runtime_error(std::to_string(status))
    (include/Common.h:116-123), which pybind surfaces as RuntimeError; same here, plus the text."""
    if status != 0:
        raise RuntimeError(f"{status}: {lib().tsvpp_strerror(status).decode()}")
