"""Python host side of the VPP path: enums + FrameParameters mirroring the reference's Python
surface (tensor_stream/tensor_stream.py:48-136), and `VideoProcessor`, the counterpart of the C++
class (include/VideoProcessor.h:120-149) that drives the HIP kernels through the C ABI.

PyTorch is plumbing here: it owns device memory and the current stream; all compute is in libtsvpp.so.
"""
import ctypes
from enum import Enum

import torch

from . import _native as N


OPT_INPUTS_READY = N.TSVPP_OPT_INPUTS_READY
OPT_COLOR_G_TERM = N.TSVPP_OPT_COLOR_G_TERM
OPT_UNSAFE_COEFFS = N.TSVPP_OPT_UNSAFE_COEFFS


class FourCC(Enum):  # reference tensor_stream/tensor_stream.py:48-62
    Y800 = 0
    RGB24 = 1
    BGR24 = 2
    NV12 = 3
    UYVY = 4
    YUV444 = 5
    HSV = 6


class ResizeType(Enum):  # :67-75
    NEAREST = 0
    BILINEAR = 1
    BICUBIC = 2
    AREA = 3


class Planes(Enum):  # :79-83
    PLANAR = 0
    MERGED = 1


def _v(x):
    return x.value if isinstance(x, Enum) else int(x)


class FrameParameters:
    """Same constructor as the reference's FrameParameters (tensor_stream/tensor_stream.py:101-136).
    normalization=None keeps the C++ default: False, except True for HSV (include/VideoProcessor.h:40-47)."""

    def __init__(self, width=0, height=0, crop_coords=(0, 0, 0, 0), resize_type=ResizeType.NEAREST,
                 pixel_format=FourCC.RGB24, planes_pos=Planes.MERGED, normalization=None):
        if normalization is None:
            normalization = _v(pixel_format) == FourCC.HSV.value
        self.parameters = N.Params(int(crop_coords[0]), int(crop_coords[1]), int(crop_coords[2]), int(crop_coords[3]),
                                   int(width), int(height), _v(resize_type), _v(pixel_format), _v(planes_pos),
                                   int(bool(normalization)))

    def __repr__(self):
        p = self.parameters
        return (f"FrameParameters(width={p.dst_width}, height={p.dst_height}, "
                f"crop=({p.crop_left},{p.crop_top},{p.crop_right},{p.crop_bottom}), resize_type={p.resize_type}, "
                f"pixel_format={p.fourcc}, planes_pos={p.planes}, normalization={bool(p.normalization)})")


def output_shape(p, out_w, out_h):
    """Tensor shape of TensorStream::getFrame (reference src/Wrappers/WrapperPython.cpp:317-341)."""
    ch = N.lib().tsvpp_channels(p.fourcc)
    if p.fourcc in (FourCC.RGB24.value, FourCC.BGR24.value):
        return (out_h, out_w, 3) if p.planes == Planes.MERGED.value else (3, out_h, out_w)
    if p.fourcc in (FourCC.YUV444.value, FourCC.HSV.value):
        return (out_h, out_w, 3)
    return (1, int(out_h * ch), out_w)


class VideoProcessor:
    """Counterpart of the reference's `VideoProcessor` (Init / Convert / Close), with
    caller-visible torch tensors instead of AVFrame::opaque.

    Convert() takes the NV12 planes as uint8 CUDA tensors (2-D: rows x pitch) and returns a new
    tensor, or fills `out`.  Work is enqueued on torch's current stream unless a consumer name is
    given, in which case that consumer's pooled stream is used (reference src/VideoProcessor.cpp:98-104)
    after making it wait for the current stream.
    """

    def __init__(self, device=None, max_consumers=5):
        self._ctx = ctypes.c_void_p()
        self._tables = []    # persistent frame tables (make_table): destroyed before the context
        self._lib = N.lib()  # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("VideoProcessor needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        N.check(self._lib.tsvpp_create(self.device, int(max_consumers), ctypes.byref(self._ctx)))

    # reference naming
    def Close(self):
        if self._ctx:
            if self._tables:
                torch.cuda.synchronize(self.device)
                for h in self._tables:
                    self._lib.tsvpp_table_destroy(h)
                self._tables = []
            self._lib.tsvpp_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    close = Close

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass

    def out_dims(self, params, in_w, in_h):
        p = params.parameters if isinstance(params, FrameParameters) else params
        w, h = ctypes.c_int(0), ctypes.c_int(0)
        N.check(self._lib.tsvpp_out_dims(ctypes.byref(p), in_w, in_h, ctypes.byref(w), ctypes.byref(h)))
        return w.value, h.value

    def prepare(self, params, in_w, in_h, n_frames=0, stream=None):
        """Pre-builds what the request needs (AREA tables, the geometry tables of the 2x2-tap kernel for a batch of `n_frames`;
        with n_frames > 0 also the UYVY / YUV444 scratch of `stream`, default torch's current stream) so that later
        conversions allocate nothing -- e.g. before graph capture."""
        p = params.parameters if isinstance(params, FrameParameters) else params
        if n_frames and stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.tsvpp_prepare_batch(self._ctx, ctypes.byref(p), in_w, in_h, int(n_frames), stream or 0))

    def enable_markers(self, on=True):
        """roctx ranges around every conversion (the reference's NVTX ranges); RuntimeError if no roctx library exists."""
        N.check(self._lib.tsvpp_enable_markers(self._ctx, 1 if on else 0))

    def consumer_stream(self, name):
        s = ctypes.c_void_p()
        N.check(self._lib.tsvpp_consumer_stream(self._ctx, name.encode(), ctypes.byref(s)))
        return s.value or 0

    def consumer_next_stream(self, name, launch_bytes=0):
        """The stream the consumer's next conversion should go to (tsvpp_consumer_next_stream): its one stream by default; under OPT_INPUTS_READY small launches
        alternate between the consumer's two streams."""
        s = ctypes.c_void_p()
        N.check(self._lib.tsvpp_consumer_next_stream(self._ctx, name.encode(), int(launch_bytes), ctypes.byref(s)))
        return s.value or 0

    def consumer_synchronize(self, name):
        N.check(self._lib.tsvpp_consumer_synchronize(self._ctx, name.encode()))

    def set_option(self, option, value):
        """Context options of include/tsvpp.h, e.g. set_option(OPT_INPUTS_READY, 1)."""
        N.check(self._lib.tsvpp_set_option(self._ctx, int(option), int(value)))

    def get_option(self, option):
        v = ctypes.c_int(0)
        N.check(self._lib.tsvpp_get_option(self._ctx, int(option), ctypes.byref(v)))
        return v.value

    def _alloc(self, p, in_w, in_h, n=None):
        ow, oh = self.out_dims(p, in_w, in_h)
        shape = output_shape(p, ow, oh)
        dtype = torch.float32 if (p.normalization or p.fourcc == FourCC.HSV.value) else torch.uint8
        if n is None:
            return torch.empty(shape, dtype=dtype, device=f"cuda:{self.device}")
        # batch: every frame must start 16-byte aligned or its launch group falls back to the element-wise kernel
        # (uint8 frames of e.g. 250 x 250 x 3 bytes are not a multiple of 16): pad the frame stride, return a view
        numel = 1
        for d in shape:
            numel *= d
        esz = 4 if dtype == torch.float32 else 1
        stride = (numel * esz + 15) // 16 * 16 // esz
        flat = torch.empty(n * stride, dtype=dtype, device=f"cuda:{self.device}")
        inner = []
        acc = 1
        for d in reversed(shape):
            inner.insert(0, acc)
            acc *= d
        return flat.as_strided((n,) + tuple(shape), (stride,) + tuple(inner))

    @staticmethod
    def _frame(y, uv, width, height):
        assert y.is_cuda and uv.is_cuda and y.dtype == torch.uint8 and uv.dtype == torch.uint8
        assert y.dim() == 2 and uv.dim() == 2 and y.stride(1) == 1 and uv.stride(1) == 1
        h = y.shape[0] if height is None else height
        w = y.shape[1] if width is None else width
        return N.NV12(y.data_ptr(), uv.data_ptr(), y.stride(0), uv.stride(0), w, h)

    def Convert(self, y, uv, params, out=None, width=None, height=None, consumer=None):
        p = params.parameters if isinstance(params, FrameParameters) else params
        fr = self._frame(y, uv, width, height)
        if out is None:
            out = self._alloc(p, fr.width, fr.height)
        cur = torch.cuda.current_stream(self.device)
        stream = cur.cuda_stream
        ext = None
        if consumer is not None:
            stream, ext = self._on_consumer_stream(consumer)
        N.check(self._lib.tsvpp_convert(self._ctx, ctypes.byref(fr), ctypes.byref(p), out.data_ptr(), stream))
        if ext is not None:
            cur.wait_stream(ext)  # later work on torch's stream sees the finished tensor (no host sync)
        return out

    convert = Convert

    def convert_batch(self, ys, uvs, params, out=None, width=None, height=None):
        """ys / uvs: 3-D uint8 tensors (n, rows, pitch) or lists of 2-D tensors; one launch per 128 frames.
        Returns (or fills `out`, indexed out[i]) a tensor of shape (n, ...frame shape).  NOTE the frame stride of the tensor this
        method allocates: every frame starts 16-byte aligned (the vector-store kernels need that), so when a frame's bytes are
        not a multiple of 16 (e.g. 250 x 250 x 3 uint8) the batch tensor is a strided VIEW with a padded stride(0) -- out[i] is
        contiguous, the batch as a whole is not (`.contiguous()` compacts it; pass your own `out` to choose the layout: frames
        that are not 16-byte aligned then take the element-wise kernel)."""
        p = params.parameters if isinstance(params, FrameParameters) else params
        n = len(ys)
        frames = (N.NV12 * n)(*[self._frame(ys[i], uvs[i], width, height) for i in range(n)])
        if out is None:
            out = self._alloc(p, frames[0].width, frames[0].height, n)
        outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.tsvpp_convert_batch(self._ctx, n, frames, ctypes.byref(p), outs, stream))
        return out

    def make_batch(self, ys, uvs, params, out=None, width=None, height=None):
        """Pre-builds the descriptor arrays of a batch (the per-frame structs are built once, not per
        call): returns a handle for run_batch().  Keeps the tensors alive."""
        p = params.parameters if isinstance(params, FrameParameters) else params
        n = len(ys)
        frames = (N.NV12 * n)(*[self._frame(ys[i], uvs[i], width, height) for i in range(n)])
        if out is None:
            out = self._alloc(p, frames[0].width, frames[0].height, n)
        outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        return {"n": n, "frames": frames, "outs": outs, "params": p, "out": out, "keep": (ys, uvs)}

    def run_batch(self, batch, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.tsvpp_convert_batch(self._ctx, batch["n"], batch["frames"], ctypes.byref(batch["params"]),
                                              batch["outs"], stream))
        return batch["out"]

    def make_table(self, ys, uvs, params, out=None, width=None, height=None):
        """A persistent, device-resident frame table (tsvpp_table_*, include/tsvpp.h) over a POOL of input frames and output tensors: registered once, any
        run of its entries is then converted by run_table() with launches of up to 1024 frames and no per-call pointer traffic.  ys / uvs / out as for
        convert_batch; the handle keeps the tensors alive; free_table() (or Close) releases the table."""
        p = params.parameters if isinstance(params, FrameParameters) else params
        n = len(ys)
        frames = (N.NV12 * n)(*[self._frame(ys[i], uvs[i], width, height) for i in range(n)])
        if out is None:
            out = self._alloc(p, frames[0].width, frames[0].height, n)
        outs = (ctypes.c_void_p * n)(*[out[i].data_ptr() for i in range(n)])
        h = ctypes.c_void_p()
        N.check(self._lib.tsvpp_table_create(self._ctx, n, ctypes.byref(h)))
        self._tables.append(h)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.tsvpp_table_set(h, 0, n, frames, outs, stream))
        return {"n": n, "handle": h, "params": p, "out": out, "keep": (ys, uvs), "width": frames[0].width}

    def run_table(self, table, first=0, n=None, params=None, stream=None):
        """Converts entries [first, first + n) of a table made by make_table (default: all of it, with the parameters it was made with)."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        p = table["params"] if params is None else (params.parameters if isinstance(params, FrameParameters) else params)
        cnt = table["n"] - first if n is None else n
        if params is not None:
            #  the table's output tensors were allocated for the parameters it was made with and their addresses live in the device table: other
            # parameters are fine only while every frame still fits its registered output
            y0 = table["keep"][0][0]
            w = table.get("width") or y0.shape[1]
            need = int(self._lib.tsvpp_out_bytes(ctypes.byref(p), int(w), int(y0.shape[0])))
            have = table["out"][0].numel() * table["out"].element_size()
            if need == 0 or need > have:
                raise RuntimeError(f"-3: run_table(params=...) needs {need} bytes per frame, the table's outputs hold {have}")
        N.check(self._lib.tsvpp_convert_table(self._ctx, table["handle"], first, cnt, ctypes.byref(p), stream))
        return table["out"]

    def free_table(self, table):
        h = table.get("handle")
        if h is not None and any(h is t for t in self._tables):
            torch.cuda.synchronize(self.device)
            self._lib.tsvpp_table_destroy(h)
            self._tables = [t for t in self._tables if t is not h]
            table["handle"] = None

    def trim(self):
        """Releases the memory the context has retired (geometry tables pushed out of its cache, outgrown scratch buffers): tsvpp_trim.  Synchronises the device
        first -- the C call itself leaves that to the caller."""
        torch.cuda.synchronize(self.device)
        n = ctypes.c_size_t(0)
        N.check(self._lib.tsvpp_trim(self._ctx, ctypes.byref(n)))
        return int(n.value)

    def _on_consumer_stream(self, name):
        raw = self.consumer_stream(name)
        ext = torch.cuda.ExternalStream(raw, device=self.device)
        ext.wait_stream(torch.cuda.current_stream(self.device))
        return raw, ext

    def get_coeffs(self):
        c = N.Coeffs()
        N.check(self._lib.tsvpp_get_coeffs(self._ctx, ctypes.byref(c)))
        return [getattr(c, f[0]) for f in N.Coeffs._fields_]

    def set_coeffs(self, values):
        c = N.Coeffs(*[float(v) for v in values])
        N.check(self._lib.tsvpp_set_coeffs(self._ctx, ctypes.byref(c)))


def describe(params, in_w, in_h, pitch=0, n_frames=64, aligned_outputs=True):
    """What a convert_batch of this request would launch (stage selection, kernel, workgroup shape, LDS bytes, grid)
    as a dict -- host logic only, works without a GPU (tsvpp_describe)."""
    p = params.parameters if isinstance(params, FrameParameters) else params
    buf = ctypes.create_string_buffer(512)
    N.check(N.lib().tsvpp_describe(ctypes.byref(p), in_w, in_h, pitch, pitch, n_frames, 1 if aligned_outputs else 0, buf, len(buf)))
    out = {}
    for item in buf.value.decode().split(" "):
        k, _, v = item.partition("=")
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


def default_coeffs():
    c = N.Coeffs()
    N.lib().tsvpp_default_coeffs(ctypes.byref(c))
    return [getattr(c, f[0]) for f in N.Coeffs._fields_]
