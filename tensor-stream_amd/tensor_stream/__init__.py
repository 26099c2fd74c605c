"""tensor_stream: MI355X-native Video Post Processing path behind the reference's Python surface.

Same package name as the reference's (`from tensor_stream import ...`), so user code switches by
putting `tensor-stream_amd/` on sys.path.
"""
from .vpp import FourCC, FrameParameters, Planes, ResizeType, VideoProcessor, default_coeffs, describe, output_shape  # noqa: F401

from .tensor_stream import FrameRate, FrameRing, LogsLevel, LogsType, StatusLevel, TensorStreamConverter  # noqa: F401

__version__ = "0.1.0"
