"""TensorStreamConverter: the reference's Python surface (tensor_stream/tensor_stream.py:153-340) on ROCm.

Same constructor, same methods, same `read()` signature and tensor shapes/dtypes.  Behind it:
  * a producer thread standing in for Parser + Decoder (see sources.py) that keeps the last `buffer_size` NV12 frames
    in device memory and hands them to named consumers with the semantics of Decoder::GetFrame
    (reference src/Decoder.cpp:97-131): every consumer sees each published frame at most once, `delay` in
    [-buffer_size, 0] selects older frames, and reads after the end raise RuntimeError("Decoding finished");
  * the HIP VideoProcessor for the conversion, one pooled stream per consumer name
    (reference src/VideoProcessor.cpp:98-104), output written straight into a torch-owned tensor -- which also
    retires the reference's use_count()-based garbage collection of output buffers
    (src/Wrappers/WrapperPython.cpp:173-184).
"""
import logging
import threading
import time
from enum import Enum

import torch

from .sources import open_source
from .vpp import FourCC, FrameParameters, Planes, ResizeType, VideoProcessor


class StatusLevel(Enum):
    OK = 0
    REPEAT = 1
    ERROR = 2


class LogsLevel(Enum):
    NONE = 0
    LOW = 1
    MEDIUM = 2
    HIGH = 3


class LogsType(Enum):
    FILE = 1
    CONSOLE = 2


class FrameRate(Enum):
    NATIVE = 0
    NATIVE_SIMPLE = 1
    NATIVE_LOW_DELAY = 2
    FAST = 3
    BLOCKING = 4


class FrameRing:
    """The decoder's frame ring + per-consumer hand-off (reference src/Decoder.cpp:97-131, 150-162)."""

    def __init__(self, depth):
        self.depth = max(1, int(depth))
        self.frames = [None] * self.depth
        self.current = 0  # number of frames published so far
        self.finished = False
        self.status = {}  # consumer name -> "a new frame is available"
        self.cv = threading.Condition()

    def publish(self, frame):
        with self.cv:
            self.frames[self.current % self.depth] = frame
            self.current += 1
            for k in self.status:
                self.status[k] = True
            self.cv.notify_all()

    def finish(self):
        with self.cv:
            for k in self.status:
                self.status[k] = True
            self.finished = True
            self.cv.notify_all()

    def get(self, name, index=0, timeout=None):
        """-> (frame, frame_number).  Blocks until a frame this consumer has not seen is published."""
        with self.cv:
            if name not in self.status:
                self.status[name] = self.current > 0  # a late joiner may take the latest frame at once
            deadline = None if timeout is None else time.monotonic() + timeout
            while not self.finished and not self.status[name]:
                if not self.cv.wait(None if deadline is None else max(0.0, deadline - time.monotonic())):
                    if deadline is not None and time.monotonic() >= deadline:
                        raise RuntimeError("Timeout waiting for a frame")
            if self.finished:
                raise RuntimeError("Decoding finished")
            self.status[name] = False
            index = min(int(index), 0)  # reference: positive delays are forced to 0
            aligned = (self.current - 1) % self.depth + index
            if aligned < 0 or self.frames[aligned] is None:
                return None, -1  # VREADER_REPEAT
            return self.frames[aligned], self.current

    def all_consumed(self):
        with self.cv:
            return all(not v for v in self.status.values()) and len(self.status) > 0


class TensorStreamConverter:
    def __init__(self, stream_url, max_consumers=5, cuda_device=None, buffer_size=5, framerate_mode=FrameRate.NATIVE, timeout=None):
        self.log = logging.getLogger(__name__)
        self.thread = None
        self.fps = None
        self.frame_size = None
        self.max_consumers = max_consumers
        self.cuda_device = torch.cuda.current_device() if (cuda_device is None and torch.cuda.is_available()) else (cuda_device or 0)
        self.buffer_size = buffer_size
        self.stream_url = stream_url
        self.framerate_mode = framerate_mode
        self._timeout = None
        self.set_timeout(timeout)
        self._source = None
        self._vpp = None
        self._ring = None
        self._stop = threading.Event()
        self._logs = (LogsLevel.NONE, LogsType.CONSOLE)
        self._markers = False

    # ---- lifecycle -------------------------------------------------------------------------------
    def initialize(self, repeat_number=1):
        last = None
        for _ in range(max(1, repeat_number)):
            try:
                self._source = open_source(self.stream_url)
                self._vpp = VideoProcessor(device=self.cuda_device, max_consumers=self.max_consumers)
                if self._markers:
                    self._vpp.enable_markers(True)
                self._ring = FrameRing(self.buffer_size)
                self._stop.clear()
                self.fps = self._source.fps_num / self._source.fps_den
                self.frame_size = (self._source.width, self._source.height)
                return
            except RuntimeError as e:
                last = e
                self.stop()
        raise RuntimeError(f"Can't initialize TensorStream: {last}")

    def enable_logs(self, level, log_type):
        self._logs = (level, log_type)

    def enable_nvtx(self):
        """Reference: NVTX ranges around the pipeline stages (tensor_stream/tensor_stream.py enable_nvtx,
        include/Common.h:72-105).  Here: roctx ranges around every conversion, visible to rocprofv3 --marker-trace.
        May be called before or after initialize(), like the reference's."""
        self._markers = True
        if self._vpp is not None:
            self._vpp.enable_markers(True)

    def set_timeout(self, timeout):
        self._timeout = None if timeout is None else float(timeout)

    def skip_analyze(self):
        pass  # bitstream analysis belongs to the H.264 parser, which the frame sources do not have

    def _start(self):
        src, ring = self._source, self._ring
        if src is None or ring is None:
            return
        period = src.fps_den / src.fps_num if src.fps_num else 0.0
        nxt = time.monotonic()
        dev = torch.device("cuda", self.cuda_device)
        while not self._stop.is_set():
            f = src.next_frame()
            if f is None:
                break
            y = torch.from_numpy(f[0]).to(dev, non_blocking=False)
            uv = torch.from_numpy(f[1]).to(dev, non_blocking=False)
            ring.publish((y, uv))
            if self.framerate_mode == FrameRate.BLOCKING:
                while not self._stop.is_set() and not ring.all_consumed():  # frame by frame, nobody skips
                    time.sleep(0.0005)
            elif self.framerate_mode != FrameRate.FAST and period > 0:
                nxt += period
                delay = nxt - time.monotonic()
                if delay > 0:
                    self._stop.wait(delay)
        ring.finish()

    def start(self):
        self.thread = threading.Thread(target=self._start, daemon=True)
        self.thread.start()

    def stop(self):
        self._stop.set()
        if self._ring is not None:
            self._ring.finish()
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self._source is not None:
            self._source.close()
            self._source = None
        if self._vpp is not None:
            self._vpp.Close()
            self._vpp = None

    # ---- reading ---------------------------------------------------------------------------------
    def read(self, name="default", width=0, height=0, resize_type=ResizeType.NEAREST, crop_coords=(0, 0, 0, 0), pixel_format=FourCC.RGB24,
             planes_pos=Planes.MERGED, normalization=None, delay=0, return_index=False):
        fp = FrameParameters(width=width, height=height, crop_coords=crop_coords, resize_type=resize_type, pixel_format=pixel_format,
                             planes_pos=planes_pos, normalization=normalization)
        return self.param_read(fp, name=name, delay=delay, return_index=return_index)

    def param_read(self, frame_parameters, name="default", delay=0, return_index=False):
        if self._ring is None or self._vpp is None:
            raise RuntimeError("-3")  # the reference throws std::to_string(VREADER_ERROR) when the pipeline is not up
        frame, index = None, -1
        while frame is None:  # VREADER_REPEAT loop of TensorStream::getFrame (src/Wrappers/WrapperPython.cpp:300-306)
            frame, index = self._ring.get(name, delay, self._timeout)
        y, uv = frame
        tensor = self._vpp.Convert(y, uv, frame_parameters, consumer=name)
        return (tensor, index) if return_index else tensor

    def dump(self, tensor, name="default", width=0, height=0, crop_coords=(0, 0, 0, 0), resize_type=ResizeType.NEAREST,
             pixel_format=FourCC.RGB24, planes_pos=Planes.MERGED, normalization=None):
        """Appends the tensor's raw elements to <name>.yuv (reference src/Wrappers/WrapperPython.cpp:421-456)."""
        torch.cuda.synchronize(tensor.device)  # conversions run asynchronously on the consumer's stream
        with open(name + ".yuv", "ab+") as f:
            f.write(tensor.contiguous().cpu().numpy().tobytes())
