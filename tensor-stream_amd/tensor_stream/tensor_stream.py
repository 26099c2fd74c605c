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

Batching (round 3).  The reference converts one frame per read() per consumer -- 1-3 launches and up to five cudaMalloc each.
A single-frame launch of the fused kernel is launch-bound (~0.3 of the roofline), so the production entry
reaches the kernel rate only if conversions with identical FrameParameters travel together:
  * read_many(names, ...): ONE call serves many consumer names -- one hand-off under one lock, one tsvpp_convert_batch
    launch, one output allocation; every consumer gets its own tensor (a view of the batch tensor, which it keeps alive);
  * concurrent read() calls with identical parameters are coalesced by a leader / follower rendezvous (coalesce_window_us:
    the first caller waits that long for others, then launches for all of them) -- same results, fewer launches; Python
    threads stay bound by the interpreter lock (~10^4 reads/s), which read_many avoids;
  * the producer uploads through pinned staging buffers on a copy stream (frames carry an event that consumers wait on),
    and a synthetic source's frame pool is uploaded once.
"""
import ctypes
import logging
import threading
import time
from enum import Enum

import numpy as np
import torch

from . import _native as N

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
            self.cv.notify_all()  # (a FAST producer waits for the hand-off)
            index = min(int(index), 0)  # reference: positive delays are forced to 0
            aligned = (self.current - 1) % self.depth + index
            if aligned < 0 or self.frames[aligned] is None:
                return None, -1  # VREADER_REPEAT
            return self.frames[aligned], self.current

    def get_many(self, names, index=0, timeout=None):
        """get() for several consumers under ONE lock: blocks until every one of them has an unseen frame (they all see the
        same one: the hand-off flags are set together by publish) -> ([frame per name], frame_number)."""
        with self.cv:
            for name in names:
                if name not in self.status:
                    self.status[name] = self.current > 0
            deadline = None if timeout is None else time.monotonic() + timeout
            while not self.finished and not all(self.status[n] for n in names):
                if not self.cv.wait(None if deadline is None else max(0.0, deadline - time.monotonic())):
                    if deadline is not None and time.monotonic() >= deadline:
                        raise RuntimeError("Timeout waiting for a frame")
            if self.finished:
                raise RuntimeError("Decoding finished")
            for name in names:
                self.status[name] = False
            self.cv.notify_all()
            index = min(int(index), 0)
            aligned = (self.current - 1) % self.depth + index
            if aligned < 0 or self.frames[aligned] is None:
                return None, -1
            return [self.frames[aligned]] * len(names), self.current

    def get_batch(self, name, k, index=0, timeout=None):
        """The consumer's next hand-off as a TEMPORAL window: the frame get() would return and the k - 1 frames published before it, oldest first ->
        ([k frames], frame_number of the newest).  Frames are addressed by their absolute number (frame f lives in slot f mod depth): the window
        [newest - k + 1, newest] must still be in the ring (k - index <= depth) and already published, else (None, -1) = VREADER_REPEAT, as get() answers for a
        delay that reaches before the first frame."""
        k = int(k)
        if k < 1 or k - min(int(index), 0) > self.depth:
            raise ValueError(f"batch={k}, delay={index}: the ring holds {self.depth} frames")
        with self.cv:
            if name not in self.status:
                self.status[name] = self.current > 0
            deadline = None if timeout is None else time.monotonic() + timeout
            while not self.finished and not self.status[name]:
                if not self.cv.wait(None if deadline is None else max(0.0, deadline - time.monotonic())):
                    if deadline is not None and time.monotonic() >= deadline:
                        raise RuntimeError("Timeout waiting for a frame")
            if self.finished:
                raise RuntimeError("Decoding finished")
            self.status[name] = False
            self.cv.notify_all()
            index = min(int(index), 0)
            newest = self.current - 1 + index
            oldest = newest - k + 1
            if oldest < 0 or oldest <= self.current - 1 - self.depth:
                return None, -1
            frames = [self.frames[f % self.depth] for f in range(oldest, newest + 1)]
            if any(f is None for f in frames):
                return None, -1
            return frames, self.current

    def all_consumed(self):
        with self.cv:
            return all(not v for v in self.status.values()) and len(self.status) > 0


_NV12_DTYPE = np.dtype([("y", np.uint64), ("uv", np.uint64), ("pitch_y", np.int32), ("pitch_uv", np.int32), ("width", np.int32), ("height", np.int32)])
assert _NV12_DTYPE.itemsize == ctypes.sizeof(N.NV12)


class _Group:
    __slots__ = ("reqs", "closed", "done", "error", "out")

    def __init__(self):
        self.reqs = []      # (frame, consumer name) in arrival order
        self.closed = False
        self.done = threading.Event()
        self.error = None
        self.out = None


class _Coalescer:
    """Leader / follower rendezvous of concurrent conversions with identical parameters and frame geometry: the first caller
    (leader) waits up to `window` seconds -- or until `max_batch` requests have joined -- then issues ONE batched launch for the
    whole group and hands every caller its own slice."""

    def __init__(self, convert_group, window, max_batch=N.TSVPP_MAX_BATCH):
        self.convert_group = convert_group
        self.window = float(window)
        self.max_batch = int(max_batch)
        self.lock = threading.Lock()
        self.cv = threading.Condition(self.lock)
        self.open = {}
        self.launches = 0
        self.requests = 0

    def convert(self, key, frame, name, fp):
        """The caller's own result: convert_group's return value indexed by the caller's slot (a batch tensor, a list, or a _Converted: all subscriptable)."""
        out, slot = self.convert_slot(key, frame, name, fp)
        return out[slot]

    def convert_slot(self, key, frame, name, fp):
        """-> (what convert_group returned for the caller's group, the caller's index in it)."""
        with self.cv:
            grp = self.open.get(key)
            leader = grp is None
            if leader:
                grp = self.open[key] = _Group()
            slot = len(grp.reqs)
            grp.reqs.append((frame, name))
            self.requests += 1
            if leader:
                if len(grp.reqs) >= self.max_batch:  # max_batch == 1: the leader fills its own group -- nobody else would close it
                    grp.closed = True
                    del self.open[key]
                deadline = time.monotonic() + self.window
                while not grp.closed:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self.cv.wait(left)
                if not grp.closed:  # the window ran out: close under the lock, later arrivals open a new group
                    grp.closed = True
                    del self.open[key]
                self.launches += 1
            elif len(grp.reqs) >= self.max_batch:
                # full: the FOLLOWER that filled it closes it, under the lock -- the leader still has to re-acquire the lock, and until it
                # did later arrivals would keep appending to a group that is already at max_batch
                grp.closed = True
                del self.open[key]
                self.cv.notify_all()
        if leader:
            try:
                grp.out = self.convert_group([r[0] for r in grp.reqs], fp)
            except Exception as e:  # noqa: BLE001
                grp.error = e
            grp.done.set()
        else:
            grp.done.wait()
        if grp.error is not None:
            raise grp.error
        return grp.out, slot


class _Converted:
    """A group's batch tensor + the event recorded behind its launch on the leader's stream."""
    __slots__ = ("out", "event")

    def __getitem__(self, slot):  # (_Coalescer.convert: the production coalescer returns this object)
        return self.out[slot]

    def __init__(self, out, event):
        self.out, self.event = out, event


class TensorStreamConverter:
    def __init__(self, stream_url, max_consumers=5, cuda_device=None, buffer_size=5, framerate_mode=FrameRate.NATIVE, timeout=None,
                 coalesce_window_us=0):
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
        self.coalesce_window_us = float(coalesce_window_us)  # > 0: concurrent read() calls with identical parameters share a launch
        self._coalescer = None
        self._copy_stream = None
        self._desc_cache = {}

    # ---- lifecycle -------------------------------------------------------------------------------
    def initialize(self, repeat_number=1):
        last = None
        for _ in range(max(1, repeat_number)):
            try:
                self._source = open_source(self.stream_url)
                self._vpp = VideoProcessor(device=self.cuda_device, max_consumers=self.max_consumers)
                if self._markers:
                    self._enable_markers()
                self._ring = FrameRing(self.buffer_size)
                self._coalescer = _Coalescer(self._convert_group_event, self.coalesce_window_us * 1e-6) if self.coalesce_window_us > 0 else None
                self._desc_cache = {}
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
        include/Common.h:72-105), a harmless toggle.  Here: roctx ranges around every conversion, visible to
        rocprofv3 --marker-trace; on a machine without a roctx library the toggle stays harmless (logged, not raised).
        May be called before or after initialize(), like the reference's."""
        self._markers = True
        if self._vpp is not None:
            self._enable_markers()

    def _enable_markers(self):
        try:
            self._vpp.enable_markers(True)
        except RuntimeError as e:  # no roctx library: VideoProcessor.enable_markers raises, the facade's toggle does not
            self.log.warning("enable_nvtx: %s", e)

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
        # Uploads: pinned staging buffers, asynchronous copies on a copy stream, an event per frame that the consumers' streams
        # wait on (never the host).  Device frames are fresh allocations of the copy stream (consumers record their own streams on
        # them, so the allocator does not recycle a frame under a running conversion).  A source that cycles through a fixed pool
        # of arrays (synthetic) is uploaded once.
        copy_stream = torch.cuda.Stream(device=dev)
        resident = {}                       # id(host array pair) -> frame, for sources with a fixed pool
        nslots = 3                          # pinned staging pairs
        staging, staged_ev, k = [None] * nslots, [None] * nslots, 0
        fixed_pool = getattr(src, "pool", None) is not None
        while not self._stop.is_set():
            f = src.next_frame()
            if f is None:
                break
            key = (id(f[0]), id(f[1]))
            item = resident.get(key) if fixed_pool else None
            if item is None:
                if staging[k] is None:
                    staging[k] = (torch.empty(f[0].shape, dtype=torch.uint8).pin_memory(), torch.empty(f[1].shape, dtype=torch.uint8).pin_memory())
                elif staged_ev[k] is not None:
                    staged_ev[k].synchronize()  # the copy that last read this staging pair
                hy, huv = staging[k]
                hy.numpy()[...] = f[0]
                huv.numpy()[...] = f[1]
                ev = torch.cuda.Event()
                with torch.cuda.stream(copy_stream):
                    dy = torch.empty(f[0].shape, dtype=torch.uint8, device=dev)
                    duv = torch.empty(f[1].shape, dtype=torch.uint8, device=dev)
                    dy.copy_(hy, non_blocking=True)
                    duv.copy_(huv, non_blocking=True)
                    ev.record(copy_stream)
                staged_ev[k] = ev
                k = (k + 1) % nslots
                if fixed_pool:
                    ev.synchronize()
                    resident[key] = item = (dy, duv, None)
                else:
                    item = (dy, duv, ev)
            ring.publish(item)
            if self.framerate_mode == FrameRate.BLOCKING:
                while not self._stop.is_set() and not ring.all_consumed():  # frame by frame, nobody skips
                    time.sleep(0.0002)
            elif self.framerate_mode != FrameRate.FAST and period > 0:
                nxt += period
                delay = nxt - time.monotonic()
                if delay > 0:
                    self._stop.wait(delay)
            elif self.framerate_mode == FrameRate.FAST:
                # no pacing -- but a producer without real decoding work must neither spin on the interpreter lock the consumers need
                # nor ping-pong with them frame by frame (two thread wake-ups per frame): ~10^4 frames/s, ahead of any consumer
                time.sleep(5e-5)
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
             planes_pos=Planes.MERGED, normalization=None, delay=0, return_index=False, batch=None):
        """The reference's read() (tensor_stream/tensor_stream.py:248-273) plus `batch` (round 6): batch=k (1 <= k <= buffer_size) returns the consumer's next
        frame TOGETHER with the k - 1 frames before it -- the decoder ring already holds them (reference include/Decoder.h:19, tensor_stream.py:165) -- as one
        (k, ...) tensor, oldest first, converted by ONE tsvpp_convert_batch launch: a clip for a temporal model, at 0.6-0.7 of the HBM roofline for k = 4..8
        where k single-frame reads run at 0.27 (profiles/r06_curve_*.txt).  return_index gives the index of the newest frame."""
        fp = FrameParameters(width=width, height=height, crop_coords=crop_coords, resize_type=resize_type, pixel_format=pixel_format,
                             planes_pos=planes_pos, normalization=normalization)
        return self.param_read(fp, name=name, delay=delay, return_index=return_index, batch=batch)

    def param_read(self, frame_parameters, name="default", delay=0, return_index=False, batch=None):
        if self._ring is None or self._vpp is None:
            raise RuntimeError("-3")  # the reference throws std::to_string(VREADER_ERROR) when the pipeline is not up
        if batch is not None:
            if not 1 <= int(batch) <= self.buffer_size:
                raise RuntimeError(f"-3: batch={batch} exceeds the decoder ring (buffer_size={self.buffer_size})")
            self._vpp.consumer_stream(name)  # claims the consumer's pool slot, as every read does
            frames, index = None, -1
            while frames is None:  # VREADER_REPEAT until the ring holds k frames
                frames, index = self._ring.get_batch(name, int(batch), delay, self._timeout)
            tensor = self._convert_group(frames, frame_parameters)
            return (tensor, index) if return_index else tensor
        frame, index = None, -1
        while frame is None:  # VREADER_REPEAT loop of TensorStream::getFrame (src/Wrappers/WrapperPython.cpp:300-306)
            frame, index = self._ring.get(name, delay, self._timeout)
        y, uv, ev = frame
        if self._coalescer is not None:
            self._vpp.consumer_stream(name)  # claims the consumer's pool slot (reference: a 6th name on a pool of 5 is an error)
            p = frame_parameters.parameters
            key = (bytes(p), tuple(y.shape), y.stride(0), uv.stride(0))
            grp, slot = self._coalescer.convert_slot(key, frame, name, frame_parameters)
            # The group was converted on the LEADER thread's current stream.  This caller may run under another stream
            # (`with torch.cuda.stream(s)`): order its stream behind the launch, and tell the allocator that the batch tensor is in
            # use on it, exactly the contract of the non-coalesced path (Convert(consumer=...) ends with cur.wait_stream) -- ADVICE r03.
            cur = torch.cuda.current_stream(self.cuda_device)
            cur.wait_event(grp.event)
            tensor = grp.out[slot]
            tensor.record_stream(cur)
        else:
            if ev is not None:
                torch.cuda.current_stream(self.cuda_device).wait_event(ev)  # (the consumer's pooled stream waits for the current one)
            tensor = self._vpp.Convert(y, uv, frame_parameters, consumer=name)
            if ev is not None:
                # The frame's memory must outlive the conversion enqueued on the consumer's pooled stream.  Convert() has made torch's
                # current stream wait for that stream, so recording the CURRENT stream (torch-owned, never destroyed -- the pooled one
                # dies with the context, before the allocator may want to query it) on the frame orders its reuse after the conversion.
                cur = torch.cuda.current_stream(self.cuda_device)
                y.record_stream(cur)
                uv.record_stream(cur)
        return (tensor, index) if return_index else tensor

    def _convert_group(self, frames, fp):
        """ONE batched launch for `frames` (list of (y, uv, event)) with parameters `fp` on torch's current stream -> batch tensor
        (n, ...): slice i belongs to request i and keeps the batch alive."""
        vpp, n = self._vpp, len(frames)
        cur = torch.cuda.current_stream(self.cuda_device)
        waited = set()
        for f in frames:
            if f[2] is not None and id(f[2]) not in waited:
                cur.wait_event(f[2])
                waited.add(id(f[2]))
                f[0].record_stream(cur)  # the frame's memory must outlive the conversion enqueued below
                f[1].record_stream(cur)
        y0, uv0 = frames[0][0], frames[0][1]
        w, h = y0.shape[1], y0.shape[0]
        p = fp.parameters
        out = vpp._alloc(p, w, h, n)
        # descriptor arrays without a Python loop over struct fields: numpy records with the layout of tsvpp_nv12
        rec = np.empty(n, dtype=_NV12_DTYPE)
        rec["y"] = [f[0].data_ptr() for f in frames]
        rec["uv"] = [f[1].data_ptr() for f in frames]
        rec["pitch_y"], rec["pitch_uv"], rec["width"], rec["height"] = y0.stride(0), uv0.stride(0), w, h
        outs = np.arange(n, dtype=np.uint64) * np.uint64(out.stride(0) * out.element_size()) + np.uint64(out.data_ptr())
        N.check(vpp._lib.tsvpp_convert_batch(vpp._ctx, n, rec.ctypes.data_as(ctypes.POINTER(N.NV12)), ctypes.byref(p),
                                             outs.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)), cur.cuda_stream))
        return out

    def _convert_group_event(self, frames, fp):
        """_convert_group + an event behind the launch: what the coalescer's followers order their own streams after."""
        out = self._convert_group(frames, fp)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.cuda_device))
        return _Converted(out, ev)

    def read_many(self, names, width=0, height=0, resize_type=ResizeType.NEAREST, crop_coords=(0, 0, 0, 0), pixel_format=FourCC.RGB24,
                  planes_pos=Planes.MERGED, normalization=None, delay=0, return_index=False):
        """read() for several consumer names at once: every consumer takes its next frame (the hand-off semantics of read(), under
        one lock), ONE batched launch converts them all, and the result is a list of tensors -- one per name, each a view of one
        batch tensor that it keeps alive.  This is how the facade reaches the kernel's rate: a launch per frame is launch-bound."""
        if self._ring is None or self._vpp is None:
            raise RuntimeError("-3")
        names = list(names)
        if len(names) > self.max_consumers:
            raise RuntimeError("-3")  # more consumers than the pool holds (reference src/VideoProcessor.cpp:100-103)
        fp = FrameParameters(width=width, height=height, crop_coords=crop_coords, resize_type=resize_type, pixel_format=pixel_format,
                             planes_pos=planes_pos, normalization=normalization)
        frames, index = None, -1
        while frames is None:
            frames, index = self._ring.get_many(names, delay, self._timeout)
        out = self._convert_group(frames, fp)
        tensors = [out[i] for i in range(len(names))]
        return (tensors, index) if return_index else tensors

    def dump(self, tensor, name="default", width=0, height=0, crop_coords=(0, 0, 0, 0), resize_type=ResizeType.NEAREST,
             pixel_format=FourCC.RGB24, planes_pos=Planes.MERGED, normalization=None):
        """Appends the tensor's raw elements to <name>.yuv (reference src/Wrappers/WrapperPython.cpp:421-456).  Like the reference, a width or
        height of 0 is filled in from the tensor ((H, W, C) for the three-channel formats, (1, H * channels, W) otherwise) and the element
        count written is that of width x height x channels; unlike the reference (which would read past a smaller tensor) a tensor that does
        not hold that many elements is an error."""
        fp = FrameParameters(width=width, height=height, crop_coords=crop_coords, resize_type=resize_type, pixel_format=pixel_format,
                             planes_pos=planes_pos, normalization=normalization)
        ch = N.lib().tsvpp_channels(fp.parameters.fourcc)
        three = ch == 3
        # (3, H, W) of planar RGB24 / BGR24: the reference reads size(1) / size(0) here as well and dumps 3 x H elements -- not copied
        planar3 = three and fp.parameters.planes == Planes.PLANAR.value and fp.parameters.fourcc in (FourCC.RGB24.value, FourCC.BGR24.value)
        if not width:
            width = tensor.shape[2] if (planar3 or not three) else tensor.shape[1]
        if not height:
            height = tensor.shape[1] if planar3 else (tensor.shape[0] if three else int(tensor.shape[1] / ch))
        if tensor.numel() != int(int(width) * int(height) * ch):
            raise RuntimeError(f"-3: tensor of {tensor.numel()} elements does not match the dump parameters {width}x{height}x{ch}")
        torch.cuda.synchronize(tensor.device)  # conversions run asynchronously on the consumer's stream
        with open(name + ".yuv", "ab+") as f:
            f.write(tensor.contiguous().cpu().numpy().tobytes())
