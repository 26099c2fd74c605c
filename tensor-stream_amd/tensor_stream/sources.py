"""Frame sources for the TensorStreamConverter facade.

The reference demuxes and decodes H.264 with FFmpeg + NVDEC (src/Parser.cpp, src/Decoder.cpp); neither FFmpeg
nor a VA-API/VCN decode stack exists in this image, so the facade is fed by sources that produce what the decoder
would leave in device memory: NV12 frames.

    synthetic://1920x1080?seed=0&frames=100&fps=25&pool=8     deterministic full-range random NV12
    /path/clip.nv12?w=1920&h=1080&fps=30                      raw NV12 file (Y plane, then interleaved UV, per frame)

Anything else (".h264", ".mp4", "rtmp://...") raises RuntimeError at initialize(), exactly like the reference's
"Can't initialize TensorStream" (tensor_stream/tensor_stream.py:187-202).
"""
import os
import urllib.parse

import numpy as np


class FrameSource:
    width = 0
    height = 0
    fps_num = 25
    fps_den = 1

    def next_frame(self):
        """-> (y, uv) uint8 arrays of shape (H, W) and (H/2, W), or None at end of stream."""
        raise NotImplementedError

    def close(self):
        pass


class SyntheticSource(FrameSource):
    def __init__(self, width, height, seed=0, frames=0, fps=25, pool=8):
        if width <= 0 or height <= 0 or (width | height) & 1:
            raise RuntimeError("synthetic source needs positive even dimensions")
        self.width, self.height = width, height
        self.fps_num, self.fps_den = int(fps), 1
        self.total = int(frames)  # 0 = endless
        self.index = 0
        rng = np.random.default_rng(int(seed))
        self.pool = [(rng.integers(0, 256, (height, width), dtype=np.uint8),
                      rng.integers(0, 256, (height // 2, width), dtype=np.uint8)) for _ in range(max(1, int(pool)))]

    def next_frame(self):
        if self.total and self.index >= self.total:
            return None
        f = self.pool[self.index % len(self.pool)]
        self.index += 1
        return f


class RawNV12Source(FrameSource):
    def __init__(self, path, width, height, fps=25, loop=False):
        if not os.path.isfile(path):
            raise RuntimeError(f"no such file: {path}")
        if width <= 0 or height <= 0 or (width | height) & 1:
            raise RuntimeError("raw NV12 source needs ?w=&h= with even values")
        self.width, self.height = width, height
        self.fps_num, self.fps_den = int(fps), 1
        self.f = open(path, "rb")
        self.frame_bytes = width * height * 3 // 2
        self.loop = bool(loop)

    def next_frame(self):
        buf = self.f.read(self.frame_bytes)
        if len(buf) < self.frame_bytes:
            if not self.loop:
                return None
            self.f.seek(0)
            buf = self.f.read(self.frame_bytes)
            if len(buf) < self.frame_bytes:
                return None
        a = np.frombuffer(buf, dtype=np.uint8)
        n = self.width * self.height
        return a[:n].reshape(self.height, self.width), a[n:].reshape(self.height // 2, self.width)

    def close(self):
        self.f.close()


def open_source(url):
    """Parse a stream URL into a FrameSource; RuntimeError for anything that would need a real demuxer/decoder."""
    if url.startswith("synthetic://"):
        rest = url[len("synthetic://"):]
        size, _, query = rest.partition("?")
        q = {k: v[0] for k, v in urllib.parse.parse_qs(query).items()}
        try:
            w, h = (int(x) for x in size.lower().split("x"))
        except ValueError:
            raise RuntimeError(f"bad synthetic size in {url!r}")
        return SyntheticSource(w, h, seed=int(q.get("seed", 0)), frames=int(q.get("frames", 0)), fps=float(q.get("fps", 25)),
                               pool=int(q.get("pool", 8)))
    path, _, query = url.partition("?")
    if path.lower().endswith((".nv12", ".yuv")):
        q = {k: v[0] for k, v in urllib.parse.parse_qs(query).items()}
        return RawNV12Source(path, int(q.get("w", 0)), int(q.get("h", 0)), fps=float(q.get("fps", 25)), loop=q.get("loop", "0") == "1")
    raise RuntimeError(f"cannot open {url!r}: this build has no demuxer/decoder (FFmpeg + VA-API/VCN are not in the image); "
                       "use synthetic://WxH or a raw .nv12 file")
