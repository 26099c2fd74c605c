"""Decodes frame 0 (the IDR picture) of the reference's test clip tests/resources/bbb_1080x608_420_10.h264 with the intra
decoder of h264_intra.py and writes it as a tight NV12 fixture (tests/golden/bbb_1080x608_frame0.nv12, 984 960 bytes):
the input of the reference's 38 CRC goldens (tests/golden/reference_crcs.py).

Self-validation, all against literals the REFERENCE holds:
  * plane CRCs of the decoded frame: tests/src/DecoderTests.cpp:63-65 (Y 3265466497, UV 2183362287) -- H.264 decoding is
    bit-exact by specification, so a match means the frame is the one NVDEC produces;
  * the NEAREST-resized 320x240 frame recovered from tests/resources/test_references/NV12Normalization_320x240.yuv
    (tests/golden/ref_320x240.npz): 76 800 luma + 38 400 chroma samples of the decoded picture.
Runs only where /root/reference exists (the build container); the fixture travels.

    python tests/golden/make_bbb_frame0.py [path/to/bbb_1080x608_420_10.h264]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from h264_intra import IntraDecoder  # noqa: E402
from oracle import oracle as O  # noqa: E402  (av_crc restatement; this script is test infrastructure)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/resources/bbb_1080x608_420_10.h264"
    y, uv = IntraDecoder(open(src, "rb").read()).decode_first_idr()
    assert y.shape == (608, 1080) and uv.shape == (304, 1080)
    crc_y, crc_uv = O.av_crc32_ieee(y), O.av_crc32_ieee(uv)
    print("plane CRCs:", crc_y, crc_uv)
    assert (crc_y, crc_uv) == (3265466497, 2183362287), "decoded frame does not match reference tests/src/DecoderTests.cpp:63-65"
    g = np.load(os.path.join(HERE, "ref_320x240.npz"))["input_nv12_u8"]
    Y, UV = g[: 320 * 240].reshape(240, 320), g[320 * 240:].reshape(120, 320)
    xr, yr = np.float32(1080) / np.float32(320), np.float32(608) / np.float32(240)
    xs, ys = (xr * np.arange(320, dtype=np.float32)).astype(int), (yr * np.arange(240, dtype=np.float32)).astype(int)
    assert np.array_equal(y[np.ix_(ys, xs)], Y)
    cx, cy = xs[:160], ys[:120]
    assert np.array_equal(uv[np.ix_(cy, 2 * cx)], UV[:, 0::2]) and np.array_equal(uv[np.ix_(cy, 2 * cx + 1)], UV[:, 1::2])
    out = os.path.join(HERE, "bbb_1080x608_frame0.nv12")
    with open(out, "wb") as f:
        f.write(y.tobytes())
        f.write(uv.tobytes())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
