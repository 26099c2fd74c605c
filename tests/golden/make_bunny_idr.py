"""BASELINE.json configs[0] (C1) names the reference's own demo clip, tests/resources/bunny.mp4 (1280x720 H.264 Main profile, CABAC, 241 frames) decoded to NV12.
No FFmpeg exists in this image, so this script pulls ONE picture out of it with what the repository has: a minimal ISO-BMFF (MP4) box walk -- stsz / stco /
stsc / stss of the video track, avcC for the parameter sets -- and the intra decoder of h264_intra.py (test infrastructure, validated bit for bit on the
reference's decoder-test CRCs of its other clip, tests/golden/make_bbb_frame0.py).  The clip has two IDR pictures: sample 1 is a 209-byte blank, sample 129
(51 242 bytes, 5.4 s in) is a full picture of the film -- that one is decoded and written as tests/golden/bunny_idr129_1280x720.npz (Y 720x1280, UV 360x1280).

What this fixture is and is not: real 1280x720 content of the reference's own clip for the parity tests and the C1 leg of bench.py (NV12 -> RGB24 MERGED uint8
at native size).  The reference holds NO literal for this picture, so nothing here pins the DECODE against the reference (H.264 decoding is bit-exact by
specification and the decoder is pinned on the other clip); the conversion is checked against the oracle as everywhere else.
Runs only where /root/reference exists (the build container); the fixture travels.

    python tests/golden/make_bunny_idr.py [path/to/bunny.mp4]
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from h264_intra import IntraDecoder  # noqa: E402


def find(b, path, off=0, end=None):
    """(payload start, box end) of the first box at `path` (a list of 4-byte types), depth first."""
    end = len(b) if end is None else end
    while off + 8 <= end:
        sz, tp = struct.unpack(">I4s", b[off:off + 8])
        hdr = 8
        if sz == 1:
            sz, hdr = struct.unpack(">Q", b[off + 8:off + 16])[0], 16
        if sz == 0:
            sz = end - off
        if tp == path[0]:
            if len(path) == 1:
                return off + hdr, off + sz
            r = find(b, path[1:], off + hdr, off + sz)
            if r:
                return r
        off += sz
    return None


def video_samples(f):
    """-> (SPS list, PPS list, NAL length size, [(offset, size)] of every sample, sync sample numbers) of the first track (the video track of this clip)."""
    stbl = find(f, [b"moov", b"trak", b"mdia", b"minf", b"stbl"])
    box = lambda name: find(f, [name], stbl[0], stbl[1])  # noqa: E731
    s, _ = box(b"stsd")
    p = s + 8
    esz, fmt = struct.unpack(">I4s", f[p:p + 8])
    assert fmt == b"avc1"
    q, sps, pps, nal_len = p + 8 + 78, [], [], 4
    while q < p + esz:
        s2, t2 = struct.unpack(">I4s", f[q:q + 8])
        if t2 == b"avcC":
            a = f[q + 8:q + s2]
            nal_len = (a[4] & 3) + 1
            o = 6
            for _ in range(a[5] & 31):
                n = struct.unpack(">H", a[o:o + 2])[0]
                sps.append(a[o + 2:o + 2 + n])
                o += 2 + n
            npps = a[o]
            o += 1
            for _ in range(npps):
                n = struct.unpack(">H", a[o:o + 2])[0]
                pps.append(a[o + 2:o + 2 + n])
                o += 2 + n
        q += s2
    s, _ = box(b"stsz")
    _, fixed, cnt = struct.unpack(">III", f[s:s + 12])
    sizes = [fixed] * cnt if fixed else list(struct.unpack(">%dI" % cnt, f[s + 12:s + 12 + 4 * cnt]))
    s, _ = box(b"stco")
    n = struct.unpack(">I", f[s + 4:s + 8])[0]
    chunks = struct.unpack(">%dI" % n, f[s + 8:s + 8 + 4 * n])
    s, _ = box(b"stsc")
    nc = struct.unpack(">I", f[s + 4:s + 8])[0]
    stsc = [struct.unpack(">III", f[s + 8 + 12 * i:s + 20 + 12 * i]) for i in range(nc)]
    s, _ = box(b"stss")
    ns = struct.unpack(">I", f[s + 4:s + 8])[0]
    sync = list(struct.unpack(">%dI" % ns, f[s + 8:s + 8 + 4 * ns]))
    samples, si = [], 0
    for ci in range(n):
        per_chunk = [x for x in stsc if x[0] <= ci + 1][-1][1]
        o = chunks[ci]
        for _ in range(per_chunk):
            if si >= cnt:
                break
            samples.append((o, sizes[si]))
            o += sizes[si]
            si += 1
    return sps, pps, nal_len, samples, sync


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/resources/bunny.mp4"
    f = open(src, "rb").read()
    sps, pps, nal_len, samples, sync = video_samples(f)
    assert len(samples) == 241 and sync == [1, 129], (len(samples), sync)
    off, size = samples[sync[-1] - 1]
    annexb = b"".join(b"\x00\x00\x01" + x for x in sps + pps)
    p = off
    while p < off + size:  # length-prefixed NAL units -> Annex B
        n = int.from_bytes(f[p:p + nal_len], "big")
        annexb += b"\x00\x00\x01" + f[p + nal_len:p + nal_len + n]
        p += nal_len + n
    y, uv = IntraDecoder(annexb).decode_first_idr()
    assert y.shape == (720, 1280) and uv.shape == (360, 1280)
    assert 8 < y.std() and y.min() >= 0  # a picture, not the blank first IDR
    out = os.path.join(HERE, "bunny_idr129_1280x720.npz")
    np.savez_compressed(out, y=y, uv=uv)
    print("wrote", out, os.path.getsize(out), "bytes; luma mean %.2f std %.2f" % (y.mean(), y.std()))


if __name__ == "__main__":
    main()
