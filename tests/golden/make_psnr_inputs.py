"""Decodes the two JPEGs of the reference's PSNR tests (tests/resources/test_resize/{forest,tv_template}.jpg,
used by tests/src/VPPTests.cpp:673-911) into the NV12 frames its decode front-end hands to VideoProcessor, and
stores them as tests/golden/psnr_inputs.npz.  Run in the build container only (reads /root/reference):

    python tests/golden/make_psnr_inputs.py

The decoder below is a plain baseline-JPEG decoder (Huffman, dequantisation, exact float IDCT rounded to nearest):
the reference decodes with NVDEC, whose integer IDCT may differ by one LSB in a few samples, which moves the PSNR
figures of the tests by a few thousandths of a dB.  forest.jpg is 4:2:0, so its planes ARE the NV12 planes;
tv_template.jpg is 4:4:4 and NVDEC's chroma down-sampling is not documented -- a 2x2 box average is stored, and the
test built on it uses a wider tolerance.
"""
import os
import sys

import numpy as np
from scipy.fft import idctn

SRC = "/root/reference/tests/resources/test_resize"
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


class Bits:
    def __init__(self, data):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.p]
            self.p += 1
            if b == 0xFF:
                assert self.d[self.p] == 0x00, "marker inside a scan segment"
                self.p += 1
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v


def huff_table(counts, symbols):
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


def decode_symbol(br, table):
    code = 0
    for length in range(1, 17):
        code = (code << 1) | br.bit()
        s = table.get((length, code))
        if s is not None:
            return s
    raise ValueError("bad Huffman code")


def extend(v, t):
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def decode_jpeg(path):
    data = open(path, "rb").read()
    assert data[:2] == b"\xff\xd8"
    p, qt, ht, comps, restart = 2, {}, {}, None, 0
    while True:
        assert data[p] == 0xFF
        m = data[p + 1]
        p += 2
        if m == 0xD9:
            raise ValueError("no scan")
        n = (data[p] << 8) | data[p + 1]
        seg = data[p + 2:p + n]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                assert seg[q] >> 4 == 0, "16-bit quantisation tables not supported"
                qt[seg[q] & 15] = np.array(list(seg[q + 1:q + 65]), dtype=np.float64)
                q += 65
        elif m == 0xC0 or m == 0xC1:
            assert seg[0] == 8
            h, w, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(nc)]
        elif m == 0xC2:
            raise ValueError("progressive JPEG not supported")
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc_th, counts = seg[q], list(seg[q + 1:q + 17])
                ns = sum(counts)
                ht[tc_th] = huff_table(counts, list(seg[q + 17:q + 17 + ns]))
                q += 17 + ns
        elif m == 0xDD:
            restart = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            for c in range(ns):
                cid, t = seg[1 + 2 * c], seg[2 + 2 * c]
                for comp in comps:
                    if comp["id"] == cid:
                        comp["td"], comp["ta"] = t >> 4, t & 15
            p += n
            break
        p += n
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = (w + 8 * hmax - 1) // (8 * hmax), (h + 8 * vmax - 1) // (8 * vmax)
    for c in comps:
        c["plane"] = np.zeros((mcuy * c["v"] * 8, mcux * c["h"] * 8), dtype=np.float64)
        c["pred"] = 0
    # split the entropy-coded data at restart markers
    scan = data[p:]
    br = Bits(scan)
    count = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and count and count % restart == 0:
                # byte-align, expect RSTn
                br.n = 0
                assert scan[br.p] == 0xFF and 0xD0 <= scan[br.p + 1] <= 0xD7, "restart marker expected"
                br.p += 2
                for c in comps:
                    c["pred"] = 0
            count += 1
            for c in comps:
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = np.zeros(64, dtype=np.float64)
                        t = decode_symbol(br, ht[c["td"]])
                        c["pred"] += extend(br.bits(t), t) if t else 0
                        blk[0] = c["pred"]
                        k = 1
                        while k < 64:
                            rs = decode_symbol(br, ht[0x10 | c["ta"]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            blk[ZIGZAG[k]] = extend(br.bits(s), s)
                            k += 1
                        q = np.zeros(64, dtype=np.float64)
                        q[ZIGZAG] = qt[c["tq"]]
                        coef = (blk * q).reshape(8, 8)
                        pix = idctn(coef, norm="ortho") + 128.0
                        y0, x0 = (my * c["v"] + by) * 8, (mx * c["h"] + bx) * 8
                        c["plane"][y0:y0 + 8, x0:x0 + 8] = pix
    planes = []
    for c in comps:
        pw, ph = (w * c["h"] + hmax - 1) // hmax, (h * c["v"] + vmax - 1) // vmax
        planes.append(np.clip(np.floor(c["plane"][:ph, :pw] + 0.5), 0, 255).astype(np.uint8))
    return w, h, comps, planes


def to_nv12(w, h, comps, planes):
    y, cb, cr = planes
    if cb.shape != (h // 2, w // 2):  # 4:4:4 -> 4:2:0 by a 2x2 box average (NVDEC's filter is not documented)
        def half(c):
            c = c[: h // 2 * 2, : w // 2 * 2].astype(np.uint32)
            return ((c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) // 4).astype(np.uint8)
        cb, cr = half(cb), half(cr)
    uv = np.empty((h // 2, w // 2 * 2), dtype=np.uint8)
    uv[:, 0::2], uv[:, 1::2] = cb, cr
    return y[: h // 2 * 2, : w // 2 * 2].copy(), uv


def main():
    out = {}
    for name in ("forest", "tv_template"):
        w, h, comps, planes = decode_jpeg(os.path.join(SRC, name + ".jpg"))
        y, uv = to_nv12(w, h, comps, planes)
        print(name, w, h, [(c["h"], c["v"]) for c in comps], y.shape, uv.shape, int(y.mean()), int(uv.mean()))
        out[name + "_y"], out[name + "_uv"] = y, uv
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psnr_inputs.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    sys.exit(main())
