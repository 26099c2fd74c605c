"""Minimal H.264 intra-picture decoder -- TEST INFRASTRUCTURE, written from the standard (ITU-T H.264, clauses 7-9).

Purpose: the reference's only byte-exact goldens for the interpolating resize kernels are 38 CRC-32 literals of outputs
computed from frame 0 of its own test clip `tests/resources/bbb_1080x608_420_10.h264` (reference tests/src/VPPTests.cpp:134-299,
tests/src/PythonTests.cpp:183-244).  No H.264 decoder exists in this image (no FFmpeg, no VCN user space), and H.264 decoding
is bit-exact by specification -- so this file decodes that one IDR picture: High profile, CABAC, 4:2:0 8-bit, frame
macroblocks only, one slice, flat scaling lists, Intra 4x4 / 8x8 (transform_size_8x8_flag) / 16x16 / I_PCM, in-loop
deblocking.  It validates itself through the reference's own literals: the plane CRCs of the decoded frame
(tests/src/DecoderTests.cpp:63-65: Y 3265466497, UV 2183362287) and the NEAREST-resized 320x240 frame recoverable from
tests/resources/test_references/NV12Normalization_320x240.yuv.  Only tests/golden/make_bbb_frame0.py runs it (in the build
container, where /root/reference exists); the decoded frame travels as a fixture.

Not supported (not needed for that picture): inter prediction, CAVLC, fields / MBAFF, multiple slices, FMO, scaling matrices,
4:2:2 / 4:4:4, bit depths above 8.
"""
import numpy as np

# ---------------------------------------------------------------------------------------------- bit reader

def nal_units(data):
    """Annex B byte stream -> list of NAL unit payloads (header byte included), emulation prevention removed."""
    out, i, n = [], 0, len(data)
    starts = []
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = (starts[k + 1] - 3) if k + 1 < len(starts) else n
        raw = data[s:e]
        while len(raw) and raw[-1] == 0:
            raw = raw[:-1]
        rb = bytearray()
        z = 0
        for b in raw:
            if z >= 2 and b == 3:
                z = 0
                continue
            rb.append(b)
            z = z + 1 if b == 0 else 0
        out.append(bytes(rb))
    return out


class BitReader:
    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def u(self, n):
        v = 0
        for _ in range(n):
            byte = self.b[self.p >> 3] if (self.p >> 3) < len(self.b) else 0
            v = (v << 1) | ((byte >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


# ---------------------------------------------------------------------------------------------- parameter sets

def parse_sps(rbsp):
    r = BitReader(rbsp, 8)
    s = {"profile": r.u(8)}
    r.u(8)
    s["level"] = r.u(8)
    r.ue()
    s["chroma_format_idc"] = 1
    if s["profile"] in (100, 110, 122, 244, 44, 83, 86, 118, 128):
        s["chroma_format_idc"] = r.ue()
        assert s["chroma_format_idc"] == 1, "only 4:2:0"
        assert r.ue() == 0 and r.ue() == 0, "only 8-bit"
        r.u(1)
        assert r.u(1) == 0, "scaling matrices not supported"
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    if s["poc_type"] == 0:
        s["log2_max_poc_lsb"] = r.ue() + 4
    elif s["poc_type"] == 1:
        raise NotImplementedError("poc type 1")
    r.ue()
    r.u(1)
    s["mb_w"] = r.ue() + 1
    s["mb_h"] = r.ue() + 1
    assert r.u(1) == 1, "frame_mbs_only"
    r.u(1)
    s["crop"] = (0, 0, 0, 0)
    if r.u(1):
        s["crop"] = (r.ue(), r.ue(), r.ue(), r.ue())  # left right top bottom, in chroma sample units (x2 for luma)
    return s


def parse_pps(rbsp):
    r = BitReader(rbsp, 8)
    p = {}
    r.ue()
    r.ue()
    p["cabac"] = r.u(1)
    p["bottom_field_pic_order"] = r.u(1)
    assert r.ue() == 0, "slice groups"
    r.ue()
    r.ue()
    r.u(1)
    r.u(2)
    p["pic_init_qp"] = r.se() + 26
    r.se()
    p["chroma_qp_index_offset"] = r.se()
    p["deblocking_filter_control_present"] = r.u(1)
    p["constrained_intra_pred"] = r.u(1)
    p["redundant_pic_cnt_present"] = r.u(1)
    p["transform_8x8_mode"] = 0
    p["second_chroma_qp_index_offset"] = p["chroma_qp_index_offset"]
    # more_rbsp_data(): anything before the trailing stop bit
    total = len(rbsp) * 8
    last = rbsp[-1]
    tz = 0
    while not (last >> tz) & 1:
        tz += 1
    if r.p < total - tz - 1:
        p["transform_8x8_mode"] = r.u(1)
        assert r.u(1) == 0, "pic scaling matrix"
        p["second_chroma_qp_index_offset"] = r.se()
    return p


# ---------------------------------------------------------------------------------------------- CABAC (9.3)

RANGE_LPS = [
    (128, 176, 208, 240), (128, 167, 197, 227), (128, 158, 187, 216), (123, 150, 178, 205), (116, 142, 169, 195), (111, 135, 160, 185),
    (105, 128, 152, 175), (100, 122, 144, 166), (95, 116, 137, 158), (90, 110, 130, 150), (85, 104, 123, 142), (81, 99, 117, 135),
    (77, 94, 111, 128), (73, 89, 105, 122), (69, 85, 100, 116), (66, 80, 95, 110), (62, 76, 90, 104), (59, 72, 86, 99),
    (56, 69, 81, 94), (53, 65, 77, 89), (51, 62, 73, 85), (48, 59, 69, 80), (46, 56, 66, 76), (43, 53, 63, 72),
    (41, 50, 59, 69), (39, 48, 56, 65), (37, 45, 54, 62), (35, 43, 51, 59), (33, 41, 48, 56), (32, 39, 46, 53),
    (30, 37, 43, 50), (29, 35, 41, 48), (27, 33, 39, 45), (26, 31, 37, 43), (24, 30, 35, 41), (23, 28, 33, 39),
    (22, 27, 32, 37), (21, 26, 30, 35), (20, 24, 29, 33), (19, 23, 27, 31), (18, 22, 26, 30), (17, 21, 25, 28),
    (16, 20, 23, 27), (15, 19, 22, 25), (14, 18, 21, 24), (14, 17, 20, 23), (13, 16, 19, 22), (12, 15, 18, 21),
    (12, 14, 17, 20), (11, 14, 16, 19), (11, 13, 15, 18), (10, 12, 15, 17), (10, 12, 14, 16), (9, 11, 13, 15),
    (9, 11, 12, 14), (8, 10, 12, 14), (8, 9, 11, 13), (7, 9, 11, 12), (7, 9, 10, 12), (7, 8, 10, 11),
    (6, 8, 9, 11), (6, 7, 9, 10), (6, 7, 8, 9), (2, 2, 2, 2)]
TRANS_LPS = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27,
             28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63]

# (m, n) of the context variables an I slice uses (Tables 9-12 .. 9-23, columns for I slices); ctxIdx -> (m, n)
CTX_I = {}


def _put(first, pairs):
    for k, mn in enumerate(pairs):
        CTX_I[first + k] = mn


_put(0, [(20, -15), (2, 54), (3, 74), (20, -15), (2, 54), (3, 74), (-28, 127), (-23, 104), (-6, 53), (-1, 54), (7, 51)])
_put(60, [(0, 41), (0, 63), (0, 63), (0, 63), (-9, 83), (4, 86), (0, 97), (-7, 72), (13, 41), (3, 62)])
_put(70, [(0, 11), (1, 55), (0, 69), (-17, 127), (-13, 102), (0, 82), (-7, 74), (-21, 107), (-27, 127), (-31, 127), (-24, 127), (-18, 95),
          (-27, 127), (-21, 114), (-30, 127), (-17, 123), (-12, 115), (-16, 122)])
_put(88, [(-11, 115), (-12, 63), (-2, 68), (-15, 84), (-13, 104), (-3, 70), (-8, 93), (-10, 90), (-30, 127), (-1, 74), (-6, 97), (-7, 91),
          (-20, 127), (-4, 56), (-5, 82), (-7, 76), (-22, 125)])
_put(105, [(-7, 93), (-11, 87), (-3, 77), (-5, 71), (-4, 63), (-4, 68), (-12, 84), (-7, 62), (-7, 65), (8, 61), (5, 56), (-2, 66),
           (1, 64), (0, 61), (-2, 78), (1, 50), (7, 52), (10, 35), (0, 44), (11, 38), (1, 45), (0, 46), (5, 44), (31, 17),
           (1, 51), (7, 50), (28, 19), (16, 33), (14, 62), (-13, 108), (-15, 100)])
_put(136, [(-13, 101), (-13, 91), (-12, 94), (-10, 88), (-16, 84), (-10, 86), (-7, 83), (-13, 87), (-19, 94), (1, 70), (0, 72), (-5, 74),
           (18, 59), (-8, 102), (-15, 100), (0, 95), (-4, 75), (2, 72), (-11, 75), (-3, 71), (15, 46), (-13, 69), (0, 62), (0, 65),
           (21, 37), (-15, 72), (9, 57), (16, 54), (0, 62), (12, 72)])
_put(166, [(24, 0), (15, 9), (8, 25), (13, 18), (15, 9), (13, 19), (10, 37), (12, 18), (6, 29), (20, 33), (15, 30), (4, 45),
           (1, 58), (0, 62), (7, 61), (12, 38), (11, 45), (15, 39), (11, 42), (13, 44), (16, 45), (12, 41), (10, 49), (30, 34),
           (18, 42), (10, 55), (17, 51), (17, 46), (0, 89), (26, -19), (22, -17)])
_put(197, [(26, -17), (30, -25), (28, -20), (33, -23), (37, -27), (33, -23), (40, -28), (38, -17), (33, -11), (40, -15), (41, -6), (38, 1),
           (41, 17), (30, -6), (27, 3), (26, 22), (37, -16), (35, -4), (38, -8), (38, -3), (37, 3), (38, 5), (42, 0), (35, 16),
           (39, 22), (14, 48), (27, 37), (21, 60), (12, 68), (2, 97)])
_put(227, [(-3, 71), (-6, 42), (-5, 50), (-3, 54), (-2, 62), (0, 58), (1, 63), (-2, 72), (-1, 74), (-9, 91), (-5, 67), (-5, 27),
           (-3, 39), (-2, 44), (0, 46), (-16, 64), (-8, 68), (-10, 78), (-6, 77), (-10, 86), (-12, 92), (-15, 55), (-10, 60), (-6, 62),
           (-4, 65)])
_put(252, [(-12, 73), (-8, 76), (-7, 80), (-9, 88), (-17, 110), (-11, 97), (-20, 84), (-11, 79), (-6, 73), (-4, 74), (-13, 86), (-13, 96),
           (-11, 97), (-19, 117), (-8, 78), (-5, 33), (-4, 48), (-2, 53), (-3, 62), (-13, 71), (-10, 79), (-12, 86), (-13, 90), (-14, 97)])
_put(399, [(31, 21), (31, 31), (25, 50),
           (-17, 120), (-20, 112), (-18, 114), (-11, 85), (-15, 92), (-14, 89), (-26, 71), (-15, 81), (-14, 80), (0, 68), (-14, 70), (-24, 56),
           (-23, 68), (-24, 50), (-11, 74),
           (23, -13), (26, -13), (40, -15), (49, -14), (44, 3), (45, 6), (44, 34), (33, 54), (19, 82),
           (-3, 75), (-1, 23), (1, 34), (1, 43), (0, 54), (-2, 55), (0, 61), (1, 64), (0, 68), (-9, 92)])

# ctxIdxInc of significant_coeff_flag / last_significant_coeff_flag for 8x8 blocks, frame coding (Table 9-43)
SIG8 = [0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11,
        12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12]
LAST8 = [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,
         5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8]
CBF_OFF = 85
SIG_OFF = {0: 105, 1: 120, 2: 134, 3: 149, 4: 152, 5: 402}
LAST_OFF = {0: 166, 1: 181, 2: 195, 3: 210, 4: 213, 5: 417}
ABS_OFF = {0: 227, 1: 237, 2: 247, 3: 257, 4: 266, 5: 426}


class Cabac:
    def __init__(self, buf, bitpos, slice_qp):
        assert bitpos % 8 == 0
        self.r = BitReader(buf, bitpos)
        self.range = 510
        self.offset = self.r.u(9)
        self.state = {}
        q = min(max(slice_qp, 0), 51)
        for idx, (m, n) in CTX_I.items():
            pre = min(max(((m * q) >> 4) + n, 1), 126)
            self.state[idx] = [63 - pre, 0] if pre <= 63 else [pre - 64, 1]
        self.bins = 0

    def decision(self, ctx):
        st = self.state[ctx]
        lps = RANGE_LPS[st[0]][(self.range >> 6) & 3]
        self.range -= lps
        if self.offset >= self.range:
            b = 1 - st[1]
            self.offset -= self.range
            self.range = lps
            if st[0] == 0:
                st[1] = 1 - st[1]
            st[0] = TRANS_LPS[st[0]]
        else:
            b = st[1]
            if st[0] < 62:
                st[0] += 1
        while self.range < 256:
            self.range <<= 1
            self.offset = (self.offset << 1) | self.r.u(1)
        self.bins += 1
        return b

    def bypass(self):
        self.offset = (self.offset << 1) | self.r.u(1)
        if self.offset >= self.range:
            self.offset -= self.range
            return 1
        return 0

    def terminate(self):
        self.range -= 2
        if self.offset >= self.range:
            return 1
        while self.range < 256:
            self.range <<= 1
            self.offset = (self.offset << 1) | self.r.u(1)
        return 0


# ---------------------------------------------------------------------------------------------- tables for reconstruction

ZIGZAG4 = [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1), (2, 0), (3, 0), (2, 1), (1, 2), (0, 3), (1, 3), (2, 2), (3, 1), (3, 2), (2, 3), (3, 3)]  # (x, y)
ZIGZAG8 = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
           57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]  # y * 8 + x
NORM4 = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]
NORM8 = [(20, 18, 32, 19, 25, 24), (22, 19, 35, 21, 28, 26), (26, 23, 42, 24, 33, 31), (28, 25, 45, 26, 35, 33), (32, 28, 51, 30, 40, 38),
         (36, 32, 58, 34, 46, 43)]
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226,
                    255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 0, 1), (0, 1, 1), (0, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 2), (1, 1, 2),
                          (1, 1, 2), (1, 1, 2), (1, 2, 3), (1, 2, 3), (2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4), (3, 3, 5), (3, 4, 6), (3, 4, 6),
                          (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18), (10, 13, 20),
                          (11, 15, 23), (13, 17, 25)]


def level_scale4(qp_rem):
    v = NORM4[qp_rem]
    m = np.empty((4, 4), np.int64)
    for i in range(4):
        for j in range(4):
            m[i, j] = 16 * (v[0] if (i % 2 == 0 and j % 2 == 0) else v[1] if (i % 2 == 1 and j % 2 == 1) else v[2])
    return m


def level_scale8(qp_rem):
    v = NORM8[qp_rem]
    m = np.empty((8, 8), np.int64)
    for i in range(8):
        for j in range(8):
            if i % 4 == 0 and j % 4 == 0:
                c = 0
            elif i % 2 == 1 and j % 2 == 1:
                c = 1
            elif i % 4 == 2 and j % 4 == 2:
                c = 2
            elif (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
                c = 3
            elif (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
                c = 4
            else:
                c = 5
            m[i, j] = 16 * v[c]
    return m


LS4 = [level_scale4(k) for k in range(6)]
LS8 = [level_scale8(k) for k in range(6)]


def idct4(d):
    """8.5.12.2: d[y][x] scaled coefficients -> residual r[y][x]."""
    d = d.astype(np.int64)
    f = np.empty((4, 4), np.int64)
    for i in range(4):  # rows
        e0 = d[i, 0] + d[i, 2]
        e1 = d[i, 0] - d[i, 2]
        e2 = (d[i, 1] >> 1) - d[i, 3]
        e3 = d[i, 1] + (d[i, 3] >> 1)
        f[i] = (e0 + e3, e1 + e2, e1 - e2, e0 - e3)
    r = np.empty((4, 4), np.int64)
    for j in range(4):  # columns
        g0 = f[0, j] + f[2, j]
        g1 = f[0, j] - f[2, j]
        g2 = (f[1, j] >> 1) - f[3, j]
        g3 = f[1, j] + (f[3, j] >> 1)
        r[:, j] = (g0 + g3, g1 + g2, g1 - g2, g0 - g3)
    return (r + 32) >> 6


def _idct8_1d(a):
    e0 = a[0] + a[4]
    e1 = -a[3] + a[5] - a[7] - (a[7] >> 1)
    e2 = a[0] - a[4]
    e3 = a[1] + a[7] - a[3] - (a[3] >> 1)
    e4 = (a[2] >> 1) - a[6]
    e5 = -a[1] + a[7] + a[5] + (a[5] >> 1)
    e6 = a[2] + (a[6] >> 1)
    e7 = a[3] + a[5] + a[1] + (a[1] >> 1)
    f0 = e0 + e6
    f1 = e1 + (e7 >> 2)
    f2 = e2 + e4
    f3 = e3 + (e5 >> 2)
    f4 = e2 - e4
    f5 = (e3 >> 2) - e5
    f6 = e0 - e6
    f7 = e7 - (e1 >> 2)
    return (f0 + f7, f2 + f5, f4 + f3, f6 + f1, f6 - f1, f4 - f3, f2 - f5, f0 - f7)


def idct8(d):
    """8.5.13: 8x8 scaled coefficients d[y][x] -> residual."""
    d = d.astype(np.int64)
    g = np.empty((8, 8), np.int64)
    for i in range(8):
        g[i] = _idct8_1d([int(v) for v in d[i]])
    m = np.empty((8, 8), np.int64)
    for j in range(8):
        m[:, j] = _idct8_1d([int(v) for v in g[:, j]])
    return (m + 32) >> 6


def clip1(a):
    return np.clip(a, 0, 255)


# ---------------------------------------------------------------------------------------------- intra prediction (8.3)

def pred4x4(mode, top, left, tl, tr):
    """top: 4 samples or None, left: 4 or None, tl: sample or None, tr: 4 samples (already substituted) or None."""
    P = np.zeros((4, 4), np.int64)
    if mode == 0:
        P[:] = np.array(top)[None, :]
    elif mode == 1:
        P[:] = np.array(left)[:, None]
    elif mode == 2:
        if top is not None and left is not None:
            P[:] = (sum(top) + sum(left) + 4) >> 3
        elif left is not None:
            P[:] = (sum(left) + 2) >> 2
        elif top is not None:
            P[:] = (sum(top) + 2) >> 2
        else:
            P[:] = 128
    elif mode == 3:  # diagonal down-left
        t = list(top) + list(tr)
        for y in range(4):
            for x in range(4):
                if x == 3 and y == 3:
                    P[y, x] = (t[6] + 3 * t[7] + 2) >> 2
                else:
                    P[y, x] = (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2
    else:
        # p(x, -1) = top[x], p(-1, y) = left[y], p(-1, -1) = tl
        def p(x, y):
            if y == -1:
                return tl if x == -1 else top[x]
            return left[y]
        for y in range(4):
            for x in range(4):
                if mode == 4:  # diagonal down-right
                    if x > y:
                        v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                    elif x < y:
                        v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                    else:
                        v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
                elif mode == 5:  # vertical-right
                    z = 2 * x - y
                    if z >= 0 and z % 2 == 0:
                        v = (p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 1) >> 1
                    elif z >= 0:
                        v = (p(x - (y >> 1) - 2, -1) + 2 * p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 2) >> 2
                    elif z == -1:
                        v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                    else:
                        v = (p(-1, y - 1) + 2 * p(-1, y - 2) + p(-1, y - 3) + 2) >> 2
                elif mode == 6:  # horizontal-down
                    z = 2 * y - x
                    if z >= 0 and z % 2 == 0:
                        v = (p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 1) >> 1
                    elif z >= 0:
                        v = (p(-1, y - (x >> 1) - 2) + 2 * p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 2) >> 2
                    elif z == -1:
                        v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                    else:
                        v = (p(x - 1, -1) + 2 * p(x - 2, -1) + p(x - 3, -1) + 2) >> 2
                elif mode == 7:  # vertical-left
                    t = list(top) + list(tr)
                    if y % 2 == 0:
                        v = (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1
                    else:
                        v = (t[x + (y >> 1)] + 2 * t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2
                else:  # 8: horizontal-up
                    z = x + 2 * y
                    if z > 5:
                        v = left[3]
                    elif z == 5:
                        v = (left[2] + 3 * left[3] + 2) >> 2
                    elif z % 2 == 0:
                        v = (left[y + (x >> 1)] + left[y + (x >> 1) + 1] + 1) >> 1
                    else:
                        v = (left[y + (x >> 1)] + 2 * left[y + (x >> 1) + 1] + left[y + (x >> 1) + 2] + 2) >> 2
                P[y, x] = v
    return P


def pred8x8l(mode, top, left, tl, tr):
    """Intra 8x8 with the reference sample filter (8.3.2.2.1).  top / left: 8 samples or None, tl: sample or None,
    tr: 8 samples or None (None: substituted by top[7] if top exists)."""
    # assemble p'[x, -1] for x = -1..15 and p'[-1, y]
    have_top, have_left, have_tl = top is not None, left is not None, tl is not None
    if have_top:
        t = list(top) + (list(tr) if tr is not None else [top[7]] * 8)
        ft = [0] * 16
        ft[0] = ((tl if have_tl else t[0]) + 2 * t[0] + t[1] + 2) >> 2 if have_tl else (3 * t[0] + t[1] + 2) >> 2
        for x in range(1, 15):
            ft[x] = (t[x - 1] + 2 * t[x] + t[x + 1] + 2) >> 2
        ft[15] = (t[14] + 3 * t[15] + 2) >> 2
    if have_tl:
        if have_top and have_left:
            ftl = (top[0] + 2 * tl + left[0] + 2) >> 2
        elif have_top:
            ftl = (3 * tl + top[0] + 2) >> 2
        elif have_left:
            ftl = (3 * tl + left[0] + 2) >> 2
        else:
            ftl = tl
    if have_left:
        fl = [0] * 8
        fl[0] = (tl + 2 * left[0] + left[1] + 2) >> 2 if have_tl else (3 * left[0] + left[1] + 2) >> 2
        for y in range(1, 7):
            fl[y] = (left[y - 1] + 2 * left[y] + left[y + 1] + 2) >> 2
        fl[7] = (left[6] + 3 * left[7] + 2) >> 2

    def p(x, y):
        if y == -1:
            return ftl if x == -1 else ft[x]
        return fl[y]

    P = np.zeros((8, 8), np.int64)
    for y in range(8):
        for x in range(8):
            if mode == 0:
                v = ft[x]
            elif mode == 1:
                v = fl[y]
            elif mode == 2:
                if have_top and have_left:
                    v = (sum(ft[:8]) + sum(fl) + 8) >> 4
                elif have_left:
                    v = (sum(fl) + 4) >> 3
                elif have_top:
                    v = (sum(ft[:8]) + 4) >> 3
                else:
                    v = 128
            elif mode == 3:
                v = (ft[14] + 3 * ft[15] + 2) >> 2 if (x == 7 and y == 7) else (ft[x + y] + 2 * ft[x + y + 1] + ft[x + y + 2] + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                elif x < y:
                    v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                else:
                    v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
            elif mode == 5:
                z = 2 * x - y
                if z >= 0 and z % 2 == 0:
                    v = (p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 1) >> 1
                elif z >= 0:
                    v = (p(x - (y >> 1) - 2, -1) + 2 * p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(-1, y - 2 * x - 1) + 2 * p(-1, y - 2 * x - 2) + p(-1, y - 2 * x - 3) + 2) >> 2
            elif mode == 6:
                z = 2 * y - x
                if z >= 0 and z % 2 == 0:
                    v = (p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 1) >> 1
                elif z >= 0:
                    v = (p(-1, y - (x >> 1) - 2) + 2 * p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(x - 2 * y - 1, -1) + 2 * p(x - 2 * y - 2, -1) + p(x - 2 * y - 3, -1) + 2) >> 2
            elif mode == 7:
                if y % 2 == 0:
                    v = (ft[x + (y >> 1)] + ft[x + (y >> 1) + 1] + 1) >> 1
                else:
                    v = (ft[x + (y >> 1)] + 2 * ft[x + (y >> 1) + 1] + ft[x + (y >> 1) + 2] + 2) >> 2
            else:
                z = x + 2 * y
                if z > 13:
                    v = fl[7]
                elif z == 13:
                    v = (fl[6] + 3 * fl[7] + 2) >> 2
                elif z % 2 == 0:
                    v = (fl[y + (x >> 1)] + fl[y + (x >> 1) + 1] + 1) >> 1
                else:
                    v = (fl[y + (x >> 1)] + 2 * fl[y + (x >> 1) + 1] + fl[y + (x >> 1) + 2] + 2) >> 2
            P[y, x] = v
    return P


def pred_plane(top, left, tl, n):
    """Plane prediction for an n x n block (16: luma, 8: chroma).  top / left: n samples, tl = p[-1, -1]."""
    h = n // 2
    t = [tl] + list(top)   # t[x + 1] = p[x, -1]
    l = [tl] + list(left)  # l[y + 1] = p[-1, y]
    H = sum((x + 1) * (t[h + 1 + x] - t[h - 1 - x]) for x in range(h))
    V = sum((y + 1) * (l[h + 1 + y] - l[h - 1 - y]) for y in range(h))
    if n == 16:
        b, c = (5 * H + 32) >> 6, (5 * V + 32) >> 6
    else:
        b, c = (34 * H + 32) >> 6, (34 * V + 32) >> 6
    a = 16 * (left[n - 1] + top[n - 1])
    yy, xx = np.mgrid[0:n, 0:n]
    return clip1((a + b * (xx - (h - 1)) + c * (yy - (h - 1)) + 16) >> 5)


# ---------------------------------------------------------------------------------------------- the picture decoder

I_NXN, I_16X16, I_PCM = 0, 1, 2


def blk_idx(bx, by):
    """4x4 block index (decoding order inside a macroblock) of the block at column bx, row by."""
    return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1)


BLK_XY = [None] * 16
for _by in range(4):
    for _bx in range(4):
        BLK_XY[blk_idx(_bx, _by)] = (_bx, _by)


class IntraDecoder:
    def __init__(self, data):
        self.nals = nal_units(data)
        self.sps = self.pps = None
        for n in self.nals:
            t = n[0] & 31
            if t == 7 and self.sps is None:
                self.sps = parse_sps(n)
            elif t == 8 and self.pps is None:
                self.pps = parse_pps(n)
        assert self.sps and self.pps and self.pps["cabac"]

    def decode_first_idr(self, deblock=True, progress=None):
        idr = next(n for n in self.nals if (n[0] & 31) == 5)
        sps, pps = self.sps, self.pps
        r = BitReader(idr, 8)
        assert r.ue() == 0, "first_mb_in_slice"
        st = r.ue()
        assert st % 5 == 2, "I slice expected"
        r.ue()
        r.u(sps["log2_max_frame_num"])
        r.ue()  # idr_pic_id
        if sps["poc_type"] == 0:
            r.u(sps["log2_max_poc_lsb"])
        r.u(1)
        r.u(1)  # dec_ref_pic_marking of an IDR picture
        self.slice_qp = pps["pic_init_qp"] + r.se()
        self.disable_deblock, self.alpha_off, self.beta_off = 0, 0, 0
        if pps["deblocking_filter_control_present"]:
            self.disable_deblock = r.ue()
            if self.disable_deblock != 1:
                self.alpha_off = 2 * r.se()
                self.beta_off = 2 * r.se()
        while r.p % 8:
            assert r.u(1) == 1, "cabac_alignment_one_bit"
        self.c = Cabac(idr, r.p, self.slice_qp)
        W, H = sps["mb_w"], sps["mb_h"]
        self.W, self.H = W, H
        self.Y = np.zeros((H * 16, W * 16), np.int64)
        self.C = [np.zeros((H * 8, W * 8), np.int64), np.zeros((H * 8, W * 8), np.int64)]
        # per-macroblock state
        self.mb_type = np.full((H, W), -1, np.int64)
        self.mb_qp = np.zeros((H, W), np.int64)
        self.mb_t8 = np.zeros((H, W), np.int64)
        self.mb_cbp = np.zeros((H, W), np.int64)          # luma bits 0-3, chroma value in bits 4-5
        self.mb_chroma_mode = np.zeros((H, W), np.int64)
        self.pred_mode = np.full((H * 4, W * 4), 2, np.int64)  # Intra4x4/8x8PredMode per 4x4 block; 2 = DC for the others
        self.nz_luma = np.zeros((H * 4, W * 4), np.int64)      # coded_block_flag of luma 4x4 blocks (cat 1 / 2 / 5)
        self.nz_dc = np.zeros((H, W, 3), np.int64)             # coded_block_flag of Intra16x16 DC, Cb DC, Cr DC
        self.nz_chroma = np.zeros((2, H * 2, W * 2), np.int64)  # chroma AC blocks
        qp, last_dqp = self.slice_qp, 0
        for mby in range(H):
            for mbx in range(W):
                qp, last_dqp = self._macroblock(mbx, mby, qp, last_dqp)
                end = self.c.terminate()
                if end:
                    assert (mbx, mby) == (W - 1, H - 1), f"end_of_slice at macroblock ({mbx}, {mby})"
            if progress:
                progress(mby)
        self.Y_pre, self.C_pre = self.Y.copy(), [c.copy() for c in self.C]
        if deblock and self.disable_deblock != 1:
            self._deblock()
        cl, cr, ct, cb = sps["crop"]
        y = self.Y[2 * ct: H * 16 - 2 * cb, 2 * cl: W * 16 - 2 * cr].astype(np.uint8)
        u = self.C[0][ct: H * 8 - cb, cl: W * 8 - cr].astype(np.uint8)
        v = self.C[1][ct: H * 8 - cb, cl: W * 8 - cr].astype(np.uint8)
        uv = np.empty((u.shape[0], u.shape[1] * 2), np.uint8)
        uv[:, 0::2], uv[:, 1::2] = u, v
        return y, uv

    # ---- availability helpers
    def _avail(self, mbx, mby):
        return 0 <= mbx < self.W and 0 <= mby < self.H and self.mb_type[mby, mbx] >= 0

    # ---- syntax elements
    def _mb_type(self, mbx, mby):
        c = self.c
        inc = 0
        if self._avail(mbx - 1, mby) and self.mb_type[mby, mbx - 1] != I_NXN:
            inc += 1
        if self._avail(mbx, mby - 1) and self.mb_type[mby - 1, mbx] != I_NXN:
            inc += 1
        if c.decision(3 + inc) == 0:
            return I_NXN, 0, 0, 0
        if c.terminate():
            return I_PCM, 0, 0, 0
        luma = 15 if c.decision(3 + 3) else 0
        chroma = 0
        if c.decision(3 + 4):
            chroma = 2 if c.decision(3 + 5) else 1
        pm = 2 * c.decision(3 + 6)
        pm += c.decision(3 + 7)
        return I_16X16, pm, luma, chroma

    def _pred_mode_syntax(self):
        c = self.c
        if c.decision(68):
            return -1
        m = c.decision(69)
        m |= c.decision(69) << 1
        m |= c.decision(69) << 2
        return m

    def _cbp(self, mbx, mby):
        c = self.c
        # neighbouring luma bits: unavailable -> treated as coded (condTermFlag 0); I_PCM -> coded
        def luma_bits(x, y):
            if not self._avail(x, y):
                return 15
            if self.mb_type[y, x] == I_PCM:
                return 15
            return int(self.mb_cbp[y, x]) & 15
        a, b = luma_bits(mbx - 1, mby), luma_bits(mbx, mby - 1)
        cbp = 0
        cbp |= c.decision(73 + (0 if a & 2 else 1) + 2 * (0 if b & 4 else 1))
        cbp |= c.decision(73 + (0 if cbp & 1 else 1) + 2 * (0 if b & 8 else 1)) << 1
        cbp |= c.decision(73 + (0 if a & 8 else 1) + 2 * (0 if cbp & 1 else 1)) << 2
        cbp |= c.decision(73 + (0 if cbp & 4 else 1) + 2 * (0 if cbp & 2 else 1)) << 3

        def chroma_val(x, y):
            if not self._avail(x, y):
                return 0
            if self.mb_type[y, x] == I_PCM:
                return 2
            return int(self.mb_cbp[y, x]) >> 4
        ca, cb_ = chroma_val(mbx - 1, mby), chroma_val(mbx, mby - 1)
        chroma = 0
        if c.decision(77 + (1 if ca else 0) + 2 * (1 if cb_ else 0)):
            chroma = 1 + c.decision(77 + 4 + (1 if ca == 2 else 0) + 2 * (1 if cb_ == 2 else 0))
        return cbp, chroma

    def _chroma_pred_mode(self, mbx, mby):
        c = self.c
        inc = 0
        for (x, y) in ((mbx - 1, mby), (mbx, mby - 1)):
            if self._avail(x, y) and self.mb_type[y, x] != I_PCM and self.mb_chroma_mode[y, x] != 0:
                inc += 1
        if c.decision(64 + inc) == 0:
            return 0
        if c.decision(64 + 3) == 0:
            return 1
        return 2 + c.decision(64 + 3)

    def _qp_delta(self, last_dqp):
        c = self.c
        if c.decision(60 + (1 if last_dqp != 0 else 0)) == 0:
            return 0
        val, ctx = 1, 60 + 2
        while c.decision(ctx):
            ctx = 60 + 3
            val += 1
            assert val < 200
        return (val + 1) >> 1 if val & 1 else -(val >> 1)

    def _residual(self, cat, max_coeff, cbf_ctx_inc):
        """residual_block_cabac: returns the list of max_coeff levels in scan order, and coded_block_flag."""
        c = self.c
        coeffs = [0] * max_coeff
        if cbf_ctx_inc is not None:
            if c.decision(CBF_OFF + 4 * cat + cbf_ctx_inc) == 0:
                return coeffs, 0
        sig = []
        num = max_coeff
        i = 0
        while i < num - 1:
            if cat == 5:
                s_inc, l_inc = SIG8[i], LAST8[i]
            elif cat == 3:
                s_inc = l_inc = min(i, 2)
            else:
                s_inc = l_inc = i
            if c.decision(SIG_OFF[cat] + s_inc):
                sig.append(i)
                if c.decision(LAST_OFF[cat] + l_inc):
                    num = i + 1
                    break
            i += 1
        else:
            sig.append(num - 1)  # reached the last position: inferred significant
        eq1 = gt1 = 0
        for pos in reversed(sig):
            ctx = ABS_OFF[cat] + (0 if gt1 else min(4, 1 + eq1))
            val = 0
            if c.decision(ctx):
                val = 1
                ctx = ABS_OFF[cat] + 5 + min(4 - (1 if cat == 3 else 0), gt1)
                while val < 14 and c.decision(ctx):
                    val += 1
                if val == 14:
                    k = 0
                    while c.bypass():
                        val += 1 << k
                        k += 1
                        assert k < 32
                    while k:
                        k -= 1
                        val += c.bypass() << k
            level = val + 1
            if level == 1:
                eq1 += 1
            else:
                gt1 += 1
            coeffs[pos] = -level if c.bypass() else level
        return coeffs, 1

    # ---- coded_block_flag contexts
    def _cbf_luma_inc(self, gx, gy, mbx, mby):
        """gx, gy: global 4x4 block coordinates of the current block."""
        def cond(x, y, inside):
            if inside:
                return int(self.nz_luma[y, x])
            nx, ny = x >> 2, y >> 2
            if not self._avail(nx, ny):
                return 1  # unavailable, current macroblock is intra
            if self.mb_type[ny, nx] == I_PCM:
                return 1
            return int(self.nz_luma[y, x])
        a = cond(gx - 1, gy, (gx - 1) >> 2 == mbx) if gx > 0 else 1
        b = cond(gx, gy - 1, (gy - 1) >> 2 == mby) if gy > 0 else 1
        return a + 2 * b

    def _cbf_dc_inc(self, mbx, mby, comp):
        def cond(x, y):
            if not self._avail(x, y):
                return 1
            if self.mb_type[y, x] == I_PCM:
                return 1
            return int(self.nz_dc[y, x, comp])
        return cond(mbx - 1, mby) + 2 * cond(mbx, mby - 1)

    def _cbf_chroma_ac_inc(self, comp, cx, cy, mbx, mby):
        def cond(x, y, inside):
            if inside:
                return int(self.nz_chroma[comp, y, x])
            nx, ny = x >> 1, y >> 1
            if not self._avail(nx, ny):
                return 1
            if self.mb_type[ny, nx] == I_PCM:
                return 1
            return int(self.nz_chroma[comp, y, x])
        a = cond(cx - 1, cy, (cx - 1) >> 1 == mbx) if cx > 0 else 1
        b = cond(cx, cy - 1, (cy - 1) >> 1 == mby) if cy > 0 else 1
        return a + 2 * b

    # ---- one macroblock
    def _macroblock(self, mbx, mby, qp, last_dqp):
        c = self.c
        mbt, pm16, cbp_l, cbp_c = self._mb_type(mbx, mby)
        x0, y0 = mbx * 16, mby * 16
        self.mb_type[mby, mbx] = mbt
        if mbt == I_PCM:
            r = c.r
            # pcm alignment: the arithmetic decoder has consumed 9 + renormalisation bits; per 9.3.1.2 the decoding engine is
            # re-initialised after the PCM samples.  The bit position of the samples is the next byte boundary.
            r.p = (r.p + 7) & ~7
            raise NotImplementedError("I_PCM macroblock (not present in the reference clip)")
        t8 = 0
        modes = None
        if mbt == I_NXN:
            if self.pps["transform_8x8_mode"]:
                inc = sum(1 for (x, y) in ((mbx - 1, mby), (mbx, mby - 1)) if self._avail(x, y) and self.mb_t8[y, x])
                t8 = c.decision(399 + inc)
            self.mb_t8[mby, mbx] = t8
            modes = [self._pred_mode_syntax() for _ in range(4 if t8 else 16)]
        chroma_mode = self._chroma_pred_mode(mbx, mby)
        self.mb_chroma_mode[mby, mbx] = chroma_mode
        if mbt == I_NXN:
            cbp_l, cbp_c = self._cbp(mbx, mby)
        self.mb_cbp[mby, mbx] = cbp_l | (cbp_c << 4)
        if mbt == I_16X16 or cbp_l or cbp_c:
            dqp = self._qp_delta(last_dqp)
            qp = (qp + dqp + 52) % 52
            last_dqp = dqp
        else:
            last_dqp = 0
        self.mb_qp[mby, mbx] = qp
        qpc = QPC[min(max(qp + self.pps["chroma_qp_index_offset"], 0), 51)]
        qpc2 = QPC[min(max(qp + self.pps["second_chroma_qp_index_offset"], 0), 51)]

        # ---------------- luma
        if mbt == I_16X16:
            dc, cbf = self._residual(0, 16, self._cbf_dc_inc(mbx, mby, 0))
            self.nz_dc[mby, mbx, 0] = cbf
            ac = {}
            for idx in range(16):
                bx, by = BLK_XY[idx]
                gx, gy = mbx * 4 + bx, mby * 4 + by
                if cbp_l:
                    lv, f = self._residual(1, 15, self._cbf_luma_inc(gx, gy, mbx, mby))
                    self.nz_luma[gy, gx] = f
                    ac[idx] = lv
                else:
                    self.nz_luma[gy, gx] = 0
            # prediction
            top = self.Y[y0 - 1, x0:x0 + 16] if mby > 0 else None
            left = self.Y[y0:y0 + 16, x0 - 1] if mbx > 0 else None
            if pm16 == 0:
                P = np.tile(np.array(top)[None, :], (16, 1))
            elif pm16 == 1:
                P = np.tile(np.array(left)[:, None], (1, 16))
            elif pm16 == 2:
                if top is not None and left is not None:
                    v = (int(top.sum()) + int(left.sum()) + 16) >> 5
                elif left is not None:
                    v = (int(left.sum()) + 8) >> 4
                elif top is not None:
                    v = (int(top.sum()) + 8) >> 4
                else:
                    v = 128
                P = np.full((16, 16), v, np.int64)
            else:
                P = pred_plane([int(v) for v in top], [int(v) for v in left], int(self.Y[y0 - 1, x0 - 1]), 16)
            # DC: inverse 4x4 Hadamard + scaling (8.5.10)
            cm = np.zeros((4, 4), np.int64)
            for k, (zx, zy) in enumerate(ZIGZAG4):
                cm[zy, zx] = dc[k]
            A = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], np.int64)
            f = A @ cm @ A
            ls = int(LS4[qp % 6][0, 0])
            if qp >= 36:
                dcy = (f * ls) << (qp // 6 - 6)
            else:
                dcy = (f * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)
            for idx in range(16):
                bx, by = BLK_XY[idx]
                d = np.zeros((4, 4), np.int64)
                if idx in ac:
                    for k, (zx, zy) in enumerate(ZIGZAG4[1:]):
                        d[zy, zx] = ac[idx][k]
                    d = self._scale4(d, qp)
                d[0, 0] = dcy[by, bx]
                res = idct4(d)
                self.Y[y0 + 4 * by: y0 + 4 * by + 4, x0 + 4 * bx: x0 + 4 * bx + 4] = clip1(P[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] + res)
        elif t8:
            for b8 in range(4):
                bx8, by8 = b8 & 1, b8 >> 1
                gx, gy = mbx * 4 + 2 * bx8, mby * 4 + 2 * by8
                # prediction mode
                predm = self._predicted_mode(gx, gy)
                m = modes[b8]
                mode = predm if m < 0 else (m if m < predm else m + 1)
                self.pred_mode[gy:gy + 2, gx:gx + 2] = mode
                px, py = x0 + 8 * bx8, y0 + 8 * by8
                top = [int(v) for v in self.Y[py - 1, px:px + 8]] if py > 0 else None
                left = [int(v) for v in self.Y[py:py + 8, px - 1]] if px > 0 else None
                tl = int(self.Y[py - 1, px - 1]) if (px > 0 and py > 0) else None
                tr = None
                if py > 0 and px + 8 < self.W * 16:
                    if by8 == 0:
                        tr_ok = (bx8 == 0) or self._avail(mbx + 1, mby - 1)
                    else:
                        tr_ok = (bx8 == 0)
                    if tr_ok:
                        tr = [int(v) for v in self.Y[py - 1, px + 8:px + 16]]
                P = pred8x8l(mode, top, left, tl, tr)
                if cbp_l & (1 << b8):
                    lv, _ = self._residual(5, 64, None)
                    self.nz_luma[gy:gy + 2, gx:gx + 2] = 1
                    d = np.zeros((8, 8), np.int64)
                    for k, zz in enumerate(ZIGZAG8):
                        d[zz >> 3, zz & 7] = lv[k]
                    ls = LS8[qp % 6]
                    if qp >= 36:
                        d = (d * ls) << (qp // 6 - 6)
                    else:
                        d = (d * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)
                    P = clip1(P + idct8(d))
                else:
                    self.nz_luma[gy:gy + 2, gx:gx + 2] = 0
                self.Y[py:py + 8, px:px + 8] = P
        else:
            for idx in range(16):
                bx, by = BLK_XY[idx]
                gx, gy = mbx * 4 + bx, mby * 4 + by
                predm = self._predicted_mode(gx, gy)
                m = modes[idx]
                mode = predm if m < 0 else (m if m < predm else m + 1)
                self.pred_mode[gy, gx] = mode
                px, py = x0 + 4 * bx, y0 + 4 * by
                top = [int(v) for v in self.Y[py - 1, px:px + 4]] if py > 0 else None
                left = [int(v) for v in self.Y[py:py + 4, px - 1]] if px > 0 else None
                tl = int(self.Y[py - 1, px - 1]) if (px > 0 and py > 0) else None
                tr = None
                if top is not None:
                    ok = False
                    if px + 4 < self.W * 16:
                        if by == 0:
                            ok = (bx < 3) or self._avail(mbx + 1, mby - 1)
                        elif bx < 3:
                            ok = blk_idx(bx + 1, by - 1) < idx
                    tr = [int(v) for v in self.Y[py - 1, px + 4:px + 8]] if ok else [top[3]] * 4
                P = pred4x4(mode, top, left, tl, tr)
                if cbp_l & (1 << (idx >> 2)):
                    lv, f = self._residual(2, 16, self._cbf_luma_inc(gx, gy, mbx, mby))
                    self.nz_luma[gy, gx] = f
                    if f:
                        d = np.zeros((4, 4), np.int64)
                        for k, (zx, zy) in enumerate(ZIGZAG4):
                            d[zy, zx] = lv[k]
                        P = clip1(P + idct4(self._scale4(d, qp)))
                else:
                    self.nz_luma[gy, gx] = 0
                self.Y[py:py + 4, px:px + 4] = P

        # ---------------- chroma
        cx0, cy0 = mbx * 8, mby * 8
        dcs = [[0] * 4, [0] * 4]
        if cbp_c:
            for comp in range(2):
                lv, f = self._residual(3, 4, self._cbf_dc_inc(mbx, mby, 1 + comp))
                self.nz_dc[mby, mbx, 1 + comp] = f
                dcs[comp] = lv
        else:
            self.nz_dc[mby, mbx, 1:] = 0
        acs = [{}, {}]
        for comp in range(2):
            for b in range(4):
                bx, by = b & 1, b >> 1
                gx, gy = mbx * 2 + bx, mby * 2 + by
                if cbp_c == 2:
                    lv, f = self._residual(4, 15, self._cbf_chroma_ac_inc(comp, gx, gy, mbx, mby))
                    self.nz_chroma[comp, gy, gx] = f
                    acs[comp][b] = lv
                else:
                    self.nz_chroma[comp, gy, gx] = 0
        for comp in range(2):
            plane = self.C[comp]
            q = qpc if comp == 0 else qpc2
            top = [int(v) for v in plane[cy0 - 1, cx0:cx0 + 8]] if mby > 0 else None
            left = [int(v) for v in plane[cy0:cy0 + 8, cx0 - 1]] if mbx > 0 else None
            P = np.zeros((8, 8), np.int64)
            if chroma_mode == 0:  # DC, per 4x4 block (8.3.4.1-3)
                for b in range(4):
                    bx, by = b & 1, b >> 1
                    st = sum(top[4 * bx:4 * bx + 4]) if top is not None else None
                    sl = sum(left[4 * by:4 * by + 4]) if left is not None else None
                    if (bx, by) in ((0, 0), (1, 1)):
                        if st is not None and sl is not None:
                            v = (st + sl + 4) >> 3
                        elif st is not None:
                            v = (st + 2) >> 2
                        elif sl is not None:
                            v = (sl + 2) >> 2
                        else:
                            v = 128
                    elif (bx, by) == (1, 0):
                        v = (st + 2) >> 2 if st is not None else ((sl + 2) >> 2 if sl is not None else 128)
                    else:
                        v = (sl + 2) >> 2 if sl is not None else ((st + 2) >> 2 if st is not None else 128)
                    P[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = v
            elif chroma_mode == 1:
                P[:] = np.array(left)[:, None]
            elif chroma_mode == 2:
                P[:] = np.array(top)[None, :]
            else:
                P = pred_plane(top, left, int(plane[cy0 - 1, cx0 - 1]), 8)
            # DC transform 2x2 + scaling (8.5.11)
            cdc = np.array([[dcs[comp][0], dcs[comp][1]], [dcs[comp][2], dcs[comp][3]]], np.int64)
            A2 = np.array([[1, 1], [1, -1]], np.int64)
            f = A2 @ cdc @ A2
            dcc = ((f * int(LS4[q % 6][0, 0])) << (q // 6)) >> 5
            for b in range(4):
                bx, by = b & 1, b >> 1
                d = np.zeros((4, 4), np.int64)
                if b in acs[comp]:
                    for k, (zx, zy) in enumerate(ZIGZAG4[1:]):
                        d[zy, zx] = acs[comp][b][k]
                    d = self._scale4(d, q)
                d[0, 0] = dcc[by, bx]
                res = idct4(d)
                plane[cy0 + 4 * by:cy0 + 4 * by + 4, cx0 + 4 * bx:cx0 + 4 * bx + 4] = clip1(P[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] + res)
        return qp, last_dqp

    def _predicted_mode(self, gx, gy):
        """predIntra4x4PredMode / predIntra8x8PredMode (8.3.1.1, 8.3.2.1): Min(A, B); when EITHER neighbouring macroblock is not
        available both count as DC (dcPredModePredictedFlag)."""
        if gx == 0 or gy == 0:  # one slice covering the picture: unavailable == outside the picture
            return 2
        return min(self._neigh_mode(gx - 1, gy), self._neigh_mode(gx, gy - 1))

    def _neigh_mode(self, gx, gy):
        """Intra4x4/8x8PredMode of the neighbouring 4x4 block (8.3.1.1 / 8.3.2.1): DC (2) when the macroblock is not
        available or not coded in Intra 4x4 / 8x8 mode."""
        if gx < 0 or gy < 0:
            return 2
        nx, ny = gx >> 2, gy >> 2
        if not self._avail(nx, ny) or self.mb_type[ny, nx] != I_NXN:
            return 2
        return int(self.pred_mode[gy, gx])

    @staticmethod
    def _scale4(c, qp):
        ls = LS4[qp % 6]
        if qp >= 24:
            return (c * ls) << (qp // 6 - 4)
        return (c * ls + (1 << (3 - qp // 6))) >> (4 - qp // 6)

    # ---- deblocking filter (8.7), intra pictures: bS 4 on macroblock edges, 3 inside
    def _deblock(self):
        W, H = self.W, self.H
        co = self.pps["chroma_qp_index_offset"]
        co2 = self.pps["second_chroma_qp_index_offset"]

        def qpc_of(qpy, off):
            return QPC[min(max(qpy + off, 0), 51)]

        for mby in range(H):
            for mbx in range(W):
                qp_q = int(self.mb_qp[mby, mbx])
                t8 = int(self.mb_t8[mby, mbx])
                # vertical edges then horizontal edges, luma
                for vertical in (True, False):
                    for e in range(4):
                        if t8 and (e & 1):
                            continue
                        if e == 0:
                            if vertical and mbx == 0:
                                continue
                            if not vertical and mby == 0:
                                continue
                            qp_p = int(self.mb_qp[mby, mbx - 1]) if vertical else int(self.mb_qp[mby - 1, mbx])
                            bs = 4
                        else:
                            qp_p, bs = qp_q, 3
                        self._filter_edge(self.Y, mbx * 16, mby * 16, 16, e * 4, vertical, bs, (qp_p + qp_q + 1) >> 1, True)
                    # chroma: edges 0 and 2 of the 8x8 block (in units of 4 chroma samples: 0, 4)
                    for comp, off in ((0, co), (1, co2)):
                        for e in (0, 1):
                            if e == 0:
                                if vertical and mbx == 0:
                                    continue
                                if not vertical and mby == 0:
                                    continue
                                qp_p = int(self.mb_qp[mby, mbx - 1]) if vertical else int(self.mb_qp[mby - 1, mbx])
                                bs = 4
                            else:
                                qp_p, bs = qp_q, 3
                            qav = (qpc_of(qp_p, off) + qpc_of(qp_q, off) + 1) >> 1
                            self._filter_edge(self.C[comp], mbx * 8, mby * 8, 8, e * 4, vertical, bs, qav, False)

    def _filter_edge(self, plane, x0, y0, n, e, vertical, bs, qav, luma):
        ia = min(max(qav + self.alpha_off, 0), 51)
        ib = min(max(qav + self.beta_off, 0), 51)
        alpha, beta = ALPHA[ia], BETA[ib]
        if alpha == 0 or beta == 0:
            return
        tc0 = TC0[ia][bs - 1] if bs < 4 else 0
        for k in range(n):
            if vertical:
                xs, ys, dx, dy = x0 + e, y0 + k, 1, 0
            else:
                xs, ys, dx, dy = x0 + k, y0 + e, 0, 1

            def g(i):  # i >= 0: q_i, i < 0: p_(-i-1)
                return int(plane[ys + dy * i, xs + dx * i])

            p0, p1, p2 = g(-1), g(-2), g(-3)
            q0, q1, q2 = g(0), g(1), g(2)
            if not (abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
                continue
            if bs < 4:
                ap, aq = abs(p2 - p0) < beta, abs(q2 - q0) < beta
                tc = tc0 + (int(ap) + int(aq) if luma else 1)
                delta = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc)
                plane[ys - dy, xs - dx] = min(max(p0 + delta, 0), 255)
                plane[ys, xs] = min(max(q0 - delta, 0), 255)
                if luma:
                    if ap:
                        plane[ys - 2 * dy, xs - 2 * dx] = p1 + min(max((p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1, -tc0), tc0)
                    if aq:
                        plane[ys + dy, xs + dx] = q1 + min(max((q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1, -tc0), tc0)
            else:
                if luma:
                    p3, q3 = g(-4), g(3)
                    small = abs(p0 - q0) < ((alpha >> 2) + 2)
                    if abs(p2 - p0) < beta and small:
                        plane[ys - dy, xs - dx] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3
                        plane[ys - 2 * dy, xs - 2 * dx] = (p2 + p1 + p0 + q0 + 2) >> 2
                        plane[ys - 3 * dy, xs - 3 * dx] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3
                    else:
                        plane[ys - dy, xs - dx] = (2 * p1 + p0 + q1 + 2) >> 2
                    if abs(q2 - q0) < beta and small:
                        plane[ys, xs] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3
                        plane[ys + dy, xs + dx] = (p0 + q0 + q1 + q2 + 2) >> 2
                        plane[ys + 2 * dy, xs + 2 * dx] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3
                    else:
                        plane[ys, xs] = (2 * q1 + q0 + p1 + 2) >> 2
                else:
                    plane[ys - dy, xs - dx] = (2 * p1 + p0 + q1 + 2) >> 2
                    plane[ys, xs] = (2 * q1 + q0 + p1 + 2) >> 2
