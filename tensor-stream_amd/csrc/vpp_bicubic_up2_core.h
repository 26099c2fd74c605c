// vpp_bicubic_up2_core.h -- the resize of ONE thread tile of vpp_bicubic_up2.hip (BICUBIC at the exact ratio 1 : 2 -- 540p -> 1080p, 1080p -> 4K) as
// plain C++ on arrays of dwords: the kernel calls it on registers, and tests/host/bicubic_up2_host.cpp compiles the SAME text with g++ (the hardware
// operations emulated, vpp_bicubic_r32_core.h) and runs it over whole frames against the oracle.  Product code: includes nothing from oracle/.
//
// Arithmetic: that of vpp_bicubic_r32_core.h (Keys' cubic in fp64, every 4-tap sum rounded half away from zero and clamped to a byte, horizontally
// first, then down the column; reference src/Resize.cu:27-91, 314-357).  At ratio 1/2 the coordinate (j + 0.5) / 2 - 0.5 is exactly j / 2 - 1/4: output
// 2 m has p = m - 1, w = 3/4 -- coefficients (-9, 67, 225, -27) / 256 --, output 2 m + 1 has p = m, w = 1/4 -- (-27, 225, 67, -9) / 256: byte
// coefficients again, so a 4-tap sum is v_dot4_u32_u8 on the source dwords with compile-time byte masks (positive taps on the data, negative taps on
// the complemented data, one accumulator chain started at 128 - 255 * 36).
//
// Geometry of a thread tile: 8 output columns x 4 output rows from 4 source samples per row (ONE dword) extended by one dword on each side -- ext[r][0..2],
// byte e = source byte 4 q - 4 + e (luma: samples 4 q - 2 .. 4 q + 5 are tapped; chroma: pairs 2 q - 2 .. 2 q + 3 = all twelve bytes).  Rows: output rows
// 4 n .. 4 n + 3 tap source rows 2 n - 2 .. 2 n + 3 (6), the tile's two chroma output rows tap chroma rows n - 2 .. n + 2 (5).
//
// Edge rules -- NOT those of the down-scales (src/Resize.cu:32-43, 325-347 with a coordinate that starts at -1/4):
//   * output 0 of an axis: the coordinate is negative, the reference clamps it to 0 with w = 0: the value is the first sample itself (rows: the
//     horizontal result of source row 0) -- `first` / `first_row` below replace the sum by that byte;
//   * output 1 (and 2): p = 0, the -1 tap reads p: the caller replicates the first sample into the byte before the row / loads rows above the
//     plane clamped;
//   * the LAST THREE outputs of an axis have p + 2 beyond the plane: BOTH the +1 and +2 taps read p.  Columns: outputs 5, 6 of the row's last thread
//     (p = the run's sample 2) take their own copy of dwords 1, 2 (xa), output 7 (p = sample 3) another copy of dword 2 (xb) -- equal to the extended
//     row everywhere else; chroma alike on pair columns (1, 2) and 3.  Rows: the centre byte of the packed vertical window replicated over its
//     +1 / +2 bytes for output rows 1, 2, 3 of the last tile row -- chroma: its rows 0, 1, and row 1 of the tile row BEFORE the last.
#pragma once
#include "vpp_bilinear_up2_core.h"

namespace tsvpp {

constexpr int B2_NYR = 6, B2_NCR = 5; // luma / chroma source rows of a tile

// window start of output index c (0..7) along an axis, in samples relative to the thread's run
constexpr int b2_ws(int c) { return (c & 1) ? ((c - 1) >> 1) - 1 : (c >> 1) - 2; }
// 256 x coefficient of tap t of output index c
constexpr int b2_coef(int c, int t) {
    return (c & 1) ? (t == 0 ? -27 : t == 1 ? 225 : t == 2 ? 67 : -9) : (t == 0 ? -9 : t == 1 ? 67 : t == 2 ? 225 : -27);
}
constexpr uint32_t b2_acc0() { return (uint32_t)(128 - 255 * 36); }
template <bool CHROMA> constexpr uint32_t b2_hmask(int v, int d, bool neg) {
    const int c = CHROMA ? (v >> 1) : v;
    uint32_t m = 0;
    for (int t = 0; t < 4; t++) {
        const int e = CHROMA ? 2 * (b2_ws(c) + t) + (v & 1) + 4 : b2_ws(c) + t + 4;
        const int C = b2_coef(c, t);
        if ((e >> 2) == d && (neg ? C < 0 : C > 0)) m |= (uint32_t)(neg ? -C : C) << (8 * (e & 3));
    }
    return m;
}
constexpr uint32_t b2_vmask(int r, bool neg) {
    uint32_t m = 0;
    for (int t = 0; t < 4; t++) {
        const int C = b2_coef(r, t);
        if (neg ? C < 0 : C > 0) m |= (uint32_t)(neg ? -C : C) << (8 * t);
    }
    return m;
}
// which copy of dwords 1 / 2 output value v reads: 0 = the extended row, 1 = xa (the last thread's outputs 5, 6 / pair columns 1, 2), 2 = xb (7 / 3)
template <bool CHROMA> constexpr int b2_variant(int v) { return CHROMA ? (v >= 6 ? 2 : (v >= 2 ? 1 : 0)) : (v == 7 ? 2 : (v >= 5 ? 1 : 0)); }

// column edges (see the header): in place for the low edge, the last thread's own copies xa / xb for the high edge
template <bool CHROMA, int NROWS> BC_HD void b2_fix_rows(uint32_t (&ext)[NROWS][3], uint32_t (&xa)[NROWS][2], uint32_t (&xb)[NROWS], bool first, bool last) {
    const uint32_t sf = first ? bc_sel_first<CHROMA>() : BC_SEL_ID;
    const uint32_t s0 = last ? bc_sel_last0<CHROMA>() : BC_SEL_ID, s1 = last ? bc_sel_last1<CHROMA>() : BC_SEL_ID;
    const uint32_t s2 = last ? (CHROMA ? 0x07060706u : 0x03020707u) : BC_SEL_ID; // dword 2 := the run's last sample over its first two positions
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = bc_perm(ext[r][1], ext[r][0], sf);
        xa[r][0] = bc_perm(ext[r][1], ext[r][1], s0);
        xa[r][1] = bc_perm(ext[r][1], ext[r][2], s1);
        xb[r] = bc_perm(ext[r][1], ext[r][2], s2);
    }
}

// Horizontal pass of NROWS source rows -> packed columns: D[g][v] = bytes (rows 4 g .. 4 g + 3) of output value v
template <bool CHROMA, int NROWS>
BC_HD void b2_hpass(const uint32_t (&ext)[NROWS][3], const uint32_t (&xa)[NROWS][2], const uint32_t (&xb)[NROWS], bool first, uint32_t (&D)[(NROWS + 3) / 4][8]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int g = 0; g < (NROWS + 3) / 4; g++) {
        int S[8][4]; // [value][row of the group]
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int rr = 0; rr < 4; rr++) {
            const int r = 4 * g + rr;
            if (r >= NROWS) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int v = 0; v < 8; v++) S[v][rr] = 0;
                continue;
            }
            // [copy][dword]: what each variant reads as dwords 0, 1, 2, and the complements for the negative taps
            const uint32_t src[3][3] = { { ext[r][0], ext[r][1], ext[r][2] }, { ext[r][0], xa[r][0], xa[r][1] }, { ext[r][0], ext[r][1], xb[r] } };
            uint32_t neg[3][3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 3; k++)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int d = 0; d < 3; d++) neg[k][d] = ~src[k][d];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int v = 0; v < 8; v++) {
                const int k = b2_variant<CHROMA>(v);
                uint32_t acc = b2_acc0();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int d = 0; d < 3; d++) {
                    const uint32_t mp = b2_hmask<CHROMA>(v, d, false), mn = b2_hmask<CHROMA>(v, d, true);
                    if (mp != 0u) acc = bc_udot4(src[k][d], mp, acc);
                    if (mn != 0u) acc = bc_udot4(neg[k][d], mn, acc);
                }
                // output 0 of the row (chroma: both components of pair column 0): the first sample itself
                if (v < (CHROMA ? 2 : 1)) acc = first ? ((((ext[r][1] >> (8 * v)) & 255u) << 8) | 128u) : acc;
                S[v][rr] = (int)acc;
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int v = 0; v < 8; v += 2) {
            const int s8[8] = { S[v][0], S[v][1], S[v][2], S[v][3], S[v + 1][0], S[v + 1][1], S[v + 1][2], S[v + 1][3] };
            bc_pack8(s8, D[g][v], D[g][v + 1]);
        }
    }
}

// first row of output row r's vertical window, relative to the tile's first source row
constexpr int b2_vstart(int r) { return (r & 1) ? ((r - 1) >> 1) + 1 : (r >> 1); }

// Vertical pass: one output row of 8 values -> two dwords of bytes.  vsel: BC_SEL_VLAST where the row is one of the plane's last three, else BC_SEL_ID;
// copy: the row is the plane's first output row (the horizontal result of source row 0 itself: every row above it was loaded clamped).
BC_HD void b2_vrow(const uint32_t (&D)[2][8], int r, uint32_t vsel, bool copy, uint32_t &lo, uint32_t &hi) {
    int s8[8];
    const int sh = b2_vstart(r);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int v = 0; v < 8; v++) {
        uint32_t w = sh == 0 ? D[0][v] : bc_alignbyte(D[1][v], D[0][v], (uint32_t)sh);
        w = bc_perm(w, w, vsel);
        uint32_t acc = bc_udot4(w, b2_vmask(r, false), b2_acc0());
        acc = bc_udot4(~w, b2_vmask(r, true), acc);
        if (r == 0) acc = copy ? ((((w >> 8) & 255u) << 8) | 128u) : acc;
        s8[v] = (int)acc;
    }
    bc_pack8(s8, lo, hi);
}

// The whole resize of a thread tile from its extended rows (column edges fixed: b2_fix_rows).  first: the row's first thread; first_row / last_row /
// before_last_row: the tile row is the plane's first / last / last but one.
template <bool WITH_CHROMA>
BC_HD void b2_tile(const uint32_t (&ey)[B2_NYR][3], const uint32_t (&xay)[B2_NYR][2], const uint32_t (&xby)[B2_NYR], const uint32_t (&ec)[B2_NCR][3],
                   const uint32_t (&xac)[B2_NCR][2], const uint32_t (&xbc)[B2_NCR], bool first, bool first_row, bool last_row, bool before_last_row,
                   uint32_t (&ylo)[4], uint32_t (&yhi)[4], uint32_t (&clo)[2], uint32_t (&chi)[2]) {
    const uint32_t vl = last_row ? BC_SEL_VLAST : BC_SEL_ID, vl2 = (last_row || before_last_row) ? BC_SEL_VLAST : BC_SEL_ID;
    {
        uint32_t D[2][8];
        b2_hpass<false, B2_NYR>(ey, xay, xby, first, D);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int r = 0; r < 4; r++) b2_vrow(D, r, r == 0 ? BC_SEL_ID : vl, first_row, ylo[r], yhi[r]);
    }
    if (WITH_CHROMA) {
        uint32_t D[2][8];
        b2_hpass<true, B2_NCR>(ec, xac, xbc, first, D);
        b2_vrow(D, 0, vl, first_row, clo[0], chi[0]);
        b2_vrow(D, 1, vl2, false, clo[1], chi[1]);
    }
}

} // namespace tsvpp
