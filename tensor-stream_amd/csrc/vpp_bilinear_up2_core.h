// vpp_bilinear_up2_core.h -- the resize of ONE thread tile of vpp_bilinear_up2.hip (BILINEAR at the exact ratio 1 : 2 -- 540p -> 1080p, 1080p -> 4K) as
// plain C++ on arrays of dwords: the kernel calls it on registers, and tests/host/bilinear_up2_host.cpp compiles the SAME text with g++ (the hardware
// operations emulated, vpp_bicubic_r32_core.h) and runs it over whole frames against the oracle -- every mask, window and edge rule is checked on the
// CPU before a GPU ever sees it.  Product code: includes nothing from oracle/.
//
// Arithmetic (reference src/Resize.cu:5-25, 269-312).  At ratio 1/2 the coordinate (j + 0.5) / 2 - 0.5 is exactly j / 2 - 1/4: output 2 m has p = m - 1,
// w = 3/4, output 2 m + 1 has p = m, w = 1/4, on both axes.  Every weight product is a multiple of 1/16, so the reference's float sum
//     A (1 - wx)(1 - wy) + B wx (1 - wy) + C wy (1 - wx) + D wx wy      (whatever nvcc fused in it)
// is EXACT, and its (int) cast is floor(S / 16) with S = the same sum in sixteenths: per source row h = S[p] (4 - kx) + S[p + 1] kx (kx = 3 or 1) is ONE
// v_dot4_u32_u8 on the source dword with a compile-time byte mask (two where the pair straddles a dword), an output is (h_upper (4 - ky) + h_lower ky) >> 4:
// one multiply-add, the shift in the pack instruction.  The clamps of the reference (x < 0 -> x = 0, w = 0; x + 1 >= width -> the second tap reads the
// first) give exactly what edge REPLICATION gives (S[0] (1 + 3) / 4 = S[0]): the caller replicates the first / last sample into the byte before / after
// the row and loads rows above / below the plane clamped.
//
// Geometry of a thread tile: 8 output columns x 4 output rows from 4 source samples per row (ONE dword: the thread's run) extended by one dword on each
// side -- ext[r][0..2], byte e = source byte 4 q - 4 + e; used: bytes 3 .. 8 (luma: samples 4 q - 1 .. 4 q + 4), 2 .. 9 (chroma: pairs 2 q - 1 .. 2 q + 2).
// Rows: output rows 4 n .. 4 n + 3 tap source rows 2 n - 1 .. 2 n + 2 (4 rows), the tile's two chroma output rows tap chroma rows n - 1 .. n + 1 (3 rows).
#pragma once
#include "vpp_bicubic_r32_core.h"

namespace tsvpp {

constexpr int U2_NYR = 4, U2_NCR = 3; // luma / chroma source rows of a tile

// first sample of the pair output index c (0..7) taps, relative to the thread's run; its weights in quarters (first, second)
constexpr int u2_first(int c) { return (c & 1) ? (c >> 1) : (c >> 1) - 1; }
constexpr int u2_wfirst(int c) { return (c & 1) ? 3 : 1; } // even outputs: w = 3/4 -> (1/4, 3/4); odd: w = 1/4 -> (3/4, 1/4)
// byte mask of dword d of the extended row for output value v (luma: column v; chroma: component v & 1 of pair column v / 2)
template <bool CHROMA> constexpr uint32_t u2_hmask(int v, int d) {
    const int c = CHROMA ? (v >> 1) : v;
    uint32_t m = 0;
    for (int t = 0; t < 2; t++) {
        const int s = u2_first(c) + t;
        const int e = CHROMA ? 2 * s + (v & 1) + 4 : s + 4;
        const int w = t == 0 ? u2_wfirst(c) : 4 - u2_wfirst(c);
        if ((e >> 2) == d) m |= (uint32_t)w << (8 * (e & 3));
    }
    return m;
}
// the two source rows (tile rows) of output row r and the weight of the first, in quarters
constexpr int u2_row_first(int r) { return (r & 1) ? (r >> 1) + 1 : (r >> 1); }
constexpr int u2_row_wfirst(int r) { return (r & 1) ? 3 : 1; }

// Eight sums in sixteenths -> two dwords of bytes s >> 4 (s[0] in byte 0 of lo); cf. bc_pack8 (the same instruction with shift 4)
BC_HD void u2_pack8(const int (&s)[8], uint32_t &lo, uint32_t &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a, b;
    asm("s_nop 2\n\t"
        "v_ashr_pk_u8_i32 %0, %4, %5, 4\n\t"
        "v_ashr_pk_u8_i32 %1, %8, %9, 4\n\t"
        "v_lshlrev_b32 %0, 16, %0\n\t"
        "v_lshlrev_b32 %1, 16, %1\n\t"
        "v_ashr_pk_u8_i32 %0, %2, %3, 4\n\t"
        "v_ashr_pk_u8_i32 %1, %6, %7, 4"
        : "=&v"(a), "=&v"(b)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]));
    lo = a;
    hi = b;
#else
    uint32_t r[2] = { 0, 0 };
    for (int k = 0; k < 8; k++) {
        int v = s[k] >> 4;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        r[k >> 2] |= (uint32_t)v << (8 * (k & 3));
    }
    lo = r[0];
    hi = r[1];
#endif
}

// horizontal sums of one extended source row, in quarters: h[v] for the 8 output values of the thread
template <bool CHROMA> BC_HD void u2_hrow(const uint32_t (&ext)[3], uint32_t (&h)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int v = 0; v < 8; v++) {
        uint32_t acc = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int d = 0; d < 3; d++) {
            const uint32_t m = u2_hmask<CHROMA>(v, d);
            if (m != 0u) acc = bc_udot4(ext[d], m, acc);
        }
        h[v] = acc;
    }
}

// source row of tile row r (rows above / below the plane: its first / last row -- the reference's clamps, see the header)
BC_HD int u2_row(int row0, int r, int plane_rows) {
    int row = row0 + r;
    row = row < 0 ? 0 : row;
    return row > plane_rows - 1 ? plane_rows - 1 : row;
}

// The extended rows of a thread tile straight from the plane: NROWS rows from source row `row0`.  Every address read lies inside its row: the dword
// before the run is not read by the row's first thread, the dword after it not by its last (they re-read their own dword; u2_fix_rows replaces the
// bytes that matter).  Workgroups 64 threads wide take the two neighbour dwords from the adjacent lanes instead (vpp_bilinear_up2.hip).
template <int NROWS> BC_HD void u2_load_rows(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, uint32_t (&ext)[NROWS][3]) {
    const uint32_t col = 4u * (uint32_t)q, before = first ? 0u : 4u, after = last ? 0u : 4u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < NROWS; r++) {
        const uint32_t off = (uint32_t)u2_row(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<1>(plane + off, &ext[r][1]);
        bc_ld<1>(plane + (off - before), &ext[r][0]);
        bc_ld<1>(plane + (off + after), &ext[r][2]);
    }
}
// column edges: the sample before the row := its first one, the sample after it := its last one (luma: one byte; chroma: one U V pair)
template <bool CHROMA, int NROWS> BC_HD void u2_fix_rows(uint32_t (&ext)[NROWS][3], bool first, bool last) {
    const uint32_t sf = first ? bc_sel_first<CHROMA>() : BC_SEL_ID;
    const uint32_t sl = last ? (CHROMA ? 0x03020706u : 0x03020107u) : BC_SEL_ID; // ext[2] bytes 0 (0, 1) := the run's byte 3 (bytes 2, 3)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = bc_perm(ext[r][1], ext[r][0], sf);
        ext[r][2] = bc_perm(ext[r][1], ext[r][2], sl);
    }
}

// The whole resize of a thread tile from its extended rows (edges fixed): ylo / yhi[r] = the 8 luma bytes of output row r (0..3), clo / chi[rc] = U0 V0 U1 V1 |
// U2 V2 U3 V3 of chroma output row rc (0..1).
template <bool WITH_CHROMA>
BC_HD void u2_tile(const uint32_t (&ey)[U2_NYR][3], const uint32_t (&ec)[U2_NCR][3], uint32_t (&ylo)[4], uint32_t (&yhi)[4], uint32_t (&clo)[2], uint32_t (&chi)[2]) {
    {
        uint32_t h[U2_NYR][8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int r = 0; r < U2_NYR; r++) u2_hrow<false>(ey[r], h[r]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int r = 0; r < 4; r++) {
            const int a = u2_row_first(r), w = u2_row_wfirst(r);
            int s[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int v = 0; v < 8; v++) s[v] = (int)(h[a][v] * (uint32_t)w + h[a + 1][v] * (uint32_t)(4 - w));
            u2_pack8(s, ylo[r], yhi[r]);
        }
    }
    if (WITH_CHROMA) {
        uint32_t h[U2_NCR][8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int r = 0; r < U2_NCR; r++) u2_hrow<true>(ec[r], h[r]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int rc = 0; rc < 2; rc++) {
            const int a = u2_row_first(rc), w = u2_row_wfirst(rc);
            int s[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int v = 0; v < 8; v++) s[v] = (int)(h[a][v] * (uint32_t)w + h[a + 1][v] * (uint32_t)(4 - w));
            u2_pack8(s, clo[rc], chi[rc]);
        }
    }
}

} // namespace tsvpp
