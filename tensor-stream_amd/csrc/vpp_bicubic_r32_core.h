// vpp_bicubic_r32_core.h -- the resize of ONE thread tile of vpp_bicubic_r32.hip (BICUBIC at the exact ratios 3 : 2 and 2 : 1) as plain
// C++ on arrays of dwords: the kernel calls it on registers, and tests/host/bicubic_r32_host.cpp compiles the SAME text with g++ (the four
// hardware operations below emulated) and runs it over whole frames against the oracle -- every mask, window and edge rule of the kernel is
// checked on the CPU before a GPU ever sees it.  Product code: includes nothing from oracle/.
//
// Arithmetic (reference src/Resize.cu:27-91, 311-357: Keys' cubic a = -0.75 in fp64, every 4-tap sum rounded half away from zero and clamped to a
// byte, horizontally first, then down the column).  At ratio 1.5 the coordinate (j + 0.5) * 1.5 - 0.5 is exactly 1.5 j + 0.25: output 2k has
// p = 3k, w = 1/4, output 2k + 1 has p = 3k + 1, w = 3/4; at ratio 2 it is 2 j + 0.5: p = 2 j, w = 1/2.  For these weights the coefficients are
// integers over 256:  w = 1/4: (-27, 225, 67, -9),  w = 3/4: (-9, 67, 225, -27),  w = 1/2: (-24, 152, 152, -24)  -- magnitudes that fit a BYTE, so a
// 4-tap sum is v_dot4_u32_u8 on the source dwords themselves with compile-time byte masks: the positive taps on the data, the negative taps on
// the COMPLEMENTED data (-C p = C (255 - p) - 255 C; the constant part sits in the accumulator's start value), all in one accumulator chain:
//     acc = 128 - 255 * sum|C_neg|;  acc = dot4(d, mask_pos, acc);  acc = dot4(~d, mask_neg, acc);   value = clamp(acc >> 8, 0, 255)
// (exact: every product and sum is an integer; acc >> 8 with the +128 is round-half-up, and a negative sum clamps to 0 either way).
// No LDS, no barrier, no coordinate arithmetic, no tables: tap positions are compile-time byte positions of a dword-aligned run.
//
// Geometry of a thread tile: 8 output columns x 4 output rows.  Source run of the thread: RUN = 4 * P2 bytes per row (P2 = 3: 12, P2 = 4: 16),
// extended by one dword on each side (taps reach one sample before and one after the run): NDW = P2 + 2 dwords per row, byte e of the extended
// run = source byte RUN * q - 4 + e.  Rows: output rows 4 n .. 4 n + 3 tap source rows R0 .. R0 + NYR - 1 with R0 = 2 P2 n - 1, NYR = 8 (3 : 2) or
// 10 (2 : 1); the tile's two chroma output rows tap chroma rows P2 n - 1 .. P2 n - 1 + NCR - 1, NCR = 5 or 6.
// Edge rule (src/Resize.cu:32-43): at p = 0 the -1 tap reads p itself; where p + 2 would leave the plane BOTH the +1 and +2 taps read p.  At these
// ratios that happens exactly at the first and the last output index of an axis.  Low edge: the caller replicates the first sample into the byte
// before the run (columns) / loads row 0 twice (rows): the sample before the plane is tapped by the first output only.  High edge: the last
// output gets its own copy of the last two dwords with the centre tap replicated over the +1 / +2 positions (xr below; the caller builds it with
// one v_perm per dword whose selector is the identity everywhere else), rows alike on the packed window.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BC_HD __host__ __device__ __forceinline__
#else
#define BC_HD inline
#endif

namespace tsvpp {

BC_HD uint32_t bc_udot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 255u) * ((b >> (8 * k)) & 255u);
    return c;
#endif
}
// bytes [sh, sh + 4) of the 8-byte value hi:lo
BC_HD uint32_t bc_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh));
#endif
}
// v_perm_b32: result byte k = byte sel[k] of the 8-byte value a:b (0-3 = b, 4-7 = a)
BC_HD uint32_t bc_perm(uint32_t a, uint32_t b, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(a, b, sel);
#else
    const uint64_t v = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * k)) & 7u))) & 255u) << (8 * k);
    return r;
#endif
}
// Eight biased sums -> two dwords of bytes clamp(s >> 8, 0, 255) (s[0] in byte 0 of lo).  gfx950's V_ASHR_PK_U8_I32 shifts, saturates and packs
// two values into the LOW half of its destination and preserves the upper half (vpp_bicubic_int.hip, round_clamp_pack4) -- used on purpose: the
// upper pair is packed first and shifted up, the lower pair lands below it.  ONE s_nop for the whole block: a VALU read of a dot result needs
// three wait states and the compiler does not look into inline assembly.
BC_HD void bc_pack8(const int (&s)[8], uint32_t &lo, uint32_t &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a, b;
    asm("s_nop 2\n\t"
        "v_ashr_pk_u8_i32 %0, %4, %5, 8\n\t"
        "v_ashr_pk_u8_i32 %1, %8, %9, 8\n\t"
        "v_lshlrev_b32 %0, 16, %0\n\t"
        "v_lshlrev_b32 %1, 16, %1\n\t"
        "v_ashr_pk_u8_i32 %0, %2, %3, 8\n\t"
        "v_ashr_pk_u8_i32 %1, %6, %7, 8"
        : "=&v"(a), "=&v"(b)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]));
    lo = a;
    hi = b;
#else
    uint32_t r[2] = { 0, 0 };
    for (int k = 0; k < 8; k++) {
        int v = s[k] >> 8;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        r[k >> 2] |= (uint32_t)v << (8 * (k & 3));
    }
    lo = r[0];
    hi = r[1];
#endif
}

// window start of output index c (0..7) of the thread along an axis, in samples relative to the thread's run
template <int P2> constexpr int bc_ws(int c) { return P2 == 4 ? 2 * c - 1 : 3 * (c >> 1) + (c & 1) - 1; }
// 256 x coefficient of tap t of output index c
template <int P2> constexpr int bc_coef(int c, int t) {
    return P2 == 4 ? ((t == 0 || t == 3) ? -24 : 152)
                   : ((c & 1) ? (t == 0 ? -9 : t == 1 ? 67 : t == 2 ? 225 : -27) : (t == 0 ? -27 : t == 1 ? 225 : t == 2 ? 67 : -9));
}
// start value of the accumulator chain: the rounding bias and the constant part of the complemented negative taps
template <int P2> constexpr uint32_t bc_acc0() { return (uint32_t)(128 - 255 * (P2 == 4 ? 48 : 36)); }
// byte mask of dword d of the extended run for output value v of a row: luma value v = column v; chroma value v = component (v & 1) of pair
// column v / 2 (bytes U V U V ...: sample s of the pair grid = bytes 2 s, 2 s + 1).  NEG: magnitudes of the negative coefficients (applied to ~d).
template <int P2, bool CHROMA> constexpr uint32_t bc_hmask(int v, int d, bool neg) {
    const int c = CHROMA ? (v >> 1) : v;
    uint32_t m = 0;
    for (int t = 0; t < 4; t++) {
        const int e = CHROMA ? 2 * (bc_ws<P2>(c) + t) + (v & 1) + 4 : bc_ws<P2>(c) + t + 4;
        const int C = bc_coef<P2>(c, t);
        if ((e >> 2) == d && (neg ? C < 0 : C > 0)) m |= (uint32_t)(neg ? -C : C) << (8 * (e & 3));
    }
    return m;
}
// byte mask of a packed vertical window (tap t in byte t) for output row r of the tile
template <int P2> constexpr uint32_t bc_vmask(int r, bool neg) {
    uint32_t m = 0;
    for (int t = 0; t < 4; t++) {
        const int C = bc_coef<P2>(r, t);
        if (neg ? C < 0 : C > 0) m |= (uint32_t)(neg ? -C : C) << (8 * t);
    }
    return m;
}

template <int P2> struct BcGeom {
    static constexpr int NDW = P2 + 2;            // dwords of an extended row
    static constexpr int NYR = P2 == 4 ? 10 : 8;  // luma source rows of a tile
    static constexpr int NCR = P2 == 4 ? 6 : 5;   // chroma source rows of a tile
    static constexpr int RUN = 4 * P2;            // source bytes of the thread's run per row
};

// v_perm selectors of the edge rules (identity = 0x03020100 on the SECOND operand).
// Low column edge: ext[0] = perm(ext[1], ext[0], sel): the byte(s) before the run := the run's first sample.
template <bool CHROMA> constexpr uint32_t bc_sel_first() { return CHROMA ? 0x05040100u : 0x04020100u; }
// High column edge, the last output's own copy of the last two dwords: xr[0] = perm(ext[P2], ext[P2], sel0) -- the centre tap (luma: byte 2 of the
// run's last dword; chroma: its bytes 0, 1) over the +1 position --, xr[1] = perm(ext[P2], ext[P2 + 1], sel1) -- over the +2 position.
template <bool CHROMA> constexpr uint32_t bc_sel_last0() { return CHROMA ? 0x01000100u : 0x02020100u; }
template <bool CHROMA> constexpr uint32_t bc_sel_last1() { return CHROMA ? 0x03020504u : 0x03020106u; }
constexpr uint32_t BC_SEL_ID = 0x03020100u;
// High row edge on a packed vertical window: taps +1, +2 := the centre (byte 1)
constexpr uint32_t BC_SEL_VLAST = 0x01010100u;

// Horizontal pass of NROWS source rows -> packed columns: D[g][v] = bytes (rows 4 g .. 4 g + 3) of output value v.  ext[r] = extended run of row r,
// xr[r] = the last output's copy of dwords P2, P2 + 1 (== ext[r][P2], ext[r][P2 + 1] away from the right edge).
template <int P2, bool CHROMA, int NROWS>
BC_HD void bc_hpass(const uint32_t (&ext)[NROWS][P2 + 2], const uint32_t (&xr)[NROWS][2], uint32_t (&D)[(NROWS + 3) / 4][8]) {
    constexpr int NDW = P2 + 2;
    constexpr int LASTV = CHROMA ? 6 : 7; // first output value that belongs to the last output index (chroma: both components of pair column 3)
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
            uint32_t nd[NDW], nx[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int d = 0; d < NDW; d++) nd[d] = ~ext[r][d];
            nx[0] = ~xr[r][0];
            nx[1] = ~xr[r][1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int v = 0; v < 8; v++) {
                uint32_t acc = bc_acc0<P2>();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int d = 0; d < NDW; d++) {
                    const uint32_t mp = bc_hmask<P2, CHROMA>(v, d, false), mn = bc_hmask<P2, CHROMA>(v, d, true);
                    const bool own = v >= LASTV && d >= P2; // the last output's own copy
                    if (mp != 0u) acc = bc_udot4(own ? xr[r][d - P2] : ext[r][d], mp, acc);
                    if (mn != 0u) acc = bc_udot4(own ? nx[d - P2] : nd[d], mn, acc);
                }
                S[v][rr] = (int)acc;
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int v = 0; v < 8; v += 2) { // two columns per pack block: (4 rows of value v, 4 rows of value v + 1)
            const int s8[8] = { S[v][0], S[v][1], S[v][2], S[v][3], S[v + 1][0], S[v + 1][1], S[v + 1][2], S[v + 1][3] };
            bc_pack8(s8, D[g][v], D[g][v + 1]);
        }
    }
}

// Packed vertical window (tap t in byte t) of output row r (luma: 0..3; chroma: 0..1 -- its windows are the luma windows of rows 0, 1) from the
// packed columns of one output value.
template <int P2, int NG> BC_HD uint32_t bc_vwindow(const uint32_t (&D)[NG][8], int v, int r) {
    // first row of the window relative to the tile's first source row: 3 : 2 -> 0, 1, 3, 4;  2 : 1 -> 0, 2, 4, 6
    const int start = P2 == 4 ? 2 * r : 3 * (r >> 1) + (r & 1);
    const int g = start >> 2, sh = start & 3;
    if (sh == 0) return D[g][v];
    return bc_alignbyte(D[g + 1 < NG ? g + 1 : g][v], D[g][v], (uint32_t)sh);
}

// Vertical pass: one output row of 8 values (luma columns, or U0 V0 .. U3 V3) -> two dwords of bytes.  edge_row: this output row can be the plane's
// last one (a compile-time fact after unrolling: row 3 / chroma row 1); vsel then is BC_SEL_VLAST in the plane's last tile row, else BC_SEL_ID.
template <int P2, int NG> BC_HD void bc_vrow(const uint32_t (&D)[NG][8], int r, bool edge_row, uint32_t vsel, uint32_t &lo, uint32_t &hi) {
    int s8[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int v = 0; v < 8; v++) {
        uint32_t w = bc_vwindow<P2, NG>(D, v, r);
        if (edge_row) w = bc_perm(w, w, vsel);
        uint32_t acc = bc_udot4(w, bc_vmask<P2>(r, false), bc_acc0<P2>());
        acc = bc_udot4(~w, bc_vmask<P2>(r, true), acc);
        s8[v] = (int)acc;
    }
    bc_pack8(s8, lo, hi);
}

// source row of tile row r: only the first row of a tile can lie above the plane, only its last rows below it
template <int NROWS> BC_HD int bc_row(int row0, int r, int plane_rows) {
    int row = row0 + r;
    if (r == 0) row = row < 0 ? 0 : row;
    if (r >= NROWS - 2) row = row > plane_rows - 1 ? plane_rows - 1 : row;
    return row;
}

#ifdef BC_HOST_BOUNDS
static const uint8_t *bc_host_lo = nullptr, *bc_host_hi = nullptr;
static long bc_host_oob = 0;
#endif
// N dwords from a 4-byte aligned address
template <int N> BC_HD void bc_ld(const uint8_t *p, uint32_t *dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (N == 1) {
        dst[0] = *(const uint32_t *)p;
    } else if constexpr (N == 2) {
        typedef uint32_t v2 __attribute__((ext_vector_type(2), aligned(4)));
        const v2 v = *(const v2 *)p;
        dst[0] = v.x; dst[1] = v.y;
    } else if constexpr (N == 3) {
        typedef uint32_t v3 __attribute__((ext_vector_type(3), aligned(4)));
        const v3 v = *(const v3 *)p;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
    } else {
        typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(4)));
        const v4 v = *(const v4 *)p;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
#else
#ifdef BC_HOST_BOUNDS // host test build: every byte read must lie inside the plane the harness announced
    if (p < bc_host_lo || p + 4 * N > bc_host_hi) bc_host_oob++;
#endif
    for (int k = 0; k < N; k++) {
        const uint8_t *b = p + 4 * k;
        dst[k] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    }
#endif
}

// The extended rows of a thread tile straight from the plane: NROWS rows from source row `row0` (-1 in the first tile row: clamped, i.e. row 0
// twice -- the low row edge rule), rows past the plane clamp onto its last row (their samples carry replicated data below).  Every address read
// lies inside its row: the dword before the run is not read by the row's first thread, the dword after it not by its last (they re-read their
// own first / last dword; the value is replaced / never selected).  Workgroups 64 threads wide take the two neighbour dwords from the adjacent
// lanes instead of loading them (vpp_bicubic_r32.hip, bcr_load_rows).
template <int P2, int NROWS>
BC_HD void bc_load_rows(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, uint32_t (&ext)[NROWS][P2 + 2]) {
    constexpr int RUN = 4 * P2;
    // 32-bit byte offsets from the plane's (uniform) base: the loads take the SGPR-base + VGPR-offset form (planes are far below 4 GiB)
    const uint32_t col = (uint32_t)(RUN * q), before = first ? 0u : 4u, after = last ? (uint32_t)(RUN - 4) : (uint32_t)RUN;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < NROWS; r++) {
        const uint32_t off = (uint32_t)bc_row<NROWS>(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<P2>(plane + off, &ext[r][1]);
        bc_ld<1>(plane + (off - before), &ext[r][0]);
        bc_ld<1>(plane + (off + after), &ext[r][P2 + 1]);
    }
}
// ... and the column edge rules on them (see the header comment): in place for the low edge, the last output's own copy xr for the high edge.
template <int P2, bool CHROMA, int NROWS>
BC_HD void bc_fix_rows(uint32_t (&ext)[NROWS][P2 + 2], uint32_t (&xr)[NROWS][2], bool first, bool last) {
    const uint32_t sf = first ? bc_sel_first<CHROMA>() : BC_SEL_ID, s0 = last ? bc_sel_last0<CHROMA>() : BC_SEL_ID, s1 = last ? bc_sel_last1<CHROMA>() : BC_SEL_ID;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = bc_perm(ext[r][1], ext[r][0], sf);
        xr[r][0] = bc_perm(ext[r][P2], ext[r][P2], s0);
        xr[r][1] = bc_perm(ext[r][P2], ext[r][P2 + 1], s1);
    }
}

// The whole resize of a thread tile from its extended rows: ylo/yhi[r] = the 8 luma bytes of output row r (0..3), clo/chi[rc] = U0 V0 U1 V1 | U2 V2 U3
// V3 of chroma output row rc (0..1).  last_row: the tile holds the plane's last output rows.
template <int P2, bool WITH_CHROMA>
BC_HD void bc_tile(const uint32_t (&ey)[BcGeom<P2>::NYR][P2 + 2], const uint32_t (&xy)[BcGeom<P2>::NYR][2], const uint32_t (&ec)[BcGeom<P2>::NCR][P2 + 2],
                   const uint32_t (&xc)[BcGeom<P2>::NCR][2], bool last_row, uint32_t (&ylo)[4], uint32_t (&yhi)[4], uint32_t (&clo)[2], uint32_t (&chi)[2]) {
    constexpr int NYR = BcGeom<P2>::NYR, NCR = BcGeom<P2>::NCR, NGY = (NYR + 3) / 4, NGC = (NCR + 3) / 4;
    const uint32_t vlast = last_row ? BC_SEL_VLAST : BC_SEL_ID;
    {
        uint32_t D[NGY][8];
        bc_hpass<P2, false, NYR>(ey, xy, D);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int r = 0; r < 4; r++) bc_vrow<P2, NGY>(D, r, r == 3, vlast, ylo[r], yhi[r]);
    }
    if (WITH_CHROMA) {
        uint32_t D[NGC][8];
        bc_hpass<P2, true, NCR>(ec, xc, D);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int rc = 0; rc < 2; rc++) bc_vrow<P2, NGC>(D, rc, rc == 1, vlast, clo[rc], chi[rc]);
    }
}

} // namespace tsvpp
