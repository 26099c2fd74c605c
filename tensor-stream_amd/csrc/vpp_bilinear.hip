// vpp_bilinear.hip -- the 2x2-tap kernel family (BILINEAR and the AREA up-scale variant; the headline kernel) in its own
// translation unit, compiled with LDS accesses of unknown alignment split into byte reads (target feature `no-unaligned-access-mode`,
// applied to this file's functions by the pragma below -- device pass only).  Measured on MI355X: a ds_read_u16 / ds_read_b32 whose address is not naturally aligned is
// executed lane by lane (~64 cycles per wave instruction); at ratio 1.5 half of all tap addresses are odd, and the first
// version of the integer thread tile, which read each horizontal tap pair as one 16-bit word, ran at 195 us per launch
// whatever the output type (LDS-bound) against 154 us for the float tile (profiles/r02_bilinear_int_ab.txt).
#include "vpp_device.h"

// (device pass only: the host compiler does not know the feature)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-unaligned-access-mode"))), apply_to = function)
#endif

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#pragma clang fp contract(off)

namespace tsvpp {

// ----------------------------------------------------------------------------------------------
// Fast kernel for the 2x2-tap family (BILINEAR and the AREA up-scale variant).
//   * per-workgroup coordinate tables in LDS: every output column / row of the tile gets its
//     source offset and weight computed ONCE (one lane each) instead of once per thread;
//   * one LDS address per tap row, the horizontal neighbours at immediate offsets;
//   * blend and colour arithmetic on float pairs (packed VALU).
// Right-edge rule of the reference (x + 1 >= width -> B = A, src/Resize.cu:7-8): the column just
// past the plane is written into LDS as a copy of the last one, so B is always "the next byte".

template <bool AREAUP>
__device__ __forceinline__ void axis2(int idx, float ratio, int limit, int &p, float &w) {
    if constexpr (AREAUP) areaup_axis(idx, ratio, p, w);
    else bilinear_axis(idx, ratio, limit, p, w);
}

// Blend + colour-convert + store one thread tile of the 2x2-tap family from the staged LDS planes
// and the coordinate tables.
template <int OUT>
__device__ __forceinline__ void bilinear_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const XEntry *xtab,
                                                     const XEntry *cxtab, const YEntry *ytab, const YEntry *cytab, int lx, int ly,
                                                     typename OutT<OUT>::type *out, int i0, int j0) {
    // this thread's table entries: 4 luma columns, 2 luma rows, 2 chroma columns, 1 chroma row
    XEntry xe[PXW], cxe[2];
    YEntry ye[PXH], cye;
    {
        const uint4 a = *(const uint4 *)(xtab + lx * PXW), b = *(const uint4 *)(xtab + lx * PXW + 2);
        xe[0] = XEntry{ (int)a.x, __uint_as_float(a.y) };
        xe[1] = XEntry{ (int)a.z, __uint_as_float(a.w) };
        xe[2] = XEntry{ (int)b.x, __uint_as_float(b.y) };
        xe[3] = XEntry{ (int)b.z, __uint_as_float(b.w) };
        const uint4 c = *(const uint4 *)(cxtab + lx * 2);
        cxe[0] = XEntry{ (int)c.x, __uint_as_float(c.y) };
        cxe[1] = XEntry{ (int)c.z, __uint_as_float(c.w) };
        const uint4 y0 = *(const uint4 *)(ytab + ly * PXH), y1 = *(const uint4 *)(ytab + ly * PXH + 1);
        ye[0] = YEntry{ (int)y0.x, (int)y0.y, __uint_as_float(y0.z), 0 };
        ye[1] = YEntry{ (int)y1.x, (int)y1.y, __uint_as_float(y1.z), 0 };
        const uint4 cy = *(const uint4 *)(cytab + ly);
        cye = YEntry{ (int)cy.x, (int)cy.y, __uint_as_float(cy.z), 0 };
    }

    // chroma: (U, V) of one block blended as a float pair
    float Uf[2], Vf[2], Yf[PXH][PXW];
    {
        const f2 wy = { cye.w, cye.w }, omy = (f2){ 1.0f, 1.0f } - wy;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            // one address per row, taps at immediate offsets 0..3 (U0 V0 U1 V1); unaligned wide LDS
            // reads are serialised by the hardware, so the taps are byte reads
            const uint8_t *top = lds_uv + cye.top + cxe[c].off, *bot = lds_uv + cye.bot + cxe[c].off;
            const f2 A = { (float)top[0], (float)top[1] }, B = { (float)top[2], (float)top[3] };
            const f2 C = { (float)bot[0], (float)bot[1] }, D = { (float)bot[2], (float)bot[3] };
            const f2 wx = { cxe[c].w, cxe[c].w }, omx = (f2){ 1.0f, 1.0f } - wx;
            const f2 sum = bilerp2(A, B, C, D, wx, omx, wy, omy);
            Uf[c] = __builtin_truncf(sum.x);
            Vf[c] = __builtin_truncf(sum.y);
        }
    }
    // luma: horizontally adjacent pixel pairs
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const f2 wy = { ye[r].w, ye[r].w }, omy = (f2){ 1.0f, 1.0f } - wy;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint8_t *t0 = lds_y + ye[r].top + xe[2 * p].off, *t1 = lds_y + ye[r].top + xe[2 * p + 1].off;
            const uint8_t *b0 = lds_y + ye[r].bot + xe[2 * p].off, *b1 = lds_y + ye[r].bot + xe[2 * p + 1].off;
            const f2 A = { (float)t0[0], (float)t1[0] }, B = { (float)t0[1], (float)t1[1] };
            const f2 C = { (float)b0[0], (float)b1[0] }, D = { (float)b0[1], (float)b1[1] };
            const f2 wx = { xe[2 * p].w, xe[2 * p + 1].w }, omx = (f2){ 1.0f, 1.0f } - wx;
            const f2 sum = bilerp2(A, B, C, D, wx, omx, wy, omy);
            Yf[r][2 * p] = __builtin_truncf(sum.x);
            Yf[r][2 * p + 1] = __builtin_truncf(sum.y);
        }
    }
    color_store_tile<OUT, true, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
}

// Integer form of the thread tile for requests whose weights are all multiples of 1/16 (LaunchDesc::bil_int; ratios 1.5,
// 2, 2.5, 4, 0.5, 1.25 ...: 1080p -> 720p, 4K -> 1080p, 2x up-scales).  The reference's blend
//     (int)( A (1-wx)(1-wy) + B wx (1-wy) + C wy (1-wx) + D (wx wy) )        (src/Resize.cu:17-23)
// is then exact in fp32 -- every product has at most 16 significant bits -- so it equals the integer
//     ( (A wx0 + B wx1) wy0 + (C wx0 + D wx1) wy1 ) >> 8,   wx0 = 16 - 16 wx, wx1 = 16 wx, likewise wy.
// Per value: the two horizontal taps are two LDS byte reads merged by one v_perm_b32 into (A, 0, B, 0) and fed to
// v_dot4_u32_u8 with the column's packed weights; the vertical pair is one v_dot2_u32_u16 on (top | bottom << 16); the
// final shift and the conversion for the colour stage are one v_cvt_f32_ubyte1.  7 VALU instructions against ~11 (4 byte ->
// float conversions + 7 packed multiplies / adds per value).  Table entries: XEntry::w / YEntry::w carry the packed integer
// weights (wx0 | wx1 << 16).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t tap_pair(const uint8_t *p, int second) { // (p[0], 0, p[second], 0)
    return __builtin_bit_cast(uint32_t, (u16x2){ (unsigned short)p[0], (unsigned short)p[second] });
}
// top * wy0 + bot * wy1 (both horizontal sums are < 2^12, the weights <= 16): one v_dot2_u32_u16 on (top | bot << 16) -- the
// 32-bit integer multiplies (v_mul_lo_u32, v_mad_u64_u32) run at a quarter of the VALU rate
__device__ __forceinline__ uint32_t vpair(uint32_t top, uint32_t bot, uint32_t wyp) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, top | (bot << 16)), __builtin_bit_cast(u16x2, wyp), 0u, false);
}
template <int OUT>
__device__ __forceinline__ void bilinear_int_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const XEntry *xtab,
                                                         const XEntry *cxtab, const YEntry *ytab, const YEntry *cytab, int lx, int ly,
                                                         typename OutT<OUT>::type *out, int i0, int j0) {
    int xo[PXW], cxo[2];
    uint32_t xw[PXW], cxw[2];
    {
        const uint4 a = *(const uint4 *)(xtab + lx * PXW), b = *(const uint4 *)(xtab + lx * PXW + 2);
        xo[0] = (int)a.x; xw[0] = a.y; xo[1] = (int)a.z; xw[1] = a.w;
        xo[2] = (int)b.x; xw[2] = b.y; xo[3] = (int)b.z; xw[3] = b.w;
        const uint4 c = *(const uint4 *)(cxtab + lx * 2);
        cxo[0] = (int)c.x; cxw[0] = c.y; cxo[1] = (int)c.z; cxw[1] = c.w;
    }
    float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) {
        const uint4 cy = *(const uint4 *)(cytab + ly);
        const uint32_t wyp = cy.z;
#pragma unroll
        for (int c = 0; c < 2; c++) { // bytes U0 V0 U1 V1 of each row: U taps at +0 / +2, V taps at +1 / +3
            const uint8_t *top = lds_uv + (int)cy.x + cxo[c], *bot = lds_uv + (int)cy.y + cxo[c];
            const uint32_t tu = __builtin_amdgcn_udot4(tap_pair(top, 2), cxw[c], 0u, false), tv = __builtin_amdgcn_udot4(tap_pair(top + 1, 2), cxw[c], 0u, false);
            const uint32_t bu = __builtin_amdgcn_udot4(tap_pair(bot, 2), cxw[c], 0u, false), bv = __builtin_amdgcn_udot4(tap_pair(bot + 1, 2), cxw[c], 0u, false);
            const uint32_t su = vpair(tu, bu, wyp), sv = vpair(tv, bv, wyp);
            Uf[c] = (float)((su >> 8) & 255u);
            Vf[c] = (float)((sv >> 8) & 255u);
        }
    }
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const uint4 ye = *(const uint4 *)(ytab + ly * PXH + r);
        const uint32_t wyp = ye.z;
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            const uint32_t tt = __builtin_amdgcn_udot4(tap_pair(lds_y + (int)ye.x + xo[c], 1), xw[c], 0u, false);
            const uint32_t bb = __builtin_amdgcn_udot4(tap_pair(lds_y + (int)ye.y + xo[c], 1), xw[c], 0u, false);
            const uint32_t sv = vpair(tt, bb, wyp);
            Yf[r][c] = (float)((sv >> 8) & 255u);
        }
    }
    color_store_tile<OUT, true, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
}


// Window form of the integer thread tile (LaunchDesc::bil_int == 2: dyadic weights and a horizontal ratio <= 2).  The byte form
// above issues 48 ds_read_u8 per thread tile, and at a lane stride of 6 bytes (ratio 1.5) every one of them is a two-way bank
// conflict: the LDS pipe, not the VALU, set the pace of the uint8 outputs.  Here each source row of the thread tile is read ONCE
// as three ALIGNED dwords (ds_read2_b32 + ds_read_b32), two v_alignbyte_b32 shift them into the 8 bytes that start at the
// thread's first tap, and one v_perm_b32 per output column -- its selector depends on the column only, so it is built once per
// thread -- picks (A, 0, B, 0) for v_dot4_u32_u8.  With ratio <= 2 the four columns' taps span at most 8 bytes (luma: column 3
// starts <= 6 bytes after column 0; chroma: U0 V0 U1 V1 of the second block end <= 8 bytes after the first block's U0).
// LDS instructions per thread tile: 18 instead of 60; the arithmetic after the perm is unchanged (same integers as the byte form).
struct TapWindow {
    uint32_t lo, hi;
};
__device__ __forceinline__ TapWindow window8(const uint8_t *lds, int a) {
    const uint32_t *q = (const uint32_t *)(lds + (a & ~3));
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    // v_alignbyte_b32 shifts by 8 * (operand & 3) (probed, tools/dbg_cvt.hip): the address itself is the shift operand
    return TapWindow{ __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)a), __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)a) };
}
// this thread's columns: first tap offset, per-column selectors and packed weights (constant over the thread's row pairs)
struct WinColumns {
    int xo0, cxo0;
    uint32_t sel[PXW], w[PXW], csel[2], cw[2];
};
__device__ __forceinline__ WinColumns win_columns(const XEntry *xtab, const XEntry *cxtab, int lx) {
    WinColumns k;
    const uint4 a = *(const uint4 *)(xtab + lx * PXW), b = *(const uint4 *)(xtab + lx * PXW + 2);
    const uint4 c = *(const uint4 *)(cxtab + lx * 2);
    k.xo0 = (int)a.x;
    k.cxo0 = (int)c.x;
    const uint32_t rel[PXW] = { 0u, a.z - a.x, b.x - a.x, b.z - a.x };
    k.w[0] = a.y; k.w[1] = a.w; k.w[2] = b.y; k.w[3] = b.w;
#pragma unroll
    for (int i = 0; i < PXW; i++) k.sel[i] = rel[i] * 0x00010001u + 0x0c010c00u; // bytes (rel, 0x0c -> zero, rel + 1, zero)
    k.csel[0] = 0x0c020c00u;                                                     // U taps at +0 / +2 (V: every selector + 1)
    k.csel[1] = (c.z - c.x) * 0x00010001u + 0x0c020c00u;
    k.cw[0] = c.y; k.cw[1] = c.w;
    return k;
}
// The rows of one thread tile: LDS byte offsets of the top / bottom tap rows (without the column part, which is WinColumns::xo0 /
// cxo0) and the row weights (packed integers or float bits), for the two luma rows and the one chroma row.  Filled from the
// workgroup's LDS coordinate tables (vpp_bilinear_kernel) or from the host-built geometry tables (vpp_bilinear_geo_kernel).
struct RowPairGeo {
    int top[PXH], bot[PXH];
    uint32_t wy[PXH];
    int ctop, cbot;
    uint32_t cwy;
};
__device__ __forceinline__ RowPairGeo rows_from_lds(const YEntry *ytab, const YEntry *cytab, int ly) {
    RowPairGeo g;
    const uint4 cy = *(const uint4 *)(cytab + ly);
    g.ctop = (int)cy.x; g.cbot = (int)cy.y; g.cwy = cy.z;
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const uint4 ye = *(const uint4 *)(ytab + ly * PXH + r);
        g.top[r] = (int)ye.x; g.bot[r] = (int)ye.y; g.wy[r] = ye.z;
    }
    return g;
}
// TAP22: the weights in the tables are those of LaunchDesc::tap22 (AREA down-scale at 3 : 2 or 2 : 1) and the value is the weighted sum divided
// by its weight total, truncated (area_quot with the one reciprocal of the request: the dyadic AREA kernels' exact division) instead of >> 8.
template <bool TAP22> __device__ __forceinline__ float win_value(uint32_t sum, float rcp) {
    if constexpr (TAP22) return area_quot(sum, 0, 0, rcp);
    else return (float)((sum >> 8) & 255u);
}
template <int OUT, bool TAP22 = false>
__device__ __forceinline__ void bilinear_win_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const WinColumns &k,
                                                         const RowPairGeo &g, typename OutT<OUT>::type *out, int i0, int j0) {
    float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) {
        const TapWindow T = window8(lds_uv, g.ctop + k.cxo0), B = window8(lds_uv, g.cbot + k.cxo0);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t su = k.csel[c], sv = su + 0x00010001u;
            const uint32_t tu = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, su), k.cw[c], 0u, false);
            const uint32_t tv = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, sv), k.cw[c], 0u, false);
            const uint32_t bu = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, su), k.cw[c], 0u, false);
            const uint32_t bv = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, sv), k.cw[c], 0u, false);
            Uf[c] = win_value<TAP22>(vpair(tu, bu, g.cwy), d.area_rcp);
            Vf[c] = win_value<TAP22>(vpair(tv, bv, g.cwy), d.area_rcp);
        }
    }
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const TapWindow T = window8(lds_y, g.top[r] + k.xo0), B = window8(lds_y, g.bot[r] + k.xo0);
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            const uint32_t tt = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, k.sel[c]), k.w[c], 0u, false);
            const uint32_t bb = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, k.sel[c]), k.w[c], 0u, false);
            Yf[r][c] = win_value<TAP22>(vpair(tt, bb, g.wy[r]), d.area_rcp);
        }
    }
    color_store_tile<OUT, true, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
}

// Window form of the FLOAT thread tile (LaunchDesc::bil_win: any weights, horizontal ratio <= 2): the same row windows and
// per-column selectors as the integer window tile; v_perm_b32 yields (A, 0, B, 0) and the two taps are converted by
// v_cvt_f32_ubyte0 / v_cvt_f32_ubyte2.  The blend is bilerp2 -- the float tile's arithmetic, unchanged.
template <int OUT>
__device__ __forceinline__ void bilinear_winf_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const WinColumns &k,
                                                          const RowPairGeo &g, typename OutT<OUT>::type *out, int i0, int j0) {
    float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) {
        const TapWindow T = window8(lds_uv, g.ctop + k.cxo0), B = window8(lds_uv, g.cbot + k.cxo0);
        const float wyf = __uint_as_float(g.cwy);
        const f2 wy = { wyf, wyf }, omy = (f2){ 1.0f, 1.0f } - wy;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t su = k.csel[c], sv = su + 0x00010001u;
            const uint32_t tu = __builtin_amdgcn_perm(T.hi, T.lo, su), tv = __builtin_amdgcn_perm(T.hi, T.lo, sv);
            const uint32_t bu = __builtin_amdgcn_perm(B.hi, B.lo, su), bv = __builtin_amdgcn_perm(B.hi, B.lo, sv);
            const f2 A = { (float)(tu & 255u), (float)(tv & 255u) }, Bq = { (float)((tu >> 16) & 255u), (float)((tv >> 16) & 255u) };
            const f2 C = { (float)(bu & 255u), (float)(bv & 255u) }, D = { (float)((bu >> 16) & 255u), (float)((bv >> 16) & 255u) };
            const float wxf = __uint_as_float(k.cw[c]);
            const f2 wx = { wxf, wxf }, omx = (f2){ 1.0f, 1.0f } - wx;
            const f2 sum = bilerp2(A, Bq, C, D, wx, omx, wy, omy);
            Uf[c] = __builtin_truncf(sum.x);
            Vf[c] = __builtin_truncf(sum.y);
        }
    }
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const TapWindow T = window8(lds_y, g.top[r] + k.xo0), B = window8(lds_y, g.bot[r] + k.xo0);
        const float wyf = __uint_as_float(g.wy[r]);
        const f2 wy = { wyf, wyf }, omy = (f2){ 1.0f, 1.0f } - wy;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint32_t t0 = __builtin_amdgcn_perm(T.hi, T.lo, k.sel[2 * p]), t1 = __builtin_amdgcn_perm(T.hi, T.lo, k.sel[2 * p + 1]);
            const uint32_t b0 = __builtin_amdgcn_perm(B.hi, B.lo, k.sel[2 * p]), b1 = __builtin_amdgcn_perm(B.hi, B.lo, k.sel[2 * p + 1]);
            const f2 A = { (float)(t0 & 255u), (float)(t1 & 255u) }, Bq = { (float)((t0 >> 16) & 255u), (float)((t1 >> 16) & 255u) };
            const f2 C = { (float)(b0 & 255u), (float)(b1 & 255u) }, D = { (float)((b0 >> 16) & 255u), (float)((b1 >> 16) & 255u) };
            const f2 wx = { __uint_as_float(k.w[2 * p]), __uint_as_float(k.w[2 * p + 1]) }, omx = (f2){ 1.0f, 1.0f } - wx;
            const f2 sum = bilerp2(A, Bq, C, D, wx, omx, wy, omy);
            Yf[r][2 * p] = __builtin_truncf(sum.x);
            Yf[r][2 * p + 1] = __builtin_truncf(sum.y);
        }
    }
    color_store_tile<OUT, true, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
}

// Integer weights of an axis index for the other users of the integer window tile (LaunchDesc::tap22): packed (w0 | w1 << 16)
__device__ __forceinline__ uint32_t tap22_weights(int kind, int idx) {
    return kind == 1 ? ((idx & 1) ? (1u | (2u << 16)) : (2u | (1u << 16))) : (1u | (1u << 16));
}
// table weight field: the float weight, or for the integer tiles the packed pair (16 - 16 w) | (16 w) << 16
__device__ __forceinline__ float table_weight(const LaunchDesc &d, float w) {
    const uint32_t k16 = (uint32_t)(w * 16.0f);
    return d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w;
}

template <bool AREAUP, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr int MODE = AREAUP ? M_AREA_UP : M_BILINEAR;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH * d.rpt;
    const Footprint f = tile_footprint<MODE>(d, id);
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    XEntry *xtab = (XEntry *)(lds_uv + d.lds_rows_uv * d.lds_cpr_uv * 16);
    XEntry *cxtab = xtab + tw;
    YEntry *ytab = (YEntry *)(cxtab + (tw >> 1));
    YEntry *cytab = ytab + th;

    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, f.ylo, f.xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, f.cylo, 2 * f.cxlo, d.lds_cpr_uv, auv);
    const int ny = min(f.yhi - f.ylo + 1, d.lds_rows_y), nuv = d.luma_only ? 0 : min(f.cyhi - f.cylo + 1, d.lds_rows_uv);
    if (d.dma) {
        stage_plane_dma(lds_y, ay, py, d.pitch_y, ny, min(f.xhi - f.xlo + 1, d.lds_span_y), d.lds_magic_y, nthreads);
        stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, nuv, min(2 * (f.cxhi - f.cxlo + 1), d.lds_span_uv), d.lds_magic_uv, nthreads);
    } else {
        stage_planes<2, 1>(d, lds_y, ay, py, ny, min(f.xhi - f.xlo + 1, d.lds_span_y), lds_uv, auv, puv, nuv,
                           min(2 * (f.cxhi - f.cxlo + 1), d.lds_span_uv), nthreads);
    }

    // coordinate tables (one entry per lane)
    const int ntab = tw + (tw >> 1) + th + (th >> 1);
    for (int e = threadIdx.x; e < ntab; e += nthreads) {
        int p;
        float w;
        // bil_int: the weight field carries packed integer weights (16 - 16 w) | (16 w) << 16 instead of the float (see
        // bilinear_int_thread_tile)
        if (e < tw) {
            axis2<AREAUP>(f.j_first + e, d.xr, d.src_w, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            xtab[e] = XEntry{ p - f.xlo, d.tap22 ? __uint_as_float(tap22_weights(d.tap22, f.j_first + e)) : d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w };
        } else if (e < tw + (tw >> 1)) {
            const int k = e - tw;
            axis2<AREAUP>((f.j_first >> 1) + k, d.xr, d.src_w, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            cxtab[k] = XEntry{ 2 * (p - f.cxlo), d.tap22 ? __uint_as_float(tap22_weights(d.tap22, (f.j_first >> 1) + k)) : d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w };
        } else if (e < tw + (tw >> 1) + th) {
            const int k = e - tw - (tw >> 1);
            axis2<AREAUP>(f.i_first + k, d.yr, d.src_h, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            const int r0 = p - f.ylo, r1 = ((p + 1 >= d.src_h) ? p : p + 1) - f.ylo; // y + 1 >= height -> same row
            ytab[k] = YEntry{ r0 * py.lp + ((py.m0 + r0 * py.pm) & 15), r1 * py.lp + ((py.m0 + r1 * py.pm) & 15),
                              d.tap22 ? __uint_as_float(tap22_weights(d.tap22, f.i_first + k)) : d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w, 0 };
        } else {
            const int k = e - tw - (tw >> 1) - th;
            axis2<AREAUP>((f.i_first >> 1) + k, d.yr, d.src_h, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            const int r0 = p - f.cylo, r1 = ((p + 1 >= chh) ? p : p + 1) - f.cylo;
            cytab[k] = YEntry{ r0 * puv.lp + ((puv.m0 + r0 * puv.pm) & 15), r1 * puv.lp + ((puv.m0 + r1 * puv.pm) & 15),
                               d.tap22 ? __uint_as_float(tap22_weights(d.tap22, (f.i_first >> 1) + k)) : d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w, 0 };
        }
    }
    if (d.dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA chunks have landed
    __syncthreads();
    // replicate the last column / chroma pair one step past the plane (tiles on the right edge only)
    const bool edge_y = (f.xhi == d.src_w - 1), edge_uv = (f.cxhi == cw - 1);
    if (edge_y || edge_uv) {
        if (edge_y)
            for (int r = threadIdx.x; r < ny; r += nthreads) {
                uint8_t *q = lds_y + r * py.lp + ((py.m0 + r * py.pm) & 15) + (d.src_w - f.xlo);
                q[0] = q[-1];
            }
        if (edge_uv)
            for (int r = threadIdx.x; r < nuv; r += nthreads) {
                uint8_t *q = lds_uv + r * puv.lp + ((puv.m0 + r * puv.pm) & 15) + 2 * (cw - f.cxlo);
                q[0] = q[-2];
                q[1] = q[-1];
            }
        __syncthreads();
    }

    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = f.j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    // thread tile = 4 columns x (2 * rpt) rows: the tile decode, the staging set-up and the table build are
    // paid once per 8 * rpt pixels
    if (d.bil_int == 2 || d.bil_win) {
        const WinColumns k = win_columns(xtab, cxtab, lx);
        for (int rp = 0; rp < d.rpt; rp++) {
            const int lyr = ly * d.rpt + rp, i0 = f.i_first + lyr * PXH;
            if (i0 >= d.dst_h) break;
            const RowPairGeo g = rows_from_lds(ytab, cytab, lyr);
            if constexpr (!AREAUP) {
                if (d.tap22) {
                    bilinear_win_thread_tile<OUT, true>(d, lds_y, lds_uv, k, g, (T *)t.out[id.frame], i0, j0);
                    continue;
                }
            }
            if (d.bil_int == 2) bilinear_win_thread_tile<OUT>(d, lds_y, lds_uv, k, g, (T *)t.out[id.frame], i0, j0);
            else bilinear_winf_thread_tile<OUT>(d, lds_y, lds_uv, k, g, (T *)t.out[id.frame], i0, j0);
        }
        return;
    }
    for (int rp = 0; rp < d.rpt; rp++) {
        const int lyr = ly * d.rpt + rp, i0 = f.i_first + lyr * PXH;
        if (i0 >= d.dst_h) break;
        if (d.bil_int) bilinear_int_thread_tile<OUT>(d, lds_y, lds_uv, xtab, cxtab, ytab, cytab, lx, lyr, (T *)t.out[id.frame], i0, j0);
        else bilinear_thread_tile<OUT>(d, lds_y, lds_uv, xtab, cxtab, ytab, cytab, lx, lyr, (T *)t.out[id.frame], i0, j0);
    }
}

// ----------------------------------------------------------------------------------------------
// Geometry-table variant of the 2x2-tap kernel (window tiles, LDS-DMA staging, pitches that are multiples of 16).
// In vpp_bilinear_kernel more than half of a wave's VALU instructions are NOT blends or colour arithmetic: they are the tile's
// footprint (eight coordinate evaluations on wave-uniform values -- gfx950 has no scalar float unit), the workgroup's
// coordinate tables (one evaluation per lane, written to LDS, a barrier, read back, turned into selectors) -- about 300 of the 645
// VALU instructions a wave of the uint8 1080p -> 720p launch executes, and that launch keeps the VALUs 76 % busy
// (profiles/r02_u8_window_pmc.txt).  None of it depends on the frame: for a given request and tile shape it is the same for every
// tile column / tile row / output column / output row.  The host therefore evaluates the coordinate functions ONCE per
// (request, tile shape) -- the same __host__ __device__ functions of vpp_axis.h, IEEE operation for IEEE operation -- and
// keeps the results in device memory (GeoCache, owned by the context; built by tsvpp_prepare_batch or on first use):
//   geo_tx[tile column] = { xlo, xhi, cxlo, cxhi }   geo_ty[tile row] = { ylo, yhi, cylo, cyhi }         -- scalar loads
//   geo_col[output column quad] : first tap column, v_perm selectors and weights of the 4 luma / 2 chroma columns   (64 B)
//   geo_row[output row pair]    : LDS row offsets (row * LDS pitch) and weights of the 2 luma rows / 1 chroma row  (48 B)
// A thread loads its column record and its first row-pair record before the staging (the latency hides behind the LDS-DMA),
// no coordinate table is built in LDS and the kernel is the same for BILINEAR and the AREA up-scale.
struct RowRec { uint4 a, b, c; };
__device__ __forceinline__ RowRec load_row_rec(const uint4 *geo_row, int pair) {
    const uint4 *g = geo_row + 3 * pair;
    return RowRec{ g[0], g[1], g[2] };
}
__device__ __forceinline__ RowPairGeo rows_from_rec(const RowRec &r) {
    RowPairGeo g;
    g.top[0] = (int)r.a.x; g.bot[0] = (int)r.a.y; g.wy[0] = r.a.z; g.ctop = (int)r.a.w;
    g.top[1] = (int)r.b.x; g.bot[1] = (int)r.b.y; g.wy[1] = r.b.z; g.cbot = (int)r.b.w;
    g.cwy = r.c.x;
    return g;
}
// SROWS: the workgroup is at least 64 thread tiles wide, so a wave's lanes share their output rows: the row-pair records are read
// through a wave-uniform index -- scalar loads into SGPRs instead of 24 VGPRs per lane (80 -> 64 VGPRs: eight waves per SIMD).
template <int OUT, bool SROWS>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_geo_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int4 gx = d.geo_tx[id.tx], gy = d.geo_ty[id.ty]; // uniform addresses: scalar loads
    const int xlo = gx.x, xhi = gx.y, cxlo = gx.z, cxhi = gx.w, ylo = gy.x, yhi = gy.y, cylo = gy.z, cyhi = gy.w;
    const int j_first = id.tx * d.tx * PXW, i_first = id.ty * d.ty * PXH * d.rpt;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = j_first + lx * PXW, lyr0 = ly * d.rpt;
    const int nquads = (d.dst_w + 3) >> 2, npairs = d.dst_h >> 1;
    const uint4 *gc = d.geo_col + 4 * min(j0 >> 2, nquads - 1);
    const uint4 c0 = gc[0], c1 = gc[1], c2 = gc[2], c3 = gc[3];
    // row-pair records of this thread's first two row pairs (rpt <= 2 for every default shape: both are in flight during the staging)
    const int pair0 = SROWS ? __builtin_amdgcn_readfirstlane((i_first >> 1) + lyr0) : (i_first >> 1) + lyr0;
    RowRec ra = load_row_rec(d.geo_row, min(pair0, npairs - 1)), rb = load_row_rec(d.geo_row, min(pair0 + 1, npairs - 1));

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    const int cw = d.src_w >> 1;
    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, ylo, xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, cylo, 2 * cxlo, d.lds_cpr_uv, auv);
    const int ny = min(yhi - ylo + 1, d.lds_rows_y), nuv = d.luma_only ? 0 : min(cyhi - cylo + 1, d.lds_rows_uv);
    stage_plane_dma(lds_y, ay, py, d.pitch_y, ny, min(xhi - xlo + 1, d.lds_span_y), d.lds_magic_y, nthreads);
    stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, nuv, min(2 * (cxhi - cxlo + 1), d.lds_span_uv), d.lds_magic_uv, nthreads);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA chunks have landed
    __syncthreads();
    // replicate the last column / chroma pair one step past the plane (tiles on the right edge only); pitch % 16 == 0: every
    // row has the misalignment m0
    const bool edge_y = (xhi == d.src_w - 1), edge_uv = (cxhi == cw - 1);
    if (edge_y || edge_uv) {
        if (edge_y)
            for (int r = threadIdx.x; r < ny; r += nthreads) {
                uint8_t *q = lds_y + r * py.lp + py.m0 + (d.src_w - xlo);
                q[0] = q[-1];
            }
        if (edge_uv)
            for (int r = threadIdx.x; r < nuv; r += nthreads) {
                uint8_t *q = lds_uv + r * puv.lp + puv.m0 + 2 * (cw - cxlo);
                q[0] = q[-2];
                q[1] = q[-1];
            }
        __syncthreads();
    }
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    // column record -> WinColumns; the tile's origin (uniform) folds into the column offsets: LDS address of source (row, col) =
    // (row - ylo) * lp + m0 + (col - xlo), and geo_row carries row * lp
    WinColumns k;
    k.xo0 = (int)c0.x + (py.m0 - xlo - ylo * py.lp);
    k.cxo0 = (int)c0.y + (puv.m0 - 2 * cxlo - cylo * puv.lp);
    k.sel[0] = c0.z; k.sel[1] = c0.w; k.sel[2] = c1.x; k.sel[3] = c1.y;
    k.w[0] = c1.z; k.w[1] = c1.w; k.w[2] = c2.x; k.w[3] = c2.y;
    k.csel[0] = 0x0c020c00u;
    k.csel[1] = c2.z;
    k.cw[0] = c2.w; k.cw[1] = c3.x;
    for (int rp = 0; rp < d.rpt; rp += 2) {
        const int i0 = i_first + (lyr0 + rp) * PXH;
        if (i0 >= d.dst_h) break;
        const bool second = rp + 1 < d.rpt && i0 + PXH < d.dst_h;
        if (d.bil_int == 2) {
            bilinear_win_thread_tile<OUT>(d, lds_y, lds_uv, k, rows_from_rec(ra), (T *)t.out[id.frame], i0, j0);
            if (second) bilinear_win_thread_tile<OUT>(d, lds_y, lds_uv, k, rows_from_rec(rb), (T *)t.out[id.frame], i0 + PXH, j0);
        } else {
            bilinear_winf_thread_tile<OUT>(d, lds_y, lds_uv, k, rows_from_rec(ra), (T *)t.out[id.frame], i0, j0);
            if (second) bilinear_winf_thread_tile<OUT>(d, lds_y, lds_uv, k, rows_from_rec(rb), (T *)t.out[id.frame], i0 + PXH, j0);
        }
        if (rp + 2 < d.rpt) { // TSVPP_RPT > 2 only
            ra = load_row_rec(d.geo_row, min((i0 >> 1) + 2, npairs - 1));
            rb = load_row_rec(d.geo_row, min((i0 >> 1) + 3, npairs - 1));
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Host side of the geometry tables.
struct GeoHost {
    std::vector<int4> tx, ty;
    std::vector<uint4> col, row;
};
// Evaluates the coordinate functions for one (request, tile shape).  false: some thread tile's taps do not fit the 8-byte
// windows (cannot happen for ratios <= 2; checked, not assumed) -- the caller keeps vpp_bilinear_kernel.
static bool geo_tables_host(bool areaup, const LaunchDesc &d, GeoHost &g) {
    auto axis = [&](int idx, float ratio, int limit, int &p, float &w) {
        if (areaup) areaup_axis(idx, ratio, p, w);
        else bilinear_axis(idx, ratio, limit, p, w);
    };
    auto weight = [&](float w) -> uint32_t { // table_weight()
        if (d.bil_int) {
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            return (16u - k16) | (k16 << 16);
        }
        uint32_t u;
        memcpy(&u, &w, 4);
        return u;
    };
    const int tw = d.tx * PXW, th = d.ty * PXH * d.rpt;
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;
    const int lp_y = 16 * d.lds_cpr_y, lp_uv = 16 * d.lds_cpr_uv;
    // tile footprints: tile_footprint<M_BILINEAR / M_AREA_UP>, axis_span
    auto span = [&](int o0, int o1, float ratio, int limit, int hi_max, int &lo, int &hi) {
        float w;
        axis(o0, ratio, limit, lo, w);
        axis(o1, ratio, limit, hi, w);
        hi += 1;
        lo = std::max(lo, 0);
        hi = std::min(hi, hi_max);
    };
    g.tx.resize((size_t)d.tiles_x);
    for (int c = 0; c < d.tiles_x; c++) {
        const int j_first = c * tw, j_last = std::min(j_first + tw, d.dst_w) - 1;
        int4 v;
        span(j_first, j_last, d.xr, d.src_w, d.src_w - 1, v.x, v.y);
        span(j_first >> 1, j_last >> 1, d.xr, d.src_w, cw - 1, v.z, v.w);
        g.tx[(size_t)c] = v;
    }
    g.ty.resize((size_t)d.tiles_y);
    for (int r = 0; r < d.tiles_y; r++) {
        const int i_first = r * th, i_last = std::min(i_first + th, d.dst_h) - 1;
        int4 v;
        span(i_first, i_last, d.yr, d.src_h, d.src_h - 1, v.x, v.y);
        span(i_first >> 1, i_last >> 1, d.yr, d.src_h, chh - 1, v.z, v.w);
        g.ty[(size_t)r] = v;
    }
    // column records: win_columns() of every output column quad, first tap as an absolute source column
    const int nquads = (d.dst_w + 3) >> 2;
    g.col.assign((size_t)nquads * 4, make_uint4(0, 0, 0, 0));
    for (int q = 0; q < nquads; q++) {
        int px[PXW], cp[2];
        float wx[PXW], cwx[2];
        for (int i = 0; i < PXW; i++) axis(std::min(4 * q + i, d.dst_w - 1), d.xr, d.src_w, px[i], wx[i]);
        for (int c = 0; c < 2; c++) axis(std::min(2 * q + c, (d.dst_w >> 1) - 1), d.xr, d.src_w, cp[c], cwx[c]);
        uint32_t sel[PXW];
        for (int i = 0; i < PXW; i++) {
            const int rel = px[i] - px[0];
            if (rel < 0 || rel > 6) return false; // taps rel, rel + 1 of an 8-byte window
            sel[i] = (uint32_t)rel * 0x00010001u + 0x0c010c00u;
        }
        const int crel = 2 * (cp[1] - cp[0]);
        if (crel < 0 || crel > 4) return false; // bytes crel .. crel + 3 (U V U V) of an 8-byte window
        uint4 *o = &g.col[(size_t)q * 4];
        o[0] = make_uint4((uint32_t)px[0], (uint32_t)(2 * cp[0]), sel[0], sel[1]);
        o[1] = make_uint4(sel[2], sel[3], weight(wx[0]), weight(wx[1]));
        o[2] = make_uint4(weight(wx[2]), weight(wx[3]), (uint32_t)crel * 0x00010001u + 0x0c020c00u, weight(cwx[0]));
        o[3] = make_uint4(weight(cwx[1]), 0u, 0u, 0u);
    }
    // row-pair records: the YEntry pairs of vpp_bilinear_kernel with absolute rows (times the LDS row pitch)
    const int npairs = d.dst_h >> 1;
    g.row.assign((size_t)npairs * 3, make_uint4(0, 0, 0, 0));
    for (int ip = 0; ip < npairs; ip++) {
        int p[PXH], b[PXH], cpv, cb;
        float w[PXH], cwv;
        for (int r = 0; r < PXH; r++) {
            axis(2 * ip + r, d.yr, d.src_h, p[r], w[r]);
            b[r] = (p[r] + 1 >= d.src_h) ? p[r] : p[r] + 1; // y + 1 >= height -> same row
        }
        axis(ip, d.yr, d.src_h, cpv, cwv);
        cb = (cpv + 1 >= chh) ? cpv : cpv + 1;
        uint4 *o = &g.row[(size_t)ip * 3];
        o[0] = make_uint4((uint32_t)(p[0] * lp_y), (uint32_t)(b[0] * lp_y), weight(w[0]), (uint32_t)(cpv * lp_uv));
        o[1] = make_uint4((uint32_t)(p[1] * lp_y), (uint32_t)(b[1] * lp_y), weight(w[1]), (uint32_t)(cb * lp_uv));
        o[2] = make_uint4(weight(cwv), 0u, 0u, 0u);
    }
    return true;
}

GeoCache *geo_cache_create() { return new GeoCache(); }
void geo_cache_destroy(GeoCache *c) {
    if (!c) return;
    for (auto &e : c->map)
        if (e.second.dev) (void)hipFree(e.second.dev);
    for (uint8_t *p : c->retired) (void)hipFree(p);
    delete c;
}

size_t geo_cache_trim(GeoCache *c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    for (uint8_t *p : c->retired) (void)hipFree(p);
    const size_t n = c->retired_bytes;
    c->retired.clear();
    c->retired_bytes = 0;
    c->warned = false;
    return n;
}

// The tables of this launch (d: shape and LDS layout already chosen), built and uploaded on first use.  Never allocates while
// the stream is being captured into a graph (the launch then keeps vpp_bilinear_kernel; tsvpp_prepare_batch avoids that).
static bool geo_lookup(GeoCache *cache, bool areaup, const LaunchDesc &d, hipStream_t stream, bool may_build, LaunchDesc &out) {
    GeoKey key;
    memset(&key, 0, sizeof(key));
    uint32_t xb, yb;
    memcpy(&xb, &d.xr, 4);
    memcpy(&yb, &d.yr, 4);
    const int kv[16] = { areaup ? 1 : 0, d.src_w, d.src_h, d.dst_w, d.dst_h, (int)xb, (int)yb, d.tx, d.ty, d.rpt, d.bil_int, d.lds_cpr_y, d.lds_cpr_uv,
                         d.tiles_x, d.tiles_y, 0 };
    memcpy(key.v, kv, sizeof(kv));
    std::lock_guard<std::mutex> lk(cache->mu);
    auto it = cache->map.find(key);
    if (it == cache->map.end()) {
        if (!may_build) return false;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        // (the NULL stream is never queried: asking the legacy stream while another stream captures in global mode invalidates that capture)
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return false;
        if (!geo_cache_make_room(cache)) return false;
        GeoHost g;
        GeoEntry e;
        if (geo_tables_host(areaup, d, g)) {
            auto up16 = [](size_t n) { return (n + 255) & ~(size_t)255; };
            const size_t b_tx = g.tx.size() * sizeof(int4), b_ty = g.ty.size() * sizeof(int4), b_col = g.col.size() * sizeof(uint4),
                         b_row = g.row.size() * sizeof(uint4);
            e.off_ty = up16(b_tx);
            e.off_col = e.off_ty + up16(b_ty);
            e.off_row = e.off_col + up16(b_col);
            const size_t total = e.off_row + up16(b_row);
            std::vector<uint8_t> host(total, 0);
            memcpy(host.data(), g.tx.data(), b_tx);
            memcpy(host.data() + e.off_ty, g.ty.data(), b_ty);
            memcpy(host.data() + e.off_col, g.col.data(), b_col);
            memcpy(host.data() + e.off_row, g.row.data(), b_row);
            if (hipMalloc((void **)&e.dev, total) != hipSuccess) return false;
            if (hipMemcpy(e.dev, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(e.dev);
                return false;
            }
            e.bytes = total;
        }
        it = cache->map.emplace(key, e).first;
    }
    it->second.stamp = ++cache->clock;
    const GeoEntry &e = it->second;
    if (!e.dev) return false;
    out.geo_tx = (const int4 *)e.dev;
    out.geo_ty = (const int4 *)(e.dev + e.off_ty);
    out.geo_col = (const uint4 *)(e.dev + e.off_col);
    out.geo_row = (const uint4 *)(e.dev + e.off_row);
    return true;
}

static hipError_t launch_bilinear_geo(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    switch (out) {
#define TSVPP_GEO(O)                                                                                      \
    case O:                                                                                               \
        if (d.tx >= 64) TSVPP_LAUNCH((vpp_bilinear_geo_kernel<O, true>), grid, block, lds, stream, d, t); \
        else TSVPP_LAUNCH((vpp_bilinear_geo_kernel<O, false>), grid, block, lds, stream, d, t);     \
        break;
        TSVPP_GEO(O_U8_PLANAR) TSVPP_GEO(O_U8_MERGED) TSVPP_GEO(O_F32_PLANAR) TSVPP_GEO(O_F32_MERGED) TSVPP_GEO(O_NV12_U8)
        TSVPP_GEO(O_NV12_F32) TSVPP_GEO(O_Y800_U8) TSVPP_GEO(O_Y800_F32) TSVPP_GEO(O_HSV_F32)
#undef TSVPP_GEO
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <bool AREAUP>
static hipError_t launch_bilinear_a(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    switch (out) {
#define TSVPP_BIL(O)                                                                                                            \
    case O:                                                                                                                     \
        TSVPP_LAUNCH((vpp_bilinear_kernel<AREAUP, O>), grid, block, lds, stream, d, t);                                   \
        break;
        TSVPP_BIL(O_U8_PLANAR) TSVPP_BIL(O_U8_MERGED) TSVPP_BIL(O_F32_PLANAR) TSVPP_BIL(O_F32_MERGED) TSVPP_BIL(O_NV12_U8)
        TSVPP_BIL(O_NV12_F32) TSVPP_BIL(O_Y800_U8) TSVPP_BIL(O_Y800_F32) TSVPP_BIL(O_HSV_F32)
#undef TSVPP_BIL
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bilinear(bool areaup, OutKind out, const LaunchDesc &din, const FrameTable &t, unsigned grid_x, size_t lds_bytes,
                           hipStream_t stream, LaunchInfo *info) {
    LaunchDesc d = din;
    dim3 grid(grid_x), block((unsigned)(d.tx * d.ty));
    // geometry tables: window tiles on LDS-DMA staging with pitches that are multiples of 16 (every row then has one
    // misalignment); the coordinate tables of vpp_bilinear_kernel no longer occupy LDS
    d.geo = 0;
    // Measured (profiles/r02_geo_ab.txt, same box): uint8 outputs on the integer window tile gain 2..5 % (1080p -> 720p planar 0.528 ->
    // 0.554, 4K -> 1080p 0.695 -> 0.710: the kernel is VALU-bound there and a wave executes ~19 % fewer VALU instructions), but
    // fp32 outputs LOSE 8..19 % (headline 0.742 -> 0.658): they are bound by the memory pipeline, and a thread's records are ten
    // 16-byte loads per 8 or 16 pixels -- several times the source bytes the tile stages.  So: uint8 flavours with dyadic weights
    // only; TSVPP_GEO=2 forces the tables wherever they apply, TSVPP_GEO=0 disables them.
    const bool u8_out = (out == O_U8_PLANAR || out == O_U8_MERGED || out == O_NV12_U8 || out == O_Y800_U8);
    const bool geo_want = d.geo_pref == 2 || (d.geo_pref == 1 && u8_out && d.bil_int == 2);
    const bool geo_ok = geo_want && !d.last_col0 && d.dma && (d.bil_int == 2 || d.bil_win) && (d.pitch_y & 15) == 0 && (d.pitch_uv & 15) == 0;
    if (geo_ok) {
        if (d.geo_cache) d.geo = geo_lookup(d.geo_cache, areaup, d, stream, !info || d.geo_build, d) ? 1 : 0;
        else if (info) { // tsvpp_describe: no device -- eligibility only
            GeoHost g;
            d.geo = geo_tables_host(areaup, d, g) ? 1 : 0;
        }
    }
    if (d.geo) { // no coordinate tables in LDS
        const size_t cols = (size_t)d.tx * PXW, rows = (size_t)d.ty * PXH * d.rpt;
        lds_bytes -= (cols + cols / 2) * sizeof(XEntry) + (rows + rows / 2) * sizeof(YEntry);
    }
    if (info) {
        info->kernel = areaup ? "vpp_bilinear_kernel<areaup,OUT>" : d.tap22 ? "vpp_bilinear_kernel<bilinear,OUT>[area-weights]" : "vpp_bilinear_kernel<bilinear,OUT>";
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        info->geo = d.geo;
        return hipSuccess;
    }
    if (d.geo) return launch_bilinear_geo(out, d, t, grid, block, lds_bytes, stream);
    return areaup ? launch_bilinear_a<true>(out, d, t, grid, block, lds_bytes, stream)
                  : launch_bilinear_a<false>(out, d, t, grid, block, lds_bytes, stream);
}

} // namespace tsvpp

#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
