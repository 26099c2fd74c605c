// vpp_bilinear.hip -- the 2x2-tap kernel family (BILINEAR and the AREA up-scale variant; the headline kernel) in its own
// translation unit, compiled with LDS accesses of unknown alignment split into byte reads (Makefile: -unaligned-access-mode
// off for this file only).  Measured on MI355X: a ds_read_u16 / ds_read_b32 whose address is not naturally aligned is
// executed lane by lane (~64 cycles per wave instruction); at ratio 1.5 half of all tap addresses are odd, and the first
// version of the integer thread tile, which read each horizontal tap pair as one 16-bit word, ran at 195 us per launch
// whatever the output type (LDS-bound) against 154 us for the float tile (profiles/r02_bilinear_int_ab.txt).
#include "vpp_device.h"

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
#ifdef TSVPP_ABLATION
    if (d.ablate & 4) { // profiling: staging + stores only
        for (int c = 0; c < 2; c++) Uf[c] = Vf[c] = (float)lds_uv[cye.top + cxe[c].off];
        for (int r = 0; r < PXH; r++)
            for (int c = 0; c < PXW; c++) Yf[r][c] = (float)lds_y[ye[r].top + xe[c].off];
        if (d.ablate & 1) { // loads only
            float acc = Uf[0] + Vf[1];
            for (int r = 0; r < PXH; r++)
                for (int c = 0; c < PXW; c++) acc += Yf[r][c];
            if (acc == -1.0f) ((float *)out)[0] = acc;
            return;
        }
        const size_t plane = (size_t)d.dst_w * d.dst_h;
        float *o = (float *)out;
        for (int r = 0; r < PXH; r++)
            for (int p = 0; p < 3; p++) st4(o + p * plane + (size_t)(i0 + r) * d.dst_w + j0, Yf[r][0], Yf[r][1], Yf[r][2], Yf[r][3], d.nt_stores);
        return;
    }
#endif
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
#ifdef TSVPP_ABLATION
    if (d.ablate & 1) { // profiling: keep the arithmetic alive without the HBM writes
        float acc = Uf[0] + Vf[0] + Uf[1] + Vf[1];
        for (int r = 0; r < PXH; r++)
            for (int c = 0; c < PXW; c++) acc += Yf[r][c];
        if (acc == -1.0f) ((float *)out)[0] = acc;
        return;
    }
#endif
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
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
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
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
template <int OUT>
__device__ __forceinline__ void bilinear_win_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const WinColumns &k,
                                                         const YEntry *ytab, const YEntry *cytab, int ly, typename OutT<OUT>::type *out, int i0,
                                                         int j0) {
    float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) {
        const uint4 cy = *(const uint4 *)(cytab + ly);
        const TapWindow T = window8(lds_uv, (int)cy.x + k.cxo0), B = window8(lds_uv, (int)cy.y + k.cxo0);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t su = k.csel[c], sv = su + 0x00010001u;
            const uint32_t tu = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, su), k.cw[c], 0u, false);
            const uint32_t tv = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, sv), k.cw[c], 0u, false);
            const uint32_t bu = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, su), k.cw[c], 0u, false);
            const uint32_t bv = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, sv), k.cw[c], 0u, false);
            Uf[c] = (float)((vpair(tu, bu, cy.z) >> 8) & 255u);
            Vf[c] = (float)((vpair(tv, bv, cy.z) >> 8) & 255u);
        }
    }
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const uint4 ye = *(const uint4 *)(ytab + ly * PXH + r);
        const TapWindow T = window8(lds_y, (int)ye.x + k.xo0), B = window8(lds_y, (int)ye.y + k.xo0);
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            const uint32_t tt = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(T.hi, T.lo, k.sel[c]), k.w[c], 0u, false);
            const uint32_t bb = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(B.hi, B.lo, k.sel[c]), k.w[c], 0u, false);
            Yf[r][c] = (float)((vpair(tt, bb, ye.z) >> 8) & 255u);
        }
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
}

// Window form of the FLOAT thread tile (LaunchDesc::bil_win: any weights, horizontal ratio <= 2): the same row windows and
// per-column selectors as the integer window tile; v_perm_b32 yields (A, 0, B, 0) and the two taps are converted by
// v_cvt_f32_ubyte0 / v_cvt_f32_ubyte2.  The blend is bilerp2 -- the float tile's arithmetic, unchanged.
template <int OUT>
__device__ __forceinline__ void bilinear_winf_thread_tile(const LaunchDesc &d, const uint8_t *lds_y, const uint8_t *lds_uv, const WinColumns &k,
                                                          const YEntry *ytab, const YEntry *cytab, int ly, typename OutT<OUT>::type *out, int i0,
                                                          int j0) {
    float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) {
        const uint4 cy = *(const uint4 *)(cytab + ly);
        const TapWindow T = window8(lds_uv, (int)cy.x + k.cxo0), B = window8(lds_uv, (int)cy.y + k.cxo0);
        const float wyf = __uint_as_float(cy.z);
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
        const uint4 ye = *(const uint4 *)(ytab + ly * PXH + r);
        const TapWindow T = window8(lds_y, (int)ye.x + k.xo0), B = window8(lds_y, (int)ye.y + k.xo0);
        const float wyf = __uint_as_float(ye.z);
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
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, out, i0, j0, PXW);
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
#ifdef TSVPP_ABLATION
    if (!(d.ablate & 2))
#endif
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
            xtab[e] = XEntry{ p - f.xlo, d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w };
        } else if (e < tw + (tw >> 1)) {
            const int k = e - tw;
            axis2<AREAUP>((f.j_first >> 1) + k, d.xr, d.src_w, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            cxtab[k] = XEntry{ 2 * (p - f.cxlo), d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w };
        } else if (e < tw + (tw >> 1) + th) {
            const int k = e - tw - (tw >> 1);
            axis2<AREAUP>(f.i_first + k, d.yr, d.src_h, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            const int r0 = p - f.ylo, r1 = ((p + 1 >= d.src_h) ? p : p + 1) - f.ylo; // y + 1 >= height -> same row
            ytab[k] = YEntry{ r0 * py.lp + ((py.m0 + r0 * py.pm) & 15), r1 * py.lp + ((py.m0 + r1 * py.pm) & 15),
                              d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w, 0 };
        } else {
            const int k = e - tw - (tw >> 1) - th;
            axis2<AREAUP>((f.i_first >> 1) + k, d.yr, d.src_h, p, w);
            const uint32_t k16 = (uint32_t)(w * 16.0f);
            const int r0 = p - f.cylo, r1 = ((p + 1 >= chh) ? p : p + 1) - f.cylo;
            cytab[k] = YEntry{ r0 * puv.lp + ((puv.m0 + r0 * puv.pm) & 15), r1 * puv.lp + ((puv.m0 + r1 * puv.pm) & 15),
                               d.bil_int ? __uint_as_float((16u - k16) | (k16 << 16)) : w, 0 };
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
            if (d.bil_int == 2) bilinear_win_thread_tile<OUT>(d, lds_y, lds_uv, k, ytab, cytab, lyr, (T *)t.out[id.frame], i0, j0);
            else bilinear_winf_thread_tile<OUT>(d, lds_y, lds_uv, k, ytab, cytab, lyr, (T *)t.out[id.frame], i0, j0);
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
// Persistent variant of the 2x2-tap kernel (opt-in, TSVPP_PERSIST=k workgroups per CU).  A fixed
// grid of resident workgroups walks the tile list (tile = block + i * grid: neighbouring
// workgroups still write neighbouring tiles).  LDS holds TWO tile sets; while set `cur` is blended,
// colour-converted and stored, the NEXT tile's chunks stream into the other set by LDS-DMA -- no
// registers held across the compute phase -- and its coordinate tables are built.  One barrier per
// tile (two on right-edge tiles).
struct TileCtx {
    TileId id;
    Footprint f;
    LdsPlane py, puv;
    const uint8_t *ay, *auv;
    int ny, nuv, span_y, span_uv;
};
template <int MODE>
__device__ __forceinline__ void tile_ctx(const LaunchDesc &d, const FrameTable &t, int tile, uint8_t *lds_y, uint8_t *lds_uv, TileCtx &c) {
    const int tiles = d.tiles_x * d.tiles_y;
    c.id.frame = tile / tiles;
    const int rem = tile - c.id.frame * tiles;
    c.id.ty = rem / d.tiles_x;
    c.id.tx = rem - c.id.ty * d.tiles_x;
    c.id.valid = true;
    c.f = tile_footprint<MODE>(d, c.id);
    c.py = describe_plane(lds_y, t.y[c.id.frame], d.pitch_y, c.f.ylo, c.f.xlo, d.lds_cpr_y, c.ay);
    c.puv = describe_plane(lds_uv, t.uv[c.id.frame], d.pitch_uv, c.f.cylo, 2 * c.f.cxlo, d.lds_cpr_uv, c.auv);
    c.ny = min(c.f.yhi - c.f.ylo + 1, d.lds_rows_y);
    c.nuv = d.luma_only ? 0 : min(c.f.cyhi - c.f.cylo + 1, d.lds_rows_uv);
    c.span_y = min(c.f.xhi - c.f.xlo + 1, d.lds_span_y);
    c.span_uv = min(2 * (c.f.cxhi - c.f.cxlo + 1), d.lds_span_uv);
}

template <bool AREAUP, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_persistent_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr int MODE = AREAUP ? M_AREA_UP : M_BILINEAR;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH;
    const int total = d.tiles_x * d.tiles_y * d.n_frames;
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;
    const int y_bytes = d.lds_rows_y * d.lds_cpr_y * 16, uv_bytes = d.lds_rows_uv * d.lds_cpr_uv * 16;
    const int tab_bytes = (tw + (tw >> 1)) * (int)sizeof(XEntry) + (th + (th >> 1)) * (int)sizeof(YEntry);
    const int set_bytes = y_bytes + uv_bytes + tab_bytes;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;

    int tile = blockIdx.x;
    if (tile >= total) return;

    // stream a tile into LDS set `set` (DMA, asynchronous) and build its coordinate tables
    auto issue = [&](int tl, int set) {
        uint8_t *ly_ = lds_raw + set * set_bytes, *luv_ = ly_ + y_bytes;
        XEntry *xtab = (XEntry *)(luv_ + uv_bytes), *cxtab = xtab + tw;
        YEntry *ytab = (YEntry *)(cxtab + (tw >> 1)), *cytab = ytab + th;
        TileCtx c;
        tile_ctx<MODE>(d, t, tl, ly_, luv_, c);
        stage_plane_dma(ly_, c.ay, c.py, d.pitch_y, c.ny, c.span_y, d.lds_magic_y, nthreads);
        stage_plane_dma(luv_, c.auv, c.puv, d.pitch_uv, c.nuv, c.span_uv, d.lds_magic_uv, nthreads);
        const Footprint &f = c.f;
        const int ntab = tw + (tw >> 1) + th + (th >> 1);
        for (int e = threadIdx.x; e < ntab; e += nthreads) {
            int p;
            float w;
            if (e < tw) {
                axis2<AREAUP>(f.j_first + e, d.xr, d.src_w, p, w);
                xtab[e] = XEntry{ p - f.xlo, table_weight(d, w) };
            } else if (e < tw + (tw >> 1)) {
                const int k = e - tw;
                axis2<AREAUP>((f.j_first >> 1) + k, d.xr, d.src_w, p, w);
                cxtab[k] = XEntry{ 2 * (p - f.cxlo), table_weight(d, w) };
            } else if (e < tw + (tw >> 1) + th) {
                const int k = e - tw - (tw >> 1);
                axis2<AREAUP>(f.i_first + k, d.yr, d.src_h, p, w);
                const int r0 = p - f.ylo, r1 = ((p + 1 >= d.src_h) ? p : p + 1) - f.ylo;
                ytab[k] = YEntry{ r0 * c.py.lp + ((c.py.m0 + r0 * c.py.pm) & 15), r1 * c.py.lp + ((c.py.m0 + r1 * c.py.pm) & 15), table_weight(d, w), 0 };
            } else {
                const int k = e - tw - (tw >> 1) - th;
                axis2<AREAUP>((f.i_first >> 1) + k, d.yr, d.src_h, p, w);
                const int r0 = p - f.cylo, r1 = ((p + 1 >= chh) ? p : p + 1) - f.cylo;
                cytab[k] = YEntry{ r0 * c.puv.lp + ((c.puv.m0 + r0 * c.puv.pm) & 15), r1 * c.puv.lp + ((c.puv.m0 + r1 * c.puv.pm) & 15), table_weight(d, w), 0 };
            }
        }
    };

    issue(tile, 0);
    int cur = 0;
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA chunks of set `cur` have landed
        __syncthreads();                                  // ... everybody's, and the tables are visible
        const int next = tile + (int)gridDim.x;
        if (next < total) issue(next, cur ^ 1);           // in flight during everything below
        uint8_t *lds_y = lds_raw + cur * set_bytes, *lds_uv = lds_y + y_bytes;
        XEntry *xtab = (XEntry *)(lds_uv + uv_bytes), *cxtab = xtab + tw;
        YEntry *ytab = (YEntry *)(cxtab + (tw >> 1)), *cytab = ytab + th;
        TileCtx c; // uniform; recomputed rather than carried across the loop
        tile_ctx<MODE>(d, t, tile, lds_y, lds_uv, c);
        const Footprint &f = c.f;
        const bool edge_y = (f.xhi == d.src_w - 1), edge_uv = (f.cxhi == cw - 1);
        if (edge_y || edge_uv) {
            if (edge_y)
                for (int r = threadIdx.x; r < c.ny; r += nthreads) {
                    uint8_t *q = lds_y + r * c.py.lp + ((c.py.m0 + r * c.py.pm) & 15) + (d.src_w - f.xlo);
                    q[0] = q[-1];
                }
            if (edge_uv)
                for (int r = threadIdx.x; r < c.nuv; r += nthreads) {
                    uint8_t *q = lds_uv + r * c.puv.lp + ((c.puv.m0 + r * c.puv.pm) & 15) + 2 * (cw - f.cxlo);
                    q[0] = q[-2];
                    q[1] = q[-1];
                }
            __syncthreads();
        }
        const int j0 = f.j_first + lx * PXW, i0 = f.i_first + ly * PXH;
        if (j0 < d.dst_w && i0 < d.dst_h) {
            if (d.bil_int == 2) bilinear_win_thread_tile<OUT>(d, lds_y, lds_uv, win_columns(xtab, cxtab, lx), ytab, cytab, ly, (T *)t.out[c.id.frame], i0, j0);
            else if (d.bil_win) bilinear_winf_thread_tile<OUT>(d, lds_y, lds_uv, win_columns(xtab, cxtab, lx), ytab, cytab, ly, (T *)t.out[c.id.frame], i0, j0);
            else if (d.bil_int) bilinear_int_thread_tile<OUT>(d, lds_y, lds_uv, xtab, cxtab, ytab, cytab, lx, ly, (T *)t.out[c.id.frame], i0, j0);
            else bilinear_thread_tile<OUT>(d, lds_y, lds_uv, xtab, cxtab, ytab, cytab, lx, ly, (T *)t.out[c.id.frame], i0, j0);
        }
        if (next >= total) break;
        tile = next;
        cur ^= 1;
    }
}

template <bool AREAUP>
static hipError_t launch_bilinear_a(OutKind out, bool persistent, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    switch (out) {
#define TSVPP_BIL(O)                                                                                                            \
    case O:                                                                                                                     \
        if (persistent) hipLaunchKernelGGL((vpp_bilinear_persistent_kernel<AREAUP, O>), grid, block, lds, stream, d, t);         \
        else hipLaunchKernelGGL((vpp_bilinear_kernel<AREAUP, O>), grid, block, lds, stream, d, t);                              \
        break;
        TSVPP_BIL(O_U8_PLANAR) TSVPP_BIL(O_U8_MERGED) TSVPP_BIL(O_F32_PLANAR) TSVPP_BIL(O_F32_MERGED) TSVPP_BIL(O_NV12_U8)
        TSVPP_BIL(O_NV12_F32) TSVPP_BIL(O_Y800_U8) TSVPP_BIL(O_Y800_F32) TSVPP_BIL(O_HSV_F32)
#undef TSVPP_BIL
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bilinear(bool areaup, OutKind out, bool persistent, const LaunchDesc &d, const FrameTable &t, unsigned grid_x, size_t lds_bytes,
                           hipStream_t stream, LaunchInfo *info) {
    dim3 grid(grid_x), block((unsigned)(d.tx * d.ty));
    if (info) {
        info->kernel = persistent ? (areaup ? "vpp_bilinear_persistent_kernel<areaup,OUT>" : "vpp_bilinear_persistent_kernel<bilinear,OUT>")
                                  : (areaup ? "vpp_bilinear_kernel<areaup,OUT>" : "vpp_bilinear_kernel<bilinear,OUT>");
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    return areaup ? launch_bilinear_a<true>(out, persistent, d, t, grid, block, lds_bytes, stream)
                  : launch_bilinear_a<false>(out, persistent, d, t, grid, block, lds_bytes, stream);
}

} // namespace tsvpp
