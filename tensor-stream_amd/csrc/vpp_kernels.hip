// vpp_kernels.hip -- fused NV12 crop -> resize -> YUV->RGB kernels for gfx950 (MI355X / CDNA4).
//
// One launch replaces the reference's cropKernel (src/Crop.cu:4-21), its five resize kernels
// (src/Resize.cu:180-357) and NV12ToRGB24Kernel{Planar,Merged}<T> (src/ColorConversion.cu:6-93),
// which the reference runs as up to three launches with uint8 NV12 intermediates in global
// memory.  The fused kernel keeps those intermediates *as values*: the resized luma of every
// output pixel and the resized chroma of every 2x2 output block are rounded to uint8 exactly
// where the reference stores them, so results are bit-identical without the round trips.
//
// Arithmetic contract: every float/double operation below is a single IEEE-754 operation in the
// order the reference's source text gives it -- NO fused multiply-add (contraction is off for
// this whole file), truncating float->int conversions, round-half-away for the bicubic stage.
// The only fma()s are explicit ones in the exact x/255 sequence.
//
// Written for wave64 / CDNA4 only; no other target is supported.

#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

// ----------------------------------------------------------------------------------------------
// Fallback kernel: taps gathered byte-wise from global memory (any size, any alignment).
template <int MODE, int OUT, bool VEC>
__global__ __launch_bounds__(MAX_THREADS) void vpp_fused_gather_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = (VEC ? tile_col0(d, id.tx, d.tx * PXW) : id.tx * d.tx * PXW + d.col0) + lx * PXW;
    const int i0 = (id.ty * d.ty + ly) * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (VEC && is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    GlobalSrc s;
    s.y = t.y[id.frame];
    s.uv = t.uv[id.frame];
    s.py = d.pitch_y;
    s.puv = d.pitch_uv;
    s.w = d.src_w;
    s.h = d.src_h;
    convert_thread_tile<MODE, OUT, VEC>(s, d, (T *)t.out[id.frame], i0, j0);
}


// ----------------------------------------------------------------------------------------------
// AREA down-scale with float (non-dyadic) weights and RX x RY <= 3 x 3 taps -- ratios below 3 on both axes that the
// integer kernels cannot take (4/3: 1440p -> 1080p; 2.25 x 1.69: 1080x608 -> 480x360; 2.4; ...).  The 2x2-tap skeleton
// once more -- staged footprint, per-workgroup tables, float pairs -- with the reference's weighted box instead of
// the lerp (src/Resize.cu:160-178):
//     sum = 0; div = 0; for a in rows: for b in cols: wgt = wx[b] * wy[a]; div += wgt; sum += p[a][b] * wgt
//     out = (int)(sum / div)
// in exactly that order (the initial 0 + x is exact).  Each tile column / row looks its pattern row up ONCE
// (j % nx, i % ny) instead of once per pixel, and the division is the only IEEE division left per value.
struct AFXEntry { int off; float w[3]; };  // LDS byte offset from the row base, the column weights
struct AFYEntry { int row; float w[3]; };  // first staged row, the row weights

template <int RX, int RY>
__device__ __forceinline__ f2 areaf_pair(const uint8_t *const t0[RY], const uint8_t *const t1[RY], int step, const float wx0[3], const float wx1[3],
                                         const float wy[3]) {
    f2 sum = { 0.0f, 0.0f }, div = { 0.0f, 0.0f };
#pragma unroll
    for (int a = 0; a < RY; a++) {
        const f2 y = { wy[a], wy[a] };
#pragma unroll
        for (int b = 0; b < RX; b++) {
            const f2 wgt = (f2){ wx0[b], wx1[b] } * y;
            const f2 p = { (float)t0[a][b * step], (float)t1[a][b * step] };
            if (a == 0 && b == 0) { // 0 + x == x
                div = wgt;
                sum = p * wgt;
            } else {
                div = div + wgt;
                sum = __builtin_elementwise_fma(p, wgt, sum); // colorSum += data * weight is one fma in the reference's binary
            }
        }
    }
    f2 q;
    q.x = sum.x / div.x;
    q.y = sum.y / div.y;
    return trunc2(q);
}

template <int RX, int RY, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_areaf_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH * d.rpt;
    const Footprint f = tile_footprint<M_AREA_DOWN>(d, id);

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    AFXEntry *xtab = (AFXEntry *)(lds_uv + d.lds_rows_uv * d.lds_cpr_uv * 16);
    AFXEntry *cxtab = xtab + tw;
    AFYEntry *ytab = (AFYEntry *)(cxtab + (tw >> 1));
    AFYEntry *cytab = ytab + th;
    int *rby = (int *)(cytab + (th >> 1)); // LDS byte offset of every staged luma row (incl. its misalignment)
    int *rbuv = rby + d.lds_rows_y;

    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, f.ylo, f.xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, f.cylo, 2 * f.cxlo, d.lds_cpr_uv, auv);
    const int ny = min(f.yhi - f.ylo + 1, d.lds_rows_y), nuv = d.luma_only ? 0 : min(f.cyhi - f.cylo + 1, d.lds_rows_uv);
    const int spy = min(f.xhi - f.xlo + 1, d.lds_span_y), spuv = min(2 * (f.cxhi - f.cxlo + 1), d.lds_span_uv);
    if (d.dma) {
        stage_plane_dma(lds_y, ay, py, d.pitch_y, ny, spy, d.lds_magic_y, nthreads);
        stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, nuv, spuv, d.lds_magic_uv, nthreads);
    } else {
        stage_planes<2, 1>(d, lds_y, ay, py, ny, spy, lds_uv, auv, puv, nuv, spuv, nthreads);
    }
    const int ntab = tw + (tw >> 1) + th + (th >> 1) + d.lds_rows_y + d.lds_rows_uv;
    for (int e = threadIdx.x; e < ntab; e += nthreads) {
        int k = e;
        if (k < tw + (tw >> 1)) { // columns: luma then chroma pairs -- the SAME formulas and pattern rows on their own index
            const bool chroma = k >= tw;
            if (chroma) k -= tw;
            const int j = (chroma ? (f.j_first >> 1) : f.j_first) + k;
            const float *w = d.patx + (j % d.nx) * d.rx;
            const int x = (int)(d.xr * (float)j);
            const AFXEntry en = { chroma ? 2 * (x - f.cxlo) : x - f.xlo, { w[0], w[1], RX > 2 ? w[2] : 0.0f } };
            if (!chroma) xtab[k] = en;
            else cxtab[k] = en;
            continue;
        }
        k -= tw + (tw >> 1);
        if (k < th + (th >> 1)) {
            const bool chroma = k >= th;
            if (chroma) k -= th;
            const int i = (chroma ? (f.i_first >> 1) : f.i_first) + k;
            const float *w = d.paty + (i % d.ny) * d.ry;
            const AFYEntry en = { (int)(d.yr * (float)i) - (chroma ? f.cylo : f.ylo), { w[0], w[1], RY > 2 ? w[2] : 0.0f } };
            if (!chroma) ytab[k] = en;
            else cytab[k] = en;
            continue;
        }
        k -= th + (th >> 1);
        if (k < d.lds_rows_y) rby[k] = k * py.lp + ((py.m0 + k * py.pm) & 15);
        else {
            k -= d.lds_rows_y;
            rbuv[k] = k * puv.lp + ((puv.m0 + k * puv.pm) & 15);
        }
    }
    if (d.dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = f.j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    AFXEntry xe[PXW], cxe[2];
#pragma unroll
    for (int c = 0; c < PXW; c++) xe[c] = xtab[lx * PXW + c];
    cxe[0] = cxtab[lx * 2];
    cxe[1] = cxtab[lx * 2 + 1];
    for (int rp = 0; rp < d.rpt; rp++) {
        const int lyr = ly * d.rpt + rp, i0 = f.i_first + lyr * PXH;
        if (i0 >= d.dst_h) break;
        float Uf[2], Vf[2], Yf[PXH][PXW];
        {
            const AFYEntry ye = cytab[lyr];
            int rb[RY];
#pragma unroll
            for (int a = 0; a < RY; a++) rb[a] = rbuv[ye.row + a];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint8_t *tu[RY], *tv[RY];
#pragma unroll
                for (int a = 0; a < RY; a++) {
                    tu[a] = lds_uv + rb[a] + cxe[c].off;
                    tv[a] = tu[a] + 1;
                }
                const f2 v = areaf_pair<RX, RY>(tu, tv, 2, cxe[c].w, cxe[c].w, ye.w);
                Uf[c] = v.x;
                Vf[c] = v.y;
            }
        }
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const AFYEntry ye = ytab[lyr * PXH + r];
            int rb[RY];
#pragma unroll
            for (int a = 0; a < RY; a++) rb[a] = rby[ye.row + a];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const uint8_t *t0[RY], *t1[RY];
#pragma unroll
                for (int a = 0; a < RY; a++) {
                    t0[a] = lds_y + rb[a] + xe[2 * p].off;
                    t1[a] = lds_y + rb[a] + xe[2 * p + 1].off;
                }
                const f2 v = areaf_pair<RX, RY>(t0, t1, 1, xe[2 * p].w, xe[2 * p + 1].w, ye.w);
                Yf[r][2 * p] = v.x;
                Yf[r][2 * p + 1] = v.y;
            }
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

// ----------------------------------------------------------------------------------------------
// AREA down-scale with dyadic weights (integer ratios -> all ones; 1.5 -> {1,.5}; 2.25 -> quarters...).
// The reference accumulates float(data) * (wx*wy) tap by tap (src/Resize.cu:160-178); when every
// weight is k / 2^s all partial sums are exactly representable, so the result equals
// (int)( float(SUM) / float(SX*SY) ) with SUM = sum_a wy[a] * sum_b wx[b] * p[a][b] in integers --
// and the inner sum over four packed source bytes is ONE v_dot4_u32_u8.  The box starts at an
// arbitrary byte: v_alignbyte_b32 shifts the aligned LDS dwords so that tap 0 sits in byte 0.
// SUM / S truncated, S = sx * sy.  The reference's IEEE division of the two exact floats followed by
// truncation equals the integer quotient (the quotient is either an integer or at least 1/S away
// from one, far more than the division's rounding error).  With one divisor for the whole frame
// (rcp = 1/S, S < 4096) the quotient is floor((SUM + 0.5) * rcp): the half keeps exact multiples
// above their integer, and the product's error (< 255 * 2^-22) cannot reach the next one.
// (area_quot itself lives in vpp_device.h: the box kernel of vpp_area_box.hip shares it)
struct AXEntry { int off, sum; uint32_t w0, w1; };                // luma column: LDS offset, sum(wx), packed weights
struct ACEntry { int off, sum; uint32_t wu[4]; int pad0, pad1; }; // chroma pair column: weights on even bytes
struct AYEntry { int row, sum; uint32_t w0, w1; };                // output row: first staged row, sum(wy), packed weights

// RY > 0: the number of vertical taps is a compile-time constant (2 and 3 cover ratios up to 3): the row loops
// unroll and the dependent LDS reads (row base -> dwords) of all rows are in flight together.
template <int NW, int RY, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_dyadic_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const int ry = RY ? RY : d.ry;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH * d.rpt;
    const Footprint f = tile_footprint<M_AREA_DOWN>(d, id);

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    AXEntry *xtab = (AXEntry *)(lds_uv + d.lds_rows_uv * d.lds_cpr_uv * 16);
    ACEntry *cxtab = (ACEntry *)(xtab + tw);
    AYEntry *ytab = (AYEntry *)(cxtab + (tw >> 1));
    AYEntry *cytab = ytab + th;
    int *rby = (int *)(cytab + (th >> 1)); // LDS byte offset of every staged luma row (incl. misalignment)
    int *rbuv = rby + d.lds_rows_y;

    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, f.ylo, f.xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, f.cylo, 2 * f.cxlo, d.lds_cpr_uv, auv);
    const int ny = min(f.yhi - f.ylo + 1, d.lds_rows_y), nuv = d.luma_only ? 0 : min(f.cyhi - f.cylo + 1, d.lds_rows_uv);
    // LDS-DMA issues every chunk without holding registers; the register path (TSVPP_DMA=0, or a layout that
    // only fits compact) keeps 4 + 2 chunks per lane in flight.
    // (Deeper variants for small workgroups cost the WHOLE kernel 168 VGPRs = 3 waves per SIMD, although
    // the DMA path that normally runs needs fewer than 100.)
    const int spy = min(f.xhi - f.xlo + 1, d.lds_span_y), spuv = min(2 * (f.cxhi - f.cxlo + 1), d.lds_span_uv);
    if (d.dma) {
        stage_plane_dma(lds_y, ay, py, d.pitch_y, ny, spy, d.lds_magic_y, nthreads);
        stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, nuv, spuv, d.lds_magic_uv, nthreads);
    } else
        stage_planes<4, 2>(d, lds_y, ay, py, ny, spy, lds_uv, auv, puv, nuv, spuv, nthreads);
    const int ntab = tw + (tw >> 1) + th + (th >> 1) + d.lds_rows_y + d.lds_rows_uv;
    for (int e = threadIdx.x; e < ntab; e += nthreads) {
        int k = e;
        if (k < tw) {
            const int j = f.j_first + k;
            const AreaQRow q = d.qx[j % d.nx];
            xtab[k] = AXEntry{ (int)(d.xr * (float)j) - f.xlo, q.sum, q.w[0], q.w[1] };
            continue;
        }
        k -= tw;
        if (k < (tw >> 1)) {
            const int cj = (f.j_first >> 1) + k;
            const AreaQRow q = d.qx[cj % d.nx];
            cxtab[k] = ACEntry{ 2 * ((int)(d.xr * (float)cj) - f.cxlo), q.sum, { q.wu[0], q.wu[1], q.wu[2], q.wu[3] }, 0, 0 };
            continue;
        }
        k -= tw >> 1;
        if (k < th) {
            const int i = f.i_first + k;
            const AreaQRow q = d.qy[i % d.ny];
            ytab[k] = AYEntry{ (int)(d.yr * (float)i) - f.ylo, q.sum, q.w[0], q.w[1] };
            continue;
        }
        k -= th;
        if (k < (th >> 1)) {
            const int ci = (f.i_first >> 1) + k;
            const AreaQRow q = d.qy[ci % d.ny];
            cytab[k] = AYEntry{ (int)(d.yr * (float)ci) - f.cylo, q.sum, q.w[0], q.w[1] };
            continue;
        }
        k -= th >> 1;
        if (k < d.lds_rows_y) rby[k] = k * py.lp + ((py.m0 + k * py.pm) & 15);
        else { k -= d.lds_rows_y; rbuv[k] = k * puv.lp + ((puv.m0 + k * puv.pm) & 15); }
    }
    if (d.dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = f.j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    // thread tile = 4 columns x (2 * rpt) rows: tile decode, staging set-up and table build are paid once per 8 * rpt pixels
    for (int rp = 0; rp < d.rpt; rp++) {
    const int lyr = ly * d.rpt + rp, i0 = f.i_first + lyr * PXH;
    if (i0 >= d.dst_h) break;
    float Uf[2], Vf[2], Yf[PXH][PXW];
    { // chroma: U on even bytes, V on odd bytes of the same dwords
        const AYEntry ye = cytab[lyr];
        uint32_t su[2] = { 0, 0 }, sv[2] = { 0, 0 };
        ACEntry ce[2] = { cxtab[lx * 2], cxtab[lx * 2 + 1] };
#pragma unroll
        for (int a = 0; a < ry; a++) {
            const uint32_t wy = ((a < 4 ? ye.w0 : ye.w1) >> (8 * (a & 3))) & 255u;
            const int rb = rbuv[ye.row + a];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int A = rb + ce[c].off;
                const uint32_t *p = (const uint32_t *)(lds_uv + (A & ~3));
                const uint32_t sh = (uint32_t)A & 3u;
                uint32_t ru = 0, rv = 0;
#pragma unroll
                for (int k = 0; k < 2 * NW; k++) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(p[k + 1], p[k], sh);
                    ru = __builtin_amdgcn_udot4(v, ce[c].wu[k], ru, false);
                    rv = __builtin_amdgcn_udot4(v, ce[c].wu[k] << 8, rv, false);
                }
                su[c] += wy * ru;
                sv[c] += wy * rv;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Uf[c] = area_quot(su[c], ce[c].sum, ye.sum, d.area_rcp);
            Vf[c] = area_quot(sv[c], ce[c].sum, ye.sum, d.area_rcp);
        }
    }
    {
        AXEntry xe[PXW];
#pragma unroll
        for (int c = 0; c < PXW; c++) xe[c] = xtab[lx * PXW + c];
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const AYEntry ye = ytab[lyr * PXH + r];
            uint32_t sum[PXW] = { 0, 0, 0, 0 };
#pragma unroll
            for (int a = 0; a < ry; a++) {
                const uint32_t wy = ((a < 4 ? ye.w0 : ye.w1) >> (8 * (a & 3))) & 255u;
                const int rb = rby[ye.row + a];
#pragma unroll
                for (int c = 0; c < PXW; c++) {
                    const int A = rb + xe[c].off;
                    const uint32_t *p = (const uint32_t *)(lds_y + (A & ~3));
                    const uint32_t sh = (uint32_t)A & 3u;
                    uint32_t rs = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(p[1], p[0], sh), xe[c].w0, 0u, false);
                    if constexpr (NW == 2) rs = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(p[2], p[1], sh), xe[c].w1, rs, false);
                    sum[c] += wy * rs;
                }
            }
#pragma unroll
            for (int c = 0; c < PXW; c++) Yf[r][c] = area_quot(sum[c], xe[c].sum, ye.sum, d.area_rcp);
        }
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

// ----------------------------------------------------------------------------------------------
// Dyadic AREA for LARGE ratios (>= 2 in both axes), straight from global memory.  A 6x6 box tile
// needs 36 source bytes per output pixel: staged in LDS that is ~30 KiB per 64 threads, i.e. five
// waves per CU, and the kernel becomes latency-bound.  Here every thread reads the 2-3 aligned
// dwords that cover one box row itself (adjacent lanes read adjacent, slightly overlapping spans,
// so the wave's requests still coalesce into full lines) and reduces them with
// v_alignbyte_b32 + v_dot4_u32_u8 as above: no LDS, no barrier, occupancy limited by VGPRs only.
// A dword that the box does not reach is re-pointed at dword 0, so nothing is read beyond the
// last needed byte's dword (no over-read past the end of the plane).
typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
// N + 1 aligned dwords starting at the dword that holds byte `a`.  WIDE: one vector load (the row is
// not the plane's last one, so the few bytes past the box still belong to the plane); otherwise
// dword by dword, and a dword the box does not reach is re-pointed at dword 0 -- nothing is read
// beyond the dword of the last needed byte.
// `plane` is the frame's (wave-uniform) plane pointer rounded DOWN to a dword, `pm` the bytes it was
// rounded by and `off` a 32-bit byte offset from the true pointer: the loads use the SGPR-base +
// VGPR-offset addressing mode and the offset (off + pm) & ~3 can never go negative.
template <int N, bool WIDE>
__device__ __forceinline__ void load_span(const uint8_t *plane, uint32_t pm, uint32_t off, int nbytes, uint32_t (&dw)[N + 1], uint32_t &sh) {
    const uint32_t o = off + pm;
    sh = o & 3u;
    const uint32_t *p = (const uint32_t *)(plane + (o - sh)); // aligned dword of the first byte
    if constexpr (WIDE) {
        if constexpr (N == 1) {
            const u32x2a4 v = *(const u32x2a4 *)p;
            dw[0] = v.x; dw[1] = v.y;
        } else if constexpr (N == 2) {
            const u32x3a4 v = *(const u32x3a4 *)p;
            dw[0] = v.x; dw[1] = v.y; dw[2] = v.z;
        } else {
            static_assert(N == 4, "span width");
            const u32x4a4 v = *(const u32x4a4 *)p;
            dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
            dw[4] = p[4];
        }
    } else {
#pragma unroll
        for (int k = 0; k <= N; k++) dw[k] = p[(4 * k < (int)sh + nbytes) ? k : 0];
    }
}

template <int NW, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_direct_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = tile_col0(d, id.tx, d.tx * PXW) + lx * PXW, i0 = (id.ty * d.ty + ly) * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const uint32_t ym = (uint32_t)((uintptr_t)t.y[id.frame] & 3), uvm = (uint32_t)((uintptr_t)t.uv[id.frame] & 3);
    const uint8_t *Y = t.y[id.frame] - ym, *UV = t.uv[id.frame] - uvm; // dword-aligned bases (see load_span)
    const int ci = i0 >> 1, cj0 = j0 >> 1;

    float Uf[2], Vf[2], Yf[PXH][PXW];
    { // chroma: U on even bytes, V on odd bytes of the same dwords
        const AreaQRow qy = d.qy[ci % d.ny];
        const int y0 = (int)(d.yr * (float)ci);
        AreaQRow qx[2];
        int xo[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            qx[c] = d.qx[(cj0 + c) % d.nx];
            xo[c] = 2 * (int)(d.xr * (float)(cj0 + c));
        }
        uint32_t su[2] = { 0, 0 }, sv[2] = { 0, 0 };
        for (int a = 0; a < d.ry; a++) {
            const uint32_t wy = (qy.w[a >> 2] >> (8 * (a & 3))) & 255u;
            const uint32_t row = (uint32_t)(y0 + a) * (uint32_t)d.pitch_uv;
            const bool wide = (y0 + a) < (d.src_h >> 1) - 1; // not the plane's last row
#pragma unroll
            for (int c = 0; c < 2; c++) {
                uint32_t dw[2 * NW + 1], sh;
                if (wide) load_span<2 * NW, true>(UV, uvm, row + (uint32_t)xo[c], 2 * d.rx, dw, sh);
                else load_span<2 * NW, false>(UV, uvm, row + (uint32_t)xo[c], 2 * d.rx, dw, sh);
                uint32_t ru = 0, rv = 0;
#pragma unroll
                for (int k = 0; k < 2 * NW; k++) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(dw[k + 1], dw[k], sh);
                    ru = __builtin_amdgcn_udot4(v, qx[c].wu[k], ru, false);
                    rv = __builtin_amdgcn_udot4(v, qx[c].wu[k] << 8, rv, false);
                }
                su[c] += wy * ru;
                sv[c] += wy * rv;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Uf[c] = area_quot(su[c], qx[c].sum, qy.sum, d.area_rcp);
            Vf[c] = area_quot(sv[c], qx[c].sum, qy.sum, d.area_rcp);
        }
    }
    {
        int xo[PXW], xs[PXW];
        uint32_t w0[PXW], w1[PXW];
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            const AreaQRow q = d.qx[(j0 + c) % d.nx];
            xo[c] = (int)(d.xr * (float)(j0 + c));
            xs[c] = q.sum;
            w0[c] = q.w[0];
            w1[c] = q.w[1];
        }
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const AreaQRow qy = d.qy[(i0 + r) % d.ny];
            const int y0 = (int)(d.yr * (float)(i0 + r));
            uint32_t sum[PXW] = { 0, 0, 0, 0 };
            for (int a = 0; a < d.ry; a++) {
                const uint32_t wy = (qy.w[a >> 2] >> (8 * (a & 3))) & 255u;
                const uint32_t row = (uint32_t)(y0 + a) * (uint32_t)d.pitch_y;
                const bool wide = (y0 + a) < d.src_h - 1; // not the plane's last row
#pragma unroll
                for (int c = 0; c < PXW; c++) {
                    uint32_t dw[NW + 1], sh;
                    if (wide) load_span<NW, true>(Y, ym, row + (uint32_t)xo[c], d.rx, dw, sh);
                    else load_span<NW, false>(Y, ym, row + (uint32_t)xo[c], d.rx, dw, sh);
                    uint32_t rs = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(dw[1], dw[0], sh), w0[c], 0u, false);
                    if constexpr (NW == 2) rs = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(dw[2], dw[1], sh), w1[c], rs, false);
                    sum[c] += wy * rs;
                }
            }
#pragma unroll
            for (int c = 0; c < PXW; c++) Yf[r][c] = area_quot(sum[c], xs[c], qy.sum, d.area_rcp);
        }
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

// ----------------------------------------------------------------------------------------------
// AREA down-scale with arbitrary (non-dyadic) weights at ratios >= 2, straight from global memory.
// The reference's float accumulation order (rows outer, taps inner; src/Resize.cu:164-173) is kept
// per value, so sums are bit-identical: sum = fma(float(tap), wx[b] * wy[a], sum) -- fused, as in the reference's
// binary -- and div += wx[b] * wy[a] (not fused there: pinned by the CRC goldens).
// Same memory scheme as vpp_area_direct_kernel (per-pixel aligned dwords + v_alignbyte so that tap 0
// is byte 0); the weight rows are zero-padded to 4 * NK taps, and a zero weight adds exactly 0 to
// both accumulators, so four taps are always processed per shifted dword.  Pixel pairs / (U, V)
// pairs share the packed VALU.
typedef float vf4a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int NK, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_direct_float_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = tile_col0(d, id.tx, d.tx * PXW) + lx * PXW, i0 = (id.ty * d.ty + ly) * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const uint32_t ym = (uint32_t)((uintptr_t)t.y[id.frame] & 3), uvm = (uint32_t)((uintptr_t)t.uv[id.frame] & 3);
    const uint8_t *Y = t.y[id.frame] - ym, *UV = t.uv[id.frame] - uvm; // dword-aligned bases (see load_span)
    const int ci = i0 >> 1, cj0 = j0 >> 1;

    float Uf[2], Vf[2], Yf[PXH][PXW];
    { // chroma: U and V of one pair share every weight
        const float *wyrow = d.paty4 + (ci % d.ny) * 4 * d.nky;
        const int y0 = (int)(d.yr * (float)ci);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int x0 = 2 * (int)(d.xr * (float)(cj0 + c));
            const float *wxrow = d.patx4 + ((cj0 + c) % d.nx) * 4 * NK;
            vf4 wx[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) wx[k] = *(const vf4a4 *)(wxrow + 4 * k);
            f2 acc = { 0.0f, 0.0f };
            float div = 0.0f;
            for (int a = 0; a < d.ry; a++) {
                const float wy = wyrow[a];
                const uint32_t row = (uint32_t)(y0 + a) * (uint32_t)d.pitch_uv + (uint32_t)x0;
                uint32_t dw[2 * NK + 1], sh;
                const bool wide = (y0 + a) < (d.src_h >> 1) - 1;
                if constexpr (NK == 1) {
                    if (wide) load_span<2, true>(UV, uvm, row, 2 * d.rx, dw, sh); else load_span<2, false>(UV, uvm, row, 2 * d.rx, dw, sh);
                } else {
                    sh = (uvm + row) & 3u; // 2 * NK + 1 dwords, one by one (spans of 4-6 dwords)
                    const uint32_t *p = (const uint32_t *)(UV + (row + uvm - sh));
#pragma unroll
                    for (int k = 0; k <= 2 * NK; k++) dw[k] = p[(wide || 4 * k < (int)sh + 2 * d.rx) ? k : 0];
                }
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t v0 = __builtin_amdgcn_alignbyte(dw[2 * k + 1], dw[2 * k], sh);     // U0 V0 U1 V1
                    const uint32_t v1 = __builtin_amdgcn_alignbyte(dw[2 * k + 2], dw[2 * k + 1], sh); // U2 V2 U3 V3
                    const float wv[4] = { wx[k].x, wx[k].y, wx[k].z, wx[k].w };
                    const uint32_t vv[2] = { v0, v1 };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t q = vv[b >> 1] >> (16 * (b & 1));
                        const float wgt = wv[b] * wy;
                        div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)(q & 255), (float)((q >> 8) & 255) }, (f2){ wgt, wgt }, acc);
                    }
                }
            }
            Uf[c] = __builtin_truncf(acc.x / div);
            Vf[c] = __builtin_truncf(acc.y / div);
        }
    }
    { // luma: horizontally adjacent pixel pairs share the packed VALU
        int x0[PXW];
        vf4 wx[PXW][NK];
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            x0[c] = (int)(d.xr * (float)(j0 + c));
            const float *wxrow = d.patx4 + ((j0 + c) % d.nx) * 4 * NK;
#pragma unroll
            for (int k = 0; k < NK; k++) wx[c][k] = *(const vf4a4 *)(wxrow + 4 * k);
        }
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const float *wyrow = d.paty4 + ((i0 + r) % d.ny) * 4 * d.nky;
            const int y0 = (int)(d.yr * (float)(i0 + r));
            f2 acc[2] = { { 0.0f, 0.0f }, { 0.0f, 0.0f } }, div[2] = { { 0.0f, 0.0f }, { 0.0f, 0.0f } };
            for (int a = 0; a < d.ry; a++) {
                const float wy = wyrow[a];
                const uint32_t rowo = (uint32_t)(y0 + a) * (uint32_t)d.pitch_y;
                const bool wide = (y0 + a) < d.src_h - 1;
                uint32_t dw[PXW][NK + 1], sh[PXW];
#pragma unroll
                for (int c = 0; c < PXW; c++) {
                    if constexpr (NK <= 2) {
                        if (wide) load_span<NK, true>(Y, ym, rowo + (uint32_t)x0[c], d.rx, dw[c], sh[c]);
                        else load_span<NK, false>(Y, ym, rowo + (uint32_t)x0[c], d.rx, dw[c], sh[c]);
                    } else {
                        sh[c] = (ym + rowo + (uint32_t)x0[c]) & 3u;
                        const uint32_t *p = (const uint32_t *)(Y + (rowo + (uint32_t)x0[c] + ym - sh[c]));
#pragma unroll
                        for (int k = 0; k <= NK; k++) dw[c][k] = p[(wide || 4 * k < (int)sh[c] + d.rx) ? k : 0];
                    }
                }
#pragma unroll
                for (int k = 0; k < NK; k++) {
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const uint32_t va = __builtin_amdgcn_alignbyte(dw[2 * p][k + 1], dw[2 * p][k], sh[2 * p]);
                        const uint32_t vb = __builtin_amdgcn_alignbyte(dw[2 * p + 1][k + 1], dw[2 * p + 1][k], sh[2 * p + 1]);
                        const float wa[4] = { wx[2 * p][k].x, wx[2 * p][k].y, wx[2 * p][k].z, wx[2 * p][k].w };
                        const float wb[4] = { wx[2 * p + 1][k].x, wx[2 * p + 1][k].y, wx[2 * p + 1][k].z, wx[2 * p + 1][k].w };
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const f2 wgt = (f2){ wa[b], wb[b] } * (f2){ wy, wy };
                            div[p] = div[p] + wgt;
                            acc[p] = __builtin_elementwise_fma((f2){ (float)((va >> (8 * b)) & 255), (float)((vb >> (8 * b)) & 255) }, wgt, acc[p]);
                        }
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                Yf[r][2 * p] = __builtin_truncf(acc[p].x / div[p].x);
                Yf[r][2 * p + 1] = __builtin_truncf(acc[p].y / div[p].y);
            }
        }
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

// ----------------------------------------------------------------------------------------------
// Large-ratio AREA with float weights, one output COLUMN per lane.  The kernel above gives every lane a 4-column
// thread tile: neighbouring lanes then read windows 4 * xr (34 bytes at 1080p -> 224) apart and each load instruction
// of a wave touches ~34 cache lines for 128-512 useful bytes -- the texture-address path, not HBM or the VALU, bounds
// it (1080p -> 224x224: 30 % of the ROI roofline).  Here the sampling runs with lane = column (stride xr bytes: a wave's
// load covers ONE contiguous 64 * xr byte run, ~9 lines), two output rows per lane at a time on float pairs, and the
// resized samples go through a 12 KiB LDS tile into the usual 2 x 4 thread tiles for colour conversion and stores.
// Workgroup = 16 x 16 thread tiles = 64 columns x 32 rows; wave w samples rows 8 w .. 8 w + 7 (chroma 4 w .. 4 w + 3).
// Arithmetic and accumulation order are the reference's (src/Resize.cu:160-178), as in the kernel above.
template <int NK> struct ColWeights { vf4 w[NK]; };

template <int NK>
__device__ __forceinline__ void cols_load(const uint8_t *plane, uint32_t pm, uint32_t off, int nbytes, bool wide, uint32_t (&dw)[NK + 1], uint32_t &sh) {
    if constexpr (NK <= 2) {
        if (wide) load_span<NK, true>(plane, pm, off, nbytes, dw, sh);
        else load_span<NK, false>(plane, pm, off, nbytes, dw, sh);
    } else {
        const uint32_t o = off + pm;
        sh = o & 3u;
        const uint32_t *p = (const uint32_t *)(plane + (o - sh));
#pragma unroll
        for (int k = 0; k <= NK; k++) dw[k] = p[(wide || 4 * k < (int)sh + nbytes) ? k : 0];
    }
}

// TH = 32: the round-1 tile (each wave samples 8 rows: 8 values per lane, as many as a 2 x 4 thread tile).  TH = 8 (round 2):
// each wave samples ONE row pair -- four times as many waves for the same frame.  At these ratios a value is a serial chain
// of 45-100 taps behind strided loads, and 1080p -> 224 x 224 x 64 frames is only 7168 waves of the round-1 shape: on average
// 2.2 resident waves per SIMD (profiles/r02_a224_pmc.txt), i.e. latency-bound by lack of parallelism, not by VALU or HBM.
template <int NK, int TH, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_cols_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr int TW = 64; // output tile of the workgroup: 64 columns x TH rows (16 x TH / 2 thread tiles), always 256 threads
    static_assert(TH == 32 || TH == 8, "tile height");
    __shared__ __attribute__((aligned(16))) float yt[TH][TW];
    __shared__ __attribute__((aligned(16))) f2 uvt[TH / 2][TW / 2];
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int j_first = tile_col0(d, id.tx, TW), i_first = id.ty * TH;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t ym = (uint32_t)((uintptr_t)t.y[id.frame] & 3), uvm = (uint32_t)((uintptr_t)t.uv[id.frame] & 3);
    const uint8_t *Y = t.y[id.frame] - ym, *UV = t.uv[id.frame] - uvm; // dword-aligned bases (see load_span)

    { // luma: lane = column; columns / rows past the frame are computed on the last valid one and never used
        const int j = min(j_first + lane, d.dst_w - 1);
        const int x0 = (int)(d.xr * (float)j);
        const float *wxrow = d.patx4 + (j % d.nx) * 4 * NK;
        vf4 wx[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) wx[k] = *(const vf4a4 *)(wxrow + 4 * k);
        for (int rp = 0; rp < TH / 8; rp++) {
            const int r0 = (TH / 4) * wave + 2 * rp;
            const int iA = min(i_first + r0, d.dst_h - 1), iB = min(i_first + r0 + 1, d.dst_h - 1);
            const float *wyA = d.paty4 + (iA % d.ny) * 4 * d.nky, *wyB = d.paty4 + (iB % d.ny) * 4 * d.nky;
            // (wave-uniform, but evaluated on the vector unit -- gfx950 has no scalar float unit: back into SGPRs, so that the loads' edge test is a scalar branch)
            const int yA = __builtin_amdgcn_readfirstlane((int)(d.yr * (float)iA)), yB = __builtin_amdgcn_readfirstlane((int)(d.yr * (float)iB));
            f2 acc = { 0.0f, 0.0f }, div = { 0.0f, 0.0f };
            // (round 6) the rows of tap row a + 1 are requested BEFORE tap row a is accumulated: the kernel waits (61 % of its wave cycles in SQ_WAIT_ANY at 1080p -> 300 x 300,
            // VALU a third of the launch: profiles/HISTORY.md) -- two row pairs in flight per wave instead of one: 1080p -> 300 x 300 543 -> 491 us per 512 frames
            // (profiles/r06_area_cols_ab.txt).  Measured and NOT taken: the divisor from the host-built table (LaunchDesc::area_div) instead of the running sum -- its two
            // scattered loads and integer modulos per lane cost more than the packed add they save (300^2: 524 us, 416^2: 713 against 643).
            // tap rows [0, a_wide) lie above the plane's last row for BOTH output rows: unconditional 12- / 8-byte loads, no branch between them (the edge test
            // inside the loop made the compiler wait for every load before it issued the next: `s_waitcnt vmcnt(0)` in front of each in round 5's ISA); the plane's
            // last row -- bottom tiles only -- keeps the dword-by-dword loads that never read past it
            const int a_wide = min(max(d.src_h - 1 - yB, 0), d.ry);
            auto accum = [&](int a, const uint32_t (&qa)[NK + 1], const uint32_t (&qb)[NK + 1], uint32_t qsa, uint32_t qsb) {
                const f2 wy = { wyA[a], wyB[a] };
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t va = __builtin_amdgcn_alignbyte(qa[k + 1], qa[k], qsa), vb = __builtin_amdgcn_alignbyte(qb[k + 1], qb[k], qsb);
                    const float wk[4] = { wx[k].x, wx[k].y, wx[k].z, wx[k].w };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const f2 wgt = (f2){ wk[b], wk[b] } * wy;
                        div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)((va >> (8 * b)) & 255), (float)((vb >> (8 * b)) & 255) }, wgt, acc);
                    }
                }
            };
            auto load_wide = [&](int a, uint32_t (&qa)[NK + 1], uint32_t (&qb)[NK + 1], uint32_t &qsa, uint32_t &qsb) {
                if constexpr (NK <= 2) {
                    load_span<NK, true>(Y, ym, (uint32_t)(yA + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, qa, qsa);
                    load_span<NK, true>(Y, ym, (uint32_t)(yB + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, qb, qsb);
                } else {
                    cols_load<NK>(Y, ym, (uint32_t)(yA + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, true, qa, qsa);
                    cols_load<NK>(Y, ym, (uint32_t)(yB + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, true, qb, qsb);
                }
            };
            int a = 0;
            if (a_wide > 0) {
                uint32_t da[NK + 1], db[NK + 1], sa, sb;
                load_wide(0, da, db, sa, sb);
                for (; a + 1 < a_wide; a++) { // tap row a + 1 is in flight while tap row a is accumulated
                    uint32_t na[NK + 1], nb[NK + 1], nsa, nsb;
                    load_wide(a + 1, na, nb, nsa, nsb);
                    accum(a, da, db, sa, sb);
#pragma unroll
                    for (int k = 0; k <= NK; k++) { da[k] = na[k]; db[k] = nb[k]; }
                    sa = nsa;
                    sb = nsb;
                }
                accum(a, da, db, sa, sb);
                a++;
            }
            for (; a < d.ry; a++) { // the plane's last row
                uint32_t da[NK + 1], db[NK + 1], sa, sb;
                cols_load<NK>(Y, ym, (uint32_t)(yA + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, (yA + a) < d.src_h - 1, da, sa);
                cols_load<NK>(Y, ym, (uint32_t)(yB + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, (yB + a) < d.src_h - 1, db, sb);
                accum(a, da, db, sa, sb);
            }
            yt[r0][lane] = __builtin_truncf(acc.x / div.x);
            yt[r0 + 1][lane] = __builtin_truncf(acc.y / div.y);
        }
    }
    if (!d.luma_only && (TH == 32 || wave < 2)) { // chroma: lanes 0-31 / 32-63 = the 32 chroma columns of two chroma rows; (U, V) as a pair
        const int cw = d.dst_w >> 1, chh = d.dst_h >> 1;
        const int cj = min((j_first >> 1) + (lane & 31), cw - 1);
        const int x0 = 2 * (int)(d.xr * (float)cj);
        const float *wxrow = d.patx4 + (cj % d.nx) * 4 * NK;
        vf4 wx[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) wx[k] = *(const vf4a4 *)(wxrow + 4 * k);
        for (int q = 0; q < (TH == 32 ? 2 : 1); q++) {
            const int cr = (TH == 32 ? 4 * wave + 2 * q : 2 * wave) + (lane >> 5);
            const int ci = min((i_first >> 1) + cr, chh - 1);
            const float *wyrow = d.paty4 + (ci % d.ny) * 4 * d.nky;
            const int y0 = (int)(d.yr * (float)ci);
            f2 acc = { 0.0f, 0.0f };
            float div = 0.0f;
            const int ch_rows = d.src_h >> 1;
            // (as the luma loop: the rows above the plane's last one -- for the LOWER of the wave's two chroma rows, lanes 32-63 -- with unconditional loads, one tap row ahead)
            const int a_wide = min(max(ch_rows - 1 - __builtin_amdgcn_readlane(y0, 63), 0), d.ry);
            auto accum = [&](int a, const uint32_t (&q)[2 * NK + 1], uint32_t qsh) {
                const float wy = wyrow[a];
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t v0 = __builtin_amdgcn_alignbyte(q[2 * k + 1], q[2 * k], qsh);     // U0 V0 U1 V1
                    const uint32_t v1 = __builtin_amdgcn_alignbyte(q[2 * k + 2], q[2 * k + 1], qsh); // U2 V2 U3 V3
                    const float wv[4] = { wx[k].x, wx[k].y, wx[k].z, wx[k].w };
                    const uint32_t vv[2] = { v0, v1 };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t qq = vv[b >> 1] >> (16 * (b & 1));
                        const float wgt = wv[b] * wy;
                        div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)(qq & 255), (float)((qq >> 8) & 255) }, (f2){ wgt, wgt }, acc);
                    }
                }
            };
            auto load_row = [&](int a, bool wide, uint32_t (&q)[2 * NK + 1], uint32_t &qsh) {
                const uint32_t row = (uint32_t)(y0 + a) * (uint32_t)d.pitch_uv + (uint32_t)x0;
                if constexpr (NK == 1) {
                    if (wide) load_span<2, true>(UV, uvm, row, 2 * d.rx, q, qsh);
                    else load_span<2, false>(UV, uvm, row, 2 * d.rx, q, qsh);
                } else {
                    qsh = (uvm + row) & 3u;
                    const uint32_t *p = (const uint32_t *)(UV + (row + uvm - qsh));
#pragma unroll
                    for (int k = 0; k <= 2 * NK; k++) q[k] = p[(wide || 4 * k < (int)qsh + 2 * d.rx) ? k : 0];
                }
            };
            int a = 0;
            if (a_wide > 0) {
                uint32_t dw[2 * NK + 1], sh;
                load_row(0, true, dw, sh);
                for (; a + 1 < a_wide; a++) {
                    uint32_t nw[2 * NK + 1], nsh;
                    load_row(a + 1, true, nw, nsh);
                    accum(a, dw, sh);
#pragma unroll
                    for (int k = 0; k <= 2 * NK; k++) dw[k] = nw[k];
                    sh = nsh;
                }
                accum(a, dw, sh);
                a++;
            }
            for (; a < d.ry; a++) {
                uint32_t dw[2 * NK + 1], sh;
                load_row(a, (y0 + a) < ch_rows - 1, dw, sh);
                accum(a, dw, sh);
            }
            uvt[cr][lane & 31] = (f2){ __builtin_truncf(acc.x / div), __builtin_truncf(acc.y / div) };
        }
    }
    __syncthreads();

    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    if (ly >= TH / 2) return; // TH = 8: one wave converts and stores the 64 x 8 tile
    const int j0 = j_first + lx * PXW, i0 = i_first + ly * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    float Uf[2], Vf[2], Yf[PXH][PXW];
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        const vf4 v = *(const vf4 *)&yt[ly * PXH + r][lx * PXW];
        Yf[r][0] = v.x;
        Yf[r][1] = v.y;
        Yf[r][2] = v.z;
        Yf[r][3] = v.w;
    }
    {
        const vf4 c = *(const vf4 *)&uvt[ly][lx * 2];
        Uf[0] = c.x;
        Vf[0] = c.y;
        Uf[1] = c.z;
        Vf[1] = c.w;
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

// ----------------------------------------------------------------------------------------------
// Point-sampling kernel: NEAREST, and BILINEAR / BICUBIC requests whose weights are all zero.
// Every output row needs exactly ONE source row and every output column one source byte (pair),
// so only those rows are staged -- a 3x down-scale reads a third of the luma plane -- one LDS row
// per output row, and a tap is a single LDS byte read through small per-tile offset tables.
template <int KIND, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_point_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH;
    const int j_first = tile_col0(d, id.tx, tw), i_first = id.ty * th;
    const int j_last = min(j_first + tw, d.dst_w) - 1, i_last = min(i_first + th, d.dst_h) - 1;
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;
    // column extent of the tile in both planes (coordinates are monotonic in the output index)
    const int xlo = min(max(point_coord<KIND>(j_first, d.xr, d.src_w), 0), d.src_w - 1);
    const int xhi = min(max(point_coord<KIND>(j_last, d.xr, d.src_w), 0), d.src_w - 1);
    const int cxlo = min(max(point_coord<KIND>(j_first >> 1, d.xr, d.src_w), 0), cw - 1);
    const int cxhi = min(max(point_coord<KIND>(j_last >> 1, d.xr, d.src_w), 0), cw - 1);
    const int span_y = min(xhi - xlo + 1, d.lds_span_y), span_uv = min(2 * (cxhi - cxlo + 1), d.lds_span_uv);
    const int ny = i_last - i_first + 1, nuv = d.luma_only ? 0 : (i_last >> 1) - (i_first >> 1) + 1;

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + th * d.lds_cpr_y * 16;
    int *xtab = (int *)(lds_uv + (th >> 1) * d.lds_cpr_uv * 16);
    int *cxtab = xtab + tw;
    int *ytab = cxtab + (tw >> 1); // LDS byte offset of each staged luma row (incl. its misalignment)
    int *cytab = ytab + th;

    // stage: luma row r <- source row y(i_first + r); chroma row r <- source row y(ci_first + r)
    {
        const StageLane ln = stage_lane(d.lds_slot_y, nthreads);
        const int lp = 16 * d.lds_cpr_y;
        for (int r = ln.r0; r < ny; r += ln.rstep) {
            const int y = min(max(point_coord<KIND>(i_first + r, d.yr, d.src_h), 0), d.src_h - 1);
            const uint8_t *a = t.y[id.frame] + (size_t)y * (size_t)d.pitch_y + (size_t)xlo;
            const int mis = (int)((uintptr_t)a & 15);
            if (ln.ch == 0) ytab[r] = r * lp + mis;
            if (ln.ch < d.lds_cpr_y && 16 * ln.ch < mis + span_y) *(uint4 *)(lds_y + r * lp + 16 * ln.ch) = *(const uint4 *)(a - mis + 16 * ln.ch);
        }
    }
    {
        const StageLane ln = stage_lane(d.lds_slot_uv, nthreads);
        const int lp = 16 * d.lds_cpr_uv;
        for (int r = ln.r0; r < nuv; r += ln.rstep) {
            const int y = min(max(point_coord<KIND>((i_first >> 1) + r, d.yr, d.src_h), 0), chh - 1);
            const uint8_t *a = t.uv[id.frame] + (size_t)y * (size_t)d.pitch_uv + (size_t)(2 * cxlo);
            const int mis = (int)((uintptr_t)a & 15);
            if (ln.ch == 0) cytab[r] = r * lp + mis;
            if (ln.ch < d.lds_cpr_uv && 16 * ln.ch < mis + span_uv) *(uint4 *)(lds_uv + r * lp + 16 * ln.ch) = *(const uint4 *)(a - mis + 16 * ln.ch);
        }
    }
    for (int e = threadIdx.x; e < tw + (tw >> 1); e += nthreads) {
        if (e < tw) xtab[e] = min(max(point_coord<KIND>(j_first + e, d.xr, d.src_w), 0), d.src_w - 1) - xlo;
        else cxtab[e - tw] = 2 * (min(max(point_coord<KIND>((j_first >> 1) + e - tw, d.xr, d.src_w), 0), cw - 1) - cxlo);
    }
    __syncthreads();

    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = j_first + lx * PXW, i0 = i_first + ly * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const int4 xo = *(const int4 *)(xtab + lx * PXW);
    const int2 cxo = *(const int2 *)(cxtab + lx * 2);
    const int2 yo = *(const int2 *)(ytab + ly * PXH);
    const int cyo = cytab[ly];
    float Uf[2], Vf[2], Yf[PXH][PXW];
    const uint8_t *cr = lds_uv + cyo;
    Uf[0] = (float)cr[cxo.x];
    Vf[0] = (float)cr[cxo.x + 1];
    Uf[1] = (float)cr[cxo.y];
    Vf[1] = (float)cr[cxo.y + 1];
    const int xs[PXW] = { xo.x, xo.y, xo.z, xo.w }, ys[PXH] = { yo.x, yo.y };
#pragma unroll
    for (int r = 0; r < PXH; r++)
#pragma unroll
        for (int c = 0; c < PXW; c++) Yf[r][c] = (float)lds_y[ys[r] + xs[c]];
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

// No resize, uint8 Y800 / NV12 outputs: the output IS the (cropped) source planes made tight (reference src/ColorConversion.cu:95-105, 211-233) -- a
// strided copy.  The colour-only kernel below moves it 4 bytes per lane and instruction and reached 0.53 of the roofline (Y800, r03_output_matrix.txt):
// memory instructions cost the texture addresser per LANE, not per byte (profiles/r04_bicubic_r32_pmc.txt), so the copy takes 16 bytes per lane and
// four rows per thread (all loads first): thread tile = 16 columns x 4 rows (+ 2 chroma rows), workgroup tile = (16 tx) x (4 ty).
typedef uint32_t cp_x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t cp_x4a __attribute__((ext_vector_type(4)));
template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_copy16_kernel(const LaunchDesc d, const FrameTable t) {
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = (id.tx * d.tx + lx) * 16, i0 = (id.ty * d.ty + ly) * 4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    const uint8_t *py = t.y[id.frame] + (size_t)i0 * (size_t)d.pitch_y + (size_t)j0;
    cp_x4 y[4], c[2];
#pragma unroll
    for (int r = 0; r < 4; r++) y[r] = *(const cp_x4 *)(py + (size_t)r * (size_t)d.pitch_y);
    if constexpr (OUT == O_NV12_U8) {
        const uint8_t *pc = t.uv[id.frame] + (size_t)(i0 >> 1) * (size_t)d.pitch_uv + (size_t)j0;
#pragma unroll
        for (int r = 0; r < 2; r++) c[r] = *(const cp_x4 *)(pc + (size_t)r * (size_t)d.pitch_uv);
    }
    uint8_t *out = (uint8_t *)t.out[id.frame];
    const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;
    auto st = [&](uint32_t off, const cp_x4 &v) {
        const cp_x4a a = { v.x, v.y, v.z, v.w };
        if (d.nt_stores) st16_nt(out, off, (nt_u32x4){ v.x, v.y, v.z, v.w }, d.nt_stores); // (inline asm: see st8_nt, vpp_device.h)
        else *(cp_x4a *)(out + off) = a;
    };
#pragma unroll
    for (int r = 0; r < 4; r++) st((uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0, y[r]);
    if constexpr (OUT == O_NV12_U8) {
#pragma unroll
        for (int r = 0; r < 2; r++) st(plane + (uint32_t)((i0 >> 1) + r) * (uint32_t)d.dst_w + (uint32_t)j0, c[r]);
    }
}

// ----------------------------------------------------------------------------------------------
// Colour-only kernel (no resize; crop is already folded into the pointers): thread = 2 rows x 4
// pixels, 4-byte coalesced luma loads, one 4-byte chroma load (2 pairs) shared by the two rows.
// Every store instruction of a wave covers one contiguous run (lane stride 16 B): two 16-byte
// pieces per lane would leave each instruction writing half lines.  Needs 4-byte aligned rows.
template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_color_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = (id.tx * d.tx + lx) * PXW;
    const int i0 = (id.ty * d.ty + ly) * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused; no shifted tile here: the plane reads are aligned dwords)
    const uint8_t *py = t.y[id.frame] + (size_t)i0 * (size_t)d.pitch_y + (size_t)j0;
    const uint32_t yw[2] = { *(const uint32_t *)py, *(const uint32_t *)(py + d.pitch_y) };
    const uint32_t c = *(const uint32_t *)(t.uv[id.frame] + (size_t)(i0 >> 1) * (size_t)d.pitch_uv + (size_t)j0);
    const float Uf[2] = { (float)(c & 255), (float)((c >> 16) & 255) };
    const float Vf[2] = { (float)((c >> 8) & 255), (float)(c >> 24) };
    float Yf[PXH][PXW];
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        Yf[r][0] = (float)(yw[r] & 255);
        Yf[r][1] = (float)((yw[r] >> 8) & 255);
        Yf[r][2] = (float)((yw[r] >> 16) & 255);
        Yf[r][3] = (float)(yw[r] >> 24);
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

// Launch, or -- when the caller only wants to know what WOULD run (tsvpp_describe, CPU tests of the selection
// logic) -- record the choice and launch nothing.
#define TSVPP_LAUNCH_OR_DESCRIBE(NAME, KERNEL, GRID, BLOCK, LDS)                                   \
    do {                                                                               \
        if (info) {                                                                    \
            info->kernel = NAME;                                                       \
            info->grid = (int)(GRID).x;                                                \
            info->lds_bytes = (int)(LDS);                                              \
        } else {                                                                       \
            TSVPP_LAUNCH(KERNEL, GRID, BLOCK, LDS, stream, d, t);                \
        }                                                                              \
    } while (0)

template <int OUT>
static hipError_t launch_point(const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    switch (d.point_kind) {
    case PK_NEAREST: TSVPP_LAUNCH_OR_DESCRIBE("vpp_point_kernel<PK_NEAREST, OUT>", (vpp_point_kernel<PK_NEAREST, OUT>), grid, block, lds_bytes); break;
    case PK_BILINEAR0: TSVPP_LAUNCH_OR_DESCRIBE("vpp_point_kernel<PK_BILINEAR0, OUT>", (vpp_point_kernel<PK_BILINEAR0, OUT>), grid, block, lds_bytes); break;
    case PK_BICUBIC0: TSVPP_LAUNCH_OR_DESCRIBE("vpp_point_kernel<PK_BICUBIC0, OUT>", (vpp_point_kernel<PK_BICUBIC0, OUT>), grid, block, lds_bytes); break;
    default: return hipErrorInvalidValue;
    }
    return info ? hipSuccess : hipGetLastError();
}

template <int MODE, int OUT>
static hipError_t launch_mo(bool vec, bool staged, LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if constexpr (MODE == M_NEAREST || MODE == M_BILINEAR || MODE == M_BICUBIC) {
        if (d.r32 >= 100) return launch_point_rn((OutKind)OUT, d, t, stream, info);
    }
    if constexpr (MODE == M_NEAREST || MODE == M_AREA_UP) {
        if (d.r32 == 20) return launch_point_rn((OutKind)OUT, d, t, stream, info); // pixel replication at 1 : 2 (vpp_point_rn.hip)
    }
    if constexpr (MODE == M_NEAREST || MODE == M_BILINEAR || MODE == M_BICUBIC) { // row-segment kernel: sparse BILINEAR, and (round 6) the point samplers at sparse ratios
        if (d.bil_rows && (MODE == M_BILINEAR || d.point_kind != PK_NONE)) return launch_bilinear_rows((OutKind)OUT, d, t, lds_bytes, stream, info);
    }
    if (staged && d.point_kind != PK_NONE && (MODE == M_NEAREST || MODE == M_BILINEAR || MODE == M_BICUBIC))
        return launch_point<OUT>(d, t, lds_bytes, stream, info);
    if constexpr (MODE == M_BILINEAR || MODE == M_AREA_DOWN || MODE == M_NEAREST) {
        if (d.r32 >= 1 && d.r32 <= 6) return launch_bilinear_r32((OutKind)OUT, d, t, stream, info);
    }
    if constexpr (MODE == M_BILINEAR) {
        if (d.bil_rows) return launch_bilinear_rows((OutKind)OUT, d, t, lds_bytes, stream, info);
    }
    if constexpr (MODE == M_BILINEAR || MODE == M_AREA_UP) {
        if (staged) {
            return launch_bilinear(MODE == M_AREA_UP, (OutKind)OUT, d, t, grid.x, lds_bytes, stream, info);
        }
    } else if constexpr (MODE == M_BICUBIC) {
        if (d.r32 >= 7) return launch_bicubic_r32((OutKind)OUT, d, t, stream, info);
        if (d.bicubic_cols) return launch_bicubic_cols((OutKind)OUT, d.bicubic_cols == 2, d, t, lds_bytes, stream, info);
        if (staged && d.bicubic_int) return launch_bicubic_int((OutKind)OUT, d, t, lds_bytes, stream, info);
    } else if constexpr (MODE != M_NONE) {
        if constexpr (MODE == M_AREA_DOWN) {
            if (d.area_stream) return launch_area_stream((OutKind)OUT, d, t, lds_bytes, stream, info);
            if (vec && d.area_direct == 1 && d.area_box && !d.force_gather) // integer ratio: contiguous dword runs
                return launch_area_box((OutKind)OUT, d, t, stream, info);
            if (vec && d.area_direct == 1 && d.qx && d.qy && !d.force_gather) { // large dyadic ratios: no LDS at all
                if (d.rx <= 4) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_direct_kernel<1, OUT>", (vpp_area_direct_kernel<1, OUT>), grid, block, 0);
                else TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_direct_kernel<2, OUT>", (vpp_area_direct_kernel<2, OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
            if (vec && d.area_direct == 2 && d.area_cols && !d.force_gather) { // one output column per lane, taps straight from global memory
                const dim3 cblock(MAX_THREADS); // 256 threads whatever the tile height
                if (d.area_cols_rows == 32) {
                    if (d.nkx == 1) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<1, 32, OUT>", (vpp_area_cols_kernel<1, 32, OUT>), grid, cblock, 0);
                    else if (d.nkx == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<2, 32, OUT>", (vpp_area_cols_kernel<2, 32, OUT>), grid, cblock, 0);
                    else TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<3, 32, OUT>", (vpp_area_cols_kernel<3, 32, OUT>), grid, cblock, 0);
                } else {
                    if (d.nkx == 1) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<1, 8, OUT>", (vpp_area_cols_kernel<1, 8, OUT>), grid, cblock, 0);
                    else if (d.nkx == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<2, 8, OUT>", (vpp_area_cols_kernel<2, 8, OUT>), grid, cblock, 0);
                    else TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_cols_kernel<3, 8, OUT>", (vpp_area_cols_kernel<3, 8, OUT>), grid, cblock, 0);
                }
                return info ? hipSuccess : hipGetLastError();
            }
            if (vec && d.area_direct == 2 && !d.force_gather) { // large non-dyadic ratios: float sums straight from global memory
                if (d.nkx == 1) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_direct_float_kernel<1, OUT>", (vpp_area_direct_float_kernel<1, OUT>), grid, block, 0);
                else if (d.nkx == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_direct_float_kernel<2, OUT>", (vpp_area_direct_float_kernel<2, OUT>), grid, block, 0);
                else TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_direct_float_kernel<3, OUT>", (vpp_area_direct_float_kernel<3, OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
            if (staged && d.qx && d.qy) {
                if (d.rx <= 4) {
                    if (d.ry == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_dyadic_kernel<1, 2, OUT>", (vpp_area_dyadic_kernel<1, 2, OUT>), grid, block, lds_bytes);
                    else if (d.ry == 3) TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_dyadic_kernel<1, 3, OUT>", (vpp_area_dyadic_kernel<1, 3, OUT>), grid, block, lds_bytes);
                    else TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_dyadic_kernel<1, 0, OUT>", (vpp_area_dyadic_kernel<1, 0, OUT>), grid, block, lds_bytes);
                } else {
                    TSVPP_LAUNCH_OR_DESCRIBE("vpp_area_dyadic_kernel<2, 0, OUT>", (vpp_area_dyadic_kernel<2, 0, OUT>), grid, block, lds_bytes);
                }
                return info ? hipSuccess : hipGetLastError();
            }
        }
        if constexpr (MODE == M_AREA_DOWN) {
            if (staged && d.area2) { // float weights, at most 3 x 3 taps
                if (d.rx == 2 && d.ry == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_areaf_kernel<2, 2, OUT>", (vpp_areaf_kernel<2, 2, OUT>), grid, block, lds_bytes);
                else if (d.rx == 3 && d.ry == 2) TSVPP_LAUNCH_OR_DESCRIBE("vpp_areaf_kernel<3, 2, OUT>", (vpp_areaf_kernel<3, 2, OUT>), grid, block, lds_bytes);
                else if (d.rx == 2 && d.ry == 3) TSVPP_LAUNCH_OR_DESCRIBE("vpp_areaf_kernel<2, 3, OUT>", (vpp_areaf_kernel<2, 3, OUT>), grid, block, lds_bytes);
                else TSVPP_LAUNCH_OR_DESCRIBE("vpp_areaf_kernel<3, 3, OUT>", (vpp_areaf_kernel<3, 3, OUT>), grid, block, lds_bytes);
                return info ? hipSuccess : hipGetLastError();
            }
        }
    } else {
        if constexpr (OUT == O_Y800_U8 || OUT == O_NV12_U8) {
            if (staged && d.copy16) {
                TSVPP_LAUNCH_OR_DESCRIBE("vpp_copy16_kernel<OUT>", (vpp_copy16_kernel<OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
        }
        if (staged) { // colour-only fast path ("staged" = eligible)
            TSVPP_LAUNCH_OR_DESCRIBE("vpp_color_kernel<OUT>", (vpp_color_kernel<OUT>), grid, block, 0);
            return info ? hipSuccess : hipGetLastError();
        }
    }
    if (vec)
        TSVPP_LAUNCH_OR_DESCRIBE("vpp_fused_gather_kernel<MODE, OUT, true>", (vpp_fused_gather_kernel<MODE, OUT, true>), grid, block, 0);
    else // outputs that are not 16-byte aligned: element-wise stores
        TSVPP_LAUNCH_OR_DESCRIBE("vpp_fused_gather_kernel<MODE, OUT, false>", (vpp_fused_gather_kernel<MODE, OUT, false>), grid, block, 0);
    return info ? hipSuccess : hipGetLastError();
}

template <int MODE>
static hipError_t launch_m(OutKind out, bool vec, bool staged, LaunchDesc &d, const FrameTable &t, size_t lds, hipStream_t stream, LaunchInfo *info) {
    switch (out) {
    case O_U8_PLANAR: return launch_mo<MODE, O_U8_PLANAR>(vec, staged, d, t, lds, stream, info);
    case O_U8_MERGED: return launch_mo<MODE, O_U8_MERGED>(vec, staged, d, t, lds, stream, info);
    case O_F32_PLANAR: return launch_mo<MODE, O_F32_PLANAR>(vec, staged, d, t, lds, stream, info);
    case O_F32_MERGED: return launch_mo<MODE, O_F32_MERGED>(vec, staged, d, t, lds, stream, info);
    case O_NV12_U8: return launch_mo<MODE, O_NV12_U8>(vec, staged, d, t, lds, stream, info);
    case O_NV12_F32: return launch_mo<MODE, O_NV12_F32>(vec, staged, d, t, lds, stream, info);
    case O_Y800_U8: return launch_mo<MODE, O_Y800_U8>(vec, staged, d, t, lds, stream, info);
    case O_Y800_F32: return launch_mo<MODE, O_Y800_F32>(vec, staged, d, t, lds, stream, info);
    case O_HSV_F32: return launch_mo<MODE, O_HSV_F32>(vec, staged, d, t, lds, stream, info);
    default: return hipErrorInvalidValue;
    }
}

// The kernel of a (mode, output flavour) pair as chosen by launch_fused (vpp_select.hip) -- or, with `info`, only its name and grid.
hipError_t launch_mode(Mode mode, OutKind out, bool vec, bool staged, LaunchDesc &d, const FrameTable &t, size_t lds, hipStream_t stream, LaunchInfo *info) {
    switch (mode) {
    case M_NONE: return launch_m<M_NONE>(out, vec, staged, d, t, lds, stream, info);
    case M_NEAREST: return launch_m<M_NEAREST>(out, vec, staged, d, t, lds, stream, info);
    case M_BILINEAR: return launch_m<M_BILINEAR>(out, vec, staged, d, t, lds, stream, info);
    case M_BICUBIC: return launch_m<M_BICUBIC>(out, vec, staged, d, t, lds, stream, info);
    case M_AREA_DOWN: return launch_m<M_AREA_DOWN>(out, vec, staged, d, t, lds, stream, info);
    case M_AREA_UP: return launch_m<M_AREA_UP>(out, vec, staged, d, t, lds, stream, info);
    default: return hipErrorInvalidValue;
    }
}
// LDS bytes of the coordinate tables of the dyadic / small float AREA kernels for a tile of cols x rows outputs (their entry types live here)
size_t area_dyadic_table_bytes(size_t cols, size_t rows) { return cols * sizeof(AXEntry) + cols / 2 * sizeof(ACEntry) + (rows + rows / 2) * sizeof(AYEntry); }
size_t areaf_table_bytes(size_t cols, size_t rows) { return (cols + cols / 2) * sizeof(AFXEntry) + (rows + rows / 2) * sizeof(AFYEntry); }

} // namespace tsvpp
