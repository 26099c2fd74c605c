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
    const int j0 = (id.tx * d.tx + lx) * PXW + d.col0;
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
    const int j0 = (id.tx * d.tx + lx) * PXW, i0 = (id.ty * d.ty + ly) * PXH;
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
    const int j0 = (id.tx * d.tx + lx) * PXW, i0 = (id.ty * d.ty + ly) * PXH;
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
    const int j_first = id.tx * TW, i_first = id.ty * TH;
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
            const int yA = (int)(d.yr * (float)iA), yB = (int)(d.yr * (float)iB);
            f2 acc = { 0.0f, 0.0f }, div = { 0.0f, 0.0f };
            for (int a = 0; a < d.ry; a++) {
                const f2 wy = { wyA[a], wyB[a] };
                uint32_t da[NK + 1], db[NK + 1], sa, sb;
                cols_load<NK>(Y, ym, (uint32_t)(yA + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, (yA + a) < d.src_h - 1, da, sa);
                cols_load<NK>(Y, ym, (uint32_t)(yB + a) * (uint32_t)d.pitch_y + (uint32_t)x0, d.rx, (yB + a) < d.src_h - 1, db, sb);
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t va = __builtin_amdgcn_alignbyte(da[k + 1], da[k], sa), vb = __builtin_amdgcn_alignbyte(db[k + 1], db[k], sb);
                    const float wk[4] = { wx[k].x, wx[k].y, wx[k].z, wx[k].w };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const f2 wgt = (f2){ wk[b], wk[b] } * wy;
                        div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)((va >> (8 * b)) & 255), (float)((vb >> (8 * b)) & 255) }, wgt, acc);
                    }
                }
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
            for (int a = 0; a < d.ry; a++) {
                const float wy = wyrow[a];
                const uint32_t row = (uint32_t)(y0 + a) * (uint32_t)d.pitch_uv + (uint32_t)x0;
                const bool wide = (y0 + a) < (d.src_h >> 1) - 1;
                uint32_t dw[2 * NK + 1], sh;
                if constexpr (NK == 1) {
                    if (wide) load_span<2, true>(UV, uvm, row, 2 * d.rx, dw, sh);
                    else load_span<2, false>(UV, uvm, row, 2 * d.rx, dw, sh);
                } else {
                    sh = (uvm + row) & 3u;
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
                        const uint32_t qq = vv[b >> 1] >> (16 * (b & 1));
                        const float wgt = wv[b] * wy;
                        div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)(qq & 255), (float)((qq >> 8) & 255) }, (f2){ wgt, wgt }, acc);
                    }
                }
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
    const int j_first = id.tx * tw, i_first = id.ty * th;
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
        if (d.nt_stores) __builtin_nontemporal_store(a, (cp_x4a *)(out + off));
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
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
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

// ----------------------------------------------------------------------------------------------
// Host side of the launch: tile geometry, LDS budget, kernel selection.
static int span_bound(Mode m, int n_out, float ratio, int taps) {
    // max over tile positions of (last tap - first tap + 1) for n_out consecutive outputs, +1 spare
    int ext = 0;
    switch (m) {
    case M_BILINEAR: case M_AREA_UP: ext = 1; break;
    case M_BICUBIC: ext = 3; break;
    case M_AREA_DOWN: ext = taps - 1; break;
    default: ext = 0; break;
    }
    return (int)((double)ratio * (double)(n_out - 1)) + ext + 3;
}
static int slot_shift_for(int cpr) {
    int s = 0;
    while ((1 << s) < cpr) s++;
    return s;
}

// Launch, or -- when the caller only wants to know what WOULD run (tsvpp_describe, CPU tests of the selection
// logic) -- record the choice and launch nothing.
#define TSVPP_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS)                                   \
    do {                                                                               \
        if (info) {                                                                    \
            info->kernel = NAME;                                                       \
            info->grid = (int)(GRID).x;                                                \
            info->lds_bytes = (int)(LDS);                                              \
        } else {                                                                       \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, stream, d, t);                \
        }                                                                              \
    } while (0)

template <int OUT>
static hipError_t launch_point(const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    switch (d.point_kind) {
    case PK_NEAREST: TSVPP_LAUNCH("vpp_point_kernel<PK_NEAREST, OUT>", (vpp_point_kernel<PK_NEAREST, OUT>), grid, block, lds_bytes); break;
    case PK_BILINEAR0: TSVPP_LAUNCH("vpp_point_kernel<PK_BILINEAR0, OUT>", (vpp_point_kernel<PK_BILINEAR0, OUT>), grid, block, lds_bytes); break;
    case PK_BICUBIC0: TSVPP_LAUNCH("vpp_point_kernel<PK_BICUBIC0, OUT>", (vpp_point_kernel<PK_BICUBIC0, OUT>), grid, block, lds_bytes); break;
    default: return hipErrorInvalidValue;
    }
    return info ? hipSuccess : hipGetLastError();
}

template <int MODE, int OUT>
static hipError_t launch_mo(bool vec, bool staged, LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (staged && d.point_kind != PK_NONE && (MODE == M_NEAREST || MODE == M_BILINEAR || MODE == M_BICUBIC))
        return launch_point<OUT>(d, t, lds_bytes, stream, info);
    if constexpr (MODE == M_BILINEAR || MODE == M_AREA_DOWN || MODE == M_NEAREST) {
        if (d.r32 >= 1 && d.r32 <= 6) return launch_bilinear_r32((OutKind)OUT, d, t, stream, info);
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
                if (d.rx <= 4) TSVPP_LAUNCH("vpp_area_direct_kernel<1, OUT>", (vpp_area_direct_kernel<1, OUT>), grid, block, 0);
                else TSVPP_LAUNCH("vpp_area_direct_kernel<2, OUT>", (vpp_area_direct_kernel<2, OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
            if (vec && d.area_direct == 2 && d.area_cols && !d.force_gather) { // one output column per lane, taps straight from global memory
                const dim3 cblock(MAX_THREADS); // 256 threads whatever the tile height
                if (d.area_cols_rows == 32) {
                    if (d.nkx == 1) TSVPP_LAUNCH("vpp_area_cols_kernel<1, 32, OUT>", (vpp_area_cols_kernel<1, 32, OUT>), grid, cblock, 0);
                    else if (d.nkx == 2) TSVPP_LAUNCH("vpp_area_cols_kernel<2, 32, OUT>", (vpp_area_cols_kernel<2, 32, OUT>), grid, cblock, 0);
                    else TSVPP_LAUNCH("vpp_area_cols_kernel<3, 32, OUT>", (vpp_area_cols_kernel<3, 32, OUT>), grid, cblock, 0);
                } else {
                    if (d.nkx == 1) TSVPP_LAUNCH("vpp_area_cols_kernel<1, 8, OUT>", (vpp_area_cols_kernel<1, 8, OUT>), grid, cblock, 0);
                    else if (d.nkx == 2) TSVPP_LAUNCH("vpp_area_cols_kernel<2, 8, OUT>", (vpp_area_cols_kernel<2, 8, OUT>), grid, cblock, 0);
                    else TSVPP_LAUNCH("vpp_area_cols_kernel<3, 8, OUT>", (vpp_area_cols_kernel<3, 8, OUT>), grid, cblock, 0);
                }
                return info ? hipSuccess : hipGetLastError();
            }
            if (vec && d.area_direct == 2 && !d.force_gather) { // large non-dyadic ratios: float sums straight from global memory
                if (d.nkx == 1) TSVPP_LAUNCH("vpp_area_direct_float_kernel<1, OUT>", (vpp_area_direct_float_kernel<1, OUT>), grid, block, 0);
                else if (d.nkx == 2) TSVPP_LAUNCH("vpp_area_direct_float_kernel<2, OUT>", (vpp_area_direct_float_kernel<2, OUT>), grid, block, 0);
                else TSVPP_LAUNCH("vpp_area_direct_float_kernel<3, OUT>", (vpp_area_direct_float_kernel<3, OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
            if (staged && d.qx && d.qy) {
                if (d.rx <= 4) {
                    if (d.ry == 2) TSVPP_LAUNCH("vpp_area_dyadic_kernel<1, 2, OUT>", (vpp_area_dyadic_kernel<1, 2, OUT>), grid, block, lds_bytes);
                    else if (d.ry == 3) TSVPP_LAUNCH("vpp_area_dyadic_kernel<1, 3, OUT>", (vpp_area_dyadic_kernel<1, 3, OUT>), grid, block, lds_bytes);
                    else TSVPP_LAUNCH("vpp_area_dyadic_kernel<1, 0, OUT>", (vpp_area_dyadic_kernel<1, 0, OUT>), grid, block, lds_bytes);
                } else {
                    TSVPP_LAUNCH("vpp_area_dyadic_kernel<2, 0, OUT>", (vpp_area_dyadic_kernel<2, 0, OUT>), grid, block, lds_bytes);
                }
                return info ? hipSuccess : hipGetLastError();
            }
        }
        if constexpr (MODE == M_AREA_DOWN) {
            if (staged && d.area2) { // float weights, at most 3 x 3 taps
                if (d.rx == 2 && d.ry == 2) TSVPP_LAUNCH("vpp_areaf_kernel<2, 2, OUT>", (vpp_areaf_kernel<2, 2, OUT>), grid, block, lds_bytes);
                else if (d.rx == 3 && d.ry == 2) TSVPP_LAUNCH("vpp_areaf_kernel<3, 2, OUT>", (vpp_areaf_kernel<3, 2, OUT>), grid, block, lds_bytes);
                else if (d.rx == 2 && d.ry == 3) TSVPP_LAUNCH("vpp_areaf_kernel<2, 3, OUT>", (vpp_areaf_kernel<2, 3, OUT>), grid, block, lds_bytes);
                else TSVPP_LAUNCH("vpp_areaf_kernel<3, 3, OUT>", (vpp_areaf_kernel<3, 3, OUT>), grid, block, lds_bytes);
                return info ? hipSuccess : hipGetLastError();
            }
        }
    } else {
        if constexpr (OUT == O_Y800_U8 || OUT == O_NV12_U8) {
            if (staged && d.copy16) {
                TSVPP_LAUNCH("vpp_copy16_kernel<OUT>", (vpp_copy16_kernel<OUT>), grid, block, 0);
                return info ? hipSuccess : hipGetLastError();
            }
        }
        if (staged) { // colour-only fast path ("staged" = eligible)
            TSVPP_LAUNCH("vpp_color_kernel<OUT>", (vpp_color_kernel<OUT>), grid, block, 0);
            return info ? hipSuccess : hipGetLastError();
        }
    }
    if (vec)
        TSVPP_LAUNCH("vpp_fused_gather_kernel<MODE, OUT, true>", (vpp_fused_gather_kernel<MODE, OUT, true>), grid, block, 0);
    else // outputs that are not 16-byte aligned: element-wise stores
        TSVPP_LAUNCH("vpp_fused_gather_kernel<MODE, OUT, false>", (vpp_fused_gather_kernel<MODE, OUT, false>), grid, block, 0);
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

// ---- streaming kernels at the exact ratios 3 : 2 / 2 : 1 (vpp_bilinear_r32.hip, vpp_bicubic_r32.hip) ----------------------------------------------
// One row per (resize mode, ratio): the kernel instance (LaunchDesc::r32) and the same-box A/B that put the row here.  What a row needs beyond the
// exact ratio is in stream_select below.
struct StreamRow {
    Mode mode;
    int p2; // twice the ratio: 3 = 3 : 2, 4 = 2 : 1
    int r32;
    const char *evidence;
};
static const StreamRow kStreamRows[] = {
    { M_BILINEAR, 3, 1, "profiles/r02_r32_ab.txt: uint8 1080p -> 720p planar 0.563 -> 0.687, merged 0.509 -> 0.673, NV12 0.578 -> 0.684, Y800 0.490 -> 0.686" },
    { M_AREA_DOWN, 3, 2, "profiles/r02_r32_ab.txt: AREA planar 0.442 -> 0.679, merged 0.418 -> 0.660 (weight rows {1, 1/2}, {1/2, 1})" },
    { M_NEAREST, 3, 3, "profiles/r02_r32_ab.txt: NEAREST 0.624 -> 0.800" },
    { M_BILINEAR, 4, 4, "profiles/r02_r32_ab.txt: 1080p -> 540p 0.579 -> 0.630, 4K -> 1080p merged 0.681 vs 0.684; NOT planar >= 1.5 Mpixel (0.721 on the LDS kernel vs 0.688)" },
    { M_AREA_DOWN, 4, 5, "profiles/r02_r32_ab.txt: 4K -> 1080p AREA 0.558 -> 0.684, merged 0.430 -> 0.623 (one weight row {1, 1})" },
    { M_NEAREST, 4, 6, "profiles/r02_r32_ab.txt: NEAREST merged 0.83 -> 0.99" },
    { M_BICUBIC, 3, 7, "profiles/r04_bicubic_r32_ab.txt: 1080p -> 720p fp32 planar 0.670 -> 0.711, uint8 merged 0.362 -> 0.548, fp32 merged 0.573 -> 0.702" },
    { M_BICUBIC, 4, 8, "profiles/r04_bicubic_r32_ab.txt: 4K -> 1080p fp32 planar 0.652 -> 0.730, uint8 merged 0.368 -> 0.631; 1080p -> 540p 0.619 -> 0.691" },
};
// The streaming kernel instance of this request, or 0.  `mode` is the mode the launch runs as (an AREA request that took the 2x2-tap integer tile,
// LaunchDesc::tap22, arrives here as BILINEAR and is not eligible: fp32 RGB).
static int stream_select(Mode mode, OutKind out, bool vec, const LaunchDesc &d) {
    if (!vec || d.force_gather || !d.in_aligned4 || (d.dst_w & 7) != 0 || (d.dst_h & 3) != 0) return 0;
    const int p2 = (2L * d.src_w == 3L * d.dst_w && 2L * d.src_h == 3L * d.dst_h) ? 3 : ((d.src_w == 2 * d.dst_w && d.src_h == 2 * d.dst_h) ? 4 : 0);
    if (!p2) return 0;
    const bool f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    const bool u8_flavour = (out == O_U8_PLANAR || out == O_U8_MERGED || out == O_NV12_U8 || out == O_Y800_U8 || out == O_UYVY_U8 || out == O_YUV444_U8);
    int r32 = 0;
    for (const StreamRow &row : kStreamRows)
        if (row.mode == mode && row.p2 == p2) r32 = row.r32;
    if (!r32) return 0;
    if (r32 >= 7) // BICUBIC: every flavour of the colour back end (TSVPP_BICUBIC_INT=2 keeps the LDS integer kernel, TSVPP_BICUBIC_COLS=2 the column kernel)
        return (d.w_dyadic && d.bicubic_int_pref == 1 && d.bicubic_cols_pref != 2 && out < O_COUNT) ? r32 : 0;
    // 2x2-tap kinds: uint8 flavours; fp32 flavours (round 4, through the shared output side vpp_r32_store.h) tie or lose against the LDS kernels for RGB / BGR
    // (profiles/r04_r32_f32_ab.txt: AREA 1080p -> 720p 0.681 vs 0.683, 4K -> 1080p 0.65 vs 0.72, BILINEAR 0.69 vs 0.77) and win for HSV, whose three
    // divisions per pixel make the launch VALU-bound (AREA 0.56 -> 0.65, BILINEAR 0.59 -> 0.67): HSV takes them, the rest only under TSVPP_R32=2
    if (!d.r32_pref) return 0;
    if (!(u8_flavour || (f32_out && out < O_COUNT && (out == O_HSV_F32 || d.r32_pref == 2)))) return 0;
    if (mode == M_AREA_DOWN) { // the weight pattern the kernel instance has compiled in
        const bool pat = d.qx && d.qy && d.rx == 2 && d.ry == 2 && d.area_rcp != 0.0f && (p2 == 3 ? (d.nx == 2 && d.ny == 2) : (d.nx == 1 && d.ny == 1));
        if (!pat) return 0;
    }
    if (r32 == 4 && out == O_U8_PLANAR && (long)d.dst_w * d.dst_h >= 1500000L && d.r32_pref != 2) return 0; // (see its row)
    return r32;
}
// Workgroup shape of a streaming launch (thread tile = 8 columns x 4 rows).  Measured: 64 x 4 threads (512 x 16 pixels) +0.4..3 % over 32 x 8 and 16 x 16 for
// the uint8 flavours of the 2x2-tap kinds (profiles/r02_r32_ab.txt); the exceptions below each carry their file.
static void stream_shape(int r32, OutKind out, const LaunchDesc &d, int &tx, int &ty) {
    const bool f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    const int n = d.dst_w / 8; // threads per output row
    auto waste = [&](int w) { return (double)((n + w - 1) / w * w) / (double)n - 1.0; };
    tx = 64;
    ty = 4;
    if (r32 < 7) {
        if (f32_out) ty = out == O_HSV_F32 ? 4 : 2; // fp32 outputs want short tiles (as the BICUBIC kernel below)
        // YUV444, the one VALU-bound flavour of the 2x2-tap kinds (125-137 VGPRs): lanes past the right edge cost what they idle -- 1280 columns = 2.5
        // rows of 64 threads: 0.465 -> 0.567 on 32 x 4 (profiles/r04_r32_shape_1280.txt); the other flavours do not care
        if (out == O_YUV444_U8 && waste(64) > 0.08 && waste(32) <= 0.08) {
            tx = 32;
            ty = 4;
        }
        return;
    }
    // BICUBIC.  uint8 outputs are VALU-bound (77 % busy, profiles/r04_bicubic_r32_pmc.txt), so idle lanes cost what they idle: 1280 columns = 160
    // threads -- 32-wide workgroups +9 % there; 1920 and 960 columns (240 / 120 threads) lose 6.7 % of a 64-wide row and still prefer it (longer store
    // runs).  fp32 outputs are bound by the issue of memory instructions: always 64 wide (neighbour dwords by wave shuffle instead of two loads per
    // row) and SHORT tiles, as the 2x2-tap kernel: two thread rows (profiles/r04_bicubic_r32_shapes.txt, r04_bicubic_r32_ab.txt); HSV -- three
    // divisions per pixel, VALU-bound again -- four (0.60 vs 0.53).
    tx = (f32_out || waste(64) <= 0.08) ? 64 : (waste(32) <= 0.08 ? 32 : (waste(16) < waste(32) ? 16 : 32));
    ty = f32_out ? (out == O_HSV_F32 ? 4 : 2) : 256 / tx;
    if (ty > 8) ty = 8;
}

// ---- launch_fused, step by step -------------------------------------------------------------------------------------------------------------------
// What launch_fused decides besides the LaunchDesc fields.  Each sel_* function below is one decision with the measurements that set its thresholds;
// launch_fused calls them in order (tests/golden/describe_snapshot.json pins the outcome for 417 requests).
struct FusedSel {
    int shapes[5][2] = { { 32, 8 }, { 32, 4 }, { 16, 4 }, { 0, 0 }, { 0, 0 } }; // candidate workgroup shapes, largest first
    size_t lds_budget = 40 * 1024, lds_bytes = 0, as_lds = 0;
    bool staged = false, f32_out = false, two_tap = false;
};
static long fused_workgroups(const LaunchDesc &d, const int *sh, int rpt) {
    return (long)((d.dst_w + sh[0] * PXW - 1) / (sh[0] * PXW)) * ((d.dst_h + sh[1] * PXH * rpt - 1) / (sh[1] * PXH * rpt)) * d.n_frames;
}

// 1. AREA on the 2x2-tap integer tile (may turn `mode` into M_BILINEAR)
static void sel_tap22(Mode &mode, OutKind out, bool vec, LaunchDesc &d) {
    // AREA down-scale at exactly 3 : 2 / 2 : 1 with fp32 RGB / BGR / NV12 outputs: it taps the SAME two samples per axis as BILINEAR at that ratio
    // ((int)(r j) == floor((j + 0.5) r - 0.5) for r = 1.5 and 2), so it runs on the 2x2-tap kernel's integer window tile with its own integer weights and a
    // division instead of the shift (LaunchDesc::tap22, vpp_bilinear.hip).  Same-box A/B (profiles/r04_tap22_ab.txt): 1080p -> 720p planar 0.692 -> 0.766
    // (vpp_area_dyadic_kernel before), merged 0.699 -> 0.715, NV12 0.692 -> 0.722, 4K -> 1080p 0.717 -> 0.736 (vpp_area_box_kernel<2> before), merged
    // 0.731 -> 0.782, 1080p -> 540p 0.718 -> 0.750, 4K -> 1440p 0.700 -> 0.762.  Not taken: Y800 (0.623 -> 0.593), NEAREST (measured with weights (1, 0):
    // 0.689 -> 0.678, merged 0.702 -> 0.630 -- the point sampler reads fewer rows), uint8 outputs and HSV (the streaming kernel below).
    // TSVPP_BILINEAR_INT=0 / 2 switch it off.
    d.tap22 = 0;
    {
        const bool f32 = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32);
        const bool r32x = 2L * d.src_w == 3L * d.dst_w && 2L * d.src_h == 3L * d.dst_h, r21 = d.src_w == 2 * d.dst_w && d.src_h == 2 * d.dst_h;
        // (4 k + 2 columns: the tail launch samples by MODE)
        if (f32 && vec && !d.force_gather && d.bil_int_pref == 1 && d.r32_pref != 2 && (r32x || r21) && (d.dst_w & 3) == 0 && mode == M_AREA_DOWN && d.qx && d.qy &&
            d.rx == 2 && d.ry == 2 && d.area_rcp != 0.0f && ((r32x && d.nx == 2 && d.ny == 2) || (r21 && d.nx == 1 && d.ny == 1)))
            d.tap22 = r32x ? 1 : 2;
        if (d.tap22) {
            mode = M_BILINEAR;
            d.w_dyadic = 1;
            d.point_kind = PK_NONE;
            d.geo_pref = 0; // (the host-built geometry tables carry BILINEAR's weights)
        }
    }
}

// 2. thread-tile forms of the 2x2-tap kernel
static void sel_two_tap_forms(Mode mode, LaunchDesc &d) {
    d.bil_int = ((mode == M_BILINEAR || mode == M_AREA_UP) && d.w_dyadic && d.bil_int_pref) ? 1 : 0;
    // window form (one aligned 12-byte read per source row instead of byte reads): the four columns of a thread must span <= 8
    // bytes, i.e. horizontal ratio <= 2 (vpp_bilinear.hip); TSVPP_BILINEAR_INT=2 keeps the byte form
    if (d.bil_int && d.bil_int_pref != 2 && d.xr <= 2.0f) d.bil_int = 2;
    // float weights, same windows: measured +4..6 % at ratios 1.2 / 1.4 (1080p -> 1600x900, 1366x768), -2..5 % at 1.5 x 1.27 and 1.92,
    // even below 1 (profiles/r02_bilinear_winf_ab.txt) -- used between 1 and 1.45
    d.bil_win = (!d.bil_int && (mode == M_BILINEAR || mode == M_AREA_UP) && d.xr <= 2.0f &&
                 (d.bil_win_pref == 2 || (d.bil_win_pref == 1 && d.xr > 1.0f && d.xr <= 1.45f))) ? 1 : 0;
}

// 3. store policy
static void sel_store_policy(Mode mode, OutKind out, bool vec, LaunchDesc &d) {
    if (d.nt_stores < 0) { // per-kernel default
        // fp32 outputs: every store instruction of a wave covers whole 128-byte lines (planar: 16 contiguous
        // bytes per lane; merged: after the in-wave exchange of MergedRun) and nothing re-reads them.  The
        // scalar fallback of merged outputs (!vec) interleaves partial lines: plain stores, L2 combines them.
        const bool f32_lines = (out == O_F32_PLANAR || out == O_NV12_F32 || out == O_Y800_F32) || (vec && (out == O_F32_MERGED || out == O_HSV_F32));
        const bool f32_partial = !vec && (out == O_F32_MERGED || out == O_HSV_F32);
        d.nt_stores = f32_partial ? 0 : ((mode == M_NONE && f32_lines) ? 2 : 1);
    }
}

// 4. candidate workgroup shapes
static void sel_shapes(Mode mode, OutKind out, LaunchDesc &d, FusedSel &S) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    (void)staged; (void)lds_bytes; (void)shapes; (void)kLdsBudget; (void)f32_out; (void)two_tap;
    // Candidate workgroup shapes, largest first; the staged kernels take the first whose source
    // footprint fits the LDS budget (several workgroups per CU must stay resident to overlap one
    // group's loads with another's arithmetic).
    // The 2x2-tap kernel with fp32 outputs runs AT the HBM floor of its tile pattern, and that floor depends on the tile
    // shape (tools/membench2.hip, the headline's 22 % read / 78 % write mix with no arithmetic, 64 frames, rotating buffers):
    // 128 x 32 tiles 165 us, 128 x 16 159 us, 256 x 8 153 us (1 KiB row segments per plane, half the in-flight footprint),
    // whole rows 170 us.  The kernel follows: 1080p -> 720p 164.7 us on 128 x 32 tiles (0.688 of 8 TB/s), 151.6 us on
    // 256 x 8 (0.748) -- shape sweep in profiles/r02_shape_sweep.txt.  So: 256-wide, 8-row tiles wherever the output width
    // fills them (a half-empty last tile column costs more than the pattern gains: 1920-wide outputs stay on 128).
    // (the point samplers write the same pattern: NEAREST 1080p -> 720p 0.770 -> 0.795 on 64 x 4, profiles/r04_shape_sweep_misc.txt)
    if ((two_tap || (d.point_kind != PK_NONE && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC))) && f32_out && d.dst_w % 256 == 0) {
        shapes[3][0] = shapes[2][0]; shapes[3][1] = shapes[2][1];
        shapes[2][0] = shapes[1][0]; shapes[2][1] = shapes[1][1];
        shapes[1][0] = shapes[0][0]; shapes[1][1] = shapes[0][1];
        shapes[0][0] = 64; shapes[0][1] = 4;
    }
    // uint8 outputs on the integer window tile read host-built geometry tables (vpp_bilinear.hip): with workgroups 64 thread tiles
    // wide a wave's lanes share their output rows and the row records are scalar loads -- measured (profiles/r02_geo_ab.txt)
    // 1080p -> 720p planar 0.545 -> 0.570, merged 0.481 -> 0.506, 4K -> 1080p 0.706 -> 0.717 against 32 x 8
    if (two_tap && !f32_out && d.bil_int == 2 && d.geo_pref && d.dma && (d.pitch_y & 15) == 0 && (d.pitch_uv & 15) == 0 && d.dst_w >= 256) {
        shapes[3][0] = shapes[2][0]; shapes[3][1] = shapes[2][1];
        shapes[2][0] = shapes[1][0]; shapes[2][1] = shapes[1][1];
        shapes[1][0] = shapes[0][0]; shapes[1][1] = shapes[0][1];
        shapes[0][0] = 64; shapes[0][1] = 4;
    }
    if (d.shape_tx > 0 && d.shape_ty > 0 && (d.shape_tx & (d.shape_tx - 1)) == 0 && d.shape_tx * d.shape_ty <= MAX_THREADS &&
        d.shape_tx * d.shape_ty >= 64) {
        shapes[0][0] = d.shape_tx;
        shapes[0][1] = d.shape_ty;
    }
}

// 5. AREA down-scale: which of the un-staged samplers (direct / box / streaming / column-per-lane), if any
static void sel_area(Mode mode, bool vec, LaunchDesc &d, FusedSel &S) {
    size_t &as_lds = S.as_lds;
    if (mode == M_AREA_DOWN && d.qx && d.qy && vec && !d.force_gather && d.area_direct_min > 0.0f && d.xr >= d.area_direct_min &&
        d.yr >= d.area_direct_min)
        d.area_direct = 1;
    else if (mode == M_AREA_DOWN && !(d.qx && d.qy) && vec && !d.force_gather && d.area_direct_fmin > 0.0f && d.xr >= d.area_direct_fmin &&
             d.yr >= d.area_direct_fmin && d.nkx >= 1 && d.nkx <= 8 && d.patx4 && d.paty4)
        d.area_direct = 2; // float weights
    else
        d.area_direct = 0;
    // integer horizontal ratio (one all-ones weight row), 4-byte aligned planes and pitches: the box kernel's contiguous runs
    d.area_box = (d.area_direct == 1 && d.area_box_pref && d.box_rx >= 4 && d.box_rx <= 8 && d.box_rx == d.rx && d.in_aligned4 && d.ry <= 8) ? 1 : 0;
    // integer horizontal ratios 2 and 3 (1080p -> 960x540, 1080p -> 640x360, 4K -> 720p): below area_direct_min these went to the LDS
    // kernel (measured against the round-1 direct kernel); the box kernel wins there too (profiles/r02_area_box23_ab.txt: 1080p -> 640x360
    // fp32 0.602 -> 0.715, uint8 0.492 -> 0.666, 4K -> 720p uint8 merged 0.536 -> 0.711, 1080p -> 960x540 fp32 0.673 -> 0.707).
    // TSVPP_AREA_BOX=4 keeps it to ratios >= 4.
    if (!d.area_box && d.area_box_pref && d.area_box_pref != 4 && mode == M_AREA_DOWN && d.qx && d.qy && vec && !d.force_gather && (d.box_rx == 2 || d.box_rx == 3) &&
        d.box_rx == d.rx && d.in_aligned4 && d.ry <= 8 && d.yr >= 2.0f) {
        d.area_direct = 1;
        d.area_box = 1;
    }
    // Float-weight AREA: one wave per 128-column tile, the source rows streamed through a wave-private ring (vpp_area_stream.hip).  Needs the
    // host-built divisor table, pitches that are multiples of 16 (LDS-DMA chunks), a row segment of at most 128 chunks (ratio <= ~15).
    d.area_stream = 0;
    // TSVPP_AREA_STREAM: 1 = from `as_min_taps` taps per value on (measured cross-over, profiles/r03_area_stream_ab*.txt), 2 = wherever it applies
    if (mode == M_AREA_DOWN && !(d.qx && d.qy) && vec && !d.force_gather && (d.area_stream_pref == 2 ||
         (d.area_stream_pref == 1 && (d.rx * d.ry >= d.as_min_taps || d.nkx > 3 || (d.area_direct != 2 && !(d.rx <= 3 && d.ry <= 3 && d.area2_pref))))) && d.area_div && d.patx4 && d.paty4 && d.nkx >= 1 && d.nkx <= 8 &&
        (d.pitch_y & 15) == 0 && (d.pitch_uv & 15) == 0) {
        const int nk = d.nkx <= 4 ? d.nkx : (d.nkx <= 6 ? 6 : 8);
        auto rowb_of = [&](int cols) {
            const int seg_y = (int)((double)d.xr * (cols - 1)) + 1 + 4 * nk, seg_uv = 2 * ((int)((double)d.xr * (cols / 2 - 1)) + 1) + 8 * nk;
            return ((seg_y > seg_uv ? seg_y : seg_uv) + 15 + 15) & ~15;
        };
        const int two = rowb_of(128) <= 2048 ? 1 : 0; // a row segment = at most two DMA instructions (128 chunks)
        const int rowb = rowb_of(two ? 128 : 64);
        const int nkmin = nk == 6 ? 5 : (nk == 8 ? 7 : nk);
        // (the kernel skips the multiply of column taps 1 .. 4 * nkmin - 5: they weigh 1.0f for every ratio -- checked against the table itself)
        if (rowb <= 2048 && d.as_ones_x >= 4 * nkmin - 4) {
            // tile height 4 (measured: 8 rows never win -- 4K -> 608x342 0.618 against 0.572, 1080p -> 160^2 0.59 against 0.47; TSVPP_AREA_STREAM_ROWS=8)
            int r = 4;
            if (d.as_rows == 4 || d.as_rows == 8) r = d.as_rows;
            if (!two) r = 8;
            d.area_stream = 1;
            d.as_nk = nk;
            d.as_two = two;
            d.bc_ring_bytes = AS_RING_ROWS * rowb + 16;
            d.bc_wave_bytes = d.bc_ring_bytes + 128 * (r + r / 2);
            as_lds = 4 * (size_t)d.bc_wave_bytes;
            d.area_direct = 0;
            d.tx = two ? 32 : 16; // colour phase: a wave = 32 x 2 thread tiles per 4-row slab, or 16 x 4 per 8-row slab (MergedRun: runs of 32 / 16 lanes)
            d.ty = two ? 2 : 4;
            d.rpt = two ? r / 4 : 1;
        }
    }
    // Below the streaming kernel's cross-over (fewer than 40 taps per value; <= 12 horizontal taps): measured in round 2
    // (profiles/r02_area_cols_ab.txt; TSVPP_AREA_COLS=0/1/2, TSVPP_AREA_COLS_ROWS=8/32) the column-per-lane kernel wins from 5 horizontal
    // taps on -- 1080p -> 300^2 +18 %, -> 416^2 +21 % -- and is even at 2-4 taps
    if (d.area_direct == 2 && d.nkx > 3) d.area_direct = 0; // (13+ taps without the streaming kernel: generic path)
    d.area_cols = (!d.area_stream && d.area_direct == 2 && (d.area_cols_pref == 2 || (d.area_cols_pref == 1 && d.nkx >= 2))) ? 1 : 0;
    if (d.area_cols_rows != 8 && d.area_cols_rows != 32) d.area_cols_rows = d.nkx >= 3 ? 8 : 32;
    if (d.area_cols) { // fixed workgroup of 256 threads; tile = 16 x (rows / 2) thread tiles = 64 columns x 32 or 8 rows
        d.tx = 16;
        d.ty = d.area_cols_rows / 2;
    }
}

// 6. point samplers (NEAREST; BILINEAR / BICUBIC whose weights are all zero): one LDS row per output row
static void sel_point(Mode mode, bool vec, LaunchDesc &d, FusedSel &S) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    (void)staged; (void)lds_bytes; (void)shapes; (void)kLdsBudget; (void)f32_out; (void)two_tap;
    const bool point = d.point_kind != PK_NONE && (mode == M_NEAREST || mode == M_BILINEAR || mode == M_BICUBIC);
    if (!point) d.point_kind = PK_NONE;
    if (point && vec && !d.force_gather) {
        for (auto &sh : shapes) {
            if (sh[0] == 0) break;
            const int tw = sh[0] * PXW, th = sh[1] * PXH, nthreads = sh[0] * sh[1];
            const int span_y = (int)((double)d.xr * (tw - 1)) + 3, span_uv = 2 * ((int)((double)d.xr * (tw / 2 - 1)) + 3);
            const int cpr_y = (span_y + 15 + 15) / 16, cpr_uv = (span_uv + 15 + 15) / 16;
            if (cpr_y > nthreads || cpr_uv > nthreads) continue;
            const size_t need = (size_t)16 * ((size_t)th * cpr_y + (size_t)(th / 2) * cpr_uv) + sizeof(int) * (size_t)(tw + tw / 2 + th + th / 2);
            if (need <= kLdsBudget) {
                staged = true;
                lds_bytes = need;
                d.tx = sh[0];
                d.ty = sh[1];
                d.lds_span_y = span_y;
                d.lds_cpr_y = cpr_y;
                d.lds_slot_y = slot_shift_for(cpr_y);
                d.lds_span_uv = span_uv;
                d.lds_cpr_uv = cpr_uv;
                d.lds_slot_uv = slot_shift_for(cpr_uv);
                break;
            }
        }
        if (!staged) d.point_kind = PK_NONE; // footprint too large: generic paths below
    } else {
        d.point_kind = PK_NONE;
    }
}

// 7. the LDS-staged kernels (2x2-tap, integer BICUBIC, dyadic / small float AREA): first workgroup shape, rows per thread and staging layout that fit
static void sel_staged(Mode mode, bool vec, bool bicubic_staged, bool sparse_gather, LaunchDesc &d, FusedSel &S) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    (void)staged; (void)lds_bytes; (void)shapes; (void)kLdsBudget; (void)f32_out; (void)two_tap;
    auto workgroups = [&](const int *sh, int rpt) { return fused_workgroups(d, sh, rpt); };
    if (!staged && mode != M_NONE && vec && !d.force_gather && !d.area_direct && !sparse_gather && !d.area_stream) {
        const int want_dma = d.dma;
        for (auto &sh : shapes) {
            if (sh[0] == 0 || staged) break;
            const bool bint = bicubic_staged; // dyadic weights: integer kernel
            const bool area2 = mode == M_AREA_DOWN && !(d.qx && d.qy) && d.rx >= 2 && d.rx <= 3 && d.ry >= 2 && d.ry <= 3 && d.area2_pref;
            const bool dyadic = mode == M_AREA_DOWN && d.qx && d.qy;
            if ((mode == M_AREA_DOWN && !dyadic && !area2) || mode == M_NEAREST) break; // (no staged kernel: streaming kernel above, or gathers)
            // Row pairs per thread (TSVPP_RPT; 0 = per kernel).  Taller thread tiles amortise the tile decode, staging set-up and
            // table build over more pixels -- that pays where the kernel is VALU-bound (uint8 outputs, separable BICUBIC, the
            // AREA kernels: two row pairs) -- but the fp32 2x2-tap kernel is bound by the HBM write pattern, which prefers
            // SHORT tiles (round 2 sweep: one row pair wins by 2..9 % on 1080p -> 720p, 4K -> 1080p and 720p -> 1080p).
            const int rpt_auto = (two_tap && f32_out) ? 1 : 2;
            const int rpt_want = d.rpt_pref >= 1 && d.rpt_pref <= 8 ? d.rpt_pref : rpt_auto;
            int rpt_max = (two_tap || bint || area2 || dyadic) ? rpt_want : 1;
            // taller thread tiles only while the launch still has at least two full rounds of workgroups
            // (8 per CU): small outputs (C3: 256x256) need the parallelism more than the amortisation
            // (the 2x2-tap kernel wants six rounds)
            const long rounds = two_tap ? 48L : 16L;
            while (rpt_max > 1 && workgroups(sh, rpt_max) < rounds * d.num_cus) rpt_max--;
            // One (rows per thread, staging layout) candidate of this shape: its LDS need and descriptor fields.
            struct Cand { bool ok; size_t need; int rpt, dma, span_y, rows_y, cpr_y, span_uv, rows_uv, cpr_uv, hcs_y, hcs_uv; };
            auto candidate = [&](int rpt, int layout) {
                Cand c = {};
                c.rpt = rpt;
                c.span_y = span_bound(mode, sh[0] * PXW, d.xr, d.rx);
                const int rows_y = span_bound(mode, sh[1] * PXH * rpt, d.yr, d.ry);
                c.span_uv = 2 * span_bound(mode, sh[0] * PXW / 2, d.xr, d.rx);
                const int rows_uv = span_bound(mode, sh[1] * PXH * rpt / 2, d.yr, d.ry);
                const int nthreads = sh[0] * sh[1];
                c.cpr_y = (c.span_y + 15 + 15) / 16;
                c.cpr_uv = (c.span_uv + 15 + 15) / 16;
                if (c.cpr_y > nthreads || c.cpr_uv > nthreads) return c;
                c.rows_y = rows_y;
                c.rows_uv = rows_uv;
                c.dma = (layout == 1 && nthreads >= 64) ? 1 : 0;
                if (layout == 1 && !c.dma) return c;
                if (c.dma) { // a wave instruction fills 64 consecutive chunk slots: the plane is allocated up to a multiple of 64 slots
                    c.rows_y = ((rows_y * c.cpr_y + 63) / 64 * 64 + c.cpr_y - 1) / c.cpr_y;
                    c.rows_uv = ((rows_uv * c.cpr_uv + 63) / 64 * 64 + c.cpr_uv - 1) / c.cpr_uv;
                }
                const size_t cols = (size_t)sh[0] * PXW, rows = (size_t)sh[1] * PXH * rpt;
                size_t need = (size_t)16 * ((size_t)c.rows_y * c.cpr_y + (size_t)c.rows_uv * c.cpr_uv);
                if (dyadic) // tables + row bases + slack for the dword over-read
                    need += cols * sizeof(AXEntry) + cols / 2 * sizeof(ACEntry) + (rows + rows / 2) * sizeof(AYEntry) +
                            sizeof(int) * (size_t)(c.rows_y + c.rows_uv) + 32;
                if (area2) // column / row tables and row bases of the float AREA kernel
                    need += (cols + cols / 2) * sizeof(AFXEntry) + (rows + rows / 2) * sizeof(AFYEntry) + sizeof(int) * (size_t)(c.rows_y + c.rows_uv);
                if (mode == M_BILINEAR || mode == M_AREA_UP) // coordinate tables
                    need += (cols + cols / 2) * sizeof(XEntry) + (rows + rows / 2) * sizeof(YEntry);
                if (bint) { // column-major H planes (column stride: an odd number of dwords >= rows + 8 bytes), tables, row bases
                    c.hcs_y = 4 * (((rows_y + 3) / 4 + 2) | 1);
                    c.hcs_uv = 4 * (((rows_uv + 3) / 4 + 2) | 1);
                    need += bicubic_int_table_bytes((int)cols, (int)rows, c.rows_y, c.rows_uv, c.hcs_y, c.hcs_uv);
                }
                c.need = need;
                c.ok = need <= kLdsBudget;
                return c;
            };
            Cand best = {};
            if (dyadic) {
                // the integer AREA kernel is sensitive to how many workgroups a CU holds (160 KiB of LDS): 1080p -> 960x540
                // runs 560 k frames/s on the compact two-row layout (19.7 KiB, 8 workgroups), 523 k on the compact
                // four-row one (34 KiB) and 494 k on the two-row LDS-DMA one (37.5 KiB).  Most resident workgroups
                // (up to five) wins; ties go to the taller tile, then to LDS-DMA.
                long best_key = -1;
                for (int rpt = rpt_max; rpt >= 1; rpt--)
                    for (int layout = want_dma ? 1 : 0; layout >= 0; layout--) {
                        const Cand c = candidate(rpt, layout);
                        if (!c.ok) continue;
                        long occ = (long)(160 * 1024 / c.need);
                        if (occ > 5) occ = 5; // beyond five resident workgroups the layout matters more (1080p -> 1536x864:
                                              // four-row LDS-DMA tiles at 6 per CU, 283 k, vs compact ones at 8, 268 k)
                        const long key = occ * 100 + rpt * 10 + c.dma;
                        if (key > best_key) {
                            best_key = key;
                            best = c;
                        }
                    }
            } else {
                // first fit: the LDS-DMA layout, then the compact register-staged one; the separable BICUBIC kernel tries
                // a shorter tile of the SAME workgroup shape before a smaller workgroup (1080p -> 640x640: 375 k vs 288 k)
                for (int rpt = rpt_max; rpt >= 1 && !best.ok; rpt = bint ? rpt - 1 : 0)
                    for (int layout = want_dma ? 1 : 0; layout >= 0 && !best.ok; layout--) best = candidate(rpt, layout);
            }
            if (best.ok) {
                staged = true;
                lds_bytes = best.need;
                d.tx = sh[0];
                d.ty = sh[1];
                d.rpt = best.rpt;
                d.lds_span_y = best.span_y;
                d.lds_rows_y = best.rows_y;
                d.lds_cpr_y = best.cpr_y;
                d.lds_slot_y = slot_shift_for(best.cpr_y);
                d.lds_span_uv = best.span_uv;
                d.lds_rows_uv = best.rows_uv;
                d.lds_cpr_uv = best.cpr_uv;
                d.lds_slot_uv = slot_shift_for(best.cpr_uv);
                d.lds_magic_y = 0xFFFFFFFFu / (uint32_t)best.cpr_y + 1u;
                d.lds_magic_uv = 0xFFFFFFFFu / (uint32_t)best.cpr_uv + 1u;
                d.dma = best.dma;
                d.bicubic_int = bint ? 1 : 0;
                d.hcs_y = best.hcs_y;
                d.hcs_uv = best.hcs_uv;
                d.area2 = area2 ? 1 : 0;
            }
        }
    }
}

// 8. the wave-per-tile BICUBIC kernel
static void sel_bicubic_cols(Mode mode, bool vec, int bc_r32, LaunchDesc &d, FusedSel &S, hipStream_t stream, LaunchInfo *info) {
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    int (&shapes)[5][2] = S.shapes;
    const size_t kLdsBudget = S.lds_budget;
    const bool f32_out = S.f32_out, two_tap = S.two_tap;
    (void)staged; (void)lds_bytes; (void)shapes; (void)kLdsBudget; (void)f32_out; (void)two_tap;
    // BICUBIC that the integer kernel above did not take (non-dyadic weights -- or TSVPP_BICUBIC_COLS=2: every request): one wave per
    // 64-column tile, one lane per output column, H sums in a wave-private column-major LDS plane (vpp_bicubic_cols.hip).  A taller
    // tile re-evaluates fewer H rows at its seams (3 / (R yr) of them), a shorter one keeps more waves in flight.
    if (mode == M_BICUBIC && !staged && vec && !d.force_gather && d.bicubic_cols_pref && !bc_r32) {
        const bool sparse = d.yr >= 4.0f;
        const bool exact = d.w_dyadic != 0; // every weight a multiple of 1/16: the quantised coefficients are exact, no tie test
        // LDS-DMA ring: a row segment is (64 columns at ratio xr + window + a misalignment of up to 15 bytes) rounded up to 16-byte chunks,
        // at most 16 of them (horizontal ratios up to 3.68); bc_dma = chunks (lanes) per row; the kernel takes that path when the pitch is
        // a multiple of 16
        const int seg_bytes = (int)((double)d.xr * 63.0) + 2 + 7 + 15 + 1;
        const bool dma = d.bc_dma_pref && seg_bytes <= 256;
        const int dma_lanes = (seg_bytes + 15) / 16 < 2 ? 2 : (seg_bytes + 15) / 16;
        const int ring_bytes = dma ? 3 * 1024 + 16 : 0;
        auto col_stride = [&](int nout) { // bytes of one H-plane column: its dwords + one (phase 2 reads dword pairs), an odd number of them
            const int rows = sparse ? 4 * nout : (int)((double)d.yr * (nout - 1)) + 6;
            return 4 * ((((rows + 3) >> 2) + 1) | 1);
        };
        auto wave_bytes_of = [&](int r) { return ring_bytes + 64 * (col_stride(r) + r + r / 2); };
        // Tile height, measured (profiles/r03_bicubic_cols_ab*.txt): 32 rows for up-scales (few source rows per tile: the seams cost
        // most there; 720p -> 1080p 0.519 against 0.502 at 16), 16 rows while the launch still has 8 waves per SIMD (1080p -> 640^2 0.620
        // against 0.609 at 8, 4K -> 1080p 0.651 against 0.624), else 8 (1080p -> 224^2 0.748 against 0.642, -> 300^2 0.404 against 0.367)
        int best_r = 0;
        {
            auto waves_of = [&](int r) { return (long)((d.dst_w + 63) / 64) * ((d.dst_h + r - 1) / r) * d.n_frames; };
            int r = d.yr <= 1.0f ? 32 : 16;
            while (r > 8 && (waves_of(r) < 32L * d.num_cus || 4 * wave_bytes_of(r) > 40 * 1024)) r -= 8;
            if (d.bc_rows >= 8 && d.bc_rows <= 32 && (d.bc_rows & 7) == 0) r = d.bc_rows;
            if (4 * wave_bytes_of(r) <= 64 * 1024) best_r = r;
        }
        // the request's column / row tables (host-built, cached in the context): a real launch -- and the dry run of
        // tsvpp_prepare_batch -- looks them up or builds them; while the stream is capturing and they do not exist yet the
        // request takes the generic path below
        if (best_r && (!info || d.geo_build)) {
            d.bc_tab = bicubic_cols_tables(d, stream, true);
            if (!d.bc_tab) best_r = 0;
        }
        if (best_r) {
            d.bicubic_cols = exact ? 2 : 1;
            d.bc_sparse = sparse ? 1 : 0;
            d.bc_dma = dma ? (d.bc_dma_pref > 1 ? 16 : dma_lanes) : 0; // TSVPP_BICUBIC_DMA=2: 256-byte segments whatever the ratio (A/B)
            d.bc_ring_bytes = ring_bytes;
            d.bc_npy = bicubic_cols_rows_padded(d.dst_h);
            d.bc_npc = bicubic_cols_rows_padded(d.dst_h >> 1);
            d.hcs_y = col_stride(best_r);
            d.hcs_uv = d.hcs_y;
            d.bc_wave_bytes = wave_bytes_of(best_r);
            lds_bytes = 4 * (size_t)d.bc_wave_bytes;
            d.tx = 16; // colour phase: a wave = 16 x 4 thread tiles per 8-row slab (MergedRun: runs of 16 lanes)
            d.ty = 4;
            d.rpt = best_r / 8;
            d.dma = 0;
        }
    }
}

hipError_t launch_fused(Mode mode, OutKind out, bool vec, const LaunchDesc &din, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    LaunchDesc d = din;
    d.rpt = 1;
    d.bicubic_int = 0;
    FusedSel S;
    sel_tap22(mode, out, vec, d);
    sel_two_tap_forms(mode, d);
    d.luma_only = (out == O_Y800_U8 || out == O_Y800_F32) ? 1 : 0;
    sel_store_policy(mode, out, vec, d);
    S.f32_out = (out == O_F32_PLANAR || out == O_F32_MERGED || out == O_NV12_F32 || out == O_Y800_F32 || out == O_HSV_F32);
    S.two_tap = (mode == M_BILINEAR || mode == M_AREA_UP);
    S.lds_budget = (size_t)(d.lds_budget_kb > 0 ? d.lds_budget_kb : 40) * 1024; // TSVPP_LDS_KB
    sel_shapes(mode, out, d, S);
    bool &staged = S.staged;
    size_t &lds_bytes = S.lds_bytes;
    d.tx = S.shapes[0][0];
    d.ty = S.shapes[0][1];
    sel_area(mode, vec, d, S);
    sel_point(mode, vec, d, S);
    d.bicubic_cols = 0;
    d.bc_sparse = 0;
    d.bc_dma = 0;
    // Interpolating kernels at large down-scale ratios tap only a few bytes of each source line: staging the whole
    // footprint through LDS then moves (and waits for) mostly unused bytes with few waves in flight, while plain
    // gathers touch each needed line once with full occupancy.  Measured cross-over (tools/matrix.sh, 1080p ->
    // 224^2 / 300^2 / 640^2, 4K -> 640x360; C3: 720p crop -> 256^2 = 14, gathers +8 %): BILINEAR gathers win from
    // xr*yr ~ 12 (3.5x at 41), BICUBIC from ~ 30.
    const float ratio_area = d.xr * d.yr;
    // the streaming kernels at the exact ratios 3 : 2 / 2 : 1 (stream_select above); a BICUBIC request that takes one needs neither staging nor tables
    const int stream_r32 = stream_select(mode, out, vec, d);
    const int bc_r32 = stream_r32 >= 7 ? stream_r32 : 0;
    // (BICUBIC: only the integer kernel for dyadic weights stages; everything else is vpp_bicubic_cols.hip)
    const bool bicubic_staged = mode == M_BICUBIC && d.w_dyadic && d.bicubic_int_pref && d.bicubic_cols_pref != 2 && ratio_area < 30.0f && !bc_r32;
    const bool sparse_gather = (mode == M_BILINEAR && ratio_area >= 12.0f) || (mode == M_BICUBIC && !bicubic_staged);
    sel_staged(mode, vec, bicubic_staged, sparse_gather, d, S);
    if (!staged) d.dma = 0;
    sel_bicubic_cols(mode, vec, bc_r32, d, S, stream, info);
    const size_t bc_lds = lds_bytes;
    if (mode == M_NONE && vec && d.in_aligned4 && !d.force_gather) staged = true; // colour-only fast path
    // ... and for the outputs that are the planes themselves (uint8 Y800 / NV12) a copy of 16 bytes per lane (vpp_copy16_kernel)
    d.copy16 = (mode == M_NONE && staged && (out == O_Y800_U8 || out == O_NV12_U8) && (d.dst_w & 15) == 0 && (d.dst_h & 3) == 0) ? 1 : 0;
    if (d.copy16 && !(d.shape_tx > 0 && d.shape_ty > 0)) {
        d.tx = 64;
        d.ty = 4;
    }
    d.r32 = stream_r32;
    if (d.r32) {
        d.point_kind = PK_NONE;
        d.area_direct = 0;
        staged = false;
        lds_bytes = 0;
        d.dma = 0;
        d.rpt = 1;
        d.geo = 0;
        if (!(d.shape_tx > 0 && d.shape_ty > 0)) stream_shape(d.r32, out, d, d.tx, d.ty); // (TSVPP_SHAPE overrides: shapes[0] above)
    }
    d.tx_shift = slot_shift_for(d.tx);
    if (d.r32) d.bicubic_cols = 0;
    if (d.bicubic_cols) lds_bytes = bc_lds;
    if (d.r32) d.area_stream = 0;
    if (d.area_stream) lds_bytes = S.as_lds;
    const int tile_w = d.area_stream ? (d.as_two ? 256 : 128) : d.bicubic_cols ? 256 : d.tx * (d.copy16 ? 16 : d.r32 ? 8 : PXW); // bicubic_cols: four waves side by side; area_stream: 2 x 2 waves
    const int tile_h = d.area_stream ? (d.as_two ? 8 * d.rpt : 16) : d.bicubic_cols ? 8 * d.rpt : (d.r32 || d.copy16) ? d.ty * 4 : d.ty * PXH * d.rpt;
    d.tiles_x = (d.dst_w + tile_w - 1) / tile_w;
    d.tiles_y = (d.dst_h + tile_h - 1) / tile_h;
    const long total = (long)d.tiles_x * d.tiles_y * d.n_frames;
    d.blocks_per_xcd = (int)((total + NUM_XCD - 1) / NUM_XCD);
    if (d.tile_order == 0) { // whole tile rows per XCD: rows padded to a multiple of 8
        const long rows = (long)d.tiles_y * d.n_frames;
        d.blocks_per_xcd = (int)(((rows + NUM_XCD - 1) / NUM_XCD) * d.tiles_x);
    }
    if (info) {
        info->tx = d.tx;
        info->ty = d.ty;
        info->rpt = d.rpt;
        info->dma = d.dma;
        info->staged = staged ? 1 : 0;
        info->tiles_x = d.tiles_x;
        info->tiles_y = d.tiles_y;
    }
    auto dispatch = [&](bool v, bool st, LaunchDesc &dd, size_t lds, LaunchInfo *inf) {
        switch (mode) {
        case M_NONE: return launch_m<M_NONE>(out, v, st, dd, t, lds, stream, inf);
        case M_NEAREST: return launch_m<M_NEAREST>(out, v, st, dd, t, lds, stream, inf);
        case M_BILINEAR: return launch_m<M_BILINEAR>(out, v, st, dd, t, lds, stream, inf);
        case M_BICUBIC: return launch_m<M_BICUBIC>(out, v, st, dd, t, lds, stream, inf);
        case M_AREA_DOWN: return launch_m<M_AREA_DOWN>(out, v, st, dd, t, lds, stream, inf);
        case M_AREA_UP: return launch_m<M_AREA_UP>(out, v, st, dd, t, lds, stream, inf);
        default: return hipErrorInvalidValue;
        }
    };
    if (out >= O_COUNT) { // flavours of the streaming kernel alone (O_UYVY_U8, O_YUV444_U8): the caller falls back to two passes
        if (!d.r32) return hipErrorNotSupported;
        return launch_bilinear_r32(out, d, t, stream, info);
    }
    hipError_t e = dispatch(vec, staged, d, lds_bytes, info);
    if (e != hipSuccess || !vec || (d.dst_w & 3) == 0) return e;
    // dst_w = 4 k + 2: the vector-store kernels left the last two columns of every row alone (is_row_tail); one more,
    // tiny launch of the element-wise gather kernel converts them: workgroup = 1 x 64 thread tiles of 2 columns x 2 rows
    if (info) {
        info->tail = 1;
        return e;
    }
    LaunchDesc td = d;
    td.col0 = d.dst_w & ~3;
    td.tx = 1;
    td.ty = 64; // one wave per workgroup: the few thousand tail threads spread over all CUs
    td.tx_shift = 0;
    td.rpt = 1;
    td.dma = 0;
    td.point_kind = PK_NONE; // the generic samplers give the point samplers' values (all weights are zero)
    td.area_direct = 0;
    td.bicubic_cols = 0;
    td.area_stream = 0;
    td.tiles_x = 1;
    td.tiles_y = (d.dst_h + td.ty * PXH - 1) / (td.ty * PXH);
    const long rows = (long)td.tiles_y * td.n_frames;
    td.blocks_per_xcd = (int)((rows + NUM_XCD - 1) / NUM_XCD); // tiles_x == 1: the same for every tile order
    return dispatch(false, false, td, 0, nullptr);
}

} // namespace tsvpp
