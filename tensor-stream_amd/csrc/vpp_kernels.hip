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
#include "vpp_kernels.h"

#pragma clang fp contract(off)

namespace tsvpp {

// Thread tile: 2 output rows x 4 output columns (= one resized-chroma row of 2 pairs).
// Workgroup: 256 threads as 32 x 8 thread tiles -> 128 x 16 output pixels.
constexpr int TX = 32, TY = 8, PXW = 4, PXH = 2;
constexpr int TILE_W = TX * PXW, TILE_H = TY * PXH;
constexpr int NUM_XCD = 8;

// ----------------------------------------------------------------------------------------------
// Source access.  Coordinates are clamped into the logical source so that no input, however odd,
// can fault; for every valid (even-sized) request the clamps never fire.
struct Src {
    const uint8_t *y, *uv;
    int py, puv; // pitches in bytes
    int w, h;    // logical source size in luma pixels
};

__device__ __forceinline__ int ld_y(const Src &s, int row, int col) {
    row = min(max(row, 0), s.h - 1);
    col = min(max(col, 0), s.w - 1);
    return s.y[(uint32_t)row * (uint32_t)s.py + (uint32_t)col];
}
// col is a BYTE column of the interleaved UV plane
__device__ __forceinline__ int ld_uv(const Src &s, int row, int col) {
    row = min(max(row, 0), (s.h >> 1) - 1);
    col = min(max(col, 0), s.w - 1);
    return s.uv[(uint32_t)row * (uint32_t)s.puv + (uint32_t)col];
}

// ----------------------------------------------------------------------------------------------
// Bilinear blend, reference src/Resize.cu:17-23: four products summed left to right, truncated.
__device__ __forceinline__ int bilerp(int A, int B, int C, int D, float wx, float wy) {
    float omx = 1.0f - wx, omy = 1.0f - wy;
    float t1 = ((float)A * omx) * omy;
    float t2 = ((float)B * wx) * omy;
    float t3 = ((float)C * wy) * omx;
    float t4 = (float)D * (wx * wy);
    float s = t1 + t2;
    s = s + t3;
    s = s + t4;
    return (int)s;
}

// Source coordinate + weight of one axis for BILINEAR (src/Resize.cu:276-303).
__device__ __forceinline__ void bilinear_axis(int idx, float ratio, int limit, int &p, float &w) {
    float f = ((float)idx + 0.5f) * ratio;
    f = f - 0.5f;
    p = (int)floorf(f);
    w = f - (float)p;
    if (p < 0) { p = 0; w = 0.f; }
    if (p > limit - 1) { p = limit - 1; w = 0.f; }
}
// ... for the AREA up-scale variant (src/Resize.cu:221-234).
__device__ __forceinline__ void areaup_axis(int idx, float ratio, int &p, float &w) {
    p = (int)floorf(ratio * (float)idx);
    float q = (float)(p + 1) / ratio;
    float f = (float)(idx + 1) - q;
    if (f <= 0.f) f = 0.f; else f = f - floorf(f);
    w = f;
}
// ... for BICUBIC (src/Resize.cu:321-347): fp32 coordinate widened to double.
__device__ __forceinline__ void bicubic_axis(int idx, float ratio, int limit, int &p, double &w) {
    float ff = ((float)idx + 0.5f) * ratio;
    ff = ff - 0.5f;
    double f = (double)ff;
    p = (int)floor(f);
    w = f - (double)p;
    if (p < 0) { p = 0; w = 0.0; }
    if (p > limit - 1) { p = limit - 1; w = 0.0; }
}

// Keys cubic, a = -0.75 (src/Resize.cu:45-50).  pow(w,2), pow(w,3) are the exact square and the
// correctly rounded cube: w has <= 24 significant bits (oracle/pow_pin.c).
__device__ __forceinline__ void cubic_coeffs(double w, double c[4]) {
    const double a = -0.75;
    double w2 = w * w, w3 = w2 * w;
    c[0] = (a * w - (2 * a) * w2) + a * w3;
    c[1] = (1 - (a + 3) * w2) + (a + 2) * w3;
    c[2] = ((-a) * w + (2 * a + 3) * w2) - (a + 2) * w3;
    c[3] = a * w2 - a * w3;
}
__device__ __forceinline__ int clamp255(int v) { return max(min(v, 255), 0); }
__device__ __forceinline__ int cubic4(const double c[4], int p0, int p1, int p2, int p3) {
    double a0 = c[0] * (double)p0, a1 = c[1] * (double)p1, a2 = c[2] * (double)p2, a3 = c[3] * (double)p3;
    double s = a0 + a1;
    s = s + a2;
    s = s + a3;
    return clamp255((int)round(s));
}
// Tap offsets with the reference's edge rule (src/Resize.cu:32-43): the +1 AND +2 taps collapse
// onto the centre when either would leave the plane; the -1 tap collapses at the low edge.
__device__ __forceinline__ void bicubic_offsets(int p, int step, int limit, int &lo, int &hi) {
    hi = step;
    lo = step;
    if (p + step >= limit) hi = 0;
    if (p + hi * 2 >= limit) hi = 0;
    if (p - step < 0) lo = 0;
}

// ----------------------------------------------------------------------------------------------
// Samplers: resized LUMA at output pixel (i, j) and resized CHROMA pair at chroma-grid (ci, cj).
// The chroma grid reuses the luma formulas on its own indices (the reference runs the same
// thread for both, guarded by i < H/2 && j < W/2).

template <int MODE>
__device__ __forceinline__ int sample_luma(const Src &s, const LaunchDesc &d, int i, int j) {
    if constexpr (MODE == M_NONE) {
        return ld_y(s, i, j);
    } else if constexpr (MODE == M_NEAREST) { // src/Resize.cu:249-258
        int y = (int)(d.yr * (float)i), x = (int)(d.xr * (float)j);
        return ld_y(s, y, x);
    } else if constexpr (MODE == M_BILINEAR || MODE == M_AREA_UP) {
        int x, y;
        float wx, wy;
        if constexpr (MODE == M_BILINEAR) {
            bilinear_axis(j, d.xr, s.w, x, wx);
            bilinear_axis(i, d.yr, s.h, y, wy);
        } else {
            areaup_axis(j, d.xr, x, wx);
            areaup_axis(i, d.yr, y, wy);
        }
        int xd = (x + 1 >= s.w) ? 0 : 1;
        int y2 = (y + 1 >= s.h) ? y : y + 1;
        return bilerp(ld_y(s, y, x), ld_y(s, y, x + xd), ld_y(s, y2, x), ld_y(s, y2, x + xd), wx, wy) & 0xff;
    } else if constexpr (MODE == M_BICUBIC) {
        int x, y;
        double wx, wy;
        bicubic_axis(j, d.xr, s.w, x, wx);
        bicubic_axis(i, d.yr, s.h, y, wy);
        int xl, xh, yl, yh;
        bicubic_offsets(x, 1, s.w, xl, xh);
        bicubic_offsets(y, 1, s.h, yl, yh);
        double cx[4], cy[4];
        cubic_coeffs(wx, cx);
        cubic_coeffs(wy, cy);
        const int rows[4] = { y - yl, y, y + yh, y + 2 * yh };
        int b[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            b[r] = cubic4(cx, ld_y(s, rows[r], x - xl), ld_y(s, rows[r], x), ld_y(s, rows[r], x + xh), ld_y(s, rows[r], x + 2 * xh));
        return cubic4(cy, b[0], b[1], b[2], b[3]);
    } else { // M_AREA_DOWN, src/Resize.cu:160-178, 186-201
        int y = (int)(d.yr * (float)i), x = (int)(d.xr * (float)j);
        const float *px = d.patx + (j % d.nx) * d.rx;
        const float *py = d.paty + (i % d.ny) * d.ry;
        float sum = 0.f, div = 0.f;
        for (int a = 0; a < d.ry; a++) {
            float wy = py[a];
            for (int b = 0; b < d.rx; b++) {
                float wgt = px[b] * wy;
                div = div + wgt;
                float v = (float)ld_y(s, y + a, x + b) * wgt;
                sum = sum + v;
            }
        }
        sum = sum / div;
        return (int)sum & 0xff;
    }
}

template <int MODE>
__device__ __forceinline__ void sample_chroma(const Src &s, const LaunchDesc &d, int ci, int cj, int &U, int &V) {
    const int ch = s.h >> 1; // rows of the UV plane
    if constexpr (MODE == M_NONE) {
        U = ld_uv(s, ci, 2 * cj);
        V = ld_uv(s, ci, 2 * cj + 1);
    } else if constexpr (MODE == M_NEAREST) { // src/Resize.cu:262-265
        int y = (int)(d.yr * (float)ci), x = (int)(d.xr * (float)cj);
        U = ld_uv(s, y, 2 * x);
        V = ld_uv(s, y, 2 * x + 1);
    } else if constexpr (MODE == M_BILINEAR || MODE == M_AREA_UP) { // src/Resize.cu:308-309, 236-237
        int x, y;
        float wx, wy;
        if constexpr (MODE == M_BILINEAR) {
            bilinear_axis(cj, d.xr, s.w, x, wx);
            bilinear_axis(ci, d.yr, s.h, y, wy);
        } else {
            areaup_axis(cj, d.xr, x, wx);
            areaup_axis(ci, d.yr, y, wy);
        }
        int xu = 2 * x, xv = 2 * x + 1;
        int du = (xu + 2 >= s.w) ? 0 : 2;
        int dv = (xv + 2 >= s.w) ? 0 : 2;
        int y2 = (y + 1 >= ch) ? y : y + 1;
        U = bilerp(ld_uv(s, y, xu), ld_uv(s, y, xu + du), ld_uv(s, y2, xu), ld_uv(s, y2, xu + du), wx, wy) & 0xff;
        V = bilerp(ld_uv(s, y, xv), ld_uv(s, y, xv + dv), ld_uv(s, y2, xv), ld_uv(s, y2, xv + dv), wx, wy) & 0xff;
    } else if constexpr (MODE == M_BICUBIC) { // src/Resize.cu:353-354
        int x, y;
        double wx, wy;
        bicubic_axis(cj, d.xr, s.w, x, wx);
        bicubic_axis(ci, d.yr, s.h, y, wy);
        int yl, yh;
        bicubic_offsets(y, 1, ch, yl, yh);
        double cx[4], cy[4];
        cubic_coeffs(wx, cx);
        cubic_coeffs(wy, cy);
        const int rows[4] = { y - yl, y, y + yh, y + 2 * yh };
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            int xc = 2 * x + comp, xl, xh;
            bicubic_offsets(xc, 2, s.w, xl, xh);
            int b[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                b[r] = cubic4(cx, ld_uv(s, rows[r], xc - xl), ld_uv(s, rows[r], xc), ld_uv(s, rows[r], xc + xh), ld_uv(s, rows[r], xc + 2 * xh));
            int v = cubic4(cy, b[0], b[1], b[2], b[3]);
            if (comp == 0) U = v; else V = v;
        }
    } else { // M_AREA_DOWN, src/Resize.cu:204-210: same x, y and the SAME weight rows, stride 2
        int y = (int)(d.yr * (float)ci), x = (int)(d.xr * (float)cj);
        const float *px = d.patx + (cj % d.nx) * d.rx;
        const float *py = d.paty + (ci % d.ny) * d.ry;
        float su = 0.f, sv = 0.f, div = 0.f;
        for (int a = 0; a < d.ry; a++) {
            float wy = py[a];
            for (int b = 0; b < d.rx; b++) {
                float wgt = px[b] * wy;
                div = div + wgt;
                float vu = (float)ld_uv(s, y + a, 2 * x + 2 * b) * wgt;
                float vv = (float)ld_uv(s, y + a, 2 * x + 2 * b + 1) * wgt;
                su = su + vu;
                sv = sv + vv;
            }
        }
        su = su / div;
        sv = sv / div;
        U = (int)su & 0xff;
        V = (int)sv & 0xff;
    }
}

// ----------------------------------------------------------------------------------------------
// BT.601 limited-range YUV -> RGB, reference src/ColorConversion.cu:23-38.
__device__ __forceinline__ void yuv2rgb(int Y, int U, int V, const tsvpp_coeffs &k, int &R, int &G, int &B) {
    float yv = fmaxf(0.f, (float)Y - k.y_offset) * k.y_scale;
    float fu = (float)U - k.c_offset, fv = (float)V - k.c_offset;
    float rv = k.v_to_r * fv;
    rv = rv + k.round_bias;
    float bv = k.u_to_b * fu;
    bv = bv + k.round_bias;
    float g1 = k.v_to_g * fv;
    float g2 = k.u_to_g * fu; // u_to_g is negative: g1 + g2 == g1 - |u_to_g|*fu exactly
    float gv = g1 + g2;
    gv = gv + k.round_bias;
    R = clamp255((int)(yv + rv));
    B = clamp255((int)(yv + bv));
    G = clamp255((int)(yv + gv));
}

// v / 255 for an integer v in [0, 255], correctly rounded (== IEEE division, verified for all
// 256 inputs in tests): reciprocal multiply + one explicit-FMA Newton correction, 3 VALU ops
// instead of the ~10-instruction v_div_scale/fmas/fixup sequence.
__device__ __forceinline__ float norm255(int v) {
    const float r = 1.0f / 255.0f;
    float f = (float)v;
    float q = f * r;
    float e = __builtin_fmaf(-q, 255.0f, f);
    return __builtin_fmaf(e, r, q);
}

template <int OUT> struct OutT { using type = uint8_t; };
template <> struct OutT<O_F32_PLANAR> { using type = float; };
template <> struct OutT<O_F32_MERGED> { using type = float; };

__device__ __forceinline__ float cvt_out(int v, float *) { return norm255(v); }
__device__ __forceinline__ uint8_t cvt_out(int v, uint8_t *) { return (uint8_t)v; }

// ----------------------------------------------------------------------------------------------
template <int MODE, int OUT, bool VEC>
__global__ __launch_bounds__(TX * TY) void vpp_fused_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr bool PLANAR = (OUT == O_U8_PLANAR || OUT == O_F32_PLANAR);

    // XCD-aware decomposition: consecutive workgroup ids land on different XCDs (id % 8), so give
    // every XCD a contiguous run of (frame, tile) work -- neighbouring tiles then share their
    // source halo rows in ONE L2 instead of fetching them into two.
    const int total = d.tiles_x * d.tiles_y * d.n_frames;
    const int logical = (blockIdx.x % NUM_XCD) * d.blocks_per_xcd + blockIdx.x / NUM_XCD;
    if (logical >= total) return;
    const int tiles = d.tiles_x * d.tiles_y;
    const int frame = logical / tiles;
    const int rem = logical - frame * tiles;
    const int ty = rem / d.tiles_x, tx = rem - ty * d.tiles_x;

    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int j0 = tx * TILE_W + lx * PXW;
    const int i0 = ty * TILE_H + ly * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;

    Src s;
    s.y = t.y[frame];
    s.uv = t.uv[frame];
    s.py = d.pitch_y;
    s.puv = d.pitch_uv;
    s.w = d.src_w;
    s.h = d.src_h;
    T *out = (T *)t.out[frame];

    const int ncol = VEC ? PXW : min(PXW, d.dst_w - j0); // dst_w is even: 2 or 4
    const int ci = i0 >> 1, cj0 = j0 >> 1;

    int U[2], V[2], Y[PXH][PXW];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        U[c] = V[c] = 0;
        if (VEC || 2 * c < ncol) sample_chroma<MODE>(s, d, ci, cj0 + c, U[c], V[c]);
    }
#pragma unroll
    for (int r = 0; r < PXH; r++)
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            Y[r][c] = 0;
            if (VEC || c < ncol) Y[r][c] = sample_luma<MODE>(s, d, i0 + r, j0 + c);
        }

    const size_t plane = (size_t)d.dst_w * (size_t)d.dst_h;
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        T c0[PXW], c1[PXW], c2[PXW];
#pragma unroll
        for (int c = 0; c < PXW; c++) {
            int R, G, B;
            yuv2rgb(Y[r][c], U[c >> 1], V[c >> 1], d.k, R, G, B);
            c0[c] = cvt_out(d.swap_rb ? B : R, (T *)nullptr);
            c1[c] = cvt_out(G, (T *)nullptr);
            c2[c] = cvt_out(d.swap_rb ? R : B, (T *)nullptr);
        }
        const size_t pix = (size_t)(i0 + r) * (size_t)d.dst_w + (size_t)j0;
        if constexpr (PLANAR) {
            if constexpr (VEC) {
                if constexpr (sizeof(T) == 4) {
                    *(float4 *)(out + pix) = make_float4(c0[0], c0[1], c0[2], c0[3]);
                    *(float4 *)(out + plane + pix) = make_float4(c1[0], c1[1], c1[2], c1[3]);
                    *(float4 *)(out + 2 * plane + pix) = make_float4(c2[0], c2[1], c2[2], c2[3]);
                } else {
                    *(uchar4 *)(out + pix) = make_uchar4(c0[0], c0[1], c0[2], c0[3]);
                    *(uchar4 *)(out + plane + pix) = make_uchar4(c1[0], c1[1], c1[2], c1[3]);
                    *(uchar4 *)(out + 2 * plane + pix) = make_uchar4(c2[0], c2[1], c2[2], c2[3]);
                }
            } else {
                for (int c = 0; c < ncol; c++) {
                    out[pix + c] = c0[c];
                    out[plane + pix + c] = c1[c];
                    out[2 * plane + pix + c] = c2[c];
                }
            }
        } else {
            T *o = out + 3 * pix;
            if constexpr (VEC) {
                if constexpr (sizeof(T) == 4) {
                    ((float4 *)o)[0] = make_float4(c0[0], c1[0], c2[0], c0[1]);
                    ((float4 *)o)[1] = make_float4(c1[1], c2[1], c0[2], c1[2]);
                    ((float4 *)o)[2] = make_float4(c2[2], c0[3], c1[3], c2[3]);
                } else {
                    ((uchar4 *)o)[0] = make_uchar4(c0[0], c1[0], c2[0], c0[1]);
                    ((uchar4 *)o)[1] = make_uchar4(c1[1], c2[1], c0[2], c1[2]);
                    ((uchar4 *)o)[2] = make_uchar4(c2[2], c0[3], c1[3], c2[3]);
                }
            } else {
                for (int c = 0; c < ncol; c++) {
                    o[3 * c] = c0[c];
                    o[3 * c + 1] = c1[c];
                    o[3 * c + 2] = c2[c];
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
template <int MODE, int OUT>
static hipError_t launch_mo(bool vec, const LaunchDesc &d, const FrameTable &t, hipStream_t stream) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block(TX * TY);
    if (vec)
        hipLaunchKernelGGL((vpp_fused_kernel<MODE, OUT, true>), grid, block, 0, stream, d, t);
    else
        hipLaunchKernelGGL((vpp_fused_kernel<MODE, OUT, false>), grid, block, 0, stream, d, t);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_m(OutKind out, bool vec, const LaunchDesc &d, const FrameTable &t, hipStream_t stream) {
    switch (out) {
    case O_U8_PLANAR: return launch_mo<MODE, O_U8_PLANAR>(vec, d, t, stream);
    case O_U8_MERGED: return launch_mo<MODE, O_U8_MERGED>(vec, d, t, stream);
    case O_F32_PLANAR: return launch_mo<MODE, O_F32_PLANAR>(vec, d, t, stream);
    case O_F32_MERGED: return launch_mo<MODE, O_F32_MERGED>(vec, d, t, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fused(Mode mode, OutKind out, bool vec, const LaunchDesc &din, const FrameTable &t, hipStream_t stream) {
    LaunchDesc d = din;
    d.tiles_x = (d.dst_w + TILE_W - 1) / TILE_W;
    d.tiles_y = (d.dst_h + TILE_H - 1) / TILE_H;
    const long total = (long)d.tiles_x * d.tiles_y * d.n_frames;
    d.blocks_per_xcd = (int)((total + NUM_XCD - 1) / NUM_XCD);
    switch (mode) {
    case M_NONE: return launch_m<M_NONE>(out, vec, d, t, stream);
    case M_NEAREST: return launch_m<M_NEAREST>(out, vec, d, t, stream);
    case M_BILINEAR: return launch_m<M_BILINEAR>(out, vec, d, t, stream);
    case M_BICUBIC: return launch_m<M_BICUBIC>(out, vec, d, t, stream);
    case M_AREA_DOWN: return launch_m<M_AREA_DOWN>(out, vec, d, t, stream);
    case M_AREA_UP: return launch_m<M_AREA_UP>(out, vec, d, t, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace tsvpp
