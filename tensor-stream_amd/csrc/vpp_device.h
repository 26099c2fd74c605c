// vpp_device.h -- device-side building blocks shared by the kernel translation units (vpp_kernels.hip,
// vpp_bicubic_int.hip): source readers, samplers, the colour back end (color_store_tile), the work decomposition
// (decode_tile), footprints and LDS staging.  Product code -- never includes anything from oracle/.
//
// Arithmetic contract: every float/double operation is a single IEEE-754 operation in the order the reference's source
// text gives it -- NO fused multiply-add (contraction is off in every file that includes this header), truncating
// float->int conversions, round-half-away for the bicubic stage.  The only fma()s are explicit ones.
#pragma once
#include "vpp_kernels.h"
#include "vpp_axis.h"

#pragma clang fp contract(off)

namespace tsvpp {

// Thread tile: 2 output rows x 4 output columns (= one resized-chroma row of 2 pairs).
// Workgroup: tx x ty thread tiles (launch-time choice, tx a power of two, tx*ty <= 256), i.e.
// (4*tx) x (2*ty) output pixels; 32 x 8 threads -> 128 x 16 pixels is the default.
constexpr int PXW = 4, PXH = 2;
constexpr int MAX_THREADS = 256;
constexpr int NUM_XCD = 8;

// Two floats per lane: gfx950 executes v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on such pairs in
// the time of one scalar op, and each half is an independent IEEE operation -- results are
// bit-identical to the scalar sequence.  The VPP kernels are VALU-issue bound before they are
// HBM bound, so the blend and colour arithmetic is written on horizontally adjacent pixel pairs.
typedef float f2 __attribute__((ext_vector_type(2)));

// ----------------------------------------------------------------------------------------------
// Source access.  Two readers with the same interface:
//   GlobalSrc -- byte gathers straight from global memory (fallback: huge footprints, odd widths).
//                Coordinates are clamped into the logical source so that no input can fault; for
//                every valid (even-sized) request the clamps never fire.
//   LdsSrc    -- the workgroup's source footprint staged in LDS by 16-byte coalesced loads.
struct GlobalSrc {
    const uint8_t *y, *uv;
    int py, puv; // pitches in bytes
    int w, h;    // logical source size in luma pixels
    __device__ __forceinline__ int Y(int row, int col) const {
        row = min(max(row, 0), h - 1);
        col = min(max(col, 0), w - 1);
        return y[(uint32_t)row * (uint32_t)py + (uint32_t)col];
    }
    // col is a BYTE column of the interleaved UV plane
    __device__ __forceinline__ int UV(int row, int col) const {
        row = min(max(row, 0), (h >> 1) - 1);
        col = min(max(col, 0), w - 1);
        return uv[(uint32_t)row * (uint32_t)puv + (uint32_t)col];
    }
};

// One staged plane: LDS row r holds source row (y0 + r); because global loads are 16-byte aligned
// and the pitch need not be, each row is shifted by its own misalignment (m0 + r*pm) & 15.
struct LdsPlane {
    const uint8_t *base;
    int x0, y0; // source byte-column / row of the footprint origin
    int lp;     // LDS row pitch in bytes (multiple of 16)
    int m0, pm; // misalignment of row 0, pitch & 15
    __device__ __forceinline__ int at(int row, int col) const {
        const int r = row - y0;
        return base[r * lp + ((m0 + r * pm) & 15) + (col - x0)];
    }
};
struct LdsSrc {
    LdsPlane py_, puv_;
    int w, h;
    __device__ __forceinline__ int Y(int row, int col) const { return py_.at(row, col); }
    __device__ __forceinline__ int UV(int row, int col) const { return puv_.at(row, col); }
};

// ----------------------------------------------------------------------------------------------
// Bilinear blend, reference src/Resize.cu:17-23, with the operation tree of the reference's binary: nvcc (fmad on) fuses
//     A (1-wx) (1-wy) + B wx (1-wy) + C wy (1-wx) + D (wx wy)
// into  fma(D, wx wy, fma(C wy, 1-wx, fma(A (1-wx), 1-wy, (B wx) (1-wy)))) -- the left product of the first sum, then each
// further product, exactly the pattern that reproduces all of the reference's BILINEAR / AREA CRC goldens and the only
// one of the 96 candidate patterns that does (oracle/vpp_oracle.c).  Truncated.
__device__ __forceinline__ int bilerp(int A, int B, int C, int D, float wx, float wy) {
    float omx = 1.0f - wx, omy = 1.0f - wy;
    float s = __builtin_fmaf((float)A * omx, omy, ((float)B * wx) * omy);
    s = __builtin_fmaf((float)C * wy, omx, s);
    s = __builtin_fmaf((float)D, wx * wy, s);
    return (int)s;
}
// the same on float pairs (v_pk_mul_f32 / v_pk_fma_f32)
__device__ __forceinline__ f2 bilerp2(f2 A, f2 B, f2 C, f2 D, f2 wx, f2 omx, f2 wy, f2 omy) {
    f2 s = __builtin_elementwise_fma(A * omx, omy, (B * wx) * omy);
    s = __builtin_elementwise_fma(C * wy, omx, s);
    return __builtin_elementwise_fma(D, wx * wy, s);
}

// Keys cubic, a = -0.75 (src/Resize.cu:45-50).  pow(w,2), pow(w,3) are the exact square and the
// correctly rounded cube: w has <= 24 significant bits (oracle/pow_pin.c).
__host__ __device__ __forceinline__ void cubic_coeffs(double w, double c[4]) {
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
// Mixed-precision form of cubic4.  The reference's value is round(s64) of a 4-term fp64 sum that is
// at most a few hundred in magnitude; an fp32 evaluation s32 of the same sum differs from it by far
// less than CUBIC_DELTA (coefficient rounding 255*4*6e-8, products and sums < 2e-4 in total), so
// wherever s32 is further than CUBIC_DELTA from a half-integer round(s32) == round(s64) and the
// fp64 arithmetic (half rate, plus the coefficient polynomials) is skipped.  The few lanes that are
// close to a tie recompute exactly as the reference does.  `w` is the weight as a float: it IS
// exact (fraction of a float coordinate).
constexpr float CUBIC_DELTA = 5e-4f;
__device__ __forceinline__ void cubic_coeffs_f(float w, float c[4]) {
    const float a = -0.75f;
    const float w2 = w * w, w3 = w2 * w;
    c[0] = (a * w - (2 * a) * w2) + a * w3;
    c[1] = (1 - (a + 3) * w2) + (a + 2) * w3;
    c[2] = ((-a) * w + (2 * a + 3) * w2) - (a + 2) * w3;
    c[3] = a * w2 - a * w3;
}
__device__ __forceinline__ int cubic4_mixed(float w, const float cf[4], int p0, int p1, int p2, int p3) {
    const float s = ((cf[0] * (float)p0 + cf[1] * (float)p1) + cf[2] * (float)p2) + cf[3] * (float)p3;
    const float fl = floorf(s);
    if (fabsf((s - fl) - 0.5f) < CUBIC_DELTA) { // within reach of a tie: the reference's fp64 arithmetic decides
        double c[4];
        cubic_coeffs((double)w, c);
        return cubic4(c, p0, p1, p2, p3);
    }
    return clamp255((int)fl + ((s - fl) > 0.5f ? 1 : 0)); // round to nearest; negative sums clamp to 0 either way
}

// Two 4-tap sums at once on float pairs (packed VALU).  Rounding by the 1.5 * 2^23 trick (nearest
// even -- it differs from the reference's half-away rule only AT a tie, and anything within
// CUBIC_DELTA of a tie is recomputed exactly); results stay integer-valued floats, clamped.
__device__ __forceinline__ f2 cubic4_pair(float w0, float w1, const float c0[4], const float c1[4], const f2 p[4], const int q0[4], const int q1[4]) {
    f2 s = ((f2){ c0[0], c1[0] } * p[0] + (f2){ c0[1], c1[1] } * p[1]) + (f2){ c0[2], c1[2] } * p[2];
    s = s + (f2){ c0[3], c1[3] } * p[3];
    const f2 magic = { 12582912.0f, 12582912.0f };
    f2 r = (s + magic) - magic;
    const f2 dd = s - r;
    r.x = __builtin_amdgcn_fmed3f(r.x, 0.0f, 255.0f);
    r.y = __builtin_amdgcn_fmed3f(r.y, 0.0f, 255.0f);
    const bool tie0 = fabsf(dd.x) > 0.5f - CUBIC_DELTA, tie1 = fabsf(dd.y) > 0.5f - CUBIC_DELTA;
    if (tie0 || tie1) { // one (rarely taken) branch for the pair
        if (tie0) {
            double c[4];
            cubic_coeffs((double)w0, c);
            r.x = (float)cubic4(c, q0[0], q0[1], q0[2], q0[3]);
        }
        if (tie1) {
            double c[4];
            cubic_coeffs((double)w1, c);
            r.y = (float)cubic4(c, q1[0], q1[1], q1[2], q1[3]);
        }
    }
    return r;
}

// Tap offsets with the reference's edge rule (src/Resize.cu:32-43): the +1 AND +2 taps collapse
// onto the centre when either would leave the plane; the -1 tap collapses at the low edge.
__host__ __device__ __forceinline__ void bicubic_offsets(int p, int step, int limit, int &lo, int &hi) {
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

template <int MODE, class S>
__device__ __forceinline__ int sample_luma(const S &s, const LaunchDesc &d, int i, int j) {
    if constexpr (MODE == M_NONE) {
        return s.Y(i, j);
    } else if constexpr (MODE == M_NEAREST) { // src/Resize.cu:249-258
        int y = (int)(d.yr * (float)i), x = (int)(d.xr * (float)j);
        return s.Y(y, x);
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
        if constexpr (MODE == M_BILINEAR) {
            // an axis whose weights are all zero (host: odd integer ratio on that axis): the taps the zero multiplies are not fetched -- any finite
            // value gives the same bits ((float)B * 0 == +0) -- which halves the gathers of such a request (BASELINE config C3)
            if (d.wx_zero) {
                const int A = s.Y(y, x), C = s.Y(y2, x);
                return bilerp(A, A, C, C, wx, wy) & 0xff;
            }
            if (d.wy_zero) {
                const int A = s.Y(y, x), B = s.Y(y, x + xd);
                return bilerp(A, B, A, B, wx, wy) & 0xff;
            }
        }
        return bilerp(s.Y(y, x), s.Y(y, x + xd), s.Y(y2, x), s.Y(y2, x + xd), wx, wy) & 0xff;
    } else if constexpr (MODE == M_BICUBIC) {
        int x, y;
        double wx, wy;
        bicubic_axis(j, d.xr, s.w, x, wx);
        bicubic_axis(i, d.yr, s.h, y, wy);
        int xl, xh, yl, yh;
        bicubic_offsets(x, 1, s.w, xl, xh);
        bicubic_offsets(y, 1, s.h, yl, yh);
        const float wxf = (float)wx, wyf = (float)wy; // exact
        float cx[4], cy[4];
        cubic_coeffs_f(wxf, cx);
        cubic_coeffs_f(wyf, cy);
        const int rows[4] = { y - yl, y, y + yh, y + 2 * yh };
        int b[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            b[r] = cubic4_mixed(wxf, cx, s.Y(rows[r], x - xl), s.Y(rows[r], x), s.Y(rows[r], x + xh), s.Y(rows[r], x + 2 * xh));
        return cubic4_mixed(wyf, cy, b[0], b[1], b[2], b[3]);
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
                sum = __builtin_fmaf((float)s.Y(y + a, x + b), wgt, sum); // colorSum += data * weight: fused in the reference's binary
            }
        }
        sum = sum / div;
        return (int)sum & 0xff;
    }
}

template <int MODE, class S>
__device__ __forceinline__ void sample_chroma(const S &s, const LaunchDesc &d, int ci, int cj, int &U, int &V) {
    const int ch = s.h >> 1; // rows of the UV plane
    if constexpr (MODE == M_NONE) {
        U = s.UV(ci, 2 * cj);
        V = s.UV(ci, 2 * cj + 1);
    } else if constexpr (MODE == M_NEAREST) { // src/Resize.cu:262-265
        int y = (int)(d.yr * (float)ci), x = (int)(d.xr * (float)cj);
        U = s.UV(y, 2 * x);
        V = s.UV(y, 2 * x + 1);
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
        if constexpr (MODE == M_BILINEAR) {
            if (d.wx_zero) {
                const int Au = s.UV(y, xu), Cu = s.UV(y2, xu), Av = s.UV(y, xv), Cv = s.UV(y2, xv);
                U = bilerp(Au, Au, Cu, Cu, wx, wy) & 0xff;
                V = bilerp(Av, Av, Cv, Cv, wx, wy) & 0xff;
                return;
            }
            if (d.wy_zero) {
                const int Au = s.UV(y, xu), Bu = s.UV(y, xu + du), Av = s.UV(y, xv), Bv = s.UV(y, xv + dv);
                U = bilerp(Au, Bu, Au, Bu, wx, wy) & 0xff;
                V = bilerp(Av, Bv, Av, Bv, wx, wy) & 0xff;
                return;
            }
        }
        U = bilerp(s.UV(y, xu), s.UV(y, xu + du), s.UV(y2, xu), s.UV(y2, xu + du), wx, wy) & 0xff;
        V = bilerp(s.UV(y, xv), s.UV(y, xv + dv), s.UV(y2, xv), s.UV(y2, xv + dv), wx, wy) & 0xff;
    } else if constexpr (MODE == M_BICUBIC) { // src/Resize.cu:353-354
        int x, y;
        double wx, wy;
        bicubic_axis(cj, d.xr, s.w, x, wx);
        bicubic_axis(ci, d.yr, s.h, y, wy);
        int yl, yh;
        bicubic_offsets(y, 1, ch, yl, yh);
        const float wxf = (float)wx, wyf = (float)wy; // exact
        float cx[4], cy[4];
        cubic_coeffs_f(wxf, cx);
        cubic_coeffs_f(wyf, cy);
        const int rows[4] = { y - yl, y, y + yh, y + 2 * yh };
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            int xc = 2 * x + comp, xl, xh;
            bicubic_offsets(xc, 2, s.w, xl, xh);
            int b[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                b[r] = cubic4_mixed(wxf, cx, s.UV(rows[r], xc - xl), s.UV(rows[r], xc), s.UV(rows[r], xc + xh), s.UV(rows[r], xc + 2 * xh));
            int v = cubic4_mixed(wyf, cy, b[0], b[1], b[2], b[3]);
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
                su = __builtin_fmaf((float)s.UV(y + a, 2 * x + 2 * b), wgt, su);
                sv = __builtin_fmaf((float)s.UV(y + a, 2 * x + 2 * b + 1), wgt, sv);
            }
        }
        su = su / div;
        sv = sv / div;
        U = (int)su & 0xff;
        V = (int)sv & 0xff;
    }
}

// ----------------------------------------------------------------------------------------------
// Colour back end shared by every kernel.  Inputs are the resized samples as integer-valued floats:
// luma Yf[2][4] and one (U, V) pair per 2x2 block.  BT.601 limited range, reference
// src/ColorConversion.cu:23-38:
//     Yv = max(0, Y - 16) * 1.164;  R = (int)(Yv + (1.596 (V-128) + 0.5)) ... min 255, max 0.
// (int)x truncates toward zero == trunc(x); the clamp is done on the truncated float.

template <int OUT> struct OutT { using type = uint8_t; };
template <> struct OutT<O_F32_PLANAR> { using type = float; };
template <> struct OutT<O_F32_MERGED> { using type = float; };
template <> struct OutT<O_NV12_F32> { using type = float; };
template <> struct OutT<O_Y800_F32> { using type = float; };
template <> struct OutT<O_HSV_F32> { using type = float; };
template <int OUT> constexpr bool kLumaOnly = (OUT == O_Y800_U8 || OUT == O_Y800_F32);

// Per-block chroma terms: t0 / t2 are added to luma for the first / third stored channel
// (R,B or B,R when swapped), tg for green.  The operation tree is the reference's AS COMPILED (reference src/ColorConversion.cu:25-36 under nvcc's default
// -fmad=true; oracle/vpp_oracle.c, CT_NVCC -- the fusion rule the resize goldens pin, applied to this stage):
//     RVal = fma(1.596, V - 128, 0.5), BVal = fma(2.018, U - 128, 0.5)                      (single-use products: fused)
//     GVal = fma(-0.813, V - 128, -(0.391 (U - 128))) + 0.5                                 (the subtraction's left product fused, the right one rounded)
// and the luma product Y' = max(0, Y - 16) * 1.164 stays a rounded product (it feeds three sums).  R and B do not depend on the contraction for any
// (Y, U, V); G differs from the plain-IEEE tree of rounds 1-4 by one on 36 of the 2^24 triples (tests/test_oracle_contract.py).
// `g_term` (LaunchDesc::color_g, wave-uniform: a scalar branch) selects the other two trees of the green term for holders of goldens of the real reference binary
// (TSVPP_OPT_COLOR_G_TERM, include/tsvpp.h): 1 = both products rounded, 2 = the right product fused.
__device__ __forceinline__ void chroma_terms(float Uf, float Vf, const tsvpp_coeffs &k, int swap_rb, int g_term, float &t0, float &tg, float &t2) {
    f2 uv = { Uf, Vf };
    uv = uv - (f2){ k.c_offset, k.c_offset };
    const f2 br = __builtin_elementwise_fma(uv, (f2){ k.u_to_b, k.v_to_r }, (f2){ k.round_bias, k.round_bias }); // { 2.018 (U-128) + .5, 1.596 (V-128) + .5 }
    float gv;
    if (g_term == 0) {
        const float gu = uv.x * k.u_to_g;        // u_to_g < 0: -(0.391 (U-128)), rounded
        gv = __builtin_fmaf(uv.y, k.v_to_g, gu); // -0.813 (V-128) - 0.391 (U-128): ONE rounding of the left product's sum
    } else if (g_term == 1) {
        const float g1 = uv.y * k.v_to_g, g2 = uv.x * k.u_to_g;
        gv = g1 + g2;                            // == g1 - 0.391 (U-128): negation is exact
    } else {
        const float g1 = uv.y * k.v_to_g;
        gv = __builtin_fmaf(uv.x, k.u_to_g, g1);
    }
    tg = gv + k.round_bias;
    t0 = swap_rb ? br.x : br.y;
    t2 = swap_rb ? br.y : br.x;
}

__device__ __forceinline__ f2 trunc_clamp255(f2 v) {
    f2 r;
    r.x = __builtin_amdgcn_fmed3f(__builtin_truncf(v.x), 0.0f, 255.0f);
    r.y = __builtin_amdgcn_fmed3f(__builtin_truncf(v.y), 0.0f, 255.0f);
    return r;
}

// v / 255 for an integer-valued v in [0, 255], correctly rounded (== the reference's IEEE
// `/= 255`; all 256 inputs are checked by the tests): 1/255 = hi + lo, e = v*lo, q = fma(v, hi, e).
// Two (packed) VALU ops instead of the ~10-instruction v_div_scale/fmas/fixup sequence; a zero of
// either sign yields +0.0 like the reference's int -> float conversion.
__device__ __forceinline__ f2 norm255(f2 v) {
    const float hi = 0x1.010102p-8f, lo = -0x1.fdfdfep-33f;
    f2 e = v * (f2){ lo, lo };
    f2 q;
    q.x = __builtin_fmaf(v.x, hi, e.x);
    q.y = __builtin_fmaf(v.y, hi, e.y);
    return q;
}

// a / b, correctly rounded, for operands whose quotient needs no scaling: this is the refinement chain of the
// compiler's own IEEE fp32 division (rcp, one Newton step on the reciprocal, two on the quotient) without the
// v_div_scale / v_div_fixup bracket that only matters for denormal, overflowing or special operands.  HSV
// operands are k/255 fractions and hues in [-60, 420] over deltas >= 1/255; lanes with b == 0 produce NaN
// here and are discarded by the caller's selects, as their IEEE counterparts are.
__device__ __forceinline__ float div_inrange(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}

// Normalised RGB -> HSV in [0, 1], the reference's RGBMergedToHSVMerged (src/ColorConversion.cu:235-278),
// branch-free: the hue sector is chosen with selects, every operation keeps its place and its IEEE
// rounding.
__device__ __forceinline__ void hsv_pixel(float R, float G, float B, float &H, float &S, float &V) {
    const float mn = __builtin_fminf(__builtin_fminf(R, G), B), mx = __builtin_fmaxf(__builtin_fmaxf(R, G), B);
    const float delta = mx - mn;
    V = mx;
    const float q = div_inrange(mn, mx);
    S = (mx != 0.0f) ? 1.0f - q : 0.0f;
    const bool r = (R == mx), g = (G == mx);
    const float diff = r ? G - B : (g ? B - R : R - G);
    const float off = r ? (G < B ? 360.0f : 0.0f) : (g ? 120.0f : 240.0f);
    float h = 60.0f * diff;
    h = div_inrange(h, delta);
    h = h + off;
    if (h < 0.0f) h = h + 360.0f;
    h = div_inrange(h, 360.0f);
    H = (mx == mn) ? 0.0f : h;
}

typedef float vf4 __attribute__((ext_vector_type(4)));
// Output store policy (LaunchDesc::nt_stores): 0 plain, 1 non-temporal, 2 `sc1` write-through.  The
// output is written once and never re-read by the kernel, so it should not displace the input
// lines in L2: measured on MI355X (tools/ab.sh TSVPP_NT=0/1/2) non-temporal stores gain 3-20 % on
// the resize kernels, `sc1` 14 % on the colour-only fp32 kernel (25 MB written per 3 MB read) but
// lose 36 % on 4-byte uint8 stores (each becomes its own fabric write).  launch_fused picks
// accordingly; TSVPP_NT overrides.
__device__ __forceinline__ void st4(float *p, float a, float b, float c, float e, int nt) {
    vf4 v = { a, b, c, e };
    if (nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (nt == 1) __builtin_nontemporal_store(v, (vf4 *)p);
    else *(vf4 *)p = v;
}
// same with a uniform base pointer and a 32-bit byte offset per lane (SGPR base + VGPR offset)
__device__ __forceinline__ void st4o(uint8_t *base, uint32_t off, float a, float b, float c, float e, int nt) {
    vf4 v = { a, b, c, e };
    if (nt == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if (nt == 1) __builtin_nontemporal_store(v, (vf4 *)(base + off));
    else *(vf4 *)(base + off) = v;
}
// 8- and 16-byte stores of packed uint8 rows with the non-temporal hint, as inline asm: written as `if (nt) __builtin_nontemporal_store(..) else plain store` the two
// branches hold the same store and the compiler merges them into ONE PLAIN store -- the hint was silently dropped from every uint8 flavour of the streaming kernels
// until round 5 (found with a -D build that had no else branch: merged uint8 outputs +5..27 %, profiles/r05_prn_nt_variants.txt).  `base` is wave-uniform.
typedef uint32_t nt_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
// (`pol`: the launch's store policy, non-zero here -- 2 = `sc1`, anything else non-temporal)
__device__ __forceinline__ void st8_nt(uint8_t *base, uint32_t off, uint32_t lo, uint32_t hi, int pol = 1) {
    const nt_u32x2 v = { lo, hi };
    if (pol == 2) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void st16_nt(uint8_t *base, uint32_t off, nt_u32x4 v, int pol = 1) {
    if (pol == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
}
// 4-byte uint8 stores are always plain: `sc1` turned each into its own fabric write (-36 %), non-temporal bought nothing
// (re-measured in round 5 with the hint as inline asm, profiles/r05_st1_nt_variants.txt: the colour-only kernel's uint8 planar output +4 % (0.738 -> 0.766), AREA 1080p ->
// 960x544 +3 %, every other kernel -1..-5 %: bit 2 of the store policy, set by launch_fused for the colour-only kernel alone)
__device__ __forceinline__ void st1o(uint8_t *base, uint32_t off, uint32_t v, int nt) {
    if (nt & 4) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
    else *(uint32_t *)(base + off) = v;
}
// Four integer-valued floats -> packed bytes with v_cvt_pk_u8_f32 (one instruction per byte; it saturates to
// [0, 255], so the uint8 paths need no separate clamp after the truncation; it rounds to nearest, so the
// truncation itself stays -- measured: without v_trunc the parity tests fail)
__device__ __forceinline__ uint32_t pack_u8x4(float a, float b, float c, float e) {
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(a, 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(b, 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(c, 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(e, 3u, r);
}
// Twelve floats (one output row of a thread tile: 4 pixels x 3 channels) -> three dwords of truncated, saturated bytes.
// v_cvt_pk_u8_f32 follows the wave's fp32 rounding mode (probed on gfx950, tools/dbg_cvt.hip: under round-toward-zero 254.9
// -> 254, 2.5 -> 2, 300 -> 255, -3 -> 0; the switch is effective for the very next VALU instruction in both directions), so
// with MODE.fp_round = toward zero it IS the reference's (uint8) cast of the clamped value and the twelve v_trunc_f32
// disappear.  The two s_setreg bracket the conversions inside ONE asm statement: no float instruction of the surrounding code
// can be scheduled between them.
__device__ __forceinline__ void pack_u8x12_rtz(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3, float c0, float c1,
                                               float c2, float c3, uint32_t &pa, uint32_t &pb, uint32_t &pc) {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(pa), "=&v"(pb), "=&v"(pc)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
}
__device__ __forceinline__ f2 trunc2(f2 v) { return (f2){ __builtin_truncf(v.x), __builtin_truncf(v.y) }; }

// Merged fp32 outputs (RGB / HSV triples): a thread owns 48 contiguous bytes of an output row, so its three
// 16-byte stores would interleave with its neighbours' at a 48-byte stride -- every store instruction
// touches every 128-byte line of the row segment without completing one.  Instead the A active lanes of a
// "run" (the lanes of one wave that share an output row: min(tx, 64)) swap data through a per-wave LDS slab
// and store instruction k writes bytes [16 A k, 16 A (k + 1)) of the run's 48 A contiguous bytes.  Only
// lanes of one wave exchange data and LDS operations of a wave execute in order: no barrier.
struct MergedRun {
    uint8_t *lds; // the run's slab
    int m, a;     // lane index within the run, active lanes of the run (right edge: fewer than the run length)
};
template <int OUT, bool VEC> __device__ __forceinline__ MergedRun merged_run(const LaunchDesc &d, int j0) {
    MergedRun r{ nullptr, 0, 1 };
    if constexpr (VEC && (OUT == O_F32_MERGED || OUT == O_HSV_F32)) {
        __shared__ __attribute__((aligned(16))) uint8_t slab[MAX_THREADS * 48];
        const int len = min(d.tx, 64);
        r.m = threadIdx.x & (len - 1);
        r.a = min(len, (d.dst_w - (j0 - PXW * r.m)) / PXW);
        r.lds = slab + ((int)threadIdx.x - r.m) * 48;
    }
    // uint8 triples: a thread owns 12 contiguous bytes of the row; the same exchange turns the run's three 4-byte stores at a
    // 12-byte stride into ONE 16-byte store by three quarters of its lanes (runs whose length is a multiple of 4 lanes)
    if constexpr (VEC && OUT == O_U8_MERGED) {
        if (d.u8_xchg) {
            __shared__ __attribute__((aligned(16))) uint8_t slab8[MAX_THREADS * 12];
            const int len = min(d.tx, 64);
            r.m = threadIdx.x & (len - 1);
            r.a = min(len, (d.dst_w - (j0 - PXW * r.m)) / PXW);
            r.lds = slab8 + ((int)threadIdx.x - r.m) * 12;
        }
    }
    return r;
}

// Four pixels of one row -> three dwords of truncated, saturated bytes: PLANAR (channel 0 x 4, channel 1 x 4, channel 2 x 4) or
// MERGED (the row's 12 contiguous bytes).  The arithmetic of color_store_row's uint8 vector branch, without the stores.
template <bool PLANAR>
__device__ __forceinline__ void color_pack_row_u8(const float *Yf, const float *t0, const float *tg, const float *t2, const tsvpp_coeffs &k,
                                                  uint32_t &pa, uint32_t &pb, uint32_t &pc) {
    f2 c0[2], c1[2], c2[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        f2 y = { Yf[2 * p], Yf[2 * p + 1] };
        y = y - (f2){ k.y_offset, k.y_offset };
        y.x = __builtin_fmaxf(0.0f, y.x);
        y.y = __builtin_fmaxf(0.0f, y.y);
        y = y * (f2){ k.y_scale, k.y_scale };
        c0[p] = y + (f2){ t0[p], t0[p] };
        c1[p] = y + (f2){ tg[p], tg[p] };
        c2[p] = y + (f2){ t2[p], t2[p] };
    }
    if constexpr (PLANAR) pack_u8x12_rtz(c0[0].x, c0[0].y, c0[1].x, c0[1].y, c1[0].x, c1[0].y, c1[1].x, c1[1].y, c2[0].x, c2[0].y, c2[1].x, c2[1].y, pa, pb, pc);
    else pack_u8x12_rtz(c0[0].x, c1[0].x, c2[0].x, c0[0].y, c1[0].y, c2[0].y, c0[1].x, c1[1].x, c2[1].x, c0[1].y, c1[1].y, c2[1].y, pa, pb, pc);
}

// One output row of this thread: 4 pixels -> 3 channels, converted and stored.
template <int OUT, bool VEC>
__device__ __forceinline__ void color_store_row(const float Yf[PXW], const float t0[2], const float tg[2], const float t2[2],
                                                const tsvpp_coeffs &k, typename OutT<OUT>::type *out, uint32_t pix, uint32_t plane, int ncol, int nt,
                                                const MergedRun &run) {
    using T = typename OutT<OUT>::type;
    constexpr bool PLANAR = (OUT == O_U8_PLANAR || OUT == O_F32_PLANAR);
    f2 c0[2], c1[2], c2[2]; // channel values of pixel pairs (0,1) and (2,3)
#pragma unroll
    for (int p = 0; p < 2; p++) {
        f2 y = { Yf[2 * p], Yf[2 * p + 1] };
        y = y - (f2){ k.y_offset, k.y_offset };
        y.x = __builtin_fmaxf(0.0f, y.x);
        y.y = __builtin_fmaxf(0.0f, y.y);
        y = y * (f2){ k.y_scale, k.y_scale };
        if constexpr (sizeof(T) == 1 && VEC) { // truncation and clamp are the round-toward-zero saturating pack below
            c0[p] = y + (f2){ t0[p], t0[p] };
            c1[p] = y + (f2){ tg[p], tg[p] };
            c2[p] = y + (f2){ t2[p], t2[p] };
        } else if constexpr (sizeof(T) == 1) { // element-wise flavour: the saturating pack below clamps
            c0[p] = trunc2(y + (f2){ t0[p], t0[p] });
            c1[p] = trunc2(y + (f2){ tg[p], tg[p] });
            c2[p] = trunc2(y + (f2){ t2[p], t2[p] });
        } else {
            c0[p] = trunc_clamp255(y + (f2){ t0[p], t0[p] });
            c1[p] = trunc_clamp255(y + (f2){ tg[p], tg[p] });
            c2[p] = trunc_clamp255(y + (f2){ t2[p], t2[p] });
        }
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            c0[p] = norm255(c0[p]);
            c1[p] = norm255(c1[p]);
            c2[p] = norm255(c2[p]);
        }
        if constexpr (OUT == O_HSV_F32) { // (c0, c1, c2) = normalised (R, G, B) -> merged (H, S, V)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                float h[2], sa[2], v[2];
                hsv_pixel(c0[p].x, c1[p].x, c2[p].x, h[0], sa[0], v[0]);
                hsv_pixel(c0[p].y, c1[p].y, c2[p].y, h[1], sa[1], v[1]);
                c0[p] = (f2){ h[0], h[1] };
                c1[p] = (f2){ sa[0], sa[1] };
                c2[p] = (f2){ v[0], v[1] };
            }
        }
        float *o = (float *)out;
        if constexpr (VEC) {
            if constexpr (PLANAR) {
                // uniform plane bases (SGPR pairs) + one 32-bit byte offset per lane
                const uint32_t boff = pix * 4u;
                uint8_t *b0 = (uint8_t *)o, *b1 = (uint8_t *)(o + plane), *b2 = (uint8_t *)(o + 2 * (size_t)plane);
                st4o(b0, boff, c0[0].x, c0[0].y, c0[1].x, c0[1].y, nt);
                st4o(b1, boff, c1[0].x, c1[0].y, c1[1].x, c1[1].y, nt);
                st4o(b2, boff, c2[0].x, c2[0].y, c2[1].x, c2[1].y, nt);
            } else {
                // merged: the lanes of a run exchange their 48-byte pixel quads through LDS so that each
                // store instruction of the run writes one contiguous 16 * A byte span (see MergedRun)
                uint8_t *w = run.lds;
                *(vf4 *)(w + 48 * run.m) = (vf4){ c0[0].x, c1[0].x, c2[0].x, c0[0].y };
                *(vf4 *)(w + 48 * run.m + 16) = (vf4){ c1[0].y, c2[0].y, c0[1].x, c1[1].x };
                *(vf4 *)(w + 48 * run.m + 32) = (vf4){ c2[1].x, c0[1].y, c1[1].y, c2[1].y };
                __builtin_amdgcn_wave_barrier();
                const uint32_t q = (pix - 4u * (uint32_t)run.m) * 12u; // first byte of the run in this row
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    const uint32_t off = (uint32_t)(kk * run.a + run.m) * 16u;
                    const vf4 v = *(const vf4 *)(w + off);
                    st4o((uint8_t *)o, q + off, v.x, v.y, v.z, v.w, nt);
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            const float v0[4] = { c0[0].x, c0[0].y, c0[1].x, c0[1].y }, v1[4] = { c1[0].x, c1[0].y, c1[1].x, c1[1].y },
                        v2[4] = { c2[0].x, c2[0].y, c2[1].x, c2[1].y };
            for (int c = 0; c < ncol; c++) {
                if constexpr (PLANAR) {
                    o[pix + c] = v0[c];
                    o[plane + pix + c] = v1[c];
                    o[2 * plane + pix + c] = v2[c];
                } else {
                    o[3 * (pix + c)] = v0[c];
                    o[3 * (pix + c) + 1] = v1[c];
                    o[3 * (pix + c) + 2] = v2[c];
                }
            }
        }
    } else {
        uint8_t *o = (uint8_t *)out;
        if constexpr (VEC) {
            uint32_t pa, pb, pc;
            if constexpr (PLANAR) {
                pack_u8x12_rtz(c0[0].x, c0[0].y, c0[1].x, c0[1].y, c1[0].x, c1[0].y, c1[1].x, c1[1].y, c2[0].x, c2[0].y, c2[1].x, c2[1].y, pa, pb, pc);
                st1o(o, pix, pa, nt);
                st1o(o + plane, pix, pb, nt);
                st1o(o + 2 * (size_t)plane, pix, pc, nt);
            } else {
                pack_u8x12_rtz(c0[0].x, c1[0].x, c2[0].x, c0[0].y, c1[0].y, c2[0].y, c0[1].x, c1[1].x, c2[1].x, c0[1].y, c1[1].y, c2[1].y, pa, pb, pc);
                if (run.lds && (run.a & 3) == 0) {
                    uint32_t *w = (uint32_t *)(run.lds + 12 * run.m);
                    w[0] = pa;
                    w[1] = pb;
                    w[2] = pc;
                    __builtin_amdgcn_wave_barrier();
                    if (4 * run.m < 3 * run.a) { // 12 a bytes = 3 a / 4 chunks of 16
                        typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 v = *(const u32x4 *)(run.lds + 16 * run.m);
                        const uint32_t off16 = 3u * (pix - 4u * (uint32_t)run.m) + 16u * (uint32_t)run.m; // rows start on 4-byte boundaries only
                        if (nt) st16_nt(o, off16, (nt_u32x4){ v.x, v.y, v.z, v.w }, nt); // (round 5: the exchanged 16-byte stores cover whole lines like the fp32 ones)
                        else *(u32x4a4 *)(o + off16) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    st1o(o, 3u * pix, pa, nt);
                    st1o(o, 3u * pix + 4u, pb, nt);
                    st1o(o, 3u * pix + 8u, pc, nt);
                }
            }
        } else {
            const uint32_t p0 = pack_u8x4(c0[0].x, c0[0].y, c0[1].x, c0[1].y), p1 = pack_u8x4(c1[0].x, c1[0].y, c1[1].x, c1[1].y),
                           p2 = pack_u8x4(c2[0].x, c2[0].y, c2[1].x, c2[1].y);
            for (int c = 0; c < ncol; c++) {
                const uint8_t v0 = (uint8_t)(p0 >> (8 * c)), v1 = (uint8_t)(p1 >> (8 * c)), v2 = (uint8_t)(p2 >> (8 * c));
                if constexpr (PLANAR) {
                    o[pix + c] = v0;
                    o[plane + pix + c] = v1;
                    o[2 * plane + pix + c] = v2;
                } else {
                    o[3 * (pix + c)] = v0;
                    o[3 * (pix + c) + 1] = v1;
                    o[3 * (pix + c) + 2] = v2;
                }
            }
        }
    }
}

// Colour-convert + store a thread tile (2 rows x 4 columns) given its resized samples.  PLANE_MAJOR (fp32 planar outputs of the
// 2x2-tap kernels): both rows' colours first, then the six stores plane by plane instead of row by row -- the same arithmetic;
// measured +0.4..1.3 % on the headline, +0.3..0.6 % on 4K -> 1080p (profiles/r02_plane_major_ab.txt), so only where 24 live
// floats cost no occupancy.
template <int OUT, bool VEC, bool PLANE_MAJOR = false>
__device__ __forceinline__ void color_store_tile(const float Yf[PXH][PXW], const float Uf[2], const float Vf[2], const LaunchDesc &d,
                                                 typename OutT<OUT>::type *out, int i0, int j0, int ncol) {
    if constexpr (OUT == O_NV12_U8 || OUT == O_NV12_F32 || OUT == O_Y800_U8 || OUT == O_Y800_F32) {
        // no colour conversion: the resized samples themselves (fp32: / 255), Y plane then UV plane
        using T = typename OutT<OUT>::type;
        constexpr bool CHROMA = (OUT == O_NV12_U8 || OUT == O_NV12_F32);
        const int nt = d.nt_stores;
        const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;
        const uint32_t cpix = plane + (uint32_t)(i0 >> 1) * (uint32_t)d.dst_w + (uint32_t)j0;
        if constexpr (sizeof(T) == 1) {
            uint8_t *o = (uint8_t *)out;
            auto pk = [](float a, float b, float c, float e) { return pack_u8x4(a, b, c, e); };
#pragma unroll
            for (int r = 0; r < PXH; r++) {
                const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
                if constexpr (VEC) st1o(o, pix, pk(Yf[r][0], Yf[r][1], Yf[r][2], Yf[r][3]), nt);
                else
                    for (int c = 0; c < ncol; c++) o[pix + c] = (uint8_t)Yf[r][c];
            }
            if constexpr (CHROMA) {
                if constexpr (VEC) st1o(o, cpix, pk(Uf[0], Vf[0], Uf[1], Vf[1]), nt);
                else {
                    o[cpix] = (uint8_t)Uf[0];
                    o[cpix + 1] = (uint8_t)Vf[0];
                    if (ncol > 2) {
                        o[cpix + 2] = (uint8_t)Uf[1];
                        o[cpix + 3] = (uint8_t)Vf[1];
                    }
                }
            }
        } else {
            float *o = (float *)out;
#pragma unroll
            for (int r = 0; r < PXH; r++) {
                const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
                const f2 a = norm255((f2){ Yf[r][0], Yf[r][1] }), b = norm255((f2){ Yf[r][2], Yf[r][3] });
                if constexpr (VEC) st4o((uint8_t *)o, pix * 4u, a.x, a.y, b.x, b.y, nt);
                else {
                    const float v[4] = { a.x, a.y, b.x, b.y };
                    for (int c = 0; c < ncol; c++) o[pix + c] = v[c];
                }
            }
            if constexpr (CHROMA) {
                const f2 a = norm255((f2){ Uf[0], Vf[0] }), b = norm255((f2){ Uf[1], Vf[1] });
                if constexpr (VEC) st4o((uint8_t *)o, cpix * 4u, a.x, a.y, b.x, b.y, nt);
                else {
                    o[cpix] = a.x;
                    o[cpix + 1] = a.y;
                    if (ncol > 2) {
                        o[cpix + 2] = b.x;
                        o[cpix + 3] = b.y;
                    }
                }
            }
        }
        return;
    }
    float t0[2], tg[2], t2[2];
#pragma unroll
    for (int c = 0; c < 2; c++) chroma_terms(Uf[c], Vf[c], d.k, d.swap_rb, d.color_g, t0[c], tg[c], t2[c]);
    // 32-bit element offsets from the frame's (uniform) base pointer: the stores use the
    // SGPR-base + VGPR-offset addressing mode instead of per-lane 64-bit pointer arithmetic
    // (host side guarantees 3 * W * H * sizeof(T) < 4 GiB)
    const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;
    if constexpr (PLANE_MAJOR && OUT == O_F32_PLANAR && VEC) {
        f2 c[PXH][3][2];
#pragma unroll
        for (int r = 0; r < PXH; r++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                f2 y = { Yf[r][2 * p], Yf[r][2 * p + 1] };
                y = y - (f2){ d.k.y_offset, d.k.y_offset };
                y.x = __builtin_fmaxf(0.0f, y.x);
                y.y = __builtin_fmaxf(0.0f, y.y);
                y = y * (f2){ d.k.y_scale, d.k.y_scale };
                c[r][0][p] = norm255(trunc_clamp255(y + (f2){ t0[p], t0[p] }));
                c[r][1][p] = norm255(trunc_clamp255(y + (f2){ tg[p], tg[p] }));
                c[r][2][p] = norm255(trunc_clamp255(y + (f2){ t2[p], t2[p] }));
            }
        uint8_t *b = (uint8_t *)out;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int r = 0; r < PXH; r++) {
                const uint32_t boff = ((uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0) * 4u;
                st4o(b + (size_t)ch * plane * 4u, boff, c[r][ch][0].x, c[r][ch][0].y, c[r][ch][1].x, c[r][ch][1].y, d.nt_stores);
            }
        return;
    }
    const MergedRun run = merged_run<OUT, VEC>(d, j0);
#pragma unroll
    for (int r = 0; r < PXH; r++)
        color_store_row<OUT, VEC>(Yf[r], t0, tg, t2, d.k, out, (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0, plane, ncol, d.nt_stores, run);
}

// ----------------------------------------------------------------------------------------------
// Work decomposition shared by all kernels.
struct TileId {
    int frame, tx, ty;
    bool valid;
};
// tile_order 0 (default): workgroup id -> tile such that (a) the workgroups resident at any moment
// cover a contiguous run of tile rows of one or two frames (HBM write locality: measured +2..5 %
// over giving each XCD its own frames) and (b) a whole tile ROW lands on one XCD (id % 8), so the
// 128-byte lines that horizontally adjacent tiles share are fetched into one L2 once.
// tile_order 1: plain raster.  tile_order 2: XCD-contiguous runs of (frame, tile).
__device__ __forceinline__ TileId decode_tile(const LaunchDesc &d) {
    TileId t;
    const int tiles = d.tiles_x * d.tiles_y;
    if (d.tile_order == 0) {
        const int x = blockIdx.x % NUM_XCD, q = blockIdx.x / NUM_XCD;
        const int group = q / d.tiles_x;
        t.tx = q - group * d.tiles_x;
        const int row = group * NUM_XCD + x; // global tile row = frame * tiles_y + ty
        t.frame = row / d.tiles_y;
        t.ty = row - t.frame * d.tiles_y;
        t.valid = row < d.tiles_y * d.n_frames;
        return t;
    }
    if (d.tile_order >= 3) { // as 0, but an XCD takes G = 2 / 4 / 8 CONSECUTIVE tile rows (order 3 / 4 / 5): longer runs of output rows per XCD, vertical halos in one L2
        const int sh = d.tile_order - 2, G = 1 << sh;
        const int x = blockIdx.x % NUM_XCD, q = blockIdx.x / NUM_XCD;
        const int group = q / (G * d.tiles_x), within = q - group * G * d.tiles_x;
        t.tx = within >> sh;
        const int row = G * (group * NUM_XCD + x) + (within & (G - 1));
        t.frame = row / d.tiles_y;
        t.ty = row - t.frame * d.tiles_y;
        t.valid = row < d.tiles_y * d.n_frames;
        return t;
    }
    const int total = tiles * d.n_frames;
    const int logical = d.tile_order == 1 ? (int)blockIdx.x : (int)((blockIdx.x % NUM_XCD) * d.blocks_per_xcd + blockIdx.x / NUM_XCD);
    t.valid = logical < total;
    t.frame = logical / tiles;
    const int rem = logical - t.frame * tiles;
    t.ty = rem / d.tiles_x;
    t.tx = rem - t.ty * d.tiles_x;
    return t;
}

// Generic thread tile: samplers (any mode, any reader) -> colour back end.
template <int MODE, int OUT, bool VEC, class S>
__device__ __forceinline__ void convert_thread_tile(const S &s, const LaunchDesc &d, typename OutT<OUT>::type *out, int i0, int j0) {
    // VEC: the thread tile has its four columns.  Element-wise flavour: dst_w is even, so 2 or 4 columns (2 in the last
    // thread tile of a row when dst_w = 4 k + 2); a column that does not exist is sampled at the row's last one
    // instead -- in bounds, branch-free -- and never stored.
    const int ncol = VEC ? PXW : min(PXW, d.dst_w - j0);
    const int ci = i0 >> 1, cj0 = j0 >> 1, jmax = d.dst_w - 1, cjmax = (d.dst_w >> 1) - 1;
    float Uf[2], Vf[2], Yf[PXH][PXW];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        int U = 128, V = 128;
        if constexpr (!kLumaOnly<OUT>) sample_chroma<MODE>(s, d, ci, VEC ? cj0 + c : min(cj0 + c, cjmax), U, V);
        Uf[c] = (float)U;
        Vf[c] = (float)V;
    }
#pragma unroll
    for (int r = 0; r < PXH; r++)
#pragma unroll
        for (int c = 0; c < PXW; c++) Yf[r][c] = (float)sample_luma<MODE>(s, d, i0 + r, VEC ? j0 + c : min(j0 + c, jmax));
    color_store_tile<OUT, VEC>(Yf, Uf, Vf, d, out, i0, j0, ncol);
}

// dst_w = 4 k + 2 (854, 1366, ...): the last thread tile of every row has only two columns.  The vector-store kernels
// skip it (is_row_tail) and launch_fused covers those two columns with a second, tiny launch of the element-wise
// gather kernel (one thread tile per row pair; LaunchDesc::col0).  Both cheaper alternatives were measured and
// rejected: a fallback at store time keeps every sample live across its branch (gather / direct / colour-only
// kernels 4-10 % slower although it is almost never taken), and an early exit into an inlined generic path drags that
// path's registers into every kernel (C4's point kernel: 30 -> 110 VGPRs, 25 % slower).
__device__ __forceinline__ bool is_row_tail(const LaunchDesc &d, int j0) { return d.dst_w - j0 < PXW; }
// Round 4: that tail launch costs 10-50 us per step (profiles/r04_tail_probe.txt).  Wherever the output is at least one tile wide, launch_fused now
// shifts the launch's LAST TILE COLUMN to the left so that it ends at the frame's right edge (LaunchDesc::last_col0 = dst_w - tile width, 2 mod 4):
// every thread tile of that column then has its four columns, no row tail is left and no second launch is needed.  The columns the shifted tile shares
// with its left neighbour are computed and stored twice with the same values; its stores start 8 bytes (fp32) / 2 bytes (uint8) off the vector
// alignment, as every second row of such an output always did.  First column of tile `tx` of a launch whose tiles are `tile_w` columns wide:
__device__ __forceinline__ int tile_col0(const LaunchDesc &d, int tx, int tile_w) {
    const int j = tx * tile_w;
    return (d.last_col0 && j + tile_w > d.dst_w) ? d.last_col0 : j; // (wave-uniform)
}

// ----------------------------------------------------------------------------------------------
// Source footprint of a run of outputs [o0, o1] along one axis: first and last source sample any
// of them taps, from the SAME coordinate functions the samplers use (they are monotonic in o).
template <int MODE>
__device__ __forceinline__ void axis_span(int o0, int o1, float ratio, int limit, int taps, int &lo, int &hi) {
    if constexpr (MODE == M_NONE) {
        lo = o0;
        hi = o1;
    } else if constexpr (MODE == M_NEAREST) {
        lo = (int)(ratio * (float)o0);
        hi = (int)(ratio * (float)o1);
    } else if constexpr (MODE == M_AREA_DOWN) {
        lo = (int)(ratio * (float)o0);
        hi = (int)(ratio * (float)o1) + taps - 1;
    } else if constexpr (MODE == M_BILINEAR) {
        float w;
        bilinear_axis(o0, ratio, limit, lo, w);
        bilinear_axis(o1, ratio, limit, hi, w);
        hi += 1;
    } else if constexpr (MODE == M_AREA_UP) {
        float w;
        areaup_axis(o0, ratio, lo, w);
        areaup_axis(o1, ratio, hi, w);
        hi += 1;
    } else { // M_BICUBIC
        double w;
        bicubic_axis(o0, ratio, limit, lo, w);
        bicubic_axis(o1, ratio, limit, hi, w);
        lo -= 1;
        hi += 2;
    }
}

// Footprint of one workgroup tile in both planes (luma columns/rows; chroma PAIR columns/rows).
struct Footprint {
    int j_first, i_first, j_last, i_last;
    int xlo, xhi, ylo, yhi, cxlo, cxhi, cylo, cyhi;
};
template <int MODE>
__device__ __forceinline__ Footprint tile_footprint(const LaunchDesc &d, const TileId &id) {
    Footprint f;
    f.j_first = tile_col0(d, id.tx, d.tx * PXW);
    f.i_first = id.ty * d.ty * PXH * d.rpt;
    f.j_last = min(f.j_first + d.tx * PXW, d.dst_w) - 1;
    f.i_last = min(f.i_first + d.ty * PXH * d.rpt, d.dst_h) - 1;
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;
    axis_span<MODE>(f.j_first, f.j_last, d.xr, d.src_w, d.rx, f.xlo, f.xhi);
    axis_span<MODE>(f.i_first, f.i_last, d.yr, d.src_h, d.ry, f.ylo, f.yhi);
    f.xlo = max(f.xlo, 0);
    f.ylo = max(f.ylo, 0);
    f.xhi = min(f.xhi, d.src_w - 1);
    f.yhi = min(f.yhi, d.src_h - 1);
    // chroma: the same formulas on the chroma grid (dst_w, dst_h are even)
    axis_span<MODE>(f.j_first >> 1, f.j_last >> 1, d.xr, d.src_w, d.rx, f.cxlo, f.cxhi);
    axis_span<MODE>(f.i_first >> 1, f.i_last >> 1, d.yr, d.src_h, d.ry, f.cylo, f.cyhi);
    f.cxlo = max(f.cxlo, 0);
    f.cylo = max(f.cylo, 0);
    f.cxhi = min(f.cxhi, cw - 1);
    f.cyhi = min(f.cyhi, chh - 1);
    return f;
}

// Describe / stage rows [row_lo, row_lo + nrows) x byte columns [col_lo, col_lo + span) of one
// plane into LDS with 16-byte aligned global loads.  A 16-byte aligned chunk that overlaps at
// least one valid byte lies in the same page as that byte, so the over-read at the ends of a row
// never faults.  2^slot_shift lanes serve one row (lanes >= cpr idle): no integer division.
__device__ __forceinline__ LdsPlane describe_plane(uint8_t *lds, const uint8_t *plane, int pitch, int row_lo, int col_lo, int cpr,
                                                   const uint8_t *&a0) {
    a0 = plane + (size_t)row_lo * (size_t)pitch + (size_t)col_lo;
    LdsPlane lp;
    lp.base = lds;
    lp.x0 = col_lo;
    lp.y0 = row_lo;
    lp.lp = 16 * cpr;
    lp.m0 = (int)((uintptr_t)a0 & 15);
    lp.pm = pitch & 15;
    return lp;
}
// All loads of a round -- both planes -- are issued before the first LDS write (K rows per lane in
// flight): with a plain load->write loop the compiler has to wait for each chunk before issuing
// the next and the workgroup pays the HBM latency once per row group instead of once.  The loads
// are branch-free (clamped addresses; only the LDS write is predicated).
struct StageLane {
    int ch, r0, rstep;
};
__device__ __forceinline__ StageLane stage_lane(int slot_shift, int nthreads) {
    return StageLane{ (int)(threadIdx.x & ((1 << slot_shift) - 1)), (int)(threadIdx.x >> slot_shift), nthreads >> slot_shift };
}
template <int K>
__device__ __forceinline__ void stage_issue(const uint8_t *a0, const LdsPlane &lp, int pitch, int nrows, int span, int cpr, const StageLane &ln,
                                            int base, uint4 v[K], bool ok[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int r = base + k * ln.rstep + ln.r0;
        const int rc = min(r, nrows - 1);
        const int mis = (lp.m0 + rc * lp.pm) & 15;
        const int chmax = (mis + span - 1) >> 4;
        ok[k] = (ln.ch < cpr) && (r < nrows) && (ln.ch <= chmax);
        v[k] = make_uint4(0, 0, 0, 0);
        if (base + k * ln.rstep < nrows) // uniform
            v[k] = *(const uint4 *)(a0 + (size_t)rc * (size_t)pitch - mis + 16 * min(ln.ch, chmax));
    }
}
template <int K>
__device__ __forceinline__ void stage_commit(uint8_t *lds, const LdsPlane &lp, const StageLane &ln, int base, const uint4 v[K], const bool ok[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int r = base + k * ln.rstep + ln.r0;
        if (ok[k]) *(uint4 *)(lds + r * lp.lp + 16 * ln.ch) = v[k];
    }
}
template <int KY, int KUV>
__device__ __forceinline__ void stage_planes(const LaunchDesc &d, uint8_t *lds_y, const uint8_t *ay, const LdsPlane &py, int ny, int span_y,
                                             uint8_t *lds_uv, const uint8_t *auv, const LdsPlane &puv, int nuv, int span_uv, int nthreads) {
    const StageLane ly = stage_lane(d.lds_slot_y, nthreads), luv = stage_lane(d.lds_slot_uv, nthreads);
    for (int it = 0;; it++) {
        const int by = it * KY * ly.rstep, buv = it * KUV * luv.rstep;
        if (by >= ny && buv >= nuv) break;
        uint4 vy[KY], vuv[KUV];
        bool oky[KY], okuv[KUV];
        stage_issue<KY>(ay, py, d.pitch_y, ny, span_y, d.lds_cpr_y, ly, by, vy, oky);
        stage_issue<KUV>(auv, puv, d.pitch_uv, nuv, span_uv, d.lds_cpr_uv, luv, buv, vuv, okuv);
        stage_commit<KY>(lds_y, py, ly, by, vy, oky);
        stage_commit<KUV>(lds_uv, puv, luv, buv, vuv, okuv);
    }
}

// LDS-DMA variant (global_load_lds_dwordx4): the chunks go straight from HBM into LDS, no VGPR round trip and no
// ds_write.  Lane l of a wave instruction lands in LDS slot (wave base + 16 l), so one instruction fills 64 CONSECUTIVE
// 16-byte slots of the plane's chunk array; slot s belongs to row s / cpr, chunk s % cpr (cpr = chunks per row, any
// number: the division is a multiply-high by the host's `magic` = 2^32 / cpr + 1).  Round 1 padded cpr to a power of two
// so that the slot of (row, chunk) was the thread index: 19-35 % more LDS per tile and a power-of-two row pitch, i.e. the
// two half-waves of a wave (neighbouring rows) on the same banks -- 63 % of all LDS cycles of the uint8 2x2-tap kernel
// were bank-conflict cycles (profiles/r02_u8p_pmc.txt).  Idle lanes (slots past the footprint) fetch the last valid chunk
// into their padding slot: the plane is allocated up to a multiple of 64 slots.  The caller waits (vmcnt(0)) and barriers.
__device__ __forceinline__ void stage_plane_dma(uint8_t *lds, const uint8_t *a0, const LdsPlane &lp, int pitch, int nrows, int span,
                                                uint32_t magic, int nthreads) {
    const int cpr = lp.lp >> 4;
    const int total = nrows * cpr;                       // chunk slots of this tile's footprint (uniform)
    const int wave0 = (int)(threadIdx.x & ~63u);
    if (lp.pm == 0) {
        // pitch % 16 == 0 (every decoder output): all rows share one misalignment.  The slot -> (row, chunk) division is done
        // once per thread; every further round advances by nthreads slots = qs rows + rs chunks (both uniform) with a
        // conditional wrap: ~9 full-rate VALU instructions per round instead of ~20 with three quarter-rate multiplies.
        const uint32_t chmax = (uint32_t)(lp.m0 + span - 1) >> 4;
        const uint8_t *origin = a0 - lp.m0;
        const uint32_t qs = __umulhi((uint32_t)nthreads, magic), rs = (uint32_t)nthreads - qs * (uint32_t)cpr;
        const uint32_t row_step = qs * (uint32_t)pitch;
        const uint32_t off_last = (uint32_t)(nrows - 1) * (uint32_t)pitch + 16u * chmax; // idle lanes re-fetch the last valid chunk
        const uint32_t row0 = __umulhi(threadIdx.x, magic);
        uint32_t ch = threadIdx.x - row0 * (uint32_t)cpr, rowoff = row0 * (uint32_t)pitch;
        for (int base = 0; base + wave0 < total; base += nthreads) { // wave-uniform trip count
            const uint32_t off = min(rowoff + 16u * min(ch, chmax), off_last);
            const uint8_t *src = origin + (size_t)off;
            uint8_t *dst = lds + (base + wave0) * 16; // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            ch += rs;
            const bool wrap = ch >= (uint32_t)cpr;
            ch -= wrap ? (uint32_t)cpr : 0u;
            rowoff += row_step + (wrap ? (uint32_t)pitch : 0u);
        }
        return;
    }
    for (int base = 0; base + wave0 < total; base += nthreads) {
        const uint32_t sl = (uint32_t)min(base + (int)threadIdx.x, total - 1);
        const int rc = (int)__umulhi(sl, magic);
        const int ch = (int)sl - rc * cpr;
        const int mis = (lp.m0 + rc * lp.pm) & 15;
        const int chmax = (mis + span - 1) >> 4;
        const uint8_t *src = a0 + (size_t)rc * (size_t)pitch - mis + 16 * min(ch, chmax);
        uint8_t *dst = lds + (base + wave0) * 16; // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
}

// Integer box sum / divisor, truncated (dyadic AREA kernels; the derivation is at vpp_area_dyadic_kernel): with one
// divisor S for the whole frame (rcp = 1 / S, S < 4096) the quotient is floor((SUM + 0.5) * rcp), else one IEEE division.
__device__ __forceinline__ float area_quot(uint32_t sum, int sx, int sy, float rcp) {
    if (rcp != 0.0f) return __builtin_truncf(((float)sum + 0.5f) * rcp);
    return __builtin_truncf((float)sum / (float)(sx * sy));
}

// coordinate-table entries of the 2x2-tap kernel (vpp_bilinear.hip); launch_fused sizes the LDS tables with them
struct XEntry { int off; float w; };                  // LDS byte offset from the row base, weight
struct YEntry { int top, bot; float w; int pad; };    // LDS row bases of y and y2, weight

extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];

} // namespace tsvpp
