// vpp_formats.hip -- the reference's other output formats, from (already cropped / resized) NV12:
//   Y800      reference src/ColorConversion.cu:95-105
//   NV12      (planes packed back to back)  :211-233
//   UYVY      4:2:0 -> 4:2:2, vertical (-1, 9, 9, -1)/16 chroma filter on odd chroma rows  :107-127, 177-209
//   YUV444    UYVY -> planar 4:4:4, horizontal (-1, 9, 9, -1)/16 filter on odd pixels      :129-173
//   HSV       normalised RGB -> HSV                                                           :235-278, 357-370
// These are the "next" rows of SURVEY.md 8(f): one straightforward kernel each (thread = one
// horizontal pixel pair), not tuned.  Same arithmetic contract as vpp_kernels.hip: plain IEEE,
// no contraction; integer paths use C integer semantics exactly as the reference's <uchar> code.
#include "vpp_kernels.h"

#pragma clang fp contract(off)

namespace tsvpp {

struct Nv12View {
    const uint8_t *y, *uv;
    int py, puv, w, h;
};

// Vertical chroma filter of the 4:2:0 -> 4:2:2 step.  `col` is a BYTE column of the UV plane.
__device__ __forceinline__ int chroma_422(const Nv12View &s, int i, int col) {
    const int row = i >> 1, last = (s.h >> 1) - 1;
    int v = s.uv[(size_t)row * s.puv + col];
    if (row & 1) {
        const int r2 = min(row + 1, last), r3 = max(row - 1, 0), r4 = min(row + 2, last);
        const int a = v + s.uv[(size_t)r2 * s.puv + col];
        const int b = s.uv[(size_t)r3 * s.puv + col] + s.uv[(size_t)r4 * s.puv + col];
        v = min(max((9 * a - b + 8) >> 4, 0), 255);
    }
    return v;
}

template <class T> __device__ __forceinline__ T fin(int v, bool norm);
template <> __device__ __forceinline__ uint8_t fin<uint8_t>(int v, bool norm) { return norm ? (uint8_t)((uint8_t)v / 255) : (uint8_t)v; }
template <> __device__ __forceinline__ float fin<float>(int v, bool norm) { return norm ? (float)v / 255.0f : (float)v; }

template <class T>
__global__ __launch_bounds__(256) void fmt_y800(Nv12View s, T *out, bool norm) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= s.w) return;
    out[(size_t)i * s.w + j] = fin<T>(s.y[(size_t)i * s.py + j], norm);
}

template <class T>
__global__ __launch_bounds__(256) void fmt_nv12(Nv12View s, T *out, bool norm) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= s.w) return;
    out[(size_t)i * s.w + j] = fin<T>(s.y[(size_t)i * s.py + j], norm);
    if ((i & 1) == 0) out[(size_t)s.w * s.h + (size_t)(i >> 1) * s.w + j] = fin<T>(s.uv[(size_t)(i >> 1) * s.puv + j], norm);
}

// thread = pixel pair (2q, 2q+1) in flat order (w is even: a pair never straddles rows)
template <class T>
__global__ __launch_bounds__(256) void fmt_uyvy(Nv12View s, T *out, bool norm) {
    const int jp = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (2 * jp >= s.w) return;
    const int j = 2 * jp;
    T *o = out + ((size_t)i * s.w + j) * 2;
    o[0] = fin<T>(chroma_422(s, i, j), norm);
    o[1] = fin<T>(s.y[(size_t)i * s.py + j], norm);
    o[2] = fin<T>(chroma_422(s, i, j + 1), norm);
    o[3] = fin<T>(s.y[(size_t)i * s.py + j + 1], norm);
}

// U (comp 0) / V (comp 1) of flat pixel pair q of the intermediate UYVY image; pairs past the end
// read as 0 (the reference reads past its buffer there, src/ColorConversion.cu:131-138).
__device__ __forceinline__ int uyvy_chroma(const Nv12View &s, long q, int comp) {
    const long npairs = (long)s.w * s.h / 2;
    if (q < 0 || q >= npairs) return 0;
    const long idx = 2 * q;
    const int i = (int)(idx / s.w), j = (int)(idx - (long)i * s.w);
    return chroma_422(s, i, j + comp);
}

template <class T>
__global__ __launch_bounds__(256) void fmt_yuv444(Nv12View s, T *out, bool norm) {
    const int jp = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (2 * jp >= s.w) return;
    const size_t wh = (size_t)s.w * s.h;
    const long idx0 = (long)i * s.w + 2 * jp, q = idx0 >> 1, npairs = (long)(wh / 2);
    // even pixel: its own pair's chroma
    const int u0 = uyvy_chroma(s, q, 0), v0 = uyvy_chroma(s, q, 1);
    out[idx0] = fin<T>(s.y[(size_t)i * s.py + 2 * jp], norm);
    out[wh + idx0] = fin<T>(u0, norm);
    out[2 * wh + idx0] = fin<T>(v0, norm);
    // odd pixel: (9 (p1 + p2) - (p3 + p4) + 8) / 16 over the neighbouring pairs in FLAT order
    const long idx1 = idx0 + 1, src = 2 * idx1 + 1;
    out[idx1] = fin<T>(s.y[(size_t)i * s.py + 2 * jp + 1], norm);
#pragma unroll
    for (int comp = 0; comp < 2; comp++) {
        const int shift = 2 * comp;
        const int p1 = comp ? v0 : u0;
        const int p2 = uyvy_chroma(s, q + 1, comp);
        const int p3 = (src - 7 + shift < 0) ? p1 : uyvy_chroma(s, q - 1, comp);
        const int p4 = (src + 5 + shift > (long)(2 * wh) - 1) ? p2 : uyvy_chroma(s, q + 2, comp);
        (void)npairs;
        if constexpr (sizeof(T) == 1) {
            uint8_t v = (uint8_t)((9 * (p1 + p2) - (p3 + p4) + 8) / 16); // C division, then wrap to uchar
            if (norm) v = (uint8_t)(v / 255);
            out[(1 + comp) * wh + idx1] = (T)v;
        } else {
            float a = (float)p1 + (float)p2;
            a = 9.0f * a;
            const float b = (float)p3 + (float)p4;
            float v = a - b;
            v = v + 8.0f;
            v = v / 16.0f;
            v = fminf(v, 255.0f);
            v = fmaxf(v, 0.0f);
            if (norm) v = v / 255.0f;
            out[(1 + comp) * wh + idx1] = (T)v;
        }
    }
}

__device__ __forceinline__ int clamp_byte(int v) { return max(min(v, 255), 0); }

// NV12 -> normalised RGB (src/ColorConversion.cu:6-39, 68-93 with normalization) -> HSV (:235-278)
__global__ __launch_bounds__(256) void fmt_hsv(Nv12View s, float *out, tsvpp_coeffs k) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= s.w) return;
    const int Y = s.y[(size_t)i * s.py + j];
    const int U = s.uv[(size_t)(i >> 1) * s.puv + (j & ~1)], V = s.uv[(size_t)(i >> 1) * s.puv + (j & ~1) + 1];
    const float yv = fmaxf(0.0f, (float)Y - k.y_offset) * k.y_scale;
    const float fu = (float)U - k.c_offset, fv = (float)V - k.c_offset;
    float rv = k.v_to_r * fv;
    rv = rv + k.round_bias;
    float bv = k.u_to_b * fu;
    bv = bv + k.round_bias;
    const float g1 = k.v_to_g * fv, g2 = k.u_to_g * fu;
    float gv = g1 + g2;
    gv = gv + k.round_bias;
    const float R = (float)clamp_byte((int)(yv + rv)) / 255.0f;
    const float G = (float)clamp_byte((int)(yv + gv)) / 255.0f;
    const float B = (float)clamp_byte((int)(yv + bv)) / 255.0f;
    const float mn = fminf(fminf(R, G), B), mx = fmaxf(fmaxf(R, G), B);
    const float delta = mx - mn;
    float *o = out + ((size_t)i * s.w + j) * 3;
    o[2] = mx;
    float S = 0.0f;
    if (mx != 0.0f) {
        const float q = mn / mx;
        S = 1.0f - q;
    }
    o[1] = S;
    if (mx == mn) {
        o[0] = 0.0f;
        return;
    }
    float H = 0.0f;
    if (R == mx && G >= B) {
        H = 60.0f * (G - B);
        H = H / delta;
    } else if (R == mx && G < B) {
        H = 60.0f * (G - B);
        H = H / delta;
        H = H + 360.0f;
    } else if (G == mx) {
        H = 60.0f * (B - R);
        H = H / delta;
        H = H + 120.0f;
    } else if (B == mx) {
        H = 60.0f * (R - G);
        H = H / delta;
        H = H + 240.0f;
    }
    if (H < 0.0f) H = H + 360.0f;
    H = H / 360.0f;
    o[0] = H;
}

hipError_t launch_format(int fourcc, bool f32, bool norm, const uint8_t *y, const uint8_t *uv, int py, int puv, int w, int h, void *out,
                         const tsvpp_coeffs &k, hipStream_t stream) {
    Nv12View s{ y, uv, py, puv, w, h };
    const dim3 block(256), gpix((w + 255) / 256, h), gpair((w / 2 + 255) / 256, h);
    switch (fourcc) {
    case TSVPP_Y800:
        if (f32) hipLaunchKernelGGL(fmt_y800<float>, gpix, block, 0, stream, s, (float *)out, norm);
        else hipLaunchKernelGGL(fmt_y800<uint8_t>, gpix, block, 0, stream, s, (uint8_t *)out, norm);
        break;
    case TSVPP_NV12:
        if (f32) hipLaunchKernelGGL(fmt_nv12<float>, gpix, block, 0, stream, s, (float *)out, norm);
        else hipLaunchKernelGGL(fmt_nv12<uint8_t>, gpix, block, 0, stream, s, (uint8_t *)out, norm);
        break;
    case TSVPP_UYVY:
        if (f32) hipLaunchKernelGGL(fmt_uyvy<float>, gpair, block, 0, stream, s, (float *)out, norm);
        else hipLaunchKernelGGL(fmt_uyvy<uint8_t>, gpair, block, 0, stream, s, (uint8_t *)out, norm);
        break;
    case TSVPP_YUV444:
        if (f32) hipLaunchKernelGGL(fmt_yuv444<float>, gpair, block, 0, stream, s, (float *)out, norm);
        else hipLaunchKernelGGL(fmt_yuv444<uint8_t>, gpair, block, 0, stream, s, (uint8_t *)out, norm);
        break;
    case TSVPP_HSV:
        hipLaunchKernelGGL(fmt_hsv, gpix, block, 0, stream, s, (float *)out, k);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
