// vpp_formats.hip -- the two output formats whose chroma filters reach across rows / pixel pairs and
// therefore run as a second pass over the (already cropped / resized) NV12:
//   UYVY      4:2:0 -> 4:2:2, vertical (-1, 9, 9, -1)/16 chroma filter on odd chroma rows  reference src/ColorConversion.cu:107-127, 177-209
//   YUV444    UYVY -> planar 4:4:4, horizontal (-1, 9, 9, -1)/16 filter on odd pixels      :129-173
// (Y800, NV12 and HSV are output flavours of the fused kernels, vpp_kernels.hip.)  One launch per batch
// of <= 64 frames (pointer table in the kernarg segment), thread = two horizontal pixel pairs.  Same arithmetic contract as vpp_kernels.hip: plain IEEE,
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

// uint8 outputs are the raw values, fp32 outputs are normalised (the C ABI ties fp32 to normalization)
// The same for the four bytes (U0 V0 U1 V1) at byte columns col .. col + 3 of the UV plane, one load per
// tap row; A4: the plane base, the pitch and col are multiples of 4 (dword loads).
template <bool A4> __device__ __forceinline__ uint32_t ld4(const uint8_t *p) {
    if constexpr (A4) return *(const uint32_t *)p;
    else return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
template <bool A4> __device__ __forceinline__ void chroma_422x4(const Nv12View &s, int i, int col, int v[4]) {
    const int row = i >> 1, last = (s.h >> 1) - 1;
    const uint32_t a = ld4<A4>(s.uv + (size_t)row * s.puv + col);
    if (row & 1) { // uniform per wave: a wave serves one output row
        const uint32_t b = ld4<A4>(s.uv + (size_t)min(row + 1, last) * s.puv + col);
        const uint32_t c = ld4<A4>(s.uv + (size_t)max(row - 1, 0) * s.puv + col);
        const uint32_t e = ld4<A4>(s.uv + (size_t)min(row + 2, last) * s.puv + col);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int in = (int)((a >> (8 * k)) & 255) + (int)((b >> (8 * k)) & 255);
            const int out = (int)((c >> (8 * k)) & 255) + (int)((e >> (8 * k)) & 255);
            v[k] = min(max((9 * in - out + 8) >> 4, 0), 255);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (int)((a >> (8 * k)) & 255);
    }
}

template <class T> __device__ __forceinline__ T fin(int v);
template <> __device__ __forceinline__ uint8_t fin<uint8_t>(int v) { return (uint8_t)v; }
// x / 255, correctly rounded, for x = k/16 in [0, 255] (all 4081 values checked against exact rational arithmetic in
// tests/test_oracle_properties.py): 1/255 = hi + lo, e = x * lo, q = fma(x, hi, e) -- two VALU operations instead of
// the IEEE division sequence.  Every fp32 value these kernels normalise is an integer or a multiple of 1/16.
__device__ __forceinline__ float div255(float x) {
    const float hi = 0x1.010102p-8f, lo = -0x1.fdfdfep-33f;
    const float e = x * lo;
    return __builtin_fmaf(x, hi, e);
}
template <> __device__ __forceinline__ float fin<float>(int v) { return div255((float)v); }

struct FmtGeom {
    int py, puv, w, h;
    int aligned4; // every plane pointer and both pitches are multiples of 4
};
constexpr int FMT_BX = 64, FMT_BY = 4; // workgroup = 64 x 4 threads, thread = PAIRS horizontal pixel pairs

// UYVY: thread = PAIRS pixel pairs of row i (w is even: a pair never straddles rows); PAIRS == 2 needs
// w % 4 == 0 and 16-byte aligned outputs (one 8-byte / two 16-byte stores per thread)
template <class T, int PAIRS>
__global__ __launch_bounds__(FMT_BX *FMT_BY) void fmt_uyvy(const FrameTable t, const FmtGeom g) {
    const int f = blockIdx.z, i = blockIdx.y * FMT_BY + threadIdx.y, j = (blockIdx.x * FMT_BX + threadIdx.x) * 2 * PAIRS;
    if (i >= g.h || j >= g.w) return;
    const Nv12View s{ t.y[f], t.uv[f], g.py, g.puv, g.w, g.h };
    int v[4 * PAIRS];
    if constexpr (PAIRS == 2) { // j % 4 == 0: one load per luma / chroma row
        int c[4];
        uint32_t yw;
        if (g.aligned4) {
            chroma_422x4<true>(s, i, j, c);
            yw = ld4<true>(s.y + (size_t)i * s.py + j);
        } else {
            chroma_422x4<false>(s, i, j, c);
            yw = ld4<false>(s.y + (size_t)i * s.py + j);
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            v[4 * p] = c[2 * p];
            v[4 * p + 1] = (int)((yw >> (16 * p)) & 255);
            v[4 * p + 2] = c[2 * p + 1];
            v[4 * p + 3] = (int)((yw >> (16 * p + 8)) & 255);
        }
    } else {
#pragma unroll
        for (int p = 0; p < PAIRS; p++) {
            v[4 * p] = chroma_422(s, i, j + 2 * p);
            v[4 * p + 1] = s.y[(size_t)i * s.py + j + 2 * p];
            v[4 * p + 2] = chroma_422(s, i, j + 2 * p + 1);
            v[4 * p + 3] = s.y[(size_t)i * s.py + j + 2 * p + 1];
        }
    }
    T *o = (T *)t.out[f] + ((size_t)i * s.w + j) * 2;
    if constexpr (PAIRS == 2 && sizeof(T) == 1) { // one 8-byte store per lane: a wave instruction covers 512 contiguous bytes
        const uint64_t lo = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        const uint64_t hi = (uint32_t)v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
        __builtin_nontemporal_store(lo | (hi << 32), (uint64_t *)o);
    } else if constexpr (PAIRS == 2) {
        // 32 bytes per lane in two 16-byte stores: each instruction writes half of every line, so plain stores
        // (L2 combines the halves; non-temporal ones would go out as partial lines, cf. MergedRun in vpp_kernels.hip)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const vf4 a = { fin<float>(v[0]), fin<float>(v[1]), fin<float>(v[2]), fin<float>(v[3]) };
        const vf4 b = { fin<float>(v[4]), fin<float>(v[5]), fin<float>(v[6]), fin<float>(v[7]) };
        *(vf4 *)o = a;
        *((vf4 *)o + 1) = b;
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] = fin<T>(v[c]);
    }
}

// U (comp 0) / V (comp 1) of the pixel pair at (row i, even column j) of the intermediate UYVY image,
// where j may run off either end of the row: the reference indexes that image FLAT, so the pair before
// the first of a row is the last of the previous row; pairs before the image or past its end read as 0
// (the reference reads outside its buffer there, src/ColorConversion.cu:131-138).
__device__ __forceinline__ int uyvy_chroma(const Nv12View &s, int i, int j, int comp) {
    while (j < 0) {
        j += s.w;
        i--;
    }
    while (j >= s.w) {
        j -= s.w;
        i++;
    }
    if (i < 0 || i >= s.h) return 0;
    return chroma_422(s, i, j + comp);
}

template <class T> __device__ __forceinline__ T yuv444_odd(int p1, int p2, int p3, int p4) {
    if constexpr (sizeof(T) == 1) {
        return (T)(uint8_t)((9 * (p1 + p2) - (p3 + p4) + 8) / 16); // C division, then wrap to uchar
    } else {
        float a = (float)p1 + (float)p2;
        a = 9.0f * a;
        const float b = (float)p3 + (float)p4;
        float v = a - b;
        v = v + 8.0f;
        v = v / 16.0f;
        v = fminf(v, 255.0f);
        v = fmaxf(v, 0.0f);
        return (T)div255(v);
    }
}

// YUV444 planar: even pixels take their pair's chroma, odd pixels (9 (p1 + p2) - (p3 + p4) + 8) / 16 over the
// neighbouring pairs in FLAT order.  Thread = PAIRS pairs; the chroma of pairs -1 .. PAIRS + 1 is computed once.
template <class T, int PAIRS>
__global__ __launch_bounds__(FMT_BX *FMT_BY) void fmt_yuv444(const FrameTable t, const FmtGeom g) {
    const int f = blockIdx.z, i = blockIdx.y * FMT_BY + threadIdx.y, j = (blockIdx.x * FMT_BX + threadIdx.x) * 2 * PAIRS;
    if (i >= g.h || j >= g.w) return;
    const Nv12View s{ t.y[f], t.uv[f], g.py, g.puv, g.w, g.h };
    const size_t wh = (size_t)s.w * s.h;
    int c[PAIRS + 3][2]; // chroma of pairs -1 .. PAIRS + 1 relative to this thread's first
    if constexpr (PAIRS == 2) {
        // bytes j - 2 .. j + 7 of the chroma row(s): three loads per tap row (columns clamped into the row); the
        // pairs that fall off either end of the row are then patched from the neighbouring rows (edge lanes only)
        int l[4], m[4], r[4];
        const int jl = max(j - 4, 0), jr = min(j + 4, s.w - 4);
        if (g.aligned4) {
            chroma_422x4<true>(s, i, jl, l);
            chroma_422x4<true>(s, i, j, m);
            chroma_422x4<true>(s, i, jr, r);
        } else {
            chroma_422x4<false>(s, i, jl, l);
            chroma_422x4<false>(s, i, j, m);
            chroma_422x4<false>(s, i, jr, r);
        }
        c[0][0] = l[2], c[0][1] = l[3];
        c[1][0] = m[0], c[1][1] = m[1];
        c[2][0] = m[2], c[2][1] = m[3];
        c[3][0] = r[0], c[3][1] = r[1];
        c[4][0] = r[2], c[4][1] = r[3];
        if (j == 0) {
            c[0][0] = uyvy_chroma(s, i, -2, 0);
            c[0][1] = uyvy_chroma(s, i, -2, 1);
        }
        if (j + 4 >= s.w) {
#pragma unroll
            for (int k = 3; k < 5; k++) {
                c[k][0] = uyvy_chroma(s, i, j + 2 * (k - 1), 0);
                c[k][1] = uyvy_chroma(s, i, j + 2 * (k - 1), 1);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PAIRS + 3; k++) {
            c[k][0] = uyvy_chroma(s, i, j + 2 * (k - 1), 0);
            c[k][1] = uyvy_chroma(s, i, j + 2 * (k - 1), 1);
        }
    }
    T yv[2 * PAIRS], uo[2 * PAIRS], vo[2 * PAIRS];
    if constexpr (PAIRS == 2) {
        const uint32_t yw = g.aligned4 ? ld4<true>(s.y + (size_t)i * s.py + j) : ld4<false>(s.y + (size_t)i * s.py + j);
#pragma unroll
        for (int q = 0; q < 4; q++) yv[q] = fin<T>((int)((yw >> (8 * q)) & 255));
    } else {
        yv[0] = fin<T>(s.y[(size_t)i * s.py + j]);
        yv[1] = fin<T>(s.y[(size_t)i * s.py + j + 1]);
    }
    // The reference tests its flat source index against the ends of the UYVY buffer (src - 7 + shift < 0 and
    // src + 5 + shift > 2 w h - 1 with src = 2 idx + 1, shift = 0 / 2 for U / V, src/ColorConversion.cu:150-160):
    // for both components that is "the frame's first odd pixel" and "its last two odd pixels".  w h < 2^31 (the ABI
    // bounds the output at 4 GiB), so the flat index fits 32 bits.
    const uint32_t wh32 = (uint32_t)s.w * (uint32_t)s.h, row0 = (uint32_t)i * (uint32_t)s.w + (uint32_t)j;
#pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        const uint32_t idx1 = row0 + 2 * p + 1;
        const bool first = idx1 == 1u, last2 = idx1 + 3u >= wh32;
        uo[2 * p] = fin<T>(c[p + 1][0]);
        vo[2 * p] = fin<T>(c[p + 1][1]);
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            const int p1 = c[p + 1][comp], p2 = c[p + 2][comp];
            const int p3 = first ? p1 : c[p][comp];
            const int p4 = last2 ? p2 : c[p + 3][comp];
            (comp ? vo : uo)[2 * p + 1] = yuv444_odd<T>(p1, p2, p3, p4);
        }
    }
    T *o = (T *)t.out[f] + (size_t)i * s.w + j;
    if constexpr (PAIRS == 2 && sizeof(T) == 1) {
        auto pk = [](const T *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
        __builtin_nontemporal_store(pk(yv), (uint32_t *)o);
        __builtin_nontemporal_store(pk(uo), (uint32_t *)(o + wh));
        __builtin_nontemporal_store(pk(vo), (uint32_t *)(o + 2 * wh));
    } else if constexpr (PAIRS == 2) {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((vf4){ yv[0], yv[1], yv[2], yv[3] }, (vf4 *)o);
        __builtin_nontemporal_store((vf4){ uo[0], uo[1], uo[2], uo[3] }, (vf4 *)(o + wh));
        __builtin_nontemporal_store((vf4){ vo[0], vo[1], vo[2], vo[3] }, (vf4 *)(o + 2 * wh));
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            o[q] = yv[q];
            o[wh + q] = uo[q];
            o[2 * wh + q] = vo[q];
        }
    }
}

hipError_t launch_format(int fourcc, bool f32, const FrameTable &t, int n, int py, int puv, int w, int h, hipStream_t stream) {
    bool wide = (w % 4) == 0, a4 = (py % 4) == 0 && (puv % 4) == 0;
    for (int f = 0; f < n && wide; f++) wide = ((uintptr_t)t.out[f] & 15) == 0;
    for (int f = 0; f < n && a4; f++) a4 = (((uintptr_t)t.y[f] | (uintptr_t)t.uv[f]) & 3) == 0;
    const FmtGeom g{ py, puv, w, h, a4 ? 1 : 0 };
    const int pairs = wide ? 2 : 1;
    const dim3 block(FMT_BX, FMT_BY), grid((w / (2 * pairs) + FMT_BX - 1) / FMT_BX, (h + FMT_BY - 1) / FMT_BY, n);
#define TSVPP_FMT(K)                                                                                      \
    do {                                                                                                  \
        if (f32 && wide) hipLaunchKernelGGL((K<float, 2>), grid, block, 0, stream, t, g);                 \
        else if (f32) hipLaunchKernelGGL((K<float, 1>), grid, block, 0, stream, t, g);                    \
        else if (wide) hipLaunchKernelGGL((K<uint8_t, 2>), grid, block, 0, stream, t, g);                 \
        else hipLaunchKernelGGL((K<uint8_t, 1>), grid, block, 0, stream, t, g);                           \
    } while (0)
    switch (fourcc) {
    case TSVPP_UYVY: TSVPP_FMT(fmt_uyvy); break;
    case TSVPP_YUV444: TSVPP_FMT(fmt_yuv444); break;
    default: return hipErrorInvalidValue;
    }
#undef TSVPP_FMT
    return hipGetLastError();
}

} // namespace tsvpp
