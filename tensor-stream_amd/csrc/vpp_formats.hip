// vpp_formats.hip -- the two output formats whose chroma filters reach across rows / pixel pairs and
// therefore run as a second pass over the (already cropped / resized) NV12:
//   UYVY      4:2:0 -> 4:2:2, vertical (-1, 9, 9, -1)/16 chroma filter on odd chroma rows  reference src/ColorConversion.cu:107-127, 177-209
//   YUV444    UYVY -> planar 4:4:4, horizontal (-1, 9, 9, -1)/16 filter on odd pixels      :129-173
// (Y800, NV12 and HSV are output flavours of the fused kernels, vpp_kernels.hip.)  One launch per batch
// of <= TSVPP_MAX_BATCH (128) frames (pointer table in the kernarg segment), thread = two horizontal pixel pairs.  Same arithmetic contract as vpp_kernels.hip: plain IEEE,
// no contraction; integer paths use C integer semantics exactly as the reference's <uchar> code.
#include "vpp_kernels.h"

#include <cstdlib>

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
// gfx950 under HSA runs with unaligned global access enabled: a dword / dwordx2 / dwordx4 load or store at any byte address is one instruction (the compiler emits
// exactly that for a memcpy of 4 / 8 / 16 bytes with alignment 1).  Round 6: every vector access of this file goes through these helpers, so that no kernel choice below
// depends on the alignment of a plane, a pitch, a crop origin or an output pointer any more (before: byte-assembled loads and the one-pair kernels for all of those).
template <bool A4> __device__ __forceinline__ uint32_t ld4(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
template <int NB> __device__ __forceinline__ void ld_bytes(const uint8_t *p, uint32_t *v) { // NB bytes at any address: one instruction (see ld4)
    if constexpr (NB == 2) {
        uint16_t q;
        __builtin_memcpy(&q, p, 2);
        v[0] = q;
    } else {
        __builtin_memcpy(v, p, NB);
    }
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
    int out4;     // every output pointer is a multiple of 4
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
    } else if constexpr (sizeof(T) == 1) {
        // one pair = four bytes at byte offset 2 (i w + j): a multiple of 4 (w and j are even) -- one dword store where the output base allows it (g.aligned4 covers it)
        const uint32_t pk = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        if (g.out4) *(uint32_t *)o = pk;
        else {
#pragma unroll
            for (int c = 0; c < 4; c++) o[c] = (uint8_t)v[c];
        }
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
    if (s.w >= 8) { // the callers reach at most 6 columns past either end: one wrap (round 6: no loop in front of the loads)
        const bool before = j < 0, past = j >= s.w;
        j += before ? s.w : (past ? -s.w : 0);
        i += before ? -1 : (past ? 1 : 0);
    } else {
        while (j < 0) {
            j += s.w;
            i--;
        }
        while (j >= s.w) {
            j -= s.w;
            i++;
        }
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
        // pairs -1 .. 2 = bytes j - 2 .. j + 5 of the chroma row(s): two dword loads per tap row (any alignment: see ld4) where they stay inside the row, the
        // flat-order walk (uyvy_chroma, eight calls) in the three threads at a row's ends (round 6: every thread walked -- 8 .. 32 byte loads for two pixels)
        if (j >= 2 && j + 6 <= s.w) {
            int l[4], r[4];
            chroma_422x4<false>(s, i, j - 2, l);
            chroma_422x4<false>(s, i, j + 2, r);
            c[0][0] = l[0], c[0][1] = l[1];
            c[1][0] = l[2], c[1][1] = l[3];
            c[2][0] = r[0], c[2][1] = r[1];
            c[3][0] = r[2], c[3][1] = r[3];
        } else {
#pragma unroll
            for (int k = 0; k < PAIRS + 3; k++) {
                c[k][0] = uyvy_chroma(s, i, j + 2 * (k - 1), 0);
                c[k][1] = uyvy_chroma(s, i, j + 2 * (k - 1), 1);
            }
        }
    }
    T yv[2 * PAIRS], uo[2 * PAIRS], vo[2 * PAIRS];
    if constexpr (PAIRS == 2) {
        const uint32_t yw = g.aligned4 ? ld4<true>(s.y + (size_t)i * s.py + j) : ld4<false>(s.y + (size_t)i * s.py + j);
#pragma unroll
        for (int q = 0; q < 4; q++) yv[q] = fin<T>((int)((yw >> (8 * q)) & 255));
    } else {
        uint32_t yw;
        ld_bytes<2>(s.y + (size_t)i * s.py + j, &yw);
        yv[0] = fin<T>((int)(yw & 255));
        yv[1] = fin<T>((int)(yw >> 8));
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
        bool done = false;
        if constexpr (sizeof(T) == 1) { // one pair = two bytes per plane at offset i w + j: even, and w h is a multiple of 4 (w and h are even) -- 2-byte stores
            if (g.out4) {
                *(uint16_t *)o = (uint16_t)((uint32_t)yv[0] | ((uint32_t)yv[1] << 8));
                *(uint16_t *)(o + wh) = (uint16_t)((uint32_t)uo[0] | ((uint32_t)uo[1] << 8));
                *(uint16_t *)(o + 2 * wh) = (uint16_t)((uint32_t)vo[0] | ((uint32_t)vo[1] << 8));
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                o[q] = yv[q];
                o[wh + q] = uo[q];
                o[2 * wh + q] = vo[q];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Row-pair kernels (the default wherever the planes allow vector loads): output rows 2r and 2r + 1 share chroma row r, so a
// thread converts PX pixels of BOTH rows -- the vertical (-1, 9, 9, -1) filter of an odd chroma row runs once, not twice -- and
// PX is chosen such that every store is 16 bytes per lane and contiguous across the wave (1 KiB per store instruction):
//   UYVY   uint8: PX = 8 (8-byte luma / chroma loads)      fp32: PX = 2 (the single-row kernel above writes 32 bytes per lane in
//   YUV444 uint8: PX = 16 (16-byte loads)                  fp32: PX = 4                two stores that interleave across lanes)
// YUV444 needs the filtered chroma of the pair before and the two pairs after the thread's own: they come from the neighbouring
// lanes by wave shuffles (ds_bpermute); the first / last lane of a wave reload one dword instead, and the first / last thread of a
// ROW -- where the reference's flat indexing wraps into the previous / next row -- takes the scalar path (uyvy_chroma) per row.
__device__ __forceinline__ uint32_t byte_of(uint32_t v, int k) { return (v >> (8 * k)) & 255u; }
// (9 (a + b) - (c + e) + 8) >> 4, clamped, on the four bytes of a dword (src/ColorConversion.cu:107-127).  Shift, clamp and
// pack are gfx950's v_ashr_pk_u8_i32 -- through inline assembly: ROCm 7.2's clang selects that instruction by itself for
// `min(max(x >> 4, 0), 255) | ... << 8` but assumes it zeroes the upper half of its destination, which the hardware PRESERVES
// (measured; vpp_bicubic_int.hip has the full story).  Here the preserved half is put to use: the second instruction packs
// bytes 0-1 under the bytes 2-3 already in place.
__device__ __forceinline__ uint32_t vfilt4(uint32_t a, uint32_t b, uint32_t c, uint32_t e) {
    int sum[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int in = (int)byte_of(a, k) + (int)byte_of(b, k), out = (int)byte_of(c, k) + (int)byte_of(e, k);
        sum[k] = 9 * in - out + 8;
    }
    uint32_t hi;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 4" : "=v"(hi) : "v"(sum[2]), "v"(sum[3])); // upper half: whatever the register held, shifted out below
    uint32_t r = hi << 16;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 4" : "+v"(r) : "v"(sum[0]), "v"(sum[1]));
    return r;
}
// NB bytes of the 4:2:2 chroma of chroma row r (= output rows 2r, 2r + 1) at byte column col: chroma_422 on vectors
template <int NB> __device__ __forceinline__ void chroma_rows(const Nv12View &s, int r, int col, uint32_t *v) {
    constexpr int ND = (NB + 3) / 4;
    const int last = (s.h >> 1) - 1;
    const uint8_t *base = s.uv + col;
    ld_bytes<NB>(base + (size_t)r * s.puv, v);
    if (r & 1) { // uniform per wave
        uint32_t b[ND], c[ND], e[ND];
        ld_bytes<NB>(base + (size_t)min(r + 1, last) * s.puv, b);
        ld_bytes<NB>(base + (size_t)max(r - 1, 0) * s.puv, c);
        ld_bytes<NB>(base + (size_t)min(r + 2, last) * s.puv, e);
#pragma unroll
        for (int d = 0; d < ND; d++) v[d] = vfilt4(v[d], b[d], c[d], e[d]);
    }
}
// One dword of ANY chroma row cr, in two halves so that a caller can put the four loads of several such dwords in flight before it touches any of them (round 6:
// uyvy_chroma's byte loads behind its loops and early returns went out one at a time -- up to 32 dependent round trips in every wave that holds a row's first or last
// lane).  No branch: an even row takes all four taps from itself, which the filter maps onto itself ((9 (a + a) - (a + a) + 8) >> 4 = a); cr outside the plane reads
// as 0 (the reference's flat indexing before / past its UYVY image).
struct ChromaTaps {
    uint32_t a, b, c, e;
};
__device__ __forceinline__ ChromaTaps chroma_taps_any(const Nv12View &s, int cr, int col) {
    const int last = (s.h >> 1) - 1, r = min(max(cr, 0), last);
    const bool odd = (r & 1) != 0;
    const uint8_t *base = s.uv + col;
    ChromaTaps t;
    t.a = ld4<false>(base + (size_t)r * s.puv);
    t.b = ld4<false>(base + (size_t)(odd ? min(r + 1, last) : r) * s.puv);
    t.c = ld4<false>(base + (size_t)(odd ? max(r - 1, 0) : r) * s.puv);
    t.e = ld4<false>(base + (size_t)(odd ? min(r + 2, last) : r) * s.puv);
    return t;
}
__device__ __forceinline__ uint32_t chroma_taps_filter(const Nv12View &s, int cr, const ChromaTaps &t) {
    return (cr < 0 || cr > (s.h >> 1) - 1) ? 0u : vfilt4(t.a, t.b, t.c, t.e);
}
typedef float fvf4_a __attribute__((ext_vector_type(4)));
typedef uint32_t fvu4_a __attribute__((ext_vector_type(4)));
typedef uint32_t fvu2_a __attribute__((ext_vector_type(2)));
// the same vectors behind pointers that promise no alignment (stores at any address: see ld4)
typedef fvf4_a fvf4 __attribute__((aligned(1)));
typedef fvu4_a fvu4 __attribute__((aligned(1)));
typedef fvu2_a fvu2 __attribute__((aligned(1)));
typedef uint32_t fvu1 __attribute__((aligned(1)));

// Every load of the thread is requested before the first use (round 6: the luma loads used to follow the chroma filter, a second round trip -- the fp32 instance alone at
// 1080p 0.63 -> 0.74 of the roofline, uint8 0.72 -> 0.735).  RPW row pairs per thread: 2 gains at 720p (0.74 -> 0.82) and loses at 1080p / 4K (0.742 -> 0.717, 0.747 -> 0.713,
// profiles/r06_formats_ab.txt), so only RPW = 1 is instantiated.
template <class T, int PX, int RPW>
__global__ __launch_bounds__(FMT_BX *FMT_BY) void fmt_uyvy_rp(const FrameTable t, const FmtGeom g) {
    const int f = blockIdx.z, r0 = (blockIdx.y * FMT_BY + threadIdx.y) * RPW, j = (blockIdx.x * FMT_BX + threadIdx.x) * PX;
    if (2 * r0 >= g.h || j >= g.w) return;
    const Nv12View s{ t.y[f], t.uv[f], g.py, g.puv, g.w, g.h };
    constexpr int ND = (PX + 3) / 4;
    const int last_pair = (g.h >> 1) - 1;
    uint32_t c[RPW][ND], yy[RPW][2][ND];
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = min(r0 + k, last_pair); // (a row pair past the frame's last: loaded again, not stored)
        ld_bytes<PX>(s.y + (size_t)(2 * r) * s.py + j, yy[k][0]);
        ld_bytes<PX>(s.y + (size_t)(2 * r + 1) * s.py + j, yy[k][1]);
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) chroma_rows<PX>(s, min(r0 + k, last_pair), j, c[k]);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int r = r0 + k;
        if (r > last_pair) break;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            T *o = (T *)t.out[f] + ((size_t)(2 * r + rr) * s.w + j) * 2;
            if constexpr (sizeof(T) == 1 && PX == 8) { // (U Y V Y) x 4 = 16 bytes; v_perm_b32 interleaves chroma (S1) and luma (S0) bytes
                fvu4_a v;
                v.x = __builtin_amdgcn_perm(yy[k][rr][0], c[k][0], 0x05010400u);
                v.y = __builtin_amdgcn_perm(yy[k][rr][0], c[k][0], 0x07030602u);
                v.z = __builtin_amdgcn_perm(yy[k][rr][1], c[k][1], 0x05010400u);
                v.w = __builtin_amdgcn_perm(yy[k][rr][1], c[k][1], 0x07030602u);
                __builtin_nontemporal_store(v, (fvu4 *)o);
            } else if constexpr (sizeof(T) == 1 && PX == 4) { // (round 6: widths 4 k that are no multiple of 8 -- 300, 1100, 1364): 8 bytes per lane
                const fvu2_a v = { __builtin_amdgcn_perm(yy[k][rr][0], c[k][0], 0x05010400u), __builtin_amdgcn_perm(yy[k][rr][0], c[k][0], 0x07030602u) };
                __builtin_nontemporal_store(v, (fvu2 *)o);
            } else if constexpr (sizeof(T) == 1) { // PX = 2 (widths 4 k + 2: 1366, 854): one pair, 4 bytes per lane
                __builtin_nontemporal_store(__builtin_amdgcn_perm(yy[k][rr][0], c[k][0], 0x05010400u), (fvu1 *)o);
            } else { // PX = 2: U Y V Y as four floats
                const fvf4_a v = { fin<float>((int)byte_of(c[k][0], 0)), fin<float>((int)byte_of(yy[k][rr][0], 0)), fin<float>((int)byte_of(c[k][0], 1)),
                                   fin<float>((int)byte_of(yy[k][rr][0], 1)) };
                __builtin_nontemporal_store(v, (fvf4 *)o);
            }
        }
    }
}

template <class T, int PX>
__global__ __launch_bounds__(FMT_BX *FMT_BY) void fmt_yuv444_rp(const FrameTable t, const FmtGeom g) {
    const int f = blockIdx.z, r = blockIdx.y * FMT_BY + threadIdx.y, lane = threadIdx.x, j = (blockIdx.x * FMT_BX + lane) * PX;
    if (2 * r >= g.h || j >= g.w) return;
    const Nv12View s{ t.y[f], t.uv[f], g.py, g.puv, g.w, g.h };
    constexpr int ND = PX / 4, NP = PX / 2;
    // Everything the thread reads is requested before anything is used (one round trip per wave): both luma rows, then -- in the lanes that cannot get them from a
    // neighbour -- the four taps of up to four more chroma dwords, then the thread's own chroma.
    //   lane 0 / lane 63 of a wave inside a row: the dword before / after the thread's own ones (chroma row r)
    //   a row's first / last thread: the reference indexes its intermediate UYVY image FLAT, so output rows 2 r, 2 r + 1 take the last pair of chroma rows r - 1, r
    //   and the first two pairs of chroma rows r, r + 1 (w % 4 == 0 here: the dwords at byte columns w - 4 and 0)
    uint32_t yw[2][ND];
    ld_bytes<PX>(s.y + (size_t)(2 * r) * s.py + j, yw[0]);
    ld_bytes<PX>(s.y + (size_t)(2 * r + 1) * s.py + j, yw[1]);
    const bool row_first = (j == 0), row_last = (j + PX >= s.w);
    const bool extra_l = lane == 0, extra_r = lane == FMT_BX - 1 || row_last; // (a row's first thread is lane 0 of its wave)
    const int l_row0 = row_first ? r - 1 : r, l_col = row_first ? s.w - 4 : j - 4;
    const int r_row1 = row_last ? r + 1 : r, r_col = row_last ? 0 : j + PX;
    ChromaTaps tl[2] = {}, tr[2] = {};
    if (extra_l) {
        tl[0] = chroma_taps_any(s, l_row0, l_col);
        tl[1] = chroma_taps_any(s, r, l_col);
    }
    if (extra_r) {
        tr[0] = chroma_taps_any(s, r, r_col);
        tr[1] = chroma_taps_any(s, r_row1, r_col);
    }
    uint32_t own[ND];
    chroma_rows<PX>(s, r, j, own);
    // filtered chroma of the neighbouring threads' nearest pairs: (pairs -2, -1) and (pairs NP, NP + 1), per output row of the pair
    uint32_t nbL[2], nbR[2];
    nbL[0] = nbL[1] = (uint32_t)__shfl_up((int)own[ND - 1], 1);
    nbR[0] = nbR[1] = (uint32_t)__shfl_down((int)own[0], 1);
    if (extra_l) {
        nbL[0] = chroma_taps_filter(s, l_row0, tl[0]);
        nbL[1] = chroma_taps_filter(s, r, tl[1]);
    }
    if (extra_r) {
        nbR[0] = chroma_taps_filter(s, r, tr[0]);
        nbR[1] = chroma_taps_filter(s, r_row1, tr[1]);
    }
    int cu[NP + 3], cv[NP + 3]; // chroma of pairs -1 .. NP + 1
#pragma unroll
    for (int p = 0; p < NP; p++) {
        cu[p + 1] = (int)byte_of(own[p >> 1], 2 * (p & 1));
        cv[p + 1] = (int)byte_of(own[p >> 1], 2 * (p & 1) + 1);
    }
    const size_t wh = (size_t)s.w * s.h;
    const uint32_t wh32 = (uint32_t)s.w * (uint32_t)s.h;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int i = 2 * r + rr;
        cu[0] = (int)byte_of(nbL[rr], 2); cv[0] = (int)byte_of(nbL[rr], 3);
        cu[NP + 1] = (int)byte_of(nbR[rr], 0); cv[NP + 1] = (int)byte_of(nbR[rr], 1);
        cu[NP + 2] = (int)byte_of(nbR[rr], 2); cv[NP + 2] = (int)byte_of(nbR[rr], 3);
        T uo[PX], vo[PX];
        const uint32_t row0 = (uint32_t)i * (uint32_t)s.w + (uint32_t)j;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint32_t idx1 = row0 + 2 * p + 1;
            const bool first = idx1 == 1u, last2 = idx1 + 3u >= wh32; // see fmt_yuv444
            uo[2 * p] = fin<T>(cu[p + 1]);
            vo[2 * p] = fin<T>(cv[p + 1]);
            uo[2 * p + 1] = yuv444_odd<T>(cu[p + 1], cu[p + 2], first ? cu[p + 1] : cu[p], last2 ? cu[p + 2] : cu[p + 3]);
            vo[2 * p + 1] = yuv444_odd<T>(cv[p + 1], cv[p + 2], first ? cv[p + 1] : cv[p], last2 ? cv[p + 2] : cv[p + 3]);
        }
        T *o = (T *)t.out[f] + (size_t)i * s.w + j;
        if constexpr (sizeof(T) == 1) { // PX = 16; 8 / 4 (round 6) where the width is no multiple of 16 (1080, 600; 300, 1100)
            auto pk = [](const T *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
            if constexpr (PX == 16) {
                __builtin_nontemporal_store((fvu4_a){ yw[rr][0], yw[rr][1], yw[rr][2], yw[rr][3] }, (fvu4 *)o);
                __builtin_nontemporal_store((fvu4_a){ pk(uo), pk(uo + 4), pk(uo + 8), pk(uo + 12) }, (fvu4 *)(o + wh));
                __builtin_nontemporal_store((fvu4_a){ pk(vo), pk(vo + 4), pk(vo + 8), pk(vo + 12) }, (fvu4 *)(o + 2 * wh));
            } else if constexpr (PX == 8) {
                __builtin_nontemporal_store((fvu2_a){ yw[rr][0], yw[rr][1] }, (fvu2 *)o);
                __builtin_nontemporal_store((fvu2_a){ pk(uo), pk(uo + 4) }, (fvu2 *)(o + wh));
                __builtin_nontemporal_store((fvu2_a){ pk(vo), pk(vo + 4) }, (fvu2 *)(o + 2 * wh));
            } else {
                __builtin_nontemporal_store(yw[rr][0], (fvu1 *)o);
                __builtin_nontemporal_store(pk(uo), (fvu1 *)(o + wh));
                __builtin_nontemporal_store(pk(vo), (fvu1 *)(o + 2 * wh));
            }
        } else { // PX = 4
            __builtin_nontemporal_store((fvf4_a){ fin<float>((int)byte_of(yw[rr][0], 0)), fin<float>((int)byte_of(yw[rr][0], 1)), fin<float>((int)byte_of(yw[rr][0], 2)),
                                                fin<float>((int)byte_of(yw[rr][0], 3)) },
                                        (fvf4 *)o);
            __builtin_nontemporal_store((fvf4_a){ uo[0], uo[1], uo[2], uo[3] }, (fvf4 *)(o + wh));
            __builtin_nontemporal_store((fvf4_a){ vo[0], vo[1], vo[2], vo[3] }, (fvf4 *)(o + 2 * wh));
        }
    }
}

// YUV444 at widths 4 k + 2 (1366, 854: round 6; before, the one-pair kernel fmt_yuv444<T, 1> -- two pixels per thread, 0.09-0.15 of the roofline): row pairs and four
// pixels per thread as fmt_yuv444_rp<T, 4>, but a row's last thread holds ONE pair, so "the thread's pairs and their neighbours" is read as ten bytes of the row
// EXTENDED in flat order -- the previous row's last pair in front of it, the next row's first two pairs behind it.  No shuffles: every thread loads the dwords before,
// at and after its own columns itself (columns clamped into the row, any alignment: see ld4) and the threads at a row's ends splice the wrapped pairs in by a 64-bit
// shift.  Three chroma dwords per tap row instead of one, and still 4 x fewer memory instructions per pixel than the one-pair kernel.
template <class T>
__global__ __launch_bounds__(FMT_BX *FMT_BY) void fmt_yuv444_rp_tail(const FrameTable t, const FmtGeom g) {
    constexpr int PX = 4;
    const int f = blockIdx.z, r = blockIdx.y * FMT_BY + threadIdx.y, j = (blockIdx.x * FMT_BX + threadIdx.x) * PX;
    if (2 * r >= g.h || j >= g.w) return;
    const Nv12View s{ t.y[f], t.uv[f], g.py, g.puv, g.w, g.h };
    const bool tail = j + PX > s.w;                       // one pair: columns w - 2, w - 1
    const int jo = min(j, s.w - 4), so = 8 * (j - jo);    // own dword: loaded at jo, wanted at j (so = 0 or 16 bits)
    const int jr = min(j + 4, s.w - 4), sr = 8 * (j + 4 - jr); // right dword: wanted at j + 4 (sr = 0, 16, 32 or 48 bits)
    const int jl = max(j - 4, 0);
    const bool row_first = j == 0, near_end = j + 8 > s.w;
    uint32_t yw[2];
    yw[0] = ld4<false>(s.y + (size_t)(2 * r) * s.py + jo) >> so;
    yw[1] = ld4<false>(s.y + (size_t)(2 * r + 1) * s.py + jo) >> so;
    ChromaTaps tl[2] = {}, tr[2] = {};
    if (row_first) {
        tl[0] = chroma_taps_any(s, r - 1, s.w - 4);
        tl[1] = chroma_taps_any(s, r, s.w - 4);
    }
    if (near_end) {
        tr[0] = chroma_taps_any(s, r, 0);
        tr[1] = chroma_taps_any(s, r + 1, 0);
    }
    uint32_t cl, co, cr;
    { // chroma_rows<4> at the three columns, the loads of all three before the first filter (one round trip on odd chroma rows too)
        const int last = (s.h >> 1) - 1, cols[3] = { jl, jo, jr };
        uint32_t a[3], b[3], c[3], e[3];
#pragma unroll
        for (int k = 0; k < 3; k++) a[k] = ld4<false>(s.uv + (size_t)r * s.puv + cols[k]);
        if (r & 1) { // uniform per wave
#pragma unroll
            for (int k = 0; k < 3; k++) {
                b[k] = ld4<false>(s.uv + (size_t)min(r + 1, last) * s.puv + cols[k]);
                c[k] = ld4<false>(s.uv + (size_t)max(r - 1, 0) * s.puv + cols[k]);
                e[k] = ld4<false>(s.uv + (size_t)min(r + 2, last) * s.puv + cols[k]);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) a[k] = vfilt4(a[k], b[k], c[k], e[k]);
        }
        cl = a[0]; co = a[1]; cr = a[2];
    }
    uint32_t wl[2] = { cl, cl }, wr[2] = { 0u, 0u };
    if (row_first) {
        wl[0] = chroma_taps_filter(s, r - 1, tl[0]);
        wl[1] = chroma_taps_filter(s, r, tl[1]);
    }
    if (near_end) {
        wr[0] = chroma_taps_filter(s, r, tr[0]);
        wr[1] = chroma_taps_filter(s, r + 1, tr[1]);
    }
    const size_t wh = (size_t)s.w * s.h;
    const uint32_t wh32 = (uint32_t)s.w * (uint32_t)s.h;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int i = 2 * r + rr;
        // pairs -1 | 0, 1 | 2, 3 of the extended row
        const uint32_t own = (uint32_t)((((uint64_t)wr[rr] << 32) | co) >> so);
        const uint32_t right = (uint32_t)((((uint64_t)wr[rr] << 32) | cr) >> sr);
        int cu[5], cv[5];
        cu[0] = (int)byte_of(wl[rr], 2); cv[0] = (int)byte_of(wl[rr], 3);
        cu[1] = (int)byte_of(own, 0); cv[1] = (int)byte_of(own, 1);
        cu[2] = (int)byte_of(own, 2); cv[2] = (int)byte_of(own, 3);
        cu[3] = (int)byte_of(right, 0); cv[3] = (int)byte_of(right, 1);
        cu[4] = (int)byte_of(right, 2); cv[4] = (int)byte_of(right, 3);
        T uo[PX], vo[PX];
        const uint32_t row0 = (uint32_t)i * (uint32_t)s.w + (uint32_t)j;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint32_t idx1 = row0 + 2 * p + 1;
            const bool first = idx1 == 1u, last2 = idx1 + 3u >= wh32; // see fmt_yuv444
            uo[2 * p] = fin<T>(cu[p + 1]);
            vo[2 * p] = fin<T>(cv[p + 1]);
            uo[2 * p + 1] = yuv444_odd<T>(cu[p + 1], cu[p + 2], first ? cu[p + 1] : cu[p], last2 ? cu[p + 2] : cu[p + 3]);
            vo[2 * p + 1] = yuv444_odd<T>(cv[p + 1], cv[p + 2], first ? cv[p + 1] : cv[p], last2 ? cv[p + 2] : cv[p + 3]);
        }
        T *o = (T *)t.out[f] + (size_t)i * s.w + j;
        T yo[PX];
#pragma unroll
        for (int q = 0; q < PX; q++) yo[q] = fin<T>((int)byte_of(yw[rr], q));
        if constexpr (sizeof(T) == 1) {
            auto pk = [](const T *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
            typedef uint16_t fvh1 __attribute__((aligned(1)));
            if (!tail) {
                __builtin_nontemporal_store(pk(yo), (fvu1 *)o);
                __builtin_nontemporal_store(pk(uo), (fvu1 *)(o + wh));
                __builtin_nontemporal_store(pk(vo), (fvu1 *)(o + 2 * wh));
            } else {
                *(fvh1 *)o = (uint16_t)pk(yo);
                *(fvh1 *)(o + wh) = (uint16_t)pk(uo);
                *(fvh1 *)(o + 2 * wh) = (uint16_t)pk(vo);
            }
        } else {
            typedef float fvf2_a __attribute__((ext_vector_type(2)));
            typedef fvf2_a fvf2 __attribute__((aligned(1)));
            if (!tail) {
                __builtin_nontemporal_store((fvf4_a){ yo[0], yo[1], yo[2], yo[3] }, (fvf4 *)o);
                __builtin_nontemporal_store((fvf4_a){ uo[0], uo[1], uo[2], uo[3] }, (fvf4 *)(o + wh));
                __builtin_nontemporal_store((fvf4_a){ vo[0], vo[1], vo[2], vo[3] }, (fvf4 *)(o + 2 * wh));
            } else {
                __builtin_nontemporal_store((fvf2_a){ yo[0], yo[1] }, (fvf2 *)o);
                __builtin_nontemporal_store((fvf2_a){ uo[0], uo[1] }, (fvf2 *)(o + wh));
                __builtin_nontemporal_store((fvf2_a){ vo[0], vo[1] }, (fvf2 *)(o + 2 * wh));
            }
        }
    }
}

hipError_t launch_format(int fourcc, bool f32, const FrameTable &t, int n, int py, int puv, int w, int h, hipStream_t stream) {
    // row-pair kernels: PX pixels per thread (see above), the widest PX that divides the width -- uint8 YUV444 16 / 8 / 4, uint8 UYVY 8 / 4 / 2 (round 6: only 16 and 8,
    // and only on 16-byte aligned planes, pitches and outputs: widths like 1080, 600, 300, 1366 and every odd crop origin fell to the one-row kernels below, 2-5 x
    // slower).  No alignment condition: see ld4.  What is left for the one-row kernels: odd heights, YUV444 at widths 4 k + 2, frames narrower than two threads.
    auto rp_fits = [&](int px) { return (fourcc == TSVPP_UYVY || fourcc == TSVPP_YUV444) && (w % px) == 0 && (h % 2) == 0 && w >= 2 * px; };
    int px = fourcc == TSVPP_UYVY ? (f32 ? 2 : 8) : (f32 ? 4 : 16);
    const int px_min = fourcc == TSVPP_UYVY ? 2 : 4;
    bool rp = rp_fits(px);
    while (!rp && px > px_min) {
        px >>= 1;
        rp = rp_fits(px);
    }
    if (rp) {
        const dim3 block(FMT_BX, FMT_BY), grid((w / px + FMT_BX - 1) / FMT_BX, (h / 2 + FMT_BY - 1) / FMT_BY, n);
        const FmtGeom g{ py, puv, w, h, 1, 1 };
        if (fourcc == TSVPP_UYVY) {
            if (f32) hipLaunchKernelGGL((fmt_uyvy_rp<float, 2, 1>), grid, block, 0, stream, t, g);
            else if (px == 8) hipLaunchKernelGGL((fmt_uyvy_rp<uint8_t, 8, 1>), grid, block, 0, stream, t, g);
            else if (px == 4) hipLaunchKernelGGL((fmt_uyvy_rp<uint8_t, 4, 1>), grid, block, 0, stream, t, g);
            else hipLaunchKernelGGL((fmt_uyvy_rp<uint8_t, 2, 1>), grid, block, 0, stream, t, g);
        } else {
            if (f32) hipLaunchKernelGGL((fmt_yuv444_rp<float, 4>), grid, block, 0, stream, t, g);
            else if (px == 16) hipLaunchKernelGGL((fmt_yuv444_rp<uint8_t, 16>), grid, block, 0, stream, t, g);
            else if (px == 8) hipLaunchKernelGGL((fmt_yuv444_rp<uint8_t, 8>), grid, block, 0, stream, t, g);
            else hipLaunchKernelGGL((fmt_yuv444_rp<uint8_t, 4>), grid, block, 0, stream, t, g);
        }
        return hipGetLastError();
    }
    if (fourcc == TSVPP_YUV444 && (w % 4) == 2 && (h % 2) == 0 && w >= 10) { // widths 4 k + 2
        const dim3 block(FMT_BX, FMT_BY), grid(((w + 3) / 4 + FMT_BX - 1) / FMT_BX, (h / 2 + FMT_BY - 1) / FMT_BY, n);
        const FmtGeom g{ py, puv, w, h, 1, 1 };
        if (f32) hipLaunchKernelGGL((fmt_yuv444_rp_tail<float>), grid, block, 0, stream, t, g);
        else hipLaunchKernelGGL((fmt_yuv444_rp_tail<uint8_t>), grid, block, 0, stream, t, g);
        return hipGetLastError();
    }
    bool wide = (w % 4) == 0, a4 = (py % 4) == 0 && (puv % 4) == 0;
    for (int f = 0; f < n && wide; f++) wide = ((uintptr_t)t.out[f] & 15) == 0;
    for (int f = 0; f < n && a4; f++) a4 = (((uintptr_t)t.y[f] | (uintptr_t)t.uv[f]) & 3) == 0;
    bool o4 = (h % 2) == 0;
    for (int f = 0; f < n && o4; f++) o4 = ((uintptr_t)t.out[f] & 3) == 0;
    const FmtGeom g{ py, puv, w, h, a4 ? 1 : 0, o4 ? 1 : 0 };
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
