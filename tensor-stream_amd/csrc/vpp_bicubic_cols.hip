// vpp_bicubic_cols.hip -- BICUBIC, any ratio: one wave = one output tile, one lane = one output COLUMN.
//
// The reference's value is V(H(row y-1), H(row y), H(row y+1), H(row y+2)) with H the horizontal 4-tap Keys sum (a = -0.75)
// of ONE source row, rounded and clamped to a byte, and V the same sum down the column (src/Resize.cu:27-91, 314-357).  H
// depends only on (source row, output column) and neighbouring output rows share most of their source rows, so the
// workgroup kernels of rounds 1 / 2 (vpp_bicubic_sep_kernel, vpp_bicubic_int_kernel) evaluated H once per (staged row, tile
// column) into an LDS plane -- behind a whole-workgroup pipeline: stage the footprint, barrier, tables, H phase, barrier, V
// phase.  Measured (profiles/r02_bicubic_fallback_bound.txt, r02_bicubic_pmc.txt): VALU ~30 % busy, four workgroups per CU,
// 35 % of the LDS cycles bank conflicts -- bound by that structure, not by arithmetic or HBM.
//
// Here nothing is staged and no workgroup barrier exists.  A wave owns 64 output columns x R output rows:
//   phase 1  lane j walks DOWN the source rows of the tile; per row it loads the eight bytes around its own four taps
//            straight from global memory (adjacent lanes read adjacent, overlapping windows: every 128-byte line is fetched
//            once per wave instruction), selects the taps with one v_perm_b32 (the reference's edge rule -- the +1 AND +2
//            taps collapse, src/Resize.cu:32-43 -- lives in the selector), and writes H of four consecutive rows as ONE dword
//            of its private column in a wave-private, column-major LDS plane (column stride = an odd number of dwords: no
//            bank conflicts);
//   phase 2  the four vertical taps of an output row are four consecutive bytes of that column: two dword reads,
//            v_alignbyte_b32, v_perm_b32 (vertical edge rule), the same 4-tap sum; per-row parameters are wave-uniform: each
//            lane evaluates ONE row's coordinates, the loop fetches them with v_readlane_b32 (no table in LDS, no barrier);
//   chroma   is the same two phases on the interleaved UV plane with lane = (pair column, U | V) and a tap stride of 2;
//   colour   the resized tile goes through a small wave-private byte tile into the usual 2 x 4 thread tiles (colour
//            conversion, every output flavour, vector stores: color_store_tile).
// LDS operations of one wave execute in order, so nothing but wave-level fences separates the phases; 8 workgroups of 4
// independent waves fit a CU.  Rows the vertical taps skip (ratios >= 4) are not evaluated: there the H plane holds the
// four taps of each output row instead of a contiguous run of source rows ("sparse" mode).
//
// Arithmetic: exactly the round-1 scheme (vpp_device.h: cubic4_pair): the reference's fp64 sum is evaluated in fp32 pairs and
// redone in fp64, as the reference does it, whenever the fp32 sum is within 5e-4 of a rounding tie.  INT = true (host: every
// weight of the request is a multiple of 1/16): the integer evaluation of vpp_bicubic_int.hip -- coefficients x 2^14 in
// int16, two v_dot2_i32_i16 per sum, v_ashr_pk_u8_i32 -- in the same wave structure.
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef short s16x2c __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2a4c __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3a4c __attribute__((ext_vector_type(3), aligned(4)));

// One axis position: first sample of the 4-sample window, the tap selector (byte k = offset of tap k from the window start,
// 0..3: the reference's edge rule), weight and coefficients.
struct BcAxis {
    int ws;        // window start (sample units of that grid)
    uint32_t sel;  // tap offsets from ws, one per byte
    int maxoff;    // largest tap offset
    float w;
    float c[4];
    int c01, c23;  // INT: folded integer coefficients per WINDOW offset (vpp_bicubic_int.hip)
};

// `clamp_limit`: the size the coordinate clamps use (the LUMA size on both grids, src/Resize.cu:325-347); `tap_limit`: the
// size the edge rule uses, in samples of this grid.
template <bool INT>
__device__ __forceinline__ BcAxis bc_axis(int idx, float ratio, int clamp_limit, int tap_limit) {
    BcAxis a;
    int p;
    double w;
    bicubic_axis(idx, ratio, clamp_limit, p, w);
    int lo, hi;
    bicubic_offsets(p, 1, tap_limit, lo, hi);
    a.ws = p - lo;
    a.sel = (uint32_t)lo << 8 | (uint32_t)(lo + hi) << 16 | (uint32_t)(lo + 2 * hi) << 24;
    a.maxoff = lo + 2 * hi;
    a.w = (float)w; // exact: the fraction of a float coordinate
    cubic_coeffs_f(a.w, a.c);
    a.c01 = a.c23 = 0;
    if constexpr (INT) { // exact for w = k / 16; a collapsed tap's coefficient moves onto the sample it collapsed on
        const int C[4] = { (int)(a.c[0] * 16384.0f), (int)(a.c[1] * 16384.0f), (int)(a.c[2] * 16384.0f), (int)(a.c[3] * 16384.0f) };
        int wg[4] = { 0, 0, 0, 0 };
        const int off[4] = { 0, lo, lo + hi, lo + 2 * hi };
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int o = 0; o < 4; o++) wg[o] += (off[k] == o) ? C[k] : 0;
        a.c01 = (wg[0] & 0xffff) | (wg[1] << 16);
        a.c23 = (wg[2] & 0xffff) | (wg[3] << 16);
        a.maxoff = min(3, tap_limit - 1 - a.ws); // the window is always four consecutive samples; those past the plane carry weight 0
    }
    return a;
}

// Two 4-tap sums on packed taps (tap k of sum 0 / 1 in byte k of t0 / t1) -> two integer-valued floats in [0, 255].
__device__ __forceinline__ f2 bc_sum_pair(uint32_t t0, uint32_t t1, float w0, float w1, const float c0[4], const float c1[4]) {
    f2 p[4];
    p[0] = (f2){ __builtin_amdgcn_cvt_f32_ubyte0(t0), __builtin_amdgcn_cvt_f32_ubyte0(t1) };
    p[1] = (f2){ __builtin_amdgcn_cvt_f32_ubyte1(t0), __builtin_amdgcn_cvt_f32_ubyte1(t1) };
    p[2] = (f2){ __builtin_amdgcn_cvt_f32_ubyte2(t0), __builtin_amdgcn_cvt_f32_ubyte2(t1) };
    p[3] = (f2){ __builtin_amdgcn_cvt_f32_ubyte3(t0), __builtin_amdgcn_cvt_f32_ubyte3(t1) };
    f2 s = ((f2){ c0[0], c1[0] } * p[0] + (f2){ c0[1], c1[1] } * p[1]) + (f2){ c0[2], c1[2] } * p[2];
    s = s + (f2){ c0[3], c1[3] } * p[3];
    const f2 magic = { 12582912.0f, 12582912.0f };
    f2 r = (s + magic) - magic; // nearest even: differs from the reference's half-away rule only AT a tie, and those are redone
    const f2 dd = s - r;
    r.x = __builtin_amdgcn_fmed3f(r.x, 0.0f, 255.0f);
    r.y = __builtin_amdgcn_fmed3f(r.y, 0.0f, 255.0f);
    const bool tie0 = fabsf(dd.x) > 0.5f - CUBIC_DELTA, tie1 = fabsf(dd.y) > 0.5f - CUBIC_DELTA;
    if (tie0 || tie1) { // rarely taken: the reference's own fp64 evaluation decides
        if (tie0) {
            double c[4];
            cubic_coeffs((double)w0, c);
            r.x = (float)cubic4(c, (int)(t0 & 255u), (int)((t0 >> 8) & 255u), (int)((t0 >> 16) & 255u), (int)(t0 >> 24));
        }
        if (tie1) {
            double c[4];
            cubic_coeffs((double)w1, c);
            r.y = (float)cubic4(c, (int)(t1 & 255u), (int)((t1 >> 8) & 255u), (int)((t1 >> 16) & 255u), (int)(t1 >> 24));
        }
    }
    return r;
}
// INT: S + 8192 of the 4-tap sum over the window bytes of q; the value is clamp(that >> 14, 0, 255)
__device__ __forceinline__ int bc_isum(uint32_t q, int c01, int c23) {
    const uint32_t p01 = __builtin_amdgcn_perm(0u, q, 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, q, 0x0c030c02u);
    int s = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2c, p01), __builtin_bit_cast(s16x2c, c01), 8192, false);
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2c, p23), __builtin_bit_cast(s16x2c, c23), s, false);
}
// four biased sums -> four bytes (vpp_bicubic_int.hip: round_clamp_pack4; the instruction keeps the upper half of its destination)
__device__ __forceinline__ uint32_t bc_pack4(int s0, int s1, int s2, int s3) {
    uint32_t hi;
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 14" : "=v"(hi) : "v"(s2), "v"(s3));
    uint32_t r = hi << 16;
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 14" : "+v"(r) : "v"(s0), "v"(s1));
    return r;
}
__device__ __forceinline__ uint32_t bc_pack4f(f2 a, f2 b) { // integer-valued floats in [0, 255]
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(a.x, 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(a.y, 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(b.x, 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(b.y, 3u, r);
}
__device__ __forceinline__ uint32_t rep4(uint32_t b) { return __builtin_amdgcn_perm(0u, b, 0u); } // byte 0 into all four bytes

// The window bytes of one lane in one source row.  `plane` is the frame's plane pointer rounded down to a dword (uniform),
// `a` the byte offset of the lane's window start from it (row * pitch + window start + rounding), `need` the last window byte
// the taps read.  NDW = 2 (luma: 8 bytes cover any 4-byte window) or 3 (chroma: taps 2 apart, 7-byte window).  `last`: the
// row is the plane's last one -- a dword past the last needed byte is then re-pointed at the first, nothing is read beyond the
// dword of the last needed byte; every other row may over-read into the next row of the same plane.
template <int NDW>
__device__ __forceinline__ void bc_load(const uint8_t *plane, uint32_t a, int need, bool last, uint32_t (&dw)[NDW]) {
    const uint32_t *p = (const uint32_t *)(plane + (a & ~3u));
    if (!last) {
        if constexpr (NDW == 2) {
            const u32x2a4c v = *(const u32x2a4c *)p;
            dw[0] = v.x; dw[1] = v.y;
        } else {
            const u32x3a4c v = *(const u32x3a4c *)p;
            dw[0] = v.x; dw[1] = v.y; dw[2] = v.z;
        }
    } else {
        const int hi = (int)(a & 3u) + need; // last needed byte, from the aligned dword
#pragma unroll
        for (int k = 0; k < NDW; k++) dw[k] = p[(4 * k <= hi) ? k : 0];
    }
}

// One plane (luma: STEP 1, lane = column; chroma: STEP 2, lane = (pair column, component)) through phases 1 and 2.
// hp: this wave's H plane for the plane (byte address of the LANE's column); res: the wave's result tile (row-major, 64 bytes
// per row), nout output rows.  ax: the lane's column; rows come from the lanes `rl0 ...` of the row registers (v_readlane).
template <bool INT, int STEP>
__device__ __forceinline__ void bc_plane(const uint8_t *plane, uint32_t pm, int pitch, int rows_in_plane, const BcAxis &ax, int comp, bool sparse,
                                         uint8_t *hcol, uint8_t *res, int lane, int nout, int rl0, const BcAxis &rw) {
    constexpr int NDW = STEP == 1 ? 2 : 3;
    // ---- phase 1: H of every needed source row of this lane's column, four rows per dword
    const uint32_t xoff = pm + (uint32_t)(STEP * ax.ws + comp);
    const int xneed = STEP * ax.maxoff;
    const int ws0 = __builtin_amdgcn_readlane(rw.ws, rl0), wsl = __builtin_amdgcn_readlane(rw.ws, rl0 + nout - 1);
    const int ylo = ws0, yhi = min(wsl + 3, rows_in_plane - 1); // dense: every row from the first window's start to the last window's end
    const int ng = sparse ? nout : ((yhi - ylo + 1 + 3) >> 2);
    for (int g = 0; g < ng; g++) {
        int rr[4];
        if (sparse) { // the four taps of output row g, in tap order
            const int ws = __builtin_amdgcn_readlane(rw.ws, rl0 + g);
            const uint32_t sel = (uint32_t)__builtin_amdgcn_readlane((int)rw.sel, rl0 + g);
#pragma unroll
            for (int k = 0; k < 4; k++) rr[k] = ws + (int)((sel >> (8 * k)) & 255u);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) rr[k] = min(ylo + 4 * g + k, yhi);
        }
        uint32_t dw[4][NDW], sh[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t a = (uint32_t)rr[k] * (uint32_t)pitch + xoff;
            sh[k] = a & 3u;
            bc_load<NDW>(plane, a, xneed, rr[k] >= rows_in_plane - 1, dw[k]);
        }
        uint32_t tp[4]; // packed taps (float) / window bytes (INT) of the four rows
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (STEP == 1) {
                if constexpr (INT) tp[k] = __builtin_amdgcn_alignbyte(dw[k][1], dw[k][0], sh[k]);
                else tp[k] = __builtin_amdgcn_perm(dw[k][1], dw[k][0], ax.sel + rep4(sh[k]));
            } else { // bytes sh .. sh + 7 of the three dwords, then every second one
                const uint32_t a0 = __builtin_amdgcn_alignbyte(dw[k][1], dw[k][0], sh[k]), a1 = __builtin_amdgcn_alignbyte(dw[k][2], dw[k][1], sh[k]);
                if constexpr (INT) tp[k] = __builtin_amdgcn_perm(a1, a0, 0x06040200u);
                else tp[k] = __builtin_amdgcn_perm(a1, a0, ax.sel + ax.sel); // tap offsets in bytes: 2 x
            }
        }
        uint32_t h4;
        if constexpr (INT) {
            h4 = bc_pack4(bc_isum(tp[0], ax.c01, ax.c23), bc_isum(tp[1], ax.c01, ax.c23), bc_isum(tp[2], ax.c01, ax.c23), bc_isum(tp[3], ax.c01, ax.c23));
        } else {
            h4 = bc_pack4f(bc_sum_pair(tp[0], tp[1], ax.w, ax.w, ax.c, ax.c), bc_sum_pair(tp[2], tp[3], ax.w, ax.w, ax.c, ax.c));
        }
        *(uint32_t *)(hcol + 4 * g) = h4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2: vertical sums, two output rows per step
    for (int i = 0; i < nout; i += 2) {
        uint32_t tp[2];
        float wv[2], cv[2][4];
        int ic01[2], ic23[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int rl = rl0 + min(i + e, nout - 1);
            const int off = sparse ? 4 * min(i + e, nout - 1) : __builtin_amdgcn_readlane(rw.ws, rl) - ylo;
            const uint32_t *p = (const uint32_t *)(hcol + (off & ~3));
            const uint32_t win = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)off & 3u);
            if constexpr (INT) {
                tp[e] = win;
                ic01[e] = __builtin_amdgcn_readlane(rw.c01, rl);
                ic23[e] = __builtin_amdgcn_readlane(rw.c23, rl);
            } else {
                const uint32_t sel = sparse ? 0x03020100u : (uint32_t)__builtin_amdgcn_readlane((int)rw.sel, rl);
                tp[e] = __builtin_amdgcn_perm(0u, win, sel);
                wv[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rw.w), rl));
#pragma unroll
                for (int k = 0; k < 4; k++) cv[e][k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rw.c[k]), rl));
            }
        }
        uint32_t v0, v1;
        if constexpr (INT) {
            const uint32_t pk = bc_pack4(bc_isum(tp[0], ic01[0], ic23[0]), bc_isum(tp[1], ic01[1], ic23[1]), 0, 0);
            v0 = pk & 255u;
            v1 = (pk >> 8) & 255u;
        } else {
            const f2 v = bc_sum_pair(tp[0], tp[1], wv[0], wv[1], cv[0], cv[1]);
            v0 = (uint32_t)v.x;
            v1 = (uint32_t)v.y;
        }
        res[i * 64 + lane] = (uint8_t)v0;
        if (i + 1 < nout) res[(i + 1) * 64 + lane] = (uint8_t)v1;
    }
}

template <int OUT, bool INT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bicubic_cols_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lane = (int)(threadIdx.x & 63u), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int R = 8 * d.rpt; // output rows of the tile (multiple of 8, <= 32)
    const int j_first = id.tx * 256 + wave * 64, i_first = id.ty * R;
    if (j_first >= d.dst_w) return; // (no workgroup barrier anywhere in this kernel)
    const int nrows = min(R, d.dst_h - i_first), ncrows = nrows >> 1;
    const bool sparse = d.bc_sparse != 0;

    // wave-private LDS: H planes (column-major, one column per lane), result tiles (row-major)
    uint8_t *wl = lds_raw + wave * d.bc_wave_bytes;
    uint8_t *hy = wl + lane * d.hcs_y, *huv = wl + 64 * d.hcs_y + lane * d.hcs_uv;
    uint8_t *yt = wl + 64 * (d.hcs_y + d.hcs_uv), *uvt = yt + 64 * R;

    // rows: lane r < 32 evaluates luma output row i_first + r, lane 32 + r chroma row i_first / 2 + r (the luma formulas on
    // the chroma grid, clamps against the luma height: src/Resize.cu:325-347)
    const bool crow = lane >= 32;
    const int ridx = crow ? min((i_first >> 1) + (lane - 32), (d.dst_h >> 1) - 1) : min(i_first + lane, d.dst_h - 1);
    const BcAxis rw = bc_axis<INT>(ridx, d.yr, d.src_h, crow ? (d.src_h >> 1) : d.src_h);

    const uint32_t ym = (uint32_t)((uintptr_t)t.y[id.frame] & 3), uvm = (uint32_t)((uintptr_t)t.uv[id.frame] & 3);
    { // luma: lane = column (columns past the frame repeat the last one; never stored)
        const BcAxis ax = bc_axis<INT>(min(j_first + lane, d.dst_w - 1), d.xr, d.src_w, d.src_w);
        bc_plane<INT, 1>(t.y[id.frame] - ym, ym, d.pitch_y, d.src_h, ax, 0, sparse, hy, yt, lane, nrows, 0, rw);
    }
    if constexpr (!kLumaOnly<OUT>) { // chroma: lane = (pair column, component); taps in pair units, 2 bytes apart
        const BcAxis ax = bc_axis<INT>(min((j_first >> 1) + (lane >> 1), (d.dst_w >> 1) - 1), d.xr, d.src_w, d.src_w >> 1);
        bc_plane<INT, 2>(t.uv[id.frame] - uvm, uvm, d.pitch_uv, d.src_h >> 1, ax, lane & 1, sparse, huv, uvt, lane, ncrows, 32, rw);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // colour conversion + stores: 16 x 4 thread tiles of 4 x 2 pixels per 8-row slab
    const int lx = lane & 15, ly = lane >> 4;
    const int j0 = j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    for (int s = 0; s < d.rpt; s++) {
        const int r0 = s * 8 + ly * PXH, i0 = i_first + r0;
        if (i0 >= d.dst_h) break;
        float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const uint32_t v = *(const uint32_t *)(yt + (r0 + r) * 64 + lx * PXW);
            Yf[r][0] = __builtin_amdgcn_cvt_f32_ubyte0(v);
            Yf[r][1] = __builtin_amdgcn_cvt_f32_ubyte1(v);
            Yf[r][2] = __builtin_amdgcn_cvt_f32_ubyte2(v);
            Yf[r][3] = __builtin_amdgcn_cvt_f32_ubyte3(v);
        }
        if constexpr (!kLumaOnly<OUT>) {
            const uint32_t c = *(const uint32_t *)(uvt + (r0 >> 1) * 64 + lx * PXW);
            Uf[0] = __builtin_amdgcn_cvt_f32_ubyte0(c);
            Vf[0] = __builtin_amdgcn_cvt_f32_ubyte1(c);
            Uf[1] = __builtin_amdgcn_cvt_f32_ubyte2(c);
            Vf[1] = __builtin_amdgcn_cvt_f32_ubyte3(c);
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

hipError_t launch_bicubic_cols(OutKind out, bool integer, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block(MAX_THREADS);
    if (info) {
        info->kernel = integer ? "vpp_bicubic_cols_kernel<OUT, true>" : "vpp_bicubic_cols_kernel<OUT, false>";
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    switch (out) {
#define TSVPP_BC(O)                                                                                                \
    case O:                                                                                                        \
        if (integer) hipLaunchKernelGGL((vpp_bicubic_cols_kernel<O, true>), grid, block, lds_bytes, stream, d, t);  \
        else hipLaunchKernelGGL((vpp_bicubic_cols_kernel<O, false>), grid, block, lds_bytes, stream, d, t);         \
        break;
        TSVPP_BC(O_U8_PLANAR) TSVPP_BC(O_U8_MERGED) TSVPP_BC(O_F32_PLANAR) TSVPP_BC(O_F32_MERGED) TSVPP_BC(O_NV12_U8)
        TSVPP_BC(O_NV12_F32) TSVPP_BC(O_Y800_U8) TSVPP_BC(O_Y800_F32) TSVPP_BC(O_HSV_F32)
#undef TSVPP_BC
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
