// vpp_bicubic_r32.hip -- BICUBIC at the exact ratios 3 : 2 (1920x1080 -> 1280x720, 3840x2160 -> 2560x1440) and 2 : 1 (3840x2160 -> 1920x1080,
// 1920x1080 -> 960x540) on both axes as a streaming kernel: no LDS staging, no barrier, no coordinate arithmetic, no tables.
//
// Why (round 4): the integer kernel of vpp_bicubic_int.hip executes 7.3e7 VALU wave-instructions per 64-frame 1080p -> 720p launch -- at one
// instruction per SIMD per 4 cycles that alone is ~120 us against an HBM time of ~150 us for fp32 and ~63 us for uint8 outputs -- and 35 % of its
// LDS cycles are bank conflicts (profiles/r03_bicubic_pmc.txt).  At these two ratios every weight is 1/4, 3/4 or 1/2: Keys' coefficients are
// integers over 256 whose magnitudes fit a byte, and a thread's taps sit at compile-time byte positions of a dword-aligned run, so each 4-tap sum
// is two to four v_dot4_u32_u8 on the source dwords as they were loaded (vpp_bicubic_r32_core.h: the arithmetic, the geometry and the edge rules,
// shared with the host build the CPU suite checks against the oracle).
//
// A thread converts 8 output columns x 4 output rows from 8 (10) luma and 5 (6) chroma source rows of 12 (16) bytes + one dword before and after,
// loaded straight from global memory (dwordx3 / dwordx4 + two dwords per row, all issued before the first use).  Horizontal sums of four source
// rows are packed into one dword per output column, so a vertical 4-tap sum is again ONE dword (v_alignbyte of two) through the same dot4 chain:
// the north-star's "wavefront shuffles for the bicubic taps" are not needed -- no lane ever needs another lane's samples.
// Outputs: every flavour of the colour back end.  uint8 planar 8-byte stores; uint8 / fp32 merged rows are exchanged through LDS inside the wave so
// that each store instruction writes one contiguous run (cf. MergedRun, vpp_device.h); fp32 planar: the resized bytes are dealt out between the lanes
// of a wave by shuffles before the colour conversion, so that 16-byte stores cover whole lines (BcDeal below).
#include "vpp_device.h"
#include "vpp_bicubic_r32_core.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef uint32_t bq2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t bq4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void bc_st8(uint8_t *base, uint32_t off, uint32_t lo, uint32_t hi, int nt) {
    const bq2 v = { lo, hi };
    if (nt) __builtin_nontemporal_store(v, (bq2 *)(base + off));
    else *(bq2 *)(base + off) = v;
}
// the four bytes of a dword as integer-valued floats (v_cvt_f32_ubyte0..3)
__device__ __forceinline__ void bc_unpack4(uint32_t w, float *f) {
    f[0] = (float)(w & 255u);
    f[1] = (float)((w >> 8) & 255u);
    f[2] = (float)((w >> 16) & 255u);
    f[3] = (float)(w >> 24);
}
// Four pixels of one row -> normalised (c0, c1, c2) per pixel, the fp32 arithmetic of color_store_row (vpp_device.h): merged order c0 c1 c2 c0 ...
template <bool HSV>
__device__ __forceinline__ void bc_color4_f32(const float *Yf, const float *t0, const float *tg, const float *t2, const tsvpp_coeffs &k, float (&o)[12]) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        f2 y = { Yf[2 * p], Yf[2 * p + 1] };
        y = y - (f2){ k.y_offset, k.y_offset };
        y.x = __builtin_fmaxf(0.0f, y.x);
        y.y = __builtin_fmaxf(0.0f, y.y);
        y = y * (f2){ k.y_scale, k.y_scale };
        f2 c0 = norm255(trunc_clamp255(y + (f2){ t0[p], t0[p] }));
        f2 c1 = norm255(trunc_clamp255(y + (f2){ tg[p], tg[p] }));
        f2 c2 = norm255(trunc_clamp255(y + (f2){ t2[p], t2[p] }));
        if constexpr (HSV) {
            float h0, s0, v0, h1, s1, v1;
            hsv_pixel(c0.x, c1.x, c2.x, h0, s0, v0);
            hsv_pixel(c0.y, c1.y, c2.y, h1, s1, v1);
            c0 = (f2){ h0, h1 };
            c1 = (f2){ s0, s1 };
            c2 = (f2){ v0, v1 };
        }
        o[6 * p + 0] = c0.x; o[6 * p + 1] = c1.x; o[6 * p + 2] = c2.x;
        o[6 * p + 3] = c0.y; o[6 * p + 4] = c1.y; o[6 * p + 5] = c2.y;
    }
}

constexpr int BCR_COLS = 8, BCR_ROWS = 4;

// fp32 PLANAR rows: a lane owns 8 consecutive floats (32 bytes) of a plane row, so its two 16-byte stores would each cover half of every 128-byte
// line the wave touches -- measured 0.23 of the roofline against 0.67 for full-line stores (profiles/r04_bicubic_r32_ab.txt, first version; an
// exchange of the FLOATS through LDS, 12 b128 operations per row, reached 0.67).  The lanes of a run (the lanes of a wave that share the output
// rows, A of them active) therefore trade their RESIZED BYTES before the colour conversion, one wave shuffle (ds_bpermute) per dword: the run's
// 2 A groups of four columns are dealt out so that lane m converts groups m and A + m -- group g is the (g & 1) half of lane g / 2 -- and every
// store instruction of the wave writes ONE contiguous span of 16 A bytes per plane row.  No LDS memory, no barrier.
struct BcDeal {
    int src[2], odd[2]; // source lane (byte address for ds_bpermute) and half of the two groups this lane receives
};
__device__ __forceinline__ BcDeal bc_deal(int lane, int m, int a) {
    BcDeal x;
    const int base = lane - m; // first lane of the run
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int g = k * a + m;
        x.src[k] = 4 * (base + (g >> 1));
        x.odd[k] = g & 1;
    }
    return x;
}
// the dword of group k: (lo, hi) = the two halves every lane holds
__device__ __forceinline__ uint32_t bc_dealt(const BcDeal &x, int k, uint32_t lo, uint32_t hi) {
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(x.src[k], (int)lo), b = (uint32_t)__builtin_amdgcn_ds_bpermute(x.src[k], (int)hi);
    return x.odd[k] ? b : a;
}

// The extended rows of a thread tile, neighbours by wave shuffle.  Measured (profiles/r04_bicubic_r32_pmc.txt): with three loads per row (the run + the
// dword before + the dword after: bc_load_rows) the fp32 launch keeps the texture addresser 79 % busy -- a wave's 39 loads and 24 stores cost ~63 TA
// cycles per instruction: the kernel is bound by the ISSUE of memory instructions, not by bytes.  The dword before a thread's run IS the last dword
// of its left neighbour's run and the dword after it the first of its right neighbour's: one v_mov_b32_dpp wave_shr:1 / wave_shl:1 each instead of a
// load (fp32 planar 1080p -> 720p 0.64 -> 0.71, 4K -> 1080p 0.68 -> 0.73: profiles/r04_bicubic_r32_ab.txt).  Only the first / last lane of a wave
// -- their neighbour is in another wave -- load theirs: two single-lane loads per row.  For workgroups 64 threads wide (a wave = one run of lanes
// that share their output rows); narrower workgroups (uint8 outputs of widths that would idle lanes: VALU-bound, not TA-bound) keep the three loads.
template <int P2, int NROWS>
__device__ __forceinline__ void bcr_load_rows(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, bool wide, bool run_first,
                                              bool run_last, uint32_t (&ext)[NROWS][P2 + 2], uint32_t (&nb)[NROWS][2]) {
    if (!wide) {
        bc_load_rows<P2, NROWS>(plane, pitch, row0, plane_rows, q, first, last, ext);
        return;
    }
    constexpr int RUN = 4 * P2;
    const uint32_t col = (uint32_t)(RUN * q);
    uint32_t off[NROWS];
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        off[r] = (uint32_t)bc_row<NROWS>(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<P2>(plane + off[r], &ext[r][1]);
        nb[r][0] = 0u; // (any defined value: the lanes that do not load below take their neighbour's dword)
        nb[r][1] = 0u;
    }
    if (run_first && !first) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) bc_ld<1>(plane + (off[r] - 4u), &nb[r][0]);
    }
    if (run_last && !last) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) bc_ld<1>(plane + (off[r] + (uint32_t)RUN), &nb[r][1]);
    }
}
// ... the shuffles, once the loads have landed (call after the scheduling barrier that keeps the loads together).  v_mov_b32_dpp leaves a lane
// without a source lane (the wave's first for wave_shr, its last for wave_shl) on the `old` operand: the dword that lane loaded itself.  (A run that
// ends before lane 63 ends at the frame's right edge: its last lane's "dword after" is never selected.)
template <int P2, int NROWS>
__device__ __forceinline__ void bcr_neighbours(uint32_t (&ext)[NROWS][P2 + 2], const uint32_t (&nb)[NROWS][2], bool wide) {
    if (!wide) return;
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][0], (int)ext[r][P2], 0x138, 0xf, 0xf, false);     // wave_shr:1: lane i <- lane i - 1
        ext[r][P2 + 1] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][1], (int)ext[r][1], 0x130, 0xf, 0xf, false); // wave_shl:1: lane i <- lane i + 1
    }
}

template <int OUT, int P2>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bicubic_r32_kernel(const LaunchDesc d, const FrameTable t) {
    using G = BcGeom<P2>;
    constexpr bool LUMA_ONLY = kLumaOnly<OUT>;
    const TileId id = decode_tile(d); // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = BCR_COLS * q, i0 = BCR_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    const bool first = q == 0, last = j0 + BCR_COLS >= d.dst_w, last_row = i0 + BCR_ROWS >= d.dst_h;
    const int nt = d.nt_stores;
    uint8_t *out = (uint8_t *)t.out[id.frame];
    const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;

    // the run this lane belongs to: the lanes of the wave that share its output rows (A of them active)
    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - BCR_COLS * run_m)) / BCR_COLS);
    const bool run_first = run_m == 0, run_last = run_m == run_a - 1;
    // every load of the tile first, then the neighbour shuffles and the column edge fix-ups, then the arithmetic
    uint32_t ey[G::NYR][P2 + 2], xy[G::NYR][2], ec[G::NCR][P2 + 2], xc[G::NCR][2], ny[G::NYR][2], nc[G::NCR][2];
    const bool wide = d.tx >= 64; // wave-uniform: a wave is one run
    bcr_load_rows<P2, G::NYR>(t.y[id.frame], d.pitch_y, 2 * P2 * n4 - 1, d.src_h, q, first, last, wide, run_first, run_last, ey, ny);
    if constexpr (!LUMA_ONLY) bcr_load_rows<P2, G::NCR>(t.uv[id.frame], d.pitch_uv, P2 * n4 - 1, d.src_h >> 1, q, first, last, wide, run_first, run_last, ec, nc);
#ifndef TSVPP_BCR_NO_SCHED_BARRIER
    // nothing is scheduled across this point: without it the compiler sinks the loads into three batches between the arithmetic of the row groups
    // (fewer live registers, but a wave then waits for memory three times per tile)
    __builtin_amdgcn_sched_barrier(0);
#endif
    bcr_neighbours<P2, G::NYR>(ey, ny, wide);
    if constexpr (!LUMA_ONLY) bcr_neighbours<P2, G::NCR>(ec, nc, wide);
    bc_fix_rows<P2, false, G::NYR>(ey, xy, first, last);
    if constexpr (!LUMA_ONLY) bc_fix_rows<P2, true, G::NCR>(ec, xc, first, last);
    uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
    bc_tile<P2, !LUMA_ONLY>(ey, xy, ec, xc, last_row, ylo, yhi, clo, chi);

    // Merged rows leave as several 16-byte pieces per lane (uint8: 24 bytes, fp32: 96 bytes): exchanged through LDS inside the wave so that each store
    // instruction writes one contiguous span (cf. MergedRun, vpp_device.h).  fp32 planar rows: the resized bytes are dealt out by wave shuffles (BcDeal).
    constexpr bool MERGED8 = OUT == O_U8_MERGED, MERGED32 = (OUT == O_F32_MERGED || OUT == O_HSV_F32);
    constexpr int PIECE = MERGED8 ? 24 : (MERGED32 ? 96 : 0); // LDS bytes per lane
    __shared__ __attribute__((aligned(16))) uint8_t slab[PIECE ? MAX_THREADS * PIECE : 16];
    uint8_t *run_lds = nullptr;
    if constexpr (PIECE != 0) run_lds = slab + ((int)threadIdx.x - run_m) * PIECE;

    if constexpr (OUT == O_NV12_U8 || OUT == O_Y800_U8) { // the resized planes themselves: packed bytes as they are
#pragma unroll
        for (int r = 0; r < BCR_ROWS; r++) bc_st8(out, (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0, ylo[r], yhi[r], nt);
        if constexpr (OUT == O_NV12_U8) {
#pragma unroll
            for (int rc = 0; rc < 2; rc++) bc_st8(out, plane + (uint32_t)((i0 >> 1) + rc) * (uint32_t)d.dst_w + (uint32_t)j0, clo[rc], chi[rc], nt);
        }
        return;
    } else if constexpr (OUT == O_NV12_F32 || OUT == O_Y800_F32) { // ... / 255, the bytes dealt out first (BcDeal): full-line stores
        const BcDeal deal = bc_deal((int)threadIdx.x & 63, run_m, run_a);
        const uint32_t run_j0 = (uint32_t)(j0 - BCR_COLS * run_m);
        auto row_f32 = [&](uint32_t lo, uint32_t hi, uint32_t rowpix) { // rowpix: element index of the row's first pixel in its plane
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float f[4];
                bc_unpack4(bc_dealt(deal, k, lo, hi), f);
                const f2 a = norm255((f2){ f[0], f[1] }), b = norm255((f2){ f[2], f[3] });
                st4o(out, (rowpix + run_j0 + 4u * (uint32_t)(k * run_a + run_m)) * 4u, a.x, a.y, b.x, b.y, nt);
            }
        };
#pragma unroll
        for (int r = 0; r < BCR_ROWS; r++) row_f32(ylo[r], yhi[r], (uint32_t)(i0 + r) * (uint32_t)d.dst_w);
        if constexpr (OUT == O_NV12_F32) {
#pragma unroll
            for (int rc = 0; rc < 2; rc++) row_f32(clo[rc], chi[rc], plane + (uint32_t)((i0 >> 1) + rc) * (uint32_t)d.dst_w);
        }
        return;
    } else if constexpr (OUT == O_F32_PLANAR) {
        // the resized bytes dealt out (BcDeal), then the usual four-pixel colour rows: lane m stores columns run_j0 + 4 (k A + m) .. + 3, k = 0, 1
        const BcDeal deal = bc_deal((int)threadIdx.x & 63, run_m, run_a);
        const uint32_t run_j0 = (uint32_t)(j0 - BCR_COLS * run_m);
        const MergedRun none{ nullptr, 0, 1 };
#pragma unroll
        for (int rc = 0; rc < 2; rc++) {
            uint32_t cx[2];
#pragma unroll
            for (int k = 0; k < 2; k++) cx[k] = bc_dealt(deal, k, clo[rc], chi[rc]);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int r = 2 * rc + rr;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    float uvf[4], yf[4], t0[2], tg[2], t2[2];
                    bc_unpack4(cx[k], uvf); // U0 V0 U1 V1
                    bc_unpack4(bc_dealt(deal, k, ylo[r], yhi[r]), yf);
#pragma unroll
                    for (int c = 0; c < 2; c++) chroma_terms(uvf[2 * c], uvf[2 * c + 1], d.k, d.swap_rb, t0[c], tg[c], t2[c]);
                    const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + run_j0 + 4u * (uint32_t)(k * run_a + run_m);
                    color_store_row<O_F32_PLANAR, true>(yf, t0, tg, t2, d.k, (float *)out, pix, plane, 4, nt, none);
                }
            }
        }
        return;
    } else {
        // colour flavours
#pragma unroll
        for (int rc = 0; rc < 2; rc++) { // chroma output row rc of the tile = luma output rows 2 rc, 2 rc + 1
            float uvf[8]; // U0 V0 U1 V1 U2 V2 U3 V3
            bc_unpack4(clo[rc], uvf);
            bc_unpack4(chi[rc], uvf + 4);
            float t0[4], tg[4], t2[4];
#pragma unroll
            for (int c = 0; c < 4; c++) chroma_terms(uvf[2 * c], uvf[2 * c + 1], d.k, d.swap_rb, t0[c], tg[c], t2[c]);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int r = 2 * rc + rr;
                float yf[8];
                bc_unpack4(ylo[r], yf);
                bc_unpack4(yhi[r], yf + 4);
                const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
                if constexpr (OUT == O_U8_PLANAR || OUT == O_U8_MERGED) {
                    uint32_t pa[2], pb[2], pc[2];
#pragma unroll
                    for (int h = 0; h < 2; h++) color_pack_row_u8<OUT == O_U8_PLANAR>(yf + 4 * h, t0 + 2 * h, tg + 2 * h, t2 + 2 * h, d.k, pa[h], pb[h], pc[h]);
                    if constexpr (OUT == O_U8_PLANAR) {
                        bc_st8(out, pix, pa[0], pa[1], nt);
                        bc_st8(out + plane, pix, pb[0], pb[1], nt);
                        bc_st8(out + 2 * (size_t)plane, pix, pc[0], pc[1], nt);
                    } else if ((run_a & 1) == 0) { // 24 A bytes = 3 A / 2 chunks of 16
                        uint32_t *w = (uint32_t *)(run_lds + 24 * run_m);
                        w[0] = pa[0]; w[1] = pb[0]; w[2] = pc[0]; w[3] = pa[1]; w[4] = pb[1]; w[5] = pc[1];
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t row0 = 3u * (pix - (uint32_t)(BCR_COLS * run_m)); // first byte of the run in this row
                        const bq4 v0 = *(const bq4 *)(run_lds + 16 * run_m);
                        *(bq4 *)(out + row0 + 16u * (uint32_t)run_m) = v0;
                        if (2 * run_m < run_a) {
                            const bq4 v1 = *(const bq4 *)(run_lds + 16 * (run_a + run_m));
                            *(bq4 *)(out + row0 + 16u * (uint32_t)(run_a + run_m)) = v1;
                        }
                        __builtin_amdgcn_wave_barrier();
                    } else {
                        bc_st8(out, 3u * pix, pa[0], pb[0], 0);
                        bc_st8(out, 3u * pix + 8u, pc[0], pa[1], 0);
                        bc_st8(out, 3u * pix + 16u, pb[1], pc[1], 0);
                    }
                } else { // fp32 merged triples (RGB / BGR, or HSV of the normalised RGB)
                    float o[24];
                    {
                        float a[12], b[12];
                        bc_color4_f32<OUT == O_HSV_F32>(yf, t0, tg, t2, d.k, a);
                        bc_color4_f32<OUT == O_HSV_F32>(yf + 4, t0 + 2, tg + 2, t2 + 2, d.k, b);
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            o[k] = a[k];
                            o[12 + k] = b[k];
                        }
                    }
                    vf4 *w = (vf4 *)(run_lds + 96 * run_m);
#pragma unroll
                    for (int k = 0; k < 6; k++) w[k] = (vf4){ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t row0 = 12u * (pix - (uint32_t)(BCR_COLS * run_m)); // first byte of the run in this row
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const uint32_t off = 16u * (uint32_t)(k * run_a + run_m);
                        const vf4 v = *(const vf4 *)(run_lds + off);
                        st4o(out, row0 + off, v.x, v.y, v.z, v.w, nt);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
}

template <int P2> static hipError_t launch_bcr_k(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, hipStream_t stream) {
    switch (out) {
#define TSVPP_BCR(O) case O: hipLaunchKernelGGL((vpp_bicubic_r32_kernel<O, P2>), grid, block, 0, stream, d, t); break;
        TSVPP_BCR(O_U8_PLANAR) TSVPP_BCR(O_U8_MERGED) TSVPP_BCR(O_F32_PLANAR) TSVPP_BCR(O_F32_MERGED) TSVPP_BCR(O_NV12_U8) TSVPP_BCR(O_NV12_F32)
        TSVPP_BCR(O_Y800_U8) TSVPP_BCR(O_Y800_F32) TSVPP_BCR(O_HSV_F32)
#undef TSVPP_BCR
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// d.r32: 7 = BICUBIC at 3 : 2, 8 = BICUBIC at 2 : 1 (launch_fused)
hipError_t launch_bicubic_r32(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (d.r32 != 7 && d.r32 != 8) return hipErrorInvalidValue;
    if (info) {
        info->kernel = d.r32 == 7 ? "vpp_bicubic_r32_kernel<OUT,3:2>" : "vpp_bicubic_r32_kernel<OUT,2:1>";
        info->grid = (int)grid.x;
        info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : 16;
        return hipSuccess;
    }
    return d.r32 == 7 ? launch_bcr_k<3>(out, d, t, grid, block, stream) : launch_bcr_k<4>(out, d, t, grid, block, stream);
}

} // namespace tsvpp
