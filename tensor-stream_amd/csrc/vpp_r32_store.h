// vpp_r32_store.h -- the output side of the streaming kernels whose thread tile is 8 output columns x 4 output rows (vpp_bicubic_r32.hip, and the fp32
// flavours of vpp_bilinear_r32.hip): from the RESIZED BYTES of a tile -- ylo / yhi[r] = the 8 luma bytes of output row r, clo / chi[rc] = U0 V0 U1 V1 |
// U2 V2 U3 V3 of chroma output row rc -- through the colour back end to stores in which every store instruction of a wave writes whole lines,
// for every output flavour:
//   uint8 planar     8-byte stores (512 contiguous bytes per wave and plane row);
//   uint8 / fp32 merged (RGB / BGR triples, HSV)   a lane's row piece (24 / 96 bytes) is exchanged through LDS inside the wave so that each store instruction
//                    writes one contiguous span (cf. MergedRun, vpp_device.h);
//   fp32 planar (colour planes, NV12, Y800)   the resized bytes are dealt out between the lanes of the wave by shuffles BEFORE the colour conversion
//                    (BcDeal below), so that a lane converts two groups of four columns a run-length apart and 16-byte stores cover whole lines;
//   uint8 NV12 / Y800   the packed bytes as they are.
#pragma once
#include "vpp_device.h"

namespace tsvpp {

typedef uint32_t bq2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t bq4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void bc_st8(uint8_t *base, uint32_t off, uint32_t lo, uint32_t hi, int nt) {
    const bq2 v = { lo, hi };
    if (nt) st8_nt(base, off, lo, hi, nt); // (inline asm: see st8_nt, vpp_device.h -- the builtin's hint did not survive)
    else *(bq2 *)(base + off) = v;
}
__device__ __forceinline__ void bc_st16(uint8_t *base, uint32_t off, bq4 v, int nt) {
    if (nt) st16_nt(base, off, (nt_u32x4){ v.x, v.y, v.z, v.w }, nt);
    else *(bq4 *)(base + off) = v;
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

// i0 / j0: the tile's first output row / column; run_m / run_a: this lane's index in its run (the lanes of the wave that share its output rows) and the
// run's active lanes.
template <int OUT>
__device__ __forceinline__ void r32_store_tile(const LaunchDesc &d, uint8_t *out, const uint32_t (&ylo)[4], const uint32_t (&yhi)[4], const uint32_t (&clo)[2],
                                               const uint32_t (&chi)[2], int i0, int j0, int run_m, int run_a) {
    const int nt = d.nt_stores;
    const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;
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
                    for (int c = 0; c < 2; c++) chroma_terms(uvf[2 * c], uvf[2 * c + 1], d.k, d.swap_rb, d.color_g, t0[c], tg[c], t2[c]);
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
            for (int c = 0; c < 4; c++) chroma_terms(uvf[2 * c], uvf[2 * c + 1], d.k, d.swap_rb, d.color_g, t0[c], tg[c], t2[c]);
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
                        // (round 5) non-temporal like every other whole-line store of the library: the exchanged 16-byte stores were plain until then, and that alone
                        // cost the uint8 merged flavours of the streaming kernels 5..27 % (profiles/r05_prn_nt_variants.txt, r05_u8_merged_nt_ab.txt)
                        bc_st16(out, row0 + 16u * (uint32_t)run_m, v0, nt);
                        if (2 * run_m < run_a) {
                            const bq4 v1 = *(const bq4 *)(run_lds + 16 * (run_a + run_m));
                            bc_st16(out, row0 + 16u * (uint32_t)(run_a + run_m), v1, nt);
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

} // namespace tsvpp
