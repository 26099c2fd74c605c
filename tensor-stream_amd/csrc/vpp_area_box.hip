// vpp_area_box.hip -- AREA down-scale at INTEGER horizontal ratios from 4 up (4K -> 640x360 = 6 x 6, BASELINE config C5;
// 4K -> 960x540; 1080p -> 480x270 ...): a pure streaming reduction, every source byte is needed exactly once.
//
// The reference sums float(tap) * (wx * wy) over a ceil(xr) x ceil(yr) box and divides by the weight sum
// (src/Resize.cu:160-178); with all-ones horizontal weights and dyadic vertical ones every partial sum is an exact
// integer, so the value is (int)(SUM / S) with SUM taken in integers (the argument is at vpp_area_dyadic_kernel).
//
// vpp_area_direct_kernel (the general large-ratio kernel) gives each of a thread's four columns its own window: RX + 3
// bytes loaded as 2-3 dwords per column and row, shifted into place with v_alignbyte -- at ratio 6 that is four
// 12-byte loads for 24 useful bytes per source row, 10x the load instructions of the staged kernels for 4x the bytes
// (profiles/r01_c5_pmc.txt) inside a loop whose trip count the compiler does not know: ~4 loads in flight per wave, 58 %
// of all wave cycles waiting on memory.  Here the four columns of a thread tile are ONE contiguous, dword-aligned run of
// 4 RX bytes = RX dwords (the row pitch and the plane pointers are multiples of 4, see launch_fused; the column
// boundaries fall at compile-time byte positions): one or two vector loads per row, no shifts, the column sums come
// from v_dot4_u32_u8 with constant 0/1 byte masks -- accumulated across the rows of the box through the
// instruction's accumulator operand when the vertical weights are all one (SQUARE: yr == xr, the rows unroll and all
// loads of a thread tile are in flight together).  Chroma: the two (U, V) pair columns of the tile are the same RX dwords
// of the UV plane, U on the even and V on the odd bytes of the masks.
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

// N consecutive dwords in as few loads as possible (dwordx4 / x3 / x2 / x1)
template <int N>
__device__ __forceinline__ void load_dwords(const uint8_t *base, uint32_t off, uint32_t (&dw)[N]) {
    const uint32_t *p = (const uint32_t *)(base + off);
    constexpr int Q = N / 4, R = N % 4;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const u32x4a4 v = *(const u32x4a4 *)(p + 4 * q);
        dw[4 * q] = v.x; dw[4 * q + 1] = v.y; dw[4 * q + 2] = v.z; dw[4 * q + 3] = v.w;
    }
    if constexpr (R == 3) {
        const u32x3a4 v = *(const u32x3a4 *)(p + 4 * Q);
        dw[4 * Q] = v.x; dw[4 * Q + 1] = v.y; dw[4 * Q + 2] = v.z;
    } else if constexpr (R == 2) {
        const u32x2a4 v = *(const u32x2a4 *)(p + 4 * Q);
        dw[4 * Q] = v.x; dw[4 * Q + 1] = v.y;
    } else if constexpr (R == 1) {
        dw[4 * Q] = p[4 * Q];
    }
}

// 0/1 byte weights of dword k for the bytes of [lo, hi) with the given parity (-1: all, 0: even, 1: odd byte positions)
constexpr uint32_t box_mask(int lo, int hi, int k, int parity) {
    uint32_t m = 0;
    for (int b = 0; b < 4; b++) {
        const int B = 4 * k + b;
        if (B >= lo && B < hi && (parity < 0 || (B & 1) == parity)) m |= 1u << (8 * b);
    }
    return m;
}
// Column sums of one or two rows at once.  acc[s][c] += bytes [lo_c, hi_c) of row s.  The loops run dword-major with the
// two rows innermost: consecutive v_dot4 then never feed each other (a dependent v_dot4 pair costs a wait state; with
// one accumulator per column the compiler's chains came out as dot4 / s_nop / dot4 / s_nop ...).
//   luma:   column c = bytes [RX c, RX (c + 1)),           all byte positions
//   chroma: column c = bytes [2 RX c, 2 RX (c + 1)), U on the even and V on the odd positions (acc index 2 c + parity)
template <int RX, int NROWS, bool CHROMA>
__device__ __forceinline__ void box_rows(const uint32_t (&dw)[2][RX], uint32_t (&acc)[2][4]) {
#pragma unroll
    for (int k = 0; k < RX; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = CHROMA ? (q >> 1) : q;
            const uint32_t m = CHROMA ? box_mask(2 * RX * c, 2 * RX * (c + 1), k, q & 1) : box_mask(RX * c, RX * (c + 1), k, -1);
            if (m != 0u) {
#pragma unroll
                for (int sidx = 0; sidx < NROWS; sidx++) acc[sidx][q] = __builtin_amdgcn_udot4(dw[sidx][k], m, acc[sidx][q], false);
            }
        }
}

// all RX rows of a SQUARE box (vertical weights all one), two rows per step on separate accumulators
template <int RX, bool CHROMA>
__device__ __forceinline__ void box_square(const uint8_t *plane, uint32_t first, uint32_t pitch, uint32_t (&sum)[4]) {
    uint32_t acc[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
    for (int a = 0; a + 1 < RX; a += 2) {
        uint32_t dw[2][RX];
        load_dwords<RX>(plane, first + (uint32_t)a * pitch, dw[0]);
        load_dwords<RX>(plane, first + (uint32_t)(a + 1) * pitch, dw[1]);
        box_rows<RX, 2, CHROMA>(dw, acc);
    }
    if constexpr (RX & 1) {
        uint32_t dw[2][RX];
        load_dwords<RX>(plane, first + (uint32_t)(RX - 1) * pitch, dw[0]);
        box_rows<RX, 1, CHROMA>(dw, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) sum[q] = acc[0][q] + acc[1][q];
}

template <int RX, bool SQUARE, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_box_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = (id.tx * d.tx + lx) * PXW, i0 = (id.ty * d.ty + ly) * PXH;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const uint8_t *Y = t.y[id.frame], *UV = t.uv[id.frame];
    const uint32_t xoff = (uint32_t)(RX * j0); // first byte of the tile's run, in both planes (chroma: 2 * RX * (j0 / 2))
    constexpr float kRcpSquare = 1.0f / (float)(RX * RX); // the host's area_rcp for S = RX * RX

    float Uf[2], Vf[2], Yf[PXH][PXW];
    if constexpr (!kLumaOnly<OUT>) { // chroma row i0 / 2: U on even bytes, V on odd bytes of the same dwords
        const int ci = i0 >> 1;
        uint32_t s4[4] = { 0, 0, 0, 0 }; // U0 V0 U1 V1
        int sy = RX;
        if constexpr (SQUARE) {
            box_square<RX, true>(UV, (uint32_t)(RX * ci) * (uint32_t)d.pitch_uv + xoff, (uint32_t)d.pitch_uv, s4);
        } else {
            const AreaQRow qy = d.qy[ci % d.ny];
            const int y0 = (int)(d.yr * (float)ci);
            sy = qy.sum;
            for (int a = 0; a < d.ry; a++) {
                uint32_t dw[2][RX], rs[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
                load_dwords<RX>(UV, (uint32_t)(y0 + a) * (uint32_t)d.pitch_uv + xoff, dw[0]);
                box_rows<RX, 1, true>(dw, rs);
                const uint32_t wy = (qy.w[a >> 2] >> (8 * (a & 3))) & 255u;
#pragma unroll
                for (int q = 0; q < 4; q++) s4[q] += wy * rs[0][q];
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            Uf[c] = SQUARE ? area_quot(s4[2 * c], RX, RX, kRcpSquare) : area_quot(s4[2 * c], RX, sy, d.area_rcp);
            Vf[c] = SQUARE ? area_quot(s4[2 * c + 1], RX, RX, kRcpSquare) : area_quot(s4[2 * c + 1], RX, sy, d.area_rcp);
        }
    } else {
        Uf[0] = Uf[1] = Vf[0] = Vf[1] = 128.0f;
    }
#pragma unroll
    for (int r = 0; r < PXH; r++) {
        uint32_t s4[4] = { 0, 0, 0, 0 };
        int sy = RX;
        if constexpr (SQUARE) {
            box_square<RX, false>(Y, (uint32_t)(RX * (i0 + r)) * (uint32_t)d.pitch_y + xoff, (uint32_t)d.pitch_y, s4);
        } else {
            const AreaQRow qy = d.qy[(i0 + r) % d.ny];
            const int y0 = (int)(d.yr * (float)(i0 + r));
            sy = qy.sum;
            for (int a = 0; a < d.ry; a++) {
                uint32_t dw[2][RX], rs[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
                load_dwords<RX>(Y, (uint32_t)(y0 + a) * (uint32_t)d.pitch_y + xoff, dw[0]);
                box_rows<RX, 1, false>(dw, rs);
                const uint32_t wy = (qy.w[a >> 2] >> (8 * (a & 3))) & 255u;
#pragma unroll
                for (int q = 0; q < 4; q++) s4[q] += wy * rs[0][q];
            }
        }
#pragma unroll
        for (int c = 0; c < PXW; c++) Yf[r][c] = SQUARE ? area_quot(s4[c], RX, RX, kRcpSquare) : area_quot(s4[c], RX, sy, d.area_rcp);
    }
    color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
}

template <int RX, bool SQUARE>
static hipError_t launch_box_rs(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, hipStream_t stream) {
    switch (out) {
#define TSVPP_BOX(O) case O: TSVPP_LAUNCH((vpp_area_box_kernel<RX, SQUARE, O>), grid, block, 0, stream, d, t); break;
        TSVPP_BOX(O_U8_PLANAR) TSVPP_BOX(O_U8_MERGED) TSVPP_BOX(O_F32_PLANAR) TSVPP_BOX(O_F32_MERGED) TSVPP_BOX(O_NV12_U8)
        TSVPP_BOX(O_NV12_F32) TSVPP_BOX(O_Y800_U8) TSVPP_BOX(O_Y800_F32) TSVPP_BOX(O_HSV_F32)
#undef TSVPP_BOX
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_area_box(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    const bool square = d.box_ry == d.box_rx;
    if (info) {
        static const char *const names[2][7] = { { "vpp_area_box_kernel<2,0,OUT>", "vpp_area_box_kernel<3,0,OUT>", "vpp_area_box_kernel<4,0,OUT>", "vpp_area_box_kernel<5,0,OUT>",
                                                   "vpp_area_box_kernel<6,0,OUT>", "vpp_area_box_kernel<7,0,OUT>", "vpp_area_box_kernel<8,0,OUT>" },
                                                 { "vpp_area_box_kernel<2,1,OUT>", "vpp_area_box_kernel<3,1,OUT>", "vpp_area_box_kernel<4,1,OUT>", "vpp_area_box_kernel<5,1,OUT>",
                                                   "vpp_area_box_kernel<6,1,OUT>", "vpp_area_box_kernel<7,1,OUT>", "vpp_area_box_kernel<8,1,OUT>" } };
        if (d.box_rx < 2 || d.box_rx > 8) return hipErrorInvalidValue;
        info->kernel = names[square ? 1 : 0][d.box_rx - 2];
        info->grid = (int)grid.x;
        info->lds_bytes = 0;
        return hipSuccess;
    }
    switch (d.box_rx * 2 + (square ? 1 : 0)) {
    case 4: return launch_box_rs<2, false>(out, d, t, grid, block, stream);
    case 5: return launch_box_rs<2, true>(out, d, t, grid, block, stream);
    case 6: return launch_box_rs<3, false>(out, d, t, grid, block, stream);
    case 7: return launch_box_rs<3, true>(out, d, t, grid, block, stream);
    case 8: return launch_box_rs<4, false>(out, d, t, grid, block, stream);
    case 9: return launch_box_rs<4, true>(out, d, t, grid, block, stream);
    case 10: return launch_box_rs<5, false>(out, d, t, grid, block, stream);
    case 11: return launch_box_rs<5, true>(out, d, t, grid, block, stream);
    case 12: return launch_box_rs<6, false>(out, d, t, grid, block, stream);
    case 13: return launch_box_rs<6, true>(out, d, t, grid, block, stream);
    case 14: return launch_box_rs<7, false>(out, d, t, grid, block, stream);
    case 15: return launch_box_rs<7, true>(out, d, t, grid, block, stream);
    case 16: return launch_box_rs<8, false>(out, d, t, grid, block, stream);
    case 17: return launch_box_rs<8, true>(out, d, t, grid, block, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace tsvpp
