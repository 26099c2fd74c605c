// vpp_area_cols.hip -- large-ratio AREA with float weights, one output column per lane, source footprint staged in LDS.
//
// The column-per-lane kernel of vpp_kernels.hip (vpp_area_cols_kernel) reads its taps straight from global memory: lane =
// output column, so a wave's dword load covers one contiguous run of 64 * xr bytes -- but every lane still issues its own
// dword request, NK + 1 of them per source row, and each request occupies the vector cache for the whole 128-byte line it
// touches.  1080p -> 224 x 224 x 64 frames is 1.4 M such wave loads of ~9 lines each: the texture-addresser / L1 path is busy
// 69 % of the time (profiles/r01_area224_pmc.txt) while the VALU work of the tap chains is ~25 us and the HBM time ~30 us.
// With cached inputs AND outputs the kernel still takes 66 us (bench.py --alias 3, profiles/r02_alias_diag.txt).
//
// Here the workgroup's source footprint -- the (TH * yr + ry) rows x (64 * xr + rx) bytes its 64 x TH outputs tap -- goes
// to LDS once, by LDS-DMA in 16-byte chunks (stage_plane_dma: the compact layout of the other staged kernels), and the
// lanes read their windows from there as aligned dwords + v_alignbyte_b32.  Source rows that neighbouring output rows
// share (ry - yr of them per output row) are fetched once per tile instead of once per output row.  The arithmetic --
// weights, tap order, divisor accumulation, the final IEEE division -- is the global kernel's, i.e. the reference's
// (src/Resize.cu:160-178 as compiled: colorSum = fma(data, weight, colorSum), divide += weight).
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef float vf4a4 __attribute__((ext_vector_type(4), aligned(4)));

// NK + 1 aligned dwords from LDS byte address a (any alignment); the caller shifts with v_alignbyte_b32, whose shift operand
// is (a & 3) (the instruction ignores the upper bits: tools/dbg_cvt.hip)
template <int N> __device__ __forceinline__ void lds_window(const uint8_t *lds, int a, uint32_t (&dw)[N]) {
    const uint32_t *q = (const uint32_t *)(lds + (a & ~3));
#pragma unroll
    for (int k = 0; k < N; k++) dw[k] = q[k];
}

// Workgroup: TH = 32 -> 4 waves, each samples 8 luma rows and then 4 chroma rows (balanced); TH = 8 -> SIX waves: waves 0-3
// sample one luma row pair each, waves 4-5 the four chroma rows at the same time (with four waves two of them would run the
// chroma chains after their luma chains while the other two idle: the tile's latency is what bounds this kernel).
template <int TH> constexpr int cols_threads() { return TH == 8 ? 384 : (TH == 4 ? 192 : 256); }
// DIVTAB: the divisor of every (column pattern, row pattern) comes from a host-built table (LaunchDesc::area_div, built in
// tsvpp_api.cpp with the same fp32 products and the same summation order) instead of one more add per tap and lane.
template <int NK, int TH, bool DIVTAB, int OUT>
__global__ __launch_bounds__(cols_threads<TH>()) void vpp_area_cols_lds_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr int TW = 64; // output tile of the workgroup: 64 columns x TH rows
    constexpr int NTHREADS = cols_threads<TH>();
    static_assert(TH == 32 || TH == 8 || TH == 4, "tile height");
    __shared__ __attribute__((aligned(16))) float yt[TH][TW];
    __shared__ __attribute__((aligned(16))) f2 uvt[TH / 2][TW / 2];
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int j_first = id.tx * TW, i_first = id.ty * TH;
    const int j_last = min(j_first + TW, d.dst_w) - 1, i_last = min(i_first + TH, d.dst_h) - 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cw = d.dst_w >> 1, chh = d.dst_h >> 1;

    // footprint of the tile, from the same coordinate expressions the lanes use below
    const int xlo = (int)(d.xr * (float)j_first), ylo = (int)(d.yr * (float)i_first);
    const int span_y = min((int)(d.xr * (float)j_last) + d.rx, d.src_w) - xlo;
    const int ny = min((int)(d.yr * (float)i_last) + d.ry, d.src_h) - ylo;
    const int cj_first = j_first >> 1, ci_first = i_first >> 1;
    const int cj_last = min(cj_first + TW / 2, cw) - 1, ci_last = min(ci_first + TH / 2, chh) - 1;
    const int cxlo = 2 * (int)(d.xr * (float)cj_first), cylo = (int)(d.yr * (float)ci_first);
    const int span_uv = min(2 * ((int)(d.xr * (float)cj_last) + d.rx), d.src_w) - cxlo;
    const int nuv = d.luma_only ? 0 : min((int)(d.yr * (float)ci_last) + d.ry, d.src_h >> 1) - cylo;

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, ylo, xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, cylo, cxlo, d.lds_cpr_uv, auv);
    stage_plane_dma(lds_y, ay, py, d.pitch_y, min(ny, d.lds_rows_y), min(span_y, d.lds_span_y), d.lds_magic_y, NTHREADS);
    stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, min(nuv, d.lds_rows_uv), min(span_uv, d.lds_span_uv), d.lds_magic_uv, NTHREADS);

    // per-lane constants while the chunks are in flight
    const int j = min(j_first + lane, d.dst_w - 1);
    const int xrel = (int)(d.xr * (float)j) - xlo;
    vf4 wx[NK];
    {
        const float *wxrow = d.patx4 + (j % d.nx) * 4 * NK;
#pragma unroll
        for (int k = 0; k < NK; k++) wx[k] = *(const vf4a4 *)(wxrow + 4 * k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA chunks have landed
    __syncthreads();

    if (TH == 32 || wave < TH / 2) { // luma: lane = column, two output rows per lane on float pairs; rows past the frame are computed on the last valid one
        for (int rp = 0; rp < (TH == 32 ? 4 : 1); rp++) {
            const int r0 = (TH == 32 ? 8 : 2) * wave + 2 * rp;
            const int iA = min(i_first + r0, d.dst_h - 1), iB = min(i_first + r0 + 1, d.dst_h - 1);
            const float *wyA = d.paty4 + (iA % d.ny) * 4 * d.nky, *wyB = d.paty4 + (iB % d.ny) * 4 * d.nky;
            const int yA = (int)(d.yr * (float)iA) - ylo, yB = (int)(d.yr * (float)iB) - ylo;
            f2 acc = { 0.0f, 0.0f }, div = { 0.0f, 0.0f };
            if constexpr (DIVTAB) div = (f2){ d.area_div[(j % d.nx) * d.ny + iA % d.ny], d.area_div[(j % d.nx) * d.ny + iB % d.ny] };
            auto row_addr = [&](int r) { // uniform: LDS row base of staged row r (clamped: a tap past the plane has weight 0)
                r = min(r, ny - 1);
                return r * py.lp + ((py.m0 + r * py.pm) & 15) + xrel;
            };
            uint32_t da[NK + 1], db[NK + 1];
            int aa = row_addr(yA), ab = row_addr(yB);
            lds_window<NK + 1>(lds_y, aa, da);
            lds_window<NK + 1>(lds_y, ab, db);
            for (int a = 0; a < d.ry; a++) {
                const f2 wy = { wyA[a], wyB[a] };
                uint32_t ca[NK + 1], cb[NK + 1];
                const int sa = aa, sb = ab;
#pragma unroll
                for (int k = 0; k <= NK; k++) { ca[k] = da[k]; cb[k] = db[k]; }
                // the next row's windows are requested before this row's taps are consumed
                aa = row_addr(yA + a + 1);
                ab = row_addr(yB + a + 1);
                lds_window<NK + 1>(lds_y, aa, da);
                lds_window<NK + 1>(lds_y, ab, db);
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t va = __builtin_amdgcn_alignbyte(ca[k + 1], ca[k], (uint32_t)sa), vb = __builtin_amdgcn_alignbyte(cb[k + 1], cb[k], (uint32_t)sb);
                    const float wk[4] = { wx[k].x, wx[k].y, wx[k].z, wx[k].w };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const f2 wgt = (f2){ wk[b], wk[b] } * wy;
                        if constexpr (!DIVTAB) div = div + wgt;
                        acc = __builtin_elementwise_fma((f2){ (float)((va >> (8 * b)) & 255), (float)((vb >> (8 * b)) & 255) }, wgt, acc);
                    }
                }
            }
            yt[r0][lane] = __builtin_truncf(acc.x / div.x);
            yt[r0 + 1][lane] = __builtin_truncf(acc.y / div.y);
        }
    }
    if (!d.luma_only && (TH == 32 || wave >= TH / 2)) { // chroma: lanes 0-31 / 32-63 = the 32 chroma columns of two chroma rows; (U, V) as a pair
        const int cj = min(cj_first + (lane & 31), cw - 1);
        const int cxrel = 2 * (int)(d.xr * (float)cj) - cxlo;
        const float *wxrow = d.patx4 + (cj % d.nx) * 4 * NK;
        vf4 cwx[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) cwx[k] = *(const vf4a4 *)(wxrow + 4 * k);
        for (int q = 0; q < (TH == 32 ? 2 : 1); q++) {
            const int cr = (TH == 32 ? 4 * wave + 2 * q : 2 * (wave - TH / 2)) + (lane >> 5);
            const int ci = min(ci_first + cr, chh - 1);
            const float *wyrow = d.paty4 + (ci % d.ny) * 4 * d.nky;
            const int y0 = (int)(d.yr * (float)ci) - cylo;
            f2 acc = { 0.0f, 0.0f };
            float div = 0.0f;
            if constexpr (DIVTAB) div = d.area_div[(cj % d.nx) * d.ny + ci % d.ny];
            auto row_addr = [&](int r) { // per half-wave
                r = min(r, nuv - 1);
                return r * puv.lp + ((puv.m0 + r * puv.pm) & 15) + cxrel;
            };
            uint32_t dn[2 * NK + 1];
            int an = row_addr(y0);
            lds_window<2 * NK + 1>(lds_uv, an, dn);
            for (int a = 0; a < d.ry; a++) {
                const float wy = wyrow[a];
                uint32_t dw[2 * NK + 1];
                const int sh = an;
#pragma unroll
                for (int k = 0; k <= 2 * NK; k++) dw[k] = dn[k];
                an = row_addr(y0 + a + 1);
                lds_window<2 * NK + 1>(lds_uv, an, dn);
#pragma unroll
                for (int k = 0; k < NK; k++) {
                    const uint32_t v0 = __builtin_amdgcn_alignbyte(dw[2 * k + 1], dw[2 * k], (uint32_t)sh);     // U0 V0 U1 V1
                    const uint32_t v1 = __builtin_amdgcn_alignbyte(dw[2 * k + 2], dw[2 * k + 1], (uint32_t)sh); // U2 V2 U3 V3
                    const float wv[4] = { cwx[k].x, cwx[k].y, cwx[k].z, cwx[k].w };
                    const uint32_t vv[2] = { v0, v1 };
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t qq = vv[b >> 1] >> (16 * (b & 1));
                        const float wgt = wv[b] * wy;
                        if constexpr (!DIVTAB) div = div + wgt;
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

template <int NK, int TH, bool DIVTAB>
static hipError_t launch_cols_lds_nt(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    switch (out) {
#define TSVPP_COLS(O) case O: hipLaunchKernelGGL((vpp_area_cols_lds_kernel<NK, TH, DIVTAB, O>), grid, block, lds, stream, d, t); break;
        TSVPP_COLS(O_U8_PLANAR) TSVPP_COLS(O_U8_MERGED) TSVPP_COLS(O_F32_PLANAR) TSVPP_COLS(O_F32_MERGED) TSVPP_COLS(O_NV12_U8)
        TSVPP_COLS(O_NV12_F32) TSVPP_COLS(O_Y800_U8) TSVPP_COLS(O_Y800_F32) TSVPP_COLS(O_HSV_F32)
#undef TSVPP_COLS
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_area_cols_lds(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    const bool four = d.area_cols_rows == 4;
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block(four ? 192u : 384u);
    if (d.nkx < 1 || d.nkx > 3 || (d.area_cols_rows != 8 && !four)) return hipErrorInvalidValue;
    const bool divtab = d.area_div != nullptr;
    if (info) {
        static const char *const names[2][2][3] = {
            { { "vpp_area_cols_lds_kernel<1,8,0,OUT>", "vpp_area_cols_lds_kernel<2,8,0,OUT>", "vpp_area_cols_lds_kernel<3,8,0,OUT>" },
              { "vpp_area_cols_lds_kernel<1,8,1,OUT>", "vpp_area_cols_lds_kernel<2,8,1,OUT>", "vpp_area_cols_lds_kernel<3,8,1,OUT>" } },
            { { "vpp_area_cols_lds_kernel<1,4,0,OUT>", "vpp_area_cols_lds_kernel<2,4,0,OUT>", "vpp_area_cols_lds_kernel<3,4,0,OUT>" },
              { "vpp_area_cols_lds_kernel<1,4,1,OUT>", "vpp_area_cols_lds_kernel<2,4,1,OUT>", "vpp_area_cols_lds_kernel<3,4,1,OUT>" } } };
        info->kernel = names[four ? 1 : 0][divtab ? 1 : 0][d.nkx - 1];
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    switch ((four ? 8 : 0) + d.nkx * 2 + (divtab ? 1 : 0)) {
    case 2: return launch_cols_lds_nt<1, 8, false>(out, d, t, grid, block, lds_bytes, stream);
    case 3: return launch_cols_lds_nt<1, 8, true>(out, d, t, grid, block, lds_bytes, stream);
    case 4: return launch_cols_lds_nt<2, 8, false>(out, d, t, grid, block, lds_bytes, stream);
    case 5: return launch_cols_lds_nt<2, 8, true>(out, d, t, grid, block, lds_bytes, stream);
    case 6: return launch_cols_lds_nt<3, 8, false>(out, d, t, grid, block, lds_bytes, stream);
    case 7: return launch_cols_lds_nt<3, 8, true>(out, d, t, grid, block, lds_bytes, stream);
    case 10: return launch_cols_lds_nt<1, 4, false>(out, d, t, grid, block, lds_bytes, stream);
    case 11: return launch_cols_lds_nt<1, 4, true>(out, d, t, grid, block, lds_bytes, stream);
    case 12: return launch_cols_lds_nt<2, 4, false>(out, d, t, grid, block, lds_bytes, stream);
    case 13: return launch_cols_lds_nt<2, 4, true>(out, d, t, grid, block, lds_bytes, stream);
    case 14: return launch_cols_lds_nt<3, 4, false>(out, d, t, grid, block, lds_bytes, stream);
    default: return launch_cols_lds_nt<3, 4, true>(out, d, t, grid, block, lds_bytes, stream);
    }
}

} // namespace tsvpp
