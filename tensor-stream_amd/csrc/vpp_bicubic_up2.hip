// vpp_bicubic_up2.hip -- BICUBIC at the exact ratio 1 : 2 on both axes (540p -> 1080p, 720p -> 1440p, 1080p -> 4K) as a streaming kernel: no LDS
// staging, no barrier, no coordinate arithmetic, no tables -- the up-scale twin of vpp_bicubic_r32.hip on the loading scheme of vpp_bilinear_up2.hip.
//
// At 1 : 2 every weight is 1/4 or 3/4: Keys' coefficients are the byte coefficients of the 3 : 2 kernel, and a 4-tap sum is two to four
// v_dot4_u32_u8 on the source dwords as they were loaded (vpp_bicubic_up2_core.h: the arithmetic, the geometry and the up-scale's own edge rules,
// shared with the host build the CPU suite checks against the oracle).  A thread converts 8 output columns x 4 output rows from 6 luma and 5 chroma
// source rows of ONE dword each + the dword before and after it (workgroups 64 threads wide: from the neighbour lanes by v_mov_b32_dpp), through the
// shared 8 x 4 output side (vpp_r32_store.h).
#include "vpp_device.h"
#include "vpp_bicubic_up2_core.h"
#include "vpp_r32_store.h"
#include "vpp_up2.h"

#pragma clang fp contract(off)

namespace tsvpp {

// (the loader and the neighbour shuffles of vpp_bilinear_up2.hip, for this kernel's row counts)
template <int NROWS>
__device__ __forceinline__ void b2k_load_rows(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, bool wide, bool run_first,
                                              bool run_last, uint32_t (&ext)[NROWS][3], uint32_t (&nb)[NROWS][2]) {
    if (!wide) {
        u2_load_rows<NROWS>(plane, pitch, row0, plane_rows, q, first, last, ext);
        return;
    }
    const uint32_t col = 4u * (uint32_t)q;
    uint32_t off[NROWS];
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        off[r] = (uint32_t)u2_row(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<1>(plane + off[r], &ext[r][1]);
        nb[r][0] = 0u;
        nb[r][1] = 0u;
    }
    if (run_first && !first) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) bc_ld<1>(plane + (off[r] - 4u), &nb[r][0]);
    }
    if (run_last && !last) {
#pragma unroll
        for (int r = 0; r < NROWS; r++) bc_ld<1>(plane + (off[r] + 4u), &nb[r][1]);
    }
}
template <int NROWS> __device__ __forceinline__ void b2k_neighbours(uint32_t (&ext)[NROWS][3], const uint32_t (&nb)[NROWS][2], bool wide) {
    if (!wide) return;
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][0], (int)ext[r][1], 0x138, 0xf, 0xf, false); // wave_shr:1: lane i <- lane i - 1
        ext[r][2] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][1], (int)ext[r][1], 0x130, 0xf, 0xf, false); // wave_shl:1: lane i <- lane i + 1
    }
}

template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bicubic_up2_kernel(const LaunchDesc d, const FrameTable t) {
    constexpr bool LUMA_ONLY = kLumaOnly<OUT>;
    const TileId id = decode_tile(d); // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = BCR_COLS * q, i0 = BCR_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    const bool first = q == 0, last = j0 + BCR_COLS >= d.dst_w;
    const int ntiles = d.dst_h >> 2;
    const bool first_row = n4 == 0, last_row = n4 == ntiles - 1, before_last_row = n4 == ntiles - 2;
    uint8_t *out = (uint8_t *)t.out[id.frame];

    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - BCR_COLS * run_m)) / BCR_COLS);
    const bool run_first = run_m == 0, run_last = run_m == run_a - 1;
    uint32_t ey[B2_NYR][3], xay[B2_NYR][2], xby[B2_NYR], ec[B2_NCR][3], xac[B2_NCR][2], xbc[B2_NCR], ny[B2_NYR][2], nc[B2_NCR][2];
    const bool wide = d.tx >= 64; // wave-uniform: a wave is one run
    b2k_load_rows<B2_NYR>(t.y[id.frame], d.pitch_y, 2 * n4 - 2, d.src_h, q, first, last, wide, run_first, run_last, ey, ny);
    if constexpr (!LUMA_ONLY) b2k_load_rows<B2_NCR>(t.uv[id.frame], d.pitch_uv, n4 - 2, d.src_h >> 1, q, first, last, wide, run_first, run_last, ec, nc);
    __builtin_amdgcn_sched_barrier(0); // nothing is scheduled across this point: all loads of the tile are in flight together
    b2k_neighbours<B2_NYR>(ey, ny, wide);
    if constexpr (!LUMA_ONLY) b2k_neighbours<B2_NCR>(ec, nc, wide);
    b2_fix_rows<false, B2_NYR>(ey, xay, xby, first, last);
    if constexpr (!LUMA_ONLY) b2_fix_rows<true, B2_NCR>(ec, xac, xbc, first, last);
    if constexpr (LUMA_ONLY) {
#pragma unroll
        for (int r = 0; r < B2_NCR; r++) ec[r][0] = ec[r][1] = ec[r][2] = xac[r][0] = xac[r][1] = xbc[r] = 0u;
    }
    uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
    b2_tile<!LUMA_ONLY>(ey, xay, xby, ec, xac, xbc, first, first_row, last_row, before_last_row, ylo, yhi, clo, chi);

    r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
}

// d.r32: 9 = BICUBIC at 1 : 2 (launch_fused)
hipError_t launch_bicubic_up2(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (d.r32 != 9) return hipErrorInvalidValue;
    if (info) {
        info->kernel = "vpp_bicubic_up2_kernel<OUT>";
        info->grid = (int)grid.x;
        info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : 16;
        return hipSuccess;
    }
    switch (out) {
#define TSVPP_B2(O) case O: hipLaunchKernelGGL((vpp_bicubic_up2_kernel<O>), grid, block, 0, stream, d, t); break;
        TSVPP_B2(O_U8_PLANAR) TSVPP_B2(O_U8_MERGED) TSVPP_B2(O_F32_PLANAR) TSVPP_B2(O_F32_MERGED) TSVPP_B2(O_NV12_U8) TSVPP_B2(O_NV12_F32)
        TSVPP_B2(O_Y800_U8) TSVPP_B2(O_Y800_F32) TSVPP_B2(O_HSV_F32)
#undef TSVPP_B2
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
