// vpp_bilinear_up2.hip -- BILINEAR at the exact ratio 1 : 2 on both axes (540p -> 1080p, 720p -> 1440p, 1080p -> 4K) as a streaming kernel: no LDS
// staging, no barrier, no coordinate arithmetic, no tables.
//
// Why (round 4): up-scales with uint8 outputs ran at 0.31-0.34 of the roofline on the LDS kernels whatever the interpolation
// (profiles/r04_upscale_u8_probe.txt): seven eighths of the bytes of such a launch are OUTPUT, and the LDS kernels' 4 x 2-pixel thread tiles spend
// their time in the colour / pack / store phase.  At 1 : 2 every weight is 1/4 or 3/4: a horizontal pair sum is one v_dot4_u32_u8 on the source dword
// as it was loaded, an output one multiply-add of two of them (vpp_bilinear_up2_core.h: the arithmetic, the geometry and the edge rules, shared with
// the host build the CPU suite checks against the oracle), and the thread tile is the 8 x 4 pixels of the streaming kernels' output side
// (vpp_r32_store.h: 8-byte planar stores, merged rows exchanged through LDS inside the wave).
//
// A thread converts 8 output columns x 4 output rows from 4 luma and 3 chroma source rows of ONE dword each + the dword before and after it.  In
// workgroups 64 threads wide a wave is one run of lanes that share their output rows: the dword before a thread's own IS its left neighbour's and the
// dword after it its right neighbour's -- one v_mov_b32_dpp wave_shr:1 / wave_shl:1 each instead of a load (only the first / last lane of a wave load
// theirs); narrower workgroups load all three (cf. vpp_bicubic_r32.hip, where the same trade was measured).
#include "vpp_device.h"
#include "vpp_bilinear_up2_core.h"
#include "vpp_r32_store.h"
#include "vpp_up2.h"

#pragma clang fp contract(off)

namespace tsvpp {

// (round 6: the wide / narrow choice is made ONCE around both planes' loads in the kernel and the neighbour dwords are zeroed before any load is issued -- see
// bcr_load_rows_wide, vpp_bicubic_r32.hip: behind the merge of a per-plane `if (!wide)` the compiler made every wave wait for its main loads before the last lane's.)
template <int NROWS>
__device__ __forceinline__ void u2k_load_rows_wide(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, bool run_first, bool run_last,
                                                   uint32_t (&ext)[NROWS][3], uint32_t (&nb)[NROWS][2]) {
    const uint32_t col = 4u * (uint32_t)q;
    uint32_t off[NROWS];
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        off[r] = (uint32_t)u2_row(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<1>(plane + off[r], &ext[r][1]);
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
// ... the shuffles, once the loads have landed.  v_mov_b32_dpp leaves a lane without a source lane (the wave's first for wave_shr, its last for wave_shl)
// on the `old` operand: the dword that lane loaded itself.  (A run that ends before lane 63 ends at the frame's right edge: the bytes its last lane
// would take from the dword after are replaced by u2_fix_rows.)
template <int NROWS> __device__ __forceinline__ void u2k_neighbours(uint32_t (&ext)[NROWS][3], const uint32_t (&nb)[NROWS][2], bool wide) {
    if (!wide) return;
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        ext[r][0] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][0], (int)ext[r][1], 0x138, 0xf, 0xf, false); // wave_shr:1: lane i <- lane i - 1
        ext[r][2] = (uint32_t)__builtin_amdgcn_update_dpp((int)nb[r][1], (int)ext[r][1], 0x130, 0xf, 0xf, false); // wave_shl:1: lane i <- lane i + 1
    }
}

template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_up2_kernel(const LaunchDesc d, const FrameTable t) {
    constexpr bool LUMA_ONLY = kLumaOnly<OUT>;
    const TileId id = decode_tile(d); // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = BCR_COLS * q, i0 = BCR_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    const bool first = q == 0, last = j0 + BCR_COLS >= d.dst_w;
    uint8_t *out = (uint8_t *)t.out[id.frame];

    // the run this lane belongs to: the lanes of the wave that share its output rows (A of them active)
    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - BCR_COLS * run_m)) / BCR_COLS);
    const bool run_first = run_m == 0, run_last = run_m == run_a - 1;
    // every load of the tile first, then the neighbour shuffles and the column edge fix-ups, then the arithmetic
    uint32_t ey[U2_NYR][3], ec[U2_NCR][3];
    uint32_t ny[U2_NYR][2] = {}, nc[U2_NCR][2] = {}; // (any defined value: the lanes that do not load theirs take their neighbour's dword)
    const bool wide = d.tx >= 64; // wave-uniform: a wave is one run
    if (wide) {
        u2k_load_rows_wide<U2_NYR>(t.y[id.frame], d.pitch_y, 2 * n4 - 1, d.src_h, q, first, last, run_first, run_last, ey, ny);
        if constexpr (!LUMA_ONLY) u2k_load_rows_wide<U2_NCR>(t.uv[id.frame], d.pitch_uv, n4 - 1, d.src_h >> 1, q, first, last, run_first, run_last, ec, nc);
    } else {
        u2_load_rows<U2_NYR>(t.y[id.frame], d.pitch_y, 2 * n4 - 1, d.src_h, q, first, last, ey);
        if constexpr (!LUMA_ONLY) u2_load_rows<U2_NCR>(t.uv[id.frame], d.pitch_uv, n4 - 1, d.src_h >> 1, q, first, last, ec);
    }
    __builtin_amdgcn_sched_barrier(0); // nothing is scheduled across this point: all loads of the tile are in flight together
    u2k_neighbours<U2_NYR>(ey, ny, wide);
    if constexpr (!LUMA_ONLY) u2k_neighbours<U2_NCR>(ec, nc, wide);
    u2_fix_rows<false, U2_NYR>(ey, first, last);
    if constexpr (!LUMA_ONLY) u2_fix_rows<true, U2_NCR>(ec, first, last);
    uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
    if constexpr (LUMA_ONLY) {
#pragma unroll
        for (int r = 0; r < U2_NCR; r++) ec[r][0] = ec[r][1] = ec[r][2] = 0u;
    }
    u2_tile<!LUMA_ONLY>(ey, ec, ylo, yhi, clo, chi);

    r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
}

// d.r32: 10 = BILINEAR at 1 : 2 (launch_fused)
hipError_t launch_bilinear_up2(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (d.r32 != 10) return hipErrorInvalidValue;
    if (info) {
        info->kernel = "vpp_bilinear_up2_kernel<OUT>";
        info->grid = (int)grid.x;
        info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : 16;
        return hipSuccess;
    }
    switch (out) {
#define TSVPP_U2(O) case O: TSVPP_LAUNCH((vpp_bilinear_up2_kernel<O>), grid, block, 0, stream, d, t); break;
        TSVPP_U2(O_U8_PLANAR) TSVPP_U2(O_U8_MERGED) TSVPP_U2(O_F32_PLANAR) TSVPP_U2(O_F32_MERGED) TSVPP_U2(O_NV12_U8) TSVPP_U2(O_NV12_F32)
        TSVPP_U2(O_Y800_U8) TSVPP_U2(O_Y800_F32) TSVPP_U2(O_HSV_F32)
#undef TSVPP_U2
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
