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
#include "vpp_r32_store.h"

#pragma clang fp contract(off)

namespace tsvpp {

// The extended rows of a thread tile, neighbours by wave shuffle.  Measured (profiles/r04_bicubic_r32_pmc.txt): with three loads per row (the run + the
// dword before + the dword after: bc_load_rows) the fp32 launch keeps the texture addresser 79 % busy -- a wave's 39 loads and 24 stores cost ~63 TA
// cycles per instruction: the kernel is bound by the ISSUE of memory instructions, not by bytes.  The dword before a thread's run IS the last dword
// of its left neighbour's run and the dword after it the first of its right neighbour's: one v_mov_b32_dpp wave_shr:1 / wave_shl:1 each instead of a
// load (fp32 planar 1080p -> 720p 0.64 -> 0.71, 4K -> 1080p 0.68 -> 0.73: profiles/r04_bicubic_r32_ab.txt).  Only the first / last lane of a wave
// -- their neighbour is in another wave -- load theirs: two single-lane loads per row.  For workgroups 64 threads wide (a wave = one run of lanes
// that share their output rows); narrower workgroups (uint8 outputs of widths that would idle lanes: VALU-bound, not TA-bound) keep the three loads.
// (round 6) The two flavours are chosen ONCE around both planes' loads in the kernel, and the neighbour dwords are zeroed before any load is issued: with `if (!wide)` inside a
// per-plane function the chroma call came behind the merge of the luma call's two branches, where the compiler has to assume the other branch's loads in flight to the same
// registers -- it put `s_waitcnt vmcnt(8)` in front of the last lane's loads, i.e. made every wave wait for its main loads before it requested its last neighbour dwords (a
// second memory round trip per tile).
template <int P2, int NROWS>
__device__ __forceinline__ void bcr_load_rows_wide(const uint8_t *plane, int pitch, int row0, int plane_rows, int q, bool first, bool last, bool run_first, bool run_last,
                                                   uint32_t (&ext)[NROWS][P2 + 2], uint32_t (&nb)[NROWS][2]) {
    constexpr int RUN = 4 * P2;
    const uint32_t col = (uint32_t)(RUN * q);
    uint32_t off[NROWS];
#pragma unroll
    for (int r = 0; r < NROWS; r++) {
        off[r] = (uint32_t)bc_row<NROWS>(row0, r, plane_rows) * (uint32_t)pitch + col;
        bc_ld<P2>(plane + off[r], &ext[r][1]);
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
    uint8_t *out = (uint8_t *)t.out[id.frame];

    // the run this lane belongs to: the lanes of the wave that share its output rows (A of them active)
    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - BCR_COLS * run_m)) / BCR_COLS);
    const bool run_first = run_m == 0, run_last = run_m == run_a - 1;
    // every load of the tile first, then the neighbour shuffles and the column edge fix-ups, then the arithmetic
    uint32_t ey[G::NYR][P2 + 2], xy[G::NYR][2], ec[G::NCR][P2 + 2], xc[G::NCR][2];
    uint32_t ny[G::NYR][2] = {}, nc[G::NCR][2] = {}; // (any defined value: the lanes that do not load theirs take their neighbour's dword)
    const bool wide = d.tx >= 64; // wave-uniform: a wave is one run
    if (wide) {
        bcr_load_rows_wide<P2, G::NYR>(t.y[id.frame], d.pitch_y, 2 * P2 * n4 - 1, d.src_h, q, first, last, run_first, run_last, ey, ny);
        if constexpr (!LUMA_ONLY) bcr_load_rows_wide<P2, G::NCR>(t.uv[id.frame], d.pitch_uv, P2 * n4 - 1, d.src_h >> 1, q, first, last, run_first, run_last, ec, nc);
    } else {
        bc_load_rows<P2, G::NYR>(t.y[id.frame], d.pitch_y, 2 * P2 * n4 - 1, d.src_h, q, first, last, ey);
        if constexpr (!LUMA_ONLY) bc_load_rows<P2, G::NCR>(t.uv[id.frame], d.pitch_uv, P2 * n4 - 1, d.src_h >> 1, q, first, last, ec);
    }
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

    r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
}

template <int P2> static hipError_t launch_bcr_k(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, hipStream_t stream) {
    switch (out) {
#define TSVPP_BCR(O) case O: TSVPP_LAUNCH((vpp_bicubic_r32_kernel<O, P2>), grid, block, 0, stream, d, t); break;
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
