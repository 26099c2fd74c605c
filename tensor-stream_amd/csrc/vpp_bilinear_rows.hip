// vpp_bilinear_rows.hip -- BILINEAR at large down-scale ratios (BASELINE config C3: crop 1280x720 -> 256x256 = 5.0 x 2.8125; 1080p -> 300x300,
// 224x224 ...): one wave = one output tile of 64 columns x R rows, the source rows it TAPS staged as contiguous row segments by LDS-DMA.
//
// What it replaces.  Until round 4 these requests ran on vpp_fused_gather_kernel: every tap a byte load from global memory.  At a horizontal
// ratio of 5 the 64 lanes of such a load fetch 64 bytes out of a dozen 128-byte lines -- one byte per lane and texture-addresser cycle, 24 such loads per
// thread tile: C3's 64-frame launch issued 1.97e5 of those wave instructions for 113 MB of traffic and ran 23.9 us (0.60 of 8 TB/s on moved bytes) --
// and 50.1 us for 128 frames, i.e. bound by the issue of those gathers, not by the launch ramp (profiles/r04_batch128_ab.txt).
// The LDS-staged 2x2-tap kernel (vpp_bilinear.hip) does not fit either: it stages the DENSE footprint of a workgroup tile, at a vertical ratio of
// 2.8 that is 40 % more rows than the two per output row that are tapped.
//
// Here no byte leaves global memory on its own:
//   stage    the tile taps rows y(i), y(i) + 1 for each of its R luma rows and R / 2 chroma rows (src/Resize.cu:5-25, 269-312: the chroma grid
//            reuses the luma formulas on its own indices) -- 3 R row segments of L 16-byte chunks (L = 64 columns x ratio + taps + misalignment,
//            host: LaunchDesc::bil_rows).  ONE global_load_lds_dwordx4 fetches floor(64 / L) segments: lane l = (segment l / L, chunk l % L),
//            landing at LDS byte 16 l of the instruction's base -- so segment s lives at 16 L s whatever the grouping.  All of a tile's
//            instructions (8 for C3) are issued back to back, then ONE s_waitcnt: 8 KiB in flight per wave, 16-20 waves per CU.
//   sample   the usual thread tile (4 columns x 2 rows, one chroma pair row) picks its taps out of the staged segments with LDS byte reads and
//            blends them with `bilerp` -- the reference's operation tree as nvcc compiled it (vpp_device.h) -- then the shared colour back end
//            (color_store_tile: every output flavour, 16-byte stores).
//   WX0      every horizontal weight is zero (odd integer ratio, C3): the right-hand taps are not read (any finite value gives the same bits:
//            (float)B * 0 == +0), as in the samplers of vpp_device.h.
// No workgroup barrier: a wave's LDS operations execute in order and nothing is shared between waves.  Rows are staged in PAIRS per output row even
// when two output rows share a source row (vertical ratios below 2): the kernel is selected for ratio products >= 12 (launch_fused), where that
// is rare; it stays correct for any ratio whose segment fits one instruction (L <= 64, horizontal ratios up to ~15.7).
//   POINT    (round 6) the pure point samplers -- NEAREST (src/Resize.cu:242-267), BILINEAR / BICUBIC requests whose weights are all zero -- on the same front
//            end: ONE segment per output row (R luma + R / 2 chroma segments), the sample is a byte read.  Until round 6 they ran on vpp_point_kernel, which stages
//            one LDS row per output row per WORKGROUP tile behind a barrier: 1080p -> 224 x 224 NEAREST moved 0.51 of the roofline at 512 frames per launch where
//            BILINEAR, with twice the rows to fetch, moved 0.71 on this kernel (profiles/r06_nn_matrix.txt).
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

// source coordinate of one output index for the variants of this kernel: the 2x2-tap ones and the zero-weight point samplers share bilinear_axis's (BICUBIC's
// coordinate is the same fused expression, vpp_axis.h), NEAREST has its own
template <bool NEAR> __device__ __forceinline__ int rows_coord(int idx, float ratio, int limit) {
    if constexpr (NEAR) return point_coord<PK_NEAREST>(idx, ratio, limit);
    int p;
    float w_;
    bilinear_axis(idx, ratio, limit, p, w_);
    return p;
}

enum { BRK_2X2 = 0, BRK_WX0 = 1, BRK_POINT = 2, BRK_NEAREST = 3 }; // (BRK_POINT: zero-weight BILINEAR / BICUBIC, bilinear_axis coordinates; BRK_NEAREST: (int)(r j))
template <int OUT, int KIND>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_rows_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    constexpr bool WX0 = KIND == BRK_WX0, POINT = KIND >= BRK_POINT, NEAR = KIND == BRK_NEAREST;
    constexpr int SPR = POINT ? 1 : 2; // staged segments per output row
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lane = (int)(threadIdx.x & 63u), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int R = 8 * d.rpt;                                  // output rows of the tile
    const int L = d.bil_rows;                                 // 16-byte chunks per row segment
    const int j_first = tile_col0(d, id.tx, 64 * d.br_waves) + wave * 64, i_first = id.ty * R;
    if (j_first >= d.dst_w) return; // (no workgroup barrier anywhere in this kernel)
    const int nrows = min(R, d.dst_h - i_first);              // even: dst_h is
    const int ch = d.src_h >> 1;

    // the tile's byte columns: luma [x_lo, x(j_last) + 1], chroma [2 cx_lo, 2 cx(cj_last) + 3] -- the host sized L for both
    const int x_lo = rows_coord<NEAR>(j_first, d.xr, d.src_w), cx_lo = rows_coord<NEAR>(j_first >> 1, d.xr, d.src_w);
    // plane pointers rounded down to 16 bytes (wave-uniform) + the bytes they were rounded by: every offset below is >= 0
    const uint8_t *fy = t.y[id.frame], *fc = t.uv[id.frame];
    const uint32_t pm_y = (uint32_t)((uintptr_t)fy & 15), pm_c = (uint32_t)((uintptr_t)fc & 15);
    const uint8_t *plane_y = fy - pm_y, *plane_c = fc - pm_c;
    const uint32_t seg0_y = (pm_y + (uint32_t)x_lo) & ~15u, seg0_c = (pm_c + 2u * (uint32_t)cx_lo) & ~15u; // a segment's first chunk (byte column from `plane`)

    uint8_t *wl = lds_raw + wave * d.bc_wave_bytes; // wave-private: segment s at 16 L s
    const int seg_bytes = 16 * L;
    const int nluma = SPR * nrows, nchroma = SPR * (nrows >> 1);
    // ---- stage: floor(64 / L) segments per instruction, the luma segments [0, 2 nrows), then the chroma segments [2 nrows, 3 nrows) (POINT: [0, nrows), [nrows, 3 nrows / 2))
    {
        // Lane q evaluates ONCE where luma segment q and chroma segment q start (row offset in bytes): the loops below fetch a segment's offset from the lane that
        // holds it (ds_bpermute) instead of evaluating a coordinate per lane and instruction (the first version: 50 VALU instructions per DMA instruction, ~400 per
        // wave -- with its reads served from cache the launch still took 14 us of its 23: instruction issue, profiles/r05_c3_diag.txt).
        uint32_t ro_y, ro_c;
        {
            const int rq = POINT ? lane : (lane >> 1); // the output row whose segment(s) lane q describes
            int y = rows_coord<NEAR>(min(i_first + rq, d.dst_h - 1), d.yr, d.src_h);
            int c = rows_coord<NEAR>(min((i_first >> 1) + rq, (d.dst_h >> 1) - 1), d.yr, d.src_h);
            y = min(y, d.src_h - 1);
            c = min(c, ch - 1); // (never fires for a valid request, see sample_chroma)
            if (!POINT && (lane & 1)) { // the second row of the pair: y2 = (y + 1 >= rows) ? y : y + 1
                y = min(y + 1, d.src_h - 1);
                c = min(c + 1, ch - 1);
            }
            ro_y = (uint32_t)y * (uint32_t)d.pitch_y;
            ro_c = (uint32_t)c * (uint32_t)d.pitch_uv;
        }
        const int lrow = (lane * (65536 / L + 1)) >> 16, lchunk = lane - lrow * L; // lane / L, lane % L (lane < 64 <= 65536 / L: exact)
        const int rpi = d.br_rpi;                                                  // 64 / L; 0: a segment is wider than one instruction (L > 64)
        const uint32_t last_y = ((uint32_t)(d.src_h - 1) * (uint32_t)d.pitch_y + pm_y + (uint32_t)d.src_w - 1u) & ~15u;  // the planes' last valid chunks
        const uint32_t last_c = ((uint32_t)(ch - 1) * (uint32_t)d.pitch_uv + pm_c + (uint32_t)d.src_w - 1u) & ~15u;
        const uint32_t cb_y = seg0_y + 16u * (uint32_t)lchunk, cb_c = seg0_c + 16u * (uint32_t)lchunk;
        // chunks this tile really needs (<= L, the host's bound for any tile): up to its last column's right-hand tap -- a chunk past it can start a 128-byte line that
        // no tile needs (C3: bytes 1280 .. 1295 of every row, +10 % read traffic in the first version)
        int ncy, ncc;
        {
            const int j_last = min(j_first + 63, d.dst_w - 1);
            const int xl = rows_coord<NEAR>(j_last, d.xr, d.src_w), cl = rows_coord<NEAR>(j_last >> 1, d.xr, d.src_w);
            const uint32_t by = pm_y + (uint32_t)min(xl + ((WX0 || POINT) ? 0 : 1), d.src_w - 1), bc = pm_c + (uint32_t)min(2 * cl + ((WX0 || POINT) ? 1 : 3), d.src_w - 1);
            ncy = (int)((by - seg0_y) >> 4) + 1;
            ncc = (int)((bc - seg0_c) >> 4) + 1;
        }
        if (rpi > 0) {
            for (int s0 = 0; s0 < nluma; s0 += rpi) {
                const int q = s0 + lrow;
                const uint32_t voff = min((uint32_t)__builtin_amdgcn_ds_bpermute(q << 2, (int)ro_y) + cb_y, last_y); // chunks past a plane's end hold bytes no tap reads
                uint8_t *dst = wl + s0 * seg_bytes; // wave-uniform
                if (lrow < rpi && q < nluma && lchunk < ncy)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane_y + voff), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
            if constexpr (!kLumaOnly<OUT>) {
                for (int s0 = 0; s0 < nchroma; s0 += rpi) {
                    const int q = s0 + lrow;
                    const uint32_t voff = min((uint32_t)__builtin_amdgcn_ds_bpermute(q << 2, (int)ro_c) + cb_c, last_c);
                    uint8_t *dst = wl + (nluma + s0) * seg_bytes;
                    if (lrow < rpi && q < nchroma && lchunk < ncc)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane_c + voff), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                }
            }
        } else {
            // (round 6) a segment of 65 .. 128 chunks -- horizontal ratios 15.7 .. 31: 4K -> 224 x 224 is 17.1 -- takes TWO instructions: lane l fetches chunk l, then chunk 64 + l; the
            // segment's row offset is wave-uniform (v_readlane of the lane that evaluated it).  These requests ran on the byte-gather kernel until round 6.
            for (int sg = 0; sg < nluma; sg++) {
                const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)ro_y, sg);
                for (int k = 0; 64 * k < L; k++) {
                    const int chunk = lane + 64 * k;
                    const uint32_t voff = min(row + seg0_y + 16u * (uint32_t)chunk, last_y);
                    uint8_t *dst = wl + sg * seg_bytes + 1024 * k;
                    if (chunk < ncy)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane_y + voff), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                }
            }
            if constexpr (!kLumaOnly<OUT>) {
                for (int sg = 0; sg < nchroma; sg++) {
                    const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)ro_c, sg);
                    for (int k = 0; 64 * k < L; k++) {
                        const int chunk = lane + 64 * k;
                        const uint32_t voff = min(row + seg0_c + 16u * (uint32_t)chunk, last_c);
                        uint8_t *dst = wl + (nluma + sg) * seg_bytes + 1024 * k;
                        if (chunk < ncc)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane_c + voff), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                    }
                }
            }
        }
    }

    // ---- sample + colour: 16 x 4 thread tiles of 4 x 2 pixels per 8-row slab.  The thread's coordinates are evaluated while its segments are in flight.
    const int lx = lane & 15, ly = lane >> 4;
    const int j0 = j_first + lx * PXW;
    const bool live = j0 < d.dst_w && !is_row_tail(d, j0); // (the two-column row tail belongs to the tail launch, launch_fused)
    const int jc = live ? j0 : j_first;                    // idle lanes: any column of the tile (never stored)
    // the thread's four luma columns and two chroma pair columns: LDS address of the left tap in segment 0, step to the right tap, weight
    const uint8_t *py[PXW], *pc[2];
    int xdy[PXW], du[2], dv[2];
    float wxy[PXW], wxc[2];
#pragma unroll
    for (int c = 0; c < PXW; c++) {
        int x;
        if constexpr (POINT) { x = min(rows_coord<NEAR>(jc + c, d.xr, d.src_w), d.src_w - 1); wxy[c] = 0.0f; }
        else bilinear_axis(jc + c, d.xr, d.src_w, x, wxy[c]);
        xdy[c] = (x + 1 >= d.src_w) ? 0 : 1;
        py[c] = wl + (pm_y + (uint32_t)x - seg0_y);
    }
    if constexpr (!kLumaOnly<OUT>) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            int x;
            if constexpr (POINT) { x = min(rows_coord<NEAR>((jc >> 1) + c, d.xr, d.src_w), (d.src_w >> 1) - 1); wxc[c] = 0.0f; }
            else bilinear_axis((jc >> 1) + c, d.xr, d.src_w, x, wxc[c]);
            const int xu = 2 * x, xv = 2 * x + 1;
            du[c] = (xu + 2 >= d.src_w) ? 0 : 2;
            dv[c] = (xv + 2 >= d.src_w) ? 0 : 2;
            pc[c] = wl + nluma * seg_bytes + (pm_c + (uint32_t)xu - seg0_c);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!live) return;
    for (int sl = 0; sl < d.rpt; sl++) {
        const int r0 = sl * 8 + ly * PXH, i0 = i_first + r0;
        if (i0 >= d.dst_h) break;
        float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
        if constexpr (POINT) { // the sample is the byte itself: luma row r0 + r = segment r0 + r, chroma row r0 / 2 = segment nluma + r0 / 2
#pragma unroll
            for (int r = 0; r < PXH; r++)
#pragma unroll
                for (int c = 0; c < PXW; c++) Yf[r][c] = (float)py[c][(r0 + r) * seg_bytes];
            if constexpr (!kLumaOnly<OUT>) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    Uf[c] = (float)pc[c][(r0 >> 1) * seg_bytes];
                    Vf[c] = (float)pc[c][(r0 >> 1) * seg_bytes + 1];
                }
            }
            color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
            continue;
        }
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            int y_;
            float wy;
            bilinear_axis(i0 + r, d.yr, d.src_h, y_, wy);
            const int top = (2 * (r0 + r)) * seg_bytes, bot = top + seg_bytes;
            if constexpr (WX0) {
                // wx == 0 for every column (host: LaunchDesc::wx_zero): bilerp(A, B, C, D, 0, wy) = fma(D, 0 wy, fma(C wy, 1, fma(A 1, 1 - wy, (B 0)(1 - wy)))) is, rounding
                // for rounding, (C wy) + (A (1 - wy)) -- two products and one sum, on column pairs (v_pk_mul_f32 / v_pk_add_f32): 5 instructions per value instead of 16
                const float omy = 1.0f - wy;
#pragma unroll
                for (int c = 0; c < PXW; c += 2) {
                    const f2 A = { (float)py[c][top], (float)py[c + 1][top] }, C = { (float)py[c][bot], (float)py[c + 1][bot] };
                    const f2 v = C * (f2){ wy, wy } + A * (f2){ omy, omy };
                    Yf[r][c] = (float)((int)v.x & 0xff);
                    Yf[r][c + 1] = (float)((int)v.y & 0xff);
                }
            } else {
                const f2 wy2 = { wy, wy }, omy2 = { 1.0f - wy, 1.0f - wy };
#pragma unroll
                for (int c = 0; c < PXW; c += 2) {
                    const f2 A = { (float)py[c][top], (float)py[c + 1][top] }, C = { (float)py[c][bot], (float)py[c + 1][bot] };
                    const f2 B = { (float)py[c][top + xdy[c]], (float)py[c + 1][top + xdy[c + 1]] }, D = { (float)py[c][bot + xdy[c]], (float)py[c + 1][bot + xdy[c + 1]] };
                    const f2 wx2 = { wxy[c], wxy[c + 1] };
                    const f2 v = bilerp2(A, B, C, D, wx2, (f2){ 1.0f, 1.0f } - wx2, wy2, omy2);
                    Yf[r][c] = (float)((int)v.x & 0xff);
                    Yf[r][c + 1] = (float)((int)v.y & 0xff);
                }
            }
        }
        if constexpr (!kLumaOnly<OUT>) {
            int y_;
            float wy;
            bilinear_axis(i0 >> 1, d.yr, d.src_h, y_, wy);
            const int top = r0 * seg_bytes, bot = top + seg_bytes; // chroma row r0 / 2 of the tile: segments 2 (r0 / 2), + 1 (pc points at the first chroma segment)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const f2 A = { (float)pc[c][top], (float)pc[c][top + 1] }, C = { (float)pc[c][bot], (float)pc[c][bot + 1] }; // (U, V) pairs
                f2 v;
                if constexpr (WX0) {
                    v = C * (f2){ wy, wy } + A * (f2){ 1.0f - wy, 1.0f - wy };
                } else {
                    const f2 B = { (float)pc[c][top + du[c]], (float)pc[c][top + 1 + dv[c]] }, D = { (float)pc[c][bot + du[c]], (float)pc[c][bot + 1 + dv[c]] };
                    const f2 wx2 = { wxc[c], wxc[c] };
                    v = bilerp2(A, B, C, D, wx2, (f2){ 1.0f, 1.0f } - wx2, (f2){ wy, wy }, (f2){ 1.0f - wy, 1.0f - wy });
                }
                Uf[c] = (float)((int)v.x & 0xff);
                Vf[c] = (float)((int)v.y & 0xff);
            }
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

hipError_t launch_bilinear_rows(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(64 * d.br_waves));
    const int kind = d.point_kind == PK_NEAREST ? BRK_NEAREST : (d.point_kind != PK_NONE ? BRK_POINT : (d.wx_zero != 0 ? BRK_WX0 : BRK_2X2));
    if (info) {
        static const char *const names[4] = { "vpp_bilinear_rows_kernel<OUT, 2x2>", "vpp_bilinear_rows_kernel<OUT, wx0>", "vpp_bilinear_rows_kernel<OUT, point>",
                                              "vpp_bilinear_rows_kernel<OUT, nearest>" };
        info->kernel = names[kind];
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    switch (out) {
#define TSVPP_BR(O)                                                                                                     \
    case O:                                                                                                             \
        if (kind == BRK_NEAREST) TSVPP_LAUNCH((vpp_bilinear_rows_kernel<O, BRK_NEAREST>), grid, block, lds_bytes, stream, d, t);  \
        else if (kind == BRK_POINT) TSVPP_LAUNCH((vpp_bilinear_rows_kernel<O, BRK_POINT>), grid, block, lds_bytes, stream, d, t); \
        else if (kind == BRK_WX0) TSVPP_LAUNCH((vpp_bilinear_rows_kernel<O, BRK_WX0>), grid, block, lds_bytes, stream, d, t);     \
        else TSVPP_LAUNCH((vpp_bilinear_rows_kernel<O, BRK_2X2>), grid, block, lds_bytes, stream, d, t);                          \
        break;
        TSVPP_BR(O_U8_PLANAR) TSVPP_BR(O_U8_MERGED) TSVPP_BR(O_F32_PLANAR) TSVPP_BR(O_F32_MERGED) TSVPP_BR(O_NV12_U8)
        TSVPP_BR(O_NV12_F32) TSVPP_BR(O_Y800_U8) TSVPP_BR(O_Y800_F32) TSVPP_BR(O_HSV_F32)
#undef TSVPP_BR
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
