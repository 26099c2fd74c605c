// vpp_area_stream.hip -- AREA down-scale with float (non-dyadic) weights, any ratio up to 15: one wave = one output tile,
// source rows STREAMED through a small wave-private LDS ring.
//
// The reference's value (src/Resize.cu:160-178 as compiled) is
//     sum = 0; div = 0; for a in rows: for b in cols: wgt = wx[b] * wy[a]; div += wgt; sum = fma(p[a][b], wgt, sum)
//     out = (int)(sum / div)
// in exactly that order -- a serial chain of rx * ry fused multiply-adds per value (45 at 1080p -> 224 x 224, 180 at 4K ->
// 224 x 224) that cannot be split or re-ordered.  The column-per-lane kernels of rounds 1 / 2 (vpp_area_cols_kernel,
// vpp_area_cols_lds_kernel, vpp_area_direct_float_kernel) kept that order but ran every chain behind its own exposed memory
// latencies (one or two loads per source row, consumed at once) or behind a workgroup-wide staging barrier with two
// workgroups per CU: 0.41-0.45 of the roofline with the VALU a quarter busy and the texture addresser at 69 %.
//
// Here (the structure of vpp_bicubic_cols.hip): a wave owns 128 output columns x R output rows and no workgroup barrier exists.
//   * Every source row the tile taps is fetched ONCE, as one contiguous segment, by LDS-DMA (global_load_lds_dwordx4: lane l
//     fetches chunk l of the segment; one or two instructions per row, no VGPRs) into a ring of four rows -- three rows are in
//     flight while the lanes consume the fourth.  A row that two consecutive output rows share stays in the ring.
//   * Lane l accumulates the chains of TWO output columns (l and l + 64) of the current output row as one float pair: the
//     column weights of both sit in registers, the row weight is a scalar, a tap costs two byte conversions, one packed
//     multiply and one packed fused multiply-add for the two values.  Chroma: lane = pair column, (U, V) as the pair.
//   * The divisor of a (column pattern, row pattern) pair comes from the host-built table (LaunchDesc::area_div: the same fp32
//     products and summation order), fetched one output row ahead.
//   * The resized tile goes through a small wave-private byte tile into the usual 2 x 4 thread tiles (color_store_tile).
// Taps are padded to 4 * NK with zero weights (an exact no-op in both accumulators, as in the kernels this replaces).
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef float as_f4a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int AS_RING = AS_RING_ROWS; // rows of the ring (vpp_kernels.h)

__device__ __forceinline__ float asb(uint32_t v, int b) { return (float)((v >> (8 * b)) & 255u); }

// The taps of one source row for a lane's two chains (luma: columns A and B; chroma: U and V of one pair column), in tap order.
// Two facts about the weight rows (build_area_rows, the reference's pattern: [rest] 1 1 ... 1 [last fraction], `taps` = ceil(ratio)
// entries, zero-padded to 4 * nkx) allow a third less arithmetic without touching a single rounding:
//   * taps 1 .. taps - 3 are exactly 1.0f in EVERY row of the table (the host checks it: AreaTable::ones_end), so their weight is
//     the row weight itself (1.0f * wy == wy): no multiply.  At compile time only nkx is known (taps >= 4 * nkx - 3), hence
//     taps 1 .. 4 * nkx - 5; a row whose own weight is 1.0f (ROW_ONE: every row strictly inside the footprint) multiplies nothing.
//   * taps >= `taps` have weight 0 (an exact no-op on a non-negative sum): skipped, by wave-uniform branches in the last group(s).
// Measured (profiles/r03_area_stream_lean_ab.txt): from 17 horizontal taps on (ratios > 16, the instances NK = 6 / 8) this is worth
// +20 % (4K -> 224 x 224: 0.44 -> 0.53); below, the kernel is bound by its row streaming (without ANY arithmetic it runs 0.64 at
// 1080p -> 224 x 224 against 0.60 with), and the branches cost more than the multiplies they save (0.60 -> 0.54): AS_LEAN.
template <int NK> constexpr bool AS_LEAN = NK >= 6;

template <int NK, bool CHROMA, bool ROW_ONE>
__device__ __forceinline__ void as_row_taps(f2 &acc, const uint32_t *dA, const uint32_t *dB, uint32_t aA, uint32_t aB, const as_f4a4 *wxA, const as_f4a4 *wxB,
                                            float wy, int taps) {
    constexpr bool LEAN = AS_LEAN<NK>;
    constexpr int NKMIN = NK == 6 ? 5 : (NK == 8 ? 7 : NK); // the smallest nkx this instance serves
    constexpr int ALWAYS = LEAN ? 4 * NKMIN - 3 : 4 * NK;                 // taps below this index exist for every ratio served
    constexpr int ONES_END = LEAN ? 4 * NKMIN - 4 : 0;               // taps 1 .. ONES_END - 1 weigh 1.0f
    auto tap = [&](int k, int b, uint32_t va, uint32_t vb) {
        const int t = 4 * k + b;
        const float wa = b == 0 ? wxA[k].x : (b == 1 ? wxA[k].y : (b == 2 ? wxA[k].z : wxA[k].w));
        if constexpr (!CHROMA) {
            const float wb = b == 0 ? wxB[k].x : (b == 1 ? wxB[k].y : (b == 2 ? wxB[k].z : wxB[k].w));
            f2 wgt = { wa, wb };
            if constexpr (!ROW_ONE) {
                if (t >= 1 && t < ONES_END) wgt = (f2){ wy, wy };
                else wgt = wgt * (f2){ wy, wy };
            }
            acc = __builtin_elementwise_fma((f2){ asb(va, b), asb(vb, b) }, wgt, acc); // colorSum = fma(data, weight, colorSum)
        } else {
            float wgt = wa;
            if constexpr (!ROW_ONE) wgt = (t >= 1 && t < ONES_END) ? wy : wa * wy;
            const uint32_t qq = (b & 2 ? vb : va) >> (16 * (b & 1)); // va = U0 V0 U1 V1, vb = U2 V2 U3 V3
            acc = __builtin_elementwise_fma((f2){ (float)(qq & 255u), (float)((qq >> 8) & 255u) }, (f2){ wgt, wgt }, acc);
        }
    };
    auto group = [&](int k, uint32_t &va, uint32_t &vb) {
        if constexpr (!CHROMA) {
            va = __builtin_amdgcn_alignbyte(dA[k + 1], dA[k], aA);
            vb = __builtin_amdgcn_alignbyte(dB[k + 1], dB[k], aB);
        } else {
            va = __builtin_amdgcn_alignbyte(dA[2 * k + 1], dA[2 * k], aA);
            vb = __builtin_amdgcn_alignbyte(dA[2 * k + 2], dA[2 * k + 1], aA);
        }
    };
#pragma unroll
    for (int k = 0; k < NK; k++) {
        if (4 * k + 3 < ALWAYS) {
            uint32_t va, vb;
            group(k, va, vb);
#pragma unroll
            for (int b = 0; b < 4; b++) tap(k, b, va, vb);
        } else {
            if (4 * k < ALWAYS || 4 * k < taps) { // (uniform)
                uint32_t va, vb;
                group(k, va, vb);
                // the chain is serial and the padding is a suffix: the first missing tap ends the row
                if (4 * k + 0 < ALWAYS || 4 * k + 0 < taps) {
                    tap(k, 0, va, vb);
                    if (4 * k + 1 < ALWAYS || 4 * k + 1 < taps) {
                        tap(k, 1, va, vb);
                        if (4 * k + 2 < ALWAYS || 4 * k + 2 < taps) {
                            tap(k, 2, va, vb);
                            if (4 * k + 3 < taps) tap(k, 3, va, vb);
                        }
                    }
                }
            }
        }
    }
}

// One plane of a wave's tile.  CHROMA = false: lane l owns luma columns jf + l and jf + 64 + l; true: pair column jf + l
// (jf in pair units), taps two bytes apart, (U, V) share every weight.  i0: first output row (plane grid), nout rows;
// res: result tile, 128 bytes per row (luma: column order; chroma: U V interleaved).
template <int NK, bool CHROMA>
__device__ __forceinline__ void as_plane(const LaunchDesc &d, const uint8_t *frame_plane, int pitch, int rows_in_plane, int row_bytes, int ncols, int jf, int i0,
                                         int nout, int nrows_grid, uint8_t *ring, int rowb, uint8_t *res, int lane, bool two) {
    constexpr int NDW = CHROMA ? 2 * NK + 1 : NK + 1; // dwords that cover a lane's window of 4 NK taps at any byte alignment
    const uint32_t pm = (uint32_t)((uintptr_t)frame_plane & 15);
    const uint8_t *plane = frame_plane - pm; // rounded down to 16 bytes (wave-uniform): every offset below is >= 0

    // ---- columns of this lane: first tap byte, weights
    // (two == false -- ratios beyond ~8, where a 128-column segment would not fit two DMA instructions: the wave's tile is 64 columns wide and
    // both halves of the pair compute column A)
    // (chroma of a 64-column tile: 32 pair columns -- lanes 32..63 repeat lanes 0..31, their results land in unused bytes of the tile)
    const int jA = min(jf + ((CHROMA && !two) ? (lane & 31) : lane), ncols - 1), jB = (CHROMA || !two) ? jA : min(jf + 64 + lane, ncols - 1);
    const int rp = (two || CHROMA) ? 128 : 64; // result tile pitch (chroma of a 64-column tile: 32 pairs = lanes 0..31 valid)
    const int xA = (CHROMA ? 2 : 1) * (int)(d.xr * (float)jA), xB = (CHROMA ? 2 : 1) * (int)(d.xr * (float)jB);
    const int pxA = jA % d.nx, pxB = jB % d.nx;
    as_f4a4 wxA[NK], wxB[NK];
    const as_f4a4 zero4 = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int k = 0; k < NK; k++) { // the table rows hold 4 * d.nkx <= 4 * NK weights (NK: the next instantiated tap count)
        wxA[k] = k < d.nkx ? *(const as_f4a4 *)(d.patx4 + pxA * 4 * d.nkx + 4 * k) : zero4;
        wxB[k] = CHROMA ? wxA[k] : (k < d.nkx ? *(const as_f4a4 *)(d.patx4 + pxB * 4 * d.nkx + 4 * k) : zero4);
    }
    // ---- the wave's row segment: bytes [xs0, xs1] of a source row, as nchunks aligned 16-byte chunks
    const int xs0 = __builtin_amdgcn_readlane(xA, 0);
    const int xs1 = __builtin_amdgcn_readlane(CHROMA ? xA : xB, 63) + (CHROMA ? 8 : 4) * NK - 1;
    const uint32_t mis = (pm + (uint32_t)xs0) & 15u, seg0 = (pm + (uint32_t)xs0) & ~15u;
    const int nchunks = (int)((mis + (uint32_t)(xs1 - xs0 + 1) + 15u) >> 4); // <= rowb / 16 (host bound), <= 128
    const int ipr = nchunks > 64 ? 2 : 1;                                    // DMA instructions per row
    const uint32_t last_chunk = ((uint32_t)(rows_in_plane - 1) * (uint32_t)pitch + pm + (uint32_t)row_bytes - 1u) & ~15u; // the plane's last valid chunk
    const uint32_t aA = (uint32_t)(xA - xs0) + mis, aB = (uint32_t)(xB - xs0) + mis; // the lane's windows inside a ring row
    const uint32_t lofs = seg0 + 16u * (uint32_t)lane;
    auto issue = [&](int r) { // row r -> ring slot r % 4
        const uint32_t rbase = (uint32_t)r * (uint32_t)pitch;
        uint8_t *dst = ring + (r & (AS_RING - 1)) * rowb; // wave-uniform
        if (lane < nchunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane + min(rbase + lofs, last_chunk)),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        if (ipr > 1 && lane + 64 < nchunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane + min(rbase + lofs + 1024u, last_chunk)),
                                             (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
    };
    // NOTE: with ipr == 2 and lane + 64 >= nchunks for every lane the second instruction is skipped by all lanes: the count below
    // then over-waits by one, which is safe.

    // ---- rows: stream [y_first, y_last] through the ring
    auto row_y = [&](int io) { return (int)(d.yr * (float)min(i0 + io, nrows_grid - 1)); };
    const int y_last = min(row_y(nout - 1) + d.ry - 1, rows_in_plane - 1);
    int hi = row_y(0); // next row to issue
    for (int n = 0; n < AS_RING - 1 && hi <= y_last; n++) issue(hi++);
    // per output row, fetched one row ahead: the divisors of this lane's columns and the row weights (lane a holds wy[a]: ONE
    // coalesced load instead of a scalar load -- and its latency -- per source row; the steps fetch them with v_readlane_b32)
    float divA_n = 0.0f, divB_n = 0.0f, wy_n = 0.0f;
    const int wlane = min(lane, 4 * d.nky - 1);
    {
        const int iy = min(i0, nrows_grid - 1) % d.ny;
        divA_n = d.area_div[pxA * d.ny + iy];
        divB_n = CHROMA ? divA_n : d.area_div[pxB * d.ny + iy];
        wy_n = d.paty4[iy * 4 * d.nky + wlane];
    }
    for (int io = 0; io < nout; io++) {
        const int i = min(i0 + io, nrows_grid - 1);
        const int y = (int)(d.yr * (float)i);
        const int y_next = io + 1 < nout ? row_y(io + 1) : 0x7fffffff;
        const f2 div = { divA_n, divB_n };
        const float wyv = wy_n;
        if (io + 1 < nout) {
            const int iy = min(i0 + io + 1, nrows_grid - 1) % d.ny;
            divA_n = d.area_div[pxA * d.ny + iy];
            divB_n = CHROMA ? divA_n : d.area_div[pxB * d.ny + iy];
            wy_n = d.paty4[iy * 4 * d.nky + wlane];
        }
        f2 acc = { 0.0f, 0.0f };
        for (int a = 0; a < d.ry; a++) {
            const int r = min(y + a, rows_in_plane - 1);
            while (hi <= r) issue(hi++); // (only if the ring ran dry: ry > 4 at the start of a tile)
            // row r has landed when at most the rows issued after it are outstanding
            const int ahead = (hi - 1 - r) * ipr; // steady state: 3 rows = 3 or 6 instructions
            switch (ahead < 14 ? ahead : 14) { // (the count is an immediate)
#define AS_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
                AS_W(14) AS_W(13) AS_W(12) AS_W(11) AS_W(10) AS_W(9) AS_W(8) AS_W(7) AS_W(6) AS_W(5) AS_W(4) AS_W(3) AS_W(2) AS_W(1)
#undef AS_W
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            const uint8_t *slot = ring + (r & (AS_RING - 1)) * rowb;
            uint32_t dA[NDW], dB[CHROMA ? 1 : NDW];
            {
                const uint32_t *p = (const uint32_t *)(slot + (aA & ~3u));
#pragma unroll
                for (int q = 0; q < NDW; q++) dA[q] = p[q];
                if constexpr (!CHROMA) {
                    const uint32_t *pb = (const uint32_t *)(slot + (aB & ~3u));
#pragma unroll
                    for (int q = 0; q < NDW; q++) dB[q] = pb[q];
                }
            }
            // rows below min(r + 1, y_next) are dead: refill the ring (after the reads above have returned: the slot that row
            // hi lands in may be the one just read)
            {
                const int lo = min(r + 1, y_next);
                if (hi < lo + AS_RING && hi <= y_last) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    do issue(hi++);
                    while (hi < lo + AS_RING && hi <= y_last);
                }
            }
            // the row weight is wave-uniform (an SGPR): a row inside the footprint has weight exactly 1.0f and its taps use the column
            // weights as they are (wx * 1.0f == wx)
            const int wyb = __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyv), a);
            if (AS_LEAN<NK> && wyb == 0x3f800000) as_row_taps<NK, CHROMA, true>(acc, dA, dB, aA, aB, wxA, wxB, 1.0f, d.rx);
            else as_row_taps<NK, CHROMA, false>(acc, dA, dB, aA, aB, wxA, wxB, __builtin_bit_cast(float, wyb), d.rx);
        }
        const uint32_t q0 = (uint32_t)__builtin_truncf(acc.x / div.x), q1 = (uint32_t)__builtin_truncf(acc.y / (CHROMA ? div.x : div.y));
        if constexpr (!CHROMA) {
            res[io * rp + lane] = (uint8_t)q0;
            if (two) res[io * rp + 64 + lane] = (uint8_t)q1;
        } else {
            *(uint16_t *)(res + io * rp + 2 * lane) = (uint16_t)(q0 | (q1 << 8));
        }
    }
}

template <int NK, int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_area_stream_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lane = (int)(threadIdx.x & 63u), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // a wave's tile: 128 columns x R rows (R = 4 or 8), or -- as_two == 0 -- 64 columns x 8 rows; the workgroup = 2 x 2 waves
    const bool two = d.as_two != 0;
    const int CW = two ? 128 : 64, R = two ? 4 * d.rpt : 8;
    const int j_first = tile_col0(d, id.tx, 2 * CW) + (wave & 1) * CW, i_first = id.ty * 2 * R + (wave >> 1) * R;
    if (j_first >= d.dst_w || i_first >= d.dst_h) return; // (no workgroup barrier anywhere in this kernel)
    const int nrows = min(R, d.dst_h - i_first);

    uint8_t *wl = lds_raw + wave * d.bc_wave_bytes;
    uint8_t *ring = wl, *yt = wl + d.bc_ring_bytes, *uvt = yt + 128 * R;
    const int rowb = (d.bc_ring_bytes - 16) / AS_RING;

    as_plane<NK, false>(d, t.y[id.frame], d.pitch_y, d.src_h, d.src_w, d.dst_w, j_first, i_first, nrows, d.dst_h, ring, rowb, yt, lane, two);
    if constexpr (!kLumaOnly<OUT>)
        as_plane<NK, true>(d, t.uv[id.frame], d.pitch_uv, d.src_h >> 1, d.src_w, d.dst_w >> 1, j_first >> 1, i_first >> 1, nrows >> 1, d.dst_h >> 1, ring, rowb, uvt, lane, two);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // colour conversion + stores: thread tiles of 4 x 2 pixels, 32 x 2 of them per 4-row slab (128 columns) or 16 x 4 per 8-row slab
    const int lx = two ? (lane & 31) : (lane & 15), ly = two ? (lane >> 5) : (lane >> 4);
    const int j0 = j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const int slab = two ? 4 : 8, yp = two ? 128 : 64;
    for (int s = 0; s * slab < R; s++) {
        const int r0 = s * slab + ly * PXH, i0 = i_first + r0;
        if (i0 >= d.dst_h) break;
        float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const uint32_t v = *(const uint32_t *)(yt + (r0 + r) * yp + lx * PXW);
#pragma unroll
            for (int c = 0; c < PXW; c++) Yf[r][c] = asb(v, c);
        }
        if constexpr (!kLumaOnly<OUT>) {
            const uint32_t c = *(const uint32_t *)(uvt + (r0 >> 1) * 128 + lx * PXW);
            Uf[0] = asb(c, 0);
            Vf[0] = asb(c, 1);
            Uf[1] = asb(c, 2);
            Vf[1] = asb(c, 3);
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

template <int NK>
static hipError_t launch_area_stream_nk(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, size_t lds, hipStream_t stream) {
    switch (out) {
#define TSVPP_AS(O) case O: TSVPP_LAUNCH((vpp_area_stream_kernel<NK, O>), grid, dim3(MAX_THREADS), lds, stream, d, t); break;
        TSVPP_AS(O_U8_PLANAR) TSVPP_AS(O_U8_MERGED) TSVPP_AS(O_F32_PLANAR) TSVPP_AS(O_F32_MERGED) TSVPP_AS(O_NV12_U8)
        TSVPP_AS(O_NV12_F32) TSVPP_AS(O_Y800_U8) TSVPP_AS(O_Y800_F32) TSVPP_AS(O_HSV_F32)
#undef TSVPP_AS
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// d.as_nk: the instantiated tap counts are 4 x {1, 2, 3, 4, 6, 8}: launch_fused has rounded d.nkx up to one of them (the kernel
// zero-fills the weights past the table's 4 * d.nkx).
hipError_t launch_area_stream(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD));
    if (info) {
        static const char *const names[9] = { "", "vpp_area_stream_kernel<1,OUT>", "vpp_area_stream_kernel<2,OUT>", "vpp_area_stream_kernel<3,OUT>", "vpp_area_stream_kernel<4,OUT>",
                                              "", "vpp_area_stream_kernel<6,OUT>", "", "vpp_area_stream_kernel<8,OUT>" };
        info->kernel = (d.as_nk >= 1 && d.as_nk <= 8) ? names[d.as_nk] : "";
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    switch (d.as_nk) {
    case 1: return launch_area_stream_nk<1>(out, d, t, grid, lds_bytes, stream);
    case 2: return launch_area_stream_nk<2>(out, d, t, grid, lds_bytes, stream);
    case 3: return launch_area_stream_nk<3>(out, d, t, grid, lds_bytes, stream);
    case 4: return launch_area_stream_nk<4>(out, d, t, grid, lds_bytes, stream);
    case 6: return launch_area_stream_nk<6>(out, d, t, grid, lds_bytes, stream);
    case 8: return launch_area_stream_nk<8>(out, d, t, grid, lds_bytes, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace tsvpp
