// vpp_bicubic_int.hip -- BICUBIC for requests whose interpolation weights are all dyadic (multiples of 1/16):
// ratios 1.5, 2, 2.5, 4, 0.5, 1.25, 2.25 ... -- 1080p -> 720p, 4K -> 1080p, 1080p -> 540p, 2x up-scales.
//
// The reference evaluates Keys' cubic (a = -0.75) in fp64 (src/Resize.cu:27-91): per output value four horizontal
// 4-tap sums, each rounded (half away from zero) and clamped to a byte, then the same sum down the column.  For a weight
// w = k/16 every coefficient is an integer multiple of 2^-14 (w^3 / 4096, times a = -3/4), every product and partial
// sum is exact in fp64, and round() sees the exact value -- so the whole evaluation can be done in INTEGERS, bit for bit:
//     S = sum_t C_t p_t  with C_t = 16384 c_t (|C_t| <= 18811, fits int16),   value = clamp((S + 8192) >> 14, 0, 255)
// (S < 0 rounds to <= 0 and clamps to 0 either way).  Two v_dot2_i32_i16 per 4-tap sum, the rounding bias in the
// accumulator input, and v_ashr_pk_u8_i32 shifts, clamps and packs two values at once -- against ~14 packed/scalar float operations, a tie test and an
// fp64 fallback per sum in the general kernel of rounds 1 / 2 (every other ratio: vpp_bicubic_cols.hip since round 3).
//
// Structure (separable, like the general kernel): the footprint is staged in LDS (LDS-DMA or registers); phase 1
// evaluates H once per (staged source row, tile column) into a byte plane; phase 2 takes each thread's vertical sums
// from that plane.  Here the H plane is COLUMN-major so that the four vertical taps of a column are four consecutive
// bytes -- two aligned dword reads, v_alignbyte_b32 to put tap 0 in byte 0, v_perm_b32 to widen byte pairs to int16
// pairs.  The reference's edge rule (src/Resize.cu:32-43: the +1 AND +2 taps collapse onto the centre when either would
// leave the plane, the -1 tap at the low edge) moves WEIGHTS instead of taps -- exact in integers: the window is always
// four consecutive samples starting at max(p - 1, 0), a collapsed tap's coefficient is added to the centre's, and the
// samples the window reaches past the plane carry weight 0.
#include "vpp_device.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef short s16x2 __attribute__((ext_vector_type(2)));

struct ICol { int off, c01, c23, pad; };   // luma: LDS byte offset of the window start from the row base; chroma: of its U byte
struct IRow { int al, sh, c01, c23; };     // H-plane byte offset of the window start within a column, split into dword + byte

// Window start and folded integer coefficients of one output index along one axis (sample units; the chroma grid
// uses pair / chroma-row units with its own limit).
__device__ __forceinline__ void bicubic_int_axis(int idx, float ratio, int clamp_limit, int tap_limit, int &ws, int &c01, int &c23) {
    int p;
    double w;
    bicubic_axis(idx, ratio, clamp_limit, p, w); // clamps use the LUMA size on both grids (src/Resize.cu:325-347)
    float c[4];
    cubic_coeffs_f((float)w, c); // exact for w = k / 16
    const int C[4] = { (int)(c[0] * 16384.0f), (int)(c[1] * 16384.0f), (int)(c[2] * 16384.0f), (int)(c[3] * 16384.0f) };
    const bool hi = !(p + 1 >= tap_limit || p + 2 >= tap_limit), lo = !(p - 1 < 0);
    // taps (p - lo, p, p + hi, p + 2 hi) -> coefficients per window offset; the window starts at ws = p - lo
    const int m1 = hi ? C[1] : C[1] + C[2] + C[3], m2 = hi ? C[2] : 0, m3 = hi ? C[3] : 0; // offsets lo, lo + 1, lo + 2
    ws = lo ? p - 1 : p;
    const int wg[4] = { lo ? C[0] : C[0] + m1, lo ? m1 : m2, lo ? m2 : m3, lo ? m3 : 0 };
    c01 = (wg[0] & 0xffff) | (wg[1] << 16);
    c23 = (wg[2] & 0xffff) | (wg[3] << 16);
}

// S + 8192 of the 4-tap sum over the bytes of q (tap t in byte t); the value is clamp(that >> 14, 0, 255)
__device__ __forceinline__ int cubic_sum4(uint32_t q, int c01, int c23) {
    const uint32_t p01 = __builtin_amdgcn_perm(0u, q, 0x0c010c00u), p23 = __builtin_amdgcn_perm(0u, q, 0x0c030c02u);
    int s = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p01), __builtin_bit_cast(s16x2, c01), 8192, false);
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p23), __builtin_bit_cast(s16x2, c23), s, false);
}
// the same for interleaved (U, V) bytes: q0 = U0 V0 U1 V1, q1 = U2 V2 U3 V3
__device__ __forceinline__ void cubic_sum4_uv(uint32_t q0, uint32_t q1, int c01, int c23, int &su, int &sv) {
    const uint32_t u01 = __builtin_amdgcn_perm(0u, q0, 0x0c020c00u), u23 = __builtin_amdgcn_perm(0u, q1, 0x0c020c00u);
    const uint32_t v01 = __builtin_amdgcn_perm(0u, q0, 0x0c030c01u), v23 = __builtin_amdgcn_perm(0u, q1, 0x0c030c01u);
    su = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, u01), __builtin_bit_cast(s16x2, c01), 8192, false);
    su = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, u23), __builtin_bit_cast(s16x2, c23), su, false);
    sv = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, v01), __builtin_bit_cast(s16x2, c01), 8192, false);
    sv = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, v23), __builtin_bit_cast(s16x2, c23), sv, false);
}
// Four biased sums -> four bytes clamp(s >> 14, 0, 255), s0 in byte 0.  gfx950's V_ASHR_PK_U8_I32 shifts, saturates to
// [0, 255] and packs two values into the LOW 16 bits of its destination and leaves the upper 16 bits as they were
// (measured: tools/dbg_prims2 -- and the compiler of ROCm 7.2, which selects this instruction for the C expression
// max(min(s >> 14, 255), 0) | ... << 8, assumes they are zeroed: the first version of this kernel, written that way,
// OR-ed stale register contents into every third byte).  Written out as inline assembly, the preserved upper half is used
// on purpose: pair (s2, s3) is packed first and shifted up, pair (s0, s1) then lands below it.  Three instructions for
// four values instead of four shifts, four v_med3 and three v_lshl_or.
__device__ __forceinline__ uint32_t round_clamp_pack4(int s0, int s1, int s2, int s3) {
    uint32_t hi;
    // s_nop 2: a VALU read of a v_dot2c result needs three wait states; the compiler inserts them in its own code but does
    // not look into inline assembly (the sums usually come straight out of a v_dot2c_i32_i16)
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 14" : "=v"(hi) : "v"(s2), "v"(s3)); // upper half: don't care, shifted out below
    uint32_t r = hi << 16;
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 14" : "+v"(r) : "v"(s0), "v"(s1));
    return r;
}

// physical column of tile column c in the H planes: thread lx's k-th column (c = 4 lx + k) lives at k * tx + lx, so the
// lanes of a wave read columns one (odd number of dwords) stride apart in phase 2
__device__ __forceinline__ int hcol(int c, int tx) { return (c & 3) * tx + (c >> 2); }

template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bicubic_int_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int nthreads = d.tx * d.ty;
    const int tw = d.tx * PXW, th = d.ty * PXH * d.rpt;
    const Footprint f = tile_footprint<M_BICUBIC>(d, id);
    const int cw = d.src_w >> 1, chh = d.src_h >> 1;

    uint8_t *lds_y = lds_raw;
    uint8_t *lds_uv = lds_raw + d.lds_rows_y * d.lds_cpr_y * 16;
    uint8_t *hy = lds_uv + d.lds_rows_uv * d.lds_cpr_uv * 16 + 16; // 16 bytes of slack: the staged planes' dword over-read
    uint8_t *huv = hy + tw * d.hcs_y;
    ICol *xtab = (ICol *)(huv + tw * d.hcs_uv + 16);               // ... and the H planes'
    ICol *cxtab = xtab + tw;
    IRow *ytab = (IRow *)(cxtab + (tw >> 1));
    IRow *cytab = ytab + th;
    int *rby = (int *)(cytab + (th >> 1)); // LDS byte offset of every staged luma row (incl. its misalignment), padded to 4 k entries
    int *rbuv = rby + ((d.lds_rows_y + 3) & ~3);

    const uint8_t *ay, *auv;
    const LdsPlane py = describe_plane(lds_y, t.y[id.frame], d.pitch_y, f.ylo, f.xlo, d.lds_cpr_y, ay);
    const LdsPlane puv = describe_plane(lds_uv, t.uv[id.frame], d.pitch_uv, f.cylo, 2 * f.cxlo, d.lds_cpr_uv, auv);
    const int ny = min(f.yhi - f.ylo + 1, d.lds_rows_y), nuv = d.luma_only ? 0 : min(f.cyhi - f.cylo + 1, d.lds_rows_uv);
    const int spy = min(f.xhi - f.xlo + 1, d.lds_span_y), spuv = min(2 * (f.cxhi - f.cxlo + 1), d.lds_span_uv);
    if (d.dma) {
        stage_plane_dma(lds_y, ay, py, d.pitch_y, ny, spy, d.lds_magic_y, nthreads);
        stage_plane_dma(lds_uv, auv, puv, d.pitch_uv, nuv, spuv, d.lds_magic_uv, nthreads);
    } else {
        stage_planes<4, 2>(d, lds_y, ay, py, ny, spy, lds_uv, auv, puv, nuv, spuv, nthreads);
    }
    const int nrb_y = (d.lds_rows_y + 3) & ~3, nrb_uv = (d.lds_rows_uv + 3) & ~3;
    const int ntab = tw + (tw >> 1) + th + (th >> 1) + nrb_y + nrb_uv;
    for (int e = threadIdx.x; e < ntab; e += nthreads) {
        int k = e, ws, c01, c23;
        if (k < tw) { // luma columns
            bicubic_int_axis(f.j_first + k, d.xr, d.src_w, d.src_w, ws, c01, c23);
            xtab[k] = ICol{ ws - f.xlo, c01, c23, 0 };
            continue;
        }
        k -= tw;
        if (k < (tw >> 1)) { // chroma pair columns: the luma formulas on the chroma grid, taps in pair units
            bicubic_int_axis((f.j_first >> 1) + k, d.xr, d.src_w, cw, ws, c01, c23);
            cxtab[k] = ICol{ 2 * (ws - f.cxlo), c01, c23, 0 };
            continue;
        }
        k -= tw >> 1;
        if (k < th + (th >> 1)) { // rows: window start inside an H-plane column
            const bool chroma = k >= th;
            if (chroma) k -= th;
            bicubic_int_axis((chroma ? (f.i_first >> 1) : f.i_first) + k, d.yr, d.src_h, chroma ? chh : d.src_h, ws, c01, c23);
            const int ro = ws - (chroma ? f.cylo : f.ylo);
            const IRow en = { ro & ~3, ro & 3, c01, c23 };
            if (!chroma) ytab[k] = en;
            else cytab[k] = en;
            continue;
        }
        k -= th + (th >> 1);
        if (k < nrb_y) { // rows past the footprint repeat its last row (their H values are never used)
            const int r = min(k, max(ny - 1, 0));
            rby[k] = r * py.lp + ((py.m0 + r * py.pm) & 15);
        } else {
            k -= nrb_y;
            const int r = min(k, max(nuv - 1, 0));
            rbuv[k] = r * puv.lp + ((puv.m0 + r * puv.pm) & 15);
        }
    }
    if (d.dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // phase 1: H sums.  Work item = (tile column, four consecutive staged rows) -> one dword of the column-major H plane.
    {
        const int tw_shift = d.tx_shift + 2;
        const int ng = (ny + 3) >> 2;
        for (int it = threadIdx.x; it < (ng << tw_shift); it += nthreads) {
            const int c = it & (tw - 1), g = it >> tw_shift;
            if (f.j_first + c >= d.dst_w) continue;
            const ICol e = xtab[c];
            const int4 rb = *(const int4 *)(rby + 4 * g);
            const int rbs[4] = { rb.x, rb.y, rb.z, rb.w };
            int sm[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int A = rbs[rr] + e.off;
                const uint32_t *p = (const uint32_t *)(lds_y + (A & ~3));
                sm[rr] = cubic_sum4(__builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)A & 3u), e.c01, e.c23);
            }
            *(uint32_t *)(hy + hcol(c, d.tx) * d.hcs_y + 4 * g) = round_clamp_pack4(sm[0], sm[1], sm[2], sm[3]);
        }
        const int ngc = (nuv + 3) >> 2;
        const int cw_shift = tw_shift - 1;
        for (int it = threadIdx.x; it < (ngc << cw_shift); it += nthreads) {
            const int cp = it & ((tw >> 1) - 1), g = it >> cw_shift;
            if (f.j_first + 2 * cp >= d.dst_w) continue;
            const ICol e = cxtab[cp];
            const int4 rb = *(const int4 *)(rbuv + 4 * g);
            const int rbs[4] = { rb.x, rb.y, rb.z, rb.w };
            int su[4], sv[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int A = rbs[rr] + e.off;
                const uint32_t *p = (const uint32_t *)(lds_uv + (A & ~3));
                const uint32_t sh = (uint32_t)A & 3u;
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                cubic_sum4_uv(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh), e.c01, e.c23, su[rr], sv[rr]);
            }
            const uint32_t pu = round_clamp_pack4(su[0], su[1], su[2], su[3]), pv = round_clamp_pack4(sv[0], sv[1], sv[2], sv[3]);
            *(uint32_t *)(huv + hcol(2 * cp, d.tx) * d.hcs_uv + 4 * g) = pu;
            *(uint32_t *)(huv + hcol(2 * cp + 1, d.tx) * d.hcs_uv + 4 * g) = pv;
        }
    }
    __syncthreads();
#ifdef TSVPP_DEBUG_LDS // debugging aid (make DEBUG_LDS=1): workgroup 0 copies its whole LDS image to the buffer smuggled in the kernarg table's LAST slot --
    // only in launches that do not use that slot for a frame (a full 128-frame batch has a real output pointer there)
    if (blockIdx.x == 0 && !t.out.ext && d.n_frames < TSVPP_MAX_BATCH && t.out.v[TSVPP_MAX_BATCH - 1] != nullptr) {
        uint32_t *dst = (uint32_t *)t.out.v[TSVPP_MAX_BATCH - 1];
        const int nd = (int)(((uint8_t *)(rbuv + nrb_uv) - lds_raw + 3) / 4);
        for (int i = threadIdx.x; i < nd; i += nthreads) dst[i] = ((const uint32_t *)lds_raw)[i];
    }
#endif

    // phase 2: vertical sums of this thread's 4 columns, two output rows (one chroma row) per step
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int j0 = f.j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    const uint8_t *colY[PXW], *colC[PXW];
#pragma unroll
    for (int k = 0; k < PXW; k++) {
        colY[k] = hy + (k * d.tx + lx) * d.hcs_y;
        colC[k] = huv + (k * d.tx + lx) * d.hcs_uv;
    }
    for (int rp = 0; rp < d.rpt; rp++) {
        const int lyr = ly * d.rpt + rp, i0 = f.i_first + lyr * PXH;
        if (i0 >= d.dst_h) break;
        float cv[PXW] = { 128.0f, 128.0f, 128.0f, 128.0f }, Yf[PXH][PXW];
        if constexpr (!kLumaOnly<OUT>) {
            const IRow e = cytab[lyr];
            int sm[PXW];
#pragma unroll
            for (int k = 0; k < PXW; k++) {
                const uint32_t *p = (const uint32_t *)(colC[k] + e.al);
                sm[k] = cubic_sum4(__builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)e.sh), e.c01, e.c23);
            }
            const uint32_t w = round_clamp_pack4(sm[0], sm[1], sm[2], sm[3]);
#pragma unroll
            for (int k = 0; k < PXW; k++) cv[k] = (float)((w >> (8 * k)) & 255u);
        }
        const float Uf[2] = { cv[0], cv[2] }, Vf[2] = { cv[1], cv[3] };
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const IRow e = ytab[lyr * PXH + r];
            int sm[PXW];
#pragma unroll
            for (int k = 0; k < PXW; k++) {
                const uint32_t *p = (const uint32_t *)(colY[k] + e.al);
                sm[k] = cubic_sum4(__builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)e.sh), e.c01, e.c23);
            }
            const uint32_t w = round_clamp_pack4(sm[0], sm[1], sm[2], sm[3]);
#pragma unroll
            for (int k = 0; k < PXW; k++) Yf[r][k] = (float)((w >> (8 * k)) & 255u);
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

size_t bicubic_int_table_bytes(int tw, int th, int rows_y, int rows_uv, int hcs_y, int hcs_uv) {
    return 16 + (size_t)tw * hcs_y + (size_t)tw * hcs_uv + 16 + (size_t)(tw + tw / 2) * sizeof(ICol) + (size_t)(th + th / 2) * sizeof(IRow) +
           sizeof(int) * (size_t)(((rows_y + 3) & ~3) + ((rows_uv + 3) & ~3)) + 16;
}

hipError_t launch_bicubic_int(OutKind out, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (info) {
        info->kernel = "vpp_bicubic_int_kernel<OUT>";
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes;
        return hipSuccess;
    }
    switch (out) {
#define TSVPP_BI(O) case O: TSVPP_LAUNCH((vpp_bicubic_int_kernel<O>), grid, block, lds_bytes, stream, d, t); break;
        TSVPP_BI(O_U8_PLANAR) TSVPP_BI(O_U8_MERGED) TSVPP_BI(O_F32_PLANAR) TSVPP_BI(O_F32_MERGED) TSVPP_BI(O_NV12_U8)
        TSVPP_BI(O_NV12_F32) TSVPP_BI(O_Y800_U8) TSVPP_BI(O_Y800_F32) TSVPP_BI(O_HSV_F32)
#undef TSVPP_BI
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
