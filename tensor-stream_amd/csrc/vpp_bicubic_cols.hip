// vpp_bicubic_cols.hip -- BICUBIC, any ratio: one wave = one output tile, one lane = one output COLUMN.
//
// The reference's value is V(H(row y-1), H(row y), H(row y+1), H(row y+2)) with H the horizontal 4-tap Keys sum (a = -0.75)
// of ONE source row, rounded and clamped to a byte, and V the same sum down the column (src/Resize.cu:27-91, 314-357).  H
// depends only on (source row, output column) and neighbouring output rows share most of their source rows, so the
// workgroup kernels of rounds 1 / 2 evaluated H once per (staged row, tile column) into an LDS plane -- behind a
// whole-workgroup pipeline: stage the footprint, barrier, tables, H phase, barrier, V phase.  Measured
// (profiles/r02_bicubic_fallback_bound.txt, r02_bicubic_pmc.txt): VALU ~30 % busy, four workgroups per CU, 35 % of the LDS
// cycles bank conflicts -- bound by that structure, not by arithmetic or HBM.
//
// Here no workgroup barrier exists.  A wave owns 64 output columns x R output rows:
//   phase 1  lane j walks DOWN the source rows of the tile, four rows per step.  The row segments the wave needs are fetched
//            by LDS-DMA (global_load_lds_dword: lane l fetches dword l of the segment, one instruction per row, no VGPRs) into
//            a wave-private ring of 16 rows, so that 12-16 rows are in flight per wave while it computes; at horizontal
//            ratios >= 3.8 (a lane's taps are then disjoint from its neighbours') every lane loads its own eight bytes
//            straight into registers, one group ahead.  A lane picks its four taps out of two aligned dwords with one
//            v_perm_b32 -- the reference's edge rule (the +1 AND +2 taps collapse, src/Resize.cu:32-43) lives in the selector --
//            and writes H of four consecutive rows as ONE dword of its private column in a wave-private, column-major LDS
//            plane (column stride = an odd number of dwords: no bank conflicts);
//   phase 2  the four vertical taps of an output row are four consecutive bytes of that column: two dword reads,
//            v_alignbyte_b32, v_perm_b32 (vertical edge rule), the same 4-tap sum; per-row parameters are wave-uniform: each
//            lane evaluates ONE row's coordinates and coefficients, the loop fetches them with v_readlane_b32 (no table in
//            LDS, no barrier);
//   chroma   is the same two phases on the interleaved UV plane with lane = (pair column, U | V) and a tap stride of 2;
//   colour   the resized tile goes through a small wave-private byte tile into the usual 2 x 4 thread tiles (colour
//            conversion, every output flavour, vector stores: color_store_tile).
// LDS operations of one wave execute in order, so nothing but wave-level fences separates the phases.  Rows the vertical
// taps skip (ratios >= 4) are not evaluated: there the H plane holds the four taps of each output row instead of a contiguous
// run of source rows ("sparse" mode).
//
// Arithmetic.  The reference evaluates sum_k c_k(w) p_k in fp64 and rounds half away from zero.  Here the coefficients are
// evaluated in fp64 ONCE per column / row and quantised to C_k = rint(c_k 2^22); the sum of C_k p_k is then exact in 32-bit
// integers (|sum| <= 255 * 1.375 * 2^22 < 2^31) and differs from 2^22 times the reference's sum by at most 4 * 255 * 2^-23
// * 2^22 = 510 units.  Whenever the integer sum is further than that from a rounding tie, (S + 2^21) >> 22 IS the
// reference's value; the few lanes within reach of a tie redo the sum in fp64 exactly as the reference does (~0.03 % of
// the sums).  The integer sum is three v_dot4_u32_u8 on the four packed taps: |C_k| = l2 2^16 + l1 2^8 + l0 in base-256
// digits, and since Keys' coefficients have fixed signs (c0, c3 <= 0 <= c1, c2) the two negative ones are applied to the
// complemented taps 255 - t (one XOR on the packed taps; the constant 255 (|C_0| + |C_3|) is part of the bias) -- 6 VALU
// operations per sum instead of 15 in packed fp32.  EXACT = true (host: every weight of the request is a
// multiple of 1/16, so every C_k is exact): no tie test at all, a tie rounds up as round() does for positive values and
// negative values clamp to 0 either way.
#include "vpp_device.h"
#include "vpp_r32_store.h"

#include <vector>

#pragma clang fp contract(off)

namespace tsvpp {

// The tables are written by the host before the launch and never by a kernel: read through the CONSTANT address space, loads
// at wave-uniform addresses (the rows' entries) become scalar loads (s_load_dwordx8: no VALU, no VGPRs).
typedef const __attribute__((address_space(4))) BcEntry *BcTab;
typedef int bc_i4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) int *BcRows;     // row blocks (see bc_row_ws)
typedef const __attribute__((address_space(4))) bc_i4 *BcRows4;
__device__ __forceinline__ BcTab bc_const(const BcEntry *p) { return (BcTab)(uintptr_t)p; }
__device__ __forceinline__ BcEntry bc_ld(BcTab p) {
    BcEntry e;
    e.ws = p->ws; e.sel = p->sel; e.l0 = p->l0; e.l1 = p->l1; e.l2 = p->l2; e.bias = p->bias; e.w = p->w; e.maxoff = p->maxoff;
    return e;
}

typedef uint32_t u32x2a4c __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3a4c __attribute__((ext_vector_type(3), aligned(4)));

constexpr int BC_SHIFT = 22;             // coefficient scale 2^22
constexpr int BC_TIE = 560;              // tie zone in units of 2^-22 (error bound 510)

// byte k of a dword as a float: the compiler selects v_cvt_f32_ubyte<k>
__device__ __forceinline__ float ub0(uint32_t v) { return (float)(v & 255u); }
__device__ __forceinline__ float ub1(uint32_t v) { return (float)((v >> 8) & 255u); }
__device__ __forceinline__ float ub2(uint32_t v) { return (float)((v >> 16) & 255u); }
__device__ __forceinline__ float ub3(uint32_t v) { return (float)(v >> 24); }

// One output index along one axis -> its table entry (vpp_kernels.h: BcEntry).  Evaluated on the HOST, once per request
// (bicubic_cols_tables): round 3's first version evaluated it per wave and spent a third of its VALU instructions here.
// `clamp_limit`: the size the coordinate clamps use (the LUMA size on both grids, src/Resize.cu:325-347); `tap_limit`: the
// size the edge rule uses, in samples of this grid.
static BcEntry bc_axis(int idx, float ratio, int clamp_limit, int tap_limit) {
    BcEntry a;
    int p;
    double w;
    bicubic_axis(idx, ratio, clamp_limit, p, w);
    int lo, hi;
    bicubic_offsets(p, 1, tap_limit, lo, hi);
    a.ws = p - lo;
    a.sel = (uint32_t)lo << 8 | (uint32_t)(lo + hi) << 16 | (uint32_t)(lo + 2 * hi) << 24;
    a.maxoff = lo + 2 * hi;
    a.w = (float)w; // exact: the fraction of a float coordinate
    double c[4];
    cubic_coeffs(w, c);
    // Keys' coefficients have fixed signs (a = -0.75, w in [0, 1)): c0, c3 <= 0 <= c1, c2.  With m_k = |C_k| as three planes of
    // UNSIGNED base-256 digits and the taps 0 and 3 complemented (255 - t: one XOR 0xFF0000FF on the packed taps),
    //     sum_k C_k t_k = sum_k m_k x_k - 255 (m_0 + m_3)      -- three v_dot4_u32_u8, the constant is part of the bias
    uint32_t d0 = 0, d1 = 0, d2 = 0;
    int neg = 0;
    bool exact = true; // every coefficient is a whole number of 2^-22 units (weights that are multiples of 1/16: 1/2, 1/4, 3/8 ...)
    for (int k = 0; k < 4; k++) {
        if (c[k] * 4194304.0 != __builtin_rint(c[k] * 4194304.0)) exact = false;
        const int C = (int)__builtin_rint(c[k] * 4194304.0);
        const int m = (k == 0 || k == 3) ? -C : C; // >= 0, <= 2^22
        d0 |= (uint32_t)(m & 255) << (8 * k);
        d1 |= (uint32_t)((m >> 8) & 255) << (8 * k);
        d2 |= (uint32_t)((m >> 16) & 255) << (8 * k);
        if (k == 0 || k == 3) neg += m;
    }
    a.l0 = (int)d0;
    a.l1 = (int)d1;
    a.l2 = (int)d2;
    a.bias = (1 << (BC_SHIFT - 1)) - 255 * neg;
    // An index with exact coefficients needs no tie test: its integer sum IS 2^22 times the reference's fp64 sum, and (S + 2^21) >> 22 rounds
    // a tie the way round() does.  That matters because such indices are exactly where ties are COMMON, not rare: at w = 1/2 the sum is a
    // multiple of 1/32 and one in 32 is a true tie -- at ratio 2 / 3 (720p -> 1080p) a third of all columns and rows, so that nearly every
    // wave-wide step found a lane inside the tie zone and ran the fp64 path (measured: 2785 VALU instructions per wave against 1650 in the
    // listing).  Flag: the sign bit of `maxoff`.
    if (exact) a.maxoff |= (int)0x80000000;
    return a;
}

// 2^22 x the 4-tap sum over the packed taps (tap k in byte k) + 2^21: the value is clamp(that >> 22, 0, 255)
__device__ __forceinline__ int bc_isum(uint32_t taps, int l0, int l1, int l2, int bias) {
    const uint32_t x = taps ^ 0xff0000ffu; // taps 0 and 3 complemented: their coefficients are <= 0
    // ((d2 << 8) + d1 << 8) + d0 + bias as a chain through the dot products' accumulator inputs: dot, shift, dot, shift-add, dot
    // (written as sums the compiler re-associates it into two shifts + v_add3 and a separate bias add)
    const uint32_t s2 = __builtin_amdgcn_udot4(x, (uint32_t)l2, 0u, false);
    const uint32_t s1 = __builtin_amdgcn_udot4(x, (uint32_t)l1, s2 << 8, false);
    return (int)__builtin_amdgcn_udot4(x, (uint32_t)l0, (s1 << 8) + (uint32_t)bias, false);
}
// Four sums at once, STAGE by stage (round 6).  gfx950 needs wait states between a v_dot4 and the instruction that consumes its result; with the four chains of a group
// written one after the other (bc_isum x 4) the compiler kept them sequential and filled every gap with `s_nop 2` -- 49 scalar-slot instructions per 4 sums in the ISA of round 5,
// a dependent chain of ~40 cycles per sum.  Written stage-wise the four independent chains fill each other's wait states.
__device__ __forceinline__ void bc_isum4(const uint32_t (&tp)[4], const int (&l0)[4], const int (&l1)[4], const int (&l2)[4], const int (&bias)[4], int (&s)[4]) {
    uint32_t x[4], a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = tp[k] ^ 0xff0000ffu;
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = __builtin_amdgcn_udot4(x[k], (uint32_t)l2[k], 0u, false);
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = __builtin_amdgcn_udot4(x[k], (uint32_t)l1[k], a[k] << 8, false);
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = (int)__builtin_amdgcn_udot4(x[k], (uint32_t)l0[k], (b[k] << 8) + (uint32_t)bias[k], false);
}
__device__ __forceinline__ void bc_isum4(const uint32_t (&tp)[4], int l0, int l1, int l2, int bias, int (&s)[4]) {
    const int a0[4] = { l0, l0, l0, l0 }, a1[4] = { l1, l1, l1, l1 }, a2[4] = { l2, l2, l2, l2 }, ab[4] = { bias, bias, bias, bias };
    bc_isum4(tp, a0, a1, a2, ab, s);
}
// distance-to-tie key: small (< 2 BC_TIE << 10) iff the sum is within BC_TIE units of a rounding tie
__device__ __forceinline__ uint32_t bc_tie_key(int s) { return (uint32_t)(s + BC_TIE) << (32 - BC_SHIFT); }
constexpr uint32_t BC_TIE_KEY = (2u * BC_TIE) << (32 - BC_SHIFT);
// the reference's own evaluation (fp64, round half away, clamp) of one sum
__device__ __forceinline__ uint32_t bc_exact(uint32_t taps, float w) {
    double c[4];
    cubic_coeffs((double)w, c);
    return (uint32_t)cubic4(c, (int)(taps & 255u), (int)((taps >> 8) & 255u), (int)((taps >> 16) & 255u), (int)(taps >> 24));
}
// Four sums -> four bytes clamp(s >> 22, 0, 255), s0 in byte 0.  gfx950's V_ASHR_PK_U8_I32 shifts, saturates and packs two
// values into the LOW half of its destination and keeps the upper half (measured; see vpp_bicubic_int.hip): pair (s2, s3) is
// packed first and shifted up, pair (s0, s1) then lands below it.
__device__ __forceinline__ uint32_t bc_pack4(int s0, int s1, int s2, int s3) {
    uint32_t hi;
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 22" : "=v"(hi) : "v"(s2), "v"(s3));
    uint32_t r = hi << 16;
    asm("s_nop 2\n\tv_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(r) : "v"(s0), "v"(s1));
    return r;
}
// four sums -> four result bytes; the (rare) sums within reach of a tie are redone as the reference does them
template <bool EXACT>
__device__ __forceinline__ uint32_t bc_finish4(const int s[4], const uint32_t tp[4], float w0, float w1, float w2, float w3, uint32_t exm) {
    uint32_t r = bc_pack4(s[0], s[1], s[2], s[3]);
    if constexpr (!EXACT) {
        // exm: all ones for a lane whose coefficients are exact (bc_axis): its keys never fall below the threshold
        const uint32_t k0 = bc_tie_key(s[0]) | exm, k1 = bc_tie_key(s[1]) | exm, k2 = bc_tie_key(s[2]) | exm, k3 = bc_tie_key(s[3]) | exm;
        if (min(min(k0, k1), min(k2, k3)) < BC_TIE_KEY) { // rarely taken
            if (k0 < BC_TIE_KEY) r = (r & 0xffffff00u) | bc_exact(tp[0], w0);
            if (k1 < BC_TIE_KEY) r = (r & 0xffff00ffu) | bc_exact(tp[1], w1) << 8;
            if (k2 < BC_TIE_KEY) r = (r & 0xff00ffffu) | bc_exact(tp[2], w2) << 16;
            if (k3 < BC_TIE_KEY) r = (r & 0x00ffffffu) | bc_exact(tp[3], w3) << 24;
        }
    }
    return r;
}
__device__ __forceinline__ uint32_t rep4(uint32_t b) { return __builtin_amdgcn_perm(0u, b, 0u); } // byte 0 into all four bytes

// packed taps of one row from the dwords around the window: `sh` = byte offset of the window start in dw[0]; STEP 1: `selx` =
// tap selector + sh in every byte; STEP 2 (chroma, taps 2 bytes apart): `selx` = twice the tap selector
template <int STEP, int NDW>
__device__ __forceinline__ uint32_t bc_taps(const uint32_t (&dw)[NDW], uint32_t sh, uint32_t selx) {
    if constexpr (STEP == 1) {
        return __builtin_amdgcn_perm(dw[1], dw[0], selx);
    } else {
        const uint32_t a0 = __builtin_amdgcn_alignbyte(dw[1], dw[0], sh), a1 = __builtin_amdgcn_alignbyte(dw[2], dw[1], sh);
        return __builtin_amdgcn_perm(a1, a0, selx);
    }
}

// Row parameters (wave-uniform: scalar loads).  Rows come in blocks of four: 32 ints = [ws x4 | sel x4 | l0 x4 | l1 x4 | l2 x4 |
// bias x4 | w x4 | exact x4], so that ONE base address serves a whole phase-2 step.  `rb` = the block of the tile's first row.
__device__ __forceinline__ int bc_row_ws(BcRows rb, int i) { return rb[(i >> 2) * 32 + (i & 3)]; }
__device__ __forceinline__ uint32_t bc_row_sel(BcRows rb, int i) { return (uint32_t)rb[(i >> 2) * 32 + 4 + (i & 3)]; }

// One plane (luma: STEP 1, lane = column; chroma: STEP 2, lane = (pair column, component)) through phases 1 and 2.
// ax: the lane's column; rb: the tile's first row block; hcol: this lane's column of the wave's H plane; ring: the wave's
// LDS-DMA ring (DMA mode); res: the wave's result tile (row-major, 64 bytes per row).
template <bool EXACT, bool SPARSE, int STEP>
__device__ __forceinline__ void bc_plane(const uint8_t *frame_plane, int pitch, int rows_in_plane, int row_bytes, const BcEntry &ax, int comp, bool dma, int lanes_per_row,
                                         uint8_t *ring, uint8_t *hcol, uint8_t *res, int lane, int nout, BcRows rb) {
    constexpr int NDW = STEP == 1 ? 2 : 3;
    // the plane pointer rounded down to 16 bytes (wave-uniform) + the bytes it was rounded by: every offset below is >= 0
    const uint32_t pm = (uint32_t)((uintptr_t)frame_plane & 15);
    const uint8_t *plane = frame_plane - pm;
    const int ylo = bc_row_ws(rb, 0), yhi = min(bc_row_ws(rb, nout - 1) + 3, rows_in_plane - 1); // dense: from the first window's start to the last window's end
    const int ng = SPARSE ? nout : ((yhi - ylo + 1 + 3) >> 2);
    const int xb = STEP * ax.ws + comp; // the lane's window start, byte column of the plane
    const uint32_t exm = (uint32_t)(ax.maxoff >> 31); // all ones: this lane's coefficients are exact (no tie test)
    const uint32_t tapsel = STEP == 1 ? ax.sel : ax.sel + ax.sel;
    // ---- phase 1: H of every needed source row of this lane's column, four rows per dword of the column
    if (dma && (pitch & 15) == 0) {
        // LDS-DMA: ONE global_load_lds_dwordx4 fetches a BATCH of row segments -- L lanes per row, 16 bytes each, L = the 16-byte chunks
        // a segment needs at this ratio (host: 64 columns x ratio + window + misalignment), 4 * floor(16 / L) rows per batch (4, 8 or
        // 16: one, two or four groups) -- into a ring of three batches of 1 KiB.  Lane addresses are loop-invariant up to the batch's
        // first row.  (The first version fetched 256 bytes per row whatever the ratio: 2.5 x the needed bytes at ratio 1.5, and the
        // launch ran 40 us longer than with its reads served from cache.)
        // WIDE (round 5; dense mode, 17 .. 32 chunks per segment: horizontal ratios 3.7 .. 7.4, which took per-lane loads before): a batch is ONE group of four rows
        // fetched by TWO instructions of two rows each, a ring slot is 4 L chunks
        const bool wide = !SPARSE && lanes_per_row > 16;
        const int L = SPARSE ? 16 : lanes_per_row, rpb = (SPARSE || wide) ? 4 : 4 * (16 / L), gpb = rpb >> 2; // rows, groups per batch
        const int xb0 = __builtin_amdgcn_readlane(xb, 0);
        const uint32_t seg0 = (pm + (uint32_t)xb0) & ~15u;           // the segment's first 16-byte chunk (byte column from `plane`)
        const uint32_t a_c = (uint32_t)(xb - xb0) + ((pm + (uint32_t)xb0) & 15u); // the lane's window start inside a ring row
        const uint32_t al_c = a_c & ~3u, sh_c = a_c & 3u, selx_c = STEP == 1 ? tapsel + rep4(sh_c) : tapsel;
        const uint32_t last_chunk = ((uint32_t)(rows_in_plane - 1) * (uint32_t)pitch + pm + (uint32_t)row_bytes - 1u) & ~15u; // the plane's last valid chunk
        const int lrow = (lane * (65536 / L + 1)) >> 16, lchunk = lane - lrow * L; // lane / L, lane % L (L <= 16, lane < 64: exact)
        const bool lane_on = wide ? lrow < 2 : lrow < rpb;
        const uint32_t lconst = seg0 + 16u * (uint32_t)lchunk + (SPARSE ? 0u : (uint32_t)lrow * (uint32_t)pitch);
        const int rstride = 16 * L; // ring row pitch
        const int slot_bytes = wide ? 4 * rstride : 1024;
        const int nb = (ng + gpb - 1) / gpb;
        auto issue = [&](int bt, int slot) {
            uint32_t voff;
            if constexpr (SPARSE) { // one output row per batch: lane group k fetches its tap k
                const uint32_t tapoff = __builtin_amdgcn_ubfe(bc_row_sel(rb, bt), 8u * (uint32_t)lrow, 8u);
                voff = ((uint32_t)bc_row_ws(rb, bt) + tapoff) * (uint32_t)pitch + lconst;
            } else {
                voff = (uint32_t)(ylo + rpb * bt) * (uint32_t)pitch + lconst; // rows past yhi are real rows of the plane (or clamped below): never used
            }
            uint8_t *dst = ring + slot * slot_bytes; // wave-uniform
            const uint32_t v0 = min(voff, last_chunk);
            if (lane_on)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane + v0), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            if constexpr (!SPARSE) {
                if (wide) { // rows 2, 3 of the group: the second instruction, landing behind the first one's two rows
                    const uint32_t v1 = min(voff + 2u * (uint32_t)pitch, last_chunk);
                    if (lane_on)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(plane + v1),
                                                         (__attribute__((address_space(3))) void *)(dst + 2 * rstride), 16, 0, 0);
                }
            }
        };
        issue(0, 0);
        if (nb > 1) issue(1, 1);
        if (nb > 2) issue(2, 2);
        int slot = 0, bt = 0, gin = 0; // ring slot and index of the current batch, group inside it
        for (int g = 0; g < ng; g++) {
            if (gin == 0) { // batch bt has landed when at most the later batches' loads are outstanding (one per batch; WIDE: two)
                if (wide) {
                    if (bt + 2 < nb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (bt + 1 < nb) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    if (bt + 2 < nb) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (bt + 1 < nb) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            const uint8_t *p0 = ring + slot * slot_bytes + 4 * gin * rstride + al_c;
            uint32_t tp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t *p = (const uint32_t *)(p0 + k * rstride);
                uint32_t dw[NDW];
#pragma unroll
                for (int q = 0; q < NDW; q++) dw[q] = p[q];
                tp[k] = bc_taps<STEP, NDW>(dw, sh_c, selx_c);
            }
            if (++gin == gpb) { // the batch's last group: its slot is free once the reads above have returned
                if (bt + 3 < nb) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue(bt + 3, slot);
                }
                gin = 0;
                bt++;
                slot = slot == 2 ? 0 : slot + 1;
            }
            int s[4];
            bc_isum4(tp, ax.l0, ax.l1, ax.l2, ax.bias, s);
            *(uint32_t *)(hcol + 4 * g) = bc_finish4<EXACT>(s, tp, ax.w, ax.w, ax.w, ax.w, exm);
        }
    } else {
        // direct mode (horizontal ratios >= 3.7, or a pitch that is no multiple of 16): every lane loads the NDW dwords around its own
        // window.  Groups that do not touch the plane's last row (all but the bottom tiles' last) run software-pipelined, one group
        // ahead, with unconditional vector loads (a fixed number per group: the compiler's wait counts stay exact); the others load
        // dword by dword and never read past the last needed byte's dword.
        const uint32_t xoff = pm + (uint32_t)xb;
        const int xneed = STEP * (ax.maxoff & 0x7fffffff);
        const bool aligned = (pitch & 3) == 0; // every row then has the same misalignment: lane offsets and selectors are loop-invariant
        const uint32_t sh_c = xoff & 3u, selx_c = STEP == 1 ? tapsel + rep4(sh_c) : tapsel;
        auto group_rows = [&](int g, int (&rr)[4]) {
            if constexpr (SPARSE) { // the four taps of output row g, in tap order
                const int ws = bc_row_ws(rb, g);
                const uint32_t sel = bc_row_sel(rb, g);
#pragma unroll
                for (int k = 0; k < 4; k++) rr[k] = ws + (int)((sel >> (8 * k)) & 255u);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) rr[k] = min(ylo + 4 * g + k, yhi);
            }
        };
        auto row_ptr = [&](int row, uint32_t &sh) { // the aligned dword holding the lane's window start in `row`, its byte offset there
            const uint32_t a = (uint32_t)row * (uint32_t)pitch + xoff; // aligned: a & 3 == sh_c whatever the row
            sh = aligned ? sh_c : (a & 3u);
            // aligned: (row * pitch + xoff) & ~3 == row * pitch + (xoff & ~3): a wave-uniform row pointer + a loop-invariant lane offset
            return aligned ? (const uint32_t *)(plane + (size_t)((uint32_t)row * (uint32_t)pitch) + (xoff & ~3u)) : (const uint32_t *)(plane + (a & ~3u));
        };
        auto fetch = [&](int g, uint32_t (&dw)[4][NDW], uint32_t (&sh)[4]) { // rows below the plane's last: the bytes after the window belong to the plane
            int rr[4];
            group_rows(g, rr);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t *p = row_ptr(rr[k], sh[k]);
                if constexpr (NDW == 2) {
                    const u32x2a4c v = *(const u32x2a4c *)p;
                    dw[k][0] = v.x; dw[k][1] = v.y;
                } else {
                    const u32x3a4c v = *(const u32x3a4c *)p;
                    dw[k][0] = v.x; dw[k][1] = v.y; dw[k][2] = v.z;
                }
            }
        };
        auto finish = [&](int g, const uint32_t (&dw)[4][NDW], const uint32_t (&sh)[4]) {
            uint32_t tp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) tp[k] = bc_taps<STEP, NDW>(dw[k], sh[k], aligned ? selx_c : (STEP == 1 ? tapsel + rep4(sh[k]) : tapsel));
            int s[4];
            bc_isum4(tp, ax.l0, ax.l1, ax.l2, ax.bias, s);
            *(uint32_t *)(hcol + 4 * g) = bc_finish4<EXACT>(s, tp, ax.w, ax.w, ax.w, ax.w, exm);
        };
        // leading groups whose rows all lie above the plane's last row
        int ngf;
        if constexpr (SPARSE) ngf = (bc_row_ws(rb, nout - 1) + 3 < rows_in_plane - 1) ? ng : 0;
        else ngf = (yhi < rows_in_plane - 1) ? ng : max(min((rows_in_plane - 1 - ylo) >> 2, ng), 0);
        if (ngf > 0) {
            uint32_t cur[4][NDW], csh[4];
            fetch(0, cur, csh);
            for (int g = 0; g + 1 < ngf; g++) {
                uint32_t nxt[4][NDW], nsh[4];
                fetch(g + 1, nxt, nsh); // one group ahead
                finish(g, cur, csh);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    csh[k] = nsh[k];
#pragma unroll
                    for (int q = 0; q < NDW; q++) cur[k][q] = nxt[k][q];
                }
            }
            finish(ngf - 1, cur, csh);
        }
        for (int g = ngf; g < ng; g++) {
            int rr[4];
            group_rows(g, rr);
            uint32_t dw[4][NDW], sh[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t *p = row_ptr(rr[k], sh[k]);
                const int hi = (int)sh[k] + xneed; // a dword past the last needed byte is re-pointed at the first
#pragma unroll
                for (int q = 0; q < NDW; q++) dw[k][q] = p[(rr[k] < rows_in_plane - 1 || 4 * q <= hi) ? q : 0];
            }
            finish(g, dw, sh);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2: vertical sums, four output rows per step.  The rows' parameters are wave-uniform: three scalar loads fetch
    // them for the step's four rows (the table is padded by a block, so a partial last step reads valid entries it does not store).
    for (int i = 0; i < nout; i += 4) {
        const BcRows blk = rb + (i >> 2) * 32;
        const bc_i4 ws4 = *(BcRows4)(blk), sel4 = *(BcRows4)(blk + 4), a4 = *(BcRows4)(blk + 8), b4 = *(BcRows4)(blk + 12), c4 = *(BcRows4)(blk + 16),
                    bias4 = *(BcRows4)(blk + 20), ex4 = *(BcRows4)(blk + 28);
        uint32_t tp[4];
        int s[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int off = SPARSE ? 4 * (i + e) : max(ws4[e] - ylo, 0); // (rows past the tile's last one: anywhere inside the column)
            const uint32_t *p = (const uint32_t *)(hcol + (off & ~3));
            const uint32_t win = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)off & 3u);
            tp[e] = SPARSE ? win : __builtin_amdgcn_perm(0u, win, (uint32_t)sel4[e]);
        }
        {
            const int q0[4] = { a4[0], a4[1], a4[2], a4[3] }, q1[4] = { b4[0], b4[1], b4[2], b4[3] }, q2[4] = { c4[0], c4[1], c4[2], c4[3] },
                      qb[4] = { bias4[0], bias4[1], bias4[2], bias4[3] };
            bc_isum4(tp, q0, q1, q2, qb, s);
        }
        uint32_t r = bc_pack4(s[0], s[1], s[2], s[3]);
        if constexpr (!EXACT) {
            // (rows with exact coefficients: mask of ones from the table, see bc_axis)
            const uint32_t k0 = bc_tie_key(s[0]) | (uint32_t)ex4[0], k1 = bc_tie_key(s[1]) | (uint32_t)ex4[1], k2 = bc_tie_key(s[2]) | (uint32_t)ex4[2],
                           k3 = bc_tie_key(s[3]) | (uint32_t)ex4[3];
            if (min(min(k0, k1), min(k2, k3)) < BC_TIE_KEY) { // rarely taken
                // (four scalar dword loads: ROCm 7.2's clang, given ONE dwordx4 load here, used element 0 for all four weights)
                const BcRows wp = blk + 24;
                if (k0 < BC_TIE_KEY) r = (r & 0xffffff00u) | bc_exact(tp[0], __builtin_bit_cast(float, wp[0]));
                if (k1 < BC_TIE_KEY) r = (r & 0xffff00ffu) | bc_exact(tp[1], __builtin_bit_cast(float, wp[1])) << 8;
                if (k2 < BC_TIE_KEY) r = (r & 0xff00ffffu) | bc_exact(tp[2], __builtin_bit_cast(float, wp[2])) << 16;
                if (k3 < BC_TIE_KEY) r = (r & 0x00ffffffu) | bc_exact(tp[3], __builtin_bit_cast(float, wp[3])) << 24;
            }
        }
        res[i * 64 + lane] = (uint8_t)r;
        if (i + 1 < nout) res[(i + 1) * 64 + lane] = (uint8_t)(r >> 8);
        if (i + 2 < nout) res[(i + 2) * 64 + lane] = (uint8_t)(r >> 16);
        if (i + 3 < nout) res[(i + 3) * 64 + lane] = (uint8_t)(r >> 24);
    }
}

template <int OUT, bool EXACT, bool SPARSE>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bicubic_cols_kernel(const LaunchDesc d, const FrameTable t) {
    using T = typename OutT<OUT>::type;
    const TileId id = decode_tile(d);
    if (!id.valid) return;
    const int lane = (int)(threadIdx.x & 63u), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int R = 8 * d.rpt; // output rows of the tile (multiple of 8, <= 32)
    const int j_first = tile_col0(d, id.tx, 256) + wave * 64, i_first = id.ty * R;
    if (j_first >= d.dst_w) return; // (no workgroup barrier anywhere in this kernel)
    const int nrows = min(R, d.dst_h - i_first), ncrows = nrows >> 1;
    const bool dma = d.bc_dma != 0;

    // wave-private LDS: DMA ring, H plane (column-major, one column per lane; luma then chroma), result tiles (row-major)
    uint8_t *wl = lds_raw + wave * d.bc_wave_bytes;
    uint8_t *ring = wl, *hcol = wl + d.bc_ring_bytes + lane * d.hcs_y;
    uint8_t *yt = wl + d.bc_ring_bytes + 64 * d.hcs_y, *uvt = yt + 64 * R;

    // the request's tables (host-built): luma columns | chroma pair columns | luma row blocks | chroma row blocks
    const BcTab col_y = bc_const(d.bc_tab), col_c = col_y + d.dst_w;
    const BcRows row_y = (BcRows)(col_c + (d.dst_w >> 1)), row_c = row_y + 32 * d.bc_npy;
    { // luma: lane = column (columns past the frame repeat the last one; never stored)
        const BcEntry ax = bc_ld(col_y + min(j_first + lane, d.dst_w - 1));
        bc_plane<EXACT, SPARSE, 1>(t.y[id.frame], d.pitch_y, d.src_h, d.src_w, ax, 0, dma, d.bc_dma, ring, hcol, yt, lane, nrows, row_y + (i_first >> 2) * 32);
    }
    if constexpr (!kLumaOnly<OUT>) { // chroma: lane = (pair column, component); taps in pair units, 2 bytes apart
        const BcEntry ax = bc_ld(col_c + min((j_first >> 1) + (lane >> 1), (d.dst_w >> 1) - 1));
        bc_plane<EXACT, SPARSE, 2>(t.uv[id.frame], d.pitch_uv, d.src_h >> 1, d.src_w, ax, lane & 1, dma, d.bc_dma, ring, hcol, uvt, lane, ncrows, row_c + (i_first >> 3) * 32);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // uint8 outputs whose width is a multiple of 8 and height a multiple of 4 (host: LaunchDesc::bc_u8x): the output side of the streaming kernels (vpp_r32_store.h) --
    // a lane takes 8 columns x 4 rows of the result tiles as packed dwords, 8-byte planar stores / merged rows exchanged into 16-byte stores, instead of 4 x 2-pixel
    // thread tiles with 4-byte stores (round 5; 720p -> 1080p uint8: profiles/r05_bicubic_cols_u8_ab.txt)
    if constexpr (OUT == O_U8_PLANAR || OUT == O_U8_MERGED || OUT == O_NV12_U8 || OUT == O_Y800_U8) {
        if (d.bc_u8x) {
            const int gx = lane & 7, gy = lane >> 3; // 8 column groups x 8 row groups of a 64 x 32 tile
            const int j0 = j_first + 8 * gx, r0 = 4 * gy;
            if (j0 >= d.dst_w || r0 >= nrows) return;
            uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t *p = (const uint32_t *)(yt + (r0 + r) * 64 + 8 * gx);
                ylo[r] = p[0];
                yhi[r] = p[1];
            }
            if constexpr (!kLumaOnly<OUT>) {
#pragma unroll
                for (int rc = 0; rc < 2; rc++) {
                    const uint32_t *p = (const uint32_t *)(uvt + ((r0 >> 1) + rc) * 64 + 8 * gx);
                    clo[rc] = p[0];
                    chi[rc] = p[1];
                }
            }
            const int run_a = min(8, (d.dst_w - j_first) >> 3);
            r32_store_tile<OUT>(d, (uint8_t *)t.out[id.frame], ylo, yhi, clo, chi, i_first + r0, j0, gx, run_a);
            return;
        }
    }
    // colour conversion + stores: 16 x 4 thread tiles of 4 x 2 pixels per 8-row slab
    const int lx = lane & 15, ly = lane >> 4;
    const int j0 = j_first + lx * PXW;
    if (j0 >= d.dst_w) return;
    if (is_row_tail(d, j0)) return; // the two-column row tail belongs to the tail launch (launch_fused)
    for (int s = 0; s < d.rpt; s++) {
        const int r0 = s * 8 + ly * PXH, i0 = i_first + r0;
        if (i0 >= d.dst_h) break;
        float Uf[2] = { 128.0f, 128.0f }, Vf[2] = { 128.0f, 128.0f }, Yf[PXH][PXW];
#pragma unroll
        for (int r = 0; r < PXH; r++) {
            const uint32_t v = *(const uint32_t *)(yt + (r0 + r) * 64 + lx * PXW);
            Yf[r][0] = ub0(v);
            Yf[r][1] = ub1(v);
            Yf[r][2] = ub2(v);
            Yf[r][3] = ub3(v);
        }
        if constexpr (!kLumaOnly<OUT>) {
            const uint32_t c = *(const uint32_t *)(uvt + (r0 >> 1) * 64 + lx * PXW);
            Uf[0] = ub0(c);
            Vf[0] = ub1(c);
            Uf[1] = ub2(c);
            Vf[1] = ub3(c);
        }
        color_store_tile<OUT, true>(Yf, Uf, Vf, d, (T *)t.out[id.frame], i0, j0, PXW);
    }
}

int bicubic_cols_rows_padded(int n) { return (n + 3) / 4 + 1; } // row blocks (of four rows) for n rows: one spare block

// The request's tables, evaluated on the host with the functions the kernels of rounds 1 / 2 evaluated per tile
// (bicubic_axis, bicubic_offsets, cubic_coeffs: plain IEEE operations, identical on host and device) and cached in the
// context: (dst_w + dst_w / 2) column records of 32 bytes + 7 ints per row -- ~240 KiB for a 4K output.
const BcEntry *bicubic_cols_tables(const LaunchDesc &d, hipStream_t stream, bool may_build) {
    GeoCache *cache = d.geo_cache;
    if (!cache) return nullptr;
    GeoKey key;
    memset(&key, 0, sizeof(key));
    uint32_t xb, yb;
    memcpy(&xb, &d.xr, 4);
    memcpy(&yb, &d.yr, 4);
    const int kv[16] = { 2 /* kind: BICUBIC tables */, d.src_w, d.src_h, d.dst_w, d.dst_h, (int)xb, (int)yb, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    memcpy(key.v, kv, sizeof(kv));
    std::lock_guard<std::mutex> lk(cache->mu);
    auto it = cache->map.find(key);
    if (it == cache->map.end()) {
        if (!may_build) return nullptr;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        // (the NULL stream is never queried: asking the legacy stream while another stream captures in global mode invalidates that capture)
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
        if (!geo_cache_make_room(cache)) return nullptr;
        std::vector<uint8_t> buf;
        auto append = [&](const void *p, size_t n) {
            const size_t at = buf.size();
            buf.resize(at + n);
            memcpy(buf.data() + at, p, n);
        };
        for (int j = 0; j < d.dst_w; j++) { // luma columns, then chroma pair columns: records of 32 bytes
            const BcEntry e = bc_axis(j, d.xr, d.src_w, d.src_w);
            append(&e, sizeof(e));
        }
        for (int j = 0; j < (d.dst_w >> 1); j++) {
            const BcEntry e = bc_axis(j, d.xr, d.src_w, d.src_w >> 1);
            append(&e, sizeof(e));
        }
        auto row_blocks = [&](int n, int tap_limit) { // blocks of four rows x eight fields (rows past the last repeat it)
            const int nb = bicubic_cols_rows_padded(n);
            std::vector<int> a((size_t)32 * nb, 0);
            for (int i = 0; i < 4 * nb; i++) {
                const BcEntry e = bc_axis(i < n ? i : n - 1, d.yr, d.src_h, tap_limit);
                int wbits;
                memcpy(&wbits, &e.w, 4);
                const int v[8] = { e.ws, (int)e.sel, e.l0, e.l1, e.l2, e.bias, wbits, e.maxoff >> 31 /* exact: all ones */ };
                for (int f = 0; f < 8; f++) a[(size_t)(i >> 2) * 32 + 4 * f + (i & 3)] = v[f];
            }
            append(a.data(), a.size() * sizeof(int));
        };
        row_blocks(d.dst_h, d.src_h);
        row_blocks(d.dst_h >> 1, d.src_h >> 1);
        GeoEntry e;
        if (hipMalloc((void **)&e.dev, buf.size()) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemcpy(e.dev, buf.data(), buf.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(e.dev);
            return nullptr;
        }
        e.bytes = buf.size();
        it = cache->map.emplace(key, e).first;
    }
    it->second.stamp = ++cache->clock;
    return (const BcEntry *)it->second.dev;
}

hipError_t launch_bicubic_cols(OutKind out, bool exact, const LaunchDesc &d, const FrameTable &t, size_t lds_bytes, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block(MAX_THREADS);
    const bool sparse = d.bc_sparse != 0;
    if (info) {
        info->kernel = exact ? (sparse ? "vpp_bicubic_cols_kernel<OUT, exact, sparse>" : "vpp_bicubic_cols_kernel<OUT, exact, dense>")
                             : (sparse ? "vpp_bicubic_cols_kernel<OUT, tie, sparse>" : "vpp_bicubic_cols_kernel<OUT, tie, dense>");
        info->grid = (int)grid.x;
        info->lds_bytes = (int)lds_bytes + (out == O_U8_MERGED ? MAX_THREADS * 24 : 0); // + the static exchange slab of the uint8 merged output side
        return hipSuccess;
    }
    if (!d.bc_tab) return hipErrorInvalidValue;
    switch (out) {
#define TSVPP_BC(O)                                                                                                               \
    case O:                                                                                                                       \
        if (exact && sparse) TSVPP_LAUNCH((vpp_bicubic_cols_kernel<O, true, true>), grid, block, lds_bytes, stream, d, t);   \
        else if (exact) TSVPP_LAUNCH((vpp_bicubic_cols_kernel<O, true, false>), grid, block, lds_bytes, stream, d, t);       \
        else if (sparse) TSVPP_LAUNCH((vpp_bicubic_cols_kernel<O, false, true>), grid, block, lds_bytes, stream, d, t);      \
        else TSVPP_LAUNCH((vpp_bicubic_cols_kernel<O, false, false>), grid, block, lds_bytes, stream, d, t);                 \
        break;
        TSVPP_BC(O_U8_PLANAR) TSVPP_BC(O_U8_MERGED) TSVPP_BC(O_F32_PLANAR) TSVPP_BC(O_F32_MERGED) TSVPP_BC(O_NV12_U8)
        TSVPP_BC(O_NV12_F32) TSVPP_BC(O_Y800_U8) TSVPP_BC(O_Y800_F32) TSVPP_BC(O_HSV_F32)
#undef TSVPP_BC
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace tsvpp
