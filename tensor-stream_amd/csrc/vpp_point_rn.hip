// vpp_point_rn.hip -- pure point samplers at an EXACT integer ratio N on both axes, as a streaming kernel without LDS staging: NEAREST (src/Resize.cu:249-265:
// source sample (int)(N j) = N j) and the BILINEAR / BICUBIC requests whose every weight is zero (odd N: (j + 0.5) N - 0.5 = N j + (N - 1) / 2 exactly, so the
// interpolation formulas of src/Resize.cu:5-25, 27-91 reduce to the centre tap -- BASELINE config C4: 3840x2160 -> 1280x720 BICUBIC -> BGR24 MERGED uint8).
//
// Until round 4 these ran on vpp_point_kernel (vpp_kernels.hip): one LDS row per output row, register-staged, a workgroup barrier, per-tile offset tables, thread
// tiles of 4 x 2 pixels -- C4's 64-frame launch executed 3.6e7 VALU instructions (39 per output pixel, the VALU busy 65 % of the launch) and moved 0.61 of
// the HBM roofline (profiles/r04_c4_pmc.txt).  At an exact integer ratio nothing needs computing: a thread converts 8 output columns x 4 output rows from the
// 8 N source bytes [8 N q, 8 N q + 8 N) of 4 luma rows (N i + OFF) and 2 chroma rows -- dword-aligned runs, loaded with dwordx4 / dwordx2 straight from global
// memory, contiguous across the wave -- picks its bytes at COMPILE-TIME positions (v_perm_b32 with constant selectors: luma byte N v + OFF, chroma pair at
// byte 2 (N c + OFF)) and hands the packed bytes to the output side of the 8 x 4 streaming tiles (vpp_r32_store.h: every flavour, whole-line stores).
// No staging, no barrier, no tables, no coordinate arithmetic.  Only the tapped ROWS are read (a third of them at N = 3); of a tapped row every line is.
#include "vpp_device.h"
#include "vpp_r32_store.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef uint32_t prn_x2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t prn_x4 __attribute__((ext_vector_type(4), aligned(4)));

// ND consecutive dwords from a dword-aligned address: dwordx4 while four remain, then dwordx2
template <int ND> __device__ __forceinline__ void prn_load(const uint8_t *p, uint32_t (&dw)[ND]) {
    static_assert(ND % 2 == 0, "runs are 8 N bytes");
#pragma unroll
    for (int k = 0; k + 4 <= ND; k += 4) {
        const prn_x4 v = *(const prn_x4 *)(p + 4 * k); // (plain: non-temporal loads measured -1..-14 %, profiles/r05_prn_nt_variants.txt)
        dw[k] = v.x; dw[k + 1] = v.y; dw[k + 2] = v.z; dw[k + 3] = v.w;
    }
    if constexpr (ND % 4 == 2) {
        const prn_x2 v = *(const prn_x2 *)(p + 4 * (ND - 2));
        dw[ND - 2] = v.x; dw[ND - 1] = v.y;
    }
}

// Bytes b[0..3] of the run (compile-time positions after unrolling) gathered into one dword, byte e <- run byte b[e]: the first two distinct source dwords
// merge in one v_perm_b32, every further one costs one more (N = 3: two per dword, N = 5: three).
template <int ND> __device__ __forceinline__ uint32_t prn_pick(const uint32_t (&dw)[ND], const int (&b)[4]) {
    uint32_t r = 0;
    bool have = false;       // r holds the bytes of the positions in `filled`
    bool filled[4] = { false, false, false, false };
#pragma unroll
    for (int e0 = 0; e0 < 4; e0++) {
        if (filled[e0]) continue;
        const int da = b[e0] >> 2; // the next source dword that is still needed
        if (!have) {
            // is there a second distinct dword?  merge both in one instruction
            int db = -1;
#pragma unroll
            for (int e = e0 + 1; e < 4; e++)
                if (db < 0 && (b[e] >> 2) != da) db = b[e] >> 2;
            uint32_t sel = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t s = 0x0cu; // constant 0x00
                if ((b[e] >> 2) == da) { s = (uint32_t)(b[e] & 3); filled[e] = true; }           // bytes 0..3: the second operand
                else if (db >= 0 && (b[e] >> 2) == db) { s = 4u + (uint32_t)(b[e] & 3); filled[e] = true; } // bytes 4..7: the first operand
                sel |= s << (8 * e);
            }
            r = __builtin_amdgcn_perm(db >= 0 ? dw[db] : 0u, dw[da], sel);
            have = true;
        } else {
            uint32_t sel = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t s = (uint32_t)e; // keep what r holds
                if (!filled[e] && (b[e] >> 2) == da) { s = 4u + (uint32_t)(b[e] & 3); filled[e] = true; }
                sel |= s << (8 * e);
            }
            r = __builtin_amdgcn_perm(dw[da], r, sel);
        }
    }
    return r;
}

constexpr int PRN_COLS = 8, PRN_ROWS = 4;

template <int OUT, int N, int OFF>
__global__ __launch_bounds__(MAX_THREADS) void vpp_point_rn_kernel(const LaunchDesc d, const FrameTable t) {
    constexpr int RUN = 8 * N, ND = 2 * N; // source bytes / dwords of a thread's run
    const TileId id = decode_tile(d);      // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = PRN_COLS * q, i0 = PRN_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    uint8_t *out = (uint8_t *)t.out[id.frame];

    // all loads of the thread tile first: luma rows N (i0 + r) + OFF, chroma rows N (i0 / 2 + rc) + OFF, RUN bytes each
    uint32_t ys[PRN_ROWS][ND], cs[2][ND];
    const uint8_t *py = t.y[id.frame] + (size_t)(N * i0 + OFF) * (size_t)d.pitch_y + (size_t)(RUN * q);
#pragma unroll
    for (int r = 0; r < PRN_ROWS; r++) prn_load<ND>(py + (size_t)(N * r) * (size_t)d.pitch_y, ys[r]);
    if constexpr (!kLumaOnly<OUT>) {
        const uint8_t *pc = t.uv[id.frame] + (size_t)(N * (i0 >> 1) + OFF) * (size_t)d.pitch_uv + (size_t)(RUN * q);
#pragma unroll
        for (int rc = 0; rc < 2; rc++) prn_load<ND>(pc + (size_t)(N * rc) * (size_t)d.pitch_uv, cs[rc]);
    }

    uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
#pragma unroll
    for (int r = 0; r < PRN_ROWS; r++) {
        const int b0[4] = { OFF, N + OFF, 2 * N + OFF, 3 * N + OFF }, b1[4] = { 4 * N + OFF, 5 * N + OFF, 6 * N + OFF, 7 * N + OFF };
        ylo[r] = prn_pick<ND>(ys[r], b0);
        yhi[r] = prn_pick<ND>(ys[r], b1);
    }
    if constexpr (!kLumaOnly<OUT>) {
#pragma unroll
        for (int rc = 0; rc < 2; rc++) { // pair c of the thread: (U, V) at bytes 2 (N c + OFF), + 1
            const int b0[4] = { 2 * OFF, 2 * OFF + 1, 2 * (N + OFF), 2 * (N + OFF) + 1 }, b1[4] = { 2 * (2 * N + OFF), 2 * (2 * N + OFF) + 1, 2 * (3 * N + OFF), 2 * (3 * N + OFF) + 1 };
            clo[rc] = prn_pick<ND>(cs[rc], b0);
            chi[rc] = prn_pick<ND>(cs[rc], b1);
        }
    }
    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - PRN_COLS * run_m)) / PRN_COLS);
    r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
}

// ---- the same idea at the exact ratio 1 : 2 (540p -> 1080p, 1080p -> 4K): NEAREST (source sample (int)(0.5 j) = j / 2) and the AREA up-scale, whose weights are ALL zero at
// this ratio -- src/Resize.cu:221-234: x = floor(0.5 j), xFloat = (j + 1) - (x + 1) / 0.5 is -1 for even j and 0 for odd j, both "<= 0 -> 0" -- so that its bilinear blend
// returns tap A itself: both are 2 x 2 pixel REPLICATION, on the chroma grid alike.  Seven eighths of such a launch's bytes are output and the LDS kernels' 4 x 2-pixel
// thread tiles spent the launch in their colour / pack / store phase (uint8 outputs 0.31-0.34 of the roofline, profiles/r04_upscale_u8_probe.txt).  Here a thread's
// 8 x 4 output pixels are ONE luma dword of two source rows and ONE chroma dword, each byte (pair) doubled by a v_perm_b32 with a constant selector.
template <int OUT>
__global__ __launch_bounds__(MAX_THREADS) void vpp_rep2_kernel(const LaunchDesc d, const FrameTable t) {
    const TileId id = decode_tile(d); // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = PRN_COLS * q, i0 = PRN_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    uint8_t *out = (uint8_t *)t.out[id.frame];
    // output rows 4 n .. 4 n + 3 <- source rows 2 n, 2 n + 1; columns 8 q .. 8 q + 7 <- source bytes 4 q .. 4 q + 3; chroma output rows 2 n, 2 n + 1 <- chroma row n,
    // pairs 4 q .. 4 q + 3 <- source pairs 2 q, 2 q + 1 = bytes 4 q .. 4 q + 3
    const uint8_t *py = t.y[id.frame] + (size_t)(2 * n4) * (size_t)d.pitch_y + (size_t)(4 * q);
    const uint32_t y0 = *(const uint32_t *)py, y1 = *(const uint32_t *)(py + d.pitch_y);
    uint32_t c = 0x80808080u;
    if constexpr (!kLumaOnly<OUT>) c = *(const uint32_t *)(t.uv[id.frame] + (size_t)n4 * (size_t)d.pitch_uv + (size_t)(4 * q));
    uint32_t ylo[4], yhi[4], clo[2], chi[2];
    ylo[0] = ylo[1] = __builtin_amdgcn_perm(0u, y0, 0x01010000u); // s0 s0 s1 s1
    yhi[0] = yhi[1] = __builtin_amdgcn_perm(0u, y0, 0x03030202u); // s2 s2 s3 s3
    ylo[2] = ylo[3] = __builtin_amdgcn_perm(0u, y1, 0x01010000u);
    yhi[2] = yhi[3] = __builtin_amdgcn_perm(0u, y1, 0x03030202u);
    clo[0] = clo[1] = __builtin_amdgcn_perm(0u, c, 0x01000100u);   // U0 V0 U0 V0
    chi[0] = chi[1] = __builtin_amdgcn_perm(0u, c, 0x03020302u);   // U1 V1 U1 V1
    const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
    const int run_a = min(run_len, (d.dst_w - (j0 - PRN_COLS * run_m)) / PRN_COLS);
    r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
}

template <int N, int OFF>
static hipError_t launch_prn(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, hipStream_t stream) {
    switch (out) {
#define TSVPP_PRN(O) case O: TSVPP_LAUNCH((vpp_point_rn_kernel<O, N, OFF>), grid, block, 0, stream, d, t); break;
        TSVPP_PRN(O_U8_PLANAR) TSVPP_PRN(O_U8_MERGED) TSVPP_PRN(O_NV12_U8) TSVPP_PRN(O_Y800_U8)
        TSVPP_PRN(O_F32_PLANAR) TSVPP_PRN(O_F32_MERGED) TSVPP_PRN(O_NV12_F32) TSVPP_PRN(O_Y800_F32) TSVPP_PRN(O_HSV_F32)
#undef TSVPP_PRN
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// d.r32 = 100 + 10 N + OFF, or 20: replication at 1 : 2 (launch_fused)
hipError_t launch_point_rn(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    const char *name = nullptr;
    if (d.r32 == 20) { // pixel replication at 1 : 2 (NEAREST, AREA up-scale)
        if (info) {
            info->kernel = "vpp_rep2_kernel<OUT>";
            info->grid = (int)grid.x;
            info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : 16;
            return hipSuccess;
        }
        switch (out) {
#define TSVPP_REP2(O) case O: TSVPP_LAUNCH((vpp_rep2_kernel<O>), grid, block, 0, stream, d, t); break;
            TSVPP_REP2(O_U8_PLANAR) TSVPP_REP2(O_U8_MERGED) TSVPP_REP2(O_NV12_U8) TSVPP_REP2(O_Y800_U8)
            TSVPP_REP2(O_F32_PLANAR) TSVPP_REP2(O_F32_MERGED) TSVPP_REP2(O_NV12_F32) TSVPP_REP2(O_Y800_F32) TSVPP_REP2(O_HSV_F32)
#undef TSVPP_REP2
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (d.r32) {
    case 130: name = "vpp_point_rn_kernel<OUT,3:1,nearest>"; break;
    case 131: name = "vpp_point_rn_kernel<OUT,3:1,centre>"; break;
    case 140: name = "vpp_point_rn_kernel<OUT,4:1,nearest>"; break;
    case 150: name = "vpp_point_rn_kernel<OUT,5:1,nearest>"; break;
    case 152: name = "vpp_point_rn_kernel<OUT,5:1,centre>"; break;
    default: return hipErrorInvalidValue;
    }
    if (info) {
        info->kernel = name;
        info->grid = (int)grid.x;
        info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : 16;
        return hipSuccess;
    }
    switch (d.r32) {
    case 130: return launch_prn<3, 0>(out, d, t, grid, block, stream);
    case 131: return launch_prn<3, 1>(out, d, t, grid, block, stream);
    case 140: return launch_prn<4, 0>(out, d, t, grid, block, stream);
    case 150: return launch_prn<5, 0>(out, d, t, grid, block, stream);
    default: return launch_prn<5, 2>(out, d, t, grid, block, stream);
    }
}

} // namespace tsvpp
