// vpp_bilinear_r32.hip -- BILINEAR at the exact ratio 3 : 2 on both axes (1920x1080 -> 1280x720, 3840x2160 -> 2560x1440, 1280x720 ->
// 854x480 is NOT one: 853.33) with uint8 outputs, as a streaming kernel without LDS.
//
// At ratio 1.5 the reference's coordinate (j + 0.5) * 1.5 - 0.5 (src/Resize.cu:276-303, one fma in its binary) is exactly 1.5 j + 0.25:
// output column 2k taps source columns (3k, 3k + 1) with weights (3/4, 1/4), column 2k + 1 taps (3k + 1, 3k + 2) with (1/4, 3/4); no
// clamp ever fires (the last tap is the last source column).  Rows alike.  Every weight is a multiple of 1/4, so the reference's float
// blend is exact and equals ( sum of tap * (4 wx) (4 wy) ) >> 4 -- here with sixteenths like the integer thread tile of
// vpp_bilinear.hip: ( sum tap * (16 wx)(16 wy) ) >> 8, products 144 / 48 / 48 / 16.  Those fit a byte: the WHOLE 2x2 blend of an output
// value is v_dot4_u32_u8 on the source dwords with compile-time byte weights, accumulated across the two source rows through the
// instruction's accumulator -- 2.5 instructions per value instead of 7 (two v_perm, two v_dot4, shift-or, v_dot2, convert).
//
// A thread converts 8 output columns x 4 output rows: source bytes [12 q, 12 q + 12) of 6 luma rows -- three aligned dwords per row,
// dwordx3 loads straight from global memory, contiguous across the wave -- and the same 12 bytes of 3 chroma rows (4 output pairs = 6
// source pairs; the two chroma output rows of the tile are the even / odd case of the same pattern).  No staging, no barrier, no
// coordinate arithmetic: the uint8 1080p -> 720p launch of the LDS kernel is bound by exactly those.
// The same tap positions serve the AREA down-scale at 3 : 2 (weight rows {1, 1/2}, {1/2, 1}: src/Resize.cu:160-212 with the pattern of
// generateResizePattern(1.5); integer weights (2, 1) / (1, 2), value = SUM / 9 truncated -- area_quot, as in the dyadic AREA kernels) and
// NEAREST (src/Resize.cu:249-265: the first tap alone), so KIND selects the weights and the final step; for NEAREST the untouched
// third source row / column costs nothing (its loads are dead code).
// RATIO 4 (template parameter P2 = twice the ratio) is the 2 : 1 case (3840x2160 -> 1920x1080, 1920x1080 -> 960x540) on the same skeleton:
// 1.5 j + 0.25 becomes 2 j + 0.5 -- taps (2 j, 2 j + 1) with weights (1/2, 1/2) for every index --, a thread's 8 columns are 16 source bytes
// (dwordx4 loads) and its 4 rows 8 luma / 4 chroma source rows; AREA sums the 2 x 2 box and divides by 4, NEAREST takes tap 2 j.
// UYVY and YUV444 (uint8) are two more outputs of the same pass: see the kernel body (the vertical chroma filter needs the tile below's two chroma
// rows, YUV444's horizontal one the pairs around the thread's four).
// Outputs: RGB24 / BGR24 uint8 planar (8-byte stores) and merged (24 bytes per lane and row, exchanged through LDS inside the wave so
// that every store instruction writes a contiguous run), NV12, Y800.  fp32 outputs stay on vpp_bilinear_kernel: it sits on the
// HBM floor of its write pattern already, and 8 fp32 columns per lane would split every line between two store instructions.
#include "vpp_device.h"
#include "vpp_r32_store.h"

#pragma clang fp contract(off)

namespace tsvpp {

typedef uint32_t r32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t r32x2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t r32x4 __attribute__((ext_vector_type(4), aligned(4)));

// byte weights of dword d for the two taps of one output value: taps at bytes b0 and b0 + step of the 12-byte run, weights wa, wb
constexpr uint32_t r32_w(int b0, int step, int wa, int wb, int d) {
    uint32_t m = 0;
    if (b0 / 4 == d) m |= (uint32_t)wa << (8 * (b0 % 4));
    if ((b0 + step) / 4 == d) m |= (uint32_t)wb << (8 * ((b0 + step) % 4));
    return m;
}
enum R32Kind : int { R32_BILINEAR = 0, R32_AREA = 1, R32_NEAREST = 2 };
// integer weights of the (first, second) tap of an output index along one axis: BILINEAR sixteenths, AREA halves (3 : 2) or ones (2 : 1),
// NEAREST the tap itself.  At 2 : 1 (P2 == 4) every index has the same pair.
template <int KIND, int P2> constexpr int r32_wfirst(bool odd) {
    return P2 == 4 ? (KIND == R32_BILINEAR ? 8 : 1) : KIND == R32_BILINEAR ? (odd ? 4 : 12) : KIND == R32_AREA ? (odd ? 1 : 2) : 1;
}
template <int KIND, int P2> constexpr int r32_wsecond(bool odd) {
    return P2 == 4 ? (KIND == R32_BILINEAR ? 8 : KIND == R32_AREA ? 1 : 0) : KIND == R32_BILINEAR ? (odd ? 12 : 4) : KIND == R32_AREA ? (odd ? 2 : 1) : 0;
}
// first tap of output index c within the thread's run, in samples: 3 : 2 -> 3 (c / 2) + (c & 1); 2 : 1 -> 2 c
template <int P2> constexpr int r32_first(int c) { return P2 == 4 ? 2 * c : 3 * (c >> 1) + (c & 1); }

// One output row of 8 values (luma: STEP 1) or 4 (U, V) pairs (chroma: STEP 2, values U0 V0 U1 V1 ...) from its two source rows of P2
// dwords.  ODD: the output row index is odd; columns alike.
template <int KIND, int P2, bool CHROMA, bool ODD>
__device__ __forceinline__ void r32_row(const uint32_t (&top)[P2], const uint32_t (&bot)[P2], float (&out)[8]) {
    constexpr int wy0 = r32_wfirst<KIND, P2>(ODD), wy1 = r32_wsecond<KIND, P2>(ODD);
#pragma unroll
    for (int v = 0; v < 8; v++) {
        // luma: value v = column v.  chroma: value v = component (v & 1) of pair column c = v / 2: byte 2 * pair + component, second tap
        // two bytes on
        const int c = CHROMA ? (v >> 1) : v;
        const int first = r32_first<P2>(c);
        const int b0 = CHROMA ? 2 * first + (v & 1) : first;
        const int step = CHROMA ? 2 : 1;
        const int wa = r32_wfirst<KIND, P2>((c & 1) != 0), wb = r32_wsecond<KIND, P2>((c & 1) != 0);
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < P2; d++) {
            const uint32_t mt = r32_w(b0, step, wa * wy0, wb * wy0, d), mb = r32_w(b0, step, wa * wy1, wb * wy1, d);
            if (mt != 0u) acc = __builtin_amdgcn_udot4(top[d], mt, acc, false);
            if (mb != 0u) acc = __builtin_amdgcn_udot4(bot[d], mb, acc, false);
        }
        if constexpr (KIND == R32_BILINEAR) out[v] = (float)((acc >> 8) & 255u); // v_cvt_f32_ubyte1
        else if constexpr (KIND == R32_AREA) out[v] = P2 == 4 ? area_quot(acc, 2, 2, 0.25f) : area_quot(acc, 3, 3, 1.0f / 9.0f);
        else out[v] = (float)(acc & 255u);
    }
}
// YUV444 needs the 4:2:2 chroma of the pair BEFORE the thread's four and of the two pairs AFTER them (its odd pixels filter across pairs): the
// same evaluation on an extended run -- one dword before the thread's P2, two after -- for pairs -1 .. 5 (14 values U-1 V-1 U0 V0 ... U5 V5).
template <int KIND, int P2, bool ODD>
__device__ __forceinline__ void r32_row_cx(const uint32_t (&top)[P2 + 3], const uint32_t (&bot)[P2 + 3], float (&out)[14]) {
    constexpr int wy0 = r32_wfirst<KIND, P2>(ODD), wy1 = r32_wsecond<KIND, P2>(ODD);
#pragma unroll
    for (int v = 0; v < 14; v++) {
        const int c = (v >> 1) - 1; // pair column relative to the thread's first
        const int first = P2 == 4 ? 2 * c : 3 * ((c + 2) / 2 - 1) + (c & 1); // r32_first for c >= -1 (no shift of a negative value)
        const int b0 = 2 * first + (v & 1) + 4;                               // byte in the extended run
        const int wa = r32_wfirst<KIND, P2>((c & 1) != 0), wb = r32_wsecond<KIND, P2>((c & 1) != 0);
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < P2 + 3; d++) {
            const uint32_t mt = r32_w(b0, 2, wa * wy0, wb * wy0, d), mb = r32_w(b0, 2, wa * wy1, wb * wy1, d);
            if (mt != 0u) acc = __builtin_amdgcn_udot4(top[d], mt, acc, false);
            if (mb != 0u) acc = __builtin_amdgcn_udot4(bot[d], mb, acc, false);
        }
        if constexpr (KIND == R32_BILINEAR) out[v] = (float)((acc >> 8) & 255u);
        else if constexpr (KIND == R32_AREA) out[v] = P2 == 4 ? area_quot(acc, 2, 2, 0.25f) : area_quot(acc, 3, 3, 1.0f / 9.0f);
        else out[v] = (float)(acc & 255u);
    }
}
// the vertical 4:2:0 -> 4:2:2 filter of an odd chroma row (src/ColorConversion.cu:107-127): exact in fp32, >> 4 of a possibly negative sum = floor
__device__ __forceinline__ float r32_vfilt(float cm1, float c, float cp1, float cp2) {
    const float f = __builtin_floorf((9.0f * (c + cp1) - (cm1 + cp2) + 8.0f) * 0.0625f);
    return __builtin_fminf(__builtin_fmaxf(f, 0.0f), 255.0f);
}
// Row ends of YUV444 (one thread per row end): the resized (U, V) of chroma row cr, pair cp straight from the source plane by the same integer
// weights -- the pair's two taps are four consecutive bytes U V U V of a source row: two aligned dwords, v_alignbyte, one v_dot4 per component and
// source row -- and the 4:2:2 chroma of a chroma row there.  Rows outside the image read as 0 (the reference's flat indexing runs off its buffer
// there, src/ColorConversion.cu:131-138).  The plane base and the pitch are multiples of 4 (a condition of this kernel).
// In two halves (round 6): r32_pair_taps requests the pair's dwords -- no branch, every address valid (cr clamped into the plane; the second dword of a row is asked for
// only where the pair straddles it) -- and r32_pair_eval turns them into (U, V); a caller puts the taps of ALL its pairs in flight before it evaluates the first.
// Written as one function per pair behind `if (cr in range)` / `if (cr odd)` the up to 96 loads of a row-end thread went out one round trip at a time.
struct R32PairTaps {
    uint32_t t0, t1, b0, b1;
};
template <int P2> __device__ __forceinline__ const uint8_t *r32_pair_ptr(const uint8_t *uv, int pitch, int cr, int cp) {
    const int fr = P2 == 4 ? 2 * cr : 3 * (cr >> 1) + (cr & 1), fc = P2 == 4 ? 2 * cp : 3 * (cp >> 1) + (cp & 1);
    return uv + (size_t)fr * (size_t)pitch + (size_t)(2 * fc);
}
template <int KIND, int P2> __device__ __forceinline__ R32PairTaps r32_pair_taps(const uint8_t *uv, int pitch, int cr, int cp) {
    const uint8_t *p = r32_pair_ptr<P2>(uv, pitch, cr, cp);
    const uint32_t sh = (uint32_t)(uintptr_t)p & 3u; // 0 or 2
    const uint32_t *a = (const uint32_t *)(p - sh);
    const int second = sh ? 1 : 0;
    R32PairTaps t;
    t.t0 = a[0];
    t.t1 = a[second];
    t.b0 = t.b1 = 0u;
    if (KIND != R32_NEAREST) {
        const uint32_t *b = (const uint32_t *)(p - sh + pitch);
        t.b0 = b[0];
        t.b1 = b[second];
    }
    return t;
}
template <int KIND, int P2> __device__ __forceinline__ void r32_pair_eval(const R32PairTaps &t, const uint8_t *uv, int pitch, int cr, int cp, float &u, float &v) {
    const bool orow = (cr & 1) != 0, ocol = (cp & 1) != 0;
    constexpr uint32_t xe = (uint32_t)r32_wfirst<KIND, P2>(false) | ((uint32_t)r32_wsecond<KIND, P2>(false) << 16);
    constexpr uint32_t xo = (uint32_t)r32_wfirst<KIND, P2>(true) | ((uint32_t)r32_wsecond<KIND, P2>(true) << 16);
    const uint32_t wx = ocol ? xo : xe; // byte weights of U0 . U1 . (V: one byte up)
    const uint32_t wy0 = (uint32_t)(orow ? r32_wfirst<KIND, P2>(true) : r32_wfirst<KIND, P2>(false));
    const uint32_t wy1 = (uint32_t)(orow ? r32_wsecond<KIND, P2>(true) : r32_wsecond<KIND, P2>(false));
    const uint32_t sh = (uint32_t)(uintptr_t)r32_pair_ptr<P2>(uv, pitch, cr, cp) & 3u;
    const uint32_t top = __builtin_amdgcn_alignbyte(t.t1, t.t0, sh), bot = __builtin_amdgcn_alignbyte(t.b1, t.b0, sh); // (shift 0: the first dword)
    uint32_t au = __builtin_amdgcn_udot4(top, wx * wy0, 0u, false), av = __builtin_amdgcn_udot4(top, (wx * wy0) << 8, 0u, false);
    if (KIND != R32_NEAREST) {
        au = __builtin_amdgcn_udot4(bot, wx * wy1, au, false);
        av = __builtin_amdgcn_udot4(bot, (wx * wy1) << 8, av, false);
    }
    if constexpr (KIND == R32_BILINEAR) {
        u = (float)((au >> 8) & 255u);
        v = (float)((av >> 8) & 255u);
    } else if constexpr (KIND == R32_AREA) {
        u = P2 == 4 ? area_quot(au, 2, 2, 0.25f) : area_quot(au, 3, 3, 1.0f / 9.0f);
        v = P2 == 4 ? area_quot(av, 2, 2, 0.25f) : area_quot(av, 3, 3, 1.0f / 9.0f);
    } else {
        u = (float)(au & 255u);
        v = (float)(av & 255u);
    }
}
// The resized (U, V) of pair cp in the N consecutive chroma rows first .. first + N - 1, each clamped into the image: all taps first, then the arithmetic.
template <int KIND, int P2, int N>
__device__ __forceinline__ void r32_pair_column(const uint8_t *uv, int pitch, int rows, int first, int cp, float (&u)[N], float (&v)[N]) {
    R32PairTaps t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = r32_pair_taps<KIND, P2>(uv, pitch, min(max(first + k, 0), rows - 1), cp);
#pragma unroll
    for (int k = 0; k < N; k++) r32_pair_eval<KIND, P2>(t[k], uv, pitch, min(max(first + k, 0), rows - 1), cp, u[k], v[k]);
}
template <int P2> __device__ __forceinline__ void r32_load(const uint8_t *p, uint32_t (&dw)[P2]) {
    if constexpr (P2 == 3) {
        const r32x3 v = *(const r32x3 *)p;
        dw[0] = v.x; dw[1] = v.y; dw[2] = v.z;
    } else {
        const r32x4 v = *(const r32x4 *)p;
        dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
    }
}
__device__ __forceinline__ void st8(uint8_t *base, uint32_t off, uint32_t lo, uint32_t hi, int nt) {
    const r32x2 v = { lo, hi };
    if (nt) st8_nt(base, off, lo, hi, nt); // (inline asm: see st8_nt, vpp_device.h -- the builtin's hint did not survive)
    else *(r32x2 *)(base + off) = v;
}

constexpr int R32_COLS = 8, R32_ROWS = 4;

template <int OUT, int KIND, int P2>
__global__ __launch_bounds__(MAX_THREADS) void vpp_bilinear_r32_kernel(const LaunchDesc d, const FrameTable t) {
    constexpr int NYR = 2 * P2, NCR = P2, RUN = 4 * P2; // luma / chroma source rows of a thread tile, source bytes per row
    const TileId id = decode_tile(d); // tiles of (8 tx) x (4 ty) output pixels
    if (!id.valid) return;
    const int lx = threadIdx.x & (d.tx - 1), ly = threadIdx.x >> d.tx_shift;
    const int q = id.tx * d.tx + lx, n4 = id.ty * d.ty + ly;
    const int j0 = R32_COLS * q, i0 = R32_ROWS * n4;
    if (j0 >= d.dst_w || i0 >= d.dst_h) return;
    const int nt = d.nt_stores;
    uint8_t *out = (uint8_t *)t.out[id.frame];
    const uint32_t plane = (uint32_t)d.dst_w * (uint32_t)d.dst_h;

    // all loads of the thread tile first (3 : 2: 6 luma rows, 3 chroma rows, 12 bytes each; 2 : 1: 8 and 4 rows of 16 bytes)
    uint32_t ys[NYR][P2], cs[NCR][P2];
    const uint8_t *py = t.y[id.frame] + (size_t)(NYR * n4) * (size_t)d.pitch_y + (size_t)(RUN * q);
#pragma unroll
    for (int r = 0; r < NYR; r++) r32_load<P2>(py + (size_t)r * (size_t)d.pitch_y, ys[r]);
    if constexpr (!kLumaOnly<OUT>) {
        const uint8_t *pc = t.uv[id.frame] + (size_t)(NCR * n4) * (size_t)d.pitch_uv + (size_t)(RUN * q);
#pragma unroll
        for (int r = 0; r < NCR; r++) r32_load<P2>(pc + (size_t)r * (size_t)d.pitch_uv, cs[r]);
    }

    if constexpr (OUT == O_YUV444_U8) {
        // YUV444 planar (uint8) of the resized frame in the same pass (reference src/ColorConversion.cu:129-173 on the 4:2:2 image above): Y as it is;
        // U / V of an even pixel = its pair's 4:2:2 chroma, of an odd pixel (9 (p1 + p2) - (p3 + p4) + 8) / 16 (C division, wrapped to a byte) over
        // the pair, the next, the previous and the next but one IN FLAT ORDER: before a row's first pair comes the previous row's last, after its
        // last the next row's first two; outside the image 0; the frame's first odd pixel takes p3 = p1, its last two p4 = p2.
        // The thread evaluates the 4:2:2 chroma of pairs -1 .. 5 around its four from runs extended by one dword before and two after (its neighbours'
        // loads: cache hits); the thread at a row's start / end takes the wrapped pairs from the source plane directly (r32_uyvy_px).
        const bool below = i0 + R32_ROWS < d.dst_h;
        const bool row_start = j0 == 0, row_end = j0 + R32_COLS >= d.dst_w;
        uint32_t ce[NCR][P2 + 3], cn[NCR][P2 + 3];
        {
            const uint8_t *pc = t.uv[id.frame] + (size_t)(RUN * q);
            const size_t r_this = (size_t)(NCR * n4), r_next = (size_t)(NCR * (n4 + (below ? 1 : 0)));
#pragma unroll
            for (int r = 0; r < NCR; r++) {
                const uint8_t *a = pc + (r_this + r) * (size_t)d.pitch_uv, *b = pc + (r_next + r) * (size_t)d.pitch_uv;
                // (round 6) the dwords before / after the run are loaded UNCONDITIONALLY -- from the run's own first dword in the threads at a row's start / end, whose
                // neighbours do not exist -- and zeroed by a select: written as `row_start ? 0 : load` these 36 loads sat in exec-masked branches and the compiler waited
                // for each before it issued the next (`s_waitcnt vmcnt(0)` in front of 33 of the kernel's 94 loads)
                const int before = row_start ? 0 : -4, after = row_end ? 0 : RUN;
                const uint32_t eb = *(const uint32_t *)(a + before), nb = *(const uint32_t *)(b + before);
                ce[r][0] = row_start ? 0u : eb;
                cn[r][0] = row_start ? 0u : nb;
#pragma unroll
                for (int k = 0; k < P2; k++) {
                    ce[r][1 + k] = cs[r][k];
                    cn[r][1 + k] = *(const uint32_t *)(b + 4 * k);
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t ea = *(const uint32_t *)(a + after + (row_end ? 0 : 4 * k)), na = *(const uint32_t *)(b + after + (row_end ? 0 : 4 * k));
                    ce[r][1 + P2 + k] = row_end ? 0u : ea;
                    cn[r][1 + P2 + k] = row_end ? 0u : na;
                }
            }
        }
        // the wrapped pairs (their loads are issued here, behind the tile's own: one wait covers both): luma row i0 + r takes pair -1 from row
        // i0 + r - 1 (chroma rows 2 n4 - 1, 2 n4, 2 n4, 2 n4 + 1) and pairs 4, 5 from row i0 + r + 1 (chroma rows 2 n4, 2 n4 + 1, 2 n4 + 1, 2 n4 + 2)
        const int crows = d.dst_h >> 1, cpairs = d.dst_w >> 1, cr0 = 2 * n4;
        float ws[3][2] = {}, we[3][4] = {};
        // 4:2:2 chroma of chroma row cr = the resized pair when cr is even, the vertical filter over the resized pairs of rows cr - 1, cr, min(cr + 1, last),
        // min(cr + 2, last) when it is odd, 0 outside the image.  cr0 is even and crows is even (dst_h % 4 == 0), so of the rows this tile needs only cr0 - 1 (above the
        // first tile) and cr0 + 2 (below the last) can fall outside.
        if (row_start) { // rows cr0 - 1 .. cr0 + 1 of the row's LAST pair: resized rows cr0 - 2 .. cr0 + 3
            float pu[6], pv[6];
            r32_pair_column<KIND, P2, 6>(t.uv[id.frame], d.pitch_uv, crows, cr0 - 2, cpairs - 1, pu, pv);
            const bool above = cr0 > 0;
            ws[0][0] = above ? r32_vfilt(pu[0], pu[1], pu[2], pu[3]) : 0.0f;
            ws[0][1] = above ? r32_vfilt(pv[0], pv[1], pv[2], pv[3]) : 0.0f;
            ws[1][0] = pu[2];
            ws[1][1] = pv[2];
            ws[2][0] = r32_vfilt(pu[2], pu[3], pu[4], pu[5]);
            ws[2][1] = r32_vfilt(pv[2], pv[3], pv[4], pv[5]);
        }
        if (row_end) { // rows cr0 .. cr0 + 2 of the row's FIRST two pairs: resized rows cr0 .. cr0 + 3
            const bool under = cr0 + 2 < crows;
#pragma unroll
            for (int cp = 0; cp < 2; cp++) {
                float pu[4], pv[4];
                r32_pair_column<KIND, P2, 4>(t.uv[id.frame], d.pitch_uv, crows, cr0, cp, pu, pv);
                we[0][2 * cp] = pu[0];
                we[0][2 * cp + 1] = pv[0];
                we[1][2 * cp] = r32_vfilt(pu[0], pu[1], pu[2], pu[3]);
                we[1][2 * cp + 1] = r32_vfilt(pv[0], pv[1], pv[2], pv[3]);
                we[2][2 * cp] = under ? pu[2] : 0.0f;
                we[2][2 * cp + 1] = under ? pv[2] : 0.0f;
            }
        }
        float c0[14], c1[14], c2[14], c3[14], cf[14];
        r32_row_cx<KIND, P2, false>(ce[0], ce[1], c0);
        r32_row_cx<KIND, P2, true>(ce[r32_first<P2>(1)], ce[r32_first<P2>(1) + 1], c1);
        r32_row_cx<KIND, P2, false>(cn[0], cn[1], c2);
        r32_row_cx<KIND, P2, true>(cn[r32_first<P2>(1)], cn[r32_first<P2>(1) + 1], c3);
#pragma unroll
        for (int v = 0; v < 14; v++) cf[v] = r32_vfilt(c0[v], c1[v], below ? c2[v] : c1[v], below ? c3[v] : c1[v]);
        const uint32_t wh = plane;
        const bool edge = row_start || row_end;
        uint32_t pu[2] = { 0, 0 }, pv[2] = { 0, 0 };
#pragma unroll
        for (int r = 0; r < R32_ROWS; r++) {
            float yf[8];
            if (r == 0) r32_row<KIND, P2, false, false>(ys[r32_first<P2>(0)], ys[r32_first<P2>(0) + 1], yf);
            else if (r == 1) r32_row<KIND, P2, false, true>(ys[r32_first<P2>(1)], ys[r32_first<P2>(1) + 1], yf);
            else if (r == 2) r32_row<KIND, P2, false, false>(ys[r32_first<P2>(2)], ys[r32_first<P2>(2) + 1], yf);
            else r32_row<KIND, P2, false, true>(ys[r32_first<P2>(3)], ys[r32_first<P2>(3) + 1], yf);
            const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
            // luma rows 2 rc, 2 rc + 1 share their chroma planes -- except in the threads at a row's ends, where the wrapped pairs (and the frame's
            // first / last odd pixels) differ from row to row
            if ((r & 1) == 0 || edge) {
                float ch[14];
#pragma unroll
                for (int v = 0; v < 14; v++) ch[v] = r < 2 ? c0[v] : cf[v];
                if (row_start) {
                    const int k = r == 0 ? 0 : (r == 3 ? 2 : 1);
                    ch[0] = ws[k][0];
                    ch[1] = ws[k][1];
                }
                if (row_end) {
                    const int k = r == 0 ? 0 : (r == 3 ? 2 : 1);
#pragma unroll
                    for (int v = 0; v < 4; v++) ch[10 + v] = we[k][v];
                }
                float uo[8], vo[8];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const uint32_t idx1 = pix + 2u * (uint32_t)p + 1u; // flat index of the pair's odd pixel
                    const bool first = idx1 == 1u, last2 = idx1 + 3u >= wh;
#pragma unroll
                    for (int comp = 0; comp < 2; comp++) {
                        const float p1 = ch[2 * (p + 1) + comp], p2 = ch[2 * (p + 2) + comp];
                        const float p3 = first ? p1 : ch[2 * p + comp], p4 = last2 ? p2 : ch[2 * (p + 3) + comp];
                        // (9 (p1 + p2) - (p3 + p4) + 8) / 16 in C: truncation towards zero, then the conversion to uchar wraps
                        const int quot = (int)__builtin_truncf((9.0f * (p1 + p2) - (p3 + p4) + 8.0f) * 0.0625f);
                        (comp ? vo : uo)[2 * p] = p1;
                        (comp ? vo : uo)[2 * p + 1] = (float)(quot & 255);
                    }
                }
                pu[0] = pack_u8x4(uo[0], uo[1], uo[2], uo[3]);
                pu[1] = pack_u8x4(uo[4], uo[5], uo[6], uo[7]);
                pv[0] = pack_u8x4(vo[0], vo[1], vo[2], vo[3]);
                pv[1] = pack_u8x4(vo[4], vo[5], vo[6], vo[7]);
            }
            st8(out, pix, pack_u8x4(yf[0], yf[1], yf[2], yf[3]), pack_u8x4(yf[4], yf[5], yf[6], yf[7]), nt);
            st8(out + wh, pix, pu[0], pu[1], nt);
            st8(out + 2 * (size_t)wh, pix, pv[0], pv[1], nt);
        }
        return;
    }
    if constexpr (OUT == O_UYVY_U8 || OUT == O_UYVY_F32) {
        // UYVY (4:2:2) of the resized frame in the same pass (reference src/ColorConversion.cu:107-127, 177-209 on the resized NV12): luma rows
        // 2 r, 2 r + 1 take chroma row r as it is when r is even and clamp((9 (c[r] + c[r+1]) - (c[r-1] + c[r+2]) + 8) >> 4) when r is odd (rows
        // clamped to the last).  The tile's chroma rows are r = 2 n4 (even) and 2 n4 + 1 (odd): the odd one needs the two chroma rows of the tile
        // BELOW -- NCR more source rows of this thread's run (they are the next thread row's loads: cache hits), evaluated here a second time.
        const bool below = i0 + R32_ROWS < d.dst_h; // (dst_h is a multiple of 4: the tile below is whole or absent)
        uint32_t cn[NCR][P2];
        {
            const uint8_t *pc = t.uv[id.frame] + (size_t)(NCR * (n4 + (below ? 1 : 0))) * (size_t)d.pitch_uv + (size_t)(RUN * q);
#pragma unroll
            for (int r = 0; r < NCR; r++) r32_load<P2>(pc + (size_t)r * (size_t)d.pitch_uv, cn[r]);
        }
        float c0[8], c1[8], c2[8], c3[8], cf[8];
        r32_row<KIND, P2, true, false>(cs[0], cs[1], c0);
        r32_row<KIND, P2, true, true>(cs[r32_first<P2>(1)], cs[r32_first<P2>(1) + 1], c1);
        r32_row<KIND, P2, true, false>(cn[0], cn[1], c2);
        r32_row<KIND, P2, true, true>(cn[r32_first<P2>(1)], cn[r32_first<P2>(1) + 1], c3);
#pragma unroll
        for (int v = 0; v < 8; v++) {
            // the last chroma row: r + 1 and r + 2 clamp onto r itself (cn then holds this tile's rows again: c3 == c1)
            const float in = c1[v] + (below ? c2[v] : c1[v]), outer = c0[v] + (below ? c3[v] : c1[v]);
            // exact in fp32 (integers below 2^13); >> 4 of a possibly negative sum = floor
            float f = __builtin_floorf((9.0f * in - outer + 8.0f) * 0.0625f);
            cf[v] = __builtin_fminf(__builtin_fmaxf(f, 0.0f), 255.0f);
        }
#pragma unroll
        for (int r = 0; r < R32_ROWS; r++) {
            float yf[8];
            if (r == 0) r32_row<KIND, P2, false, false>(ys[r32_first<P2>(0)], ys[r32_first<P2>(0) + 1], yf);
            else if (r == 1) r32_row<KIND, P2, false, true>(ys[r32_first<P2>(1)], ys[r32_first<P2>(1) + 1], yf);
            else if (r == 2) r32_row<KIND, P2, false, false>(ys[r32_first<P2>(2)], ys[r32_first<P2>(2) + 1], yf);
            else r32_row<KIND, P2, false, true>(ys[r32_first<P2>(3)], ys[r32_first<P2>(3) + 1], yf);
            const float *c = r < 2 ? c0 : cf; // U0 V0 U1 V1 U2 V2 U3 V3
            const r32x4 v = { pack_u8x4(c[0], yf[0], c[1], yf[1]), pack_u8x4(c[2], yf[2], c[3], yf[3]), pack_u8x4(c[4], yf[4], c[5], yf[5]),
                              pack_u8x4(c[6], yf[6], c[7], yf[7]) };
            const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
            if constexpr (OUT == O_UYVY_U8) {
                // 16 bytes per lane, 1 KiB contiguous per wave
                if (nt) st16_nt(out, 2u * pix, (nt_u32x4){ v.x, v.y, v.z, v.w }, nt); // (inline asm: see st8_nt, vpp_device.h)
                else *(r32x4 *)(out + 2u * (size_t)pix) = v;
            } else {
                // fp32 (round 6: was a second pass over the uint8 NV12 of this kernel): the lane's 16 values would leave as four 16-byte stores 64 bytes apart -- a quarter of
                // every line per instruction.  The lanes of a run (the lanes of the wave that share the output rows, A of them) trade their packed dwords through LDS instead:
                // the run's 4 A dwords = groups of four values, lane m converts groups m, A + m, 2 A + m, 3 A + m, and every store instruction writes 16 A contiguous bytes.
                __shared__ __attribute__((aligned(16))) uint8_t uslab[MAX_THREADS * 16];
                const int len = min(d.tx, 64), run_m = (int)threadIdx.x & (len - 1);
                const int run_a = min(len, (d.dst_w - (j0 - R32_COLS * run_m)) / R32_COLS);
                uint8_t *run_lds = uslab + ((int)threadIdx.x - run_m) * 16;
                *(r32x4 *)(run_lds + 16 * run_m) = v;
                __builtin_amdgcn_wave_barrier();
                const uint32_t row0 = 2u * (pix - (uint32_t)(R32_COLS * run_m)); // the run's first float of this row
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t g = (uint32_t)(k * run_a + run_m);
                    const uint32_t w = *(const uint32_t *)(run_lds + 4u * g);
                    const f2 a = norm255((f2){ (float)(w & 255u), (float)((w >> 8) & 255u) }), b = norm255((f2){ (float)((w >> 16) & 255u), (float)(w >> 24) });
                    st4o(out, (row0 + 4u * g) * 4u, a.x, a.y, b.x, b.y, nt);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }

    if constexpr (OUT == O_F32_PLANAR || OUT == O_F32_MERGED || OUT == O_NV12_F32 || OUT == O_Y800_F32 || OUT == O_HSV_F32) {
        // fp32 flavours (round 4): the tile's resized values as packed bytes, then the shared output side of the 8 x 4 streaming tiles (vpp_r32_store.h:
        // planar rows dealt out between the lanes by shuffles before the colour conversion, merged rows exchanged through LDS -- whole-line stores)
        uint32_t ylo[4], yhi[4], clo[2] = { 0x80808080u, 0x80808080u }, chi[2] = { 0x80808080u, 0x80808080u };
        {
            float yf[8];
            r32_row<KIND, P2, false, false>(ys[r32_first<P2>(0)], ys[r32_first<P2>(0) + 1], yf);
            ylo[0] = pack_u8x4(yf[0], yf[1], yf[2], yf[3]); yhi[0] = pack_u8x4(yf[4], yf[5], yf[6], yf[7]);
            r32_row<KIND, P2, false, true>(ys[r32_first<P2>(1)], ys[r32_first<P2>(1) + 1], yf);
            ylo[1] = pack_u8x4(yf[0], yf[1], yf[2], yf[3]); yhi[1] = pack_u8x4(yf[4], yf[5], yf[6], yf[7]);
            r32_row<KIND, P2, false, false>(ys[r32_first<P2>(2)], ys[r32_first<P2>(2) + 1], yf);
            ylo[2] = pack_u8x4(yf[0], yf[1], yf[2], yf[3]); yhi[2] = pack_u8x4(yf[4], yf[5], yf[6], yf[7]);
            r32_row<KIND, P2, false, true>(ys[r32_first<P2>(3)], ys[r32_first<P2>(3) + 1], yf);
            ylo[3] = pack_u8x4(yf[0], yf[1], yf[2], yf[3]); yhi[3] = pack_u8x4(yf[4], yf[5], yf[6], yf[7]);
        }
        if constexpr (!kLumaOnly<OUT>) {
            float uvf[8];
            r32_row<KIND, P2, true, false>(cs[0], cs[1], uvf);
            clo[0] = pack_u8x4(uvf[0], uvf[1], uvf[2], uvf[3]); chi[0] = pack_u8x4(uvf[4], uvf[5], uvf[6], uvf[7]);
            r32_row<KIND, P2, true, true>(cs[r32_first<P2>(1)], cs[r32_first<P2>(1) + 1], uvf);
            clo[1] = pack_u8x4(uvf[0], uvf[1], uvf[2], uvf[3]); chi[1] = pack_u8x4(uvf[4], uvf[5], uvf[6], uvf[7]);
        }
        const int run_len = min(d.tx, 64), run_m = (int)threadIdx.x & (run_len - 1);
        const int run_a = min(run_len, (d.dst_w - (j0 - R32_COLS * run_m)) / R32_COLS);
        r32_store_tile<OUT>(d, out, ylo, yhi, clo, chi, i0, j0, run_m, run_a);
        return;
    }

    // merged uint8: the lanes of a run (the lanes of a wave that share the output rows) exchange their 24-byte row pieces through LDS
    // so that the run's 24 A contiguous bytes leave as 16-byte stores (cf. MergedRun, vpp_device.h)
    __shared__ __attribute__((aligned(16))) uint8_t slab[OUT == O_U8_MERGED ? MAX_THREADS * 24 : 16];
    int run_m = 0, run_a = 1;
    uint8_t *run_lds = nullptr;
    if constexpr (OUT == O_U8_MERGED) {
        const int len = min(d.tx, 64);
        run_m = (int)threadIdx.x & (len - 1);
        run_a = min(len, (d.dst_w - (j0 - R32_COLS * run_m)) / R32_COLS);
        run_lds = slab + ((int)threadIdx.x - run_m) * 24;
    }

#pragma unroll
    for (int rc = 0; rc < 2; rc++) { // chroma output row rc of the tile = luma output rows 2 rc, 2 rc + 1
        float uvf[8] = { 128.0f, 128.0f, 128.0f, 128.0f, 128.0f, 128.0f, 128.0f, 128.0f }; // U0 V0 U1 V1 U2 V2 U3 V3
        if constexpr (!kLumaOnly<OUT>) {
            // chroma output row rc: first source row r32_first(rc) = 0 / 1 (3 : 2) or 0 / 2 (2 : 1)
            if (rc == 0) r32_row<KIND, P2, true, false>(cs[0], cs[1], uvf);
            else r32_row<KIND, P2, true, true>(cs[r32_first<P2>(1)], cs[r32_first<P2>(1) + 1], uvf);
        }
        float t0[4], tg[4], t2[4];
        if constexpr (OUT == O_U8_PLANAR || OUT == O_U8_MERGED) {
#pragma unroll
            for (int c = 0; c < 4; c++) chroma_terms(uvf[2 * c], uvf[2 * c + 1], d.k, d.swap_rb, d.color_g, t0[c], tg[c], t2[c]);
        }
        if constexpr (OUT == O_NV12_U8) {
            const uint32_t cpix = plane + (uint32_t)((i0 >> 1) + rc) * (uint32_t)d.dst_w + (uint32_t)j0;
            st8(out, cpix, pack_u8x4(uvf[0], uvf[1], uvf[2], uvf[3]), pack_u8x4(uvf[4], uvf[5], uvf[6], uvf[7]), nt);
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int r = 2 * rc + rr; // luma output row of the tile: source rows r32_first(r), r32_first(r) + 1
            float yf[8];
            if (rr == 0) r32_row<KIND, P2, false, false>(ys[r32_first<P2>(2 * rc)], ys[r32_first<P2>(2 * rc) + 1], yf);
            else r32_row<KIND, P2, false, true>(ys[r32_first<P2>(2 * rc + 1)], ys[r32_first<P2>(2 * rc + 1) + 1], yf);
            const uint32_t pix = (uint32_t)(i0 + r) * (uint32_t)d.dst_w + (uint32_t)j0;
            if constexpr (OUT == O_NV12_U8 || OUT == O_Y800_U8) {
                st8(out, pix, pack_u8x4(yf[0], yf[1], yf[2], yf[3]), pack_u8x4(yf[4], yf[5], yf[6], yf[7]), nt);
            } else {
                uint32_t pa[2], pb[2], pc[2];
#pragma unroll
                for (int h = 0; h < 2; h++)
                    color_pack_row_u8<OUT == O_U8_PLANAR>(yf + 4 * h, t0 + 2 * h, tg + 2 * h, t2 + 2 * h, d.k, pa[h], pb[h], pc[h]);
                if constexpr (OUT == O_U8_PLANAR) {
                    st8(out, pix, pa[0], pa[1], nt);
                    st8(out + plane, pix, pb[0], pb[1], nt);
                    st8(out + 2 * (size_t)plane, pix, pc[0], pc[1], nt);
                } else {
                    if ((run_a & 1) == 0) { // 24 A bytes = 3 A / 2 chunks of 16
                        uint32_t *w = (uint32_t *)(run_lds + 24 * run_m);
                        w[0] = pa[0]; w[1] = pb[0]; w[2] = pc[0]; w[3] = pa[1]; w[4] = pb[1]; w[5] = pc[1];
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t row0 = 3u * (pix - (uint32_t)(R32_COLS * run_m)); // first byte of the run in this row
                        const r32x4 v0 = *(const r32x4 *)(run_lds + 16 * run_m);
                        if (nt) st16_nt(out, row0 + 16u * (uint32_t)run_m, (nt_u32x4){ v0.x, v0.y, v0.z, v0.w }, nt); // (round 5: see vpp_r32_store.h)
                        else *(r32x4 *)(out + row0 + 16u * (uint32_t)run_m) = v0;
                        if (2 * run_m < run_a) {
                            const r32x4 v1 = *(const r32x4 *)(run_lds + 16 * (run_a + run_m));
                            if (nt) st16_nt(out, row0 + 16u * (uint32_t)(run_a + run_m), (nt_u32x4){ v1.x, v1.y, v1.z, v1.w }, nt);
                            else *(r32x4 *)(out + row0 + 16u * (uint32_t)(run_a + run_m)) = v1;
                        }
                        __builtin_amdgcn_wave_barrier();
                    } else {
                        st8(out, 3u * pix, pa[0], pb[0], 0);
                        st8(out, 3u * pix + 8u, pc[0], pa[1], 0);
                        st8(out, 3u * pix + 16u, pb[1], pc[1], 0);
                    }
                }
            }
        }
    }
}

template <int KIND, int P2>
static hipError_t launch_r32_k(OutKind out, const LaunchDesc &d, const FrameTable &t, dim3 grid, dim3 block, hipStream_t stream) {
    switch (out) {
#define TSVPP_R32(O) case O: TSVPP_LAUNCH((vpp_bilinear_r32_kernel<O, KIND, P2>), grid, block, 0, stream, d, t); break;
        TSVPP_R32(O_U8_PLANAR) TSVPP_R32(O_U8_MERGED) TSVPP_R32(O_NV12_U8) TSVPP_R32(O_Y800_U8) TSVPP_R32(O_UYVY_U8) TSVPP_R32(O_YUV444_U8) TSVPP_R32(O_UYVY_F32)
        TSVPP_R32(O_F32_PLANAR) TSVPP_R32(O_F32_MERGED) TSVPP_R32(O_NV12_F32) TSVPP_R32(O_Y800_F32) TSVPP_R32(O_HSV_F32)
#undef TSVPP_R32
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// d.r32: 1 BILINEAR, 2 AREA, 3 NEAREST at 3 : 2; 4 / 5 / 6 the same at 2 : 1 (launch_fused)
hipError_t launch_bilinear_r32(OutKind out, const LaunchDesc &d, const FrameTable &t, hipStream_t stream, LaunchInfo *info) {
    dim3 grid((unsigned)(d.blocks_per_xcd * NUM_XCD)), block((unsigned)(d.tx * d.ty));
    if (info) {
        static const char *const names[6] = { "vpp_bilinear_r32_kernel<OUT,bilinear,3:2>", "vpp_bilinear_r32_kernel<OUT,area,3:2>", "vpp_bilinear_r32_kernel<OUT,nearest,3:2>",
                                              "vpp_bilinear_r32_kernel<OUT,bilinear,2:1>", "vpp_bilinear_r32_kernel<OUT,area,2:1>", "vpp_bilinear_r32_kernel<OUT,nearest,2:1>" };
        if (d.r32 < 1 || d.r32 > 6) return hipErrorInvalidValue;
        info->kernel = names[d.r32 - 1];
        info->grid = (int)grid.x;
        info->lds_bytes = out == O_U8_MERGED ? MAX_THREADS * 24 : (out == O_F32_MERGED || out == O_HSV_F32) ? MAX_THREADS * 96 : out == O_UYVY_F32 ? MAX_THREADS * 16 : 16;
        return hipSuccess;
    }
    switch (d.r32) {
    case 1: return launch_r32_k<R32_BILINEAR, 3>(out, d, t, grid, block, stream);
    case 2: return launch_r32_k<R32_AREA, 3>(out, d, t, grid, block, stream);
    case 3: return launch_r32_k<R32_NEAREST, 3>(out, d, t, grid, block, stream);
    case 4: return launch_r32_k<R32_BILINEAR, 4>(out, d, t, grid, block, stream);
    case 5: return launch_r32_k<R32_AREA, 4>(out, d, t, grid, block, stream);
    case 6: return launch_r32_k<R32_NEAREST, 4>(out, d, t, grid, block, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace tsvpp
