// membench2.hip -- settles the HBM floor the headline kernel is priced against: does a float4
// copy reach the guide's 6.29 TB/s on this box, and what does the headline's own 22 % read / 78 % write mix with its
// exact store pattern (three planes 3.69 MB apart, 512-byte row segments per tile) reach with no arithmetic at all?
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/bin/membench2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const vf4 *__restrict__ in, vf4 *__restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256 * U;
    for (; i + (size_t)(U - 1) * 256 < n; i += st) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&in[i + (size_t)u * 256]) : in[i + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(v[u], &o[i + (size_t)u * 256]);
            else o[i + (size_t)u * 256] = v[u];
        }
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_read(const vf4 *__restrict__ in, float *sink, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256 * U;
    float acc = 0.f;
    for (; i + (size_t)(U - 1) * 256 < n; i += st) {
#pragma unroll
        for (int u = 0; u < U; u++) { vf4 v = in[i + (size_t)u * 256]; acc += v.x + v.w; }
    }
    if (acc == 123.456f) *sink = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(vf4 *__restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256 * U;
    const vf4 v = { 1.f, 2.f, 3.f, (float)threadIdx.x };
    for (; i + (size_t)(U - 1) * 256 < n; i += st) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(v, &o[i + (size_t)u * 256]);
            else o[i + (size_t)u * 256] = v;
        }
    }
}
// The headline's traffic shape, no arithmetic: one workgroup = one 128 x ROWS output tile of one 1280 x 720 fp32 planar
// frame (3 planes), reading the 192 x (1.5 ROWS) luma + 192 x (0.75 ROWS) chroma bytes under it from a 2048-pitch 1080p NV12
// frame with 16-byte loads, writing float4 per lane (32 lanes = one 512-byte row segment) with non-temporal stores.
// XCD-aware order as in decode_tile (whole tile row per XCD).
template <int ROWS, int TW>
__global__ __launch_bounds__(256) void k_mix(const uint8_t *__restrict__ src, float *__restrict__ dst, int n_frames, int do_read, int do_write) {
    constexpr int W = 1280, H = 720, SP = 2048, SH = 1080;
    constexpr int tiles_x = W / TW, tiles_y = (H + ROWS - 1) / ROWS;
    const int x = blockIdx.x % 8, q = blockIdx.x / 8;
    const int group = q / tiles_x, tx = q - group * tiles_x;
    const int row = group * 8 + x;
    const int frame = row / tiles_y, ty = row - frame * tiles_y;
    if (frame >= n_frames) return;
    const uint8_t *y = src + (size_t)frame * (SP * SH * 3 / 2), *uv = y + SP * SH;
    float *o = dst + (size_t)frame * (3 * W * H);
    // reads: luma rows [1.5 ty ROWS, +1.5 ROWS + 1) x bytes [1.5 tx TW, + 1.5 TW + 16), 16-byte chunks
    constexpr int SW16 = (TW * 3 / 2 + 16 + 15) / 16; // chunks per row
    const int y0 = ty * ROWS * 3 / 2, x0 = (tx * TW * 3 / 2) & ~15;
    const int ny = min(ROWS * 3 / 2 + 1, SH - y0), nuv = min(ROWS * 3 / 4 + 1, SH / 2 - y0 / 2);
    uint32_t acc = 0;
    if (do_read) for (int e = threadIdx.x; e < (ny + nuv) * SW16; e += 256) {
        const int r = e / SW16, c = e - r * SW16;
        const uint8_t *p = r < ny ? y + (size_t)(y0 + r) * SP : uv + (size_t)(y0 / 2 + r - ny) * SP;
        const uint4 v = *(const uint4 *)(p + x0 + 16 * c);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    const float f = (float)(acc & 255);
    const vf4 v = { f, f + 1.f, f + 2.f, f + 3.f };
    constexpr int LPR = TW / 4; // lanes per tile row
    if (!do_write) { if (acc == 0x12345u) dst[0] = f; return; }
    for (int e = threadIdx.x; e < ROWS * LPR; e += 256) {
        const int r = e / LPR, lx = e - r * LPR;
        const int i = ty * ROWS + r;
        if (i >= H) break;
        float *p = o + (size_t)i * W + tx * TW + lx * 4;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) __builtin_nontemporal_store(v, (vf4 *)(p + (size_t)pl * W * H));
    }
}

int main() {
    const size_t bytes = (size_t)1 << 31; // 2 GiB buffers >> 256 MiB Infinity Cache
    vf4 *a, *b; float *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double moved, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        float best = 1e30f, tot = 0.f;
        const int it = 7;
        for (int i = 0; i < it; i++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best; tot += ms;
        }
        printf("%-44s avg %7.1f GB/s  best %7.1f GB/s (%.3f ms)\n", name, moved * it / (tot * 1e-3) / 1e9, moved / (best * 1e-3) / 1e9, best);
        fflush(stdout);
    };
    for (int g : {2048, 65536}) {
        char nm[96];
        snprintf(nm, sizeof nm, "copy float4 U=1 grid %d (r+w bytes)", g); run(nm, 2.0 * bytes, [&] { k_copy<1, false><<<g, 256>>>(a, b, n); });
        snprintf(nm, sizeof nm, "copy float4 U=4 grid %d", g); run(nm, 2.0 * bytes, [&] { k_copy<4, false><<<g, 256>>>(a, b, n); });
        snprintf(nm, sizeof nm, "copy float4 U=4 nontemporal grid %d", g); run(nm, 2.0 * bytes, [&] { k_copy<4, true><<<g, 256>>>(a, b, n); });
        snprintf(nm, sizeof nm, "copy float4 U=8 grid %d", g); run(nm, 2.0 * bytes, [&] { k_copy<8, false><<<g, 256>>>(a, b, n); });
    }
    run("copy one pass, 1 float4/thread", 2.0 * bytes, [&] { k_copy<1, false><<<(unsigned)(n / 256), 256>>>(a, b, n); });
    run("copy one pass, 4 float4/thread", 2.0 * bytes, [&] { k_copy<4, false><<<(unsigned)(n / 1024), 256>>>(a, b, n); });
    run("copy one pass, 4 float4/thread, nontemporal", 2.0 * bytes, [&] { k_copy<4, true><<<(unsigned)(n / 1024), 256>>>(a, b, n); });
    run("hipMemcpyDtoDAsync", 2.0 * bytes, [&] { (void)hipMemcpyDtoDAsync(b, a, bytes, 0); });
    run("read float4 U=4 grid 4096", (double)bytes, [&] { k_read<4><<<4096, 256>>>(a, sink, n); });
    run("read one pass, 4 float4/thread", (double)bytes, [&] { k_read<4><<<(unsigned)(n / 1024), 256>>>(a, sink, n); });
    run("write float4 U=4 grid 4096", (double)bytes, [&] { k_write<4, false><<<4096, 256>>>(a, n); });
    run("write one pass nontemporal, 4 float4/thread", (double)bytes, [&] { k_write<4, true><<<(unsigned)(n / 1024), 256>>>(a, n); });
    {   // the headline mix: 64 frames, 199 MB read + 708 MB written per launch.  Rotating sets as in bench.py (the 256 MiB
        // Infinity Cache must not serve the reads): 8 source sets of 64 frames in a (8 x 212 MB), 2 destination sets in b.
        const int F = 64;
        const double moved = (double)F * 14169600.0;
        int it = 0;
        auto src = [&] { return (const uint8_t *)a + (size_t)(it % 8) * ((size_t)F * 2048 * 1080 * 3 / 2); };
        auto dst = [&] { return (float *)b + (size_t)(it++ % 2) * ((size_t)F * 3 * 1280 * 720); };
        run("headline mix 128x32 tiles (no arithmetic)", moved, [&] { k_mix<32, 128><<<8 * ((23 * F + 7) / 8) * 10, 256>>>(src(), dst(), F, 1, 1); });
        run("  reads only (199 MB)", (double)F * 3110400.0, [&] { k_mix<32, 128><<<8 * ((23 * F + 7) / 8) * 10, 256>>>(src(), dst(), F, 1, 0); });
        run("  writes only (708 MB)", (double)F * 11059200.0, [&] { k_mix<32, 128><<<8 * ((23 * F + 7) / 8) * 10, 256>>>(src(), dst(), F, 0, 1); });
        run("headline mix 128x16 tiles", moved, [&] { k_mix<16, 128><<<8 * ((45 * F + 7) / 8) * 10, 256>>>(src(), dst(), F, 1, 1); });
        run("headline mix 128x48 tiles (720 = 15 x 48)", moved, [&] { k_mix<48, 128><<<8 * ((15 * F + 7) / 8) * 10, 256>>>(src(), dst(), F, 1, 1); });
        run("headline mix 256x8 tiles (1 KiB segments)", moved, [&] { k_mix<8, 256><<<8 * ((90 * F + 7) / 8) * 5, 256>>>(src(), dst(), F, 1, 1); });
        run("headline mix 256x16 tiles", moved, [&] { k_mix<16, 256><<<8 * ((45 * F + 7) / 8) * 5, 256>>>(src(), dst(), F, 1, 1); });
        run("headline mix 256x24 tiles (720 = 30 x 24)", moved, [&] { k_mix<24, 256><<<8 * ((30 * F + 7) / 8) * 5, 256>>>(src(), dst(), F, 1, 1); });
        run("headline mix 1280x8 tiles (whole rows)", moved, [&] { k_mix<8, 1280><<<8 * ((90 * F + 7) / 8) * 1, 256>>>(src(), dst(), F, 1, 1); });
    }
    return 0;
}
