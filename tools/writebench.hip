// writebench.hip -- which output write pattern does HBM like?  64 frames of 1280x720x3 fp32 planar.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int W = 1280, H = 720, NF = 64;
constexpr size_t PLANE = (size_t)W * H, FRAME = 3 * PLANE;
typedef float vf4 __attribute__((ext_vector_type(4)));

// A: 128x16 tiles, thread = 2 rows x 4 px, 3 planes (the kernel's pattern). tiles: 10 x 45 per frame
template <int TXS, int TYS, bool NT>
__global__ void k_tile(float *out, int xcd) {
    constexpr int TW = TXS * 4, TH = TYS * 2;
    constexpr int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles = tiles_x * tiles_y;
    int total = tiles * NF, per = (total + 7) / 8;
    int logical = xcd ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    if (logical >= total) return;
    int f = logical / tiles, rem = logical % tiles, ty = rem / tiles_x, tx = rem % tiles_x;
    int lx = threadIdx.x % TXS, ly = threadIdx.x / TXS;
    int j0 = tx * TW + lx * 4, i0 = ty * TH + ly * 2;
    if (j0 >= W || i0 >= H) return;
    float *o = out + f * FRAME;
    vf4 v = { (float)j0, 1.f, 2.f, 3.f };
    for (int r = 0; r < 2; r++)
        for (int p = 0; p < 3; p++) {
            vf4 *q = (vf4 *)(o + p * PLANE + (size_t)(i0 + r) * W + j0);
            if (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
}
// B: one WG = ROWS full output rows x 3 planes; 320 threads, thread = 4 px of each row
template <int ROWS, bool NT>
__global__ void k_rows(float *out) {
    int groups = H / ROWS;
    int f = blockIdx.x / groups, g = blockIdx.x % groups;
    float *o = out + f * FRAME;
    vf4 v = { (float)threadIdx.x, 1.f, 2.f, 3.f };
    for (int r = 0; r < ROWS; r++)
        for (int p = 0; p < 3; p++) {
            vf4 *q = (vf4 *)(o + p * PLANE + (size_t)(g * ROWS + r) * W + threadIdx.x * 4);
            if (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
}
// D: single plane per WG (3x more WGs), 128x16 tiles
__global__ void k_tile_1plane(float *out) {
    constexpr int tiles_x = 10, tiles_y = 45, tiles = 450;
    int logical = blockIdx.x;
    int f = logical / (tiles * 3), rem = logical % (tiles * 3), p = rem / tiles, t = rem % tiles, ty = t / tiles_x, tx = t % tiles_x;
    int lx = threadIdx.x % 32, ly = threadIdx.x / 32;
    int j0 = tx * 128 + lx * 4, i0 = ty * 16 + ly * 2;
    float *o = out + f * FRAME + p * PLANE;
    vf4 v = { (float)j0, 1.f, 2.f, 3.f };
    for (int r = 0; r < 2; r++) *(vf4 *)(o + (size_t)(i0 + r) * W + j0) = v;
}
// E: linear slab: WG b writes 12 KiB contiguous (= the bytes of one tile) at b * 12 KiB
__global__ void k_linear(float *out) {
    vf4 *q = (vf4 *)out + (size_t)blockIdx.x * 1536 + threadIdx.x;
    vf4 v = { (float)threadIdx.x, 1.f, 2.f, 3.f };
    for (int k = 0; k < 6; k++) q[k * 256] = v;
}

int main() {
    float *buf[3];
    for (auto &b : buf) { CK(hipMalloc(&b, NF * FRAME * 4)); CK(hipMemset(b, 0, NF * FRAME * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)NF * FRAME * 4;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch(buf[i % 3]);
        hipEventRecord(e0);
        const int it = 12;
        for (int i = 0; i < it; i++) launch(buf[i % 3]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s  (%.1f us per 64 frames)\n", name, bytes * it / (ms * 1e-3) / 1e9, ms / it * 1e3);
    };
    const int nt = 450 * NF, ntp = ((nt + 7) / 8) * 8;
    run("A tile 128x16 (32x8 thr), raster", [&](float *b) { k_tile<32, 8, false><<<nt, 256>>>(b, 0); });
    run("A tile 128x16, xcd remap", [&](float *b) { k_tile<32, 8, false><<<ntp, 256>>>(b, 1); });
    run("A tile 128x16, nontemporal", [&](float *b) { k_tile<32, 8, true><<<nt, 256>>>(b, 0); });
    run("A tile 256x8 (64x4 thr)", [&](float *b) { k_tile<64, 4, false><<<5 * 90 * NF, 256>>>(b, 0); });
    run("A tile 256x8, nontemporal", [&](float *b) { k_tile<64, 4, true><<<5 * 90 * NF, 256>>>(b, 0); });
    run("B 2 full rows per WG (320 thr)", [&](float *b) { k_rows<2, false><<<NF * H / 2, 320>>>(b); });
    run("B 2 full rows, nontemporal", [&](float *b) { k_rows<2, true><<<NF * H / 2, 320>>>(b); });
    run("B 4 full rows per WG", [&](float *b) { k_rows<4, false><<<NF * H / 4, 320>>>(b); });
    run("B 8 full rows per WG", [&](float *b) { k_rows<8, false><<<NF * H / 8, 320>>>(b); });
    run("B 8 full rows, nontemporal", [&](float *b) { k_rows<8, true><<<NF * H / 8, 320>>>(b); });
    run("D tile 128x16, one plane per WG", [&](float *b) { k_tile_1plane<<<nt * 3, 256>>>(b); });
    run("E linear 24 KiB slab per WG", [&](float *b) { k_linear<<<nt, 256>>>(b); });
    return 0;
}
