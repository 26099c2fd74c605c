// storepolicy.hip -- does the cache policy of the output stores change the HBM write rate?  (MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int W = 1280, H = 720, NF = 64;
constexpr size_t PLANE = (size_t)W * H, FRAME = 3 * PLANE;
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int POL>
__device__ inline void st(vf4 *p, vf4 v) {
    if (POL == 0) *p = v;
    else if (POL == 1) __builtin_nontemporal_store(v, p);
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ void k_tile(float *out, const float *in) {
    constexpr int tiles_x = 10, tiles = 450;
    int logical = blockIdx.x;
    int f = logical / tiles, rem = logical % tiles, ty = rem / tiles_x, tx = rem % tiles_x;
    int lx = threadIdx.x % 32, ly = threadIdx.x / 32;
    int j0 = tx * 128 + lx * 4, i0 = ty * 16 + ly * 2;
    float *o = out + f * FRAME;
    // a little read traffic like the real kernel (22 %): 3.5 bytes per output pixel
    float r = in[(size_t)logical * 256 + threadIdx.x];
    vf4 v = { (float)j0, r, 2.f, 3.f };
    for (int rr = 0; rr < 2; rr++)
        for (int p = 0; p < 3; p++) st<POL>((vf4 *)(o + p * PLANE + (size_t)(i0 + rr) * W + j0), v);
}
int main() {
    float *buf[3], *in;
    for (auto &b : buf) { hipMalloc(&b, NF * FRAME * 4); hipMemset(b, 0, NF * FRAME * 4); }
    hipMalloc(&in, (size_t)450 * NF * 256 * 4 * 8); hipMemset(in, 0, (size_t)450 * NF * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)NF * FRAME * 4;
    auto run = [&](const char *name, auto launch) {
        for (int rep = 0; rep < 2; rep++) {
            for (int i = 0; i < 3; i++) launch(buf[i % 3]);
            hipEventRecord(e0);
            const int it = 20;
            for (int i = 0; i < it; i++) launch(buf[i % 3]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s %8.1f GB/s  (%.1f us per 64 frames)\n", name, bytes * it / (ms * 1e-3) / 1e9, ms / it * 1e3);
        }
    };
    const int nt = 450 * NF;
    run("plain", [&](float *b) { k_tile<0><<<nt, 256>>>(b, in); });
    run("nt", [&](float *b) { k_tile<1><<<nt, 256>>>(b, in); });
    run("sc1", [&](float *b) { k_tile<2><<<nt, 256>>>(b, in); });
    run("sc0 sc1", [&](float *b) { k_tile<3><<<nt, 256>>>(b, in); });
    run("sc0", [&](float *b) { k_tile<4><<<nt, 256>>>(b, in); });
    run("nt sc1", [&](float *b) { k_tile<5><<<nt, 256>>>(b, in); });
    run("plain", [&](float *b) { k_tile<0><<<nt, 256>>>(b, in); });
    return 0;
}
