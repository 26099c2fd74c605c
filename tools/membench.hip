// membench.hip -- HBM streaming ceilings on MI355X for the access shapes the VPP kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float4 v, float4 *p) { vf4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (vf4 *)p); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_write(float4 *o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (; i < n; i += st) o[i] = v;
}
__global__ void k_write_nt(float4 *o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (; i < n; i += st) nt_store(v, &o[i]);
}
__global__ void k_read(const float4 *in, float *sink, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += st) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_copy(const float4 *in, float4 *o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) o[i] = in[i];
}
// one block = one contiguous 16 KiB slab (no grid stride): the "tile per workgroup" shape
__global__ void k_write_slab(float4 *o, size_t n) {
    size_t base = (size_t)blockIdx.x * 1024;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
#pragma unroll
    for (int k = 0; k < 4; k++) { size_t i = base + k * 256 + threadIdx.x; if (i < n) o[i] = v; }
}
// read 1 byte-quad + write 3 float4 (C2-like: 4 px in, 3 planes out), tile per block
__global__ void k_expand(const uint32_t *in, float4 *o, size_t npx4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npx4) return;
    uint32_t p = in[i];
    float4 v = make_float4((float)(p & 255), (float)((p >> 8) & 255), (float)((p >> 16) & 255), (float)(p >> 24));
    o[i] = v; o[i + npx4] = v; o[i + 2 * npx4] = v;
}
__global__ void k_expand_nt(const uint32_t *in, float4 *o, size_t npx4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= npx4) return;
    uint32_t p = in[i];
    float4 v = make_float4((float)(p & 255), (float)((p >> 8) & 255), (float)((p >> 16) & 255), (float)(p >> 24));
    nt_store(v, &o[i]); nt_store(v, &o[i + npx4]); nt_store(v, &o[i + 2 * npx4]);
}

int main() {
    const size_t bytes = (size_t)1 << 31; // 2 GiB buffers >> 256 MiB Infinity Cache
    float4 *a, *b; float *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double moved, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        const int it = 10;
        for (int i = 0; i < it; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f GB/s  (%.3f ms)\n", name, moved * it / (ms * 1e-3) / 1e9, ms / it);
    };
    for (int g : {2048, 8192, 65536}) {
        printf("grid %d x 256\n", g);
        run("read float4", (double)bytes, [&] { k_read<<<g, 256>>>(a, sink, n); });
        run("write float4", (double)bytes, [&] { k_write<<<g, 256>>>(a, n); });
        run("write float4 nontemporal", (double)bytes, [&] { k_write_nt<<<g, 256>>>(a, n); });
        run("copy float4 (r+w bytes)", 2.0 * bytes, [&] { k_copy<<<g, 256>>>(a, b, n); });
    }
    run("write slab/block 16KiB", (double)bytes, [&] { k_write_slab<<<(unsigned)(n / 1024), 256>>>(a, n); });
    {
        const size_t npx4 = bytes / 16 / 3; // out = 3 * npx4 float4
        run("expand u8x4 -> 3 x float4", (double)npx4 * 52, [&] { k_expand<<<(unsigned)((npx4 + 255) / 256), 256>>>((const uint32_t *)b, a, npx4); });
        run("expand, nontemporal stores", (double)npx4 * 52, [&] { k_expand_nt<<<(unsigned)((npx4 + 255) / 256), 256>>>((const uint32_t *)b, a, npx4); });
    }
    return 0;
}
