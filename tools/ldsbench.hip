// ldsbench.hip -- LDS read throughput per wave instruction on MI355X for the access shapes of the tap reads:
// ds_read_u8 / ds_read_u16 / ds_read_b32 / ds_read_b64 at a given byte stride between lanes (inline asm: nothing is merged).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ldsbench.hip -o tools/bin/ldsbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048;

template <int KIND> // 0 u8, 1 u16, 2 b32, 3 b64
__global__ __launch_bounds__(256) void k_lds(uint32_t *sink, int stride, int align) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (uint8_t)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t addr = (uint32_t)(uintptr_t)(lds + wave * 4096) + (uint32_t)((lane * stride) & ~(align - 1));
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    uint64_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
        if (KIND == 0)
            asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:16\n ds_read_u8 %2, %8 offset:32\n ds_read_u8 %3, %8 offset:48\n"
                         "ds_read_u8 %4, %8 offset:64\n ds_read_u8 %5, %8 offset:80\n ds_read_u8 %6, %8 offset:96\n ds_read_u8 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));
        else if (KIND == 1)
            asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:16\n ds_read_u16 %2, %8 offset:32\n ds_read_u16 %3, %8 offset:48\n"
                         "ds_read_u16 %4, %8 offset:64\n ds_read_u16 %5, %8 offset:80\n ds_read_u16 %6, %8 offset:96\n ds_read_u16 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));
        else if (KIND == 2)
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:16\n ds_read_b32 %2, %8 offset:32\n ds_read_b32 %3, %8 offset:48\n"
                         "ds_read_b32 %4, %8 offset:64\n ds_read_b32 %5, %8 offset:80\n ds_read_b32 %6, %8 offset:96\n ds_read_b32 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));
        else
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:16\n ds_read_b64 %2, %4 offset:32\n ds_read_b64 %3, %4 offset:48\n"
                         "ds_read_b64 %0, %4 offset:64\n ds_read_b64 %1, %4 offset:80\n ds_read_b64 %2, %4 offset:96\n ds_read_b64 %3, %4 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(b0 + b1 + b2 + b3);
}

__global__ __launch_bounds__(256) void k_valu(float *sink) { // clock probe: 64 dependent-free FMAs per iteration
    float a[8];
    for (int k = 0; k < 8; k++) a[k] = (float)threadIdx.x + k;
#pragma unroll 1
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = __builtin_fmaf(a[k], 1.0001f, 0.5f);
    sink[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
}

int main() {
    uint32_t *sink;
    CK(hipMalloc(&sink, 2048 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    double ghz = 2.4;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu, dim3(2048), dim3(256), 0, 0, (float *)sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        // 2048 blocks * 4 waves * ITER * 64 wave instructions, 4 cycles each, on 256 CUs * 4 SIMDs
        ghz = 2048.0 * 4 * ITER * 64 * 4 / (256.0 * 4) / (ms * 1e-3) / 1e9;
    }
    printf("VALU probe: %.3f ms -> %.2f GHz effective (assuming 4 cycles per wave64 fp32 instruction)\n", ms, ghz);
    const char *names[4] = { "ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_b64" };
    for (int kind = 0; kind < 4; kind++)
        for (int stride = 1; stride <= 16; stride = stride < 4 ? stride + 1 : stride * 2) {
            const int align = kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 4 : 8;
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_lds<0>, dim3(2048), dim3(256), 0, 0, sink, stride, align);
                else if (kind == 1) hipLaunchKernelGGL(k_lds<1>, dim3(2048), dim3(256), 0, 0, sink, stride, align);
                else if (kind == 2) hipLaunchKernelGGL(k_lds<2>, dim3(2048), dim3(256), 0, 0, sink, stride, align);
                else hipLaunchKernelGGL(k_lds<3>, dim3(2048), dim3(256), 0, 0, sink, stride, align);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double reads = 2048.0 * 4 * ITER * 8 / 256.0; // wave-level read instructions per CU
            printf("%-12s lane stride %2d B (addresses rounded down to %d): %7.3f ms  %5.2f cycles per wave instruction per CU\n", names[kind], stride, align,
                   best, best * 1e-3 * ghz * 1e9 / reads);
        }
    return 0;
}
