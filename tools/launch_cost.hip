// launch_cost -- what one kernel launch costs the HOST on this box, by kernarg size and launch API (round 6: the 1-frame-per-launch regime is host-bound at
// ~4.2 us per tsvpp_convert_batch call; how much of that is the runtime's, how much the 3.7 KiB kernarg segment's, how much ours?).
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/launch_cost tools/launch_cost.hip && tools/bin/launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int BYTES> struct Blob { unsigned char b[BYTES]; };
template <int BYTES> __global__ void k_args(const Blob<BYTES> a, int *sink) {
    if (a.b[0] == 255 && sink) *sink = 1; // (never true: the blob is zero)
}

template <int BYTES> static void run(const char *label, int mode, int nstreams, hipStream_t *st) {
    Blob<BYTES> blob;
    memset(&blob, 0, sizeof(blob));
    const int iters = 20000;
    int *sink = nullptr;
    for (int rep = 0; rep < 2; rep++) { // first repetition warms up
        const double t0 = now_us();
        for (int i = 0; i < iters; i++) {
            hipStream_t s = st[i % nstreams];
            if (mode == 0) hipLaunchKernelGGL((k_args<BYTES>), dim3(1), dim3(64), 0, s, blob, sink);
            else hipExtLaunchKernelGGL((k_args<BYTES>), dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, blob, sink);
        }
        const double t1 = now_us();
        (void)hipDeviceSynchronize();
        const double t2 = now_us();
        if (rep) printf("%-34s kernarg %5d B  streams %d  host %.2f us/launch  wall %.2f us/launch\n", label, BYTES, nstreams, (t1 - t0) / iters, (t2 - t0) / iters);
    }
}

int main() {
    hipStream_t st[4];
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    run<16>("hipLaunchKernelGGL", 0, 1, st);
    run<512>("hipLaunchKernelGGL", 0, 1, st);
    run<1024>("hipLaunchKernelGGL", 0, 1, st);
    run<2048>("hipLaunchKernelGGL", 0, 1, st);
    run<3700>("hipLaunchKernelGGL", 0, 1, st);
    run<16>("hipLaunchKernelGGL", 0, 2, st);
    run<3700>("hipLaunchKernelGGL", 0, 2, st);
    run<16>("hipExtLaunchKernelGGL any-order", 1, 1, st);
    run<3700>("hipExtLaunchKernelGGL any-order", 1, 1, st);
    run<16>("hipExtLaunchKernelGGL any-order", 1, 2, st);
    // the same launches issued through a pre-resolved function handle and one packed argument buffer (hipModuleLaunchKernel + HIP_LAUNCH_PARAM_BUFFER_POINTER)
    {
        hipFunction_t f = nullptr;
        if (hipGetFuncBySymbol(&f, (const void *)k_args<3700>) == hipSuccess && f) {
            struct { Blob<3700> a; int *sink; } args;
            memset(&args, 0, sizeof(args));
            size_t sz = sizeof(args);
            void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END };
            const int iters = 20000;
            for (int rep = 0; rep < 2; rep++) {
                const double t0 = now_us();
                for (int i = 0; i < iters; i++) (void)hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, st[0], nullptr, extra);
                const double t1 = now_us();
                (void)hipDeviceSynchronize();
                const double t2 = now_us();
                if (rep) printf("%-34s kernarg %5d B  streams %d  host %.2f us/launch  wall %.2f us/launch\n", "hipModuleLaunchKernel(extra)", 3700, 1, (t1 - t0) / iters, (t2 - t0) / iters);
            }
        } else {
            printf("hipGetFuncBySymbol failed\n");
        }
    }
    return 0;
}
