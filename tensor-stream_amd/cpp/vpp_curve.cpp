// vpp_curve -- the hot path at the launch sizes the reference's calling pattern produces: 1 .. 64 frames per launch.
//
// The reference converts ONE frame per VideoProcessor::Convert (reference src/Wrappers/WrapperPython.cpp:265-363) out of a decoder ring of 5
// (tensor_stream/tensor_stream.py:165) or 10 (include/Decoder.h:19) frames, on one stream per consumer name (src/VideoProcessor.cpp:98-104).  bench.py's headline
// runs 64-frame launches; this driver measures what lies between: for every n in a list, launches of n frames through the C ABI (tsvpp_convert_batch) over a ROTATING
// pool of distinct frames whose moved bytes exceed 640 MiB per issuing thread (the Infinity Cache holds 256 MiB: every launch reads and writes HBM), in two shapes:
//   TxS = 1x1   one consumer: every launch on ONE stream, back to back (each waits for its predecessor: the dependent-launch boundary is inside the figure);
//   TxS = T x S T host threads, each with S streams of its own, issuing round-robin (the reference's concurrency model: one stream per consumer name, one host
//               thread per consumer) -- launches of different streams overlap, the figure is wall time over all launches;
//   Txc         T host threads, each a NAMED CONSUMER of the context: every launch goes to tsvpp_consumer_next_stream(name) -- what VideoProcessor::ConvertInto
//               does.  Without TSVPP_OPT_INPUTS_READY that is the consumer's one stream (== Tx1 on the pool's blocking streams); with it (last argument 1) the
//               consumer alternates between its two streams and the launches carry no barrier bit.
// Every point ends with the CRC-32 (zlib polynomial) of the first and the last output frame of thread 0's LAST launch; the whole output pool is overwritten with 0xCD
// before the point, so a CRC can only match the oracle's if this point's launches wrote the frame.  Inputs are a counter hash (lowbias32) that bench.py
// regenerates on the host for the oracle.  Prints ONE JSON line.
//
// usage: vpp_curve W H pitch cl ct cr cb dstW dstH resizeType fourcc planes norm movedBytesPerFrame nList modeList [target_ms [any_order]]
//        nList "1,2,4,8"; modeList "1x1,4x1,1xc" (threads x streams per thread; c = through the consumer pool); movedBytesPerFrame sizes the pool (see main)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tsvpp.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__device__ __host__ inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// byte i of plane `plane` (0 = Y, 1 = UV) of pool frame `frame`
__global__ void fill_plane(uint8_t *dst, size_t n, uint32_t frame, uint32_t plane) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint8_t)(lowbias32((uint32_t)i + frame * 0x9E3779B1U + plane * 0x85EBCA6BU) >> 24);
}

static uint32_t crc_table[256];
static void crc_init() {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320U ^ (c >> 1) : c >> 1;
        crc_table[i] = c;
    }
}
static uint32_t crc32_of(const uint8_t *p, size_t n) {
    uint32_t c = 0xFFFFFFFFU;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFU;
}

// Restricts the calling thread to the CPUs local to the GPU's PCIe root (doorbell and kernarg writes cross the socket otherwise: the host cost of a launch moved
// between 2.9 and 5.0 us from run to run before this).  Returns the number of CPUs in the set, 0 if it could not be read.
#include <sched.h>
static cpu_set_t g_local_cpus;
static int g_local_count = 0;
static void read_local_cpus() {
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) != hipSuccess) return;
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return;
    char buf[4096] = "";
    if (!fgets(buf, sizeof(buf), f)) { fclose(f); return; }
    fclose(f);
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    (void)sched_getaffinity(0, sizeof(allowed), &allowed);
    CPU_ZERO(&g_local_cpus);
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = -1;
        if (sscanf(tok, "%d-%d", &a, &b) < 2) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &g_local_cpus); g_local_count++; }
    }
}
static void pin_local() {
    if (g_local_count > 0) (void)sched_setaffinity(0, sizeof(g_local_cpus), &g_local_cpus);
}

static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t a = 0;
    while (a <= s.size()) {
        size_t b = s.find(sep, a);
        if (b == std::string::npos) b = s.size();
        if (b > a) out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "vpp_curve: %s failed: %s\n", #x, hipGetErrorString(e_));             \
            return 3;                                                                               \
        }                                                                                           \
    } while (0)

struct Slice { // one issuing thread's pool
    int frames = 0;
    uint8_t *in = nullptr, *out = nullptr;
    std::vector<tsvpp_nv12> desc;
    std::vector<void *> outs;
    std::vector<hipStream_t> streams;
};

int main(int argc, char **argv) {
    if (argc < 17) {
        fprintf(stderr, "usage: vpp_curve W H pitch cl ct cr cb dstW dstH rt fourcc planes norm movedBytesPerFrame nList modeList [target_ms [any_order]]\n");
        return 2;
    }
    const int W = atoi(argv[1]), H = atoi(argv[2]), pitch = atoi(argv[3]);
    tsvpp_params p = {};
    p.crop_left = atoi(argv[4]); p.crop_top = atoi(argv[5]); p.crop_right = atoi(argv[6]); p.crop_bottom = atoi(argv[7]);
    p.dst_width = atoi(argv[8]); p.dst_height = atoi(argv[9]); p.resize_type = atoi(argv[10]); p.fourcc = atoi(argv[11]); p.planes = atoi(argv[12]);
    p.normalization = atoi(argv[13]);
    const double moved = atof(argv[14]);
    std::vector<int> ns;
    for (auto &s : split(argv[15], ',')) ns.push_back(atoi(s.c_str()));
    std::vector<std::pair<int, int>> modes; // (threads, streams per thread; 0 = through the consumer pool)
    for (auto &s : split(argv[16], ',')) {
        int t = 1, st = 1;
        if (s.find('c') != std::string::npos) { sscanf(s.c_str(), "%dx", &t); st = 0; }
        else sscanf(s.c_str(), "%dx%d", &t, &st);
        modes.emplace_back(std::max(1, t), std::max(0, st));
    }
    const double target_ms = argc > 17 ? atof(argv[17]) : 30.0;
    const int any_order = argc > 18 ? atoi(argv[18]) : 0;
    int max_n = 1, max_t = 1, max_s = 1;
    for (int n : ns) max_n = std::max(max_n, n);
    for (auto &m : modes) { max_t = std::max(max_t, m.first); max_s = std::max(max_s, m.second); }
    if (max_n > TSVPP_MAX_BATCH) return 2;
    crc_init();
    if (!getenv("VPP_CURVE_NO_PIN")) read_local_cpus();
    pin_local();

    tsvpp_ctx *ctx = nullptr;
    if (tsvpp_create(0, max_t, &ctx) != 0) return 4;
#ifdef TSVPP_HAVE_OPTIONS
    if (any_order && tsvpp_set_option(ctx, TSVPP_OPT_INPUTS_READY, any_order) != 0) return 4;
#else
    (void)any_order;
#endif
    const size_t out_bytes = tsvpp_out_bytes(&p, W, H);
    if (!out_bytes) { fprintf(stderr, "vpp_curve: unsupported request\n"); return 5; }
    const size_t out_stride = (out_bytes + 255) & ~(size_t)255;
    const size_t y_bytes = (size_t)pitch * H, in_stride = y_bytes * 3 / 2;
    // Frames per issuing thread: the bytes a pass over the pool READS exceed 768 MiB (three times the 256 MiB Infinity Cache: a cyclic sweep then finds nothing of
    // its previous pass -- the first version of this driver sized the pool by moved bytes alone, 64 headline frames, whose 212 MB of inputs stayed cache-resident
    // and flattered the large launches: 0.90 "of the roofline" at n = 64) and the bytes it moves exceed 2 GiB; a multiple of 64, at most 1024.
    const double read_est = std::max(moved - (double)out_bytes, 1.0);
    int per_thread = (int)std::max((768.0 * 1048576.0) / read_est, (2048.0 * 1048576.0) / (moved > 0 ? moved : 1.0)) + 1;
    const int gran = std::max(64, max_n);
    per_thread = std::min(1024, (per_thread + gran - 1) / gran * gran);

    std::vector<Slice> pool((size_t)max_t);
    for (int t = 0; t < max_t; t++) {
        Slice &s = pool[(size_t)t];
        s.frames = per_thread;
        CK(hipMalloc((void **)&s.in, in_stride * per_thread));
        CK(hipMalloc((void **)&s.out, out_stride * per_thread));
        for (int f = 0; f < per_thread; f++) {
            uint8_t *y = s.in + (size_t)f * in_stride;
            const uint32_t id = (uint32_t)(t * per_thread + f);
            fill_plane<<<(unsigned)((y_bytes + 255) / 256), 256>>>(y, y_bytes, id, 0);
            fill_plane<<<(unsigned)((y_bytes / 2 + 255) / 256), 256>>>(y + y_bytes, y_bytes / 2, id, 1);
            s.desc.push_back(tsvpp_nv12{ y, y + y_bytes, pitch, pitch, W, H });
            s.outs.push_back(s.out + (size_t)f * out_stride);
        }
        for (int k = 0; k < std::max(1, max_s); k++) {
            hipStream_t st = nullptr;
            CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            s.streams.push_back(st);
        }
    }
    CK(hipDeviceSynchronize());

    printf("{\"driver\": \"vpp_curve\", \"pinned_to_gpu_local_cpus\": %d, \"pool_frames_per_thread\": %d, \"pool_moved_MiB_per_thread\": %.1f, \"out_bytes\": %zu, \"any_order\": %d, \"points\": [", g_local_count, per_thread,
           per_thread * moved / 1048576.0, out_bytes, any_order);
    bool first_point = true;
    std::vector<uint8_t> host(out_bytes);
    for (auto &mode : modes) {
        const int T = mode.first, S = std::max(1, mode.second);
        const bool pool_mode = mode.second == 0;
        for (int n : ns) {
            if (tsvpp_prepare_batch(ctx, &p, W, H, n, nullptr) != 0) return 6;
            for (int t = 0; t < max_t; t++) CK(hipMemsetAsync(pool[(size_t)t].out, 0xCD, out_stride * per_thread, nullptr));
            CK(hipDeviceSynchronize());
            // launch i of a thread converts its frames [(i n) mod F, +n) on its stream i mod S
            const size_t launch_bytes = (size_t)((double)n * moved); // what the caller knows the launch moves
            auto issue = [&](Slice &s, long i, int t = 0) {
                const int base = (int)((i * (long)n) % s.frames);
                void *st = s.streams[(size_t)(i % S)];
                if (pool_mode) {
                    char name[16];
                    snprintf(name, sizeof(name), "c%d", t);
                    if (tsvpp_consumer_next_stream(ctx, name, launch_bytes, &st) != 0) return -1;
                }
                return tsvpp_convert_batch(ctx, n, s.desc.data() + base, &p, s.outs.data() + base, st);
            };
            auto sync_thread = [&](Slice &s, int t) {
                if (pool_mode) {
                    char name[16];
                    snprintf(name, sizeof(name), "c%d", t);
                    (void)tsvpp_consumer_synchronize(ctx, name);
                    return;
                }
                for (int q = 0; q < S; q++) (void)hipStreamSynchronize(s.streams[(size_t)q]);
            };
            // calibration + warm-up on thread 0's slice: ~25 ms of work settles the clocks
            long done0 = 0;
            double per_launch_us = 0;
            {
                const double t0 = now_us();
                long k = 0;
                while (now_us() - t0 < 25e3) {
                    for (int j = 0; j < 16; j++)
                        if (issue(pool[0], k++) != 0) return 7;
                    sync_thread(pool[0], 0);
                }
                per_launch_us = (now_us() - t0) / (double)k;
                done0 = k;
            }
            long K = (long)(target_ms * 1e3 / per_launch_us * (T > 1 ? 1.0 : 1.0));
            K = std::max<long>(32, std::min<long>(K, 20000));
            // every thread starts its launch counter where a full pool pass begins, so the last launch is well defined
            std::vector<double> reps_us, host_us;
            long last_i = 0;
            const int kReps = 5;
            for (int r = 0; r < kReps; r++) {
                std::atomic<int> ready{ 0 };
                std::atomic<bool> go{ false };
                std::vector<double> t_end((size_t)T, 0.0), t_host((size_t)T, 0.0);
                std::vector<int> err((size_t)T, 0);
                float ev_ms = 0.0f;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (T == 1 && S == 1 && !pool_mode) {
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                }
                double t_start = 0;
                auto body = [&](int t) {
                    pin_local();
                    Slice &s = pool[(size_t)t];
                    ready.fetch_add(1);
                    while (!go.load(std::memory_order_acquire)) {}
                    const double h0 = now_us();
                    if (e0) (void)hipEventRecord(e0, s.streams[0]);
                    const long i0 = done0 + (long)r * K;
                    for (long i = 0; i < K; i++)
                        if (issue(s, i0 + i, t) != 0) { err[(size_t)t] = 1; break; }
                    if (e1) (void)hipEventRecord(e1, s.streams[0]);
                    t_host[(size_t)t] = now_us() - h0;
                    sync_thread(s, t);
                    t_end[(size_t)t] = now_us();
                };
                std::vector<std::thread> th;
                for (int t = 1; t < T; t++) th.emplace_back(body, t);
                while (ready.load() < T - 1) {}
                t_start = now_us();
                go.store(true, std::memory_order_release);
                body(0);
                for (auto &x : th) x.join();
                for (int t = 0; t < T; t++)
                    if (err[(size_t)t]) return 8;
                double wall = 0, hmax = 0;
                for (int t = 0; t < T; t++) { wall = std::max(wall, t_end[(size_t)t] - t_start); hmax = std::max(hmax, t_host[(size_t)t]); }
                if (e0) {
                    CK(hipEventElapsedTime(&ev_ms, e0, e1));
                    wall = ev_ms * 1e3; // one stream: the device time between the events (the launch stream's own clock)
                    CK(hipEventDestroy(e0));
                    CK(hipEventDestroy(e1));
                }
                reps_us.push_back(wall / (double)(K * T));
                host_us.push_back(hmax / (double)K);
                last_i = done0 + (long)r * K + K - 1;
            }
            std::vector<double> sorted = reps_us;
            std::sort(sorted.begin(), sorted.end());
            const double med = sorted[sorted.size() / 2];
            std::sort(host_us.begin(), host_us.end());
            // parity material: first and last frame of thread 0's last launch
            const int base = (int)((last_i * (long)n) % pool[0].frames);
            const int chk[2] = { base, base + n - 1 };
            uint32_t crc[2];
            for (int c = 0; c < 2; c++) {
                CK(hipMemcpy(host.data(), pool[0].outs[(size_t)chk[c]], out_bytes, hipMemcpyDeviceToHost));
                crc[c] = crc32_of(host.data(), out_bytes);
            }
            printf("%s{\"n\": %d, \"threads\": %d, \"streams_per_thread\": %d, \"consumer_pool\": %d, \"launches_per_thread\": %ld, \"us_per_launch\": %.3f, \"us_per_launch_reps\": [%.3f, %.3f, %.3f, %.3f, %.3f], "
                   "\"host_issue_us_per_launch\": %.3f, \"timer\": \"%s\", \"check\": [{\"frame\": %d, \"crc32\": %u}, {\"frame\": %d, \"crc32\": %u}]}",
                   first_point ? "" : ", ", n, T, pool_mode ? 0 : S, pool_mode ? 1 : 0, K, med, reps_us[0], reps_us[1], reps_us[2], reps_us[3], reps_us[4], host_us[host_us.size() / 2],
                   (T == 1 && S == 1 && !pool_mode) ? "hip events on the launch stream" : "wall clock, start barrier -> every stream synchronised", chk[0], crc[0], chk[1], crc[1]);
            first_point = false;
            fflush(stdout);
        }
    }
    printf("]}\n");
    for (auto &s : pool) {
        for (auto st : s.streams) (void)hipStreamDestroy(st);
        (void)hipFree(s.in);
        (void)hipFree(s.out);
    }
    tsvpp_destroy(ctx);
    return 0;
}
