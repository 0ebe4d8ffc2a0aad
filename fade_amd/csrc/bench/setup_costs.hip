// What the first calls of the file path pay for: device allocations, pinned allocations, streams (HSA queues), first launches.
//   hipcc --offload-arch=gfx950 -O2 -o setup_costs fade_amd/csrc/bench/setup_costs.hip && ./setup_costs
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_a(int *p) { if (p) p[threadIdx.x] = 1; }
__global__ void k_b(int *p) { if (p) p[threadIdx.x] = 2; }
__global__ void k_scratch(int *p, int n) {  // a private array the compiler cannot keep in registers
    volatile int a[64];
    for (int i = 0; i < 64; i++) a[i] = i * n;
    int s = 0;
    for (int i = 0; i < n; i++) s += a[(i * 7 + threadIdx.x) & 63];
    if (p) p[threadIdx.x] = s;
}
__global__ void k_write(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, (unsigned)i);
}
int main(int argc, char **argv) {
    double t = now();
    const bool blocking = argc > 1 && atoi(argv[1]) == 1;
    if (blocking) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    printf("blocking sync: %d\n", (int)blocking);
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    printf("hipSetDevice + hipFree(0): %.2f ms\n", (now() - t) * 1e3);
    hipStream_t s0;
    t = now(); CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); printf("stream 0 create: %.2f ms\n", (now() - t) * 1e3);
    t = now(); hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s0, nullptr); CK(hipStreamSynchronize(s0)); printf("first launch on it (code object load + queue): %.2f ms\n", (now() - t) * 1e3);
    t = now(); hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s0, nullptr); CK(hipStreamSynchronize(s0)); printf("second kernel, same stream: %.3f ms\n", (now() - t) * 1e3);
    {
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0)));
        for (int rep = 0; rep < 3; rep++) {
            t = now(); hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s0, nullptr); CK(hipEventRecord(ev, s0)); CK(hipEventSynchronize(ev)); printf("kernel + event wait %d: %.3f ms\n", rep, (now() - t) * 1e3);
        }
        int *hp, *dp;
        CK(hipHostMalloc((void **)&hp, 4096)); CK(hipMalloc((void **)&dp, 4096));
        for (int rep = 0; rep < 3; rep++) {
            t = now(); CK(hipMemcpyAsync(dp, hp, 4096, hipMemcpyHostToDevice, s0)); hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s0, dp); CK(hipMemcpyAsync(hp, dp, 256, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0));
            printf("small copy up, kernel, small copy down, stream wait %d: %.3f ms\n", rep, (now() - t) * 1e3);
        }
        void *big, *dbig;
        CK(hipHostMalloc(&big, 32 << 20)); CK(hipMalloc(&dbig, 32 << 20));
        for (int rep = 0; rep < 3; rep++) {
            t = now(); CK(hipMemcpyAsync(dbig, big, 32 << 20, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
            printf("32 MB up, stream wait %d: %.3f ms\n", rep, (now() - t) * 1e3);
        }
    }
    if (argc > 3 && atoi(argv[3]) == 1) {
        // is there a stall some time AFTER streams have been made?  three streams side by side, then 60 ms of small waits
        hipStream_t ss[3];
        void *hp2, *dp2;
        CK(hipHostMalloc(&hp2, 1 << 20)); CK(hipMalloc(&dp2, 1 << 20));
        const double tz = now();
        std::vector<std::thread> th;
        for (int q = 0; q < 3; q++) th.emplace_back([&, q] { CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&ss[q], hipStreamNonBlocking)); });
        for (auto &x : th) x.join();
        printf("three streams side by side: %.3f ms\n", (now() - tz) * 1e3);
        double worst = 0, worst_at = 0;
        int n_it = 0;
        while (now() - tz < 0.080) {
            const double a = now();
            CK(hipMemcpyAsync(dp2, hp2, 1 << 20, hipMemcpyHostToDevice, s0)); hipLaunchKernelGGL(k_b, dim3(64), dim3(64), 0, s0, nullptr); CK(hipStreamSynchronize(s0));
            const double d = now() - a;
            if (d > worst) { worst = d; worst_at = a - tz; }
            if (d > 0.002) printf("  at %.2f ms: a copy + kernel + wait took %.3f ms\n", (a - tz) * 1e3, d * 1e3);
            n_it++;
        }
        printf("%d rounds in 80 ms, the slowest %.3f ms at %.2f ms\n", n_it, worst * 1e3, worst_at * 1e3);
        // and on the NEW streams: first use of each
        for (int q = 0; q < 3; q++) {
            const double a = now();
            CK(hipMemcpyAsync(dp2, hp2, 1 << 20, hipMemcpyHostToDevice, ss[q])); hipLaunchKernelGGL(k_b, dim3(64), dim3(64), 0, ss[q], nullptr); CK(hipStreamSynchronize(ss[q]));
            printf("new stream %d first use: %.3f ms\n", q, (now() - a) * 1e3);
        }
    }
    for (int q = 1; q <= 3; q++) {
        hipStream_t s;
        t = now(); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); const double tc = now() - t;
        t = now(); hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, nullptr); CK(hipStreamSynchronize(s));
        printf("stream %d: create %.2f ms, first launch %.2f ms\n", q, tc * 1e3, (now() - t) * 1e3);
    }
    {
        uint32_t mask[8]; for (auto &m : mask) m = 0xffff0000u;
        hipStream_t s;
        t = now(); CK(hipExtStreamCreateWithCUMask(&s, 8, mask)); const double tc = now() - t;
        t = now(); hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, nullptr); CK(hipStreamSynchronize(s));
        printf("CU-masked stream: create %.2f ms, first launch %.2f ms\n", tc * 1e3, (now() - t) * 1e3);
    }
    for (size_t mb : {1, 4, 32, 64, 256, 1024, 4096}) {
        void *p;
        t = now(); CK(hipMalloc(&p, mb << 20)); const double ta = now() - t;
        t = now(); CK(hipMemsetAsync(p, 0, mb << 20, s0)); CK(hipStreamSynchronize(s0)); const double tm = now() - t;
        printf("hipMalloc %5zu MB: %.3f ms (first memset %.3f ms)\n", mb, ta * 1e3, tm * 1e3);
    }
    {
        t = now();
        std::vector<void *> ps(40);
        for (auto &p : ps) CK(hipMalloc(&p, 3 << 20));
        printf("40 x hipMalloc 3 MB: %.3f ms\n", (now() - t) * 1e3);
    }
    for (size_t mb : {1, 32, 40, 128}) {
        void *p;
        t = now(); CK(hipHostMalloc(&p, mb << 20)); const double ta = now() - t;
        t = now(); memset(p, 1, mb << 20); const double tm = now() - t;
        printf("hipHostMalloc %4zu MB: %.3f ms (first touch %.3f ms)\n", mb, ta * 1e3, tm * 1e3);
    }
    {
        const size_t n = (size_t)32 << 20;
        void *p = aligned_alloc(4096, n);
        t = now(); memset(p, 1, n); const double tt = now() - t;
        t = now(); CK(hipHostRegister(p, n, hipHostRegisterDefault)); printf("hipHostRegister 32 MB (touched in %.3f ms): %.3f ms\n", tt * 1e3, (now() - t) * 1e3);
    }
    {
        // two pinned allocations side by side
        void *a, *b;
        t = now();
        std::thread th([&] { CK(hipSetDevice(0)); CK(hipHostMalloc(&a, (size_t)32 << 20)); });
        CK(hipHostMalloc(&b, (size_t)32 << 20));
        th.join();
        printf("2 x hipHostMalloc 32 MB on two threads: %.3f ms\n", (now() - t) * 1e3);
        hipStream_t s1, s2;
        t = now();
        std::thread t2([&] { CK(hipSetDevice(0)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s1, nullptr); CK(hipStreamSynchronize(s1)); });
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s2, nullptr); CK(hipStreamSynchronize(s2));
        t2.join();
        printf("2 streams made and first used on two threads: %.3f ms\n", (now() - t) * 1e3);
    }
    {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, nullptr); CK(hipStreamSynchronize(s));
        t = now(); hipLaunchKernelGGL(k_scratch, dim3(1), dim3(64), 0, s, nullptr, 5); CK(hipStreamSynchronize(s)); printf("first scratch kernel on a used stream (1 wave): %.3f ms\n", (now() - t) * 1e3);
        t = now(); hipLaunchKernelGGL(k_scratch, dim3(1), dim3(64), 0, s, nullptr, 5); CK(hipStreamSynchronize(s)); printf("again: %.3f ms\n", (now() - t) * 1e3);
        t = now(); hipLaunchKernelGGL(k_scratch, dim3(4096), dim3(256), 0, s, nullptr, 5); CK(hipStreamSynchronize(s)); printf("4096 x 256 threads of it: %.3f ms\n", (now() - t) * 1e3);
        t = now(); hipLaunchKernelGGL(k_scratch, dim3(4096), dim3(256), 0, s, nullptr, 5); CK(hipStreamSynchronize(s)); printf("again: %.3f ms\n", (now() - t) * 1e3);
        t = now(); hipLaunchKernelGGL(k_scratch, dim3(4096), dim3(256), 0, s0, nullptr, 5); CK(hipStreamSynchronize(s0)); printf("on another stream: %.3f ms\n", (now() - t) * 1e3);
    }
    {
        // staging memory: hipHostMalloc against malloc + touch + hipHostRegister: copies up, copies down, a kernel storing into it
        const size_t n = (size_t)32 << 20;
        void *hm, *hr = aligned_alloc(1 << 21, n), *d;
        CK(hipHostMalloc(&hm, n)); memset(hm, 1, n); memset(hr, 1, n);
        CK(hipHostRegister(hr, n, hipHostRegisterDefault));
        void *hr_dev = nullptr;
        CK(hipHostGetDevicePointer(&hr_dev, hr, 0));
        CK(hipMalloc(&d, n));
        for (int rep = 0; rep < 2; rep++)
            for (int which = 0; which < 2; which++) {
                void *h = which ? hr : hm;
                t = now(); CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); const double up = now() - t;
                t = now(); CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); const double dn = now() - t;
                t = now(); hipLaunchKernelGGL(k_write, dim3(512), dim3(256), 0, s0, (uint4 *)(which ? hr_dev : hm), n / 16); CK(hipStreamSynchronize(s0)); const double kw = now() - t;
                printf("%s 32 MB: up %.3f ms (%.1f GB/s), down %.3f ms (%.1f GB/s), kernel stores %.3f ms (%.1f GB/s)\n", which ? "registered " : "hipHostMalloc", up * 1e3, n / up / 1e9, dn * 1e3, n / dn / 1e9, kw * 1e3, n / kw / 1e9);
            }
        // first touch on 8 threads, then register
        void *p2 = aligned_alloc(1 << 21, n);
        t = now();
        std::vector<std::thread> th;
        for (int q = 0; q < 8; q++) th.emplace_back([=] { memset((char *)p2 + n / 8 * q, 0, n / 8); });
        for (auto &x : th) x.join();
        const double tt = now() - t;
        t = now(); CK(hipHostRegister(p2, n, hipHostRegisterDefault)); printf("32 MB touched on 8 threads in %.3f ms, registered in %.3f ms\n", tt * 1e3, (now() - t) * 1e3);
        void *p3 = aligned_alloc(1 << 21, n);
        t = now(); CK(hipHostRegister(p3, n, hipHostRegisterDefault)); printf("32 MB untouched registered in %.3f ms\n", (now() - t) * 1e3);
        t = now(); CK(hipHostUnregister(p2)); printf("unregistered in %.3f ms\n", (now() - t) * 1e3);
    }
    {
        // the file path's input buffers: filled by many threads, registered afterwards, copied up — with a large block given
        // back to the system in between (the FASTA's text)
        const size_t n = (size_t)32 << 20;
        void *bufs[3], *d;
        CK(hipMalloc(&d, n));
        char *big = (char *)malloc((size_t)200 << 20);
        memset(big, 1, (size_t)200 << 20);
        for (int b = 0; b < 3; b++) {
            bufs[b] = aligned_alloc(1 << 21, n);
            std::vector<std::thread> th;
            for (int q = 0; q < 16; q++) th.emplace_back([=] { memset((char *)bufs[b] + n / 16 * q, b + 1, n / 16); });
            for (auto &x : th) x.join();
        }
        for (int b = 0; b < 3; b++) { t = now(); CK(hipHostRegister(bufs[b], n, hipHostRegisterDefault)); printf("input buffer %d registered in %.3f ms\n", b, (now() - t) * 1e3); }
        if (argc > 2 && atoi(argv[2]) == 1) { t = now(); free(big); printf("200 MB given back in %.3f ms\n", (now() - t) * 1e3); }
        for (int rep = 0; rep < 2; rep++)
            for (int b = 0; b < 3; b++) {
                t = now(); CK(hipMemcpyAsync(d, bufs[b], n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
                printf("input buffer %d up (round %d): %.3f ms\n", b, rep, (now() - t) * 1e3);
            }
        // rewritten by the threads, copied again
        for (int b = 0; b < 3; b++) {
            std::vector<std::thread> th;
            for (int q = 0; q < 16; q++) th.emplace_back([=] { memset((char *)bufs[b] + n / 16 * q, b + 7, n / 16); });
            for (auto &x : th) x.join();
            t = now(); CK(hipMemcpyAsync(d, bufs[b], n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
            printf("input buffer %d rewritten, up: %.3f ms\n", b, (now() - t) * 1e3);
        }
    }
    hipEvent_t evt;
    t = now(); for (int i = 0; i < 16; i++) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming)); printf("16 events: %.3f ms\n", (now() - t) * 1e3);
    fflush(stdout);
    _exit(0);
}
