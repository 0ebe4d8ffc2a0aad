// What one more HIP stream (an HSA queue) costs a short-lived process, start to finish: queue_cost <n_streams> [pinned MB]
// makes the streams, uses each once and leaves by _exit; time the whole process from outside (tools/r04/queue_cost.py).
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(int *p) { if (p) p[threadIdx.x] = 1; }
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1;
    const size_t pin_mb = argc > 2 ? (size_t)atoi(argv[2]) : 0;
    if (hipSetDevice(0) != hipSuccess) return 1;
    hipStream_t s[16];
    for (int i = 0; i < n && i < 16; i++) {
        if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) return 1;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[i], nullptr);
        if (hipStreamSynchronize(s[i]) != hipSuccess) return 1;
    }
    if (pin_mb) {
        void *p = aligned_alloc(1 << 21, pin_mb << 20);
        if (hipHostRegister(p, pin_mb << 20, hipHostRegisterDefault) != hipSuccess) return 1;
    }
    _exit(0);
}
