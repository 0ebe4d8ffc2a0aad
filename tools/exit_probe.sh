#!/bin/bash
# How long a process that used the GPU takes to go away after main() returns (GPU box): a minimal HIP program that
# prints its own elapsed time, against the wall time its parent sees.
set -e
cat > /tmp/exit_probe.hip <<'HIP'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k(int *p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv) {
    const auto t0 = std::chrono::steady_clock::now();
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    int *d = nullptr;
    void *h = nullptr;
    hipSetDevice(0);
    hipMalloc(&d, 1 << 20);
    if (mode >= 1) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); }
    if (mode >= 2) { hipHostMalloc(&h, (size_t)1 << 30, 0); for (size_t i = 0; i < ((size_t)1 << 30); i += 4096) ((char *)h)[i] = 1; }
    if (mode >= 3) { hipHostFree(h); hipFree(d); }
    fprintf(stderr, "mode %d: main ran %.3f s\n", mode, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
HIP
hipcc --offload-arch=gfx950 -O2 -o /tmp/exit_probe /tmp/exit_probe.hip 2>/dev/null
hipcc --offload-arch=gfx950 -O2 -o /tmp/exit_probe_rccl /tmp/exit_probe.hip -Wl,--no-as-needed -L/opt/rocm/lib -lrccl 2>/dev/null
python3 - <<'PY'
import subprocess, time
for exe in ("/tmp/exit_probe", "/tmp/exit_probe_rccl", "/tmp/exit_probe", "/tmp/exit_probe_rccl"):
    for m in (0, 1, 2, 3):
        t = time.time()
        subprocess.run([exe, str(m)])
        print("%s mode %d: wall %.3f s" % (exe, m, time.time() - t), flush=True)
PY
