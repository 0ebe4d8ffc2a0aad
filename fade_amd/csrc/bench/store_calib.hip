// store_calib.hip — known-byte-count kernels in the forward kernel's own access shapes, to calibrate
// rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM: widths other than 16 B/lane
// are uncalibrated).  write_dwords: one dword per lane, 256 B per wave-instruction, like the trace
// stores.  read_bytes: scattered byte loads like the window staging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void calib_write_dwords(unsigned *out, size_t n_dwords) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_dwords; i += stride) out[i] = (unsigned)i;
}
__global__ void calib_read_dwords(const unsigned *in, size_t n_dwords, unsigned *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n_dwords; i += stride) acc += in[i];
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB, beyond the 256 MiB Infinity Cache
    unsigned *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    for (int rep = 0; rep < 3; rep++) {
        calib_write_dwords<<<4096, 256>>>(buf, bytes / 4);
        calib_read_dwords<<<4096, 256>>>(buf, bytes / 4, sink);
    }
    hipDeviceSynchronize();
    printf("calib: each kernel moves %zu bytes\n", bytes);
    return 0;
}
