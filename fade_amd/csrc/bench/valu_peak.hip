// valu_peak.hip — measures the int32 VALU issue rate the DP kernel is bounded by.
// Independent chains of the exact opcodes of the SW inner loop; prints lane-ops/s per occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void valu_loop(int iters, int *out) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int c = blockIdx.x | 3;
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                              \
    if (MODE == 0) {                                                         \
        asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_sub_u32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, 31" \
                     : "+v"(x) : "v"(c));                                  \
    } else if (MODE == 1) {                                                  \
        asm volatile("v_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %1\n\tv_pk_sub_i16 %0, %0, %1\n\tv_pk_mad_u16 %0, %0, %1, %1" \
                     : "+v"(x) : "v"(c));                                  \
    } else if (MODE == 2) {                                                  \
        asm volatile("v_max3_i32 %0, %0, %1, %1\n\tv_mad_u32_u24 %0, %0, %1, %1\n\tv_bfe_u32 %0, %0, %1, 4\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" \
                     : "+v"(x) : "v"(c));                                  \
    } else if (MODE == 3) {                                                  \
        asm volatile("v_pk_add_f16 %0, %0, %1\n\tv_pk_max_f16 %0, %0, %1\n\tv_pk_add_f16 %0, %0, %1\n\tv_pk_min_f16 %0, %0, %1" \
                     : "+v"(x) : "v"(c));                                  \
    } else if (MODE == 4) {                                                  \
        asm volatile("v_add_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_sub_f32 %0, %0, %1" \
                     : "+v"(x) : "v"(c));                                  \
    } else if (MODE == 5) {                                                  \
        asm volatile("v_pk_fma_f16 %0, %0, %1, %1\n\tv_pk_mul_f16 %0, %0, %1\n\tv_pk_max_f16 %0, %0, %1\n\tv_pk_fma_f16 %0, %0, %1, %1" \
                     : "+v"(x) : "v"(c));                                  \
    } else {                                                                 \
        asm volatile("v_pk_maximum3_f16 %0, %0, %1, %1\n\tv_perm_b32 %0, %0, %1, %1\n\tv_pk_maximum3_f16 %0, %0, %1, %1\n\tv_pk_sub_u16 %0, %0, %1 clamp" \
                     : "+v"(x) : "v"(c));                                  \
    }
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, int cus) {
    int *out;
    hipMalloc(&out, sizeof(int) * 256 * cus * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {  // 256-thread blocks per CU == waves per SIMD
        const int blocks = cus * per_cu;
        valu_loop<MODE><<<blocks, 256>>>(100, out);
        hipEventRecord(e0);
        valu_loop<MODE><<<blocks, 256>>>(iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * 256 * iters * 32.0;  // lane-instructions
        printf("%s waves/SIMD=%d  %.2f T lane-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, per_cu,
               ops / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / ((double)per_cu * iters * 32.0));
    }
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("int32 add/max/sub/alignbit", p.multiProcessorCount);
    run<1>("packed i16 add/max/sub/mad ", p.multiProcessorCount);
    run<2>("max3/mad24/bfe/mov_dpp      ", p.multiProcessorCount);
    run<3>("packed f16 add/max/add/min  ", p.multiProcessorCount);
    run<4>("f32 add/max/fma/sub         ", p.multiProcessorCount);
    run<5>("packed f16 fma/mul/max/fma  ", p.multiProcessorCount);
    run<6>("pk_maximum3_f16/perm/max3/sat", p.multiProcessorCount);
    return 0;
}
