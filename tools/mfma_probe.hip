// Cycles per v_mfma_f32_16x16x4_f32 for one wave per SIMD, NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC, int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters, unsigned long long *cyc) {
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void probe32(float *out, int iters, unsigned long long *cyc) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
#define RUN(K, N, LABEL, PER)                                                         \
    for (int grid : {1, 256}) {                                                       \
        float ms;                                                                     \
        K<<<grid, 256>>>(out, iters, cyc);                                            \
        hipEventRecord(e0); K<<<grid, 256>>>(out, iters, cyc); hipEventRecord(e1);    \
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                    \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                 \
        printf("%s acc=%d grid=%d: %.1f cycles/mfma (memtime), %.3f ms, %.1f TFLOP/s\n", LABEL, N, grid, \
               (double)h / ((double)iters * N), ms, 2.0 * PER * iters * N * grid * 4 / ms / 1e9); \
    }
    RUN((probe<4, 0>), 4, "16x16x4", 1024)
    RUN((probe<12, 0>), 12, "16x16x4", 1024)
    RUN((probe<36, 0>), 36, "16x16x4", 1024)
    RUN((probe32<2>), 2, "32x32x2", 2048)
    RUN((probe32<8>), 8, "32x32x2", 2048)
    return 0;
}
