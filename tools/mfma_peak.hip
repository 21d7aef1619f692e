// What rate do back-to-back f32 MFMAs reach on the whole chip (clock under load included)?
// 256 or 512 workgroups of 4 waves, each wave 16 independent accumulators, nothing else.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 1) void peak(float *out, int iters) {
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out;
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {256, 512, 1024})
        for (int iters : {20000, 200000}) {
            peak<<<grid, 256>>>(out, 1000);
            hipEventRecord(e0);
            peak<<<grid, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 * iters * 16 * 2048;
            printf("grid %4d iters %6d: %.2f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD\n", grid, iters, ms, flop / ms / 1e9,
                   ms * 1e6 / ((double)iters * 16 * (grid / 256.0)));
        }
    return 0;
}
