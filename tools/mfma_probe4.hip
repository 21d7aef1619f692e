// Do MFMAs of one wave overlap with vector-ALU / LDS work of ANOTHER wave on the same SIMD?
// 512 threads per workgroup = 2 waves per SIMD.  Waves 0-3 run MFMAs (optionally with K
// VALU ops each); waves 4-7 run VALU ops or LDS reads only.  Reports clocks per MFMA for the
// first kind and clocks per instruction for the second.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define M_AGPR(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b))
#define V_ADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a))

template <int MODE, int K>
__global__ __launch_bounds__(512, 1) void probe(float *out, int iters, unsigned long long *cyc) {
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) sh[i] = i;
    __syncthreads();
    constexpr int NACC = 16;
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    const int wave = threadIdx.x >> 6;
    float l = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || MODE == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_ADD(j);
            }
        }
    } else if (MODE == 0) {          // VALU-only companion: 8 v_add per step
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) V_ADD(j);
            }
        }
    } else if (MODE == 1) {          // LDS-read companion: 4 ds_read_b32 + 4 v_add per step
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) l += sh[(threadIdx.x * 5 + i * 67 + j * 129 + it) & 8191];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 4000;
#define RUN(MODE, K, WHAT)                                                            \
    {                                                                                 \
        probe<MODE, K><<<256, 512>>>(out, iters, cyc);                                \
        hipDeviceSynchronize();                                                       \
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);                                 \
        printf("%-44s K=%d: mfma wave %.1f clocks/mfma, companion %.1f clocks/step\n", WHAT, K, \
               (double)h[0] / ((double)iters * 16), (double)h[4] / ((double)iters * 16)); \
    }
    RUN(0, 0, "mfma wave + wave of 8 v_add per step")
    RUN(0, 2, "mfma wave + wave of 8 v_add per step")
    RUN(1, 0, "mfma wave + wave of 4 ds_read+v_add per step")
    RUN(2, 0, "two mfma waves per SIMD")
    RUN(2, 2, "two mfma waves per SIMD")
    RUN(2, 4, "two mfma waves per SIMD")
    RUN(2, 6, "two mfma waves per SIMD")
    return 0;
}
