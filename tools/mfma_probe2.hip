// Cycles per v_mfma_f32_16x16x4_f32 with K independent VALU ops (and optionally one LDS read) per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int K, int LDS>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters, unsigned long long *cyc) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = i;
    __syncthreads();
    constexpr int NACC = 36;
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    float l = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = v[j] * 3 + it;          // one VALU each (mad)
            if (LDS && (i % LDS) == 0) l += sh[(v[0] + i) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 5000;
#define RUN(K, L)                                                                     \
    {                                                                                 \
        probe<K, L><<<256, 256>>>(out, iters, cyc);                                   \
        hipDeviceSynchronize();                                                       \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                 \
        printf("valu/mfma=%d lds_every=%d: %.1f cycles/mfma\n", K, L, (double)h / ((double)iters * 36)); \
    }
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(4, 0) RUN(5, 0) RUN(6, 0) RUN(8, 0)
    RUN(0, 4) RUN(2, 4) RUN(2, 1)
    return 0;
}
