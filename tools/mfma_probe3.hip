// Issue behaviour of v_mfma_f32_16x16x4_f32 next to other instructions of the same wave
// (one wave per SIMD).  Every variant runs 36 MFMAs per loop iteration on independent
// accumulators and reports shader clocks per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define M_AGPR(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b))
#define M_VGPR(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define M_VB_A(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(bacc))
#define V_ADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(a))
#define V_MINU(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]))
#define V_CMPSEL(j) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(a) : "vcc")
#define V_MED3(j) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]))
#define V_LSHLADD(j) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]))
#define V_PKADD(j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(p[(j + 1) & 3]))
#define ACCRD(j, i) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[j]) : "a"(bacc))

template <int MODE, int K>
__global__ __launch_bounds__(256, 1) void probe(float *out, int iters, unsigned long long *cyc) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = i;
    __syncthreads();
    constexpr int NACC = 36;
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f, bacc = b;
    float v[8];
    unsigned u[8];
    float2 p[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = threadIdx.x + j; u[j] = threadIdx.x * 7 + j; }
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = make_float2(j, threadIdx.x);
    float l = 0;
    unsigned sc[4] = {1, 2, 3, 4};
    float gl[4] = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (MODE == 0) {          // MFMA (AGPR acc) + K v_add_f32
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_ADD(j);
            } else if constexpr (MODE == 1) {   // MFMA (VGPR acc) + K v_add_f32
                M_VGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_ADD(j);
            } else if constexpr (MODE == 2) {   // MFMA (VGPR acc, B from AGPR) + K v_add_f32
                M_VB_A(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_ADD(j);
            } else if constexpr (MODE == 3) {   // MFMA + K v_min_u32
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_MINU(j);
            } else if constexpr (MODE == 4) {   // MFMA + K (v_cmp, s_nop 1, v_cndmask)
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_CMPSEL(j);
            } else if constexpr (MODE == 5) {   // MFMA + K v_med3_u32 (VOP3)
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_MED3(j);
            } else if constexpr (MODE == 6) {   // groups: 3 MFMA then 3K v_add_f32
                M_AGPR(i);
                if (i % 3 == 2) {
#pragma unroll
                    for (int j = 0; j < 3 * K; ++j) V_ADD(j & 7);
                }
            } else if constexpr (MODE == 7) {   // MFMA + K v_pk_add_f32
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_PKADD(j & 3);
            } else if constexpr (MODE == 8) {   // MFMA + K ds_read_b32 (waited for one iteration later)
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) l += sh[(u[j] + i) & 4095];
            } else if constexpr (MODE == 9) {   // MFMA + K v_accvgpr_read
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) ACCRD(j, i);
            } else if constexpr (MODE == 11) {  // MFMA + K scalar adds
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sc[j & 3]) : "s"(it) : "scc");
            } else if constexpr (MODE == 12) {  // MFMA + K global loads (waited for at the end of the iteration)
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) asm volatile("global_load_dword %0, %1, off" : "=v"(gl[j & 3]) : "v"(out + ((threadIdx.x + 64 * j) & 1023)) : "memory");
                if (i == NACC - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if constexpr (MODE == 10) {  // MFMA + K v_lshl_add_u32
                M_AGPR(i);
#pragma unroll
                for (int j = 0; j < K; ++j) V_LSHLADD(j);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = l;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j] + u[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) s += p[j].x + p[j].y + sc[j] + gl[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 3000;
    const char *names[] = {"agpr acc + v_add_f32", "vgpr acc + v_add_f32", "vgpr acc, B in agpr + v_add_f32", "v_min_u32",
                           "v_cmp/s_nop 1/v_cndmask", "v_med3_u32", "3 mfma then 3K v_add_f32", "v_pk_add_f32",
                           "ds_read_b32", "v_accvgpr_read", "v_lshl_add_u32", "s_add_u32", "global_load_dword"};
#define RUN(MODE, K)                                                                  \
    {                                                                                 \
        probe<MODE, K><<<256, 256>>>(out, iters, cyc);                                \
        hipDeviceSynchronize();                                                       \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                 \
        printf("%-34s K=%d: %.1f clocks/mfma\n", names[MODE], K, (double)h / ((double)iters * 36)); \
    }
#define RUNS(MODE) RUN(MODE, 0) RUN(MODE, 2) RUN(MODE, 4) RUN(MODE, 5) RUN(MODE, 6) RUN(MODE, 7) RUN(MODE, 8)
    RUNS(0) RUNS(1) RUNS(5) RUNS(7) RUNS(8) RUNS(6)
    RUN(11, 2) RUN(11, 4) RUN(11, 8) RUN(12, 1) RUN(12, 2) RUN(12, 4)
    return 0;
}
