// HBM streaming ceilings of the box: float4 fill, float4 read-reduce, float4 copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill(float4 *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(256) void rsum(const float4 *p, size_t n, float *out) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void copy(const float4 *a, float4 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
    const size_t bytes = (size_t)48 << 30;
    const size_t n = bytes / 16;
    float4 *a, *b;
    float *o;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&o, 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {2048, 4096, 8192, 65536}) {
        for (int k = 0; k < 3; ++k) {
            float ms;
            hipEventRecord(e0);
            if (k == 0) fill<<<grid, 256>>>(a, n, 1.f);
            if (k == 1) rsum<<<grid, 256>>>(a, n, o);
            if (k == 2) copy<<<grid, 256>>>(a, b, n);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            hipEventElapsedTime(&ms, e0, e1);
            // second, warm run
            hipEventRecord(e0);
            if (k == 0) fill<<<grid, 256>>>(a, n, 1.f);
            if (k == 1) rsum<<<grid, 256>>>(a, n, o);
            if (k == 2) copy<<<grid, 256>>>(a, b, n);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            hipEventElapsedTime(&ms, e0, e1);
            const double moved = (k == 2 ? 2.0 : 1.0) * bytes;
            printf("grid %6d %s: %.2f ms, %.0f GB/s\n", grid, k == 0 ? "fill" : k == 1 ? "read" : "copy", ms, moved / ms / 1e6);
        }
    }
    return 0;
}
