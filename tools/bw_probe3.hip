// HBM write ceiling of two large buffers allocated in one process (the physical region a buffer
// lands in matters): persistent workgroups writing contiguous 192 KB parts, and a grid-stride
// sweep of float4 stores, on each buffer.   usage: bw_probe3 [GiB per buffer]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T>
__global__ __launch_bounds__(T) void parts(float4 *p, size_t n_items, size_t per_item, float v) {
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        float4 *q = p + it * per_item;
        for (size_t i = threadIdx.x; i < per_item; i += T) q[i] = make_float4(v, v, v, v);
    }
}
template <int T>
__global__ __launch_bounds__(T) void stride(float4 *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) p[i] = make_float4(v, v, v, v);
}
template <int T>
__global__ __launch_bounds__(T) void rd(const float4 *p, size_t n, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) *out = s;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 114) << 30, n = bytes / 16;
    float4 *buf[2];
    float *out;
    CK(hipMalloc(&buf[0], bytes));
    CK(hipMalloc(&buf[1], bytes));
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](auto launch) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            float ms;
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        return bytes / best / 1e6;
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int b = 0; b < 2; ++b) {
            const size_t per = 192 * 1024 / 16;
            const double p1 = time([&] { parts<512><<<1024, 512>>>(buf[b], n / per, per, 1.f); });
            const double p2 = time([&] { stride<256><<<16384, 256>>>(buf[b], n, 1.f); });
            const double p3 = time([&] { rd<256><<<16384, 256>>>(buf[b], n, out); });
            printf("buffer %d (%p): parts512 192 KB x1024 %.0f GB/s, stride 16384x256 %.0f GB/s, read 16384x256 %.0f GB/s\n", b,
                   (void *)buf[b], p1, p2, p3);
        }
    return 0;
}
