// HBM write patterns: which way of handing out a big buffer to workgroups streams fastest?
//   chunk  one workgroup per contiguous chunk, as many workgroups as chunks (not persistent)
//   parts  persistent workgroups, each taking contiguous parts in turn (item = block + k*grid)
//   stride persistent workgroups, grid-stride over float4s
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T>
__global__ __launch_bounds__(T) void chunk(float4 *p, size_t per_block, float v) {
    float4 *q = p + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += T) q[i] = make_float4(v, v, v, v);
}
template <int T>
__global__ __launch_bounds__(T) void parts(float4 *p, size_t n_items, size_t per_item, float v) {
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        float4 *q = p + it * per_item;
        for (size_t i = threadIdx.x; i < per_item; i += T) q[i] = make_float4(v, v, v, v);
    }
}
// parts, but every float4 comes from a small table (period floats, 4 rotated copies) read
// through the vector cache, or from a copy of it in LDS
template <int T, bool LDS>
__global__ __launch_bounds__(T) void ptab(float4 *p, size_t n_items, size_t per_item, const float *tab, int period, int tstride) {
    __shared__ float s[4 * 896];
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const float *src = tab + (it % 256) * 4 * tstride;
        if (LDS) {
            __syncthreads();
            for (int i = threadIdx.x; i < 4 * tstride; i += T) s[i] = src[i];
            __syncthreads();
            src = s;
        }
        float4 *q = p + it * per_item;
        int j = (4 * threadIdx.x) % period;
        const int jstep = (4 * T) % period;
        for (size_t i = threadIdx.x; i < per_item; i += T) {
            const int r = j & 3;
            q[i] = *(const float4 *)(src + r * tstride + (j - r));
            j += jstep;
            j = j >= period ? j - period : j;
        }
    }
}
// parts with non-temporal stores
typedef float fx4 __attribute__((ext_vector_type(4)));
template <int T>
__global__ __launch_bounds__(T) void parts_nt(fx4 *p, size_t n_items, size_t per_item, float v) {
    const fx4 val = {v, v, v, v};
    for (size_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        fx4 *q = p + it * per_item;
        for (size_t i = threadIdx.x; i < per_item; i += T) __builtin_nontemporal_store(val, q + i);
    }
}
template <int T>
__global__ __launch_bounds__(T) void chunk_nt(fx4 *p, size_t per_block, float v) {
    const fx4 val = {v, v, v, v};
    fx4 *q = p + (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += T) __builtin_nontemporal_store(val, q + i);
}
template <int T>
__global__ __launch_bounds__(T) void stride(float4 *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) p[i] = make_float4(v, v, v, v);
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 32) << 30, n = bytes / 16;
    printf("buffer %zu GiB\n", bytes >> 30);
    float4 *a;
    CK(hipMalloc(&a, bytes));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](auto launch, const char *what, long x, long y) {
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
        printf("%-8s %8ld %6ld: %.0f GB/s\n", what, x, y, bytes / best / 1e6);
    };
    for (long kb : {4, 8, 16, 64}) {
        const size_t per = (size_t)kb * 1024 / 16;
        time([&] { chunk<256><<<(unsigned)(n / per), 256>>>(a, per, 1.f); }, "chunk256", kb, 0);
    }
    for (long kb : {192}) {
        const size_t per = (size_t)kb * 1024 / 16;
        for (long grid : {512, 1024, 2048}) time([&] { parts<512><<<(unsigned)grid, 512>>>(a, n / per, per, 1.f); }, "parts512", kb, grid);
    }
    for (long kb : {4, 16}) {
        const size_t per = (size_t)kb * 1024 / 16;
        time([&] { chunk_nt<256><<<(unsigned)(n / per), 256>>>((fx4 *)a, per, 1.f); }, "chunk-nt", kb, 0);
    }
    for (long kb : {192, 768}) {
        const size_t per = (size_t)kb * 1024 / 16;
        for (long grid : {512, 1024}) {
            time([&] { parts<512><<<(unsigned)grid, 512>>>(a, n / per, per, 1.f); }, "parts512", kb, grid);
            time([&] { parts_nt<512><<<(unsigned)grid, 512>>>((fx4 *)a, n / per, per, 1.f); }, "parts-nt", kb, grid);
        }
    }
    float *tab;
    CK(hipMalloc(&tab, 256 * 4 * 896 * 4));
    CK(hipMemset(tab, 0, 256 * 4 * 896 * 4));
    for (long grid : {512, 1024}) {
        const size_t per = 192 * 1024 / 16;
        time([&] { ptab<512, false><<<(unsigned)grid, 512>>>(a, n / per, per, tab, 882, 896); }, "ptab-l1", 192, grid);
        time([&] { ptab<512, true><<<(unsigned)grid, 512>>>(a, n / per, per, tab, 882, 896); }, "ptab-lds", 192, grid);
    }
    for (long grid : {1024, 4096, 16384}) time([&] { stride<256><<<(unsigned)grid, 256>>>(a, n, 1.f); }, "stride", 0, grid);
    return 0;
}
