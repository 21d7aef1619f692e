// A plain store kernel with few registers and a chosen LDS footprint, to see whether its
// workgroups run beside the detection kernel's (one 467-register wave per SIMD, 131 KiB LDS)
// on the same CUs.  Built as a shared library and driven by tools/coresident_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void store_kernel(float4 *p, unsigned long long n_items, unsigned per_item, float v) {
    extern __shared__ float pad[];
    if (v == 12345.f) pad[threadIdx.x] = v;
    const float4 val = make_float4(v, v, v, v);
    for (unsigned long long it = blockIdx.x; it < n_items; it += gridDim.x) {
        float4 *q = p + it * per_item;
        for (unsigned i = threadIdx.x; i < per_item; i += 256) q[i] = val;
    }
}

extern "C" int store_launch(void *p, unsigned long long bytes, unsigned per_item_bytes, unsigned grid, int lds_bytes,
                            void *stream) {
    const unsigned per_item = per_item_bytes / 16;
    const unsigned long long n_items = bytes / per_item_bytes;
    hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (float4 *)p, n_items,
                       per_item, 1.0f);
    return (int)hipGetLastError();
}
