// Window staging by LDS-DMA (global_load_lds): shared by the detection and sync kernels.
#pragma once
#include <hip/hip_runtime.h>

// CNT loads of 16 (X4) or 4 bytes per lane from p + j*STRIDE bytes straight into LDS at
// lds + j*STRIDE + lane*(16 or 4): the immediate offset applies to both addresses.  M0 holds
// the LDS base for the duration and is restored.
template <int CNT, int STRIDE, bool X4>
__device__ __forceinline__ void dma_group(const float *p, unsigned lds) {
    static_assert(CNT >= 1 && CNT <= 4, "group size");
    unsigned keep_m0;
#define WN_DMA_ASM(OP)                                                                                       \
    if constexpr (CNT == 1)                                                                                  \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" OP " %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_m0) : "v"(p), "s"(lds) : "memory");                                        \
    else if constexpr (CNT == 2)                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" OP " %1, off\n\t" OP              \
                     " %1, off offset:%3\n\ts_mov_b32 m0, %0"                                                \
                     : "=&s"(keep_m0) : "v"(p), "s"(lds), "n"(STRIDE) : "memory");                           \
    else if constexpr (CNT == 3)                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" OP " %1, off\n\t" OP              \
                     " %1, off offset:%3\n\t" OP " %1, off offset:%4\n\ts_mov_b32 m0, %0"                    \
                     : "=&s"(keep_m0) : "v"(p), "s"(lds), "n"(STRIDE), "n"(2 * STRIDE) : "memory");          \
    else                                                                                                     \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" OP " %1, off\n\t" OP              \
                     " %1, off offset:%3\n\t" OP " %1, off offset:%4\n\t" OP                                 \
                     " %1, off offset:%5\n\ts_mov_b32 m0, %0"                                                \
                     : "=&s"(keep_m0) : "v"(p), "s"(lds), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE)      \
                     : "memory");
    if constexpr (X4) {
        WN_DMA_ASM("global_load_lds_dwordx4")
    } else {
        WN_DMA_ASM("global_load_lds_dword")
    }
#undef WN_DMA_ASM
}

// The same transfers one instruction at a time, for issuing them between other instructions:
// the LDS base is set once per run of pieces (wn_m0_set), a piece adds its immediate offset to
// both addresses, and the global address is a scalar base plus a per-lane byte offset.
__device__ __forceinline__ unsigned wn_m0_get() {
    unsigned v;
    asm volatile("s_mov_b32 %0, m0" : "=s"(v));
    return v;
}
__device__ __forceinline__ void wn_m0_set(unsigned v) {
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);   // the same in every lane; no cost when already scalar
    // The compiler does not know that M0 changes here (it refuses M0 in a clobber list as a
    // reserved register).  Between a wn_m0_set and the one that restores the saved value there must
    // be no instruction of the compiler's own that uses M0 (indexed register moves, lane reads by
    // M0, LDS instructions that take their base from it): tools/check_m0.py disassembles the built
    // kernels and fails if any instruction other than these moves and the DMA loads touches M0.
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(v) : "memory");
}
template <int OFF, bool X4>
__device__ __forceinline__ void dma_piece(const float *base, unsigned lane_bytes) {
    static_assert(OFF >= 0 && OFF < 4096, "immediate offset");
    // the base is the same in every lane; said so here, because the instruction needs it in scalar
    // registers whatever the compiler has concluded about it (no cost when it already is there)
    const unsigned long long bv = (unsigned long long)(uintptr_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bv);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bv >> 32));
    base = (const float *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    if constexpr (X4)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane_bytes), "s"(base), "n"(OFF) : "memory");
    else
        asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(lane_bytes), "s"(base), "n"(OFF) : "memory");
}
