// K2: per-window tone detection = argmax_k |DFT(x)[k]| over k in [0, N/2), plus the class
// of that bin.  Replaces detectar_frecuencia (c1_shared.py:286-299; capa1.py:54-59) and
// the decisions taken from its result (freq_to_byte, ping and EOF tests,
// c1_shared.py:331-341) through a per-bin class table.
//
// The whole spectrum is needed: in a noisy or mixed window the reference's argmax can
// land on any bin, and almost every bin maps to some byte.  The window length N factors
// into coprime n1*n2 (7055 = 83*85, 2205 = 45*49), so the N-point DFT is computed by the
// prime-factor (Good-Thomas) algorithm, with no twiddles between its two stages:
//
//   x2[a][b] = x[(n2*a + n1*b) mod N]                         a < n1, b < n2
//   Y[a][k2] = sum_b x2[a][b] * exp(-2*pi*i*b*k2/n2)          k2 <= n2/2 (real input)
//   Z[k1][k2] = sum_a exp(-2*pi*i*a*k1/n1) * Y[a][k2]
//   X[k] = Z[k mod n1][k mod n2]; bins with (k mod n2) > n2/2 are mirrors of these.
//
// Both stages are small dense contractions and run on the matrix cores in f32
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation), 3.6 MFLOP per window
// instead of the 99.5 MFLOP of a dense N x N/2 DFT.  The stage-1 result tile is already
// laid out as the B operand of stage 2 (rows on register index and lane group), so Y
// never leaves the registers.
//
// One wave owns one window at a time; a workgroup is four independent waves (one per
// SIMD, the whole 512-entry register file each).  The window is staged in LDS because
// the prime-factor index map is a gather; the next window is fetched from HBM into
// registers while stage 2 runs.  Bound: matrix cores (f32 MFMA, 64 FLOP/clk/SIMD).
//
// Decisions must equal the reference's float64 FFT.  Windows whose two largest bins are
// closer than a relative band (default 5e-5, ~10x the f32 error) are re-evaluated for
// those bins in float64 by direct summation with exactly reduced phases; if a third bin
// is in the band too, the whole spectrum is.
#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "wn_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define FLAG_TOP2 (1 << 30)
#define FLAG_FULL (1 << 29)
#define FLAG_MASK (FLAG_TOP2 | FLAG_FULL)

template <int N1, int N2>
struct Plan {
    static constexpr int N = N1 * N2;
    static constexpr int T1 = (N1 + 15) / 16;     // 16-row tiles over n1 (stage 1) / k1 (stage 2)
    static constexpr int KS1 = (N2 + 3) / 4;      // stage-1 k-steps (4 values of b each)
    static constexpr int K2 = N2 / 2 + 1;         // kept output columns k2
    static constexpr int C2T = (K2 + 15) / 16;    // 16-column tiles over k2
    static constexpr int KS2 = (N1 + 3) / 4;      // stage-2 k-steps (4 values of a each)
    static constexpr int NV4 = ((N + 3 + 3) / 4 + 63) / 64;  // float4 loads per lane per window
    static constexpr int XW = NV4 * 64 * 4;       // floats of LDS per wave
    static constexpr int TW2 = (N2 + 1) & ~1;     // float2 entries, 16-byte multiple
    static constexpr int EP = T1 * 4 * 64 * 4;    // uint16 entries of the output-bin table
    // LDS map (bytes): stage-2 twiddles, stage-1 twiddles, output-bin table, class table, a
    // zero word, then one window buffer per wave
    static constexpr int OFF_TW2 = N1 * 16;
    static constexpr int OFF_EP = OFF_TW2 + TW2 * 8;
    static constexpr int OFF_LUT = OFF_EP + EP * 2;
    static constexpr int OFF_ZERO = OFF_LUT + (((N / 2) * 2 + 15) & ~15);
    static constexpr int OFF_XW = OFF_ZERO + 16;
    static constexpr size_t LDS = (size_t)OFF_XW + (size_t)4 * XW * 4;
};

struct Top3 {
    float m1, m2, m3;
    int b1, b2;
};

// Keeps the three largest values seen and the bins of the first two.  Equal values are not
// ordered by bin here: an exact tie leaves m2 == m1, which sends the window to the float64
// pass, and that pass breaks ties towards the lower bin as np.argmax does.
__device__ __forceinline__ void top3_insert(Top3 &t, float m, int b) {
    const bool c1 = m > t.m1, c2 = m > t.m2;
    // selects, never branches: the conditions are data-dependent per lane
    const int b2a = __builtin_unpredictable(c2) ? b : t.b2;
    t.b2 = __builtin_unpredictable(c1) ? t.b1 : b2a;
    t.m3 = fmaxf(t.m3, fminf(m, t.m2));
    t.m2 = __builtin_amdgcn_fmed3f(t.m1, t.m2, m);
    t.b1 = __builtin_unpredictable(c1) ? b : t.b1;
    t.m1 = fmaxf(t.m1, m);
}

template <int N1, int N2>
__global__ __launch_bounds__(256, 1) void detect_kernel(const float *__restrict__ audio, long long audio_len,
                                                        long long n_streams, long long stream_stride,
                                                        long long first_sample, long long n_windows,
                                                        const float4 *__restrict__ g_tw1,
                                                        const float2 *__restrict__ g_tw2,
                                                        const uint16_t *__restrict__ lut, int e1, int e2,
                                                        float keep, uint16_t *__restrict__ out_cls,
                                                        float *__restrict__ out_mag, int32_t *__restrict__ out_bin,
                                                        int dbg) {
    typedef Plan<N1, N2> P;
    constexpr int N = P::N;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *smf = (float *)smem;                        // everything below is indexed from here
    float4 *tw1 = (float4 *)smem;                      // n1 x (cos, cos-sin, cos+sin, sin)(2*pi*m/n1)
    float2 *tw2 = (float2 *)(smem + P::OFF_TW2);       // n2 x (cos, -sin)(2*pi*m/n2)
    uint16_t *ep = (uint16_t *)(smem + P::OFF_EP);     // [T1][4][64][4]: folded bin of an output, 0xFFFF = unused
    uint16_t *lut_s = (uint16_t *)(smem + P::OFF_LUT); // class word per bin
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < N1; i += 256) tw1[i] = g_tw1[i];
    for (int i = threadIdx.x; i < N2; i += 256) tw2[i] = g_tw2[i];
    for (int i = threadIdx.x; i < N / 2; i += 256) lut_s[i] = lut[i];
    if (threadIdx.x < 4) smf[P::OFF_ZERO / 4 + threadIdx.x] = 0.f;
    // Output element (tile t, register r, column tile c) of lane l is Z[k1][k2] with
    // k1 = 16t + 4(l>>4) + r, k2 = 16c + (l&15); its bin is (k1*e1 + k2*e2) mod N folded to
    // [0, N/2].  Bins the reference never looks at, padding rows/columns and the mirrored
    // half of column k2 = 0 are marked unused.
    for (int i = threadIdx.x; i < P::EP; i += 256) {
        const int c = i & 3, l = (i >> 2) & 63, r = (i >> 8) & 3, t = i >> 10;
        const int k1 = 16 * t + 4 * (l >> 4) + r, k2 = 16 * c + (l & 15);
        const int kb = (int)(((long long)k1 * e1 + (long long)k2 * e2) % N);
        const int kk = min(kb, N - kb);
        const bool ok = (c < P::C2T) & (k1 < N1) & (k2 < P::K2) & (kk < N / 2) & ((k2 != 0) | (k1 <= N1 / 2));
        ep[i] = ok ? (uint16_t)kk : (uint16_t)0xFFFF;
    }
    __syncthreads();

    const int li = lane & 15;  // A row / B column / D column
    const int lg = lane >> 4;  // A,B k index / D row group
    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    long long w = (long long)blockIdx.x * 4 + wave;
    const int xw_f = P::OFF_XW / 4 + wave * P::XW;     // float index of this wave's window buffer

    // Window staging by LDS-DMA: 16 bytes per lane, 1 KiB per wave-instruction, written
    // to the wave's own LDS buffer at lds_base + 1024*j + 16*lane.  The loads are issued
    // from inline asm so that the compiler does not order the table reads of stage 2
    // behind them; the wait is the explicit vmcnt(0) at the top of the window loop.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(smf + xw_f);
    auto fetch = [&](long long win) {
        const long long s = win / n_windows;
        const long long wi = win - s * n_windows;
        const long long g0 = s * stream_stride + first_sample + wi * (long long)N;
        const long long a0 = g0 & ~3ll;
        if (a0 + 256ll * P::NV4 <= audio_len) {
            // whole span inside the buffer: one address per 4 KiB, immediate offsets (which the
            // hardware adds to the global and to the LDS address alike) for the 1-KiB pieces
            const float *src = audio + a0 + 4 * lane;
#pragma unroll
            for (int j = 0; j < P::NV4; j += 4) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 1024u * j);
                unsigned keep_m0;
                if (j + 4 <= P::NV4) {
                    asm volatile(
                        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                        "global_load_lds_dwordx4 %1, off\n\t"
                        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                        "s_mov_b32 m0, %0"
                        : "=&s"(keep_m0)
                        : "v"(src + 256 * j), "s"(dst)
                        : "memory");
                } else {
#pragma unroll
                    for (int u = 0; u < P::NV4 - j; ++u) {
                        const unsigned d2 = __builtin_amdgcn_readfirstlane(lds_base + 1024u * (j + u));
                        asm volatile(
                            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                            "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                            : "=&s"(keep_m0)
                            : "v"(src + 256 * (j + u)), "s"(d2)
                            : "memory");
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < P::NV4; ++j) {
                long long e = a0 + 4 * (j * 64 + lane);
                e = e + 4 <= audio_len ? e : audio_len - 4;  // pieces past the window are never read
                const float *src = audio + e;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 1024u * j);
                unsigned keep_m0;
                asm volatile(
                    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                    : "=&s"(keep_m0)
                    : "v"(src), "s"(dst)
                    : "memory");
            }
        }
        return (int)(g0 - a0);
    };

    // per-lane starting indices of stage 1 (the same for every window)
    int idxA0[P::T1];   // LDS index of x2[a(tile, li)][lg]; rows a >= n1 read real samples, their
                        // products are dropped in stage 2 by a zero twiddle
    int m20[P::C2T], st2[P::C2T];  // (b*k2) mod n2 at b = lg, and its step per k-step
#pragma unroll
    for (int t = 0; t < P::T1; ++t) idxA0[t] = (N2 * (16 * t + 4 * (li & 3) + (li >> 2)) + N1 * lg) % N;
#pragma unroll
    for (int c = 0; c < P::C2T; ++c) {
        int k2 = 16 * c + li;
        k2 = k2 < P::K2 ? k2 : 0;
        m20[c] = (lg * k2) % N2;
        st2[c] = (4 * k2) % N2;
    }
    constexpr int STEP_A = (4 * N1) % N;

    // results of a window are stored one window later, just ahead of the next DMA, so that
    // the vmcnt(0) at the top of the loop never waits for a store it has just issued
    bool pend = false;
    long long pend_w = 0;
    int pend_bin = 0;
    unsigned pend_cls = 0;
    float pend_mag = 0.f;
    auto flush = [&]() {
        if (pend && lane == 0) {
            out_bin[pend_w] = pend_bin;
            out_cls[pend_w] = (uint16_t)pend_cls;
            out_mag[pend_w] = pend_mag;
        }
    };

    int next_off = 0;
    if (w < total) next_off = fetch(w);
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();

    while (w < total) {
        const int xo = xw_f + next_off;   // float index of sample 0 of this window
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this window's DMA has landed

        // ------------------------------------------------------------------ stage 1
        floatx4 Y[P::C2T][2][P::T1];
        int idxA[P::T1], m2[P::C2T];
#pragma unroll
        for (int t = 0; t < P::T1; ++t) idxA[t] = idxA0[t];
#pragma unroll
        for (int c = 0; c < P::C2T; ++c) m2[c] = m20[c];
        // LDS operands of one k-step (4 values of b): the samples of every row tile and the
        // twiddles of every column tile, double-buffered in registers and read one k-step
        // ahead of their MFMAs; the reads are pinned ahead of the MFMAs by sched_barrier.
        int pa[P::T1], pt[P::C2T];
        // address of row tile t / column tile c for k-step ks, then advance the index to ks + 1.
        // Columns b >= n2 (only in the last k-step) read a zero word; the choice is made with
        // bit masks so that it stays straight-line code.
        auto addr_a = [&](int t, int ks) {
            const int keep_m = -(int)(4 * ks + lg < N2);
            pa[t] = ((xo + idxA[t]) & keep_m) | ((int)(P::OFF_ZERO / 4) & ~keep_m);
            const int nx = idxA[t] + STEP_A;
            idxA[t] = nx >= N ? nx - N : nx;
        };
        auto addr_t = [&](int c) {
            pt[c] = m2[c];
            const int nx = m2[c] + st2[c];
            m2[c] = nx >= N2 ? nx - N2 : nx;
        };
        float aA[P::T1], aB[P::T1];
        float2 tA[P::C2T], tB[P::C2T];
#define S1_ISSUE(A_, T_)                                                       \
    do {                                                                       \
        _Pragma("unroll") for (int t = 0; t < P::T1; ++t) A_[t] = smf[pa[t]];  \
        _Pragma("unroll") for (int c = 0; c < P::C2T; ++c) T_[c] = tw2[pt[c]]; \
        __builtin_amdgcn_sched_barrier(0);                                     \
    } while (0)
        // One k-step: 2 MFMAs per (column tile, row tile) slot; each slot also carries a piece of
        // the address arithmetic of k-step KS_NEXT_, pinned there so that it issues in the MFMA
        // shadow instead of after the last MFMA.
#define S1_MFMA(A_, T_, FIRST_, KS_NEXT_)                                                                     \
    do {                                                                                                      \
        _Pragma("unroll") for (int c = 0; c < P::C2T; ++c) {                                                  \
            _Pragma("unroll") for (int t = 0; t < P::T1; ++t) {                                               \
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};                                                       \
                Y[c][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[t], T_[c].x, FIRST_ ? z : Y[c][0][t], 0, 0, 0); \
                Y[c][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[t], T_[c].y, FIRST_ ? z : Y[c][1][t], 0, 0, 0); \
                if (c == 0) addr_a(t, KS_NEXT_);                                                              \
                if (c == P::C2T - 1 && t < P::C2T) addr_t(t);                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#pragma unroll
        for (int t = 0; t < P::T1; ++t) addr_a(t, 0);
#pragma unroll
        for (int c = 0; c < P::C2T; ++c) addr_t(c);
        S1_ISSUE(aA, tA);                 // k-step 0
#pragma unroll
        for (int t = 0; t < P::T1; ++t) addr_a(t, 1);
#pragma unroll
        for (int c = 0; c < P::C2T; ++c) addr_t(c);
        S1_ISSUE(aB, tB);                 // k-step 1
        S1_MFMA(aA, tA, true, 2);         // k-step 0 starts the accumulators from a literal zero
        const int ks_end = (dbg & 1) ? 2 : P::KS1;
#pragma unroll 1
        for (int ks = 1; ks + 1 < ks_end; ks += 2) {
            S1_ISSUE(aA, tA);             // k-step ks + 1
            S1_MFMA(aB, tB, false, ks + 2);   // k-step ks
            S1_ISSUE(aB, tB);             // k-step ks + 2
            S1_MFMA(aA, tA, false, ks + 3);   // k-step ks + 1
        }
        if ((P::KS1 - 1) % 2 == 1) S1_MFMA(aB, tB, false, P::KS1 + 1);   // last k-step when their number is even
#undef S1_ISSUE
#undef S1_MFMA

        // every LDS read of this window has returned (its values fed the MFMAs above):
        // store the previous window's results, then start the next window's DMA so that it
        // lands while stage 2 runs
        const long long wn = w + wstep;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flush();
        if (wn < total && !(dbg & 4)) next_off = fetch(wn);

        // ------------------------------------------------------------------ stage 2
        // Z = W1 * (Yr + i*Yi) with three real products instead of four:
        //   P1 = C*(Yr+Yi), P2 = (C-S)*Yi, P3 = (C+S)*Yr;  Re Z = P1 - P2, Im Z = P1 - P3.
        floatx4 Ys[P::C2T][P::T1];
#pragma unroll
        for (int c = 0; c < P::C2T; ++c)
#pragma unroll
            for (int t = 0; t < P::T1; ++t) Ys[c][t] = Y[c][0][t] + Y[c][1][t];

        Top3 best;
        best.m1 = best.m2 = best.m3 = -1.f;
        best.b1 = best.b2 = 0x7fffffff;
        // results of the previous output tile: their epilogue runs under this tile's MFMAs
        float pre[P::C2T][4], pim[P::C2T][4];
#pragma unroll
        for (int c = 0; c < P::C2T; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[c][r] = pim[c][r] = 0.f;

        // The epilogue of one output tile (12 elements: |Z|^2, bin lookup, top-3 update) is cut
        // into five pieces per element and dealt out over the MFMA slots of the next tile, so
        // that it issues in the shadow of the MFMAs.  Slots and pieces are generated at compile
        // time (templated lambdas), every register index below is a constant.
        uint2 ep_pk[4];
        float ep_m[4 * P::C2T];
        int ep_k[4 * P::C2T];
        bool ep_c1[4 * P::C2T], ep_c2[4 * P::C2T];
        int ep_t = -1;          // tile whose results are in pre/pim (-1: none)
        auto piece = [&]<int Q>(std::integral_constant<int, Q>) {
            constexpr int e = Q / 5, k = Q % 5, r = e / P::C2T, c = e % P::C2T;
            if constexpr (k == 0) {
                if constexpr (c == 0)
                    ep_pk[r] = *(const uint2 *)(ep + ((((ep_t < 0 ? 0 : ep_t) * 4 + r) * 64 + lane) << 2));
            } else if constexpr (k == 1) {
                const unsigned word = c < 2 ? ep_pk[r].x : ep_pk[r].y;
                const unsigned kk = (c & 1) ? (word >> 16) : (word & 0xFFFFu);
                const bool ok = (kk != 0xFFFFu) & (ep_t >= 0);
                const float mg = pre[c][r] * pre[c][r] + pim[c][r] * pim[c][r];
                ep_m[e] = __builtin_unpredictable(ok) ? mg : -2.f;
                ep_k[e] = (int)kk;
            } else if constexpr (k == 2) {
                ep_c1[e] = ep_m[e] > best.m1;
                ep_c2[e] = ep_m[e] > best.m2;
                const int b2a = __builtin_unpredictable(ep_c2[e]) ? ep_k[e] : best.b2;
                best.b2 = __builtin_unpredictable(ep_c1[e]) ? best.b1 : b2a;
            } else if constexpr (k == 3) {
                best.m3 = fmaxf(best.m3, fminf(ep_m[e], best.m2));
                best.m2 = __builtin_amdgcn_fmed3f(best.m1, best.m2, ep_m[e]);
            } else {
                best.b1 = __builtin_unpredictable(ep_c1[e]) ? ep_k[e] : best.b1;
                best.m1 = fmaxf(best.m1, ep_m[e]);
            }
        };
        constexpr int NPIECE = 5 * 4 * P::C2T;          // pieces per tile
        constexpr int NSLOT = 3 * P::KS2;               // MFMA slots per tile (C2T MFMAs each)
        auto pieces = [&]<int LO, int... J>(std::integral_constant<int, LO>, std::integer_sequence<int, J...>) {
            (piece(std::integral_constant<int, LO + J>{}), ...);
        };

#pragma unroll 1
        for (int t = 0; t < ((dbg & 2) ? 1 : P::T1); ++t) {
            floatx4 p1[P::C2T], p2[P::C2T], p3[P::C2T];
            const int k1a = 16 * t + li;  // A row of this lane
            int m1 = (k1a * lg) % N1;
            const int st1 = (4 * k1a) % N1;
            float4 tw_next = tw1[m1], tw = tw_next;
            ep_t = t - 1;
            __builtin_amdgcn_sched_barrier(0);
            // slot = (k-step s, product part): C2T MFMAs plus its share of the pending epilogue
            auto slot = [&]<int SL>(std::integral_constant<int, SL>) {
                constexpr int s = SL / 3, part = SL % 3;
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                if constexpr (part == 0) {
                    tw = tw_next;
                    if constexpr (4 * s + 3 >= N1) {   // rows a >= n1 of Y are not data: drop them
                        const int keep_m = -(int)(4 * s + lg < N1);
                        tw.x = __int_as_float(__float_as_int(tw.x) & keep_m);
                        tw.y = __int_as_float(__float_as_int(tw.y) & keep_m);
                        tw.z = __int_as_float(__float_as_int(tw.z) & keep_m);
                    }
                }
#pragma unroll
                for (int c = 0; c < P::C2T; ++c) {
                    if constexpr (part == 0)
                        p1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw.x, Ys[c][s >> 2][s & 3], s == 0 ? z : p1[c], 0, 0, 0);
                    else if constexpr (part == 1)
                        p2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw.y, Y[c][1][s >> 2][s & 3], s == 0 ? z : p2[c], 0, 0, 0);
                    else
                        p3[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw.z, Y[c][0][s >> 2][s & 3], s == 0 ? z : p3[c], 0, 0, 0);
                }
                if constexpr (part == 0) {   // twiddles of the next k-step, read ahead of its MFMAs
                    const int nx = m1 + st1;
                    m1 = nx >= N1 ? nx - N1 : nx;
                    if constexpr (s + 1 < P::KS2) tw_next = tw1[m1];
                }
                constexpr int lo = SL * NPIECE / NSLOT, hi = (SL + 1) * NPIECE / NSLOT;
                pieces(std::integral_constant<int, lo>{}, std::make_integer_sequence<int, hi - lo>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (slot(std::integral_constant<int, I>{}), ...);
            }(std::make_integer_sequence<int, NSLOT>{});
#pragma unroll
            for (int c = 0; c < P::C2T; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pre[c][r] = p1[c][r] - p2[c][r];
                    pim[c][r] = p1[c][r] - p3[c][r];
                }
        }
        ep_t = P::T1 - 1;
        pieces(std::integral_constant<int, 0>{}, std::make_integer_sequence<int, NPIECE>{});
        // wave-wide merge of the per-lane top three
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float om1 = __shfl_xor(best.m1, d), om2 = __shfl_xor(best.m2, d), om3 = __shfl_xor(best.m3, d);
            const int ob1 = __shfl_xor(best.b1, d), ob2 = __shfl_xor(best.b2, d);
            top3_insert(best, om1, ob1);
            top3_insert(best, om2, ob2);
            best.m3 = fmaxf(best.m3, om3);
        }
        {
            const bool silent = best.m1 <= 0.f;   // every bin is zero: np.argmax gives bin 0
            const bool full = best.m3 >= best.m1 * keep, top2 = best.m2 >= best.m1 * keep;
            const int b = silent ? 0 : best.b1;
            const unsigned cls = lut_s[b];
            pend_bin = silent ? 0 : (full ? (b | FLAG_FULL) : (top2 ? (b | FLAG_TOP2) : b));
            pend_cls = silent ? cls : (full ? 0u : (top2 ? (unsigned)best.b2 : cls));
            pend_mag = sqrtf(fmaxf(best.m1, 0.f));
            pend_w = w;
            pend = true;
        }
        w = wn;
    }
    flush();
    if ((dbg & 8) && blockIdx.x == 0 && threadIdx.x == 0) {   // diagnostic: shader clock = 100 MHz * [0]/[1]
        out_bin[0] = (int)((__builtin_amdgcn_s_memtime() - dbg_t0) >> 8);
        out_bin[1] = (int)((__builtin_amdgcn_s_memrealtime() - dbg_r0) >> 8);
    }
}

// ---------------------------------------------------------------------------------------
// float64 re-evaluation of flagged windows.  X[b] = sum_n x[n] * exp(-2*pi*i*((b*n) mod N)/N)
// with the phase table built by sincospi in LDS the first time a workgroup needs it.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ __launch_bounds__(256) void refine_kernel(const float *__restrict__ audio, long long n_streams,
                                                     long long stream_stride, long long first_sample,
                                                     long long n_windows, int N,
                                                     const uint16_t *__restrict__ lut,
                                                     uint16_t *__restrict__ cls, float *__restrict__ mag,
                                                     int32_t *__restrict__ bin, unsigned long long *stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    double2 *tw = (double2 *)smem;  // N entries
    __shared__ double red[4][4];
    __shared__ double best_m[4];
    __shared__ int best_b[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long total = n_streams * n_windows;
    bool table_ready = false;
    const int nbins = N / 2;
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&stats[0], (unsigned long long)total);
    for (long long base = (long long)blockIdx.x * 256; base < total; base += (long long)gridDim.x * 256) {
        const long long idx = base + tid;
        const int fl = idx < total ? (bin[idx] & FLAG_MASK) : 0;
        if (!__syncthreads_or(fl)) continue;
        if (!table_ready) {
            for (int m = tid; m < N; m += 256) {
                double s, c;
                sincospi(2.0 * (double)m / (double)N, &s, &c);
                tw[m] = make_double2(c, s);
            }
            table_ready = true;
            __syncthreads();
        }
        const int cnt = (int)((total - base) < 256 ? (total - base) : 256);
        for (int j = 0; j < cnt; ++j) {
            const long long wi = base + j;
            const int f = bin[wi];  // same address in every thread
            if (!(f & FLAG_MASK)) continue;
            const long long s = wi / n_windows;
            const float *x = audio + s * stream_stride + first_sample + (wi - s * n_windows) * (long long)N;
            int rb;
            double rm;
            if (f & FLAG_TOP2) {
                const int b1 = f & 0xFFFF, b2 = cls[wi];
                double r1 = 0, i1 = 0, r2 = 0, i2 = 0;
                for (int n = tid; n < N; n += 256) {
                    const double xv = (double)x[n];
                    const double2 t1 = tw[(unsigned)(b1 * n) % (unsigned)N];
                    const double2 t2 = tw[(unsigned)(b2 * n) % (unsigned)N];
                    r1 += xv * t1.x; i1 += xv * t1.y;
                    r2 += xv * t2.x; i2 += xv * t2.y;
                }
                r1 = wave_sum(r1); i1 = wave_sum(i1); r2 = wave_sum(r2); i2 = wave_sum(i2);
                __syncthreads();
                if (lane == 0) { red[wave][0] = r1; red[wave][1] = i1; red[wave][2] = r2; red[wave][3] = i2; }
                __syncthreads();
                double v[4];
                for (int q = 0; q < 4; ++q) v[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
                const double ma = v[0] * v[0] + v[1] * v[1], mb = v[2] * v[2] + v[3] * v[3];
                const bool first = (ma > mb) || (ma == mb && b1 < b2);
                rb = first ? b1 : b2;
                rm = first ? ma : mb;
                if (tid == 0) atomicAdd(&stats[1], 1ull);
            } else {
                double bm = -1.0;
                int bb = 0x7fffffff;
                for (int b = tid; b < nbins; b += 256) {
                    double re = 0, im = 0;
                    int m = 0;
                    for (int n = 0; n < N; ++n) {
                        const double xv = (double)x[n];
                        const double2 t = tw[m];
                        re += xv * t.x; im += xv * t.y;
                        m += b;
                        m = m >= N ? m - N : m;
                    }
                    const double mg = re * re + im * im;
                    if (mg > bm) { bm = mg; bb = b; }  // b ascends: first maximum is kept
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const double om = __shfl_xor(bm, d);
                    const int ob = __shfl_xor(bb, d);
                    if (om > bm || (om == bm && ob < bb)) { bm = om; bb = ob; }
                }
                __syncthreads();
                if (lane == 0) { best_m[wave] = bm; best_b[wave] = bb; }
                __syncthreads();
                bm = best_m[0]; bb = best_b[0];
                for (int q = 1; q < 4; ++q)
                    if (best_m[q] > bm || (best_m[q] == bm && best_b[q] < bb)) { bm = best_m[q]; bb = best_b[q]; }
                rb = bb;
                rm = bm;
                if (tid == 0) atomicAdd(&stats[2], 1ull);
            }
            if (tid == 0) {
                bin[wi] = rb;
                cls[wi] = lut[rb];
                mag[wi] = (float)sqrt(rm);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
template <int N1, int N2>
static int launch_detect(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                         int64_t stream_stride, int64_t first_sample, int64_t n_windows, uint16_t *d_cls,
                         float *d_mag, int32_t *d_bin, hipStream_t st) {
    typedef Plan<N1, N2> P;
    static bool attr = false;
    if (!attr) {
        WN_HIP_TRY(hipFuncSetAttribute((const void *)detect_kernel<N1, N2>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::LDS));
        attr = true;
    }
    const long long total = n_streams * n_windows;
    long long blocks = (total + 3) / 4;
    if (blocks > ctx->num_cus) blocks = ctx->num_cus;
    const double k = (1.0 - ctx->tie_band) * (1.0 - ctx->tie_band);
    hipLaunchKernelGGL((detect_kernel<N1, N2>), dim3((unsigned)blocks), dim3(256), P::LDS, st, d_audio,
                       (long long)audio_len, (long long)n_streams, (long long)stream_stride,
                       (long long)first_sample, (long long)n_windows, ctx->d_tw1, ctx->d_tw2, ctx->d_cls_lut,
                       ctx->e1, ctx->e2, (float)k, d_cls, d_mag, d_bin,
                       getenv("WN_DETECT_DEBUG") ? atoi(getenv("WN_DETECT_DEBUG")) : 0);
    return wn_launch_check("detect_kernel");
}

int wn_detect_prepare(wn_ctx *ctx) {
    // the float64 phase table of the re-evaluation kernel lives in dynamic LDS
    const size_t lds = (size_t)ctx->n * sizeof(double2);
    if (lds > 150 * 1024) {
        wn_set_error("window length %d too large for the float64 phase table", ctx->n);
        return WN_ERR_UNSUPPORTED;
    }
    WN_HIP_TRY(hipFuncSetAttribute((const void *)refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    return WN_OK;
}

extern "C" int wn_detect_windows(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                                 int64_t stream_stride, int64_t first_sample, int64_t n_windows, uint16_t *d_cls,
                                 float *d_peak_mag, int32_t *d_peak_bin, void *stream) {
    WN_REQUIRE(ctx && n_streams >= 0 && n_windows >= 0 && first_sample >= 0 && stream_stride >= 0, "ctx / sizes");
    if (n_streams == 0 || n_windows == 0) return WN_OK;
    WN_REQUIRE(d_audio && d_cls && d_peak_mag && d_peak_bin, "null device pointer");
    WN_REQUIRE(((uintptr_t)d_audio % 16) == 0 && audio_len % 4 == 0, "d_audio 16-byte aligned, audio_len % 4 == 0");
    const int64_t span = first_sample + n_windows * (int64_t)ctx->n;
    WN_REQUIRE(n_streams == 1 || span <= stream_stride, "windows of a stream overlap the next stream");
    WN_REQUIRE((n_streams - 1) * stream_stride + span <= audio_len, "windows reach past audio_len");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (ctx->n1 == 83 && ctx->n2 == 85)
        rc = launch_detect<83, 85>(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls,
                                   d_peak_mag, d_peak_bin, st);
    else if (ctx->n1 == 45 && ctx->n2 == 49)
        rc = launch_detect<45, 49>(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls,
                                   d_peak_mag, d_peak_bin, st);
    else {
        wn_set_error("no detection kernel for %d x %d", ctx->n1, ctx->n2);
        return WN_ERR_UNSUPPORTED;
    }
    if (rc != WN_OK) return rc;
    const long long total = n_streams * n_windows;
    long long blocks = (total + 255) / 256;
    const long long cap = (long long)ctx->num_cus * 4;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)blocks), dim3(256), (size_t)ctx->n * sizeof(double2), st,
                       d_audio, (long long)n_streams, (long long)stream_stride, (long long)first_sample,
                       (long long)n_windows, ctx->n, ctx->d_cls_lut, d_cls, d_peak_mag, d_peak_bin, ctx->d_stats);
    return wn_launch_check("refine_kernel");
}

// ---------------------------------------------------------------------------------------
// Lane-layout self test: D = A*B for A[i][k] = 1 + i + 16k, B[k][j] = 1 + 3j + 100k
// through one v_mfma_f32_16x16x4_f32; lane l writes its four D registers to out[4l..].
// ---------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(float *out) {
    const int l = threadIdx.x;
    const float a = 1.0f + (float)(l & 15) + 16.0f * (float)(l >> 4);
    const float b = 1.0f + 3.0f * (float)(l & 15) + 100.0f * (float)(l >> 4);
    floatx4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[4 * l + r] = d[r];
}

extern "C" int wn_selftest_mfma(float *d_out256, void *stream) {
    WN_REQUIRE(d_out256, "d_out256");
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out256);
    return wn_launch_check("mfma_selftest_kernel");
}
