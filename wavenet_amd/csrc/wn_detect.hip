// K2: per-window tone detection = argmax_k |DFT(x)[k]| over k in [0, N/2), plus the class
// of that bin.  Replaces detectar_frecuencia (c1_shared.py:286-299; capa1.py:54-59) and
// the decisions taken from its result (freq_to_byte, ping and EOF tests,
// c1_shared.py:331-341) through a per-bin class table.
//
// The whole spectrum is needed: in a noisy or mixed window the reference's argmax can
// land on any bin, and almost every bin maps to some byte.  The window length N factors
// into coprime odd n1*n2 (7055 = 83*85, 2205 = 45*49), so the N-point DFT is computed by
// the prime-factor (Good-Thomas) algorithm, with no twiddles between its two stages:
//
//   x2[a][b] = x[(n2*a + n1*b) mod N]                         a < n1, b < n2
//   Y[a][k2] = sum_b x2[a][b] * exp(-2*pi*i*b*k2/n2)          k2 <= h2 = n2/2 (real input)
//   Z[k1][k2] = sum_a exp(-2*pi*i*a*k1/n1) * Y[a][k2]
//   X[k] = Z[k mod n1][k mod n2]; bins with (k mod n2) > h2 are mirrors of these.
//
// Both stages fold their summation index in mirrored pairs (cos is even, sin is odd in the
// index), which halves the multiplications, and stage 2 also produces rows k1 and n1-k1
// from the same four real sums:
//
//   stage 1, b' in [0, h2]:  Yr = sum cos(b'k2) (x2[a][b'] + x2[a][n2-b']),
//                            Yi = -sum sin(b'k2) (x2[a][b'] - x2[a][n2-b'])
//   stage 2, a' in [0, h1], k1 in [0, h1], with S = Y[a'] + Y[n1-a'], D = Y[a'] - Y[n1-a']:
//       A = sum cos(a'k1) Sr, B = sum sin(a'k1) Di, C = sum cos(a'k1) Si, E = sum sin(a'k1) Dr
//       Z[k1] = (A + B) + i(C - E),   Z[n1-k1] = (A - B) + i(C + E)
//
// All sums are small dense contractions and run on the matrix cores in f32
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation): 792 instructions of
// 2048 FLOP per 7055-sample window (1.2 MFLOP useful) instead of the 99.5 MFLOP of a dense
// N x N/2 DFT.  A stage-1 result tile is already laid out as the B operand of stage 2
// (rows on register index and lane group); the mirrored rows n1-a' are computed as separate
// row tiles with the same lane/register positions, so S and D are plain register adds and
// Y never leaves the registers.
//
// One wave owns one window at a time; a workgroup is four independent waves (one per
// SIMD, the whole 512-entry register file each).  The window is staged in LDS by DMA
// because the prime-factor index map is a gather; the next window is fetched while stage 2
// runs.  With one wave per SIMD every instruction costs an issue slot, so the address
// arithmetic, LDS reads and the arg-max epilogue are dealt out in fixed shares between the
// MFMAs (compile-time generated slots) to issue in their shadow.
// Bound: matrix cores (f32 MFMA, 64 FLOP/clk/SIMD).
//
// Decisions must equal the reference's float64 FFT.  The per-lane running top three are
// kept as integer keys (|Z|^2 bits with the low 12 bits replaced by the bin), so windows
// whose largest bins are closer than the relative band (default 5e-5, ~10x the f32 error)
// plus the 2^-12 the keys drop are re-evaluated in float64: the two candidate bins by
// direct summation with exactly reduced phases; if a third bin is in the band too, the
// whole spectrum.
#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "wn_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define FLAG_TOP2 (1 << 30)
#define FLAG_FULL (1 << 29)
#define FLAG_MASK (FLAG_TOP2 | FLAG_FULL)
#define KEY_BIN 0xFFFu   // low bits of a key hold the bin

template <int I>
using ic = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
// f(ic<0>), f(ic<1>), ... f(ic<N-1>): a loop whose index is a constant in every iteration
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int N1, int N2>
struct Plan {
    static_assert(N1 % 2 == 1 && N2 % 2 == 1, "mirrored-pair folding needs odd factors");
    static constexpr int N = N1 * N2;
    static constexpr int H1 = N1 / 2;             // a' and k1 run over [0, H1]
    static constexpr int H2 = N2 / 2;             // b' and k2 run over [0, H2]
    static constexpr int TA = (H1 + 1 + 15) / 16; // 16-row tiles over a' (stage 1) / k1 (stage 2)
    static constexpr int T1 = 2 * TA;             // stage-1 row tiles: rows a', then rows n1-a'
    static constexpr int KS1 = (H2 + 1 + 3) / 4;  // stage-1 k-steps (4 values of b' each)
    static constexpr int K2 = H2 + 1;             // kept output columns k2
    static constexpr int C2T = (K2 + 15) / 16;    // 16-column tiles over k2
    static constexpr int KS2 = (H1 + 1 + 3) / 4;  // stage-2 k-steps (4 values of a' each)
    static_assert(16 * TA <= N1 && KS2 <= 4 * TA && C2T >= 2 && C2T <= 4 && C2T <= T1 && N / 2 <= (int)KEY_BIN, "plan limits");
    static constexpr int NV4 = ((N + 3 + 3) / 4 + 63) / 64;  // float4 loads per lane per window
    static constexpr int XW = NV4 * 64 * 4;       // floats of LDS per wave
    static constexpr int TW1 = (N1 + 1) & ~1;     // float2 entries, 16-byte multiple
    static constexpr int TW2 = (N2 + 1) & ~1;
    static constexpr int EP = TA * 4 * 64 * 8;    // uint16 entries of the output-bin table
    // LDS map (bytes): stage-2 twiddles, stage-1 twiddles, output-bin table, class table, a
    // zero word, then one window buffer per wave
    static constexpr int OFF_TW2 = TW1 * 8;
    static constexpr int OFF_EP = OFF_TW2 + TW2 * 8;
    static constexpr int OFF_LUT = OFF_EP + EP * 2;
    static constexpr int OFF_ZERO = OFF_LUT + (((N / 2) * 2 + 15) & ~15);
    static constexpr int OFF_XW = OFF_ZERO + 16;
    static constexpr size_t LDS = (size_t)OFF_XW + (size_t)4 * XW * 4;
};

// The three largest keys seen.  A key is the bit pattern of a non-negative float with its
// low 12 bits replaced by the bin: keys order like the values (to 2^-11), ties by bin.
struct Top3 {
    unsigned k1, k2, k3;
    float mx;   // exact largest value
};
__device__ __forceinline__ void top3_insert(Top3 &t, unsigned key) {
    t.k3 = max(t.k3, min(key, t.k2));
    t.k2 = max(min(t.k1, t.k2), min(max(t.k1, t.k2), key));   // median of three
    t.k1 = max(t.k1, key);
}

template <int N1, int N2>
__global__ __launch_bounds__(256, 1) void detect_kernel(const float *__restrict__ audio, long long audio_len,
                                                        long long n_streams, long long stream_stride,
                                                        long long first_sample, long long n_windows,
                                                        const float2 *__restrict__ g_tw1,
                                                        const float2 *__restrict__ g_tw2,
                                                        const uint16_t *__restrict__ lut, int e1, int e2,
                                                        float keep, uint16_t *__restrict__ out_cls,
                                                        float *__restrict__ out_mag, int32_t *__restrict__ out_bin,
                                                        int dbg) {
    typedef Plan<N1, N2> P;
    constexpr int N = P::N, TA = P::TA, T1 = P::T1, C2T = P::C2T, KS1 = P::KS1, KS2 = P::KS2;
    constexpr int H1 = P::H1, H2 = P::H2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *smf = (float *)smem;                        // everything below is indexed from here
    float2 *tw1 = (float2 *)smem;                      // n1 x (cos, sin)(2*pi*m/n1)
    float2 *tw2 = (float2 *)(smem + P::OFF_TW2);       // n2 x (cos, -sin)(2*pi*m/n2)
    uint16_t *ep = (uint16_t *)(smem + P::OFF_EP);     // [TA][4][64][8]: folded bin of an output, 0xFFFF = unused
    uint16_t *lut_s = (uint16_t *)(smem + P::OFF_LUT); // class word per bin
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < N1; i += 256) tw1[i] = g_tw1[i];
    for (int i = threadIdx.x; i < N2; i += 256) tw2[i] = g_tw2[i];
    for (int i = threadIdx.x; i < N / 2; i += 256) lut_s[i] = lut[i];
    if (threadIdx.x < 4) smf[P::OFF_ZERO / 4 + threadIdx.x] = 0.f;
    // Accumulator element (output tile t, register r, column tile c) of lane l holds the sums
    // for k1 = 16t + 4(l>>4) + r, k2 = 16c + (l&15); from them come Z[k1][k2] (entry 2c) and
    // Z[n1-k1][k2] (entry 2c+1).  The bin of Z[j][k2] is (j*e1 + k2*e2) mod N folded to
    // [0, N/2].  Padding rows and columns, bins the reference never looks at, and the second
    // copy of self-mirrored outputs (k1 = 0 or k2 = 0) are marked unused.
    for (int i = threadIdx.x; i < P::EP; i += 256) {
        const int sg = i & 1, c = (i >> 1) & 3, l = (i >> 3) & 63, r = (i >> 9) & 3, t = i >> 11;
        const int k1 = 16 * t + 4 * (l >> 4) + r, k2 = 16 * c + (l & 15);
        const int j = sg ? (N1 - k1) % N1 : k1;
        const int kb = (int)(((long long)j * e1 + (long long)k2 * e2) % N);
        const int kk = min(kb, N - kb);
        const bool ok = (c < C2T) & (k1 <= H1) & (k2 <= H2) & (kk < N / 2) & (!sg | ((k1 != 0) & (k2 != 0)));
        ep[i] = ok ? (uint16_t)kk : (uint16_t)0xFFFF;
    }
    __syncthreads();

    const int li = lane & 15;  // A row / B column / D column
    const int lg = lane >> 4;  // A,B k index / D row group
    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    long long w = (long long)blockIdx.x * 4 + wave;
    const int xw_f = P::OFF_XW / 4 + wave * P::XW;     // float index of this wave's window buffer
    constexpr int ZERO_F = P::OFF_ZERO / 4;

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

    // per-lane starting indices of stage 1 (the same for every window).  A-operand row li of
    // row tile t is a' = 16(t mod TA) + 4(li&3) + (li>>2), i.e. result register r of lane group
    // g holds a' = 16t + 4r + g, so that stage-2 k-step s covers a' = 4s .. 4s+3.  Tiles
    // t >= TA hold row n1-a' at the position of a' (row 0 again for a' = 0).  Rows a' > H1
    // are real samples too; their products are dropped in stage 2 by a zero twiddle.
    int ipos0[T1], ineg0[T1];      // index of x2[a][b'] and x2[a][n2-b'] at b' = lg
    int m20[C2T], st2[C2T];        // (b'*k2) mod n2 at b' = lg, and its step per k-step
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        const int ap = 16 * (t % TA) + 4 * (li & 3) + (li >> 2);
        const int a = t < TA ? ap : (N1 - ap) % N1;
        ipos0[t] = (N2 * a + N1 * lg) % N;
        ineg0[t] = (N2 * a + N - N1 * lg) % N;
    }
#pragma unroll
    for (int c = 0; c < C2T; ++c) {
        int k2 = 16 * c + li;
        k2 = k2 <= H2 ? k2 : 0;
        m20[c] = (lg * k2) % N2;
        st2[c] = (4 * k2) % N2;
    }
    constexpr int STEP_A = (4 * N1) % N;
    const float c_half = lg == 0 ? 0.5f : 1.0f;   // S of a' = 0 is 2*Y[0]

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
    const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};

    while (w < total) {
        const int xo = xw_f + next_off;   // float index of sample 0 of this window
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this window's DMA has landed

        // ------------------------------------------------------------------ stage 1
        floatx4 Yr[C2T][T1], Yi[C2T][T1];
        int ipos[T1], ineg[T1], m2[C2T];
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            ipos[t] = ipos0[t];
            ineg[t] = ineg0[t];
        }
#pragma unroll
        for (int c = 0; c < C2T; ++c) m2[c] = m20[c];
        // LDS operands are read two k-steps ahead of their MFMAs (raw samples), folded one
        // k-step ahead (xs = x[b'] + x[n2-b'], xd = x[b'] - x[n2-b']); twiddles one ahead.
        float raw_p[T1], raw_m[T1], xs[2][T1], xd[2][T1];
        float2 tw[2][C2T];
        auto rd = [&](auto KS, auto T) {   // issue the sample reads of row tile T for k-step KS
            constexpr int ks = decltype(KS)::value, t = decltype(T)::value;
            int ap = xo + ipos[t], an = xo + ineg[t];
            if constexpr (ks == 0) an = __builtin_unpredictable(lg == 0) ? ZERO_F : an;   // b' = 0 pairs with itself
            if constexpr (4 * ks + 3 > H2) {                                               // b' > h2: no such column
                const bool over = 4 * ks + lg > H2;
                ap = __builtin_unpredictable(over) ? ZERO_F : ap;
                an = __builtin_unpredictable(over) ? ZERO_F : an;
            }
            raw_p[t] = smf[ap];
            raw_m[t] = smf[an];
        };
        auto adv = [&](auto T) {           // advance the sample indices of row tile T by one k-step
            constexpr int t = decltype(T)::value;
            const int np = ipos[t] + STEP_A, nn = ineg[t] - STEP_A;
            ipos[t] = np >= N ? np - N : np;
            ineg[t] = nn < 0 ? nn + N : nn;
        };
        auto fold = [&](auto B, auto T) {  // raw samples of row tile T -> operands of buffer B
            constexpr int b = decltype(B)::value, t = decltype(T)::value;
            xs[b][t] = raw_p[t] + raw_m[t];
            xd[b][t] = raw_p[t] - raw_m[t];
        };
        auto rdtw = [&](auto B, auto C) {  // twiddles of column tile C into buffer B, advance
            constexpr int b = decltype(B)::value, c = decltype(C)::value;
            tw[b][c] = tw2[m2[c]];
            const int nx = m2[c] + st2[c];
            m2[c] = nx >= N2 ? nx - N2 : nx;
        };
        static_for<T1>([&](auto T) { rd(ic<0>{}, T); adv(T); });
        static_for<C2T>([&](auto C) { rdtw(ic<0>{}, C); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<T1>([&](auto T) { fold(ic<0>{}, T); });
        static_for<T1>([&](auto T) { rd(ic<1>{}, T); adv(T); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<KS1>([&](auto KS) {
            constexpr int ks = decltype(KS)::value, cur = ks & 1, nxt = cur ^ 1;
            static_for<C2T>([&](auto C) {
                static_for<T1>([&](auto T) {
                    constexpr int c = decltype(C)::value, t = decltype(T)::value;
                    Yr[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[cur][t], tw[cur][c].x, ks == 0 ? z4 : Yr[c][t], 0, 0, 0);
                    Yi[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd[cur][t], tw[cur][c].y, ks == 0 ? z4 : Yi[c][t], 0, 0, 0);
                    if constexpr (ks + 1 < KS1) {
                        if constexpr (c == 0) {
                            fold(ic<nxt>{}, T);
                            if constexpr (ks + 2 < KS1) rd(ic<ks + 2>{}, T);
                        }
                        if constexpr (c == 1) {
                            if constexpr (ks + 2 < KS1) adv(T);
                            if constexpr (t < C2T) rdtw(ic<nxt>{}, T);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });

        // every LDS read of this window has returned (its values fed the MFMAs above):
        // store the previous window's results, then start the next window's DMA so that it
        // lands while stage 2 runs
        const long long wn = w + wstep;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flush();
        if (wn < total && !(dbg & 4)) next_off = fetch(wn);

        // ------------------------------------------------------------------ stage 2
        // rows a' and n1-a' -> sum and difference, in place: tiles [0, TA) hold S, [TA, 2TA) hold D
#pragma unroll
        for (int c = 0; c < C2T; ++c)
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const floatx4 ar = Yr[c][t], br = Yr[c][t + TA], ai = Yi[c][t], bi = Yi[c][t + TA];
                Yr[c][t] = ar + br;
                Yr[c][t + TA] = ar - br;
                Yi[c][t] = ai + bi;
                Yi[c][t + TA] = ai - bi;
            }

        Top3 best;
        best.k1 = best.k2 = best.k3 = 0u;
        best.mx = 0.f;
        // The four sums of an output tile; two sets, so that the epilogue of tile t (bins, |Z|^2,
        // top-three update: 6 pieces per accumulator element) runs in the MFMA slots of tile t+1.
        floatx4 acc[2][4][C2T];
        uint4 ep_pk[4];
        float e_rp, e_ip, e_rm, e_im, e_mp, e_mm;
        unsigned e_kp, e_km;
        auto piece = [&](auto TT, auto Q) {
            constexpr int tt = decltype(TT)::value, q = decltype(Q)::value;
            constexpr int e = q / 6, k = q % 6, r = e / C2T, c = e % C2T, set = tt & 1;
            if constexpr (k == 0) {
                if constexpr (c == 0) ep_pk[r] = *(const uint4 *)(ep + (((tt * 4 + r) * 64 + lane) << 3));
                const float A = acc[set][0][c][r], B = acc[set][1][c][r], C = acc[set][2][c][r], E = acc[set][3][c][r];
                e_rp = A + B;
                e_ip = C - E;
                e_rm = A - B;
                e_im = C + E;
            } else if constexpr (k == 1) {
                e_mp = e_rp * e_rp + e_ip * e_ip;
                e_mm = e_rm * e_rm + e_im * e_im;
            } else if constexpr (k == 2) {
                const unsigned word = c == 0 ? ep_pk[r].x : (c == 1 ? ep_pk[r].y : (c == 2 ? ep_pk[r].z : ep_pk[r].w));
                const unsigned kk = word & 0xFFFFu;
                e_mp = __builtin_unpredictable(kk != 0xFFFFu) ? e_mp : 0.f;
                e_kp = __builtin_unpredictable(kk != 0xFFFFu) ? ((__float_as_uint(e_mp) & ~KEY_BIN) | kk) : 0u;
            } else if constexpr (k == 3) {
                const unsigned word = c == 0 ? ep_pk[r].x : (c == 1 ? ep_pk[r].y : (c == 2 ? ep_pk[r].z : ep_pk[r].w));
                const unsigned kk = word >> 16;
                e_mm = __builtin_unpredictable(kk != 0xFFFFu) ? e_mm : 0.f;
                e_km = __builtin_unpredictable(kk != 0xFFFFu) ? ((__float_as_uint(e_mm) & ~KEY_BIN) | kk) : 0u;
                best.mx = fmaxf(best.mx, fmaxf(e_mp, e_mm));
            } else if constexpr (k == 4) {
                top3_insert(best, e_kp);
            } else {
                top3_insert(best, e_km);
            }
        };
        constexpr int NPIECE = 6 * 4 * C2T;   // pieces per output tile
        constexpr int NSLOT = 2 * KS2;        // MFMA slots per output tile (2*C2T MFMAs each)
        // twiddles (cos, sin)(a'k1) of a k-step are read one k-step ahead (tw_pre), masked once
        // (tw_cur) and serve its cosine half (sums A, C) and its sine half (sums B, E)
        float2 tw_pre, tw_cur;
        int m1, st1;
        auto tile_start = [&](int tt) {   // twiddle index of k-step 0 and its step for output tile tt
            const int k1a = 16 * tt + li; // A row of this lane
            m1 = (k1a * lg) % N1;
            st1 = (4 * k1a) % N1;
            tw_pre = tw1[m1];
        };
        tile_start(0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<TA>([&](auto TT) {
            constexpr int tt = decltype(TT)::value, set = tt & 1;
            static_for<NSLOT>([&](auto SL) {
                constexpr int sl = decltype(SL)::value, s = sl >> 1, half = sl & 1, tau = s >> 2, r = s & 3;
                if constexpr (half == 0) {
                    tw_cur = tw_pre;
                    if constexpr (s == 0) tw_cur.x *= c_half;
                    if constexpr (4 * s + 3 > H1) {   // rows a' > h1 of S and D are not data: drop them
                        const int keep_m = -(int)(4 * s + lg <= H1);
                        tw_cur.x = __int_as_float(__float_as_int(tw_cur.x) & keep_m);
                        tw_cur.y = __int_as_float(__float_as_int(tw_cur.y) & keep_m);
                    }
#pragma unroll
                    for (int c = 0; c < C2T; ++c) {
                        acc[set][0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw_cur.x, Yr[c][tau][r], s == 0 ? z4 : acc[set][0][c], 0, 0, 0);
                        acc[set][2][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw_cur.x, Yi[c][tau][r], s == 0 ? z4 : acc[set][2][c], 0, 0, 0);
                    }
                    if constexpr (s + 1 < KS2) {
                        const int nx = m1 + st1;
                        m1 = nx >= N1 ? nx - N1 : nx;
                        tw_pre = tw1[m1];
                    } else if constexpr (tt + 1 < TA) {
                        tile_start(tt + 1);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C2T; ++c) {
                        acc[set][1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw_cur.y, Yi[c][tau + TA][r], s == 0 ? z4 : acc[set][1][c], 0, 0, 0);
                        acc[set][3][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(tw_cur.y, Yr[c][tau + TA][r], s == 0 ? z4 : acc[set][3][c], 0, 0, 0);
                    }
                }
                if constexpr (tt > 0) {
                    constexpr int lo = sl * NPIECE / NSLOT, hi = (sl + 1) * NPIECE / NSLOT;
                    static_for<hi - lo>([&](auto J) { piece(ic<tt - 1>{}, ic<lo + decltype(J)::value>{}); });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        static_for<NPIECE>([&](auto Q) { piece(ic<TA - 1>{}, Q); });
        // wave-wide merge of the per-lane top three
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned o1 = __shfl_xor(best.k1, d), o2 = __shfl_xor(best.k2, d), o3 = __shfl_xor(best.k3, d);
            best.mx = fmaxf(best.mx, __shfl_xor(best.mx, d));
            top3_insert(best, o1);
            top3_insert(best, o2);
            top3_insert(best, o3);
        }
        {
            const bool silent = best.mx <= 0.f;   // every bin is zero: np.argmax gives bin 0
            const float f1 = __uint_as_float(best.k1 & ~KEY_BIN), f2 = __uint_as_float(best.k2 & ~KEY_BIN),
                        f3 = __uint_as_float(best.k3 & ~KEY_BIN);
            const bool full = f3 >= f1 * keep, top2 = f2 >= f1 * keep;
            const int b = silent ? 0 : (int)(best.k1 & KEY_BIN);
            const unsigned cls = lut_s[b];
            pend_bin = silent ? 0 : (full ? (b | FLAG_FULL) : (top2 ? (b | FLAG_TOP2) : b));
            pend_cls = silent ? cls : (full ? 0u : (top2 ? (best.k2 & KEY_BIN) : cls));
            pend_mag = sqrtf(best.mx);
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
    // squared magnitudes are compared, through keys that drop up to 2^-11 of the value
    const double k = (1.0 - ctx->tie_band) * (1.0 - ctx->tie_band) / (1.0 + 1.0 / 2048.0) * (1.0 - 1e-6);
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
