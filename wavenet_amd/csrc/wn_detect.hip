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
// closer than a relative band (default 3e-5, 10x the f32 error) are re-evaluated for
// those bins in float64 by direct summation with exactly reduced phases; if a third bin
// is in the band too, the whole spectrum is.
#include <math.h>

#include <type_traits>

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
    static constexpr int TW1 = (N1 + 1) & ~1;     // float2 entries, 16-byte multiple
    static constexpr int TW2 = (N2 + 1) & ~1;
    static constexpr size_t LDS = (size_t)(TW1 + TW2) * 8 + (size_t)4 * XW * 4;
};

struct Top3 {
    float m1, m2, m3;
    int b1, b2;
};

__device__ __forceinline__ void top3_insert(Top3 &t, float m, int b) {
    const bool gt1 = (m > t.m1) | ((m == t.m1) & (b < t.b1));
    const bool gt2 = (m > t.m2) | ((m == t.m2) & (b < t.b2));
    t.m3 = gt2 ? t.m2 : fmaxf(t.m3, m);
    const float nm2 = gt1 ? t.m1 : (gt2 ? m : t.m2);
    const int nb2 = gt1 ? t.b1 : (gt2 ? b : t.b2);
    t.m1 = gt1 ? m : t.m1;
    t.b1 = gt1 ? b : t.b1;
    t.m2 = nm2;
    t.b2 = nb2;
}

template <int N1, int N2>
__global__ __launch_bounds__(256, 1) void detect_kernel(const float *__restrict__ audio, long long audio_len,
                                                        long long n_streams, long long stream_stride,
                                                        long long first_sample, long long n_windows,
                                                        const float2 *__restrict__ g_tw1,
                                                        const float2 *__restrict__ g_tw2,
                                                        const uint16_t *__restrict__ lut, int e1, int e2,
                                                        float keep, uint16_t *__restrict__ out_cls,
                                                        float *__restrict__ out_mag, int32_t *__restrict__ out_bin) {
    typedef Plan<N1, N2> P;
    constexpr int N = P::N;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float2 *tw1 = (float2 *)smem;
    float2 *tw2 = tw1 + P::TW1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *xw = (float *)(tw2 + P::TW2) + wave * P::XW;
    for (int i = threadIdx.x; i < N1; i += 256) tw1[i] = g_tw1[i];
    for (int i = threadIdx.x; i < N2; i += 256) tw2[i] = g_tw2[i];
    __syncthreads();

    const int li = lane & 15;  // A row / B column / D column
    const int lg = lane >> 4;  // A,B k index / D row group
    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    long long w = (long long)blockIdx.x * 4 + wave;

    // per-lane constants of the epilogue: bin(k1,k2) = (k1*e1 + k2*e2) mod N
    int q2[P::C2T];
#pragma unroll
    for (int c = 0; c < P::C2T; ++c) q2[c] = (int)(((long long)(16 * c + li) * e2) % N);

    // Window staging by LDS-DMA: 16 bytes per lane, 1 KiB per wave-instruction, written
    // to the wave's own LDS buffer at lds_base + 1024*j + 16*lane.  The loads are issued
    // from inline asm so that the compiler does not order the table reads of stage 2
    // behind them; the wait is the explicit vmcnt(0) at the top of the window loop.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)xw;
    auto fetch = [&](long long win) {
        const long long s = win / n_windows;
        const long long wi = win - s * n_windows;
        const long long g0 = s * stream_stride + first_sample + wi * (long long)N;
        const long long a0 = g0 & ~3ll;
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
        return (int)(g0 - a0);
    };
    int off = 0;
    int next_off = 0;
    if (w < total) next_off = fetch(w);

    while (w < total) {
        off = next_off;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this window's DMA has landed

        // ------------------------------------------------------------------ stage 1
        floatx4 Y[P::C2T][2][P::T1];
        int idxA[P::T1];   // LDS index of x2[a(tile, li)][4*ks + lg]
        bool okA[P::T1];
#pragma unroll
        for (int t = 0; t < P::T1; ++t) {
            const int a = 16 * t + 4 * (li & 3) + (li >> 2);
            okA[t] = a < N1;
            idxA[t] = (N2 * a + N1 * lg) % N;
        }
        int m2[P::C2T], st2[P::C2T];  // (b*k2) mod n2 and its step per k-step
#pragma unroll
        for (int c = 0; c < P::C2T; ++c) {
            int k2 = 16 * c + li;
            k2 = k2 < P::K2 ? k2 : 0;
            m2[c] = (lg * k2) % N2;
            st2[c] = (4 * k2) % N2;
        }
        constexpr int STEP_A = (4 * N1) % N;
        // LDS operands of one k-step (4 values of b): the samples of every row tile and the
        // twiddles of every column tile.  They are read one k-step ahead of their MFMAs.
        float a_nx[P::T1];
        float2 tw_nx[P::C2T];
        auto load_step = [&](int ks) {
            const bool okb = 4 * ks + lg < N2;
#pragma unroll
            for (int t = 0; t < P::T1; ++t) {
                const float v = xw[off + idxA[t]];
                a_nx[t] = (okA[t] & okb) ? v : 0.f;
                int nx = idxA[t] + STEP_A;
                idxA[t] = nx >= N ? nx - N : nx;
            }
#pragma unroll
            for (int c = 0; c < P::C2T; ++c) {
                tw_nx[c] = tw2[m2[c]];
                int nx = m2[c] + st2[c];
                m2[c] = nx >= N2 ? nx - N2 : nx;
            }
        };
        // The first step starts the accumulators from a literal zero (an inline constant
        // of the MFMA), so they need no initialisation.
        auto kstep = [&](int ks, auto first) {
            float a[P::T1];
            float2 tw[P::C2T];
#pragma unroll
            for (int t = 0; t < P::T1; ++t) a[t] = a_nx[t];
#pragma unroll
            for (int c = 0; c < P::C2T; ++c) tw[c] = tw_nx[c];
            load_step(ks + 1);  // past the last step this reads valid, unused entries
#pragma unroll
            for (int c = 0; c < P::C2T; ++c) {
#pragma unroll
                for (int t = 0; t < P::T1; ++t) {
                    if constexpr (decltype(first)::value) {
                        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                        Y[c][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tw[c].x, z, 0, 0, 0);
                        Y[c][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tw[c].y, z, 0, 0, 0);
                    } else {
                        Y[c][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tw[c].x, Y[c][0][t], 0, 0, 0);
                        Y[c][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tw[c].y, Y[c][1][t], 0, 0, 0);
                    }
                }
            }
        };
        load_step(0);
        kstep(0, std::true_type{});
#pragma unroll 1
        for (int ks = 1; ks < P::KS1; ++ks) kstep(ks, std::false_type{});

        // every LDS read of this window has returned (its values fed the MFMAs above):
        // start the next window's DMA so that it lands while stage 2 runs
        const long long wn = w + wstep;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (wn < total) next_off = fetch(wn);

        // ------------------------------------------------------------------ stage 2
        Top3 best;
        best.m1 = best.m2 = best.m3 = -1.f;
        best.b1 = best.b2 = 0x7fffffff;
#pragma unroll 1
        for (int t = 0; t < P::T1; ++t) {
            floatx4 accR[P::C2T], accI[P::C2T];
#pragma unroll
            for (int c = 0; c < P::C2T; ++c) {
                accR[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
                accI[c] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
            const int k1a = 16 * t + li;  // A row of this lane
            int m1 = (k1a * lg) % N1;
            const int st1 = (4 * k1a) % N1;
            float2 tw_next = tw1[m1];
#pragma unroll
            for (int s = 0; s < P::KS2; ++s) {
                const float2 tw = tw_next;
                {  // twiddle of the next k-step, read ahead of this step's MFMAs
                    int nx = m1 + st1;
                    m1 = nx >= N1 ? nx - N1 : nx;
                    if (s + 1 < P::KS2) tw_next = tw1[m1];
                }
                const float cm = tw.x, sm = tw.y, nsm = -tw.y;
#pragma unroll
                for (int c = 0; c < P::C2T; ++c)
                    accR[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cm, Y[c][0][s >> 2][s & 3], accR[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < P::C2T; ++c)
                    accI[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cm, Y[c][1][s >> 2][s & 3], accI[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < P::C2T; ++c)
                    accR[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm, Y[c][1][s >> 2][s & 3], accR[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < P::C2T; ++c)
                    accI[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(nsm, Y[c][0][s >> 2][s & 3], accI[c], 0, 0, 0);
            }
            // epilogue: |Z|^2 of rows k1 = 16t + 4*lg + r, columns k2 = 16c + li
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k1 = 16 * t + 4 * lg + r;
                const int p1 = (k1 * e1) % N;
#pragma unroll
                for (int c = 0; c < P::C2T; ++c) {
                    const int k2 = 16 * c + li;
                    int kb = p1 + q2[c];
                    kb = kb >= N ? kb - N : kb;
                    const int kk = min(kb, N - kb);
                    const bool ok = (k1 < N1) & (k2 < P::K2) & (kk < N / 2) & ((k2 != 0) | (k1 <= N1 / 2));
                    const float re = accR[c][r], im = accI[c][r];
                    const float mg = ok ? re * re + im * im : -2.f;
                    top3_insert(best, mg, kk);
                }
            }
        }
        // wave-wide merge of the per-lane top three
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float om1 = __shfl_xor(best.m1, d), om2 = __shfl_xor(best.m2, d), om3 = __shfl_xor(best.m3, d);
            const int ob1 = __shfl_xor(best.b1, d), ob2 = __shfl_xor(best.b2, d);
            top3_insert(best, om1, ob1);
            top3_insert(best, om2, ob2);
            best.m3 = fmaxf(best.m3, om3);
        }
        if (lane == 0) {
            int b = best.b1;
            float mag = sqrtf(fmaxf(best.m1, 0.f));
            if (best.m1 <= 0.f) {  // silent window: every bin is zero, np.argmax gives 0
                out_bin[w] = 0;
                out_cls[w] = lut[0];
            } else if (best.m3 >= best.m1 * keep) {
                out_bin[w] = b | FLAG_FULL;
                out_cls[w] = 0;
            } else if (best.m2 >= best.m1 * keep) {
                out_bin[w] = b | FLAG_TOP2;
                out_cls[w] = (uint16_t)best.b2;
            } else {
                out_bin[w] = b;
                out_cls[w] = lut[b];
            }
            out_mag[w] = mag;
        }
        w = wn;
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
                       ctx->e1, ctx->e2, (float)k, d_cls, d_mag, d_bin);
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
