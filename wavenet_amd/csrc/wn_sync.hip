// K3: frame-sync correlator.  Replaces the search phase of the receive loops: the test
// "dominant frequency within 10 Hz of the ping tone" of escuchar_ping
// (c1_shared.py:368-384) and of escuchar_y_retornar_trama before the ping was heard
// (c1_shared.py:330-335), without computing the spectrum.
//
// Per window the kernel reads the samples once (HBM-read-bound, 4 bytes per sample) and
// accumulates sum(x), sum(x^2) and a handful of DFT bins by direct correlation: the ping
// bins, a few rival bins (nearest bins of the separator, EOF and byte-0 tones) and the
// one bin of [1, N/2] that np.argmax never looks at.  Parseval then bounds every bin not
// computed:   sum_{k=1..(N-1)/2} |X_k|^2 = (N*sum(x^2) - X_0^2) / 2   (N odd), so
//   - ping certainly dominant:      max ping |X|^2 > everything else could hold;
//   - ping certainly not dominant:  a computed rival (or bin 0) is larger, or the average
//     of the bins not computed is larger than the best ping bin;
//   - otherwise undecided (2): only the full detection kernel can tell.
//
// One wave per window; see the kernel for how the lanes walk it.
#include <math.h>
#include <string.h>

#include <type_traits>

#include "wn_common.h"

#define SYNC_NB 8  // correlator bins per window

typedef float floatx2 __attribute__((ext_vector_type(2)));

struct SyncBins {
    int n_ping;               // bins [0, n_ping) are the ping bins
    int n_used;               // bins [n_ping, n_used) are rivals; slot n_used = excluded bin
    int bin[SYNC_NB];
    float2 step[SYNC_NB];     // rotation by 256 samples
    float2 rot[SYNC_NB][4];   // rotation by q = 0..3 samples
    float inv_n2;             // 2 / N
    int others;               // bins of [1, N/2) that are not computed
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// One wave per window.  Lanes take consecutive float4s, four per step (a wave has 4 KB of
// contiguous HBM in flight); the 4 samples of a float4 share one twiddle per bin, their
// offsets inside the float4 are applied once at the end, and the twiddle advances by a fixed
// rotation (256 samples) from one float4 of the lane to its next.  A window's first twiddles
// come from a table in LDS (bin x window alignment x lane).  Only the first and the last
// step of a window can hold samples outside it; the steps between run without masks or
// bounds checks.
template <int NB>
__global__ __launch_bounds__(256, 3) void sync_kernel(const float *__restrict__ audio, long long audio_len,
                                                   long long n_streams, long long stream_stride,
                                                   long long first_sample, long long n_windows, int N,
                                                   SyncBins sb, float *__restrict__ ping_energy,
                                                   float *__restrict__ win_energy, uint8_t *__restrict__ verdict) {
    __shared__ float2 tw0[NB][4][64];   // twiddle of sample 4*lane - off, window alignment off
    for (int i = threadIdx.x; i < NB * 4 * 64; i += 256) {
        const int b = i >> 8, off = (i >> 6) & 3, l = i & 63;
        int m = (sb.bin[b] * (4 * l - off)) % N;
        m = m < 0 ? m + N : m;
        float sn, cs;
        sincospif((float)m * sb.inv_n2, &sn, &cs);
        tw0[b][off][l] = make_float2(cs, sn);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    for (long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += wstep) {
        const long long s = w / n_windows;
        const long long g0 = s * stream_stride + first_sample + (w - s * n_windows) * (long long)N;
        const long long a0 = g0 & ~3ll;
        const int off = (int)(g0 - a0);
        const float *base = audio + a0 + 4 * lane;
        floatx2 tw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float2 t = tw0[b][off][lane];
            tw[b] = (floatx2){t.x, t.y};
        }
        floatx2 acc[NB][4];   // (re, im) sums per bin and position in the float4: packed FMAs
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[b][q] = (floatx2){0.f, 0.f};
        float sx = 0.f, sxx = 0.f;
        const int nv = (N + off + 3) / 4;  // float4s covering the window
        auto step = [&](auto EDGE, int it) {
            constexpr bool edge = decltype(EDGE)::value;
            float4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = lane + 64 * u + 256 * it;
                if constexpr (edge) {
                    vv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < nv && a0 + 4 * j + 4 <= audio_len) vv[u] = *(const float4 *)(base + 256 * u + 1024 * it);
                } else {
                    vv[u] = *(const float4 *)(base + 256 * u + 1024 * it);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
                if constexpr (edge) {
                    const int n0 = 4 * (lane + 64 * u + 256 * it) - off;
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = (n0 + q >= 0 && n0 + q < N) ? x[q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sx += x[q];
                    sxx = fmaf(x[q], x[q], sxx);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[b][q] = __builtin_elementwise_fma((floatx2){x[q], x[q]}, tw[b], acc[b][q]);
                    const floatx2 t = tw[b];
                    tw[b] = (floatx2){t.x * sb.step[b].x - t.y * sb.step[b].y, t.x * sb.step[b].y + t.y * sb.step[b].x};
                }
            }
        };
        const int last = (nv - 1) / 256;
        step(std::true_type{}, 0);
        for (int it = 1; it < last; ++it) step(std::false_type{}, it);
        if (last > 0) step(std::true_type{}, last);
        // apply the in-float4 offsets, reduce over the wave
        float mag2[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (are + i*aim) * (rot.x + i*rot.y)
                re += acc[b][q].x * sb.rot[b][q].x - acc[b][q].y * sb.rot[b][q].y;
                im += acc[b][q].x * sb.rot[b][q].y + acc[b][q].y * sb.rot[b][q].x;
            }
            re = wave_sum_f(re);
            im = wave_sum_f(im);
            mag2[b] = re * re + im * im;
        }
        sx = wave_sum_f(sx);
        sxx = wave_sum_f(sxx);
        if (lane == 0) {
            float pmax = 0.f, psum = 0.f, rival = sx * sx;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < sb.n_ping) {
                    pmax = fmaxf(pmax, mag2[b]);
                    psum += mag2[b];
                } else if (b < sb.n_used) {
                    rival = fmaxf(rival, mag2[b]);
                }
            }
            float known = psum + mag2[NB - 1];
#pragma unroll
            for (int b = 0; b < NB - 1; ++b)
                if (b >= sb.n_ping && b < sb.n_used) known += mag2[b];
            const float half = 0.5f * ((float)N * sxx - sx * sx);
            const float rest = fmaxf(half - known, 0.f);  // energy of the bins not computed
            const float margin = 1e-3f;
            uint8_t v = 2;
            if (sxx == 0.f) {
                v = 0;  // silence: every bin is zero, the dominant bin is bin 0
            } else if (pmax > (fmaxf(rival, rest) + 1e-3f * half) * (1.f + margin)) {
                v = 1;
            } else if (rival > pmax * (1.f + margin) ||
                       rest * (1.f - margin) - 1e-3f * half > pmax * (float)sb.others * (1.f + margin)) {
                v = 0;
            }
            ping_energy[w] = pmax;
            win_energy[w] = sxx;
            verdict[w] = v;
        }
    }
}

static void fill_rot(SyncBins &sb, int slot, int bin, int N) {
    sb.bin[slot] = bin;
    const double a = 2.0 * M_PI * (double)(((long long)bin * 256) % N) / (double)N;
    sb.step[slot] = make_float2((float)cos(a), (float)sin(a));
    for (int q = 0; q < 4; ++q) {
        const double aq = 2.0 * M_PI * (double)(((long long)bin * q) % N) / (double)N;
        sb.rot[slot][q] = make_float2((float)cos(aq), (float)sin(aq));
    }
}

extern "C" int wn_sync_scan(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                            int64_t stream_stride, int64_t first_sample, int64_t n_windows, float *d_ping_energy,
                            float *d_win_energy, uint8_t *d_verdict, void *stream) {
    WN_REQUIRE(ctx && n_streams >= 0 && n_windows >= 0 && first_sample >= 0 && stream_stride >= 0, "ctx / sizes");
    if (ctx->ping_lo < 0) {
        wn_set_error("the scheme has no ping tone");
        return WN_ERR_UNSUPPORTED;
    }
    if (n_streams == 0 || n_windows == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    WN_REQUIRE(d_audio && d_ping_energy && d_win_energy && d_verdict, "null device pointer");
    WN_REQUIRE(((uintptr_t)d_audio % 16) == 0 && audio_len % 4 == 0, "d_audio 16-byte aligned, audio_len % 4 == 0");
    const int64_t span = first_sample + n_windows * (int64_t)ctx->n;
    WN_REQUIRE(n_streams == 1 || span <= stream_stride, "windows of a stream overlap the next stream");
    WN_REQUIRE((n_streams - 1) * stream_stride + span <= audio_len, "windows reach past audio_len");
    const int N = ctx->n;
    WN_REQUIRE(N % 2 == 1, "sync scan needs an odd window length");
    const int n_ping = ctx->ping_hi - ctx->ping_lo + 1;
    WN_REQUIRE(n_ping >= 1 && n_ping <= 3, "1..3 ping bins");

    SyncBins sb;
    memset(&sb, 0, sizeof(sb));
    sb.n_ping = n_ping;
    int slot = 0;
    for (int k = ctx->ping_lo; k <= ctx->ping_hi; ++k) fill_rot(sb, slot++, k, N);
    // rivals: nearest bins of the separator, EOF and byte-0 tones (distinct from the ping bins)
    const double spacing = (double)ctx->s.sample_rate / (double)N;
    const int tones[3] = {ctx->s.sep_freq, ctx->s.eof_freq, ctx->s.base_freq};
    for (int t = 0; t < 3 && slot < SYNC_NB - 1; ++t) {
        const int k = (int)llround((double)tones[t] / spacing);
        bool dup = k <= 0 || k >= ctx->nbins;
        for (int j = 0; j < slot; ++j) dup = dup || sb.bin[j] == k;
        if (!dup) fill_rot(sb, slot++, k, N);
    }
    sb.n_used = slot;
    const int nb = slot + 1;
    fill_rot(sb, slot, (N - 1) / 2, N);  // last slot: the bin of [1, N/2] that argmax never sees
    sb.inv_n2 = (float)(2.0 / (double)N);
    // bins of [1, N/2) not computed: all but ping and rivals
    sb.others = (ctx->nbins - 1) - sb.n_used;

    const long long total = n_streams * n_windows;
    long long blocks = (total + 3) / 4;
    const long long cap = (long long)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
#define SYNC_LAUNCH(NBV)                                                                                         \
    hipLaunchKernelGGL(sync_kernel<NBV>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_audio,     \
                       (long long)audio_len, (long long)n_streams, (long long)stream_stride,                     \
                       (long long)first_sample, (long long)n_windows, N, sb, d_ping_energy, d_win_energy, d_verdict)
    switch (nb) {
        case 2: SYNC_LAUNCH(2); break;
        case 3: SYNC_LAUNCH(3); break;
        case 4: SYNC_LAUNCH(4); break;
        case 5: SYNC_LAUNCH(5); break;
        case 6: SYNC_LAUNCH(6); break;
        case 7: SYNC_LAUNCH(7); break;
        default: SYNC_LAUNCH(8); break;
    }
#undef SYNC_LAUNCH
    return wn_launch_check("sync_kernel");
}
