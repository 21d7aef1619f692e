// K3: frame-sync correlator.  Replaces the search phase of the receive loops: the test
// "dominant frequency within 10 Hz of the ping tone" of escuchar_ping
// (c1_shared.py:368-384) and of escuchar_y_retornar_trama before the ping was heard
// (c1_shared.py:330-335), without computing the spectrum.
//
// Per window the kernel reads the samples once (HBM-read-bound, 4 bytes per sample) and
// computes sum(x), sum(x^2) and a handful of DFT bins by direct correlation: the ping
// bins, a few rival bins (nearest bins of the separator, EOF and byte-0 tones) and the
// one bin of [1, N/2] that np.argmax never looks at.  Parseval then bounds every bin not
// computed:   sum_{k=1..(N-1)/2} |X_k|^2 = (N*sum(x^2) - X_0^2) / 2   (N odd), so
//   - ping certainly dominant:      max ping |X|^2 > everything else could hold;
//   - ping certainly not dominant:  a computed rival (or bin 0) is larger, or the average
//     of the bins not computed is larger than the best ping bin;
//   - otherwise undecided (2): only the full detection kernel can tell.
//
// The correlations run on the matrix cores; see the kernel.
#include <math.h>
#include <string.h>

#include <type_traits>

#include "wn_common.h"
#include "wn_dma.h"

#define SYNC_NB 8  // correlator bins per window (at most 7 are used: rows 0..13 of the MFMA tile)

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct SyncBins {
    int n_ping;               // bins [0, n_ping) are the ping bins
    int n_used;               // bins [n_ping, n_used) are rivals; slot n_used = excluded bin
    int n_bins;               // n_used + 1
    int bin[SYNC_NB];
    float inv_n2;             // 2 / N
    int others;               // bins of [1, N/2) that are not computed
};

template <int N>
struct SyncPlan {
    static constexpr int CH = 256;                          // samples per chunk = one 1-KiB DMA piece
    static constexpr int PITCH = CH + 4;                    // floats between chunks in LDS (bank spread)
    static constexpr int NV4MAX = (N + 3 + 3) / 4;          // float4s of a window at the worst alignment
    static constexpr int NCH = (4 * NV4MAX + CH - 1) / CH;  // chunks that can hold data
    static constexpr int G = (NCH + 15) / 16;               // column groups of 16 chunks
    static constexpr int BUF = G * 16 * PITCH;              // floats of LDS per wave
    static constexpr int R1 = NCH > 16 ? NCH - 16 : 0;      // DMA pieces of the second group
    static_assert(G == 1 || G == 2, "window length");
    static constexpr size_t LDS = (size_t)4 * BUF * 4;
};

template <int CTRL>
__device__ __forceinline__ float row_add(float v) {   // v + the value of the lane the DPP control pairs it with
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {   // v of a given lane, in every lane
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// sum over the 16 lanes of a DPP row, left in every lane of the row
__device__ __forceinline__ float row_sum(float v) {
    v = row_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = row_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = row_add<0x141>(v);   // row_half_mirror
    v = row_add<0x140>(v);   // row_mirror
    return v;
}

// One wave per window, on the matrix cores.  The window is cut into chunks of 256 samples;
// a correlation over the window is the sum over chunks c of (chunk correlation) x
// exp(-i*theta*256*c), and the chunk correlations of 16 chunks at once are one small matrix
// product: A (16 x 256) holds cos / -sin of up to 7 bins and a row of ones (sum of x), the
// same for every chunk, so each lane keeps its 64 entries in registers for the whole launch;
// B (256 x 16) is 16 chunks of the window, read from LDS.  v_mfma_f32_16x16x4_f32, 64 per 16
// chunks.  The window is staged by DMA, one 1-KiB piece per chunk at a pitch of 260 floats
// (conflict-free 16-byte reads); the buffer has two halves (chunks 0-15, 16-31) and the
// next window's half is fetched as soon as this window's half has been read.  Window
// starts are not 16-byte aligned: position p of the buffer is sample p - off, which only
// turns the phase of every bin; the up to 3 positions before the window and the few after
// it are zeroed.  sum(x^2) is accumulated by the vector ALU from the same reads.
template <int N>
__global__ __launch_bounds__(256, 1) void sync_kernel(const float *__restrict__ audio, long long n_streams,
                                                      long long stream_stride, long long first_sample,
                                                      long long n_windows, SyncBins sb,
                                                      float *__restrict__ ping_energy, float *__restrict__ win_energy,
                                                      uint8_t *__restrict__ verdict) {
    typedef SyncPlan<N> P;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lg = lane >> 4;
    float *buf = (float *)smem + wave * P::BUF;
    for (int i = lane; i < P::BUF; i += 64) buf[i] = 0.f;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)buf;

    // A operand: row li, k index (4j + e) <-> position 16j + 4lg + e of a chunk
    const int row_bin = li >> 1;
    const bool row_live = row_bin < sb.n_bins, row_ones = li == 14;
    float a_tab[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const int p = 16 * (k >> 2) + 4 * lg + (k & 3);
        float sn, cs;
        sincospif((float)((sb.bin[row_bin & 7] * p) % N) * sb.inv_n2, &sn, &cs);
        a_tab[k] = row_ones ? 1.f : (row_live ? ((li & 1) ? -sn : cs) : 0.f);
    }
    // chunk c = 16g + li of this lane's column: exp(-i*theta*256*c) for the lane's two bins
    // (rows 4lg, 4lg+1 and 4lg+2, 4lg+3 of the result tile)
    float wc[P::G][2], ws[P::G][2];
#pragma unroll
    for (int g = 0; g < P::G; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = 2 * lg + h;
            float sn, cs;
            sincospif((float)(((long long)sb.bin[b & 7] * 256 * (16 * g + li)) % N) * sb.inv_n2, &sn, &cs);
            wc[g][h] = b < sb.n_bins ? cs : 0.f;
            ws[g][h] = b < sb.n_bins ? -sn : 0.f;
        }

    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    long long w = (long long)blockIdx.x * 4 + wave;
    // (stream, window) of the window to fetch next, advanced without divisions
    const long long adv_s = wstep / n_windows, adv_w = wstep - adv_s * n_windows;
    long long f_s = w / n_windows, f_w = w - f_s * n_windows;
    const float *f_base = nullptr;   // 16-byte aligned start of the window being fetched
    int f_off = 0, f_nv4 = 0;
    auto fetch_begin = [&]() {
        const long long g0 = f_s * stream_stride + first_sample + f_w * (long long)N;
        f_off = (int)(g0 & 3);
        f_base = audio + (g0 - f_off) + 4 * lane;
        f_nv4 = (N + f_off + 3) / 4;
        f_s += adv_s;
        f_w += adv_w;
        if (f_w >= n_windows) {
            f_w -= n_windows;
            f_s += 1;
        }
    };
    auto fetch_half = [&](auto H) {   // chunks [16h, 16h+16) of the window: one piece per chunk
        constexpr int h = decltype(H)::value;
        constexpr int c0 = 16 * h, c1 = (P::NCH < c0 + 16 ? P::NCH : c0 + 16);
#pragma unroll
        for (int c = c0; c < c1; ++c) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 4u * P::PITCH * c);
            if (64 * (c + 1) <= P::NV4MAX - 1) {            // whole piece inside every window
                dma_group<1, 1024, true>(f_base + 256 * c, dst);
            } else {                                         // the last piece: only the float4s of the window
                if (64 * c + lane < f_nv4) dma_group<1, 1024, true>(f_base + 256 * c, dst);
            }
        }
    };
    auto fix_head = [&](int off) {   // positions before the window
        if (lane < off) buf[lane] = 0.f;
    };
    auto fix_tail = [&](int off) {   // positions after it, up to the end of the last float4 any alignment writes
        const int p = off + N + lane;
        if (p < 4 * P::NV4MAX) buf[P::PITCH * (p >> 8) + (p & 255)] = 0.f;
    };

    bool pend = false;
    long long pend_w = 0;
    float pend_pe = 0.f, pend_we = 0.f;
    unsigned pend_v = 0;
    auto flush = [&]() {
        if (pend && lane == 0) {
            ping_energy[pend_w] = pend_pe;
            win_energy[pend_w] = pend_we;
            verdict[pend_w] = (uint8_t)pend_v;
        }
    };

    int off = 0;
    if (w < total) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill above is in LDS
        fetch_begin();
        off = f_off;
        fetch_half(std::integral_constant<int, 0>{});
        if constexpr (P::G == 2) fetch_half(std::integral_constant<int, 1>{});
    }
    while (w < total) {
        const long long wn = w + wstep;
        const bool more = wn < total;
        float xr[2] = {0.f, 0.f}, xi[2] = {0.f, 0.f}, sxx = 0.f;
        int next_off = 0;
        auto group = [&](auto GG) {
            constexpr int g = decltype(GG)::value;
            const float *col = buf + P::PITCH * (16 * g + li) + 4 * lg;
            floatx4 acc[4];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 v = *(const float4 *)(col + 16 * j);
                sxx = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, sxx))));
                const float x[4] = {v.x, v.y, v.z, v.w};
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_tab[4 * j + e], x[e], j == 0 ? z : acc[e], 0, 0, 0);
            }
            const floatx4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);   // rows 4lg..4lg+3 of column 16g+li
            // (cos-sum + i * minus-sin-sum) * exp(-i*theta*256*c)
            xr[0] += t[0] * wc[g][0] - t[1] * ws[g][0];
            xi[0] += t[0] * ws[g][0] + t[1] * wc[g][0];
            if (lg < 3) {
                xr[1] += t[2] * wc[g][1] - t[3] * ws[g][1];
                xi[1] += t[2] * ws[g][1] + t[3] * wc[g][1];
            } else {
                xr[1] += t[2];                                     // row 14: sum of x
            }
        };
        // ---- first half
        if constexpr (P::G == 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::R1) : "memory");   // all but the second half's pieces
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flush();
        fix_head(off);
        if constexpr (P::G == 1) fix_tail(off);
        group(std::integral_constant<int, 0>{});
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this half are done
            fetch_begin();
            next_off = f_off;
            fetch_half(std::integral_constant<int, 0>{});
        }
        // ---- second half
        if constexpr (P::G == 2) {
            if (more)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // everything before the 16 pieces just issued
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fix_tail(off);
            group(std::integral_constant<int, 1>{});
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fetch_half(std::integral_constant<int, 1>{});
            }
        }
        // ---- sums over the 16 columns (one DPP row per lane group), then over the wave
        float m0 = row_sum(xr[0]), m1 = row_sum(xi[0]), m2 = row_sum(xr[1]), m3 = row_sum(xi[1]);
        float sq = row_sum(sxx);
        sq = lane_value(sq, 0) + lane_value(sq, 16) + lane_value(sq, 32) + lane_value(sq, 48);
        const float e0 = m0 * m0 + m1 * m1, e1 = m2 * m2 + m3 * m3;   // |X|^2 of bins 2lg, 2lg+1
        float mag2[SYNC_NB];
#pragma unroll
        for (int b = 0; b < 7; ++b) mag2[b] = lane_value((b & 1) ? e1 : e0, 16 * (b >> 1));
        const float sx = lane_value(m2, 48);
        {
            float pmax = 0.f, rival = sx * sx, known = 0.f;
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                if (b < sb.n_ping) {
                    pmax = fmaxf(pmax, mag2[b]);
                } else if (b < sb.n_used) {
                    rival = fmaxf(rival, mag2[b]);
                }
                if (b < sb.n_bins) known += mag2[b];
            }
            const float half = 0.5f * ((float)N * sq - sx * sx);
            const float rest = fmaxf(half - known, 0.f);  // energy of the bins not computed
            const float margin = 1e-3f;
            unsigned v = 2;
            const float tot = (float)N * sq;
            if (!(tot >= 1e-28f && tot <= 1e30f)) {
                // Silence, or too quiet / too loud for float32 energies.  The bin sums themselves
                // (not squared) tell the two apart well enough: if every ping bin is exactly
                // zero the dominant bin is not a ping bin (silence gives bin 0); anything else
                // is left to the spectrum.
                const float l0 = fabsf(m0) + fabsf(m1), l1 = fabsf(m2) + fabsf(m3);
                float plin = 0.f;
#pragma unroll
                for (int b = 0; b < 7; ++b)
                    if (b < sb.n_ping) plin += lane_value((b & 1) ? l1 : l0, 16 * (b >> 1));
                v = plin == 0.f ? 0 : 2;
            } else if (pmax > (fmaxf(rival, rest) + 1e-3f * half) * (1.f + margin)) {
                v = 1;
            } else if (rival > pmax * (1.f + margin) ||
                       rest * (1.f - margin) - 1e-3f * half > pmax * (float)sb.others * (1.f + margin)) {
                v = 0;
            }
            pend_pe = pmax;
            pend_we = sq;
            pend_v = v;
            pend_w = w;
            pend = true;
        }
        off = next_off;
        w = wn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flush();
}

int wn_sync_device_setup(int) {
    WN_HIP_TRY(hipFuncSetAttribute((const void *)sync_kernel<7055>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)SyncPlan<7055>::LDS));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)sync_kernel<2205>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)SyncPlan<2205>::LDS));
    return WN_OK;
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
    for (int k = ctx->ping_lo; k <= ctx->ping_hi; ++k) sb.bin[slot++] = k;
    // rivals: nearest bins of the separator, EOF and byte-0 tones (distinct from the ping bins)
    const double spacing = (double)ctx->s.sample_rate / (double)N;
    const int tones[3] = {ctx->s.sep_freq, ctx->s.eof_freq, ctx->s.base_freq};
    for (int t = 0; t < 3 && slot < 6; ++t) {
        const int k = (int)llround((double)tones[t] / spacing);
        bool dup = k <= 0 || k >= ctx->nbins;
        for (int j = 0; j < slot; ++j) dup = dup || sb.bin[j] == k;
        if (!dup) sb.bin[slot++] = k;
    }
    sb.n_used = slot;
    sb.bin[slot] = (N - 1) / 2;  // last slot: the bin of [1, N/2] that argmax never sees
    sb.n_bins = slot + 1;
    sb.inv_n2 = (float)(2.0 / (double)N);
    // bins of [1, N/2) not computed: all but ping and rivals
    sb.others = (ctx->nbins - 1) - sb.n_used;

    const long long total = n_streams * n_windows;
    long long blocks = (total + 3) / 4;
    if (blocks > ctx->num_cus) blocks = ctx->num_cus;
    if (N == 7055)
        hipLaunchKernelGGL(sync_kernel<7055>, dim3((unsigned)blocks), dim3(256), SyncPlan<7055>::LDS,
                           (hipStream_t)stream, d_audio, (long long)n_streams, (long long)stream_stride,
                           (long long)first_sample, (long long)n_windows, sb, d_ping_energy, d_win_energy, d_verdict);
    else if (N == 2205)
        hipLaunchKernelGGL(sync_kernel<2205>, dim3((unsigned)blocks), dim3(256), SyncPlan<2205>::LDS,
                           (hipStream_t)stream, d_audio, (long long)n_streams, (long long)stream_stride,
                           (long long)first_sample, (long long)n_windows, sb, d_ping_energy, d_win_energy, d_verdict);
    else {
        wn_set_error("no sync kernel for windows of %d samples", N);
        return WN_ERR_UNSUPPORTED;
    }
    return wn_launch_check("sync_kernel");
}
