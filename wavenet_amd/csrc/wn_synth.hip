// K1: batched tone synthesis, and the synthetic channel's noise.
//
// Replaces transmite_freq / transmitir_silencio / emitir_trama (c1_shared.py:386-434),
// the WAV variant guardar_trama_como_wav (c1_shared.py:440-471) and capa1's
// transmite_freq / emitir_trama / WAV block (capa1.py:33-46, 163-173).
//
// Every tone restarts at t = 0 and every frequency is a whole number of Hz, so sample i
// of a tone of f Hz is amp*sin(2*pi*((f*i) mod sr)/sr): an exact table lookup in one
// quarter period of sr entries (built on the host in float64, so each output sample is
// the correctly rounded value).  The kernels are HBM-write-bound: 4 bytes (float) or
// 2 bytes (int16) per sample out, nothing but the ~1.5 KB segment table per frame in.
//
// Layout: one workgroup owns a contiguous part of one frame's samples, keeps the frame's
// segment table in LDS (and the quarter-wave table too when the scheme's tones are short),
// finds its first segment once, and every thread writes 4 consecutive samples per step from a
// one-period table of the segment's tone.
#include <stdlib.h>

#include "wn_common.h"

#define SYN_BLOCK 512
#define SYN_MAX_SEGS 768  // segments of one frame the LDS table holds (2*274+4 fits)

// ---------------------------------------------------------------------------------------
// Segment layout: one thread per frame walks its bytes (<= 274) once.
// ---------------------------------------------------------------------------------------
__global__ void synth_layout_kernel(wn_scheme s, const uint8_t *__restrict__ frames,
                                    const int32_t *__restrict__ lens, int stride, long long n, int variant,
                                    int seg_cap, int32_t *__restrict__ seg_start, uint16_t *__restrict__ seg_freq,
                                    int32_t *__restrict__ n_segs, long long *__restrict__ n_samples) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const uint8_t *fr = frames + f * (long long)stride;
    int len = lens[f];
    len = len < 0 ? 0 : (len > stride ? stride : len);
    int32_t *st = seg_start + f * (long long)(seg_cap + 1);
    uint16_t *fq = seg_freq + f * (long long)seg_cap;
    int j = 0;
    int32_t t = 0;
#define PUSH(freq, cnt)                  \
    do {                                 \
        if (j < seg_cap) {               \
            st[j] = t;                   \
            fq[j] = (uint16_t)(freq);    \
            ++j;                         \
            t += (cnt);                  \
        }                                \
    } while (0)
    if (s.sep_freq == 0) {  // prototype scheme (capa1.py:40-46; its WAV block has no end tone)
        for (int i = 0; i < len; ++i) PUSH(s.base_freq + fr[i] * s.freq_step, s.tone_n);
        if (variant == WN_EMIT_LIVE) PUSH(s.eof_freq, s.eof_n);
    } else if (variant == WN_EMIT_LIVE) {  // c1_shared.py:420-433
        PUSH(s.ping_freq, s.ping_n);
        int prev = -1;
        for (int i = 0; i < len; ++i) {
            const int b = fr[i];
            PUSH(s.sep_freq, b == prev ? s.sep_short_n : s.sep_n);
            PUSH(s.base_freq + b * s.freq_step, s.tone_n);
            prev = b;
        }
        PUSH(s.sep_freq, s.sep_n);
        PUSH(s.eof_freq, s.eof_n);
    } else {  // c1_shared.py:445-466
        PUSH(s.ping_freq, s.sep_n);
        PUSH(s.sep_freq, s.sep_n);
        for (int i = 0; i < len; ++i) {
            PUSH(s.base_freq + fr[i] * s.freq_step, s.tone_n);
            PUSH(s.sep_freq, s.sep_n);
        }
        PUSH(s.eof_freq, s.tone_n);
        PUSH(s.sep_freq, s.sep_n);
    }
#undef PUSH
    st[j] = t;
    n_segs[f] = j;
    n_samples[f] = t;
}

// ---------------------------------------------------------------------------------------
// Synthesis
// ---------------------------------------------------------------------------------------
template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
    typedef float4 type;
};
template <>
struct Vec4<int16_t> {
    typedef short4 type;
};

// (f*d) mod sr for f, d < 2^15 (product < 2^30), sr a run-time constant.
__device__ __forceinline__ int mod_sr(unsigned x, int sr, float inv_sr) {
    int q = (int)((float)x * inv_sr);
    int r = (int)x - q * sr;
    r = r < 0 ? r + sr : r;
    r = r >= sr ? r - sr : r;
    return r;
}

// LDS position of quarter-wave table entry i.  Lanes of a wave read entries f*4 apart (mod sr);
// for the many tones whose 4*f is a multiple of 32 that is one bank for the whole wave, so the
// low five bits are XOR-ed with the next two groups of five (a bijection inside each 32-entry row).
__device__ __forceinline__ int tab_pos(int i) { return i ^ (((i >> 5) ^ (i >> 10)) & 31); }

// sin(2*pi*m/sr) for 0 <= m <= quarter = sr/4, w = 2*pi/sr, in float64 by the power series on
// [0, pi/4] (the sine of m or the cosine of quarter - m, whichever argument is the smaller):
// within a few units of the last place, which a test compares entry by entry, after rounding
// to the output format, with the table the host builds with sin().
__device__ __forceinline__ double sin_quarter(int m, int quarter, double w) {
    const bool upper = 2 * m > quarter;
    const double x = w * (double)(upper ? quarter - m : m), z = x * x;
    double s = -1.0 / 1307674368000.0;                      // sine: -x^15/15! ... -x^3/3!
    s = fma(s, z, 1.0 / 6227020800.0);
    s = fma(s, z, -1.0 / 39916800.0);
    s = fma(s, z, 1.0 / 362880.0);
    s = fma(s, z, -1.0 / 5040.0);
    s = fma(s, z, 1.0 / 120.0);
    s = fma(s, z, -1.0 / 6.0);
    s = fma(x * z, s, x);
    double c = 1.0 / 20922789888000.0;                      // cosine: x^16/16! ... -x^2/2!
    c = fma(c, z, -1.0 / 87178291200.0);
    c = fma(c, z, 1.0 / 479001600.0);
    c = fma(c, z, -1.0 / 3628800.0);
    c = fma(c, z, 1.0 / 40320.0);
    c = fma(c, z, -1.0 / 720.0);
    c = fma(c, z, 1.0 / 24.0);
    c = fma(c, z, -0.5);
    c = fma(c, z, 1.0);
    return upper ? c : s;
}

#define SYN_MAX_STRIDE 2304  // longest tone period kept as a table: 2301 samples + 3 wrapped entries
// LDS map for a segment table of `segs` entries: starts (segs + 1 int32), frequencies, periods, then the tables
constexpr int syn_off_freq(int segs) { return ((segs + 1) * 4 + 15) & ~15; }
constexpr int syn_off_period(int segs) { return syn_off_freq(segs) + segs * 2; }
constexpr int syn_off_tab(int segs) { return syn_off_period(segs) + segs * 2; }
// The small-footprint instance: 256 threads, a segment table of 320 entries (frames of up to 158
// bytes), period tables of up to 893 samples (every tone of the live scheme), sines from the
// quarter-wave table in global memory: under 48 registers and 10 KB of LDS, so that its workgroups
// find room on a CU whose SIMDs each hold one wave of the detection kernel.
#define SYN_LEAN_BLOCK 256
#define SYN_LEAN_SEGS 320
#define SYN_LEAN_STRIDE 896

// One workgroup owns a contiguous part of one frame's samples and goes through the segments
// that overlap it.  LDS holds the frame's segment table and two buffers of one tone period
// each, and with TAB_LDS the quarter-wave table (otherwise a sine is computed in float64 where it
// is needed, which only the building of a period table and the rare sample-by-sample paths do).  A tone of f Hz repeats every sr/gcd(f, sr) samples: the
// period of the current segment's tone (plus three wrapped entries) sits in one buffer, the
// next segment's is built from the quarter-wave table into the other before this segment is
// written, so a segment costs one barrier, and the steady state issues no global loads at all
// (a wait for a load would also wait for every store issued before it).  Inside a segment a
// thread keeps its phase j and advances it by the workgroup's step modulo the period; the four
// samples of an aligned vector are s[j..j+3].  Samples before the first and after the last
// aligned vector of a segment are written one by one.  Segments shorter than four periods (and
// tones whose period does not fit a buffer) take their samples straight from the quarter-wave
// table.
// BLOCK threads, a segment table of MAXSEGS entries, period tables of MAXSTRIDE entries; SINE_GLOBAL:
// single sines are read from the quarter-wave table in global memory instead of being computed.
template <typename T, bool TAB_LDS, int BLOCK, int MAXSEGS, int MAXSTRIDE, bool SINE_GLOBAL>
__device__ __forceinline__ void synth_body(
    const int32_t *__restrict__ seg_start, const uint16_t *__restrict__ seg_freq, const int32_t *__restrict__ n_segs,
    int seg_cap, long long n_frames, int parts, int part_len, const long long *__restrict__ dst_off,
    const long long *__restrict__ fill_to, T *__restrict__ audio, long long audio_len,
    const T *__restrict__ g_tab, int quarter, int tab_entries, double amp, double gain, double w_unit) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int32_t *s_start = (int32_t *)smem;                        // MAXSEGS + 1
    uint16_t *s_freq = (uint16_t *)(smem + syn_off_freq(MAXSEGS));      // frequency mod sr
    uint16_t *s_period = (uint16_t *)(smem + syn_off_period(MAXSEGS));  // period of the tone, 0 = no table
    T *tab_s = (T *)(smem + syn_off_tab(MAXSEGS));                      // tab_entries (quarter + 1 rounded up to 32)
    T *per = tab_s + (TAB_LDS ? tab_entries : 0);              // [2][MAXSTRIDE]
    const int tid = threadIdx.x;
    if constexpr (TAB_LDS)
        for (int i = tid; i <= quarter; i += BLOCK) tab_s[tab_pos(i)] = g_tab[i];
    const int sr = 4 * quarter, half = 2 * quarter;
    const float inv_sr = 1.0f / (float)sr;
    typedef typename Vec4<T>::type V4;
    auto sine = [&](int m) -> T {          // amp*sin(2*pi*m/sr), 0 <= m < sr
        const bool neg = m >= half;
        const int m2 = neg ? m - half : m;
        const int mq = m2 > quarter ? half - m2 : m2;
        T v;
        if constexpr (TAB_LDS) {
            v = tab_s[tab_pos(mq)];
        } else if constexpr (SINE_GLOBAL) {
            v = g_tab[mq];
        } else {
            const double sn = sin_quarter(mq, quarter, w_unit);
            if constexpr (sizeof(T) == 4) {
                double w = amp * sn;
                if (gain != 1.0) {
                    w = w * gain;
                    w = w < -1.0 ? -1.0 : (w > 1.0 ? 1.0 : w);
                }
                v = (T)w;
            } else {
                v = (T)(amp * sn * 32767.0);
            }
        }
        return neg ? (T)(-v) : v;
    };
    auto build = [&](T *dst, int f, int period) {   // one period of the f-Hz tone and its first three samples again
        for (int i = tid; i < period + 3; i += BLOCK) {
            const int ii = i >= period ? i - period : i;
            dst[i] = sine(mod_sr((unsigned)f * (unsigned)ii, sr, inv_sr));
        }
    };

    const long long n_items = n_frames * parts;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const long long f = item / parts;
        const int part = (int)(item - f * parts);
        __syncthreads();  // quarter-wave table ready / previous item's tables no longer read
        const int ns = n_segs[f];
        for (int i = tid; i <= ns; i += BLOCK) s_start[i] = seg_start[f * (long long)(seg_cap + 1) + i];
        for (int i = tid; i < ns; i += BLOCK) {
            const unsigned fq = (unsigned)seg_freq[f * (long long)seg_cap + i] % (unsigned)sr;
            unsigned g = (unsigned)sr, r = fq;     // gcd(sr, fq) by Euclid; fq = 0 gives sr, a period of one sample
            while (r) {
                const unsigned t = g % r;
                g = r;
                r = t;
            }
            const unsigned period = (unsigned)sr / g;
            s_freq[i] = (uint16_t)fq;
            s_period[i] = period + 3 <= MAXSTRIDE ? (uint16_t)period : (uint16_t)0;
        }
        __syncthreads();
        const int total = s_start[ns];
        long long end = fill_to ? fill_to[f] : (long long)total;
        if (end < total) end = total;
        const long long base = dst_off[f];
        if (base < 0) continue;                              // never write outside the buffer
        if (base + end > audio_len) end = audio_len - base;
        const long long p0l = (long long)part * part_len;
        long long p1l = p0l + part_len;
        if (p1l > end) p1l = end;
        if (p0l >= p1l) continue;
        const int p0 = (int)p0l, p1 = (int)p1l;
        T *out = audio + base;
        const int mis = (int)(base & 3);       // sample p is vector-aligned when (mis + p) % 4 == 0

        // first segment of this part: binary search (same result in every thread)
        int sg = ns;
        if (p0 < total) {
            int lo = 0, hi = ns - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_start[mid] <= p0) lo = mid; else hi = mid - 1;
            }
            sg = lo;
        }
        if (!SINE_GLOBAL && (long long)total < (long long)ns * (16 * BLOCK)) {
            // Short segments (fewer than four steps of the workgroup each, as in the prototype
            // scheme): going segment by segment with the whole workgroup would leave most lanes
            // idle, so each wave takes a segment of its own (every BLOCK/64-th one of the part):
            // the segment's bounds and frequency are the same in all its lanes, a lane keeps its
            // phase and advances it by the wave's step of 256 samples, and the samples come
            // straight from the quarter-wave table.
            const int wave = tid >> 6, lane = tid & 63;
            for (int g = sg + wave; g < ns; g += BLOCK / 64) {
                const int lo = s_start[g], hi = s_start[g + 1];
                if (lo >= p1) break;
                const int a = lo > p0 ? lo : p0, b = hi < p1 ? hi : p1;
                const int fq = s_freq[g];
                int va = a + ((4 - ((mis + a) & 3)) & 3);    // first aligned vector of the segment
                va = va > b ? b : va;
                const int vb = va + ((b - va) & ~3);         // end of the last one
                int v = va + 4 * lane;
                if (v < vb) {
                    int m = mod_sr((unsigned)fq * (unsigned)(v - lo), sr, inv_sr);
                    const int mstep = mod_sr((unsigned)fq * 256u, sr, inv_sr);
                    for (; v < vb; v += 256) {
                        int mk = m;
                        T e[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            e[k] = sine(mk);
                            mk += fq;
                            mk = mk >= sr ? mk - sr : mk;
                        }
                        V4 o;
                        o.x = e[0]; o.y = e[1]; o.z = e[2]; o.w = e[3];
                        *(V4 *)(out + v) = o;
                        m += mstep;
                        m = m >= sr ? m - sr : m;
                    }
                }
                if (lane < 8) {                 // up to three samples before va and after vb
                    const int pos = lane < 4 ? a + lane : vb + (lane - 4);
                    const int lim = lane < 4 ? va : b;
                    if (pos < lim) out[pos] = sine(mod_sr((unsigned)fq * (unsigned)(pos - lo), sr, inv_sr));
                }
            }
            sg = ns;   // nothing left for the segment loop below; the zero fill follows it
        }
        // a table pays when the segment is several periods long; otherwise straight from the
        // quarter-wave table
        auto table_period = [&](int g) -> int {
            const int period = s_period[g];
            return (period && 4 * (period + 3) <= s_start[g + 1] - s_start[g]) ? period : 0;
        };
        int cur = 0;
        int period = sg < ns ? table_period(sg) : 0;
        // frequency whose table a buffer holds (-1: none): the separator comes back every other
        // segment and finds its table still there
        int held[2] = {-1, -1};
        if (period) {
            build(per, s_freq[sg], period);
            held[0] = s_freq[sg];
        }
        __syncthreads();
        while (sg < ns) {
            const int lo = s_start[sg], hi = s_start[sg + 1];
            if (lo >= p1) break;
            const int a = lo > p0 ? lo : p0, b = hi < p1 ? hi : p1;
            const int nx = sg + 1;
            const int period_nx = (nx < ns && hi < p1) ? table_period(nx) : 0;
            if (period_nx && held[cur ^ 1] != (int)s_freq[nx]) {
                build(per + (cur ^ 1) * MAXSTRIDE, s_freq[nx], period_nx);
                held[cur ^ 1] = s_freq[nx];
            }
            int va = a + ((4 - ((mis + a) & 3)) & 3);    // first aligned vector of the segment
            va = va > b ? b : va;
            const int vb = va + ((b - va) & ~3);         // end of the last one
            if (period) {
                const T *tone = per + cur * MAXSTRIDE;
                int v = va + 4 * tid;
                if (v < vb) {
                    int j = (int)((unsigned)(v - lo) % (unsigned)period);
                    const int jstep = (4 * BLOCK) % period;
                    for (; v < vb; v += 4 * BLOCK) {
                        V4 o;
                        o.x = tone[j]; o.y = tone[j + 1]; o.z = tone[j + 2]; o.w = tone[j + 3];
                        *(V4 *)(out + v) = o;
                        j += jstep;
                        j = j >= period ? j - period : j;
                    }
                }
                if (tid < 8) {                 // up to three samples before va and after vb
                    const int pos = tid < 4 ? a + tid : vb + (tid - 4);
                    const int lim = tid < 4 ? va : b;
                    if (pos < lim) out[pos] = tone[(unsigned)(pos - lo) % (unsigned)period];
                }
            } else if constexpr (SINE_GLOBAL) {
                // no table for this tone (its period is too long, or the segment too short): sample by
                // sample from the quarter-wave table; correct, and not meant to be fast
                const int fq = s_freq[sg];
                for (int q = a + tid; q < b; q += BLOCK) out[q] = sine(mod_sr((unsigned)fq * (unsigned)(q - lo), sr, inv_sr));
            } else {
                const int fq = s_freq[sg];
                for (int v = va + 4 * tid; v < vb; v += 4 * BLOCK) {
                    int m = mod_sr((unsigned)fq * (unsigned)(v - lo), sr, inv_sr);
                    T e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        e[k] = sine(m);
                        m += fq;
                        m = m >= sr ? m - sr : m;
                    }
                    V4 o;
                    o.x = e[0]; o.y = e[1]; o.z = e[2]; o.w = e[3];
                    *(V4 *)(out + v) = o;
                }
                if (tid < 8) {
                    const int pos = tid < 4 ? a + tid : vb + (tid - 4);
                    const int lim = tid < 4 ? va : b;
                    if (pos < lim) out[pos] = sine(mod_sr((unsigned)fq * (unsigned)(pos - lo), sr, inv_sr));
                }
            }
            __syncthreads();  // next tone visible / everyone is done with this one
            cur ^= 1;
            period = period_nx;
            sg = nx;
        }
        // past the frame: zero fill
        const int a = total > p0 ? total : p0;
        if (a < p1) {
            int va = a + ((4 - ((mis + a) & 3)) & 3);
            va = va > p1 ? p1 : va;
            const int vb = va + ((p1 - va) & ~3);
            V4 z = {};
            for (int v = va + 4 * tid; v < vb; v += 4 * BLOCK) *(V4 *)(out + v) = z;
            if (tid < 8) {
                const int pos = tid < 4 ? a + tid : vb + (tid - 4);
                const int lim = tid < 4 ? va : p1;
                if (pos < lim) out[pos] = (T)0;
            }
        }
    }
}

#define SYN_ARGS_DECL                                                                                                 \
    const int32_t *__restrict__ seg_start, const uint16_t *__restrict__ seg_freq, const int32_t *__restrict__ n_segs,   \
        int seg_cap, long long n_frames, int parts, int part_len, const long long *__restrict__ dst_off,               \
        const long long *__restrict__ fill_to, T *__restrict__ audio, long long audio_len, const T *__restrict__ g_tab, \
        int quarter, int tab_entries, double amp, double gain, double w_unit
#define SYN_ARGS_PASS                                                                                             \
    seg_start, seg_freq, n_segs, seg_cap, n_frames, parts, part_len, dst_off, fill_to, audio, audio_len, g_tab, quarter, \
        tab_entries, amp, gain, w_unit
template <typename T, bool TAB_LDS>
__global__ __launch_bounds__(SYN_BLOCK) void synth_kernel(SYN_ARGS_DECL) {
    synth_body<T, TAB_LDS, SYN_BLOCK, SYN_MAX_SEGS, SYN_MAX_STRIDE, false>(SYN_ARGS_PASS);
}
// the small-footprint instance, held to 40 registers: a SIMD whose one detection wave holds 464
// of its 512 registers has 48 left
template <typename T>
__global__ __launch_bounds__(SYN_LEAN_BLOCK) __attribute__((amdgpu_num_vgpr(40))) void synth_lean_kernel(SYN_ARGS_DECL) {
    synth_body<T, false, SYN_LEAN_BLOCK, SYN_LEAN_SEGS, SYN_LEAN_STRIDE, true>(SYN_ARGS_PASS);
}

// ---------------------------------------------------------------------------------------
// Sines computed on the GPU against the table the host built: one thread per table entry, both
// output formats.  Used once when a context is made (wn_synth_sines_agree).
// ---------------------------------------------------------------------------------------
__global__ void sine_check_kernel(const float *__restrict__ tab_f, const int16_t *__restrict__ tab_h, int quarter,
                                  double amp, double gain, double w_unit, unsigned *__restrict__ mismatches) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m > quarter) return;
    const double sn = sin_quarter(m, quarter, w_unit);
    double w = amp * sn;
    if (gain != 1.0) {
        w = w * gain;
        w = w < -1.0 ? -1.0 : (w > 1.0 ? 1.0 : w);
    }
    const float f = (float)w;
    const int16_t h = (int16_t)(amp * sn * 32767.0);
    if (f != tab_f[m] || h != tab_h[m]) atomicAdd(mismatches, 1u);
}

// WN_OK with *agree set: whether every sine the synthesis kernel would compute for this scheme (its
// sample rate, amplitude and gain) equals, after rounding to float32 and to int16, the entry of
// the quarter-wave table the host built with sin().  The tests pin that for the shipped schemes;
// for any other scheme this check decides at run time, and a context whose sines do not all agree
// takes its samples from the table instead (synth_table_lds).
int wn_synth_sines_agree(wn_ctx *ctx, bool *agree) {
    unsigned *d = nullptr, h = 0;
    WN_HIP_TRY(hipMalloc((void **)&d, sizeof(unsigned)));
    hipError_t e = hipMemset(d, 0, sizeof(unsigned));
    if (e == hipSuccess) {
        const int n = ctx->quarter + 1;
        hipLaunchKernelGGL(sine_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ctx->d_sin_f32,
                           ctx->d_sin_i16, ctx->quarter, ctx->s.amp, ctx->s.gain,
                           2.0 * M_PI / (double)ctx->s.sample_rate, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        wn_set_error("sine check failed: %s", hipGetErrorString(e));
        return WN_ERR_HIP;
    }
    *agree = h == 0;
    return WN_OK;
}

// ---------------------------------------------------------------------------------------
// Channel noise: sample i gets sigma * z(seed, i), z from splitmix64 bits by Box-Muller.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void awgn_kernel(float *__restrict__ audio, long long n, float sigma,
                                                   unsigned long long seed) {
    // one thread makes two normals (samples 2i, 2i+1) from one 64-bit draw
    const long long pairs = (n + 1) / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs;
         i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r = splitmix64(seed ^ splitmix64((unsigned long long)i));
        const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
        const float u2 = (float)(unsigned)((r >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
        const float rad = sigma * sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        audio[2 * i] += rad * cs;
        if (2 * i + 1 < n) audio[2 * i + 1] += rad * sn;
    }
}

// ---------------------------------------------------------------------------------------
extern "C" int wn_synth_layout(wn_ctx *ctx, const uint8_t *d_frames, const int32_t *d_lens, int32_t stride,
                               int64_t n, int32_t variant, int32_t seg_cap, int32_t *d_seg_start,
                               uint16_t *d_seg_freq, int32_t *d_n_segs, int64_t *d_n_samples, void *stream) {
    WN_REQUIRE(ctx && n >= 0 && stride > 0, "ctx / n >= 0 / stride > 0");
    WN_REQUIRE(variant == WN_EMIT_LIVE || variant == WN_EMIT_WAV, "variant");
    WN_REQUIRE(seg_cap >= 2 * stride + 4 && seg_cap <= SYN_MAX_SEGS, "2*stride+4 <= seg_cap <= 768");
    WN_REQUIRE(n == 0 || (d_frames && d_lens && d_seg_start && d_seg_freq && d_n_segs && d_n_samples),
               "null device pointer");
    if (n == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    const unsigned blocks = (unsigned)((n + 127) / 128);
    hipLaunchKernelGGL(synth_layout_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, ctx->s, d_frames,
                       d_lens, (int)stride, (long long)n, (int)variant, (int)seg_cap, d_seg_start, d_seg_freq,
                       d_n_segs, (long long *)d_n_samples);
    return wn_launch_check("synth_layout_kernel");
}

// the dynamic LDS of synth_kernel depends on the scheme's sample rate and is checked against
// 150 KB at launch: allow that much once
int wn_synth_device_setup(int) {
    WN_HIP_TRY(hipFuncSetAttribute((const void *)synth_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   150 * 1024));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)synth_kernel<int16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   150 * 1024));
    return WN_OK;
}

extern "C" int wn_synth_batch(wn_ctx *ctx, const int32_t *d_seg_start, const uint16_t *d_seg_freq,
                              const int32_t *d_n_segs, int32_t seg_cap, int64_t n, const int64_t *d_dst_off,
                              const int64_t *d_fill_to, int64_t slot_len, void *d_audio, int64_t audio_len,
                              int32_t fmt, void *stream) {
    WN_REQUIRE(ctx && n >= 0, "ctx / n >= 0");
    const bool beside = (fmt & WN_SYNTH_BESIDE) != 0;
    fmt &= ~WN_SYNTH_BESIDE;
    WN_REQUIRE(fmt == WN_FMT_F32 || fmt == WN_FMT_I16, "fmt");
    WN_REQUIRE(seg_cap > 0 && seg_cap <= SYN_MAX_SEGS, "0 < seg_cap <= 768");
    WN_REQUIRE(slot_len > 0 && slot_len < (1ll << 31) && audio_len > 0, "0 < slot_len < 2^31 / audio_len > 0");
    WN_REQUIRE(n == 0 || (d_seg_start && d_seg_freq && d_n_segs && d_dst_off && d_audio), "null device pointer");
    if (n == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    const size_t esz = fmt == WN_FMT_F32 ? 4 : 2;
    const size_t tab_entries = ((size_t)ctx->quarter + 1 + 31) & ~(size_t)31;
    // Tones of many periods are written from period tables built once per segment, and sines are
    // needed only to build those (two per thread and segment at most): they are then computed
    // where they are needed (sin_quarter), LDS holds no
    // quarter-wave table and four workgroups fit a CU.  Measured on 122 GB: 6.0 TB/s; with the
    // table read from global memory 5.7 (every load waits for the stores issued before it), with
    // the table in LDS and two workgroups per CU 5.7; plain stores in the same pattern reach 6.3.
    // Schemes with short tones take every sample from the quarter-wave table and keep it in LDS
    // (WN_SYNTH_TABLE=lds forces that for any scheme: the two must give the same samples).
    const bool tab_lds = ctx->s.tone_n < 16 * SYN_BLOCK || ctx->synth_table_lds;
    const size_t lds = syn_off_tab(SYN_MAX_SEGS) + ((tab_lds ? tab_entries : 0) + 2 * SYN_MAX_STRIDE) * esz;
    WN_REQUIRE(lds <= 150 * 1024, "sample_rate too large for the LDS sine table");
    // Work items are (frame, part) pairs, parts of 48 Ki samples (192 KB) when there is plenty of
    // work, down to 16 Ki for few frames; a workgroup takes every blocks-th item.  The grid is an
    // eighth of the items (at least what is resident at once: four workgroups per CU by LDS, two
    // by registers, the rest queue and start in order as others finish): the dispatcher hands
    // out the items in address order then, and the chip sweeps through the buffer instead of
    // writing a thousand far-apart regions at once.  Measured on 122 GB (the faster placement):
    // 1024 persistent workgroups 19.6 ms at parts of 768 Ki, 20.0 at 192 Ki, 22.5 at 48 Ki; a
    // quarter of the items 18.35 ms at 192 Ki, 17.7 at 48 Ki; an eighth of them 18.0 at 96 Ki, 17.3
    // at 48 Ki, 17.3-17.7 at 36 Ki; a sixteenth 17.65 at 48 Ki; smaller parts (24 Ki: 19 ms, 12 Ki:
    // 29 ms) pay too much set-up; one workgroup per item 21.3 ms (every workgroup of a round in
    // its set-up at the same time).
    // WN_SYNTH_PART / WN_SYNTH_PER_CU (workgroups per CU of the grid) override the choice for tuning.
    const long long resident = (long long)ctx->num_cus * (tab_lds ? 2 : 4);
    long long part_want = (n * slot_len / (resident * 4) + 4095) & ~4095ll;
    // a workgroup with the table in LDS loads 44 KB first: larger parts (20 000 prototype-scheme messages of
    // 282 240 samples: 4.54 ms at parts of 70 560, 4.73-4.83 at 35 280, 141 120 and 282 240)
    const long long part_max = tab_lds ? 73728 : 49152;
    part_want = part_want < 16384 ? 16384 : (part_want > part_max ? part_max : part_want);
    if (ctx->synth_part >= 4096) part_want = ctx->synth_part;
    long long parts = (slot_len + part_want - 1) / part_want;
    if (parts < 1) parts = 1;
    long long part_len = (slot_len + parts - 1) / parts;
    part_len = (part_len + 3) & ~3ll;
    parts = (slot_len + part_len - 1) / part_len;
    const long long items = n * parts;
    const long long share = tab_lds ? 4 : 8;
    long long blocks = items / share > resident ? items / share : resident;
    if (const int v = ctx->synth_per_cu)
        if (v >= 1 && v <= 4096) blocks = (long long)ctx->num_cus * v;
    if (blocks > items) blocks = items;
    hipStream_t st = (hipStream_t)stream;
    if (beside && !tab_lds && seg_cap <= SYN_LEAN_SEGS) {
        // The small-footprint instance (same samples): parts of 48 Ki samples, a grid of an eighth of
        // the items and at least one workgroup per CU, as above.
        long long lb = items / 8 > ctx->num_cus ? items / 8 : ctx->num_cus;
        if (lb > items) lb = items;
        const size_t llds = syn_off_tab(SYN_LEAN_SEGS) + 2 * SYN_LEAN_STRIDE * esz;
#define WN_SYNTH_LEAN(T, TABLE)                                                                                    \
    hipLaunchKernelGGL((synth_lean_kernel<T>),                                                                   \
                       dim3((unsigned)lb), dim3(SYN_LEAN_BLOCK), llds, st, d_seg_start, d_seg_freq, d_n_segs,      \
                       (int)seg_cap, (long long)n, (int)parts, (int)part_len, (const long long *)d_dst_off,        \
                       (const long long *)d_fill_to, (T *)d_audio, (long long)audio_len, (const T *)(TABLE),       \
                       ctx->quarter, (int)tab_entries, ctx->s.amp, ctx->s.gain, 2.0 * M_PI / (double)ctx->s.sample_rate)
        if (fmt == WN_FMT_F32) WN_SYNTH_LEAN(float, ctx->d_sin_f32);
        else WN_SYNTH_LEAN(int16_t, ctx->d_sin_i16);
#undef WN_SYNTH_LEAN
        return wn_launch_check("synth_kernel (small footprint)");
    }
#define WN_SYNTH_LAUNCH(T, TAB, TABLE)                                                                          \
    hipLaunchKernelGGL((synth_kernel<T, TAB>), dim3((unsigned)blocks), dim3(SYN_BLOCK), lds, st, d_seg_start,      \
                       d_seg_freq, d_n_segs, (int)seg_cap, (long long)n, (int)parts, (int)part_len,                \
                       (const long long *)d_dst_off, (const long long *)d_fill_to, (T *)d_audio,                   \
                       (long long)audio_len, (const T *)(TABLE), ctx->quarter, (int)tab_entries, ctx->s.amp, ctx->s.gain, 2.0 * M_PI / (double)ctx->s.sample_rate)
    if (fmt == WN_FMT_F32) {
        if (tab_lds) WN_SYNTH_LAUNCH(float, true, ctx->d_sin_f32);
        else WN_SYNTH_LAUNCH(float, false, ctx->d_sin_f32);
    } else {
        if (tab_lds) WN_SYNTH_LAUNCH(int16_t, true, ctx->d_sin_i16);
        else WN_SYNTH_LAUNCH(int16_t, false, ctx->d_sin_i16);
    }
#undef WN_SYNTH_LAUNCH
    return wn_launch_check("synth_kernel");
}

extern "C" int wn_channel_awgn(wn_ctx *ctx, float *d_audio, int64_t n, float sigma, uint64_t seed, void *stream) {
    WN_REQUIRE(ctx && n >= 0 && sigma >= 0.0f, "ctx / n >= 0 / sigma >= 0");
    WN_REQUIRE(n == 0 || d_audio, "null device pointer");
    if (n == 0 || sigma == 0.0f) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    const long long pairs = (n + 1) / 2;
    long long blocks = (pairs + 255) / 256;
    const long long cap = (long long)ctx->num_cus * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(awgn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_audio,
                       (long long)n, sigma, (unsigned long long)seed);
    return wn_launch_check("awgn_kernel");
}
