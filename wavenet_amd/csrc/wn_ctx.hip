// Context creation: host-side tables (class LUT, sine tables, DFT twiddles, CRC tables)
// and their upload.  No kernels here.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <numeric>
#include <vector>

#include "wn_common.h"

static thread_local char g_err[512] = "";

void wn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *wn_last_error(void) { return g_err; }

extern "C" const char *wn_strerror(int code) {
    switch (code) {
        case WN_OK: return "ok";
        case WN_ERR_ARG: return "bad argument";
        case WN_ERR_HIP: return "HIP runtime error";
        case WN_ERR_UNSUPPORTED: return "scheme not supported by the kernels";
        case WN_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

// ---------------------------------------------------------------------------------------
// Class LUT: the decision of the receive loop as a function of the dominant bin.
// freq = |fftfreq(n, 1/sr)[k]| = k * (1/(n*(1/sr)))  (c1_shared.py:296-299, NumPy's
// fftfreq multiplies the integer index by val = 1.0/(n*d)).
// ---------------------------------------------------------------------------------------
static int scheme_ok(const wn_scheme *s) {
    if (!s) return 0;
    if (s->sample_rate <= 0 || s->sample_rate % 4 != 0 || s->sample_rate > 160000) return 0;
    if (s->freq_step <= 0 || s->base_freq < 0 || s->max_byte < 0 || s->max_byte > 255) return 0;
    if (s->win_n < 16 || s->tone_n <= 0 || s->eof_n <= 0) return 0;
    if (s->base_freq + 255 * s->freq_step >= 65536) return 0;
    if (s->sep_freq < 0 || s->ping_freq < 0 || s->eof_freq < 0) return 0;
    if (s->sep_freq > 0 && (s->sep_n <= 0 || s->sep_short_n <= 0)) return 0;
    if (s->ping_freq > 0 && s->ping_n <= 0) return 0;
    return 1;
}

extern "C" int wn_host_class_lut(const wn_scheme *s, uint16_t *out, int32_t cap) {
    if (!scheme_ok(s) || !out) {
        wn_set_error("bad argument: scheme/out");
        return WN_ERR_ARG;
    }
    const int32_t n = s->win_n, nb = n / 2;
    if (cap < nb) {
        wn_set_error("bad argument: cap < win_n/2");
        return WN_ERR_ARG;
    }
    const double d = 1.0 / (double)s->sample_rate;
    const double val = 1.0 / ((double)n * d);
    for (int32_t k = 0; k < nb; ++k) {
        const double f = fabs((double)k * val);
        // freq_to_byte: int(round((f - base)/step)), round-half-even like Python/NumPy
        const double q = (f - (double)s->base_freq) / (double)s->freq_step;
        const double r = nearbyint(q);
        uint16_t c = WN_CLS_NONE;
        if (r >= 0.0 && r <= (double)s->max_byte) {
            const long b = (long)r;
            if (!s->tol_check) {
                c = (uint16_t)b;
            } else {
                const double fc = (double)(s->base_freq + b * s->freq_step);
                if (fabs(f - fc) < (double)s->freq_step / 2.0) c = (uint16_t)b;
            }
        }
        if (s->ping_freq > 0) {  // the scheme with a receive loop (c1_shared.py:331, 337)
            if (fabs(f - (double)s->ping_freq) < 10.0) c |= WN_CLS_PING;
            if (fabs(f - (double)s->eof_freq) < 10.0) c |= WN_CLS_EOF;
        }
        out[k] = c;
    }
    return nb;
}

// ---------------------------------------------------------------------------------------
// DFT plan: the detection kernel is instantiated for these (n1, n2) pairs.
// ---------------------------------------------------------------------------------------
static const int32_t kPlans[][2] = {{83, 85}, {45, 49}};

int wn_plan_lookup(int32_t n, int32_t *n1, int32_t *n2) {
    for (auto &p : kPlans)
        if ((int64_t)p[0] * p[1] == n) {
            *n1 = p[0];
            *n2 = p[1];
            return WN_OK;
        }
    wn_set_error("window length %d has no built factorisation (built: 83x85, 45x49)", n);
    return WN_ERR_UNSUPPORTED;
}

extern "C" int wn_host_dft_plan(const wn_scheme *s, int32_t *n1, int32_t *n2) {
    if (!scheme_ok(s) || !n1 || !n2) {
        wn_set_error("bad argument: scheme/n1/n2");
        return WN_ERR_ARG;
    }
    return wn_plan_lookup(s->win_n, n1, n2);
}

static int32_t inv_mod(int32_t a, int32_t m) {
    a %= m;
    for (int32_t x = 1; x < m; ++x)
        if ((int64_t)a * x % m == 1) return x;
    return 0;
}

// ---------------------------------------------------------------------------------------
// Sine tables: every tone frequency is an integer number of Hz and every tone starts at
// t = 0, so sample i of a tone of f Hz has phase 2*pi*((f*i) mod sr)/sr exactly; one
// quarter period of sr entries covers all of them.
// ---------------------------------------------------------------------------------------
extern "C" int wn_host_sine_tables(const wn_scheme *s, float *out_f32, int16_t *out_i16, int32_t cap) {
    if (!scheme_ok(s)) {
        wn_set_error("bad argument: scheme");
        return WN_ERR_ARG;
    }
    const int32_t q = s->sample_rate / 4;
    if (cap < q + 1) {
        wn_set_error("bad argument: cap < sample_rate/4+1");
        return WN_ERR_ARG;
    }
    for (int32_t m = 0; m <= q; ++m) {
        const double sn = (m == q) ? 1.0 : sin(2.0 * M_PI * (double)m / (double)s->sample_rate);
        if (out_f32) {
            double v = s->amp * sn;  // 0.5*np.sin(...)   (c1_shared.py:391)
            if (s->gain != 1.0) {
                v = v * s->gain;     // tono*1.95          (c1_shared.py:392)
                v = v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
            }
            out_f32[m] = (float)v;
        }
        if (out_i16) out_i16[m] = (int16_t)(s->amp * sn * 32767.0);  // np.int16(x*32767)
    }
    return q + 1;
}

extern "C" int64_t wn_host_frame_samples(const wn_scheme *s, const uint8_t *frame, int32_t len, int32_t variant) {
    if (!scheme_ok(s) || (len > 0 && !frame) || len < 0) return WN_ERR_ARG;
    int64_t t = 0;
    if (s->sep_freq == 0) {  // prototype scheme: one tone per byte, then the end tone
        t = (int64_t)len * s->tone_n;
        if (variant == WN_EMIT_LIVE) t += s->eof_n;
        return t;
    }
    if (variant == WN_EMIT_LIVE) {
        t = s->ping_n;
        for (int32_t i = 0; i < len; ++i) {
            t += (i > 0 && frame[i] == frame[i - 1]) ? s->sep_short_n : s->sep_n;
            t += s->tone_n;
        }
        t += s->sep_n + s->eof_n;
    } else {
        t = 2 * (int64_t)s->sep_n + (int64_t)len * (s->tone_n + s->sep_n) + s->tone_n + s->sep_n;
    }
    return t;
}

void wn_host_crc_tables(uint32_t tab[4][256]) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 4; ++t) tab[t][i] = (tab[t - 1][i] >> 8) ^ tab[0][tab[t - 1][i] & 0xFF];
}

// ---------------------------------------------------------------------------------------
template <typename T>
static int upload(T **dptr, const T *host, size_t count) {
    WN_HIP_TRY(hipMalloc((void **)dptr, count * sizeof(T)));
    WN_HIP_TRY(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    return WN_OK;
}

int wn_device_check(const wn_ctx *ctx) {
    int dev = -1;
    WN_HIP_TRY(hipGetDevice(&dev));
    if (dev != ctx->device || dev < 0 || dev >= WN_MAX_DEVICES) {
        wn_set_error("bad argument: the context belongs to device %d, the current device is %d", ctx->device, dev);
        return WN_ERR_ARG;
    }
    return WN_OK;
}

extern "C" int wn_ctx_destroy(wn_ctx *ctx) {
    if (!ctx) return WN_OK;
    (void)hipFree(ctx->d_sin_f32);
    (void)hipFree(ctx->d_sin_i16);
    (void)hipFree(ctx->d_cls_lut);
    (void)hipFree(ctx->d_cand);
    (void)hipFree(ctx->d_tw1);
    (void)hipFree(ctx->d_tw2);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_crc_tab);
    (void)hipFree(ctx->d_crc_combine_tab);
    free(ctx);
    return WN_OK;
}

extern "C" int wn_scheme_init(const wn_scheme *s, wn_ctx **out) {
    if (!out) {
        wn_set_error("bad argument: out");
        return WN_ERR_ARG;
    }
    *out = nullptr;
    if (!scheme_ok(s)) {
        wn_set_error("bad argument: scheme fields out of range");
        return WN_ERR_ARG;
    }
    int32_t n1 = 0, n2 = 0;
    int rc = wn_plan_lookup(s->win_n, &n1, &n2);
    if (rc != WN_OK) return rc;

    wn_ctx *ctx = (wn_ctx *)calloc(1, sizeof(wn_ctx));
    if (!ctx) return WN_ERR_NOMEM;
    ctx->s = *s;
    ctx->n = s->win_n;
    ctx->n1 = n1;
    ctx->n2 = n2;
    ctx->nbins = s->win_n / 2;
    ctx->e1 = n2 * inv_mod(n2, n1);
    ctx->e2 = n1 * inv_mod(n1, n2);
    ctx->quarter = s->sample_rate / 4;
    ctx->tie_band = 0.0;

#define INIT_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != WN_OK) {       \
            wn_ctx_destroy(ctx);  \
            return rc_;           \
        }                         \
    } while (0)
#define INIT_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            wn_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
            wn_ctx_destroy(ctx);                                                    \
            return WN_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

    INIT_HIP(hipGetDevice(&ctx->device));
    if (ctx->device >= WN_MAX_DEVICES) {
        wn_set_error("device index %d not supported", ctx->device);
        wn_ctx_destroy(ctx);
        return WN_ERR_UNSUPPORTED;
    }
    hipDeviceProp_t prop;
    INIT_HIP(hipGetDeviceProperties(&prop, ctx->device));
    ctx->num_cus = prop.multiProcessorCount;

    {
        std::vector<float> f(ctx->quarter + 1);
        std::vector<int16_t> h(ctx->quarter + 1);
        wn_host_sine_tables(s, f.data(), h.data(), ctx->quarter + 1);
        INIT_TRY(upload(&ctx->d_sin_f32, f.data(), f.size()));
        INIT_TRY(upload(&ctx->d_sin_i16, h.data(), h.size()));
    }
    {
        std::vector<uint16_t> lut(ctx->nbins);
        wn_host_class_lut(s, lut.data(), ctx->nbins);
        INIT_TRY(upload(&ctx->d_cls_lut, lut.data(), lut.size()));
        ctx->ping_lo = ctx->ping_hi = -1;
        for (int32_t k = 0; k < ctx->nbins; ++k)
            if (lut[k] & WN_CLS_PING) {
                if (ctx->ping_lo < 0) ctx->ping_lo = k;
                ctx->ping_hi = k;
            }
    }
    {
        std::vector<float2> t1(n1);
        std::vector<float2> t2(n2);
        for (int32_t m = 0; m < n1; ++m) {
            const double a = 2.0 * M_PI * (double)m / (double)n1;
            t1[m] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int32_t m = 0; m < n2; ++m) {
            const double a = 2.0 * M_PI * (double)m / (double)n2;
            t2[m] = make_float2((float)cos(a), (float)(-sin(a)));
        }
        INIT_TRY(upload(&ctx->d_tw1, t1.data(), t1.size()));
        INIT_TRY(upload(&ctx->d_tw2, t2.data(), t2.size()));
    }
    {
        // The scheme's tones (bytes, ping, end; the separator has a test of its own) sorted by the spectrum
        // column their nearest bin falls in, k mod n2 folded, with the row frequency k mod n1 folded: what
        // the detection kernel needs to name the one bin a window's energy may sit in.  Two tones of one
        // column with the same folded row frequency cannot be told apart there and are both left out.
        const int K2 = n2 / 2 + 1;
        std::vector<int32_t> cand((size_t)K2 * 8, 0);
        auto add = [&](long f) {
            if (f <= 0) return;
            const long k = (f * ctx->n * 2 + s->sample_rate) / (2l * s->sample_rate);
            if (k <= 0 || k >= ctx->n / 2 || k >= 65536) return;
            int c = (int)(k % n2), r = (int)(k % n1);
            c = c < n2 - c ? c : n2 - c;
            r = r < n1 - r ? r : n1 - r;
            int32_t *row = &cand[(size_t)c * 8];
            for (int j = 0; j < 8; ++j) {
                if (row[j] == 0) {
                    row[j] = (int32_t)k | (r << 16);
                    return;
                }
                if ((row[j] & 0xFFFF) == k) return;              // the same bin again
                if ((row[j] >> 16) == r && row[j] != (r << 16)) {
                    row[j] = r << 16;                             // keeps the place, names no bin
                    return;
                }
                if (row[j] == (r << 16)) return;
            }
        };
        for (int b = 0; b <= s->max_byte; ++b) add((long)s->base_freq + (long)s->freq_step * b);
        add(s->ping_freq);
        add(s->eof_freq);
        INIT_TRY(upload(&ctx->d_cand, cand.data(), cand.size()));
    }
    {
        static uint32_t tab[4][256];
        wn_host_crc_tables(tab);
        INIT_TRY(upload(&ctx->d_crc_tab, &tab[0][0], (size_t)4 * 256));
    }
    {
        std::vector<uint32_t> t(wn_host_crc_combine_words());
        wn_host_crc_combine_tables(t.data());
        INIT_TRY(upload(&ctx->d_crc_combine_tab, t.data(), t.size()));
    }
    if (const char *e = getenv("WN_CRC_COMBINE")) ctx->crc_combine = atoi(e) != 0;
    INIT_HIP(hipMalloc((void **)&ctx->d_stats, 16 * sizeof(unsigned long long)));
    INIT_HIP(hipMemset(ctx->d_stats, 0, 16 * sizeof(unsigned long long)));
    if ((size_t)(ctx->n + n1 + n2) * sizeof(double2) > 150 * 1024) {
        wn_set_error("window length %d too large for the float64 phase table", ctx->n);
        wn_ctx_destroy(ctx);
        return WN_ERR_UNSUPPORTED;
    }
    if (const char *e = getenv("WN_DETECT_PROFILE")) ctx->detect_profile = atoi(e) == 2 ? 2 : (atoi(e) != 0);   // 2: clocks per window and way only
    ctx->detect_share = 0.7f;
    if (const char *e = getenv("WN_DETECT_SHARE")) {
        const float v = (float)atof(e);
        if (v > 0.34f && v <= 1.0f) ctx->detect_share = v;
    }
    // The weakest of three column tiles is left out when the pilot finds at most this share in it.  On
    // the round trip: 29.4 ms at 0 (never), 28.7-30.0 at 0.1, 28.04 at 0.2, 28.01 at 0.3, and from 0.25 on
    // white noise (shares of 37, 37 and 26 % give or take two points) tries two tiles in vain.
    ctx->detect_two_share = 0.2f;
    if (const char *e = getenv("WN_DETECT_TWO")) {
        const float v = (float)atof(e);
        if (v >= 0.0f && v < 0.34f) ctx->detect_two_share = v;
    }
    // the reference's bin of a frequency f is the nearest one: round(f * N / sample_rate)
    ctx->sep_bin = s->sep_freq > 0 ? (int)(((long long)s->sep_freq * ctx->n * 2 + s->sample_rate) / (2ll * s->sample_rate)) : 0;
    // The separator's bin is tried alone when the pilot finds this share of its energy in that bin's column.
    // On the round trip: 25.10 / 24.35 / 24.52 / 24.66 / 24.91 / 25.02 ms at 0.4 / 0.5 / 0.6 / 0.7 / 0.8 / 0.9
    // (26.0 without); at 0.4 six windows in a hundred come back undecided and start again.
    ctx->detect_sep_share = 0.5f;
    if (const char *e = getenv("WN_DETECT_SEP")) {
        const float v = (float)atof(e);
        if (v >= 0.0f && v <= 1.0f) ctx->detect_sep_share = v;
    }
    // A scheme tone's bin is tried alone when the pilot finds this share of its energy in one column that holds
    // such tones.  On the round trip: 23.78 / 22.82 / 22.94 / 23.21 / 23.46 ms at 0.4 / 0.5 / 0.6 / 0.7 / 0.8 (24.0
    // without); at 0.4 one window in six comes back undecided, at 0.5 one in twenty-two.
    ctx->detect_bin_share = 0.55f;
    if (const char *e = getenv("WN_DETECT_BIN")) {
        const float v = (float)atof(e);
        if (v >= 0.0f && v <= 1.0f) ctx->detect_bin_share = v;
    }
    ctx->detect_pick_share = 1.0f / 128.0f;
    if (const char *e = getenv("WN_DETECT_PICK")) {
        const float v = (float)atof(e);
        if (v >= 0.0f && v <= 0.25f) ctx->detect_pick_share = v;
    }
    if (const char *e = getenv("WN_DETECT_PER_CU")) ctx->detect_per_cu = atoi(e);
    if (const char *e = getenv("WN_SYNTH_PER_CU")) ctx->synth_per_cu = atoi(e);
    if (const char *e = getenv("WN_SYNTH_TABLE")) ctx->synth_table_lds = e[0] == 'l';
    if (const char *e = getenv("WN_SYNTH_PART")) ctx->synth_part = atoll(e);
    {
        // everything that is per device and process-wide happens here, once, under a lock:
        // afterwards launches from any number of threads only read
        static std::mutex mu;
        static bool ready[WN_MAX_DEVICES];
        std::lock_guard<std::mutex> lock(mu);
        if (!ready[ctx->device]) {
            INIT_TRY(wn_detect_device_setup(ctx->device));
            INIT_TRY(wn_sync_device_setup(ctx->device));
            INIT_TRY(wn_synth_device_setup(ctx->device));
            INIT_TRY(wn_crc_device_setup(ctx->device));
            ready[ctx->device] = true;
        }
    }
    if (!ctx->synth_table_lds) {
        // the synthesis of long-tone schemes computes its sines on the GPU: only if every one of them
        // equals the host's table entry (pinned by tests for the shipped schemes, checked here for any)
        bool agree = false;
        INIT_TRY(wn_synth_sines_agree(ctx, &agree));
        if (!agree) ctx->synth_table_lds = 1;
    }
    *out = ctx;
    return WN_OK;
}

extern "C" int wn_detect_set_tie_band(wn_ctx *ctx, double rel) {
    WN_REQUIRE(ctx && rel >= 0.0 && rel < 1.0, "ctx / 0 <= rel < 1");
    ctx->tie_band = rel;
    return WN_OK;
}

extern "C" int wn_detect_stats(wn_ctx *ctx, int64_t out[12], void *stream) {
    WN_REQUIRE(ctx && out, "ctx/out");
    unsigned long long h[12];
    hipStream_t st = (hipStream_t)stream;
    WN_HIP_TRY(hipMemcpyAsync(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost, st));
    WN_HIP_TRY(hipMemsetAsync(ctx->d_stats, 0, sizeof(h), st));
    WN_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < 12; ++i) out[i] = (int64_t)h[i];
    return WN_OK;
}
