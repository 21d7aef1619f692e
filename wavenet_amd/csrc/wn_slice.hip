// Receive state machine over window classes.
//
// Replaces the loop of escuchar_y_retornar_trama (c1_shared.py:312-351): wait for a
// ping-class window (B:330-335, which also restarts the timeout clock), stop at an
// EOF-class window (B:337-339), take a byte-class window as a symbol only after a
// separator (non-byte) window was seen (expect_silence, B:341-351).  Calls follow each
// other over one capture the way repeated listen calls consume one audio device: the
// next call starts at the window after the previous call's last window.
//
// For the prototype scheme (no separators, capa1.py has no receive loop) every window is
// one symbol and a stream yields one record with the byte-class windows in order.
//
// The machine looks sequential, but between a ping and the EOF it only asks, per window,
// "is this a byte window and was the previous one not": an edge detection on a bit mask, which
// does not depend on where frames start, except for the first window of a frame (which counts as
// preceded by a byte window).  So the work splits into three phases (slice_scan_kernel):
//   1. every wave, all chunks of 64 windows at once: ballots give the ping, EOF and byte masks of a
//      chunk; the symbol mask is byte & ~(byte << 1 | last bit of the chunk before); scans over the
//      chunks give, per chunk, the number of symbols before it and the first ping / first EOF at or
//      after it;
//   2. one wave goes from call to call in a few dozen instructions each: first ping at or after the
//      start within the timeout, first EOF after it within the timeout, number of symbols between
//      them from the prefix counts;
//   3. every wave again: the symbols of each call are placed, one lane per window.
// A one-hour capture (22 503 windows, 52 listen calls) took 4.3 ms with one lane stepping through
// the windows and 0.2 ms with one wave taking 64 windows per step (a lone wave issues an
// instruction every few clocks, wherever the class words come from).
// Captures too long for the tables in LDS (98 304 windows) take slice_wave_kernel, the same
// machine run by one wave, 64 windows per step, over class words staged in LDS.
// (The prototype scheme, one symbol per window, keeps one lane per stream: symbols_kernel.)
#include "wn_common.h"

// Prototype scheme: every byte-class window is one symbol; one lane per stream, one record.
__global__ __launch_bounds__(64) void symbols_kernel(const uint16_t *__restrict__ cls, long long n_streams,
                                                     long long n_windows, int max_calls, int frame_cap,
                                                     uint8_t *__restrict__ out_bytes, int32_t *__restrict__ out_lens,
                                                     int32_t *__restrict__ out_start, int32_t *__restrict__ out_end,
                                                     int32_t *__restrict__ out_reason, int32_t *__restrict__ n_calls) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const uint16_t *c = cls + s * n_windows;
    uint8_t *ob = out_bytes + (s * max_calls) * (long long)frame_cap;
    int len = 0, over = 0;
    for (long long w = 0; w < n_windows; ++w) {
        const int b = c[w] & WN_CLS_BYTE_MASK;
        if (b == WN_CLS_NONE) continue;
        if (len < frame_cap) ob[len++] = (uint8_t)b; else over = WN_END_OVERFLOW;
    }
    out_lens[s * max_calls] = len;
    out_start[s * max_calls] = 0;
    out_end[s * max_calls] = (int32_t)n_windows;
    out_reason[s * max_calls] = WN_END_TIMEOUT | over;
    n_calls[s] = 1;
}

// The machine run by one wave, for captures of any length: all THREADS/64 waves stage 16*THREADS class
// words in LDS at a time, wave 0 runs the machine over them, 64 windows per step.
// The machine is written as one loop over pieces of at most 64 windows with its state in
// wave-uniform registers, so that it can stop at the end of a stage and go on in the next one.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void slice_wave_kernel(const uint16_t *__restrict__ cls, long long n_windows,
                                                             int timeout_windows, int max_calls, int frame_cap,
                                                             uint8_t *__restrict__ out_bytes,
                                                             int32_t *__restrict__ out_lens,
                                                             int32_t *__restrict__ out_start,
                                                             int32_t *__restrict__ out_end,
                                                             int32_t *__restrict__ out_reason,
                                                             int32_t *__restrict__ n_calls) {
    constexpr int SC = 16 * THREADS;
    __shared__ uint16_t stage[SC];
    __shared__ int stop;
    const long long s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool machine = tid < 64;                            // wave 0
    const unsigned long long below = (1ull << lane) - 1ull;   // lanes before this one
    const uint16_t *c = cls + s * n_windows;
    // state of the machine: the call in work, whether its ping has been heard, the window at which
    // it gives up, what it has collected
    int calls = 0, len = 0, over = 0;
    long long w = 0, start_w = 0;
    long long limit = min((long long)timeout_windows, n_windows);
    bool heard = false, finished = false;
    unsigned long long carry = 1ull;
    auto end_call = [&](int reason) {     // true: nothing more to do
        const long long rec = s * max_calls + calls;
        if (lane == 0) {
            out_lens[rec] = len;
            out_start[rec] = (int32_t)start_w;
            out_end[rec] = (int32_t)w;
            out_reason[rec] = reason | over;
        }
        ++calls;
        heard = false;
        start_w = w;
        len = over = 0;
        limit = min(w + (long long)timeout_windows, n_windows);
        return w >= n_windows || calls >= max_calls;
    };
    for (long long base = 0; base < n_windows; base += SC) {
        const long long sc_end = min(base + (long long)SC, n_windows);
        const int cnt = (int)(sc_end - base);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + k * THREADS;
            if (i < cnt) stage[i] = c[base + i];
        }
        __syncthreads();
        if (machine && !finished) {
            for (;;) {
                if (w >= limit) {                 // timeout, or the capture is exhausted
                    if ((finished = end_call(WN_END_TIMEOUT))) break;
                    continue;
                }
                if (w >= sc_end) break;           // goes on in the next stage
                const int nv = (int)min(64ll, min(limit, sc_end) - w);
                const int cw = lane < nv ? stage[(int)(w - base) + lane] : 0;
                if (!heard) {
                    // before the ping: the first ping-class window within the timeout opens the frame
                    const unsigned long long pm = __ballot((cw & WN_CLS_PING) != 0);
                    if (pm) {
                        w += __ffsll((long long)pm);     // the ping window is consumed
                        heard = true;
                        // after the ping the clock restarts; expect_silence starts true
                        limit = min(w + (long long)timeout_windows, n_windows);
                        carry = 1ull;
                    } else {
                        w += nv;
                    }
                    continue;
                }
                // a byte window is a symbol iff the window before it (since the ping) was not a byte
                // window; an EOF-class window ends the frame
                const int b = cw & WN_CLS_BYTE_MASK;
                const unsigned long long em = __ballot(lane < nv && (cw & WN_CLS_EOF) != 0);
                const unsigned long long bm = __ballot(lane < nv && b != WN_CLS_NONE);
                const int upto = em ? __ffsll((long long)em) - 1 : nv;   // windows of this piece before the EOF
                const unsigned long long live = upto >= 64 ? ~0ull : (1ull << upto) - 1ull;
                const unsigned long long sym = bm & ~((bm << 1) | carry) & live;
                if ((sym >> lane) & 1ull) {
                    const int pos = len + __popcll(sym & below);
                    if (pos < frame_cap) out_bytes[(s * max_calls + calls) * (long long)frame_cap + pos] = (uint8_t)b;
                }
                const int total = len + __popcll(sym);
                if (total > frame_cap) over = WN_END_OVERFLOW;
                len = total < frame_cap ? total : frame_cap;
                if (em) {
                    w += upto + 1;               // the EOF window is consumed
                    if ((finished = end_call(WN_END_EOF))) break;
                } else {
                    carry = (bm >> (nv - 1)) & 1ull;
                    w += nv;
                }
            }
        }
        if constexpr (THREADS > 64) {
            if (tid == 0) stop = finished;
            __syncthreads();                      // also: the stage is no longer read
            if (stop) break;
        } else {
            if (finished) break;
        }
    }
    if (tid == 0) n_calls[s] = calls;
}

// The three-phase form (see the top of the file).  THREADS = 64 with room for 32 chunks (many short
// streams, one wave each) or 1024 with room for 1536 chunks (long captures, one workgroup each).
template <int THREADS, int NCH>
__global__ __launch_bounds__(THREADS) void slice_scan_kernel(const uint16_t *__restrict__ cls, long long n_windows,
                                                             int timeout_windows, int max_calls, int frame_cap,
                                                             uint8_t *__restrict__ out_bytes,
                                                             int32_t *__restrict__ out_lens,
                                                             int32_t *__restrict__ out_start,
                                                             int32_t *__restrict__ out_end,
                                                             int32_t *__restrict__ out_reason,
                                                             int32_t *__restrict__ n_calls) {
    constexpr int WAVES = THREADS / 64, CT = 128, INF = 0x7fffffff;
    __shared__ unsigned long long pm[NCH], em[NCH], sym[NCH + 1];   // sym first holds the byte masks
    __shared__ int csum[NCH + 1], nping[NCH + 1], neof[NCH + 1];
    __shared__ int4 table[CT];     // calls of a batch: first window after the ping, end of the symbols, rank to subtract, record
    __shared__ int batch_n, all_done;
    const long long s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint16_t *c = cls + s * n_windows;
    const int n = (int)n_windows, nch = (n + 63) >> 6;
    // ---- phase 1: masks per chunk; the class words of eight chunks are requested before the first
    // ballot waits for them
    for (int k0 = wave; k0 < nch; k0 += 8 * WAVES) {
        int cw[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int x = 64 * (k0 + q * WAVES) + lane;
            cw[q] = x < n ? c[x] : WN_CLS_NONE;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q * WAVES;
            const unsigned long long p = __ballot((cw[q] & WN_CLS_PING) != 0);
            const unsigned long long e = __ballot((cw[q] & WN_CLS_EOF) != 0);
            const unsigned long long b = __ballot((cw[q] & WN_CLS_BYTE_MASK) != WN_CLS_NONE);
            if (lane == 0 && k < nch) {
                pm[k] = p;
                em[k] = e;
                sym[k] = b;
            }
        }
    }
    __syncthreads();
    // byte masks -> symbol masks (a byte window after a window that is not one); every thread keeps
    // its chunks' results in registers until all have read their neighbours
    {
        unsigned long long mine[(NCH + THREADS - 1) / THREADS];
#pragma unroll
        for (int q = 0; q < (NCH + THREADS - 1) / THREADS; ++q) {
            const int k = tid + q * THREADS;
            if (k < nch) {
                const unsigned long long b = sym[k];
                mine[q] = b & ~((b << 1) | (k > 0 ? sym[k - 1] >> 63 : 0ull));
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (NCH + THREADS - 1) / THREADS; ++q) {
            const int k = tid + q * THREADS;
            if (k < nch) sym[k] = mine[q];
        }
        if (tid == 0) sym[nch] = 0ull;
    }
    __syncthreads();
    // scans over the chunks, one wave each (the same wave does all three when there is only one):
    // symbols before a chunk; first ping / first EOF at or after a chunk's first window
    if (wave == 0) {
        int carry = 0;
        for (int g = 0; g <= nch; g += 64) {
            const int k = g + lane;
            const int v = k < nch ? __popcll(sym[k]) : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (k <= nch) csum[k] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
    if (wave == (1 % WAVES) || wave == (2 % WAVES)) {
        for (int which = 0; which < 2; ++which) {
            if (wave != ((1 + which) % WAVES)) continue;
            const unsigned long long *mk = which ? em : pm;
            int *nx = which ? neof : nping;
            int carry = INF;
            for (int g = ((nch - 1) >> 6) << 6; g >= 0; g -= 64) {
                const int k = g + lane;
                const unsigned long long m = k < nch ? mk[k] : 0ull;
                int v = m ? 64 * k + __ffsll((long long)m) - 1 : INF;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_down(v, d);
                    if (lane + d < 64) v = min(v, o);
                }
                v = min(v, carry);
                if (k < nch) nx[k] = v;
                carry = __shfl(v, 0);
            }
            if (lane == 0) nx[nch] = INF;
        }
    }
    __syncthreads();
    // ---- phases 2 and 3 in batches of CT calls
    int calls = 0, w = 0;
    auto rank = [&](int x) { return csum[x >> 6] + __popcll(sym[x >> 6] & ((1ull << (x & 63)) - 1ull)); };
    for (;;) {
        if (wave == 0) {
            int nb = 0;
            while (w < n && calls < max_calls && nb < CT) {
                const long long rec = s * max_calls + calls;
                const int start_w = w;
                int limit = (int)min((long long)w + timeout_windows, (long long)n);
                // the first ping-class window within the timeout opens the frame and is consumed
                int k = w >> 6;
                unsigned long long m = pm[k] & (~0ull << (w & 63));
                const int p = m ? 64 * k + __ffsll((long long)m) - 1 : nping[k + 1];
                int len = 0, reason = WN_END_TIMEOUT;
                if (p >= limit) {
                    w = limit;
                } else {
                    // after the ping the clock restarts; an EOF-class window ends the frame and is consumed
                    const int a = p + 1;
                    limit = (int)min((long long)a + timeout_windows, (long long)n);
                    int e = INF;
                    if (a < limit) {
                        k = a >> 6;
                        m = em[k] & (~0ull << (a & 63));
                        e = m ? 64 * k + __ffsll((long long)m) - 1 : neof[k + 1];
                    }
                    const bool eof = e < limit;
                    const int end_x = eof ? e : limit;          // symbols come from windows [a, end_x)
                    // the first window of a frame counts as preceded by a byte window
                    const int drop = a < end_x ? (int)((sym[a >> 6] >> (a & 63)) & 1ull) : 0;
                    const int r0 = rank(a) + drop;
                    const int cnt = a < end_x ? rank(end_x) - r0 : 0;
                    len = cnt < frame_cap ? cnt : frame_cap;
                    reason = (eof ? WN_END_EOF : WN_END_TIMEOUT) | (cnt > frame_cap ? WN_END_OVERFLOW : 0);
                    w = eof ? e + 1 : limit;
                    if (cnt > 0) {
                        if (lane == 0) table[nb] = make_int4(a, end_x, r0, calls);
                        ++nb;
                    }
                }
                if (lane == 0) {
                    out_lens[rec] = len;
                    out_start[rec] = start_w;
                    out_end[rec] = w;
                    out_reason[rec] = reason;
                }
                ++calls;
            }
            if (lane == 0) {
                batch_n = nb;
                all_done = !(w < n && calls < max_calls);
            }
        }
        __syncthreads();
        const int nb = batch_n;
        const bool done = all_done != 0;
        for (int j = wave; j < nb; j += WAVES) {
            const int4 t = table[j];
            uint8_t *ob = out_bytes + (s * max_calls + t.w) * (long long)frame_cap;
            for (int k = t.x >> 6; k <= (t.y - 1) >> 6; ++k) {
                const int x = 64 * k + lane;
                const unsigned long long sk = sym[k];
                const int pos = csum[k] + __popcll(sk & below) - t.z;
                if (((sk >> lane) & 1ull) && x >= t.x && x < t.y && pos >= 0 && pos < frame_cap)
                    ob[pos] = (uint8_t)(c[x] & WN_CLS_BYTE_MASK);
            }
        }
        if (done) break;
        __syncthreads();
    }
    if (tid == 0) n_calls[s] = calls;
}

extern "C" int wn_slice_frames(wn_ctx *ctx, const uint16_t *d_cls, int64_t n_streams, int64_t n_windows,
                               int32_t timeout_windows, int32_t max_calls, int32_t frame_cap, uint8_t *d_bytes,
                               int32_t *d_lens, int32_t *d_start, int32_t *d_end, int32_t *d_reason,
                               int32_t *d_n_calls, void *stream) {
    WN_REQUIRE(ctx && n_streams >= 0 && n_windows >= 0 && n_windows < (1ll << 31), "ctx / sizes");
    WN_REQUIRE(timeout_windows > 0 && max_calls > 0 && frame_cap > 0, "timeout_windows, max_calls, frame_cap > 0");
    WN_REQUIRE(n_streams == 0 || (d_bytes && d_lens && d_start && d_end && d_reason && d_n_calls &&
                                  (d_cls || n_windows == 0)),
               "null device pointer");
    if (n_streams == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    if (ctx->s.sep_freq != 0) {
        WN_REQUIRE(n_streams < (1ll << 31), "n_streams < 2^31");
#define WN_SLICE_ARGS d_cls, (long long)n_windows, (int)timeout_windows, (int)max_calls, (int)frame_cap, d_bytes, d_lens, \
                      d_start, d_end, d_reason, d_n_calls
        if (n_windows <= 2048)
            hipLaunchKernelGGL((slice_scan_kernel<64, 32>), dim3((unsigned)n_streams), dim3(64), 0, (hipStream_t)stream,
                               WN_SLICE_ARGS);
        else if (n_windows <= 64 * 1536)
            hipLaunchKernelGGL((slice_scan_kernel<1024, 1536>), dim3((unsigned)n_streams), dim3(1024), 0,
                               (hipStream_t)stream, WN_SLICE_ARGS);
        else
            hipLaunchKernelGGL(slice_wave_kernel<1024>, dim3((unsigned)n_streams), dim3(1024), 0, (hipStream_t)stream,
                               WN_SLICE_ARGS);
#undef WN_SLICE_ARGS
        return wn_launch_check("slice kernel");
    }
    const unsigned blocks = (unsigned)((n_streams + 63) / 64);
    hipLaunchKernelGGL(symbols_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, d_cls, (long long)n_streams,
                       (long long)n_windows, (int)max_calls, (int)frame_cap, d_bytes, d_lens, d_start, d_end, d_reason,
                       d_n_calls);
    return wn_launch_check("symbols_kernel");
}
