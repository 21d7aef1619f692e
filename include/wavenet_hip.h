/*
 * wavenet_hip.h — C ABI of libwavenet_hip.so: the layer-1 acoustic modem of
 * yosgarcia/WaveNet as batched HIP kernels for MI355X (gfx950).
 *
 * The reference has no FFI of its own: its hot path is Python/NumPy inside
 * DispositivoWaveNET/dispositivo_wavenet/c1_shared.py ("B" below) and
 * DispositivoWaveNET/capa1.py ("A").  Each entry point here names the reference
 * function(s) it replaces.  A Python binding (ctypes) that keeps the reference's call
 * surface lives in wavenet_amd/; any other host can bind the same symbols.
 *
 * Conventions
 *  - Every function returns 0 (WN_OK) or a negative WN_ERR_* code and never throws.
 *  - Pointers named d_* are DEVICE pointers (hipMalloc'ed by the caller, e.g. the storage
 *    of a torch tensor); everything else is host memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *    enqueue work; they do not synchronise unless stated.
 *  - Threads.  wn_scheme_init does everything that is per device and process-wide (kernel
 *    attributes, symbol addresses) under a lock, and reads the environment switches once; the
 *    calls that launch work only read the context.  A wn_ctx may therefore be shared by threads,
 *    each enqueueing on its own stream.  Exceptions, all stated at their declarations:
 *    wn_detect_set_tie_band changes the context (a set-up call: not while another thread
 *    detects with it), the counters read by wn_detect_stats are summed over all callers of the
 *    context, and graph replays of wn_detect_windows_noalloc must not overlap other calls of it.
 */
#ifndef WAVENET_HIP_H
#define WAVENET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_OK 0
#define WN_ERR_ARG (-1)          /* bad argument (null pointer, size, alignment)           */
#define WN_ERR_HIP (-2)          /* a HIP runtime call failed: see wn_last_error()          */
#define WN_ERR_UNSUPPORTED (-3)  /* scheme outside what the kernels are built for          */
#define WN_ERR_NOMEM (-4)

/* Tone plan and timing of one modem variant (B:15-44, A:12-18), in samples and Hz. */
typedef struct wn_scheme {
    int32_t sample_rate;   /* 44100; must be a multiple of 4                              */
    int32_t base_freq;     /* B 350 / A 500: tone of byte 0                               */
    int32_t freq_step;     /* B 70 / A 20                                                 */
    int32_t max_byte;      /* B 255 / A 127: highest value freq_to_byte returns           */
    int32_t tol_check;     /* B 1: |f - tone| < step/2 required (B:161); A 0              */
    int32_t sep_freq;      /* B 200; 0 = no separators and no ping (A)                    */
    int32_t ping_freq;     /* B 250; 0 = none                                             */
    int32_t eof_freq;      /* B 300 / A 900                                               */
    int32_t tone_n;        /* samples per data tone: B 10584 / A 2205                     */
    int32_t sep_n;         /* B 13230                                                     */
    int32_t sep_short_n;   /* B 10584 (separator before a repeated byte, B:428)           */
    int32_t ping_n;        /* B 21168                                                     */
    int32_t eof_n;         /* B 21168 / A 2205                                            */
    int32_t win_n;         /* receive window: B 7055 (B:323) / A 2205                     */
    double amp;            /* 0.5                                                         */
    double gain;           /* B 1.95 (then clipped to [-1,1], B:392-393); A 1.0           */
} wn_scheme;

typedef struct wn_ctx wn_ctx;

/* Class word of a window (what the receive loop decides from its dominant bin). */
#define WN_CLS_BYTE_MASK 0x1FF   /* 0..255 = byte value, 256 = no byte                     */
#define WN_CLS_NONE 256
#define WN_CLS_PING (1 << 9)     /* |f - ping| < 10 (B:331)                                */
#define WN_CLS_EOF (1 << 10)     /* |f - eof| < 10 (B:337)                                 */

/* Frame record written by wn_frame_parse_validate_batch (8 bytes). */
typedef struct wn_frame_rec {
    uint32_t crc;          /* CRC-32 of bytes [0, 15+length); 0 unless status == WN_ST_OK  */
    uint8_t status;        /* WN_ST_*                                                      */
    uint8_t crc_ok;        /* 1 iff the 4 stored bytes equal crc, big-endian (B:119-125)   */
    uint8_t tipo;          /* byte 13                                                      */
    uint8_t length;        /* byte 14                                                      */
} wn_frame_rec;
#define WN_ST_OK 0
#define WN_ST_EMPTY 1            /* no bytes: "Faltan argumentos..." (B:72-77)             */
#define WN_ST_SHORT_HEADER 2     /* < 15 bytes: "Trama muy corta" (B:90-91)                */
#define WN_ST_SHORT_BODY 3       /* < 15+length+4 bytes (B:99-100)                         */

/* Why a listen call ended (wn_slice_frames). */
#define WN_END_EOF 0             /* EOF tone seen (B:337-339)                              */
#define WN_END_TIMEOUT 1         /* clock test failed or capture exhausted (B:319-321)     */
#define WN_END_OVERFLOW 0x100    /* or-ed in: more bytes than frame_cap, extra dropped     */

/* Emission variants for wn_synth_layout. */
#define WN_EMIT_LIVE 0           /* emitir_trama (B:415-434; A:40-46)                      */
#define WN_EMIT_WAV 1            /* guardar_trama_como_wav (B:445-466; A:163-173)          */
/* Sample formats for wn_synth_batch. */
#define WN_FMT_F32 0             /* clip(amp*sin*gain) as float (B:391-393)                */
#define WN_FMT_I16 1             /* int16(amp*sin*32767), truncated (B:467; A:173)         */
#define WN_SYNTH_BESIDE 0x100    /* or-ed in: the small-footprint kernel (256 threads, < 48 registers, < 10 KB LDS),
                                    whose workgroups fit on CUs that a wn_detect_windows launch on another stream
                                    occupies; the same samples (no counterpart in the reference)                */

const char *wn_strerror(int code);
/* Text of the last failure on the calling thread (HIP error string or argument name). */
const char *wn_last_error(void);

/* ---- host-only helpers (no GPU needed): the tables the kernels use ----------------- */
/* Per-bin class words for a window of s->win_n samples, bins [0, win_n/2): freq_to_byte
 * (B:152-163; A:26-30) and the ping/EOF tests (B:331, 337) applied to
 * |fftfreq(win_n, 1/sample_rate)[k]| (B:296-299).  Returns the number of bins. */
int wn_host_class_lut(const wn_scheme *s, uint16_t *out, int32_t cap);
/* Factorisation win_n = n1*n2 (coprime, n2 odd) used by the detection kernel. */
int wn_host_dft_plan(const wn_scheme *s, int32_t *n1, int32_t *n2);
/* Quarter-wave tables, sample_rate/4+1 entries: out_f32[m] = clip(amp*sin(2*pi*m/sr)*gain),
 * out_i16[m] = (int16)(amp*sin(2*pi*m/sr)*32767).  Either pointer may be NULL. */
int wn_host_sine_tables(const wn_scheme *s, float *out_f32, int16_t *out_i16, int32_t cap);
/* Samples one frame of `len` bytes occupies (B:415-434 / 445-466 / A:40-46). */
int64_t wn_host_frame_samples(const wn_scheme *s, const uint8_t *frame, int32_t len, int32_t variant);

/* ---- context ---------------------------------------------------------------------- */
/* Builds the device-side tables for one scheme on the current device. */
int wn_scheme_init(const wn_scheme *s, wn_ctx **out);
int wn_ctx_destroy(wn_ctx *ctx);
/* Extra relative band (on magnitudes) under which the largest bins of a window count as tied
 * and are re-evaluated in float64, on top of the error bound the kernel applies by itself
 * (default 0).  Tests raise it to force that path.  Changes the context: call it before the
 * context is shared, not concurrently with detection on it. */
int wn_detect_set_tie_band(wn_ctx *ctx, double rel);
/* Counters since the last call (synchronises `stream`): out[0] windows detected,
 * out[1] windows whose two or three candidate bins were re-evaluated in float64, out[2] windows
 * whose whole spectrum was; and how much of the two DFT stages the kernel computed: out[3] (row
 * tile of 16 outputs, spectrum column) products of the second stage (((n1+1)/2 + 15)/16 tiles times
 * (n2+1)/2 columns k mod n2, folded, per window at most),
 * out[4] (row tile of 16, real or imaginary part, column) products of the first stage
 * (((n1+1)/2 + 15)/16*2 tiles times 2 times (n2+1)/2 columns per window at most), out[5] windows
 * that were started on part of the spectrum (one bin, one column tile, sixteen chosen columns, or all
 * tiles but the weakest) and had to be done again on more (their work is counted again in out[3],
 * out[4]), out[6] passes "the long way" (first stage of every column tile, or of all but the
 * weakest; those of out[5] included), out[7] of them those that left the weakest column tile out,
 * out[8] windows that took one tile of sixteen columns chosen for them, out[9] windows decided from
 * the separator's bin alone, out[10] windows decided from the bin of another tone of the scheme alone;
 * out[11] is reserved and zero.  The rest was left out because the energy outside the outputs already seen could not reach
 * the decision threshold.  One set of counters per context, summed over every thread and stream
 * that detects with it. */
int wn_detect_stats(wn_ctx *ctx, int64_t out[12], void *stream);

/* ---- K4: checksum and frames ------------------------------------------------------- */
/* calcular_checksum (B:178-187; A:89) over n messages: d_out[i] = zlib CRC-32 of
 * d_msgs[i*stride .. +d_lens[i]).  0 <= lens[i] <= stride. */
int wn_crc32_batch(wn_ctx *ctx, const uint8_t *d_msgs, const int32_t *d_lens, int32_t stride,
                   int64_t n, uint32_t *d_out, void *stream);
/* Trama(version, mac_origen, mac_destino, tipo, payload)._build (B:105-111; A:80-91).
 * d_fields: n rows of 16 bytes = version, mac_o[6], mac_d[6], tipo, payload length, 0.
 * d_payloads: n rows of payload_stride bytes.  Writes header+payload+CRC32 (big-endian)
 * into d_frames rows of frame_stride bytes and the frame length 19+length; a frame that
 * does not fit its row (or whose length exceeds payload_stride) gets length -1. */
int wn_frame_build_batch(wn_ctx *ctx, const uint8_t *d_fields, const uint8_t *d_payloads,
                         int32_t payload_stride, int64_t n, uint8_t *d_frames, int32_t frame_stride,
                         int32_t *d_frame_lens, void *stream);
/* Trama(bytes_trama)._parse + get_checksum_valido (B:86-103, 119-125) over n received
 * byte strings (d_nbytes[i] bytes in row i; bytes beyond 15+length+4 are ignored). */
int wn_frame_parse_validate_batch(wn_ctx *ctx, const uint8_t *d_frames, const int32_t *d_nbytes,
                                  int32_t stride, int64_t n, wn_frame_rec *d_out, void *stream);

/* ---- K1: tone synthesis ------------------------------------------------------------ */
/* Tone schedule of n frames (emitir_trama's loop, B:420-433, or the WAV variant):
 * segment j of frame i starts at sample d_seg_start[i*(seg_cap+1)+j], has frequency
 * d_seg_freq[i*seg_cap+j] Hz; d_seg_start[..+n_segs] is the frame's sample count, also
 * written to d_n_samples[i].  seg_cap >= 2*max_len+4. */
int wn_synth_layout(wn_ctx *ctx, const uint8_t *d_frames, const int32_t *d_lens, int32_t stride,
                    int64_t n, int32_t variant, int32_t seg_cap, int32_t *d_seg_start,
                    uint16_t *d_seg_freq, int32_t *d_n_segs, int64_t *d_n_samples, void *stream);
/* transmite_freq / transmitir_silencio for every segment (B:386-406; A:33-37): frame i is
 * written to d_audio[d_dst_off[i] ..], then zeros up to d_fill_to[i] samples when
 * d_fill_to != NULL.  slot_len bounds the samples any frame may occupy (incl. fill);
 * audio_len is the size of d_audio in samples. */
int wn_synth_batch(wn_ctx *ctx, const int32_t *d_seg_start, const uint16_t *d_seg_freq,
                   const int32_t *d_n_segs, int32_t seg_cap, int64_t n, const int64_t *d_dst_off,
                   const int64_t *d_fill_to, int64_t slot_len, void *d_audio, int64_t audio_len,
                   int32_t fmt, void *stream);
/* Synthetic channel: adds seeded white Gaussian noise (splitmix64 + Box-Muller per
 * sample index) of standard deviation sigma to d_audio[0..n). */
int wn_channel_awgn(wn_ctx *ctx, float *d_audio, int64_t n, float sigma, uint64_t seed, void *stream);

/* ---- K2: tone detection ------------------------------------------------------------ */
/* detectar_frecuencia (B:286-299; A:54-59) + the class of the detected frequency
 * (freq_to_byte and the ping/EOF tests, B:331-341) for window w of stream s =
 * d_audio[s*stream_stride + first_sample + w*win_n .. +win_n), all s < n_streams,
 * w < n_windows.  Outputs are [n_streams][n_windows]: class word, |X| of the dominant
 * bin, and its index (first maximum over bins [0, win_n/2), as np.argmax).
 * d_audio must be 16-byte aligned and audio_len a multiple of 4.  Samples must be finite; any
 * finite amplitude gives the float64 decision (windows too quiet or too loud for float32
 * energies are re-evaluated in float64, all-zero windows give bin 0). */
int wn_detect_windows(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                      int64_t stream_stride, int64_t first_sample, int64_t n_windows,
                      uint16_t *d_cls, float *d_peak_mag, int32_t *d_peak_bin, void *stream);

/* No counterpart in the reference.  For callers that cannot allocate device memory or let a
 * library do so (code under HIP graph capture, tight serving loops, harnesses that forbid it) and
 * that only need the class words: wn_detect_windows without the magnitude and bin outputs.  What
 * that call keeps in them between its two kernels, and its counters, live here in a workspace
 * inside the library (one array per GPU, 8 MiB + 256 bytes, held until the process ends), so
 * the call makes no allocation of any kind.  At most WN_DETECT_NOALLOC_MAX_WINDOWS windows
 * (n_streams * n_windows) per call; more is refused with WN_ERR_ARG.  d_cls equals what
 * wn_detect_windows writes.  Calls on one GPU share the workspace.  Outside graph capture the library
 * orders them itself: a call's work waits, on the GPU, for the previous call's, whichever stream or
 * thread enqueued it (an event recorded behind every call, no allocation).  A call captured into a
 * graph takes no part in that ordering: replays of such a graph must not run at the same time as
 * other calls of this function on the same GPU.  The counters are not reported. */
#define WN_DETECT_NOALLOC_MAX_WINDOWS (1 << 20)
int wn_detect_windows_noalloc(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                              int64_t stream_stride, int64_t first_sample, int64_t n_windows,
                              uint16_t *d_cls, void *stream);

/* ---- K3: frame sync ----------------------------------------------------------------- */
/* Sync-tone correlator over the same window grid: per window the energy sum(x^2), the
 * largest |X_k|^2 over the ping bins, and a verdict on "dominant bin is a ping bin"
 * (the test of escuchar_ping, B:380-382, and of B:330-335): 1 = certainly, 0 = certainly
 * not, 2 = undecided (needs wn_detect_windows on that window). */
int wn_sync_scan(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                 int64_t stream_stride, int64_t first_sample, int64_t n_windows,
                 float *d_ping_energy, float *d_win_energy, uint8_t *d_verdict, void *stream);

/* ---- receive state machine ----------------------------------------------------------- */
/* escuchar_y_retornar_trama's loop (B:312-351) run over class words, call after call:
 * each call starts at the window after the previous call's last one and ends at an EOF
 * window, after timeout_windows windows (counted again from the ping window, B:334), or
 * at the end of the capture.  Per stream up to max_calls records: bytes
 * [n_streams][max_calls][frame_cap], their count, first window, end window (exclusive),
 * reason (WN_END_*); d_n_calls[s] = records written. */
int wn_slice_frames(wn_ctx *ctx, const uint16_t *d_cls, int64_t n_streams, int64_t n_windows,
                    int32_t timeout_windows, int32_t max_calls, int32_t frame_cap,
                    uint8_t *d_bytes, int32_t *d_lens, int32_t *d_start, int32_t *d_end,
                    int32_t *d_reason, int32_t *d_n_calls, void *stream);

/* ---- diagnostics --------------------------------------------------------------------- */
/* Runs one v_mfma_f32_16x16x4_f32 on index-coded operands and writes the 64x4 result
 * registers, so a test can confirm the lane layout the detection kernel assumes. */
int wn_selftest_mfma(float *d_out256, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_HIP_H */
