// Shared internals of libwavenet_hip.so: the context object and error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wavenet_hip.h"

struct wn_ctx {
    wn_scheme s;
    int device;
    int num_cus;
    // synthesis: quarter-wave tables, quarter+1 entries each
    int32_t quarter;
    float *d_sin_f32;
    int16_t *d_sin_i16;
    // detection: window length n = n1*n2 (coprime, n2 odd), CRT idempotents e1, e2
    int32_t n, n1, n2, nbins;
    int32_t e1, e2;
    uint16_t *d_cls_lut;  // nbins class words
    float2 *d_tw1;        // n1 entries (cos, sin)(2*pi*m/n1)
    float2 *d_tw2;        // n2 entries (cos, -sin)(2*pi*m/n2)
    double tie_band;
    unsigned long long *d_stats;  // 16 counters (11 used)
    // sync: ping bins [lo, hi]
    int32_t ping_lo, ping_hi;
    // checksum: slicing-by-4 tables [4][256]
    uint32_t *d_crc_tab;
    uint32_t *d_crc_combine_tab;   // per-distance tables of the lane-parallel kernel (wn_crc_frames.hip)
    int crc_combine;               // WN_CRC_COMBINE=1: that kernel instead of one lane per row (stride 128)
    // tuning / diagnostic switches, read from the environment once, when the context is made
    // (WN_DETECT_PROFILE, WN_SYNTH_PER_CU, WN_SYNTH_PART); 0 = not set
    int detect_profile;
    float detect_share;    // share of the pilot's energy in one column tile that sends a window the one-tile way
    float detect_two_share;   // the weakest column tile is left out when the pilot finds at most this share in it (0: never)
    int32_t *d_cand;          // the scheme's tones other than the separator by spectrum column: 8 entries each,
                              // bin | folded row frequency << 16, 0 where there is none (null: not built)
    float detect_bin_share;   // share of the pilot's energy in one column from which the fitting tone's bin is tried alone (0: never)
    int sep_bin;              // bin nearest to the separator tone (0: the scheme has none)
    float detect_sep_share;   // share of the pilot's energy in the separator's column from which that bin is tried alone (0: never)
    float detect_pick_share;  // class width of the choice of sixteen spectrum columns, as a share of the pilot's energy (0: never)
    int detect_per_cu;     // 1: one detection workgroup per CU even where two fit (diagnostic)
    int synth_per_cu;
    int synth_table_lds;   // 1: keep the quarter-wave table in LDS whatever the scheme (WN_SYNTH_TABLE=lds)
    long long synth_part;
};

void wn_set_error(const char *fmt, ...);

#define WN_HIP_TRY(expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            wn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                       \
            return WN_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define WN_REQUIRE(cond, what)                       \
    do {                                             \
        if (!(cond)) {                               \
            wn_set_error("bad argument: %s", what);  \
            return WN_ERR_ARG;                       \
        }                                            \
    } while (0)

// Entry points that launch work: the current device must be the one the context was built
// on (its tables live there).  Returns WN_OK or WN_ERR_ARG with the two device numbers.
int wn_device_check(const wn_ctx *ctx);
#define WN_REQUIRE_DEVICE(ctx)            \
    do {                                  \
        int rc_dev_ = wn_device_check(ctx); \
        if (rc_dev_ != WN_OK) return rc_dev_; \
    } while (0)
#define WN_MAX_DEVICES 64   // per-device bookkeeping of function attributes

static inline int wn_launch_check(const char *kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        wn_set_error("launch of %s failed: %s", kernel, hipGetErrorString(e));
        return WN_ERR_HIP;
    }
    return WN_OK;
}

// host tables (wn_ctx.hip)
void wn_host_crc_tables(uint32_t tab[4][256]);
void wn_host_crc_combine_tables(uint32_t *out);   // wn_host_crc_combine_words() entries
size_t wn_host_crc_combine_words();
int wn_plan_lookup(int32_t n, int32_t *n1, int32_t *n2);

// Per-device set-up of each kernel file (dynamic-LDS limits of its kernels, addresses of its
// device symbols).  wn_scheme_init calls them once per device, with that device current, under
// one lock; launches never change process-wide state.
int wn_detect_device_setup(int device);
int wn_sync_device_setup(int device);
int wn_synth_device_setup(int device);
int wn_crc_device_setup(int device);
// whether the sines the synthesis kernel computes equal the host-built table for this context's scheme
int wn_synth_sines_agree(wn_ctx *ctx, bool *agree);
