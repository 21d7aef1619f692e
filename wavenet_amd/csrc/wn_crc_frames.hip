// K4: batched CRC-32, frame build, frame parse + checksum validation.
//
// Replaces calcular_checksum (c1_shared.py:178-187), Trama._build / _parse /
// get_checksum_valido (c1_shared.py:86-125) and capa1.crear_trama (capa1.py:80-91).
//
// Data layout: frames are rows of `stride` bytes.  A wave takes 64 consecutive rows: the
// tile is one contiguous piece of HBM, read with 16-byte loads (only the pieces a row's
// length needs), staged in LDS with one dword of padding per row so that lane r can walk
// row r dword by dword without bank conflicts, and checksummed with slicing-by-4 tables
// held in LDS.  HBM-bound integer work: (19+L) bytes in, 8 bytes out per frame.
#include "wn_common.h"

#define CRC_BLOCK 256
#define CRC_WAVES (CRC_BLOCK / 64)

__device__ __forceinline__ uint32_t crc_step_dword(const uint32_t *tab, uint32_t crc, uint32_t w) {
    crc ^= w;
    return tab[3 * 256 + (crc & 0xFF)] ^ tab[2 * 256 + ((crc >> 8) & 0xFF)] ^
           tab[1 * 256 + ((crc >> 16) & 0xFF)] ^ tab[crc >> 24];
}

__device__ __forceinline__ uint32_t crc_step_byte(const uint32_t *tab, uint32_t crc, uint32_t b) {
    return (crc >> 8) ^ tab[(crc ^ b) & 0xFF];
}

// CRC-32 (zlib) of `len` bytes of an LDS row whose base is 4-byte aligned.
__device__ __forceinline__ uint32_t crc_lds_row(const uint32_t *tab, const uint8_t *row, int len) {
    uint32_t crc = 0xFFFFFFFFu;
    const uint32_t *rw = (const uint32_t *)row;
    const int nd = len >> 2;
    for (int j = 0; j < nd; ++j) crc = crc_step_dword(tab, crc, rw[j]);
    for (int j = nd << 2; j < len; ++j) crc = crc_step_byte(tab, crc, row[j]);
    return ~crc;
}

// MODE 0: d_out32[row] = crc32(row[0..len)).  MODE 1: parse + validate into wn_frame_rec.
// PPR > 0: the stride is 16*PPR bytes and the loads of a tile are all issued before the first
// LDS write (one trip to HBM per tile instead of one per piece); PPR = 0: any stride.
template <int MODE, int PPR>
__global__ __launch_bounds__(CRC_BLOCK) void crc_rows_kernel(const uint8_t *__restrict__ rows,
                                                             const int32_t *__restrict__ lens, int stride,
                                                             long long n, const uint32_t *__restrict__ g_tab,
                                                             uint32_t *__restrict__ out32,
                                                             wn_frame_rec *__restrict__ out_rec) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *tab = (uint32_t *)smem;  // [4][256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = stride + 4;  // padded LDS row
    uint8_t *tile = smem + 4096 + (size_t)wave * 64 * lrow;
    for (int i = threadIdx.x; i < 1024; i += CRC_BLOCK) tab[i] = g_tab[i];
    __syncthreads();

    const int ppr = stride >> 4;  // 16-byte pieces per row
    const long long n_tiles = (n + 63) / 64;
    for (long long t = (long long)blockIdx.x * CRC_WAVES + wave; t < n_tiles;
         t += (long long)gridDim.x * CRC_WAVES) {
        const long long row0 = t * 64;
        const long long my_row = row0 + lane;
        int my_len = 0;
        if (my_row < n) {
            my_len = lens[my_row];
            my_len = my_len < 0 ? 0 : (my_len > stride ? stride : my_len);
        }
        // stage: piece p = (row p/ppr, 16-byte column p%ppr)
        const uint4 *src = (const uint4 *)(rows + row0 * (long long)stride);
        if constexpr (PPR > 0) {
            uint4 v[PPR];
#pragma unroll
            for (int i = 0; i < PPR; ++i) {
                const int p = lane + 64 * i, r = p / PPR, c = p % PPR;
                v[i] = make_uint4(0u, 0u, 0u, 0u);
                if (c * 16 < __shfl(my_len, r)) v[i] = src[p];
            }
#pragma unroll
            for (int i = 0; i < PPR; ++i) {
                const int p = lane + 64 * i, r = p / PPR, c = p % PPR;
                uint32_t *dst = (uint32_t *)(tile + r * lrow + c * 16);
                dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
            }
        } else {
            for (int p = lane; p < 64 * ppr; p += 64) {
                const int r = p / ppr, c = p - r * ppr;
                const int rl = __shfl(my_len, r);
                if (c * 16 < rl) {
                    const uint4 v = src[p];
                    uint32_t *dst = (uint32_t *)(tile + r * lrow + c * 16);
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                }
            }
        }
        // a wave only reads what it wrote itself; LDS ops of one wave complete in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (my_row < n) {
            const uint8_t *row = tile + lane * lrow;
            if (MODE == 0) {
                out32[my_row] = crc_lds_row(tab, row, my_len);
            } else {
                wn_frame_rec rec;
                rec.crc = 0; rec.crc_ok = 0; rec.tipo = 0; rec.length = 0;
                if (my_len == 0) {
                    rec.status = WN_ST_EMPTY;
                } else if (my_len < 15) {
                    rec.status = WN_ST_SHORT_HEADER;
                } else {
                    rec.tipo = row[13];
                    rec.length = row[14];
                    const int body = 15 + rec.length;
                    if (my_len < body + 4) {
                        rec.status = WN_ST_SHORT_BODY;
                    } else {
                        rec.status = WN_ST_OK;
                        rec.crc = crc_lds_row(tab, row, body);
                        const uint32_t stored = ((uint32_t)row[body] << 24) | ((uint32_t)row[body + 1] << 16) |
                                                ((uint32_t)row[body + 2] << 8) | (uint32_t)row[body + 3];
                        rec.crc_ok = stored == rec.crc;
                    }
                }
                out_rec[my_row] = rec;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same two operations with a lane-parallel combine instead of one lane walking a row
// (selected by WN_CRC_COMBINE=1 when the context is made; stride 128 only).
//
// A CRC register is linear over GF(2) in the message bits, so the checksum of an L-byte row is
//     ~( XOR_i  T[L-1-i][ M_i ]  ^  INIT[L] ),
// where T[e][b] is the register after byte b followed by e zero bytes (start value 0) and INIT[L]
// the register after L zero bytes from the start value 0xFFFFFFFF.  Every byte's term is one
// look-up that depends on nothing else: no carried state, no order.  Eight lanes hold the eight
// 16-byte pieces of a row exactly as one coalesced load delivers them (a wave-wide 16-byte load
// is eight whole rows), each lane XORs the 16 terms of its own piece, and three DPP steps XOR the
// eight partial registers together.  The tables for every distance a 128-byte row can have
// (128 x 1 KiB, plus 15 all-zero ones below them so that bytes past the end of the message need no
// mask: their distance is negative and lands there) fill 143 KiB of LDS per CU.
// Nothing is staged, no lane waits for a longer row in its wave, and the look-ups of a lane are
// independent; the price is 16 look-ups per lane whatever the row's length (2 per row and wave
// instruction, about what the longest row of 64 costs the one-lane-per-row kernel).
// ---------------------------------------------------------------------------------------
#define CMB_ZERO_ROWS 15
#define CMB_MAX_LEN 128
#define CMB_TAB_ROWS (CMB_ZERO_ROWS + CMB_MAX_LEN)
#define CMB_WORDS (CMB_TAB_ROWS * 256 + CMB_MAX_LEN + 4)   // tables, then INIT[0..128]
#define CMB_BLOCK 1024
#define CMB_WAVES (CMB_BLOCK / 64)
#define CMB_UNROLL 4   // groups of eight rows a wave has in flight

// the tables of the combine kernel, as the kernel lays them out (host side; wn_scheme_init uploads them)
void wn_host_crc_combine_tables(uint32_t *out) {
    static uint32_t tab[4][256];
    wn_host_crc_tables(tab);
    for (int i = 0; i < CMB_ZERO_ROWS * 256; ++i) out[i] = 0u;
    uint32_t *t = out + CMB_ZERO_ROWS * 256;
    for (int b = 0; b < 256; ++b) t[b] = tab[0][b];
    for (int e = 1; e < CMB_MAX_LEN; ++e)
        for (int b = 0; b < 256; ++b) {
            const uint32_t v = t[(e - 1) * 256 + b];
            t[e * 256 + b] = (v >> 8) ^ tab[0][v & 0xFF];
        }
    uint32_t *init = out + CMB_TAB_ROWS * 256;
    uint32_t st = 0xFFFFFFFFu;
    for (int l = 0; l <= CMB_MAX_LEN; ++l) {
        init[l] = st;
        st = (st >> 8) ^ tab[0][st & 0xFF];
    }
    for (int l = CMB_MAX_LEN + 1; l < CMB_MAX_LEN + 4; ++l) init[l] = 0u;
}
size_t wn_host_crc_combine_words() { return CMB_WORDS; }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_xor(uint32_t v) {
    return v ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

template <int MODE>
__global__ __launch_bounds__(CMB_BLOCK) void crc_combine_kernel(const uint8_t *__restrict__ rows,
                                                                const int32_t *__restrict__ lens, long long n,
                                                                const uint32_t *__restrict__ g_tabs,
                                                                uint32_t *__restrict__ out32,
                                                                wn_frame_rec *__restrict__ out_rec) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *tabs = (uint32_t *)smem;
    {
        const uint4 *src = (const uint4 *)g_tabs;
        uint4 *dst = (uint4 *)smem;
        for (int i = threadIdx.x; i < CMB_WORDS / 4; i += CMB_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t *init = tabs + CMB_TAB_ROWS * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 7, sub = lane >> 3;   // piece of the row, row of the group
    constexpr int TILE = 8 * CMB_UNROLL;       // rows a wave takes at a time
    const long long n_tiles = (n + TILE - 1) / TILE;
    const long long tstep = (long long)gridDim.x * CMB_WAVES;
    long long t = (long long)blockIdx.x * CMB_WAVES + wave;
    uint4 v[CMB_UNROLL], nv[CMB_UNROLL];
    int ln[CMB_UNROLL], nl[CMB_UNROLL];
    auto load = [&](long long tile, uint4 (&vv)[CMB_UNROLL], int (&ll)[CMB_UNROLL]) {
#pragma unroll
        for (int u = 0; u < CMB_UNROLL; ++u) {
            const long long row = tile * TILE + 8 * u + sub;
            vv[u] = make_uint4(0u, 0u, 0u, 0u);
            ll[u] = 0;
            if (row < n) {
                // the whole 128-byte line of a row comes from HBM whatever part of it is asked for
                vv[u] = ((const uint4 *)(rows + row * 128))[c];
                ll[u] = lens[row];
            }
        }
    };
    if (t < n_tiles) load(t, v, ln);
    for (; t < n_tiles; t += tstep) {
        if (t + tstep < n_tiles) load(t + tstep, nv, nl);
#pragma unroll
        for (int u = 0; u < CMB_UNROLL; ++u) {
            const long long row = t * TILE + 8 * u + sub;
            int len = ln[u];
            len = len < 0 ? 0 : (len > CMB_MAX_LEN ? CMB_MAX_LEN : len);
            int crc_len = len;
            wn_frame_rec rec;
            rec.crc = 0; rec.crc_ok = 0; rec.tipo = 0; rec.length = 0; rec.status = WN_ST_OK;
            int body = 0;
            if constexpr (MODE == 1) {
                // type and length bytes sit in piece 0: every lane of the row gets them from its lane
                const uint32_t w3 = (uint32_t)__shfl((int)v[u].w, lane & ~7);
                crc_len = 0;
                if (len == 0) {
                    rec.status = WN_ST_EMPTY;
                } else if (len < 15) {
                    rec.status = WN_ST_SHORT_HEADER;
                } else {
                    rec.tipo = (uint8_t)(w3 >> 8);
                    rec.length = (uint8_t)(w3 >> 16);
                    body = 15 + rec.length;
                    if (len < body + 4) rec.status = WN_ST_SHORT_BODY;
                    else crc_len = body;
                }
            }
            // table of the last byte of this piece is base + 0, of its first byte base + 15; pieces
            // that start at or after the end of the message contribute nothing
            const int base = crc_len - 16 * c - 16 + CMB_ZERO_ROWS;   // table row (with the zero rows) of byte 15 of the piece
            const bool live = 16 * c < crc_len;
            const uint32_t *tb = tabs + (live ? base : 0) * 256;
            const uint32_t d[4] = {live ? v[u].x : 0u, live ? v[u].y : 0u, live ? v[u].z : 0u, live ? v[u].w : 0u};
            uint32_t x = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) x ^= tb[(15 - k) * 256 + ((d[k >> 2] >> (8 * (k & 3))) & 0xFF)];
            x = dpp_xor<0xB1>(x);    // lanes 2j, 2j+1
            x = dpp_xor<0x4E>(x);    // quads
            x = dpp_xor<0x141>(x);   // both quads of the row's eight lanes
            const uint32_t crc = ~(x ^ init[crc_len]);
            if (c == 0 && row < n) {
                if constexpr (MODE == 0) {
                    out32[row] = crc;
                } else {
                    if (rec.status == WN_ST_OK) {
                        const uint8_t *p = rows + row * 128 + body;   // the line is in the cache
                        const uint32_t stored = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
                        rec.crc = crc;
                        rec.crc_ok = stored == crc;
                    }
                    out_rec[row] = rec;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CMB_UNROLL; ++u) {
            v[u] = nv[u];
            ln[u] = nl[u];
        }
    }
}

// Fallback for strides the tiled kernel does not take: one lane per row, bytes from HBM.
template <int MODE>
__global__ __launch_bounds__(CRC_BLOCK) void crc_rows_generic_kernel(const uint8_t *__restrict__ rows,
                                                                     const int32_t *__restrict__ lens, int stride,
                                                                     long long n, const uint32_t *__restrict__ g_tab,
                                                                     uint32_t *__restrict__ out32,
                                                                     wn_frame_rec *__restrict__ out_rec) {
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += CRC_BLOCK) tab[i] = g_tab[i];
    __syncthreads();
    for (long long r = (long long)blockIdx.x * CRC_BLOCK + threadIdx.x; r < n; r += (long long)gridDim.x * CRC_BLOCK) {
        const uint8_t *row = rows + r * (long long)stride;
        int len = lens[r];
        len = len < 0 ? 0 : (len > stride ? stride : len);
        if (MODE == 0) {
            uint32_t crc = 0xFFFFFFFFu;
            for (int j = 0; j < len; ++j) crc = crc_step_byte(tab, crc, row[j]);
            out32[r] = ~crc;
        } else {
            wn_frame_rec rec;
            rec.crc = 0; rec.crc_ok = 0; rec.tipo = 0; rec.length = 0;
            if (len == 0) {
                rec.status = WN_ST_EMPTY;
            } else if (len < 15) {
                rec.status = WN_ST_SHORT_HEADER;
            } else {
                rec.tipo = row[13];
                rec.length = row[14];
                const int body = 15 + rec.length;
                if (len < body + 4) {
                    rec.status = WN_ST_SHORT_BODY;
                } else {
                    rec.status = WN_ST_OK;
                    uint32_t crc = 0xFFFFFFFFu;
                    for (int j = 0; j < body; ++j) crc = crc_step_byte(tab, crc, row[j]);
                    rec.crc = ~crc;
                    const uint32_t stored = ((uint32_t)row[body] << 24) | ((uint32_t)row[body + 1] << 16) |
                                            ((uint32_t)row[body + 2] << 8) | (uint32_t)row[body + 3];
                    rec.crc_ok = stored == rec.crc;
                }
            }
            out_rec[r] = rec;
        }
    }
}

// Frame build: one lane per frame (the transmit side handles few frames per audio second).
__global__ __launch_bounds__(CRC_BLOCK) void frame_build_kernel(const uint8_t *__restrict__ fields,
                                                                const uint8_t *__restrict__ payloads,
                                                                int payload_stride, long long n,
                                                                const uint32_t *__restrict__ g_tab,
                                                                uint8_t *__restrict__ frames, int frame_stride,
                                                                int32_t *__restrict__ frame_lens) {
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += CRC_BLOCK) tab[i] = g_tab[i];
    __syncthreads();
    for (long long r = (long long)blockIdx.x * CRC_BLOCK + threadIdx.x; r < n; r += (long long)gridDim.x * CRC_BLOCK) {
        const uint8_t *f = fields + r * 16;
        const uint8_t *p = payloads + r * (long long)payload_stride;
        uint8_t *o = frames + r * (long long)frame_stride;
        const int len = f[14];
        if (19 + len > frame_stride || len > payload_stride) {  // does not fit the caller's rows
            frame_lens[r] = -1;
            continue;
        }
        uint32_t crc = 0xFFFFFFFFu;
        for (int j = 0; j < 15; ++j) {
            const uint8_t b = f[j];
            o[j] = b;
            crc = crc_step_byte(tab, crc, b);
        }
        for (int j = 0; j < len; ++j) {
            const uint8_t b = p[j];
            o[15 + j] = b;
            crc = crc_step_byte(tab, crc, b);
        }
        crc = ~crc;
        o[15 + len] = (uint8_t)(crc >> 24);
        o[16 + len] = (uint8_t)(crc >> 16);
        o[17 + len] = (uint8_t)(crc >> 8);
        o[18 + len] = (uint8_t)crc;
        for (int j = 19 + len; j < frame_stride; ++j) o[j] = 0;
        frame_lens[r] = 19 + len;
    }
}

// the tiled kernel for any stride stages 256 rows of up to 512 + 4 bytes next to the tables: allow that much
int wn_crc_device_setup(int) {
    WN_HIP_TRY(hipFuncSetAttribute((const void *)crc_rows_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   4096 + CRC_WAVES * 64 * (512 + 4)));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)crc_rows_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   4096 + CRC_WAVES * 64 * (512 + 4)));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)crc_combine_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   CMB_WORDS * 4));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)crc_combine_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   CMB_WORDS * 4));
    return WN_OK;
}

template <int MODE>
static int launch_rows(wn_ctx *ctx, const uint8_t *rows, const int32_t *lens, int32_t stride, int64_t n,
                       uint32_t *out32, wn_frame_rec *out_rec, hipStream_t st) {
    if (n == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    const bool tiled = (stride % 16 == 0) && stride >= 16 && stride <= 512 && (((uintptr_t)rows) % 16 == 0);
    if (ctx->crc_combine && stride == 128 && (((uintptr_t)rows) % 16 == 0)) {
        const long long tiles = (n + 8 * CMB_UNROLL - 1) / (8 * CMB_UNROLL);
        long long blocks = (tiles + CMB_WAVES - 1) / CMB_WAVES;
        if (blocks > (long long)ctx->num_cus) blocks = ctx->num_cus;   // the tables fill a CU's LDS: one workgroup each
        hipLaunchKernelGGL(crc_combine_kernel<MODE>, dim3((unsigned)blocks), dim3(CMB_BLOCK), CMB_WORDS * 4, st, rows, lens,
                           (long long)n, ctx->d_crc_combine_tab, out32, out_rec);
        return wn_launch_check("crc_combine_kernel");
    }
    if (tiled) {
        const size_t lds = 4096 + (size_t)CRC_WAVES * 64 * (stride + 4);
        const long long tiles = (n + 63) / 64;
        long long blocks = (tiles + CRC_WAVES - 1) / CRC_WAVES;
        const long long cap = (long long)ctx->num_cus * 8;
        if (blocks > cap) blocks = cap;
        if (stride == 128)
            hipLaunchKernelGGL((crc_rows_kernel<MODE, 8>), dim3((unsigned)blocks), dim3(CRC_BLOCK), lds, st, rows, lens,
                               (int)stride, (long long)n, ctx->d_crc_tab, out32, out_rec);
        else
            hipLaunchKernelGGL((crc_rows_kernel<MODE, 0>), dim3((unsigned)blocks), dim3(CRC_BLOCK), lds, st, rows, lens,
                               (int)stride, (long long)n, ctx->d_crc_tab, out32, out_rec);
        return wn_launch_check("crc_rows_kernel");
    }
    long long blocks = (n + CRC_BLOCK - 1) / CRC_BLOCK;
    const long long cap = (long long)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(crc_rows_generic_kernel<MODE>, dim3((unsigned)blocks), dim3(CRC_BLOCK), 0, st, rows, lens,
                       (int)stride, (long long)n, ctx->d_crc_tab, out32, out_rec);
    return wn_launch_check("crc_rows_generic_kernel");
}

extern "C" int wn_crc32_batch(wn_ctx *ctx, const uint8_t *d_msgs, const int32_t *d_lens, int32_t stride, int64_t n,
                              uint32_t *d_out, void *stream) {
    WN_REQUIRE(ctx && n >= 0 && stride > 0, "ctx / n >= 0 / stride > 0");
    WN_REQUIRE(n == 0 || (d_msgs && d_lens && d_out), "null device pointer");
    return launch_rows<0>(ctx, d_msgs, d_lens, stride, n, d_out, nullptr, (hipStream_t)stream);
}

extern "C" int wn_frame_parse_validate_batch(wn_ctx *ctx, const uint8_t *d_frames, const int32_t *d_nbytes,
                                             int32_t stride, int64_t n, wn_frame_rec *d_out, void *stream) {
    WN_REQUIRE(ctx && n >= 0 && stride > 0, "ctx / n >= 0 / stride > 0");
    WN_REQUIRE(n == 0 || (d_frames && d_nbytes && d_out), "null device pointer");
    return launch_rows<1>(ctx, d_frames, d_nbytes, stride, n, nullptr, d_out, (hipStream_t)stream);
}

extern "C" int wn_frame_build_batch(wn_ctx *ctx, const uint8_t *d_fields, const uint8_t *d_payloads,
                                    int32_t payload_stride, int64_t n, uint8_t *d_frames, int32_t frame_stride,
                                    int32_t *d_frame_lens, void *stream) {
    WN_REQUIRE(ctx && n >= 0, "ctx / n >= 0");
    WN_REQUIRE(payload_stride >= 0 && frame_stride >= 19, "payload_stride >= 0 / frame_stride >= 19");
    WN_REQUIRE(n == 0 || (d_fields && d_frames && d_frame_lens && (d_payloads || payload_stride == 0)),
               "null device pointer");
    if (n == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    long long blocks = (n + CRC_BLOCK - 1) / CRC_BLOCK;
    const long long cap = (long long)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(frame_build_kernel, dim3((unsigned)blocks), dim3(CRC_BLOCK), 0, (hipStream_t)stream, d_fields,
                       d_payloads, (int)payload_stride, (long long)n, ctx->d_crc_tab, d_frames, (int)frame_stride,
                       d_frame_lens);
    return wn_launch_check("frame_build_kernel");
}
