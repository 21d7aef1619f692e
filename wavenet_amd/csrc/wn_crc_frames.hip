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
    return WN_OK;
}

template <int MODE>
static int launch_rows(wn_ctx *ctx, const uint8_t *rows, const int32_t *lens, int32_t stride, int64_t n,
                       uint32_t *out32, wn_frame_rec *out_rec, hipStream_t st) {
    if (n == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    const bool tiled = (stride % 16 == 0) && stride >= 16 && stride <= 512 && (((uintptr_t)rows) % 16 == 0);
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
