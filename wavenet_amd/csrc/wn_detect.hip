// K2: per-window tone detection = argmax_k |DFT(x)[k]| over k in [0, N/2), plus the class
// of that bin.  Replaces detectar_frecuencia (c1_shared.py:286-299; capa1.py:54-59) and
// the decisions taken from its result (freq_to_byte, ping and EOF tests,
// c1_shared.py:331-341) through a per-bin class table.
//
// The whole spectrum is needed: in a noisy or mixed window the reference's argmax can
// land on any bin, and almost every bin maps to some byte.  The window length N factors
// into coprime odd n1*n2 (7055 = 83*85, 2205 = 45*49), so the N-point DFT is computed by
// the prime-factor (Good-Thomas) algorithm, with no twiddles between its two stages:
//
//   x2[a][b] = x[(n2*a + n1*b) mod N]                         a < n1, b < n2
//   Y[a][k2] = sum_b x2[a][b] * exp(-2*pi*i*b*k2/n2)          k2 <= h2 = n2/2 (real input)
//   Z[k1][k2] = sum_a exp(-2*pi*i*a*k1/n1) * Y[a][k2]
//   X[k] = Z[k mod n1][k mod n2]; bins with (k mod n2) > h2 are mirrors of these.
//
// Both stages fold their summation index in mirrored pairs (cos is even, sin is odd in the
// index), which halves the multiplications, and stage 2 also produces rows k1 and n1-k1
// from the same four real sums.  With xs/xd = sum/difference of the mirrored columns b' and
// n2-b' of x2, folded once more over the mirrored rows a' and n1-a':
//
//   stage 1, b' in [0, h2]:  Sr = sum cos(b'k2) (xs[a'] + xs[n1-a']),  Dr = sum cos(b'k2) (xs[a'] - xs[n1-a'])
//                            Si = -sum sin(b'k2) (xd[a'] + xd[n1-a']), Di = -sum sin(b'k2) (xd[a'] - xd[n1-a'])
//   stage 2, a' in [0, h1], k1 in [0, h1]:
//       A = sum cos(a'k1) Sr, B = sum sin(a'k1) Di, C = sum cos(a'k1) Si, E = sum sin(a'k1) Dr
//       Z[k1] = (A + B) + i(C - E),   Z[n1-k1] = (A - B) + i(C + E)
//
// All sums are small dense contractions and run on the matrix cores in f32
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation): 792 instructions of
// 2048 FLOP per 7055-sample window (1.2 MFLOP useful) instead of the 99.5 MFLOP of a dense
// N x N/2 DFT.  A stage-1 result tile is already laid out as the B operand of stage 2
// (rows on register index and lane group), so S and D never leave the registers.
//
// One wave owns one window at a time; a workgroup is four independent waves (one per
// SIMD, the whole 512-entry register file each).  On this hardware a wave's own vector-ALU
// and LDS instructions do not overlap with its matrix instructions, so the kernel is built
// to issue as few of them as possible: every twiddle factor is a per-lane constant held in
// registers for the whole launch (stage 1's in the accumulation registers, stage 2's in the
// ordinary ones), the gather addresses of stage 1 are per-lane bases plus immediates (the
// window is staged in LDS with a circular extension so that the index never wraps), and the
// matrix instructions are issued in long uninterrupted runs.  The window is staged by DMA;
// the next window is fetched while stage 2 runs.
// Bound: matrix cores (f32 MFMA, 64 FLOP/clk/SIMD).
//
// Only the part of the spectrum that can matter is computed.  By Parseval the outputs not yet
// seen hold at most E - sum|Z seen|^2 (E from the samples); once that is below what could still
// change the decision, the rest of both stages is left out and bins, classes, flags and
// magnitudes are still those of the whole spectrum.  Stage 1 starts with a pilot (one row tile
// of every column tile) that says where the energy is: a window with nearly all of it in one
// column tile gets stage 1 and stage 2 of that tile only, cell by cell (row tile x column
// tile) with the check after each; a window with nearly all of it in a dozen columns of
// different tiles (two tones meeting) gets one tile made of its sixteen strongest columns;
// other windows get stage 1 of all tiles (or all but the weakest) and stage 2 in the order
// of the pilot's energies.  A window that fails the check after a partial pass is done again
// with more.  Before any of that, a window in which the pilot finds most of the energy in the
// column of the separator tone, or in a column that holds other tones of the scheme, has one bin
// computed alone (the separator's, or that of the tone whose row frequency fits the pilot's rows;
// chunk sums on the vector ALU) and is decided from it if the rest of the window's energy cannot
// compete: three windows in five of a transmission.  LDS and global-memory instructions issue beside the wave's matrix
// instructions (vector-ALU ones do not), so sample reads and the next window's DMA pieces are
// placed one after each MFMA.
//
// Decisions must equal the reference's float64 FFT.  The per-lane running top three are
// kept as integer keys (|Z|^2 bits with the low 12 bits replaced by the bin), and the kernel
// sums |Z|^2 over all its outputs: by Parseval that is the energy of the window, and the
// float32 error of every output is bounded by a multiple of its square root (derivation at
// launch_detect).  Windows whose runner-up comes within that bound of the largest bin (plus
// the 2^-11 the keys drop) are re-evaluated in float64: the two candidate bins by direct
// summation with exactly reduced phases; if a third bin is that close too, the whole
// spectrum.  Windows whose largest |Z|^2 is zero or below 1e-30, whose energy is above 1e30,
// or whose searched maximum holds less than 1e-6 of the energy take the float64 path too
// (all-zero windows end there without a spectrum), so the result does not depend on how loud
// the capture is or on where its energy sits.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <utility>

#include "wn_common.h"
#include "wn_dma.h"

// 1: the pilot keeps the folded operands of row-tile pair 0 for stage 1 (44 more registers, 509 of
// a SIMD lane's 512 in all: about 1.5 % faster alone); 0: stage 1 gathers and folds that pair again
// (448 registers, which leaves room on every SIMD for a wave of a small kernel on another stream).
// 1: the circular extension of a window (its first samples again behind its last) is copied inside LDS
// when the window has landed; 0: it is fetched a second time by 4-byte DMA pieces (thirteen more memory
// instructions per window: about 1 % slower on the round trip)
#ifndef WN_EXT_COPY
#define WN_EXT_COPY 1
#endif
// 1: while one bin is computed alone, the next window's transfer follows the chunk reads piece by piece
// (a window that is not decided that way after all has to be fetched again); 0: the transfer starts when
// the window is decided (about 3 % slower on the round trip: the transfers of a CU then come in bursts)
#ifndef WN_BIN_PIPE
#define WN_BIN_PIPE 1
#endif
#ifndef WN_DETECT_KEEP0
#define WN_DETECT_KEEP0 1
#endif
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

#define FLAG_TOP2 (1 << 30)   // two candidate bins: this one and the one in the class word
#define FLAG_FULL (1 << 29)   // the whole spectrum
#define FLAG_TOP3 (1 << 28)   // three candidates: the third in the magnitude word (as an integer)
#define FLAG_MASK (FLAG_TOP2 | FLAG_FULL | FLAG_TOP3)
#define KEY_BIN 0xFFFu   // low bits of a key hold the bin
#define EP_UNUSED 0x8000u

template <int I>
using ic = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(ic<I>{}), ...);
}
// f(ic<0>), f(ic<1>), ... f(ic<N-1>): a loop whose index is a constant in every iteration
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int N1, int N2>
struct Plan {
    static_assert(N1 % 2 == 1 && N2 % 2 == 1, "mirrored-pair folding needs odd factors");
    static constexpr int N = N1 * N2;
    static constexpr int H1 = N1 / 2;             // a' and k1 run over [0, H1]
    static constexpr int H2 = N2 / 2;             // b' and k2 run over [0, H2]
    static constexpr int TA = (H1 + 1 + 15) / 16; // 16-row tiles over a' (stage 1) / k1 (stage 2)
    static constexpr int T1 = 2 * TA;             // stage-1 row tiles: sums over (a', n1-a'), then differences
    static constexpr int KS1 = (H2 + 1 + 3) / 4;  // stage-1 k-steps; lane group g of k-step ks has b' = ks + KS1*g
    static constexpr int K2 = H2 + 1;             // kept output columns k2
    static constexpr int C2T = (K2 + 15) / 16;    // 16-column tiles over k2
    static constexpr int KS2 = (H1 + 1 + 3) / 4;  // stage-2 k-steps; lane group g of k-step s has a' = 4s + g
    // window staging: SM floats by 16-byte DMA pieces, then NDW 4-byte pieces that finish the
    // window and append its first EXT samples again (stage 1 reads up to EXT past the end)
    static constexpr int SM = (N / 4) * 4;
    static constexpr int NV4 = SM / 4;
    static constexpr int EXT = N1 * (KS1 - 1);
    static constexpr int NDW = (3 + N + EXT - SM + 63) / 64;
    static constexpr int XW = (SM + 64 * NDW + 3) & ~3;   // floats of LDS per wave
    static_assert(16 * TA <= N1 && KS2 <= 4 * TA && C2T >= 2 && C2T <= 4 && N / 2 <= (int)KEY_BIN &&
                      256 * (NDW - 1) < 4096 && 4 * N1 * (KS1 - 1) < 65536,
                  "plan limits");
    // pilot: bf16 matrix instructions of K = 32, k-steps 8m .. 8m+7 of lane group g in slots 8g .. 8g+7
    static constexpr int NM = (KS1 + 7) / 8;
    // LDS map (bytes): class table, the pilot's B operands (16 bytes per lane for each of
    // NM x C2T x {cos, -sin}), two small tables, then one window buffer per wave
    static constexpr int OFF_LUT = 0;
    static constexpr int OFF_PTW = OFF_LUT + (((N / 2) * 2 + 15) & ~15);
    // for windows that take sixteen chosen spectrum columns: the stage-1 twiddle table and the bin
    // term (k2*E2) mod N of every kept column
    static constexpr int OFF_TW2 = OFF_PTW + NM * C2T * 2 * 1024;
    static constexpr int OFF_CB = OFF_TW2 + ((N2 * 8 + 15) & ~15);
    // for the one-bin test of separator windows: exp(2*pi*i*kc*m/N) for the 256 positions m of a chunk,
    // then (cos, sin, -sin, cos) of exp(2*pi*i*kc*256*j/N) for every chunk j
    static constexpr int NCH = (N + 3 + 255) / 256;
    static constexpr int OFF_SEP = OFF_CB + ((K2 * 4 + 15) & ~15);
    // for the one-bin test of a scheme tone: cos(2*pi*m/n1); the scheme's tones by spectrum column (8 entries
    // each: bin | folded row frequency << 16); exp(2*pi*i*128*a/N) and exp(2*pi*i*b/N), b < 128; one table
    // of chunk factors per wave
    static constexpr int NTH = (N + 127) / 128;
    static constexpr int OFF_TW1C = OFF_SEP + 256 * 8 + NCH * 16;
    static constexpr int OFF_CAND = OFF_TW1C + ((N1 * 4 + 15) & ~15);
    static constexpr int OFF_TH = OFF_CAND + K2 * 8 * 4;
    static constexpr int OFF_TL = OFF_TH + NTH * 8;
    static constexpr int OFF_WU = OFF_TL + 128 * 8;
    static constexpr int OFF_XW = OFF_WU + 4 * NCH * 16;
    static constexpr size_t LDS = (size_t)OFF_XW + (size_t)4 * XW * 4;
    // idempotents of the index map: bin of Z[j][k2] is (j*E1 + k2*E2) mod N
    static constexpr int idem(int own, int other) {
        for (int x = other;; x += other)
            if (x % own == 1) return x;
    }
    static constexpr int E1 = idem(N1, N2), E2 = idem(N2, N1);
    // The one bin in [0, N/2] the reference never looks at is N/2 (N odd).  Its output sits at
    // (k1x, k2x) on the plus side (Z[k1]) or the minus side (Z[n1-k1]) of a sum pair.
    static constexpr int XB = ((N / 2) % N2 <= H2) ? N / 2 : N - N / 2;
    static constexpr int XK2 = XB % N2, XJ = XB % N1;
    static constexpr bool XMINUS = XJ > H1;
    static constexpr int XK1 = XMINUS ? N1 - XJ : XJ;
};

// The three largest keys seen.  A key is the bit pattern of a non-negative float with its
// low 12 bits replaced by the bin: keys order like the values (to 2^-11), ties by bin.
struct Top3 {
    unsigned k1, k2, k3;
    unsigned mx;   // bit pattern of the exact largest value (non-negative floats order like their bits)
};
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// k1 >= k2 >= k3 before and after: the new third is the median of (k2, k3, key), the new
// second the median of (k1, k2, key)
__device__ __forceinline__ void top3_insert(Top3 &t, unsigned key) {
    t.k3 = med3_u32(t.k2, t.k3, key);
    t.k2 = med3_u32(t.k1, t.k2, key);
    t.k1 = max(t.k1, key);
}

// The matrix instruction, written out so that the register files are fixed: stage 1 reads its
// B operand (constant twiddles) and keeps its sums in the accumulation registers; stage 2
// reads those sums as its B operand where they are and keeps its own sums in ordinary
// registers, where the arg-max code reads them.  (Left to the compiler, operands are copied
// between the two files inside the window loop.)  The compiler does not see inside: the
// code below keeps 16 wait states between the last MFMA writing a register and the first
// other instruction reading it.
__device__ __forceinline__ void mfma1_acc(floatx4 &y, float a, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(y) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void mfma1_new(floatx4 &y, float a, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&a"(y) : "v"(a), "a"(b_acc));
}
// the same with the B operand in an ordinary register
__device__ __forceinline__ void mfma1v_acc(floatx4 &y, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(y) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma1v_new(floatx4 &y, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&a"(y) : "v"(a), "v"(b));
}
// declares an accumulation register written (no instruction): the statements below update it in place
__device__ __forceinline__ void acc_declare(floatx4 &y) { asm volatile("" : "=a"(y)); }
// One k-step of stage 1 for one column tile (six row tiles, real and imaginary part: twelve matrix
// instructions), skipped as a whole when bit BIT of `need` is clear.  The branch sits inside the
// one asm statement: to the compiler the accumulation registers are updated in place whatever
// happens, so it has nothing to copy where the two ways meet (the same instructions under an
// `if` of the source made it spill over a hundred registers).
template <bool FIRST, int BIT>
__device__ __forceinline__ void mfma1_tile_if(unsigned need, floatx4 (&yr)[6], floatx4 (&yi)[6], const float (&as)[6],
                                              const float (&ad)[6], float tc_acc, float ts_acc) {
#define WN_T(I, Y, A, B) "v_mfma_f32_16x16x4_f32 %" #Y ", %" #A ", %" #B ", " I "\n\t"
    if constexpr (FIRST)
        asm volatile("s_bitcmp1_b32 %26, %27\n\ts_cbranch_scc0 9f\n\t"
                     WN_T("0", 0, 12, 24) WN_T("0", 6, 18, 25) WN_T("0", 1, 13, 24) WN_T("0", 7, 19, 25)
                     WN_T("0", 2, 14, 24) WN_T("0", 8, 20, 25) WN_T("0", 3, 15, 24) WN_T("0", 9, 21, 25)
                     WN_T("0", 4, 16, 24) WN_T("0", 10, 22, 25) WN_T("0", 5, 17, 24) WN_T("0", 11, 23, 25) "9:"
                     : "+a"(yr[0]), "+a"(yr[1]), "+a"(yr[2]), "+a"(yr[3]), "+a"(yr[4]), "+a"(yr[5]), "+a"(yi[0]),
                       "+a"(yi[1]), "+a"(yi[2]), "+a"(yi[3]), "+a"(yi[4]), "+a"(yi[5])
                     : "v"(as[0]), "v"(as[1]), "v"(as[2]), "v"(as[3]), "v"(as[4]), "v"(as[5]), "v"(ad[0]), "v"(ad[1]),
                       "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "a"(tc_acc), "a"(ts_acc), "s"(need), "n"(BIT)
                     : "scc");
    else
        asm volatile("s_bitcmp1_b32 %26, %27\n\ts_cbranch_scc0 9f\n\t"
                     WN_T("%0", 0, 12, 24) WN_T("%6", 6, 18, 25) WN_T("%1", 1, 13, 24) WN_T("%7", 7, 19, 25)
                     WN_T("%2", 2, 14, 24) WN_T("%8", 8, 20, 25) WN_T("%3", 3, 15, 24) WN_T("%9", 9, 21, 25)
                     WN_T("%4", 4, 16, 24) WN_T("%10", 10, 22, 25) WN_T("%5", 5, 17, 24) WN_T("%11", 11, 23, 25) "9:"
                     : "+a"(yr[0]), "+a"(yr[1]), "+a"(yr[2]), "+a"(yr[3]), "+a"(yr[4]), "+a"(yr[5]), "+a"(yi[0]),
                       "+a"(yi[1]), "+a"(yi[2]), "+a"(yi[3]), "+a"(yi[4]), "+a"(yi[5])
                     : "v"(as[0]), "v"(as[1]), "v"(as[2]), "v"(as[3]), "v"(as[4]), "v"(as[5]), "v"(ad[0]), "v"(ad[1]),
                       "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "a"(tc_acc), "a"(ts_acc), "s"(need), "n"(BIT)
                     : "scc");
#undef WN_T
}
__device__ __forceinline__ void mfma2_acc(floatx4 &acc, float a, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void mfma2_new(floatx4 &acc, float a, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b_acc));
}
// the same with the A operand in an accumulation register (half of the stage-2 twiddles live there)
__device__ __forceinline__ void mfma2a_acc(floatx4 &acc, float a_acc, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a_acc), "a"(b_acc));
}
__device__ __forceinline__ void mfma2a_new(floatx4 &acc, float a_acc, float b_acc) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(a_acc), "a"(b_acc));
}
// The pilot's matrix instruction: bf16 operands (8 per lane: A[row lane&15][k = 8(lane>>4) + j],
// B[k = 8(lane>>4) + j][column lane&15]), f32 sums laid out like those of the f32 instruction.
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_bf16_new(floatx4 &acc, uintx4 a, uintx4 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_bf16_acc(floatx4 &acc, uintx4 a, uintx4 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// (bf16(lo), bf16(hi)) in one register
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// |a - b| for a per-lane a and a wave-uniform b
__device__ __forceinline__ unsigned absdiff_u32(unsigned a, unsigned b) {
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }
// a value that lives in an accumulation register from here on
__device__ __forceinline__ void acc_const(float &dst, float v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(v)); }
// ties a register's later readers to this point of the (ordered) asm sequence
__device__ __forceinline__ void after_drain(floatx4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void after_drain_a(floatx4 &v) { asm volatile("" : "+a"(v)); }

// (v.x + v.y, v.x - v.y) in one packed instruction
__device__ __forceinline__ floatx2 pk_butterfly(floatx2 v) {
    floatx2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(v));
    return r;
}

// the DPP-selected lane's value where the row mask writes, 0 elsewhere (0 is neutral for the
// unsigned maxima and key merges it feeds, and lets the move fold into the instruction using it)
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);
}

// the DPP-selected lane's value where the row mask writes, 0 elsewhere
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, false));
}

// Sum / maximum over the wave, valid in lane 63 (same DPP steps as the top-three merge: pairs,
// quads, eights, sixteens inside each row, then the rows collected in the last one).
__device__ __forceinline__ float wave_sum63(float v) {
    v += dpp_f0<0xB1, 0xF>(v);
    v += dpp_f0<0x4E, 0xF>(v);
    v += dpp_f0<0x141, 0xF>(v);
    v += dpp_f0<0x140, 0xF>(v);
    v += dpp_f0<0x142, 0xA>(v);
    v += dpp_f0<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_max63(unsigned v) {
    v = max(v, dpp_u<0xB1, 0xF>(v));
    v = max(v, dpp_u<0x4E, 0xF>(v));
    v = max(v, dpp_u<0x141, 0xF>(v));
    v = max(v, dpp_u<0x140, 0xF>(v));
    v = max(v, dpp_u<0x142, 0xA>(v));
    v = max(v, dpp_u<0x143, 0xC>(v));
    return v;
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ unsigned lane63(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
template <class T>
__device__ __forceinline__ float lane_f(float v, T lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

template <int N1, int N2, bool PROF>
__global__ __launch_bounds__(256, 1) void detect_kernel(const float *__restrict__ audio, long long n_streams,
                                                        long long stream_stride, long long first_sample,
                                                        long long n_windows, const float2 *__restrict__ g_tw1,
                                                        const float2 *__restrict__ g_tw2,
                                                        const uint16_t *__restrict__ lut, float keep,
                                                        float err_lin, float err_sq, float one_share, float two_share,
                                                        float pick_share, int sep_bin, float sep_share,
                                                        const int *__restrict__ g_cand, float bin_share, int prof_mode,
                                                        uint16_t *__restrict__ out_cls, float *__restrict__ out_mag,
                                                        int32_t *__restrict__ out_bin,
                                                        unsigned long long *__restrict__ stats) {
    typedef Plan<N1, N2> P;
    constexpr int N = P::N, TA = P::TA, T1 = P::T1, C2T = P::C2T, KS1 = P::KS1, KS2 = P::KS2;
    constexpr int H1 = P::H1, H2 = P::H2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float *smf = (float *)smem;
    uint16_t *lut_s = (uint16_t *)(smem + P::OFF_LUT);  // class word per bin
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < N / 2; i += 256) lut_s[i] = lut[i];
    // the pilot's B operands: entry (m, c, part) of lane l holds, for j < 8, the bf16 twiddle of
    // folded sample column b' = (8m + j) + KS1*(l>>4) and spectrum column k2 = 16c + (l&15)
    // (cos, halved for b' = 0, or -sin; zero for padding)
    for (int e = threadIdx.x; e < P::NM * C2T * 2 * 64; e += 256) {
        const int idx = e >> 6, ln = e & 63, part = idx & 1, c = (idx >> 1) % C2T, m = idx / (2 * C2T);
        const int k2 = 16 * c + (ln & 15);
        unsigned h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ks = 8 * m + j, bp = ks + KS1 * (ln >> 4);
            float v = 0.f;
            if (ks < KS1 && bp <= H2 && k2 <= H2) {
                const float2 t = g_tw2[(bp * k2) % N2];
                v = part ? t.y : (bp == 0 ? 0.5f * t.x : t.x);
            }
            const unsigned u = __float_as_uint(v);
            h[j] = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;   // round to nearest even (finite values)
        }
        uintx4 w4 = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        *(uintx4 *)(smem + P::OFF_PTW + idx * 1024 + ln * 16) = w4;
    }

    for (int i = threadIdx.x; i < N2; i += 256) ((float2 *)(smem + P::OFF_TW2))[i] = g_tw2[i];
    for (int i = threadIdx.x; i < P::K2; i += 256) ((int *)(smem + P::OFF_CB))[i] = (int)(((long long)i * P::E2) % N);
    if (bin_share > 0.f) {
        for (int i = threadIdx.x; i < N1; i += 256) ((float *)(smem + P::OFF_TW1C))[i] = g_tw1[i].x;
        for (int i = threadIdx.x; i < P::K2 * 8; i += 256) ((int *)(smem + P::OFF_CAND))[i] = g_cand[i];
        double sn, cs;
        if (threadIdx.x < 128) {
            sincospi(2.0 * (double)threadIdx.x / (double)N, &sn, &cs);
            ((float2 *)(smem + P::OFF_TL))[threadIdx.x] = make_float2((float)cs, (float)sn);
        } else if (threadIdx.x - 128 < P::NTH) {
            sincospi(2.0 * (double)(128 * (threadIdx.x - 128)) / (double)N, &sn, &cs);
            ((float2 *)(smem + P::OFF_TH))[threadIdx.x - 128] = make_float2((float)cs, (float)sn);
        }
    }
    if (sep_share > 0.f) {
        double sn, cs;
        sincospi(2.0 * (double)(((long long)sep_bin * threadIdx.x) % N) / (double)N, &sn, &cs);
        ((float2 *)(smem + P::OFF_SEP))[threadIdx.x] = make_float2((float)cs, (float)sn);
        if (threadIdx.x < P::NCH) {
            sincospi(2.0 * (double)(((long long)sep_bin * 256 * threadIdx.x) % N) / (double)N, &sn, &cs);
            ((float4 *)(smem + P::OFF_SEP + 256 * 8))[threadIdx.x] = make_float4((float)cs, (float)sn, -(float)sn, (float)cs);
        }
    }

    const int li = lane & 15;  // A row / B column / D column
    const int lg = lane >> 4;  // A,B k index / D row group
    const long long total = n_streams * n_windows;
    const long long wstep = (long long)gridDim.x * 4;
    long long w = (long long)blockIdx.x * 4 + wave;
    const int xw_f = P::OFF_XW / 4 + wave * P::XW;     // float index of this wave's window buffer
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(smf + xw_f);

    // ---- per-lane constants, held in registers for the whole launch
    // stage-1 twiddles (B operand: row b' = ks + KS1*lg, column k2 = 16c + li), in accumulation
    // registers: (cos, -sin)(2*pi*b'*k2/n2); zero for padding rows and columns; cos halved
    // for b' = 0, whose folded sample is x2[a][0] twice
    float t2c[KS1][C2T], t2s[KS1][C2T];
    static_for<KS1>([&](auto KS) {
        static_for<C2T>([&](auto C) {
            constexpr int ks = decltype(KS)::value, c = decltype(C)::value;
            const int bp = ks + KS1 * lg, k2 = 16 * c + li;
            float2 t = g_tw2[(bp * k2) % N2];
            if (bp > H2 || k2 > H2) t = make_float2(0.f, 0.f);
            if (bp == 0) t.x *= 0.5f;
            acc_const(t2c[ks][c], t.x);
            acc_const(t2s[ks][c], t.y);
        });
    });
    // stage-2 twiddles (A operand: row k1 = 16t + li, column a' = 4s + lg), in ordinary
    // registers: (cos, sin)(2*pi*a'*k1/n1); zero for padding rows and columns; cos halved
    // for a' = 0, whose folded row is Y[0] twice
    float t1c[TA][KS2], t1s[TA][KS2];
    static_for<TA>([&](auto TT) {
        static_for<KS2>([&](auto S) {
            constexpr int tt = decltype(TT)::value, s = decltype(S)::value;
            const int k1 = 16 * tt + li, ap = 4 * s + lg;
            float2 t = g_tw1[(k1 * ap) % N1];
            if (ap > H1 || k1 > H1) t = make_float2(0.f, 0.f);
            if (ap == 0) t.x *= 0.5f;
            t1c[tt][s] = t.x;
            acc_const(t1s[tt][s], t.y);
        });
    });
    // stage-1 gather: A-operand row li of row tile t is a' = 16(t mod TA) + 4(li&3) + (li>>2),
    // i.e. result register r of lane group g holds a' = 16t + 4r + g, so that stage-2 k-step s
    // covers a' = 4s .. 4s+3; tiles t >= TA read row n1-a' (row 0 again for a' = 0: its
    // difference is zero and its sum is halved by the twiddle).  Column b' of row a is sample
    // (n2*a + n1*b') mod N and column n2-b' is sample (n2*a - n1*b') mod N; over the k-steps
    // the first index climbs by n1 from its value at ks = 0 and the second descends by n1 to
    // its value at ks = KS1-1, both without reduction thanks to the circular extension.
    int cb_p[T1], cb_m[T1];   // byte offsets from sample 0 of the window
#pragma unroll
    for (int t = 0; t < T1; ++t) {
        const int ap = 16 * (t % TA) + 4 * (li & 3) + (li >> 2);
        const int a = t < TA ? ap : (N1 - ap) % N1;
        cb_p[t] = 4 * ((N2 * a + N1 * (KS1 * lg)) % N);
        cb_m[t] = 4 * ((N2 * a + N * N1 - N1 * (KS1 - 1 + KS1 * lg)) % N);
    }
    // Energy of the window from the folded samples of stage 1: the four A-operand values of a
    // lane (sums and differences over a mirrored column pair and a mirrored row pair) are an
    // orthogonal transform of four samples scaled by 2, so their squares sum to 4 * (sum of the
    // four samples' squares).  Weights undo what is read twice or is padding: row a' = 0 is its
    // own mirror (1/2), rows a' > h1 repeat other rows (0); column b' = 0 is its own mirror
    // (1/2, first k-step, lane group 0), column b' > h2 is padding (0, last k-step).
    float w_row[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int ap = 16 * t + 4 * (li & 3) + (li >> 2);
        w_row[t] = ap > H1 ? 0.f : (ap == 0 ? 0.5f : 1.f);
    }
    const float w_first = lg == 0 ? 0.5f : 1.f;
    const float w_last = (KS1 - 1 + KS1 * lg) > H2 ? 0.f : 1.f;
    unsigned long long cells_done = 0;  // (row tile, column) products of stage 2 this wave computed (statistics)
    __syncthreads();

    // Bins of this lane's outputs (tile, register, column tile, plus/minus side), kept in
    // registers for the whole launch.  Accumulator element (output tile t, register r, column
    // tile c) of lane l holds the sums for k1 = 16t + 4(l>>4) + r, k2 = 16c + (l&15); from them
    // come Z[k1][k2] (side 0) and Z[n1-k1][k2] (side 1).  The bin of Z[j][k2] is
    // (j*E1 + k2*E2) mod N folded to [0, N/2].  Padding rows and columns (whose sums are
    // exactly zero) carry the bin field alone; the bin the reference never looks at and the
    // second copy of self-mirrored outputs (k1 = 0 or k2 = 0) are marked unused.  The two sides of
    // an element share a register: plus side in the low half, minus side in the high half.
    unsigned tag[TA][4][C2T];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < C2T; ++c) {
                unsigned both = 0u;
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    const int k1 = 16 * t + 4 * lg + r, k2 = 16 * c + li;
                    const int j = sg ? (N1 - k1) % N1 : k1;
                    const int kb = (int)(((long long)j * P::E1 + (long long)k2 * P::E2) % N);
                    const int kk = min(kb, N - kb);
                    const bool ok = (k1 <= H1) & (k2 <= H2) & (kk < N / 2) & (!sg | ((k1 != 0) & (k2 != 0)));
                    const bool pad = (k1 > H1) | (k2 > H2);
                    both |= (ok ? (unsigned)kk : (pad ? KEY_BIN : (EP_UNUSED | KEY_BIN))) << (16 * sg);
                }
                tag[t][r][c] = both;
            }

    // ---- window staging by LDS-DMA
    auto fetch = [&](long long s, long long wi) {
        const long long g0 = s * stream_stride + first_sample + wi * (long long)N;
        const int off = (int)(g0 & 3);
        const float *x0 = audio + g0;
        // 16-byte pieces: floats [0, SM) of the buffer <- samples [-off, SM - off)
        const float *src = x0 - off + 4 * lane;
        constexpr int FULL = P::NV4 / 64, REM = P::NV4 % 64;
        static_for<(FULL + 3) / 4>([&](auto G) {
            constexpr int g = decltype(G)::value, cnt = FULL - 4 * g < 4 ? FULL - 4 * g : 4;
            dma_group<cnt, 1024, true>(src + 1024 * g, __builtin_amdgcn_readfirstlane(lds_base + 4096u * g));
        });
        if constexpr (REM > 0) {
            if (lane < REM)
                dma_group<1, 1024, true>(src + 256 * FULL, __builtin_amdgcn_readfirstlane(lds_base + 1024u * FULL));
        }
        // 4-byte pieces: floats [SM, SM + 64*NDW) <- the last samples of the window, then its
        // first ones again.  Only the first piece contains the seam.
        const int i0 = P::SM + lane - off;
        dma_group<1, 256, false>(x0 + (i0 < N ? i0 : i0 - N), __builtin_amdgcn_readfirstlane(lds_base + 4u * P::SM));
        const float *ext = x0 + (P::SM + 64 + lane - off - N);
        if constexpr (!WN_EXT_COPY) static_for<(P::NDW - 1 + 3) / 4>([&](auto G) {
            constexpr int g = decltype(G)::value, cnt = P::NDW - 1 - 4 * g < 4 ? P::NDW - 1 - 4 * g : 4;
            dma_group<cnt, 256, false>(ext + 256 * g,
                                       __builtin_amdgcn_readfirstlane(lds_base + 4u * (P::SM + 64) + 1024u * g));
        });
        return off;
    };
    // The same transfer as single instructions, issued between the matrix instructions of the first
    // stage-2 cell of a window (one per MFMA: the address work of a piece runs while the matrix
    // pipe is busy).  prepare() issues the two pieces that need per-lane handling and keeps the
    // scalar bases of the rest; piece(k) issues the k-th of the others.
    constexpr int DMA_FULL = P::NV4 / 64, DMA_REM = P::NV4 % 64, DMA_PIECES = DMA_FULL + P::NDW - 1;
    static_assert(DMA_PIECES <= 4 * KS2, "one piece per matrix instruction of a cell");
    const float *nx_src = audio, *nx_ext = audio;
    unsigned m0_keep = 0;
    // (the two pieces that need per-lane handling land in the last 256 floats a window occupies: a caller that
    // is still reading there issues them afterwards, with fetch_late)
    auto fetch_late = [&](int off) {
        if constexpr (DMA_REM > 0) {
            if (lane < DMA_REM)
                dma_group<1, 1024, true>(nx_src + 4 * lane + 256 * DMA_FULL,
                                         __builtin_amdgcn_readfirstlane(lds_base + 1024u * DMA_FULL));
        }
        const int i0 = P::SM + lane - off;
        dma_group<1, 256, false>(nx_src + off + (i0 < N ? i0 : i0 - N), __builtin_amdgcn_readfirstlane(lds_base + 4u * P::SM));
    };
    auto fetch_prepare = [&](long long s, long long wi, auto LATE) {
        const long long g0 = s * stream_stride + first_sample + wi * (long long)N;
        const int off = (int)(g0 & 3);
        const float *x0 = audio + g0;
        nx_src = x0 - off;
        nx_ext = x0 + (P::SM + 64 - off - N);
        if constexpr (!decltype(LATE)::value) fetch_late(off);
        m0_keep = wn_m0_get();
        return off;
    };
    auto fetch_piece = [&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < DMA_FULL) {
            if constexpr (k % 4 == 0) wn_m0_set(__builtin_amdgcn_readfirstlane(lds_base + 4096u * (k / 4)));
            dma_piece<1024 * (k % 4), true>(nx_src + 1024 * (k / 4), 16u * lane);
        } else if constexpr (k < DMA_PIECES && !WN_EXT_COPY) {
            constexpr int j = k - DMA_FULL;
            if constexpr (j % 4 == 0)
                wn_m0_set(__builtin_amdgcn_readfirstlane(lds_base + 4u * (P::SM + 64) + 1024u * (j / 4)));
            dma_piece<256 * (j % 4), false>(nx_ext + 256 * (j / 4), 4u * lane);
        }
        if constexpr (k == DMA_PIECES - 1) wn_m0_set(m0_keep);
    };
    auto extend = [&]() {
        if constexpr (WN_EXT_COPY) {
            // float f of the buffer holds sample f - off; the extension repeats samples 0 .. from float off + N on
            float *bufw = smf + xw_f;
            float keepf[P::NDW - 1];
#pragma unroll
            for (int j = 0; j < P::NDW - 1; ++j) keepf[j] = bufw[P::SM + 64 + 64 * j + lane - N];
#pragma unroll
            for (int j = 0; j < P::NDW - 1; ++j) bufw[P::SM + 64 + 64 * j + lane] = keepf[j];
        }
    };
    // (stream, window) of the window in work and of the one after it, advanced without divisions
    const long long adv_s = wstep / n_windows, adv_w = wstep - adv_s * n_windows;
    long long cur_s = w / n_windows, cur_w = w - cur_s * n_windows;
    long long f_s = cur_s, f_w = cur_w;
    auto advance = [&]() {   // the window after this wave's next one
        cur_s = f_s;
        cur_w = f_w;
        f_s += adv_s;
        f_w += adv_w;
        if (f_w >= n_windows) {
            f_w -= n_windows;
            f_s += 1;
        }
    };

    // Results of a window are stored one window later, just ahead of the next DMA issue, so
    // that the wait for DMA data never waits for a store just issued.
    bool pend = false;
    long long pend_w = 0;
    int pend_bin = 0;
    unsigned pend_cls = 0;
    float pend_mag = 0.f;
    auto flush = [&]() {
        if (pend && lane == 63) {
            out_bin[pend_w] = pend_bin;
            out_cls[pend_w] = (uint16_t)pend_cls;
            out_mag[pend_w] = pend_mag;
        }
    };

    // PROF build only: shader clocks spent per phase of the window loop, summed over this wave's windows
    // (the sums live in LDS behind the window buffers, 32 per wave: in registers they cost the build the
    // registers its kernel needs)
    volatile unsigned long long *prof = (volatile unsigned long long *)(smem + P::LDS) + 32 * wave;
    unsigned long long prof_t = 0;
    if constexpr (PROF) {
        for (int i = 0; i < 32; ++i) prof[i] = 0ull;
    }
    const unsigned long long prof_r0 = PROF ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long prof_c0 = PROF ? __builtin_amdgcn_s_memtime() : 0ull;
    // (prof_mode 2: only the clocks per window and way, taken at the end of a window; the laps in between
    // cost a window about a tenth more)
    auto lap = [&](auto I) {
        if constexpr (PROF) {
            __builtin_amdgcn_sched_barrier(0);
            if (prof_mode == 1 || decltype(I)::value == 4) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if (prof_mode == 1) prof[decltype(I)::value] = prof[decltype(I)::value] + (now - prof_t);
                prof_t = now;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // PROF build only: windows and clocks by the way a window went (0 one column tile; 1 the long way without
    // the weakest tile; 2 the long way with every tile; 3 one tile, then the long way; 4 the long way twice;
    // 5 sixteen chosen columns; 6 those, then the long way; 7 the separator's bin alone)
    // (prof[12 + way]: windows, prof[20 + way]: clocks)
    unsigned long long path_t0 = 0;
    unsigned long long s1_units = 0;    // (row tile, real or imaginary part, column) products of stage 1 this wave computed (statistics)
    unsigned long long redone = 0;      // windows done a second time, the whole way
    unsigned long long whole_n = 0;     // passes the long way (windows done twice included)
    unsigned long long two_n = 0;       // of them: with one column tile left out
    unsigned long long pick_n = 0;      // windows that took sixteen chosen columns
    unsigned long long sep_n = 0;       // windows decided from the separator's bin alone
    unsigned long long tone_n = 0;      // windows decided from the bin of another tone of the scheme alone
    int next_off = 0;
    bool sep_tried = false;   // this window has been through the separator's test and came back
    if (w < total) {
        next_off = fetch(f_s, f_w);
        advance();
    }
    while (w < total) {
        if constexpr (PROF) path_t0 = prof_t = __builtin_amdgcn_s_memtime();
        const int xb = 4 * (xw_f + next_off);   // byte address of sample 0 of this window
        int ad_p[T1], ad_m[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            ad_p[t] = xb + cb_p[t];
            ad_m[t] = xb + cb_m[t];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this window's DMA has landed
        extend();
        lap(ic<0>{});

        // ------------------------------------------------------------------ stage 1
        // results: S (row tiles < TA) and D (row tiles >= TA), real and imaginary.  Whatever part of
        // stage 1 writes them, the part of stage 2 that reads them follows on the same path through
        // the code below, so that no two ways of producing them ever meet (where they would, the
        // compiler copies whole sets of accumulation registers around).
        floatx4 Yr[C2T][T1], Yi[C2T][T1];
        // samples and A operands as register pairs (row tile t, row tile t + TA), so that the
        // folds are packed f32 instructions: 4 per pair of row tiles instead of 8
        floatx2 raw_p[2][TA], raw_m[2][TA], xs[TA], xd[TA];   // samples: two sets, for even and odd k-steps
        floatx2 e4[TA];            // running sums of the squared A operands, per row-tile pair
        floatx2 sx[TA];            // running sums of the A operands (.x: the sum of the four samples)
        floatx2 xs0[KS1], xd0[KS1];   // A operands of row-tile pair 0 for every k-step, folded by the pilot
        // stage-1 twiddles of the sixteen chosen columns: read from the table k-step by k-step (byte offset
        // pk_i8 of this lane's entry for the next k-step, climbing by pk_k8 modulo the table)
        unsigned pk_i8 = 0u, pk_k8 = 0u;
        // Stage 1 comes in three cuts.  WHOLE: every row tile of every column tile.  ONE<c>: every row
        // tile of column tile c only, for windows whose energy the pilot (below) finds in one
        // column tile.  PICK: every row tile of one tile made of sixteen columns chosen for this window
        // (twiddles in dtc / dts, results where column tile 0 has its own).
        // The sample reads of k-step ks+2 are issued one after each matrix instruction of k-step
        // ks (an LDS instruction issues while the matrix pipe works on the one before it; the
        // vector instructions of the folds do not), into the register set that the fold of
        // k-step ks has already consumed.
        // (Doing the k-steps in pairs, for half as many changes between matrix and vector work, was
        // slower: 24 LDS reads in one go exceed the 15 the wave may have outstanding.)
        constexpr int WHOLE = 1, ONE = 2, PICK = 3;
        auto stage1 = [&](auto CUT_, auto CSEL_, unsigned need) {   // need (WHOLE only): the column tiles to compute, wave-uniform
            constexpr int CUT = decltype(CUT_)::value, CSEL = decltype(CSEL_)::value;
            // row-tile pairs gathered here: all of them, or all but pair 0 when the pilot kept its operands
            constexpr int P0 = WN_DETECT_KEEP0 ? 1 : 0, NP = TA - P0;
            auto rd1 = [&](auto KS, auto M) {   // issue sample read M (of 4*NP) of k-step KS
                constexpr int ks = decltype(KS)::value, m = decltype(M)::value, pr = P0 + m / 4, q = m % 4;
                if constexpr (ks < KS1 && m < 4 * NP) {
                    if constexpr (q == 0) raw_p[ks & 1][pr][0] = *(const float *)(smem + ad_p[pr] + 4 * N1 * ks);
                    if constexpr (q == 1) raw_m[ks & 1][pr][0] = *(const float *)(smem + ad_m[pr] + 4 * N1 * (KS1 - 1 - ks));
                    if constexpr (q == 2) raw_p[ks & 1][pr][1] = *(const float *)(smem + ad_p[pr + TA] + 4 * N1 * ks);
                    if constexpr (q == 3) raw_m[ks & 1][pr][1] = *(const float *)(smem + ad_m[pr + TA] + 4 * N1 * (KS1 - 1 - ks));
                }
            };
            auto rd = [&](auto KS) {   // all sample reads of k-step KS
                static_for<4 * NP>([&](auto M) { rd1(KS, M); });
            };
            auto fold = [&](auto KS) {   // samples of k-step KS -> its A operands: fold columns, then rows
                constexpr int ks = decltype(KS)::value;
                static_for<NP>([&](auto T) {
                    constexpr int t = P0 + decltype(T)::value;
                    const floatx2 sab = raw_p[ks & 1][t] + raw_m[ks & 1][t], dab = raw_p[ks & 1][t] - raw_m[ks & 1][t];   // (sa, sb), (da, db)
                    xs[t] = pk_butterfly(sab);                                            // (sa + sb, sa - sb)
                    xd[t] = pk_butterfly(dab);
                    if constexpr (ks == 0) {
                        e4[t] = (xs[t] * xs[t] + xd[t] * xd[t]) * w_first;
                        sx[t] = xs[t] * w_first;
                    } else if constexpr (ks == KS1 - 1) {
                        e4[t] += (xs[t] * xs[t] + xd[t] * xd[t]) * w_last;
                        sx[t] += xs[t] * w_last;
                    } else {
                        e4[t] += xs[t] * xs[t];
                        e4[t] += xd[t] * xd[t];
                        sx[t] += xs[t];
                    }
                    // keeps the sums here, between the matrix instructions of two k-steps (left to
                    // itself the compiler holds every k-step's operands and squares them after stage 1)
                    asm volatile("" : "+v"(e4[t]), "+v"(sx[t]));
                });
            };
            floatx2 dtw[2];   // PICK: (cos, -sin) of this lane's column for k-steps ks (even: [0], odd: [1])
            auto tw_rd = [&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                if constexpr (CUT == PICK && ks < KS1) {
                    const float2 t = *(const float2 *)(smem + P::OFF_TW2 + pk_i8);
                    pk_i8 += pk_k8;
                    pk_i8 = min(pk_i8, pk_i8 - 8u * N2);
                    dtw[ks & 1][0] = t.x;
                    dtw[ks & 1][1] = t.y;
                }
            };
            tw_rd(ic<0>{});
            if constexpr (CUT == WHOLE && C2T > 2) {
                static_for<C2T>([&](auto C) {
                    static_for<T1>([&](auto T) {
                        acc_declare(Yr[decltype(C)::value][decltype(T)::value]);
                        acc_declare(Yi[decltype(C)::value][decltype(T)::value]);
                    });
                });
            }
            rd(ic<0>{});
            __builtin_amdgcn_sched_barrier(0);
            fold(ic<0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KS1 > 1) rd(ic<1>{});
            __builtin_amdgcn_sched_barrier(0);
            static_for<KS1>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                static_for<C2T>([&](auto C) {
                    constexpr int c = decltype(C)::value;
                    auto tile = [&]() {
                        float pc = 0.f, ps = 0.f;   // PICK: this k-step's twiddles, padding and the halved first column applied
                        if constexpr (CUT == PICK) {
                            pc = dtw[ks & 1][0];
                            ps = dtw[ks & 1][1];
                            if constexpr (ks == 0) pc *= w_first;
                            if constexpr (ks == KS1 - 1) {
                                pc *= w_last;
                                ps *= w_last;
                            }
                            asm volatile("s_nop 1" ::"v"(pc), "v"(ps));   // just written, read by the next MFMA
                        }
                        static_for<T1>([&](auto T) {
                            constexpr int t = decltype(T)::value;
                            // row-tile pair 0 takes the operands the pilot folded and kept
                            const float as = (WN_DETECT_KEEP0 && t % TA == 0) ? xs0[ks][t / TA] : xs[t % TA][t / TA];
                            const float ad = (WN_DETECT_KEEP0 && t % TA == 0) ? xd0[ks][t / TA] : xd[t % TA][t / TA];
                            if constexpr (CUT == PICK) {
                                if constexpr (ks == 0) mfma1v_new(Yr[c][t], as, pc);
                                else mfma1v_acc(Yr[c][t], as, pc);
                                if constexpr (t == 0) tw_rd(ic<ks + 1>{});   // the next k-step's pair, while this one's matrix instructions run
                            } else {
                                if constexpr (ks == 0) mfma1_new(Yr[c][t], as, t2c[ks][c]);
                                else mfma1_acc(Yr[c][t], as, t2c[ks][c]);
                            }
                            // the sample reads of k-step ks+2 ride on the first matrix instructions of a
                            // column tile; every tile that runs issues them (the same values again:
                            // which tiles run is not known when the code is laid out)
                            rd1(ic<ks + 2>{}, ic<2 * t>{});
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (CUT == PICK) {
                                if constexpr (ks == 0) mfma1v_new(Yi[c][t], ad, ps);
                                else mfma1v_acc(Yi[c][t], ad, ps);
                            } else {
                                if constexpr (ks == 0) mfma1_new(Yi[c][t], ad, t2s[ks][c]);
                                else mfma1_acc(Yi[c][t], ad, t2s[ks][c]);
                            }
                            rd1(ic<ks + 2>{}, ic<2 * t + 1>{});
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    };
                    if constexpr (CUT == WHOLE && C2T > 2) {
                        if constexpr (T1 == 6) {
                            float asv[6], adv[6];
                            static_for<6>([&](auto T) {
                                constexpr int t = decltype(T)::value;
                                asv[t] = (WN_DETECT_KEEP0 && t % TA == 0) ? xs0[ks][t / TA] : xs[t % TA][t / TA];
                                adv[t] = (WN_DETECT_KEEP0 && t % TA == 0) ? xd0[ks][t / TA] : xd[t % TA][t / TA];
                            });
                            mfma1_tile_if<ks == 0, c>(need, Yr[c], Yi[c], asv, adv, t2c[ks][c], t2s[ks][c]);
                            // the sample reads of k-step ks+2, a third of them behind each tile
                            static_for<4 * NP>([&](auto M) {
                                if constexpr (decltype(M)::value % C2T == c) rd1(ic<ks + 2>{}, M);
                            });
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if constexpr (CUT == WHOLE || c == CSEL) {
                        tile();
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks + 1 < KS1) {
                    fold(ic<ks + 1>{});                        // operands of k-step ks+1
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 1" ::: "memory");      // operands just written, read by the next MFMA
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        // every LDS read of this window has returned (its values fed the MFMAs of stage 1): store
        // the previous window's results, then start the next window's DMA so that it lands
        // while stage 2 runs
        const long long wn = w + wstep;
        auto next_window_dma = [&]() {
            flush();
            if (wn < total) next_off = fetch(f_s, f_w);
        };
        // Energy of the half spectrum the reference searches, E = sum_{k <= N/2} |X_k|^2
        // = (N * sum x^2 + X_0^2) / 2 (N odd), from the samples gathered by stage 1: an upper
        // bound for every bin and the scale of the float32 error.  Wave-uniform.
        float e_half = 0.f, sqrt_e = 0.f;
        auto energies = [&]() {
            float e4l = 0.f, sxl = 0.f;
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                e4l = fmaf(e4[t].x + e4[t].y, w_row[t], e4l);
                sxl = fmaf(sx[t].x, w_row[t], sxl);
            }
            const float sx2 = 0.25f * lane63(wave_sum63(e4l)), x0 = lane63(wave_sum63(sxl));
            e_half = 0.5f * fmaf((float)N, sx2, x0 * x0) * (1.f + 1e-5f);
            sqrt_e = __builtin_amdgcn_sqrtf(e_half);
        };

        // The pilot says where the energy is before any full-precision work is done: the sum row
        // tile 0 (rows a' < 16 and their mirror images) of every column tile in bf16, whose matrix
        // instructions are sixteen times as fast as the f32 ones.  Its result only chooses which
        // column tile stage 1 computes and in which order stage 2 goes: the energy check makes
        // the window's result independent of it, and everything that reaches a decision or a
        // magnitude is computed in f32 below.  (With this row tile in f32 and its results kept for
        // stage 2 the pilot took 3.4 k of a window's 15.9 k clocks.)
        floatx4 pr[C2T], pi[C2T];
        {
            unsigned pa_s[4 * P::NM], pa_d[4 * P::NM];   // A operands: (sum, difference) of the mirrored columns, 8 k-steps per 4 registers
            // every sample read is issued before the first one is waited for (nothing else runs
            // meanwhile that could hide one read's latency behind the next one's)
            floatx2 rp[KS1], rm[KS1];   // rows a' and n1-a': column b', column n2-b'
            static_for<KS1>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                rp[ks][0] = *(const float *)(smem + ad_p[0] + 4 * N1 * ks);
                rm[ks][0] = *(const float *)(smem + ad_m[0] + 4 * N1 * (KS1 - 1 - ks));
                rp[ks][1] = *(const float *)(smem + ad_p[TA] + 4 * N1 * ks);
                rm[ks][1] = *(const float *)(smem + ad_m[TA] + 4 * N1 * (KS1 - 1 - ks));
            });
            uintx4 bc[P::NM][C2T], bs[P::NM][C2T];   // the twiddles: those of the first k-steps are read while the folds run
            auto twiddles = [&](auto MM) {
                static_for<C2T>([&](auto C) {
                    constexpr int m = decltype(MM)::value, c = decltype(C)::value;
                    bc[m][c] = *(const uintx4 *)(smem + P::OFF_PTW + ((m * C2T + c) * 2) * 1024 + lane * 16);
                    bs[m][c] = *(const uintx4 *)(smem + P::OFF_PTW + ((m * C2T + c) * 2 + 1) * 1024 + lane * 16);
                });
            };
            if constexpr (WN_DETECT_KEEP0) twiddles(ic<0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PROF) {
                if (prof_mode == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            lap(ic<8>{});
            float hs = 0.f, hd = 0.f;
            static_for<8 * P::NM>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                float sv = 0.f, dv = 0.f;
                if constexpr (ks < KS1 && !WN_DETECT_KEEP0) {
                    const floatx2 sab = rp[ks] + rm[ks], dab = rp[ks] - rm[ks];
                    sv = sab.x + sab.y;
                    dv = dab.x + dab.y;
                } else if constexpr (ks < KS1) {
                    // the fold of stage 1 for this row-tile pair, kept for its matrix instructions below
                    const floatx2 sab = rp[ks] + rm[ks], dab = rp[ks] - rm[ks];
                    xs0[ks] = pk_butterfly(sab);
                    xd0[ks] = pk_butterfly(dab);
                    if constexpr (ks == 0) {
                        e4[0] = (xs0[ks] * xs0[ks] + xd0[ks] * xd0[ks]) * w_first;
                        sx[0] = xs0[ks] * w_first;
                    } else if constexpr (ks == KS1 - 1) {
                        e4[0] += (xs0[ks] * xs0[ks] + xd0[ks] * xd0[ks]) * w_last;
                        sx[0] += xs0[ks] * w_last;
                    } else {
                        e4[0] += xs0[ks] * xs0[ks];
                        e4[0] += xd0[ks] * xd0[ks];
                        sx[0] += xs0[ks];
                    }
                    sv = xs0[ks].x;
                    dv = xd0[ks].x;
                }
                if constexpr (ks % 2 == 0) {
                    hs = sv;
                    hd = dv;
                } else if constexpr (ks - 1 < KS1) {
                    pa_s[ks / 2] = pk_bf16(hs, sv);
                    pa_d[ks / 2] = pk_bf16(hd, dv);
                } else {
                    pa_s[ks / 2] = 0u;
                    pa_d[ks / 2] = 0u;
                }
            });
            if constexpr (PROF) asm volatile("" :: "v"(pa_s[0]), "v"(pa_d[0]), "v"(pa_s[4 * P::NM - 1]), "v"(pa_d[4 * P::NM - 1]));
            lap(ic<9>{});
            static_for<P::NM>([&](auto MM) {
                constexpr int m = decltype(MM)::value;
                const uintx4 as = {pa_s[4 * m], pa_s[4 * m + 1], pa_s[4 * m + 2], pa_s[4 * m + 3]};
                const uintx4 ad = {pa_d[4 * m], pa_d[4 * m + 1], pa_d[4 * m + 2], pa_d[4 * m + 3]};
                if constexpr (!WN_DETECT_KEEP0) twiddles(ic<m>{});
                else if constexpr (m + 1 < P::NM) twiddles(ic<m + 1>{});   // arrive while this group of matrix instructions runs
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1" ::: "memory");      // operands just written, read by the next MFMA
                __builtin_amdgcn_sched_barrier(0);
                static_for<C2T>([&](auto C) {
                    constexpr int c = decltype(C)::value;
                    if constexpr (m == 0) {
                        mfma_bf16_new(pr[c], as, bc[m][c]);
                        mfma_bf16_new(pi[c], ad, bs[m][c]);
                    } else {
                        mfma_bf16_acc(pr[c], as, bc[m][c]);
                        mfma_bf16_acc(pi[c], ad, bs[m][c]);
                    }
                });
            });
        }
        mfma_drain();
        lap(ic<10>{});
        // rough energy of each column tile (rows a' < 16 of it, mirrored pairs summed): where the
        // second stage starts, and whether one column tile holds nearly everything
        float ec[C2T];
        unsigned eo[C2T];
#pragma unroll
        for (int c = 0; c < C2T; ++c) {
            after_drain(pr[c]);
            after_drain(pi[c]);
            const floatx4 yr = pr[c], yi = pi[c];
            ec[c] = (yr.x * yr.x + yi.x * yi.x) + (yr.y * yr.y + yi.y * yi.y) + (yr.z * yr.z + yi.z * yi.z) +
                    (yr.w * yr.w + yi.w * yi.w);
        }
        // per column (the four lane groups summed by two lane swaps: every lane then holds the energy of column
        // 16c + li), then per tile (the sixteen columns of a row of lanes)
        float ce[C2T];
#pragma unroll
        for (int c = 0; c < C2T; ++c) {
            const unsigned a = __float_as_uint(ec[c]);
            const auto h = __builtin_amdgcn_permlane16_swap(a, a, false, false);      // rows (0, 0, 2, 2) and (1, 1, 3, 3)
            const float s2 = __uint_as_float(h[0]) + __uint_as_float(h[1]);
            const auto f = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
            ce[c] = __uint_as_float(f[0]) + __uint_as_float(f[1]);
            float v = ce[c];
            v += dpp_f0<0xB1, 0xF>(v);
            v += dpp_f0<0x4E, 0xF>(v);
            v += dpp_f0<0x141, 0xF>(v);
            v += dpp_f0<0x140, 0xF>(v);
            eo[c] = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v));   // non-negative floats order like their bits
        }
        int c_first = 0;
        bool one_tile;
        {
            unsigned top = eo[0];
            float tot = __uint_as_float(eo[0]);
#pragma unroll
            for (int c = 1; c < C2T; ++c) {
                tot += __uint_as_float(eo[c]);
                if (eo[c] > top) {
                    top = eo[c];
                    c_first = c;
                }
            }
            one_tile = __uint_as_float(top) >= one_share * tot && tot <= 1e30f;   // false for anything that is not a number
        }
        lap(ic<1>{});

        auto refetch = [&]() {
            // The window's buffer is by now being overwritten by the next window: wait for that
            // DMA, fetch this window again (the long way then fetches the next window again).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            (void)fetch(cur_s, cur_w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            extend();
            ++redone;
        };
        // One bin alone.  Two windows in five of a transmission hold little else than the separator tone,
        // one in four little else than one data tone.  When the pilot finds half of its energy in the
        // separator's column, or more than `bin_share` of it in one column that holds tones of the
        // scheme, one bin is computed by itself: the separator's, or the scheme tone of that column whose
        // row frequency fits the pilot's rows best (a tone makes the pilot's sixteen row sums S[a'] of a
        // column swing like cos(2*pi*k1*a'/n1) against S[0]; every candidate's pattern is laid against
        // them).  Per chunk of 256 samples every lane multiplies its four samples by fixed phase factors,
        // the chunk's sum is turned by the chunk's own factor, and the window's energy is summed beside
        // it (no matrix instructions, a tenth of the one-tile way).  If everything outside that bin
        // holds less than the limit below which nothing can compete, the window is decided: same bound,
        // same limit as after any part of the spectrum.  Otherwise the window starts again without
        // this attempt.
        int path = 7;   // PROF build only
        bool sep_done = false;
        {
            float tot = 0.f;
#pragma unroll
            for (int c = 0; c < C2T; ++c) tot += __uint_as_float(eo[c]);
            if (!sep_tried && tot <= 1e30f && (sep_share > 0.f || bin_share > 0.f)) {
                int cand = 0;   // the bin to try, 0: none
                bool is_sep = false;
                floatx4 p01 = {0.f, 0.f, 0.f, 0.f}, p23 = p01;       // this lane's phase factors for floats 4*lane + q of a chunk
                const floatx4 *utab = (const floatx4 *)(smem + P::OFF_SEP + 256 * 8);   // (cos, sin, -sin, cos) of the chunks' factors
                const int scol = min(sep_bin % N2, N2 - sep_bin % N2);   // the separator bin's spectrum column
                const float in_tile = (scol >> 4) == 0 ? ce[0] : ((scol >> 4) == 1 ? ce[1] : ce[C2T - 1]);
                if (sep_share > 0.f && lane_f(in_tile, scol & 15) >= sep_share * tot) {
                    cand = sep_bin;
                    is_sep = true;
                    p01 = ((const floatx4 *)(smem + P::OFF_SEP))[2 * lane];
                    p23 = ((const floatx4 *)(smem + P::OFF_SEP))[2 * lane + 1];
                } else if (bin_share > 0.f) {
                    // the strongest column (lane 16c + li speaks for column 16c + li; computed from a copy of the
                    // lane number the compiler cannot see through, see the choice of sixteen columns below)
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    const int lgq = ln >> 4;
                    const float mine = lgq == 0 ? ce[0] : (lgq == 1 ? ce[1] : ce[C2T - 1]);
                    const unsigned top = lane63(wave_max63(ln <= H2 ? ((__float_as_uint(mine) & ~63u) | (unsigned)ln) : 0u));
                    const int cs = (int)(top & 63u);
                    if (__uint_as_float(top & ~63u) >= bin_share * tot) {
                        // S[a'] of that column: register r of lane (group g, column lane ls) of its tile holds row a' = 4r + g
                        const int ct = cs >> 4, ls = cs & 15;
                        const floatx4 yr = ct == 0 ? pr[0] : (ct == 1 ? pr[1] : pr[C2T - 1]), yi = ct == 0 ? pi[0] : (ct == 1 ? pi[1] : pi[C2T - 1]);
                        const float r0 = lane_f(yr.x, ls), i0 = lane_f(yi.x, ls);
                        const floatx4 g = yr * r0 + yi * i0;   // Re(S[a'] conj(S[0])), in the four lanes that hold the column
                        // lane 4j + p lays candidate j against rows a' = p, p + 4, p + 8, p + 12 (all in lane group p)
                        const int jc = (ln >> 2) & 7, pp = ln & 3;
                        const int src = 4 * (ls + 16 * pp);
                        const float g0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g.x)));
                        const float g1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g.y)));
                        const float g2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g.z)));
                        const float g3 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g.w)));
                        const int entry = ((const int *)(smem + P::OFF_CAND))[8 * cs + jc];   // bin | folded row frequency << 16
                        const int k1f = entry >> 16;
                        const float *cw = (const float *)(smem + P::OFF_TW1C);
                        float sc = g0 * cw[(k1f * pp) % N1] + g1 * cw[(k1f * (pp + 4)) % N1] + g2 * cw[(k1f * (pp + 8)) % N1] +
                                   g3 * cw[(k1f * (pp + 12)) % N1];
                        sc += dpp_f0<0xB1, 0xF>(sc);
                        sc += dpp_f0<0x4E, 0xF>(sc);
                        unsigned u = __float_as_uint(sc);
                        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // orders like the value
                        const unsigned best_c = lane63(wave_max63((entry & 0xFFFF) && ln < 32 ? ((u & ~7u) | (unsigned)jc) : 0u));
                        if (best_c) cand = ((const int *)(smem + P::OFF_CAND))[8 * cs + (int)(best_c & 7u)] & 0xFFFF;
                    }
                    if (cand) {
                        // phase factors of that bin: exp(2*pi*i*m/N), m = cand * position mod N, as the product of two
                        // table entries (m = 128a + b); this wave's chunk factors go to its own table
                        auto turn = [&](unsigned m) {
                            const float2 h = ((const float2 *)(smem + P::OFF_TH))[m >> 7], l = ((const float2 *)(smem + P::OFF_TL))[m & 127u];
                            return make_float2(h.x * l.x - h.y * l.y, h.x * l.y + h.y * l.x);
                        };
                        const unsigned kc = (unsigned)cand;
                        const float2 t0 = turn((kc * (4u * ln + 0u)) % (unsigned)N), t1 = turn((kc * (4u * ln + 1u)) % (unsigned)N);
                        const float2 t2 = turn((kc * (4u * ln + 2u)) % (unsigned)N), t3 = turn((kc * (4u * ln + 3u)) % (unsigned)N);
                        p01 = floatx4{t0.x, t0.y, t1.x, t1.y};
                        p23 = floatx4{t2.x, t2.y, t3.x, t3.y};
                        floatx4 *mine_u = (floatx4 *)(smem + P::OFF_WU) + P::NCH * wave;
                        if (ln < P::NCH) {
                            const float2 uj = turn((kc * 256u * (unsigned)ln) % (unsigned)N);
                            mine_u[ln] = floatx4{uj.x, uj.y, -uj.y, uj.x};
                        }
                        utab = mine_u;
                    }
                }
                if (cand) {
                const floatx4 *buf = (const floatx4 *)(smf + xw_f) + lane;   // float 256*j + 4*lane + q of the buffer is sample 256*j + 4*lane + q - off
                const int off = next_off;
                if constexpr (WN_BIN_PIPE) {
                    // The next window's transfer follows the reads chunk by chunk (a 16-byte piece of it covers
                    // exactly one chunk; the pieces that land in the last chunk go out after the loop): the wave
                    // has one window buffer and nothing else to do while the next window is under way.
                    static_assert(DMA_FULL >= P::NCH - 1 && P::SM >= 256 * (P::NCH - 1),
                                  "every piece from the last chunk's on lands in or behind the last chunk");
                    flush();
                    pend = false;
                    const bool nxt = wn < total;
                    next_off = fetch_prepare(nxt ? f_s : cur_s, nxt ? f_w : cur_w, ic<1>{});
                }
                floatx2 acc = {0.f, 0.f}, e2 = {0.f, 0.f}, sx2 = {0.f, 0.f};
                floatx4 xr[4], ur[4];
                auto ld = [&](auto J) {
                    constexpr int j = decltype(J)::value;
                    if constexpr (j < P::NCH) {
                        xr[j & 3] = buf[64 * j];
                        ur[j & 3] = utab[j];
                    }
                };
                ld(ic<0>{});
                ld(ic<1>{});
                ld(ic<2>{});
                static_for<P::NCH>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    ld(ic<j + 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                    floatx4 x = xr[j & 3];
                    const floatx4 u = ur[j & 3];
                    if constexpr (j == 0 || 256 * (j + 1) > N) {   // floats before the window's first sample, after its last
                        const int i0 = 256 * j + 4 * lane - off;
                        x.x = (unsigned)(i0 + 0) < (unsigned)N ? x.x : 0.f;
                        x.y = (unsigned)(i0 + 1) < (unsigned)N ? x.y : 0.f;
                        x.z = (unsigned)(i0 + 2) < (unsigned)N ? x.z : 0.f;
                        x.w = (unsigned)(i0 + 3) < (unsigned)N ? x.w : 0.f;
                    }
                    floatx2 sj = floatx2{x.x, x.x} * p01.lo;
                    sj += floatx2{x.y, x.y} * p01.hi;
                    sj += floatx2{x.z, x.z} * p23.lo;
                    sj += floatx2{x.w, x.w} * p23.hi;
                    acc += floatx2{sj.x, sj.x} * u.lo;
                    acc += floatx2{sj.y, sj.y} * u.hi;
                    e2 += x.lo * x.lo;
                    e2 += x.hi * x.hi;
                    sx2 += x.lo;
                    sx2 += x.hi;
                    // keeps the sums here (left to itself the compiler holds every chunk's samples in registers
                    // and adds them up after the loop)
                    asm volatile("" : "+v"(acc), "+v"(e2), "+v"(sx2));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (WN_BIN_PIPE && j < DMA_FULL && j < P::NCH - 1) {
                        fetch_piece(ic<j>{});   // chunk j has been read (its values are in the sums pinned above)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                if constexpr (WN_BIN_PIPE) {
                    static_for<DMA_PIECES - (P::NCH - 1)>([&](auto K) { fetch_piece(ic<P::NCH - 1 + decltype(K)::value>{}); });
                    fetch_late(next_off);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float zr = lane63(wave_sum63(acc.x)), zi = lane63(wave_sum63(acc.y));
                const float s2 = lane63(wave_sum63(e2.x + e2.y)), s1 = lane63(wave_sum63(sx2.x + sx2.y));
                const float m = zr * zr + zi * zi;
                const float eh = 0.5f * fmaf((float)N, s2, s1 * s1) * (1.f + 1e-5f), sq = __builtin_amdgcn_sqrtf(eh);
                const float lim_s = (m - (err_lin * (__builtin_amdgcn_sqrtf(m) * sq) + err_sq * eh)) * keep;
                if (m >= 1e-30f && eh <= 1e30f && eh * 1.005f - m * (1.f - 1e-5f) < lim_s) {
                    if constexpr (!WN_BIN_PIPE) next_window_dma();
                    if (wn < total) advance();
                    pend_bin = cand;
                    pend_cls = lut_s[cand];
                    pend_mag = __builtin_amdgcn_sqrtf(m);
                    pend_w = w;
                    pend = true;
                    if (is_sep) ++sep_n;
                    else ++tone_n;
                    sep_done = true;
                    lap(ic<11>{});
                } else {
                    // not decided after all (the pilot sees a part of the rows): the window starts again,
                    // this time without this attempt
                    if constexpr (WN_BIN_PIPE) {
                        refetch();
                        next_off = off;
                    } else {
                        ++redone;
                    }
                    sep_tried = true;
                    continue;
                }
                }
            }
        }
        do {
        if (sep_done) break;
        Top3 best;
        float eproc;   // energy of the outputs this lane has looked at (each bin once)
        // What is still to do cannot matter once everything outside the outputs already seen
        // holds less energy than the smallest value that could still change the decision (`lim`,
        // below).  The float32 energy of the outputs seen can exceed the exact one by
        // 2B*sum|Z| + n*B^2 <= 2B*sqrt(n*E) < 0.005 E (B = 724u*sqrt(E), n < 3600 outputs).
        auto rest_cannot_matter = [&](unsigned &mx_out) {
            mx_out = lane63(wave_max63(best.mx));
            const float f1u = __uint_as_float(mx_out);
            const float seen = lane63(wave_sum63(eproc));
            const float lim_now = (f1u - (err_lin * (__builtin_amdgcn_sqrtf(f1u) * sqrt_e) + err_sq * e_half)) * keep;
            return e_half * 1.005f - seen * (1.f - 1e-5f) < lim_now;
        };
        constexpr int NCOL[3] = {16, C2T > 2 ? 16 : P::K2 - 16, P::K2 - 16 * (C2T - 1)};
        unsigned mxu = 0u;
        bool have_max = false;
        // Stage 2 of column tile C, one row tile of 16 outputs k1 at a time (44 matrix instructions
        // and the arg-max work of 8 outputs per lane), with the check after each: true as soon as
        // the rest cannot matter.  `more` says whether other column tiles would follow (after the
        // very last row tile of the window there is nothing left to rule out).  The last column
        // tile starts with the last row tile: that is where the separator, ping and end tones are.
        // With PICKED the tile is the one of sixteen chosen columns (results where column tile 0 has
        // its own): lane column `pk_col`, bins computed from the per-window terms pk_u / pk_v below.
        unsigned pk_u = 0u, pk_v = 0u;     // ((4*lg*E1 + k2*E2) mod N, (k2*E2 - 4*lg*E1) mod N) of this lane's column k2
        bool pk_col0 = false, pk_x = false;   // this lane's column is k2 = 0; this lane holds the bin that is never looked at
        auto col_tile = [&](auto C, bool more, auto DMA, auto PICKED) -> bool {
            constexpr int c = decltype(C)::value;
            constexpr bool dma = decltype(DMA)::value != 0;   // the first cell issues the next window's transfer
            constexpr bool picked = decltype(PICKED)::value != 0;
            bool done = false;
            static_for<TA>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int tt = (picked || c == C2T - 1) ? (i == 0 ? TA - 1 : i - 1) : i;
                if (!done) {
                    floatx4 acc[4];   // the sums A, B, C, E
                    static_for<KS2>([&](auto S) {
                        constexpr int s = decltype(S)::value, tau = s >> 2, r = s & 3;
                        auto piece = [&](auto Q) {
                            if constexpr (dma && i == 0) fetch_piece(ic<4 * s + decltype(Q)::value>{});
                        };
                        if constexpr (s == 0) {
                            mfma2_new(acc[0], t1c[tt][s], Yr[c][tau][r]);
                            piece(ic<0>{});
                            mfma2_new(acc[2], t1c[tt][s], Yi[c][tau][r]);
                            piece(ic<1>{});
                            mfma2a_new(acc[1], t1s[tt][s], Yi[c][tau + TA][r]);
                            piece(ic<2>{});
                            mfma2a_new(acc[3], t1s[tt][s], Yr[c][tau + TA][r]);
                            piece(ic<3>{});
                        } else {
                            mfma2_acc(acc[0], t1c[tt][s], Yr[c][tau][r]);
                            piece(ic<0>{});
                            mfma2_acc(acc[2], t1c[tt][s], Yi[c][tau][r]);
                            piece(ic<1>{});
                            mfma2a_acc(acc[1], t1s[tt][s], Yi[c][tau + TA][r]);
                            piece(ic<2>{});
                            mfma2a_acc(acc[3], t1s[tt][s], Yr[c][tau + TA][r]);
                            piece(ic<3>{});
                        }
                    });
                    mfma_drain();
#pragma unroll
                    for (int q = 0; q < 4; ++q) after_drain(acc[q]);
                    __builtin_amdgcn_sched_barrier(0);
                    // arg-max work: two outputs per accumulator element; the float part on register
                    // pairs (packed f32 instructions), the keys one by one
                    floatx2 es_p = {0.f, 0.f}, es_m = {0.f, 0.f};
                    float dup_row0 = 0.f;
                    static_for<2>([&](auto HALF) {
                        constexpr int hf = decltype(HALF)::value;
                        const floatx2 A = hf ? acc[0].hi : acc[0].lo, B = hf ? acc[1].hi : acc[1].lo;
                        const floatx2 Cc = hf ? acc[2].hi : acc[2].lo, E = hf ? acc[3].hi : acc[3].lo;
                        const floatx2 rp = A + B, rm = A - B, ip = Cc - E, im = Cc + E;
                        const floatx2 mp2 = rp * rp + ip * ip, mm2 = rm * rm + im * im;
                        es_p += mp2;
                        es_m += mm2;
                        if constexpr (tt == 0 && hf == 0) dup_row0 = mm2.x;
                        static_for<2>([&](auto EL) {
                            constexpr int r = 2 * hf + decltype(EL)::value;
                            float mp = mp2[decltype(EL)::value], mm = mm2[decltype(EL)::value];
                            unsigned tp, tm;   // only the low 12 bits (and bit 15) of each are used
                            if constexpr (picked) {
                                // bin of Z[j][k2] = (j*E1 + k2*E2) mod N folded to [0, N/2], j = k1 (plus side) or
                                // n1 - k1 (minus side: -k1*E1 mod N), k1 = 16*tt + 4*lg + r: the lane's part of the
                                // sum comes per window, the rest is a constant of the code.  x in [0, 2N] folds to
                                // its distance from the nearest of 0, N, 2N.
                                constexpr unsigned CT = (unsigned)(((long long)(16 * tt + r) * P::E1) % N);
                                const unsigned xp = pk_u + CT, xm = pk_v + ((unsigned)N - CT);
                                tp = min(min(xp, absdiff_u32(xp, (unsigned)N)), 2u * N - xp);
                                tm = min(min(xm, absdiff_u32(xm, (unsigned)N)), 2u * N - xm);
                                constexpr bool xrow = tt == P::XK1 / 16 && r == P::XK1 % 4;
                                if constexpr (xrow && !P::XMINUS) mp = pk_x ? 0.f : mp;
                                if constexpr (xrow && P::XMINUS) mm = pk_x ? 0.f : mm;
                                // the minus side repeats the plus side where k2 = 0 or k1 = 0
                                if constexpr (tt == 0 && r == 0) mm = (pk_col0 | (lg == 0)) ? 0.f : mm;
                                else mm = pk_col0 ? 0.f : mm;
                            } else {
                                // (the empty statement keeps the two halves from being taken apart once and for
                                // all ahead of the window loop, which would cost the registers the packing saves)
                                asm volatile("" : "+v"(tag[tt][r][c]));
                                tp = tag[tt][r][c];
                                tm = tp >> 16;
                                // outputs that must not compete although their value is not zero: known positions
                                constexpr bool xhere = tt == P::XK1 / 16 && r == P::XK1 % 4 && c == P::XK2 / 16;
                                if constexpr (xhere && !P::XMINUS) mp = __builtin_unpredictable(tp & EP_UNUSED) ? 0.f : mp;
                                if constexpr (c == 0 || (tt == 0 && r == 0) || (xhere && P::XMINUS))
                                    mm = __builtin_unpredictable(tm & EP_UNUSED) ? 0.f : mm;
                            }
                            best.mx = max(best.mx, max(__float_as_uint(mp), __float_as_uint(mm)));
                            top3_insert(best, (__float_as_uint(mp) & ~KEY_BIN) | (tp & KEY_BIN));
                            top3_insert(best, (__float_as_uint(mm) & ~KEY_BIN) | (tm & KEY_BIN));
                        });
                    });
                    // each bin once: where k2 = 0 (column tile 0, li = 0) every minus-side output repeats
                    // its plus side, and so does the one of row k1 = 0 (row tile 0, register 0, lane group 0)
                    float em = es_m.x + es_m.y;
                    if constexpr (picked) em = pk_col0 ? 0.f : em - (lg == 0 ? dup_row0 : 0.f);
                    else if constexpr (c == 0) em = li == 0 ? 0.f : em - (lg == 0 ? dup_row0 : 0.f);
                    else em -= lg == 0 ? dup_row0 : 0.f;
                    eproc += (es_p.x + es_p.y) + em;
                    cells_done += picked ? 16 : NCOL[c];
                    if (i + 1 < TA || more) {
                        if (rest_cannot_matter(mxu)) {
                            have_max = true;
                            done = true;
                        }
                    }
                }
            });
            return done;
        };
        constexpr int OTHER_TILES = 2 * T1;   // (row tile, real or imaginary part) pairs of stage 1

        // ------------------------------------------------------------------ the rest of stage 1, stage 2
        auto after_stage1 = [&](auto SPLIT) {
            lap(ic<6>{});
            if constexpr (decltype(SPLIT)::value) {
                // the transfer goes out piece by piece inside the first stage-2 cell; a wave without
                // a next window fetches this one again (the data is not used)
                flush();
                const bool nxt = wn < total;
                next_off = fetch_prepare(nxt ? f_s : cur_s, nxt ? f_w : cur_w, ic<0>{});
            } else {
                next_window_dma();
            }
            lap(ic<2>{});
            mfma_drain();
            energies();
            lap(ic<5>{});
            best.k1 = best.k2 = best.k3 = 0u;
            best.mx = 0u;
            eproc = 0.f;
        };
        constexpr unsigned ALL = (1u << C2T) - 1u;
        bool whole = !one_tile;
        // Column tiles whose stage 1 the long way computes.  With three column tiles, a window whose
        // pilot finds next to nothing in the weakest one (a tone meeting the separator, each in its
        // own tile) does without that tile; should the check still fail after the other two, the
        // window is done again with all three.
        unsigned need = ALL;
        if constexpr (C2T > 2) {
            {   // (also for a window that starts on one tile: should its check fail, the long way follows)
                int c_min = 0;
                unsigned low = eo[0];
                float tot = __uint_as_float(eo[0]);
#pragma unroll
                for (int c = 1; c < C2T; ++c) {
                    tot += __uint_as_float(eo[c]);
                    if (eo[c] < low) {
                        low = eo[c];
                        c_min = c;
                    }
                }
                if (__uint_as_float(low) <= two_share * tot && tot <= 1e30f) need = ALL & ~(1u << c_min);
            }
        }
        // Sixteen chosen columns.  A window that is not a one-tile window mostly holds two tones
        // (one ending, one beginning) whose columns lie in different tiles; nearly all of its energy
        // still sits in a dozen columns.  The pilot's energies per column (summed over the four lane
        // groups) put every column in one of four classes (above 4, 1, 1/4 times `pick_share` of the
        // total, or below); columns are taken class by class, in column order inside a class, until
        // sixteen are chosen.  If the first two classes fit in sixteen, stage 1 computes that one tile
        // with twiddles fetched for it, and stage 2 follows with the bins worked out per window.  The
        // energy check decides as everywhere else whether that was enough.
        bool picked = false;
        if constexpr (C2T > 2) {
            float tot = 0.f;
#pragma unroll
            for (int c = 0; c < C2T; ++c) tot += __uint_as_float(eo[c]);
            if (!one_tile && pick_share > 0.f && tot <= 1e30f) {
                // lane 16c + li speaks for column k2 = 16c + li.  (What follows is computed from a copy of the
                // lane number the compiler cannot see through: left to itself it keeps a dozen per-lane
                // constants of this rarely used block in registers for the whole launch.)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int lgq = ln >> 4, liq = ln & 15;
                const float mine = lgq == 0 ? ce[0] : (lgq == 1 ? ce[1] : ce[2]);
                const bool valid = ln <= H2;
                const float th = pick_share * tot;
                const bool p1 = valid & (mine > 4.f * th), p2 = valid & (mine > th), p3 = valid & (mine > 0.25f * th);
                const unsigned long long m1 = __ballot(p1), m2 = __ballot(p2), m3 = __ballot(p3), mv = __ballot(valid);
                const int n1c = __popcll(m1), n2c = __popcll(m2), n3c = __popcll(m3);
                if (n2c <= 16) {
                    auto below = [&](unsigned long long m) {
                        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    };
                    int slot = p1 ? below(m1)
                                  : (p2 ? n1c + below(m2 & ~m1) : (p3 ? n2c + below(m3 & ~m2) : n3c + below(mv & ~m3)));
                    slot = valid ? slot : ln;   // a permutation of the lanes: columns in order of choice, then the lanes that hold none
                    const int by_slot = __builtin_amdgcn_ds_permute(4 * slot, ln);   // lane s: the column of slot s
                    const int k2 = __builtin_amdgcn_ds_bpermute(4 * liq, by_slot);
                    const int kE = ((const int *)(smem + P::OFF_CB))[k2];
                    constexpr int R1 = (int)((4ll * P::E1) % N), R2 = (int)((8ll * P::E1) % N), R3 = (int)((12ll * P::E1) % N);
                    const int rl = lgq == 0 ? 0 : (lgq == 1 ? R1 : (lgq == 2 ? R2 : R3));
                    pk_u = (unsigned)(kE + rl >= N ? kE + rl - N : kE + rl);
                    pk_v = (unsigned)(kE - rl < 0 ? kE - rl + N : kE - rl);
                    pk_col0 = k2 == 0;
                    pk_x = k2 == P::XK2 && lgq == (P::XK1 % 16) / 4;
                    // twiddles (cos, -sin)(2*pi*b'*k2/n2) of b' = ks + KS1*lg: the table index climbs by k2 per k-step
                    static_assert(KS1 - 2 + KS1 * 3 <= H2, "only the last k-step holds padding");
                    pk_i8 = 8u * (unsigned)((KS1 * lgq * k2) % N2);
                    pk_k8 = 8u * (unsigned)k2;
                    picked = true;
                }
            }
        }
        path = one_tile ? 0 : (picked ? 5 : (need != ALL ? 1 : 2));
        if (one_tile) {
            // the pilot found the energy in one column tile: stage 1 and stage 2 of that tile only
            bool done = false;
            if (c_first == 0) {
                stage1(ic<ONE>{}, ic<0>{}, 0u);
                after_stage1(ic<1>{});
                done = col_tile(ic<0>{}, true, ic<1>{}, ic<0>{});
            } else if (c_first == 1) {
                stage1(ic<ONE>{}, ic<1>{}, 0u);
                after_stage1(ic<1>{});
                done = col_tile(ic<1>{}, true, ic<1>{}, ic<0>{});
            } else if constexpr (C2T > 2) {
                stage1(ic<ONE>{}, ic<2>{}, 0u);
                after_stage1(ic<1>{});
                done = col_tile(ic<2>{}, true, ic<1>{}, ic<0>{});
            }
            const int ncol = c_first == 0 ? NCOL[0] : (c_first == 1 ? NCOL[1] : NCOL[2]);
            s1_units += (unsigned)(OTHER_TILES * ncol);
            if (!done) {
                // Other column tiles are needed after all (energy in one column tile, but spread
                // over many bins: rare): the whole way.
                refetch();
                whole = true;
                path = 3;
            }
        }
        if (picked) {
            stage1(ic<PICK>{}, ic<0>{}, 0u);
            after_stage1(ic<1>{});
            const bool done = col_tile(ic<0>{}, true, ic<1>{}, ic<1>{});
            s1_units += (unsigned)(OTHER_TILES * 16);
            ++pick_n;
            if (done) {
                whole = false;
            } else {
                refetch();   // not enough: the long way
                path = 6;
            }
        }
        while (whole) {
            // the column tiles in `need` get their stage 1; stage 2 takes the one with the largest
            // rough energy first (any order is valid; this one ends the loop soonest) and chooses
            // again among the rest each time it has to go on
            stage1(ic<WHOLE>{}, ic<0>{}, need);
#pragma unroll
            for (int c = 0; c < C2T; ++c)
                if (need & (1u << c)) s1_units += (unsigned)(OTHER_TILES * NCOL[c]);
            ++whole_n;
            if (need != ALL) ++two_n;
            after_stage1(ic<0>{});
            have_max = false;
            unsigned todo = need;
            bool done = false;
#pragma nounroll
            for (int i = 0; todo != 0u; ++i) {
                int c;
                if (i == 0) {
                    c = c_first;
                } else {
                    unsigned key = 0u;   // energy bits, tile number in the low bits
#pragma unroll
                    for (int q = 0; q < C2T; ++q)
                        key = max(key, ((todo >> q) & 1u) ? ((eo[q] & ~7u) | 4u | (unsigned)q) : 0u);
                    c = (int)(key & 3u);
                }
                todo &= ~(1u << c);
                // after the very last row tile of a window there is nothing left to rule out; tiles
                // that were left out still have to be
                const bool more = todo != 0u || need != ALL;
                if (c == 0) done = col_tile(ic<0>{}, more, ic<0>{}, ic<0>{});
                else if (c == 1) done = col_tile(ic<1>{}, more, ic<0>{}, ic<0>{});
                else if constexpr (C2T > 2) done = col_tile(ic<2>{}, more, ic<0>{}, ic<0>{});
                if (done) break;
            }
            if (done || need == ALL) break;
            refetch();      // the tile that was left out is needed after all
            need = ALL;
            path = 4;
        }
        lap(ic<3>{});
        if (wn < total) advance();
        if (!have_max) mxu = lane63(wave_max63(best.mx));
        // From here on `mxu`, and everything computed from it, is the same in every lane.
        // A largest |Z|^2 below 1e-30 or an energy above 1e30 (zero, or samples so small or so
        // large that float32 squares lose their meaning, or not a number) is not decided here: the
        // window goes to the float64 pass, which also recognises all-zero windows.
        const float mxf = __uint_as_float(mxu);
        const bool odd = !(mxf >= 1e-30f && e_half <= 1e30f);
        // Keys carry |Z|^2 truncated to 2^-11 relative.  The float32 value of every |Z|^2 is
        // within err_lin*sqrt(|Z|^2 * E) + err_sq*E of the exact one (E = energy of the half
        // spectrum; derivation at launch_detect), so a bin whose key value, rounded up, reaches
        // `lim` may be the true maximum.  Outputs of column tiles that were left out are all below it.
        const float lim = (mxf - (err_lin * (__builtin_amdgcn_sqrtf(mxf) * sqrt_e) + err_sq * e_half)) * keep;
        // The magnitude that is reported must hold to 1e-4 relative as well.  Its float32 error is a
        // few 1e-9 * sqrt(E) in practice (measured; the bound above is the worst case), so a searched
        // maximum holding less than 1e-6 of the energy (nearly everything sits in the one bin that
        // is not searched) is recomputed in float64 too.
        const bool faint = e_half > 1e6f * mxf;
        const unsigned long long in1 = __ballot(__uint_as_float(best.k1 & ~KEY_BIN) >= lim);
        const unsigned long long in2 = __ballot(__uint_as_float(best.k2 & ~KEY_BIN) >= lim);
        const unsigned long long in3 = __ballot(__uint_as_float(best.k3 & ~KEY_BIN) >= lim);
        pend_mag = odd ? 0.f : __builtin_amdgcn_sqrtf(mxf);
        pend_w = w;
        pend = true;
        if (!odd && !faint && in2 == 0ull && __popcll(in1) == 1) {
            // the usual case: one output of one lane stands alone above the limit
            const int b = (int)((unsigned)__builtin_amdgcn_readlane((int)best.k1, __ffsll((long long)in1) - 1) & KEY_BIN);
            pend_bin = b;
            pend_cls = lut_s[b];
        } else {
            // wave-wide merge of the per-lane top three, all by DPP: lane pairs, quads, eights and
            // sixteens merge two disjoint lane sets each and leave every lane of a row with the
            // row's result; two row broadcasts then collect the four rows in the last row (the
            // other rows merge with zeros and are not used).  Lane 63 decides and stores.
            auto merge_with = [&](unsigned o1, unsigned o2, unsigned o3) {
                top3_insert(best, o1);
                top3_insert(best, o2);
                top3_insert(best, o3);
            };
#define WN_MERGE_DPP(CTRL, ROWS) \
    merge_with(dpp_u<CTRL, ROWS>(best.k1), dpp_u<CTRL, ROWS>(best.k2), dpp_u<CTRL, ROWS>(best.k3))
            WN_MERGE_DPP(0xB1, 0xF);    // quad_perm [1,0,3,2]
            WN_MERGE_DPP(0x4E, 0xF);    // quad_perm [2,3,0,1]
            WN_MERGE_DPP(0x141, 0xF);   // row_half_mirror
            WN_MERGE_DPP(0x140, 0xF);   // row_mirror
            WN_MERGE_DPP(0x142, 0xA);   // row_bcast:15 into rows 1 and 3
            WN_MERGE_DPP(0x143, 0xC);   // row_bcast:31 into rows 2 and 3
#undef WN_MERGE_DPP
            // Outputs that reach the limit, counted over the lanes before the merge: a lane's three
            // slots are three different outputs.  Up to three, none of them a lane's third (a lane
            // whose third reaches the limit may hold a fourth that does): exactly those are the
            // candidates, and the merged top three name them.  More: the whole spectrum.
            const int cand = __popcll(in1) + __popcll(in2) + __popcll(in3);
            const bool full = odd | faint | (in3 != 0ull) | (cand > 3);
            const int b = (int)(best.k1 & KEY_BIN);
            const unsigned cls = lut_s[odd ? 0 : b];
            pend_bin = full ? ((odd ? 0 : b) | FLAG_FULL) : (cand == 3 ? (b | FLAG_TOP3) : (cand == 2 ? (b | FLAG_TOP2) : b));
            pend_cls = full ? 0u : (cand >= 2 ? (best.k2 & KEY_BIN) : cls);
            if (!full && cand == 3) pend_mag = __int_as_float((int)(best.k3 & KEY_BIN));
        }
        } while (0);
        sep_tried = false;
        lap(ic<4>{});
        if constexpr (PROF) {
            prof[12 + path] = prof[12 + path] + 1ull;
            prof[20 + path] = prof[20 + path] + (prof_t - path_t0);
        }
        w = wn;
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no transfer into this workgroup's LDS is left in flight
    if (lane == 0 && (cells_done | sep_n | tone_n)) {
        atomicAdd(&stats[3], cells_done);
        atomicAdd(&stats[4], s1_units);
        if (redone) atomicAdd(&stats[5], redone);
        if (whole_n) atomicAdd(&stats[6], whole_n);
        if (two_n) atomicAdd(&stats[7], two_n);
        if (pick_n) atomicAdd(&stats[8], pick_n);
        if (sep_n) atomicAdd(&stats[9], sep_n);
        if (tone_n) atomicAdd(&stats[10], tone_n);
    }
    if constexpr (PROF) {
        // out_bin[0..6]: clocks/256 per phase (DMA wait, pilot of stage 1 and the column tiles' rough
        // energies, stores + next DMA issue, stage 2 with its arg-max work and checks, decision,
        // energy of the window, rest of stage 1); [7]: total clocks/256; [8]: 100 MHz ticks/256
        // out_mag[16 + 4*workgroup + wave]: 100 MHz ticks this wave ran (how far apart the waves finish)
        if (lane == 0) out_mag[16 + 4 * blockIdx.x + wave] = (float)(__builtin_amdgcn_s_memrealtime() - prof_r0);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int i = 0; i < 7; ++i) out_bin[i] = (int)(prof[i] >> 8);
            for (int i = 8; i < 12; ++i) out_bin[i + 1] = (int)(prof[i] >> 8);   // parts of the pilot: reads, folds, matrix instructions
            out_bin[7] = (int)((__builtin_amdgcn_s_memtime() - prof_c0) >> 8);
            out_bin[8] = (int)((__builtin_amdgcn_s_memrealtime() - prof_r0) >> 8);
            // [13..20]: windows by the way they went, [21..28]: their clocks/256
            for (int i = 0; i < 8; ++i) {
                out_bin[13 + i] = (int)prof[12 + i];
                out_bin[21 + i] = (int)(prof[20 + i] >> 8);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// float64 re-evaluation of flagged windows, in two passes over the flags.
// Pass 0: windows with two or three candidate bins, X[b] = sum_n x[n] * exp(2*pi*i*((b*n) mod N)/N)
// by direct summation, the phase factor of m = 128*k + j taken as the product of two table
// entries (128 + N/128 of them, built by sincospi: the LDS of a workgroup stays small and eight
// workgroups share a CU); and windows flagged for the whole spectrum that turn out to be all
// zeros (digital silence: np.argmax of zeros is 0).  Pass 1: the whole spectrum of the windows
// still flagged, by the two-stage prime-factor DFT in float64 with Y in LDS.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

template <int PASS>
__global__ __launch_bounds__(256) void refine_kernel(const float *__restrict__ audio, long long n_streams,
                                                     long long stream_stride, long long first_sample,
                                                     long long n_windows, int N, int n1, int n2, int e1, int e2,
                                                     const uint16_t *__restrict__ lut,
                                                     uint16_t *__restrict__ cls, float *__restrict__ mag,
                                                     int32_t *__restrict__ bin, unsigned long long *stats, int chunk) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // pass 0: exp(2*pi*i*j/N), j < 128, then exp(2*pi*i*128*k/N), k < ceil(N/128)
    double2 *tab_lo = (double2 *)smem, *tab_hi = tab_lo + 128;
    // pass 1: Y, n1 * (n2/2+1) entries, then exp(-2*pi*i*m/n1), m < n1, and exp(-2*pi*i*m/n2), m < n2
    double2 *Y = (double2 *)smem;
    double2 *c1 = Y + n1 * (n2 / 2 + 1), *c2 = c1 + n1;
    __shared__ double red[4][6];
    __shared__ double best_m[4];
    __shared__ int best_b[4];
    __shared__ unsigned long long flagged[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long total = n_streams * n_windows;
    const int nbins = N / 2;
    if constexpr (PASS == 0) {
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&stats[0], (unsigned long long)total);
        double sn, cs;
        if (tid < 128) {
            sincospi(2.0 * (double)tid / (double)N, &sn, &cs);
            tab_lo[tid] = make_double2(cs, sn);
        } else if (128 * (tid - 128) < N) {
            sincospi(2.0 * (double)(128 * (tid - 128)) / (double)N, &sn, &cs);
            tab_hi[tid - 128] = make_double2(cs, sn);
        }
    } else {
        for (int m = tid; m < n1 + n2; m += 256) {
            double sn, cs;
            sincospi(m < n1 ? 2.0 * (double)m / (double)n1 : 2.0 * (double)(m - n1) / (double)n2, &sn, &cs);
            c1[m] = make_double2(cs, -sn);
        }
    }
    __syncthreads();
    auto phase = [&](unsigned m) {   // exp(2*pi*i*m/N), m < N
        const double2 h = tab_hi[m >> 7], l = tab_lo[m & 127];
        return make_double2(h.x * l.x - h.y * l.y, h.x * l.y + h.y * l.x);
    };
    // a workgroup takes `chunk` windows (64 or 256) at a time: small launches use the small chunk so that
    // their flagged windows, which a workgroup re-evaluates one after the other, spread over more CUs
    for (long long base = (long long)blockIdx.x * chunk; base < total; base += (long long)gridDim.x * chunk) {
        const long long idx = base + tid;
        int fl = (tid < chunk && idx < total) ? (bin[idx] & FLAG_MASK) : 0;
        if constexpr (PASS == 1) fl &= FLAG_FULL;
        if (!__syncthreads_or(fl)) continue;
        // flagged windows of this chunk, one bit per thread
        const unsigned long long bal = __ballot(fl != 0);
        __syncthreads();
        if (lane == 0) flagged[wave] = bal;
        __syncthreads();
        for (int fw = 0; fw < chunk / 64; ++fw)
        for (unsigned long long mask = flagged[fw]; mask; mask &= mask - 1) {
            const long long wi = base + 64 * fw + __ffsll((long long)mask) - 1;
            const int f = bin[wi];  // same address in every thread
            const long long s = wi / n_windows;
            const float *x = audio + s * stream_stride + first_sample + (wi - s * n_windows) * (long long)N;
            int rb;
            double rm;
            if (f & (FLAG_TOP2 | FLAG_TOP3)) {
                if constexpr (PASS == 1) continue;
                const bool three = (f & FLAG_TOP3) != 0;
                const int b1 = f & 0xFFFF, b2 = cls[wi], b3 = three ? __float_as_int(mag[wi]) : b1;
                double r1 = 0, i1 = 0, r2 = 0, i2 = 0, r3 = 0, i3 = 0;
                // phases (b*n) mod N of this thread's samples n = tid, tid + 256, ...: advanced by
                // (256*b) mod N per step instead of reduced anew
                const unsigned uN = (unsigned)N;
                unsigned m1 = (unsigned)(b1 * tid) % uN, m2 = (unsigned)(b2 * tid) % uN, m3 = (unsigned)(b3 * tid) % uN;
                const unsigned d1 = (unsigned)(b1 * 256) % uN, d2 = (unsigned)(b2 * 256) % uN, d3 = (unsigned)(b3 * 256) % uN;
#pragma unroll 2
                for (int n = tid; n < N; n += 256) {
                    const double xv = (double)x[n];
                    const double2 t1 = phase(m1);
                    const double2 t2 = phase(m2);
                    r1 += xv * t1.x; i1 += xv * t1.y;
                    r2 += xv * t2.x; i2 += xv * t2.y;
                    if (three) {
                        const double2 t3 = phase(m3);
                        r3 += xv * t3.x; i3 += xv * t3.y;
                        m3 += d3; m3 = m3 >= uN ? m3 - uN : m3;
                    }
                    m1 += d1; m1 = m1 >= uN ? m1 - uN : m1;
                    m2 += d2; m2 = m2 >= uN ? m2 - uN : m2;
                }
                r1 = wave_sum(r1); i1 = wave_sum(i1); r2 = wave_sum(r2); i2 = wave_sum(i2);
                r3 = wave_sum(r3); i3 = wave_sum(i3);
                __syncthreads();
                if (lane == 0) {
                    red[wave][0] = r1; red[wave][1] = i1; red[wave][2] = r2; red[wave][3] = i2;
                    red[wave][4] = r3; red[wave][5] = i3;
                }
                __syncthreads();
                double v[6];
                for (int q = 0; q < 6; ++q) v[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
                const double ma = v[0] * v[0] + v[1] * v[1], mb = v[2] * v[2] + v[3] * v[3];
                const bool first = (ma > mb) || (ma == mb && b1 < b2);
                rb = first ? b1 : b2;
                rm = first ? ma : mb;
                if (three) {
                    const double mc = v[4] * v[4] + v[5] * v[5];
                    if (mc > rm || (mc == rm && b3 < rb)) { rb = b3; rm = mc; }
                }
                if (tid == 0) atomicAdd(&stats[1], 1ull);
            } else {
                // all-zero windows (digital silence) need no spectrum: np.argmax of zeros is 0
                bool nz = false;
                for (int n = tid; n < N; n += 256) nz |= (x[n] != 0.0f);
                const bool any = __syncthreads_or(nz);
                if (PASS == 0 && any) continue;   // pass 1 does this window
                double bm = -1.0;
                int bb = any ? 0x7fffffff : 0;
                if (any) {
                    // The whole spectrum in float64 by the same two-stage prime-factor DFT as the
                    // float32 kernel, unfolded: Y[a][k2] = sum_b x[(n2*a + n1*b) mod N] * w2^(b*k2),
                    // Z[k1][k2] = sum_a w1^(a*k1) * Y[a][k2] = X[(k1*e1 + k2*e2) mod N]; columns
                    // k2 <= n2/2 hold every bin or its mirror image.
                    const int K2 = n2 / 2 + 1;
                    __syncthreads();   // nobody reads the previous window's Y any more
                    for (int o = tid; o < n1 * K2; o += 256) {
                        const int a = o / K2, k2 = o - a * K2;
                        double re = 0, im = 0;
                        int pos = (n2 * a) % N, m = 0;
                        for (int bq = 0; bq < n2; ++bq) {
                            const double xv = (double)x[pos];
                            const double2 t = c2[m];
                            re += xv * t.x; im += xv * t.y;
                            pos += n1; pos = pos >= N ? pos - N : pos;
                            m += k2; m = m >= n2 ? m - n2 : m;
                        }
                        Y[o] = make_double2(re, im);
                    }
                    __syncthreads();
                    for (int o = tid; o < n1 * K2; o += 256) {
                        const int k1 = o / K2, k2 = o - k1 * K2;
                        double re = 0, im = 0;
                        int m = 0;
                        for (int a = 0; a < n1; ++a) {
                            const double2 t = c1[m], y = Y[a * K2 + k2];
                            re += t.x * y.x - t.y * y.y;
                            im += t.x * y.y + t.y * y.x;
                            m += k1; m = m >= n1 ? m - n1 : m;
                        }
                        const int kb = (int)(((long long)k1 * e1 + (long long)k2 * e2) % N);
                        const int kk = kb < N - kb ? kb : N - kb;
                        const double mg = re * re + im * im;
                        if (kk < nbins && (mg > bm || (mg == bm && kk < bb))) { bm = mg; bb = kk; }   // first maximum wins
                    }
                }
                if (!any) bm = tid == 0 ? 0.0 : -1.0;
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
                rb = bb < nbins ? bb : 0;   // no bin compared greater (samples that are not numbers)
                rm = bm > 0.0 ? bm : 0.0;
                if (tid == 0 && any) atomicAdd(&stats[2], 1ull);
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
                         float *d_mag, int32_t *d_bin, unsigned long long *stats, hipStream_t st) {
    typedef Plan<N1, N2> P;
    // ctx->detect_profile (WN_DETECT_PROFILE=1 when the context was made; diagnostics, results
    // are not usable): per-phase clock counts of wave 0 in out_bin[0..8]
    const long long total = n_streams * n_windows;
    // one workgroup per CU for the long window (its four window buffers fill the LDS); short
    // windows leave room for two, and a second wave per SIMD runs its vector-ALU work under
    // the first one's matrix instructions
    constexpr int PER_CU = (2 * P::LDS <= 160 * 1024 && 2 * (P::TA * 16 + 160) <= 512) ? 2 : 1;
    long long blocks = (total + 3) / 4;
    // WN_DETECT_PER_CU=1 (read when the context is made) holds the short-window kernel to one workgroup
    // per CU: the measure of what a second wave per SIMD is worth to this code
    const int per_cu = (ctx->detect_per_cu == 1) ? 1 : PER_CU;
    if (blocks > (long long)ctx->num_cus * per_cu) blocks = (long long)ctx->num_cus * per_cu;
    // Which windows the float32 spectrum may decide.  u = 2^-24.  A folded sample (sum or
    // difference of four samples, two roundings) is within 2u of exact relative to the sum q of
    // the four magnitudes; a stage-1 output adds a rounded twiddle (u) and at most 4*KS1 <= 44
    // accumulation steps (taken as 2u each: faithful rounding is enough): 91u * (q summed over the
    // row pair).  A stage-2 sum repeats that on the stage-1 outputs (89u) and carries their error
    // (91u): 180u * ||x||_1 for each of the four real sums; real and imaginary part of an output
    // combine two sums each (one more rounding): 362u * ||x||_1, so |dZ| <= 512u * ||x||_1
    // <= 512u * sqrt(N * sum x^2) <= 512u * sqrt(2E), E = sum of |Z|^2 over bins [0, N/2]
    // (N odd: N * sum x^2 = 2E - Z_0^2).  With B = 724u * sqrt(E):
    //   | |Z^|^2 - |Z|^2 | <= 2|Z|B + B^2 (+ 3u|Z|^2 for squaring and adding),
    // and two bins are ordered for certain when their float32 values differ by more than the sum
    // of both terms: 4B*sqrt(m1) + 2B^2 + 6u*m1.  The kernel's E is summed in float32 from the
    // float32 outputs (self-mirrored outputs counted twice, which only raises it); it can fall
    // short of the exact E by under 1 %, which the factors below absorb.
    const double u = ldexp(1.0, -24), cB = 724.0 * u * 1.01;
    const float err_lin = (float)(4.0 * cB), err_sq = (float)(2.0 * cB * cB + 6.0 * u);
    // keys drop up to 2^-11 of a value; ctx->tie_band widens the band further (tests force the
    // float64 path with it): magnitudes within that relative distance count as tied
    const double k = (1.0 - ctx->tie_band) * (1.0 - ctx->tie_band) / (1.0 + 1.0 / 2048.0) * (1.0 - 1e-6);
    // A window goes the one-column-tile way when the pilot finds this share of its energy in one
    // tile.  Measured on clean and noisy frames: 0.5 -> 9 % of the windows done twice, 0.6 -> 3 %,
    // 0.7 -> 0.25 %, 0.8 and 0.9 -> none, but more windows the whole way; 0.7 is the fastest
    // (again after the later changes: 30.0 / 29.8 / 29.6 / 30.1 / 30.5 ms at 0.6 / 0.65 / 0.7 / 0.75 / 0.8).
    const float share = ctx->detect_share;   // 0.7 unless WN_DETECT_SHARE said otherwise when the context was made
    // the separator's bin (the nearest one), for schemes that have a separator inside the searched range
    const int sep_bin = ctx->sep_bin;
    const float sep_share = (sep_bin > 0 && sep_bin < ctx->n / 2) ? ctx->detect_sep_share : 0.f;
    const float bin_share = ctx->d_cand ? ctx->detect_bin_share : 0.f;
    // the profiling build leaves its clock counts in out_bin[0..28] and out_mag[16 .. 16 + 4*blocks):
    // launches whose result arrays are shorter than that run the ordinary build
    if (ctx->detect_profile && total >= 32 + 4 * blocks)
        hipLaunchKernelGGL((detect_kernel<N1, N2, true>), dim3((unsigned)blocks), dim3(256), P::LDS + 1024, st, d_audio,
                           (long long)n_streams, (long long)stream_stride, (long long)first_sample,
                           (long long)n_windows, ctx->d_tw1, ctx->d_tw2, ctx->d_cls_lut, (float)k, err_lin, err_sq, share, ctx->detect_two_share,
                           ctx->detect_pick_share, sep_bin, sep_share, ctx->d_cand, bin_share, ctx->detect_profile, d_cls, d_mag, d_bin, stats);
    else
        hipLaunchKernelGGL((detect_kernel<N1, N2, false>), dim3((unsigned)blocks), dim3(256), P::LDS, st, d_audio,
                           (long long)n_streams, (long long)stream_stride, (long long)first_sample,
                           (long long)n_windows, ctx->d_tw1, ctx->d_tw2, ctx->d_cls_lut, (float)k, err_lin, err_sq, share, ctx->detect_two_share,
                           ctx->detect_pick_share, sep_bin, sep_share, ctx->d_cand, bin_share, ctx->detect_profile, d_cls, d_mag, d_bin, stats);
    return wn_launch_check("detect_kernel");
}

// Workspace of wn_detect_windows_noalloc: what wn_detect_windows keeps in the caller's magnitude
// and bin arrays and in the context's counter block.  Every GPU holds its own copy of the array
// at its own address; host code never uses the array's name as an address, only the pointer
// hipGetSymbolAddress returned for the device in question.
#define NOALLOC_HEAD 256   // bytes: the four counters of the float64 pass, padded
static __device__ __align__(256) unsigned char scratch[NOALLOC_HEAD + 8ull * WN_DETECT_NOALLOC_MAX_WINDOWS];
static unsigned char *g_scratch_ptr[WN_MAX_DEVICES];   // written once per device, under wn_scheme_init's lock
static std::mutex g_noalloc_mutex[WN_MAX_DEVICES];
// recorded behind the last launch of every call that is not being captured into a graph: the next such
// call makes its stream wait for it, so that calls enqueued on different streams do not run at the
// same time on the one workspace
static hipEvent_t g_noalloc_event[WN_MAX_DEVICES];
static bool g_noalloc_recorded[WN_MAX_DEVICES];

template <int N1, int N2>
static int detect_attr() {
    WN_HIP_TRY(hipFuncSetAttribute((const void *)detect_kernel<N1, N2, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Plan<N1, N2>::LDS));
    WN_HIP_TRY(hipFuncSetAttribute((const void *)detect_kernel<N1, N2, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)Plan<N1, N2>::LDS + 1024));   // + its clock sums
    return WN_OK;
}

int wn_detect_device_setup(int device) {
    int rc = detect_attr<83, 85>();
    if (rc == WN_OK) rc = detect_attr<45, 49>();
    if (rc != WN_OK) return rc;
    // the whole-spectrum pass of the re-evaluation keeps its first-stage results in dynamic LDS:
    // n1 * (n2/2 + 1) + n1 + n2 entries of 16 bytes (wn_scheme_init refuses windows that need more)
    WN_HIP_TRY(hipFuncSetAttribute((const void *)refine_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   150 * 1024));
    // called with `device` current: the address of this device's copy of the workspace
    void *p = nullptr;
    WN_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p || ((uintptr_t)p % 256) != 0) {
        wn_set_error("workspace symbol of device %d resolved to %p", device, p);
        return WN_ERR_HIP;
    }
    g_scratch_ptr[device] = (unsigned char *)p;
    WN_HIP_TRY(hipEventCreateWithFlags(&g_noalloc_event[device], hipEventDisableTiming));
    g_noalloc_recorded[device] = false;
    return WN_OK;
}

// Argument checks and the two launches shared by both entry points: `d_mag`, `d_bin` hold the
// per-window magnitude and bin (and, between the two kernels, the flags of the windows the second
// one re-evaluates), `stats` the four counters of the float64 pass.
static int detect_run(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                      int64_t stream_stride, int64_t first_sample, int64_t n_windows, uint16_t *d_cls, float *d_mag,
                      int32_t *d_bin, unsigned long long *stats, hipStream_t st) {
    WN_REQUIRE(d_audio && d_cls && d_mag && d_bin, "null device pointer");
    WN_REQUIRE(((uintptr_t)d_audio % 16) == 0 && audio_len % 4 == 0, "d_audio 16-byte aligned, audio_len % 4 == 0");
    const int64_t span = first_sample + n_windows * (int64_t)ctx->n;
    WN_REQUIRE(n_streams == 1 || span <= stream_stride, "windows of a stream overlap the next stream");
    WN_REQUIRE((n_streams - 1) * stream_stride + span <= audio_len, "windows reach past audio_len");
    int rc;
    if (ctx->n1 == 83 && ctx->n2 == 85)
        rc = launch_detect<83, 85>(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls,
                                   d_mag, d_bin, stats, st);
    else if (ctx->n1 == 45 && ctx->n2 == 49)
        rc = launch_detect<45, 49>(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls,
                                   d_mag, d_bin, stats, st);
    else {
        wn_set_error("no detection kernel for %d x %d", ctx->n1, ctx->n2);
        return WN_ERR_UNSUPPORTED;
    }
    if (rc != WN_OK) return rc;
    const long long total = n_streams * n_windows;
    const int chunk = total < (long long)ctx->num_cus * 8 * 256 ? 64 : 256;
    const long long chunks = (total + chunk - 1) / chunk;
    const long long b0 = chunks < (long long)ctx->num_cus * 8 ? chunks : (long long)ctx->num_cus * 8;
    const long long b1 = chunks < (long long)ctx->num_cus * 2 ? chunks : (long long)ctx->num_cus * 2;
    const size_t lds0 = (size_t)(128 + (ctx->n + 127) / 128) * sizeof(double2);
    const size_t lds1 = (size_t)(ctx->n1 * (ctx->n2 / 2 + 1) + ctx->n1 + ctx->n2) * sizeof(double2);
    hipLaunchKernelGGL(refine_kernel<0>, dim3((unsigned)b0), dim3(256), lds0, st, d_audio, (long long)n_streams,
                       (long long)stream_stride, (long long)first_sample, (long long)n_windows, ctx->n, ctx->n1, ctx->n2,
                       ctx->e1, ctx->e2, ctx->d_cls_lut, d_cls, d_mag, d_bin, stats, chunk);
    rc = wn_launch_check("refine_kernel<0>");
    if (rc != WN_OK) return rc;
    hipLaunchKernelGGL(refine_kernel<1>, dim3((unsigned)b1), dim3(256), lds1, st, d_audio, (long long)n_streams,
                       (long long)stream_stride, (long long)first_sample, (long long)n_windows, ctx->n, ctx->n1, ctx->n2,
                       ctx->e1, ctx->e2, ctx->d_cls_lut, d_cls, d_mag, d_bin, stats, chunk);
    return wn_launch_check("refine_kernel");
}

extern "C" int wn_detect_windows(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                                 int64_t stream_stride, int64_t first_sample, int64_t n_windows, uint16_t *d_cls,
                                 float *d_peak_mag, int32_t *d_peak_bin, void *stream) {
    WN_REQUIRE(ctx && n_streams >= 0 && n_windows >= 0 && first_sample >= 0 && stream_stride >= 0, "ctx / sizes");
    if (n_streams == 0 || n_windows == 0) return WN_OK;
    WN_REQUIRE_DEVICE(ctx);
    return detect_run(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls, d_peak_mag,
                      d_peak_bin, ctx->d_stats, (hipStream_t)stream);
}

// The same two kernels with the per-window magnitude and bin arrays and the counters in the
// library's own per-device workspace: no allocation of any kind happens in the call.  Calls on
// one GPU share that workspace and must not overlap.
extern "C" int wn_detect_windows_noalloc(wn_ctx *ctx, const float *d_audio, int64_t audio_len, int64_t n_streams,
                                         int64_t stream_stride, int64_t first_sample, int64_t n_windows,
                                         uint16_t *d_cls, void *stream) {
    WN_REQUIRE(ctx && n_streams >= 0 && n_windows >= 0 && first_sample >= 0 && stream_stride >= 0, "ctx / sizes");
    if (n_streams == 0 || n_windows == 0) return WN_OK;
    if (n_streams > WN_DETECT_NOALLOC_MAX_WINDOWS || n_windows > WN_DETECT_NOALLOC_MAX_WINDOWS ||
        n_streams * n_windows > WN_DETECT_NOALLOC_MAX_WINDOWS) {
        wn_set_error("bad argument: %lld x %lld windows exceed the %d windows the library's workspace holds; "
                     "split the call or use wn_detect_windows",
                     (long long)n_streams, (long long)n_windows, WN_DETECT_NOALLOC_MAX_WINDOWS);
        return WN_ERR_ARG;
    }
    WN_REQUIRE_DEVICE(ctx);
    unsigned char *ws = g_scratch_ptr[ctx->device];
    if (!ws) {
        wn_set_error("workspace of device %d was not set up", ctx->device);
        return WN_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    // The clearing and the three launches of a call are enqueued as one unit: two threads calling at
    // once cannot interleave them.  Outside graph capture the call also waits (on the GPU) for the
    // previous call's last launch, whatever stream that was on, and leaves an event behind its own:
    // calls on different streams then follow each other on the workspace instead of racing.  A call
    // that is being captured into a graph records and waits for nothing (an event from outside the
    // capture may not be waited for inside it): the replays of such a graph must not overlap with other
    // calls on the same GPU, which the caller has to see to.
    std::lock_guard<std::mutex> hold(g_noalloc_mutex[ctx->device]);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    WN_HIP_TRY(hipStreamIsCapturing(st, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && g_noalloc_recorded[ctx->device])
        WN_HIP_TRY(hipStreamWaitEvent(st, g_noalloc_event[ctx->device], 0));
    // the counters start at zero in every call; magnitudes and bins are written for every window
    // of the call before they are read, so what the previous call left there does not matter
    WN_HIP_TRY(hipMemsetAsync(ws, 0, NOALLOC_HEAD, st));
    int32_t *bins = (int32_t *)(ws + NOALLOC_HEAD);
    float *mags = (float *)(ws + NOALLOC_HEAD + 4ull * WN_DETECT_NOALLOC_MAX_WINDOWS);
    const int rc = detect_run(ctx, d_audio, audio_len, n_streams, stream_stride, first_sample, n_windows, d_cls, mags,
                              bins, (unsigned long long *)ws, st);
    if (rc == WN_OK && !capturing) {
        WN_HIP_TRY(hipEventRecord(g_noalloc_event[ctx->device], st));
        g_noalloc_recorded[ctx->device] = true;
    }
    return rc;
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
