// 3x3 stride-1 convolution / data gradient with f16c8 operands -- ONE WAVE PER SIMD, 4 x 4 register blocking.
//
// Same arithmetic, operand formats, weight packs and C-ABI entry as conv_f16c8.hip (x*w ~= f16(x)*f16(w) on
// v_mfma_f32_32x32x16_f16 + one K-concatenated block-scaled fp8 correction product on
// v_mfma_scale_f32_32x32x64_f8f6f4); what changes is the shape of the work on a CU:
//
//   conv_f16c8.hip      8 waves (2 per SIMD, <= 256 registers), wave tile 2 co-tiles x 2 pixel rows of 32 x 32,
//                       block tile 128 co x 8 rows x 32 px: per tap a wave reads 16 KB of fragments for 12 MFMAs
//   this file           4 waves (1 per SIMD, the whole 512-register file: 256 accumulators + 256 vector registers),
//                       wave tile 4 co-tiles x 4 pixel rows, block tile 128 co x 16 rows x 32 px: per tap a wave
//                       reads 32 KB of fragments for 48 MFMAs -- half the LDS bytes, half the weight traffic and a
//                       1.2 x halo (18 x 34 / 16 x 32) instead of 1.33 x per MFMA.
//
// Pipeline of one (tile, 32-channel chunk) = 18 PHASES of 1024 MFMA cycles each:
//     phases 0..8   the nine taps' f16 products      (32 x v_mfma_f32_32x32x16_f16, reads the f16 halo image)
//     phases 9..17  the nine taps' fp8 corrections   (16 x v_mfma_scale_f32_32x32x64_f8f6f4, reads the fp8 halo image)
// so each of the two halo images (39 KB) is idle for half a chunk, and that is when the NEXT chunk's half is written
// into it (f16 image at phase 8, fp8 image at phase 17): single-buffered images, no barrier pair around a refill, no
// bubble at chunk or tile boundaries.  The next chunk travels global -> registers (phases 0..4) -> converted in
// place (phases 3..7, fp32 inputs only) -> LDS.  Weights: 8 KB per phase, global -> registers two phases ahead ->
// a 3-slot LDS ring (slot = phase % 3: the streamed A fragments of phase p are still being read during phase p); every load is a plain load, so the compiler counts vmcnt exactly and no
// hand-counted wait is needed.  Within a phase the MFMAs run co-tile by co-tile: the A fragments of co-tile i+1 and
// one row of the next phase's B fragments are requested in front of co-tile i's MFMAs (A: 2 x 8 registers, streamed;
// B: 2 x 32 registers, double buffered across phases).
// One workgroup barrier per phase publishes the weight slot and the halo writes.
//
// Restrictions (the host falls back to conv_f16c8.hip otherwise): plain 3x3 form (no sub-pixel / space-to-depth),
// Cout % 128 == 0, Cin % 32 == 0, Hout % 16 == 0, Wout % 32 == 0, at least two tiles per CU.
#include <stdlib.h>

#include <type_traits>

#include "conv_f16c8_common.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// mx6_quant (conv_f16c8_common.h) on values that already sit in two sixteen-register tuples (k = 0..15 | 16..31)
__device__ __forceinline__ void mx6_quant16(const f32x16& a, const f32x16& b, float amax, i32x4& p0, i32x4& p1) {
  typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
  const unsigned sb = mx6_scale_byte(amax);
  const float div = __builtin_bit_cast(float, sb << 23);
  u32x6 q;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a), "v"(b), "v"(div));
  p0[0] = (int)q[0]; p0[1] = (int)q[1]; p0[2] = (int)q[2]; p0[3] = (int)q[3];
  p1[0] = (int)q[4]; p1[1] = (int)q[5]; p1[2] = (int)sb; p1[3] = 0;
}

#ifndef EOVAE_W4_SGB
#define EOVAE_W4_SGB 1   // scheduling-group pattern inside a co-tile group (experiment hook)
#endif

// Diagnostic build only (-DEOVAE_W4_STAMP=1, tools/w4_stamps.py): shader-clock stamps per tile of the first blocks,
// written to a buffer nothing else reads.  The product build has no stamp code.
#ifndef EOVAE_W4_STAMP
#define EOVAE_W4_STAMP 0
#endif
#if EOVAE_W4_STAMP
#define W4_NPT 7
#define W4_MAXREC 32
#define W4_NBLK 8
__device__ unsigned long long w4_stamps[W4_NBLK * 4 * W4_MAXREC * W4_NPT];
#define W4_STAMP(id)                                                                                          \
  do {   /* blocks 0..3 and 256..259: a CU's first and (two blocks per CU) second block */                    \
    const int sb_ = (int)((blockIdx.x >> 8) * 4 + (blockIdx.x & 255));                                         \
    if (lane == 0 && (blockIdx.x & 255) < 4 && blockIdx.x < 512 && rec < W4_MAXREC)                            \
      w4_stamps[((sb_ * 4 + wave) * W4_MAXREC + rec) * W4_NPT + (id)] =                                        \
          (id) == 5 ? __builtin_amdgcn_s_memrealtime()                                                        \
                    : ((id) == 6 ? (unsigned long long)__builtin_amdgcn_s_getreg(63492) /* HW_ID, 32 bits */           \
                                 : __builtin_amdgcn_s_memtime());                                                     \
  } while (0)
#else
#define W4_STAMP(id) do {} while (0)
#endif

// NI = pixel rows per wave: 4 (16-row tiles, one block per CU, 512 registers per wave) or 2 (8-row tiles, TWO blocks
// per CU with 256 registers per wave, started half a tile apart so that one block's epilogue -- which is bound by the
// bytes a CU can move, ~13 B/clk, whatever the instruction width (as measured in round 4; round 6 found why: the memory
// pipeline's lane-quad rule, see the epilogue) -- runs beside the other block's MFMAs).
// MX (round 5): the correction product on block-scaled FP6 operands (conv_f16c8_common.h) -- phases 9..17 run 16 x
// v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 operands (~0.6 x the cycles of the fp8 form), each lane's scale byte riding in
// dword 6 of its fragment.  Same LDS images, same fragment reads (the 32 FP6 bytes of a lane = 24 data + scale + pad sit
// where its 32 fp8 bytes sat).  The input is then MX operand records (format 2, groupnorm.hip gn_apply_mx_kernel): the f16
// octets from the F plane (a.x: [n][Cin / 8][pixel][16 B]), the FP6 pieces from the S plane (a.x6: [n][Cin / 32][piece
// 0..3][pixel][16 B], piece = wave) -- two 16-byte loads and two 16-byte LDS stores per task.
template <bool PACKED, int NI, bool MX, bool RES>   // RES: the layer adds a residual (a.res != null)
__global__ __launch_bounds__(256, NI == 2 ? 2 : 1) void conv3x3_f16c8_w4_kernel(ConvCArgs a) {
  // FQ (round 5, late): fp32 input WITH the MX-FP6 correction product -- the data gradients.  The block-scaled operands need
  // the 32 channels of a pixel's block in ONE thread (block amax, v_cvt_scalef32_2xpk16_fp6_f32 takes 32 values), so the
  // staging changes its shape: a thread takes halo pixels tid, tid + 256, tid + 512 with all 32 channels of the chunk
  // (96 dword loads per chunk instead of 80, ~2/3 of the conversion VALU of the fp8 form: one f16 pack + one residual per
  // value, two quantising instructions per pixel) and keeps, per pixel, 16 registers of f16 pairs until the f16 image is free
  // (phases 9, 11, 15) and 16 of FP6 pieces until the FP6 image is (phase 17).  The load / convert / write schedule below keeps
  // the live staging registers at the fp8 form's 80.
  constexpr bool FQ = !PACKED && MX;
  static_assert(!FQ || NI == 4, "FQ: 16-row tiles");
  constexpr int TH = 4 * NI, TW = 32, BM = 128, KC = 32;
  constexpr int XW = TW + 2, NPX = (TH + 2) * XW;          // 18 x 34 = 612 (10 x 34 = 340) halo pixels
  constexpr int IMG = 4 * NPX * 16;                         // one halo image: f16 [4 octets][pixel][8 ch] = 39168 B;
  constexpr int OFF_X8 = IMG;                               // fp8 [lo' | hi'][2 x 16 ch][pixel][16 B] = the same size
  constexpr int OFF_W = 2 * IMG, WPART = 8192, NSLOT = 3;   // weight ring: 3 slots of one phase's 8 KB (slot = phase % 3)
  constexpr int OFF_GN = OFF_W + NSLOT * WPART;
  constexpr int GNS = 4 * 2 * 16 * 16 * 4;                  // GroupNorm partials [wave][sum | sumsq][batch][lane quad]
  constexpr int NTASK = (NPX + 63) / 64;                    // halo tasks per thread: wave w stages channel octet w,
                                                            // lane l pixels l, l + 64, ... (612 = 9 x 64 + 36)
  constexpr int OFF_DUMMY = OFF_GN + GNS;                   // 1 KB that swallows the writes of the last task's idle lanes
  constexpr bool LT = NI == 4;                              // epilogue: accumulator blocks transposed through LDS (below)
  constexpr int OFF_T = OFF_DUMMY + 1024, TBUF = LT ? 4 * 4096 : 0;   // [wave][32 channels][32 pixels] fp32
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_T + TBUF];
  static_assert(OFF_T + TBUF <= (NI == 2 ? 80 : 160) * 1024, "LDS");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wrow = wave * NI;
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const size_t out_plane = (size_t)a.Hout * a.Wout;

  struct Tile { int cob, n, oy0, ox0; };
  auto decode = [&](int b) {
    Tile t;
    if (a.xcd_remap) {  // the ncob blocks sharing one halo tile take consecutive slots of ONE XCD
      const int xcd = b & 7, q = b >> 3;
      t.cob = q % a.ncob;
      b = (q / a.ncob) * 8 + xcd;
    } else {
      t.cob = b % a.ncob;
      b /= a.ncob;
    }
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    t.n = b / a.tiles_y;
    t.oy0 = ty * TH;
    t.ox0 = tx * TW;
    return t;
  };

  // operand scales, as conv_f16c8.hip
  float xs = 1.f, xinv = 1.f;
  if (a.xamax) {
    const unsigned bits = *a.xamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      xs = pow2f(13 - e);
      xinv = pow2f(e - 13);
    }
  }
  const unsigned wbits = *reinterpret_cast<const unsigned*>(a.wp + (size_t)9 * a.nchunk * a.ncob * C8_WTILE);
  const int e1 = wbits ? 7 - expo_of_bits(wbits) : 0;
  auto e8m0 = [](int k) { k += 127; return k < 0 ? 0 : (k > 254 ? 254 : k); };
  const int sc_a = e8m0(-e1 - 12);
  const int sc_b = e8m0(0);

  // ---------------------------------------------------------------- halo staging (wave = channel octet)
  constexpr unsigned UNIT = MX ? 16u : (PACKED ? 32u : 4u);   // bytes per pixel step of the source (MX planes / records / fp32 plane)
  float xr[NTASK][8];
  unsigned xoff[NTASK];
  unsigned xokm = 0;
  // FQ staging state: the pixel in conversion (two sixteen-register tuples: the quantising instruction's operands), per task
  // the f16 pairs and the four FP6 pieces, the LDS byte offset of the task's pixel in a plane (or the dummy area)
  f32x16 fa[3], fb[3];
  int fh[3][16];
  i32x4 f6[3][4];
  float fahi[3], falo[3];
  const unsigned ftid16 = (unsigned)tid * 16u;          // pixel tid + 256 k sits at ftid16 + 4096 k of a plane
  const bool fidle2 = tid >= NPX - 512;                 // the third task covers pixels 512..611: threads 0..99
  auto setup_x = [&](const Tile& t) __attribute__((always_inline)) {
    xokm = 0;
    if constexpr (FQ) {
      int th = tid;
      asm volatile("" : "+v"(th));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int pix = th + 256 * k;
        const int rr = (pix * 241) >> 13, cc = pix - XW * rr;   // pix / 34, exact below 1024
        const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
        const bool ok = (pix < NPX) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
        xokm |= (ok ? 1u : 0u) << k;
        xoff[k] = 4u * (unsigned)(ok ? rr * a.Win + cc : a.Win + 1);
      }
      return;
    }
    int ln = lane;
    asm volatile("" : "+v"(ln));   // keeps the per-task index math out of the registers that live across the chunk loop
#pragma unroll
    for (int k = 0; k < NTASK; ++k) {
      const int pix = ln + 64 * k;
      const int rr = (pix * 241) >> 13, cc = pix - XW * rr;   // pix / 34, exact for pix < 640
      const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
      const bool ok = (pix < NPX) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
      xokm |= (ok ? 1u : 0u) << k;
      // BYTE offset from the halo origin (oy0 - 1, ox0 - 1); an off-image pixel reads the tile's first pixel instead
      xoff[k] = UNIT * (unsigned)(ok ? rr * a.Win + cc : a.Win + 1);
    }
  };
  // source of a chunk's octet for this wave: (tile, first channel c0) -> pointer at the halo origin (may lie in front of
  // the plane: it is only dereferenced at offsets of valid pixels)
  auto xbase = [&](const Tile& t, int c0) __attribute__((always_inline)) {
    const long origin = (long)(t.oy0 - 1) * a.Win + (t.ox0 - 1);
    if (PACKED)
      return reinterpret_cast<const char*>(a.x) +
             (((long)t.n * (a.Cin / 8) + c0 / 8 + wave) * (long)in_plane + origin) * (long)UNIT;
    if (FQ)   // all 32 channels of the chunk from every thread: channel j at + j planes
      return reinterpret_cast<const char*>(a.x) + (((long)t.n * a.Cin + c0) * (long)in_plane + origin) * 4;
    return reinterpret_cast<const char*>(a.x) + (((long)t.n * a.Cin + c0 + 8 * wave) * (long)in_plane + origin) * 4;
  };
  // ---- FQ staging steps (each takes a task k and a quarter i = 0..3 = channel octet i of the chunk)
  auto fq_load = [&](const char* xb, int k, int half, int i) __attribute__((always_inline)) {   // 4 channels: 16 half + 4 i ..
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * i + c;
      const float v = *reinterpret_cast<const float*>(xb + (size_t)(16 * half + j) * in_plane * 4 + xoff[k]);
      if (half == 0) fa[k][j] = v; else fb[k][j] = v;
    }
  };
  // conversion A, quarter i: scale (validity folded in) and running |max| of 8 channels; behind the last quarter quant(x)
  auto fq_conv_a = [&](int k, int i) __attribute__((always_inline)) {
    const float sm = ((xokm >> k) & 1) ? xs : 0.f;
    float m = i == 0 ? 0.f : fahi[k];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = 8 * i + c;
      const float v = (j < 16 ? fa[k][j] : fb[k][j - 16]) * sm;
      if (j < 16) fa[k][j] = v; else fb[k][j - 16] = v;
      m = __builtin_fmaxf(m, __builtin_fabsf(v));
    }
    fahi[k] = m;
    if (i == 3) mx6_quant16(fa[k], fb[k], m, f6[k][2], f6[k][3]);
  };
  // conversion B, quarter i: f16 pairs of 8 channels, residuals in place, their running |max|; behind the last quarter
  // quant(residual)
  auto fq_conv_b = [&](int k, int i) __attribute__((always_inline)) {
    float m = i == 0 ? 0.f : falo[k];
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
      const int j = 8 * i + c;
      const float v0 = j < 16 ? fa[k][j] : fb[k][j - 16], v1 = j < 16 ? fa[k][j + 1] : fb[k][j - 15];
      int hh;
      float l0, l1;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(v0), "v"(v1));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hh), "v"(v0));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hh), "v"(v1));
      fh[k][j >> 1] = hh;
      if (j < 16) { fa[k][j] = l0; fa[k][j + 1] = l1; } else { fb[k][j - 16] = l0; fb[k][j - 15] = l1; }
      m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(l0), __builtin_fabsf(l1)));
    }
    falo[k] = m;
    if (i == 3) mx6_quant16(fa[k], fb[k], m, f6[k][0], f6[k][1]);
  };
  // writes: octet i of task k into the f16 image / piece i into the FP6 image (image [plane][pixel][16 B]; the third task's
  // idle threads write into the dummy area: no branch in the phase sequence)
  auto fq_write16 = [&](int k, int i) __attribute__((always_inline)) {
    i32x4 w;
    w[0] = fh[k][4 * i]; w[1] = fh[k][4 * i + 1]; w[2] = fh[k][4 * i + 2]; w[3] = fh[k][4 * i + 3];
    const unsigned dst = (k == 2 && fidle2) ? (unsigned)(OFF_DUMMY + lane * 16) : (unsigned)(i * (NPX * 16) + k * 4096) + ftid16;
    *reinterpret_cast<i32x4*>(smem + dst) = w;
  };
  auto fq_write8 = [&](int k, int i) __attribute__((always_inline)) {
    const unsigned dst = (k == 2 && fidle2) ? (unsigned)(OFF_DUMMY + lane * 16) : (unsigned)(OFF_X8 + i * (NPX * 16) + k * 4096) + ftid16;
    *reinterpret_cast<i32x4*>(smem + dst) = f6[k][i];
  };
  // MX: piece `wave` of the chunk's FP6 plane, 16 bytes per pixel (the same offsets as the F plane)
  auto x6base = [&](const Tile& t, int c0) __attribute__((always_inline)) {
    const long origin = (long)(t.oy0 - 1) * a.Win + (t.ox0 - 1);
    return reinterpret_cast<const char*>(a.x6) +
           ((((long)t.n * (a.Cin / 32) + c0 / 32) * 4 + wave) * (long)in_plane + origin) * 16;
  };
  // LDS destinations of task k: compile-time offsets from two lane bases
  const unsigned st16 = (unsigned)(wave * (NPX * 16) + lane * 16);
  const unsigned st8 = MX ? (unsigned)(OFF_X8 + (wave * NPX + lane) * 16)      // image [part][piece][pixel] = [wave][pixel]
                          : (unsigned)(OFF_X8 + ((wave >> 1) * NPX + lane) * 16 + (wave & 1) * 8);

  // fp32 path: channels 4 q .. 4 q + 3 of task k converted in place: xr[k][4q .. 4q+3] = {f16 pair, f16 pair, lo' x 4, hi' x 4}
  auto convert_half = [&](int k, int q) __attribute__((always_inline)) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const float sm = ((xokm >> k) & 1) ? xs : 0.f;   // validity folded into the scale
    float v[4], lo[4];
    int pk[2];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      v[j] = xr[k][4 * q + j] * sm;
      v[j + 1] = xr[k][4 * q + j + 1] * sm;
      int hh;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(v[j]), "v"(v[j + 1]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hh), "v"(v[j]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hh), "v"(v[j + 1]));
      pk[j >> 1] = hh;
    }
    s16x2 l0 = {0, 0};
    l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[0], lo[1], 0x1p-12f, false);
    l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[2], lo[3], 0x1p-12f, true);
    int h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
    h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h0, true);
    xr[k][4 * q] = __builtin_bit_cast(float, pk[0]);
    xr[k][4 * q + 1] = __builtin_bit_cast(float, pk[1]);
    xr[k][4 * q + 2] = __builtin_bit_cast(float, l0);
    xr[k][4 * q + 3] = __builtin_bit_cast(float, h0);
  };
  // LDS writes of task k.  Register layout per task -- fp32 path: see convert_half; records: [0..3] = 8 x f16,
  // [4..5] = lo' x 8, [6..7] = hi' x 8 (an off-image pixel is an all-zero record).
  // (no branch anywhere in the phase sequence: with basic-block boundaries inside it the compiler sinks the MFMA chains
  // of whole phases towards the epilogue and spills their operands; the idle lanes of the last task -- pixels 612..639
  // -- write into a dummy area instead)
  const bool idle9 = lane >= NPX - 64 * (NTASK - 1);
  auto write16 = [&](int k) __attribute__((always_inline)) {
    i32x4 hv;
    if (PACKED) {
      const int keep = ((xokm >> k) & 1) ? -1 : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) hv[c] = __builtin_bit_cast(int, xr[k][c]) & keep;
    } else {
      hv[0] = __builtin_bit_cast(int, xr[k][0]); hv[1] = __builtin_bit_cast(int, xr[k][1]);
      hv[2] = __builtin_bit_cast(int, xr[k][4]); hv[3] = __builtin_bit_cast(int, xr[k][5]);
    }
    const unsigned dst = (k == NTASK - 1 && idle9) ? (unsigned)(OFF_DUMMY + lane * 16) : st16 + k * 1024;
    *reinterpret_cast<i32x4*>(smem + dst) = hv;
  };
  auto write8 = [&](int k) __attribute__((always_inline)) {
    if constexpr (MX) {   // xr[k][4..7] = this wave's 16-byte piece of the pixel's FP6 block
      const int keep = ((xokm >> k) & 1) ? -1 : 0;
      i32x4 pv;
#pragma unroll
      for (int c = 0; c < 4; ++c) pv[c] = __builtin_bit_cast(int, xr[k][4 + c]) & keep;
      const unsigned dst = (k == NTASK - 1 && idle9) ? (unsigned)(OFF_DUMMY + lane * 16) : st8 + k * 1024;
      *reinterpret_cast<i32x4*>(smem + dst) = pv;
      return;
    }
    i32x2 lv, uv;
    if (PACKED) {
      const int keep = ((xokm >> k) & 1) ? -1 : 0;
      lv[0] = __builtin_bit_cast(int, xr[k][4]) & keep; lv[1] = __builtin_bit_cast(int, xr[k][5]) & keep;
      uv[0] = __builtin_bit_cast(int, xr[k][6]) & keep; uv[1] = __builtin_bit_cast(int, xr[k][7]) & keep;
    } else {
      lv[0] = __builtin_bit_cast(int, xr[k][2]); lv[1] = __builtin_bit_cast(int, xr[k][6]);
      uv[0] = __builtin_bit_cast(int, xr[k][3]); uv[1] = __builtin_bit_cast(int, xr[k][7]);
    }
    const bool idle = k == NTASK - 1 && idle9;
    const unsigned d0 = idle ? (unsigned)(OFF_DUMMY + lane * 16) : st8 + k * 1024;
    const unsigned d1 = idle ? (unsigned)(OFF_DUMMY + lane * 16 + 8) : st8 + k * 1024 + 2 * NPX * 16;
    *reinterpret_cast<i32x2*>(smem + d0) = lv;       // part 0: lo'
    *reinterpret_cast<i32x2*>(smem + d1) = uv;       // part 1: hi'
  };

  // ---------------------------------------------------------------- weights: 8 KB per phase through registers
  f32x4 wreg[2][2];
  // phase q (0..17) of chunk `chunk` of co-block `cob`: taps' f16 halves first, then their fp8 halves
  auto wsrc = [&](int cob, int chunk, int q) __attribute__((always_inline)) {
    const int tap = q < 9 ? q : q - 9;
    return a.wp + ((size_t)(tap * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE + (q < 9 ? 0 : (MX ? C8_WOFF6 : C8_WOFF8)) + tid * 16;
  };
  auto w_issue = [&](int set, const unsigned char* src) __attribute__((always_inline)) {
    wreg[set][0] = *reinterpret_cast<const f32x4*>(src);
    wreg[set][1] = *reinterpret_cast<const f32x4*>(src + 4096);
  };
  auto w_commit = [&](int set, int slot) __attribute__((always_inline)) {
    unsigned char* dst = smem + OFF_W + slot * WPART + tid * 16;
    *reinterpret_cast<f32x4*>(dst) = wreg[set][0];
    *reinterpret_cast<f32x4*>(dst + 4096) = wreg[set][1];
  };

  // ---------------------------------------------------------------- fragments
  f32x16 acc[4][NI];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  i32x4 Ab[2][2];        // streamed A fragments: [co-tile parity][k-step (f16) | 16-ch half (fp8)]
  i32x4 Bb[2][NI][2];     // B fragments: [phase parity][pixel row][k-step | half]
  const unsigned xl16 = (unsigned)(h * (NPX * 16) + (wrow * XW + r) * 16);
  const unsigned xl8 = (unsigned)(OFF_X8 + (h * 2) * (NPX * 16) + (wrow * XW + r) * 16);
  const unsigned wl16 = (unsigned)(OFF_W + h * (BM * 16) + r * 16);
  const unsigned wl8 = (unsigned)(OFF_W + (h * 2) * (BM * 16) + r * 16);
  // A fragments of co-tile i of phase q into Ab[i & 1]
  auto read_a = [&](auto qc, auto ic) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
    constexpr unsigned slot = (unsigned)((q % NSLOT) * WPART);
    if constexpr (q < 9) {
      Ab[i & 1][0] = *reinterpret_cast<const i32x4*>(smem + wl16 + slot + i * 512);
      Ab[i & 1][1] = *reinterpret_cast<const i32x4*>(smem + wl16 + slot + 2 * (BM * 16) + i * 512);
    } else {
      Ab[i & 1][0] = *reinterpret_cast<const i32x4*>(smem + wl8 + slot + i * 512);
      Ab[i & 1][1] = *reinterpret_cast<const i32x4*>(smem + wl8 + slot + BM * 16 + i * 512);
    }
  };
  // B fragments of pixel row j of phase q into Bb[q & 1][j]
  auto read_b = [&](auto qc, auto jc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, j = decltype(jc)::value;
    constexpr int tap = q < 9 ? q : q - 9, ky = tap / 3, kx = tap - 3 * ky;
    constexpr unsigned off = (unsigned)(((ky + j) * XW + kx) * 16);
    if constexpr (q < 9) {
      Bb[q & 1][j][0] = *reinterpret_cast<const i32x4*>(smem + xl16 + off);
      Bb[q & 1][j][1] = *reinterpret_cast<const i32x4*>(smem + xl16 + off + 2 * (NPX * 16));
    } else {
      Bb[q & 1][j][0] = *reinterpret_cast<const i32x4*>(smem + xl8 + off);
      Bb[q & 1][j][1] = *reinterpret_cast<const i32x4*>(smem + xl8 + off + NPX * 16);
    }
  };
  auto mma = [&](auto qc, auto ic) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
    if constexpr (q < 9) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ab[i & 1][s]),
                                                            __builtin_bit_cast(f16x8, Bb[q & 1][j][s]), acc[i][j], 0, 0, 0);
    } else {
      i32x8 av = __builtin_shufflevector(Ab[i & 1][0], Ab[i & 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
      // MX: the MFMA reads dwords 0..5 of a fragment as data and dword 6 as its scale.  Left alone, hipcc gives it a
      // SIX-register operand and copies dwords 4, 5 of the second ds_read_b128 next to the first one (two v_mov and an
      // early lgkmcnt wait per fragment: 16 per phase).  The empty asm keeps the fragment one eight-register tuple, of which
      // the operand and the scale are sub-registers.
      if constexpr (MX) asm("" : "+v"(av));
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        i32x8 bv = __builtin_shufflevector(Bb[q & 1][j][0], Bb[q & 1][j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (MX && i == 0) {   // once per fragment, written back: the later co-tiles read the same tuple
          asm("" : "+v"(bv));
          Bb[q & 1][j][0] = __builtin_shufflevector(bv, bv, 0, 1, 2, 3);
          Bb[q & 1][j][1] = __builtin_shufflevector(bv, bv, 4, 5, 6, 7);
        }
        if constexpr (MX)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 2 /* A: e2m3 */, 2 /* B: e2m3 */, 0,
                                                                      av[6], 0, bv[6]);
        else
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 0 /* A: e4m3 */, 1 /* B: e5m2 */, 0,
                                                                      sc_a, 0, sc_b);
      }
    }
  };

  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---------------------------------------------------------------- GroupNorm partial sums (as conv_f16c8.hip)
  float* const gslot = reinterpret_cast<float*>(smem + OFF_GN);
  auto gn_flush = [&](int tn, int tcb) __attribute__((always_inline)) {
    // 64 outputs (value v, batch b = (co-tile, channel octet), lane half hh) x 4 waves x 8 lane quads
    const int te = wave * 64 + fresh_lane();   // (not loop-invariant: no spill across the main loop, conv_f16c8_common.h)
    const int out = te >> 2, sub = te & 3;
    const int v = out >> 5, b = (out >> 1) & 15, hh = out & 1;
    const float* cell = gslot + ((sub * 2 + v) * 16 + b) * 16 + hh * 8;
    const float4 c0 = *reinterpret_cast<const float4*>(cell), c1 = *reinterpret_cast<const float4*>(cell + 4);
    float t = ((c0.x + c0.y) + (c0.z + c0.w)) + ((c1.x + c1.y) + (c1.z + c1.w));
    t += dpp_quad<0xB1>(t);
    t += dpp_quad<0x4E>(t);
    const int co = tcb + (b >> 2) * 32 + (b & 3) * 8 + 4 * hh;
    if (sub == 0) atomicAdd(a.gn_sums + ((size_t)tn * a.gn_groups + (co >> a.gn_shift)) * 2 + v, (double)t);
  };

  // ---------------------------------------------------------------- epilogue (full tiles only)
  // The accumulators hold one PIXEL per lane and 16 channels per block (channel = 8 qq + 4 h + k in register
  // 4 qq + k), so the natural store is a dword per lane: 256 store and (with a residual) 256 load instructions per
  // lane and tile, and the CU retires one such instruction per ~20 cycles whatever its width (stamps: 19 k cycles of
  // epilogue per tile without, 40-45 k with a residual, against 74 k MFMA cycles of a 128-channel tile; a deeper
  // residual prefetch and a start stagger between CUs changed nothing, tools/w4_stamps.py).  So each batch of
  // 4 channels x 4 pixels is TRANSPOSED inside its lane quad (DPP, 1 VALU instruction per value): lane 4 pq + m
  // then owns channel 8 qq + 4 h + m at pixels 4 pq .. 4 pq + 3, and every load / store moves 16 bytes per lane --
  // eight complete 128-byte lines per instruction, a quarter of the instructions.
  // (Swapping the MFMA operands gives a lane four consecutive pixels without a transpose, but then its 64 lanes sit in
  // 32 different channel planes: 32 scattered 32-byte sectors per instruction ran 2.5 x SLOWER than dword stores.)
  auto quad_transpose = [&](float (&v)[4], bool e0, bool e1) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c += 2) {   // lane bit 0 <-> register bit 0
      const float send = e0 ? v[c] : v[c + 1];
      const float rcv = dpp_quad<0xB1>(send);
      v[c] = e0 ? rcv : v[c];
      v[c + 1] = e0 ? v[c + 1] : rcv;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {      // lane bit 1 <-> register bit 1
      const float send = e1 ? v[c] : v[c + 2];
      const float rcv = dpp_quad<0x4E>(send);
      v[c] = e1 ? rcv : v[c];
      v[c + 2] = e1 ? v[c + 2] : rcv;
    }
  };
  // Round 6 (tools/probe_store_pattern.hip, profiles/r06o_*): the memory pipeline takes a wave's 16-byte accesses one QUAD of
  // consecutive lanes at a time, and the transposed layout above -- a quad = four CHANNELS -- costs it four lines per quad
  // (16 bytes per clock and CU against 38..40 for a quad that covers one 64-byte segment): THAT is the "one instruction
  // per ~20 cycles whatever its width" above.  The 16-row form (NI = 4, the one that ships) therefore transposes each
  // accumulator block through the wave's 4 KB of LDS instead (see the loop body): lane 32 h + 8 m + pq owns channel
  // 8 qq + 4 h + m at pixels 4 pq .. 4 pq + 3 -- eight lanes = one 128-byte channel row -- and the ~20 VALU instructions per
  // 16 values of the DPP transposition are gone (-6 / -4 / -2.4 % per layer at 128 / 256 / 512 channels,
  // profiles/r06o_w4_epilogue_ab.log).  The parked 8-row form has no LDS left for that: it keeps the DPP transposition
  // and passes each register through one ds_bpermute_b32 (lane 32 h + 8 m + pq <- lane 32 h + 4 pq + m) to get the same
  // lane order.  Bias, residual and statistics run in that layout.
  // Every vector-memory instruction of the epilogue is an asm statement with hand-counted waits (the compiler does not
  // count asm stores, so its waits for builtin residual loads drained the stores in front of them), a layer without a
  // residual issues no residual loads (they were 64 zero-range loads per wave), and the per-lane addresses derive from
  // fresh_lane() so that they are not hoisted out of the tile loop and spilled (conv_f16c8_common.h).
  // (RES is a kernel parameter, not a branch: with control flow around the epilogue the register allocator copied the
  // accumulators out of the AGPRs early and spilled hundreds of registers.)
  auto epilogue_tile = [&](const Tile& t) __attribute__((always_inline)) {
    constexpr int T = 4 * NI;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to the compiler (conv_f16c8_q16.hip epilogue_tile)
    const int co_blk = t.cob * BM;
    const int le = fresh_lane();
    const bool e0 = le & 1, e1 = le & 2;
    // channel 4 eh + em of an octet, pixel quartet epq
    const int eh = le >> 5, em = (LT || EOVAE_EPI_QUADS) ? (le >> 3) & 3 : le & 3, epq = (LT || EOVAE_EPI_QUADS) ? le & 7 : (le >> 2) & 7;
    const int psrc = (32 * eh + 4 * epq + em) * 4;
    float* const tbw = reinterpret_cast<float*>(smem + OFF_T) + wave * 1024 + eh * 4 * 32 + (le & 31);
    const float* const tbr = reinterpret_cast<const float*>(smem + OFF_T) + wave * 1024 + (le >> 3) * 32 + (le & 7) * 4;
    const unsigned span = (unsigned)(BM * out_plane * 4);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
        a.y + ((size_t)t.n * a.Cout + co_blk) * out_plane, 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? a.res + ((size_t)t.n * a.Cout + co_blk) * out_plane : a.y), 0, RES ? span : 0u, 0x00020000);
    // (no branch for an absent bias: a descriptor of zero bytes, whose loads return 0 by the hardware's range check)
    const __amdgpu_buffer_rsrc_t bsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias + co_blk : a.y), 0, a.bias ? (unsigned)(BM * 4) : 0u, 0x00020000);
    const int lot = (int)(((size_t)(4 * eh + em) * out_plane + (size_t)(t.oy0 + wrow) * a.Wout + t.ox0 + 4 * epq) * 4);
    const int lob = (4 * eh + em) * 4;
    const int oct = (int)(8 * out_plane * 4), rowb = a.Wout * 4;
    float bias1[16];
    static_for<0, 16>([&](auto bc) __attribute__((always_inline)) { load32_asm<decltype(bc)::value * 32>(bias1[decltype(bc)::value], bsrd, lob); });
    // One accumulator block (co-tile i, row j: 16 channels x this lane's pixel) at a time, so that each 16-register
    // accumulator tuple leaves the AGPRs once and is consumed at once; its residual (4 x 16 bytes) is requested two
    // blocks ahead (three rotating buffers).  Queue: L0 L1 | L2 S0 | L3 S1 | ... : behind L(t) come S(t-2) L(t+1) S(t-1) L(t+2).
    auto load_res = [&](f32x4 (&rv)[4], int t) __attribute__((always_inline)) {   // t = NI i + j
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        if constexpr (RES) load128_asm(rv[qq], rsrd, lot, ((t / NI) * 4 + qq) * oct + (t % NI) * rowb);
        else rv[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    f32x4 rv0[4], rv1[4], rv2[4];
    load_res(rv0, 0);
    load_res(rv1, 1);
    float sv[4], qv[4];
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value, i = t / NI, j = t % NI;
      f32x4 (&rv)[4] = (t % 3 == 0) ? rv0 : ((t % 3 == 1) ? rv1 : rv2);
      f32x4 (&rnx)[4] = ((t + 2) % 3 == 0) ? rv0 : (((t + 2) % 3 == 1) ? rv1 : rv2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 2 < T) load_res(rnx, t + 2);
      if constexpr (RES) {
        constexpr int behind = (t >= 2 ? 4 : 0) + (t + 1 < T ? 4 : 0) + (t >= 1 ? 4 : 0) + (t + 2 < T ? 4 : 0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(behind) : "memory");
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) landed(rv[qq]);
      } else if constexpr (t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (t == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) landed(bias1[k]);
      }
      if constexpr (j == 0) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          sv[qq] = 0.f;
          qv[qq] = 0.f;
        }
      }
      const f32x16 blk = acc[i][j];
      if constexpr (LT) {
        // The block (32 channels x the 32 pixels of row j; register 4 qq + k = channel 8 qq + 4 h + k at pixel r) goes
        // through the wave's 4 KB of LDS: 16 dword writes [channel][pixel] (a wave half = 128 consecutive bytes), 4 reads of
        // 16 bytes [channel 8 s + (lane >> 3)][pixels 4 (lane & 7) ..] -- eight lanes = one 128-byte channel row, which is
        // also what the memory pipeline wants of a store (one 64-byte segment per lane quad).  The LDS queue of a wave is
        // in order: no barrier, and the next block's writes cannot pass this block's reads.
#pragma unroll
        for (int c = 0; c < 16; ++c) tbw[((c >> 2) * 8 + (c & 3)) * 32] = blk[c];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(tbr + qq * 8 * 32);
          f32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            o[k] = __builtin_fmaf(w[k], xinv, bias1[i * 4 + qq]) + rv[qq][k];
            sv[qq] += o[k];
            qv[qq] = __builtin_fmaf(o[k], o[k], qv[qq]);
          }
          store128_guarded(o, ysrd, lot, (i * 4 + qq) * oct + j * rowb);
        }
      } else {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = blk[4 * qq + k];
        quad_transpose(w, e0, e1);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float wk = EOVAE_EPI_QUADS
              ? __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, w[k]))) : w[k];
          o[k] = __builtin_fmaf(wk, xinv, bias1[i * 4 + qq]) + rv[qq][k];
          sv[qq] += o[k];
          qv[qq] = __builtin_fmaf(o[k], o[k], qv[qq]);
        }
        // (guarded: the gfx950 store-data hazard of conv_f16c8_common.h, tools/store_war_lint.py)
        store128_guarded(o, ysrd, lot, (i * 4 + qq) * oct + j * rowb);
      }
      }
      if constexpr (j == NI - 1) {
        // statistics of the following GroupNorm; written unconditionally, read by gn_flush() only when the caller asked
        // for them.  A quad holds four pixel quartets of ONE channel; the 8 cells of a lane half (lane >> 2 = 8 h + ...)
        // together cover its 4 channels (one group: gn_shift >= 2) x 32 pixels, which is all gn_flush() sums.
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float s1 = sv[qq], q1 = qv[qq];
          s1 += dpp_quad<0xB1>(s1); q1 += dpp_quad<0xB1>(q1);
          s1 += dpp_quad<0x4E>(s1); q1 += dpp_quad<0x4E>(q1);
          float* cell = gslot + (wave * 32 + i * 4 + qq) * 16 + (le >> 2);
          if ((le & 3) == 0) {
            cell[0] = s1;
            cell[16 * 16] = q1;
          }
        }
      }
    });
  };

  // ---------------------------------------------------------------- prologue
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  // Start stagger (experiment hook EOVAE_C8_W4_STAGGER, default off): eight block groups start
  // group x stagger x 1024 cycles apart.  Measured: no gain at 3 / 6 / 10.
  if (a.stagger) {
    const int g = (blockIdx.x >> 3) & 7;
    for (int i = 0; i < g * a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // Two blocks per CU (NI == 2): the CU's second block (odd workgroup slot, HW_ID.TG_ID) starts half a tile late, so
  // that from then on one block's epilogue falls into the other's phase sequence.  Nothing is lost while it sleeps:
  // the first block has the matrix pipes to itself.  (Placement is a speed matter only.)
  if (NI == 2 && a.skew) {
    const unsigned tg = __builtin_amdgcn_s_getreg(7172);   // hwreg(HW_REG_HW_ID, 16, 4): workgroup slot on the CU
    if (tg & 1)
      for (int i = 0; i < a.skew; ++i) __builtin_amdgcn_s_sleep(64);
  }
  Tile cur = decode(tile);
  setup_x(cur);
  {
    w_issue(0, wsrc(cur.cob, 0, 0));
    w_issue(1, wsrc(cur.cob, 0, 1));
    const char* xb = xbase(cur, 0);
    const char* x6b = (MX && PACKED) ? x6base(cur, 0) : nullptr;
    if constexpr (FQ) {
      static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
          fq_load(xb, k, 0, decltype(ic)::value);
          fq_load(xb, k, 1, decltype(ic)::value);
        });
      });
    }
#pragma unroll
    for (int k = 0; k < (FQ ? 0 : NTASK); ++k) {
      if (PACKED) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(xb + xoff[k]);
        const f32x4 q1 = MX ? *reinterpret_cast<const f32x4*>(x6b + xoff[k])
                            : *reinterpret_cast<const f32x4*>(xb + xoff[k] + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { xr[k][c] = q0[c]; xr[k][4 + c] = q1[c]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[k][j] = *reinterpret_cast<const float*>(xb + (size_t)j * in_plane * 4 + xoff[k]);
      }
    }
    w_commit(0, 0);
    w_commit(1, 1);
    w_issue(0, wsrc(cur.cob, 0, 2));
    w_issue(1, wsrc(cur.cob, 0, 3));
    if constexpr (FQ) {
      static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { fq_conv_a(k, decltype(ic)::value); });
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { fq_conv_b(k, decltype(ic)::value); });
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
          fq_write16(k, decltype(ic)::value);
          fq_write8(k, decltype(ic)::value);
        });
      });
    }
#pragma unroll
    for (int k = 0; k < (FQ ? 0 : NTASK); ++k) {
      if (!PACKED) { convert_half(k, 0); convert_half(k, 1); }
      write16(k);
      write8(k);
    }
  }
  zero_acc();
  barrier();
  {
    using Q0 = std::integral_constant<int, 0>;
    read_a(Q0{}, Q0{});
    static_for<0, NI>([&](auto jc) __attribute__((always_inline)) { read_b(Q0{}, jc); });
  }

  // ---------------------------------------------------------------- main loop
#if EOVAE_W4_STAMP
  int rec = 0;
#endif
  for (; tile < a.ntiles; tile += gridDim.x) {
    W4_STAMP(0);
    W4_STAMP(5);
    W4_STAMP(6);
    const bool has_next_tile = tile + (int)gridDim.x < a.ntiles;
    Tile nxt = cur;
    if (has_next_tile) nxt = decode(tile + gridDim.x);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool last_chunk = chunk + 1 == a.nchunk;
      // the chunk staged behind this one: next chunk, the next tile's first, or (end of work) this chunk again --
      // re-staging the chunk being multiplied writes the bytes that are already there
      const bool wrap = last_chunk && has_next_tile;
      const Tile& xt = wrap ? nxt : cur;
      const int xchunk = last_chunk ? (has_next_tile ? 0 : chunk) : chunk + 1;
      if (wrap) setup_x(nxt);
      const char* xb = xbase(xt, xchunk * KC);
      const char* x6b = (MX && PACKED) ? x6base(xt, xchunk * KC) : nullptr;
      static_for<0, 18>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int pn = (p + 1) % 18;
        using PN = std::integral_constant<int, pn>;
        barrier();   // weights of phase p + 1 and the halo writes of phase p - 1 are visible; frags(p) have landed
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (i == 1) {
            // weights of phase p + 2 into slot (p + 2) % 3: its previous tenant, phase p - 1, was last read (streamed
            // A fragments of co-tiles 1..3) during phase p - 1; loads of phase p + 4 into the freed registers.  Behind
            // the first co-tile's MFMAs: in front of them (right behind the barrier) the wait for the loads, the two LDS
            // writes and the address arithmetic sat between the barrier and the phase's first MFMA.
            w_commit(p & 1, (p + 2) % NSLOT);
            if constexpr (p + 4 < 18) w_issue(p & 1, wsrc(cur.cob, chunk, p + 4));
            else w_issue(p & 1, wsrc(xt.cob, xchunk, p + 4 - 18));
          }
          // fragment requests: next co-tile's A, row i of the next phase's B
          if constexpr (i < 3) read_a(pc, std::integral_constant<int, i + 1>{});
          else read_a(PN{}, std::integral_constant<int, 0>{});
          if constexpr (i < NI) read_b(PN{}, ic);
          // staging slice of this group
          if constexpr (FQ) {
            // task k: loads in phases (4, 5 | 6, 7 | 10, 11), conversion A / B in (7, 8 | 9, 10 | 13, 14), f16 pairs into the
            // f16 image in (9 | 11 | 15), FP6 pieces of all three into the FP6 image in 17 -- quarter i of each in group i
            if constexpr (p == 4 || p == 5) fq_load(xb, 0, p - 4, i);
            if constexpr (p == 6 || p == 7) fq_load(xb, 1, p - 6, i);
            if constexpr (p == 10 || p == 11) fq_load(xb, 2, p - 10, i);
            if constexpr (p == 7) fq_conv_a(0, i);
            if constexpr (p == 8) fq_conv_b(0, i);
            if constexpr (p == 9) { fq_write16(0, i); fq_conv_a(1, i); }
            if constexpr (p == 10) fq_conv_b(1, i);
            if constexpr (p == 11) fq_write16(1, i);
            if constexpr (p == 13) fq_conv_a(2, i);
            if constexpr (p == 14) fq_conv_b(2, i);
            if constexpr (p == 15) fq_write16(2, i);
            if constexpr (p == 17) { fq_write8(0, i); fq_write8(1, i); fq_write8(2, i); }
          } else if constexpr (PACKED) {
            // records: f16 halves of tasks 0..9 in phases 0, 1; [lo' | hi'] halves in phases 9, 10
            if constexpr (p == 0 || p == 1 || p == 9 || p == 10) {
              constexpr int TPP = (NTASK + 1) / 2;   // tasks per phase
              constexpr int half = p >= 9 ? 1 : 0, k0 = TPP * (p - 9 * half);
#pragma unroll
              for (int k = k0; k < k0 + TPP && k < NTASK; ++k) {
                if (((k - k0) & 3) != i) continue;
                const f32x4 qv = (MX && half) ? *reinterpret_cast<const f32x4*>(x6b + xoff[k])
                                              : *reinterpret_cast<const f32x4*>(xb + xoff[k] + 16 * half);
#pragma unroll
                for (int c = 0; c < 4; ++c) xr[k][4 * half + c] = qv[c];
              }
            }
          } else {
            // fp32: tasks 2p, 2p+1 loaded in phases 0..4 (channels 2i, 2i+1 in group i), converted three phases later
            if constexpr (2 * p < NTASK) {
#pragma unroll
              for (int k = 2 * p; k < 2 * p + 2 && k < NTASK; ++k)
#pragma unroll
                for (int j = 2 * i; j < 2 * i + 2; ++j)
                  xr[k][j] = *reinterpret_cast<const float*>(xb + (size_t)j * in_plane * 4 + xoff[k]);
            }
            if constexpr (p >= 3 && 2 * (p - 3) + (i >> 1) < NTASK) convert_half(2 * (p - 3) + (i >> 1), i & 1);
          }
          if constexpr (p == 8 && !FQ) {   // the f16 image is idle from here to phase 17's requests
#pragma unroll
            for (int k = 0; k < NTASK; ++k)
              if ((k & 3) == i) write16(k);
          }
          if constexpr (p == 17 && !FQ) {  // the fp8 image is idle from here to the next chunk's phase 8
#pragma unroll
            for (int k = 0; k < NTASK; ++k)
              if ((k & 3) == i) write8(k);
          }
          mma(pc, ic);
#if EOVAE_W4_SGB
          // spread the requests / staging work over the MFMA gaps instead of in front of the first MFMA
#pragma unroll
          for (int m = 0; m < (p < 9 ? 2 * NI : NI); ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x020, p < 9 ? 1 : 2, 0);   // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x002, p < 9 ? 2 : 4, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
          }
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    W4_STAMP(1);
    epilogue_tile(cur);
    W4_STAMP(2);
    zero_acc();
    W4_STAMP(3);
    if (a.gn_sums) {   // uniform.  The cells are rewritten by the next epilogue, a whole tile (>= 18 barriers) later.
      barrier();
      gn_flush(cur.n, cur.cob * BM);
    }
    W4_STAMP(4);
#if EOVAE_W4_STAMP
    ++rec;
#endif
    cur = nxt;
  }
}

}  // namespace

#if EOVAE_W4_STAMP
extern "C" int eovae_debug_w4_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(w4_stamps), bytes < sizeof(w4_stamps) ? bytes : sizeof(w4_stamps));
}
#endif

// rows per wave: 4 = 16-row tiles, one block per CU (default).  2 = 8-row tiles, two skewed blocks per CU: an
// EXPERIMENT (EOVAE_C8_W4_NI=2, record inputs only) that did not pay -- the CU's older block wins the issue arbitration,
// finishes its eight tiles in 0.69 M cycles while the younger one needs 0.93 M, and the layer takes as long as with
// one block per CU (tools/w4_stamps.py).  fp32 inputs always take 4 rows per wave: the 2-row fp32
// variant spills 50 registers into the phase sequence and is not used.
static int w4_rows_per_wave(const ConvCArgs& a, int n_cu) {
  (void)n_cu;
  if (EOVAE_PARKED && eovae_opt(EOVAE_OPT_C8_W4_NI) == 2 && a.packed == 1 && a.Hout % 8 == 0) return 2;   // (parked instantiations)
  return a.Hout % 16 == 0 ? 4 : 0;
}

bool eovae_c8_w4_eligible(const ConvCArgs& a, int subpix, int n_cu) {
  const int mode = eovae_opt(EOVAE_OPT_C8_W4);   // 0: never, 1: whenever the shape allows, -1 (default): when it pays
  if (mode == 0) return false;
  if (subpix != 0 || a.mode != MODE_NORMAL) return false;
  if (a.Cout % 128 || a.Cin % 32 || a.Wout % 32 || a.Hin != a.Hout || a.Win != a.Wout) return false;
  const int ni = w4_rows_per_wave(a, n_cu);
  if (!ni) return false;
  const long long tiles = (long long)a.N * (a.Hout / (4 * ni)) * (a.Wout / 32) * (a.Cout / 128);
  if (mode == 1) return tiles >= 1;
  return tiles >= (ni == 2 ? 4LL : 2LL) * n_cu;   // fewer tiles than two per block leave CUs idle: the 8-wave kernel takes those
}

// The ONE decision between the two one-wave-per-SIMD kernels for an eligible call (eovae_c8_w4_eligible): 2 = conv_f16c8_q16.hip
// (16 x 16 MFMAs: MX records, or fp32 inputs under option c8_q16 = 2), 1 = this file's kernel.  Used by the launcher below and
// by eovae_conv3x3_f16c8_which(), so that time attributed to a kernel (bench.py) is the kernel that ran.
int eovae_c8_one_wave_kind(ConvCArgs a, int n_cu) {
  const int ni = w4_rows_per_wave(a, n_cu);
  a.mx = a.packed == 2 ? 1 : 0;
  return (ni == 4 && (a.mx || a.packed == 0) && eovae_c8_q16_eligible(a)) ? 2 : 1;
}

void eovae_c8_w4_launch(ConvCArgs a, int n_cu, hipStream_t st) {
  const int ni = w4_rows_per_wave(a, n_cu);
  a.tiles_x = a.Wout / 32;
  a.tiles_y = a.Hout / (4 * ni);
  const long long sp_tiles = (long long)a.N * a.tiles_y * a.tiles_x;
  a.xcd_remap = (a.ncob > 1 && sp_tiles % 8 == 0) ? 1 : 0;
  a.ntiles = (int)(sp_tiles * a.ncob);
  a.stagger = eovae_opt(EOVAE_OPT_C8_W4_STAGGER);
  const int sk = eovae_opt(EOVAE_OPT_C8_W4_SKEW);   // experiment hook: units of 4096 cycles; -1 = half a tile
  a.skew = ni == 2 ? (sk >= 0 ? sk : (a.nchunk * 7 + 2) / 5) : 0;
  const int nblk = ni == 2 ? 2 * n_cu : n_cu;
  const dim3 grid((unsigned)(a.ntiles < nblk ? a.ntiles : nblk)), blk(256);
  a.mx = a.packed == 2 ? 1 : 0;   // MX operand records (format 2): their FP6 plane is the only correction operand they carry
  if (eovae_c8_one_wave_kind(a, n_cu) == 2) {   // 16 x 16 MFMAs (conv_f16c8_q16.hip): MX records / fp32 inputs
    eovae_c8_q16_launch(a, n_cu, st);
    return;
  }
  auto launch = [&](auto pc, auto nc, auto mc) {
    constexpr bool P = decltype(pc)::value, M = decltype(mc)::value;
    constexpr int NI_ = decltype(nc)::value;
    if (a.res) hipLaunchKernelGGL((conv3x3_f16c8_w4_kernel<P, NI_, M, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv3x3_f16c8_w4_kernel<P, NI_, M, false>), grid, blk, 0, st, a);
  };
  using std::integral_constant;
  typedef std::true_type Y;
  typedef std::false_type N;
#if EOVAE_PARKED
  if (ni == 2) {
    if (a.packed) launch(Y{}, integral_constant<int, 2>{}, N{});
    else launch(N{}, integral_constant<int, 2>{}, N{});
  } else
#endif
  {
    if (a.mx) launch(Y{}, integral_constant<int, 4>{}, Y{});
    else if (a.packed) launch(Y{}, integral_constant<int, 4>{}, N{});
    // fp32 input (the data gradients): MX-FP6 correction product quantised in the kernel (library options c8_mx, c8_dgrad_mx)
    else if (eovae_opt(EOVAE_OPT_C8_MX) && eovae_opt(EOVAE_OPT_C8_DGRAD_MX)) launch(N{}, integral_constant<int, 4>{}, Y{});
    else launch(N{}, integral_constant<int, 4>{}, N{});
  }
}
