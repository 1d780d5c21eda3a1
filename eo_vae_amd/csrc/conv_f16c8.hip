// 3x3 stride-1 convolution as an implicit GEMM with "f16c8" operands: one f16 product plus ONE
// block-scaled fp8 product that carries both first-order corrections -- 2 matrix-core units per
// algorithmic product where the split-bf16 kernel (conv_bf16x3.hip) needs 3.
//
//   x = xh + xl,  w = wh + wl        xh = f16(x), wh = f16(w)  (11 significant bits each)
//   x*w ~= xh*wh  +  [ xl*w + x*wl ]                (dropped: xl*wl ~ 2^-24)
// The bracket is a sum over TWICE the channels, so it is one K-concatenated product of 8-bit
// operands: per 32 input channels
//     v_mfma_f32_32x32x16_f16   x 2      (channels 0-15, 16-31:   xh * wh)
//     v_mfma_scale_f32_32x32x64_f8f6f4   (K = 64 = [ xl' (32 ch) | xh' (32 ch) ] . [ wh' | wl' ])
// with xl' = e5m2(xl * 2^12), xh' = e5m2(x) (bf8 has f16's exponent range: no per-tensor scale needed)
// and wh' = e4m3(w * 2^e), wl' = e4m3(wl * 2^(e+12)), e from the weight tensor's amax (found when the
// weights are packed); the powers of two ride in the instruction's E8M0 block-scale operands.  The
// scaled MFMA runs twice the bf16 rate, so the step costs 4 MFMA-units per 32 channels against 6.
// Lane (r, h) supplies the 32 bytes of row / column r for K-block h of BOTH operands, so byte j of the
// A fragment always meets byte j of the B fragment whatever the lane -> k map is (tools/probe_f8f6f4.hip
// checks exactly this on the hardware; profiles/r02_probe_f8f6f4.log).
// Error: each correction operand is good to 2^-3 .. 2^-4 of a term that is 2^-12 of the product:
// ~1e-5 per product; whole model vs the reference's fp32 CPU output 1.2e-4 (bar 1e-3), split-bf16
// 3e-5 (tools/scheme_probe_cpu.py).
//
// Gradients (dy) can sit far below f16's normal range (1/numel of the loss), so a data-gradient call
// passes the tensor's amax (eovae_absmax_bits); the loader multiplies by the power of two that puts it
// at 2^13..2^14 and the epilogue divides it out again (both exact).
//
// Tile / loader / epilogue geometry follow conv_bf16x3.hip (128 co x TH x 32 pixels, 8 waves, halo tile
// fp32 NCHW -> registers -> LDS); what differs: K chunks of 32 channels, LDS images
//     X : f16 [4 octets][pixel][8 ch]  +  fp8 [lo'|hi'][2 x 16 ch][pixel][16 B]      (128 B / pixel)
//     W : per tap 16 KiB = f16 [4 octets][128 co][8 ch] + fp8 [wh'|wl'][2][128 co][16 B]
// and the weights move through a ring of 4 tap slots in stages of two taps (a ky-row of three taps
// double buffered would need 96 KiB; with the 78 KiB halo tile of the 16-row block that is over 160 KiB).
#include "eovae_common.h"
#include <stdlib.h>
#include <type_traits>

#include "conv_f16c8_common.h"

namespace {

// Tap window of a parity (conv_bf16x3.hip has the derivation): first active tap / number of taps of
// an axis whose parity bit is pb.
template <int SUBPIX>
__device__ __forceinline__ int win0(int pb) {
  return SUBPIX == 1 ? pb : (SUBPIX == 2 ? 1 - pb : (SUBPIX == 3 ? 1 : (SUBPIX == 4 ? pb : 0)));
}
template <int SUBPIX>
__device__ __forceinline__ int wincnt(int pb) {
  return SUBPIX == 0 ? 3 : (SUBPIX >= 3 ? 2 - pb : 2);
}

// One PERSISTENT block per CU walks tiles  blockIdx.x, blockIdx.x + gridDim.x, ...  The unit of the
// software pipeline is the (tile, 32-channel chunk) pair: while chunk c is multiplied, chunk c+1 -- the
// next tile's first chunk after a tile's last -- is loaded (stage 0), converted in registers (second to
// last stage; waves 0-3 before their MFMAs, waves 4-7 after theirs, so that on every SIMD one wave
// converts while its partner multiplies) and written to the single halo buffer between the two
// barriers that end the chunk.  Weights arrive by LDS-DMA, two taps per stage, into the other half of
// a 4-slot ring.  A tile's epilogue runs with the next tile's first chunk already staged.
// PACKED: x is not fp32 NCHW but the operand records GroupNorm-apply wrote (groupnorm.hip: 32 B per pixel and
// channel octet = 8 f16 | 8 lo' | 8 hi'): a task is two 16-byte loads and three LDS stores, no conversion.
// Diagnostic build only (-DEOVAE_C8_STAMP=1, tools/c8_stamps.py): shader-clock stamps per pipeline stage of
// the first blocks, written to a buffer nothing else reads.  The product build has no stamp code.
#ifndef EOVAE_C8_STAMP
#define EOVAE_C8_STAMP 0
#endif
#if EOVAE_C8_STAMP
#define C8_NPT 8
#define C8_MAXREC 96
#define C8_NBLK 4
__device__ unsigned long long c8_stamps[C8_NBLK * 8 * C8_MAXREC * C8_NPT];
#define C8_STAMP(id)                                                                                          \
  do {                                                                                                        \
    if (lane == 0 && blockIdx.x < C8_NBLK && rec < C8_MAXREC)                                                 \
      c8_stamps[((blockIdx.x * 8 + wave) * C8_MAXREC + rec) * C8_NPT + (id)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define C8_STAMP(id) do {} while (0)
#endif
#ifndef EOVAE_C8_GNABL
#define EOVAE_C8_GNABL 0
#endif
#ifndef EOVAE_C8_ABL
#define EOVAE_C8_ABL 0   // timing-only experiments: 1 few stores, 2 no halo loads, 4 no conversion, 8 no weight DMA, 16 no MFMAs
#endif
// MX (round 5, PACKED only): x = MX operand records (format 2: F plane a.x, S plane a.x6; groupnorm.hip gn_apply_mx_kernel) and
// the correction product runs on block-scaled FP6 operands (conv_f16c8_common.h), as in conv_f16c8_w4.hip.
template <int TH, int SUBPIX, bool PACKED, bool MX = false>
__global__ __launch_bounds__(512, 2) void conv3x3_f16c8_kernel(ConvCArgs a) {
  static_assert(!MX || (PACKED && SUBPIX == 0), "MX: record input of a plain 3x3 layer");
  constexpr int NW = 8, TW = 32, BM = 128, KC = 32, NTHR = NW * 64;
  constexpr int XH = TH + 2, XW = TW + 2, NPX = XH * XW;
  constexpr int NI = TH / (NW / 2);             // pixel-row tiles per wave
  constexpr int XF16 = 4 * NPX * 16;            // f16 image: [octet][pixel][8 ch]
  constexpr int XSET = 8 * NPX * 16;            // + fp8 images [lo'|hi'][half][pixel][16 B]
  constexpr int WTAP = 16384, WF16 = 8192;   // LDS slot of one tap: the f16 + fp8 parts of its C8_WTILE-byte tile
  // two halo buffers where they fit beside the 4-slot weight ring (8-row tile: 2 x 42.5 + 64 KiB): the staged
  // chunk is converted straight into the other buffer, no barrier pair and no packed copy in registers
  constexpr bool XDB = 2 * XSET + 4 * WTAP <= 160 * 1024;
  constexpr int OFF_W = (XDB ? 2 : 1) * XSET;
  constexpr int XTASKS = NPX * 4;               // (pixel, 8-channel octet)
  constexpr int XPT = (XTASKS + NTHR - 1) / NTHR;
  // The halo loads of the next chunk are issued in 8 groups (one channel of every task each) behind the
  // MFMAs of taps 0-3 -- a burst of all of them stalled the issuing wave for 3-5 k cycles (TA back-pressure,
  // tools/c8_stamps.py) -- and the barriers wait with a COUNTED vmcnt that covers the stage's weight DMAs
  // (issued first) but leaves the groups issued after them in flight.
  constexpr int LPG = XPT;                      // loads per group
  constexpr bool KGATHER = SUBPIX == 2 || SUBPIX == 3;  // K channels gathered at a parity
  constexpr bool MSHUFFLE = SUBPIX == 1 || SUBPIX == 4; // M rows pixel-shuffled on store
  constexpr int GNS = NW * 2 * 8 * 16 * 4;      // GroupNorm partials of the tile being stored: [wave][sum|sumsq][batch][lane quad]
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_W + 4 * WTAP + GNS];
  static_assert(OFF_W + 4 * WTAP + GNS <= 160 * 1024, "LDS");

  const int tid = threadIdx.x;
  // readfirstlane: the wave index is uniform, which lets the role branches below be scalar branches
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wco = (wave & 1) * 64, wrow = (wave >> 1) * NI;
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const int xch = KGATHER ? a.creal : a.Cin;

  struct Tile { int cob, n, oy0, ox0, par1; };
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
    t.par1 = MSHUFFLE ? t.cob / a.cpb : 0;
    return t;
  };
  auto parity_of = [&](const Tile& t, int chunk) { return KGATHER ? (chunk * KC) / a.creal : t.par1; };
  auto ntaps_of = [&](const Tile& t, int chunk) {
    const int par = parity_of(t, chunk);
    return wincnt<SUBPIX>(par >> 1) * wincnt<SUBPIX>(par & 1);
  };
  // ti-th active tap of a chunk -> (ky, kx); ti is clamped so that a dummy second tap re-reads a real one
  auto tap_of = [&](const Tile& t, int chunk, int ti, int& ky, int& kx) {
    if (SUBPIX == 0) {
      ti = ti < 9 ? ti : 8;
      ky = ti / 3;
      kx = ti - 3 * ky;
    } else {
      const int par = parity_of(t, chunk);
      const int kxc = wincnt<SUBPIX>(par & 1), nt = wincnt<SUBPIX>(par >> 1) * kxc;
      ti = ti < nt ? ti : nt - 1;
      const int q = kxc == 2 ? ti >> 1 : ti;
      ky = win0<SUBPIX>(par >> 1) + q;
      kx = win0<SUBPIX>(par & 1) + (ti - q * kxc);
    }
  };

  // operand scales.  x: 2^sx puts the tensor's amax at 2^13..2^14 (1 when no amax is given).
  // weights: the packer left amax(|w|) bits in the header behind the last tile.
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
  const int e1 = wbits ? 7 - expo_of_bits(wbits) : 0;   // amax(w) * 2^e1 in [128, 256)
  auto e8m0 = [](int k) { k += 127; return k < 0 ? 0 : (k > 254 ? 254 : k); };
  // K-block 0 (lanes h = 0): xl' * wh' = 2^12 xl * 2^e1 w    K-block 1 (h = 1): xh' * wl' = x * 2^(e1+12) wl:
  // the same power of two for both blocks, so every lane passes the same scale bytes
  const int sc_a = e8m0(-e1 - 12);
  const int sc_b = e8m0(0);

  // ---- halo loader: raw fp32 in xr[][], converted IN PLACE to the packed LDS payload
  //      (xr[i][0..3] = 8 x f16, [4..5] = 8 x lo' bytes, [6..7] = 8 x hi' bytes)
  float xr[XPT][8];
  unsigned xoff[XPT];
  unsigned xokm = 0;
  auto setup_x = [&](const Tile& t) {
    xokm = 0;
    int xt = tid;    // laundered: keeps the per-task index math out of the registers that live across the tile loop
    asm volatile("" : "+v"(xt));
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int tk = xt + i * NTHR;
      const int g = tk / NPX, pix = tk - g * NPX;
      const int rr = pix / XW, cc = pix - rr * XW;
      int si = -1;
      if (tk < XTASKS) {
        if (KGATHER) {
          const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
          if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout) si = 2 * vy * a.Win + 2 * vx;
        } else {   // MODE_NORMAL only (the host routes the other modes to the split-bf16 kernel)
          const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
          if ((unsigned)vy < (unsigned)a.Hin && (unsigned)vx < (unsigned)a.Win) si = vy * a.Win + vx;
        }
      }
      xokm |= (si >= 0 ? 1u : 0u) << i;
      // BYTE offset: scalar base + 32-bit lane offset is one instruction (saddr form), no 64-bit lane address
      xoff[i] = PACKED ? (MX ? 16u : 32u) * ((unsigned)(tk < XTASKS ? g : 0) * (unsigned)in_plane + (unsigned)(si >= 0 ? si : 0))
                       : 4u * ((unsigned)((tk < XTASKS ? g : 0) * 8) * (unsigned)in_plane + (unsigned)(si >= 0 ? si : 0));
    }
  };
  // UNCONDITIONAL loads from clamped addresses; validity is applied at conversion (conv_bf16x3.hip).
  // Group j = channel j of every task's octet.
  auto load_x_group = [&](const Tile& t, int c0, int j) __attribute__((always_inline)) {
    int cbase = c0, poff = 0;
    if (KGATHER) {
      const int par = c0 / a.creal;
      cbase = c0 - par * a.creal;
      poff = (par >> 1) * a.Win + (par & 1);
    }
    if (PACKED) {   // group j (0, 1) = the j-th 16 bytes of every task's record (MX: its F-plane octet | its S-plane piece)
      const char* cj = MX ? (j == 0 ? reinterpret_cast<const char*>(a.x) + (((size_t)t.n * (xch / 8) + cbase / 8) * in_plane) * 16
                                    : reinterpret_cast<const char*>(a.x6) + (((size_t)t.n * (xch / 32) + cbase / 32) * 4 * in_plane) * 16)
                          : reinterpret_cast<const char*>(a.x) +
                                (((size_t)t.n * (xch / 8) + cbase / 8) * in_plane) * 32 + 16 * j;
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(cj + xoff[i]);
        xr[i][4 * j] = q[0]; xr[i][4 * j + 1] = q[1]; xr[i][4 * j + 2] = q[2]; xr[i][4 * j + 3] = q[3];
      }
      return;
    }
    const char* cj = reinterpret_cast<const char*>(a.x + ((size_t)t.n * xch + cbase + j) * in_plane + poff);
#pragma unroll
    for (int i = 0; i < XPT; ++i) xr[i][j] = *reinterpret_cast<const float*>(cj + xoff[i]);
  };
  // LDS byte offsets of this thread's tasks inside a halo buffer (f16 row, fp8 lo' row; ~0u = no task):
  // lane-only quantities, computed once
  unsigned xst16[XPT], xst8[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int tk = tid + i * NTHR;
    const int g = tk / NPX, pix = tk - g * NPX;
    xst16[i] = tk < XTASKS ? (unsigned)(g * (NPX * 16) + pix * 16) : ~0u;
    xst8[i] = (unsigned)(XF16 + ((g >> 1) * NPX + pix) * 16 + (g & 1) * 8);
  }
  // convert one task's 8 channels and write its three LDS payloads (f16 x 8, lo' x 8, hi' x 8) to buffer xb.
  // 3.5 VALU instructions per element: v_mul (scale and validity), v_cvt_pk_f16_f32, v_fma_mix_f32 (the residual
  // straight from the packed f16: hipcc's own code converts back to f32 and subtracts, 5.5 per element),
  // v_cvt_scalef32_pk_bf8_f32 (lo' = e5m2(lo * 2^12): the instruction divides by its scale), v_cvt_pk_bf8_f32.
  auto convert_store_x = [&](int xb) __attribute__((always_inline)) {
    unsigned char* xsm = smem + xb * XSET;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      if (PACKED) {   // records are stored as they are; a halo pixel outside the image is an all-zero record
        const int keep = ((xokm >> i) & 1) ? -1 : 0;
        if (xst16[i] != ~0u) {
          i32x4 hv;
          i32x2 lv, uv;
#pragma unroll
          for (int k = 0; k < 4; ++k) hv[k] = __builtin_bit_cast(int, xr[i][k]) & keep;
          lv[0] = __builtin_bit_cast(int, xr[i][4]) & keep; lv[1] = __builtin_bit_cast(int, xr[i][5]) & keep;
          uv[0] = __builtin_bit_cast(int, xr[i][6]) & keep; uv[1] = __builtin_bit_cast(int, xr[i][7]) & keep;
          *reinterpret_cast<i32x4*>(xsm + xst16[i]) = hv;
          if constexpr (MX) {   // the task's 16-byte piece of the pixel's FP6 block: image [part][piece][pixel] = [octet g][pixel]
            i32x4 pv;
            pv[0] = lv[0]; pv[1] = lv[1]; pv[2] = uv[0]; pv[3] = uv[1];
            *reinterpret_cast<i32x4*>(xsm + XF16 + xst16[i]) = pv;
          } else {
            *reinterpret_cast<i32x2*>(xsm + xst8[i]) = lv;
            *reinterpret_cast<i32x2*>(xsm + xst8[i] + 2 * NPX * 16) = uv;
          }
        }
        continue;
      }
      const float sm = ((xokm >> i) & 1) ? xs : 0.f;   // validity folded into the scale
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      float v[8], lo[8];
      int pk[4];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        v[j] = xr[i][j] * sm;
        v[j + 1] = xr[i][j + 1] * sm;
        int hh;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(v[j]), "v"(v[j + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hh), "v"(v[j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hh), "v"(v[j + 1]));
        pk[j >> 1] = hh;
      }
      s16x2 l0 = {0, 0}, l1 = {0, 0};
      l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[0], lo[1], 0x1p-12f, false);
      l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[2], lo[3], 0x1p-12f, true);
      l1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l1, lo[4], lo[5], 0x1p-12f, false);
      l1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l1, lo[6], lo[7], 0x1p-12f, true);
      int h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
      h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h0, true);
      int h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], 0, false);
      h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], h1, true);
      if (xst16[i] != ~0u) {
        i32x4 hv;
        hv[0] = pk[0]; hv[1] = pk[1]; hv[2] = pk[2]; hv[3] = pk[3];
        i32x2 lv, uv;
        lv[0] = __builtin_bit_cast(int, l0); lv[1] = __builtin_bit_cast(int, l1);
        uv[0] = h0; uv[1] = h1;
        *reinterpret_cast<i32x4*>(xsm + xst16[i]) = hv;
        *reinterpret_cast<i32x2*>(xsm + xst8[i]) = lv;                     // part 0: lo'
        *reinterpret_cast<i32x2*>(xsm + xst8[i] + 2 * NPX * 16) = uv;      // part 1: hi'
      }
    }
  };
  auto convert_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const bool ok = (xokm >> i) & 1;
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      float lo[8], v[8];
      int pk[4];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        v[j] = ok ? xr[i][j] * xs : 0.f;
        v[j + 1] = ok ? xr[i][j + 1] * xs : 0.f;
        f16x2 hh;
        hh[0] = (_Float16)v[j];
        hh[1] = (_Float16)v[j + 1];
        lo[j] = (v[j] - (float)hh[0]) * 4096.f;
        lo[j + 1] = (v[j + 1] - (float)hh[1]) * 4096.f;
        pk[j >> 1] = __builtin_bit_cast(int, hh);
      }
      int l0 = __builtin_amdgcn_cvt_pk_bf8_f32(lo[0], lo[1], 0, false);
      l0 = __builtin_amdgcn_cvt_pk_bf8_f32(lo[2], lo[3], l0, true);
      int l1 = __builtin_amdgcn_cvt_pk_bf8_f32(lo[4], lo[5], 0, false);
      l1 = __builtin_amdgcn_cvt_pk_bf8_f32(lo[6], lo[7], l1, true);
      int h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
      h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h0, true);
      int h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], 0, false);
      h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], h1, true);
#pragma unroll
      for (int k = 0; k < 4; ++k) xr[i][k] = __builtin_bit_cast(float, pk[k]);
      xr[i][4] = __builtin_bit_cast(float, l0);
      xr[i][5] = __builtin_bit_cast(float, l1);
      xr[i][6] = __builtin_bit_cast(float, h0);
      xr[i][7] = __builtin_bit_cast(float, h1);
    }
  };
  auto store_x = [&]() __attribute__((always_inline)) {
    // laundered lane index: the LDS addresses below are loop invariant, and hoisted out of the tile loop
    // they were spilled and reloaded one by one (scratch load + wait per ds_write) between the two barriers
    int xt = tid;
    asm volatile("" : "+v"(xt));
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int tk = xt + i * NTHR;
      if (tk < XTASKS) {
        const int g = tk / NPX, pix = tk - g * NPX;
        f32x4 hv;
        hv[0] = xr[i][0]; hv[1] = xr[i][1]; hv[2] = xr[i][2]; hv[3] = xr[i][3];
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 lv, uv;
        lv[0] = xr[i][4]; lv[1] = xr[i][5];
        uv[0] = xr[i][6]; uv[1] = xr[i][7];
        *reinterpret_cast<f32x4*>(smem + g * (NPX * 16) + pix * 16) = hv;
        unsigned char* p8 = smem + XF16 + ((g >> 1) * NPX + pix) * 16 + (g & 1) * 8;
        *reinterpret_cast<f32x2*>(p8) = lv;                      // part 0: lo'
        *reinterpret_cast<f32x2*>(p8 + 2 * NPX * 16) = uv;       // part 1: hi'
      }
    }
  };
  // weights of one stage (taps t0, t0+1 of a chunk) by LDS-DMA into ring half `half`: a wave moves 2 KiB
  // of each 16 KiB tap tile (2 instructions of 64 lanes x 16 B, LDS image = global image)
  auto dma_w = [&](const Tile& t, int chunk, int t0, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      int ky, kx;
      tap_of(t, chunk, t0 + tt, ky, kx);
      const unsigned char* src = a.wp + ((size_t)((ky * 3 + kx) * a.nchunk + chunk) * a.ncob + t.cob) * C8_WTILE;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int piece = (wave * 2 + u) * 1024;
        // MX: the slot's second half is the tile's FP6 part (waves 4..7 = pieces 8..15)
        const int spiece = (MX && wave >= 4) ? piece + (C8_WOFF6 - C8_WOFF8) : piece;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + spiece + lane * 16),
            (__attribute__((address_space(3))) void*)(smem + OFF_W + (half * 2 + tt) * WTAP + piece), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };

  // Per-lane LDS byte offsets of the fragment reads; everything else about a read is a compile-time
  // immediate (tap, row tile, k-step), so one VGPR per image serves every ds_read of the kernel.
  const unsigned xl16 = (unsigned)(h * (NPX * 16) + (wrow * XW + r) * 16);            // f16 halo image
  const unsigned xl8 = (unsigned)(XF16 + (h * 2) * (NPX * 16) + (wrow * XW + r) * 16); // fp8 halo images
  const unsigned wl16 = (unsigned)(OFF_W + h * (BM * 16) + (wco + r) * 16);            // f16 weight image
  const unsigned wl8 = (unsigned)(OFF_W + WF16 + (h * 2) * (BM * 16) + (wco + r) * 16); // fp8 weight images

  // One tap = 2 co-tiles x NI pixel-row tiles x (2 f16 MFMAs + 1 scaled fp8 MFMA), in two passes with their
  // own fragment sets.  A stage issues the reads of the NEXT pass before the MFMAs of the current one (scheduling
  // fences keep that order): left to the compiler each pass began with its 8 ds_read_b128 and an
  // `s_waitcnt lgkmcnt(0)`, four fully exposed LDS latencies per stage with both waves of a SIMD in step.
  struct Frag16 { f16x8 a[2][2]; f16x8 b[NI][2]; };
  struct Frag8 { i32x8 a[2]; i32x8 b[NI]; };
  auto load16 = [&](Frag16& f, unsigned wslot, unsigned xsel, int ky, int kx) __attribute__((always_inline)) {
    const unsigned tapoff = (unsigned)((ky * XW + kx) * 16) + xsel;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        f.a[i][s] = *reinterpret_cast<const f16x8*>(smem + (wl16 + wslot) + (2 * s) * (BM * 16) + i * 512);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        f.b[j][s] = *reinterpret_cast<const f16x8*>(smem + (xl16 + tapoff) + (2 * s) * (NPX * 16) + j * (XW * 16));
  };
  auto load8 = [&](Frag8& f, unsigned wslot, unsigned xsel, int ky, int kx) __attribute__((always_inline)) {
    const unsigned tapoff = (unsigned)((ky * XW + kx) * 16) + xsel;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* p = smem + (wl8 + wslot) + i * 512;
      const i32x4 q0 = *reinterpret_cast<const i32x4*>(p);
      const i32x4 q1 = *reinterpret_cast<const i32x4*>(p + BM * 16);
      f.a[i] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned char* p = smem + (xl8 + tapoff) + j * (XW * 16);
      const i32x4 q0 = *reinterpret_cast<const i32x4*>(p);
      const i32x4 q1 = *reinterpret_cast<const i32x4*>(p + NPX * 16);
      f.b[j] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  // MX: keeps each FP6 fragment one eight-register tuple (data = dwords 0..5, scale = dword 6 as sub-registers); see
  // conv_f16c8_w4.hip
  auto pin8 = [&](Frag8& f) __attribute__((always_inline)) {
    if constexpr (MX) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm("" : "+v"(f.a[i]));
#pragma unroll
      for (int j = 0; j < NI; ++j) asm("" : "+v"(f.b[j]));
    }
  };
  auto mma16 = [&](const Frag16& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
  };
  auto mma8 = [&](const Frag8& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (MX)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.a[i], f.b[j], acc[i][j], 2 /* A: e2m3 */,
                                                                      2 /* B: e2m3 */, 0, f.a[i][6], 0, f.b[j][6]);
        else
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.a[i], f.b[j], acc[i][j], 0 /* A: e4m3 */,
                                                                      1 /* B: e5m2 */, 0, sc_a, 0, sc_b);
      }
  };

  // GroupNorm partial sums of a stored tile (a.gn_sums): gslot[wave][sum|sumsq][batch (i, qq)][quad of lanes] is
  // written by the epilogue (each wave its own cells, so plain stores); after the next workgroup barrier
  // gn_flush() folds the 4 row waves x 8 quads behind each (channel quartet, value) and adds the result to
  // gn_sums[n][group] in double.  Every cell is rewritten by the next epilogue, several barriers later.
  float* const gslot = reinterpret_cast<float*>(smem + OFF_W + 4 * WTAP);
  auto gn_flush = [&](int tn, int tcb) __attribute__((always_inline)) {   // tcb: first (real) channel of the block
    const int out = tid >> 3, sub = tid & 7;
    const int v = out >> 5, wc = (out >> 4) & 1, b = (out >> 1) & 7, hh = out & 1;
    const int w = (sub >> 1) * 2 + wc;
    const float4 c4 = *reinterpret_cast<const float4*>(gslot + ((w * 2 + v) * 8 + b) * 16 + hh * 8 + (sub & 1) * 4);
    float t = (c4.x + c4.y) + (c4.z + c4.w);
    t += dpp_quad<0xB1>(t);
    t += dpp_quad<0x4E>(t);
    t += __shfl_xor(t, 4, 8);
    const int co = tcb + wc * 64 + (b >> 2) * 32 + (b & 3) * 8 + 4 * hh;
    if (sub == 0 && co < (MSHUFFLE ? a.creal : a.Cout))
      atomicAdd(a.gn_sums + ((size_t)tn * a.gn_groups + (co >> a.gn_shift)) * 2 + v, (double)t);
  };

  // Epilogue of one tile: as conv_bf16x3.hip (batched unconditional bias / residual loads, plain stores on
  // full tiles), plus the exact power-of-two un-scaling of a scaled input.
  float omax = 0.f;   // max |y| over everything this lane stores (a.out_amax)
  auto epilogue_tile = [&](const Tile& t) __attribute__((always_inline)) {
    const int ox = t.ox0 + r;
    const bool okx = ox < a.Wout;
    const int chans = MSHUFFLE ? a.creal : a.Cout;
    const int co_blk = (MSHUFFLE ? (t.cob - t.par1 * a.cpb) : t.cob) * BM;
    const int co_base = co_blk + wco + 4 * h;
    const int py = t.par1 >> 1, px = t.par1 & 1;
    const size_t out_plane = MSHUFFLE ? (size_t)a.Hout * a.Wout * 4 : (size_t)a.Hout * a.Wout;
    float* yn = a.y + (size_t)t.n * chans * out_plane;
    const float* rn = a.res ? a.res + (size_t)t.n * chans * out_plane : nullptr;
    unsigned rowoff[NI];
    bool rowok[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int oy = t.oy0 + wrow + j;
      rowok[j] = okx && oy < a.Hout;
      const int oyc = oy < a.Hout ? oy : a.Hout - 1, oxc = okx ? ox : a.Wout - 1;
      rowoff[j] = MSHUFFLE ? (unsigned)((2 * oyc + py) * (2 * a.Wout) + 2 * oxc + px)
                           : (unsigned)(oyc * a.Wout + oxc);
    }
    // Full tiles address y / residual / bias as  wave-uniform pointer (block + wave channel base, then the channel of
    // the (i, qq, k) batch entry: scalar ALU)  +  per-lane 32-bit byte offset (the lane half's 4 planes + its pixel).
    // (Per-lane 64-bit channel * plane products cost ~7 VALU instructions per channel, two of them quarter-rate
    // multiplies: the epilogue was 640 VALU instructions per wave tile, more than a 128-channel tile has MFMAs.)
    unsigned lo[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) lo[j] = (unsigned)(((size_t)(4 * h) * out_plane + rowoff[j]) * 4);
    const unsigned h16 = (unsigned)h * 16u;
    char* const ywb = reinterpret_cast<char*>(yn + (size_t)(co_blk + wco) * out_plane);
    const char* const rwb = reinterpret_cast<const char*>(rn + (size_t)(co_blk + wco) * out_plane);
    const char* const bwb = reinterpret_cast<const char*>(a.bias + (co_blk + wco));
    auto body = [&](auto full_t) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_t)::value;
      // residual requested half a wave tile at a time (4 x 4 x NI loads in flight, one wait): fetched batch by
      // batch, each of the eight batches exposed a full memory latency in front of its stores
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float rv[4][4][NI];
        if (!rn) {   // uniform: no residual -> zeros, so that the add below needs no select per value
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int j = 0; j < NI; ++j) rv[qq][k][j] = 0.f;
        } else {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if constexpr (FULL) {
                const char* rp = rwb + (size_t)(i * 32 + 8 * qq + k) * out_plane * 4;   // uniform
#pragma unroll
                for (int j = 0; j < NI; ++j) rv[qq][k][j] = *reinterpret_cast<const float*>(rp + lo[j]);
              } else {
                const int co = co_base + i * 32 + k + 8 * qq;
                const int cc = co < chans ? co : chans - 1;
#pragma unroll
                for (int j = 0; j < NI; ++j) rv[qq][k][j] = rn[(size_t)cc * out_plane + rowoff[j]];
              }
            }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          if ((EOVAE_C8_ABL & 1) && (i || qq)) {   // timing-only: keep the values alive, store 1/8 of them
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
              for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(acc[i][j][4 * qq + k]));
            continue;
          }
          float bias4[4];
          int coc[4];
          bool cok[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int co = co_base + i * 32 + k + 8 * qq;
            cok[k] = FULL || co < chans;
            coc[k] = cok[k] ? co : chans - 1;
            if constexpr (FULL)
              bias4[k] = a.bias ? *reinterpret_cast<const float*>(bwb + (i * 32 + 8 * qq + k) * 4 + h16) : 0.f;
            else
              bias4[k] = a.bias ? a.bias[coc[k]] : 0.f;
          }
          float v[4][NI];
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              v[k][j] = acc[i][j][4 * qq + k] * xinv + bias4[k] + rv[qq][k][j];
              if (!FULL) asm volatile("" : "+v"(v[k][j]));
              omax = __builtin_fmaxf(omax, (FULL || (cok[k] && rowok[j])) ? __builtin_fabsf(v[k][j]) : 0.f);
            }
          if (a.gn_sums) {
            // statistics of the following GroupNorm from the values being stored: the 4 channels of this batch
            // lie in one group
            float sj[NI], qj[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) { sj[j] = 0.f; qj[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int j = 0; j < NI; ++j) {
                const float u = (FULL || (cok[k] && rowok[j])) ? v[k][j] : 0.f;
                sj[j] += u;
                qj[j] = __builtin_fmaf(u, u, qj[j]);
              }
            float sv = sj[0], qv = qj[0];
#pragma unroll
            for (int j = 1; j < NI; ++j) { sv += sj[j]; qv += qj[j]; }
            // quad sums by DPP, then one plain store per quad into this wave's own cells (LDS float atomics cost
            // +9 % kernel time at 128 channels; plain stores are free)
            sv += dpp_quad<0xB1>(sv); qv += dpp_quad<0xB1>(qv);      // lane ^ 1
            sv += dpp_quad<0x4E>(sv); qv += dpp_quad<0x4E>(qv);      // lane ^ 2
            if ((lane & 3) == 0) {
              float* cell = gslot + (wave * 16 + i * 4 + qq) * 16 + (lane >> 2);
              cell[0] = sv;
              cell[8 * 16] = qv;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if constexpr (FULL) {
              char* yp = ywb + (size_t)(i * 32 + 8 * qq + k) * out_plane * 4;   // uniform
#pragma unroll
              for (int j = 0; j < NI; ++j) *reinterpret_cast<float*>(yp + lo[j]) = v[k][j];
            } else {
#pragma unroll
              for (int j = 0; j < NI; ++j)
                if (cok[k] && rowok[j]) yn[(size_t)coc[k] * out_plane + rowoff[j]] = v[k][j];
            }
          }
        }
      }
    };
    const bool full_tile = t.oy0 + TH <= a.Hout && t.ox0 + TW <= a.Wout && co_blk + BM <= chans;
    if (full_tile) body(std::true_type{}); else body(std::false_type{});
  };

  // Workgroup barrier that publishes LDS writes and the stage's landed weight DMAs: the wave's INFLIGHT
  // youngest vector-memory operations (halo loads issued after the DMAs) are left in flight.
  auto stage_barrier = [&](auto inflight) __attribute__((always_inline)) {
    constexpr int INFLIGHT = decltype(inflight)::value;
    static_assert(INFLIGHT >= 0 && INFLIGHT < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using IC0 = std::integral_constant<int, 0>;

  // ------------------------------------------------------------------ pipeline
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  Tile cur = decode(tile);
  setup_x(cur);
  dma_w(cur, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < (PACKED ? 2 : 8); ++j) load_x_group(cur, 0, j);
  if (XDB) {
    convert_store_x(0);
  } else {
    convert_x();
    store_x();
  }
  zero_acc();
  stage_barrier(IC0{});
  int xbuf = 0;
  bool gn_pend = false;   // a stored tile's GroupNorm partials wait in LDS for the next barrier
  int gn_n = 0, gn_cb = 0;   // halo buffer of the chunk being multiplied (XDB)

  int q = 0;  // global stage counter: ring half (q & 1) holds the stage being multiplied
#if EOVAE_C8_STAMP
  int rec = 0;
#endif
  for (; tile < a.ntiles; tile += gridDim.x) {
    const bool has_next_tile = tile + (int)gridDim.x < a.ntiles;
    Tile nxt = cur;
    if (has_next_tile) nxt = decode(tile + gridDim.x);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool last_chunk = chunk + 1 == a.nchunk;
      const bool have_nx = !last_chunk || has_next_tile;       // is there a chunk to stage behind this one
      if (last_chunk && has_next_tile) setup_x(nxt);            // this tile's offsets are not needed any more
      const int nt = SUBPIX == 0 ? 9 : ntaps_of(cur, chunk), ns = (nt + 1) >> 1;
      const int cs = ns >= 3 ? ns - 2 : ns - 1;                 // stage that converts the staged chunk
      const Tile& xt_tile = last_chunk ? nxt : cur;             // where the staged chunk comes from
      const int xc0 = last_chunk ? 0 : (chunk + 1) * KC;
#pragma unroll
      for (int s = 0; s < (SUBPIX == 0 ? 5 : 2); ++s) {
        if (SUBPIX != 0 && s >= ns) break;
        const bool last_stage = s == ns - 1;
        C8_STAMP(0);
        // weights of the following stage (this chunk, next chunk, or the next tile's first chunk)
        // (waves 0-3 issue them before their first tap, waves 4-7 behind it: eight waves issuing ~65-cycle DMA
        // instructions at once left the matrix pipes idle at the top of every stage)
        auto stage_dma = [&]() __attribute__((always_inline)) {
          if (EOVAE_C8_ABL & 8) return;
          if (!last_stage) dma_w(cur, chunk, 2 * s + 2, (q + 1) & 1);
          else if (!last_chunk) dma_w(cur, chunk + 1, 0, (q + 1) & 1);
          else if (has_next_tile) dma_w(nxt, 0, 0, (q + 1) & 1);
        };
        if (wave < 4) stage_dma();
        // waves 0-3 convert BEFORE their MFMAs, waves 4-7 after theirs: on every SIMD one wave converts
        // (VALU) while its partner multiplies
        if (SUBPIX != 0 && ns == 1 && have_nx) {   // single-stage chunks (stride-2 forms): loaded and converted at once
#pragma unroll
          for (int j = 0; j < (PACKED ? 2 : 8); ++j) load_x_group(xt_tile, xc0, j);
        }
        if (!(EOVAE_C8_ABL & 4)) if (s == cs && have_nx && wave < 4) {
          if (XDB) convert_store_x(xbuf ^ 1); else convert_x();
        }
        C8_STAMP(1);
        const unsigned wslot = (unsigned)((q & 1) * 2 * WTAP);
        {
          const unsigned xsel = XDB ? (unsigned)(xbuf * XSET) : 0u;
          const bool two = 2 * s + 1 < nt;   // uniform
          int ky0, kx0, ky1, kx1;
          tap_of(cur, chunk, 2 * s, ky0, kx0);
          tap_of(cur, chunk, 2 * s + 1, ky1, kx1);
          // halo groups of the staged chunk: two behind each of the taps 0..3 (plain path); all of them in
          // stage 0 when the conversion is in stage 1
          auto stage_loads = [&](int tt) __attribute__((always_inline)) {
            if ((EOVAE_C8_ABL & 2) || !have_nx) return;
            const int t = 2 * s + tt;
            if (PACKED) {
              if (t < 2) load_x_group(xt_tile, xc0, t);      // two 16-byte groups, behind taps 0 and 1
            } else if (SUBPIX == 0) {
              if (t < 4) { load_x_group(xt_tile, xc0, 2 * t); load_x_group(xt_tile, xc0, 2 * t + 1); }
            } else if (ns == 2 && s == 0) {
#pragma unroll
              for (int j = 0; j < 4; ++j) load_x_group(xt_tile, xc0, 4 * tt + j);
            }
          };
          if (!(EOVAE_C8_ABL & 16)) {
            Frag16 f16a;
            Frag8 f8a;
            load16(f16a, wslot, xsel, ky0, kx0);
            load8(f8a, wslot, xsel, ky0, kx0);
            __builtin_amdgcn_sched_barrier(0);
            mma16(f16a);
            __builtin_amdgcn_sched_barrier(0);
            if (two) load16(f16a, wslot + WTAP, xsel, ky1, kx1);
            __builtin_amdgcn_sched_barrier(0);
            pin8(f8a);
            mma8(f8a);
            __builtin_amdgcn_sched_barrier(0);
            if (wave >= 4) stage_dma();
            stage_loads(0);
            if (two) {
              load8(f8a, wslot + WTAP, xsel, ky1, kx1);
              __builtin_amdgcn_sched_barrier(0);
              mma16(f16a);
              __builtin_amdgcn_sched_barrier(0);
              pin8(f8a);
              mma8(f8a);
              __builtin_amdgcn_sched_barrier(0);
              stage_loads(1);
            }
          } else {
            if (wave >= 4) stage_dma();
            stage_loads(0);
            if (two) stage_loads(1);
          }
        }
        C8_STAMP(2);
        if (!(EOVAE_C8_ABL & 4)) if (s == cs && have_nx && wave >= 4) {
          if (XDB) convert_store_x(xbuf ^ 1); else convert_x();
        }
        C8_STAMP(3);
        if (!XDB && last_stage && have_nx) {
          stage_barrier(IC0{});     // every wave is done with this chunk's halo tile
          C8_STAMP(4);
          store_x();
          C8_STAMP(5);
        }
        C8_STAMP(6);
        // plain path: stages 0 and 1 each issued 4 groups behind their DMAs
        if (PACKED && s == 0 && !(EOVAE_C8_ABL & 2)) {          // stage 0 issued both 16-byte groups behind its DMAs
          if (have_nx) stage_barrier(std::integral_constant<int, 2 * LPG>{}); else stage_barrier(IC0{});
        } else if (!PACKED && SUBPIX == 0 && s < 2 && !(EOVAE_C8_ABL & 2)) {
          if (have_nx) stage_barrier(std::integral_constant<int, 4 * LPG>{}); else stage_barrier(IC0{});
        } else {
          stage_barrier(IC0{});
        }
        C8_STAMP(7);
        if (gn_pend) {   // uniform; first barrier behind the epilogue that filled the cells
          gn_flush(gn_n, gn_cb);
          gn_pend = false;
        }
#if EOVAE_C8_STAMP
        ++rec;
#endif
        ++q;
      }
      if (have_nx) xbuf ^= 1;
    }
#if EOVAE_C8_STAMP
    C8_STAMP(0);
#endif
    epilogue_tile(cur);
    zero_acc();
    if (a.gn_sums) { gn_pend = true; gn_n = cur.n; gn_cb = (MSHUFFLE ? cur.cob - cur.par1 * a.cpb : cur.cob) * BM; }
#if EOVAE_C8_STAMP
    C8_STAMP(7);
    if (lane == 0 && blockIdx.x < C8_NBLK && rec < C8_MAXREC)   // 100 MHz wall clock next to the shader clock
      c8_stamps[((blockIdx.x * 8 + wave) * C8_MAXREC + rec) * C8_NPT + 1] = __builtin_amdgcn_s_memrealtime();
    ++rec;
#endif
    cur = nxt;
  }
  if (gn_pend) {
    stage_barrier(IC0{});
    gn_flush(gn_n, gn_cb);
  }
  if (a.out_amax) {   // uniform
    const float mx = wave_max(omax);
    if (lane == 0 && mx > 0.f && mx < 3.0e38f) atomicMax(a.out_amax, __builtin_bit_cast(unsigned, mx));   // NaN / inf never land
  }
}


// ---------------------------------------------------------------------------------------
// weight gradient with f16c8 operands:  dWp[tap][co][ci] += sum_{n,pix} dY[n][co][pix] * Xv[n][ci][pix @ tap]
// Same tiling, staging pipeline and sub-pixel / space-to-depth forms as wgrad3x3_bf16_kernel (conv_bf16x3.hip:
// block = 128 co x 64 ci x 9 taps, K = pixels in tiles of 2 rows x 32, both operands staged
// channel-contiguous and gathered by transposing LDS reads); what differs is the arithmetic.  Per output row of
// 32 pixels and tap: 2 x v_mfma_f32_32x32x16_f16 (dy_hi * x_hi) + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 whose
// K = 64 is [dy_lo' | dy_hi'] . [x_hi' | x_lo'] (all e5m2; the lo' images carry 2^12, undone by the A block
// scale) = 4 MFMA units where split-bf16 needs 6.  dy is pre-scaled by the power of two derived from its
// amax (same value the data-gradient call uses) and the accumulators are un-scaled on the way out.
// ---------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x8 tr_frag16(const unsigned char* p0, const unsigned char* p1) {
  using lds4 = __attribute__((address_space(3))) s16x4;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p1));
  return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ int wg_src_index(int mode, int oy0, int ox0, int r, int c, int Hin, int Win) {
  const int vy = oy0 + r - 1, vx = ox0 + c - 1;
  if (mode == MODE_NORMAL)
    return ((unsigned)vy < (unsigned)Hin && (unsigned)vx < (unsigned)Win) ? vy * Win + vx : -1;
  return ((unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win)) ? (vy >> 1) * Win + (vx >> 1) : -1;
}

#ifndef EOVAE_WG_LEAN
#define EOVAE_WG_LEAN 1
#endif
#ifndef EOVAE_WG_B8DB
#define EOVAE_WG_B8DB 1
#endif
#ifndef EOVAE_WGRAD_BLOCKS
#define EOVAE_WGRAD_BLOCKS 256
#endif


// XPACKED: x is the f16c8 operand-record tensor GroupNorm-apply wrote (32 B per pixel and channel octet): an X task is
// two 16-byte loads and three LDS stores, no conversion (plain layers only: SUBPIX 0, mode NORMAL, Cin % 64 == 0).
template <int SUBPIX, bool XPACKED>
__global__ __launch_bounds__(512, 2) void wgrad3x3_f16c8_kernel(WgCArgs a) {
  constexpr int NT = SUBPIX ? 4 : 9;  // accumulated taps: all 9, or the parity's 2x2 support
  constexpr int PW = 32, XW = 40, XWV = 34, NXP = 4 * XW;  // 2 output rows + halo; LDS row pitch 40
                                                          // pixels (34 valid) keeps every window start = kx (mod 8)
  constexpr int DPART = 4 * 64 * 64;                      // f16 [4 co-tiles][64 pix][32 co]
  constexpr int D8IMG = 4 * 64 * 32, D8 = 2 * D8IMG;      // e5m2 [dyl' | dyh'][4 co-tiles][64 pix][32 co]
  constexpr int XPART = 2 * NXP * 64;                     // f16 [2 ci-tiles][160 pix][32 ci]
  constexpr int X8IMG = 2 * NXP * 32, X8 = 2 * X8IMG;     // e5m2 [xh' | xl'][2 ci-tiles][160 pix][32 ci]
  constexpr int OFF_D8 = DPART, OFF_X = DPART + D8, OFF_X8 = OFF_X + XPART;
  constexpr int BUF = OFF_X8 + X8;                        // 72 KiB, two of them
  constexpr int XQ = NXP * 8 / 4;                         // X tasks per k-step (272)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  // Wave tile: TWO co-tiles x one ci-tile x HALF of the taps (wave >> 2 selects taps 0..NTA-1 or NTA..NT-1), i.e.
  // every X fragment read from LDS feeds two MFMAs.  (One co-tile x all nine taps per wave -- the r1 tiling -- read
  // 40 KB of fragments per 27 MFMAs: 2560 LDS clocks against 2304 MFMA clocks per half tile; now
  // 26 KB.)  Waves w and w + 4 share a SIMD: one of each tap group, so the 5 / 4 split balances per SIMD.
  const int tg = __builtin_amdgcn_readfirstlane(wave >> 2), ctp = wave & 1, cih = (wave >> 1) & 1;
  const int ct = 2 * ctp;   // first of the wave's two co-tiles
  // XCD-aware block -> (slab, tile stride slot) map.  Workgroups are dealt round-robin to the 8 XCDs, each with
  // its own L2.  The slabs (co-block x ci-block) that walk the SAME pixel tiles -- ncib of them share every dY
  // tile, ncob every X tile -- are given consecutive slots of ONE XCD, so a shared tile is fetched into one L2
  // once instead of into up to 8 of them (r1: wgrad L2-miss fetch 1.8x .. 3.4x the algorithmic bytes).
  int slab, ysub;
  {
    const int nslab = a.ncib * a.ncob_;
    const int b = blockIdx.x;
    if (a.xcd_remap) {
      const int xcd = b & 7, q = b >> 3;
      slab = q % nslab;
      ysub = (q / nslab) * 8 + xcd;
    } else {
      slab = b % nslab;
      ysub = b / nslab;
    }
  }
  const int cib = slab % a.ncib, cob = slab / a.ncib;
  const int par = SUBPIX == 1 ? cob / a.cpb : (SUBPIX == 2 ? (cib * 64) / a.creal : 0);
  const int py = par >> 1, px = par & 1;
  const int dych = SUBPIX == 1 ? a.creal : a.Cout;   // channels / plane of the stored dy tensor
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const size_t out_plane = SUBPIX == 1 ? (size_t)a.Hout * a.Wout * 4 : (size_t)a.Hout * a.Wout;
  // per-lane byte offsets inside a swizzled [pixel][32 ch] image for the two transposed 4x16 block
  // reads of a fragment (pixel rows lpart and lpart+4 of the 16-pixel k-step, 8-byte chunk lchunk)
  const int lpart = 8 * (g >> 1) + lq, lchunk = 4 * (g & 1) + lp;
  // dY image: chunk ^ ((row>>1)&7); row = 16*ks + lpart(+4) and 16*ks never reaches those bits
  const int sa0 = lpart * 64 + ((lchunk ^ ((lpart >> 1) & 7)) * 8);
  const int sa1 = (lpart + 4) * 64 + ((lchunk ^ (((lpart + 4) >> 1) & 7)) * 8);
  // X image: 16-byte unit ^ ((pix>>1)&3); pix = base + lpart(+4), so one offset per (base & 7)
  auto sx_of = [&](int res) {
    return lpart * 64 + (((lchunk >> 1) ^ (((res + lpart) >> 1) & 3)) * 16) + (lchunk & 1) * 8;
  };
  // window start = kx (mod 8): named scalars, never an indexable array (that would go to scratch)
  const int sx0 = sx_of(0), sx1 = sx_of(1), sx2 = sx_of(2), sx4 = sx_of(4), sx5 = sx_of(5), sx6 = sx_of(6);
  // fp8 fragments by ds_read_b64_tr_b8 from [pixel][32 channel bytes] images.  In a 16-lane group lane 2j+p
  // points at 8 bytes; destination lane 8p+c receives byte c of the eight p-rows (tools/probe_tr8.hip).  So
  // source lane (j, p) of group G points at pixel P+j, channel octet 2(G&1)+p of image G>>1 (= the MFMA
  // K-block h of the destination lanes), and destination lane (h, r) gets channel r at pixels P..P+7.
  const int l8 = (li >> 1) * 32 + ((g & 1) * 2 + (li & 1)) * 8;
  const int a8off = OFF_D8 + (g >> 1) * D8IMG + ct * (64 * 32) + l8;      // dyl' (h = 0) | dyh' (h = 1)
  const int b8off = OFF_X8 + (g >> 1) * X8IMG + cih * (NXP * 32) + l8;    // xh'  (h = 0) | xl'  (h = 1)
  // dy pre-scale: the power of two that puts its amax at 2^13..2^14 (gradients sit far below f16's range)
  float dys = 1.f, dyinv = 1.f;
  if (a.dyamax) {
    const unsigned bits = *a.dyamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      dys = pow2f(13 - e);
      dyinv = pow2f(e - 13);
    }
  }
  const int sc_a = 127 - 12, sc_b = 127;   // E8M0: both K blocks carry one operand scaled by 2^12
  // x pre-scale (fp32 x only: un-normalised activations -- the residual stream in front of a Downsample / Upsample
  // layer -- are not bounded like GroupNorm outputs): the power of two that puts max|x| at 2^13..2^14, undone with dy's
  float xs = 1.f;
  if (!XPACKED && a.xamax) {
    const unsigned bits = *a.xamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      xs = pow2f(13 - e);
      dyinv *= pow2f(e - 13);
    }
  }

  constexpr int NTA = (NT + 1) / 2;   // taps of group 0 (5 of 9, 2 of 4); group 1 has NT - NTA
  f32x16 acc[NTA][2];                // [tap of the wave's group][co-tile]
#pragma unroll
  for (int t = 0; t < NTA; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

  // staging registers: HALF a tile (one output row of 32 pixels + its share of the halo) is issued
  // before two k-steps (54 MFMAs per wave, ~1.7 us per SIMD pair) and committed to the other LDS
  // buffer after them.  Timing-only builds (EOVAE_WABL) show staging and MFMA time adding up
  // (0.26-0.35 ms + 0.6-0.7 ms = 1.0-1.2 ms); doubling the prefetch distance from one k-step to two
  // measured +0..3 % on the same box, so load LATENCY is not what serialises them.
  // dY: 2 tasks (4 channels x 1 pixel) per thread; X halo: 640 tasks (8 channels x 1 pixel).
  // Task maps are powers of two so that no per-tile integer division is left in the loop (the r1 maps cost ~400
  // index-math VALU instructions per thread and tile; SQ counters: VALU 2 x 23 % and MFMA 38 % of a SIMD's cycles,
  // adding up):
  //   dY  : thread -> (pixel tid & 15 of x-half i, channel quartet tid >> 4); 2 tasks per half tile
  //   halo: half hf stages ci-tile hf = octets 4 hf .. 4 hf + 3 over the 160 halo pixels (4 rows x 40, 34 valid):
  //         round 0 (all threads): octet tid >> 7, pixel tid & 127; round 1 (tid < 128): octet tid >> 5, pixel
  //         128 + (tid & 31), i.e. halo row 3, columns 8..39
  // and everything that depends on the tile only is wave-uniform (scalar base pointers, 32-bit lane offsets).
  float dyr[2][4], xr[2][8];
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};  // bias-gradient partials of this thread's 4 channels
  const bool do_db = a.db != nullptr && cib == 0;   // sub-pixel form: the four parity blocks of a channel each add their pixels
  bool okd0 = true, okd1 = true, okx0 = true, okx1 = true;  // validity of the staged tasks (lane masks in SGPRs)
  const int pixl = tid & 15, quad = tid >> 4;
  auto xgeom = [&](int i, int& octr, int& pix, int& rr, int& cc, bool& active) __attribute__((always_inline)) {
    int tid = threadIdx.x;   // laundered: the handful of shifts below is recomputed per use instead of living in
    asm volatile("" : "+v"(tid));   // ~10 registers across the MFMA phases (which need them for fragment prefetch)
    if (i == 0) {
      octr = tid >> 7;
      pix = tid & 127;
      rr = (pix * 205) >> 13;          // pix / 40 for pix < 128
      cc = pix - 40 * rr;
      active = true;
    } else {
      octr = (tid >> 5) & 3;
      pix = 128 + (tid & 31);
      rr = 3;
      cc = 8 + (tid & 31);
      active = tid < 128;
    }
  };
  struct TileC { int n, oy0, ox0; };
  auto decode = [&](int tile) __attribute__((always_inline)) {
    TileC t;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    t.n = tile / a.tiles_y;
    t.oy0 = ty * 2;
    t.ox0 = tx * PW;
    return t;
  };
  auto issue = [&](int tile, int hf) __attribute__((always_inline)) {
    const TileC tc = decode(tile);
    const int n = tc.n, oy0 = tc.oy0, ox0 = tc.ox0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = oy0 + hf, ox = ox0 + 16 * i + pixl;
      const bool ok = oy < a.Hout && ox < a.Wout;
      const size_t pixoff = SUBPIX == 1 ? (size_t)(2 * oy + py) * (2 * a.Wout) + 2 * ox + px
                                        : (size_t)oy * a.Wout + ox;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = cob * 128 + quad * 4 + j;
        dyr[i][j] = (ok && co < a.Cout)
                        ? a.dy[((size_t)n * dych + (co - (SUBPIX == 1 ? par * a.creal : 0))) * out_plane + pixoff]
                        : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int octr, pix, rr, cc;
      bool active;
      xgeom(i, octr, pix, rr, cc, active);
      if (active) {
        const int oct = hf * 4 + octr;
        int si = -1;
        if (cc < XWV) {
          if (SUBPIX == 2) {  // space-to-depth gather: low-res halo pixel (vy,vx) -> x[2vy+py][2vx+px]
            const int vy = oy0 + rr - 1, vx = ox0 + cc - 1;
            if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout)
              si = (2 * vy + py) * a.Win + 2 * vx + px;
          } else {
            si = wg_src_index(a.mode, oy0, ox0, rr, cc, a.Hin, a.Win);
          }
        }
        const int xch = SUBPIX == 2 ? a.creal : a.Cin;
        if (XPACKED) {
          const char* rec = reinterpret_cast<const char*>(a.x) +
                            (((size_t)n * (a.Cin / 8) + cib * 8 + oct) * in_plane + (si >= 0 ? si : 0)) * 32;
          const f32x4 q0 = *reinterpret_cast<const f32x4*>(rec), q1 = *reinterpret_cast<const f32x4*>(rec + 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xr[i][j] = si >= 0 ? q0[j] : 0.f;
            xr[i][4 + j] = si >= 0 ? q1[j] : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ci = cib * 64 + oct * 8 + j;
            xr[i][j] = (si >= 0 && ci < a.Cin)
                           ? a.x[((size_t)n * xch + (ci - (SUBPIX == 2 ? par * a.creal : 0))) * in_plane + si]
                           : 0.f;
          }
        }
      }
    }
    okd0 = okd1 = okx0 = okx1 = true;  // zeros already selected
  };
  // Fast path (every channel of the block's 128 co x 64 ci slab exists and the layer is not the fused-upsample
  // form -- all layers of the model): UNCONDITIONAL loads, validity kept as lane masks and applied when the
  // registers are converted (predicated loads compile to an exec-mask branch per load plus vmcnt(0) drains).
  // Address = wave-uniform pointer (image, tile origin, channel j / ci-tile: scalar ALU) + a 32-bit lane offset
  // that is a per-thread constant for valid lanes and the tile-origin pixel for the others.
  const bool fullc = cob * 128 + 128 <= a.Cout && cib * 64 + 64 <= a.Cin && a.mode == MODE_NORMAL;
  const float* dyb = a.dy + (size_t)(cob * 128 - (SUBPIX == 1 ? par * a.creal : 0)) * out_plane;
  const float* xb = a.x + (size_t)(cib * 64 - (SUBPIX == 2 ? par * a.creal : 0)) * in_plane;
  constexpr int PST = SUBPIX == 1 ? 2 : 1;        // dy pixel stride (sub-pixel form: every other high-res pixel)
  const unsigned dq = (unsigned)quad * 4u * (unsigned)out_plane;
  auto issue_fast = [&](int tile, int hf) __attribute__((always_inline)) {
    const TileC tc = decode(tile);
    const int n = tc.n, oy0 = tc.oy0, ox0 = tc.ox0;
    const int xch = SUBPIX == 2 ? a.creal : a.Cin;
    const float* dyn = dyb + (size_t)n * dych * out_plane;   // wave-uniform bases
    const int oy = oy0 + hf;
    const bool rowok = oy < a.Hout;
    const unsigned rowoff = SUBPIX == 1 ? (unsigned)((2 * oy + py) * (2 * a.Wout) + px) : (unsigned)(oy * a.Wout);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oxb = ox0 + 16 * i;
      const bool cok = rowok && oxb < a.Wout;                              // uniform
      const char* sb = reinterpret_cast<const char*>(dyn + (cok ? rowoff + (unsigned)(PST * oxb) : 0u));  // off-grid: pixel 0
      const bool ok = cok && pixl < a.Wout - oxb;
      const unsigned vo = (dq + (ok ? (unsigned)(PST * pixl) : 0u)) * 4u;   // BYTE offset: scalar base + 32-bit lane offset
#pragma unroll
      for (int j = 0; j < 4; ++j) dyr[i][j] = *reinterpret_cast<const float*>(sb + (size_t)j * out_plane * 4 + vo);
      if (i == 0) okd0 = ok; else okd1 = ok;
    }
    // halo: pixel (rr, cc) of the tile's halo sits at origin + rr * ys + cc * xs of its plane
    const int ys = SUBPIX == 2 ? 2 * a.Win : a.Win, xs = SUBPIX == 2 ? 2 : 1;
    const int lh = SUBPIX == 2 ? a.Hout : a.Hin, lw = SUBPIX == 2 ? a.Wout : a.Win;
    const long origin = SUBPIX == 2 ? (long)(2 * (oy0 - 1) + py) * a.Win + 2 * (ox0 - 1) + px
                                    : (long)(oy0 - 1) * a.Win + (ox0 - 1);
    const unsigned lin_safe = (unsigned)(ys + xs);           // halo (1, 1) = the tile's first pixel: always inside
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int octr, pix, rr, cc;
      bool active;
      xgeom(i, octr, pix, rr, cc, active);
      const bool ok = active && cc < XWV && (unsigned)(oy0 + rr - 1) < (unsigned)lh &&
                      (unsigned)(ox0 + cc - 1) < (unsigned)lw;
      const unsigned lin = ok ? (unsigned)(rr * ys + cc * xs) : lin_safe;
      if (i == 0) okx0 = ok; else okx1 = ok;
      if (i == 0 || tid < 128) {   // second round: 128 of the 512 threads
        if (XPACKED) {
          const char* recb = reinterpret_cast<const char*>(a.x) +
                             (((long)n * (a.Cin / 8) + cib * 8 + hf * 4) * (long)in_plane + origin) * 32;
          const unsigned vo = ((unsigned)octr * (unsigned)in_plane + lin) * 32u;
          const f32x4 q0 = *reinterpret_cast<const f32x4*>(recb + vo);
          const f32x4 q1 = *reinterpret_cast<const f32x4*>(recb + vo + 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) { xr[i][j] = q0[j]; xr[i][4 + j] = q1[j]; }
        } else {
          const char* xn = reinterpret_cast<const char*>(xb + ((long)n * xch + hf * 32) * (long)in_plane + origin);
          const unsigned vo = ((unsigned)octr * 8u * (unsigned)in_plane + lin) * 4u;
#pragma unroll
          for (int j = 0; j < 8; ++j) xr[i][j] = *reinterpret_cast<const float*>(xn + (size_t)j * in_plane * 4 + vo);
        }
      }
    }
  };
  // per-thread LDS offsets of the staged tasks (k-step / ci-tile parts are added per half)
  const int dc0 = (quad >> 3) * 4096 + pixl * 64 + (((quad & 7) ^ ((pixl >> 1) & 7)) * 8);
  const int dc1 = OFF_D8 + (quad >> 3) * (64 * 32) + pixl * 32 + (quad & 7) * 4;
  auto commit = [&](int buf, int hf) __attribute__((always_inline)) {
    unsigned char* base = smem + buf * BUF;
    // zero the invalid tasks -- behind a wave-uniform test: interior tiles have none
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool bd = !(i == 0 ? okd0 : okd1), bx = !(i == 0 ? okx0 : okx1) && (i == 0 || tid < 128);
      if (__builtin_amdgcn_ballot_w64(bd) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dyr[i][j] = bd ? 0.f : dyr[i][j];
      }
      if (__builtin_amdgcn_ballot_w64(bx) != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = bx ? 0.f : xr[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      float v[4], lo[4];
      int hp[2];   // 4 x f16, converted and subtracted with 3.5 VALU instructions per element (see convert_store_x)
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        v[j] = dyr[i][j] * dys;
        v[j + 1] = dyr[i][j + 1] * dys;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp[j >> 1]) : "v"(v[j]), "v"(v[j + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp[j >> 1]), "v"(v[j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp[j >> 1]), "v"(v[j + 1]));
      }
      i32x2 hi;
      hi[0] = hp[0]; hi[1] = hp[1];
      s16x2 l8v = {0, 0};
      l8v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l8v, lo[0], lo[1], 0x1p-12f, false);
      l8v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l8v, lo[2], lo[3], 0x1p-12f, true);
      int h8v = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
      h8v = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h8v, true);
      const int ks = 2 * hf + i;
      *reinterpret_cast<i32x2*>(base + dc0 + ks * 1024) = hi;
      *reinterpret_cast<int*>(base + dc1 + ks * 512) = __builtin_bit_cast(int, l8v);   // image 0: dyl'
      *reinterpret_cast<int*>(base + dc1 + ks * 512 + D8IMG) = h8v;                    // image 1: dyh'
#pragma unroll
      for (int j = 0; j < 4; ++j) dbs[j] += dyr[i][j];   // always (a select on do_db cost 8 more instructions); read only if do_db
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int octr, pix, rr, cc;
      bool active;
      xgeom(i, octr, pix, rr, cc, active);
      if (active) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        i32x4 hi;
        int h0, h1;
        s16x2 l0 = {0, 0}, l1 = {0, 0};
        if (XPACKED) {   // the record already holds the three payloads
#pragma unroll
          for (int k = 0; k < 4; ++k) hi[k] = __builtin_bit_cast(int, xr[i][k]);
          l0 = __builtin_bit_cast(s16x2, xr[i][4]);
          l1 = __builtin_bit_cast(s16x2, xr[i][5]);
          h0 = __builtin_bit_cast(int, xr[i][6]);
          h1 = __builtin_bit_cast(int, xr[i][7]);
        } else {
          float lo[8];
          if (xs != 1.f) {   // uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[i][j] *= xs;
          }
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            int hp;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(xr[i][j]), "v"(xr[i][j + 1]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp), "v"(xr[i][j]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp), "v"(xr[i][j + 1]));
            hi[j >> 1] = hp;
          }
          l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[0], lo[1], 0x1p-12f, false);
          l0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l0, lo[2], lo[3], 0x1p-12f, true);
          l1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l1, lo[4], lo[5], 0x1p-12f, false);
          l1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(l1, lo[6], lo[7], 0x1p-12f, true);
          h0 = __builtin_amdgcn_cvt_pk_bf8_f32(xr[i][0], xr[i][1], 0, false);
          h0 = __builtin_amdgcn_cvt_pk_bf8_f32(xr[i][2], xr[i][3], h0, true);
          h1 = __builtin_amdgcn_cvt_pk_bf8_f32(xr[i][4], xr[i][5], 0, false);
          h1 = __builtin_amdgcn_cvt_pk_bf8_f32(xr[i][6], xr[i][7], h1, true);
        }
        // ci-tile hf: octet octr of it; 16-byte unit ^ ((pix >> 1) & 3)
        const int off = OFF_X + hf * (NXP * 64) + pix * 64 + ((octr ^ ((pix >> 1) & 3)) * 16);
        *reinterpret_cast<i32x4*>(base + off) = hi;
        const int off8 = OFF_X8 + hf * (NXP * 32) + pix * 32 + octr * 8;
        i32x2 hv, lv;
        hv[0] = h0; hv[1] = h1;
        lv[0] = __builtin_bit_cast(int, l0); lv[1] = __builtin_bit_cast(int, l1);
        *reinterpret_cast<i32x2*>(base + off8) = hv;            // image 0: xh'
        *reinterpret_cast<i32x2*>(base + off8 + X8IMG) = lv;    // image 1: xl'
      }
    }
  };
  // PARC = compile-time parity (sub-pixel form) so every tap index / swizzle offset is static
  auto tap_ok_of = [](int t, int PY, int PX) { return !(SUBPIX == 2 && (((t >> 1) && PY) || ((t & 1) && PX))); };
  auto tap_yx = [](int t, int PY, int PX, int& ky, int& kx) {
    ky = SUBPIX == 1 ? PY + (t >> 1) : (SUBPIX == 2 ? 1 + (t >> 1) : t / 3);
    kx = SUBPIX == 1 ? PX + (t & 1) : (SUBPIX == 2 ? 1 + (t & 1) : t % 3);
  };
  // f16 main term of one 16-pixel k-step: two A fragments (the wave's co-tiles), one B fragment per tap of the wave's
  // group.  The A fragments and the first B fragment of a k-step are requested by the PREVIOUS phase in front of its
  // last MFMAs (F16 hand-over), so that a phase does not open with an exposed LDS latency.
  struct F16 { f16x8 ah[2], bh; };
  auto xptrs = [&](const unsigned char* base, int ks, const unsigned char* (&q)[6]) __attribute__((always_inline)) {
    // first pixel of a tap's k-step window: (row+ky)*40 + x0 + kx = kx (mod 8).  One lane address per kx and
    // pixel-row half (6 VALU adds per k-step); ky moves it by a whole halo row = an instruction offset.  (Left to
    // the compiler every tap cost three address instructions: 30 of the k-step's 40 VALU instructions.)
    const int row = ks >> 1, x0 = 16 * (ks & 1);
    const unsigned char* xw = base + OFF_X + cih * (NXP * 64) + (row * XW + x0) * 64;   // uniform
    q[0] = xw + sx0;
    q[1] = xw + (sx1 + 64);
    q[2] = xw + (sx2 + 128);
    q[3] = xw + (sx4 + 4 * 64);
    q[4] = xw + (sx5 + 4 * 64 + 64);
    q[5] = xw + (sx6 + 4 * 64 + 128);
  };
  // first / one-past-last tap of group TG, and the next active tap at or after t (NT if none)
  auto next_ok = [&](int t, int t1, int PY, int PX) { while (t < t1 && !tap_ok_of(t, PY, PX)) ++t; return t < t1 ? t : NT; };
  auto first16 = [&](const unsigned char* base, int ks, auto parc, auto tgc) __attribute__((always_inline)) {
    constexpr int PARC = decltype(parc)::value, PY = (PARC >> 1) & 1, PX = PARC & 1, TG = decltype(tgc)::value;
    constexpr int T0 = TG ? NTA : 0, T1 = TG ? NT : NTA;
    F16 f;
    const unsigned char* ap = base + ct * 4096 + ks * 16 * 64;
    f.ah[0] = __builtin_bit_cast(f16x8, tr_frag16(ap + sa0, ap + sa1));
    f.ah[1] = __builtin_bit_cast(f16x8, tr_frag16(ap + 4096 + sa0, ap + 4096 + sa1));
    const unsigned char* q[6];
    xptrs(base, ks, q);
    const int tf = next_ok(T0, T1, PY, PX);
    int ky = 0, kx = 0;
    if (tf < NT) tap_yx(tf, PY, PX, ky, kx);
    f.bh = __builtin_bit_cast(f16x8, tr_frag16(q[kx] + ky * (XW * 64), q[3 + kx] + ky * (XW * 64)));
    return f;
  };
  auto compute16_p = [&](const unsigned char* base, int ks, auto parc, auto tgc, F16 f, auto want_next, F16& nxt)
      __attribute__((always_inline)) {
    constexpr int PARC = decltype(parc)::value, PY = (PARC >> 1) & 1, PX = PARC & 1, TG = decltype(tgc)::value;
    constexpr int T0 = TG ? NTA : 0, T1 = TG ? NT : NTA;
    constexpr bool NEXT = decltype(want_next)::value;
    const unsigned char* q[6];
    xptrs(base, ks, q);
    auto xfrag = [&](int t) __attribute__((always_inline)) {
      int ky, kx;
      tap_yx(t, PY, PX, ky, kx);
      return __builtin_bit_cast(f16x8, tr_frag16(q[kx] + ky * (XW * 64), q[3 + kx] + ky * (XW * 64)));
    };
    f16x8 bh = f.bh, nh = f.bh;
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      if (!tap_ok_of(t, PY, PX)) continue;
      const int tn = next_ok(t + 1, T1, PY, PX);
      if (tn < NT) nh = xfrag(tn);
      else if (NEXT) nxt = first16(base, ks + 1, parc, tgc);
      acc[t - T0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[0], bh, acc[t - T0][0], 0, 0, 0);
      acc[t - T0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[1], bh, acc[t - T0][1], 0, 0, 0);
      bh = nh;
    }
  };
  // both correction terms of one 32-pixel output row: K = 64 = [dyl' | dyh'] . [xh' | xl']
  auto compute8_p = [&](const unsigned char* base, int row, auto parc, auto tgc, F16& nxt) __attribute__((always_inline)) {
    constexpr int PARC = decltype(parc)::value, PY = (PARC >> 1) & 1, PX = PARC & 1, TG = decltype(tgc)::value;
    constexpr int T0 = TG ? NTA : 0, T1 = TG ? NT : NTA;
    auto frag8 = [&](const unsigned char* p) __attribute__((always_inline)) {
      using lds2 = __attribute__((address_space(3))) i32x2;
      const i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p));
      const i32x2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 256));
      const i32x2 v2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 512));
      const i32x2 v3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 768));
      const i32x4 lo = __builtin_shufflevector(v0, v1, 0, 1, 2, 3), hi = __builtin_shufflevector(v2, v3, 0, 1, 2, 3);
      return (i32x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const i32x8 a80 = frag8(base + a8off + row * (32 * 32));
    const i32x8 a81 = frag8(base + a8off + (64 * 32) + row * (32 * 32));
    auto xfrag8 = [&](int t) __attribute__((always_inline)) {
      int ky, kx;
      tap_yx(t, PY, PX, ky, kx);
      return frag8(base + b8off + ((row + ky) * XW + kx) * 32);
    };
    // B fragment of the next tap requested before the current tap's MFMAs (one fragment ahead, fenced per tap;
    // left alone the compiler hoists the transposed reads of all taps above the first MFMA and spills them).
    const int tf = next_ok(T0, T1, PY, PX);
    if (tf >= NT) { nxt = first16(base, 2 * row, parc, tgc); return; }
    if constexpr (EOVAE_WG_LEAN == 1 && SUBPIX == 0) {
      // plain 3x3: 160 accumulator registers leave no room for a second B fragment: single buffer (each pair of
      // MFMAs behind its fragment; the SIMD's partner wave covers the read latency)
#pragma unroll
      for (int t = T0; t < T1; ++t) {
        if (!tap_ok_of(t, PY, PX)) continue;
        const i32x8 b8 = xfrag8(t);
        acc[t - T0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a80, b8, acc[t - T0][0], 1 /* A: e5m2 */,
                                                                         1 /* B: e5m2 */, 0, sc_a, 0, sc_b);
        acc[t - T0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a81, b8, acc[t - T0][1], 1, 1, 0, sc_a, 0, sc_b);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      i32x8 b8 = xfrag8(tf), nb8 = b8;
#pragma unroll
      for (int t = T0; t < T1; ++t) {
        if (!tap_ok_of(t, PY, PX)) continue;
        const int tn = next_ok(t + 1, T1, PY, PX);
        if (tn < NT) nb8 = xfrag8(tn);
        else if (!(EOVAE_WG_LEAN && SUBPIX == 0)) nxt = first16(base, 2 * row, parc, tgc);
        __builtin_amdgcn_sched_barrier(0);
        acc[t - T0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a80, b8, acc[t - T0][0], 1 /* A: e5m2 */,
                                                                         1 /* B: e5m2 */, 0, sc_a, 0, sc_b);
        acc[t - T0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a81, b8, acc[t - T0][1], 1, 1, 0, sc_a, 0, sc_b);
        __builtin_amdgcn_sched_barrier(0);
        b8 = nb8;
      }
    }
  };
  // one output row of the staged tile: its fp8 correction products, then its two f16 k-steps
  auto compute_row_p = [&](const unsigned char* base, int row, auto parc, auto tgc) __attribute__((always_inline)) {
    F16 f0, f1, fd;
    if constexpr (EOVAE_WG_LEAN && SUBPIX == 0) {   // no register room for the hand-over: every phase requests its own
      compute8_p(base, row, parc, tgc, fd);
      __builtin_amdgcn_sched_barrier(0);
      f0 = first16(base, 2 * row, parc, tgc);
      compute16_p(base, 2 * row, parc, tgc, f0, std::false_type{}, fd);
      __builtin_amdgcn_sched_barrier(0);
      f1 = first16(base, 2 * row + 1, parc, tgc);
      compute16_p(base, 2 * row + 1, parc, tgc, f1, std::false_type{}, fd);
    } else {
      compute8_p(base, row, parc, tgc, f0);
      __builtin_amdgcn_sched_barrier(0);
      compute16_p(base, 2 * row, parc, tgc, f0, std::true_type{}, f1);
      __builtin_amdgcn_sched_barrier(0);
      compute16_p(base, 2 * row + 1, parc, tgc, f1, std::false_type{}, fd);
    }
  };
  auto compute_row_g = [&](const unsigned char* base, int row, auto tgc) __attribute__((always_inline)) {
    if (!SUBPIX) { compute_row_p(base, row, std::integral_constant<int, 0>{}, tgc); return; }
    switch (par) {  // uniform
      case 0: compute_row_p(base, row, std::integral_constant<int, 0>{}, tgc); break;
      case 1: compute_row_p(base, row, std::integral_constant<int, 1>{}, tgc); break;
      case 2: compute_row_p(base, row, std::integral_constant<int, 2>{}, tgc); break;
      default: compute_row_p(base, row, std::integral_constant<int, 3>{}, tgc); break;
    }
  };

  // The tile loop is instantiated once per load path: a run-time choice between issue_fast() and
  // issue() inside ONE loop makes the loaded registers phi nodes, and the compiler then waits for
  // vmcnt(0) and copies them at the join -- right after issuing the loads, before the MFMAs.
  // ... and once per tap group: a choice inside the loop would merge the two groups' accumulator updates in phi nodes
  // (535 spilled registers).
  auto run = [&](auto full_t, auto tgc) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
    auto load = [&](int tile, int hf) __attribute__((always_inline)) {
      if constexpr (FULL) issue_fast(tile, hf); else issue(tile, hf);
    };
    int tile = ysub;
    if (tile < a.ntiles) {
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {
        load(tile, hf);
        commit(0, hf);
      }
    }
    __syncthreads();
    int cur = 0;
    for (; tile < a.ntiles; tile += a.split) {
      const int next = tile + a.split;
      const bool has_next = next < a.ntiles;
      const unsigned char* base = smem + cur * BUF;
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {
        if (has_next) load(next, hf);
        // the row's fp8 products first, then its two f16 k-steps, then the commit: the loads above have all 54
        // MFMAs of the half tile (2304 cycles per wave) to land.  (With the commit between the two groups of
        // MFMAs chosen by the compiler the register allocator spilled 230 VGPRs; scheduling fences keep this
        // order.  Staggering the commit between waves 0-3 and 4-7 -- SQ counters show VALU 2 x 23 % and MFMA
        // 38 % of a SIMD's cycles adding up -- spilled 498 in either arrangement tried.)
        __builtin_amdgcn_sched_barrier(0);
        compute_row_g(base, hf, tgc);   // the row's two k-steps share one K = 64 correction product per tap
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) commit(cur ^ 1, hf);
      }
      __syncthreads();
      cur ^= 1;
    }
  };
  if (tg == 0) {   // wave-uniform
    if (fullc) run(std::true_type{}, std::integral_constant<int, 0>{}); else run(std::false_type{}, std::integral_constant<int, 0>{});
  } else {
    if (fullc) run(std::true_type{}, std::integral_constant<int, 1>{}); else run(std::false_type{}, std::integral_constant<int, 1>{});
  }

  if (do_db) {  // lanes 16q..16q+15 hold the same 4 channels at 16 different pixels
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = dbs[j];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      const int co = cob * 128 + (tid >> 4) * 4 + j - (SUBPIX == 1 ? par * a.creal : 0);   // channel of the stored dy
      if ((tid & 15) == 0 && co < (SUBPIX == 1 ? a.creal : a.Cout)) atomicAdd(&a.db[co], v);
    }
  }
  const int ci = cib * 64 + cih * 32 + r;
  if (ci < a.Cin) {
#pragma unroll
    for (int tl = 0; tl < NTA; ++tl) {
      const int t = tg ? NTA + tl : tl;   // uniform
      if (t >= NT) continue;
      const int tap = SUBPIX == 1 ? (py + (t >> 1)) * 3 + px + (t & 1)
                                  : (SUBPIX == 2 ? (1 + (t >> 1)) * 3 + 1 + (t & 1) : t);
      if (SUBPIX == 2 && (((t >> 1) && py) || ((t & 1) && px))) continue;
      if (SUBPIX == 0 && a.part) {
        // plain stores into this block's own partial tile [tap][128 co][64 ci] (folded in a fixed order afterwards):
        // the float atomics of the 256 blocks were 75 MB per launch at the chip's 1.3 TB/s atomic rate, all at the
        // end of the launch -- 58 us of every weight-gradient launch, 30 % of the 512-channel 32^2 ones
        float* pt = a.part + (((size_t)slab * a.split + ysub) * 9 + tap) * (128 * 64) + cih * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int col = (ct + c) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            pt[col * 64] = acc[tl][c][q] * dyinv;
          }
        continue;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int co = cob * 128 + (ct + c) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          if (co < a.Cout) atomicAdd(&a.dwp[((size_t)tap * a.Cout + co) * a.Cin + ci], acc[tl][c][q] * dyinv);
        }
    }
  }
}

// dw[Cout][Cin][3][3] (PyTorch layout) = sum over the `split` partial tiles of each (co-block, ci-block) slab, in a
// fixed order: deterministic, and the unpack step is gone.  One thread per (co, ci, tap).
// One block = one slab x ONE co row: 9 waves, wave = tap, lane = ci; every read is a 256-byte run of a partial tile, the sums
// pass through LDS and leave as one 2,304-byte run of dw[co][ci0 .. ci0 + 63][tap] (a thread per output wrote 4 bytes of
// each 32-byte sector at a time, eight visits per sector spread over the launch).  Same summation order per output.
__global__ __launch_bounds__(576) void wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout,
                                                         int Cin, int ncib, int split) {
  __shared__ float t[64 * 9];
  const int slab = blockIdx.x >> 7, col = blockIdx.x & 127;
  const int cob = slab / ncib, cib = slab % ncib;
  const int tap = threadIdx.x >> 6, cil = threadIdx.x & 63;
  const float* q = part + (((size_t)slab * split) * 9 + tap) * (128 * 64) + col * 64 + cil;
  float s0 = 0.f, s1 = 0.f;
  int y = 0;
  for (; y + 1 < split; y += 2) {
    s0 += q[(size_t)y * 9 * 128 * 64];
    s1 += q[(size_t)(y + 1) * 9 * 128 * 64];
  }
  if (y < split) s0 += q[(size_t)y * 9 * 128 * 64];
  t[cil * 9 + tap] = s0 + s1;
  __syncthreads();
  const int oc = cob * 128 + col, k = threadIdx.x, ic = cib * 64 + k / 9;
  if (oc < Cout && ic < Cin) dw[((size_t)oc * Cin + cib * 64) * 9 + k] = t[k];
}


// max |x| as float bits (a non-negative float orders like its bit pattern), atomically into *out
__global__ void absmax_bits_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  const size_t n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  m = wave_max(m);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, part[i]);
    if (m > 0.f && m < 3.0e38f) atomicMax(out, __builtin_bit_cast(unsigned, m));   // NaN / inf never land here
  }
}

// per-block max |x| -> part[blockIdx.x] (plain stores: no zero-filled target, no atomics)
__global__ void absmax_part_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  float m = 0.f;
  const size_t n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  m = wave_max(m);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, part[i]);
    partial[blockIdx.x] = (m < 3.0e38f) ? m : 0.f;   // NaN / inf never become the scale
  }
}

// An UPPER BOUND of max|y| from the (sum, sum of squares) per (image, channel group) that the producing convolution's
// epilogue accumulated for the following GroupNorm: max|y| <= sqrt(max_g sum_g y^2).  Good enough for the range guard's
// power-of-two pre-scale (which only has to keep |x| * scale inside f16 / e5m2; a bound that is 2^k too large costs k of
// f16's ~27 spare binades below the amax) and costs one 512-value reduction instead of a pass over the tensor.
__global__ __launch_bounds__(256) void amax_bound_kernel(const double* __restrict__ sums, int n, unsigned* __restrict__ out) {
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double q = sums[2 * (size_t)i + 1];
    m = q > m ? q : m;                        // NaN never wins
  }
  m = wave_max_d(m);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
    float b = (float)sqrt(m);
    b = b * 1.0000002f + 1e-30f;              // never below the true root after the two roundings
    *out = (b < 3.0e38f) ? __builtin_bit_cast(unsigned, b) : 0x7f7fffffu;
  }
}

__device__ __forceinline__ int cdiv_d(int a, int b) { return (a + b - 1) / b; }
constexpr int PACK_PARTS = 256;   // amax partials kept behind the header of a packed weight buffer

// w[Cout][Cin][9] fp32 -> per (tap, 32-ch chunk, 128-co block) one 16 KiB tile:
//   [0, 8 KiB)     f16  [octet 0..3][128 m][8 k]
//   [8 KiB, 16 KiB) e4m3 [wh' | wl'][half 0..1][128 m][16 k]
// flip = 0: k = ci, m = co;  flip = 1 (data gradient): k = co, m = ci, taps reversed.
// One thread = 4 consecutive k of one (tile, m): one dword of each fp8 image, 8 bytes of the f16 image.
__device__ __forceinline__ void pack_f16c8_body(const float* __restrict__ w, unsigned char* __restrict__ wp, int Cout,
                                                int Cin, int nchunk, int ncob, int flip, float scale,
                                                const float* __restrict__ parts, int nparts, unsigned* __restrict__ hdr,
                                                unsigned bid, unsigned nblk) {
  // amax(|w|) of the UNSCALED weights = max of the partial maxima absmax_part_kernel left behind the header
  // (every block folds the <= 256 numbers itself: no fold launch, no zero-filled atomic target); `scale` is a
  // power of two.  Block 0 publishes the bits in this buffer's header for the convolution kernel.
  __shared__ float wmax_s[4];
  {
    float m = (int)threadIdx.x < nparts ? parts[threadIdx.x] : 0.f;
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax_s[threadIdx.x >> 6] = m;
    __syncthreads();
  }
  const float wmax = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
  const unsigned wbits = __builtin_bit_cast(unsigned, wmax);
  if (bid == 0 && threadIdx.x == 0) *hdr = wbits;
  const float amax = wmax * fabsf(scale);
  const int e1 = amax > 0.f ? 7 - expo_of_bits(__builtin_bit_cast(unsigned, amax)) : 0;
  const float s_hi = pow2f(e1), s_lo = pow2f(e1 + 12);   // |wl| <= 2^-11 |w|: wl * s_lo stays under 256
  const size_t total = (size_t)9 * nchunk * ncob * 128 * 8;
  for (size_t i = bid * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)nblk * blockDim.x) {
    const int k4 = i & 7;
    size_t t = i >> 3;
    const int m_ = t & 127;
    t >>= 7;
    const int cob = t % ncob;
    t /= ncob;
    const int chunk = t % nchunk;
    const int tap = t / nchunk;
    const int st = flip ? 8 - tap : tap;
    float v[4], lo[4];
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 hh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = chunk * 32 + k4 * 4 + j, m = cob * 128 + m_;
      const int co = flip ? k : m, ci = flip ? m : k;
      v[j] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + st] * scale : 0.f;
      hh[j] = (_Float16)v[j];
      lo[j] = v[j] - (float)hh[j];
    }
    int w_hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * s_hi, v[1] * s_hi, 0, false);
    w_hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * s_hi, v[3] * s_hi, w_hi, true);
    int w_lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0] * s_lo, lo[1] * s_lo, 0, false);
    w_lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2] * s_lo, lo[3] * s_lo, w_lo, true);
    unsigned char* tile = wp + ((size_t)(tap * nchunk + chunk) * ncob + cob) * C8_WTILE;
    const int kk = k4 * 4;   // first k of this thread inside the chunk
    *reinterpret_cast<f16x4*>(tile + (kk >> 3) * 2048 + m_ * 16 + (kk & 7) * 2) = hh;
    unsigned char* p8 = tile + 8192 + ((kk >> 4) * 128 + m_) * 16 + (kk & 15);
    *reinterpret_cast<int*>(p8) = w_hi;
    *reinterpret_cast<int*>(p8 + 2 * 128 * 16) = w_lo;
  }
  // MX-FP6 part of every tile (conv_f16c8_common.h): one thread = one (tile, row m), both operands
  const size_t rows = (size_t)9 * nchunk * ncob * 128;
  for (size_t i = bid * (size_t)blockDim.x + threadIdx.x; i < rows; i += (size_t)nblk * blockDim.x) {
    const int m_ = i & 127;
    size_t t = i >> 7;
    const int cob = t % ncob;
    t /= ncob;
    const int chunk = t % nchunk;
    const int tap = t / nchunk;
    const int st = flip ? 8 - tap : tap;
    float v[32], lo[32], ah = 0.f, al = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int k = chunk * 32 + j, m = cob * 128 + m_;
      const int co = flip ? k : m, ci = flip ? m : k;
      v[j] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + st] * scale : 0.f;
      lo[j] = v[j] - (float)(_Float16)v[j];
      ah = fmaxf(ah, fabsf(v[j]));
      al = fmaxf(al, fabsf(lo[j]));
    }
    i32x4 q0, q1;
    unsigned char* p6 = wp + ((size_t)(tap * nchunk + chunk) * ncob + cob) * C8_WTILE + C8_WOFF6 + m_ * 16;
    mx6_quant(v, ah, q0, q1);                       // wh6: pairs with the activations' residual part
    *reinterpret_cast<i32x4*>(p6) = q0;
    *reinterpret_cast<i32x4*>(p6 + 128 * 16) = q1;
    mx6_quant(lo, al, q0, q1);                      // wl6: pairs with the activations' leading part
    *reinterpret_cast<i32x4*>(p6 + 2 * 128 * 16) = q0;
    *reinterpret_cast<i32x4*>(p6 + 3 * 128 * 16) = q1;
  }
}

__global__ void pack_f16c8_kernel(const float* __restrict__ w, unsigned char* __restrict__ wp, int Cout, int Cin,
                                  int nchunk, int ncob, int flip, float scale, const float* __restrict__ parts,
                                  int nparts, unsigned* __restrict__ hdr) {
  pack_f16c8_body(w, wp, Cout, Cin, nchunk, ncob, flip, scale, parts, nparts, hdr, blockIdx.x, gridDim.x);
}

// Every plain 3x3 layer of a model in TWO launches per step instead of three per layer (176 launches of 5-10 us each
// in the full-size model): blockIdx.y = layer (x orientation).  Descriptors live in device memory.
struct PackDesc {
  const float* w;
  unsigned char* wp_fwd;
  unsigned char* wp_bwd;
  int Cout, Cin;
};
constexpr int PACK_MULTI_PARTS = 64;
__device__ __forceinline__ float* pack_parts_of(const PackDesc& d) {   // behind the header of the forward buffer
  return reinterpret_cast<float*>(d.wp_fwd + (size_t)9 * cdiv_d(d.Cin, 32) * cdiv_d(d.Cout, 128) * C8_WTILE + 16);
}
__global__ __launch_bounds__(256) void pack_multi_amax_kernel(const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  const size_t n = (size_t)d.Cout * d.Cin * 9;
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)PACK_MULTI_PARTS * 256) m = fmaxf(m, fabsf(d.w[i]));
  m = wave_max(m);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    pack_parts_of(d)[blockIdx.x] = (m < 3.0e38f) ? m : 0.f;
  }
}
// Both orientations of a layer from ONE coalesced read of its weights.  The per-element gather of pack_f16c8_body reads
// w[co][ci][tap] at a 36-byte stride per tap pass (9 passes over the tensor) and, for the flipped image, one cache line per
// lane: 0.94 ms per step for the 84 M weights of the full model (the step timeline of the build before this kernel; after: 0.32 ms, profiles/r04w_step_timeline.txt), 0.7 TB/s.  Here a block
// stages a 32 co x 32 ci x 9 tap sub-block in LDS (32 contiguous 1,152-byte rows) and writes the 9 x 2 tile pieces it
// owns: 512-byte runs in the packed images.  Same arithmetic per element as pack_f16c8_body (bit-identical output).
constexpr int PKB = 32, PKP = PKB * 9 + 1;
// wp_fwd / wp_bwd: either may be null (that orientation is skipped); parts: the amax partials of the UNSCALED weights
__device__ __forceinline__ void pack_tiles_body(const float* __restrict__ w, unsigned char* __restrict__ wp_fwd,
                                                unsigned char* __restrict__ wp_bwd, int Cout, int Cin, float scale,
                                                const float* __restrict__ parts, int nparts, unsigned bid, unsigned nblk) {
  __shared__ float tile[PKB * PKP];
  __shared__ float wmax_s[4];
  const int tid = threadIdx.x;
  {
    float m = tid < nparts ? parts[tid] : 0.f;
    m = wave_max(m);
    if ((tid & 63) == 0) wmax_s[tid >> 6] = m;
    __syncthreads();
  }
  const float wmax = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
  // orientation 0 (forward): k = ci, m = co;  1 (data gradient): k = co, m = ci, taps reversed
  const int nchunk0 = cdiv_d(Cin, 32), ncob0 = cdiv_d(Cout, 128);
  const int nchunk1 = cdiv_d(Cout, 32), ncob1 = cdiv_d(Cin, 128);
  if (bid == 0 && tid == 0) {
    const unsigned wbits = __builtin_bit_cast(unsigned, wmax);
    if (wp_fwd) *reinterpret_cast<unsigned*>(wp_fwd + (size_t)9 * nchunk0 * ncob0 * C8_WTILE) = wbits;
    if (wp_bwd) *reinterpret_cast<unsigned*>(wp_bwd + (size_t)9 * nchunk1 * ncob1 * C8_WTILE) = wbits;
  }
  const float amax = wmax * fabsf(scale);
  const int e1 = amax > 0.f ? 7 - expo_of_bits(__builtin_bit_cast(unsigned, amax)) : 0;
  const float s_hi = pow2f(e1), s_lo = pow2f(e1 + 12);
  const int cobk = ncob0 * 4, cibk = ncob1 * 4;          // 32-blocks over the 128-padded extents
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  for (int blk = (int)bid; blk < cobk * cibk; blk += (int)nblk) {
    const int co0 = (blk / cibk) * PKB, ci0 = (blk % cibk) * PKB;
    const bool fwd = wp_fwd && ci0 < nchunk0 * 32, bwd = wp_bwd && co0 < nchunk1 * 32;
    if (!fwd && !bwd) continue;
    const int ncol = max(0, min(PKB, Cin - ci0)) * 9;
    __syncthreads();                                     // the previous sub-block has been read out
    for (int e = tid; e < PKB * PKB * 9; e += 256) {
      const int r = e / (PKB * 9), c = e % (PKB * 9);
      tile[r * PKP + c] = (co0 + r < Cout && c < ncol) ? w[((size_t)(co0 + r) * Cin + ci0) * 9 + c] * scale : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < 18; ++it) {
      const int o = it / 9, tap = it % 9;
      if (o == 0 ? !fwd : !bwd) continue;
      const int k4 = tid & 7, ml = tid >> 3;
      float v[4], lo[4];
      f16x4 hh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kl = k4 * 4 + j;
        v[j] = o == 0 ? tile[ml * PKP + kl * 9 + tap] : tile[kl * PKP + ml * 9 + (8 - tap)];
        hh[j] = (_Float16)v[j];
        lo[j] = v[j] - (float)hh[j];
      }
      int w_hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * s_hi, v[1] * s_hi, 0, false);
      w_hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * s_hi, v[3] * s_hi, w_hi, true);
      int w_lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0] * s_lo, lo[1] * s_lo, 0, false);
      w_lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2] * s_lo, lo[3] * s_lo, w_lo, true);
      const int k0 = o == 0 ? ci0 : co0, mm = (o == 0 ? co0 : ci0) + ml;
      const int nchunk = o == 0 ? nchunk0 : nchunk1, ncob = o == 0 ? ncob0 : ncob1;
      unsigned char* t = (o == 0 ? wp_fwd : wp_bwd) + ((size_t)(tap * nchunk + (k0 >> 5)) * ncob + (mm >> 7)) * C8_WTILE;
      const int m_ = mm & 127, kk = k4 * 4;
      *reinterpret_cast<f16x4*>(t + (kk >> 3) * 2048 + m_ * 16 + (kk & 7) * 2) = hh;
      unsigned char* p8 = t + 8192 + ((kk >> 4) * 128 + m_) * 16 + (kk & 15);
      *reinterpret_cast<int*>(p8) = w_hi;
      *reinterpret_cast<int*>(p8 + 2 * 128 * 16) = w_lo;
    }
    // MX-FP6 part (conv_f16c8_common.h): a thread takes one (orientation, tap, row) with all 32 k of the sub-block --
    // 576 tasks; lanes differ in the row, i.e. by PKP (orientation 0) or 9 (orientation 1) floats: conflict-free
    for (int task = tid; task < 2 * 9 * PKB; task += 256) {
      const int o = task / (9 * PKB), tap = (task / PKB) % 9, ml = task % PKB;
      if (o == 0 ? !fwd : !bwd) continue;
      float v[32], lo[32], ah = 0.f, al = 0.f;
#pragma unroll
      for (int kl = 0; kl < 32; ++kl) {
        v[kl] = o == 0 ? tile[ml * PKP + kl * 9 + tap] : tile[kl * PKP + ml * 9 + (8 - tap)];
        lo[kl] = v[kl] - (float)(_Float16)v[kl];
        ah = fmaxf(ah, fabsf(v[kl]));
        al = fmaxf(al, fabsf(lo[kl]));
      }
      const int k0 = o == 0 ? ci0 : co0, mm = (o == 0 ? co0 : ci0) + ml;
      const int nchunk = o == 0 ? nchunk0 : nchunk1, ncob = o == 0 ? ncob0 : ncob1;
      unsigned char* p6 = (o == 0 ? wp_fwd : wp_bwd) + ((size_t)(tap * nchunk + (k0 >> 5)) * ncob + (mm >> 7)) * C8_WTILE +
                          C8_WOFF6 + (mm & 127) * 16;
      i32x4 q0, q1;
      mx6_quant(v, ah, q0, q1);
      *reinterpret_cast<i32x4*>(p6) = q0;
      *reinterpret_cast<i32x4*>(p6 + 128 * 16) = q1;
      mx6_quant(lo, al, q0, q1);
      *reinterpret_cast<i32x4*>(p6 + 2 * 128 * 16) = q0;
      *reinterpret_cast<i32x4*>(p6 + 3 * 128 * 16) = q1;
    }
  }
}
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackDesc* __restrict__ descs, float scale) {
  const PackDesc d = descs[blockIdx.y];
  pack_tiles_body(d.w, d.wp_fwd, d.wp_bwd, d.Cout, d.Cin, scale, pack_parts_of(d), PACK_MULTI_PARTS, blockIdx.x, gridDim.x);
}
// one layer, either or both orientations (eovae_conv3x3_f16c8_pack / _pack_pair)
__global__ __launch_bounds__(256) void pack_tiles_kernel(const float* __restrict__ w, unsigned char* __restrict__ wp_fwd,
                                                         unsigned char* __restrict__ wp_bwd, int Cout, int Cin, float scale,
                                                         const float* __restrict__ parts, int nparts) {
  pack_tiles_body(w, wp_fwd, wp_bwd, Cout, Cin, scale, parts, nparts, blockIdx.x, gridDim.x);
}

}  // namespace

extern "C" {

#if EOVAE_C8_STAMP
int eovae_debug_c8_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(c8_stamps), bytes < sizeof(c8_stamps) ? bytes : sizeof(c8_stamps));
}
#endif

// *out_bits = float bits of max |x[i]| (0 for an all-zero tensor); out_bits is device memory
int eovae_absmax_bits(const float* x, size_t n, unsigned* out_bits, void* stream) {
  EOVAE_CHECK_ARG(x && out_bits && n > 0, "absmax_bits: bad argument");
  EOVAE_CHECK_ARG(((uintptr_t)x & 15) == 0, "absmax_bits: x must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out_bits, 0, sizeof(unsigned), st) != hipSuccess) {
    eovae_set_error("absmax_bits: memset failed");
    return EOVAE_ERR_LAUNCH;
  }
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, out_bits);
  EOVAE_CHECK_LAUNCH("eovae_absmax_bits");
  return EOVAE_OK;
}

// *out_bits = float bits of an upper bound of max|y| from y's GroupNorm accumulators sums[n_groups][2] (eovae_conv3x3_f16c8_fwd's
// gn_sums): sqrt of the largest sum of squares.  For range pre-scales only (see amax_bound_kernel).
int eovae_amax_bound_from_sums(const double* sums, int n_groups, unsigned* out_bits, void* stream) {
  EOVAE_CHECK_ARG(sums && out_bits && n_groups > 0, "amax_bound_from_sums: bad argument");
  hipLaunchKernelGGL(amax_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, n_groups, out_bits);
  EOVAE_CHECK_LAUNCH("eovae_amax_bound_from_sums");
  return EOVAE_OK;
}

// bytes of the f16c8-packed weight buffer of a 3x3 layer (tiles + 16-byte header with amax(|w|) bits + scratch for
// the amax partials of the pack call)
size_t eovae_conv3x3_f16c8_packed_bytes(int Cout, int Cin, int flip) {
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  return (size_t)9 * cdiv(k, 32) * cdiv(m, 128) * C8_WTILE + 16 + PACK_PARTS * sizeof(float);
}

int eovae_conv3x3_f16c8_pack(const float* w, void* wp, int Cout, int Cin, int flip, float scale, void* stream) {
  EOVAE_CHECK_ARG(w && wp && Cout > 0 && Cin > 0, "conv3x3_f16c8_pack: bad argument");
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  const int nchunk = cdiv(k, 32), ncob = cdiv(m, 128);
  unsigned char* base = (unsigned char*)wp;
  unsigned* hdr = (unsigned*)(base + (size_t)9 * nchunk * ncob * C8_WTILE);
  float* parts = (float*)(hdr + 4);
  hipStream_t st = (hipStream_t)stream;
  EOVAE_CHECK_ARG(((uintptr_t)w & 15) == 0, "conv3x3_f16c8_pack: w must be 16-byte aligned");
  const size_t nw = (size_t)Cout * Cin * 9;
  size_t ab = (nw / 4 + 255) / 256;
  if (ab > PACK_PARTS) ab = PACK_PARTS;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(absmax_part_kernel, dim3((unsigned)ab), dim3(256), 0, st, w, nw, parts);
  if (eovae_opt(EOVAE_OPT_C8_PACK_GATHER)) {   // the per-element gather kernel of rounds 2-3 (reference of the LDS-staged pack)
    const size_t total = (size_t)9 * nchunk * ncob * 128 * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_f16c8_kernel, dim3(blocks), dim3(256), 0, st, w, base, Cout, Cin, nchunk, ncob, flip,
                       scale, (const float*)parts, (int)ab, hdr);
  } else {
    int blocks = cdiv(Cout, 128) * 4 * cdiv(Cin, 128) * 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(blocks), dim3(256), 0, st, w, flip ? (unsigned char*)nullptr : base,
                       flip ? base : (unsigned char*)nullptr, Cout, Cin, scale, (const float*)parts, (int)ab);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_pack");
  return EOVAE_OK;
}

// Both orientations of one layer in one call: wp_fwd (flip 0) for the forward convolution and wp_bwd (flip 1) for its
// data gradient, from ONE amax pass (4 launches instead of 6 for a layer that is multiplied forward and backward
// in the same step).  Buffer sizes: eovae_conv3x3_f16c8_packed_bytes(Cout, Cin, 0 / 1).
int eovae_conv3x3_f16c8_pack_pair(const float* w, void* wp_fwd, void* wp_bwd, int Cout, int Cin, float scale,
                                  void* stream) {
  EOVAE_CHECK_ARG(w && wp_fwd && wp_bwd && Cout > 0 && Cin > 0, "conv3x3_f16c8_pack_pair: bad argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* b0 = (unsigned char*)wp_fwd;
  unsigned char* b1 = (unsigned char*)wp_bwd;
  const int nch0 = cdiv(Cin, 32), ncb0 = cdiv(Cout, 128), nch1 = cdiv(Cout, 32), ncb1 = cdiv(Cin, 128);
  unsigned* hdr0 = (unsigned*)(b0 + (size_t)9 * nch0 * ncb0 * C8_WTILE);
  unsigned* hdr1 = (unsigned*)(b1 + (size_t)9 * nch1 * ncb1 * C8_WTILE);
  EOVAE_CHECK_ARG(((uintptr_t)w & 15) == 0, "conv3x3_f16c8_pack_pair: w must be 16-byte aligned");
  float* parts = (float*)(hdr0 + 4);
  const size_t nw = (size_t)Cout * Cin * 9;
  size_t ab = (nw / 4 + 255) / 256;
  if (ab > PACK_PARTS) ab = PACK_PARTS;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(absmax_part_kernel, dim3((unsigned)ab), dim3(256), 0, st, w, nw, parts);
  auto launch = [&](unsigned char* base, int nchunk, int ncob, int flip, unsigned* hdr) {
    const size_t total = (size_t)9 * nchunk * ncob * 128 * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_f16c8_kernel, dim3(blocks), dim3(256), 0, st, w, base, Cout, Cin, nchunk, ncob, flip,
                       scale, (const float*)parts, (int)ab, hdr);
  };
  if (eovae_opt(EOVAE_OPT_C8_PACK_GATHER)) {
    launch(b0, nch0, ncb0, 0, hdr0);
    launch(b1, nch1, ncb1, 1, hdr1);
  } else {   // both orientations from one coalesced read of the weights
    int blocks = cdiv(Cout, 128) * 4 * cdiv(Cin, 128) * 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(blocks), dim3(256), 0, st, w, b0, b1, Cout, Cin, scale, (const float*)parts, (int)ab);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_pack_pair");
  return EOVAE_OK;
}

// All plain 3x3 layers of a model at once.  descs: DEVICE array of n records {const float* w; void* wp_fwd; void* wp_bwd;
// int Cout; int Cin;} (32 bytes each, buffers sized by eovae_conv3x3_f16c8_packed_bytes(Cout, Cin, 0 / 1)); the result
// is what n calls of eovae_conv3x3_f16c8_pack_pair produce, in two launches.
int eovae_conv3x3_f16c8_pack_multi(const void* descs, int n, float scale, void* stream) {
  EOVAE_CHECK_ARG(descs && n > 0 && n <= 32767, "conv3x3_f16c8_pack_multi: bad argument");
  static_assert(sizeof(PackDesc) == 32, "PackDesc layout is part of the ABI");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_multi_amax_kernel, dim3(PACK_MULTI_PARTS, n), dim3(256), 0, st, (const PackDesc*)descs);
  hipLaunchKernelGGL(pack_multi_kernel, dim3(64, n), dim3(256), 0, st, (const PackDesc*)descs, scale);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_pack_multi");
  return EOVAE_OK;
}

static int c8_n_cu() {
  static int n_cu = 0;   // one persistent block per CU
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    n_cu -= n_cu % 8;   // keeps  tile = blockIdx.x + k * gridDim.x  on one XCD (dealt round-robin)
    if (n_cu < 8) n_cu = 8;
  }
  return n_cu;
}

// Which kernel eovae_conv3x3_f16c8_fwd launches for these arguments: 1 = conv3x3_f16c8_w4_kernel (one wave per SIMD, 16-row
// tiles; plain layers that give every CU two tiles), 2 = conv3x3_f16c8_q16_kernel (the same walk on 16 x 16 MFMAs with an
// all-DMA feed: those layers on MX operand records), 0 = conv3x3_f16c8_kernel (8 waves, 8-row tiles; every other form).
// For callers that attribute measured time to a kernel (bench.py); no launch, no device work beyond the first
// property query.
int eovae_conv3x3_f16c8_which(int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int mode, int subpix,
                              int xpacked) {
  ConvCArgs a;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout; a.mode = mode;
  a.packed = xpacked; a.x6 = nullptr; a.mx = 0; a.gn_sums = nullptr; a.gn_shift = 0;
  // a forward call on records carries the following GroupNorm(32)'s sums (ops.GN_GROUPS_HINT): the group size the launcher sees
  if (xpacked && Cout % 32 == 0) {
    a.gn_sums = reinterpret_cast<double*>(1);
    while ((1 << a.gn_shift) < Cout / 32) ++a.gn_shift;
  }
  a.nchunk = Cin / 32; a.ncob = cdiv(Cout, 128);
  const int n_cu = c8_n_cu();
  if (!(n_cu > 0 && subpix == 0 && eovae_c8_w4_eligible(a, subpix, n_cu))) return 0;
  return eovae_c8_one_wave_kind(a, n_cu);   // the launcher's own decision function
}

// 3x3 stride-1 conv / data gradient with f16c8 operands.  Arguments as eovae_conv3x3_bf16_fwd; the GEMM K
// channel count must be a multiple of 32.  xamax: device pointer to the float bits of max |x|
// (eovae_absmax_bits) when x may sit outside f16's range (gradients), else NULL.
int eovae_conv3x3_f16c8_fwd_amax(const float* x, const void* wp, const float* bias, const float* residual, float* y,
                                 const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                                 int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                                 unsigned* out_amax, void* stream);

int eovae_conv3x3_f16c8_fwd_mx(const float* x, const void* x6, const void* wp, const float* bias, const float* residual,
                               float* y, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                               int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                               unsigned* out_amax, void* stream);

int eovae_conv3x3_f16c8_fwd(const float* x, const void* wp, const float* bias, const float* residual, float* y,
                            const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                            int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                            void* stream) {
  return eovae_conv3x3_f16c8_fwd_mx(x, nullptr, wp, bias, residual, y, xamax, N, Cin, Hin, Win, Cout, Hout, Wout, mode, subpix,
                                    xpacked, gn_sums, gn_groups, nullptr, stream);
}

// As eovae_conv3x3_f16c8_fwd; out_amax (nullable, a pre-zeroed device word) receives the float bits of max|y| by atomicMax
// from the epilogue: the range pre-scale of the layer that reads y, without a reduction pass over y.
int eovae_conv3x3_f16c8_fwd_amax(const float* x, const void* wp, const float* bias, const float* residual, float* y,
                                 const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                                 int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                                 unsigned* out_amax, void* stream) {
  return eovae_conv3x3_f16c8_fwd_mx(x, nullptr, wp, bias, residual, y, xamax, N, Cin, Hin, Win, Cout, Hout, Wout, mode, subpix,
                                    xpacked, gn_sums, gn_groups, out_amax, stream);
}

// As eovae_conv3x3_f16c8_fwd_amax; x6 (nullable; record input only) = the records' MX-FP6 plane written by
// eovae_gn_silu_fwd_packed_mx: kernels that have the FP6 form (conv_f16c8_w4.hip) then run the correction product on it at
// ~1.6 x the fp8 rate; kernels that do not ignore it (the records still carry their e5m2 parts).
int eovae_conv3x3_f16c8_fwd_mx(const float* x, const void* x6, const void* wp, const float* bias, const float* residual,
                               float* y, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                               int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                               unsigned* out_amax, void* stream) {
  EOVAE_CHECK_ARG(x && wp && y, "conv3x3_f16c8_fwd: null pointer");
  EOVAE_CHECK_ARG(xpacked >= 0 && xpacked <= 2, "conv3x3_f16c8_fwd: xpacked %d (0 fp32, 1 operand records, 2 MX operand records)", xpacked);
  EOVAE_CHECK_ARG(xpacked != 2 || Cin % 32 == 0, "conv3x3_f16c8_fwd: MX operand records need Cin %% 32 == 0");
  if (xpacked == 2 && !x6) x6 = reinterpret_cast<const char*>(x) + (size_t)N * Cin * Hin * Win * 2;   // S plane behind the F plane
  EOVAE_CHECK_ARG(!x6 || xpacked == 2, "conv3x3_f16c8_fwd_mx: an FP6 plane belongs to MX operand records (xpacked = 2)");
  EOVAE_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv3x3_f16c8_fwd: bad shape");
  EOVAE_CHECK_ARG(mode == MODE_NORMAL, "conv3x3_f16c8_fwd: mode %d unsupported (stride-1 'normal' layers and the "
                  "sub-pixel / space-to-depth forms only)", mode);
  EOVAE_CHECK_ARG(subpix >= 0 && subpix <= 4, "conv3x3_f16c8_fwd: subpix %d", subpix);
  ConvCArgs a;
  a.x = x; a.wp = (const unsigned char*)wp; a.bias = bias; a.res = residual; a.y = y; a.xamax = xamax;
  a.N = N; a.Hin = Hin; a.Win = Win; a.mode = mode; a.subpix = subpix; a.creal = 0; a.cpb = 1; a.packed = xpacked;
  a.stagger = 0; a.skew = 0; a.out_amax = out_amax; a.x6 = (const unsigned char*)x6; a.mx = 0;
  EOVAE_CHECK_ARG(!xpacked || (subpix == 0 && !xamax), "conv3x3_f16c8_fwd: packed input is a forward, plain 3x3 input");
  if (subpix == 0) {
    const int eh = mode == MODE_NORMAL ? Hin : 2 * Hin, ew = mode == MODE_NORMAL ? Win : 2 * Win;
    EOVAE_CHECK_ARG(Hout == eh && Wout == ew, "conv3x3_f16c8_fwd: output %dx%d does not match mode %d", Hout,
                    Wout, mode);
    a.Cin = Cin; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  } else if (subpix == 1) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual, "conv3x3_f16c8_fwd: subpix 1 needs mode 0, no residual");
    EOVAE_CHECK_ARG(Cout % 128 == 0, "conv3x3_f16c8_fwd: subpix 1 needs Cout %% 128 == 0 (got %d)", Cout);
    EOVAE_CHECK_ARG(Hout == 2 * Hin && Wout == 2 * Win, "conv3x3_f16c8_fwd: subpix 1 output must be 2x");
    a.Cin = Cin; a.Cout = 4 * Cout; a.creal = Cout; a.cpb = Cout / 128;
    a.Hout = Hin; a.Wout = Win;
  } else if (subpix == 2 || subpix == 3) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual, "conv3x3_f16c8_fwd: subpix %d needs mode 0, no residual", subpix);
    EOVAE_CHECK_ARG(Hin == 2 * Hout && Win == 2 * Wout, "conv3x3_f16c8_fwd: subpix %d output must be 1/2", subpix);
    a.Cin = 4 * Cin; a.creal = Cin; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  } else {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual && !bias, "conv3x3_f16c8_fwd: subpix 4 is a bare data gradient");
    EOVAE_CHECK_ARG(Cout % 128 == 0, "conv3x3_f16c8_fwd: subpix 4 needs Cout %% 128 == 0 (got %d)", Cout);
    EOVAE_CHECK_ARG(Hout == 2 * Hin && Wout == 2 * Win, "conv3x3_f16c8_fwd: subpix 4 output must be 2x");
    a.Cin = Cin; a.Cout = 4 * Cout; a.creal = Cout; a.cpb = Cout / 128;
    a.Hout = Hin; a.Wout = Win;
  }
  const int kreal = (subpix == 2 || subpix == 3) ? a.creal : a.Cin;
  EOVAE_CHECK_ARG(kreal % 32 == 0, "conv3x3_f16c8_fwd: K channels (%d) must be a multiple of 32", kreal);
  a.gn_sums = gn_sums; a.gn_groups = gn_groups; a.gn_shift = 0;
  if (gn_sums) {   // GroupNorm statistics of y from the epilogue: groups of 4 .. 128 channels, a power of two
    const int creal = (subpix == 1 || subpix == 4) ? a.creal : a.Cout;   // channels of the tensor being written
    const int cpg = gn_groups > 0 && creal % gn_groups == 0 ? creal / gn_groups : 0;
    EOVAE_CHECK_ARG(cpg >= 4 && cpg <= 128 && (cpg & (cpg - 1)) == 0,
                    "conv3x3_f16c8_fwd: gn_sums needs Cout / gn_groups in {4, 8, ..., 128} (Cout %d, groups %d)", creal,
                    gn_groups);
    while ((1 << a.gn_shift) < cpg) ++a.gn_shift;
  }
  Cin = a.Cin; Cout = a.Cout; Hout = a.Hout; Wout = a.Wout;
  a.nchunk = Cin / 32; a.ncob = cdiv(Cout, 128);
  // The epilogue's GroupNorm cells are folded behind the NEXT tile's first stage barrier and rewritten by that tile's
  // epilogue: at least one more barrier must lie between the two, i.e. a tile needs two stages.  Plain layers have five
  // per 32-channel chunk; the parity forms can have ONE per chunk, so they need two chunks (the model's have >= 16).
  EOVAE_CHECK_ARG(!gn_sums || subpix == 0 || a.nchunk >= 2,
                  "conv3x3_f16c8_fwd: gn_sums with a parity form needs at least 64 K channels (got %d)", Cin);
  a.tiles_x = cdiv(Wout, 32);
  hipStream_t st = (hipStream_t)stream;
  auto xcd_ok = [&](long long tiles) { return (a.ncob > 1 && tiles % 8 == 0) ? 1 : 0; };
  const int n_cu = c8_n_cu();
  if (n_cu < 0) {
    eovae_set_error("conv3x3_f16c8_fwd: cannot query the device");
    return EOVAE_ERR_LAUNCH;
  }
  // one-wave-per-SIMD kernel (conv_f16c8_w4.hip: 4 x 4 register blocking, 16-row tiles) for plain layers that give
  // every CU at least two of its tiles
  if (eovae_c8_w4_eligible(a, subpix, n_cu)) {
    eovae_c8_w4_launch(a, n_cu, st);
    if (out_amax) {   // that kernel's epilogue has no register for a running maximum (+70 spills): one pass over y instead
      const size_t n = (size_t)N * Cout * Hout * Wout;
      size_t blocks = (n / 4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, n, out_amax);
    }
    EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_fwd (w4)");
    return EOVAE_OK;
  }
  // 16-row tiles while they still give every CU two tiles, else 8-row tiles
  const long long blocks16 = (long long)N * cdiv(Hout, 16) * a.tiles_x * a.ncob;
  // 8-row tiles only: the 16-row tile (half the weight traffic per MFMA) needs 128 accumulator + 40 staging
  // registers per lane and spilled into the loop
  const bool th16 = false;
  (void)blocks16;
  a.tiles_y = cdiv(Hout, th16 ? 16 : 8);
  const long long sp_tiles = (long long)N * a.tiles_y * a.tiles_x;
  a.xcd_remap = xcd_ok(sp_tiles);
  EOVAE_CHECK_ARG(sp_tiles * a.ncob < (1LL << 30), "conv3x3_f16c8_fwd: too many tiles");
  a.ntiles = (int)(sp_tiles * a.ncob);
  const dim3 grid((unsigned)(a.ntiles < n_cu ? a.ntiles : n_cu)), blk(512);
#define EOVAE_LAUNCH_C8(TH)                                                                                   \
  switch (subpix) {                                                                                           \
    case 0:                                                                                                   \
      if (xpacked == 2) hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 0, true, true>), grid, blk, 0, st, a);     \
      else if (xpacked) hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 0, true>), grid, blk, 0, st, a);           \
      else hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 0, false>), grid, blk, 0, st, a);                       \
      break;                                                                                                  \
    case 1: hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 1, false>), grid, blk, 0, st, a); break;             \
    case 2: hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 2, false>), grid, blk, 0, st, a); break;             \
    case 3: hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 3, false>), grid, blk, 0, st, a); break;             \
    default: hipLaunchKernelGGL((conv3x3_f16c8_kernel<TH, 4, false>), grid, blk, 0, st, a); break;            \
  }
  EOVAE_LAUNCH_C8(8)
#undef EOVAE_LAUNCH_C8
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_fwd");
  return EOVAE_OK;
}

// dwp[9][Cout][Cin] += dY (x) Xv with f16c8 operands; arguments as eovae_conv3x3_bf16_wgrad plus dyamax
// (device pointer to the float bits of max|dy|, eovae_absmax_bits / eovae_gn_silu_bwd; NULL = unscaled dy) and xamax
// (the same for an fp32 x that is not a GroupNorm output; NULL = x used unscaled, i.e. assumed inside f16's range)
}  // extern "C"

// dwp (pre-zeroed, atomics; every form) OR part + dw (plain layers: partial tiles, folded into dw[Cout][Cin][3][3])
static int c8_wgrad_impl(const float* x, const float* dy, float* dwp, float* part, float* dw, float* db,
                         const unsigned* dyamax, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout,
                         int Hout, int Wout, int mode, int subpix, int xpacked, void* stream) {
  EOVAE_CHECK_ARG(x && dy && (dwp || (part && dw)), "conv3x3_f16c8_wgrad: null pointer");
  EOVAE_CHECK_ARG(subpix >= 0 && subpix <= 2, "conv3x3_f16c8_wgrad: subpix %d", subpix);
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || mode == MODE_UP, "conv3x3_f16c8_wgrad: mode %d unsupported", mode);
  EOVAE_CHECK_ARG(!xpacked || (subpix == 0 && mode == MODE_NORMAL && Cin % 64 == 0),
                  "conv3x3_f16c8_wgrad: packed x needs a plain layer with Cin %% 64 == 0");
  EOVAE_CHECK_ARG(xpacked >= 0 && xpacked <= 2, "conv3x3_f16c8_wgrad: xpacked %d", xpacked);
  WgCArgs a;
  EOVAE_CHECK_ARG(!(xpacked && xamax), "conv3x3_f16c8_wgrad: operand records carry no scale (xamax must be NULL)");
  a.x = x; a.dy = dy; a.dwp = dwp; a.db = db; a.dyamax = dyamax; a.xamax = xamax; a.part = part;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  a.mode = mode; a.subpix = subpix; a.creal = 0; a.cpb = 1;
  if (subpix == 1) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && Cout % 128 == 0 && Hout == 2 * Hin && Wout == 2 * Win,
                    "conv3x3_f16c8_wgrad: subpix 1 needs mode 0, Cout %% 128 == 0 and a 2x dy");
    a.creal = Cout; a.cpb = Cout / 128; a.Cout = 4 * Cout; a.Hout = Hin; a.Wout = Win;
    Cout = a.Cout; Hout = a.Hout; Wout = a.Wout;
  } else if (subpix == 2) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && Cin % 64 == 0 && Hin == 2 * Hout && Win == 2 * Wout,
                    "conv3x3_f16c8_wgrad: subpix 2 needs mode 0, Cin %% 64 == 0 and a 1/2 dy");
    a.creal = Cin; a.Cin = 4 * Cin;
    Cin = a.Cin;
  }
  a.tiles_x = cdiv(Wout, 32); a.tiles_y = cdiv(Hout, 2);
  a.ntiles = N * a.tiles_x * a.tiles_y;
  a.ncib = cdiv(Cin, 64);
  const int ncob = cdiv(Cout, 128);
  int split = EOVAE_WGRAD_BLOCKS / (a.ncib * ncob);   // one block per CU (conv_bf16x3.hip)
  if (split < 1) split = 1;
  if (split > a.ntiles) split = a.ntiles;
  a.ncob_ = ncob; a.split = split;
  a.xcd_remap = (a.ncib * ncob > 1 && split % 8 == 0) ? 1 : 0;
  dim3 grid(a.ncib * ncob * split);
  hipStream_t st = (hipStream_t)stream;
  a.xpacked = xpacked;
  a.nseg = 1; a.units = 0;
  // one-wave-per-SIMD kernel (conv_f16c8_wgrad_w4.hip) for plain layers made of full tiles
  if (!part && xpacked != 2 && eovae_c8_wgrad_w4_eligible(a)) {
    eovae_c8_wgrad_w4_launch(a, st);
    EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_wgrad (w4)");
    return EOVAE_OK;
  }
  // row-streaming kernel (conv_f16c8_wgrad_rows.hip): plain layers with record input and partial-tile output
  if (eovae_c8_wgrad_rows_eligible(a)) {
    eovae_c8_wgrad_rows_launch(a, st);
    split = a.split;
  } else if (xpacked == 2) {
    eovae_set_error("conv3x3_f16c8_wgrad: MX operand records (xpacked = 2) are read by the row-streaming weight gradient only "
                    "(plain layer, Cout %% 128 == 0, Cin %% 64 == 0, W %% 32 == 0, options c8_wgrad_rows = 1, c8_wgrad_corr = 0): ask "
                    "eovae_records_mx_ok() before writing them");
    return EOVAE_ERR_UNSUPPORTED;
  } else
  if (subpix == 1) hipLaunchKernelGGL((wgrad3x3_f16c8_kernel<1, false>), grid, dim3(512), 0, st, a);
  else if (subpix == 2) hipLaunchKernelGGL((wgrad3x3_f16c8_kernel<2, false>), grid, dim3(512), 0, st, a);
  else if (xpacked) hipLaunchKernelGGL((wgrad3x3_f16c8_kernel<0, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((wgrad3x3_f16c8_kernel<0, false>), grid, dim3(512), 0, st, a);
  if (part) {
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)(a.ncib * ncob * 128)), dim3(576), 0, st, part, dw, Cout, Cin, a.ncib, split);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_wgrad");
  return EOVAE_OK;
}

extern "C" {

int eovae_conv3x3_f16c8_wgrad(const float* x, const float* dy, float* dwp, float* db, const unsigned* dyamax,
                              const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                              int mode, int subpix, int xpacked, void* stream) {
  return c8_wgrad_impl(x, dy, dwp, nullptr, nullptr, db, dyamax, xamax, N, Cin, Hin, Win, Cout, Hout, Wout, mode, subpix,
                       xpacked, stream);
}

// 1 when a GroupNorm whose only consumer is the plain 3x3 layer (N, Cin, H, W) -> Cout may hand its result over as MX
// operand records (eovae_gn_silu_fwd_packed_mx, xpacked = 2): the forward kernels read them whenever Cin % 32 == 0; the weight
// gradient must be the row-streaming kernel in its f16 form.  0: write format-1 records (eovae_gn_silu_fwd_packed).
int eovae_records_mx_ok(int N, int Cin, int H, int W, int Cout) {
  if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  if (!eovae_opt(EOVAE_OPT_C8_MX) || eovae_opt(EOVAE_OPT_C8_WGRAD_CORR)) return 0;
  WgCArgs a;
  a.part = reinterpret_cast<float*>(1); a.subpix = 0; a.mode = MODE_NORMAL; a.xpacked = 2; a.xamax = nullptr;
  a.N = N; a.Cin = Cin; a.Hin = H; a.Win = W; a.Cout = Cout; a.Hout = H; a.Wout = W;
  return eovae_c8_wgrad_rows_eligible(a) ? 1 : 0;
}

// bytes of scratch eovae_conv3x3_f16c8_wgrad_dw needs: one [9][128][64] fp32 tile per block of the launch
size_t eovae_conv3x3_f16c8_wgrad_dw_workspace(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  const int nslab = cdiv(Cin, 64) * cdiv(Cout, 128);
  int split = EOVAE_WGRAD_BLOCKS / nslab;
  if (split < 1) split = 1;
  return (size_t)nslab * split * 9 * 128 * 64 * sizeof(float);
}

// Weight gradient of a PLAIN 3x3 layer (mode 0, no sub-pixel form) straight into dw[Cout][Cin][3][3] (PyTorch layout,
// written, not accumulated): the blocks store partial tiles into ws, a second kernel folds them in a fixed order --
// no pre-zeroed accumulator, no float atomics (deterministic), no unpack step.  Other arguments as
// eovae_conv3x3_f16c8_wgrad; db (nullable) is still accumulated atomically into a pre-zeroed [Cout].
int eovae_conv3x3_f16c8_wgrad_dw(const float* x, const float* dy, float* dw, float* db, const unsigned* dyamax,
                                 const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                                 int xpacked, void* ws, void* stream) {
  EOVAE_CHECK_ARG(dw && ws, "conv3x3_f16c8_wgrad_dw: null pointer");
  return c8_wgrad_impl(x, dy, nullptr, (float*)ws, dw, db, dyamax, xamax, N, Cin, Hin, Win, Cout, Hout, Wout, MODE_NORMAL, 0,
                       xpacked, stream);
}

}  // extern "C"
