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
#include <type_traits>

namespace {

enum { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_ZINS = 3 };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct ConvCArgs {
  const float* x;
  const unsigned char* wp;   // f16c8-packed weights, 16 KiB per (tap, 32-ch chunk, 128-co block) + header
  const float* bias;
  const float* res;
  float* y;
  const unsigned* xamax;     // |x| amax as float bits (device), or null: x is used unscaled
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int nchunk, ncob;
  int mode;
  int tiles_x, tiles_y;
  int subpix, creal, cpb;    // as ConvBArgs (conv_bf16x3.hip)
  int xcd_remap;
};

__device__ __forceinline__ int src_index(int mode, int oy0, int ox0, int r, int c, int Hin, int Win) {
  if (mode == MODE_NORMAL) {
    const int vy = oy0 + r - 1, vx = ox0 + c - 1;
    return ((unsigned)vy < (unsigned)Hin && (unsigned)vx < (unsigned)Win) ? vy * Win + vx : -1;
  } else if (mode == MODE_UP) {
    const int vy = oy0 + r - 1, vx = ox0 + c - 1;
    return ((unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win))
               ? (vy >> 1) * Win + (vx >> 1)
               : -1;
  } else {  // MODE_ZINS: adjoint of stride 2 (zero-inserted dy)
    const int vy = oy0 + r - 2, vx = ox0 + c - 2;
    const bool ok = (unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win) && !((vy | vx) & 1);
    return ok ? (vy >> 1) * Win + (vx >> 1) : -1;
  }
}

// power of two 2^k as a float, k clamped to the normal range
__device__ __forceinline__ float pow2f(int k) {
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}
// binary exponent of a positive float given as bits (0 for zero / subnormal input)
__device__ __forceinline__ int expo_of_bits(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xff);
  return e == 0 ? -127 : e - 127;
}

template <int TH>
__global__ __launch_bounds__(512, 2) void conv3x3_f16c8_kernel(ConvCArgs a) {
  constexpr int NW = 8, TW = 32, BM = 128, KC = 32, NTHR = NW * 64;
  constexpr int XH = TH + 2, XW = TW + 2, NPX = XH * XW;
  constexpr int NI = TH / (NW / 2);             // pixel-row tiles per wave
  constexpr int XF16 = 4 * NPX * 16;            // f16 image: [octet][pixel][8 ch]
  constexpr int XSET = 8 * NPX * 16;            // + fp8 images [lo'|hi'][half][pixel][16 B]
  constexpr bool XDB = TH == 8;                 // second halo buffer where it fits
  constexpr int WTAP = 16384, WF16 = 8192;
  constexpr int OFF_W = (XDB ? 2 : 1) * XSET;
  constexpr int XTASKS = NPX * 4;               // (pixel, 8-channel octet)
  constexpr int XPT = (XTASKS + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_W + 4 * WTAP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  int cob;
  if (a.xcd_remap) {  // the ncob blocks sharing one halo tile take consecutive slots of ONE XCD
    const int xcd = b & 7, q = b >> 3;
    cob = q % a.ncob;
    b = (q / a.ncob) * 8 + xcd;
  } else {
    cob = b % a.ncob;
    b /= a.ncob;
  }
  const int tx = b % a.tiles_x;
  b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int wco = (wave & 1) * 64, wrow = (wave >> 1) * NI;
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const bool kgather = a.subpix == 2 || a.subpix == 3;  // K channels gathered at a parity
  const bool mshuffle = a.subpix == 1 || a.subpix == 4; // M rows pixel-shuffled on store
  const int xch = kgather ? a.creal : a.Cin;
  const float* xn = a.x + (size_t)n * xch * in_plane;
  const int par1 = mshuffle ? cob / a.cpb : 0;
  auto parity_of = [&](int chunk) { return kgather ? (chunk * KC) / a.creal : par1; };
  auto win0 = [&](int pb) {  // first active tap of an axis whose parity bit is pb (conv_bf16x3.hip)
    switch (a.subpix) {
      case 1: return pb;
      case 2: return 1 - pb;
      case 3: return 1;
      case 4: return pb;
      default: return 0;
    }
  };
  auto wincnt = [&](int pb) { return a.subpix == 0 ? 3 : (a.subpix >= 3 ? 2 - pb : 2); };
  auto ntaps_of = [&](int chunk) {
    const int par = parity_of(chunk);
    return wincnt(par >> 1) * wincnt(par & 1);
  };
  // t-th active tap of a chunk -> (ky, kx); t is clamped so that a dummy second tap re-reads a real one
  auto tap_of = [&](int chunk, int t, int& ky, int& kx) {
    const int par = parity_of(chunk);
    const int kxc = wincnt(par & 1), nt = wincnt(par >> 1) * kxc;
    t = t < nt ? t : nt - 1;
    const int q = kxc == 3 ? t / 3 : (kxc == 2 ? t >> 1 : t);
    ky = win0(par >> 1) + q;
    kx = win0(par & 1) + (t - q * kxc);
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
  const unsigned wbits = *reinterpret_cast<const unsigned*>(a.wp + (size_t)9 * a.nchunk * a.ncob * WTAP);
  const int e1 = wbits ? 7 - expo_of_bits(wbits) : 0;   // amax(w) * 2^e1 in [128, 256)
  auto e8m0 = [](int k) { k += 127; return k < 0 ? 0 : (k > 254 ? 254 : k); };
  // K-block 0 (lanes h = 0): xl' * wh' = 2^12 xl * 2^e1 w    K-block 1 (h = 1): xh' * wl' = x * 2^(e1+12) wl:
  // the same power of two for both blocks, so every lane passes the same scale bytes
  const int sc_a = e8m0(-e1 - 12);
  const int sc_b = e8m0(0);

  float xr[XPT][8];
  f32x4 wr[2][2];

  // chunk-invariant element offsets of this thread's halo tasks (loads are UNCONDITIONAL from a clamped
  // address, the validity mask is applied at conversion -- conv_bf16x3.hip explains why)
  unsigned xoff[XPT];
  bool xok[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int t = tid + i * NTHR;
    const int g = t / NPX, pix = t - g * NPX;
    const int rr = pix / XW, cc = pix - rr * XW;
    int si = -1;
    if (t < XTASKS) {
      if (kgather) {
        const int vy = oy0 + rr - 1, vx = ox0 + cc - 1;
        if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout) si = 2 * vy * a.Win + 2 * vx;
      } else {
        si = src_index(a.mode, oy0, ox0, rr, cc, a.Hin, a.Win);
      }
    }
    xok[i] = si >= 0;
    xoff[i] = (unsigned)((t < XTASKS ? g : 0) * 8) * (unsigned)in_plane + (unsigned)(si >= 0 ? si : 0);
  }
  auto load_x = [&](int c0) {
    int cbase = c0, poff = 0;
    if (kgather) {
      const int par = c0 / a.creal;
      cbase = c0 - par * a.creal;
      poff = (par >> 1) * a.Win + (par & 1);
    }
    const float* cb = xn + (size_t)cbase * in_plane + poff;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* cj = cb + (size_t)j * in_plane;
#pragma unroll
      for (int i = 0; i < XPT; ++i) xr[i][j] = cj[xoff[i]];
    }
  };
  auto store_x = [&](int xb) {
    unsigned char* xsm = smem + (XDB ? xb * XSET : 0);
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int t = tid + i * NTHR;
      if (t < XTASKS) {
        const int g = t / NPX, pix = t - g * NPX;
        f16x8 hi;
        float lo[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = xok[i] ? xr[i][j] * xs : 0.f;
          const _Float16 hh = (_Float16)v[j];
          hi[j] = hh;
          lo[j] = (v[j] - (float)hh) * 4096.f;
        }
        i32x2 l8, h8;
        l8[0] = __builtin_amdgcn_cvt_pk_bf8_f32(lo[0], lo[1], 0, false);
        l8[0] = __builtin_amdgcn_cvt_pk_bf8_f32(lo[2], lo[3], l8[0], true);
        l8[1] = __builtin_amdgcn_cvt_pk_bf8_f32(lo[4], lo[5], 0, false);
        l8[1] = __builtin_amdgcn_cvt_pk_bf8_f32(lo[6], lo[7], l8[1], true);
        h8[0] = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
        h8[0] = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h8[0], true);
        h8[1] = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], 0, false);
        h8[1] = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], h8[1], true);
        *reinterpret_cast<f16x8*>(xsm + g * (NPX * 16) + pix * 16) = hi;
        unsigned char* p8 = xsm + XF16 + ((g >> 1) * NPX + pix) * 16 + (g & 1) * 8;
        *reinterpret_cast<i32x2*>(p8) = l8;                      // part 0: lo'
        *reinterpret_cast<i32x2*>(p8 + 2 * NPX * 16) = h8;       // part 1: hi'
      }
    }
  };
  // weights of one stage = taps t0, t0+1 of a chunk (a missing second tap re-reads the first)
  auto load_w = [&](int chunk, int t0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      int ky, kx;
      tap_of(chunk, t0 + tt, ky, kx);
      const size_t tile = ((size_t)((ky * 3 + kx) * a.nchunk + chunk) * a.ncob + cob);
#pragma unroll
      for (int u = 0; u < 2; ++u)
        wr[tt][u] = *reinterpret_cast<const f32x4*>(a.wp + tile * WTAP + (tid + u * NTHR) * 16);
    }
  };
  auto store_w = [&](int slot0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *reinterpret_cast<f32x4*>(smem + OFF_W + (slot0 + tt) * WTAP + (tid + u * NTHR) * 16) = wr[tt][u];
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  load_x(0);
  load_w(0, 0);
  store_x(0);
  store_w(0);
  __syncthreads();

  // one tap: 2 co-tiles x NI pixel-row tiles x (2 f16 MFMAs + 1 scaled fp8 MFMA)
  auto compute_tap = [&](const unsigned char* wb, const unsigned char* xb, int ky, int kx)
      __attribute__((always_inline)) {
    const int arow = (wco + r) * 16;
    // --- pass 1: f16 main term.  Fragment of k-step s = octet 2s + h.
    {
      f16x8 ah[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          ah[i][s] = *reinterpret_cast<const f16x8*>(wb + (2 * s + h) * (BM * 16) + arow + i * 512);
      auto xfrag = [&](int j, f16x8 (&f)[2]) __attribute__((always_inline)) {
        const int pix = (wrow + j + ky) * XW + r + kx;
#pragma unroll
        for (int s = 0; s < 2; ++s)
          f[s] = *reinterpret_cast<const f16x8*>(xb + (2 * s + h) * (NPX * 16) + pix * 16);
      };
      f16x8 bh[2], nh[2];
      xfrag(0, bh);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (j + 1 < NI) xfrag(j + 1, nh);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][s], bh[s], acc[i][j], 0, 0, 0);
        bh[0] = nh[0];
        bh[1] = nh[1];
      }
    }
    // --- pass 2: both correction terms in one K = 64 block-scaled product.
    {
      i32x8 a8[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* p = wb + WF16 + (h * 2) * (BM * 16) + arow + i * 512;
        const i32x4 q0 = *reinterpret_cast<const i32x4*>(p);
        const i32x4 q1 = *reinterpret_cast<const i32x4*>(p + BM * 16);
        a8[i] = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      auto xfrag8 = [&](int j) __attribute__((always_inline)) {
        const int pix = (wrow + j + ky) * XW + r + kx;
        const unsigned char* p = xb + XF16 + ((h * 2) * NPX + pix) * 16;
        const i32x4 q0 = *reinterpret_cast<const i32x4*>(p);
        const i32x4 q1 = *reinterpret_cast<const i32x4*>(p + NPX * 16);
        return (i32x8)__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      i32x8 b8 = xfrag8(0), n8;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (j + 1 < NI) n8 = xfrag8(j + 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8, acc[i][j], 0 /* A: e4m3 */,
                                                                      1 /* B: e5m2 */, 0, sc_a, 0, sc_b);
        b8 = n8;
      }
    }
  };

  int q = 0;  // global stage counter: ring half (q & 1) holds the stage being multiplied
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    const int nt = ntaps_of(chunk), ns = (nt + 1) >> 1;
    const unsigned char* xb = smem + (XDB ? (chunk & 1) * XSET : 0);
    for (int s = 0; s < ns; ++s, ++q) {
      const bool last_stage = s == ns - 1;
      const bool has_next = !(last_stage && chunk + 1 == a.nchunk);
      if (has_next) {
        if (last_stage) load_w(chunk + 1, 0); else load_w(chunk, 2 * s + 2);
      }
      if (s == 0 && chunk + 1 < a.nchunk) load_x((chunk + 1) * KC);

      const unsigned char* wb = smem + OFF_W + (q & 1) * 2 * WTAP;
      {
        int ky, kx;
        tap_of(chunk, 2 * s, ky, kx);
        compute_tap(wb, xb, ky, kx);
        if (2 * s + 1 < nt) {  // uniform
          tap_of(chunk, 2 * s + 1, ky, kx);
          compute_tap(wb + WTAP, xb, ky, kx);
        }
      }

      if (has_next) store_w(((q + 1) & 1) * 2);   // other ring half: last read a barrier ago
      if (last_stage && chunk + 1 < a.nchunk) {
        if (!XDB) __syncthreads();                // one buffer: every wave is done with this chunk's tile
        store_x((chunk + 1) & 1);
      }
      __syncthreads();
    }
  }

  // Epilogue: as conv_bf16x3.hip (batched unconditional bias / residual loads, plain stores on full
  // tiles), plus the exact power-of-two un-scaling of a scaled input.
  const int ox = ox0 + r;
  const bool okx = ox < a.Wout;
  const int chans = mshuffle ? a.creal : a.Cout;
  const int co_blk = (mshuffle ? (cob - par1 * a.cpb) : cob) * BM;
  const int co_base = co_blk + wco + 4 * h;
  const int py = par1 >> 1, px = par1 & 1;
  const size_t out_plane = mshuffle ? (size_t)a.Hout * a.Wout * 4 : (size_t)a.Hout * a.Wout;
  float* yn = a.y + (size_t)n * chans * out_plane;
  const float* rn = a.res ? a.res + (size_t)n * chans * out_plane : nullptr;
  unsigned rowoff[NI];
  bool rowok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int oy = oy0 + wrow + j;
    rowok[j] = okx && oy < a.Hout;
    const int oyc = oy < a.Hout ? oy : a.Hout - 1, oxc = okx ? ox : a.Wout - 1;
    rowoff[j] = mshuffle ? (unsigned)((2 * oyc + py) * (2 * a.Wout) + 2 * oxc + px)
                         : (unsigned)(oyc * a.Wout + oxc);
  }
  auto epilogue = [&](auto full_t) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float bias4[4], rv[4][NI];
        int coc[4];
        bool cok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int co = co_base + i * 32 + k + 8 * qq;
          cok[k] = FULL || co < chans;
          coc[k] = cok[k] ? co : chans - 1;
          bias4[k] = a.bias ? a.bias[coc[k]] : 0.f;
        }
        if (rn) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NI; ++j) rv[k][j] = rn[(size_t)coc[k] * out_plane + rowoff[j]];
        }
        float v[4][NI];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            v[k][j] = acc[i][j][4 * qq + k] * xinv + bias4[k];
            if (rn) v[k][j] += rv[k][j];
            if (!FULL) asm volatile("" : "+v"(v[k][j]));
          }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            if (FULL || (cok[k] && rowok[j])) yn[(size_t)coc[k] * out_plane + rowoff[j]] = v[k][j];
      }
    }
  };
  const bool full_tile = oy0 + TH <= a.Hout && ox0 + TW <= a.Wout && co_blk + BM <= chans;
  if (full_tile) epilogue(std::true_type{}); else epilogue(std::false_type{});
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

// w[Cout][Cin][9] fp32 -> per (tap, 32-ch chunk, 128-co block) one 16 KiB tile:
//   [0, 8 KiB)     f16  [octet 0..3][128 m][8 k]
//   [8 KiB, 16 KiB) e4m3 [wh' | wl'][half 0..1][128 m][16 k]
// flip = 0: k = ci, m = co;  flip = 1 (data gradient): k = co, m = ci, taps reversed.
// One thread = 4 consecutive k of one (tile, m): one dword of each fp8 image, 8 bytes of the f16 image.
__global__ void pack_f16c8_kernel(const float* __restrict__ w, unsigned char* __restrict__ wp, int Cout, int Cin,
                                  int nchunk, int ncob, int flip, float scale, const unsigned* __restrict__ hdr) {
  const unsigned wbits = *hdr;   // amax(|w|) bits of the UNSCALED weights; `scale` is a power of two
  const float amax = __builtin_bit_cast(float, wbits) * fabsf(scale);
  const int e1 = amax > 0.f ? 7 - expo_of_bits(__builtin_bit_cast(unsigned, amax)) : 0;
  const float s_hi = pow2f(e1), s_lo = pow2f(e1 + 12);   // |wl| <= 2^-11 |w|: wl * s_lo stays under 256
  const size_t total = (size_t)9 * nchunk * ncob * 128 * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
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
    unsigned char* tile = wp + ((size_t)(tap * nchunk + chunk) * ncob + cob) * 16384;
    const int kk = k4 * 4;   // first k of this thread inside the chunk
    *reinterpret_cast<f16x4*>(tile + (kk >> 3) * 2048 + m_ * 16 + (kk & 7) * 2) = hh;
    unsigned char* p8 = tile + 8192 + ((kk >> 4) * 128 + m_) * 16 + (kk & 15);
    *reinterpret_cast<int*>(p8) = w_hi;
    *reinterpret_cast<int*>(p8 + 2 * 128 * 16) = w_lo;
  }
}

}  // namespace

extern "C" {

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

// bytes of the f16c8-packed weight buffer of a 3x3 layer (tiles + 16-byte header with amax(|w|) bits)
size_t eovae_conv3x3_f16c8_packed_bytes(int Cout, int Cin, int flip) {
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  return (size_t)9 * cdiv(k, 32) * cdiv(m, 128) * 16384 + 16;
}

int eovae_conv3x3_f16c8_pack(const float* w, void* wp, int Cout, int Cin, int flip, float scale, void* stream) {
  EOVAE_CHECK_ARG(w && wp && Cout > 0 && Cin > 0, "conv3x3_f16c8_pack: bad argument");
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  const int nchunk = cdiv(k, 32), ncob = cdiv(m, 128);
  unsigned char* base = (unsigned char*)wp;
  unsigned* hdr = (unsigned*)(base + (size_t)9 * nchunk * ncob * 16384);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(hdr, 0, 16, st) != hipSuccess) {
    eovae_set_error("conv3x3_f16c8_pack: memset failed");
    return EOVAE_ERR_LAUNCH;
  }
  const size_t nw = (size_t)Cout * Cin * 9;
  size_t ab = (nw / 4 + 255) / 256;
  if (ab > 1024) ab = 1024;
  if (ab < 1) ab = 1;
  hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)ab), dim3(256), 0, st, w, nw, hdr);
  const size_t total = (size_t)9 * nchunk * ncob * 128 * 8;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_f16c8_kernel, dim3(blocks), dim3(256), 0, st, w, base, Cout, Cin, nchunk, ncob, flip,
                     scale, (const unsigned*)hdr);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_pack");
  return EOVAE_OK;
}

// 3x3 stride-1 conv / data gradient with f16c8 operands.  Arguments as eovae_conv3x3_bf16_fwd; the GEMM K
// channel count must be a multiple of 32.  xamax: device pointer to the float bits of max |x|
// (eovae_absmax_bits) when x may sit outside f16's range (gradients), else NULL.
int eovae_conv3x3_f16c8_fwd(const float* x, const void* wp, const float* bias, const float* residual, float* y,
                            const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout,
                            int Wout, int mode, int subpix, void* stream) {
  EOVAE_CHECK_ARG(x && wp && y, "conv3x3_f16c8_fwd: null pointer");
  EOVAE_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv3x3_f16c8_fwd: bad shape");
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || mode == MODE_UP || mode == MODE_ZINS,
                  "conv3x3_f16c8_fwd: mode %d unsupported", mode);
  EOVAE_CHECK_ARG(subpix >= 0 && subpix <= 4, "conv3x3_f16c8_fwd: subpix %d", subpix);
  ConvCArgs a;
  a.x = x; a.wp = (const unsigned char*)wp; a.bias = bias; a.res = residual; a.y = y; a.xamax = xamax;
  a.N = N; a.Hin = Hin; a.Win = Win; a.mode = mode; a.subpix = subpix; a.creal = 0; a.cpb = 1;
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
  Cin = a.Cin; Cout = a.Cout; Hout = a.Hout; Wout = a.Wout;
  a.nchunk = Cin / 32; a.ncob = cdiv(Cout, 128);
  a.tiles_x = cdiv(Wout, 32);
  hipStream_t st = (hipStream_t)stream;
  auto xcd_ok = [&](long long tiles) { return (a.ncob > 1 && tiles % 8 == 0) ? 1 : 0; };
  const long long blocks16 = (long long)N * cdiv(Hout, 16) * a.tiles_x * a.ncob;
  if (blocks16 >= 512) {   // 16-row tiles while they still give every CU two blocks
    a.tiles_y = cdiv(Hout, 16);
    a.xcd_remap = xcd_ok((long long)N * a.tiles_y * a.tiles_x);
    hipLaunchKernelGGL((conv3x3_f16c8_kernel<16>), dim3((unsigned)(N * a.tiles_y * a.tiles_x * a.ncob)), dim3(512),
                       0, st, a);
  } else {
    a.tiles_y = cdiv(Hout, 8);
    a.xcd_remap = xcd_ok((long long)N * a.tiles_y * a.tiles_x);
    hipLaunchKernelGGL((conv3x3_f16c8_kernel<8>), dim3((unsigned)(N * a.tiles_y * a.tiles_x * a.ncob)), dim3(512),
                       0, st, a);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_fwd");
  return EOVAE_OK;
}

}  // extern "C"
