// Weight gradient of the wavelength-conditioned first / last 3x3 convolutions (dynamic_conv.py:186-206, 231-246 of the
// reference: DynamicConv 3..13 -> 128 channels, DynamicConv_decoder 128 -> 3..13), for the modalities with at most FOUR
// channels on the thin side (S2RGB 3, S1RTC 2):
//     dW[co][ci][ky][kx] = sum_{n, y, x} dy[n][co][y][x] * x[n][ci][y + ky - 1][x + kx - 1]
// On the matrix-core weight-gradient kernels these layers cost what a 128 -> 128 layer costs (a 128-row tile with 3
// valid rows: 0.75 ms and 0.48 ms per step at 10-15 TFLOP/s algorithmic), although the work is 7 GFLOP and one pass
// over the 128-channel tensor (537 MB at B = 16, 256^2): an HBM-bound problem.  So: plain fp32 FMAs.
//   L = the tensor with many channels (x of the 128 -> 3 layer -- fp32 or GroupNorm operand records --, dy of the
//       3 -> 128 layer), S = the thin one (<= 4 channels).
//   A thread owns one pixel column and two channels of L, walks ALL rows and keeps acc[2][S channels][9 taps] and the
//   3 x 3 window of S in registers: per pixel 2 global values of L, 3 x CS values of S (L2), 2 x CS x 9 FMAs.
//   Partial sums leave the block once, into part[image][channel pair][74]; a second tiny kernel adds them in a fixed order
//   (deterministic, no atomics) and writes dW in the PyTorch layout [Cout][Cin][3][3] (+ the bias gradient).
#include "eovae_common.h"

namespace {

constexpr int TH_W = 256;        // widest row: four waves x 64 columns
constexpr int TH_CL = 2;         // L channels per thread
constexpr int TH_NP = TH_CL * 36 + TH_CL;   // partial sums per block: tap sums + sums of L

struct ThinArgs {
  const float* L;     // [N][CL][H][W] fp32, or operand records [N][CL/8][H][W] x 32 B
  const float* S;     // [N][CS][H][W] fp32
  float* part;        // [N][CL / TH_CL][TH_NP]: TH_CL x 4 x 9 tap sums + TH_CL sums of L (bias gradient when L = dy)
  int N, CL, CS, H, W, sign;   // sign: +1 S is sampled at p + (ky-1, kx-1) (S = x), -1 at p - (...) (S = dy)
};

__device__ __forceinline__ float e5m2_f(unsigned b) {   // an e5m2 byte is the upper byte of the f16 with the same value
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(b << 8));
}

// One workgroup = one image x one pair of L channels; wave w owns columns 64 w .. 64 w + 63, a thread ONE column, and
// walks all rows top to bottom with
//   acc[TH_CL][CS][9 taps]  and the 3 x 3 window of S around its pixel
// in registers: per pixel TH_CL global values of L, 3 x CS values of S for the window's new bottom row (plain loads:
// S is a few MB and lives in L2; out-of-image positions read a clamped address and are zeroed by a select, so no
// load sits under a branch), TH_CL x CS x 9 FMAs.  No LDS tile, no barrier in the loop, one reduction per workgroup.
// (Earlier forms, both SLOWER than the matrix-core kernel they were to replace: 4 L channels + nine LDS reads per
// pixel = 256 VGPRs, one wave per SIMD: 1.1-1.3 ms per launch; 2 channels + an LDS tile of S refilled every 8 rows and
// a reduction per 32-row band: 0.73-0.99 ms -- the refills and reductions were a third of the instructions.)
template <bool LREC, int CS>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ThinArgs a) {
  __shared__ float red[4][TH_NP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ngrp = a.CL / TH_CL;
  const int grp = blockIdx.x % ngrp, n = blockIdx.x / ngrp;
  const size_t plane = (size_t)a.H * a.W;
  const int x = tid;
  const bool col_ok = x < a.W;
  const int c0 = TH_CL * grp;   // first L channel of this block
  float acc[TH_CL][CS][9];
  float lsum[TH_CL];
#pragma unroll
  for (int c = 0; c < TH_CL; ++c) {
    lsum[c] = 0.f;
#pragma unroll
    for (int s = 0; s < CS; ++s)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[c][s][t] = 0.f;
  }
  // column offsets of the window, clamped into the row; validity as multipliers
  const int xc = col_ok ? x : 0;
  const int xo[3] = {xc > 0 ? xc - 1 : 0, xc, xc + 1 < a.W ? xc + 1 : xc};
  const float xm[3] = {(col_ok && x > 0) ? 1.f : 0.f, col_ok ? 1.f : 0.f, (col_ok && x + 1 < a.W) ? 1.f : 0.f};
  const float* Sn = a.S + (size_t)n * CS * plane;
  auto srow = [&](int y, float (&w)[CS][3]) __attribute__((always_inline)) {   // row y of S around this column (0 off-image)
    const bool rok = (unsigned)y < (unsigned)a.H;
    const float* p = Sn + (size_t)(rok ? y : 0) * a.W;
    const float rm = rok ? 1.f : 0.f;
#pragma unroll
    for (int s_ = 0; s_ < CS; ++s_)
#pragma unroll
      for (int k = 0; k < 3; ++k) w[s_][k] = p[(size_t)s_ * plane + xo[k]] * (rm * xm[k]);
  };
  auto lrow = [&](int y, float (&lv)[TH_CL]) __attribute__((always_inline)) {   // L at (y, x) for the block's channel pair
    const bool rok = y < a.H;
    const size_t off = (size_t)(rok ? y : 0) * a.W + xc;
    const float m = rok ? xm[1] : 0.f;
    if (LREC) {   // channels c0, c0 + 1 of octet c0 >> 3: x = f16 + e5m2(lo') * 2^-12
      const char* rec = reinterpret_cast<const char*>(a.L) + (((size_t)n * (a.CL / 8) + (c0 >> 3)) * plane + off) * 32;
      const unsigned h = *reinterpret_cast<const unsigned*>(rec + 2 * (c0 & 7));
      const unsigned lo = *reinterpret_cast<const unsigned short*>(rec + 16 + (c0 & 7));
      lv[0] = ((float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu)) + e5m2_f(lo & 0xffu) * 0x1p-12f) * m;
      lv[1] = ((float)__builtin_bit_cast(_Float16, (unsigned short)(h >> 16)) + e5m2_f(lo >> 8) * 0x1p-12f) * m;
    } else {
      const float* p = a.L + ((size_t)n * a.CL + c0) * plane + off;
#pragma unroll
      for (int c = 0; c < TH_CL; ++c) lv[c] = p[(size_t)c * plane] * m;
    }
  };
  // rows y - 1, y, y + 1 of S and row y of L are in registers when iteration y starts; it first requests row y + 2 of
  // S and row y + 1 of L, used by the NEXT iteration: their latency hides behind this row's FMAs and the SIMD's other
  // three waves (120 VGPRs = four waves per SIMD = the whole grid resident at B = 16: 305 / 374 us per launch, about one
  // memory latency per row; a second row of look-ahead costs 26 registers, i.e. the fourth wave: 373 / 548 us)
  float win[3][CS][3];   // [row above | this | below][S channel][column x - 1 + k]
  float nxt[CS][3], lv[TH_CL], lnx[TH_CL];
#pragma unroll
  for (int s_ = 0; s_ < CS; ++s_)
#pragma unroll
    for (int k = 0; k < 3; ++k) win[0][s_][k] = 0.f;   // row -1
  srow(0, win[1]);
  srow(1, win[2]);
  lrow(0, lv);
  if (wave * 64 < a.W) {   // uniform per wave
    for (int y = 0; y < a.H; ++y) {
      srow(y + 2, nxt);
      lrow(y + 1, lnx);
#pragma unroll
      for (int c = 0; c < TH_CL; ++c) lsum[c] += lv[c];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          // S = x: sample at (y + ky - 1, x + kx - 1); S = dy: the output pixel that reads x[y][x] through tap (ky, kx)
          // is (y - ky + 1, x - kx + 1): the mirrored window position.  a.sign is uniform.
          const int t = ky * 3 + kx;
#pragma unroll
          for (int s_ = 0; s_ < CS; ++s_) {
            const float sv = a.sign > 0 ? win[ky][s_][kx] : win[2 - ky][s_][2 - kx];
#pragma unroll
            for (int c = 0; c < TH_CL; ++c) acc[c][s_][t] = __builtin_fmaf(lv[c], sv, acc[c][s_][t]);
          }
        }
#pragma unroll
      for (int s_ = 0; s_ < CS; ++s_)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          win[0][s_][k] = win[1][s_][k];
          win[1][s_][k] = win[2][s_][k];
          win[2][s_][k] = nxt[s_][k];
        }
#pragma unroll
      for (int c = 0; c < TH_CL; ++c) lv[c] = lnx[c];
    }
  }
  // block sums: wave butterflies, then the four waves through LDS (fixed order)
#pragma unroll
  for (int c = 0; c < TH_CL; ++c)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = s_ < CS ? wave_sum(acc[c][s_ < CS ? s_ : 0][t]) : 0.f;
        if (lane == 0) red[wave][(c * 4 + s_) * 9 + t] = v;
      }
#pragma unroll
  for (int c = 0; c < TH_CL; ++c) {
    const float v = wave_sum(lsum[c]);
    if (lane == 0) red[wave][TH_CL * 36 + c] = v;
  }
  __syncthreads();
  if (tid < TH_NP) a.part[(size_t)blockIdx.x * TH_NP + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// dW (PyTorch layout [Cout][Cin][3][3]) and the bias gradient from the per-block partial sums, in a fixed order.
// One thread per (L channel, S channel, tap).  l_is_dy: L = dy (the 3 -> 128 layer: dW[cl][cs], db[cl] = sum L);
// else L = x (the 128 -> 3 layer: dW[cs][cl], db[cs] = sum S is not available here -> see the host).
__global__ __launch_bounds__(256) void thin_fold_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                        float* __restrict__ db, int N, int CL, int CS, int l_is_dy) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int ngrp = CL / TH_CL;
  if (i < CL * CS * 9) {
    const int t = i % 9, s = (i / 9) % CS, cl = i / (9 * CS);
    const int grp = cl / TH_CL, c = cl % TH_CL;
    double v = 0;
    for (int n = 0; n < N; ++n) v += (double)part[((size_t)n * ngrp + grp) * TH_NP + (c * 4 + s) * 9 + t];
    if (l_is_dy) dw[((size_t)cl * CS + s) * 9 + t] = (float)v;     // [Cout = CL][Cin = CS]
    else dw[((size_t)s * CL + cl) * 9 + t] = (float)v;             // [Cout = CS][Cin = CL]
  }
  if (db && l_is_dy && i < CL) {
    const int grp = i / TH_CL, c = i % TH_CL;
    double v = 0;
    for (int n = 0; n < N; ++n) v += (double)part[((size_t)n * ngrp + grp) * TH_NP + TH_CL * 36 + c];
    db[i] = (float)v;
  }
}

}  // namespace

extern "C" {

// bytes of scratch eovae_conv3x3_thin_wgrad needs (per-block partial sums)
size_t eovae_conv3x3_thin_wgrad_workspace(int N, int Cin, int H, int W, int Cout) {
  const int CL = Cin > Cout ? Cin : Cout;
  if (N <= 0 || CL <= 0 || H <= 0) return 0;
  (void)H;
  return (size_t)N * (CL / TH_CL) * TH_NP * sizeof(float);
}

// 1 when the shape is one this kernel takes: plain 3x3, stride 1, one side with at most 4 channels, the other a
// multiple of 8 (records) / 4 (fp32), rows of at most 256 pixels
int eovae_conv3x3_thin_wgrad_eligible(int N, int Cin, int H, int W, int Cout, int xpacked) {
  if (N <= 0 || H <= 0 || W <= 0 || W > TH_W) return 0;
  const int CS = Cin < Cout ? Cin : Cout, CL = Cin < Cout ? Cout : Cin;
  if (CS < 1 || CS > 4 || CL < 32) return 0;
  if (xpacked) return (Cin > Cout && Cin % 8 == 0) ? 1 : 0;   // records can only be the many-channel x
  return CL % 4 == 0 ? 1 : 0;
}

// dw[Cout][Cin][3][3] (written, not accumulated), db[Cout] (nullable; written only for the few-input-channel layer,
// whose L tensor is dy -- for the few-output-channel layer the caller keeps using eovae_bias_grad).
// x: [N][Cin][H][W] fp32, or operand records when xpacked = 1 (then Cin > Cout).  dy: [N][Cout][H][W] fp32.
int eovae_conv3x3_thin_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int Cin, int H, int W,
                             int Cout, int xpacked, void* ws, void* stream) {
  EOVAE_CHECK_ARG(x && dy && dw && ws, "conv3x3_thin_wgrad: null pointer");
  EOVAE_CHECK_ARG(xpacked == 0 || xpacked == 1, "conv3x3_thin_wgrad: xpacked %d (0 fp32, 1 operand records; MX records are not read here)", xpacked);
  EOVAE_CHECK_ARG(eovae_conv3x3_thin_wgrad_eligible(N, Cin, H, W, Cout, xpacked),
                  "conv3x3_thin_wgrad: shape not supported (N %d Cin %d H %d W %d Cout %d xpacked %d)", N, Cin, H, W, Cout,
                  xpacked);
  hipStream_t st = (hipStream_t)stream;
  ThinArgs a;
  const bool l_is_dy = Cout > Cin;
  a.L = l_is_dy ? dy : x;
  a.S = l_is_dy ? x : dy;
  a.part = (float*)ws;
  a.N = N; a.CL = l_is_dy ? Cout : Cin; a.CS = l_is_dy ? Cin : Cout; a.H = H; a.W = W;
  a.sign = l_is_dy ? 1 : -1;
  const int blocks = N * (a.CL / TH_CL);
#define EOVAE_THIN_LAUNCH(REC)                                                                                       \
  switch (a.CS) {                                                                                                     \
    case 1: hipLaunchKernelGGL((thin_wgrad_kernel<REC, 1>), dim3(blocks), dim3(256), 0, st, a); break;                 \
    case 2: hipLaunchKernelGGL((thin_wgrad_kernel<REC, 2>), dim3(blocks), dim3(256), 0, st, a); break;                 \
    case 3: hipLaunchKernelGGL((thin_wgrad_kernel<REC, 3>), dim3(blocks), dim3(256), 0, st, a); break;                 \
    default: hipLaunchKernelGGL((thin_wgrad_kernel<REC, 4>), dim3(blocks), dim3(256), 0, st, a); break;                \
  }
  if (xpacked) { EOVAE_THIN_LAUNCH(true) } else { EOVAE_THIN_LAUNCH(false) }
#undef EOVAE_THIN_LAUNCH
  hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(a.CL * a.CS * 9, 256)), dim3(256), 0, st, a.part, dw, db, N, a.CL, a.CS,
                     l_is_dy ? 1 : 0);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_thin_wgrad");
  return EOVAE_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Forward of the LAST layer (many -> at most 4 channels; reference dynamic_conv.py:231-246, F.conv2d(img_feat, weight,
// bias, padding=1)):  y[n][co][p] = b[co] + sum_{ci, tap} w[co][ci][tap] * x[n][ci][p + tap]
// On the 8-wave matrix-core kernel this layer is a 128-row tile with 3 valid rows (0.54 ms per step, 13 TFLOP/s); the
// work is 7 GFLOP over one pass of x (537 MB): HBM-bound.  fp32 FMAs:
//   workgroup = image x 16 output rows x 64 columns; per channel octet the halo tile (18 x 66 pixels) is decoded ONCE
//   into LDS as fp32 [row][column][8 channels] (records: f16 + e5m2(lo') * 2^-12; fp32 input: eight planes), then a
//   thread (column c, row quartet q) takes its six halo rows x three columns x 8 channels into registers (36 16-byte
//   LDS reads) and runs 8 x 9 x CO x 4 FMAs whose weights are wave-uniform (scalar loads).  y accumulates in registers
//   across the octets and is stored once.
namespace {

constexpr int TF_ROWS = 16, TF_COLS = 64;

struct ThinFwdArgs {
  const float* x;      // [N][Cin][H][W] fp32, or operand records
  const float* w;      // [Cout][Cin][3][3]
  const float* bias;   // [Cout] or null
  float* y;            // [N][Cout][H][W]
  int N, Cin, Cout, H, W, tiles_x, tiles_y;
};

template <bool XREC, int CO>
__global__ __launch_bounds__(256) void thin_fwd_kernel(ThinFwdArgs a) {
  __shared__ float4 tile[(TF_ROWS + 2) * (TF_COLS + 2) * 2];   // [row][column][2 x float4 = 8 channels]
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int y0 = ty * TF_ROWS, x0 = tx * TF_COLS;
  const size_t plane = (size_t)a.H * a.W;
  const int col = tid & 63, rq = tid >> 6;           // column of the tile, row quartet (one wave each)
  float acc[4][CO];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[r][co] = 0.f;

  for (int oct = 0; oct < a.Cin / 8; ++oct) {
    __syncthreads();   // the previous octet's tile has been read
    for (int i = tid; i < (TF_ROWS + 2) * (TF_COLS + 2); i += 256) {
      const int rr = i / (TF_COLS + 2), cc = i - rr * (TF_COLS + 2);
      const int yy = y0 - 1 + rr, xx = x0 - 1 + cc;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
        const size_t pix = (size_t)yy * a.W + xx;
        if (XREC) {
          const char* rec = reinterpret_cast<const char*>(a.x) + (((size_t)n * (a.Cin / 8) + oct) * plane + pix) * 32;
          const uint4 h = *reinterpret_cast<const uint4*>(rec);
          const uint2 lo = *reinterpret_cast<const uint2*>(rec + 16);
          auto dec = [](unsigned hw, unsigned lb) __attribute__((always_inline)) {
            return (float)__builtin_bit_cast(_Float16, (unsigned short)hw) + e5m2_f(lb) * 0x1p-12f;
          };
          v0.x = dec(h.x & 0xffffu, lo.x & 0xffu);         v0.y = dec(h.x >> 16, (lo.x >> 8) & 0xffu);
          v0.z = dec(h.y & 0xffffu, (lo.x >> 16) & 0xffu); v0.w = dec(h.y >> 16, lo.x >> 24);
          v1.x = dec(h.z & 0xffffu, lo.y & 0xffu);         v1.y = dec(h.z >> 16, (lo.y >> 8) & 0xffu);
          v1.z = dec(h.w & 0xffffu, (lo.y >> 16) & 0xffu); v1.w = dec(h.w >> 16, lo.y >> 24);
        } else {
          const float* p = a.x + ((size_t)n * a.Cin + 8 * oct) * plane + pix;
          v0.x = p[0]; v0.y = p[plane]; v0.z = p[2 * plane]; v0.w = p[3 * plane];
          v1.x = p[4 * plane]; v1.y = p[5 * plane]; v1.z = p[6 * plane]; v1.w = p[7 * plane];
        }
      }
      tile[2 * i] = v0;
      tile[2 * i + 1] = v1;
    }
    __syncthreads();
    // the quartet's six halo rows x three columns x 8 channels in registers: every weight (wave-uniform: nine scalar
    // registers per (output channel, input channel)) then feeds FOUR FMAs, one per output row.  (First form: a sliding
    // 3-row window and all 216 weights of the octet live across the row loop = 225 spilled scalar registers, a
    // v_readlane per FMA: 340 us per launch.)
    float win[6][3][8];
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int e = 2 * ((4 * rq + rr) * (TF_COLS + 2) + col + k);
        const float4 u0 = tile[e], u1 = tile[e + 1];
        win[rr][k][0] = u0.x; win[rr][k][1] = u0.y; win[rr][k][2] = u0.z; win[rr][k][3] = u0.w;
        win[rr][k][4] = u1.x; win[rr][k][5] = u1.y; win[rr][k][6] = u1.z; win[rr][k][7] = u1.w;
      }
    const float* wo = a.w + (size_t)oct * 8 * 9;   // w[co][8 oct + j][tap] = wo[(co * Cin + j) * 9 + tap]
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // nine wave-uniform weights: scalar loads.  (Through LDS broadcast reads instead, the compiler hoisted all 216
        // of the octet into vector registers: 256 VGPRs, one wave per SIMD, 476 us.)
        const float* wp = wo + ((size_t)co * a.Cin + j) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float wv = wp[ky * 3 + kx];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r][co] = __builtin_fmaf(wv, win[r + ky][kx][j], acc[r][co]);
          }
      }
  }
  const int xo = x0 + col;
  if (xo < a.W) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yo = y0 + 4 * rq + r;
      if (yo < a.H) {
#pragma unroll
        for (int co = 0; co < CO; ++co)
          a.y[((size_t)n * a.Cout + co) * plane + (size_t)yo * a.W + xo] = acc[r][co] + (a.bias ? a.bias[co] : 0.f);
      }
    }
  }
}

}  // namespace

extern "C" {

// 1 when eovae_conv3x3_thin_fwd takes the shape: plain 3x3, stride 1, at most 4 output channels, Cin a multiple of 8
int eovae_conv3x3_thin_fwd_eligible(int N, int Cin, int H, int W, int Cout) {
  return (N > 0 && H > 0 && W > 0 && Cout >= 1 && Cout <= 4 && Cin >= 32 && Cin % 8 == 0) ? 1 : 0;
}

// y[N][Cout][H][W] = conv3x3(x, w[Cout][Cin][3][3]) + bias (nullable); x fp32 or operand records (xpacked = 1)
int eovae_conv3x3_thin_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                           int Cout, int xpacked, void* stream) {
  EOVAE_CHECK_ARG(x && w && y, "conv3x3_thin_fwd: null pointer");
  EOVAE_CHECK_ARG(xpacked == 0 || xpacked == 1, "conv3x3_thin_fwd: xpacked %d (0 fp32, 1 operand records; MX records are not read here)", xpacked);
  EOVAE_CHECK_ARG(eovae_conv3x3_thin_fwd_eligible(N, Cin, H, W, Cout),
                  "conv3x3_thin_fwd: shape not supported (N %d Cin %d H %d W %d Cout %d)", N, Cin, H, W, Cout);
  hipStream_t st = (hipStream_t)stream;
  ThinFwdArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.tiles_x = cdiv(W, TF_COLS); a.tiles_y = cdiv(H, TF_ROWS);
  const int blocks = N * a.tiles_x * a.tiles_y;
#define EOVAE_THIN_FWD(REC)                                                                                   \
  switch (Cout) {                                                                                              \
    case 1: hipLaunchKernelGGL((thin_fwd_kernel<REC, 1>), dim3(blocks), dim3(256), 0, st, a); break;            \
    case 2: hipLaunchKernelGGL((thin_fwd_kernel<REC, 2>), dim3(blocks), dim3(256), 0, st, a); break;            \
    case 3: hipLaunchKernelGGL((thin_fwd_kernel<REC, 3>), dim3(blocks), dim3(256), 0, st, a); break;            \
    default: hipLaunchKernelGGL((thin_fwd_kernel<REC, 4>), dim3(blocks), dim3(256), 0, st, a); break;           \
  }
  if (xpacked) { EOVAE_THIN_FWD(true) } else { EOVAE_THIN_FWD(false) }
#undef EOVAE_THIN_FWD
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_thin_fwd");
  return EOVAE_OK;
}

}  // extern "C"
