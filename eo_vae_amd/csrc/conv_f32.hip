// Implicit-GEMM 2-D convolution on the gfx950 fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 256 flop/clk/CU).
//
// Replaces the ATen call sites of the reference's hot path:
//   nn.Conv2d 3x3 s1 p1            eo_vae/models/modules/layers.py:64-66,81-83; model.py:162,260
//   Upsample  (nearest x2 + conv)  layers.py:47-50     -> MODE_UP   (upsample folded into the tile load)
//   Downsample (pad(0,1,0,1)+s2)   layers.py:33-37     -> MODE_DOWN (asymmetric zero pad = bounds check)
//   nn.Conv2d 1x1                  layers.py:84-87,123-126; model.py:165,236
//   F.conv2d with generated weight dynamic_conv.py:527-533, 699-705
// and their autograd adjoints (data-gradient = same kernel on flipped/transposed packed
// weights, MODE_ZINS for the stride-2 layer; weight-gradient = conv_wgrad_kernel).
//
// GEMM view (NCHW, fp32):   Y[co][pix] = sum_tap sum_ci  Wp[tap][ci][co] * Xv[ci][pix @ tap]
//   M = output channels (block tile 128), N = output pixels (block tile 8 rows x 32 cols),
//   K = input channels x taps, streamed in chunks of KC channels.
// Per chunk the block stages ONE halo tile of the input (all 9 taps re-use it from LDS) and
// the KC x taps x 128 slice of the packed weights; 4 waves each own a 64(co) x 128(pix)
// sub-tile = 2 x 4 MFMA tiles (128 accumulator registers).  Global loads for chunk c+1 are
// issued into registers before the MFMAs of chunk c (issue-early / write-late staging).

#include "eovae_common.h"

namespace {

enum { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_ZINS = 3 };

struct ConvArgs {
  const float* x;
  const float* wp;    // packed [taps][CinP][CoutP]
  const float* bias;  // [Cout] or null
  const float* res;   // residual, same shape as y, or null
  float* y;
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int CinP, CoutP;
  int mode;
  int tiles_x, tiles_y, ncob;
  float out_scale;
};

// virtual-coordinate -> source-index mapping shared by forward / wgrad loaders.
// (r, c) are tile-local coordinates in the staged tile whose origin corresponds to output
// pixel (oy0, ox0); returns -1 for a zero (padding / odd zero-insert position).
template <int KS, int S>
__device__ __forceinline__ int src_index(int mode, int oy0, int ox0, int r, int c, int Hin, int Win) {
  if (S == 2) {  // MODE_DOWN: iy = 2*oy + ky, zero beyond the bottom/right edge
    int iy = oy0 * 2 + r, ix = ox0 * 2 + c;
    return (iy < Hin && ix < Win) ? iy * Win + ix : -1;
  }
  constexpr int PAD = (KS - 1) / 2;
  if (mode == MODE_NORMAL) {
    int vy = oy0 + r - PAD, vx = ox0 + c - PAD;
    return ((unsigned)vy < (unsigned)Hin && (unsigned)vx < (unsigned)Win) ? vy * Win + vx : -1;
  } else if (mode == MODE_UP) {  // virtual image = nearest x2 upsample of the stored one
    int vy = oy0 + r - PAD, vx = ox0 + c - PAD;
    return ((unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win))
               ? (vy >> 1) * Win + (vx >> 1)
               : -1;
  } else {  // MODE_ZINS: virtual image = zero-inserted x2, window offset -2 (adjoint of MODE_DOWN)
    int vy = oy0 + r - 2, vx = ox0 + c - 2;
    bool ok = (unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win) &&
              !((vy | vx) & 1);
    return ok ? (vy >> 1) * Win + (vx >> 1) : -1;
  }
}

template <int KS, int S, int KC>
__global__ __launch_bounds__(256, 1) void conv_fwd_kernel(ConvArgs a) {
  constexpr int TH = 8, TW = 32, BM = 128;
  constexpr int XH = (TH - 1) * S + KS, XW = (TW - 1) * S + KS, XCH = XH * XW;
  constexpr int TAPS = KS * KS;
  constexpr int XN = KC * XCH;
  constexpr int XPT = (XN + 255) / 256;
  constexpr int WN4 = TAPS * KC * BM / 4;
  constexpr int WPT = (WN4 + 255) / 256;

  __shared__ float Xs[XN];
  __shared__ __attribute__((aligned(16))) float Ws[TAPS * KC * BM];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  int b = blockIdx.x;
  const int cob = b % a.ncob;  // fastest: every XCD (b % 8) keeps one weight slice in its L2
  b /= a.ncob;
  const int tx = b % a.tiles_x;
  b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int wco = (wave & 1) * 64, wrow = (wave >> 1) * 4;
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const float* xn = a.x + (size_t)n * a.Cin * in_plane;

  float xr[XPT];
  f32x4 wr[WPT];

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int e = tid + i * 256;
      float v = 0.f;
      if (e < XN) {
        int c = e / XCH, rem = e % XCH;
        int r = rem / XW, col = rem % XW;
        int ci = c0 + c;
        int si = src_index<KS, S>(a.mode, oy0, ox0, r, col, a.Hin, a.Win);
        if (ci < a.Cin && si >= 0) v = xn[(size_t)ci * in_plane + si];
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int f = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < WN4) {
        int row = f / (BM / 4), q = f % (BM / 4);
        int tap = row / KC, kc = row % KC;
        v = *reinterpret_cast<const f32x4*>(a.wp + ((size_t)(tap * a.CinP + c0 + kc) * a.CoutP +
                                                    cob * BM + q * 4));
      }
      wr[i] = v;
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = a.CinP / KC;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      int e = tid + i * 256;
      if (e < XN) Xs[e] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int f = tid + i * 256;
      if (f < WN4) *reinterpret_cast<f32x4*>(&Ws[f * 4]) = wr[i];
    }
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const int k = 2 * kk + hi;
        const float a0 = Ws[(tap * KC + k) * BM + wco + l31];
        const float a1 = Ws[(tap * KC + k) * BM + wco + 32 + l31];
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bv[j] = Xs[k * XCH + ((wrow + j) * S + ky) * XW + l31 * S + kx];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[j], acc[1][j], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: D[row = co][col = pixel x]; row = (r&3) + 8*(r>>2) + 4*hi
  const size_t out_plane = (size_t)a.Hout * a.Wout;
  const int ox = ox0 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * BM + wco + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (co >= a.Cout) continue;
      const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + wrow + j;
        if (oy < a.Hout && ox < a.Wout) {
          size_t o = ((size_t)n * a.Cout + co) * out_plane + (size_t)oy * a.Wout + ox;
          float v = acc[i][j][r] * a.out_scale + bias;
          if (a.res) v += a.res[o];
          a.y[o] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient:  dWp[tap][co][ci] += sum_{n,pix} dY[n][co][pix] * Xv[n][ci][pix @ tap]
//   M = co (block 128, one 32-row MFMA tile per wave), N = ci (block 32), K = pixels.
//   All TAPS accumulators of the (co, ci) tile stay in registers (9 x 16 = 144) while the
//   block walks its share of 2 x 32 pixel tiles; partial sums leave through fp32 atomics
//   shaped as 128-B row segments (ci contiguous) -- MI355X_MICROARCH "Global float atomics".
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;
  const float* dy;
  float* dwp;  // [taps][Cout][Cin], pre-zeroed (or holding a running sum)
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int mode;
  int tiles_x, tiles_y, ntiles, ncib;
};

template <int KS, int S>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int PH = 2, PW = 32, BM = 128, BN = 32, NPIX = PH * PW;
  constexpr int XH = (PH - 1) * S + KS, XW = (PW - 1) * S + KS;
  constexpr int XCH = XH * XW, XCHP = XCH | 1;  // odd channel stride: conflict-free lane=ci reads
  constexpr int DYP = NPIX + 1;                 // odd row stride: conflict-free lane=co reads
  constexpr int TAPS = KS * KS;

  __shared__ float Xs[BN * XCHP];
  __shared__ float Ds[BM * DYP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int cib = blockIdx.x % a.ncib, cob = blockIdx.x / a.ncib;
  const size_t in_plane = (size_t)a.Hin * a.Win, out_plane = (size_t)a.Hout * a.Wout;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int tile = blockIdx.y; tile < a.ntiles; tile += gridDim.y) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = ty * PH, ox0 = tx * PW;
    __syncthreads();
    // dY tile: 128 co x (2 x 32) pixels
    for (int e = tid; e < BM * NPIX; e += 256) {
      int co = e / NPIX, p = e % NPIX;
      int oy = oy0 + p / PW, ox = ox0 + p % PW;
      int cog = cob * BM + co;
      float v = 0.f;
      if (cog < a.Cout && oy < a.Hout && ox < a.Wout)
        v = a.dy[((size_t)n * a.Cout + cog) * out_plane + (size_t)oy * a.Wout + ox];
      Ds[co * DYP + p] = v;
    }
    // input halo tile: 32 ci x XH x XW
    for (int e = tid; e < BN * XCH; e += 256) {
      int c = e / XCH, rem = e % XCH;
      int r = rem / XW, col = rem % XW;
      int ci = cib * BN + c;
      int si = src_index<KS, S>(a.mode, oy0, ox0, r, col, a.Hin, a.Win);
      float v = 0.f;
      if (ci < a.Cin && si >= 0) v = a.x[((size_t)n * a.Cin + ci) * in_plane + si];
      Xs[c * XCHP + rem] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < NPIX / 2; ++kk) {
      const int p = 2 * kk + hi;
      const int py = p / PW, px = p % PW;
      const float av = Ds[(wave * 32 + l31) * DYP + p];
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = tap / KS, kx = tap % KS;
        const float bv = Xs[l31 * XCHP + (py * S + ky) * XW + px * S + kx];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
      }
    }
  }

  const int ci = cib * BN + l31;
  if (ci < a.Cin) {
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cob * BM + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co < a.Cout) atomicAdd(&a.dwp[((size_t)tap * a.Cout + co) * a.Cin + ci], acc[tap][r]);
      }
    }
  }
}

// pack [Cout][Cin][taps] -> [taps][KP][MP] (zero padded).
//   flip = 0 (forward):        k = ci, m = co, tap kept
//   flip = 1 (data gradient):  k = co, m = ci, tap reversed (180-degree rotation)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                    int Cin, int taps, int KP, int MP, int flip, float scale) {
  size_t total = (size_t)taps * KP * MP;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int m = i % MP;
    size_t t2 = i / MP;
    int k = t2 % KP;
    int tap = t2 / KP;
    int co = flip ? k : m, ci = flip ? m : k;
    int st = flip ? (taps - 1 - tap) : tap;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + st] * scale;
    wp[i] = v;
  }
}

// dWp[tap][co][ci] -> dW[co][ci][tap]
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout,
                                    int Cin, int taps) {
  size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int tap = i % taps;
    size_t t2 = i / taps;
    int ci = t2 % Cin;
    int co = t2 / Cin;
    dw[i] = dwp[((size_t)tap * Cout + co) * Cin + ci];
  }
}

// db[c] += sum over a slice of (n, hw); grid (C, nsplit), float4 loads, one atomic per block
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy,
                                                        float* __restrict__ db, int N, int C,
                                                        int HW) {
  const int c = blockIdx.x;
  const int hw4 = HW >> 2;                       // float4 units per plane (HW % 4 == 0 path)
  const bool vec = (HW & 3) == 0;
  const long long units = (long long)N * (vec ? hw4 : HW);
  const long long per = (units + gridDim.y - 1) / gridDim.y;
  const long long beg = per * blockIdx.y;
  long long end = beg + per;
  if (end > units) end = units;
  float s = 0.f;
  if (vec) {
    for (long long u = beg + threadIdx.x; u < end; u += 256) {
      const int n = (int)(u / hw4), q = (int)(u - (long long)n * hw4);
      const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)n * C + c) * HW + 4 * (size_t)q);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long u = beg + threadIdx.x; u < end; u += 256) {
      const int n = (int)(u / HW), q = (int)(u - (long long)n * HW);
      s += dy[((size_t)n * C + c) * HW + q];
    }
  }
  __shared__ double red[4];
  double d = wave_sum_d((double)s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&db[c], (float)(red[0] + red[1] + red[2] + red[3]));
}

// adjoint of nearest x2 upsampling: dx[n][c][y][x] = sum of the 2x2 block of du
__global__ void upsample2x_bwd_kernel(const float* __restrict__ du, float* __restrict__ dx,
                                      size_t planes, int H, int W) {
  size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int x = i % W;
    size_t t = i / W;
    int y = t % H;
    size_t pl = t / H;
    const float* s = du + (pl * 2 * H + 2 * y) * (2 * (size_t)W) + 2 * x;
    float2 a = *reinterpret_cast<const float2*>(s);
    float2 b = *reinterpret_cast<const float2*>(s + 2 * W);
    dx[i] = (a.x + a.y) + (b.x + b.y);
  }
}

template <int KS, int S, int KC>
int launch_fwd(const ConvArgs& a, hipStream_t st) {
  int blocks = a.N * a.tiles_y * a.tiles_x * a.ncob;
  hipLaunchKernelGGL((conv_fwd_kernel<KS, S, KC>), dim3(blocks), dim3(256), 0, st, a);
  EOVAE_CHECK_LAUNCH("eovae_conv2d_fwd");
  return EOVAE_OK;
}

}  // namespace

extern "C" {

int eovae_conv_packed_dims(int Cout, int Cin, int KS, int mode, int flip, int* KP, int* MP) {
  EOVAE_CHECK_ARG(KS == 1 || KS == 3, "conv: kernel size %d unsupported (1 or 3)", KS);
  int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  int kc = (KS == 1) ? 32 : ((mode == MODE_DOWN && !flip) ? 4 : 8);
  *KP = round_up(k, kc);
  *MP = round_up(m, 128);
  return EOVAE_OK;
}

int eovae_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int KS, int mode,
                            int flip, float scale, void* stream) {
  int KP, MP;
  int rc = eovae_conv_packed_dims(Cout, Cin, KS, mode, flip, &KP, &MP);
  if (rc) return rc;
  size_t total = (size_t)KS * KS * KP * MP;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Cout, Cin, KS * KS, KP, MP, flip, scale);
  EOVAE_CHECK_LAUNCH("eovae_conv_pack_weights");
  return EOVAE_OK;
}

// y = conv(x, wp) * out_scale + bias + residual.  (Cin, Cout) are the GEMM K/M channel counts
// of THIS call (for a data-gradient call they are the layer's Cout/Cin, with flip-packed wp).
int eovae_conv2d_fwd(const float* x, const float* wp, const float* bias, const float* residual,
                     float* y, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                     int KS, int mode, int KP, int MP, float out_scale, void* stream) {
  EOVAE_CHECK_ARG(x && wp && y, "conv2d_fwd: null pointer");
  EOVAE_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv2d_fwd: bad shape");
  EOVAE_CHECK_ARG(KS == 1 || KS == 3, "conv2d_fwd: kernel size %d unsupported", KS);
  EOVAE_CHECK_ARG(mode >= 0 && mode <= 3, "conv2d_fwd: mode %d unknown", mode);
  int eh, ew;
  if (mode == MODE_NORMAL) { eh = Hin; ew = Win; }
  else if (mode == MODE_DOWN) { eh = Hin / 2; ew = Win / 2; }
  else { eh = Hin * 2; ew = Win * 2; }
  if (mode == MODE_DOWN) {  // floor((H + 1 - 3) / 2) + 1
    eh = (Hin + 1 - 3) / 2 + 1;
    ew = (Win + 1 - 3) / 2 + 1;
  }
  EOVAE_CHECK_ARG(Hout == eh && Wout == ew, "conv2d_fwd: output %dx%d does not match mode %d on %dx%d",
                  Hout, Wout, mode, Hin, Win);
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || KS == 3, "conv2d_fwd: mode %d needs a 3x3 kernel", mode);
  ConvArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.res = residual; a.y = y;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  a.CinP = KP; a.CoutP = MP; a.mode = mode; a.out_scale = out_scale;
  a.tiles_x = cdiv(Wout, 32); a.tiles_y = cdiv(Hout, 8); a.ncob = cdiv(Cout, 128);
  EOVAE_CHECK_ARG(MP >= a.ncob * 128 && KP >= Cin, "conv2d_fwd: packed dims too small");
  hipStream_t st = (hipStream_t)stream;
  if (KS == 1) { EOVAE_CHECK_ARG(KP % 32 == 0, "conv2d_fwd: KP %% 32"); return launch_fwd<1, 1, 32>(a, st); }
  if (mode == MODE_DOWN) { EOVAE_CHECK_ARG(KP % 4 == 0, "conv2d_fwd: KP %% 4"); return launch_fwd<3, 2, 4>(a, st); }
  EOVAE_CHECK_ARG(KP % 8 == 0, "conv2d_fwd: KP %% 8");
  return launch_fwd<3, 1, 8>(a, st);
}

// dwp[taps][Cout][Cin] += dY (x) Xv ; caller zero-fills dwp (or passes a running sum)
int eovae_conv2d_wgrad(const float* x, const float* dy, float* dwp, int N, int Cin, int Hin,
                       int Win, int Cout, int Hout, int Wout, int KS, int mode, void* stream) {
  EOVAE_CHECK_ARG(x && dy && dwp, "conv2d_wgrad: null pointer");
  EOVAE_CHECK_ARG(KS == 1 || KS == 3, "conv2d_wgrad: kernel size %d unsupported", KS);
  EOVAE_CHECK_ARG(mode >= 0 && mode <= 2, "conv2d_wgrad: mode %d unsupported", mode);
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || KS == 3, "conv2d_wgrad: mode %d needs a 3x3 kernel", mode);
  WgradArgs a;
  a.x = x; a.dy = dy; a.dwp = dwp;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  a.mode = mode;
  a.tiles_x = cdiv(Wout, 32); a.tiles_y = cdiv(Hout, 2);
  a.ntiles = N * a.tiles_x * a.tiles_y;
  a.ncib = cdiv(Cin, 32);
  int ncob = cdiv(Cout, 128);
  int split = 2048 / (a.ncib * ncob);
  if (split < 1) split = 1;
  if (split > a.ntiles) split = a.ntiles;
  dim3 grid(a.ncib * ncob, split);
  hipStream_t st = (hipStream_t)stream;
  if (KS == 1) hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), grid, dim3(256), 0, st, a);
  else if (mode == MODE_DOWN) hipLaunchKernelGGL((conv_wgrad_kernel<3, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<3, 1>), grid, dim3(256), 0, st, a);
  EOVAE_CHECK_LAUNCH("eovae_conv2d_wgrad");
  return EOVAE_OK;
}

int eovae_conv_unpack_wgrad(const float* dwp, float* dw, int Cout, int Cin, int KS, void* stream) {
  size_t total = (size_t)Cout * Cin * KS * KS;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dwp, dw,
                     Cout, Cin, KS * KS);
  EOVAE_CHECK_LAUNCH("eovae_conv_unpack_wgrad");
  return EOVAE_OK;
}

int eovae_bias_grad(const float* dy, float* db, int N, int C, int HW, void* stream) {
  EOVAE_CHECK_ARG(dy && db && N > 0 && C > 0 && HW > 0, "bias_grad: bad argument");
  long long work = (long long)N * HW;
  int split = (int)((work + 16383) / 16384);
  if (split < 1) split = 1;
  if (split > 64) split = 64;
  while (split > 1 && (long long)C * split > 4096) split >>= 1;
  hipMemsetAsync(db, 0, (size_t)C * sizeof(float), (hipStream_t)stream);
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C, split), dim3(256), 0, (hipStream_t)stream, dy, db, N, C, HW);
  EOVAE_CHECK_LAUNCH("eovae_bias_grad");
  return EOVAE_OK;
}

int eovae_upsample2x_bwd(const float* du, float* dx, int N, int C, int H, int W, void* stream) {
  EOVAE_CHECK_ARG(du && dx && N > 0 && C > 0 && H > 0 && W > 0, "upsample2x_bwd: bad argument");
  size_t total = (size_t)N * C * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, du, dx,
                     (size_t)N * C, H, W);
  EOVAE_CHECK_LAUNCH("eovae_upsample2x_bwd");
  return EOVAE_OK;
}

}  // extern "C"
