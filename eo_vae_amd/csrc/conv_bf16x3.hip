// 3x3 stride-1 convolution as an implicit GEMM on the bf16 matrix cores with SPLIT-bf16
// operands ("bf16x3"): every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and
//      x*w  ~=  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w        (fp32 accumulation in the MFMA)
// which keeps ~16 mantissa bits per operand (relative error ~2^-17 per product; the dropped
// lo*lo term is 2^-18) -- enough to hold the 1e-3 parity bar through the ~60 stacked layers,
// which plain bf16 (2^-9) cannot -- at 3 bf16 MFMAs per tile-step instead of 16 fp32-MFMA
// equivalents (v_mfma_f32_32x32x16_bf16 runs 16x the rate of v_mfma_f32_32x32x2_f32).
// NTERMS = 1 drops the two correction terms (plain bf16, fast / flagged mode).
//
// Same GEMM view and tile-loader index map (normal / nearest-x2 / zero-insert) as conv_f32.hip:
//   D[co][pix] = sum_tap sum_ci Wp[tap][ci][co] * Xv[ci][pix @ tap]
// Block = 512 threads = 8 waves, tile 128 co x (TH x 32) pixels, K streamed in 16-channel chunks.
//   * X halo tile: fp32 NCHW -> registers -> split -> LDS as [k-half][pixel][8 ci] bf16 (hi and
//     lo images).  A lane's MFMA B fragment (8 consecutive ci of one pixel) is ONE 16-byte
//     ds_read_b128; the 32 lanes of a k-half read 32 consecutive pixels = 512 contiguous bytes
//     => bank-conflict free (a [pixel][16 ci] image with 32-byte rows measured 46 % conflict
//     cycles), and a tap shift moves whole pixels, so every read stays 16-byte aligned.
//   * W tile: pre-split, pre-tiled in global as [tap][chunk][co-block][hi|lo][k-half][128 co][8 ci];
//     staged per ky-row phase (3 taps, 24 KiB) by straight 16-byte copies, double buffered.
//   * per chunk: 3 phases x 3 taps x (2 x NI tiles) x NTERMS MFMAs per wave.
#include "eovae_common.h"
#include <type_traits>

namespace {

enum { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_ZINS = 3 };

struct ConvBArgs {
  const float* x;
  const unsigned short* wp;  // split-packed weights (bf16 bits)
  const float* bias;
  const float* res;
  float* y;
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int nchunk, ncob;
  int mode;
  int tiles_x, tiles_y;
  // sub-pixel form of "nearest x2 + 3x3" (4 parity kernels with 2x2 support, 4/9 of the MFMAs):
  //   subpix 1 (forward):  M = 4*creal output channels (parity-major), x = low-res input,
  //                        epilogue pixel-shuffles into y[N][creal][2H][2W]
  //   subpix 2 (data grad): K = 4*creal channels (parity-major) gathered from the high-res dy,
  //                        output = low-res dx; active taps depend on the chunk's parity
  // space-to-depth form of "pad(0,1,0,1) + 3x3 stride 2" (no MFMA spent on inserted zeros):
  //   subpix 3 (forward):  K = 4*creal channels gathered from the high-res x at the chunk's parity
  //   subpix 4 (data grad): M = 4*creal parity-major rows on the low-res dy, pixel-shuffled into dx
  int subpix, creal, cpb;
  int xcd_remap;
};

__device__ __forceinline__ int src_index3(int mode, int oy0, int ox0, int r, int c, int Hin, int Win) {
  if (mode == MODE_NORMAL) {
    int vy = oy0 + r - 1, vx = ox0 + c - 1;
    return ((unsigned)vy < (unsigned)Hin && (unsigned)vx < (unsigned)Win) ? vy * Win + vx : -1;
  } else if (mode == MODE_UP) {
    int vy = oy0 + r - 1, vx = ox0 + c - 1;
    return ((unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win))
               ? (vy >> 1) * Win + (vx >> 1)
               : -1;
  } else {  // MODE_ZINS
    int vy = oy0 + r - 2, vx = ox0 + c - 2;
    bool ok = (unsigned)vy < (unsigned)(2 * Hin) && (unsigned)vx < (unsigned)(2 * Win) &&
              !((vy | vx) & 1);
    return ok ? (vy >> 1) * Win + (vx >> 1) : -1;
  }
}

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

template <int TH, int NTERMS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv3x3_bf16_kernel(ConvBArgs a) {
  constexpr int TW = 32, BM = 128, KC = 16, NTHR = NW * 64;
  constexpr int XH = TH + 2, XW = TW + 2, NPX = XH * XW;
  constexpr int NI = TH / (NW / 2);             // pixel-row tiles per wave
  constexpr int XBYTES = NPX * KC * 2;          // one halo image (hi or lo)
  constexpr int WTAP = 2 * BM * KC * 2;         // hi+lo tile of one tap: 8 KiB
  constexpr int WPHASE = 3 * WTAP;              // one ky row of taps: 24 KiB
  constexpr int XTASKS = NPX * 2;               // (pixel, 8-channel octet)
  constexpr int XPT = (XTASKS + NTHR - 1) / NTHR;
  constexpr int WPT = 1536 / NTHR;              // 16-byte weight units per thread and phase
#ifndef EOVAE_XDB8
#define EOVAE_XDB8 0
#endif
  // the 16-row tile is alone on its CU anyway (242 VGPRs => 2 waves/SIMD), so it can afford a second
  // halo buffer: the next chunk's tile is converted and written while the current one is still being
  // read, with no barrier pair (and no all-waves-idle section) around the write.  The 8-row tile keeps
  // one buffer: it exists to run two blocks per CU.
  constexpr bool XDB = TH == 16 || EOVAE_XDB8;
  constexpr int XSET = 2 * XBYTES;              // hi + lo image of one halo tile
  constexpr int OFF_XLO = XBYTES, OFF_W = (XDB ? 2 : 1) * XSET;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_W + 2 * WPHASE];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  // XCD-aware block -> work map.  Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD
  // b % 8), each with its own L2.  The ncob blocks that share one halo tile are therefore given
  // CONSECUTIVE slots of ONE XCD (tile t lives on XCD t % 8), so the tile is fetched into one L2 once
  // instead of into ncob of them.  Needs the tile count to be a multiple of 8 (any batch of 8k images).
  int b = blockIdx.x;
  int cob;
  if (a.xcd_remap) {
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
  const int xch = kgather ? a.creal : a.Cin;  // channels of the stored input tensor
  const float* xn = a.x + (size_t)n * xch * in_plane;
  // active tap window [ky0, ky0+kyc) x [kx0, kx0+kxc) of a parity; kyn row-phases per chunk
  const int kyn = a.subpix ? 2 : 3;
  const int par1 = mshuffle ? cob / a.cpb : 0;  // parity of this block's output rows
  auto parity_of = [&](int chunk) { return kgather ? (chunk * KC) / a.creal : par1; };
  auto win0 = [&](int pb) {  // first active tap of an axis whose parity bit is pb
    switch (a.subpix) {
      case 1: return pb;          // up fwd:    {py, py+1}
      case 2: return 1 - pb;      // up dgrad:  {1-py, 2-py}
      case 3: return 1;           // down fwd:  {1, 2} (py = 0) / {1} (py = 1)
      case 4: return pb;          // down dgrad: {0, 1} (py = 0) / {1} (py = 1)
      default: return 0;
    }
  };
  auto wincnt = [&](int pb) { return a.subpix == 0 ? 3 : (a.subpix >= 3 ? 2 - pb : 2); };

  float xr[XPT][8];
  f32x4 wr[WPT];

  // chunk-invariant element offset of this thread's halo tasks inside image n (index math hoisted out
  // of the K loop), including the octet's 8-channel stride, and whether the pixel exists at all.
  // The loads themselves are UNCONDITIONAL (a padded pixel reads offset 0 and is zeroed when it is
  // converted): predicated loads made the compiler wrap every one of them in an exec-mask branch and
  // drain vmcnt(0) after every second one -- 12 exposed memory latencies per chunk, 23-32 % of the kernel.
  unsigned xoff[XPT];
  bool xok[XPT];
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int t = tid + i * NTHR;
    const int g = t / NPX, pix = t - g * NPX;
    const int rr = pix / XW, cc = pix - rr * XW;
    int si = -1;
    if (t < XTASKS) {
      if (kgather) {  // low-res halo pixel (vy, vx) -> index of high-res pixel (2vy, 2vx)
        const int vy = oy0 + rr - 1, vx = ox0 + cc - 1;
        if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout) si = 2 * vy * a.Win + 2 * vx;
      } else {
        si = src_index3(a.mode, oy0, ox0, rr, cc, a.Hin, a.Win);
      }
    }
    xok[i] = si >= 0;
    xoff[i] = (unsigned)((t < XTASKS ? g : 0) * 8) * (unsigned)in_plane + (unsigned)(si >= 0 ? si : 0);
  }
  auto load_x = [&](int c0) {
    int cbase = c0, poff = 0;
    if (kgather) {  // channel chunk of parity p reads src[c][2vy+py][2vx+px]
      const int par = c0 / a.creal;
      cbase = c0 - par * a.creal;
      poff = (par >> 1) * a.Win + (par & 1);
    }
    const float* cb = xn + (size_t)cbase * in_plane + poff;  // wave-uniform base of the chunk
    if (c0 + KC <= a.Cin) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* cj = cb + (size_t)j * in_plane;         // scalar base + 32-bit lane offset
#pragma unroll
        for (int i = 0; i < XPT; ++i) xr[i][j] = cj[xoff[i]];
      }
    } else {  // ragged channel tail (only the last chunk of a layer whose Cin is not a multiple of 16)
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int g = (tid + i * NTHR) / NPX;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ci = c0 + g * 8 + j;
          xr[i][j] = ci < a.Cin ? cb[(size_t)j * in_plane + xoff[i]] : 0.f;
        }
      }
    }
  };
  auto store_x = [&](int xb) {
    unsigned char* xs = smem + (XDB ? xb * XSET : 0);
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int t = tid + i * NTHR;
      if (t < XTASKS) {
        const int g = t / NPX, pix = t - g * NPX;
        bf16x8 hi, lo;
        if (!xok[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xr[i][j] = 0.f;
        }
        split8(xr[i], hi, lo);
        *reinterpret_cast<bf16x8*>(xs + g * (NPX * 16) + pix * 16) = hi;
        if (NTERMS == 3) *reinterpret_cast<bf16x8*>(xs + OFF_XLO + g * (NPX * 16) + pix * 16) = lo;
      }
    }
  };
  // phase p = chunk*3 + ky : taps ky*3 .. ky*3+2 of channel chunk `chunk`
  auto load_w = [&](int chunk, int ky) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int u = tid + i * NTHR;       // 16-byte unit in [0, 1536)
      const int tl = u >> 9, within = u & 511;
      const size_t tile = ((size_t)((ky * 3 + tl) * a.nchunk + chunk) * a.ncob + cob);
      wr[i] = *reinterpret_cast<const f32x4*>(
          reinterpret_cast<const unsigned char*>(a.wp) + tile * WTAP + within * 16);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int u = tid + i * NTHR;
      *reinterpret_cast<f32x4*>(smem + OFF_W + buf * WPHASE + u * 16) = wr[i];
    }
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  load_x(0);
  load_w(0, win0(parity_of(0) >> 1));
  store_x(0);
  store_w(0);
  __syncthreads();

#ifndef EOVAE_ABL
#define EOVAE_ABL 0
#endif
#ifndef EOVAE_EXP
#define EOVAE_EXP 0
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for the NEXT chunk's halo loads (issued 1-3 phases ahead on purpose) at the first
// barrier after they were issued; register uses still get their own counted vmcnt waits.
#if EOVAE_EXP & 2
#define LDS_BARRIER()                                      \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)
#else
#define LDS_BARRIER() __syncthreads()
#endif
  // phase p = (chunk, active tap row): one row of taps of one 16-channel chunk
  const int nphase = a.nchunk * kyn;
  for (int p = 0; p < nphase; ++p) {
    const int chunk = p / kyn, kyi = p - chunk * kyn;
    const int par = parity_of(chunk);
    const int ky = win0(par >> 1) + kyi, kx0 = win0(par & 1), kxn = wincnt(par & 1);
    const bool row_active = kyi < wincnt(par >> 1);  // parity-1 axes of the stride-2 forms have 1 tap
    const int buf = p & 1;
    if (!(EOVAE_ABL & 1)) if (p + 1 < nphase) {
      const int nchk = (p + 1) / kyn;
      int nky = win0(parity_of(nchk) >> 1) + (p + 1 - nchk * kyn);
      load_w(nchk, nky > 2 ? 2 : nky);
    }
    // timing-only: 8 = no global loads for the halo (stale registers re-converted), 16 = loads kept
    // alive but no conversion / LDS write
    if (!(EOVAE_ABL & (2 | 8))) if (kyi == 0 && chunk + 1 < a.nchunk) load_x((chunk + 1) * KC);

    const unsigned char* wb = smem + OFF_W + buf * WPHASE;
    const unsigned char* xb = smem + (XDB ? (chunk & 1) * XSET : 0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      if (!row_active || kx < kx0 || kx >= kx0 + kxn) continue;  // uniform: outside the parity's support
      bf16x8 ahi[2], alo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = kx * WTAP + h * (BM * 16) + (wco + i * 32 + r) * 16;
        ahi[i] = *reinterpret_cast<const bf16x8*>(wb + off);
        if (NTERMS == 3) alo[i] = *reinterpret_cast<const bf16x8*>(wb + off + BM * KC * 2);
      }
      // software pipeline over the pixel-row tiles: row j+1's X fragments are requested before row j's
      // six MFMAs are issued (the compiler on its own requests them after, then waits on lgkmcnt)
      auto xfrag = [&](int j, bf16x8& fhi, bf16x8& flo) __attribute__((always_inline)) {
        const int pix = (wrow + j + ky) * XW + r + kx;
        fhi = *reinterpret_cast<const bf16x8*>(xb + h * (NPX * 16) + pix * 16);
        if (NTERMS == 3) flo = *reinterpret_cast<const bf16x8*>(xb + OFF_XLO + h * (NPX * 16) + pix * 16);
      };
      bf16x8 bhi, blo, nhi, nlo;
      xfrag(0, bhi, blo);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (j + 1 < NI) xfrag(j + 1, nhi, nlo);
#if EOVAE_EXP & 1
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (NTERMS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bhi, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], blo, acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bhi, acc[i][j], 0, 0, 0);
        }
#if EOVAE_EXP & 1
        __builtin_amdgcn_s_setprio(0);
#endif
        bhi = nhi;
        blo = nlo;
      }
    }

    if (!(EOVAE_ABL & 1)) if (p + 1 < nphase) store_w(buf ^ 1);  // other buffer: last read in phase p-1, fenced below
    if (kyi == kyn - 1 && chunk + 1 < a.nchunk) {
      // one buffer: every wave must have finished reading this chunk's halo tile first; two buffers:
      // the other one was last read in chunk - 1, a barrier ago
      if (!XDB) if (!(EOVAE_ABL & 4)) LDS_BARRIER();
      if (EOVAE_ABL & 16) {
#pragma unroll
        for (int i = 0; i < XPT; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(xr[i][j]));
      } else if (!(EOVAE_ABL & 2)) store_x((chunk + 1) & 1);
    }
    if (!(EOVAE_ABL & 4)) LDS_BARRIER();
  }

  // Epilogue.  Bias and residual are fetched by UNCONDITIONAL loads (clamped index) in batches of
  // 4 channels x NI rows, then the batch is stored under its bounds mask.  Written the obvious way
  // (`bias ? bias[co] : 0`, `if (res) v += res[o]` per element) every one of the 128 stores of a lane
  // was preceded by s_waitcnt vmcnt(0) -- the previous store's acknowledgement and the residual's
  // full latency, 128 times per block.
  const int ox = ox0 + r;
  const bool okx = ox < a.Wout;
  const int chans = mshuffle ? a.creal : a.Cout;                    // channels of the stored tensor
  const int co_blk = (mshuffle ? (cob - par1 * a.cpb) : cob) * BM;
  const int co_base = co_blk + wco + 4 * h;
  const int py = par1 >> 1, px = par1 & 1;
  const size_t out_plane = mshuffle ? (size_t)a.Hout * a.Wout * 4 : (size_t)a.Hout * a.Wout;
  float* yn = a.y + (size_t)n * chans * out_plane;
  const float* rn = a.res ? a.res + (size_t)n * chans * out_plane : nullptr;
  unsigned rowoff[NI];   // element offset of (row j, this lane's column) inside a channel plane
  bool rowok[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int oy = oy0 + wrow + j;
    rowok[j] = okx && oy < a.Hout;
    const int oyc = oy < a.Hout ? oy : a.Hout - 1, oxc = okx ? ox : a.Wout - 1;
    rowoff[j] = mshuffle ? (unsigned)((2 * oyc + py) * (2 * a.Wout) + 2 * oxc + px)
                         : (unsigned)(oyc * a.Wout + oxc);
  }
  // FULL = the whole 128 co x TH x 32 tile lies inside the tensor (every layer of the model): plain
  // stores, nothing predicated, so no wait is needed between them
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
          bias4[k] = a.bias ? a.bias[coc[k]] : 0.f;       // uniform branch, unconditional load
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
            v[k][j] = acc[i][j][4 * qq + k] + bias4[k];
            if (rn) v[k][j] += rv[k][j];
            if (!FULL) asm volatile("" : "+v"(v[k][j]));  // finish the batch's arithmetic before its masked stores
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

// ---------------------------------------------------------------------------------------
// weight gradient with split-bf16 operands:
//   dWp[tap][co][ci] += sum_{n,pix} dY[n][co][pix] * Xv[n][ci][pix @ tap]       (K = pixels)
// Block = 8 waves, tile 128 co x 64 ci x 9 taps; wave = (32 co) x (32 ci) x 9 accumulator tiles.
// Both operands have K = pixel, which is the contiguous NCHW axis, but a tap shift of one pixel
// would break the 16-byte alignment of a K-contiguous fragment read.  So both tiles are staged
// channel-contiguous ([pixel][32 channels], 64-byte rows: a tap shift moves whole rows) and the
// fragments are gathered with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group,
// delivered pixel-major per lane); 4 consecutive pixel rows = 256 contiguous bytes => no read
// conflicts.  The staging WRITES put 16 consecutive pixels (64-byte stride) of one channel group
// in one instruction, 4-8-way conflicted on a plain image (45 % conflict cycles measured), so the
// 8-byte (dY) / 16-byte (X) channel groups of row p are XOR-swizzled by (p>>1): writes become
// conflict free and a row stays a permutation of itself, so reads stay conflict free too; the
// swizzled per-lane read offsets are loop-invariant registers.  Pixel tiles of 2 x 32 are double
// buffered in LDS; the next tile is loaded half a tile (one 32-pixel output row) at a time behind
// two k-steps (54 MFMAs per wave).
// ---------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  using lds4 = __attribute__((address_space(3))) s16x4;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p1));
  const s16x8 f = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, f);
}

#ifndef EOVAE_WGRAD_BLOCKS
#define EOVAE_WGRAD_BLOCKS 256
#endif

struct WgBArgs {
  const float* x;
  const float* dy;
  float* db;   // [Cout] bias gradient (sum of dy over n, pixels), pre-zeroed, or null
  float* dwp;  // [9][Cout][Cin] fp32, pre-zeroed
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int mode;
  int tiles_x, tiles_y, ntiles, ncib;
  // subpix 1: rows are the 4*creal parity-major output channels of the sub-pixel form; dy is the
  // high-res gradient gathered at the block's parity; only the parity's 2x2 taps are accumulated
  // subpix 2: columns are the 4*creal parity-major (space-to-depth) input channels of the stride-2
  // layer; x is the high-res input gathered at the ci-block's parity; taps (1..2, 1..2)
  int subpix, creal, cpb;
};

template <int NTERMS, int SUBPIX>
__global__ __launch_bounds__(512, 2) void wgrad3x3_bf16_kernel(WgBArgs a) {
  constexpr int NT = SUBPIX ? 4 : 9;  // accumulated taps: all 9, or the parity's 2x2 support
  constexpr int PW = 32, XW = 40, XWV = 34, NXP = 4 * XW;  // 2 output rows + halo; LDS row pitch 40
                                                          // pixels (34 valid) keeps every window start = kx (mod 8)
  constexpr int DPART = 4 * 64 * 64;                      // [4 co-tiles][64 pix][32 co] bf16
  constexpr int XPART = 2 * NXP * 64;                     // [2 ci-tiles][136 pix][32 ci] bf16
  constexpr int OFF_X = 2 * DPART;
  constexpr int BUF = 2 * DPART + 2 * XPART;
  constexpr int XQ = NXP * 8 / 4;                         // X tasks per k-step (272)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int ct = wave & 3, cih = wave >> 2;
  const int cib = blockIdx.x % a.ncib, cob = blockIdx.x / a.ncib;
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

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  // staging registers: HALF a tile (one output row of 32 pixels + its share of the halo) is issued
  // before two k-steps (54 MFMAs per wave, ~1.7 us per SIMD pair) and committed to the other LDS
  // buffer after them.  Timing-only builds (EOVAE_WABL) show staging and MFMA time adding up
  // (0.26-0.35 ms + 0.6-0.7 ms = 1.0-1.2 ms); doubling the prefetch distance from one k-step to two
  // measured +0..3 % on the same box, so load LATENCY is not what serialises them.
  // dY: 2 tasks (4 channels x 1 pixel) per thread; X halo: 640 tasks (8 channels x 1 pixel).
  constexpr int XHALF = NXP * 8 / 2;   // 640
  float dyr[2][4], xr[2][8];
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};  // bias-gradient partials of this thread's 4 channels
  const bool do_db = a.db != nullptr && cib == 0 && SUBPIX != 1;
  unsigned vmask = 0xf;  // bit i: dY task i valid; bit 2+i: X task i valid (set by issue*, used by commit)
  auto issue = [&](int tile, int hf) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = ty * 2, ox0 = tx * PW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pixl = tid & 15, quad = tid >> 4;      // 32 quads x 16 pixels, x-half i
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
      const int tt = tid + i * 512;
      if (tt < XHALF) {
        const int t2 = hf * XHALF + tt;
        const int oct = t2 / NXP, pix = t2 - oct * NXP;
        const int rr = pix / XW, cc = pix - rr * XW;
        int si = -1;
        if (cc < XWV) {
          if (SUBPIX == 2) {  // space-to-depth gather: low-res halo pixel (vy,vx) -> x[2vy+py][2vx+px]
            const int vy = oy0 + rr - 1, vx = ox0 + cc - 1;
            if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout)
              si = (2 * vy + py) * a.Win + 2 * vx + px;
          } else {
            si = src_index3(a.mode, oy0, ox0, rr, cc, a.Hin, a.Win);
          }
        }
        const int xch = SUBPIX == 2 ? a.creal : a.Cin;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ci = cib * 64 + oct * 8 + j;
          xr[i][j] = (si >= 0 && ci < a.Cin)
                         ? a.x[((size_t)n * xch + (ci - (SUBPIX == 2 ? par * a.creal : 0))) * in_plane + si]
                         : 0.f;
        }
      }
    }
    vmask = 0xf;  // zeros already selected
  };
  // Fast path (every channel of the block's 128 co x 64 ci slab exists -- all layers of the model):
  // UNCONDITIONAL loads from clamped addresses, validity kept as 4 mask bits and applied when the
  // registers are converted.  Predicated loads (`ok ? p[i] : 0`) compile to an exec-mask branch per
  // load plus vmcnt(0) drains between them, i.e. several fully exposed memory latencies per half tile.
  const bool fullc = cob * 128 + 128 <= a.Cout && cib * 64 + 64 <= a.Cin;
  const float* dyb = a.dy + (size_t)(cob * 128 - (SUBPIX == 1 ? par * a.creal : 0)) * out_plane;
  const float* xb = a.x + (size_t)(cib * 64 - (SUBPIX == 2 ? par * a.creal : 0)) * in_plane;
  auto issue_fast = [&](int tile, int hf) __attribute__((always_inline)) {
    int t = tile;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = ty * 2, ox0 = tx * PW;
    const int xch = SUBPIX == 2 ? a.creal : a.Cin;
    const float* dyn = dyb + (size_t)n * dych * out_plane;   // wave-uniform bases
    const float* xn = xb + (size_t)n * xch * in_plane;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pixl = tid & 15, quad = tid >> 4;
      const int oy = oy0 + hf, ox = ox0 + 16 * i + pixl;
      const bool ok = oy < a.Hout && ox < a.Wout;
      const unsigned pixoff = !ok ? 0u
                              : (SUBPIX == 1 ? (unsigned)((2 * oy + py) * (2 * a.Wout) + 2 * ox + px)
                                             : (unsigned)(oy * a.Wout + ox));
      m |= (ok ? 1u : 0u) << i;
      const float* src = dyn + (size_t)(quad * 4) * out_plane + pixoff;
#pragma unroll
      for (int j = 0; j < 4; ++j) dyr[i][j] = src[(size_t)j * out_plane];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tt = tid + i * 512;
      const int t2 = hf * XHALF + (tt < XHALF ? tt : 0);
      const int oct = t2 / NXP, pix = t2 - oct * NXP;
      const int rr = pix / XW, cc = pix - rr * XW;
      int si = -1;
      if (cc < XWV && tt < XHALF) {
        if (SUBPIX == 2) {
          const int vy = oy0 + rr - 1, vx = ox0 + cc - 1;
          if ((unsigned)vy < (unsigned)a.Hout && (unsigned)vx < (unsigned)a.Wout)
            si = (2 * vy + py) * a.Win + 2 * vx + px;
        } else {
          si = src_index3(a.mode, oy0, ox0, rr, cc, a.Hin, a.Win);
        }
      }
      m |= (si >= 0 ? 1u : 0u) << (2 + i);
      const float* src = xn + (size_t)(oct * 8) * in_plane + (si >= 0 ? si : 0);
      if (i == 0 || tid < XHALF - 512) {   // second round: 128 of the 512 threads
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = src[(size_t)j * in_plane];
      }
    }
    vmask = m;
  };
  auto commit = [&](int buf, int hf) __attribute__((always_inline)) {
    unsigned char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!((vmask >> i) & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dyr[i][j] = 0.f;
      }
      if (!((vmask >> (2 + i)) & 1)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pixl = tid & 15, quad = tid >> 4;
      typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
      bf16x4v hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 hb = (__bf16)dyr[i][j];
        hi[j] = hb;
        lo[j] = (__bf16)(dyr[i][j] - (float)hb);
      }
      const int ks = 2 * hf + i;
      const int off = (quad >> 3) * 4096 + (ks * 16 + pixl) * 64 + (((quad & 7) ^ ((pixl >> 1) & 7)) * 8);
      *reinterpret_cast<bf16x4v*>(base + off) = hi;
      if (NTERMS == 3) *reinterpret_cast<bf16x4v*>(base + DPART + off) = lo;
      if (do_db) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dbs[j] += dyr[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tt = tid + i * 512;
      if (tt < XHALF) {
        const int t2 = hf * XHALF + tt;
        const int oct = t2 / NXP, pix = t2 - oct * NXP;
        bf16x8 hi, lo;
        split8(xr[i], hi, lo);
        const int off = OFF_X + (oct >> 2) * (NXP * 64) + pix * 64 + (((oct & 3) ^ ((pix >> 1) & 3)) * 16);
        *reinterpret_cast<bf16x8*>(base + off) = hi;
        if (NTERMS == 3) *reinterpret_cast<bf16x8*>(base + XPART + off) = lo;
      }
    }
  };
  // PARC = compile-time parity (sub-pixel form) so every tap index / swizzle offset is static
  auto compute_p = [&](const unsigned char* base, int ks, auto parc) __attribute__((always_inline)) {
    constexpr int PARC = decltype(parc)::value, PY = (PARC >> 1) & 1, PX = PARC & 1;
    const int row = ks >> 1, x0 = 16 * (ks & 1);
    const unsigned char* ap = base + ct * 4096 + ks * 16 * 64;
    const bf16x8 ahi = tr_frag(ap + sa0, ap + sa1);
    bf16x8 alo;
    if (NTERMS == 3) alo = tr_frag(ap + DPART + sa0, ap + DPART + sa1);
    // software pipeline over the taps: the X fragments of tap t+1 are requested BEFORE tap t's three
    // MFMAs are issued (left alone the compiler requests them after, then waits: one exposed
    // transposed-read latency per 3 MFMAs)
    auto tap_ok = [&](int t) { return !(SUBPIX == 2 && (((t >> 1) && PY) || ((t & 1) && PX))); };
    auto xfrag = [&](int t, bf16x8& fhi, bf16x8& flo) __attribute__((always_inline)) {
      const int ky = SUBPIX == 1 ? PY + (t >> 1) : (SUBPIX == 2 ? 1 + (t >> 1) : t / 3);
      const int kx = SUBPIX == 1 ? PX + (t & 1) : (SUBPIX == 2 ? 1 + (t & 1) : t % 3);
      // first pixel of the k-step window: (row+ky)*40 + x0 + kx = kx (mod 8)
      const unsigned char* bp = base + OFF_X + cih * (NXP * 64) + ((row + ky) * XW + x0 + kx) * 64;
      const int o0 = kx == 0 ? sx0 : (kx == 1 ? sx1 : sx2);
      const int o1 = (kx == 0 ? sx4 : (kx == 1 ? sx5 : sx6)) + 4 * 64;
      fhi = tr_frag(bp + o0, bp + o1);
      if (NTERMS == 3) flo = tr_frag(bp + XPART + o0, bp + XPART + o1);
    };
    bf16x8 bhi, blo, nhi, nlo;
    xfrag(0, bhi, blo);   // tap 0 is active for every parity
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!tap_ok(t)) continue;  // compile-time: odd axes of the space-to-depth form have 1 tap
      int tn = t + 1;
      while (tn < NT && !tap_ok(tn)) ++tn;
      if (tn < NT) xfrag(tn, nhi, nlo);
      if (NTERMS == 3) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[t], 0, 0, 0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[t], 0, 0, 0);
      bhi = nhi;
      blo = nlo;
    }
  };
  auto compute = [&](const unsigned char* base, int ks) __attribute__((always_inline)) {
    if (!SUBPIX) { compute_p(base, ks, std::integral_constant<int, 0>{}); return; }
    switch (par) {  // uniform
      case 0: compute_p(base, ks, std::integral_constant<int, 0>{}); break;
      case 1: compute_p(base, ks, std::integral_constant<int, 1>{}); break;
      case 2: compute_p(base, ks, std::integral_constant<int, 2>{}); break;
      default: compute_p(base, ks, std::integral_constant<int, 3>{}); break;
    }
  };

  // The tile loop is instantiated once per load path: a run-time choice between issue_fast() and
  // issue() inside ONE loop makes the loaded registers phi nodes, and the compiler then waits for
  // vmcnt(0) and copies them at the join -- right after issuing the loads, before the MFMAs.
#ifndef EOVAE_WABL
#define EOVAE_WABL 0
#endif
  auto run = [&](auto full_t) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_t)::value;
    auto load = [&](int tile, int hf) __attribute__((always_inline)) {
      if constexpr (FULL) issue_fast(tile, hf); else issue(tile, hf);
    };
    int tile = blockIdx.y;
    if (tile < a.ntiles) {
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {
        load(tile, hf);
        commit(0, hf);
      }
    }
    __syncthreads();
    int cur = 0;
    for (; tile < a.ntiles; tile += gridDim.y) {
      const int next = tile + gridDim.y;
      const bool has_next = next < a.ntiles;
      const unsigned char* base = smem + cur * BUF;
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {
        if (!(EOVAE_WABL & 1)) if (has_next) load(next, hf);
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub)
          if (!(EOVAE_WABL & 4)) compute(base, 2 * hf + sub);
        if (!(EOVAE_WABL & 2)) if (has_next) commit(cur ^ 1, hf);
      }
      __syncthreads();
      cur ^= 1;
    }
  };
  if (fullc) run(std::true_type{}); else run(std::false_type{});

  if (do_db) {  // lanes 16q..16q+15 hold the same 4 channels at 16 different pixels
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = dbs[j];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      const int co = cob * 128 + (tid >> 4) * 4 + j;
      if ((tid & 15) == 0 && co < a.Cout) atomicAdd(&a.db[co], v);
    }
  }
  const int ci = cib * 64 + cih * 32 + r;
  if (ci < a.Cin) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tap = SUBPIX == 1 ? (py + (t >> 1)) * 3 + px + (t & 1)
                                  : (SUBPIX == 2 ? (1 + (t >> 1)) * 3 + 1 + (t & 1) : t);
      if (SUBPIX == 2 && (((t >> 1) && py) || ((t & 1) && px))) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = cob * 128 + ct * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (co < a.Cout) atomicAdd(&a.dwp[((size_t)tap * a.Cout + co) * a.Cin + ci], acc[t][q]);
      }
    }
  }
}

// w[Cout][Cin][9] fp32 -> [tap][chunk][cob][hi|lo][k-half][128][8] bf16 bits (zero padded)
//   flip = 0: k = ci, m = co;  flip = 1 (data gradient): k = co, m = ci, taps reversed
__global__ void pack_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp,
                                  int Cout, int Cin, int nchunk, int ncob, int flip, float scale) {
  const size_t total = (size_t)9 * nchunk * ncob * 128 * 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int kin = i & 15;
    size_t t = i >> 4;
    const int min_ = t & 127;
    t >>= 7;
    const int cob = t % ncob;
    t /= ncob;
    const int chunk = t % nchunk;
    const int tap = t / nchunk;
    const int k = chunk * 16 + kin, m = cob * 128 + min_;
    const int co = flip ? k : m, ci = flip ? m : k;
    const int st = flip ? 8 - tap : tap;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + st] * scale;
    const __bf16 hb = (__bf16)v;
    const __bf16 lb = (__bf16)(v - (float)hb);
    const size_t tile = ((size_t)(tap * nchunk + chunk) * ncob + cob) * (2 * 128 * 16);
    const size_t e = (size_t)(kin >> 3) * (128 * 8) + min_ * 8 + (kin & 7);
    wp[tile + e] = __builtin_bit_cast(unsigned short, hb);
    wp[tile + 128 * 16 + e] = __builtin_bit_cast(unsigned short, lb);
  }
}

// Sub-pixel weights of "nearest x2 + 3x3": W'[(py,px,co)][ci][ty][tx] = sum of the 3x3 taps that
// land on low-res offset (ty-1, tx-1) for output parity (py, px); zero outside the 2x2 support.
//   rows: parity 0 -> ky 0 | ky 1+2 | -      parity 1 -> - | ky 0+1 | ky 2
__device__ __forceinline__ bool sub_contrib(int par, int t, int k) {
  return par == 0 ? (t == 0 ? k == 0 : (t == 1 ? k >= 1 : false))
                  : (t == 0 ? false : (t == 1 ? k <= 1 : k == 2));
}
__global__ void subpix_expand_kernel(const float* __restrict__ w, float* __restrict__ we, int Cout,
                                     int Cin) {
  const size_t total = (size_t)4 * Cout * Cin * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int tap = i % 9;
    size_t t = i / 9;
    const int ci = t % Cin;
    t /= Cin;
    const int co = t % Cout, par = t / Cout;
    const int ty = tap / 3, tx = tap % 3, py = par >> 1, px = par & 1;
    const float* src = w + ((size_t)co * Cin + ci) * 9;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        if (sub_contrib(py, ty, ky) && sub_contrib(px, tx, kx)) v += src[ky * 3 + kx];
    we[i] = v;
  }
}
// adjoint of the expansion: dW[co][ci][ky][kx] = sum_parities dWp'[ty*3+tx][par*Cout+co][ci]
__global__ void subpix_fold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout,
                                   int Cin) {
  const size_t total = (size_t)Cout * Cin * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int tap = i % 9;
    size_t t = i / 9;
    const int ci = t % Cin, co = t / Cin;
    const int ky = tap / 3, kx = tap % 3;
    float v = 0.f;
#pragma unroll
    for (int par = 0; par < 4; ++par) {
      const int py = par >> 1, px = par & 1;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
          if (sub_contrib(py, ty, ky) && sub_contrib(px, tx, kx))
            v += dwp[((size_t)(ty * 3 + tx) * 4 * Cout + (size_t)par * Cout + co) * Cin + ci];
    }
    dw[i] = v;
  }
}

// Space-to-depth forms of "pad(0,1,0,1) + 3x3 stride 2":
//   forward  : y = conv3x3(xs, W'')   xs[(py,px,ci)] = x[ci][2i+py][2j+px]
//              W''[co][(py,px,ci)][1+a][1+b] = W[co][ci][2a+py][2b+px]        (a, b in {0,1})
//   data grad: dx[ci][2i+py][2j+px] = conv3x3(dy, Wd'')[(py,px,ci)]
//              Wd''[(py,px,ci)][co][1-a][1-b] = W[co][ci][2a+py][2b+px]
__global__ void down_expand_kernel(const float* __restrict__ w, float* __restrict__ we, int Cout,
                                   int Cin, int dgrad) {
  const size_t total = (size_t)4 * Cout * Cin * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int tap = i % 9, ty = tap / 3, tx = tap % 3;
    size_t t = i / 9;
    int co, ci, par;
    if (dgrad) {  // [4Cin][Cout][9]
      co = t % Cout; t /= Cout;
      ci = t % Cin; par = t / Cin;
    } else {      // [Cout][4Cin][9]
      ci = t % Cin; t /= Cin;
      par = t % 4; co = t / 4;
    }
    const int py = par >> 1, px = par & 1;
    const int a = dgrad ? 1 - ty : ty - 1, b = dgrad ? 1 - tx : tx - 1;
    const int ky = 2 * a + py, kx = 2 * b + px;
    float v = 0.f;
    if (a >= 0 && a <= 1 && b >= 0 && b <= 1 && ky <= 2 && kx <= 2)
      v = w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
    we[i] = v;
  }
}
// dW[co][ci][ky][kx] = dwp''[(1+(ky>>1))*3 + 1+(kx>>1)][co][((ky&1)*2+(kx&1))*Cin + ci]
__global__ void down_fold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout,
                                 int Cin) {
  const size_t total = (size_t)Cout * Cin * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int tap = i % 9, ky = tap / 3, kx = tap % 3;
    size_t t = i / 9;
    const int ci = t % Cin, co = t / Cin;
    const int tp = (1 + (ky >> 1)) * 3 + 1 + (kx >> 1), par = (ky & 1) * 2 + (kx & 1);
    dw[i] = dwp[((size_t)tp * Cout + co) * (4 * (size_t)Cin) + (size_t)par * Cin + ci];
  }
}

}  // namespace

extern "C" {

// Space-to-depth weights of the stride-2 layer: dgrad = 0 -> we[Cout][4*Cin][3][3] (forward),
// dgrad = 1 -> we[4*Cin][Cout][3][3] (data gradient, already in forward-conv orientation)
int eovae_down_expand(const float* w, float* we, int Cout, int Cin, int dgrad, void* stream) {
  EOVAE_CHECK_ARG(w && we && Cout > 0 && Cin > 0, "down_expand: bad argument");
  size_t total = (size_t)4 * Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(down_expand_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, we, Cout, Cin, dgrad);
  EOVAE_CHECK_LAUNCH("eovae_down_expand");
  return EOVAE_OK;
}

// dw[Cout][Cin][3][3] from the space-to-depth weight gradient dwp[9][Cout][4*Cin]
int eovae_down_fold(const float* dwp, float* dw, int Cout, int Cin, void* stream) {
  EOVAE_CHECK_ARG(dwp && dw && Cout > 0 && Cin > 0, "down_fold: bad argument");
  size_t total = (size_t)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(down_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dwp, dw, Cout, Cin);
  EOVAE_CHECK_LAUNCH("eovae_down_fold");
  return EOVAE_OK;
}

// we[4*Cout][Cin][3][3] = sub-pixel expansion of w[Cout][Cin][3][3] (nearest x2 + 3x3)
int eovae_subpix_expand(const float* w, float* we, int Cout, int Cin, void* stream) {
  EOVAE_CHECK_ARG(w && we && Cout > 0 && Cin > 0, "subpix_expand: bad argument");
  size_t total = (size_t)4 * Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(subpix_expand_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, we, Cout, Cin);
  EOVAE_CHECK_LAUNCH("eovae_subpix_expand");
  return EOVAE_OK;
}

// dw[Cout][Cin][3][3] from the sub-pixel weight gradient dwp[9][4*Cout][Cin]
int eovae_subpix_fold(const float* dwp, float* dw, int Cout, int Cin, void* stream) {
  EOVAE_CHECK_ARG(dwp && dw && Cout > 0 && Cin > 0, "subpix_fold: bad argument");
  size_t total = (size_t)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(subpix_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dwp, dw, Cout, Cin);
  EOVAE_CHECK_LAUNCH("eovae_subpix_fold");
  return EOVAE_OK;
}

// number of bf16 elements of the split-packed weight buffer for a 3x3 layer
size_t eovae_conv3x3_bf16_packed_elems(int Cout, int Cin, int flip) {
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  return (size_t)9 * cdiv(k, 16) * cdiv(m, 128) * 2 * 128 * 16;
}

int eovae_conv3x3_bf16_pack(const float* w, void* wp, int Cout, int Cin, int flip, float scale,
                            void* stream) {
  EOVAE_CHECK_ARG(w && wp && Cout > 0 && Cin > 0, "conv3x3_bf16_pack: bad argument");
  const int k = flip ? Cout : Cin, m = flip ? Cin : Cout;
  const int nchunk = cdiv(k, 16), ncob = cdiv(m, 128);
  size_t total = (size_t)9 * nchunk * ncob * 128 * 16;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned short*)wp, Cout, Cin, nchunk, ncob, flip, scale);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_bf16_pack");
  return EOVAE_OK;
}

// 3x3 stride-1 conv (modes NORMAL / UP / ZINS) with split-bf16 operands.
//   precision: 3 = bf16x3 (fp32-grade), 1 = plain bf16
// (Cin, Cout) = GEMM K / M channel counts of this call (swapped + flip-packed for a data gradient).
//   subpix 0: plain.  subpix 1: forward of "nearest x2 + 3x3" in sub-pixel form: x low-res
//   [N][Cin][Hin][Win], wp = pack(expand(w), flip 0), Cout = the layer's real Cout (multiple of
//   128), y = [N][Cout][2Hin][2Win] (pass Hout = 2Hin, Wout = 2Win), mode must be 0.
//   subpix 2: its data gradient: x = dy [N][Cin][Hin][Win] high-res (Cin = the layer's real Cout,
//   multiple of 16), wp = pack(expand(w), flip 1), output y = dx [N][Cout][Hin/2][Win/2].
int eovae_conv3x3_bf16_fwd(const float* x, const void* wp, const float* bias, const float* residual,
                           float* y, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                           int mode, int precision, int subpix, void* stream) {
  EOVAE_CHECK_ARG(x && wp && y, "conv3x3_bf16_fwd: null pointer");
  EOVAE_CHECK_ARG(N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv3x3_bf16_fwd: bad shape");
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || mode == MODE_UP || mode == MODE_ZINS,
                  "conv3x3_bf16_fwd: mode %d unsupported", mode);
  EOVAE_CHECK_ARG(precision == 1 || precision == 3, "conv3x3_bf16_fwd: precision %d", precision);
  EOVAE_CHECK_ARG(subpix >= 0 && subpix <= 4, "conv3x3_bf16_fwd: subpix %d", subpix);
  ConvBArgs a;
  a.x = x; a.wp = (const unsigned short*)wp; a.bias = bias; a.res = residual; a.y = y;
  a.N = N; a.Hin = Hin; a.Win = Win; a.mode = mode; a.subpix = subpix; a.creal = 0; a.cpb = 1;
  if (subpix == 0) {
    const int eh = mode == MODE_NORMAL ? Hin : 2 * Hin, ew = mode == MODE_NORMAL ? Win : 2 * Win;
    EOVAE_CHECK_ARG(Hout == eh && Wout == ew, "conv3x3_bf16_fwd: output %dx%d does not match mode %d",
                    Hout, Wout, mode);
    a.Cin = Cin; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  } else if (subpix == 1) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual, "conv3x3_bf16_fwd: subpix 1 needs mode 0, no residual");
    EOVAE_CHECK_ARG(Cout % 128 == 0, "conv3x3_bf16_fwd: subpix 1 needs Cout %% 128 == 0 (got %d)", Cout);
    EOVAE_CHECK_ARG(Hout == 2 * Hin && Wout == 2 * Win, "conv3x3_bf16_fwd: subpix 1 output must be 2x");
    a.Cin = Cin; a.Cout = 4 * Cout; a.creal = Cout; a.cpb = Cout / 128;
    a.Hout = Hin; a.Wout = Win;  // tiles run over the low-res grid
  } else if (subpix == 2 || subpix == 3) {
    // 2: data gradient of the upsample layer; 3: forward of the stride-2 layer.  Both gather the
    // K channels of chunk parity p from the high-res source x[N][Cin][Hin][Win].
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual, "conv3x3_bf16_fwd: subpix %d needs mode 0, no residual", subpix);
    EOVAE_CHECK_ARG(Cin % 16 == 0, "conv3x3_bf16_fwd: subpix %d needs Cin %% 16 == 0 (got %d)", subpix, Cin);
    EOVAE_CHECK_ARG(Hin == 2 * Hout && Win == 2 * Wout, "conv3x3_bf16_fwd: subpix %d output must be 1/2", subpix);
    a.Cin = 4 * Cin; a.creal = Cin; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  } else {
    // 4: data gradient of the stride-2 layer: x = dy low-res, y = dx [N][Cout][2Hin][2Win]
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && !residual && !bias, "conv3x3_bf16_fwd: subpix 4 is a bare data gradient");
    EOVAE_CHECK_ARG(Cout % 128 == 0, "conv3x3_bf16_fwd: subpix 4 needs Cout %% 128 == 0 (got %d)", Cout);
    EOVAE_CHECK_ARG(Hout == 2 * Hin && Wout == 2 * Win, "conv3x3_bf16_fwd: subpix 4 output must be 2x");
    a.Cin = Cin; a.Cout = 4 * Cout; a.creal = Cout; a.cpb = Cout / 128;
    a.Hout = Hin; a.Wout = Win;
  }
  Cin = a.Cin; Cout = a.Cout; Hout = a.Hout; Wout = a.Wout;
  a.nchunk = cdiv(Cin, 16); a.ncob = cdiv(Cout, 128);
  a.tiles_x = cdiv(Wout, 32);
  hipStream_t st = (hipStream_t)stream;
#ifndef EOVAE_CONV_NW4
#define EOVAE_CONV_NW4 0
#endif
#ifndef EOVAE_XCD_REMAP
#define EOVAE_XCD_REMAP 1
#endif
  a.xcd_remap = 0;
  auto xcd_ok = [&](long long tiles) { return (EOVAE_XCD_REMAP && a.ncob > 1 && tiles % 8 == 0) ? 1 : 0; };
  // 16-row tiles when that still fills the chip (>= 2 blocks per CU), else 8-row tiles
  const long long blocks16 = (long long)N * cdiv(Hout, 16) * a.tiles_x * a.ncob;
  if (EOVAE_CONV_NW4) {  // two independent 4-wave blocks of 8 x 32 pixels per CU
    a.tiles_y = cdiv(Hout, 8);
    dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x * a.ncob));
    if (precision == 3) hipLaunchKernelGGL((conv3x3_bf16_kernel<8, 3, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_bf16_kernel<8, 1, 4>), grid, dim3(256), 0, st, a);
  } else if (blocks16 >= 512) {
    a.tiles_y = cdiv(Hout, 16);
    a.xcd_remap = xcd_ok(N * a.tiles_y * a.tiles_x);
    dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x * a.ncob));
    if (precision == 3) hipLaunchKernelGGL((conv3x3_bf16_kernel<16, 3, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_bf16_kernel<16, 1, 8>), grid, dim3(512), 0, st, a);
  } else {
    a.tiles_y = cdiv(Hout, 8);
    a.xcd_remap = xcd_ok(N * a.tiles_y * a.tiles_x);
    dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x * a.ncob));
    if (precision == 3) hipLaunchKernelGGL((conv3x3_bf16_kernel<8, 3, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_bf16_kernel<8, 1, 8>), grid, dim3(512), 0, st, a);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_bf16_fwd");
  return EOVAE_OK;
}

// dwp[9][Cout][Cin] += dY (x) Xv with split-bf16 operands (modes NORMAL / UP of the forward layer)
//   subpix 1: weight gradient of "nearest x2 + 3x3" in sub-pixel form: x low-res, dy high-res
//   [N][Cout][2Hin][2Win] (pass Hout = 2Hin, Wout = 2Win), Cout %% 128 == 0, mode 0,
//   dwp = [9][4*Cout][Cin] pre-zeroed; finish with eovae_subpix_fold.
//   db (may be NULL): [Cout] pre-zeroed; receives the bias gradient sum(dy) from the dy tiles the kernel
//   stages anyway (not available with subpix 1, where dy is gathered per parity).
int eovae_conv3x3_bf16_wgrad(const float* x, const float* dy, float* dwp, float* db, int N, int Cin,
                             int Hin, int Win, int Cout, int Hout, int Wout, int mode, int precision,
                             int subpix, void* stream) {
  EOVAE_CHECK_ARG(x && dy && dwp, "conv3x3_bf16_wgrad: null pointer");
  EOVAE_CHECK_ARG(!(db && subpix == 1), "conv3x3_bf16_wgrad: db is not produced in subpix 1");
  EOVAE_CHECK_ARG(subpix >= 0 && subpix <= 2, "conv3x3_bf16_wgrad: subpix %d", subpix);
  EOVAE_CHECK_ARG(mode == MODE_NORMAL || mode == MODE_UP, "conv3x3_bf16_wgrad: mode %d unsupported", mode);
  EOVAE_CHECK_ARG(precision == 1 || precision == 3, "conv3x3_bf16_wgrad: precision %d", precision);
  WgBArgs a;
  a.x = x; a.dy = dy; a.dwp = dwp; a.db = db;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = Hout; a.Wout = Wout;
  a.mode = mode; a.subpix = subpix; a.creal = 0; a.cpb = 1;
  if (subpix == 1) {
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && Cout % 128 == 0 && Hout == 2 * Hin && Wout == 2 * Win,
                    "conv3x3_bf16_wgrad: subpix 1 needs mode 0, Cout %% 128 == 0 and a 2x dy");
    a.creal = Cout; a.cpb = Cout / 128; a.Cout = 4 * Cout; a.Hout = Hin; a.Wout = Win;
    Cout = a.Cout; Hout = a.Hout; Wout = a.Wout;
  } else if (subpix == 2) {  // stride-2 layer: x high-res, dy low-res, dwp = [9][Cout][4*Cin]
    EOVAE_CHECK_ARG(mode == MODE_NORMAL && Cin % 64 == 0 && Hin == 2 * Hout && Win == 2 * Wout,
                    "conv3x3_bf16_wgrad: subpix 2 needs mode 0, Cin %% 64 == 0 and a 1/2 dy");
    a.creal = Cin; a.Cin = 4 * Cin;
    Cin = a.Cin;
  }
  a.tiles_x = cdiv(Wout, 32); a.tiles_y = cdiv(Hout, 2);
  a.ntiles = N * a.tiles_x * a.tiles_y;
  a.ncib = cdiv(Cin, 64);
  const int ncob = cdiv(Cout, 128);
  // one block per CU: every block ends with 1152 atomic wave-instructions (~58 us at the chip's
  // ~50 ns per instruction per CU), so a second round of blocks would pay that tail twice
  int split = EOVAE_WGRAD_BLOCKS / (a.ncib * ncob);
  if (split < 1) split = 1;
  if (split > a.ntiles) split = a.ntiles;
  dim3 grid(a.ncib * ncob, split);
  hipStream_t st = (hipStream_t)stream;
  if (subpix == 1) {
    if (precision == 3) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<3, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<1, 1>), grid, dim3(512), 0, st, a);
  } else if (subpix == 2) {
    if (precision == 3) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<3, 2>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<1, 2>), grid, dim3(512), 0, st, a);
  } else {
    if (precision == 3) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<3, 0>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<1, 0>), grid, dim3(512), 0, st, a);
  }
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_bf16_wgrad");
  return EOVAE_OK;
}

}  // extern "C"
