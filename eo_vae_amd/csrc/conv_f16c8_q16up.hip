// "nearest x2 + 3x3 convolution" (layers.py:47-50, the decoder's three Upsample layers) in sub-pixel form on MX operand
// records -- the forward of conv_f16c8_q16.hip's kind (one wave per SIMD, 16 x 16 MFMAs, every byte fed by LDS-DMA) with
// BOTH COLUMN PARITIES of the output in one wave (round 6).
//
// Sub-pixel form: output pixel (2 y + py, 2 x + px) is a 2 x 2-tap convolution of the LOW-res input with the folded weights
// of parity (py, px) (eovae_subpix_expand: rows par * Cout + co of a [4 Cout][Cin][3][3] tensor whose parity block is non-zero
// on taps ky in {py, py + 1}, kx in {px, px + 1}; input pixel (y + ky - 1, x + kx - 1)) -- 4/9 of the MFMAs of the layer as the
// reference executes it.  The 8-wave kernel's sub-pixel mode (conv_f16c8.hip, SUBPIX = 1) stages and converts a full fp32
// halo chunk per parity block for 4 of 9 taps' MFMAs and stores stride-2 dwords: 360 TFLOP/s of actual MFMA work
// (profiles/r05zzz_bench_cfg2_kernel_stats.csv: 3 launches, 3.4 ms).  Here:
//
//   input      the low-res x as MX operand records (format 2: f16 plane + MX-FP6 plane), written ONCE per layer by
//              records_mx_kernel below (x is the residual stream, not a GroupNorm output; pre-scaled by the power of two of
//              its amax, undone in the epilogue); halo images and weight units arrive by buffer_load ... lds, no staging
//              registers, no conversion VALU in the matrix kernel.
//   work item  (image, 8 low-res rows x 32 columns, 128 output channels, row parity py); a wave owns 2 low-res rows x 32
//              columns x 128 channels x BOTH column parities = 2 x 4 pixel tiles x 8 channel tiles = 64 accumulators of
//              16 x 16 (256 AGPRs): acc[i][px * 4 + 2 jr + hx].
//   phases     a 32-channel chunk = 4 f16 phases + 4 correction phases.  f16 phase q = 2 a + s multiplies, for BOTH px, the
//              tap (ky, kx) = (py + a, px + s): 64 MFMAs on two weight units (one per px: the 16 KB ring slot of
//              conv_f16c8_q16.hip exactly filled) and eight B fragments.  Correction phase 4 + 2 a + px = the MX-FP6 product of
//              the tap pair (py + a, px), (py + a, px + 1) of parity (py, px): one 16 x 16 x 128 MFMA per accumulator of that
//              px (K blocks = [x_lo | x_hi] of the two taps against [w_hi | w_lo]), 32 MFMAs, two FP6 units.
//   epilogue   a lane holds 4 consecutive low-res pixels of one channel for px = 0 and for px = 1: interleaved IN the lane
//              they are 8 consecutive output pixels -- two 16-byte stores, no cross-lane work, no stride-2 dwords.
//
// Request schedule (r = phase of the request; data requested in phase r is readable from phase r + 2: the counted vmcnt wait at
// the end of phase r + 1 and the barrier that opens phase r + 2):
//   weights of phase p + 3 in phase p, ring slot (P + 3) % 4 (two units, four DMAs, in every phase);
//   f16 image of the NEXT chunk in phases 3, 4, 5 (its last readers of this chunk are phase 3's fragments, read in phase 2;
//   first reader: phase 7, which reads phase 0's fragments); FP6 image of the NEXT chunk in phase 7 (last readers: phase 7's
//   fragments, read in phase 6; first reader: phase 3 of the next chunk).
// Everything about asm statements that conv_f16c8_q16.hip's header lists applies here unchanged (conv_f16c8_q16_asm.h).
#include <stdlib.h>

#include "conv_f16c8_common.h"
#include "conv_f16c8_q16_asm.h"

namespace {

constexpr int UBM = 128, UXW = 34, UROWS = 9, UNPX = UROWS * UXW, UNTASK = 5;   // 306 halo pixels in 5 tasks of 64 lanes
constexpr int UPPL = 64 * UNTASK * 16;         // one octet's / piece's plane of a halo image: 320 pixels x 16 B
constexpr int UIMG = 4 * UPPL;                 // 20,480
constexpr int UOFF_XF = 0, UOFF_X6 = UIMG, UOFF_W = 2 * UIMG;
constexpr int UWSLOT = 16384, UNWS = 4;        // weight ring: four phase slots of two 8 KB units
constexpr int ULDS = UOFF_W + UNWS * UWSLOT;   // 106,496
constexpr int UNPH = 8;                        // phases per chunk: 0..3 f16, 4..7 correction

__device__ __host__ constexpr int u_nh(int p) { return p == 3 || p == 4 ? 2 : (p == 5 ? 1 : (p == 7 ? 5 : 0)); }

// fp32 NCHW -> MX operand records (the format groupnorm.hip's gn_apply_mx_kernel writes: F plane [n][C/8][HW] x 16 B = 8 x f16,
// S plane behind it [n][C/32][piece 0..3][HW] x 16 B = block-scaled e2m3 of (x - f16 x) | x per pixel and 32 channels), for a
// layer input that is not a GroupNorm output.  xamax (nullable): x is scaled by the power of two that puts its amax at
// 2^13..2^14 first; the consumer undoes it with its result.  A thread = a pixel x 32 channels; grid = N * C/32 * ceil(HW / 256).
__global__ __launch_bounds__(256) void records_mx_kernel(const float* __restrict__ x, const unsigned* __restrict__ xamax,
                                                         int* __restrict__ rec, int N, int C, int HW, int pblocks) {
  const int pb = blockIdx.x % pblocks, bg = blockIdx.x / pblocks;   // bg = n * C/32 + block
  const int blk = bg % (C / 32), n = bg / (C / 32);
  const int p = pb * 256 + threadIdx.x;
  if (p >= HW) return;
  float xs = 1.f;
  if (xamax) {
    const unsigned bits = *xamax;
    if (bits) xs = pow2f(13 - expo_of_bits(bits));
  }
  float v[32], lo[32];
  const float* xp = x + ((size_t)n * C + blk * 32) * HW + p;
#pragma unroll
  for (int j = 0; j < 32; ++j) v[j] = xp[(size_t)j * HW] * xs;
  i32x4* fpl = reinterpret_cast<i32x4*>(rec) + ((size_t)n * (C / 8) + blk * 4) * HW + p;
#pragma unroll
  for (int oc = 0; oc < 4; ++oc) {
    i32x4 pk;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      int hh;
      const float v0 = v[8 * oc + j], v1 = v[8 * oc + j + 1];
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(v0), "v"(v1));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[8 * oc + j]) : "v"(hh), "v"(v0));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[8 * oc + j + 1]) : "v"(hh), "v"(v1));
      pk[j >> 1] = hh;
    }
    fpl[(size_t)oc * HW] = pk;
  }
  float ah = 0.f, al = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    ah = fmaxf(ah, fabsf(v[j]));
    al = fmaxf(al, fabsf(lo[j]));
  }
  i32x4 q0, q1;
  i32x4* dst = reinterpret_cast<i32x4*>(reinterpret_cast<char*>(rec) + (size_t)N * C * HW * 2) + ((size_t)bg * 4) * HW + p;
  mx6_quant(lo, al, q0, q1);
  dst[0] = q0;
  dst[(size_t)HW] = q1;
  mx6_quant(v, ah, q0, q1);
  dst[(size_t)2 * HW] = q0;
  dst[(size_t)3 * HW] = q1;
}

// a: Cin, Hin, Win = the low-res input; Cout = output channels; Hout, Wout = 2 Hin, 2 Win; cpb = Cout / 128;
// ncob = 4 cpb (128-row blocks of the expanded weights); tiles_y = Hin / 8, tiles_x = Win / 32; ntiles = N tiles_y tiles_x 2 cpb
__global__ __launch_bounds__(256, 1) void conv3x3_f16c8_q16up_kernel(ConvCArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ULDS];
  __shared__ float gcell[4 * 2 * 4 * UBM];   // [wave][sum | sumsq][lane group][channel of the block] (an array of its own: conv_f16c8_q16.hip)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wrow = wave * 2;
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const size_t out_plane = (size_t)a.Hout * a.Wout;

  struct Tile { int cb, py, n, oy0, ox0; };
  const int ngrp = 2 * a.cpb;   // the (co block, py) items that read (nearly) the same halo tile
  auto decode = [&](int b) {
    Tile t;
    int cg;
    if (a.xcd_remap) {  // ... take consecutive slots of ONE XCD
      const int xcd = b & 7, q = b >> 3;
      cg = q % ngrp;
      b = (q / ngrp) * 8 + xcd;
    } else {
      cg = b % ngrp;
      b /= ngrp;
    }
    t.cb = cg >> 1;
    t.py = cg & 1;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    t.n = b / a.tiles_y;
    t.oy0 = ty * 8;
    t.ox0 = tx * 32;
    return t;
  };

  auto lds0_ = [&]() __attribute__((always_inline)) { return (unsigned)(size_t)smem; };
  float xinv = 1.f;   // range pre-scale of the records (amax to 2^13..2^14), undone with the result
  if (a.xamax) {
    const unsigned bits = *a.xamax;
    if (bits) xinv = pow2f(expo_of_bits(bits) - 13);
  }

  // ---------------------------------------------------------------- halo requests (wave = channel octet / FP6 piece)
  // halo = low-res rows oy0 - 1 + py .. oy0 + 7 + py (nine), columns ox0 - 1 .. ox0 + 32; task k of a wave: pixels 64 k + lane
  const unsigned OOR = 0x80000000u;
  unsigned xoff[UNTASK];
  auto setup_x = [&](const Tile& t) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // keeps the per-task index math out of the registers that live across the chunk loop
#pragma unroll
    for (int k = 0; k < UNTASK; ++k) {
      const int pix = ln + 64 * k;
      const int rr = (pix * 241) >> 13, cc = pix - UXW * rr;   // pix / 34, exact for pix < 640
      const int vy = t.oy0 + t.py + rr - 1, vx = t.ox0 + cc - 1;
      const bool ok = (pix < UNPX) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
      xoff[k] = ok ? (unsigned)(vy * a.Win + vx) * 16u : OOR;
    }
  };
  const unsigned plane_bytes = (unsigned)(in_plane * 16);
  auto srd_xf = [&](const Tile& t, int chunk) __attribute__((always_inline)) {   // F plane [n][Cin / 8][pixel][16 B], octet 4 chunk + wave
    const char* p = reinterpret_cast<const char*>(a.x) + (((size_t)t.n * (a.Cin / 8) + 4 * chunk + wave) * in_plane) * 16;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, plane_bytes, 0x00020000);
  };
  auto srd_x6 = [&](const Tile& t, int chunk) __attribute__((always_inline)) {   // S plane [n][Cin / 32][piece][pixel][16 B], piece = wave
    const char* p = reinterpret_cast<const char*>(a.x6) + ((((size_t)t.n * (a.Cin / 32) + chunk) * 4 + wave) * in_plane) * 16;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, plane_bytes, 0x00020000);
  };
  unsigned char* const xf_dst = smem + UOFF_XF + wave * UPPL;
  unsigned char* const x6_dst = smem + UOFF_X6 + wave * UPPL;
  auto dma_x = [&](const __amdgpu_buffer_rsrc_t& srd, unsigned char* dst, int k) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)(dst + k * 1024), 16, (int)xoff[k], 0, 0, 0);
  };

  // ---------------------------------------------------------------- weight requests: 8 KB units, two DMAs each
  const size_t wbytes = (size_t)9 * a.nchunk * a.ncob * C8_WTILE;
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wp), 0, (unsigned)wbytes, 0x00020000);
  const int wvoff = tid * 16;
  auto dma_unit = [&](unsigned soff, unsigned slotbyte) __attribute__((always_inline)) {
    unsigned char* dst = smem + UOFF_W + slotbyte + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr)dst, 16, wvoff, (int)soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr)(dst + 4096), 16, wvoff, (int)(soff + 4096u), 0, 0);
  };
  // byte offset of the weight tile (tap, chunk, parity par, co block cb) of the expanded pack
  auto wtile = [&](int tap, int chunk, int par, int cb) __attribute__((always_inline)) {
    return (unsigned)(((size_t)(tap * a.nchunk + chunk) * a.ncob + par * a.cpb + cb) * C8_WTILE);
  };
  // units of phase q (0..7) of chunk `chunk` of work item t into ring slot `slot`
  auto request_w = [&](auto qc, const Tile& t, int chunk, unsigned slot) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    const unsigned sb = slot * UWSLOT;
    if constexpr (q < 4) {          // tap (py + a, px + s) of parity (py, px): the px = 0 unit | the px = 1 unit
      constexpr int a_ = q >> 1, s = q & 1;
      const int ky = t.py + a_;
      dma_unit(wtile(ky * 3 + s, chunk, 2 * t.py, t.cb), sb);
      dma_unit(wtile(ky * 3 + 1 + s, chunk, 2 * t.py + 1, t.cb), sb + 8192);
    } else {                        // FP6 units of the tap pair (py + a, px), (py + a, px + 1) of parity (py, px)
      constexpr int a_ = (q - 4) >> 1, px = (q - 4) & 1;
      const int t0 = (t.py + a_) * 3 + px;
      dma_unit(wtile(t0, chunk, 2 * t.py + px, t.cb) + C8_WOFF6, sb);
      dma_unit(wtile(t0 + 1, chunk, 2 * t.py + px, t.cb) + C8_WOFF6, sb + 8192);
    }
  };

  // ---------------------------------------------------------------- fragments
  // lane (li, g) of a 16 x 16 MFMA: A row / B column li, K block g.
  //   f16:  K block g = channel octet g of the chunk.   weights: unit [octet][row][16 B] (px = 1: + 8 KB); pixels: f16 image
  //   FP6:  K block g = (tap t0 + g / 2, part g % 2).   weights: units of the two taps [tap][part][piece][row][16 B]; pixels:
  //         FP6 image pieces 2 (g % 2), 2 (g % 2) + 1 at the pixel shifted by THAT tap (the second tap = one column further)
  const unsigned lds0 = lds0_();
  const unsigned a_f = lds0 + (unsigned)(UOFF_W + g * 2048 + li * 16);
  const unsigned a_c = lds0 + (unsigned)(UOFF_W + (g >> 1) * 8192 + (g & 1) * 4096 + li * 16);
  // MFMA row li of a pixel tile = low-res column bcol (not li): the four rows 4 q .. 4 q + 3 a result lane holds are the column
  // PAIRS 2 q, 2 q + 1 and 8 + 2 q, 8 + 2 q + 1, so that each of the epilogue's two stores (a pair x both column parities =
  // 16 bytes per lane) is contiguous over the four lanes of a quad -- see epilogue_tile
  const int bcol = ((li & 2) << 2) + 2 * (li >> 2) + (li & 1);
  const unsigned b_f = lds0 + (unsigned)(UOFF_XF + g * UPPL + (wrow * UXW + bcol) * 16);
  const unsigned b_c = lds0 + (unsigned)(UOFF_X6 + 2 * (g & 1) * UPPL + (wrow * UXW + bcol) * 16) + (unsigned)(g >> 1) * 16u;

  f32x4 acc[8][8];   // [co tile of 16][px * 4 + pixel tile (row jr, half hx)]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  };
  // A fragments (weights) by blocks of 16 MFMAs, double-buffered: f16 block c2 = co tiles 2 c2, 2 c2 + 1 x both px
  // (Ar[c2 & 1][2 t + px]); correction block cb = co tiles 4 cb .. 4 cb + 3 (Ar / Ah[cb & 1][u]).  B fragments (pixels) per
  // phase: f16 phases eight (index px * 4 + 2 jr + hx), correction phases four (2 jr + hx).
  i32x4 Ar[2][4], Br[2][8];
  i32x3 Ah[2][4], Bh[2][4];

  auto read_a = [&](auto isc, auto blk, unsigned sb) __attribute__((always_inline)) {
    constexpr bool ISC = decltype(isc)::value;
    constexpr int b = decltype(blk)::value;
    if constexpr (ISC) {
      const unsigned base = a_c + sb;
      static_for<0, 4>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        rd128<(4 * b + u) * 256>(Ar[b & 1][u], base);
        rd96<(4 * b + u) * 256 + 2048>(Ah[b & 1][u], base);
      });
    } else {
      const unsigned base = a_f + sb;
      static_for<0, 4>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, t = u >> 1, px = u & 1;
        rd128<(2 * b + t) * 256 + px * 8192>(Ar[b & 1][u], base);
      });
    }
  };
  // B fragment k of phase q into Br[q & 1][k] (+ Bh)
  auto read_b = [&](auto qc, auto kc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, k = decltype(kc)::value;
    if constexpr (q < 4) {
      constexpr int a_ = q >> 1, s = q & 1, px = k >> 2, jr = (k >> 1) & 1, hx = k & 1;
      rd128<((a_ + jr) * UXW + px + s + 16 * hx) * 16>(Br[q & 1][k], b_f);
    } else {
      constexpr int a_ = (q - 4) >> 1, px = (q - 4) & 1, jr = (k >> 1) & 1, hx = k & 1;
      static_assert(k < 4, "four B fragments per correction phase");
      rd128<((a_ + jr) * UXW + px + 16 * hx) * 16>(Br[q & 1][k], b_c);
      rd96<((a_ + jr) * UXW + px + 16 * hx) * 16 + UPPL>(Bh[q & 1][k], b_c);
    }
  };
  // MFMA mm (0..15) of block b of phase q
  auto mma = [&](auto qc, auto bc, auto mc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, b = decltype(bc)::value, mm = decltype(mc)::value;
    // (pixels as the MFMA's A operand, weights as its B operand: a lane's four result registers are four consecutive
    // low-res pixels of one channel)
    if constexpr (q < 4) {
      constexpr int jj = mm >> 1, t = mm & 1, i = 2 * b + t, px = jj >> 2;
      mfma_f(acc[i][jj], Br[q & 1][jj], Ar[b & 1][2 * t + px]);
    } else {
      constexpr int px = (q - 4) & 1, k = mm >> 2, u = mm & 3, i = 4 * b + u, jj = 4 * px + k;
      i32x6 a6, b6;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a6[e] = Ar[b & 1][u][e]; b6[e] = Br[q & 1][k][e]; }
      a6[4] = Ah[b & 1][u][0]; a6[5] = Ah[b & 1][u][1];
      b6[4] = Bh[q & 1][k][0]; b6[5] = Bh[q & 1][k][1];
      mfma_c<(q == UNPH - 1 && b == 1 && mm >= 14)>(acc[i][jj], b6, a6, Bh[q & 1][k][2], Ah[b & 1][u][2]);
    }
  };

  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  auto landed_a = [&](int pr) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("" : "+v"(Ar[pr][u]));
      asm volatile("" : "+v"(Ah[pr][u]));
    }
  };
  auto landed_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(Br[par][j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(Bh[par][j]));
  };

  // ---------------------------------------------------------------- GroupNorm partial sums (as conv_f16c8_q16.hip)
  auto gn_flush = [&](int tn, int tcb) __attribute__((always_inline)) {
    const int te = wave * 64 + fresh_lane();   // (not loop-invariant: no spill across the main loop, conv_f16c8_common.h)
    const int v = te >> 7, c = te & 127;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {   // fixed order
      const float* cw = gcell + ((w * 2 + v) * 4) * UBM + c;
      t += (cw[0] + cw[UBM]) + (cw[2 * UBM] + cw[3 * UBM]);
    }
    const int cpg = 1 << a.gn_shift;   // 4..64 channels per group (eligibility): inside one wave's 64 channels
    for (int s = 1; s < cpg && s < 64; s <<= 1) t += __shfl_xor(t, s, 64);
    if ((c & (cpg - 1)) == 0) atomicAdd(a.gn_sums + ((size_t)tn * a.gn_groups + ((tcb + c) >> a.gn_shift)) * 2 + v, (double)t);
  };

  // ---------------------------------------------------------------- epilogue
  // acc[i][4 px + 2 jr + hx][k]: channel 16 i + li, low-res pixel (row wrow + jr, column 16 hx + {2 g, 2 g + 1, 8 + 2 g,
  // 9 + 2 g}[k] -- bcol above), column parity px -> output row 2 (oy0 + wrow + jr) + py, columns
  // 2 (ox0 + 16 hx + 2 g) + {0..3} = a0 b0 a1 b1 and 16 columns further a2 b2 a3 b3.
  // As conv_f16c8_q16.hip's: every register goes through one ds_bpermute_b32 (lane 4 c + p <- lane 16 p + c: channel c = lane
  // >> 2, column group p = lane & 3), so that a lane quad writes one 64-byte segment (tools/probe_store_pattern.hip: 2.5 x the
  // store rate of the MFMA layout's four lines per quad); bias loads are asm with one counted wait, per-lane addresses derive
  // from fresh_lane() (no hoisting + spill), and the compiler is shown a vmcnt(0) where nothing is outstanding.
  auto epilogue_tile = [&](const Tile& t) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to the compiler (conv_f16c8_q16.hip epilogue_tile)
    const int co_blk = t.cb * UBM;
    const unsigned span = (unsigned)(UBM * out_plane * 4);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
        a.y + ((size_t)t.n * a.Cout + co_blk) * out_plane, 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias + co_blk : a.y), 0, a.bias ? (unsigned)(UBM * 4) : 0u, 0x00020000);
    const int le = fresh_lane();
    const int ec = le >> 2, ep = le & 3;   // channel, column group
    const int psrc = ((ep << 4) | ec) << 2;
    const int lot = (int)(((size_t)ec * out_plane + (size_t)(2 * (t.oy0 + wrow) + t.py) * a.Wout + 2 * (t.ox0 + 2 * ep)) * 4);
    const int lob = ec * 4;
    const int tile16 = (int)(16 * out_plane * 4), rowb2 = 2 * a.Wout * 4;
    float* const cell = gcell + wave * 8 * UBM + ep * UBM + ec;
    float bv[8];
    static_for<0, 8>([&](auto ic) __attribute__((always_inline)) { load32_asm<decltype(ic)::value * 64>(bv[decltype(ic)::value], bsrd, lob); });
    wait_vm<0>();
#pragma unroll
    for (int k = 0; k < 8; ++k) landed(bv[k]);
    float amx = 0.f;   // max |y| of the lane's values (a.out_amax: the range pre-scale of the layer that reads y, without a pass over y)
    auto perm = [&](float v) __attribute__((always_inline)) {
      return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(psrc, __builtin_bit_cast(int, v)));
    };
    static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const float bias1 = bv[i];
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // pixel tile (jr, hx)
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = acc[i][k][e], a1 = acc[i][4 + k][e];   // (copied out: __builtin_bit_cast of a vector element, q16.hip)
          o0[e] = __builtin_fmaf(perm(a0), xinv, bias1);
          o1[e] = __builtin_fmaf(perm(a1), xinv, bias1);
          sv += o0[e] + o1[e];
          qv = __builtin_fmaf(o0[e], o0[e], qv);
          qv = __builtin_fmaf(o1[e], o1[e], qv);
          amx = __builtin_fmaxf(amx, __builtin_fmaxf(__builtin_fabsf(o0[e]), __builtin_fabsf(o1[e])));
        }
        f32x4 s0, s1;
        s0[0] = o0[0]; s0[1] = o1[0]; s0[2] = o0[1]; s0[3] = o1[1];
        s1[0] = o0[2]; s1[1] = o1[2]; s1[2] = o0[3]; s1[3] = o1[3];
        const int off = i * tile16 + (k >> 1) * rowb2 + (k & 1) * 128;
        store128_guarded(s0, ysrd, lot, off);         // (guarded: conv_f16c8_common.h -- the compiler re-uses s0 / s1 at once)
        store128_guarded(s1, ysrd, lot, off + 64);
      }
      cell[16 * i] = sv;
      cell[4 * UBM + 16 * i] = qv;
    });
    if (a.out_amax) {   // uniform
      amx = wave_max(amx);
      if (le == 0) atomicMax(a.out_amax, __builtin_bit_cast(unsigned, amx));   // non-negative floats order like their bits
    }
  };

  // ---------------------------------------------------------------- prologue
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  Tile cur = decode(tile);
  unsigned P = 0;   // running phase number (mod 4 = ring slot)
  {
    setup_x(cur);
    const __amdgpu_buffer_rsrc_t s0 = srd_xf(cur, 0), s1 = srd_x6(cur, 0);
#pragma unroll
    for (int k = 0; k < UNTASK; ++k) dma_x(s0, xf_dst, k);
#pragma unroll
    for (int k = 0; k < UNTASK; ++k) dma_x(s1, x6_dst, k);
    request_w(std::integral_constant<int, 0>{}, cur, 0, 0u);
    request_w(std::integral_constant<int, 1>{}, cur, 0, 1u);
    request_w(std::integral_constant<int, 2>{}, cur, 0, 2u);
  }
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  {
    using Q0 = std::integral_constant<int, 0>;
    read_a(std::false_type{}, Q0{}, 0u);
    static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { read_b(Q0{}, kc); });
  }

  // ---------------------------------------------------------------- main loop
  for (; tile < a.ntiles; tile += gridDim.x) {
    const bool has_next_tile = tile + (int)gridDim.x < a.ntiles;
    Tile nxt = cur;
    if (has_next_tile) nxt = decode(tile + gridDim.x);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool last_chunk = chunk + 1 == a.nchunk;
      // the chunk whose images and first weights are requested behind this one: the next chunk, the next tile's first, or
      // (end of work) this chunk again -- re-requesting it writes the bytes that are already there
      const bool wrap = last_chunk && has_next_tile;
      const Tile& xt = wrap ? nxt : cur;
      const int xchunk = last_chunk ? (has_next_tile ? 0 : chunk) : chunk + 1;
      __amdgpu_buffer_rsrc_t sf = srd_xf(cur, chunk), s6 = sf;   // the next chunk's sources (set in phase 3)
      static_for<0, UNPH>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int pn = (p + 1) % UNPH;
        constexpr bool ISC = p >= 4, NISC = pn >= 4;
        constexpr int NBLK = ISC ? 2 : 4;          // blocks of 16 MFMAs in this phase
        constexpr int NBN = NISC ? 4 : 8;          // B fragments of the next phase, read in this one
        constexpr int BPB = NBN / NBLK;            // ... per block: f16 phase 2 | 1, correction phase 4 | 2
        constexpr int BI = NISC ? 2 : 1;           // LDS instructions per B fragment of the next phase
        using PN = std::integral_constant<int, pn>;
        barrier();   // phase p's weights, the images it reads and its first fragments are there
        landed_a(0);
        landed_b(p & 1);
        const unsigned sb = ((P + p) & 3u) * UWSLOT, sbn = ((P + p + 1) & 3u) * UWSLOT;
        if constexpr (p == 3) {
          if (wrap) setup_x(nxt);   // uniform; every halo request from here on belongs to the next chunk
          sf = srd_xf(xt, xchunk);
          s6 = srd_x6(xt, xchunk);
        }
        static_for<0, NBLK>([&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          static_for<0, 16>([&](auto mc) __attribute__((always_inline)) {
            constexpr int mm = decltype(mc)::value;
            mma(pc, bc, mc);
            if constexpr (mm == 1) {
              if constexpr (b == 0) {   // weights of phase p + 3
                if constexpr (p + 3 < UNPH) request_w(std::integral_constant<int, p + 3>{}, cur, chunk, (P + p + 3) & 3u);
                else request_w(std::integral_constant<int, p + 3 - UNPH>{}, xt, xchunk, (P + p + 3) & 3u);
              }
              if constexpr (b == 1) {   // halo requests of this phase
                if constexpr (p == 3) { dma_x(sf, xf_dst, 0); dma_x(sf, xf_dst, 1); }
                if constexpr (p == 4) { dma_x(sf, xf_dst, 2); dma_x(sf, xf_dst, 3); }
                if constexpr (p == 5) { dma_x(sf, xf_dst, 4); }
                if constexpr (p == 7) {
                  dma_x(s6, x6_dst, 0); dma_x(s6, x6_dst, 1); dma_x(s6, x6_dst, 2); dma_x(s6, x6_dst, 3); dma_x(s6, x6_dst, 4);
                }
              }
            }
            if constexpr (mm == 3) {     // A fragments of the next block (of the next phase behind the last block)
              if constexpr (b + 1 < NBLK) read_a(std::integral_constant<bool, ISC>{}, std::integral_constant<int, b + 1>{}, sb);
              else read_a(std::integral_constant<bool, NISC>{}, std::integral_constant<int, 0>{}, sbn);
            }
            // B fragments of the next phase: BPB per block, behind MFMAs 5, 7, 9, 11
            if constexpr (mm >= 5 && mm <= 11 && (mm & 1) == 1) {
              constexpr int s = (mm - 5) >> 1;
              if constexpr (s < BPB) read_b(PN{}, std::integral_constant<int, b * BPB + s>{});
            }
            if constexpr (mm == 15 && b + 1 < NBLK) {   // the next block's A fragments have landed; the B reads behind them may be outstanding
              wait_lgkm<BPB * BI>();
              landed_a((b + 1) & 1);
            }
          });
        });
        // everything requested before this phase has landed (its own requests may be outstanding); behind a tile's last
        // phase all of it, and the first phase of a tile does not wait (conv_f16c8_q16.hip: the epilogue's stores)
        constexpr int NREQ = 4 + u_nh(p);
        if constexpr (p == UNPH - 1) {
          if (last_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wait_lgkm<0>();    // the next phase-0 fragments, before the compiler-scheduled epilogue may touch their registers
            landed_a(0);
            landed_b(0);
          } else wait_vm<NREQ>();
        } else if constexpr (p == 0) {
          if (chunk != 0 || (EOVAE_Q16_SAFE & 1)) wait_vm<NREQ>();
        } else {
          wait_vm<NREQ>();
        }
      });
      P += UNPH;
    }
    epilogue_tile(cur);
    zero_acc();
    if (a.gn_sums) {   // uniform.  The cells are rewritten by the next epilogue, a whole tile later.
      barrier();       // (raw: __syncthreads() carries a release fence = vmcnt(0), i.e. a wait for the epilogue's stores)
      gn_flush(cur.n, cur.cb * UBM);
    }
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests behind the last chunk
}

}  // namespace

extern "C" {

// 1 when eovae_conv3x3_f16c8_up_mx takes the layer: x [N][Cin][Hin][Win] -> y [N][Cout][2 Hin][2 Win]
int eovae_conv3x3_f16c8_up_mx_eligible(int N, int Cin, int Hin, int Win, int Cout) {
  if (!(N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0)) return 0;
  if (eovae_opt(EOVAE_OPT_C8_UP_Q16) == 0 || eovae_opt(EOVAE_OPT_C8_MX) == 0) return 0;
  if (Cin % 32 || Cout % 128 || Hin % 8 || Win % 32) return 0;
  const size_t in_plane = (size_t)Hin * Win, out_plane = 4 * in_plane;
  if (in_plane * 16 >= (1ull << 31) || (size_t)UBM * out_plane * 4 >= (1ull << 31)) return 0;   // 32-bit byte offsets
  if ((size_t)9 * (Cin / 32) * (4 * Cout / 128) * C8_WTILE >= (1ull << 32)) return 0;
  if (Cout / 32 < 4 || Cout / 32 > 64) return 0;   // GroupNorm(32) sums from the epilogue: groups inside one wave's 64 channels
  return 1;
}

// fp32 x [N][C][HW] -> MX operand records rec (N*C*HW*4 bytes: F plane | S plane), pre-scaled by the power of two of *xamax
// (float bits of max|x|, device; nullable).  C % 32 == 0.
int eovae_records_mx_from_f32(const float* x, const unsigned* xamax, float* rec, int N, int C, int HW, void* stream) {
  EOVAE_CHECK_ARG(x && rec, "records_mx_from_f32: null pointer");
  EOVAE_CHECK_ARG(N > 0 && C > 0 && C % 32 == 0 && HW > 0, "records_mx_from_f32: bad shape (N %d C %d HW %d)", N, C, HW);
  const int pblocks = cdiv(HW, 256);
  hipLaunchKernelGGL(records_mx_kernel, dim3((unsigned)(N * (C / 32) * pblocks)), dim3(256), 0, (hipStream_t)stream, x, xamax,
                     reinterpret_cast<int*>(rec), N, C, HW, pblocks);
  EOVAE_CHECK_LAUNCH("eovae_records_mx_from_f32");
  return EOVAE_OK;
}

// y [N][Cout][2 Hin][2 Win] = conv3x3(nearest_x2(x)) + bias, layers.py:47-50.  xrec = MX operand records of x
// (eovae_records_mx_from_f32 with the same xamax); wp = the f16c8 forward pack of eovae_subpix_expand(w) (4 Cout rows);
// gn_sums (nullable): [N][gn_groups][2] += (sum, sum of squares) of y per (image, group), as eovae_conv3x3_f16c8_fwd;
// out_amax (nullable, a pre-zeroed device word): float bits of max|y| by atomicMax from the epilogue.
int eovae_conv3x3_f16c8_up_mx(const float* xrec, const void* wp, const float* bias, float* y, const unsigned* xamax, int N, int Cin,
                              int Hin, int Win, int Cout, double* gn_sums, int gn_groups, unsigned* out_amax, void* stream) {
  EOVAE_CHECK_ARG(xrec && wp && y, "conv3x3_f16c8_up_mx: null pointer");
  EOVAE_CHECK_ARG(eovae_conv3x3_f16c8_up_mx_eligible(N, Cin, Hin, Win, Cout),
                  "conv3x3_f16c8_up_mx: shape not supported (N %d Cin %d %dx%d Cout %d; needs Cin %% 32, Cout %% 128, H %% 8, W %% 32)",
                  N, Cin, Hin, Win, Cout);
  ConvCArgs a;
  a.x = xrec; a.x6 = reinterpret_cast<const unsigned char*>(xrec) + (size_t)N * Cin * Hin * Win * 2;   // S plane behind the F plane
  a.wp = (const unsigned char*)wp; a.bias = bias; a.res = nullptr; a.y = y; a.xamax = xamax;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.Hout = 2 * Hin; a.Wout = 2 * Win;
  a.nchunk = Cin / 32; a.cpb = Cout / 128; a.ncob = 4 * a.cpb; a.creal = Cout;
  a.mode = MODE_NORMAL; a.subpix = 1; a.packed = 2; a.mx = 1; a.stagger = 0; a.skew = 0; a.out_amax = out_amax;
  a.gn_sums = gn_sums; a.gn_groups = gn_groups; a.gn_shift = 0;
  if (gn_sums) {
    const int cpg = gn_groups > 0 && Cout % gn_groups == 0 ? Cout / gn_groups : 0;
    EOVAE_CHECK_ARG(cpg >= 4 && cpg <= 64 && (cpg & (cpg - 1)) == 0,
                    "conv3x3_f16c8_up_mx: gn_sums needs Cout / gn_groups in {4, ..., 64} (Cout %d, groups %d)", Cout, gn_groups);
    while ((1 << a.gn_shift) < cpg) ++a.gn_shift;
  }
  a.tiles_x = Win / 32; a.tiles_y = Hin / 8;
  const long long sp_tiles = (long long)N * a.tiles_y * a.tiles_x;
  a.xcd_remap = (sp_tiles % 8 == 0) ? 1 : 0;
  EOVAE_CHECK_ARG(sp_tiles * 2 * a.cpb < (1LL << 30), "conv3x3_f16c8_up_mx: too many tiles");
  a.ntiles = (int)(sp_tiles * 2 * a.cpb);
  int dev = 0, n_cu = 256;
  hipDeviceProp_t prop;
  static int cached_cu = 0;
  if (cached_cu == 0) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    n_cu -= n_cu % 8;
    if (n_cu < 8) n_cu = 8;
    cached_cu = n_cu;
  }
  n_cu = cached_cu;
  const dim3 grid((unsigned)(a.ntiles < n_cu ? a.ntiles : n_cu)), blk(256);
  hipLaunchKernelGGL(conv3x3_f16c8_q16up_kernel, grid, blk, 0, (hipStream_t)stream, a);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_up_mx");
  return EOVAE_OK;
}

}  // extern "C"
