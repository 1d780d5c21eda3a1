// 3x3 stride-1 convolution on MX operand records -- ONE WAVE PER SIMD, 16 x 16 MFMAs, every byte fed by LDS-DMA (round 5).
//
// Same arithmetic, operand records, weight packs and C-ABI entry as conv_f16c8_w4.hip's MX instantiation
// (x*w ~= f16(x)*f16(w) + one block-scaled e2m3 correction product, conv_f16c8_common.h); same block tile (128 co x 16 rows
// x 32 px, four waves of 128 co x 4 rows) and persistent walk.  What changes:
//
//   MFMA shape     v_mfma_f32_16x16x32_f16 / v_mfma_scale_f32_16x16x128_f8f6f4 instead of the 32 x 32 forms: a wave's tile is
//                  8 x 8 accumulators of 16 x 16 (256 AGPRs, as before).  Same flops per cycle, same LDS bytes per flop -- but
//                  the chip holds a higher clock under a loop of the small shape (MI355X_MICROARCH.md 'DVFS give-back' (7);
//                  tools/probe_mfma_shape.hip: 1.85 against 1.66 GHz at equal cycles on this kernel's operand mix;
//                  profiles/r05t_w4_q16_stamps.log: 1.90 against 1.63 GHz inside conv_f16c8_w4.hip's phase sequence).
//   correction     the 16 x 16 x 128 form takes FOUR 32-element K blocks: [x_lo | x_hi] of tap t and of tap t + 1 against
//                  [w_hi | w_lo] of the same two taps (lane group g = lane / 16 supplies block g: tap t + g / 2, part g % 2 --
//                  its B address carries that tap's pixel shift).  Nine taps = four pairs + one tap whose second half is a
//                  zero weight unit.  A 32-channel chunk is 9 f16 phases + 5 correction phases of 64 MFMAs each.
//   feed           records need no conversion, so nothing passes through registers: the halo images (F plane -> f16 image,
//                  S plane -> FP6 image) and the weight units go global -> LDS by buffer_load ... lds.  No staging registers,
//                  no ds_write, no conversion VALU; a wave's instruction stream is MFMAs, fragment reads and a few DMA
//                  requests per phase.  Out-of-image halo pixels are lane offsets beyond the descriptor's range (zeros).
//   issue order    every MFMA and every fragment read is a volatile asm statement in program order; waits are counted by
//                  hand (lgkmcnt: the reads are the wave's only LDS instructions; vmcnt: DMA requests per phase are
//                  compile-time constants).  hipcc 7.2 cannot keep 64 four-register accumulators in place through the
//                  builtins (it renames destinations and repairs the loop's back edge with hundreds of v_accvgpr moves),
//                  and it does not schedule around inline-asm MFMAs.
//
// Phases of a chunk: p = 0..8 f16 product of tap p; p = 9..13 correction product of taps (2m, 2m + 1), m = p - 9.
//   weights of phase p + 3 are requested in phase p into slot (P + 3) % 4 of a four-slot ring (P = running phase number);
//   the FP6 image of the chunk is requested in phases 0..4 (first read: the B fragments of phase 9, requested in phase 8);
//   the f16 image of the NEXT chunk in phases 8..11 (its last reader of this chunk is phase 8, whose fragments were
//   requested in phase 7; first reader: phase 13, which requests phase 0's B fragments).
//   End of phase p: s_waitcnt vmcnt(requests of phase p) -- everything older has landed -- then the workgroup barrier.
//
// FP32IN (the data gradients: dY arrives as fp32): the same MFMA stream, the same LDS images -- written by the block itself.
//   A thread takes halo pixels tid, tid + 256, tid + 512 with ALL 32 channels of the chunk (one buffer_load_dword per channel:
//   64 consecutive pixels of a plane per instruction), so the 32-channel block scales of the MX-FP6 parts need no exchange
//   between threads.  One pixel is in flight at a time (32 registers); it is converted IN PLACE in ~40 steps of two to four
//   VALU instructions, one step behind each MFMA of a phase (pinned by scheduling barriers: hipcc would issue the block in one
//   piece and stall the matrix pipes for it), and leaves as four ds_write_b128.  The images are single-buffered and free
//   at different times, so a chunk is read TWICE (the second time from L2): pass 1 = f16 parts -> f16 image of the next chunk
//   (phases 8, 10, 12), pass 2 = block-scaled FP6 of x and of x - f16(x) -> FP6 image of this chunk (phases 0, 2, 4).
//   The weights still come by DMA.  conv_f16c8_w4.hip's fp32 path has the fp8 correction product (2 units); this one the
//   MX-FP6 one (1.64) on the shape the chip clocks higher on.
//   MEASURED SLOWER (profiles/r05z_q16f_*: 0.88-0.92 x conv_f16c8_w4.hip's fp32 path per layer): the seven staging phases of a chunk
//   take 2.1-3.1 k cycles instead of 1.2 k -- the 192 buffer_load_dword of a chunk cost ~30 cycles of issue each beside the MFMAs
//   (phase 6, which only loads, takes 2.3 k), the conversion steps the rest.  A planar fp32 input has no cheap way into per-pixel
//   32-channel blocks; the path stays behind option value 2, tested (tests/test_gpu_mx.py), as the record of that.
//
// Restrictions (conv_f16c8_w4.hip runs otherwise): those of that kernel; MX operand records, or fp32 input with library option
// c8_q16 = 2 (default 1: record inputs only).
#include <stdlib.h>

#include <type_traits>

#include "conv_f16c8_common.h"
#include "conv_f16c8_q16_asm.h"

namespace {

constexpr int QBM = 128, QXW = 34, QNPX = 18 * QXW, QNTASK = 10;
constexpr int QPPL = 640 * 16;                 // one octet's / piece's plane of the halo images: 640 pixels (612 + DMA padding)
constexpr int QIMG = 4 * QPPL;                 // 40,960
constexpr int QOFF_XF = 0, QOFF_X6 = QIMG, QOFF_W = 2 * QIMG;
constexpr int QWSLOT = 16384, QNWS = 4;        // weight ring: four phase slots of two 8 KB units
constexpr int QLDS = QOFF_W + QNWS * QWSLOT;   // 147,456
constexpr int QNPH = 14;                       // phases per chunk

// requests of a phase: weight DMAs for phase p + 3 (two per 8 KB unit), halo DMAs
__device__ __host__ constexpr int q_is_c(int p) { return ((p % QNPH) + QNPH) % QNPH >= 9; }
__device__ __host__ constexpr int q_nw(int p) { return q_is_c(p + 3) ? 4 : 2; }
__device__ __host__ constexpr int q_nh(int p) { return p <= 4 ? 2 : (p == 8 || p == 9) ? 3 : (p == 10 || p == 11) ? 2 : 0; }

// Diagnostic build only (-DEOVAE_Q16_STAMP=1, tools/q16_stamps.py): shader-clock stamps of block 0's waves per tile -- start,
// end of the phase sequence, end of the epilogue, 100-MHz clock -- in a buffer nothing else reads.
#ifndef EOVAE_Q16_STAMP
#define EOVAE_Q16_STAMP 0
#endif
#if EOVAE_Q16_STAMP
__device__ unsigned long long q16_phase_stamps[4 * 14 * 4];   // STAMP == 2: [wave][phase][barrier in | barrier out | MFMAs done | requests landed]
#define Q16_PSTAMP(p, id)                                                                                  \
  do {                                                                                                      \
    if (EOVAE_Q16_STAMP == 2 && blockIdx.x == 0 && lane == 0 && rec == 1 && chunk == 1)                     \
      q16_phase_stamps[(wave * 14 + (p)) * 4 + (id)] = __builtin_amdgcn_s_memtime();                        \
  } while (0)
__device__ unsigned long long q16_stamps[4 * 32 * 4];
#define Q16_STAMP(id)                                                                                       \
  do {                                                                                                      \
    if (blockIdx.x == 0 && lane == 0 && rec < 32)                                                           \
      q16_stamps[(wave * 32 + rec) * 4 + (id)] = (id) == 3 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define Q16_STAMP(id) do {} while (0)
#define Q16_PSTAMP(p, id) do {} while (0)
#endif
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
// placement of a pair's side work: index of the MFMA (0..15) it follows
#ifndef EOVAE_Q16_MM_DMA
#define EOVAE_Q16_MM_DMA 1
#endif
#ifndef EOVAE_Q16_MM_A
#define EOVAE_Q16_MM_A 3
#endif
#ifndef EOVAE_Q16_MM_B0
#define EOVAE_Q16_MM_B0 7
#endif
#ifndef EOVAE_Q16_MM_B1
#define EOVAE_Q16_MM_B1 11
#endif
template <bool FP32IN>
__global__ __launch_bounds__(256, 1) void conv3x3_f16c8_q16_kernel(ConvCArgs a) {
  // the DMA targets / fragment sources; the GroupNorm cells live in an array of their own so that the compiler does not
  // take their (plain) accesses for possible readers of a pending DMA and drain vmcnt(0) -- the epilogue's stores
  // included -- in front of them
  __shared__ __attribute__((aligned(16))) unsigned char smem[QLDS];
  __shared__ float gcell[4 * 2 * 4 * QBM];   // [wave][sum | sumsq][lane group][channel of the block]: with the images and the
                                             // weight ring exactly the CU's 160 KiB

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wrow = wave * 4;
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
    t.oy0 = ty * 16;
    t.ox0 = tx * 32;
    return t;
  };

  auto lds0_ = [&]() __attribute__((always_inline)) { return (unsigned)(size_t)smem; };
  float xs = 1.f, xinv = 1.f;   // range pre-scale of an fp32 input (amax to 2^13..2^14), undone with the result, as conv_f16c8_w4.hip
  if (a.xamax) {
    const unsigned bits = *a.xamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      xs = pow2f(13 - e);
      xinv = pow2f(e - 13);
    }
  }

  // ---------------------------------------------------------------- halo requests (wave = channel octet / FP6 piece)
  // task k of a wave: pixels 64 k + lane of the 18 x 34 halo; the lane's offset in its plane, or out of range (zeros)
  const unsigned OOR = 0x80000000u;
  unsigned xoff[QNTASK];
  auto setup_x = [&](const Tile& t) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // keeps the per-task index math out of the registers that live across the chunk loop
#pragma unroll
    for (int k = 0; k < QNTASK; ++k) {
      const int pix = ln + 64 * k;
      const int rr = (pix * 241) >> 13, cc = pix - QXW * rr;   // pix / 34, exact for pix < 640
      const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
      const bool ok = (pix < QNPX) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
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
  unsigned char* const xf_dst = smem + QOFF_XF + wave * QPPL;
  unsigned char* const x6_dst = smem + QOFF_X6 + wave * QPPL;
  auto dma_x = [&](const __amdgpu_buffer_rsrc_t& srd, unsigned char* dst, int k) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)(dst + k * 1024), 16, (int)xoff[k], 0, 0, 0);
  };

  // ---------------------------------------------------------------- fp32 input: register staging (FP32IN)
  unsigned foff[3];   // byte offset of the thread's k-th pixel in a channel plane of x, or out of range
  auto setup_f = [&](const Tile& t) __attribute__((always_inline)) {
    int th = tid;
    asm volatile("" : "+v"(th));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pix = th + 256 * k;
      const int rr = (pix * 241) >> 13, cc = pix - QXW * rr;   // pix / 34 (exact below 1024 here: 241 / 8192 = 1 / 33.99)
      const int vy = t.oy0 + rr - 1, vx = t.ox0 + cc - 1;
      const bool ok = (pix < QNPX) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
      foff[k] = ok ? (unsigned)(vy * a.Win + vx) * 4u : OOR;
    }
  };
  const unsigned plane4 = (unsigned)(in_plane * 4);
  auto srd_f = [&](const Tile& t, int chunk) __attribute__((always_inline)) {   // x [n][Cin][pixel] fp32, channels 32 chunk ..
    const float* p = a.x + ((size_t)t.n * a.Cin + 32 * chunk) * in_plane;
    // (the chunk's 32 planes: whether or not the range check adds the scalar offset, a valid lane stays inside and the
    // out-of-range marker 2^31 stays outside)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 32u * plane4, 0x00020000);
  };
  f32x16 va, vb;         // the pixel in flight: channels 0..15 | 16..31
  float ahi = 0.f, alo = 0.f;
  int hp[16];            // pass 1: the sixteen f16 pairs
  i32x4 q0, q1, q2, q3;  // pass 2: pieces of quant(x - f16 x) | quant(x)
  const unsigned wpx = lds0_() + (unsigned)tid * 16u;   // LDS address of pixel `tid` in plane 0 of the f16 image
  // two channels of the pixel's 32 per call: loads
  auto ld_step = [&](const __amdgpu_buffer_rsrc_t& srd, int k, auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;   // 0..15
    va[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, (int)foff[k], (int)(j * plane4), 0));
    vb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, (int)foff[k], (int)((16 + j) * plane4), 0));
  };
  // pass 1, step s of 36: scale, f16 pairs, four writes into the f16 image (octet o at o * QPPL)
  auto cv1_step = [&](int k, auto sc_) __attribute__((always_inline)) {
    constexpr int s_ = decltype(sc_)::value;
    if constexpr (s_ < 16) {
      va[s_] *= xs;
      vb[s_] *= xs;
    } else if constexpr (s_ < 32) {
      constexpr int q = s_ - 16;   // pair q = channels 2 q, 2 q + 1
      const float v0 = q < 8 ? va[2 * q] : vb[2 * q - 16], v1 = q < 8 ? va[2 * q + 1] : vb[2 * q - 15];
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp[q]) : "v"(v0), "v"(v1));
    } else if constexpr (s_ < 36) {
      constexpr int o = s_ - 32;
      i32x4 w;
      w[0] = hp[4 * o]; w[1] = hp[4 * o + 1]; w[2] = hp[4 * o + 2]; w[3] = hp[4 * o + 3];
      if (k < 2 || tid < 128) wr128<o * QPPL>(wpx + (unsigned)(k * 4096), w);   // (pixels beyond the plane's 640: no write)
    }
  };
  // pass 2, step s of 38: scale + amax, quant(x), residuals in place + amax, quant(residual), four writes into the FP6 image
  auto cv2_step = [&](int k, auto sc_) __attribute__((always_inline)) {
    constexpr int s_ = decltype(sc_)::value;
    if constexpr (s_ < 16) {
      if constexpr (s_ == 0) { ahi = 0.f; alo = 0.f; }
      va[s_] *= xs;
      vb[s_] *= xs;
      ahi = __builtin_fmaxf(ahi, __builtin_fmaxf(__builtin_fabsf(va[s_]), __builtin_fabsf(vb[s_])));
    } else if constexpr (s_ == 16) {
      mx6_quant16(va, vb, ahi, q2, q3);
    } else if constexpr (s_ < 33) {
      constexpr int q = s_ - 17;
      const float v0 = q < 8 ? va[2 * q] : vb[2 * q - 16];
      const float v1 = q < 8 ? va[2 * q + 1] : vb[2 * q - 15];
      int hh;
      float l0, l1;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(v0), "v"(v1));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hh), "v"(v0));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hh), "v"(v1));
      if constexpr (q < 8) { va[2 * q] = l0; va[2 * q + 1] = l1; }
      else { vb[2 * q - 16] = l0; vb[2 * q - 15] = l1; }
      alo = __builtin_fmaxf(alo, __builtin_fmaxf(__builtin_fabsf(l0), __builtin_fabsf(l1)));
    } else if constexpr (s_ == 33) {
      mx6_quant16(va, vb, alo, q0, q1);
    } else if constexpr (s_ < 38) {
      constexpr int pc_ = s_ - 34;   // piece 0, 1: residual; 2, 3: x
      const i32x4& w = pc_ == 0 ? q0 : (pc_ == 1 ? q1 : (pc_ == 2 ? q2 : q3));
      if (k < 2 || tid < 128) wr128<pc_ * QPPL>(wpx + (unsigned)(QOFF_X6 + k * 4096), w);
    }
  };

  // ---------------------------------------------------------------- weight requests: 8 KB units, two DMAs each
  const size_t wbytes = (size_t)9 * a.nchunk * a.ncob * C8_WTILE;
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wp), 0, (unsigned)wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t zsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wp), 0, 0u, 0x00020000);
  const int wvoff = tid * 16;
  auto dma_unit = [&](const __amdgpu_buffer_rsrc_t& srd, unsigned soff, unsigned slotbyte) __attribute__((always_inline)) {
    unsigned char* dst = smem + QOFF_W + slotbyte + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)dst, 16, wvoff, (int)soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr)(dst + 4096), 16, wvoff, (int)(soff + 4096u), 0, 0);
  };
  // units of phase q (0..13) of chunk `chunk` of co-block `cob` into ring slot `slot`
  auto request_w = [&](auto qc, int cob, int chunk, unsigned slot) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    const unsigned sb = slot * QWSLOT;
    if constexpr (q < 9) {
      dma_unit(wsrd, (unsigned)(((size_t)(q * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE), sb);
    } else {
      constexpr int t0 = 2 * (q - 9);
      dma_unit(wsrd, (unsigned)(((size_t)(t0 * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE + C8_WOFF6), sb);
      if constexpr (t0 + 1 < 9)
        dma_unit(wsrd, (unsigned)(((size_t)((t0 + 1) * a.nchunk + chunk) * a.ncob + cob) * C8_WTILE + C8_WOFF6), sb + 8192);
      else
        dma_unit(zsrd, 0u, sb + 8192);   // the ninth tap has no partner: a unit of zeros (descriptor of zero records)
    }
  };

  // ---------------------------------------------------------------- fragments
  // lane (li, g) of a 16 x 16 MFMA: A row / B column li, K block g.
  //   f16:  K block g = channel octet g of the chunk.   A: weight unit [octet][row][16 B]; B: f16 image [octet][pixel][16 B]
  //   FP6:  K block g = (tap t + g / 2, part g % 2).     A: units of the two taps [tap][part][piece][row][16 B]; B: FP6 image
  //         pieces 2 (g % 2), 2 (g % 2) + 1 at the pixel shifted by THAT tap
  const unsigned lds0 = lds0_();   // the asm reads take LDS byte addresses
  const unsigned a_f = lds0 + (unsigned)(QOFF_W + g * 2048 + li * 16);
  const unsigned a_c = lds0 + (unsigned)(QOFF_W + (g >> 1) * 8192 + (g & 1) * 4096 + li * 16);
  const unsigned b_f = lds0 + (unsigned)(QOFF_XF + g * QPPL + (wrow * QXW + li) * 16);
  const unsigned b_c = lds0 + (unsigned)(QOFF_X6 + 2 * (g & 1) * QPPL + (wrow * QXW + li) * 16);
  const unsigned gsel = (unsigned)(g >> 1);

  f32x4 acc[8][8];   // [co tile of 16][pixel tile: row 0..3 x half 0..1]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  };
  // fragment registers: A streamed by co-tile pairs [pair parity][tile][lo | hi], B held per phase [phase parity][tile][lo | hi]
  // FP6 fragments: 16 bytes of codes in *r[..], the remaining 8 bytes + the scale dword in *h[..] (ds_read_b96: the pad dword of
  // a 32-byte operand is never read -- an eighth less LDS traffic in the correction phases, which the LDS paces)
  i32x4 Ar[2][2], Br[2][8];
  i32x3 Ah[2][2], Bh[2][8];

  // A fragments of co-tile pair c2 of a phase whose ring slot starts at byte `sb` (runtime), into Ar[c2 & 1]
  auto read_a = [&](auto isc, auto c2c, unsigned sb) __attribute__((always_inline)) {
    constexpr bool ISC = decltype(isc)::value;
    constexpr int c2 = decltype(c2c)::value;
    const unsigned base = (ISC ? a_c : a_f) + sb;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (ISC) {
        if (t == 0) { rd128<(2 * c2) * 256>(Ar[c2 & 1][0], base); rd96<(2 * c2) * 256 + 2048>(Ah[c2 & 1][0], base); }
        else { rd128<(2 * c2 + 1) * 256>(Ar[c2 & 1][1], base); rd96<(2 * c2 + 1) * 256 + 2048>(Ah[c2 & 1][1], base); }
      } else {
        if (t == 0) rd128<(2 * c2) * 256>(Ar[c2 & 1][0], base);
        else rd128<(2 * c2 + 1) * 256>(Ar[c2 & 1][1], base);
      }
    }
  };
  // B fragment of pixel tile j (row j >> 1, half j & 1) of phase q into Br[q & 1][j]
  auto read_b = [&](auto qc, auto jc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, j = decltype(jc)::value, jr = j >> 1, hx = j & 1;
    if constexpr (q < 9) {
      constexpr int ky = q / 3, kx = q - 3 * ky;
      rd128<((ky + jr) * QXW + kx + 16 * hx) * 16>(Br[q & 1][j], b_f);
    } else {
      constexpr int t0 = 2 * (q - 9), ky = t0 / 3, kx = t0 - 3 * ky;
      constexpr int t1 = t0 + 1 < 9 ? t0 + 1 : t0, ky1 = t1 / 3, kx1 = t1 - 3 * ky1;
      constexpr int delta = ((ky1 - ky) * QXW + (kx1 - kx)) * 16;   // the second tap's pixel shift against the first's
      const unsigned base = b_c + gsel * (unsigned)delta;
      rd128<((ky + jr) * QXW + kx + 16 * hx) * 16>(Br[q & 1][j], base);
      rd96<((ky + jr) * QXW + kx + 16 * hx) * 16 + QPPL>(Bh[q & 1][j], base);
    }
  };
  // MFMA (co tile i, pixel tile j) of phase q
  auto mma = [&](auto qc, auto ic, auto jc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value, j = decltype(jc)::value;
    constexpr int pr = (i >> 1) & 1, t = i & 1;
    // (pixels as the MFMA's A operand, weights as its B operand: a lane's four result registers are then four CONSECUTIVE
    // PIXELS of one channel -- the epilogue stores them as they are)
    if constexpr (q < 9) {
      mfma_f(acc[i][j], Br[q & 1][j], Ar[pr][t]);
    } else {
      // a fragment = 24 data bytes + scale dword: the operand is a six-register tuple (16 + 8 bytes), the scale the last dword
      // of the second read
      i32x6 a6, b6;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a6[e] = Ar[pr][t][e]; b6[e] = Br[q & 1][j][e]; }
      a6[4] = Ah[pr][t][0]; a6[5] = Ah[pr][t][1];
      b6[4] = Bh[q & 1][j][0]; b6[5] = Bh[q & 1][j][1];
      mfma_c<(q == QNPH - 1 && i >= 6 && j == 7)>(acc[i][j], b6, a6, Bh[q & 1][j][2], Ah[pr][t][2]);
    }
  };

  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  // The fragment reads are asm statements: to the compiler their results exist as soon as the statement is passed, and it
  // may schedule a consumer of its own (the v_mov copies that build the FP6 operands' six-register tuples) in front of the
  // hand-placed s_waitcnt -- i.e. copy registers the LDS has not written yet.  Behind every wait the registers it covers
  // are therefore passed through an empty asm statement: their consumers depend on THAT.
  auto landed_a = [&](int pr) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      asm volatile("" : "+v"(Ar[pr][t]));
      asm volatile("" : "+v"(Ah[pr][t]));
    }
  };
  auto landed_b = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("" : "+v"(Br[par][j]));
      asm volatile("" : "+v"(Bh[par][j]));
    }
  };

  // ---------------------------------------------------------------- GroupNorm partial sums
  auto gn_flush = [&](int tn, int tcb) __attribute__((always_inline)) {
    const int te = wave * 64 + fresh_lane();
    const int v = te >> 7, c = te & 127;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {   // fixed order
      const float* cw = gcell + ((w * 2 + v) * 4) * QBM + c;
      t += (cw[0] + cw[QBM]) + (cw[2 * QBM] + cw[3 * QBM]);
    }
    const int cpg = 1 << a.gn_shift;   // 4..64 channels per group (eligibility): inside one wave's 64 channels
    for (int s = 1; s < cpg && s < 64; s <<= 1) t += __shfl_xor(t, s, 64);
    if ((c & (cpg - 1)) == 0) atomicAdd(a.gn_sums + ((size_t)tn * a.gn_groups + ((tcb + c) >> a.gn_shift)) * 2 + v, (double)t);
  };

  // ---------------------------------------------------------------- epilogue
  // acc[i][j][q]: channel 16 i + li, pixel (row j >> 1, column 16 (j & 1) + 4 g + q): four consecutive pixels of one channel
  // per lane, so every load / store moves 16 bytes per lane (a wave instruction covers 16 channel planes x 64 bytes).
  // WHICH lane holds which 16 bytes decides the rate (round 6, tools/probe_store_pattern.hip, profiles/r06o_*): the memory
  // pipeline takes a wave's store one QUAD of consecutive lanes at a time, and a quad whose four lanes go to four different
  // lines (the MFMA layout: consecutive lanes = consecutive channels) costs four times the quad that writes one 64-byte
  // segment -- 16 against 38..40 bytes per clock and CU.  Every accumulator register is therefore passed through one
  // ds_bpermute_b32 (lane 4 c + p takes the value of lane 16 p + c: channel c = lane >> 2, pixel group p = lane & 3; the
  // crossbar only, no LDS bytes) in front of the bias / residual / statistics arithmetic, which runs in the new layout.
  // Every vector-memory instruction of the epilogue is an asm statement with hand-counted waits.  The queue of one co tile i
  // (16 channels x the wave's 4 rows x 32 pixels = 8 accumulators) is: residual loads of tile i + 2 (8, only when the layer
  // has a residual), stores of tile i (8) -- so tile i's residual has landed at vmcnt(32) with the two previous tiles' stores
  // and the next tile's loads still in flight.  (With compiler-counted loads the asm stores were invisible to its wait insertion: "vmcnt(8)" drained
  // the previous tile's stores in front of every co tile, eight write latencies per tile; and a layer WITHOUT a residual
  // still issued the 64 zero-range loads -- 16 k of the epilogue's 26 k cycles, profiles/r05w_q16_epi_abl.log.)
  auto epilogue_tile = [&](const Tile& t, auto resc) __attribute__((always_inline)) {
    constexpr bool RES = decltype(resc)::value;
    // (nothing is outstanding here -- the last phase's asm wait -- but the compiler's own scoreboard still holds the main
    // loop's DMA requests as pending LDS writes and would drain vmcnt(0) in front of a later LDS access of its own, the
    // statistics cells behind the stores: a wait it can see, where it costs nothing)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    const int co_blk = t.cob * QBM;
    const unsigned span = (unsigned)(QBM * out_plane * 4);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
        a.y + ((size_t)t.n * a.Cout + co_blk) * out_plane, 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? a.res + ((size_t)t.n * a.Cout + co_blk) * out_plane : a.y), 0, RES ? span : 0u, 0x00020000);
    // (no branch for an absent bias: a descriptor of zero bytes, the loads return 0)
    const __amdgpu_buffer_rsrc_t bsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias + co_blk : a.y), 0, a.bias ? (unsigned)(QBM * 4) : 0u, 0x00020000);
    const int le = fresh_lane();
    const int ec = EOVAE_EPI_QUADS ? le >> 2 : le & 15, ep = EOVAE_EPI_QUADS ? le & 3 : le >> 4;   // channel, pixel group
    const int psrc = ((ep << 4) | ec) << 2;
    const int lot = (int)(((size_t)ec * out_plane + (size_t)(t.oy0 + wrow) * a.Wout + t.ox0 + 4 * ep) * 4);
    const int lob = ec * 4;
    const int tile16 = (int)(16 * out_plane * 4), rowb = a.Wout * 4;
    float* const cell = gcell + wave * 8 * QBM + ep * QBM + ec;
    float bv[8];
    static_for<0, 8>([&](auto ic) __attribute__((always_inline)) { load32_asm<decltype(ic)::value * 64>(bv[decltype(ic)::value], bsrd, lob); });
    auto finish = [&](f32x4& o, float& sv, float& qv, int i, int j) __attribute__((always_inline)) {
      if (!(EOVAE_Q16_ABL & 16)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sv += o[k];
          qv = __builtin_fmaf(o[k], o[k], qv);
        }
      }
      if (!(EOVAE_Q16_ABL & 2))   // (guarded store: a VALU write of the data registers right behind a register-soffset store is a
                                  // hazard hipcc does not cover on gfx950 -- conv_f16c8_common.h, tools/probe_store_war.hip)
        store128_guarded(o, ysrd, lot, i * tile16 + (j >> 1) * rowb + (j & 1) * 64);
      else
        asm volatile("" :: "v"(o));
    };
    // statistics of the following GroupNorm: every lane's partial (its four pixels x eight pixel tiles of one channel), folded
    // over pixel groups and waves by gn_flush().  Written unconditionally, read only when the caller asked for them.
    // Unit u = half a co tile (co tile u >> 1, pixel tiles 4 (u & 1) .. + 3: 16 registers).  Two things run AHEAD of the
    // arithmetic, as asm statements with counted waits:
    //   the lane permutation, one unit (issued where the compiler places the builtin -- in front of the values' use -- every
    //   batch of eight exposed the LDS latency);
    //   the residual loads, RA units (one co tile ahead left ~17 k cycles of exposed load latency per tile,
    //   profiles/r06o_q16_stamps_pipelined.log).  Queue: L0 .. L(RA-1) | L(RA) S0 | L(RA+1) S1 | ...: behind L(u) lie the
    //   stores of units u - RA .. u - 1 and the loads of units u + 1 .. u + RA.
    constexpr int RA = EOVAE_Q16_RES_AHEAD;
    f32x4 rv[RA + 1][4];
    auto load_res = [&](f32x4 (&r)[4], int u) __attribute__((always_inline)) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * (u & 1) + jj;
        if constexpr (RES) load128_asm(r[jj], rsrd, lot, (u >> 1) * tile16 + (j >> 1) * rowb + (j & 1) * 64);
        else r[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    static_for<0, RA>([&](auto uc) __attribute__((always_inline)) { load_res(rv[decltype(uc)::value], decltype(uc)::value); });
    float pv[2][4][4];
    auto perm_issue = [&](float (&d)[4][4], int u) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = acc[u >> 1][4 * (u & 1) + j][k];   // (copied out: hipcc 7.2's __builtin_bit_cast of a vector ELEMENT expression reads element 0)
          if (EOVAE_EPI_QUADS && !(EOVAE_Q16_ABL & 8)) bpermute_asm(d[j][k], psrc, a0);
          else d[j][k] = a0;
        }
    };
    perm_issue(pv[0], 0);
    float sv = 0.f, qv = 0.f;
    static_for<0, 16>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, i = u >> 1, hf = u & 1;
      wait_lgkm<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) landed(pv[u & 1][j][k]);
      if constexpr (u + 1 < 16) perm_issue(pv[(u + 1) & 1], u + 1);
      if constexpr (u + RA < 16) load_res(rv[(u + RA) % (RA + 1)], u + RA);
      if constexpr (RES) {
        wait_vm<4 * (u < RA ? u : RA) + 4 * ((u + RA < 15 ? u + RA : 15) - u)>();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) landed(rv[u % (RA + 1)][jj]);
      } else if constexpr (u == 0) {
        wait_vm<0>();
      }
      if constexpr (u == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) landed(bv[k]);
      }
      if constexpr (hf == 0) {
        sv = 0.f;
        qv = 0.f;
      }
      const float bias1 = bv[i];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(pv[u & 1][jj][k], xinv, bias1) + rv[u % (RA + 1)][jj][k];
        finish(o, sv, qv, i, 4 * hf + jj);
      }
      if constexpr (hf == 1) {
        cell[16 * i] = sv;
        cell[4 * QBM + 16 * i] = qv;
      }
    });
  };

  // ---------------------------------------------------------------- prologue
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  Tile cur = decode(tile);
  unsigned P = 0;   // running phase number (mod 4 = ring slot)
  if constexpr (FP32IN) {
    // the first chunk's f16 image (pass 1 of its three pixels per thread), then its first pixel again for pass 2 (phase 0)
    setup_f(cur);
    const __amdgpu_buffer_rsrc_t s0 = srd_f(cur, 0);
    static_for<0, 3>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      static_for<0, 16>([&](auto jc) __attribute__((always_inline)) { ld_step(s0, k, jc); });
      static_for<0, 36>([&](auto sc_) __attribute__((always_inline)) { cv1_step(k, sc_); });
    });
    static_for<0, 16>([&](auto jc) __attribute__((always_inline)) { ld_step(s0, 0, jc); });
  } else {
    setup_x(cur);
    const __amdgpu_buffer_rsrc_t s0 = srd_xf(cur, 0);
#pragma unroll
    for (int k = 0; k < QNTASK; ++k) dma_x(s0, xf_dst, k);
  }
  {
    request_w(std::integral_constant<int, 0>{}, cur.cob, 0, 0u);
    request_w(std::integral_constant<int, 1>{}, cur.cob, 0, 1u);
    request_w(std::integral_constant<int, 2>{}, cur.cob, 0, 2u);
  }
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  {
    using Q0 = std::integral_constant<int, 0>;
    read_a(std::false_type{}, Q0{}, 0u);
    static_for<0, 8>([&](auto jc) __attribute__((always_inline)) { read_b(Q0{}, jc); });
  }

  // ---------------------------------------------------------------- main loop
#if EOVAE_Q16_STAMP
  int rec = 0;
#endif
  for (; tile < a.ntiles; tile += gridDim.x) {
    Q16_STAMP(0);
    Q16_STAMP(3);
    const bool has_next_tile = tile + (int)gridDim.x < a.ntiles;
    Tile nxt = cur;
    if (has_next_tile) nxt = decode(tile + gridDim.x);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool last_chunk = chunk + 1 == a.nchunk;
      // the chunk whose f16 image and first weights are requested behind this one: the next chunk, the next tile's first, or
      // (end of work) this chunk again -- re-requesting it writes the bytes that are already there
      const bool wrap = last_chunk && has_next_tile;
      const Tile& xt = wrap ? nxt : cur;
      const int xchunk = last_chunk ? (has_next_tile ? 0 : chunk) : chunk + 1;
      const __amdgpu_buffer_rsrc_t s6 = FP32IN ? srd_f(cur, chunk) : srd_x6(cur, chunk);   // FP32IN: this chunk's planes (pass 2)
      __amdgpu_buffer_rsrc_t sf = s6;   // the next chunk's source (set at phase 5 / 7)
      static_for<0, QNPH>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int pn = (p + 1) % QNPH;
        constexpr bool ISC = p >= 9, NISC = pn >= 9;
        using PN = std::integral_constant<int, pn>;
        Q16_PSTAMP(p, 0);
        barrier();   // phase p's weights, the images it reads and its first fragments are there
        Q16_PSTAMP(p, 1);
        landed_a(0);
        landed_b(p & 1);
        const unsigned sb = ((P + p) & 3u) * QWSLOT, sbn = ((P + p + 1) & 3u) * QWSLOT;
        if constexpr (!FP32IN && p == 7) {
          if (wrap) setup_x(nxt);   // uniform; this chunk's FP6 image was requested in phases 0..4
          sf = srd_xf(xt, xchunk);
        }
        if constexpr (FP32IN && p == 5) {
          if (wrap) setup_f(nxt);   // uniform; this chunk's last pass-2 loads went out in phase 2
          sf = srd_f(xt, xchunk);
        }
        static_for<0, 4>([&](auto c2c) __attribute__((always_inline)) {
          constexpr int c2 = decltype(c2c)::value;
          static_for<0, 16>([&](auto mc) __attribute__((always_inline)) {
            constexpr int mm = decltype(mc)::value;
            constexpr int j = mm >> 1, i = 2 * c2 + (mm & 1);
            mma(pc, std::integral_constant<int, i>{}, std::integral_constant<int, j>{});
            // the phase's other work, one piece behind every MFMA pair
            if constexpr (mm == EOVAE_Q16_MM_DMA) {
              if constexpr (c2 == 0) {   // weights of phase p + 3
                if constexpr (p + 3 < QNPH) request_w(std::integral_constant<int, p + 3>{}, cur.cob, chunk, (P + p + 3) & 3u);
                else request_w(std::integral_constant<int, p + 3 - QNPH>{}, xt.cob, xchunk, (P + p + 3) & 3u);
              }
              if constexpr (c2 == 1 && !FP32IN) {   // halo requests of this phase
                if constexpr (p <= 4) { dma_x(s6, x6_dst, 2 * p); dma_x(s6, x6_dst, 2 * p + 1); }
                if constexpr (p == 8) { dma_x(sf, xf_dst, 0); dma_x(sf, xf_dst, 1); dma_x(sf, xf_dst, 2); }
                if constexpr (p == 9) { dma_x(sf, xf_dst, 3); dma_x(sf, xf_dst, 4); dma_x(sf, xf_dst, 5); }
                if constexpr (p == 10) { dma_x(sf, xf_dst, 6); dma_x(sf, xf_dst, 7); }
                if constexpr (p == 11) { dma_x(sf, xf_dst, 8); dma_x(sf, xf_dst, 9); }
              }
            }
            if constexpr (mm == EOVAE_Q16_MM_A) {     // A fragments of the next co-tile pair (of the next phase behind the last pair)
              if constexpr (c2 < 3) read_a(std::integral_constant<bool, ISC>{}, std::integral_constant<int, c2 + 1>{}, sb);
              else read_a(std::integral_constant<bool, NISC>{}, std::integral_constant<int, 0>{}, sbn);
            }
            if constexpr (mm == EOVAE_Q16_MM_B0) read_b(PN{}, std::integral_constant<int, 2 * c2>{});
            if constexpr (mm == EOVAE_Q16_MM_B1) read_b(PN{}, std::integral_constant<int, 2 * c2 + 1>{});
            if constexpr (mm == 15) {    // the next pair's A fragments have landed; the B reads behind them may be outstanding
              if constexpr (c2 < 3) {
                wait_lgkm<NISC ? 4 : 2>();
                landed_a((c2 + 1) & 1);
              }
            }
            if constexpr (FP32IN) {
              // staging step s = 16 c2 + mm of this phase (schedule in the file header): conversion steps 0..37, then the next
              // pixel's loads in steps 40..55; phase 6 only loads
              constexpr int st = 16 * c2 + mm;
              constexpr int kcv = p == 0 || p == 8 ? 0 : (p == 2 || p == 10 ? 1 : 2);
              if constexpr (p == 0 || p == 2 || p == 4) {
                if constexpr (st < 38) cv2_step(kcv, std::integral_constant<int, st>{});
              }
              if constexpr (p == 8 || p == 10 || p == 12) {
                if constexpr (st < 36) cv1_step(kcv, std::integral_constant<int, st>{});
              }
              if constexpr (st >= 40 && st < 56) {
                if constexpr (p == 0) ld_step(s6, 1, std::integral_constant<int, st - 40>{});
                if constexpr (p == 2) ld_step(s6, 2, std::integral_constant<int, st - 40>{});
                if constexpr (p == 8) ld_step(sf, 1, std::integral_constant<int, st - 40>{});
                if constexpr (p == 10) ld_step(sf, 2, std::integral_constant<int, st - 40>{});
                if constexpr (p == 12) ld_step(sf, 0, std::integral_constant<int, st - 40>{});
              }
              if constexpr (p == 6 && st < 16) ld_step(sf, 0, std::integral_constant<int, st>{});
              if constexpr (p == 0 || p == 2 || p == 4 || p == 6 || p == 8 || p == 10 || p == 12) __builtin_amdgcn_sched_barrier(0);
            }
          });
        });
        Q16_PSTAMP(p, 2);
        // everything requested before this phase has landed (its own requests may be outstanding).  Behind a tile's last
        // phase: all of it -- the epilogue's stores come next in the queue, and a counted wait behind them would wait
        // for them too; the first phase of a tile therefore does not wait (its needs were met here).
        constexpr int NREQ = q_nw(p) + (FP32IN ? ((p == 0 || p == 2 || p == 6 || p == 8 || p == 10 || p == 12) ? 32 : 0) : q_nh(p));
        if constexpr (p == QNPH - 1) {
          if (last_chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // The next phase-0 fragments (Ar[0], Ah[0], Br[0], Bh[0]) were requested by asm reads in this phase.  To the
            // compiler they exist already, so the compiler-scheduled epilogue behind this loop may copy or spill those
            // registers while the LDS return is still pending (hazard (i) of the file header; it would hit co-tile pair 0).
            // Land them here: a few hundred cycles against a ~100 k-cycle tile.
            wait_lgkm<0>();
            landed_a(0);
            landed_b(0);
          } else wait_vm<NREQ>();
        } else if constexpr (p == 0) {
          if (chunk != 0 || (EOVAE_Q16_SAFE & 1)) wait_vm<NREQ>();
        } else {
          wait_vm<NREQ>();
        }
        Q16_PSTAMP(p, 3);
      });
      P += QNPH;
    }
    Q16_STAMP(1);
    if (EOVAE_Q16_ABL & 4) {
      Tile fake = cur;
      fake.n = (int)(blockIdx.x % a.N); fake.oy0 = 0; fake.ox0 = (int)((blockIdx.x / a.N) % a.tiles_x) * 32; fake.cob = 0;
      epilogue_tile(fake, std::false_type{});
    } else if (a.res) {   // uniform
      epilogue_tile(cur, std::true_type{});
    } else {
      epilogue_tile(cur, std::false_type{});
    }
    Q16_STAMP(2);
#if EOVAE_Q16_STAMP
    ++rec;
#endif
    zero_acc();
    if (a.gn_sums) {   // uniform.  The cells are rewritten by the next epilogue, a whole tile later.
      barrier();       // (raw: __syncthreads() carries a release fence = vmcnt(0), i.e. a wait for the epilogue's stores)
      gn_flush(cur.n, cur.cob * QBM);
    }
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests behind the last chunk
}

}  // namespace

#if EOVAE_Q16_STAMP
extern "C" int eovae_debug_q16_phase_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(q16_phase_stamps), bytes < sizeof(q16_phase_stamps) ? bytes : sizeof(q16_phase_stamps));
}
extern "C" int eovae_debug_q16_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(q16_stamps), bytes < sizeof(q16_stamps) ? bytes : sizeof(q16_stamps));
}
#endif

// (a.x6 is not looked at: eovae_conv3x3_f16c8_which() asks without pointers; the entry sets it for every packed == 2 call)
bool eovae_c8_q16_eligible(const ConvCArgs& a) {
  const int opt = eovae_opt(EOVAE_OPT_C8_Q16);   // 0: never, 1: MX record inputs, 2: fp32 inputs (data gradients) too
  return opt != 0 && (a.packed == 2 || (EOVAE_PARKED && a.packed == 0 && opt >= 2 && eovae_opt(EOVAE_OPT_C8_MX))) && a.Hout % 16 == 0 &&
         (!a.gn_sums || a.gn_shift <= 6) && (size_t)a.Hin * a.Win * 32 * 4 < (1ull << 31) &&   // (the fp32 path's out-of-range marker 2^31 must lie beyond a chunk's 32 planes)
         // the epilogue's 32-bit byte arithmetic: a co-block's span of y (descriptor range) and 7 * tile16 + lot stay below 2^31
         (size_t)QBM * a.Hout * a.Wout * 4 < (1ull << 31) &&
         (size_t)a.Hin * a.Win * 16 < (1ull << 31) && (size_t)9 * a.nchunk * a.ncob * C8_WTILE < (1ull << 32);
}

// a: prepared by eovae_c8_w4_launch (tiles, ntiles, xcd_remap)
void eovae_c8_q16_launch(const ConvCArgs& a, int n_cu, hipStream_t st) {
  const dim3 grid((unsigned)(a.ntiles < n_cu ? a.ntiles : n_cu)), blk(256);
#if EOVAE_PARKED   // the fp32-input path (measured 0.88-0.92 x conv_f16c8_w4.hip's): compiled only into parked builds
  if (a.packed != 2) {
    hipLaunchKernelGGL(conv3x3_f16c8_q16_kernel<true>, grid, blk, 0, st, a);
    return;
  }
#endif
  hipLaunchKernelGGL(conv3x3_f16c8_q16_kernel<false>, grid, blk, 0, st, a);
}
