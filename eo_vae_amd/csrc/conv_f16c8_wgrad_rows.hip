// Weight gradient of a plain 3x3 layer with f16c8 operands -- ROW-STREAMING form (round 4).
//   dW[co][ci][ky][kx] = sum_{n, y, x} dY[n][co][y][x] * X[n][ci][y + ky - 1][x + kx - 1]
// Same arithmetic per product as wgrad3x3_f16c8_kernel (conv_f16c8.hip): per 32 pixels of one image row and tap
//   2 x v_mfma_f32_32x32x16_f16 (dy_hi . x_hi)  +  1 x v_mfma_scale_f32_32x32x64_f8f6f4 ([dy_lo' | dy_hi'] . [x_hi' | x_lo']),
// same block tile (128 co x 64 ci x 9 taps, 8 waves), same partial-tile output + fold.  What changes is the walk:
//
//   tile kernel (conv_f16c8.hip)            a block stages 2 output rows x 32 pixels of dY and the 4 x 34 halo rows of X,
//                                           multiplies, moves on: X is staged 2 x, every (tap, row) pair re-reads its
//                                           fragments from LDS (52 fragment reads per 72 MFMAs), both operands sit
//                                           channel-contiguous and are gathered by transposing reads, a wave owns 2
//                                           co-tiles x 5 | 4 taps.
//   this kernel                             a block walks DOWN a strip of 32 pixel columns, one image row per stage.
//                                           Stage R stages dY row R + 1 and X row R ONCE (X 1 x + 2 halo columns) and
//                                           multiplies X row R with dY rows R + 1, R, R - 1 (ky = 0, 1, 2): the three dY
//                                           rows' A fragments live in REGISTERS across stages (a ring of three), so a stage
//                                           reads 4 + 24 LDS instructions per 27 MFMAs and wave (tile kernel: 56).  A wave
//                                           owns ONE co-tile x one ci-tile x all NINE taps (144 accumulators): the two
//                                           waves of a SIMD carry the same MFMA load.
//   dY, which takes no tap shift, is staged K-contiguous ([co][32 pixels], as it lies in memory: two 16-byte loads per
//   thread and stage) and read as the MFMA A operand by plain ds_read_b128 -- conflict-free with an odd 16-byte row pitch;
//   only X, whose window moves with kx, needs the transposing reads.
//   X arrives as GroupNorm operand records and goes global -> LDS by LDS-DMA (two buffer_load ... lds per wave and
//   stage, 34 live lanes each: the f16 halves and the fp8 halves of one channel octet's 34 pixels; off-image pixels are
//   offsets beyond the descriptor's range, off-image rows a descriptor of zero records: the hardware fills in zeros): no
//   registers, no VALU, no ds_write.
//
// A stage is paced by instruction ISSUE as much as by the matrix pipes (timing-only ablations, profiles/r04g: without the
// MFMAs the first build still took 0.65-0.74 of its time: ~240 instructions per wave and stage, 120 of them on the CU's
// one scalar unit).  So the stage body carries nothing it can avoid: the walk is INCREMENTAL (a row step is two scalar
// adds; descriptors' bases and the per-lane DMA offset change only when a unit opens, in a rarely taken branch), the MFMAs
// are volatile asm statements in a fixed order with the other work placed between them by hand, the first MFMAs of a
// stage run on operands that are in registers when the barrier opens.
//
// LDS (135 KiB): three dY slots {f16 [128 co][80 B] | dyl' [128][48 B] | dyh' [128][48 B]} and four X slots
// {f16 planes [8 octets][34 px][16 B], pitch 1088 = 64 (mod 256) | fp8 planes [8 octets][34 px][xl' x 8 | xh' x 8] at
// offsets = {0, 136, 8, 128} (mod 256) per ci-tile}: every fragment read of a 32-lane group covers 64 distinct banks, and
// the dY stores of a 16-lane group (rows co, co + 4, co + 2, co + 6) cover 32.
// The stage loop is unrolled by twelve so that slot, ring and register-set indices are compile-time.
//
// Restrictions (else the tile kernel runs): plain form (subpix 0, mode NORMAL), record input, partial-tile output,
// Cout % 128 == 0, Cin % 64 == 0, Wout % 32 == 0.
#include <stdlib.h>

#include <type_traits>

#include "conv_f16c8_common.h"

#ifndef EOVAE_ROWS_DBG
#define EOVAE_ROWS_DBG 0   // bisection switches: 1 linear dY row map, 2 zeroed `old` operands of the fp8 conversions
#endif
#ifndef EOVAE_ROWS_ROLES
#define EOVAE_ROWS_ROLES 0   // 1: waves 4-7 request / convert one kx later than waves 0-3 -- measured 3-5 % SLOWER (profiles/r04j_roles_ab.log:
                             // a second copy of the 12-stage loop, 34 spilled registers); the two waves of a SIMD run in step
#endif
#ifndef EOVAE_ROWS_ABL
#define EOVAE_ROWS_ABL 0   // timing-only experiments (WRONG results): 1 no dY conversion, 2 no requests (loads, DMAs, cursor), 4 no barrier, 8 no MFMAs
#endif

namespace {

constexpr int P16 = 80, P8 = 48;             // dY row pitches (bytes): 64 + 16, 32 + 16 -- odd multiples of 16
constexpr int DY16 = 128 * P16;              // f16 image of one dY row
constexpr int DY8 = 128 * P8;                // one fp8 image
constexpr int DYSLOT = DY16 + 2 * DY8;       // 22,528
// Every LDS-DMA instruction deposits 64 lanes x 16 B = 1 KiB (lanes 34..63 carry an out-of-range offset and deposit
// zeros): each octet's plane owns a 1 KiB window, so the instruction needs no EXEC mask and the request block no branch.
constexpr int XP16 = 1088;                   // f16 plane pitch per octet: 1 KiB window + 64 (= 64 mod 256)
constexpr int X16 = 8 * XP16;                // 8,704
constexpr int X8T = 4480;                    // fp8 planes of one ci-tile (4 octets, 1 KiB windows at the offsets of x8off())
constexpr int XSLOT = X16 + 2 * X8T;         // 17,664
constexpr int NRING = 3;

// windows of 1 KiB at {0, 136, 8, 128} (mod 256), at least 1 KiB apart
__device__ __forceinline__ constexpr int x8off(int o) { return o == 0 ? 0 : (o == 1 ? 1160 : (o == 2 ? 2312 : 3456)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 rtr16(const unsigned char* p0, const unsigned char* p1) {
  using lds4 = __attribute__((address_space(3))) s16x4;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p1));
  return __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ i32x8 rtr8(const unsigned char* p) {   // 4 x 8 pixels, 16 B per pixel
  using lds2 = __attribute__((address_space(3))) i32x2;
  const i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p));
  const i32x2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 128));
  const i32x2 v2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 256));
  const i32x2 v3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 384));
  const i32x4 lo = __builtin_shufflevector(v0, v1, 0, 1, 2, 3), hi = __builtin_shufflevector(v2, v3, 0, 1, 2, 3);
  return (i32x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The stage's MFMAs as volatile asm statements: instruction selection lets the pure MFMA builtin nodes float down to their
// next use, i.e. below any block of side-effecting instructions (loads, LDS-DMAs) that follows them in the source --
// with the request block of a stage anywhere in front of MFMAs, those MFMAs were emitted BEHIND it and the block sat
// in the stage's head with the matrix pipes idle.  Volatile statements keep their order among themselves and against
// the loads.  Hazards the compiler no longer sees: (i) the same accumulator is next touched three MFMAs (>= 96 cycles)
// later, by another MFMA; (ii) A / B operands come from LDS reads (s_waitcnt by the compiler, which tracks the
// registers); (iii) the epilogue's reads of the accumulators sit behind explicit s_nops.
__device__ __forceinline__ void mfma16(f32x16& c, const f16x8& a, const f16x8& b) {
  if (EOVAE_ROWS_ABL & 8) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return; }
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 16 x 16 x 32 form (Q16): a quarter of the output tile, the whole 32-pixel row as K
__device__ __forceinline__ void mfma16q(f32x4& c, const f16x8& a, const f16x8& b) {
  if (EOVAE_ROWS_ABL & 8) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return; }
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma8(f32x16& c, const i32x8& a, const i32x8& b, int sa, int sb) {
  if (EOVAE_ROWS_ABL & 8) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return; }
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1 blgp:1"
               : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

// wave-uniform values the compiler keeps in vector registers (64-bit address arithmetic, expanded divisions): pinned to
// scalar registers, so that a buffer descriptor built from them is not loaded inside a waterfall loop
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// SP = 1: the "nearest x2 + 3x3" layer in sub-pixel form.  X is the LOW-res input (records), dY the 2x output gradient.
// Output parity (py, px) of dY against X is a 2 x 2-tap weight gradient on the low-res grid:
//     G[p][a][b] = sum dY[2 y + py][2 x + px] * X[y + a - 1 + py][x + b - 1 + px]          (a, b in {0, 1})
// i.e. the plain walk over the parity image dY' with 4 of the 9 taps.  A block takes ONE ROW parity py (blockIdx: slab * 2
// + py) and both column parities: it stages the hi-res row 2 (R + 1 - py) + py whole (64 pixels per strip, every loaded
// byte used) and de-interleaves it into two pixel planes while converting.  dY' row R + 1 - py is what the plain kernel
// calls "row R + 1", so the two live ring positions are the same for both py.  Waves = 4 co-tiles x 2 column parities;
// a wave runs BOTH ci-tiles (8 accumulators: ci-tile x a x b), its X window starts px pixels to the right: 24 MFMAs per
// stage and wave (plain 27; one parity per block was 12 behind the same staging work: issue-bound, 330-415 TFLOP/s of
// actual MFMA work).  Two dY slots of two planes (a row's fragments are read once, at the start of the stage after its
// conversion) keep the LDS at 157 KiB.  The fold (wgrad_up_fold_kernel) sums the splits and maps (parity, a, b) back to
// the nine taps of the layer's kernel.
// CORR (round 5): false = the weight gradient's product is f16(dY) * f16(X) ALONE -- no fp8 correction product: the nine (SP:
// eight) scaled MFMAs of a stage, the dY residual / e5m2 conversions and stores, the second LDS-DMA of every X row, the
// fp8 fragment reads and the A8 register ring all go; a stage is 18 (16) f16 MFMAs = half the matrix-pipe cycles.  Why that
// is sound here and nowhere else: a weight gradient is a LEAF -- its rounding errors (two f16 roundings per term, random
// over >= 10^4 pixels) do not feed another product, whereas the forward and data-gradient errors compound through ~60
// layers; tools/scheme_probe_grad.py / profiles/r05_scheme_probe_wgrad.log (emulation) and the full-size pins of
// tests/test_gpu_model.py hold the same gradient bars as before.  true = the r4 arithmetic (library option
// c8_wgrad_corr = 1).
// Q16 (round 5, plain f16-only form): the same products on v_mfma_f32_16x16x32_f16 -- a wave's 32 co x 32 ci tile of a tap is
// four 16 x 16 accumulators, one MFMA takes the whole 32-pixel row as its K (36 MFMAs of 16 cycles per stage instead of 18 of
// 32).  Same LDS images, same number of fragment reads, same registers; what changes is the clock the chip holds under the
// loop (MI355X_MICROARCH.md 'DVFS give-back' (7); tools/probe_mfma_shape.hip: 1.15 x the FLOP/s at equal cycles).
// Fragment maps: A (dY) lane (li, g) = row 16 half + li, pixels 8 g .. 8 g + 7: one ds_read_b128; B (X) lane (li, g) = channel
// 16 nh + li, pixels 8 g + kx ..: the two transposing reads of the 32 x 32 form with (channel half, k half) = (nh, g) instead
// of (g & 1, g >> 1); D lane (li, g) = ci 16 nh + li, co 16 half + 4 g + q.
template <int SP, bool CORR, bool Q16 = false>
__global__ __launch_bounds__(512, 2) void wgrad3x3_f16c8_rows_kernel(WgCArgs a) {
  static_assert(!Q16 || !CORR, "Q16: the f16-only forms");
  __shared__ __attribute__((aligned(16))) unsigned char dys[SP ? 4 * DYSLOT : NRING * DYSLOT];   // SP: 2 slots x 2 pixel planes
  // FOUR arrays for the LDS-DMA targets (X row s + 3 is requested in stage s; row s + 1 is already read at the end of
  // stage s): with one array and a computed slot offset the compiler cannot tell a DMA write from the fragment reads of
  // another slot and drains vmcnt(0) in front of every read (conv_f16c8_wgrad_w4.hip)
  __shared__ __attribute__((aligned(16))) unsigned char xs0[XSLOT];
  __shared__ __attribute__((aligned(16))) unsigned char xs1[XSLOT];
  __shared__ __attribute__((aligned(16))) unsigned char xs2[XSLOT];
  __shared__ __attribute__((aligned(16))) unsigned char xs3[XSLOT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, li = lane & 15;
  // waves w and w + 4 share a SIMD: same co-tile, the two ci-tiles (SP: the two column parities, each wave both ci-tiles)
  const int ct = wave & 3, cih = SP ? 0 : wave >> 2, pxw = SP ? wave >> 2 : 0;
  int slab, ysub;
  {   // XCD-aware block -> (slab, split slot) map, as wgrad3x3_f16c8_kernel
    const int nslab = a.ncib * a.ncob_ * (SP ? 2 : 1);
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
  const int py = SP ? (slab & 1) : 0;                               // SP: slab = (cob * ncib + cib) * 2 + row parity
  const int slab_ = SP ? (slab >> 1) : slab;
  const int cib = uni(slab_ % a.ncib), cob = uni(slab_ / a.ncib);
  const int H = SP ? a.Hin : a.Hout, W = SP ? a.Win : a.Wout;   // the grid of X (and of dY')
  const unsigned plane = (unsigned)(H * W);
  const unsigned dplane = SP ? 4u * plane : plane;               // a dY channel plane

  float dys_scale = 1.f, dyinv = 1.f;   // dy pre-scale: the power of two that puts its amax at 2^13..2^14
  if (a.dyamax) {
    const unsigned bits = *a.dyamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      dys_scale = pow2f(13 - e);
      dyinv = pow2f(e - 13);
    }
  }
  if (SP && a.xamax) {   // the records were written from x * 2^(13 - e) (f16c8_records_kernel): undone with the result
    const unsigned bits = *a.xamax;
    if (bits) dyinv *= pow2f(expo_of_bits(bits) - 13);
  }
  const int sc_a = 127 - 12, sc_b = 127;   // E8M0: both K blocks carry one operand scaled by 2^12

  // ---- the walk: units u = ysub, ysub + split, ... ; unit = (image n, strip tx, row segment seg) covers output rows
  // [r0, r1); its stages are the X rows R = r0 - 1 .. r1 (dY rows outside [r0, r1) and X rows outside the image: zeros).
  // The cursor `c` is the stage whose X row is requested next.  A row step is two scalar adds; the descriptors' bases and
  // the per-lane DMA offset change only when a unit opens.
  const int segrows = uni(H / a.nseg);
  const unsigned OOR = 0x80000000u;
  // bytes per pixel and channel octet of X: 32 = operand records (format 1: f16 | e5m2 halves), 16 = the F plane of MX operand
  // records (format 2: f16 only -- !CORR; groupnorm.hip gn_apply_mx_kernel)
  const unsigned XU = (!CORR && a.xpacked == 2) ? 16u : 32u;
  const unsigned dyspan = (unsigned)a.Cout * dplane * 4u, xspan = 8u * plane * XU;
  struct Cursor {
    int R, r0, r1, live;
    unsigned xsoff, dsoff;     // byte offsets of X row R / of dY row R + 1 at the strip's first column (scalar offsets)
    const char* xbase;         // records of image n, channel octets 8 cib .. 8 cib + 7
    const float* dbase;        // dY of image n
  } c;
  c.R = 0; c.r0 = 0; c.r1 = -1; c.live = 1; c.xsoff = 0; c.dsoff = 0;
  c.xbase = reinterpret_cast<const char*>(a.x); c.dbase = a.dy;
  int unit = uni(ysub - a.split);
  unsigned vx = OOR;           // lane's DMA offset inside the descriptor: octet `wave`, pixel column x0 - 1 + lane (or OOR)
  auto step = [&]() __attribute__((always_inline)) {
    if (c.live && c.R >= c.r1) {   // wave-uniform, taken once per unit
      asm volatile("" ::: "memory");   // a real branch: if-converted, the cursor's scalars turn into vector selects
      unit = uni(unit + a.split);
      if (unit >= a.units) {
        c.live = 0;
      } else {
        const int seg = unit % a.nseg, t = unit / a.nseg;
        const int x0 = uni((t % a.tiles_x) * 32), n = uni(t / a.tiles_x);
        c.r0 = uni(seg * segrows);
        c.r1 = c.r0 + segrows;
        c.R = c.r0 - 1;
        c.xsoff = (unsigned)uni((c.R * W) * (int)XU);
        c.dsoff = SP ? (unsigned)uni(((2 * (c.r0 - py) + py) * 2 * W + 2 * x0) * 4) : (unsigned)uni((c.r0 * W + x0) * 4);
        c.xbase = uni_ptr(reinterpret_cast<const char*>(a.x) + ((size_t)n * (a.Cin / 8) + (size_t)cib * 8) * plane * XU);
        c.dbase = uni_ptr(a.dy + (size_t)n * a.Cout * dplane);
        const unsigned ln = threadIdx.x & 63u;
        const int px = x0 - 1 + (int)ln;
        vx = (ln < 34u && (unsigned)px < (unsigned)W) ? ((unsigned)wave * plane + (unsigned)px) * XU : OOR;
      }
    } else {
      ++c.R;
      c.xsoff += (unsigned)W * XU;
      c.dsoff += (unsigned)((SP ? 4 : 1) * W * 4);
    }
  };
  int nst;   // stages of this block
  {
    const int mine = ysub < a.units ? (a.units - ysub + a.split - 1) / a.split : 0;
    nst = uni(mine * (segrows + 2));
  }

  // ---- dY staging: thread -> (row co, pixels 8 q .. 8 q + 7): two 16-byte loads, ~30 conversions, three stores.  The rows
  // of 16 consecutive lanes are co, co + 4, co + 2, co + 6 (two per 8-lane group): with the 80- / 48-byte pitches their
  // 64- / 32-byte pieces fall into disjoint banks, so the stores are conflict-free too.
  // The loads are UNCONDITIONAL: a row outside the unit is read through a descriptor of zero records and comes back as
  // zeros (a branch around the loads would make the registers phi nodes behind an s_waitcnt vmcnt(0)).  Two register
  // sets: the row of stage s + 2 is requested in stage s, the row of stage s + 1 is converted between the MFMAs of stage s.
  const int dq = tid & 3;
  const int dco = (EOVAE_ROWS_DBG & 1) ? (tid >> 2) : 8 * (tid >> 5) + ((tid >> 4) & 1) + (((tid >> 2) & 1) << 2) + (((tid >> 3) & 1) << 1);
  const unsigned dvoff = ((unsigned)(cob * 128 + dco) * dplane + (SP ? 16u : 8u) * (unsigned)dq) * 4u;
  f32x4 dr[2][SP ? 4 : 2];
  float dbs = 0.f;
  const bool do_db = a.db != nullptr && cib == 0;
  struct { const float* base; unsigned soff; int ok; } sv;   // the dY row of the cursor's stage, requested a stage later
  sv.base = a.dy; sv.soff = 0; sv.ok = 0;
  auto save_dy = [&]() __attribute__((always_inline)) {
    sv.base = c.dbase;
    sv.soff = c.dsoff;
    sv.ok = c.live & (c.R + 1 - py >= c.r0) & (c.R + 1 - py < c.r1);
  };
  auto request_dy = [&](auto pc) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    const unsigned so = (EOVAE_ROWS_DBG & 8) ? (unsigned)uni((int)sv.soff) : sv.soff;
    const unsigned nr = (EOVAE_ROWS_DBG & 8) ? (unsigned)uni(sv.ok ? (int)dyspan : 0) : (sv.ok ? dyspan : 0u);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (EOVAE_ROWS_DBG & 8) ? uni_ptr(const_cast<float*>(sv.base)) : const_cast<float*>(sv.base), 0, nr, 0x00020000);
    dr[P][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff, so, 0));
    dr[P][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff + 16, so, 0));
    if constexpr (SP != 0) {
      dr[P][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff + 32, so, 0));
      dr[P][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff + 48, so, 0));
    }
  };
  const int dw16 = dco * P16 + dq * 16, dw8 = DY16 + dco * P8 + dq * 8;
  // conversion of one register set in two pieces placed between MFMA triples: pixels 0-3 | pixels 4-7, each stored at once
  // (8 + 4 + 4 bytes): nothing of a piece lives across an MFMA
  auto conv_piece = [&](auto pc, auto piecec, unsigned char* slot) __attribute__((always_inline)) {
    // plain: piece I = pixels 4 I .. 4 I + 3 of the thread's 8; SP: piece 2 pp + I = pixels 4 I .. of column-parity plane pp
    constexpr int P = decltype(pc)::value, I = decltype(piecec)::value & 1, PP = decltype(piecec)::value >> 1;
    f32x4 d;
    if constexpr (SP != 0) {   // every other hi-res pixel
      d[0] = dr[P][2 * I][PP]; d[1] = dr[P][2 * I][PP + 2]; d[2] = dr[P][2 * I + 1][PP]; d[3] = dr[P][2 * I + 1][PP + 2];
      slot += PP * DYSLOT;
    } else {
      d = dr[P][I];
    }
    float v[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = d[j] * dys_scale;
    if (do_db) {   // uniform
      float t0 = d[0] + d[1], t1 = d[2] + d[3];
      asm volatile("" : "+v"(t0), "+v"(t1));   // scalar adds: packed f32 VALU is slow beside MFMAs
      dbs += t0 + t1;
    }
    i32x2 hi;
    if constexpr (!CORR) {   // leading parts only
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        int hp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[j]), "v"(v[j + 1]));
        hi[j >> 1] = hp;
      }
      *reinterpret_cast<i32x2*>(slot + dw16 + 8 * I) = hi;
      (void)lo;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      int hp;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[j]), "v"(v[j + 1]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp), "v"(v[j]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp), "v"(v[j + 1]));
      hi[j >> 1] = hp;
    }
    // (the `old` operands are don't-cares: both halves of each result are written)
    s16x2 lv = __builtin_bit_cast(s16x2, (EOVAE_ROWS_DBG & 2) ? 0.f : lo[0]);
    lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[0], lo[1], 0x1p-12f, false);
    lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[2], lo[3], 0x1p-12f, true);
    int hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], (EOVAE_ROWS_DBG & 2) ? 0 : __builtin_bit_cast(int, v[0]), false);
    hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], hv, true);
    *reinterpret_cast<i32x2*>(slot + dw16 + 8 * I) = hi;
    *reinterpret_cast<int*>(slot + dw8 + 4 * I) = __builtin_bit_cast(int, lv);   // image 0: dyl'
    *reinterpret_cast<int*>(slot + dw8 + DY8 + 4 * I) = hv;                     // image 1: dyh'
  };

  // ---- X by LDS-DMA: wave w moves channel octet w of the slab (ci-tile w >> 2, octet w & 3 of it)
  const int x16dst = uni(wave * XP16), x8dst = uni(X16 + (wave >> 2) * X8T + x8off(wave & 3));
  auto request_x = [&](unsigned char* xslot) __attribute__((always_inline)) {
    const int ok = c.live & ((unsigned)c.R < (unsigned)H);
    const int so = (int)c.xsoff;
    const __amdgpu_buffer_rsrc_t srd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.xbase), 0, ok ? xspan : 0u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(xslot + x16dst), 16, (int)vx, so, 0, 0);
    // (the records' second halves through the SCALAR offset: an LDS-DMA's immediate offset also moves its LDS address)
    if constexpr (CORR)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(xslot + x8dst), 16, (int)vx, so + 16, 0, 0);
  };
  // vector-memory operations of one stage's requests: dY row (2, SP: 4 loads) + X row (CORR: 2 DMAs, else 1)
  constexpr int NDY = SP ? 4 : 2, NX = CORR ? 2 : 1;

  // ---- fragment addresses (per-lane constants; slot, tap shift and k-step are immediates)
  // A (dY): lane (r, h) reads row 32 ct + r; f16: 8 pixels at 16 h of the k-step; fp8: the row's 32 bytes of image h
  const int a16off = pxw * DYSLOT + (32 * ct + r) * P16 + 16 * h;
  const int a8off = pxw * DYSLOT + DY16 + h * DY8 + (32 * ct + r) * P8;
  // B (X) f16 by ds_read_b64_tr_b16: lane 4q + p of group g supplies pixel q of its 4-pixel block, 8-byte chunk p & 1 of
  // octet 2 (g & 1) + (p >> 1) of the wave's ci-tile; k-half g >> 1
  const int b16off = (4 * cih + 2 * (g & 1) + ((li & 3) >> 1)) * XP16 + (8 * (g >> 1) + (li >> 2)) * 16 + (li & 1) * 8 + pxw * 16;
  // B fp8 by ds_read_b64_tr_b8: lane 2j + p supplies pixel j of its 8-pixel block, octet 2 (g & 1) + p; K block g >> 1:
  // block 0 = xh' (second 8 bytes of the pixel's 16), block 1 = xl'
  const int b8off = X16 + cih * X8T + x8off(2 * (g & 1) + (li & 1)) + (li >> 1) * 16 + ((g >> 1) ? 0 : 8) + pxw * 16;

  // Q16: A row li of co-half `half` (+ half * 16 P16), pixels 8 g ..; B as b16off with the k-group g and the channel half as an
  // immediate (+ nh * 2 XP16)
  const int a16q = pxw * DYSLOT + (32 * ct + li) * P16 + 16 * g;
  const int b16q = (4 * cih + ((li & 3) >> 1)) * XP16 + (8 * g + (li >> 2)) * 16 + (li & 1) * 8 + pxw * 16;

  constexpr int NT = SP ? 8 : 9;   // SP: acc[4 ci-tile + 2 a + b]
  f32x16 acc[Q16 ? 1 : NT];
  f32x4 acc4[Q16 ? NT : 1][2][2];   // Q16: [tap | SP tile][co half][ci half]
  if constexpr (Q16) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc4[t][(q >> 3) & 1][(q >> 2) & 1][q & 3] = 0.f;
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  }
  f16x8 A16[NRING][2];
  i32x8 A8[NRING];
#pragma unroll
  for (int k = 0; k < NRING; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { A16[k][0][j] = (_Float16)0.f; A16[k][1][j] = (_Float16)0.f; A8[k][j] = 0; }
  }

  auto xslot_of = [&](auto kc) __attribute__((always_inline)) -> unsigned char* {
    constexpr int k = decltype(kc)::value % 4;
    if constexpr (k == 0) return xs0;
    else if constexpr (k == 1) return xs1;
    else if constexpr (k == 2) return xs2;
    else return xs3;
  };

  // ---- prologue: X(0..2) in flight, dY(0) committed, dY(1) in registers, dY(2) noted; the cursor ends at stage 3
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
  step();
  request_x(xs0);
  save_dy();
  request_dy(C0{});
  step();
  request_x(xs1);
  save_dy();
  request_dy(C1{});
  conv_piece(C0{}, C0{}, dys);
  conv_piece(C0{}, C1{}, dys);
  if constexpr (SP != 0) {
    conv_piece(C0{}, C2{}, dys);
    conv_piece(C0{}, C3{}, dys);
  }
  step();
  request_x(xs2);
  save_dy();
  step();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the fp8 B fragment of a stage's first MFMAs is read at the end of the stage before it (X rows land two stages
  // ahead): behind the barrier a stage starts multiplying at once, on operands that are all in registers
  i32x8 b8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f16x8 nb0 = A16[0][0], nb1 = A16[0][0];   // !CORR: the B fragments of the next sub-step, read one sub-step ahead
  if constexpr (CORR) {
    b8 = rtr8(xs0 + b8off);
  } else if constexpr (Q16) {
    nb0 = rtr16(xs0 + b16q, xs0 + b16q + 4 * 16);
    nb1 = rtr16(xs0 + b16q + 2 * XP16, xs0 + b16q + 2 * XP16 + 4 * 16);
  } else {
    nb0 = rtr16(xs0 + b16off, xs0 + b16off + 4 * 16);
    nb1 = rtr16(xs0 + b16off + 16 * 16, xs0 + b16off + 20 * 16);
  }

  // stage s: S = s mod 12; K = s mod 3 (dY slots, register ring), P = s mod 2 (dY register set that receives row s + 2; row
  // s + 1 sits in the other), Q = s mod 4 (X slot)
  auto body = [&](auto sc, auto rolec) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value, K = S % 3, P = S % 2, Q = S % 4;
    // (EOVAE_ROWS_ROLES: the two waves of a SIMD place their non-MFMA work one kx apart)
    constexpr int KREQ = decltype(rolec)::value, KCNV = KREQ + 1;
    constexpr int KD = (K + 1) % 3;
    using PC = std::integral_constant<int, P>;
    using PN = std::integral_constant<int, P ^ 1>;
    const unsigned char* dsl = dys + K * DYSLOT;
    unsigned char* dnx = dys + KD * DYSLOT;
    const unsigned char* xsl = xslot_of(std::integral_constant<int, Q>{});
    // A stage opens with every wave of the CU behind the same barrier and the matrix pipes idle.  Its first two MFMAs
    // multiply the two OLD dY rows (register ring) with the B fragment read before the barrier; the new row's A
    // fragments are requested first thing and are needed by the third MFMA.  The requests of dY row s + 2 and X row
    // s + 3 and the cursor step ride between the MFMAs of kx = 0, the conversion of dY row s + 1 (requested a stage ago)
    // between the triples of kx = 1.
    constexpr int R1 = (K + 2) % 3, R2 = (K + 1) % 3;   // ring positions of ky = 1, 2 (ky = 0: K)
    if constexpr (Q16) {
      // 36 MFMAs of 16 cycles, in the order of the 32 x 32 form: per kx the two OLD rows first (eight MFMAs on operands that
      // are in registers when the barrier opens), then the new row's four; the requests ride between the first MFMAs of
      // kx = KREQ, the conversion of dY row s + 1 around the B reads of kx = KCNV.
      A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16q));
      A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16q + 16 * P16));
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f16x8 b0 = nb0, b1 = nb1;
        if (kx == KCNV) {
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + NDY + NX) : "memory");   // newer than dY(s + 1): X(s + 2), dY(s + 2), X(s + 3)
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma16q(acc4[3 + kx][0][0], A16[R1][0], b0);
        if (kx == KREQ) {
          __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
          if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
        }
        mfma16q(acc4[6 + kx][0][0], A16[R2][0], b0);
        mfma16q(acc4[3 + kx][0][1], A16[R1][0], b1);
        if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) {
          request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
          save_dy();
        }
        mfma16q(acc4[6 + kx][0][1], A16[R2][0], b1);
        mfma16q(acc4[3 + kx][1][0], A16[R1][1], b0);
        if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) step();
        mfma16q(acc4[6 + kx][1][0], A16[R2][1], b0);
        mfma16q(acc4[3 + kx][1][1], A16[R1][1], b1);
        mfma16q(acc4[6 + kx][1][1], A16[R2][1], b1);
        __builtin_amdgcn_sched_barrier(0);
        if (kx < 2) {
          nb0 = rtr16(xsl + b16q + (kx + 1) * 16, xsl + b16q + (kx + 5) * 16);
          nb1 = rtr16(xsl + b16q + 2 * XP16 + (kx + 1) * 16, xsl + b16q + 2 * XP16 + (kx + 5) * 16);
        } else {   // the next stage's first B fragments
          const unsigned char* xn = xslot_of(std::integral_constant<int, (Q + 1) % 4>{});
          nb0 = rtr16(xn + b16q, xn + b16q + 4 * 16);
          nb1 = rtr16(xn + b16q + 2 * XP16, xn + b16q + 2 * XP16 + 4 * 16);
        }
        if (kx == KCNV && !(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
        __builtin_amdgcn_sched_barrier(0);
        mfma16q(acc4[kx][0][0], A16[K][0], b0);
        mfma16q(acc4[kx][0][1], A16[K][0], b1);
        mfma16q(acc4[kx][1][0], A16[K][1], b0);
        mfma16q(acc4[kx][1][1], A16[K][1], b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDY + NX) : "memory");   // X(s + 2) has landed: only this stage's requests may be outstanding
      if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      return;
    }
    if constexpr (CORR) {
      const i32x4 l = *reinterpret_cast<const i32x4*>(dsl + a8off), u = *reinterpret_cast<const i32x4*>(dsl + a8off + 16);
      A8[K] = (i32x8)__builtin_shufflevector(l, u, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off));
    A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off + 32));
    if constexpr (!CORR) {
      // 18 f16 MFMAs.  Per kx: the two OLD rows (register ring) first -- four MFMAs on operands that are in registers when
      // the barrier opens (their B fragments were read one sub-step ahead) -- then the new row, whose A fragments were
      // requested above.  Requests ride between the MFMAs of kx = KREQ, the conversion of dY row s + 1 at kx = KCNV.
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f16x8 b0 = nb0, b1 = nb1;
        if (kx == KCNV) {
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + NDY + NX) : "memory");   // newer than dY(s + 1): X(s + 2), dY(s + 2), X(s + 3)
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc[3 + kx], A16[R1][0], b0);
        if (kx == KREQ) {
          __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
          if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
        }
        mfma16(acc[6 + kx], A16[R2][0], b0);
        if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) {
          request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
          save_dy();
        }
        mfma16(acc[3 + kx], A16[R1][1], b1);
        if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) step();
        mfma16(acc[6 + kx], A16[R2][1], b1);
        __builtin_amdgcn_sched_barrier(0);
        if (kx < 2) {
          nb0 = rtr16(xsl + b16off + (kx + 1) * 16, xsl + b16off + (kx + 5) * 16);
          nb1 = rtr16(xsl + b16off + (kx + 17) * 16, xsl + b16off + (kx + 21) * 16);
        } else {   // the next stage's first B fragments
          const unsigned char* xn = xslot_of(std::integral_constant<int, (Q + 1) % 4>{});
          nb0 = rtr16(xn + b16off, xn + b16off + 4 * 16);
          nb1 = rtr16(xn + b16off + 16 * 16, xn + b16off + 20 * 16);
        }
        if (kx == KCNV && !(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc[kx], A16[K][0], b0);
        mfma16(acc[kx], A16[K][1], b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDY + NX) : "memory");   // X(s + 2) has landed: only this stage's requests may be outstanding
      if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      return;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f16x8 b0 = rtr16(xsl + b16off + kx * 16, xsl + b16off + (kx + 4) * 16);
      if (kx == KCNV) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // newer than dY(s + 1): X(s + 2), dY(s + 2), X(s + 3) -- two operations each
        if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma8(acc[3 + kx], A8[R1], b8, sc_a, sc_b);
      if (kx == KREQ) {
        __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
        if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
      }
      mfma8(acc[6 + kx], A8[R2], b8, sc_a, sc_b);
      if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) {
        request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
        save_dy();
      }
      mfma8(acc[kx], A8[K], b8, sc_a, sc_b);
      if (kx == KREQ && !(EOVAE_ROWS_ABL & 2)) step();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 b1 = rtr16(xsl + b16off + (kx + 16) * 16, xsl + b16off + (kx + 20) * 16);
      if (kx == KCNV && !(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(acc[3 + kx], A16[R1][0], b0);
      mfma16(acc[6 + kx], A16[R2][0], b0);
      mfma16(acc[kx], A16[K][0], b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kx < 2) b8 = rtr8(xsl + b8off + (kx + 1) * 16);
      else b8 = rtr8(xslot_of(std::integral_constant<int, (Q + 1) % 4>{}) + b8off);   // the next stage's first B fragment
      __builtin_amdgcn_sched_barrier(0);
      mfma16(acc[3 + kx], A16[R1][1], b1);
      mfma16(acc[6 + kx], A16[R2][1], b1);
      mfma16(acc[kx], A16[K][1], b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // X(s + 2) has landed when at most this stage's four operations are outstanding
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    // a RAW barrier behind the counted wait above: __syncthreads() carries a release fence = s_waitcnt vmcnt(0), which made
    // every stage wait for the dY row and X row it had just requested for the stages after it
    if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // SP stage: ring of TWO rows (K = new row s + 1 - py, R1 = the row before it), two dY slots, four sub-steps (kx, ci-tile) of
  // 2 + 4 MFMAs; the requests ride in sub-step 0, the four conversion pieces of dY row s + 1 in sub-steps 1 and 2.
  auto body_sp = [&](auto sc) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value, K = S % 2, R1 = K ^ 1, P = S % 2, Q = S % 4;
    using PC = std::integral_constant<int, P>;
    using PN = std::integral_constant<int, P ^ 1>;
    const unsigned char* dsl = dys + K * (2 * DYSLOT);
    unsigned char* dnx = dys + R1 * (2 * DYSLOT);
    const unsigned char* xsl = xslot_of(std::integral_constant<int, Q>{});
    if constexpr (Q16) {
      // 32 MFMAs of 16 cycles in the order of the 32 x 32 form: per sub-step (kx, ci-tile) the old row's four (co half x ci half)
      // first, then the new row's; A fragments = co halves, B fragments = ci halves of the ci-tile
      A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16q));
      A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16q + 16 * P16));
      static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
        constexpr int IT = decltype(itc)::value, kx = IT >> 1, ci = IT & 1;
        constexpr int TK = 4 * ci + kx, TR = 4 * ci + 2 + kx;   // a = 0: the new row; a = 1: the row before it
        const f16x8 b0 = nb0, b1 = nb1;
        if constexpr (IT == 1) {
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + NDY + NX) : "memory");   // newer than dY(s + 1): X(s + 2), dY(s + 2), X(s + 3)
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
        }
        if constexpr (IT == 2) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C2{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool useR = !(SP == 2 && py == 1), useB = !(SP == 2 && kx == 1 && pxw == 1);
        if (useR && useB) mfma16q(acc4[TR][0][0], A16[R1][0], b0);
        if constexpr (IT == 0) {
          __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
          if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
        }
        if (useR && useB) mfma16q(acc4[TR][0][1], A16[R1][0], b1);
        if constexpr (IT == 0) {
          if (!(EOVAE_ROWS_ABL & 2)) {
            request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
            save_dy();
            step();
          }
        }
        if (useR && useB) mfma16q(acc4[TR][1][0], A16[R1][1], b0);
        if (useR && useB) mfma16q(acc4[TR][1][1], A16[R1][1], b1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (IT < 3) {
          constexpr int nkx = (IT + 1) >> 1, nci = (IT + 1) & 1, NO = nci * 4 * XP16;
          nb0 = rtr16(xsl + b16q + NO + nkx * 16, xsl + b16q + NO + (nkx + 4) * 16);
          nb1 = rtr16(xsl + b16q + NO + 2 * XP16 + nkx * 16, xsl + b16q + NO + 2 * XP16 + (nkx + 4) * 16);
        } else {   // the next stage's first B fragments
          const unsigned char* xn = xslot_of(std::integral_constant<int, (Q + 1) % 4>{});
          nb0 = rtr16(xn + b16q, xn + b16q + 4 * 16);
          nb1 = rtr16(xn + b16q + 2 * XP16, xn + b16q + 2 * XP16 + 4 * 16);
        }
        if constexpr (IT == 1) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
        }
        if constexpr (IT == 2) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C3{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (useB) mfma16q(acc4[TK][0][0], A16[K][0], b0);
        if (useB) mfma16q(acc4[TK][0][1], A16[K][0], b1);
        if (useB) mfma16q(acc4[TK][1][0], A16[K][1], b0);
        if (useB) mfma16q(acc4[TK][1][1], A16[K][1], b1);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDY + NX) : "memory");   // X(s + 2) has landed
      if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      return;
    }
    if constexpr (CORR) {
      const i32x4 l = *reinterpret_cast<const i32x4*>(dsl + a8off), u = *reinterpret_cast<const i32x4*>(dsl + a8off + 16);
      A8[K] = (i32x8)__builtin_shufflevector(l, u, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off));
    A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off + 32));
    if constexpr (!CORR) {
      // 16 f16 MFMAs: per sub-step (kx, ci-tile) the old row's two k-steps first (operands in registers), then the new row's
      static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
        constexpr int IT = decltype(itc)::value, kx = IT >> 1, ci = IT & 1;
        f32x16& accK = acc[4 * ci + kx];        // a = 0: the new row
        f32x16& accR = acc[4 * ci + 2 + kx];    // a = 1: the row before it
        const f16x8 b0 = nb0, b1 = nb1;
        if constexpr (IT == 1) {
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + NDY + NX) : "memory");   // newer than dY(s + 1): X(s + 2), dY(s + 2), X(s + 3)
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
        }
        if constexpr (IT == 2) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C2{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool useR = !(SP == 2 && py == 1), useB = !(SP == 2 && kx == 1 && pxw == 1);
        if (useR && useB) mfma16(accR, A16[R1][0], b0);
        if constexpr (IT == 0) {
          __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
          if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
        }
        if (useR && useB) mfma16(accR, A16[R1][1], b1);
        if constexpr (IT == 0) {
          if (!(EOVAE_ROWS_ABL & 2)) {
            request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
            save_dy();
            step();
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (IT < 3) {
          constexpr int nkx = (IT + 1) >> 1, nci = (IT + 1) & 1, NO = nci * 4 * XP16;
          nb0 = rtr16(xsl + b16off + NO + nkx * 16, xsl + b16off + NO + (nkx + 4) * 16);
          nb1 = rtr16(xsl + b16off + NO + (nkx + 16) * 16, xsl + b16off + NO + (nkx + 20) * 16);
        } else {   // the next stage's first B fragments
          const unsigned char* xn = xslot_of(std::integral_constant<int, (Q + 1) % 4>{});
          nb0 = rtr16(xn + b16off, xn + b16off + 4 * 16);
          nb1 = rtr16(xn + b16off + 16 * 16, xn + b16off + 20 * 16);
        }
        if constexpr (IT == 1) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
        }
        if constexpr (IT == 2) {
          if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C3{}, dnx);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (useB) mfma16(accK, A16[K][0], b0);
        if (useB) mfma16(accK, A16[K][1], b1);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDY + NX) : "memory");   // X(s + 2) has landed
      if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      return;
    }
    static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
      constexpr int IT = decltype(itc)::value, kx = IT >> 1, ci = IT & 1;
      constexpr int XO16 = ci * 4 * XP16, XO8 = ci * X8T;
      f32x16& accK = acc[4 * ci + kx];        // a = 0: the new row
      f32x16& accR = acc[4 * ci + 2 + kx];    // a = 1: the row before it
      const f16x8 b0 = rtr16(xsl + b16off + XO16 + kx * 16, xsl + b16off + XO16 + (kx + 4) * 16);
      if constexpr (IT == 1) {
        // newer than dY(s + 1): X(s + 2) (2 operations), dY(s + 2) (4), X(s + 3) (2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C0{}, dnx);
      }
      if constexpr (IT == 2) {
        if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C2{}, dnx);
      }
      __builtin_amdgcn_sched_barrier(0);
      // SP == 2 (stride-2 layer, operands swapped): the products (py, a) = (1, 1) and (px, b) = (1, 1) are no taps of that
      // layer -- block-uniform / wave-uniform skips of their MFMAs (the fold never reads those tiles)
      const bool useR = !(SP == 2 && py == 1), useB = !(SP == 2 && kx == 1 && pxw == 1);
      if (useR && useB) mfma8(accR, A8[R1], b8, sc_a, sc_b);
      if constexpr (IT == 0) {
        __builtin_amdgcn_sched_barrier(0);   // nothing of the request block in front of the stage's first MFMA
        if (!(EOVAE_ROWS_ABL & 2)) request_dy(PC{});
      }
      if (useB) mfma8(accK, A8[K], b8, sc_a, sc_b);
      if constexpr (IT == 0) {
        if (!(EOVAE_ROWS_ABL & 2)) {
          request_x(xslot_of(std::integral_constant<int, (Q + 3) % 4>{}));
          save_dy();
          step();
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 b1 = rtr16(xsl + b16off + XO16 + (kx + 16) * 16, xsl + b16off + XO16 + (kx + 20) * 16);
      if constexpr (IT == 1) {
        if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C1{}, dnx);
      }
      if constexpr (IT == 2) {
        if (!(EOVAE_ROWS_ABL & 1)) conv_piece(PN{}, C3{}, dnx);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (useR && useB) mfma16(accR, A16[R1][0], b0);
      if (useB) mfma16(accK, A16[K][0], b0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IT < 3) {
        constexpr int nkx = (IT + 1) >> 1, nci = (IT + 1) & 1;
        b8 = rtr8(xsl + b8off + nci * X8T + nkx * 16);
      } else {
        b8 = rtr8(xslot_of(std::integral_constant<int, (Q + 1) % 4>{}) + b8off);   // the next stage's first B fragment
      }
      __builtin_amdgcn_sched_barrier(0);
      if (useR && useB) mfma16(accR, A16[R1][1], b1);
      if (useB) mfma16(accK, A16[K][1], b1);
      __builtin_amdgcn_sched_barrier(0);
      (void)XO8;
    });
    // X(s + 2) has landed when at most this stage's operations (dY(s + 2): 4, X(s + 3): 2) are outstanding
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (!(EOVAE_ROWS_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ONE exit: with an exit behind every body (twelve exits) the compiler kept the cursor's scalars in vector registers and
  // loaded every descriptor inside a waterfall loop.  The stages that pad the count to a multiple of twelve lie behind
  // the block's last unit: zero operands.
  auto run = [&](auto rolec) __attribute__((always_inline)) {
#pragma unroll 1
    for (int s = 0; s < nst; s += 12) {
      if constexpr (SP != 0) {
        static_for<0, 12>([&](auto sc) __attribute__((always_inline)) { body_sp(sc); });
      } else {
        static_for<0, 12>([&](auto sc) __attribute__((always_inline)) { body(sc, rolec); });
      }
    }
  };
  if (EOVAE_ROWS_ROLES && wave >= 4) run(C1{}); else run(C0{});
  // requests of the stages behind the last one; the asm MFMAs' results are read below: well past their latency
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  if (do_db) {   // lanes 4 k .. 4 k + 3 hold the same channel
    float v = dbs;
    v += dpp_quad<0xB1>(v);
    v += dpp_quad<0x4E>(v);
    if (dq == 0) atomicAdd(&a.db[cob * 128 + dco], v);
  }
  // this block's partial tile [tap][128 co][64 ci], folded in a fixed order by wgrad_fold_kernel
  // (SP: eight tiles per block [px][2 a + b][128 co][64 ci]; the wave holds [ci-tile][2 a + b] of its px)
  if constexpr (Q16) {   // lane (li, g): ci = 32 (ci-tile) + 16 nh + li, co = 32 ct + 16 half + 4 g + q
    float* pq = a.part + (((size_t)slab * a.split + ysub) * (SP ? 8 : 9) + (SP ? 4 * pxw : 0)) * (128 * 64) + (32 * ct + 4 * g) * 64 + 32 * cih + li;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (SP: tile 2 a + b = t & 3 of the wave's column parity, ci-tile t >> 2)
            const size_t to = SP ? (size_t)(t & 3) * (128 * 64) + (t >> 2) * 32 : (size_t)t * (128 * 64);
            pq[to + (16 * hf + q) * 64 + 16 * nh] = acc4[t][hf][nh][q] * dyinv;
          }
    return;
  }
  float* pt = a.part + (((size_t)slab * a.split + ysub) * (SP ? 8 : 9) + (SP ? 4 * pxw : 0)) * (128 * 64) + cih * 32 + r;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int col = ct * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if constexpr (SP != 0) pt[(size_t)(t & 3) * (128 * 64) + col * 64 + (t >> 2) * 32] = acc[t][q] * dyinv;
      else pt[(size_t)t * (128 * 64) + col * 64] = acc[t][q] * dyinv;
    }
}


// fp32 NCHW -> f16c8 operand records [N][C/8][HW][32 B] (the format gn_apply_packed_kernel writes, groupnorm.hip), for a
// layer input that is not a GroupNorm output.  xamax (nullable): x is scaled by the power of two that puts its amax at
// 2^13..2^14 first (the range guard of the fp32-input kernels); the consumer undoes it with its result.
__global__ __launch_bounds__(256) void f16c8_records_kernel(const float* __restrict__ x, const unsigned* __restrict__ xamax,
                                                            int* __restrict__ rec, int C, int HW, int pblocks) {
  const int pb = blockIdx.x % pblocks, og = blockIdx.x / pblocks;   // og = n * C/8 + octet
  const int p = pb * 256 + threadIdx.x;
  if (p >= HW) return;
  float xs = 1.f;
  if (xamax) {
    const unsigned bits = *xamax;
    if (bits) xs = pow2f(13 - expo_of_bits(bits));
  }
  float v[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x[((size_t)og * 8 + j) * HW + p] * xs;
  int pk[4];
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
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
  i32x4 a0, a1;
  a0[0] = pk[0]; a0[1] = pk[1]; a0[2] = pk[2]; a0[3] = pk[3];
  a1[0] = __builtin_bit_cast(int, l0); a1[1] = __builtin_bit_cast(int, l1); a1[2] = h0; a1[3] = h1;
  i32x4* r = reinterpret_cast<i32x4*>(rec + ((size_t)og * HW + p) * 8);
  r[0] = a0;
  r[1] = a1;
}

// dW[co][ci][KY][KX] of the upsample layer from the parity tiles: part[(slab * 2 + py) * split + y][4 px + 2 a + b][128][64];
// tap (KY, KX) of the layer's kernel lands on (a, b) = (KY + py - 1 >= 1 ? .. ) of parity (py, px):
//   py = 0: KY = 0 -> a = 0, KY = 1, 2 -> a = 1;   py = 1: KY = 0, 1 -> a = 0, KY = 2 -> a = 1   (same for KX, px, b).
// Splits and parities are summed in a fixed order (deterministic).
__global__ __launch_bounds__(576) void wgrad_up_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout,
                                                            int Cin, int ncib, int split) {
  // block = one slab x one co row, wave = tap, lane = ci: 256-byte reads of the parity tiles, one 2,304-byte run of dw (LDS)
  __shared__ float t[64 * 9];
  const int slab = blockIdx.x >> 7, col = blockIdx.x & 127;
  const int cob = slab / ncib, cib = slab % ncib;
  const int tap = threadIdx.x >> 6, cil = threadIdx.x & 63;
  const int KY = tap / 3, KX = tap % 3;
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int py = p >> 1, px = p & 1;
    const int a_ = py == 0 ? (KY == 0 ? 0 : 1) : (KY == 2 ? 1 : 0);
    const int b_ = px == 0 ? (KX == 0 ? 0 : 1) : (KX == 2 ? 1 : 0);
    const float* q = part + ((((size_t)slab * 2 + py) * split) * 8 + 4 * px + (2 * a_ + b_)) * (128 * 64) + col * 64 + cil;
    float s0 = 0.f, s1 = 0.f;
    int y = 0;
    for (; y + 1 < split; y += 2) {
      s0 += q[(size_t)y * 8 * 128 * 64];
      s1 += q[(size_t)(y + 1) * 8 * 128 * 64];
    }
    if (y < split) s0 += q[(size_t)y * 8 * 128 * 64];
    s += s0 + s1;
  }
  t[cil * 9 + tap] = s;
  __syncthreads();
  const int oc = cob * 128 + col, k = threadIdx.x, ic = cib * 64 + k / 9;
  if (oc < Cout && ic < Cin) dw[((size_t)oc * Cin + cib * 64) * 9 + k] = t[k];
}


// The stride-2 layer (pad (0,1,0,1) + 3x3 stride 2, layers.py Downsample) through the SAME kernel with the operands' roles
// swapped: dW[co][ci][KY][KX] = sum dY[co][Y][X] x[ci][2Y + KY][2X + KX] is the sub-pixel product G[p][a][b] with the HI-res
// tensor x in dY's place and the low-res dY (as records) in X's place, transposed:
//     G[p][a][b][ci][co] = sum x[ci][2y + py][2x + px] * dY[co][y + a - 1 + py][x + b - 1 + px],   KY = 2 - 2a - py, KX = 2 - 2b - px
// (py, a) = (0, 1), (1, 0), (0, 0) give KY = 0, 1, 2; (1, 1) pairs x rows with no tap and is dropped here (7 of the 16 products
// per pixel are not needed: the kernel runs them anyway, at 4.4 x the useful rate of the tile kernel's space-to-depth mode,
// which multiplied 36 tap-parity products for 9).  part[(slab * 2 + py) * split + y][4 px + 2 a + b][128 ci][64 co].
__global__ __launch_bounds__(576) void wgrad_down_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout,
                                                              int Cin, int ncob64, int split) {
  // block = one slab (128 ci x 64 co) x four ci rows, wave = tap, lane = co (the tiles' columns: 256-byte reads); dw wants
  // (ci, tap) runs per co: 4 ci x 9 taps = 144 bytes each, through LDS
  __shared__ float t[64][4 * 9 + 1];
  const int slab = blockIdx.x >> 5, g4 = blockIdx.x & 31;
  const int cib128 = slab / ncob64, cob64 = slab % ncob64;     // kernel's "cob" = 128 rows of ci, "cib" = 64 columns of co
  const int tap = threadIdx.x >> 6, col = threadIdx.x & 63;
  const int KY = tap / 3, KX = tap % 3;
  const int py = KY == 1 ? 1 : 0, a_ = KY == 0 ? 1 : 0;
  const int px = KX == 1 ? 1 : 0, b_ = KX == 0 ? 1 : 0;
#pragma unroll
  for (int cr = 0; cr < 4; ++cr) {
    const int row = g4 * 4 + cr;                               // ci row of the tile
    const float* q = part + ((((size_t)slab * 2 + py) * split) * 8 + 4 * px + (2 * a_ + b_)) * (128 * 64) + row * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    int y = 0;
    for (; y + 1 < split; y += 2) {
      s0 += q[(size_t)y * 8 * 128 * 64];
      s1 += q[(size_t)(y + 1) * 8 * 128 * 64];
    }
    if (y < split) s0 += q[(size_t)y * 8 * 128 * 64];
    t[col][cr * 9 + tap] = s0 + s1;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 36; e += 576) {
    const int c = e / 36, k = e % 36;
    const int oc = cob64 * 64 + c, ic = cib128 * 128 + g4 * 4 + k / 9;
    if (oc < Cout && ic < Cin) dw[((size_t)oc * Cin + cib128 * 128 + g4 * 4) * 9 + k] = t[c][k];
  }
}

}  // namespace

bool eovae_c8_wgrad_rows_eligible(const WgCArgs& a) {
  if (a.xpacked == 2 && eovae_opt(EOVAE_OPT_C8_WGRAD_CORR)) return false;   // MX operand records carry no e5m2 parts
  return eovae_opt(EOVAE_OPT_WGRAD_ROWS) != 0 && a.part && a.subpix == 0 && a.mode == MODE_NORMAL && a.xpacked && a.Cout % 128 == 0 &&
         a.Cin % 64 == 0 && a.Wout % 32 == 0 && a.Hout >= 1 && !a.xamax &&
         (long long)a.Hout * a.Wout * 32 * 8 < (1LL << 31) && (long long)a.Cout * a.Hout * a.Wout * 4 < (1LL << 32);
}

// Fills the walk's fields of `a` (nseg, units, split) for the split the caller chose as an upper bound and launches.
void eovae_c8_wgrad_rows_launch(WgCArgs& a, hipStream_t st) {
  const int strips = a.N * a.tiles_x;
  int nseg = 1;
  while (strips * nseg < a.split && a.Hout % (2 * nseg) == 0 && a.Hout / (2 * nseg) >= 8) nseg *= 2;
  a.nseg = nseg;
  a.units = strips * nseg;
  if (a.split > a.units) a.split = a.units;
  a.xcd_remap = (a.ncib * a.ncob_ > 1 && a.split % 8 == 0) ? 1 : 0;
  if (eovae_opt(EOVAE_OPT_C8_WGRAD_CORR))
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<0, true>), dim3(a.ncib * a.ncob_ * a.split), dim3(512), 0, st, a);
  else if (eovae_opt(EOVAE_OPT_C8_Q16))
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<0, false, true>), dim3(a.ncib * a.ncob_ * a.split), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<0, false>), dim3(a.ncib * a.ncob_ * a.split), dim3(512), 0, st, a);
}

// ---- the x2-upsample layer ("nearest x2 + 3x3", layers.py Upsample) in sub-pixel row-streaming form
#ifndef EOVAE_WGRAD_BLOCKS
#define EOVAE_WGRAD_BLOCKS 256   // one block per CU (conv_f16c8.hip)
#endif
static int up_rows_split(int Cin, int Cout) {
  int split = EOVAE_WGRAD_BLOCKS / (2 * cdiv(Cin, 64) * cdiv(Cout, 128));
  return split < 1 ? 1 : split;
}

extern "C" {

int eovae_conv3x3_f16c8_wgrad_up_eligible(int N, int Cin, int H, int W, int Cout) {
  return eovae_opt(EOVAE_OPT_WGRAD_ROWS) != 0 && N > 0 && Cout % 128 == 0 && Cin % 64 == 0 && W % 32 == 0 && H >= 1 &&
         (long long)H * W * 32 * 8 < (1LL << 31) && (long long)Cout * H * W * 16 < (1LL << 32);
}

// scratch: the records of x (4 B per element) + eight [128][64] fp32 tiles per block
size_t eovae_conv3x3_f16c8_wgrad_up_workspace(int N, int Cin, int H, int W, int Cout) {
  if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const size_t rec = ((size_t)N * Cin * H * W * 4 + 255) / 256 * 256;
  return rec + (size_t)2 * cdiv(Cin, 64) * cdiv(Cout, 128) * up_rows_split(Cin, Cout) * 8 * 128 * 64 * sizeof(float);
}

// Weight gradient of conv3x3(nearest_upsample_x2(x)): x fp32 [N][Cin][H][W], dy fp32 [N][Cout][2H][2W] -> dw[Cout][Cin][3][3]
// (written).  dyamax / xamax: absmax bits of dy / x (xamax nullable = no pre-scale).  db (nullable) accumulates the bias
// gradient into a pre-zeroed [Cout] (float atomics, as the other f16c8 weight-gradient entries).
int eovae_conv3x3_f16c8_wgrad_up(const float* x, const float* dy, float* dw, float* db, const unsigned* dyamax,
                                 const unsigned* xamax, int N, int Cin, int H, int W, int Cout, void* ws, void* stream) {
  EOVAE_CHECK_ARG(x && dy && dw && ws, "conv3x3_f16c8_wgrad_up: null pointer");
  EOVAE_CHECK_ARG(eovae_conv3x3_f16c8_wgrad_up_eligible(N, Cin, H, W, Cout), "conv3x3_f16c8_wgrad_up: shape N=%d Cin=%d %dx%d Cout=%d "
                  "(Cout %% 128, Cin %% 64, W %% 32)", N, Cin, H, W, Cout);
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W, pblocks = cdiv(HW, 256);
  int* rec = (int*)ws;
  hipLaunchKernelGGL(f16c8_records_kernel, dim3(N * (Cin / 8) * pblocks), dim3(256), 0, st, x, xamax, rec, Cin, HW, pblocks);
  WgCArgs a;
  a.x = reinterpret_cast<const float*>(rec); a.dy = dy; a.dwp = nullptr; a.db = db; a.dyamax = dyamax; a.xamax = xamax;
  a.part = reinterpret_cast<float*>((char*)ws + ((size_t)N * Cin * HW * 4 + 255) / 256 * 256);
  a.N = N; a.Cin = Cin; a.Hin = H; a.Win = W; a.Cout = Cout; a.Hout = 2 * H; a.Wout = 2 * W;
  a.mode = MODE_NORMAL; a.subpix = 1; a.creal = 0; a.cpb = 1; a.xpacked = 1;
  a.tiles_x = W / 32; a.tiles_y = H; a.ntiles = N * a.tiles_x * H;
  a.ncib = Cin / 64; a.ncob_ = Cout / 128;
  a.split = up_rows_split(Cin, Cout);
  const int strips = N * a.tiles_x;
  int nseg = 1;
  while (strips * nseg < a.split && H % (2 * nseg) == 0 && H / (2 * nseg) >= 8) nseg *= 2;
  a.nseg = nseg;
  a.units = strips * nseg;
  if (a.split > a.units) a.split = a.units;
  const int nslab = 2 * a.ncib * a.ncob_;
  a.xcd_remap = (a.split % 8 == 0) ? 1 : 0;
  if (eovae_opt(EOVAE_OPT_C8_WGRAD_CORR))
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<1, true>), dim3(nslab * a.split), dim3(512), 0, st, a);
  else if (eovae_opt(EOVAE_OPT_C8_Q16))
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<1, false, true>), dim3(nslab * a.split), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<1, false>), dim3(nslab * a.split), dim3(512), 0, st, a);
  hipLaunchKernelGGL(wgrad_up_fold_kernel, dim3((unsigned)(a.ncib * a.ncob_ * 128)), dim3(576), 0, st, a.part, dw, Cout, Cin, a.ncib,
                     a.split);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_wgrad_up");
  return EOVAE_OK;
}

// ---- the stride-2 layer through the same kernel, operands swapped (wgrad_down_fold_kernel)
int eovae_conv3x3_f16c8_wgrad_down_eligible(int N, int Cin, int H, int W, int Cout) {   // H, W: the OUTPUT (low-res) grid
  return eovae_conv3x3_f16c8_wgrad_up_eligible(N, Cout, H, W, Cin);
}

size_t eovae_conv3x3_f16c8_wgrad_down_workspace(int N, int Cin, int H, int W, int Cout) {
  return eovae_conv3x3_f16c8_wgrad_up_workspace(N, Cout, H, W, Cin);
}

// Weight gradient of conv3x3_stride2(pad(x, (0,1,0,1))): x fp32 [N][Cin][2H][2W], dy fp32 [N][Cout][H][W] -> dw[Cout][Cin][3][3]
// (written).  dyamax / xamax: absmax bits of dy / x (nullable).  No bias gradient here (eovae_bias_grad).
int eovae_conv3x3_f16c8_wgrad_down(const float* x, const float* dy, float* dw, const unsigned* dyamax, const unsigned* xamax, int N,
                                   int Cin, int H, int W, int Cout, void* ws, void* stream) {
  EOVAE_CHECK_ARG(x && dy && dw && ws, "conv3x3_f16c8_wgrad_down: null pointer");
  EOVAE_CHECK_ARG(eovae_conv3x3_f16c8_wgrad_down_eligible(N, Cin, H, W, Cout), "conv3x3_f16c8_wgrad_down: shape N=%d Cin=%d out %dx%d "
                  "Cout=%d (Cin %% 128, Cout %% 64, W %% 32)", N, Cin, H, W, Cout);
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W, pblocks = cdiv(HW, 256);
  int* rec = (int*)ws;   // records of dy: the kernel's low-res "X" operand
  hipLaunchKernelGGL(f16c8_records_kernel, dim3(N * (Cout / 8) * pblocks), dim3(256), 0, st, dy, dyamax, rec, Cout, HW, pblocks);
  WgCArgs a;
  a.x = reinterpret_cast<const float*>(rec); a.dy = x; a.dwp = nullptr; a.db = nullptr; a.dyamax = xamax; a.xamax = dyamax;
  a.part = reinterpret_cast<float*>((char*)ws + ((size_t)N * Cout * HW * 4 + 255) / 256 * 256);
  a.N = N; a.Cin = Cout; a.Hin = H; a.Win = W; a.Cout = Cin; a.Hout = 2 * H; a.Wout = 2 * W;
  a.mode = MODE_NORMAL; a.subpix = 1; a.creal = 0; a.cpb = 1; a.xpacked = 1;
  a.tiles_x = W / 32; a.tiles_y = H; a.ntiles = N * a.tiles_x * H;
  a.ncib = Cout / 64; a.ncob_ = Cin / 128;
  a.split = up_rows_split(Cout, Cin);
  const int strips = N * a.tiles_x;
  int nseg = 1;
  while (strips * nseg < a.split && H % (2 * nseg) == 0 && H / (2 * nseg) >= 8) nseg *= 2;
  a.nseg = nseg;
  a.units = strips * nseg;
  if (a.split > a.units) a.split = a.units;
  const int nslab = 2 * a.ncib * a.ncob_;
  a.xcd_remap = (a.split % 8 == 0) ? 1 : 0;
  if (eovae_opt(EOVAE_OPT_C8_WGRAD_CORR))
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<2, true>), dim3(nslab * a.split), dim3(512), 0, st, a);
  else if (eovae_opt(EOVAE_OPT_C8_Q16) == 3)   // (measured 0-3 % SLOWER than the 32 x 32 form on the stride-2 layers: opt-in only, tests)
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<2, false, true>), dim3(nslab * a.split), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((wgrad3x3_f16c8_rows_kernel<2, false>), dim3(nslab * a.split), dim3(512), 0, st, a);
  hipLaunchKernelGGL(wgrad_down_fold_kernel, dim3((unsigned)(a.ncib * a.ncob_ * 32)), dim3(576), 0, st, a.part, dw, Cout, Cin, a.ncib,
                     a.split);
  EOVAE_CHECK_LAUNCH("eovae_conv3x3_f16c8_wgrad_down");
  return EOVAE_OK;
}

}  // extern "C"
