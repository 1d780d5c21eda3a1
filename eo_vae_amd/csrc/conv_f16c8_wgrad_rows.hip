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
//   stage, 34 live lanes each: the f16 halves and the fp8 halves of one channel octet's 34 pixels; off-image pixels and rows are
//   offsets beyond the descriptor's range, which the hardware fills with zeros): no registers, no VALU, no ds_write.
//   Per wave and stage: ~40 VALU, 4 vector-memory and 31 LDS instructions for 27 MFMAs (tile kernel: ~135 / 10 / 62).
//
// LDS (118 KiB): three dY slots {f16 [128 co][80 B] | dyl' [128][48 B] | dyh' [128][48 B]} and three X slots
// {f16 planes [8 octets][34 px][16 B], pitch 1088 = 64 (mod 256) | fp8 planes [8 octets][34 px][xl' x 8 | xh' x 8] at
// offsets = {0, 136, 8, 128} (mod 256) per ci-tile}: every fragment read of a 32-lane group covers 64 distinct banks.
// The stage loop is unrolled by six so that slot, ring and register-set indices are compile-time.
//
// Restrictions (else the tile kernel runs): plain form (subpix 0, mode NORMAL), record input, partial-tile output,
// Cout % 128 == 0, Cin % 64 == 0, Wout % 32 == 0.
#include <stdlib.h>

#include <type_traits>

#include "conv_f16c8_common.h"

namespace {

constexpr int P16 = 80, P8 = 48;             // dY row pitches (bytes): 64 + 16, 32 + 16 -- odd multiples of 16
constexpr int DY16 = 128 * P16;              // f16 image of one dY row
constexpr int DY8 = 128 * P8;                // one fp8 image
constexpr int DYSLOT = DY16 + 2 * DY8;       // 22,528
// Every LDS-DMA instruction deposits 64 lanes x 16 B = 1 KiB (lanes 34..63 carry an out-of-range offset and deposit
// zeros): each octet's plane owns a 1 KiB window, so the instruction needs no EXEC mask -- and the request block no
// branch (behind a branch the compiler sinks the stage's first MFMAs below the whole block).
constexpr int XP16 = 1088;                   // f16 plane pitch per octet: 1 KiB window + 64 (= 64 mod 256)
constexpr int X16 = 8 * XP16;                // 8,704
constexpr int X8T = 4480;                    // fp8 planes of one ci-tile (4 octets, 1 KiB windows at the offsets of x8off())
constexpr int XSLOT = X16 + 2 * X8T;         // 17,664
constexpr int NRING = 3;

static_assert(true, "");
// windows of 1 KiB at {0, 136, 8, 128} (mod 256), at least 1 KiB apart
__device__ __forceinline__ constexpr int x8off(int o) { return o == 0 ? 0 : (o == 1 ? 1160 : (o == 2 ? 2312 : 3456)); }

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
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma8(f32x16& c, const i32x8& a, const i32x8& b, int sa, int sb) {
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

// one stage of the walk: X row R of strip (n, x0) against dY rows R + 1, R, R - 1 of the unit's rows [r0, r1)
struct RowStage {
  int n, x0, R, r0, r1;
  int live;   // 0: padding stage behind the block's last unit (all operands zero)
};

__global__ __launch_bounds__(512, 2) void wgrad3x3_f16c8_rows_kernel(WgCArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char dys[NRING * DYSLOT];
  // THREE arrays for the LDS-DMA targets: with one array and a computed slot offset the compiler cannot tell a DMA write
  // from the fragment reads of another slot and drains vmcnt(0) in front of every read (conv_f16c8_wgrad_w4.hip)
  __shared__ __attribute__((aligned(16))) unsigned char xs0[XSLOT];
  __shared__ __attribute__((aligned(16))) unsigned char xs1[XSLOT];
  __shared__ __attribute__((aligned(16))) unsigned char xs2[XSLOT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, li = lane & 15;
  const int ct = wave & 3, cih = wave >> 2;   // waves w and w + 4 share a SIMD: same co-tile, the two ci-tiles
  int slab, ysub;
  {   // XCD-aware block -> (slab, split slot) map, as wgrad3x3_f16c8_kernel
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
  const int H = a.Hout, W = a.Wout;
  const unsigned plane = (unsigned)(H * W);

  float dys_scale = 1.f, dyinv = 1.f;   // dy pre-scale: the power of two that puts its amax at 2^13..2^14
  if (a.dyamax) {
    const unsigned bits = *a.dyamax;
    if (bits) {
      const int e = expo_of_bits(bits);
      dys_scale = pow2f(13 - e);
      dyinv = pow2f(e - 13);
    }
  }
  const int sc_a = 127 - 12, sc_b = 127;   // E8M0: both K blocks carry one operand scaled by 2^12

  // ---- the walk: units u = ysub, ysub + split, ... ; unit = (image n, strip tx, row segment seg)
  const int segrows = __builtin_amdgcn_readfirstlane(H / a.nseg);
  int unit = __builtin_amdgcn_readfirstlane(ysub - a.split);
  RowStage cur;
  cur.n = 0; cur.x0 = 0; cur.R = 0; cur.r0 = 0; cur.r1 = -1; cur.live = 1;   // R > r1: the first next() opens a unit
  // BRANCH-FREE (selects, divisions by multiply-high with host-made reciprocals): the cursor steps inside the MFMA
  // sequence, and behind a branch the compiler sinks the MFMAs in front of it into the join block.  Every field goes
  // through readfirstlane, so that a buffer descriptor built from it is not loaded inside a waterfall loop.
  auto next = [&]() __attribute__((always_inline)) {
    const int open = (cur.R >= cur.r1) & cur.live;
    unit = uni(unit + (open ? a.split : 0));
    const int dead = open & (unit >= a.units);
    const unsigned u = (unsigned)(dead ? 0 : unit);
    const unsigned t = a.nseg == 1 ? u : __umulhi(u, a.magic_nseg), seg = u - t * (unsigned)a.nseg;       // (the reciprocal
    const unsigned n = a.tiles_x == 1 ? t : __umulhi(t, a.magic_tx), tx = t - n * (unsigned)a.tiles_x;   // of 1 is 2^32)
    const int r0n = (int)seg * segrows;
    cur.live = uni(cur.live & !dead);
    cur.x0 = uni(open ? (int)tx * 32 : cur.x0);
    cur.n = uni(open ? (int)n : cur.n);
    cur.R = uni(open ? r0n - 1 : cur.R + 1);
    cur.r0 = uni(open ? r0n : cur.r0);
    cur.r1 = uni(open ? r0n + segrows : cur.r1);
  };
  int nst;   // stages of this block
  {
    const int mine = ysub < a.units ? (a.units - ysub + a.split - 1) / a.split : 0;
    nst = mine * (segrows + 2);
  }

  // ---- dY staging: thread -> (co = tid >> 2, pixels 8 (tid & 3) .. + 7): two 16-byte loads, 28 conversions, three stores.
  // The loads are UNCONDITIONAL: a row outside the unit is an offset beyond the descriptor's range and comes back as
  // zeros (a branch around the loads would make the registers phi nodes behind an s_waitcnt vmcnt(0)).  Two register
  // sets: the row of stage s + 2 is requested at the top of stage s, the row of stage s + 1 is converted piecewise
  // between the MFMAs of stage s.
  const int dco = tid >> 2, dq = tid & 3;
  const unsigned dvoff = ((unsigned)(cob * 128 + dco) * plane + 8u * (unsigned)dq) * 4u;
  const unsigned dyspan = (unsigned)a.Cout * plane * 4u;
  f32x4 dr[2][2];
  float dbs = 0.f;
  auto load_dy = [&](const RowStage& s, auto pc) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value;
    const int y = s.R + 1;
    const bool ok = uni(s.live && y >= s.r0 && y < s.r1) != 0;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        uni_ptr(const_cast<float*>(a.dy) + (size_t)(ok ? s.n : 0) * a.Cout * plane), 0, dyspan, 0x00020000);
    const int so = uni(ok ? (y * W + s.x0) * 4 : 0);
    const unsigned vo = ok ? dvoff : 0x80000000u;
    dr[P][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, vo, so, 0));
    dr[P][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, vo + 16, so, 0));
  };
  const int dw16 = dco * P16 + dq * 16, dw8 = DY16 + dco * P8 + dq * 8;
  // conversion of one register set in three pieces placed between MFMA triples: pixels 0-3 | pixels 4-7 | stores (every
  // intermediate is dead at the end of its piece: only the 8 result registers live across MFMAs)
  i32x4 chi;
  i32x2 cl8, ch8;
  auto conv_piece = [&](auto pc, auto piece, unsigned char* slot) __attribute__((always_inline)) {
    constexpr int P = decltype(pc)::value, I = decltype(piece)::value;
    if constexpr (I < 2) {
      const f32x4 d = dr[P][I];
      float v[4], lo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = d[j] * dys_scale;
      float t0 = d[0] + d[1], t1 = d[2] + d[3];
      asm volatile("" : "+v"(t0), "+v"(t1));   // scalar adds: packed f32 VALU is slow beside MFMAs
      dbs += t0 + t1;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        int hp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[j]), "v"(v[j + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp), "v"(v[j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp), "v"(v[j + 1]));
        chi[2 * I + (j >> 1)] = hp;
      }
      s16x2 lv = {0, 0};
      lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[0], lo[1], 0x1p-12f, false);
      lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[2], lo[3], 0x1p-12f, true);
      cl8[I] = __builtin_bit_cast(int, lv);
      int hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
      hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], hv, true);
      ch8[I] = hv;
    } else {
      *reinterpret_cast<i32x4*>(slot + dw16) = chi;
      *reinterpret_cast<i32x2*>(slot + dw8) = cl8;          // image 0: dyl'
      *reinterpret_cast<i32x2*>(slot + dw8 + DY8) = ch8;    // image 1: dyh'
    }
  };

  // ---- X by LDS-DMA: wave w moves channel octet w of the slab (ci-tile w >> 2, octet w & 3 of it)
  const unsigned OOR = 0x80000000u;
  const unsigned xspan = 8u * plane * 32u;
  const int x16dst = uni(wave * XP16), x8dst = uni(X16 + (wave >> 2) * X8T + x8off(wave & 3));
  auto dma_x = [&](unsigned char* xslot, const RowStage& s) __attribute__((always_inline)) {
    unsigned ln = threadIdx.x & 63u;
    asm volatile("" : "+v"(ln));   // recomputed at its use, not carried in a register across the MFMA phases
    const int px = s.x0 - 1 + (int)ln;
    const bool rowok = uni(s.live && (unsigned)s.R < (unsigned)H) != 0;
    const bool ok = rowok && ln < 34u && (unsigned)px < (unsigned)W;
    const unsigned v = ok ? ((unsigned)wave * plane + (unsigned)(uni(s.R * W) + px)) * 32u : OOR;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        uni_ptr(const_cast<char*>(reinterpret_cast<const char*>(a.x)) +
                ((size_t)(rowok ? s.n : 0) * (a.Cin / 8) + (size_t)cib * 8) * plane * 32),
        0, xspan, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(xslot + x16dst), 16, (int)v, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(xslot + x8dst), 16,
                                             (int)(ok ? v + 16u : OOR), 0, 0, 0);
  };

  // ---- fragment addresses (per-lane constants; slot, tap shift and k-step are immediates)
  // A (dY): lane (r, h) reads row 32 ct + r; f16: 8 pixels at 16 h of the k-step; fp8: the row's 32 bytes of image h
  const int a16off = (32 * ct + r) * P16 + 16 * h;
  const int a8off = DY16 + h * DY8 + (32 * ct + r) * P8;
  // B (X) f16 by ds_read_b64_tr_b16: lane 4q + p of group g supplies pixel q of its 4-pixel block, 8-byte chunk p & 1 of
  // octet 2 (g & 1) + (p >> 1) of the wave's ci-tile; k-half g >> 1
  const int b16off = (4 * cih + 2 * (g & 1) + ((li & 3) >> 1)) * XP16 + (8 * (g >> 1) + (li >> 2)) * 16 + (li & 1) * 8;
  // B fp8 by ds_read_b64_tr_b8: lane 2j + p supplies pixel j of its 8-pixel block, octet 2 (g & 1) + p; K block g >> 1:
  // block 0 = xh' (second 8 bytes of the pixel's 16), block 1 = xl'
  const int b8off = X16 + cih * X8T + x8off(2 * (g & 1) + (li & 1)) + (li >> 1) * 16 + ((g >> 1) ? 0 : 8);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  f16x8 A16[NRING][2];
  i32x8 A8[NRING];
#pragma unroll
  for (int k = 0; k < NRING; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { A16[k][0][j] = (_Float16)0.f; A16[k][1][j] = (_Float16)0.f; A8[k][j] = 0; }
  }

  auto xslot_of = [&](auto kc) __attribute__((always_inline)) -> unsigned char* {
    constexpr int k = decltype(kc)::value % 3;
    if constexpr (k == 0) return xs0;
    else if constexpr (k == 1) return xs1;
    else return xs2;
  };

  // ---- prologue: X(0), X(1) in flight, dY(0) committed, dY(1) in registers; `cur` = the stage requested next
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  next();
  dma_x(xs0, cur);
  load_dy(cur, C0{});
  next();
  dma_x(xs1, cur);
  load_dy(cur, C1{});
  conv_piece(C0{}, C0{}, dys);
  conv_piece(C0{}, C1{}, dys);
  conv_piece(C0{}, C2{}, dys);
  next();            // stage 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // stage s: K = s mod 3 (LDS slots, register ring), P = s mod 2 (dY register set that receives row s + 2; row s + 1 sits in the other)
  auto body = [&](auto kc, auto pc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value, P = decltype(pc)::value;
    constexpr int KD = (K + 1) % 3, KX = (K + 2) % 3;
    using PN = std::integral_constant<int, P ^ 1>;
    const unsigned char* dsl = dys + K * DYSLOT;
    unsigned char* dnx = dys + KD * DYSLOT;
    const unsigned char* xsl = xslot_of(kc);
    // A stage opens with every wave of the CU behind the same barrier and the matrix pipes idle: nothing but the first
    // triple's operands is requested in front of it, and that triple multiplies the two OLD dY rows (registers) first.
    // The requests of X row s + 2 and dY row s + 2 (address arithmetic, two loads, two LDS-DMAs, the cursor step: ~90
    // scalar / vector instructions) ride between the MFMAs of kx = 0; the conversion of dY row s + 1 (requested a stage
    // ago) between the triples of kx = 1.
    i32x8 b8 = rtr8(xsl + b8off);
    {
      const i32x4 l = *reinterpret_cast<const i32x4*>(dsl + a8off), u = *reinterpret_cast<const i32x4*>(dsl + a8off + 16);
      A8[K] = (i32x8)__builtin_shufflevector(l, u, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off));
    A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off + 32));
    constexpr int R1 = (K + 2) % 3, R2 = (K + 1) % 3;   // ring positions of ky = 1, 2 (ky = 0: K)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f16x8 b0 = rtr16(xsl + b16off + kx * 16, xsl + b16off + (kx + 4) * 16);
      if (kx == 1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // newer than dY(s + 1): X(s + 1), dY(s + 2), X(s + 2) -- two operations each
        conv_piece(PN{}, C0{}, dnx);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma8(acc[3 + kx], A8[R1], b8, sc_a, sc_b);
      if (kx == 0) load_dy(cur, pc);
      mfma8(acc[6 + kx], A8[R2], b8, sc_a, sc_b);
      if (kx == 0) dma_x(xslot_of(std::integral_constant<int, KX>{}), cur);
      mfma8(acc[kx], A8[K], b8, sc_a, sc_b);
      if (kx == 0) next();
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 b1 = rtr16(xsl + b16off + (kx + 16) * 16, xsl + b16off + (kx + 20) * 16);
      if (kx == 1) conv_piece(PN{}, C1{}, dnx);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(acc[3 + kx], A16[R1][0], b0);
      mfma16(acc[6 + kx], A16[R2][0], b0);
      mfma16(acc[kx], A16[K][0], b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kx < 2) b8 = rtr8(xsl + b8off + (kx + 1) * 16);
      if (kx == 1) conv_piece(PN{}, C2{}, dnx);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(acc[3 + kx], A16[R1][1], b1);
      mfma16(acc[6 + kx], A16[R2][1], b1);
      mfma16(acc[kx], A16[K][1], b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // X(s + 1) has landed when at most this stage's four operations are outstanding
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
  };

  {
    int s = 0;
#pragma unroll 1
    while (true) {
      body(C0{}, C0{}); if (++s >= nst) break;
      body(C1{}, C1{}); if (++s >= nst) break;
      body(C2{}, C0{}); if (++s >= nst) break;
      body(C0{}, C1{}); if (++s >= nst) break;
      body(C1{}, C0{}); if (++s >= nst) break;
      body(C2{}, C1{}); if (++s >= nst) break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // requests of the two stages behind
                                                                                     // the last one; the asm MFMAs' results

  if (a.db != nullptr && cib == 0) {   // lanes 4 co .. 4 co + 3 hold the same channel
    float v = dbs;
    v += dpp_quad<0xB1>(v);
    v += dpp_quad<0x4E>(v);
    if (dq == 0) atomicAdd(&a.db[cob * 128 + dco], v);
  }
  // this block's partial tile [tap][128 co][64 ci], folded in a fixed order by wgrad_fold_kernel
  float* pt = a.part + (((size_t)slab * a.split + ysub) * 9) * (128 * 64) + cih * 32 + r;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int col = ct * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      pt[(size_t)t * (128 * 64) + col * 64] = acc[t][q] * dyinv;
    }
}

}  // namespace

bool eovae_c8_wgrad_rows_eligible(const WgCArgs& a) {
  return eovae_opt(EOVAE_OPT_WGRAD_ROWS) != 0 && a.part && a.subpix == 0 && a.mode == MODE_NORMAL && a.xpacked && a.Cout % 128 == 0 &&
         a.Cin % 64 == 0 && a.Wout % 32 == 0 && a.Hout >= 1 && !a.xamax && (long long)a.N * a.tiles_x * a.Hout < 65536 &&
         (long long)a.Hout * a.Wout * 32 * 8 < (1LL << 31) && (long long)a.Cout * a.Hout * a.Wout * 4 < (1LL << 32);
}

// Fills the walk's fields of `a` (nseg, units, split) for the split the caller chose as an upper bound and launches.
void eovae_c8_wgrad_rows_launch(WgCArgs& a, hipStream_t st) {
  const int strips = a.N * a.tiles_x;
  int nseg = 1;
  while (strips * nseg < a.split && a.Hout % (2 * nseg) == 0 && a.Hout / (2 * nseg) >= 8) nseg *= 2;
  a.nseg = nseg;
  a.units = strips * nseg;
  // floor(u / d) = umulhi(u, floor(2^32 / d) + 1) for u < 2^16, d < 2^16 (units and strips are far below that)
  a.magic_nseg = (unsigned)(0x100000000ull / (unsigned)nseg) + 1u;
  a.magic_tx = (unsigned)(0x100000000ull / (unsigned)a.tiles_x) + 1u;
  if (a.split > a.units) a.split = a.units;
  a.xcd_remap = (a.ncib * a.ncob_ > 1 && a.split % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL(wgrad3x3_f16c8_rows_kernel, dim3(a.ncib * a.ncob_ * a.split), dim3(512), 0, st, a);
}
