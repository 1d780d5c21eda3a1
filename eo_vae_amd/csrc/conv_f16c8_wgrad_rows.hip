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
//   stage, 34 lanes each: the f16 halves and the fp8 halves of one channel octet's 34 pixels; off-image pixels and rows are
//   offsets beyond the descriptor's range, which the hardware fills with zeros): no registers, no VALU, no ds_write.
//   Per wave and stage: ~40 VALU, 4 vector-memory and 31 LDS instructions for 27 MFMAs (tile kernel: ~135 / 10 / 62).
//
// LDS (96 KiB): three dY slots {f16 [128 co][80 B] | dyl' [128][48 B] | dyh' [128][48 B]} and three X slots
// {f16 planes [8 octets][34 px][16 B], pitch 576 = 64 (mod 256) | fp8 planes [8 octets][34 px][xl' x 8 | xh' x 8] at
// offsets = {0, 136, 8, 128} (mod 256) per ci-tile}: every fragment read of a 32-lane group covers 64 distinct banks.
// The stage loop is unrolled by three so that slot and ring indices are compile-time.
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
constexpr int XP16 = 576;                    // f16 plane pitch per octet: 34 px x 16 B = 544, +32
constexpr int X16 = 8 * XP16;                // 4,608
constexpr int X8T = 2464;                    // fp8 planes of one ci-tile (4 octets x 544 B at the offsets of x8off())
constexpr int XSLOT = X16 + 2 * X8T;         // 9,536
constexpr int NRING = 3;

__device__ __forceinline__ constexpr int x8off(int o) { return o == 0 ? 0 : (o == 1 ? 648 : (o == 2 ? 1288 : 1920)); }

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
  const int segrows = H / a.nseg;
  int unit = ysub - a.split;
  RowStage cur;
  cur.n = 0; cur.x0 = 0; cur.R = 0; cur.r0 = 0; cur.r1 = -1; cur.live = 1;   // R > r1: the first next() opens a unit
  auto next = [&]() __attribute__((always_inline)) {
    if (!cur.live) return;
    if (cur.R >= cur.r1) {
      unit += a.split;
      if (unit >= a.units) { cur.live = 0; return; }
      const int seg = unit % a.nseg;
      const int t = unit / a.nseg;
      cur.x0 = (t % a.tiles_x) * 32;
      cur.n = t / a.tiles_x;
      cur.r0 = seg * segrows;
      cur.r1 = cur.r0 + segrows;
      cur.R = cur.r0 - 1;
    } else {
      ++cur.R;
    }
  };
  int nst;   // stages of this block, rounded up to a multiple of three
  {
    const int mine = ysub < a.units ? (a.units - ysub + a.split - 1) / a.split : 0;
    nst = mine * (segrows + 2);
    nst = (nst + 2) / 3 * 3;
  }

  // ---- dY staging: thread -> (co = tid >> 2, pixels 8 (tid & 3) .. + 7): two 16-byte loads, 28 conversions, three stores
  const int dco = tid >> 2, dq = tid & 3;
  const unsigned dvoff = ((unsigned)(cob * 128 + dco) * plane + 8u * (unsigned)dq) * 4u;
  const unsigned dyspan = (unsigned)a.Cout * plane * 4u;
  f32x4 dr0, dr1;
  int dlive = 0;
  float dbs = 0.f;
  auto load_dy = [&](const RowStage& s) __attribute__((always_inline)) {
    const int y = s.R + 1;
    dlive = s.live && y >= s.r0 && y < s.r1;   // uniform
    if (dlive) {
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.dy) + (size_t)s.n * a.Cout * plane, 0, dyspan, 0x00020000);
      const int so = (y * W + s.x0) * 4;
      dr0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff, so, 0));
      dr1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, dvoff + 16, so, 0));
    }
  };
  const int dw16 = dco * P16 + dq * 16, dw8 = DY16 + dco * P8 + dq * 8;
  auto commit_dy = [&](unsigned char* slot) __attribute__((always_inline)) {
    i32x4 hi = {0, 0, 0, 0};
    i32x2 l8 = {0, 0}, h8 = {0, 0};
    if (dlive) {
      float v[8], lo[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = dr0[j] * dys_scale; v[4 + j] = dr1[j] * dys_scale; }
      dbs += ((dr0[0] + dr0[1]) + (dr0[2] + dr0[3])) + ((dr1[0] + dr1[1]) + (dr1[2] + dr1[3]));
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        int hp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[j]), "v"(v[j + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp), "v"(v[j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp), "v"(v[j + 1]));
        hi[j >> 1] = hp;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s16x2 lv = {0, 0};
        lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[4 * j], lo[4 * j + 1], 0x1p-12f, false);
        lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[4 * j + 2], lo[4 * j + 3], 0x1p-12f, true);
        l8[j] = __builtin_bit_cast(int, lv);
        int hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[4 * j], v[4 * j + 1], 0, false);
        hv = __builtin_amdgcn_cvt_pk_bf8_f32(v[4 * j + 2], v[4 * j + 3], hv, true);
        h8[j] = hv;
      }
    }
    *reinterpret_cast<i32x4*>(slot + dw16) = hi;
    *reinterpret_cast<i32x2*>(slot + dw8) = l8;          // image 0: dyl'
    *reinterpret_cast<i32x2*>(slot + dw8 + DY8) = h8;    // image 1: dyh'
  };

  // ---- X by LDS-DMA: wave w moves channel octet w of the slab (ci-tile w >> 2, octet w & 3 of it)
  const unsigned OOR = 0x80000000u;
  const unsigned xspan = 8u * plane * 32u;
  auto dma_x = [&](unsigned char* xslot, const RowStage& s) __attribute__((always_inline)) {
    unsigned ln = threadIdx.x & 63u;
    asm volatile("" : "+v"(ln));   // recomputed at its use, not carried in a register across the MFMA phases
    const int px = s.x0 - 1 + (int)ln;
    const bool ok = s.live && (unsigned)s.R < (unsigned)H && (unsigned)px < (unsigned)W;
    const unsigned v = ok ? ((unsigned)wave * plane + (unsigned)(s.R * W + px)) * 32u : OOR;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x)) + ((size_t)s.n * (a.Cin / 8) + (size_t)cib * 8) * plane * 32, 0, xspan,
        0x00020000);
    if (ln < 34u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(xslot + wave * XP16), 16,
                                               (int)v, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          srd, (__attribute__((address_space(3))) void*)(xslot + X16 + (wave >> 2) * X8T + x8off(wave & 3)), 16,
          (int)(ok ? v + 16u : OOR), 0, 0, 0);
    }
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

  // cursors: sD = stage whose dY row is loaded next, sX = stage whose X row is DMAed next
  RowStage sD, sX;
  // ---- prologue: X(0), dY(0) committed, X(1) in flight
  next();
  sX = cur;
  dma_x(xs0, sX);
  load_dy(sX);
  next();
  sD = cur;          // stage 1
  dma_x(xs1, cur);
  commit_dy(dys);
  next();
  sX = cur;          // stage 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto body = [&](auto kc) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    constexpr int KD = (K + 1) % 3, KX = (K + 2) % 3;
    unsigned char* dsl = dys + K * DYSLOT;
    const unsigned char* xsl = xslot_of(kc);
    // dY row of the next stage -> registers, X row of the stage after it -> LDS (the DMAs are the two newest
    // vector-memory operations when the registers are waited for)
    load_dy(sD);
    dma_x(xslot_of(std::integral_constant<int, KX>{}), sX);
    // the new dY row's A fragments
    {
      A16[K][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off));
      A16[K][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(dsl + a16off + 32));
      const i32x4 l = *reinterpret_cast<const i32x4*>(dsl + a8off), u = *reinterpret_cast<const i32x4*>(dsl + a8off + 16);
      A8[K] = (i32x8)__builtin_shufflevector(l, u, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // nine B fragments per stage (per kx: fp8, f16 k-step 0, f16 k-step 1), each requested one MFMA triple ahead
    i32x8 b8 = rtr8(xsl + b8off);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f16x8 b0 = rtr16(xsl + b16off + kx * 16, xsl + b16off + (kx + 4) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8[(K + 3 - ky) % 3], b8, acc[ky * 3 + kx], 1 /* A: e5m2 */,
                                                                           1 /* B: e5m2 */, 0, sc_a, 0, sc_b);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 b1 = rtr16(xsl + b16off + (kx + 16) * 16, xsl + b16off + (kx + 20) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A16[(K + 3 - ky) % 3][0], b0, acc[ky * 3 + kx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kx < 2) b8 = rtr8(xsl + b8off + (kx + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A16[(K + 3 - ky) % 3][1], b1, acc[ky * 3 + kx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // dY registers are back when at most the two DMAs issued after them are outstanding; the DMAs of the previous stage
    // (older) have landed with them
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    commit_dy(dys + KD * DYSLOT);
    __syncthreads();
    sD = sX;
    next();
    sX = cur;
  };

#pragma unroll 1
  for (int s = 0; s < nst; s += 3) {
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
    body(std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMAs of the two stages behind the last one

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
  hipLaunchKernelGGL(wgrad3x3_f16c8_rows_kernel, dim3(a.ncib * a.ncob_ * a.split), dim3(512), 0, st, a);
}
