// Weight gradient of a plain 3x3 layer with f16c8 operands -- ONE WAVE PER SIMD.
//   dWp[tap][co][ci] += sum_{n, pixels} dY[n][co][pix] * X[n][ci][pix @ tap]
// Same arithmetic, LDS images, transposing fragment reads and C-ABI entry as wgrad3x3_f16c8_kernel (conv_f16c8.hip):
// block = slab of 128 co x 64 ci x 9 taps, K = pixels in tiles of 2 output rows x 32, per row of 32 pixels and tap
//   2 x v_mfma_f32_32x32x16_f16 (dy_hi . x_hi)  +  1 x v_mfma_scale_f32_32x32x64_f8f6f4 ([dy_lo' | dy_hi'] . [x_hi' | x_lo']).
// What changes is the shape of the work on a CU:
//   conv_f16c8.hip   8 waves (2 per SIMD), wave tile 2 co-tiles x 1 ci-tile x HALF the taps (5 | 4): 160 accumulators;
//                    the two waves of a SIMD convert (VALU) and multiply (MFMA) at the same time and the two add up
//                    (SQ counters: MFMA busy 38 %, VALU 2 x 23.5 %), one barrier per tile with both waves in step
//   this file        4 waves (1 per SIMD, 512 registers), wave tile 2 co-tiles x 1 ci-tile x ALL 9 taps: 288
//                    accumulators (balanced: 54 MFMAs = 2304 cycles per wave and row).  X arrives as GroupNorm operand
//                    records and goes global -> LDS by LDS-DMA (buffer_load ... lds: the swizzle sits on the per-lane
//                    SOURCE offset, the halo's off-image pixels are offsets beyond the descriptor's range, which the
//                    hardware fills with zeros): no registers, no VALU conversion, no ds_write for X.  dY (fp32) goes
//                    through 32 registers: loads in row 0, conversion + LDS writes in row 1, one item per tap slot of
//                    the same wave's MFMA sequence; no branch inside the tile sequence.
//                    (First build: X through registers too = 80 staging registers; with 288 accumulators the compiler
//                    spilled 260 registers INTO the tile loop, every reload behind an s_waitcnt vmcnt(0) that drained
//                    the prefetch: 3.5 x SLOWER than the 8-wave kernel.)
// STATUS: correct (tests/test_gpu_wgrad_w4.py) and PARKED -- 0.77-0.87 x the 8-wave kernel at the model's shapes, so it
// only runs with EOVAE_C8_WGRAD_W4=1.  One wave per SIMD hides ~24 issue cycles per 32-cycle MFMA; a weight-gradient
// tile (216 MFMAs = 4,608 cycles) needs ~3,600 of issue for its dY split, LDS-DMA instructions and their address math,
// and fragment reads -- the 8-wave kernel has twice the issue slots for the same MFMA time.  A/B, timing-only
// ablations (EOVAE_WG4_ABL) and SQ counters: profiles/r03k_wgrad_w4_ablation.log, profiles/r03i_pmc_sq.json, DESIGN.md s5.0.
// Restrictions (else the 8-wave kernel runs): plain form (subpix 0, mode NORMAL), record input, Cout % 128 == 0,
// Cin % 64 == 0, Wout % 32 == 0, Hout % 2 == 0.
#include <stdlib.h>

#include <type_traits>

#include "conv_f16c8_common.h"

#if EOVAE_PARKED
namespace {

#ifndef EOVAE_WG4_ABL
#define EOVAE_WG4_ABL 0   // timing-only experiments (wrong results): 1 no staging, 2 no fragment requests in the loop, 4 no barrier
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f16x8 tr16(const unsigned char* p0, const unsigned char* p1) {
  using lds4 = __attribute__((address_space(3))) s16x4;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p1));
  return __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int STEP = 256>   // bytes between 8-pixel groups: 8 x the image's pixel pitch
__device__ __forceinline__ i32x8 tr8(const unsigned char* p) {
  using lds2 = __attribute__((address_space(3))) i32x2;
  const i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p));
  const i32x2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + STEP));
  const i32x2 v2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 2 * STEP));
  const i32x2 v3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2*)(p + 3 * STEP));
  const i32x4 lo = __builtin_shufflevector(v0, v1, 0, 1, 2, 3), hi = __builtin_shufflevector(v2, v3, 0, 1, 2, 3);
  return (i32x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 1) void wgrad3x3_f16c8_w4_kernel(WgCArgs a) {
  constexpr int NT = 9, XW = 40, XWV = 34, NXP = 4 * XW;   // 2 output rows + halo; LDS row pitch 40 pixels (34 valid)
  constexpr int DPART = 4 * 64 * 64;                        // f16 [4 co-tiles][64 pix][32 co]
  constexpr int D8IMG = 4 * 64 * 32, D8 = 2 * D8IMG;        // e5m2 [dyl' | dyh'][4 co-tiles][64 pix][32 co]
  constexpr int XPART = 2 * NXP * 64;                       // f16 [2 ci-tiles][160 pix][32 ci]
  constexpr int X8 = 2 * NXP * 64;                          // e5m2 [2 ci-tiles][160 pix][4 octets][xl' x 8 | xh' x 8]: the
                                                            // records' second halves as they are (16-byte LDS-DMA granules)
  constexpr int OFF_D8 = DPART, OFF_X = DPART + D8, OFF_X8 = OFF_X + XPART;
  constexpr int BUF = OFF_X8 + X8;                          // 72 KiB, two of them
  // TWO arrays, and a tile loop unrolled by two so that every tile body reads one array and DMAs into the other by
  // name: with one array and a run-time buffer offset the compiler cannot tell an LDS-DMA write from the fragment
  // reads and put an s_waitcnt vmcnt(0) -- a drain of the whole prefetch -- in front of the reads of every slot
  __shared__ __attribute__((aligned(16))) unsigned char smem0[BUF];
  __shared__ __attribute__((aligned(16))) unsigned char smem1[BUF];
  static_assert(2 * BUF <= 160 * 1024, "LDS");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int ctp = wave & 1, cih = wave >> 1;   // wave tile: co-tiles 2 ctp, 2 ctp + 1; ci-tile cih; all nine taps
  const int ct = 2 * ctp;
  // XCD-aware block -> (slab, tile stride slot) map, as wgrad3x3_f16c8_kernel
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
  const size_t in_plane = (size_t)a.Hin * a.Win;
  const size_t out_plane = (size_t)a.Hout * a.Wout;
  // lane offsets of the transposed fragment reads (derivations: conv_f16c8.hip), RECOMPUTED where a pass starts from a
  // laundered lane id (see lane_now below): as loop-long constants they were the registers the compiler spilled
  struct LaneC { int lpart, lchunk, sa0, sa1, l8, lx8; };
  auto lane_consts = [&]() __attribute__((always_inline)) {
    unsigned l = threadIdx.x & 63u;
    asm volatile("" : "+v"(l));
    const int g_ = (int)(l >> 4), li_ = (int)(l & 15u), lq_ = li_ >> 2, lp_ = li_ & 3;
    LaneC c;
    c.lpart = 8 * (g_ >> 1) + lq_;
    c.lchunk = 4 * (g_ & 1) + lp_;
    c.sa0 = c.lpart * 64 + ((c.lchunk ^ ((c.lpart >> 1) & 7)) * 8);
    c.sa1 = (c.lpart + 4) * 64 + ((c.lchunk ^ (((c.lpart + 4) >> 1) & 7)) * 8);
    c.l8 = (li_ >> 1) * 32 + ((g_ & 1) * 2 + (li_ & 1)) * 8;                 // dY fp8 images: [pixel][32 channel bytes]
    // X fp8 image: source lane (j, p) of group G points at pixel P + j, octet 2 (G & 1) + p, half G >> 1 (K block 0 = xh',
    // the record's LAST 8 bytes; K block 1 = xl')
    c.lx8 = (li_ >> 1) * 64 + ((g_ & 1) * 2 + (li_ & 1)) * 16 + ((g_ >> 1) ? 0 : 8);
    return c;
  };
  auto sx_of = [&](const LaneC& c, int res) __attribute__((always_inline)) {
    return c.lpart * 64 + (((c.lchunk >> 1) ^ (((res + c.lpart) >> 1) & 3)) * 16) + (c.lchunk & 1) * 8;
  };
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

  f32x16 acc[NT - 1][2];   // [tap 0..7][co-tile]: AGPRs
  f32x16 accv0, accv1;     // tap 8: VGPR-form MFMAs (see row_pass)
#pragma unroll
  for (int t = 0; t < NT - 1; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accv0[q] = 0.f; accv1[q] = 0.f; }
  const int sc_av = sc_a, sc_bv = sc_b;

  // ------------------------------------------------------------------ staging
  // dY: 8 items per tile through registers -- item d = 4 hf + 2 i + qh: row hf, pixels 16 i + (tid & 15), channels
  //     4 (tid >> 4) + 64 qh + 0..3.
  // X : 25 LDS-DMA instructions per wave and tile straight from the record tensor:
  //     5 x 1 KiB of the f16 image  (wave w: ci-tile w >> 1, pieces 5 (w & 1) .. + 4; piece = 16 halo pixels x 4 octets;
  //                                  lane l -> pixel l >> 2, 16-byte unit l & 3 = octet ^ ((pixel >> 1) & 3))
  //     20 x 256 B of an fp8 image  (wave w: image w >> 1 (0 = xh', 1 = xl'), ci-tile w & 1; piece = 8 halo pixels;
  //                                  lane l -> pixel l >> 3, octet (l >> 1) & 3, 4-byte half l & 1)
  float dyr[8][4];
  float dbs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool do_db = a.db != nullptr && cib == 0;
  struct TileC { int n, oy0, ox0; };
  auto decode = [&](int tile) __attribute__((always_inline)) {
    TileC t;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    t.n = tile / a.tiles_y;
    t.oy0 = ty * 2;
    t.ox0 = tx * 32;
    return t;
  };
  auto tid_now = [&]() __attribute__((always_inline)) {
    unsigned t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    return t_;
  };
  auto issue_dy = [&](const TileC& tc, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value, hf = I >> 2, i = (I >> 1) & 1, qh = I & 1;
    const char* sb = reinterpret_cast<const char*>(
        a.dy + ((size_t)tc.n * a.Cout + cob * 128 + 64 * qh) * out_plane + (size_t)(tc.oy0 + hf) * a.Wout + tc.ox0 + 16 * i);
    const unsigned t_ = tid_now();
    const unsigned vo = (t_ >> 4) * 4u * (unsigned)out_plane * 4u + (t_ & 15u) * 4u;   // channel quartet, pixel
#pragma unroll
    for (int j = 0; j < 4; ++j) dyr[I][j] = *reinterpret_cast<const float*>(sb + (size_t)j * out_plane * 4 + vo);
  };
  // LDS destinations of the dY tasks (lane constants; k-step / quartet-half parts are compile-time offsets)
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  int dc0 = 0, dc1 = 0;   // LDS lane offsets of the dY tasks, computed by the tile's first commit
  float dys_stage = dys;   // dY scale of the tile being staged (0: dead tile at the end of an odd walk)
  float dbw = 1.f;   // 0 while the walk's last tile is staged a second time (its bytes go to the idle buffer): not summed twice
  auto commit_dy = [&](unsigned char* base, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value, hf = I >> 2, i = (I >> 1) & 1, qh = I & 1, ks = 2 * hf + i;
    if constexpr (I == 0) {
      const int t_ = (int)tid_now(), pixl = t_ & 15, quad = t_ >> 4;
      dc0 = (quad >> 3) * 4096 + pixl * 64 + (((quad & 7) ^ ((pixl >> 1) & 7)) * 8);
      dc1 = OFF_D8 + (quad >> 3) * (64 * 32) + pixl * 32 + (quad & 7) * 4;
    }
    float v[4], lo[4];
    int hp[2];   // 3.5 VALU instructions per element (conv_f16c8.hip convert_store_x)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = dyr[I][j] * dys_stage;
      dbs[qh][j] = __builtin_fmaf(dyr[I][j], dbw, dbs[qh][j]);   // always; read only if do_db
    }
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp[j >> 1]) : "v"(v[j]), "v"(v[j + 1]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j]) : "v"(hp[j >> 1]), "v"(v[j]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[j + 1]) : "v"(hp[j >> 1]), "v"(v[j + 1]));
    }
    s16x2 lv = {0, 0};
    lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[0], lo[1], 0x1p-12f, false);
    lv = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lv, lo[2], lo[3], 0x1p-12f, true);
    int h8v = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
    h8v = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h8v, true);
    i32x2 hi;
    hi[0] = hp[0]; hi[1] = hp[1];
    *reinterpret_cast<i32x2*>(base + dc0 + qh * 8192 + ks * 1024) = hi;
    *reinterpret_cast<int*>(base + dc1 + qh * (2 * 64 * 32) + ks * 512) = __builtin_bit_cast(int, lv);   // image 0: dyl'
    *reinterpret_cast<int*>(base + dc1 + qh * (2 * 64 * 32) + ks * 512 + D8IMG) = h8v;                   // image 1: dyh'
  };
  // X by LDS-DMA.  Descriptor = the slab's 8 octet planes of image n; a lane's offset is  (octet x plane + pixel) x 32
  // + byte  or, for a halo pixel outside the image (or a pad column of the 40-pixel LDS row), 2^31: out of range, so
  // the hardware writes zeros.
  const unsigned OOR = 0x80000000u;
  const unsigned xspan = (unsigned)(8 * in_plane * 32);
  const int dma_hf = wave >> 1, dma_m0 = 5 * (wave & 1);      // this wave's ci-tile and pieces (of both images)
  // (lane-derived constants are RECOMPUTED at their use from a laundered lane id: kept in registers across the tile loop
  // they were spilled and reloaded through scratch, each reload behind an s_waitcnt vmcnt(0) that drained the DMAs)
  auto lane_now = [&]() __attribute__((always_inline)) {
    unsigned l = threadIdx.x & 63u;
    asm volatile("" : "+v"(l));
    return l;
  };
  // piece k (0..4) of this wave: 16 halo pixels x 4 octets x 16 bytes = 1 KiB of the f16 image (records' first halves,
  // 16-byte unit = octet ^ ((pixel >> 1) & 3)) and 1 KiB of the fp8 image (second halves, unit = octet).  An LDS-DMA
  // instruction costs ~100 issue cycles beside MFMAs: 10 per wave and tile, not the 25 a 4-byte-granule fp8 image took.
  auto dma_x = [&](const __amdgpu_buffer_rsrc_t srd, const TileC& tc, unsigned char* wb, auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    const int m = dma_m0 + k;                                  // uniform
    const unsigned ln = lane_now(), lu = ln & 3u, lp = ln >> 2;
    const unsigned pix = 16u * (unsigned)m + lp;
    const unsigned rr = (pix * 205u) >> 13, cc = pix - 40u * rr;   // pix / 40 for pix < 160
    const int vy = tc.oy0 - 1 + (int)rr, vx = tc.ox0 - 1 + (int)cc;
    const bool ok = (cc < (unsigned)XWV) & ((unsigned)vy < (unsigned)a.Hin) & ((unsigned)vx < (unsigned)a.Win);
    const unsigned pixoff = (unsigned)(vy * a.Win + vx);
    const unsigned v16 = ok ? ((lu ^ ((pix >> 1) & 3u)) * (unsigned)in_plane + pixoff) * 32u : OOR;
    const unsigned v8 = ok ? (lu * (unsigned)in_plane + pixoff) * 32u + 16u : OOR;
    const int soff = dma_hf * 4 * (int)in_plane * 32;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        srd, (__attribute__((address_space(3))) void*)(wb + OFF_X + dma_hf * (NXP * 64) + m * 1024), 16, (int)v16, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        srd, (__attribute__((address_space(3))) void*)(wb + OFF_X8 + dma_hf * (NXP * 64) + m * 1024), 16, (int)v8, soff, 0, 0);
  };
  auto x_srd = [&](const TileC& tc) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x) + ((size_t)tc.n * (a.Cin / 8) + cib * 8) * in_plane * 32), 0, xspan,
        0x00020000);
  };
  // staging work of tap slot s (0..26) of row 0 (issue) / row 1 (commit)
  auto issue_slot = [&](const __amdgpu_buffer_rsrc_t srd, const TileC& tc, unsigned char* buf, auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s < 10 && (s & 1) == 0) dma_x(srd, tc, buf, std::integral_constant<int, s / 2>{});
    if constexpr (s >= 12 && s < 20) issue_dy(tc, std::integral_constant<int, s - 12>{});
  };
  auto commit_slot = [&](unsigned char* nbase, auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s >= 6 && s < 14) commit_dy(nbase, std::integral_constant<int, s - 6>{});
  };

  // ------------------------------------------------------------------ one output row of the staged tile
  // slot = one tap of one of the row's three passes (fp8, f16 k-step 0, f16 k-step 1): 27 per row, 2 MFMAs each (the
  // wave's two co-tiles share the slot's B fragment).  With ONE wave on the SIMD nothing else covers an LDS latency,
  // and an f16 slot is only 64 MFMA cycles long: fragments are requested TWO slots ahead (a ring of three B
  // fragments; the A fragments of a pass two slots before its first slot); `work(slot)` is the slot's staging slice.
  i32x8 a8[2];            // fp8 pass: dyl' | dyh' of the wave's two co-tiles
  f16x8 ah[2][2];         // f16 k-step kk: dy_hi of the two co-tiles
  constexpr int AHEAD = 4, RING = AHEAD + 1;   // fragments are requested AHEAD slots before their MFMAs
  i32x8 bring[RING];      // fp8 B fragments of slots s .. s + AHEAD
  f16x8 bring16[RING];    // f16 B fragments
  int o8a = 0, o8b = 0;   // lane offsets of the fp8 pass, computed where its first fragment is requested
  int qo[6];              // lane offsets of the six f16 window starts (kx, pixel-row half), computed once per row
  // requests of slot s (0..26) of output row `row` of the buffer at `base`; s >= 27 wraps into row + 1 of the SAME buffer
  auto request = [&](const unsigned char* base, auto rowc, auto sc) __attribute__((always_inline)) {
    constexpr int s_ = decltype(sc)::value, row = decltype(rowc)::value + s_ / 27, s = s_ % 27;
    if constexpr (row < 2) {
      constexpr int kind = s / 9, t = s % 9, ky = t / 3, kx = t % 3;
      if constexpr (kind == 0) {
        if constexpr (t == 0) {
          const LaneC c = lane_consts();
          const int g_ = (int)(lane_now() >> 5);   // K block of the destination lanes = image of the source lanes
          o8a = OFF_D8 + g_ * D8IMG + ct * (64 * 32) + c.l8;      // dyl' (h = 0) | dyh' (h = 1)
          o8b = OFF_X8 + cih * (NXP * 64) + c.lx8;                // xh'  (h = 0) | xl'  (h = 1)
          a8[0] = tr8(base + o8a + row * (32 * 32));
          a8[1] = tr8(base + o8a + (64 * 32) + row * (32 * 32));
        }
        bring[(27 * decltype(rowc)::value + s_) % RING] = tr8<512>(base + o8b + ((row + ky) * XW + kx) * 64);
      } else {
        constexpr int kk = kind - 1, ks = 2 * row + kk;
        if constexpr (t == 0) {
          const LaneC c = lane_consts();
          const unsigned char* ap = base + ct * 4096 + ks * 16 * 64;
          ah[kk][0] = tr16(ap + c.sa0, ap + c.sa1);
          ah[kk][1] = tr16(ap + 4096 + c.sa0, ap + 4096 + c.sa1);
          if constexpr (kk == 0) {   // the window offsets of k-step 1 are those of k-step 0 + 16 pixels
            qo[0] = sx_of(c, 0);
            qo[1] = sx_of(c, 1) + 64;
            qo[2] = sx_of(c, 2) + 128;
            qo[3] = sx_of(c, 4) + 4 * 64;
            qo[4] = sx_of(c, 5) + 4 * 64 + 64;
            qo[5] = sx_of(c, 6) + 4 * 64 + 128;
          }
        }
        const unsigned char* xw = base + OFF_X + cih * (NXP * 64) + (row * XW + 16 * kk) * 64;   // uniform
        bring16[(27 * decltype(rowc)::value + s_) % RING] = tr16(xw + qo[kx] + ky * (XW * 64), xw + qo[3 + kx] + ky * (XW * 64));
      }
    }
  };
  // first: the row's slots 0 and 1 are requested here (tile start: nothing could be requested across the barrier)
  auto row_pass = [&](const unsigned char* base, auto rowc, auto&& work) __attribute__((always_inline)) {
    constexpr int row = decltype(rowc)::value;
    if constexpr (row == 0)
      static_for<0, AHEAD>([&](auto sc) __attribute__((always_inline)) { request(base, rowc, sc); });
    static_for<0, 27>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kind = s / 9, t = s % 9;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(EOVAE_WG4_ABL & 2)) request(base, rowc, std::integral_constant<int, s + AHEAD>{});
      if constexpr (!(EOVAE_WG4_ABL & 1)) work(sc);
      // 18 accumulator tuples, 16 fit the AGPRs -- and hipcc selects the AGPR form for every MFMA builtin, so with all
      // 18 behind builtins it moved 16 registers AGPR <-> VGPR around EVERY MFMA (s_nop 13 + 32 copies each: 3.5 x
      // slower than the 8-wave kernel).  The two tuples of tap 8 are therefore VGPR-form MFMAs in inline assembly:
      // operands come from LDS reads the compiler waits for, the scale registers are loop constants, and the results
      // are next read by the same instruction eight slots later (no wait states needed in between).
      if constexpr (kind == 0) {
        const i32x8 b8 = bring[(s + 27 * row) % RING];
        if constexpr (t < NT - 1) {
          acc[t][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8, acc[t][0], 1 /* A: e5m2 */, 1 /* B: e5m2 */,
                                                                     0, sc_a, 0, sc_b);
          acc[t][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8, acc[t][1], 1, 1, 0, sc_a, 0, sc_b);
        } else {
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1 blgp:1"
                       : "+v"(accv0) : "v"(a8[0]), "v"(b8), "v"(sc_av), "v"(sc_bv));
          // (the trailing s_nops cover both instructions' result latency: the compiler schedules nothing for an asm
          // statement's hazards and may copy or spill accv right behind it)
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1 blgp:1\n\ts_nop 15\n\ts_nop 7"
                       : "+v"(accv1) : "v"(a8[1]), "v"(b8), "v"(sc_av), "v"(sc_bv));
        }
      } else {
        const f16x8 bh = bring16[(s + 27 * row) % RING];
        if constexpr (t < NT - 1) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kind - 1][0], bh, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kind - 1][1], bh, acc[t][1], 0, 0, 0);
        } else {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accv0) : "v"(ah[kind - 1][0]), "v"(bh));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(accv1) : "v"(ah[kind - 1][1]), "v"(bh));
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, kind == 0 ? 4 : 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, kind == 0 ? 10 : 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    });
  };

  // ------------------------------------------------------------------ tile loop
  int tile = ysub;
  if (tile >= a.ntiles) return;   // uniform (split <= ntiles: does not happen)
  {
    const TileC tc = decode(tile);
    const __amdgpu_buffer_rsrc_t srd = x_srd(tc);
    static_for<0, 27>([&](auto sc) __attribute__((always_inline)) { issue_slot(srd, tc, smem0, sc); });
    static_for<0, 27>([&](auto sc) __attribute__((always_inline)) { commit_slot(smem0, sc); });
  }
  __syncthreads();
  // One tile: multiply out of `rb`, stage the tile behind it into `wb`.  The walk is made of PAIRS of tiles (one loop
  // back-edge, no exit in the middle: with two exits the VGPR accumulators went through scratch at every block
  // boundary); a walk of odd length ends with a dead tile -- the last tile staged again with its dY scaled by 0.
  const int cnt = (a.ntiles - ysub + a.split - 1) / a.split;
  int k = 0;
  auto tile_body = [&](unsigned char* rb, unsigned char* wb) __attribute__((always_inline)) {
    const bool live = k + 1 < cnt;                         // uniform
    const TileC tn = decode(ysub + (live ? k + 1 : cnt - 1) * a.split);
    dbw = live ? 1.f : 0.f;
    dys_stage = live ? dys : 0.f;
    const __amdgpu_buffer_rsrc_t srd = x_srd(tn);
    __builtin_amdgcn_sched_barrier(0);
    // row 0: the next tile's X DMAs and dY loads, one slice per tap slot; row 1: dY conversion and LDS writes
    row_pass(rb, std::integral_constant<int, 0>{}, [&](auto sc) __attribute__((always_inline)) { issue_slot(srd, tn, wb, sc); });
    row_pass(rb, std::integral_constant<int, 1>{}, [&](auto sc) __attribute__((always_inline)) { commit_slot(wb, sc); });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(EOVAE_WG4_ABL & 4)) __syncthreads();
    ++k;
  };
  for (int pr = 0; pr < (cnt + 1) / 2; ++pr) {
    tile_body(smem0, smem1);
    tile_body(smem1, smem0);
  }

  // ------------------------------------------------------------------ results
  if (do_db) {  // lanes 16 q .. 16 q + 15 hold the same 4 channels at 16 different pixels
#pragma unroll
    for (int qh = 0; qh < 2; ++qh)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = dbs[qh][j];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        const int co = cob * 128 + 64 * qh + (tid >> 4) * 4 + j;
        if ((tid & 15) == 0) atomicAdd(&a.db[co], v);
      }
  }
  const int ci = cib * 64 + cih * 32 + r;
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs' results are read below: well past their latency
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = cob * 128 + (ct + c) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const float v = t < NT - 1 ? acc[t < NT - 1 ? t : 0][c][q] : (c ? accv1[q] : accv0[q]);
        atomicAdd(&a.dwp[((size_t)t * a.Cout + co) * a.Cin + ci], v * dyinv);
      }
}

}  // namespace

// OFF by default (EOVAE_C8_WGRAD_W4=1 enables it where the shape allows): correct (tests/test_gpu_wgrad_w4.py), but
// on the real layer shapes it runs at 0.62 .. 0.78 of the 8-wave kernel's speed (profiles/r03f_wgrad_w4_ab.log).  The
// tile loop holds ~1 300 non-MFMA instructions per wave against 108 MFMAs (4 608 MFMA cycles): with one wave per SIMD
// every one of them must fit an MFMA gap of the SAME wave, and 1 300 x ~4.5 issue cycles do not (dY conversion 240,
// LDS-DMA issue and addressing 200, fragment reads 176, SGPR-spill lane moves 130, waits 86, dY loads + 64-bit
// addressing 90, tile decode 80 ...), while the 8-wave kernel spreads the same staging over two waves per SIMD.
#else   // not compiled in: the option reads as "not available"
bool eovae_c8_wgrad_w4_eligible(const WgCArgs&) { return false; }
void eovae_c8_wgrad_w4_launch(WgCArgs, hipStream_t) {}
#endif
#if EOVAE_PARKED
bool eovae_c8_wgrad_w4_eligible(const WgCArgs& a) {
  if (eovae_opt(EOVAE_OPT_C8_WGRAD_W4) != 1) return false;
  if (a.subpix != 0 || a.mode != MODE_NORMAL || !a.xpacked) return false;
  if (a.Cout % 128 || a.Cin % 64 || a.Wout % 32 || a.Hout % 2 || a.Hin != a.Hout || a.Win != a.Wout) return false;
  return a.split >= 1 && a.ntiles >= a.split;
}

void eovae_c8_wgrad_w4_launch(WgCArgs a, hipStream_t st) {
  const dim3 grid(a.ncib * a.ncob_ * a.split), blk(256);
  hipLaunchKernelGGL(wgrad3x3_f16c8_w4_kernel, grid, blk, 0, st, a);
}
#endif
