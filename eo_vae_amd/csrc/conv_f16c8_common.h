// Shared by the f16c8 3x3 convolution kernels (conv_f16c8.hip: 8 waves, 2 x 2 blocks per wave;
// conv_f16c8_w4.hip: 4 waves, one per SIMD, 4 x 4 blocks per wave): argument block, operand typedefs, scale helpers.
#pragma once
#include "eovae_common.h"

struct ConvCArgs {
  const float* x;
  const unsigned char* wp;   // f16c8-packed weights, C8_WTILE bytes per (tap, 32-ch chunk, 128-co block) + header
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
  int ntiles;                // tiles x co-blocks walked by the persistent grid
  int packed;                // x = f16c8 operand records (GroupNorm-apply, forward only)
  double* gn_sums;           // [N][gn_groups][2] += (sum, sum of squares) of the OUTPUT per (image, channel group), or null
  int gn_groups, gn_shift;   // channel groups of the GroupNorm that reads y; log2(channels per group)
  int stagger;               // conv_f16c8_w4.hip: start delay per block group, in units of 1024 shader cycles
  int skew;                  // conv_f16c8_w4.hip, two blocks per CU: start delay of a CU's second block, units of 4096 cycles
  unsigned* out_amax;        // float bits of max|y| (atomicMax into a pre-zeroed word), or null: the range pre-scale of the
                             // layer that reads y (stride-2 / x2-upsample / data-gradient inputs) without a pass over y
  const unsigned char* x6;   // packed input only: the records' MX-FP6 plane (groupnorm.hip gn_apply_mx_kernel), or null: the
                             // correction product then runs on the records' e5m2 parts (fp8 rate)
  int mx;                    // correction product on block-scaled FP6 operands (below): set by the launcher
};

// weight-gradient argument block (conv_f16c8.hip wgrad3x3_f16c8_kernel, conv_f16c8_wgrad_w4.hip)
struct WgCArgs {
  const float* x;
  const float* dy;
  float* db;   // [Cout] bias gradient (sum of dy over n, pixels), pre-zeroed, or null
  float* dwp;  // [9][Cout][Cin] fp32, pre-zeroed
  const unsigned* dyamax;   // float bits of max|dy| (device) or null
  const unsigned* xamax;    // float bits of max|x| (device) for an fp32 x that may sit outside f16's range, or null
  int N, Cin, Hin, Win, Cout, Hout, Wout;
  int mode;
  int tiles_x, tiles_y, ntiles, ncib;
  int subpix, creal, cpb;   // as WgBArgs (conv_bf16x3.hip)
  int ncob_, split, xcd_remap;
  int xpacked;              // x = f16c8 operand records (GroupNorm-apply)
  float* part;              // plain layers: per-block partial sums [slab][split][9][128][64] instead of atomics into dwp (or null)
  int nseg, units;          // conv_f16c8_wgrad_rows.hip: row segments per strip, units (image x strip x segment) of the walk
};

// Packed weight tile of one (tap, 32-channel chunk, 128-row block):
//   [0, 8 KiB)        f16   [octet 0..3][128 m][8 k]
//   [8 KiB, 16 KiB)   e4m3  [wh' | wl'][half 0..1][128 m][16 k]                  (per-tensor power-of-two scale, header)
//   [16 KiB, 24 KiB)  MX-FP6 [wh6 | wl6][piece 0..1][128 m][16 B]               (round 5)
// MX-FP6 operand of one (row, 32-K block): 32 x e2m3 = 6 dwords (v_cvt_scalef32_2xpk16_fp6_f32 of k 0..15 | 16..31) and
// the block's E8M0 scale byte = max(biased exponent of the block amax - 2, 1) (OCP MX: amax lands in [4, 8), 7.5 saturates),
// stored as 32 bytes = piece 0 {d0 d1 d2 d3} | piece 1 {d4 d5 scale 0}: a lane's fragment is the two pieces, i.e. the same
// two ds_read_b128 as the 32 fp8 bytes it replaces; the MFMA reads dwords 0..5 as data and takes dword 6 as its scale
// operand (byte 0).  The FP6 x FP6 product runs ~1.6 x the fp8 rate (tools/probe_mxfp6.hip), and the per-block scales
// make it MORE accurate than the per-tensor-scaled fp8 operands.
constexpr int C8_WTILE = 24576, C8_WOFF8 = 8192, C8_WOFF6 = 16384;

namespace {

enum { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_ZINS = 3 };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));


// x of the lane whose index within its quad is permuted by CTRL (0xB1: lane ^ 1, 0x4E: lane ^ 2)
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// power of two 2^k as a float, k clamped to the normal range
__device__ __forceinline__ float pow2f(int k) {
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}
// E8M0 scale byte of an MX-FP6 (e2m3) block whose largest magnitude is amax (>= 0, finite)
__device__ __forceinline__ unsigned mx6_scale_byte(float amax) {
  const int be = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xff) - 2;
  return (unsigned)(be < 1 ? 1 : be);
}
// 32 values (k = 0..31) -> the two 16-byte pieces of the block's MX-FP6 operand (layout above)
__device__ __forceinline__ void mx6_quant(const float (&v)[32], float amax, i32x4& p0, i32x4& p1) {
  typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
  f32x16 a, b;
#pragma unroll
  for (int j = 0; j < 16; ++j) { a[j] = v[j]; b[j] = v[16 + j]; }
  const unsigned sb = mx6_scale_byte(amax);
  const float div = __builtin_bit_cast(float, sb << 23);   // the instruction divides by its scale operand
  // Inline assembly with an EARLY-CLOBBER result: ROCm 7.2's hipcc lets the builtin's 6 result registers overlap its 16-
  // register sources, and the hardware writes result dwords before it has read every source (tools/probe_cvt_fp6.hip:
  // wrong fields from the 13th on).  The 32 fields come out interleaved (a[i] -> field 2i, b[i] -> field 2i + 1): every
  // operand of a product goes through this function, so the k order inside a block is the same on both sides.
  u32x6 q;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a), "v"(b), "v"(div));
  p0[0] = (int)q[0]; p0[1] = (int)q[1]; p0[2] = (int)q[2]; p0[3] = (int)q[3];
  p1[0] = (int)q[4]; p1[1] = (int)q[5]; p1[2] = (int)sb; p1[3] = 0;
}
// 16-byte buffer store with its own wait states (round 6).  MEASURED on gfx950 (tools/probe_store_war.hip,
// profiles/r06b_probe_store_war.log): a VALU write of a buffer_store_dwordx4's DATA registers issued directly behind the store
// reaches memory in lanes 12..15 of every 16-lane row (a v_pk_fma_f32's high dword in ~5 % of the stores, a v_mov_b32 in
// 0.3 %) -- the store reads its data over several cycles.  hipcc 7.2's hazard recogniser inserts its one wait state only for
// stores WITHOUT a register soffset, and the compiler-scheduled epilogues here use one: conv_f16c8_q16up.hip's first
// epilogue stored the next pixel tile's value in exactly those lanes, and conv_f16c8_q16.hip's "(iv)" (wrong OUTPUT of the
// first co tile with a butterfly behind the stores) has the same signature.  The statement below therefore carries the wait
// states itself: s_nop 4 in front (the descriptor / soffset SGPRs may have come from v_readlane spill restores, which need 5
// wait states in front of a VMEM read that hipcc does not count for inline assembly), s_nop 3 behind.  ~10 cycles per store
// against the ~80 a CU needs to drain its 1 KB.
#ifndef EOVAE_GUARD_STORES
#define EOVAE_GUARD_STORES 1   // 0 (A/B builds only, tools/build_variant.sh): the plain builtin store
#endif
__device__ __forceinline__ void store128_guarded(const f32x4& d, const __amdgpu_buffer_rsrc_t& srd, int voff, int soff) {
  if (EOVAE_GUARD_STORES)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" :: "v"(d), "v"(voff), "s"(srd), "s"(soff) : "memory");
  else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, d), srd, voff, soff, 0);
}
// Buffer loads as asm statements (issue order = program order; the caller counts the waits and passes the result through
// landed() behind its s_waitcnt, so that no consumer is scheduled in front of the wait).  s_nop 4: as above, the SGPR operands
// may come from v_readlane.
__device__ __forceinline__ void load128_asm(f32x4& d, const __amdgpu_buffer_rsrc_t& srd, int voff, int soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int OFF>   // (an immediate offset: a register one per call is loop-invariant to the compiler, hoisted and spilled)
__device__ __forceinline__ void load32_asm(float& d, const __amdgpu_buffer_rsrc_t& srd, int voff) {
  static_assert(OFF >= 0 && OFF < 4096, "12-bit instruction offset");
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=&v"(d) : "v"(voff), "s"(srd), "n"(OFF) : "memory");
}
// d = v of lane (byte address / 4): crossbar only, counted by lgkmcnt; an asm statement so that it is ISSUED where it stands
__device__ __forceinline__ void bpermute_asm(float& d, int lane_bytes, float v) {
  asm volatile("ds_bpermute_b32 %0, %1, %2" : "=&v"(d) : "v"(lane_bytes), "v"(v));
}
__device__ __forceinline__ void landed(f32x4& d) { asm volatile("" : "+v"(d)); }
__device__ __forceinline__ void landed(float& d) { asm volatile("" : "+v"(d)); }
#ifndef EOVAE_EPI_QUADS
#define EOVAE_EPI_QUADS 1   // 0 (A/B builds only): the epilogues store from the MFMA layout (four lines per lane quad)
#endif
// The lane number, re-derived where it is called: values computed from it are NOT loop-invariant to the compiler, which
// otherwise hoists an epilogue's per-lane addresses out of the tile loop, spills them across the main loop (all 512 registers
// are taken there) and reloads them between the epilogue's stores -- behind an s_waitcnt vmcnt(0) that drains the stores.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// binary exponent of a positive float given as bits (0 for zero / subnormal input)
__device__ __forceinline__ int expo_of_bits(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xff);
  return e == 0 ? -127 : e - 127;
}

}  // namespace

// conv_f16c8_w4.hip: launches the one-wave-per-SIMD kernel for an argument block prepared by eovae_conv3x3_f16c8_fwd
// (plain 3x3 form, full 16 x 32 tiles only); returns false when the layer does not qualify.
bool eovae_c8_w4_eligible(const ConvCArgs& a, int subpix, int n_cu);
void eovae_c8_w4_launch(ConvCArgs a, int n_cu, hipStream_t st);
int eovae_c8_one_wave_kind(ConvCArgs a, int n_cu);
// conv_f16c8_q16.hip: the same tile walk on 16 x 16 MFMAs with an all-DMA feed, for MX operand records (library option c8_q16)
bool eovae_c8_q16_eligible(const ConvCArgs& a);
void eovae_c8_q16_launch(const ConvCArgs& a, int n_cu, hipStream_t st);

// conv_f16c8_wgrad_rows.hip: row-streaming weight gradient (plain 3x3 form, record input, partial-tile output); the
// launch fills a.nseg / a.units and may lower a.split (blocks per slab) to the number of units
bool eovae_c8_wgrad_rows_eligible(const WgCArgs& a);
void eovae_c8_wgrad_rows_launch(WgCArgs& a, hipStream_t st);

// conv_f16c8_wgrad_w4.hip: one-wave-per-SIMD weight gradient (plain 3x3 form, full tiles)
bool eovae_c8_wgrad_w4_eligible(const WgCArgs& a);
void eovae_c8_wgrad_w4_launch(WgCArgs a, hipStream_t st);
