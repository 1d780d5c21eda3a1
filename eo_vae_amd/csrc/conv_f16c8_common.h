// Shared by the f16c8 3x3 convolution kernels (conv_f16c8.hip: 8 waves, 2 x 2 blocks per wave;
// conv_f16c8_w4.hip: 4 waves, one per SIMD, 4 x 4 blocks per wave): argument block, operand typedefs, scale helpers.
#pragma once
#include "eovae_common.h"

struct ConvCArgs {
  const float* x;
  const unsigned char* wp;   // f16c8-packed weights, 16 KiB per (tap, 32-ch chunk, 128-co block) + header
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

// conv_f16c8_wgrad_rows.hip: row-streaming weight gradient (plain 3x3 form, record input, partial-tile output); the
// launch fills a.nseg / a.units and may lower a.split (blocks per slab) to the number of units
bool eovae_c8_wgrad_rows_eligible(const WgCArgs& a);
void eovae_c8_wgrad_rows_launch(WgCArgs& a, hipStream_t st);

// conv_f16c8_wgrad_w4.hip: one-wave-per-SIMD weight gradient (plain 3x3 form, full tiles)
bool eovae_c8_wgrad_w4_eligible(const WgCArgs& a);
void eovae_c8_wgrad_w4_launch(WgCArgs a, hipStream_t st);
