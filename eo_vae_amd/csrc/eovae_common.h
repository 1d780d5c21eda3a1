// Shared helpers for the eo_vae_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define EOVAE_OK 0
#define EOVAE_ERR_ARG (-1)
#define EOVAE_ERR_LAUNCH (-2)
#define EOVAE_ERR_UNSUPPORTED (-3)

// run-time switches (api.cpp): value of option `id`, initialised once from its environment variable, changed only by
// eovae_set_option
enum {
  EOVAE_OPT_WGRAD_ROWS = 0,   // row-streaming weight gradient (1)
  EOVAE_OPT_C8_W4,            // one-wave-per-SIMD conv kernel: -1 when it pays (default), 0 never, 1 whenever the shape allows
  EOVAE_OPT_C8_W4_NI,         // rows per wave of that kernel: 4 (default) or 2 (experiment, record inputs only)
  EOVAE_OPT_C8_W4_STAGGER,    // experiment: start delay per block group, units of 1024 cycles (0)
  EOVAE_OPT_C8_W4_SKEW,       // experiment (NI = 2): start delay of a CU's second block, units of 4096 cycles; -1 = half a tile
  EOVAE_OPT_C8_WGRAD_W4,      // parked one-wave weight gradient (0)
  EOVAE_OPT_C8_PACK_GATHER,   // per-layer weight packs by the rounds-2/3 per-element gather kernel (0; tests: reference of the LDS-staged pack)
  EOVAE_OPT_C8_MX,            // correction product on block-scaled FP6 operands where the kernel and its inputs allow it (1); 0: fp8 everywhere
  EOVAE_OPT_C8_WGRAD_CORR,    // row-streaming weight gradients WITH the fp8 correction product (the r4 arithmetic): 1; 0 (default): f16 x f16 alone
  EOVAE_OPT_C8_Q16,           // 16 x 16 MFMA shape in the kernels that have that form (1): the chip holds a higher clock on it; 0: 32 x 32
  EOVAE_OPT_C8_DGRAD_MX,      // fp32-input launches of the one-wave kernel (the data gradients) with the MX-FP6 correction product (1); 0: fp8
  EOVAE_OPT_C8_UP_Q16,        // x2-upsample forward / data gradient on MX records (conv_f16c8_q16up.hip, conv_f16c8_q16upg.hip): 1 where it pays, 2 wherever the shape allows, 0: 8-wave sub-pixel modes
  EOVAE_OPT_COUNT
};
int eovae_opt(int id);

// Build switch (round 6): measured-and-parked kernels -- the one-wave weight gradient
// (conv_f16c8_wgrad_w4.hip, 0.77-0.87 x), the 16 x 16 kernel's fp32-input path (conv_f16c8_q16.hip FP32IN, 0.88-0.92 x) and the
// two-rows-per-wave instantiations of conv_f16c8_w4.hip (round 3's two-blocks-per-CU experiment) -- are compiled only with
// -DEOVAE_PARKED=1 (EOVAE_BUILD_PARKED=1 python -m eo_vae_amd.build --force).  Their options then read as "not available":
// the launchers fall back to the default kernels, eovae_build_flags() bit 0 tells tests whether to run their cases.
#ifndef EOVAE_PARKED
#define EOVAE_PARKED 0
#endif

// thread-local last-error text, exported through eovae_last_error()
void eovae_set_error(const char* fmt, ...);

#define EOVAE_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      eovae_set_error(__VA_ARGS__);           \
      return EOVAE_ERR_ARG;                   \
    }                                         \
  } while (0)

#define EOVAE_CHECK_LAUNCH(name)                                               \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      eovae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return EOVAE_ERR_LAUNCH;                                                 \
    }                                                                          \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// wave64 butterfly sums
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// (round 6) fp64: four DPP exchanges inside the 16-lane rows (lane ^ 1, lane ^ 2, mirror of 8, mirror of 16 -- every lane of
// a row then holds the row's sum), the four row sums through v_readlane.  The __shfl_xor form was twelve ds_bpermute_b32 in six
// DEPENDENT LDS round trips (~700 cycles of latency per call; the one-pass GroupNorm backward makes three such calls in a
// row per slab).  Same value in every lane, fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_d(double x, int l) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_move_d<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_move_d<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_move_d<0x141>(v);   // row_half_mirror: the other quad of the octet
  v += dpp_move_d<0x140>(v);   // row_mirror: the other octet of the row
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
