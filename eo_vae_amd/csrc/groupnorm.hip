// GroupNorm(32 groups, eps 1e-6, affine) fused with the swish that follows it in the
// reference (layers.py:61-63,78-80,91-92,106-108; model.py:159-161,193-194,295-297,349-350):
//     y = silu( (x - mean_g) * rstd_g * gamma_c + beta_c )        (silu optional: AttnBlock norm)
// NCHW fp32: one (n, group) is a contiguous run of cpg*HW floats (up to 1 MB at 4 x 256^2), so
// statistics are a pure streaming reduction: float4 loads, per-thread fp32 partials, wave64
// shuffle trees, cross-wave/cross-block combine in double.  Memory-bound (HBM roofline).
#include "eovae_common.h"
#include "conv_f16c8_common.h"   // MX-FP6 operand layout (mx6_quant)
#include <stdlib.h>
#include <unistd.h>
#include <chrono>
#include <mutex>
#include <random>

namespace {

constexpr int GN_SPLIT_ELEMS = 16384;  // elements per stats block
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }

// stage 1: partial (sum, sumsq) per (n,g) slice -> part[ng][split][2] (double)
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x,
                                                         double* __restrict__ part, size_t gsize,
                                                         int nsplit) {
  const int ng = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const size_t per = (gsize + nsplit - 1) / nsplit;
  const size_t beg = (size_t)sp * per;
  size_t end = beg + per;
  if (end > gsize) end = gsize;
  const float* p = x + (size_t)ng * gsize;
  float s = 0.f, q = 0.f;
  // gsize is a multiple of 4 whenever HW is; fall back to scalar otherwise
  if ((gsize & 3) == 0 && (per & 3) == 0) {
    for (size_t i = beg + 4 * threadIdx.x; i < end; i += 1024) {
      float4 v = *reinterpret_cast<const float4*>(p + i);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
      float v = p[i];
      s += v;
      q += v * v;
    }
  }
  __shared__ double red[8];
  double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = ds; red[4 + w] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)ng * nsplit + sp) * 2 + 0] = red[0] + red[1] + red[2] + red[3];
    part[((size_t)ng * nsplit + sp) * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// stage 2: combine partials -> stats[ng] = (mean, rstd)
__global__ void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats,
                                   int NG, int nsplit, double inv_count, float eps) {
  int ng = blockIdx.x * blockDim.x + threadIdx.x;
  if (ng >= NG) return;
  double s = 0, q = 0;
  for (int i = 0; i < nsplit; ++i) {
    s += part[((size_t)ng * nsplit + i) * 2];
    q += part[((size_t)ng * nsplit + i) * 2 + 1];
  }
  double mean = s * inv_count;
  double var = q * inv_count - mean * mean;
  if (var < 0) var = 0;
  stats[2 * ng] = (float)mean;
  stats[2 * ng + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// stage 3: normalise + affine (+ silu); grid.x = N*C planes x chunks
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ stats,
                                                       float* __restrict__ y, int C, int HW, int cpg,
                                                       int chunks) {
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = plane % C, n = plane / C;
  const int ng = n * (C / cpg) + c / cpg;
  const float mean = stats[2 * ng], rstd = stats[2 * ng + 1];
  const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
  const size_t base = (size_t)plane * HW;
  const int per = (HW + chunks - 1) / chunks;
  const int beg = chunk * per;
  int end = beg + per;
  if (end > HW) end = HW;
  if ((HW & 3) == 0 && (per & 3) == 0) {
    for (int i = beg + 4 * threadIdx.x; i < end; i += 1024) {
      float4 v = *reinterpret_cast<const float4*>(x + base + i);
      float4 o;
      o.x = v.x * sc + sh; o.y = v.y * sc + sh; o.z = v.z * sc + sh; o.w = v.w * sc + sh;
      if (SILU) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
      *reinterpret_cast<float4*>(y + base + i) = o;
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += 256) {
      float o = x[base + i] * sc + sh;
      y[base + i] = SILU ? silu_f(o) : o;
    }
  }
}

// stage 3, f16c8 operand form: normalise + affine + silu and write the result as the OPERAND RECORDS the f16c8
// convolution kernels multiply with (conv_f16c8.hip), instead of fp32:
//   pa[n][C/8][HW] x 32 B = { 8 x f16 y | 8 x e5m2 (y - f16(y)) * 2^12 | 8 x e5m2 y }      (channels 8g .. 8g+7)
// Same 4 B per element as the fp32 tensor it replaces.  The consumers (3x3 forward convolution, weight gradient)
// then copy records into LDS with no conversion work -- the conversion is done ONCE here, in a kernel that is
// HBM-bound anyway, instead of once per 128-co block of the forward and once per block of the weight gradient.
// One thread = one pixel x 8 channels; grid = N * C/8 * ceil(HW / 256).
// FROM_SUMS: the statistics arrive as the producing convolution's (sum, sum of squares) per (image, group) in double
// (conv epilogue, gn_sums) and are finished HERE -- the arithmetic of gn_finalize_kernel, once per block for the <= 8
// groups its channel octet touches -- instead of by a 5 us launch in front of every GroupNorm (52 per step); the
// block of pixel chunk 0 also leaves (mean, rstd) in `stats` for the backward pass.
template <bool SILU, bool FROM_SUMS>
__global__ __launch_bounds__(256) void gn_apply_packed_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ stats,
                                                              const double* __restrict__ sums, double inv_count, float eps,
                                                              int* __restrict__ pa, int C, int HW, int cpg,
                                                              int pblocks) {
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const int pb = blockIdx.x % pblocks, og = blockIdx.x / pblocks;   // og = n * C/8 + octet
  const int oct = og % (C / 8), n = og / (C / 8);
  const int p = pb * 256 + threadIdx.x;
  __shared__ float st_s[8][2];
  if constexpr (FROM_SUMS) {
    if (threadIdx.x < 8) {
      const int c = oct * 8 + threadIdx.x;
      const int ng = n * (C / cpg) + c / cpg;
      const double s_ = sums[(size_t)ng * 2], q_ = sums[(size_t)ng * 2 + 1];
      const double mean = s_ * inv_count;
      double var = q_ * inv_count - mean * mean;
      if (var < 0) var = 0;
      const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
      st_s[threadIdx.x][0] = m;
      st_s[threadIdx.x][1] = r;
      if (pb == 0 && c % cpg == 0) {
        stats[2 * ng] = m;
        stats[2 * ng + 1] = r;
      }
    }
    __syncthreads();
  }
  if (p >= HW) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = oct * 8 + j;
    const int ng = n * (C / cpg) + c / cpg;
    const float mean = FROM_SUMS ? st_s[j][0] : stats[2 * ng], rstd = FROM_SUMS ? st_s[j][1] : stats[2 * ng + 1];
    const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
    float o = x[((size_t)n * C + c) * HW + p] * sc + sh;
    v[j] = SILU ? silu_f(o) : o;
  }
  int pk[4];
  float lo[8];
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
  i32x4v a0, a1;
  a0[0] = pk[0]; a0[1] = pk[1]; a0[2] = pk[2]; a0[3] = pk[3];
  a1[0] = __builtin_bit_cast(int, l0); a1[1] = __builtin_bit_cast(int, l1); a1[2] = h0; a1[3] = h1;
  i32x4v* rec = reinterpret_cast<i32x4v*>(pa + ((size_t)og * HW + p) * 8);
  rec[0] = a0;
  rec[1] = a1;
}

// stage 3, MX operand records (round 5, "format 2"): normalise + affine + silu, written as the operands of a plain 3x3 layer
// whose forward runs the MX-FP6 correction product and whose weight gradient multiplies f16 x f16 alone
// (conv_f16c8_common.h, conv_f16c8_wgrad_rows.hip) -- still 4 B per element, in two planes inside pa:
//   F  pa[n][C/8][HW] x 16 B              : 8 x f16 y                          (channels 8 g .. 8 g + 7)
//   S  pa + N*C*HW*2 B : [n][C/32][piece 0..3][HW] x 16 B : the block-scaled e2m3 operands of channels 32 b .. 32 b + 31 of a
//      pixel -- pieces 0, 1 = (y - f16(y)) ("lo", pairs with the weights' leading part), pieces 2, 3 = y ("hi", pairs with
//      the weights' residual part); a block = 6 dwords of 32 x 6 bits + its E8M0 scale byte in dword 6.
// The e5m2 parts of format 1 are gone: nothing reads them any more (the weight gradient needs the f16 parts only).
// The scale of a block is shared by the 32 channels of ONE pixel (the K axis of the forward convolution), so a thread owns
// a pixel x 32 channels: 32 coalesced dword loads, 4 + 4 sixteen-byte stores; grid = N * C/32 * ceil(HW / 256).
template <bool SILU, bool FROM_SUMS>
__global__ __launch_bounds__(256) void gn_apply_mx_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ stats,
                                                          const double* __restrict__ sums, double inv_count, float eps,
                                                          int* __restrict__ pa, int N, int C, int HW, int cpg, int pblocks) {
  const int pb = blockIdx.x % pblocks, bg = blockIdx.x / pblocks;   // bg = n * C/32 + block
  const int blk = bg % (C / 32), n = bg / (C / 32);
  const int p = pb * 256 + threadIdx.x;
  __shared__ float st_s[32][2];   // per channel of the block: scale, shift of y = x * sc + sh
  if (threadIdx.x < 32) {
    const int c = blk * 32 + threadIdx.x;
    const int ng = n * (C / cpg) + c / cpg;
    float m, r;
    if constexpr (FROM_SUMS) {
      const double s_ = sums[(size_t)ng * 2], q_ = sums[(size_t)ng * 2 + 1];
      const double mean = s_ * inv_count;
      double var = q_ * inv_count - mean * mean;
      if (var < 0) var = 0;
      m = (float)mean;
      r = (float)(1.0 / sqrt(var + (double)eps));
      if (pb == 0 && c % cpg == 0) {
        stats[2 * ng] = m;
        stats[2 * ng + 1] = r;
      }
    } else {
      m = stats[2 * ng];
      r = stats[2 * ng + 1];
    }
    const float sc = r * gamma[c];
    st_s[threadIdx.x][0] = sc;
    st_s[threadIdx.x][1] = beta[c] - m * sc;
  }
  __syncthreads();
  if (p >= HW) return;
  float v[32], lo[32];
  const float* xp = x + ((size_t)n * C + blk * 32) * HW + p;
#pragma unroll
  for (int j = 0; j < 32; ++j) v[j] = xp[(size_t)j * HW];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float o = v[j] * st_s[j][0] + st_s[j][1];
    v[j] = SILU ? silu_f(o) : o;
  }
  i32x4* fpl = reinterpret_cast<i32x4*>(pa) + ((size_t)n * (C / 8) + blk * 4) * HW + p;
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
  i32x4* dst = reinterpret_cast<i32x4*>(reinterpret_cast<char*>(pa) + (size_t)N * C * HW * 2) + ((size_t)bg * 4) * HW + p;
  mx6_quant(lo, al, q0, q1);
  dst[0] = q0;
  dst[(size_t)HW] = q1;
  mx6_quant(v, ah, q0, q1);
  dst[(size_t)2 * HW] = q0;
  dst[(size_t)3 * HW] = q1;
}

// backward pass 1: per (n,c) plane  s1 = sum dz, s2 = sum dz * xhat   (dz = dy * silu'(z))
template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ stats,
                                                            double* __restrict__ sums, int C, int HW,
                                                            int cpg) {
  const int plane = blockIdx.x;
  const int c = plane % C, n = plane / C;
  const int ng = n * (C / cpg) + c / cpg;
  const float mean = stats[2 * ng], rstd = stats[2 * ng + 1];
  const float g = gamma[c], b = beta[c];
  const size_t base = (size_t)plane * HW;
  float s1 = 0.f, s2 = 0.f;
  auto term = [&](float xv, float dyv) {
    float xh = (xv - mean) * rstd;
    float dz = dyv;
    if (SILU) {
      float z = xh * g + b;
      float sg = 1.f / (1.f + __expf(-z));
      dz *= sg * (1.f + z * (1.f - sg));
    }
    s1 += dz;
    s2 += dz * xh;
  };
  if ((HW & 3) == 0) {   // 16-byte loads: 4x fewer memory instructions for the same bytes in flight
    for (int i = 4 * threadIdx.x; i < HW; i += 1024) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
      const float4 dv = *reinterpret_cast<const float4*>(dy + base + i);
      term(xv.x, dv.x); term(xv.y, dv.y); term(xv.z, dv.z); term(xv.w, dv.w);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) term(x[base + i], dy[base + i]);
  }
  __shared__ double red[8];
  double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = d1; red[4 + w] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[2 * (size_t)plane] = red[0] + red[1] + red[2] + red[3];
    sums[2 * (size_t)plane + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// backward pass 2a: dgamma/dbeta[c] = sum_n sums[n][c]; group means m1,m2 per (n,g)
__global__ void gn_bwd_finalize_kernel(const double* __restrict__ sums,
                                       const float* __restrict__ gamma, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ gm, int N,
                                       int C, int cpg, double inv_count, unsigned* __restrict__ amax_zero) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && amax_zero) *amax_zero = 0u;   // the fold kernel's atomicMax target (saves a memset launch)
  if (t < C) {
    double a = 0, b = 0;
    for (int n = 0; n < N; ++n) {
      a += sums[2 * ((size_t)n * C + t)];
      b += sums[2 * ((size_t)n * C + t) + 1];
    }
    dbeta[t] = (float)a;
    dgamma[t] = (float)b;
  }
  int G = C / cpg;
  if (t < N * G) {
    int n = t / G, g = t % G;
    double m1 = 0, m2 = 0;
    for (int j = 0; j < cpg; ++j) {
      int c = g * cpg + j;
      m1 += sums[2 * ((size_t)n * C + c)] * gamma[c];
      m2 += sums[2 * ((size_t)n * C + c) + 1] * gamma[c];
    }
    gm[2 * t] = (float)(m1 * inv_count);
    gm[2 * t + 1] = (float)(m2 * inv_count);
  }
}

// backward pass 2b: dx = rstd * (dz*gamma - m1 - xhat*m2) (+ dx_add)
template <bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ stats, const float* __restrict__ gm,
    const float* __restrict__ dx_add, float* __restrict__ dx, int C, int HW, int cpg, int chunks,
    float* __restrict__ amax_part) {
  const int plane = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  float amax = 0.f;   // max |dx| of this wave (reported when amax_part is given)
  const int c = plane % C, n = plane / C;
  const int ng = n * (C / cpg) + c / cpg;
  const float mean = stats[2 * ng], rstd = stats[2 * ng + 1];
  const float g = gamma[c], b = beta[c];
  const float m1 = gm[2 * ng], m2 = gm[2 * ng + 1];
  const size_t base = (size_t)plane * HW;
  const int per = (HW + chunks - 1) / chunks;
  const int beg = chunk * per;
  int end = beg + per;
  if (end > HW) end = HW;
  auto elem = [&](float xv, float dyv, float addv) {
    float xh = (xv - mean) * rstd;
    float dz = dyv;
    if (SILU) {
      float z = xh * g + b;
      float sg = 1.f / (1.f + __expf(-z));
      dz *= sg * (1.f + z * (1.f - sg));
    }
    return rstd * (dz * g - m1 - xh * m2) + addv;
  };
  if ((HW & 3) == 0 && (per & 3) == 0) {
    if (dx_add) {   // uniform: two straight-line loops, no load under a per-element condition
      for (int i = beg + 4 * threadIdx.x; i < end; i += 1024) {
        const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
        const float4 dv = *reinterpret_cast<const float4*>(dy + base + i);
        const float4 av = *reinterpret_cast<const float4*>(dx_add + base + i);
        float4 o;
        o.x = elem(xv.x, dv.x, av.x); o.y = elem(xv.y, dv.y, av.y);
        o.z = elem(xv.z, dv.z, av.z); o.w = elem(xv.w, dv.w, av.w);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(dx + base + i) = o;
      }
    } else {
      for (int i = beg + 4 * threadIdx.x; i < end; i += 1024) {
        const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
        const float4 dv = *reinterpret_cast<const float4*>(dy + base + i);
        float4 o;
        o.x = elem(xv.x, dv.x, 0.f); o.y = elem(xv.y, dv.y, 0.f);
        o.z = elem(xv.z, dv.z, 0.f); o.w = elem(xv.w, dv.w, 0.f);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(dx + base + i) = o;
      }
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += 256) {
      const float o = elem(x[base + i], dy[base + i], dx_add ? dx_add[base + i] : 0.f);
      amax = fmaxf(amax, fabsf(o));
      dx[base + i] = o;
    }
  }
  // max |dx| for the f16c8 data-gradient pre-scale (conv_f16c8.hip): one plain store per wave into a partial
  // array that amax_fold_kernel folds afterwards.  (Atomics on one word cost 0.65 ms per launch when every wave
  // issued one, and a read-then-maybe-atomic kept every wave resident behind its own stores: +37 %.)
  if (amax_part) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) amax_part[blockIdx.x * 4 + (threadIdx.x >> 6)] = amax;
  }
}

// out_bits = float bits of max(part[0..n)): 16 blocks, one atomic per wave (out_bits pre-zeroed)
__global__ __launch_bounds__(256) void amax_fold_kernel(const float* __restrict__ part, int n,
                                                        unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, part[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < 3.0e38f) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));
}

// ------------------------------------------------------------------------------------------------------------------
// backward, ONE pass over the activation (12 B per element instead of the 20 of reduce + apply; 16 instead of 24 with
// a skip gradient): a persistent grid of 256-thread workgroups, two to four per CU by slice size.  A TEAM of T
// workgroups owns one (image, group) slab at a time; each member loads its slice of x and dy ONCE into registers
// (V float4 of each per thread), publishes its partial sums (s1 = sum dz, s2 = sum dz * xhat), waits until the whole
// team has published, folds the team's partials into the group means m1, m2 and writes dx from the registers.  The
// slab is never re-read, so nothing depends on what a cache still holds.
//
// Hand-off between workgroups (MI355X_MICROARCH.md, cross-workgroup visibility): a member's contribution to the group
// means travels as ONE 16-byte granule {sum s1 gamma, sum s2 gamma, 64-bit token} stored and polled with sc0 sc1 --
// the granule validates itself, so no fence, no flag and no counter to reset.  The token is unique per call, per
// process and per graph replay (gnf_token below): stale granules never match.
// Progress: logical workgroup ids are tickets taken in order of ARRIVAL, so the workgroups that are resident always
// hold the lowest ids, i.e. whole teams plus at most one partial team; a whole team needs nobody else, finishes and
// frees its slots.  (grid <= workgroups-per-CU x CUs, so normally everything is resident anyway; the tickets cover a GPU
// shared with another process or with RCCL's kernels.)  The wait is bounded all the same (two seconds of s_memrealtime, gnf_timed_out):
// on a time-out the workgroup records the token in `err`, and the finalize kernel poisons dgamma / dbeta with NaN.
// The per-channel partials (for dgamma / dbeta) are plain stores, read by the finalize kernel after the launch.
// Two forms: gn_bwd_fused_kernel (one slab per workgroup in flight; A/B reference, and the form for slabs whose team
// would not fit the other's grid) and gn_bwd_fused_pipe_kernel (software-pipelined over slabs; the default).
#ifndef EOVAE_GN_NOSYNC
#define EOVAE_GN_NOSYNC 0   // 1: timing-only build that skips the team wait (WRONG results; tools/build_variant.sh)
#endif
constexpr int GNF_NT = 256;             // threads per workgroup
constexpr int GNF_UNIT = GNF_NT * 4;    // elements one float4 per thread covers: the unit of the per-channel partials
// resident workgroups per CU by slice size (register budget: V = 8 needs ~220 VGPRs, V = 4 ~160 in the pipelined form,
// V <= 2 ~100)
constexpr int gnf_wg_per_cu(int V) { return V >= 8 ? 2 : V == 4 ? 3 : 4; }
struct GnBwdFArgs {
  const float *x, *dy, *gamma, *beta, *stats, *dx_add;
  float* dx;
  i32x4* team;             // [N*G][T] member granules
  float2* part;            // [N*G][T*V] (s1, s2) of the slab's elements [UNIT j, UNIT j + UNIT)
  float* amax_part;        // [grid][4] per-wave max |dx| (or null)
  unsigned long long* err;
  unsigned* ticket;        // logical workgroup ids are handed out in order of arrival (see the kernel)
  unsigned long long token;
  int C, HW, cpg, G, T, nteams, NG;
  float inv_count;
};

// Bounded wait of the hand-off: two seconds on the constant 100-MHz clock (s_memrealtime), confirmed by a second reading.
// Round 4: the bound used to be 4e9 SHADER cycles (s_memtime) and fired spuriously in about every second bench.py run
// -- no wave had waited anywhere near two seconds (the runs were not that long); the shader-clock counter is not a
// trustworthy interval across the chip's clock-state changes.  A reading that jumps backwards or by more than 2^40
// ticks (three hours) restarts the window instead of ending the wait.
// Round 5: a wave that does time out leaves what it SAW in the launch's 64-byte error block -- diag[1] = the
// start of its window, [2], [3] = the two elapsed readings (100-MHz ticks), [4] = the shader clock counter, [5] = how often
// its window was restarted by a jumping reading -- so that the next occurrence is diagnosed from evidence
// (eovae_gn_bwd_timeout_info, printed by ops.raise_on_gn_backward_timeouts) instead of guessed at.
__device__ __forceinline__ bool gnf_timed_out(unsigned long long& t0, unsigned long long* diag) {
  const unsigned long long d = __builtin_amdgcn_s_memrealtime() - t0;
  if (d <= 200000000ull) return false;
  if (d >= (1ull << 40)) {
    t0 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(diag + 5, 1ull);
    return false;
  }
  __builtin_amdgcn_s_sleep(8);
  const unsigned long long d2 = __builtin_amdgcn_s_memrealtime() - t0;
  if (!(d2 > 200000000ull && d2 < (1ull << 40))) return false;
  diag[1] = t0;
  diag[2] = d;
  diag[3] = d2;
  diag[4] = __builtin_amdgcn_s_memtime();
  return true;
}

// The token of THIS launch: the host's per-call value advanced by a device-side epoch that the finalize kernel bumps
// at the end of every call.  A HIP graph replays a captured launch with its kernel arguments frozen -- without the
// epoch every replay would carry the token of the capture, and the granules of the previous replay would validate.
__device__ __forceinline__ unsigned long long gnf_token(const GnBwdFArgs& a) {
  const unsigned long long epoch = *reinterpret_cast<const unsigned long long*>(a.ticket + 16);
  return a.token + epoch * 0x9E3779B97F4A7C15ull;
}
// (taking the int by VALUE matters: __builtin_bit_cast applied to the element expression gv[1] itself read element 0)
__device__ __forceinline__ float bits_f(int b) { return __builtin_bit_cast(float, b); }
__device__ __forceinline__ void granule_store(i32x4* p, float s1, float s2, unsigned long long tok) {
  i32x4 v;
  v[0] = __builtin_bit_cast(int, s1);
  v[1] = __builtin_bit_cast(int, s2);
  v[2] = (int)(unsigned)tok;
  v[3] = (int)(unsigned)(tok >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ i32x4 granule_load(const i32x4* p) {
  i32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <bool SILU, bool ADD, int V>
__global__ __launch_bounds__(GNF_NT, gnf_wg_per_cu(V)) void gn_bwd_fused_kernel(GnBwdFArgs a) {
  constexpr int SE = V * GNF_UNIT;                   // elements of one member's slice
  constexpr int NW = GNF_NT / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ double red[NW][2 * V];
  __shared__ float gmean[2];
  __shared__ unsigned ticket_s;
  // logical id = order of ARRIVAL, not blockIdx: the workgroups that are resident then always hold the lowest ids, i.e.
  // whole teams (plus at most one partial team), and whole teams finish without waiting for anybody -- the grid makes
  // progress even when it is NOT fully resident (another process' kernels on the same GPU, e.g. two ranks on one card).
  if (tid == 0) ticket_s = atomicAdd(a.ticket, 1u) % gridDim.x;
  __syncthreads();
  const int bid = (int)ticket_s;
  const unsigned long long token = gnf_token(a);
  const int team = bid / a.T, member = bid % a.T;
  const size_t slab = (size_t)a.cpg * a.HW;
  float4 xr[V], dr[V], xn[V], dn[V], ar[V];
  float amax = 0.f;

  auto slice_ptr = [&](const float* p, int s) { return p + (size_t)s * slab + (size_t)member * SE + 4 * tid; };
  auto prefetch = [&](int s) __attribute__((always_inline)) {
    if (s < a.NG) {
      const float* px = slice_ptr(a.x, s);
      const float* pd = slice_ptr(a.dy, s);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        xn[v] = *reinterpret_cast<const float4*>(px + v * GNF_UNIT);
        dn[v] = *reinterpret_cast<const float4*>(pd + v * GNF_UNIT);
      }
    }
  };

  int s = team;
  prefetch(s);
  for (; s < a.NG; s += a.nteams) {
#pragma unroll
    for (int v = 0; v < V; ++v) { xr[v] = xn[v]; dr[v] = dn[v]; }
    prefetch(s + a.nteams);   // in flight during this slab's arithmetic, hand-off and stores
    if (ADD) {
      const float* pa = slice_ptr(a.dx_add, s);
#pragma unroll
      for (int v = 0; v < V; ++v) ar[v] = *reinterpret_cast<const float4*>(pa + v * GNF_UNIT);
    }
    const int g = s % a.G;
    const float mean = a.stats[2 * s], rstd = a.stats[2 * s + 1];
    // ---- partial sums; the registers then hold  A = rstd * gamma * dz  and  B = rstd * xhat  (dx = A - rstd m1 - B m2)
    float ps1[V], ps2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = g * a.cpg + (member * SE + v * GNF_UNIT) / a.HW;   // uniform: UNIT divides HW
      const float gm = a.gamma[c], bt = a.beta[c];
      float s1 = 0.f, s2 = 0.f;
      auto term = [&](float& xv, float& dv) __attribute__((always_inline)) {
        const float xh = (xv - mean) * rstd;
        float dz = dv;
        if (SILU) {
          const float z = xh * gm + bt;
          const float sg = 1.f / (1.f + __expf(-z));
          dz *= sg * (1.f + z * (1.f - sg));
        }
        s1 += dz;
        s2 += dz * xh;
        dv = rstd * gm * dz;
        xv = rstd * xh;
      };
      term(xr[v].x, dr[v].x); term(xr[v].y, dr[v].y); term(xr[v].z, dr[v].z); term(xr[v].w, dr[v].w);
      ps1[v] = s1;
      ps2[v] = s2;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const double d1 = wave_sum_d((double)ps1[v]), d2 = wave_sum_d((double)ps2[v]);
      if (lane == 0) { red[wave][2 * v] = d1; red[wave][2 * v + 1] = d2; }
    }
    __syncthreads();
    if (wave == 0) {
      // ---- this member's partials: per unit (plain stores, for dgamma / dbeta) and folded with gamma (the granule)
      double t1 = 0, t2 = 0;
      if (lane < V) {
#pragma unroll
        for (int w = 0; w < NW; ++w) { t1 += red[w][2 * lane]; t2 += red[w][2 * lane + 1]; }
        a.part[(size_t)s * a.T * V + member * V + lane] = make_float2((float)t1, (float)t2);
        const float gmv = a.gamma[g * a.cpg + (member * SE + lane * GNF_UNIT) / a.HW];
        t1 *= gmv;
        t2 *= gmv;
      }
      t1 = wave_sum_d(t1);
      t2 = wave_sum_d(t2);
      i32x4* tg = a.team + (size_t)s * a.T;
      if (lane == 0) granule_store(tg + member, (float)t1, (float)t2, token);
      // ---- wait for the team, fold its granules (fixed order: deterministic)
      double m1 = 0, m2 = 0;
      unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100-MHz wall clock
      bool late = false;
      for (int j = lane; j < (EOVAE_GN_NOSYNC ? 0 : a.T); j += 64) {
        i32x4 gv;
        for (;;) {
          gv = granule_load(tg + j);
          const unsigned long long tk = (unsigned long long)(unsigned)gv[2] | ((unsigned long long)(unsigned)gv[3] << 32);
          if (tk == token) break;
          if (gnf_timed_out(t0, a.err)) { late = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        m1 += (double)bits_f(gv[0]);
        m2 += (double)bits_f(gv[1]);
      }
      if (late) *a.err = token;
      m1 = wave_sum_d(m1);
      m2 = wave_sum_d(m2);
      if (lane == 0) {
        gmean[0] = (float)(m1 * (double)a.inv_count);
        gmean[1] = (float)(m2 * (double)a.inv_count);
      }
    }
    __syncthreads();
    const float k1 = rstd * gmean[0], m2f = gmean[1];
    float* po = a.dx + (size_t)s * slab + (size_t)member * SE + 4 * tid;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float4 o;
      o.x = dr[v].x - k1 - xr[v].x * m2f;
      o.y = dr[v].y - k1 - xr[v].y * m2f;
      o.z = dr[v].z - k1 - xr[v].z * m2f;
      o.w = dr[v].w - k1 - xr[v].w * m2f;
      if (ADD) { o.x += ar[v].x; o.y += ar[v].y; o.z += ar[v].z; o.w += ar[v].w; }
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      *reinterpret_cast<float4*>(po + v * GNF_UNIT) = o;
    }
  }
  if (a.amax_part) {
    amax = wave_max(amax);
    if (lane == 0) a.amax_part[bid * NW + wave] = amax;
  }
}

// streamed once: x, dy, the skip gradient and dx bypass the caches' retention (EOVAE_GN_NT=0 builds the plain form)
#ifndef EOVAE_GN_NT
#define EOVAE_GN_NT 1
#endif
__device__ __forceinline__ float4 gnf_ld(const float* p) {
#if EOVAE_GN_NT
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void gnf_st(float* p, const float4& o) {
#if EOVAE_GN_NT
  f32x4 v;
  v[0] = o.x; v[1] = o.y; v[2] = o.z; v[3] = o.w;
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<float4*>(p) = o;
#endif
}
// Software-pipelined form of the kernel above (EOVAE_GN_BWD_PIPE=1): THREE register sets of V float4 pairs -- P (slab
// s: partial sums published, waiting for its team), Q (slab s + 1: being summed) and N (slab s + 2: loads in flight).
// The workgroup publishes slab s + 1 BEFORE it waits for slab s, so the hand-off latency of a slab is covered by the
// arithmetic of the next one.  Same granules, tickets, tokens, partial layout and finalize kernel.
// Round 6: the per-slab chain of DEPENDENT memory round trips is taken out of the iteration -- the kernel ran at "8.3 TB/s
// plus 1.45 ps per element", i.e. ~4.6 us of fixed latency per slab (profiles/r06q_gn_bwd_nosync.log: the skip-gradient form
// moves a third more bytes in 15 % more time).  (i) a slab's scalars (mean, rstd, the slice's gamma / beta) are requested
// with its x / dy, one stage ahead, not at the head of its stage; (ii) the granule's gamma factor comes from those
// registers, not from another load in front of the publishing store; (iii) the team's granules of slab s -- published one
// iteration ago -- are REQUESTED before the arithmetic of slab s + 1 and only looked at behind it (the bounded poll loop
// remains as the fallback for a granule that was not there yet); (iv) rstd of the waiting slab stays in a register.
template <bool SILU, bool ADD, int V>
__global__ __launch_bounds__(GNF_NT, gnf_wg_per_cu(V)) void gn_bwd_fused_pipe_kernel(GnBwdFArgs a) {
  constexpr int SE = V * GNF_UNIT;
  constexpr int NW = GNF_NT / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ double red[2][NW][2 * V];   // by stage parity: the next stage's writes cannot meet wave 0's reads of this one
  __shared__ float gmean[2];
  __shared__ unsigned ticket_s;
  if (tid == 0) ticket_s = atomicAdd(a.ticket, 1u) % gridDim.x;
  __syncthreads();
  // (readfirstlane: the id is uniform, and everything derived from it -- slab, channel, the scalars' addresses -- then lives
  // in scalar registers and is fetched by scalar loads)
  const int bid = __builtin_amdgcn_readfirstlane((int)ticket_s);
  const unsigned long long token = gnf_token(a);
  const int team = bid / a.T, member = bid % a.T;
  const size_t slab = (size_t)a.cpg * a.HW;
  float4 xp[V], dp[V], xq[V], dq[V], xn[V], dn[V], ar[V];
  float mean_n = 0.f, rstd_n = 0.f, gm_n[V], bt_n[V];   // scalars of the slab whose loads are in flight
  float rstd_q = 0.f, rstd_p = 0.f;
  float amax = 0.f;
  int par = 0;

  // In THIS kernel a granule carries the launch's token advanced by its slab: the early request below lands in a register that
  // still holds the previous slab's granule (same launch, same token) if the request's result were ever lost to a spill.
  auto slab_token = [&](int s) { return token + (unsigned long long)(s + 1) * 0xD1B54A32D192ED03ull; };
  auto slice_ptr = [&](const float* p, int s) { return p + (size_t)s * slab + (size_t)member * SE + 4 * tid; };
  auto prefetch = [&](int s) __attribute__((always_inline)) {
    if (s < a.NG) {
      const float* px = slice_ptr(a.x, s);
      const float* pd = slice_ptr(a.dy, s);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        xn[v] = gnf_ld(px + v * GNF_UNIT);
        dn[v] = gnf_ld(pd + v * GNF_UNIT);
      }
      // (constant address space: nothing in this kernel writes stats / gamma / beta, and only that lets the compiler fetch
      // the uniform values by SCALAR loads into scalar registers -- as plain global pointers they were vector loads held
      // in 10 vector registers per set)
      typedef const __attribute__((address_space(4))) float* cptr;
      const cptr stats_c = (cptr)a.stats, gamma_c = (cptr)a.gamma, beta_c = (cptr)a.beta;
      const int g = s % a.G;
      mean_n = stats_c[2 * s];
      rstd_n = stats_c[2 * s + 1];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c = g * a.cpg + (member * SE + v * GNF_UNIT) / a.HW;   // uniform: UNIT divides HW
        gm_n[v] = gamma_c[c];
        bt_n[v] = beta_c[c];
      }
    }
  };
  // Q <- N, sum slab s in Q (Q then holds A = rstd gamma dz, B = rstd xhat), publish this member's granule
  auto stage = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < V; ++v) { xq[v] = xn[v]; dq[v] = dn[v]; }
    const float mean = mean_n, rstd = rstd_n;
    float gm_q[V], bt_q[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { gm_q[v] = gm_n[v]; bt_q[v] = bt_n[v]; }
    rstd_q = rstd;
    prefetch(s + a.nteams);
    float ps1[V], ps2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float gm = gm_q[v], bt = bt_q[v];
      float s1 = 0.f, s2 = 0.f;
      auto term = [&](float& xv, float& dv) __attribute__((always_inline)) {
        const float xh = (xv - mean) * rstd;
        float dz = dv;
        if (SILU) {
          const float z = xh * gm + bt;
          const float sg = 1.f / (1.f + __expf(-z));
          dz *= sg * (1.f + z * (1.f - sg));
        }
        s1 += dz;
        s2 += dz * xh;
        dv = rstd * gm * dz;
        xv = rstd * xh;
      };
      term(xq[v].x, dq[v].x); term(xq[v].y, dq[v].y); term(xq[v].z, dq[v].z); term(xq[v].w, dq[v].w);
      ps1[v] = s1;
      ps2[v] = s2;
    }
    double (&rd)[NW][2 * V] = red[par];
    par ^= 1;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const double d1 = wave_sum_d((double)ps1[v]), d2 = wave_sum_d((double)ps2[v]);
      if (lane == 0) { rd[wave][2 * v] = d1; rd[wave][2 * v + 1] = d2; }
    }
    __syncthreads();
    if (wave == 0) {
      // (the lane number re-derived here: the address a.part + lane is loop-invariant to the compiler, which hoisted it,
      // spilled it and reloaded it right here -- behind an s_waitcnt vmcnt(0) that held the PUBLICATION until the next
      // slab's x / dy had landed)
      int fl;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(fl));
      double t1 = 0, t2 = 0;
      if (fl < V) {
#pragma unroll
        for (int w = 0; w < NW; ++w) { t1 += rd[w][2 * fl]; t2 += rd[w][2 * fl + 1]; }
        a.part[(size_t)s * a.T * V + member * V + fl] = make_float2((float)t1, (float)t2);
        float gmv = gm_q[0];   // gamma of unit `lane`
#pragma unroll
        for (int v = 1; v < V; ++v) gmv = fl == v ? gm_q[v] : gmv;
        t1 *= gmv;
        t2 *= gmv;
      }
      t1 = wave_sum_d(t1);
      t2 = wave_sum_d(t2);
      if (fl == 0) granule_store(a.team + (size_t)s * a.T + member, (float)t1, (float)t2, slab_token(s));
    }
  };
  // the team's granules of slab s, first 64 members: requested (wave 0) without a wait ...
  i32x4 gv0;
  auto team_request = [&](int s) __attribute__((always_inline)) {
    if (wave == 0 && !EOVAE_GN_NOSYNC) {
      const i32x4* p = a.team + (size_t)s * a.T + (lane < a.T ? lane : 0);
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(gv0) : "v"(p) : "memory");
    }
  };
  // ... waited for, checked, then write dx of slab s from P
  auto finish = [&](int s, bool prefetched) __attribute__((always_inline)) {
    if (ADD) {
      const float* pa = slice_ptr(a.dx_add, s);
#pragma unroll
      for (int v = 0; v < V; ++v) ar[v] = gnf_ld(pa + v * GNF_UNIT);
    }
    if (wave == 0) {
      const i32x4* tg = a.team + (size_t)s * a.T;
      const unsigned long long stok = slab_token(s);
      double m1 = 0, m2 = 0;
      unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      bool late = false;
      if (!EOVAE_GN_NOSYNC) {
        // team_request's load only: the vector loads issued behind it -- the next slab's 2 V (when stage() prefetched one)
        // and the skip gradient's V -- stay in flight
        if (prefetched) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * V + (ADD ? V : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ADD ? V : 0) : "memory");
        asm volatile("" : "+v"(gv0));
      }
      for (int j = lane; j < (EOVAE_GN_NOSYNC ? 0 : a.T); j += 64) {
        i32x4 gv = gv0;
        for (bool first = j == lane;; first = false) {
          if (!first) gv = granule_load(tg + j);
          const unsigned long long tk = (unsigned long long)(unsigned)gv[2] | ((unsigned long long)(unsigned)gv[3] << 32);
          if (tk == stok) break;
          if (gnf_timed_out(t0, a.err)) { late = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        m1 += (double)bits_f(gv[0]);
        m2 += (double)bits_f(gv[1]);
      }
      if (late) *a.err = token;
      m1 = wave_sum_d(m1);
      m2 = wave_sum_d(m2);
      if (lane == 0) {
        gmean[0] = (float)(m1 * (double)a.inv_count);
        gmean[1] = (float)(m2 * (double)a.inv_count);
      }
    }
    __syncthreads();
    const float k1 = rstd_p * gmean[0], m2f = gmean[1];
    float* po = a.dx + (size_t)s * slab + (size_t)member * SE + 4 * tid;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float4 o;
      o.x = dp[v].x - k1 - xp[v].x * m2f;
      o.y = dp[v].y - k1 - xp[v].y * m2f;
      o.z = dp[v].z - k1 - xp[v].z * m2f;
      o.w = dp[v].w - k1 - xp[v].w * m2f;
      if (ADD) { o.x += ar[v].x; o.y += ar[v].y; o.z += ar[v].z; o.w += ar[v].w; }
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      gnf_st(po + v * GNF_UNIT, o);
    }
  };

  int s = team;
  if (s < a.NG) {
    prefetch(s);
    stage(s);
    for (;;) {
#pragma unroll
      for (int v = 0; v < V; ++v) { xp[v] = xq[v]; dp[v] = dq[v]; }
      rstd_p = rstd_q;
      const int sn = s + a.nteams;
      team_request(s);
      if (sn < a.NG) stage(sn);       // uniform
      finish(s, sn < a.NG && sn + a.nteams < a.NG);
      if (sn >= a.NG) break;
      s = sn;
    }
  }
  if (a.amax_part) {
    amax = wave_max(amax);
    if (lane == 0) a.amax_part[bid * NW + wave] = amax;
  }
}

// after the fused kernel: dgamma / dbeta from the per-unit partials -- one 64-lane wave per channel, every lane a fixed
// subset of the (image, unit) partials, butterfly at the end: deterministic, and N x HW/1024 (up to 1024) dependent
// 8-byte loads per channel no longer sit in ONE thread (32 us per launch, 1.5 ms per step).  Block 0 also folds max |dx|.
__global__ __launch_bounds__(256) void gn_bwd_fused_finalize_kernel(const float2* __restrict__ part,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    const float* __restrict__ amax_part, int n_part,
                                                                    unsigned* __restrict__ amax_bits,
                                                                    const unsigned long long* __restrict__ err,
                                                                    unsigned long long token, int N, int C, int HW, int cpg,
                                                                    unsigned* __restrict__ ticket,
                                                                    unsigned long long* __restrict__ sticky) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wave;
  // the launch's token (host value + device epoch, as gnf_token); the LAST block to get here resets the ticket counter
  // and bumps the epoch for the next call -- the other blocks have read the epoch by then
  unsigned long long* const epoch = reinterpret_cast<unsigned long long*>(ticket + 16);
  token += *epoch * 0x9E3779B97F4A7C15ull;
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket + 32, 1u) == gridDim.x - 1) {
    ticket[32] = 0u;
    ticket[0] = 0u;       // (tickets are taken modulo the grid: a dirty counter is harmless)
    *epoch += 1;
  }
  if (c < C) {
    const int G = C / cpg, g = c / cpg, cj = c % cpg, per = HW / GNF_UNIT;
    const size_t slab_units = (size_t)cpg * per;
    double a1 = 0, a2 = 0;
    for (int i = lane; i < N * per; i += 64) {
      const int n = i / per, j = i - n * per;
      const float2 v = part[((size_t)n * G + g) * slab_units + (size_t)cj * per + j];
      a1 += (double)v.x;
      a2 += (double)v.y;
    }
    a1 = wave_sum_d(a1);
    a2 = wave_sum_d(a2);
    if (lane == 0) {
      const bool bad = *err == token;
      // a time-out also lands in a host-visible word that survives the call (eovae_gn_bwd_timeouts): the NaNs below poison
      // this step's gradients, the host raises at its next check instead of training on
      if (bad && c == 0) {
        volatile unsigned long long* sv = reinterpret_cast<volatile unsigned long long*>(sticky);
        for (int i = 1; i < 6; ++i) sv[i] = err[i];     // what the timed-out wave saw (gnf_timed_out)
        sv[6] = (unsigned long long)N << 48 | (unsigned long long)C << 32 | (unsigned)HW;
        sv[0] = token | 1ull;
      }
      dbeta[c] = bad ? __builtin_nanf("") : (float)a1;
      dgamma[c] = bad ? __builtin_nanf("") : (float)a2;
    }
  }
  if (blockIdx.x == 0 && amax_bits) {
    __shared__ float wm[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) m = fmaxf(m, amax_part[i]);
    m = wave_max(m);
    if (lane == 0) wm[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
      *amax_bits = (m > 0.f && m < 3.0e38f) ? __builtin_bit_cast(unsigned, m) : 0u;
    }
  }
}

// host-side state of the one-pass backward's hand-off protocol (see eovae_gn_silu_bwd)
constexpr int GN_MAX_DEV = 64, GN_MAX_CTL = 64;
struct GnCtl { int dev; hipStream_t st; unsigned* block; };
std::mutex g_gn_mu;
unsigned long long g_gn_nonce = 0, g_gn_calls = 0;
GnCtl g_gn_ctl[GN_MAX_DEV * GN_MAX_CTL];
int g_gn_nctl = 0;
unsigned* g_gn_pool[GN_MAX_DEV] = {nullptr};   // GN_MAX_CTL blocks of 256 bytes per device, zeroed at allocation
int g_gn_used[GN_MAX_DEV] = {0};
unsigned long long* g_gn_sticky_host[GN_MAX_DEV] = {nullptr};
unsigned long long* g_gn_sticky_dev[GN_MAX_DEV] = {nullptr};

// plan of the fused backward for one shape: V float4 per thread, team size T; T = 0 when the shape is not eligible
struct GnBwdPlan { int V, T, nteams, grid, pipe; };
int g_gn_bwd_fused = -1;   // -1: take EOVAE_GN_BWD_FUSED (default on) at first use
constexpr int GNF_MAX_GRID = 1024;   // 4 workgroups x 256 CUs
GnBwdPlan gn_bwd_plan(int N, int C, int HW, int G) {
  GnBwdPlan p = {0, 0, 0, 0, 0};
  static int n_cu = 0;
  int& enabled = g_gn_bwd_fused;
  if (enabled < 0) {
    const char* e = getenv("EOVAE_GN_BWD_FUSED");
    enabled = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;
  }
  if (n_cu == 0) {
    n_cu = -1;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
  }
  if (!enabled || n_cu <= 0 || HW % GNF_UNIT != 0) return p;
  // EOVAE_GN_BWD_PIPE=0: the single-stage kernel everywhere (A/B); default: the software-pipelined kernel (slices of
  // at most 4 float4 pairs per thread: three register sets, three workgroups per CU), the single-stage kernel with
  // slices of 8 only for slabs whose team would not fit the pipelined grid
  static const int pipe_on = getenv("EOVAE_GN_BWD_PIPE") ? atoi(getenv("EOVAE_GN_BWD_PIPE")) : 1;
  static const int vmax_env = getenv("EOVAE_GN_BWD_VMAX") ? atoi(getenv("EOVAE_GN_BWD_VMAX")) : 0;
  const size_t slab = (size_t)(C / G) * HW;
  const size_t q = slab / GNF_UNIT;
  const int NG = N * G;
  for (int pipe = pipe_on ? 1 : 0; pipe >= 0; --pipe) {
    const int vmax = vmax_env ? vmax_env : (pipe ? 4 : 8);
    if (pipe && vmax > 4) continue;
    const int V = (q % 8 == 0 && vmax >= 8) ? 8 : (q % 4 == 0 && vmax >= 4) ? 4 : (q % 2 == 0 && vmax >= 2) ? 2 : 1;
    int slots = gnf_wg_per_cu(V) * n_cu;        // resident workgroups
    if (slots > GNF_MAX_GRID) slots = GNF_MAX_GRID;
    const size_t T = q / V;
    if (T > (size_t)slots) continue;
    p.V = V;
    p.T = (int)T;
    p.nteams = slots / p.T < NG ? slots / p.T : NG;
    p.grid = p.nteams * p.T;
    p.pipe = pipe;
    return p;
  }
  return p;
}
// workspace of the fused backward: member granules | per-unit partials | per-wave maxima | time-out word
struct GnBwdWs { size_t team, part, amax, err, total; };
GnBwdWs gn_bwd_ws(int N, int C, int HW, int G) {
  GnBwdWs w = {0, 0, 0, 0, 0};
  if (HW % GNF_UNIT != 0) return w;
  const size_t units = (size_t)N * C * (HW / GNF_UNIT);   // >= N*G*T members
  w.team = 0;
  w.part = units * 16;
  w.amax = w.part + units * 8;
  w.err = w.amax + (size_t)GNF_MAX_GRID * 4 * sizeof(float);
  w.total = w.err + 64;
  return w;
}

int nsplit_for(size_t gsize) {
  int ns = (int)((gsize + GN_SPLIT_ELEMS - 1) / GN_SPLIT_ELEMS);
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  // keep slices float4-aligned
  while (ns > 1 && ((gsize + ns - 1) / ns) % 4 != 0) --ns;
  return ns;
}
int chunks_for(int HW) {
  int c = HW / 4096;
  if (c < 1) c = 1;
  if (c > 16) c = 16;
  while (c > 1 && ((HW + c - 1) / c) % 4 != 0) --c;
  return c;
}

}  // namespace

extern "C" {

// bytes of scratch the forward/backward need (double partials)
size_t eovae_gn_workspace_bytes(int N, int C, int HW, int G) {
  size_t gsize = (size_t)(C / G) * HW;
  size_t fwd = (size_t)N * G * nsplit_for(gsize) * 2 * sizeof(double);
  size_t bwd = (size_t)N * C * 2 * sizeof(double) + (size_t)N * G * 2 * sizeof(float) +
               (size_t)N * C * chunks_for(HW) * 4 * sizeof(float);   // + per-wave max|dx| partials
  const size_t fused = gn_bwd_ws(N, C, HW, G).total;
  if (fused > bwd) bwd = fused;
  return fwd > bwd ? fwd : bwd;
}

// stats[N*G][2] = (mean, rstd) is written and must be kept for the backward
// sums (optional): [N*G][2] doubles (sum, sum of squares) of x per (image, group) that the kernel which WROTE x
// has already accumulated (eovae_conv3x3_f16c8_fwd's gn_sums) -- the statistics pass over x is then skipped.
int eovae_gn_silu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                      const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                      void* stream) {
  EOVAE_CHECK_ARG(x && gamma && beta && y && stats && ws, "gn_silu_fwd: null pointer");
  EOVAE_CHECK_ARG(N > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "gn_silu_fwd: C=%d G=%d", C, G);
  hipStream_t st = (hipStream_t)stream;
  const int cpg = C / G;
  const size_t gsize = (size_t)cpg * HW;
  const int ns = nsplit_for(gsize);
  double* part = (double*)ws;
  if (!sums) hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G * ns), dim3(256), 0, st, x, part, gsize, ns);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, st, sums ? sums : part, stats,
                     N * G, sums ? 1 : ns, 1.0 / (double)gsize, eps);
  const int chunks = chunks_for(HW);
  if (silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(N * C * chunks), dim3(256), 0, st, x, gamma, beta,
                       stats, y, C, HW, cpg, chunks);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(N * C * chunks), dim3(256), 0, st, x, gamma,
                       beta, stats, y, C, HW, cpg, chunks);
  EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd");
  return EOVAE_OK;
}

// As eovae_gn_silu_fwd, but the output is written as f16c8 operand records (see gn_apply_packed_kernel):
// pa has N*C*HW dwords (the size of the fp32 output it replaces); C must be a multiple of 8.
int eovae_gn_silu_fwd_packed(const float* x, const float* gamma, const float* beta, void* pa, float* stats,
                             const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                             void* stream) {
  EOVAE_CHECK_ARG(x && gamma && beta && pa && stats && ws, "gn_silu_fwd_packed: null pointer");
  EOVAE_CHECK_ARG(N > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && C % 8 == 0, "gn_silu_fwd_packed: C=%d G=%d", C, G);
  hipStream_t st = (hipStream_t)stream;
  const int cpg = C / G;
  const size_t gsize = (size_t)cpg * HW;
  const int ns = nsplit_for(gsize);
  double* part = (double*)ws;
  const int pblocks = cdiv(HW, 256);
  const dim3 grid(N * (C / 8) * pblocks);
  const double inv_count = 1.0 / (double)gsize;
  if (sums) {   // statistics from the producing convolution's epilogue: finished inside the apply kernel
    if (silu)
      hipLaunchKernelGGL((gn_apply_packed_kernel<true, true>), grid, dim3(256), 0, st, x, gamma, beta, stats, sums, inv_count, eps,
                         (int*)pa, C, HW, cpg, pblocks);
    else
      hipLaunchKernelGGL((gn_apply_packed_kernel<false, true>), grid, dim3(256), 0, st, x, gamma, beta, stats, sums, inv_count,
                         eps, (int*)pa, C, HW, cpg, pblocks);
    EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd_packed");
    return EOVAE_OK;
  }
  hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G * ns), dim3(256), 0, st, x, part, gsize, ns);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, st, part, stats, N * G, ns, inv_count, eps);
  if (silu)
    hipLaunchKernelGGL((gn_apply_packed_kernel<true, false>), grid, dim3(256), 0, st, x, gamma, beta, stats, (const double*)nullptr,
                       inv_count, eps, (int*)pa, C, HW, cpg, pblocks);
  else
    hipLaunchKernelGGL((gn_apply_packed_kernel<false, false>), grid, dim3(256), 0, st, x, gamma, beta, stats,
                       (const double*)nullptr, inv_count, eps, (int*)pa, C, HW, cpg, pblocks);
  EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd_packed");
  return EOVAE_OK;
}

// As eovae_gn_silu_fwd_packed, but the output is written as MX operand records ("format 2", gn_apply_mx_kernel): pa has
// N*C*HW dwords -- the f16 plane in its first half, the MX-FP6 plane in its second; C must be a multiple of 32.  Consumers:
// eovae_conv3x3_f16c8_fwd_mx / the f16c8 weight-gradient entries with xpacked = 2.
int eovae_gn_silu_fwd_packed_mx(const float* x, const float* gamma, const float* beta, void* pa, float* stats,
                                const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                                void* stream) {
  EOVAE_CHECK_ARG(x && gamma && beta && pa && stats && ws, "gn_silu_fwd_packed_mx: null pointer");
  EOVAE_CHECK_ARG(N > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && C % 32 == 0, "gn_silu_fwd_packed_mx: C=%d G=%d", C, G);
  hipStream_t st = (hipStream_t)stream;
  const int cpg = C / G;
  const size_t gsize = (size_t)cpg * HW;
  const int ns = nsplit_for(gsize);
  double* part = (double*)ws;
  const int pblocks = cdiv(HW, 256);
  const dim3 grid(N * (C / 32) * pblocks);
  const double inv_count = 1.0 / (double)gsize;
  if (sums) {
    if (silu)
      hipLaunchKernelGGL((gn_apply_mx_kernel<true, true>), grid, dim3(256), 0, st, x, gamma, beta, stats, sums, inv_count, eps,
                         (int*)pa, N, C, HW, cpg, pblocks);
    else
      hipLaunchKernelGGL((gn_apply_mx_kernel<false, true>), grid, dim3(256), 0, st, x, gamma, beta, stats, sums, inv_count, eps,
                         (int*)pa, N, C, HW, cpg, pblocks);
    EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd_packed_mx");
    return EOVAE_OK;
  }
  hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G * ns), dim3(256), 0, st, x, part, gsize, ns);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, st, part, stats, N * G, ns, inv_count, eps);
  if (silu)
    hipLaunchKernelGGL((gn_apply_mx_kernel<true, false>), grid, dim3(256), 0, st, x, gamma, beta, stats, (const double*)nullptr,
                       inv_count, eps, (int*)pa, N, C, HW, cpg, pblocks);
  else
    hipLaunchKernelGGL((gn_apply_mx_kernel<false, false>), grid, dim3(256), 0, st, x, gamma, beta, stats,
                       (const double*)nullptr, inv_count, eps, (int*)pa, N, C, HW, cpg, pblocks);
  EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd_packed_mx");
  return EOVAE_OK;
}

// switch between the one-pass (fused, default) and the two-pass GroupNorm backward: on = 0 / 1 / 2 sets it (2 = also
// where the two-pass kernels measured faster), anything else only queries; returns the setting now in force (-1 = not decided yet: EOVAE_GN_BWD_FUSED at first use).
// eovae_gn_silu_bwd takes the fused kernel when it is on AND the shape allows it (HW a multiple of 1024, team <= 2 x CUs).
int eovae_gn_bwd_fused(int on) {
  if (on == 0 || on == 1 || on == 2) g_gn_bwd_fused = on;
  return g_gn_bwd_fused;
}
// Time-outs of the one-pass backward's team hand-off on the current device since the last clear: 0 = none, else the
// token of the latest one.  Reads host memory the kernels write through: meaningful for launches the caller has
// synchronised with; costs nothing, so a trainer checks it once per step.  clear != 0 resets the word.
unsigned long long eovae_gn_bwd_timeouts(int clear) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEV) return 0ull;
  std::lock_guard<std::mutex> lock(g_gn_mu);
  unsigned long long* h = g_gn_sticky_host[dev];
  if (!h) return 0ull;
  const unsigned long long v = *reinterpret_cast<volatile unsigned long long*>(h);
  if (clear) *reinterpret_cast<volatile unsigned long long*>(h) = 0ull;
  return v;
}

// What the wave behind the latest time-out saw (gnf_timed_out), for the error message: out[0] = token | 1, [1] = start of
// its wait window and [2], [3] = its two elapsed readings (100-MHz ticks; the bound is 2e8), [4] = shader clock counter,
// [5] = window restarts after a jumping reading, [6] = N << 48 | C << 32 | HW of the call, [7] = 0.  Returns 0, or -1
// when no one-pass backward has run on the current device.
int eovae_gn_bwd_timeout_info(unsigned long long* out8) {
  int dev = 0;
  if (!out8 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEV) return -1;
  std::lock_guard<std::mutex> lock(g_gn_mu);
  const unsigned long long* h = g_gn_sticky_host[dev];
  if (!h) return -1;
  for (int i = 0; i < 8; ++i) out8[i] = reinterpret_cast<const volatile unsigned long long*>(h)[i];
  return 0;
}

// 1 when eovae_gn_silu_bwd would take the fused kernel for this shape
int eovae_gn_bwd_fused_eligible(int N, int C, int HW, int G) {
  if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return 0;
  return gn_bwd_plan(N, C, HW, G).T > 0 ? 1 : 0;
}

// dx (+= dx_add if given), dgamma, dbeta from the saved stats
int eovae_gn_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                      const float* stats, const float* dx_add, float* dx, float* dgamma,
                      float* dbeta, unsigned* dx_amax_bits, int N, int C, int HW, int G, int silu, void* ws,
                      void* stream) {
  EOVAE_CHECK_ARG(x && dy && gamma && beta && stats && dx && dgamma && dbeta && ws,
                  "gn_silu_bwd: null pointer");
  EOVAE_CHECK_ARG(N > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "gn_silu_bwd: C=%d G=%d", C, G);
  hipStream_t st = (hipStream_t)stream;
  const int cpg = C / G;
  GnBwdPlan plan = gn_bwd_plan(N, C, HW, G);
  // measured (profiles/r03l_gn_bwd_ab.log): the SINGLE-STAGE kernel loses 2 % to the two-pass kernels with 4 channels
  // per group and no skip gradient to fold in; the software-pipelined kernel wins everywhere (1.18-1.43 x).
  // EOVAE_GN_BWD_FUSED=2 (or eovae_gn_bwd_fused(2)) forces the one-pass kernel for every eligible shape.
  if (plan.T > 0 && !plan.pipe && g_gn_bwd_fused != 2 && !dx_add && cpg < 8) plan.T = 0;
  if (plan.T > 0) {
    // token: unique per call AND per process -- the workspace may sit on device memory that an earlier (or a
    // concurrent) process filled with granules of the very same call sequence, so a plain call counter is not enough
    // (seen: two ranks sharing one GPU accepted each other's stale granules now and then).  64 random-looking bits:
    // splitmix64 of (process nonce + call number).
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEV) {
      eovae_set_error("gn_silu_bwd: device index");
      return EOVAE_ERR_LAUNCH;
    }
    // Process-wide state of the hand-off protocol, all behind one mutex (ctypes releases the GIL: two host threads may be
    // in here at once): the token sequence, and one control block (ticket counter + epoch) per (device, STREAM) -- two
    // backward passes on different streams of one device must not draw tickets from the same counter (their members
    // would interleave and the teams time out).  A block is zeroed on the stream that owns it, i.e. ordered in front of
    // its first use; the time-out word is host memory mapped into the device, one per device.
    unsigned long long z;
    unsigned* ticket_block = nullptr;
    unsigned long long* sticky_dev = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_gn_mu);
      if (g_gn_nonce == 0) {
        std::random_device rd;
        g_gn_nonce = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ ((unsigned long long)getpid() << 17) ^
                     (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        g_gn_nonce |= 1;
      }
      z = g_gn_nonce + 0x9E3779B97F4A7C15ull * ++g_gn_calls;
      // control blocks come from a per-device pool that is allocated and zeroed ONCE (first call on the device, which must
      // not sit inside a stream capture); handing a block to a new stream -- a graph capture runs on a stream of its own --
      // then needs no allocation and no memset.  A pool that is used up recycles its oldest assignment.
      if (!g_gn_pool[dev]) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
          (void)hipGetLastError();
          eovae_set_error("gn_silu_bwd: the first one-pass backward on a device allocates its control blocks and cannot run "
                          "inside a stream capture (call it once eagerly first)");
          return EOVAE_ERR_UNSUPPORTED;
        }
        unsigned* pool = nullptr;
        if (hipMalloc((void**)&pool, (size_t)GN_MAX_CTL * 256) != hipSuccess ||
            hipMemset(pool, 0, (size_t)GN_MAX_CTL * 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
          eovae_set_error("gn_silu_bwd: cannot allocate the ticket counters");
          return EOVAE_ERR_LAUNCH;
        }
        g_gn_pool[dev] = pool;
      }
      if (!g_gn_sticky_host[dev]) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
          eovae_set_error("gn_silu_bwd: cannot allocate the time-out word");
          return EOVAE_ERR_LAUNCH;
        }
        for (int i = 0; i < 8; ++i) reinterpret_cast<volatile unsigned long long*>(h)[i] = 0ull;
        g_gn_sticky_host[dev] = reinterpret_cast<unsigned long long*>(h);
        g_gn_sticky_dev[dev] = reinterpret_cast<unsigned long long*>(d);
      }
      sticky_dev = g_gn_sticky_dev[dev];
      for (int i = 0; i < g_gn_nctl; ++i)
        if (g_gn_ctl[i].dev == dev && g_gn_ctl[i].st == st) ticket_block = g_gn_ctl[i].block;
      if (!ticket_block) {
        int slot = g_gn_used[dev]++ % GN_MAX_CTL;               // (a recycled block is in its rest state: every finalize resets it)
        ticket_block = g_gn_pool[dev] + (size_t)slot * 64;
        int e = -1;
        for (int i = 0; i < g_gn_nctl; ++i)
          if (g_gn_ctl[i].dev == dev && g_gn_ctl[i].block == ticket_block) e = i;   // the assignment being recycled
        if (e >= 0) {
          // More than GN_MAX_CTL streams have run a one-pass backward on this device: the block's previous stream may
          // still be drawing tickets from it.  That stream may also be gone, so the device is drained -- once
          // per recycled block, never in a steady state; inside a stream capture that is not possible: refuse.
          hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            --g_gn_used[dev];
            eovae_set_error("gn_silu_bwd: the pool of %d ticket blocks is used up and a block cannot be recycled inside a "
                            "stream capture", GN_MAX_CTL);
            return EOVAE_ERR_UNSUPPORTED;
          }
          if (hipDeviceSynchronize() != hipSuccess) {
            eovae_set_error("gn_silu_bwd: draining the device before recycling a ticket block failed");
            return EOVAE_ERR_LAUNCH;
          }
        }
        if (e < 0) e = g_gn_nctl < GN_MAX_DEV * GN_MAX_CTL ? g_gn_nctl++ : 0;
        g_gn_ctl[e].dev = dev;
        g_gn_ctl[e].st = st;
        g_gn_ctl[e].block = ticket_block;
      }
    }
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    GnBwdFArgs a;
    a.x = x; a.dy = dy; a.gamma = gamma; a.beta = beta; a.stats = stats; a.dx_add = dx_add; a.dx = dx;
    const GnBwdWs w = gn_bwd_ws(N, C, HW, G);
    a.ticket = ticket_block;
    a.team = (i32x4*)((char*)ws + w.team);
    a.part = (float2*)((char*)ws + w.part);
    a.amax_part = dx_amax_bits ? (float*)((char*)ws + w.amax) : nullptr;
    a.err = (unsigned long long*)((char*)ws + w.err);
    a.token = z;
    a.C = C; a.HW = HW; a.cpg = cpg; a.G = G; a.T = plan.T; a.nteams = plan.nteams; a.NG = N * G;
    a.inv_count = (float)(1.0 / ((double)cpg * HW));
    void (*k)(GnBwdFArgs) = nullptr;
#define EOVAE_GN_PICK(S, A)                                                                         \
    k = plan.pipe ? (plan.V == 4 ? gn_bwd_fused_pipe_kernel<S, A, 4> : plan.V == 2 ? gn_bwd_fused_pipe_kernel<S, A, 2> \
                                                                       : gn_bwd_fused_pipe_kernel<S, A, 1>)          \
      : plan.V == 8 ? gn_bwd_fused_kernel<S, A, 8> : plan.V == 4 ? gn_bwd_fused_kernel<S, A, 4>     \
      : plan.V == 2 ? gn_bwd_fused_kernel<S, A, 2> : gn_bwd_fused_kernel<S, A, 1>
    if (silu && dx_add) EOVAE_GN_PICK(true, true);
    else if (silu) EOVAE_GN_PICK(true, false);
    else if (dx_add) EOVAE_GN_PICK(false, true);
    else EOVAE_GN_PICK(false, false);
#undef EOVAE_GN_PICK
    hipLaunchKernelGGL(k, dim3(plan.grid), dim3(GNF_NT), 0, st, a);
    hipLaunchKernelGGL(gn_bwd_fused_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, a.part, dgamma, dbeta,
                       a.amax_part, plan.grid * (GNF_NT / 64), dx_amax_bits, a.err, a.token, N, C, HW, cpg, a.ticket, sticky_dev);
    EOVAE_CHECK_LAUNCH("eovae_gn_silu_bwd (fused)");
    return EOVAE_OK;
  }
  double* sums = (double*)ws;
  float* gm = (float*)(sums + (size_t)N * C * 2);
  if (silu)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<true>, dim3(N * C), dim3(256), 0, st, x, dy, gamma, beta,
                       stats, sums, C, HW, cpg);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<false>, dim3(N * C), dim3(256), 0, st, x, dy, gamma,
                       beta, stats, sums, C, HW, cpg);
  int nt = C > N * G ? C : N * G;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(cdiv(nt, 256)), dim3(256), 0, st, sums, gamma,
                     dgamma, dbeta, gm, N, C, cpg, 1.0 / ((double)cpg * HW), dx_amax_bits);
  const int chunks = chunks_for(HW);
  float* part = dx_amax_bits ? gm + (size_t)N * G * 2 : nullptr;
  if (silu)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(N * C * chunks), dim3(256), 0, st, x, dy,
                       gamma, beta, stats, gm, dx_add, dx, C, HW, cpg, chunks, part);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(N * C * chunks), dim3(256), 0, st, x, dy,
                       gamma, beta, stats, gm, dx_add, dx, C, HW, cpg, chunks, part);
  if (dx_amax_bits)
    hipLaunchKernelGGL(amax_fold_kernel, dim3(16), dim3(256), 0, st, part, N * C * chunks * 4, dx_amax_bits);
  EOVAE_CHECK_LAUNCH("eovae_gn_silu_bwd");
  return EOVAE_OK;
}

}  // extern "C"
