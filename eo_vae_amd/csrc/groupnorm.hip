// GroupNorm(32 groups, eps 1e-6, affine) fused with the swish that follows it in the
// reference (layers.py:61-63,78-80,91-92,106-108; model.py:159-161,193-194,295-297,349-350):
//     y = silu( (x - mean_g) * rstd_g * gamma_c + beta_c )        (silu optional: AttnBlock norm)
// NCHW fp32: one (n, group) is a contiguous run of cpg*HW floats (up to 1 MB at 4 x 256^2), so
// statistics are a pure streaming reduction: float4 loads, per-thread fp32 partials, wave64
// shuffle trees, cross-wave/cross-block combine in double.  Memory-bound (HBM roofline).
#include "eovae_common.h"

namespace {

constexpr int GN_SPLIT_ELEMS = 16384;  // elements per stats block

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
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_packed_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ stats,
                                                              int* __restrict__ pa, int C, int HW, int cpg,
                                                              int pblocks) {
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const int pb = blockIdx.x % pblocks, og = blockIdx.x / pblocks;   // og = n * C/8 + octet
  const int oct = og % (C / 8), n = og / (C / 8);
  const int p = pb * 256 + threadIdx.x;
  if (p >= HW) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = oct * 8 + j;
    const int ng = n * (C / cpg) + c / cpg;
    const float mean = stats[2 * ng], rstd = stats[2 * ng + 1];
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
  if (!sums) hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G * ns), dim3(256), 0, st, x, part, gsize, ns);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, st, sums ? sums : part, stats,
                     N * G, sums ? 1 : ns, 1.0 / (double)gsize, eps);
  const int pblocks = cdiv(HW, 256);
  if (silu)
    hipLaunchKernelGGL(gn_apply_packed_kernel<true>, dim3(N * (C / 8) * pblocks), dim3(256), 0, st, x, gamma, beta,
                       stats, (int*)pa, C, HW, cpg, pblocks);
  else
    hipLaunchKernelGGL(gn_apply_packed_kernel<false>, dim3(N * (C / 8) * pblocks), dim3(256), 0, st, x, gamma,
                       beta, stats, (int*)pa, C, HW, cpg, pblocks);
  EOVAE_CHECK_LAUNCH("eovae_gn_silu_fwd_packed");
  return EOVAE_OK;
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
