// Optimiser step of EOFluxVAE.training_step (new_autoencoder.py:648-655) over ONE flat fp32
// parameter/gradient buffer: global grad-norm (clip_grad_norm_) + Adam, HBM-bound streaming.
#include "eovae_common.h"

namespace {

__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g, size_t n,
                                                            double* __restrict__ part) {
  float s = 0.f;
  double acc = 0.0;
  size_t i = (blockIdx.x * (size_t)256 + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * 1024;
  int cnt = 0;
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(g + i);
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    if (++cnt == 64) { acc += s; s = 0.f; cnt = 0; }
  }
  for (; i < n && i + 3 >= n; ++i) {  // ragged tail (at most 3 elements, one thread)
    s += g[i] * g[i];
  }
  acc += s;
  __shared__ double red[4];
  double d = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm_out[0] = sqrt(sum), norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ void norm_finalize_kernel(const double* __restrict__ part, int nparts,
                                     float* __restrict__ norm_out, float max_norm) {
  double s = 0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    float nrm = (float)sqrt(s);
    norm_out[0] = nrm;
    float coef = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.f;
    norm_out[1] = coef < 1.f ? coef : 1.f;
  }
}

// torch.optim.Adam (no amsgrad); decay > 0 = torch.optim.AdamW's decoupled weight decay
// (p *= 1 - lr * weight_decay before the update, weight_distill_train.py:317); gradient pre-scaled by *clip_coef
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt,
                                                   const float* __restrict__ clip_coef,
                                                   float grad_scale, float decay) {
  const float cs = (clip_coef ? clip_coef[1] : 1.f) * grad_scale;
  // Fault containment: a non-finite gradient norm -- a GroupNorm-backward time-out poisons dgamma / dbeta with NaN,
  // and under data parallelism the all-reduce hands the same NaN to every rank -- leaves parameters and moments untouched.
  // Every rank takes the same branch (same averaged gradient, same norm), so replicas stay identical and a checkpoint written
  // before the host raises (ops.raise_on_gn_backward_timeouts, up to three steps later) is clean.  The reference would
  // write NaN into the weights here (clip_grad_norm_ + Adam on a NaN gradient); without a clip there is no norm to look at.
  if (clip_coef && !(fabsf(clip_coef[0]) <= 3.0e38f)) return;
  const float step = lr / bc1, keep = 1.f - lr * decay;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gi = g[i] * cs;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] * keep - step * (mi / denom);
  }
}

// Gather of per-parameter gradient tensors into the flat gradient buffer: one block per table row
// (src pointer as integer or 0 for "no gradient" => zeros, destination element offset, element count
// <= 16384).  Replaces one autograd accumulate-add launch per parameter (363 per step) plus the zero fill.
__global__ __launch_bounds__(256) void multi_gather_kernel(const long long* __restrict__ table,
                                                           float* __restrict__ flat) {
  const long long* row = table + (size_t)blockIdx.x * 3;
  const float* src = reinterpret_cast<const float*>(row[0]);
  float* dst = flat + row[1];
  const int n = (int)row[2];
  const bool vec = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  if (vec) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 v = src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(dst)[i] = v;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) dst[i] = src ? src[i] : 0.f;
  } else {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src ? src[i] : 0.f;
  }
}

__global__ void scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ coef,
                             float s) {
  const float cs = (coef ? coef[1] : 1.f) * s;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    g[i] *= cs;
}

}  // namespace

extern "C" {

size_t eovae_gradnorm_workspace_bytes(void) { return 1024 * sizeof(double); }

// norm_out[0] = ||g||_2, norm_out[1] = clip coefficient for max_norm (1 if max_norm <= 0)
int eovae_grad_norm(const float* g, size_t n, float max_norm, float* norm_out, void* ws,
                    void* stream) {
  EOVAE_CHECK_ARG(g && norm_out && ws && n > 0, "grad_norm: bad argument");
  EOVAE_CHECK_ARG(((uintptr_t)g & 15) == 0, "grad_norm: buffer must be 16-byte aligned");
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3(blocks), dim3(256), 0, st, g, n, (double*)ws);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, blocks,
                     norm_out, max_norm);
  EOVAE_CHECK_LAUNCH("eovae_grad_norm");
  return EOVAE_OK;
}

// step >= 1 is the Adam step count AFTER increment (bias corrections 1 - beta^step)
int eovae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, int step, const float* clip, float grad_scale,
                    float weight_decay, void* stream) {
  EOVAE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && weight_decay >= 0.f, "adam_step: bad argument");
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), clip, grad_scale, weight_decay);
  EOVAE_CHECK_LAUNCH("eovae_adam_step");
  return EOVAE_OK;
}

// table: nrows x {src address (0 = zeros), destination element offset in flat, element count} (device, int64)
int eovae_multi_gather(const long long* table, int nrows, float* flat, void* stream) {
  EOVAE_CHECK_ARG(table && flat && nrows > 0, "multi_gather: bad argument");
  hipLaunchKernelGGL(multi_gather_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, table, flat);
  EOVAE_CHECK_LAUNCH("eovae_multi_gather");
  return EOVAE_OK;
}

int eovae_scale_inplace(float* g, size_t n, const float* clip, float s, void* stream) {
  EOVAE_CHECK_ARG(g && n > 0, "scale_inplace: bad argument");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, clip, s);
  EOVAE_CHECK_LAUNCH("eovae_scale_inplace");
  return EOVAE_OK;
}

}  // extern "C"
