// Reconstruction losses of EOConsistencyLoss (eo_vae/models/modules/consistency_loss.py):
//   L1           F.l1_loss                      :418
//   Charbonnier  mean(sqrt(d^2 + eps^2))        :12-21  (eps = 1e-3)
// Streaming reductions (HBM-bound): float4 loads, wave64 shuffle trees, double across blocks.
#include "eovae_common.h"

namespace {

template <bool CHARB>
__global__ __launch_bounds__(256) void pixel_loss_partial_kernel(const float* __restrict__ pred,
                                                                 const float* __restrict__ target,
                                                                 size_t n, float eps2,
                                                                 double* __restrict__ part) {
  double acc = 0.0;
  float s = 0.f;
  int cnt = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float d = pred[i] - target[i];
    s += CHARB ? sqrtf(d * d + eps2) : fabsf(d);
    if (++cnt == 256) { acc += s; s = 0.f; cnt = 0; }
  }
  acc += s;
  __shared__ double red[4];
  double d = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void loss_finalize_kernel(const double* __restrict__ part, int nparts, double inv_n,
                                     float* __restrict__ out) {
  double s = 0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}

// dpred = (*gout / n) * dloss/dd
template <bool CHARB>
__global__ __launch_bounds__(256) void pixel_loss_bwd_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ gout,
                                                             float* __restrict__ dpred, size_t n,
                                                             float eps2, float inv_n) {
  const float gs = gout[0] * inv_n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float d = pred[i] - target[i];
    float g;
    if (CHARB) g = d / sqrtf(d * d + eps2);
    else g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: sign(0) = 0
    dpred[i] = gs * g;
  }
}

}  // namespace

extern "C" {

size_t eovae_loss_workspace_bytes(void) { return 2048 * sizeof(double); }

// kind: 0 = L1, 1 = Charbonnier(eps); out[0] = mean loss
int eovae_pixel_loss_fwd(const float* pred, const float* target, size_t n, int kind, float eps,
                         float* out, void* ws, void* stream) {
  EOVAE_CHECK_ARG(pred && target && out && ws && n > 0, "pixel_loss_fwd: bad argument");
  EOVAE_CHECK_ARG(kind == 0 || kind == 1, "pixel_loss_fwd: kind %d unknown", kind);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 1)
    hipLaunchKernelGGL(pixel_loss_partial_kernel<true>, dim3(blocks), dim3(256), 0, st, pred, target,
                       n, eps * eps, (double*)ws);
  else
    hipLaunchKernelGGL(pixel_loss_partial_kernel<false>, dim3(blocks), dim3(256), 0, st, pred,
                       target, n, 0.f, (double*)ws);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, blocks,
                     1.0 / (double)n, out);
  EOVAE_CHECK_LAUNCH("eovae_pixel_loss_fwd");
  return EOVAE_OK;
}

// gout: device scalar (upstream gradient of the mean loss)
int eovae_pixel_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred,
                         size_t n, int kind, float eps, void* stream) {
  EOVAE_CHECK_ARG(pred && target && gout && dpred && n > 0, "pixel_loss_bwd: bad argument");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 1)
    hipLaunchKernelGGL(pixel_loss_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, pred, target,
                       gout, dpred, n, eps * eps, (float)(1.0 / (double)n));
  else
    hipLaunchKernelGGL(pixel_loss_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, pred, target,
                       gout, dpred, n, 0.f, (float)(1.0 / (double)n));
  EOVAE_CHECK_LAUNCH("eovae_pixel_loss_bwd");
  return EOVAE_OK;
}

}  // extern "C"
