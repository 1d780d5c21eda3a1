// Reconstruction losses of EOConsistencyLoss (eo_vae/models/modules/consistency_loss.py):
//   L1           F.l1_loss                      :418
//   Charbonnier  mean(sqrt(d^2 + eps^2))        :12-21  (eps = 1e-3)
//   SAM          mean(1 - cos(pred, target))    :186-209   (spectral branch)
//   GDL          gradient-difference loss       :243-269   (spatial branch,  SURVEY 8f N4)
// Streaming reductions (HBM-bound): wave64 shuffle trees, double across blocks.
#include "eovae_common.h"

namespace {

// KIND 0 = L1, 1 = Charbonnier, 2 = squared error (F.mse_loss of the distillation step,
// weight_distill_train.py:211-228)
template <int KIND>
__global__ __launch_bounds__(256) void pixel_loss_partial_kernel(const float* __restrict__ pred,
                                                                 const float* __restrict__ target,
                                                                 size_t n, float eps2,
                                                                 double* __restrict__ part) {
  double acc = 0.0;
  float s = 0.f;
  int cnt = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float d = pred[i] - target[i];
    s += KIND == 1 ? sqrtf(d * d + eps2) : (KIND == 2 ? d * d : fabsf(d));
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
template <int KIND>
__global__ __launch_bounds__(256) void pixel_loss_bwd_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ gout,
                                                             float* __restrict__ dpred, size_t n,
                                                             float eps2, float inv_n) {
  const float gs = gout[0] * inv_n;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float d = pred[i] - target[i];
    float g;
    if (KIND == 1) g = d / sqrtf(d * d + eps2);
    else if (KIND == 2) g = 2.f * d;
    else g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: sign(0) = 0
    dpred[i] = gs * g;
  }
}

// ---- spectral angle (SAMLoss, consistency_loss.py:186-209): mean over pixels of 1 - cos(pred, target)
// along the channel axis.  One thread per pixel, channel stride HW.
__device__ __forceinline__ void sam_moments(const float* p, const float* t, int C, size_t HW, float& dot,
                                            float& rr, float& tt) {
  dot = rr = tt = 0.f;
  for (int c = 0; c < C; ++c) {
    const float a = p[c * HW], b = t[c * HW];
    dot += a * b; rr += a * a; tt += b * b;
  }
}

__global__ __launch_bounds__(256) void sam_partial_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target, int N, int C,
                                                          size_t HW, float eps, double* __restrict__ part) {
  double acc = 0.0;
  const size_t total = (size_t)N * HW;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / HW, px = i - n * HW, base = n * C * HW + px;
    float dot, rr, tt;
    sam_moments(pred + base, target + base, C, HW, dot, rr, tt);
    acc += (double)(1.f - dot / (sqrtf(rr) * sqrtf(tt) + eps));
  }
  __shared__ double red[4];
  double d = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sam_bwd_kernel(const float* __restrict__ pred,
                                                      const float* __restrict__ target,
                                                      const float* __restrict__ gout, float* __restrict__ dpred,
                                                      int N, int C, size_t HW, float eps, float inv_n) {
  const float gs = gout[0] * inv_n;
  const size_t total = (size_t)N * HW;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / HW, px = i - n * HW, base = n * C * HW + px;
    float dot, rr, tt;
    sam_moments(pred + base, target + base, C, HW, dot, rr, tt);
    const float nr = sqrtf(rr), nt = sqrtf(tt), den = nr * nt + eps;
    const float k1 = gs / den, k2 = nr > 0.f ? gs * dot * nt / (den * den * nr) : 0.f;  // d|r|/dr = 0 at r = 0
    for (int c = 0; c < C; ++c)
      dpred[base + c * HW] = k2 * pred[base + c * HW] - k1 * target[base + c * HW];
  }
}

// ---- gradient-difference loss (consistency_loss.py:243-269, alpha = 1 as EOConsistencyLoss builds it):
//   mean_x | |p[x+1]-p[x]| - |t[x+1]-t[x]| |  +  mean_y | |p[y+1]-p[y]| - |t[y+1]-t[y]| |
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void gdl_partial_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target, size_t planes,
                                                          int H, int W, double* __restrict__ part) {
  double ax = 0.0, ay = 0.0;
  const size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = i % W, y = (i / W) % H;
    const float p = pred[i], t = target[i];
    if (x + 1 < W) ax += (double)fabsf(fabsf(pred[i + 1] - p) - fabsf(target[i + 1] - t));
    if (y + 1 < H) ay += (double)fabsf(fabsf(pred[i + W] - p) - fabsf(target[i + W] - t));
  }
  __shared__ double red[8];
  const double dx = wave_sum_d(ax), dy = wave_sum_d(ay);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = dx; red[4 + (threadIdx.x >> 6)] = dy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    part[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ void gdl_finalize_kernel(const double* __restrict__ part, int nparts, double inv_x, double inv_y,
                                    float* __restrict__ out) {
  double sx = 0, sy = 0;
  for (int i = threadIdx.x; i < nparts; i += 64) { sx += part[2 * i]; sy += part[2 * i + 1]; }
  sx = wave_sum_d(sx); sy = wave_sum_d(sy);
  if (threadIdx.x == 0) out[0] = (float)(sx * inv_x) + (float)(sy * inv_y);
}

// edge (a -> b): d/dp_b = +sign(u) sign(dp), d/dp_a = -sign(u) sign(dp), u = |dp| - |dt|
__device__ __forceinline__ float gdl_edge(float pa, float pb, float ta, float tb) {
  const float dp = pb - pa;
  return sgn(fabsf(dp) - fabsf(tb - ta)) * sgn(dp);
}

__global__ __launch_bounds__(256) void gdl_bwd_kernel(const float* __restrict__ pred,
                                                      const float* __restrict__ target,
                                                      const float* __restrict__ gout, float* __restrict__ dpred,
                                                      size_t planes, int H, int W, float inv_x, float inv_y) {
  const float gx = gout[0] * inv_x, gy = gout[0] * inv_y;
  const size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = i % W, y = (i / W) % H;
    const float p = pred[i], t = target[i];
    float g = 0.f;
    if (x > 0) g += gx * gdl_edge(pred[i - 1], p, target[i - 1], t);
    if (x + 1 < W) g -= gx * gdl_edge(p, pred[i + 1], t, target[i + 1]);
    if (y > 0) g += gy * gdl_edge(pred[i - W], p, target[i - W], t);
    if (y + 1 < H) g -= gy * gdl_edge(p, pred[i + W], t, target[i + W]);
    dpred[i] = g;
  }
}

}  // namespace

extern "C" {

size_t eovae_loss_workspace_bytes(void) { return 2 * 2048 * sizeof(double); }

int eovae_sam_loss_fwd(const float* pred, const float* target, int N, int C, int HW, float eps, float* out,
                       void* ws, void* stream) {
  EOVAE_CHECK_ARG(pred && target && out && ws && N > 0 && C > 0 && HW > 0, "sam_loss_fwd: bad argument");
  const size_t total = (size_t)N * HW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sam_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, target, N, C, (size_t)HW, eps,
                     (double*)ws);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, blocks,
                     1.0 / (double)total, out);
  EOVAE_CHECK_LAUNCH("eovae_sam_loss_fwd");
  return EOVAE_OK;
}

int eovae_sam_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int N, int C,
                       int HW, float eps, void* stream) {
  EOVAE_CHECK_ARG(pred && target && gout && dpred && N > 0 && C > 0 && HW > 0, "sam_loss_bwd: bad argument");
  const size_t total = (size_t)N * HW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sam_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, gout, dpred,
                     N, C, (size_t)HW, eps, (float)(1.0 / (double)total));
  EOVAE_CHECK_LAUNCH("eovae_sam_loss_bwd");
  return EOVAE_OK;
}

int eovae_gdl_loss_fwd(const float* pred, const float* target, int planes, int H, int W, float* out, void* ws,
                       void* stream) {
  EOVAE_CHECK_ARG(pred && target && out && ws && planes > 0 && H > 1 && W > 1, "gdl_loss_fwd: bad argument");
  const size_t total = (size_t)planes * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gdl_partial_kernel, dim3(blocks), dim3(256), 0, st, pred, target, (size_t)planes, H, W,
                     (double*)ws);
  hipLaunchKernelGGL(gdl_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, blocks,
                     1.0 / ((double)planes * H * (W - 1)), 1.0 / ((double)planes * (H - 1) * W), out);
  EOVAE_CHECK_LAUNCH("eovae_gdl_loss_fwd");
  return EOVAE_OK;
}

int eovae_gdl_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int planes, int H,
                       int W, void* stream) {
  EOVAE_CHECK_ARG(pred && target && gout && dpred && planes > 0 && H > 1 && W > 1, "gdl_loss_bwd: bad argument");
  const size_t total = (size_t)planes * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gdl_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, gout, dpred,
                     (size_t)planes, H, W, (float)(1.0 / ((double)planes * H * (W - 1))),
                     (float)(1.0 / ((double)planes * (H - 1) * W)));
  EOVAE_CHECK_LAUNCH("eovae_gdl_loss_bwd");
  return EOVAE_OK;
}

// kind: 0 = L1, 1 = Charbonnier(eps); out[0] = mean loss
int eovae_pixel_loss_fwd(const float* pred, const float* target, size_t n, int kind, float eps,
                         float* out, void* ws, void* stream) {
  EOVAE_CHECK_ARG(pred && target && out && ws && n > 0, "pixel_loss_fwd: bad argument");
  EOVAE_CHECK_ARG(kind >= 0 && kind <= 2, "pixel_loss_fwd: kind %d unknown", kind);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 1)
    hipLaunchKernelGGL(pixel_loss_partial_kernel<1>, dim3(blocks), dim3(256), 0, st, pred, target,
                       n, eps * eps, (double*)ws);
  else if (kind == 2)
    hipLaunchKernelGGL(pixel_loss_partial_kernel<2>, dim3(blocks), dim3(256), 0, st, pred, target,
                       n, 0.f, (double*)ws);
  else
    hipLaunchKernelGGL(pixel_loss_partial_kernel<0>, dim3(blocks), dim3(256), 0, st, pred,
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
  EOVAE_CHECK_ARG(kind >= 0 && kind <= 2, "pixel_loss_bwd: kind %d unknown", kind);
  if (kind == 1)
    hipLaunchKernelGGL(pixel_loss_bwd_kernel<1>, dim3(blocks), dim3(256), 0, st, pred, target,
                       gout, dpred, n, eps * eps, (float)(1.0 / (double)n));
  else if (kind == 2)
    hipLaunchKernelGGL(pixel_loss_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, pred, target,
                       gout, dpred, n, 0.f, (float)(1.0 / (double)n));
  else
    hipLaunchKernelGGL(pixel_loss_bwd_kernel<0>, dim3(blocks), dim3(256), 0, st, pred, target,
                       gout, dpred, n, 0.f, (float)(1.0 / (double)n));
  EOVAE_CHECK_LAUNCH("eovae_pixel_loss_bwd");
  return EOVAE_OK;
}

}  // extern "C"
