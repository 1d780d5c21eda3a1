// Latent-space glue of EOFluxVAE (tiny tensors, one fused kernel each; HBM/latency bound):
//   * posterior reparameterisation + KL   eo_vae/models/modules/distributions.py:28-32,44-46,64-67
//   * 2x2 pixel-unshuffle + BatchNorm2d(4z, affine=False)   new_autoencoder.py:466-469,533-536
//   * inverse BN with running stats (eps 1e-4) + pixel-shuffle   new_autoencoder.py:423-428,538-543
// Spatial layout  z[b][c][2i+pi][2j+pj]  <->  packed layout  zp[b][c*4+pi*2+pj][i][j].
#include "eovae_common.h"

namespace {

// z = mean + exp(0.5*clamp(logvar,-30,20)) * eps ; kl[b] = 0.5*sum(mean^2 + var - 1 - logvar)
// One block of 1024 threads per image (the per-image KL sum stays a single deterministic reduction): this kernel
// sits on the step's critical path between encoder and decoder, where nothing else runs -- with 256 threads per
// image it took 100 us for 2 MB.
__global__ __launch_bounds__(1024) void posterior_fwd_kernel(const float* __restrict__ mom,
                                                             const float* __restrict__ eps,
                                                             float* __restrict__ z,
                                                             float* __restrict__ kl, int per) {
  const int b = blockIdx.x;
  const float* mean = mom + (size_t)b * 2 * per;
  const float* lv = mean + per;
  float acc = 0.f;
  for (int i = threadIdx.x; i < per; i += 1024) {
    float m = mean[i];
    float l = fminf(fmaxf(lv[i], -30.f), 20.f);
    float var = __expf(l);
    if (z) z[(size_t)b * per + i] = eps ? m + __expf(0.5f * l) * eps[(size_t)b * per + i] : m;
    acc += m * m + var - 1.f - l;
  }
  if (kl) {
    __shared__ double red[16];
    double d = wave_sum_d((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < 16; ++w) t += red[w];
      kl[b] = (float)(0.5 * t);
    }
  }
}

// dmom from dz (may be null) and dkl[b] (may be null)
__global__ void posterior_bwd_kernel(const float* __restrict__ mom, const float* __restrict__ eps,
                                     const float* __restrict__ dz, const float* __restrict__ dkl,
                                     float* __restrict__ dmom, int per, size_t total) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / per, r = i % per;
  const float* mean = mom + (size_t)b * 2 * per;
  const float raw = mean[per + r];
  const float l = fminf(fmaxf(raw, -30.f), 20.f);
  const float g = dz ? dz[i] : 0.f;
  const float gk = dkl ? dkl[b] : 0.f;
  float dm = g + gk * mean[r];
  float dl = gk * 0.5f * (__expf(l) - 1.f);
  if (eps) dl += g * eps[i] * 0.5f * __expf(0.5f * l);
  if (raw < -30.f || raw > 20.f) dl = 0.f;  // clamp passes no gradient outside its range
  dmom[(size_t)b * 2 * per + r] = dm;
  dmom[(size_t)b * 2 * per + per + r] = dl;
}

// one block per packed channel; training: batch statistics + running-stat update
__global__ __launch_bounds__(256) void latent_bn_fwd_kernel(const float* __restrict__ z,
                                                            float* __restrict__ zp,
                                                            float* __restrict__ run_mean,
                                                            float* __restrict__ run_var,
                                                            float* __restrict__ save,  // [4Z][2]
                                                            int B, int Z, int H, int W, int training,
                                                            float eps, float momentum) {
  const int pc = blockIdx.x;
  const int c = pc >> 2, pi = (pc >> 1) & 1, pj = pc & 1;
  const int h2 = H / 2, w2 = W / 2, per = h2 * w2, n = B * per;
  float mean, invstd;
  __shared__ double red[8];
  __shared__ float bc[2];
  if (training) {
    float s = 0.f, q = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
      int b = e / per, r = e % per, i = r / w2, j = r % w2;
      float v = z[(((size_t)b * Z + c) * H + 2 * i + pi) * W + 2 * j + pj];
      s += v;
      q += v * v;
    }
    double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = ds; red[4 + w] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double S = red[0] + red[1] + red[2] + red[3], Q = red[4] + red[5] + red[6] + red[7];
      double m = S / n, var = Q / n - m * m;
      if (var < 0) var = 0;
      bc[0] = (float)m;
      bc[1] = (float)(1.0 / sqrt(var + (double)eps));
      double unbiased = n > 1 ? var * n / (n - 1) : var;
      run_mean[pc] = (1.f - momentum) * run_mean[pc] + momentum * (float)m;
      run_var[pc] = (1.f - momentum) * run_var[pc] + momentum * (float)unbiased;
    }
    __syncthreads();
    mean = bc[0];
    invstd = bc[1];
  } else {
    mean = run_mean[pc];
    invstd = 1.f / sqrtf(run_var[pc] + eps);
  }
  if (threadIdx.x == 0 && save) { save[2 * pc] = mean; save[2 * pc + 1] = invstd; }
  for (int e = threadIdx.x; e < n; e += 256) {
    int b = e / per, r = e % per, i = r / w2, j = r % w2;
    float v = z[(((size_t)b * Z + c) * H + 2 * i + pi) * W + 2 * j + pj];
    zp[((size_t)b * 4 * Z + pc) * per + r] = (v - mean) * invstd;
  }
}

// dz (spatial) from dzp (packed); training: full BN backward, eval: dzp * invstd
__global__ __launch_bounds__(256) void latent_bn_bwd_kernel(const float* __restrict__ zp,
                                                            const float* __restrict__ dzp,
                                                            const float* __restrict__ save,
                                                            float* __restrict__ dz, int B, int Z,
                                                            int H, int W, int training) {
  const int pc = blockIdx.x;
  const int c = pc >> 2, pi = (pc >> 1) & 1, pj = pc & 1;
  const int h2 = H / 2, w2 = W / 2, per = h2 * w2, n = B * per;
  const float invstd = save[2 * pc + 1];
  float m1 = 0.f, m2 = 0.f;
  if (training) {
    __shared__ double red[8];
    __shared__ float bc[2];
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
      int b = e / per, r = e % per;
      size_t o = ((size_t)b * 4 * Z + pc) * per + r;
      s1 += dzp[o];
      s2 += dzp[o] * zp[o];
    }
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = d1; red[4 + w] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      bc[0] = (float)((red[0] + red[1] + red[2] + red[3]) / n);
      bc[1] = (float)((red[4] + red[5] + red[6] + red[7]) / n);
    }
    __syncthreads();
    m1 = bc[0];
    m2 = bc[1];
  }
  for (int e = threadIdx.x; e < n; e += 256) {
    int b = e / per, r = e % per, i = r / w2, j = r % w2;
    size_t o = ((size_t)b * 4 * Z + pc) * per + r;
    float g = dzp[o];
    float v = training ? invstd * (g - m1 - zp[o] * m2) : invstd * g;
    dz[(((size_t)b * Z + c) * H + 2 * i + pi) * W + 2 * j + pj] = v;
  }
}

// z (spatial) = zp * sqrt(run_var + eps) + run_mean ; grad: dzp = dz * sqrt(run_var + eps)
__global__ void latent_inv_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                  const float* __restrict__ run_mean,
                                  const float* __restrict__ run_var, int B, int Z, int H, int W,
                                  float eps, int backward) {
  size_t total = (size_t)B * Z * H * W;
  size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= total) return;
  // idx enumerates the packed layout
  const int h2 = H / 2, w2 = W / 2, per = h2 * w2;
  int r = idx % per;
  size_t t = idx / per;
  int pc = t % (4 * Z);
  int b = t / (4 * Z);
  int c = pc >> 2, pi = (pc >> 1) & 1, pj = pc & 1, i = r / w2, j = r % w2;
  size_t sp = (((size_t)b * Z + c) * H + 2 * i + pi) * W + 2 * j + pj;
  float s = sqrtf(run_var[pc] + eps);
  if (backward) dst[idx] = src[sp] * s;
  else dst[sp] = src[idx] * s + run_mean[pc];
}

}  // namespace

extern "C" {

// moments [B][2Z][H][W]; eps [B][Z][H][W] or null (=> mode); z and/or kl may be null
int eovae_posterior_fwd(const float* moments, const float* eps, float* z, float* kl, int B, int Z,
                        int HW, void* stream) {
  EOVAE_CHECK_ARG(moments && B > 0 && Z > 0 && HW > 0, "posterior_fwd: bad argument");
  hipLaunchKernelGGL(posterior_fwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, moments, eps,
                     z, kl, Z * HW);
  EOVAE_CHECK_LAUNCH("eovae_posterior_fwd");
  return EOVAE_OK;
}

int eovae_posterior_bwd(const float* moments, const float* eps, const float* dz, const float* dkl,
                        float* dmoments, int B, int Z, int HW, void* stream) {
  EOVAE_CHECK_ARG(moments && dmoments && B > 0, "posterior_bwd: bad argument");
  size_t total = (size_t)B * Z * HW;
  hipLaunchKernelGGL(posterior_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, moments, eps, dz, dkl, dmoments, Z * HW, total);
  EOVAE_CHECK_LAUNCH("eovae_posterior_bwd");
  return EOVAE_OK;
}

// z [B][Z][H][W] -> zp [B][4Z][H/2][W/2]; save[4Z][2] = (mean, invstd) used
int eovae_latent_bn_fwd(const float* z, float* zp, float* run_mean, float* run_var, float* save,
                        int B, int Z, int H, int W, int training, float eps, float momentum,
                        void* stream) {
  EOVAE_CHECK_ARG(z && zp && run_mean && run_var && save, "latent_bn_fwd: null pointer");
  EOVAE_CHECK_ARG(B > 0 && Z > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0,
                  "latent_bn_fwd: latent %dx%d must be even", H, W);
  hipLaunchKernelGGL(latent_bn_fwd_kernel, dim3(4 * Z), dim3(256), 0, (hipStream_t)stream, z, zp,
                     run_mean, run_var, save, B, Z, H, W, training, eps, momentum);
  EOVAE_CHECK_LAUNCH("eovae_latent_bn_fwd");
  return EOVAE_OK;
}

int eovae_latent_bn_bwd(const float* zp, const float* dzp, const float* save, float* dz, int B,
                        int Z, int H, int W, int training, void* stream) {
  EOVAE_CHECK_ARG(zp && dzp && save && dz, "latent_bn_bwd: null pointer");
  hipLaunchKernelGGL(latent_bn_bwd_kernel, dim3(4 * Z), dim3(256), 0, (hipStream_t)stream, zp, dzp,
                     save, dz, B, Z, H, W, training);
  EOVAE_CHECK_LAUNCH("eovae_latent_bn_bwd");
  return EOVAE_OK;
}

// backward = 0: zp (packed, normalised) -> z (spatial);  backward = 1: dz (spatial) -> dzp
int eovae_latent_inv(const float* src, float* dst, const float* run_mean, const float* run_var,
                     int B, int Z, int H, int W, float eps, int backward, void* stream) {
  EOVAE_CHECK_ARG(src && dst && run_mean && run_var, "latent_inv: null pointer");
  EOVAE_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "latent_inv: latent %dx%d must be even", H, W);
  size_t total = (size_t)B * Z * H * W;
  hipLaunchKernelGGL(latent_inv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, run_mean, run_var, B, Z, H, W, eps, backward);
  EOVAE_CHECK_LAUNCH("eovae_latent_inv");
  return EOVAE_OK;
}

}  // extern "C"
