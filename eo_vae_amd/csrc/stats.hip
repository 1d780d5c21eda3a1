// Running per-channel latent statistics of the encode_latents.py path
// (RunningStatsButFast.update, encode_latents.py:65-94): batch mean / unbiased variance / min /
// max over (B, H, W) merged into the running state with Chan's parallel formula.
// state[5][C]: mean, var, min, max, count (count replicated per channel).
#include "eovae_common.h"

namespace {

__global__ __launch_bounds__(256) void running_stats_kernel(const float* __restrict__ z,
                                                            float* __restrict__ state, int B, int C,
                                                            int HW) {
  const int c = blockIdx.x;
  const int n = B * HW;
  float s = 0.f, q = 0.f, mn = INFINITY, mx = -INFINITY;
  for (int e = threadIdx.x; e < n; e += 256) {
    int b = e / HW, r = e % HW;
    float v = z[((size_t)b * C + c) * HW + r];
    s += v; q += v * v;
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  __shared__ double rs[4], rq[4];
  __shared__ float rmn[4], rmx[4];
  double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
  mn = -wave_max(-mn);
  mx = wave_max(mx);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { rs[w] = ds; rq[w] = dq; rmn[w] = mn; rmx[w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = rs[0] + rs[1] + rs[2] + rs[3], Q = rq[0] + rq[1] + rq[2] + rq[3];
    double bm = S / n;
    double bv = n > 1 ? (Q - S * bm) / (n - 1) : 0.0;  // torch.var: unbiased
    float bmin = fminf(fminf(rmn[0], rmn[1]), fminf(rmn[2], rmn[3]));
    float bmax = fmaxf(fmaxf(rmx[0], rmx[1]), fmaxf(rmx[2], rmx[3]));
    float mean = state[c], var = state[C + c], count = state[4 * C + c];
    float bc = (float)n;
    float nab = count + bc;
    float delta = (float)bm - mean;
    float m2 = var * count + (float)bv * bc + delta * delta * count * bc / (nab + 1e-8f);
    state[c] = (mean * count + (float)bm * bc) / nab;
    state[C + c] = m2 / nab;
    state[2 * C + c] = fminf(state[2 * C + c], bmin);
    state[3 * C + c] = fmaxf(state[3 * C + c], bmax);
    state[4 * C + c] = nab;
  }
}

}  // namespace

extern "C" int eovae_running_stats_update(const float* z, float* state, int B, int C, int HW,
                                          void* stream) {
  EOVAE_CHECK_ARG(z && state && B > 0 && C > 0 && HW > 0, "running_stats_update: bad argument");
  hipLaunchKernelGGL(running_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, state, B, C,
                     HW);
  EOVAE_CHECK_LAUNCH("eovae_running_stats_update");
  return EOVAE_OK;
}
