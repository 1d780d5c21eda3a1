// EQ-VAE latent transforms of the train step:
//   forward      z -> rot90(bilinear(z, size), k)        new_autoencoder.py:456-460, 520-531
//                (= eovae_batch_prep without z-score/clamp: same gather, batchprep.hip)
//   this file    the ADJOINT of that gather (gradient w.r.t. z), and the no-grad target
//                rot90(F.interpolate(images, size, mode='area'), k)   new_autoencoder.py:612-617
// Rotation argument `rot` = quarter turns in the torch.rot90(dims=[-2,-1]) sense, as batchprep.hip;
// the reference's rot90(k, dims=[-1,-2]) is rot = (4 - k) % 4 (the host wrapper converts).
// Tiny tensors (latents [B,32,32,32], images [B,C,256,256]): plain gather kernels, deterministic
// (no atomics), HBM/latency bound.
#include "eovae_common.h"

namespace {

struct RotMap {
  int Ht, Wt, rot;
  // position of resized-grid pixel (y, x) in the rotated destination [Hd][Wd]
  __device__ __forceinline__ size_t dst_index(int y, int x) const {
    int yd = y, xd = x, Wd = Wt;
    if (rot == 1) { yd = Wt - 1 - x; xd = y; Wd = Ht; }
    else if (rot == 2) { yd = Ht - 1 - y; xd = Wt - 1 - x; }
    else if (rot == 3) { yd = x; xd = Ht - 1 - y; Wd = Ht; }
    return (size_t)yd * Wd + xd;
  }
};

// bilinear source taps of target coordinate t (align_corners = False), identical float
// arithmetic to batch_prep_kernel
__device__ __forceinline__ void taps(int t, float scale, int S, int& i0, int& i1, float& l) {
  const float f = fmaxf(((float)t + 0.5f) * scale - 0.5f, 0.f);
  i0 = (int)f;
  if (i0 > S - 1) i0 = S - 1;
  i1 = i0 + (i0 < S - 1);
  l = f - (float)i0;
}

// weight with which target coordinate t reads source coordinate s
__device__ __forceinline__ float tap_weight(int t, int s, float scale, int S) {
  int i0, i1;
  float l;
  taps(t, scale, S, i0, i1, l);
  return (i0 == s ? 1.f - l : 0.f) + (i1 == s ? l : 0.f);
}

__global__ __launch_bounds__(256) void resize_rot_bwd_kernel(const float* __restrict__ dy,
                                                             float* __restrict__ dx, int planes,
                                                             int Hs, int Ws, RotMap m) {
  const size_t total = (size_t)planes * Hs * Ws;
  const float sy = (float)Hs / (float)m.Ht, sx = (float)Ws / (float)m.Wt;
  const float ry = (float)m.Ht / (float)Hs, rx = (float)m.Wt / (float)Ws;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xs = i % Ws;
    const size_t t = i / Ws;
    const int ys = t % Hs;
    const size_t plane = t / Hs;
    // conservative candidate window: targets whose source coordinate lies in (s - 1, s + 1)
    int ylo = (int)floorf(((float)ys - 0.5f) * ry - 0.5f) - 1, yhi = (int)ceilf(((float)ys + 1.5f) * ry - 0.5f) + 1;
    int xlo = (int)floorf(((float)xs - 0.5f) * rx - 0.5f) - 1, xhi = (int)ceilf(((float)xs + 1.5f) * rx - 0.5f) + 1;
    if (ys == 0) ylo = 0;          // the clamp at 0 folds every target above the first row onto it
    if (xs == 0) xlo = 0;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > m.Ht - 1 ? m.Ht - 1 : yhi; xhi = xhi > m.Wt - 1 ? m.Wt - 1 : xhi;
    if (ys == Hs - 1) yhi = m.Ht - 1;
    if (xs == Ws - 1) xhi = m.Wt - 1;
    const float* g = dy + plane * (size_t)m.Ht * m.Wt;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      const float wy = tap_weight(y, ys, sy, Hs);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int x = xlo; x <= xhi; ++x) {
        const float wx = tap_weight(x, xs, sx, Ws);
        if (wx != 0.f) row += wx * g[m.dst_index(y, x)];
      }
      acc += wy * row;
    }
    dx[i] = acc;
  }
}

// adaptive average pooling windows: [floor(t*S/T), ceil((t+1)*S/T))
__global__ __launch_bounds__(256) void area_rot_kernel(const float* __restrict__ src,
                                                       float* __restrict__ dst, int planes, int Hs,
                                                       int Ws, RotMap m) {
  const size_t per = (size_t)m.Ht * m.Wt, total = (size_t)planes * per;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = i % m.Wt;
    const size_t t = i / m.Wt;
    const int y = t % m.Ht;
    const size_t plane = t / m.Ht;
    const int y0 = (int)(((long long)y * Hs) / m.Ht), y1 = (int)((((long long)y + 1) * Hs + m.Ht - 1) / m.Ht);
    const int x0 = (int)(((long long)x * Ws) / m.Wt), x1 = (int)((((long long)x + 1) * Ws + m.Wt - 1) / m.Wt);
    const float* p = src + plane * (size_t)Hs * Ws;
    float s = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) s += p[(size_t)yy * Ws + xx];
    dst[plane * per + m.dst_index(y, x)] = s / (float)((y1 - y0) * (x1 - x0));
  }
}

int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : b);
}

}  // namespace

extern "C" {

// dx[planes][Hs][Ws] = adjoint of {bilinear resize to (Ht, Wt), then rot quarter turns} applied
// to dy[planes][Hd][Wd] ((Hd, Wd) = (Ht, Wt), swapped when rot is odd)
int eovae_resize_rot_bwd(const float* dy, float* dx, int planes, int Hs, int Ws, int Ht, int Wt,
                         int rot, void* stream) {
  EOVAE_CHECK_ARG(dy && dx && planes > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0,
                  "resize_rot_bwd: bad argument");
  EOVAE_CHECK_ARG(rot >= 0 && rot <= 3, "resize_rot_bwd: rot %d", rot);
  RotMap m{Ht, Wt, rot};
  hipLaunchKernelGGL(resize_rot_bwd_kernel, dim3(grid_for((size_t)planes * Hs * Ws)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, planes, Hs, Ws, m);
  EOVAE_CHECK_LAUNCH("eovae_resize_rot_bwd");
  return EOVAE_OK;
}

// dst[planes][Hd][Wd] = rot quarter turns of F.interpolate(src[planes][Hs][Ws], (Ht, Wt), 'area')
int eovae_area_rot(const float* src, float* dst, int planes, int Hs, int Ws, int Ht, int Wt, int rot,
                   void* stream) {
  EOVAE_CHECK_ARG(src && dst && planes > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0,
                  "area_rot: bad argument");
  EOVAE_CHECK_ARG(rot >= 0 && rot <= 3, "area_rot: rot %d", rot);
  RotMap m{Ht, Wt, rot};
  hipLaunchKernelGGL(area_rot_kernel, dim3(grid_for((size_t)planes * Ht * Wt)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, planes, Hs, Ws, m);
  EOVAE_CHECK_LAUNCH("eovae_area_rot");
  return EOVAE_OK;
}

}  // extern "C"
