// GPU-side batch preparation = the step BEFORE the hot path, fused into one
// gather kernel so that 8 GPUs are not fed by host-side torch ops:
//   clamp + per-band z-score            terramesh_datamodule.py:130-339 (e.g. Sentinel2L2ANorm :187-194)
//   bilinear resize, align_corners=False terramesh_datamodule.py:476-479
//   D4 augmentation (h-flip, v-flip, rot90 k)  terramesh_datamodule.py:347-369
// in the reference's order: normalise -> resize -> augment.  HBM-bound streaming kernel.
#include "eovae_common.h"

namespace {

struct PrepArgs {
  const float* src;   // [N][C][Hs][Ws] raw digital numbers
  const float* mean;  // [C]
  const float* stdv;  // [C]
  float* dst;         // [N][C][Hd][Wd]   (Hd, Wd AFTER the rotation)
  int N, C, Hs, Ws, Ht, Wt;  // Ht, Wt = resize target (before rotation)
  int hflip, vflip, rot;
  float lo, hi;       // clamp range applied to the raw values
  int do_clamp, do_norm;
};

__device__ __forceinline__ float prep_sample(const PrepArgs& a, const float* plane, int y, int x,
                                             float m, float inv) {
  float v = plane[(size_t)y * a.Ws + x];
  if (a.do_clamp) v = fminf(fmaxf(v, a.lo), a.hi);
  return a.do_norm ? (v - m) * inv : v;
}

__global__ __launch_bounds__(256) void batch_prep_kernel(PrepArgs a) {
  const int Hd = (a.rot & 1) ? a.Wt : a.Ht, Wd = (a.rot & 1) ? a.Ht : a.Wt;
  const size_t total = (size_t)a.N * a.C * Hd * Wd;
  const float sy = (float)a.Hs / (float)a.Ht, sx = (float)a.Ws / (float)a.Wt;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xd = i % Wd;
    size_t t = i / Wd;
    const int yd = t % Hd;
    t /= Hd;
    const int c = t % a.C;
    // undo torch.rot90(k, dims=[-2,-1]): out[i][j] = in[j][W-1-i] for k = 1 (counter-clockwise)
    int y = yd, x = xd;
    if (a.rot == 1) { y = xd; x = a.Wt - 1 - yd; }
    else if (a.rot == 2) { y = a.Ht - 1 - yd; x = a.Wt - 1 - xd; }
    else if (a.rot == 3) { y = a.Ht - 1 - xd; x = yd; }
    if (a.vflip) y = a.Ht - 1 - y;   // flips are applied before the rotation
    if (a.hflip) x = a.Wt - 1 - x;
    const float* plane = a.src + t * (size_t)a.Hs * a.Ws;  // t = n*C + c
    const float m = a.do_norm ? a.mean[c] : 0.f, inv = a.do_norm ? 1.f / a.stdv[c] : 1.f;
    float v;
    if (a.Hs == a.Ht && a.Ws == a.Wt) {
      v = prep_sample(a, plane, y, x, m, inv);
    } else {  // F.interpolate(mode='bilinear', align_corners=False)
      float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.f);
      int y0 = (int)fy, x0 = (int)fx;
      if (y0 > a.Hs - 1) y0 = a.Hs - 1;
      if (x0 > a.Ws - 1) x0 = a.Ws - 1;
      const int y1 = y0 + (y0 < a.Hs - 1), x1 = x0 + (x0 < a.Ws - 1);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float v00 = prep_sample(a, plane, y0, x0, m, inv), v01 = prep_sample(a, plane, y0, x1, m, inv);
      const float v10 = prep_sample(a, plane, y1, x0, m, inv), v11 = prep_sample(a, plane, y1, x1, m, inv);
      v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
    a.dst[i] = v;
  }
}

}  // namespace

// dst[N][C][Hd][Wd] with (Hd, Wd) = (Ht, Wt), swapped when rot is odd.  mean/stdv may be NULL
// (no z-score); lo > hi disables the clamp.
extern "C" int eovae_batch_prep(const float* src, const float* mean, const float* stdv, float* dst,
                                int N, int C, int Hs, int Ws, int Ht, int Wt, float lo, float hi,
                                int hflip, int vflip, int rot, void* stream) {
  EOVAE_CHECK_ARG(src && dst && N > 0 && C > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0,
                  "batch_prep: bad argument");
  EOVAE_CHECK_ARG(rot >= 0 && rot <= 3, "batch_prep: rot %d", rot);
  EOVAE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "batch_prep: mean and std go together");
  PrepArgs a;
  a.src = src; a.mean = mean; a.stdv = stdv; a.dst = dst;
  a.N = N; a.C = C; a.Hs = Hs; a.Ws = Ws; a.Ht = Ht; a.Wt = Wt;
  a.hflip = hflip; a.vflip = vflip; a.rot = rot; a.lo = lo; a.hi = hi;
  a.do_clamp = lo <= hi; a.do_norm = mean != nullptr;
  size_t total = (size_t)N * C * Ht * Wt;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(batch_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  EOVAE_CHECK_LAUNCH("eovae_batch_prep");
  return EOVAE_OK;
}
