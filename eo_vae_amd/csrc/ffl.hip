// Focal frequency loss (eo_vae/models/modules/ffl.py:17-102) with the flags EOConsistencyLoss
// builds it with (consistency_loss.py:388-395): ave_spectrum = False, batch_matrix = True,
// log_matrix = True, loss_weight = 1; alpha and patch_factor are arguments.  SURVEY.md 8f N4.
//
//   patches  y[b][i*pf+j][c] = x[b][c][i*ph:(i+1)*ph][j*pw:(j+1)*pw]            (ffl.py:41-50)
//   spectrum Y = F_h . y . F_w   (2-D DFT, norm='ortho')                        (ffl.py:53)
//   weight   w = clamp(log1p(sqrt(|Yp - Yt|^2 + 1e-8)^alpha) / max_over_batch, 0, 1), detached
//   loss     mean(w * |Yp - Yt|^2)
//
// MI355X form: the DFT is linear, so ONE transform of the difference E = pred - target is needed;
// patches are at most 128 x 128, where a DFT is two dense fp32 GEMMs against the (cos, -sin)
// matrices -- they run on the exact-fp32 MFMA GEMM (eovae_gemm_f32) instead of an FFT library:
//   stage 1   T = E . [Fr_w | Fi_w]                     one GEMM, M = planes*ph, N = 2pw, K = pw
//   stage 2   Q = [Fr_h ; Fi_h] . T_plane               batched, M = 2ph, N = 2pw, K = ph
//             Yr = Q[u][v] - Q[ph+u][pw+v],  Yi = Q[ph+u][v] + Q[u][pw+v]
// The backward pass is the same two GEMMs applied to conj(dL/dY) (dE = Re(F conj(G) F)):
//   stage 1   S = [Gr | -Gi] . [[Fr_w, Fi_w], [-Fi_w, Fr_w]]        K = 2pw
//   stage 2   Q' = [Fr_h ; Fi_h] . S_plane,  dE = Q'[u][v] - Q'[ph+u][pw+v]
// Non-finite inputs: the reference clamps the two spectra separately (nan_to_num, ffl.py:57);
// here a non-finite input gives a non-finite loss (no silent clamp).
#include "eovae_common.h"

extern "C" int eovae_gemm_f32(const float* A, const float* B, float* C, int batch, int M, int N, int K,
                              long long sam, long long sak, long long sbk, long long sbn, long long scm,
                              long long scn, long long batA, long long batB, long long batC, float alpha,
                              void* stream);

namespace {

// DFT tables in fp32 from double-precision sincospi of the exactly reduced phase
//   kind 0: Fcat  [P][2P]   = [Fr | Fi]
//   kind 1: Fst   [2P][P]   = [Fr ; Fi]
//   kind 2: Fb    [2P][2P]  = [[Fr, Fi], [-Fi, Fr]]
__global__ void dft_table_kernel(float* __restrict__ out, int P, int kind) {
  const int rows = kind == 0 ? P : 2 * P, cols = kind == 1 ? P : 2 * P;
  const double norm = 1.0 / sqrt((double)P);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * cols; i += gridDim.x * 256) {
    const int r = i / cols, c = i - r * cols;
    const int j = r % P, k = c % P, rb = r / P, cb = c / P;
    const double ph = 2.0 * (double)((j * k) % P) / (double)P;
    const double fr = cospi(ph) * norm, fi = -sinpi(ph) * norm;
    double v;
    if (kind == 0) v = cb == 0 ? fr : fi;
    else if (kind == 1) v = rb == 0 ? fr : fi;
    else v = rb == cb ? fr : (rb == 0 ? fi : -fi);
    out[i] = (float)v;
  }
}

struct Geo {
  int B, C, H, W, pf, ph, pw;
  __device__ __forceinline__ size_t image_index(size_t e) const {  // E element -> NCHW element
    const int v = e % pw;
    size_t t = e / pw;
    const int u = t % ph;
    t /= ph;
    const int c = t % C;
    t /= C;
    const int j = t % pf;
    t /= pf;
    const int i = t % pf;
    const size_t b = t / pf;
    return ((b * C + c) * H + (size_t)i * ph + u) * W + (size_t)j * pw + v;
  }
};

__global__ __launch_bounds__(256) void diff_patches_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ target,
                                                           float* __restrict__ E, size_t total, Geo g) {
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t s = g.image_index(e);
    E[e] = pred[s] - target[s];
  }
}

__device__ __forceinline__ void spectrum_at(const float* Q, size_t e, int ph, int pw, float& yr, float& yi) {
  const int v = e % pw;
  const size_t t = e / pw;
  const int u = t % ph;
  const float* q = Q + (t / ph) * (size_t)(4 * ph * pw);
  yr = q[(size_t)u * 2 * pw + v] - q[(size_t)(ph + u) * 2 * pw + pw + v];
  yi = q[(size_t)(ph + u) * 2 * pw + v] + q[(size_t)u * 2 * pw + pw + v];
}

__device__ __forceinline__ float focal_matrix(float d2, float alpha, int log_matrix) {
  float m = sqrtf(d2 + 1e-8f);
  if (alpha != 1.f) m = powf(m, alpha);
  return log_matrix ? log1pf(m) : m;
}

// pass 1: batch maximum of the un-normalised weight (>= 0, so its bit pattern orders like a uint)
__global__ __launch_bounds__(256) void ffl_max_kernel(const float* __restrict__ Q, size_t total, int ph, int pw,
                                                      float alpha, int log_matrix, unsigned* __restrict__ top) {
  float mx = 0.f;
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float yr, yi;
    spectrum_at(Q, e, ph, pw, yr, yi);
    mx = fmaxf(mx, focal_matrix(yr * yr + yi * yi, alpha, log_matrix));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) atomicMax(top, __float_as_uint(mx));
}

// pass 2: loss partials and conj(dL/dY) = [Gr | -Gi] per plane row
__global__ __launch_bounds__(256) void ffl_weight_kernel(const float* __restrict__ Q, size_t total, int ph, int pw,
                                                         float alpha, int log_matrix,
                                                         const unsigned* __restrict__ top, float inv_count,
                                                         float* __restrict__ G, double* __restrict__ part) {
  float t = __uint_as_float(top[0]);
  if (!(isfinite(t) && t > 0.f)) t = 1.f;   // ffl.py:81-85
  double acc = 0.0;
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float yr, yi;
    spectrum_at(Q, e, ph, pw, yr, yi);
    const float d2 = yr * yr + yi * yi;
    const float w = fminf(fmaxf(focal_matrix(d2, alpha, log_matrix) / t, 0.f), 1.f);
    acc += (double)(w * d2);
    const size_t row = e / pw;
    const int v = e % pw;
    G[row * 2 * pw + v] = 2.f * w * yr * inv_count;
    G[row * 2 * pw + pw + v] = -2.f * w * yi * inv_count;
  }
  __shared__ double red[4];
  const double d = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void ffl_finalize_kernel(const double* __restrict__ part, int nparts, double inv_n,
                                    float* __restrict__ out) {
  double s = 0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}

// dpred = gout * Re-part of the inverse-side transform, scattered back from patch order
__global__ __launch_bounds__(256) void ffl_scatter_kernel(const float* __restrict__ Q, const float* __restrict__ gout,
                                                          float* __restrict__ dpred, size_t total, Geo g) {
  const float gs = gout[0];
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    float yr, yi;
    spectrum_at(Q, e, g.ph, g.pw, yr, yi);
    dpred[g.image_index(e)] = gs * yr;
  }
}

__global__ void zero_f32_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}

int grid_for(size_t total, int cap) {
  size_t b = (total + 255) / 256;
  return (int)(b > (size_t)cap ? cap : b);
}

struct Plan {
  Geo g;
  size_t planes, nE, nT, nQ, tabw, tabh, tabb;  // element counts
  size_t off_tabw, off_tabh, off_tabb, off_E, off_T, off_Q, off_top, off_part, bytes;
};

constexpr int kParts = 1024;

bool make_plan(int B, int C, int H, int W, int pf, Plan& p) {
  if (B <= 0 || C <= 0 || pf <= 0 || H / pf <= 0 || W / pf <= 0) return false;
  p.g = Geo{B, C, H, W, pf, H / pf, W / pf};
  const size_t ph = p.g.ph, pw = p.g.pw;
  p.planes = (size_t)B * pf * pf * C;
  p.nE = p.planes * ph * pw; p.nT = 2 * p.nE; p.nQ = 4 * p.nE;
  p.tabw = pw * 2 * pw; p.tabh = 2 * ph * ph; p.tabb = 4 * pw * pw;
  size_t o = 0;
  auto take = [&](size_t elems) { size_t at = o; o += (elems * 4 + 255) / 256 * 256; return at; };
  p.off_tabw = take(p.tabw); p.off_tabh = take(p.tabh); p.off_tabb = take(p.tabb);
  p.off_E = take(p.nE); p.off_T = take(p.nT); p.off_Q = take(p.nQ);
  p.off_top = take(64); p.off_part = take(2 * kParts);
  p.bytes = o;
  return true;
}

}  // namespace

extern "C" {

// scratch for either pass; gy (saved between the passes) holds 2 * B*C*(H/pf*pf)*(W/pf*pf) floats
size_t eovae_ffl_workspace_bytes(int B, int C, int H, int W, int patch_factor) {
  Plan p;
  return make_plan(B, C, H, W, patch_factor, p) ? p.bytes : 0;
}

size_t eovae_ffl_gy_elems(int B, int C, int H, int W, int patch_factor) {
  Plan p;
  return make_plan(B, C, H, W, patch_factor, p) ? p.nT : 0;
}

// out[0] = focal frequency loss; gy = conj(dL/dY) for eovae_ffl_bwd
int eovae_ffl_fwd(const float* pred, const float* target, float* out, float* gy, void* ws, int B, int C, int H,
                  int W, int patch_factor, float alpha, int log_matrix, void* stream) {
  Plan p;
  EOVAE_CHECK_ARG(pred && target && out && gy && ws, "ffl_fwd: null pointer");
  EOVAE_CHECK_ARG(make_plan(B, C, H, W, patch_factor, p), "ffl_fwd: bad shape B=%d C=%d H=%d W=%d pf=%d", B, C, H,
                  W, patch_factor);
  EOVAE_CHECK_ARG(p.planes * p.g.ph < (1u << 31), "ffl_fwd: too many patch rows");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w8 = (unsigned char*)ws;
  float *tabw = (float*)(w8 + p.off_tabw), *tabh = (float*)(w8 + p.off_tabh);
  float *E = (float*)(w8 + p.off_E), *T = (float*)(w8 + p.off_T), *Q = (float*)(w8 + p.off_Q);
  unsigned* top = (unsigned*)(w8 + p.off_top);
  double* part = (double*)(w8 + p.off_part);
  const int ph = p.g.ph, pw = p.g.pw;
  hipLaunchKernelGGL(dft_table_kernel, dim3(grid_for(p.tabw, 256)), dim3(256), 0, st, tabw, pw, 0);
  hipLaunchKernelGGL(dft_table_kernel, dim3(grid_for(p.tabh, 256)), dim3(256), 0, st, tabh, ph, 1);
  hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, st, (float*)top, (size_t)64);
  hipLaunchKernelGGL(diff_patches_kernel, dim3(grid_for(p.nE, 8192)), dim3(256), 0, st, pred, target, E, p.nE, p.g);
  EOVAE_CHECK_LAUNCH("eovae_ffl_fwd(prepare)");
  int rc = eovae_gemm_f32(E, tabw, T, 1, (int)(p.planes * ph), 2 * pw, pw, pw, 1, 2 * pw, 1, 2 * pw, 1, 0, 0, 0,
                          1.f, stream);
  if (rc) return rc;
  rc = eovae_gemm_f32(tabh, T, Q, (int)p.planes, 2 * ph, 2 * pw, ph, ph, 1, 2 * pw, 1, 2 * pw, 1, 0,
                      (long long)ph * 2 * pw, (long long)4 * ph * pw, 1.f, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(ffl_max_kernel, dim3(grid_for(p.nE, kParts)), dim3(256), 0, st, Q, p.nE, ph, pw, alpha,
                     log_matrix, top);
  const int blocks = grid_for(p.nE, kParts);
  hipLaunchKernelGGL(ffl_weight_kernel, dim3(blocks), dim3(256), 0, st, Q, p.nE, ph, pw, alpha, log_matrix, top,
                     (float)(1.0 / (double)p.nE), gy, part);
  hipLaunchKernelGGL(ffl_finalize_kernel, dim3(1), dim3(64), 0, st, part, blocks, 1.0 / (double)p.nE, out);
  EOVAE_CHECK_LAUNCH("eovae_ffl_fwd");
  return EOVAE_OK;
}

// dpred[B][C][H][W] = gout[0] * dL/dpred (zero outside the patch grid when H, W are not multiples of pf)
int eovae_ffl_bwd(const float* gy, const float* gout, float* dpred, void* ws, int B, int C, int H, int W,
                  int patch_factor, void* stream) {
  Plan p;
  EOVAE_CHECK_ARG(gy && gout && dpred && ws, "ffl_bwd: null pointer");
  EOVAE_CHECK_ARG(make_plan(B, C, H, W, patch_factor, p), "ffl_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* w8 = (unsigned char*)ws;
  float *tabh = (float*)(w8 + p.off_tabh), *tabb = (float*)(w8 + p.off_tabb);
  float *S = (float*)(w8 + p.off_T), *Q = (float*)(w8 + p.off_Q);
  const int ph = p.g.ph, pw = p.g.pw;
  hipLaunchKernelGGL(dft_table_kernel, dim3(grid_for(p.tabh, 256)), dim3(256), 0, st, tabh, ph, 1);
  hipLaunchKernelGGL(dft_table_kernel, dim3(grid_for(p.tabb, 256)), dim3(256), 0, st, tabb, pw, 2);
  if (H % patch_factor || W % patch_factor)
    hipLaunchKernelGGL(zero_f32_kernel, dim3(grid_for((size_t)B * C * H * W, 8192)), dim3(256), 0, st, dpred,
                       (size_t)B * C * H * W);
  EOVAE_CHECK_LAUNCH("eovae_ffl_bwd(prepare)");
  int rc = eovae_gemm_f32(gy, tabb, S, 1, (int)(p.planes * ph), 2 * pw, 2 * pw, 2 * pw, 1, 2 * pw, 1, 2 * pw, 1, 0,
                          0, 0, 1.f, stream);
  if (rc) return rc;
  rc = eovae_gemm_f32(tabh, S, Q, (int)p.planes, 2 * ph, 2 * pw, ph, ph, 1, 2 * pw, 1, 2 * pw, 1, 0,
                      (long long)ph * 2 * pw, (long long)4 * ph * pw, 1.f, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(ffl_scatter_kernel, dim3(grid_for(p.nE, 8192)), dim3(256), 0, st, Q, gout, dpred, p.nE, p.g);
  EOVAE_CHECK_LAUNCH("eovae_ffl_bwd");
  return EOVAE_OK;
}

}  // extern "C"
