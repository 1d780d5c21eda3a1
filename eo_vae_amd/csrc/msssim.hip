// MS-SSIM loss term of EOConsistencyLoss (consistency_loss.py:24-37 -> torchmetrics
// MultiScaleStructuralSimilarityIndexMeasure(data_range=6.0, betas=5, gaussian sigma 1.5 => 11 taps,
// normalize='relu')), forward and backward.
//
// Per scale the torchmetrics algorithm reflect-pads by 5, filters the five maps
// (x, y, x^2, y^2, xy) with the 11x11 Gaussian and then CROPS 5 pixels per side before the
// per-image mean -- so only outputs whose window lies entirely inside the image are ever used:
// the reflect padding never reaches the result and the kernel filters the unpadded tile.
//
// "Gaussian pyramid" kernels: 16x16 output tiles, 26x26 input tile in LDS, separable 11-tap
// passes through LDS, wave64 shuffle reductions of the per-tile sums, fp64 atomics per image.
// Backward: adjoint maps dF/d(mu_x), dF/d(E[x^2]), dF/d(E[xy]) per scale, transposed (zero-padded)
// separable filter back to the pixel grid, coarse-to-fine accumulation through the 2x2 avg-pool.
// Memory-bound, tiny next to the convolutions (~10*C*262 KB / image).
#include "eovae_common.h"

namespace {

constexpr int TS = 16, R = 5, TIN = TS + 2 * R, NT = 2 * R + 1, MAXS = 8;

struct Taps { float g[NT]; };
struct Counts { double c[MAXS]; };

__global__ void pool2_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes,
                             int H, int W) {  // H, W = output size
  size_t total = planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int x = i % W;
    size_t t = i / W;
    int y = t % H;
    size_t pl = t / H;
    const float* s = src + (pl * 2 * H + 2 * y) * (2 * (size_t)W) + 2 * x;
    dst[i] = 0.25f * ((s[0] + s[1]) + (s[2 * W] + s[2 * W + 1]));
  }
}

// ADJ = false: accumulate per-image sums of cs and ssim over the valid (cropped) domain
// ADJ = true : write the three adjoint maps (zero outside the valid domain)
template <bool ADJ>
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y, int C, int H,
                                                        int W, Taps taps, float c1, float c2,
                                                        float* __restrict__ parts,      // ADJ = false: [plane][tile][2] block sums
                                                        const float* __restrict__ coef,  // [B]
                                                        const float* __restrict__ gout, int use_ssim,
                                                        float* __restrict__ a1, float* __restrict__ a2,
                                                        float* __restrict__ a3) {
  __shared__ float xs[TIN][TIN + 1], ys[TIN][TIN + 1];
  __shared__ float hx[5][TIN][TS + 1];
  const int tiles_x = (W + TS - 1) / TS;
  const int oy = (blockIdx.x / tiles_x) * TS, ox = (blockIdx.x % tiles_x) * TS;
  const int plane = blockIdx.y, b = plane / C;
  const float* xp = x + (size_t)plane * H * W;
  const float* yp = y + (size_t)plane * H * W;
  const int tid = threadIdx.x;
  for (int e = tid; e < TIN * TIN; e += 256) {
    int r = e / TIN, c = e % TIN;
    int iy = oy + r - R, ix = ox + c - R;
    bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    xs[r][c] = ok ? xp[(size_t)iy * W + ix] : 0.f;
    ys[r][c] = ok ? yp[(size_t)iy * W + ix] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < TIN * TS; e += 256) {
    int r = e / TS, c = e % TS;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float g = taps.g[t], xv = xs[r][c + t], yv = ys[r][c + t];
      s0 += g * xv; s1 += g * yv; s2 += g * xv * xv; s3 += g * yv * yv; s4 += g * xv * yv;
    }
    hx[0][r][c] = s0; hx[1][r][c] = s1; hx[2][r][c] = s2; hx[3][r][c] = s3; hx[4][r][c] = s4;
  }
  __syncthreads();
  const int ly = tid / TS, lx = tid % TS;
  float mx = 0, my = 0, exx = 0, eyy = 0, exy = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float g = taps.g[t];
    mx += g * hx[0][ly + t][lx]; my += g * hx[1][ly + t][lx]; exx += g * hx[2][ly + t][lx];
    eyy += g * hx[3][ly + t][lx]; exy += g * hx[4][ly + t][lx];
  }
  const int py = oy + ly, px = ox + lx;
  const bool inimg = py < H && px < W;
  const bool valid = py >= R && py < H - R && px >= R && px < W - R;
  const float vxx = exx - mx * mx, vyy = eyy - my * my;
  const float sxx = fmaxf(vxx, 0.f), syy = fmaxf(vyy, 0.f), sxy = exy - mx * my;
  const float U = 2.f * sxy + c2, Lw = sxx + syy + c2;
  const float cs = U / Lw;
  const float Nl = 2.f * mx * my + c1, Dl = mx * mx + my * my + c1;
  const float lum = Nl / Dl;
  if (!ADJ) {
    float v0 = valid ? cs : 0.f, v1 = valid ? lum * cs : 0.f;
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    __shared__ float red[8];
    if ((tid & 63) == 0) { red[tid >> 6] = v0; red[4 + (tid >> 6)] = v1; }
    __syncthreads();
    if (tid == 0) {
      // one partial pair per block, folded per image in a fixed order by ssim_fold_kernel (round 6: fp64 atomics of 49 k
      // blocks on 2 B addresses serialised -- the scale-0 launch ran at 125 GB/s, profiles/r06a_cfg3_kernels.json)
      float* o = parts + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
      o[0] = (red[0] + red[1]) + (red[2] + red[3]);
      o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  } else if (inimg) {
    float o1 = 0.f, o2 = 0.f, o3 = 0.f;
    if (valid) {
      const float w = coef[b] * gout[0];
      const float indx = vxx >= 0.f ? 1.f : 0.f;
      const float dcs_dexy = 2.f / Lw;
      const float dcs_dexx = -U / (Lw * Lw) * indx;
      const float dcs_dmx = -2.f * my / Lw + 2.f * mx * U * indx / (Lw * Lw);
      if (use_ssim) {
        const float dl_dmx = (2.f * my * Dl - Nl * 2.f * mx) / (Dl * Dl);
        o1 = w * (dl_dmx * cs + lum * dcs_dmx);
        o2 = w * lum * dcs_dexx;
        o3 = w * lum * dcs_dexy;
      } else {
        o1 = w * dcs_dmx; o2 = w * dcs_dexx; o3 = w * dcs_dexy;
      }
    }
    const size_t o = (size_t)plane * H * W + (size_t)py * W + px;
    a1[o] = o1; a2[o] = o2; a3[o] = o3;
  }
}

// dx(q) = G*A1 + 2 x(q) G*A2 + y(q) G*A3  (+ coarse(q/2)/4), G* = zero-padded 11x11 Gaussian
__global__ __launch_bounds__(256) void ssim_grad_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y,
                                                        const float* __restrict__ a1,
                                                        const float* __restrict__ a2,
                                                        const float* __restrict__ a3,
                                                        const float* __restrict__ coarse,
                                                        float* __restrict__ dx, int H, int W,
                                                        Taps taps) {
  __shared__ float as[3][TIN][TIN + 1];
  __shared__ float hs[3][TIN][TS + 1];
  const int tiles_x = (W + TS - 1) / TS;
  const int oy = (blockIdx.x / tiles_x) * TS, ox = (blockIdx.x % tiles_x) * TS;
  const size_t pbase = (size_t)blockIdx.y * H * W;
  const int tid = threadIdx.x;
  for (int e = tid; e < TIN * TIN; e += 256) {
    int r = e / TIN, c = e % TIN;
    int iy = oy + r - R, ix = ox + c - R;
    bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    size_t o = pbase + (size_t)iy * W + ix;
    as[0][r][c] = ok ? a1[o] : 0.f;
    as[1][r][c] = ok ? a2[o] : 0.f;
    as[2][r][c] = ok ? a3[o] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < TIN * TS; e += 256) {
    int r = e / TS, c = e % TS;
    float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float g = taps.g[t];
      s0 += g * as[0][r][c + t]; s1 += g * as[1][r][c + t]; s2 += g * as[2][r][c + t];
    }
    hs[0][r][c] = s0; hs[1][r][c] = s1; hs[2][r][c] = s2;
  }
  __syncthreads();
  const int ly = tid / TS, lx = tid % TS;
  const int py = oy + ly, px = ox + lx;
  if (py >= H || px >= W) return;
  float g1 = 0, g2 = 0, g3 = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float g = taps.g[t];
    g1 += g * hs[0][ly + t][lx]; g2 += g * hs[1][ly + t][lx]; g3 += g * hs[2][ly + t][lx];
  }
  const size_t o = pbase + (size_t)py * W + px;
  float v = g1 + 2.f * x[o] * g2 + y[o] * g3;
  if (coarse) v += 0.25f * coarse[(size_t)blockIdx.y * (H / 2) * (W / 2) + (size_t)(py >> 1) * (W / 2) + (px >> 1)];
  dx[o] = v;
}

// sums[(s B + b)][v] = sum over image b's C planes x tiles of the block partials of scale s, fp64, fixed order
// (bit-reproducible).  grid (B, S), 256 threads.
struct FoldArgs { size_t off[MAXS]; int tiles[MAXS]; };
__global__ __launch_bounds__(256) void ssim_fold_kernel(const float* __restrict__ parts, double* __restrict__ sums, int B, int C,
                                                        FoldArgs fa) {
  const int b = blockIdx.x, s = blockIdx.y;
  const size_t n = (size_t)C * fa.tiles[s];
  const float* p = parts + fa.off[s] + (size_t)b * n * 2;
  double a0 = 0, a1 = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) { a0 += (double)p[2 * i]; a1 += (double)p[2 * i + 1]; }
  a0 = wave_sum_d(a0);
  a1 = wave_sum_d(a1);
  __shared__ double red[8];
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a0; red[4 + (threadIdx.x >> 6)] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[((size_t)s * B + b) * 2] = (red[0] + red[1]) + (red[2] + red[3]);
    sums[((size_t)s * B + b) * 2 + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// per-image factors -> value and the backward coefficients d(out)/d(mean_s,b) / count_s
__global__ void msssim_finalize_kernel(const double* __restrict__ sums, float* __restrict__ coef,
                                       float* __restrict__ out, int B, int S,
                                       const float* __restrict__ betas, Counts cnt) {
  const double* counts = cnt.c;
  double acc = 0;
  for (int b = threadIdx.x; b < B; b += 64) {
    double v[MAXS];
    bool pos = true;
    for (int s = 0; s < S; ++s) {
      double m = sums[((size_t)s * B + b) * 2 + (s == S - 1 ? 1 : 0)] / counts[s];
      v[s] = m > 0 ? m : 0;  // normalize='relu'
      pos = pos && (m > 0);
    }
    double prod = 1;
    for (int s = 0; s < S; ++s) prod *= pow(v[s], (double)betas[s]);
    acc += prod;
    for (int s = 0; s < S; ++s) {
      double d = 0;
      if (pos) d = prod * (double)betas[s] / v[s];  // beta * v^(beta-1) * prod_{t!=s} v_t^beta_t
      coef[(size_t)s * B + b] = (float)(d / ((double)B * counts[s]));
    }
  }
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) out[0] = (float)(acc / B);
}

struct Layout {
  size_t px[MAXS], py[MAXS];  // float offsets of pyramid levels >= 1
  size_t sums, coef, amaps, dlev0, dlev1, total_bytes;
  size_t parts;   // (the forward's block partials live in the adjoint-map region: the backward overwrites it after the fold)
  int hs[MAXS], wsz[MAXS];
};

Layout make_layout(int B, int C, int H, int W, int S) {
  Layout L;
  size_t off = 0;
  int h = H, w = W;
  for (int s = 0; s < S; ++s) {
    L.hs[s] = h; L.wsz[s] = w;
    if (s > 0) {
      size_t n = (size_t)B * C * h * w;
      L.px[s] = off; off += n;
      L.py[s] = off; off += n;
    }
    h /= 2; w /= 2;
  }
  off = (off + 3) / 4 * 4;
  size_t bytes = off * sizeof(float);
  L.sums = bytes; bytes += (size_t)S * B * 2 * sizeof(double);
  L.coef = bytes; bytes += (size_t)S * B * sizeof(float);
  bytes = (bytes + 15) / 16 * 16;
  L.amaps = bytes; bytes += 3 * (size_t)B * C * H * W * sizeof(float);
  L.parts = L.amaps;   // <= 2 floats per 16 x 16 tile and scale: far below the three full-resolution maps
  size_t lev1 = (size_t)B * C * (H / 2) * (W / 2) * sizeof(float);
  L.dlev0 = bytes; bytes += lev1;
  L.dlev1 = bytes; bytes += lev1;
  L.total_bytes = bytes;
  return L;
}

Taps make_taps(float sigma) {
  Taps t;
  float sum = 0.f;
  for (int i = 0; i < NT; ++i) {
    float d = (float)(i - R);
    t.g[i] = expf(-0.5f * (d / sigma) * (d / sigma));
    sum += t.g[i];
  }
  for (int i = 0; i < NT; ++i) t.g[i] /= sum;
  return t;
}

}  // namespace

extern "C" {

size_t eovae_msssim_workspace_bytes(int B, int C, int H, int W, int scales) {
  if (scales < 1 || scales > MAXS) return 0;
  return make_layout(B, C, H, W, scales).total_bytes;
}

int eovae_msssim_fwd(const float* pred, const float* target, float* out, int B, int C, int H, int W,
                     int scales, const float* betas, float data_range, void* ws, void* stream) {
  EOVAE_CHECK_ARG(pred && target && out && betas && ws, "msssim_fwd: null pointer");
  EOVAE_CHECK_ARG(scales >= 1 && scales <= MAXS, "msssim_fwd: %d scales unsupported", scales);
  EOVAE_CHECK_ARG((H >> (scales - 1)) > 2 * R && (W >> (scales - 1)) > 2 * R,
                  "msssim_fwd: image %dx%d too small for %d scales of an 11-tap window", H, W, scales);
  EOVAE_CHECK_ARG(H % (1 << (scales - 1)) == 0 && W % (1 << (scales - 1)) == 0,
                  "msssim_fwd: image size must be divisible by 2^(scales-1)");
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(B, C, H, W, scales);
  char* base = (char*)ws;
  float* fbase = (float*)ws;
  double* sums = (double*)(base + L.sums);
  float* coef = (float*)(base + L.coef);
  float* parts = (float*)(base + L.parts);
  FoldArgs fa;
  size_t poff = 0;
  for (int s = 0; s < MAXS; ++s) {
    fa.tiles[s] = s < scales ? cdiv(L.hs[s], TS) * cdiv(L.wsz[s], TS) : 0;
    fa.off[s] = poff;
    poff += (size_t)B * C * fa.tiles[s] * 2;
  }
  Counts counts;
  for (int s = 0; s < MAXS; ++s)
    counts.c[s] = s < scales ? (double)C * (L.hs[s] - 2 * R) * (L.wsz[s] - 2 * R) : 1.0;
  const Taps taps = make_taps(1.5f);
  const float c1 = (0.01f * data_range) * (0.01f * data_range);
  const float c2 = (0.03f * data_range) * (0.03f * data_range);
  const float *x = pred, *y = target;
  for (int s = 0; s < scales; ++s) {
    const int h = L.hs[s], w = L.wsz[s];
    dim3 grid(cdiv(h, TS) * cdiv(w, TS), B * C);
    hipLaunchKernelGGL(ssim_tile_kernel<false>, grid, dim3(256), 0, st, x, y, C, h, w, taps, c1, c2,
                       parts + fa.off[s], nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    if (s + 1 < scales) {
      float* nx = fbase + L.px[s + 1];
      float* ny = fbase + L.py[s + 1];
      size_t n = (size_t)B * C * (h / 2) * (w / 2);
      int blocks = (int)((n + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(pool2_kernel, dim3(blocks), dim3(256), 0, st, x, nx, (size_t)B * C, h / 2, w / 2);
      hipLaunchKernelGGL(pool2_kernel, dim3(blocks), dim3(256), 0, st, y, ny, (size_t)B * C, h / 2, w / 2);
      x = nx; y = ny;
    }
  }
  hipLaunchKernelGGL(ssim_fold_kernel, dim3(B, scales), dim3(256), 0, st, parts, sums, B, C, fa);
  hipLaunchKernelGGL(msssim_finalize_kernel, dim3(1), dim3(64), 0, st, sums, coef, out, B, scales,
                     betas, counts);
  EOVAE_CHECK_LAUNCH("eovae_msssim_fwd");
  return EOVAE_OK;
}

int eovae_msssim_bwd(const float* pred, const float* target, const float* gout, float* dpred, int B,
                     int C, int H, int W, int scales, const float* betas, float data_range, void* ws,
                     void* stream) {
  EOVAE_CHECK_ARG(pred && target && gout && dpred && ws, "msssim_bwd: null pointer");
  EOVAE_CHECK_ARG(scales >= 1 && scales <= MAXS, "msssim_bwd: %d scales unsupported", scales);
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(B, C, H, W, scales);
  char* base = (char*)ws;
  float* fbase = (float*)ws;
  const float* coef = (const float*)(base + L.coef);
  float* a1 = (float*)(base + L.amaps);
  float* a2 = a1 + (size_t)B * C * H * W;
  float* a3 = a2 + (size_t)B * C * H * W;
  float* dlev[2] = {(float*)(base + L.dlev0), (float*)(base + L.dlev1)};
  const Taps taps = make_taps(1.5f);
  const float c1 = (0.01f * data_range) * (0.01f * data_range);
  const float c2 = (0.03f * data_range) * (0.03f * data_range);
  const float* coarse = nullptr;
  for (int s = scales - 1; s >= 0; --s) {
    const int h = L.hs[s], w = L.wsz[s];
    const float* x = s == 0 ? pred : fbase + L.px[s];
    const float* y = s == 0 ? target : fbase + L.py[s];
    dim3 grid(cdiv(h, TS) * cdiv(w, TS), B * C);
    hipLaunchKernelGGL(ssim_tile_kernel<true>, grid, dim3(256), 0, st, x, y, C, h, w, taps, c1, c2,
                       nullptr, coef + (size_t)s * B, gout, s == scales - 1 ? 1 : 0, a1, a2, a3);
    float* dst = s == 0 ? dpred : dlev[s & 1];
    hipLaunchKernelGGL(ssim_grad_kernel, grid, dim3(256), 0, st, x, y, a1, a2, a3, coarse, dst, h, w,
                       taps);
    coarse = dst;
  }
  EOVAE_CHECK_LAUNCH("eovae_msssim_bwd");
  return EOVAE_OK;
}

}  // extern "C"
