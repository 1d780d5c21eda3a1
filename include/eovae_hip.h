/* eovae_hip.h -- C ABI of libeovae_hip.so: the MI355X (gfx950) kernels behind EO-VAE's
 * data-parallel hot path.
 *
 * The reference (nilsleh/eo-vae) is pure Python: its "FFI" for this path is the set of ATen
 * operator calls made by eo_vae/models/{model.py, new_autoencoder.py} and
 * eo_vae/models/modules/{layers.py, dynamic_conv.py, distributions.py, consistency_loss.py}.
 * Each entry point below names the reference call site it replaces (file:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer of the reference adds.
 *
 * Conventions
 *   - return value: 0 = ok, negative = error; eovae_last_error() gives the thread-local text
 *   - all tensors: contiguous NCHW fp32 device memory, allocated and owned by the CALLER
 *     (PyTorch-ROCm); the library never allocates, frees or retains a pointer past the call
 *   - `stream` is a hipStream_t (0 = default); every call only ENQUEUES work on it, never
 *     synchronises, and is safe to capture into a hipGraph
 *   - re-entrant.  Process-wide state the library keeps: the option words below (eovae_set_option), the device
 *     properties queried once (CU count), the one-pass GroupNorm backward's device-side control block (tickets,
 *     epoch, time-out record -- eovae_gn_bwd_*), and the thread-local error string.  No tensor is retained.
 *   - scratch needed by a call is passed in (`ws`); sizes come from *_workspace_bytes()
 */
#ifndef EOVAE_HIP_H
#define EOVAE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* eovae_last_error(void);
int eovae_abi_version(void);
const char* eovae_target_arch(void); /* "gfx950" */
/* bit 0: built with -DEOVAE_PARKED=1, i.e. the measured-and-parked kernels are compiled in (options c8_wgrad_w4 = 1,
 * c8_q16 = 2 on fp32 inputs, c8_w4_ni = 2 reach them; without the bit those values run the default kernels) */
int eovae_build_flags(void);
/* Run-time switches for A/B runs and tests.  Each option starts from its environment variable, read once; afterwards
 * only eovae_set_option changes it.  Returns 0 / the value, or a negative status for an unknown name.
 * Options marked [arithmetic] change the PRODUCTS a kernel forms (every setting is tested against the float64 sum of
 * exactly its own products, and the defaults are the ones the full-size parity pins are quoted for); the others select
 * among kernels that form the same products.  Options (environment variable, default):
 *   "c8_mx"         (EOVAE_C8_MX, 1)          [arithmetic] correction product on block-scaled MX-FP6 (e2m3) operands where the
 *                                             kernel and its input allow it (1.64 MFMA units per product); 0: fp8 (e5m2 / e4m3, 2 units)
 *   "c8_dgrad_mx"   (EOVAE_C8_DGRAD_MX, 1)    [arithmetic] data gradients (fp32 input of the one-wave kernel) quantised to f16 + MX-FP6
 *                                             inside the kernel; 0: f16 + fp8
 *   "c8_wgrad_corr" (EOVAE_C8_WGRAD_CORR, 0)  [arithmetic] row-streaming weight gradients: 0 = ONE product f16(dY) * f16(X) (a leaf of the
 *                                             step: ~3e-4 of a tensor's largest entry, full-size pins unchanged); 1 = f16 product + fp8
 *                                             correction product (the round-4 arithmetic, ~1e-5) -- the precision / time trade-off to flip
 *                                             for optimisers that are sensitive to gradient noise
 *   "c8_q16"        (EOVAE_C8_Q16, 1)         16 x 16 MFMA forms (conv_f16c8_q16.hip, the rows weight gradient): same products, bit-identical
 *                                             weight gradients, higher held clock; 0: the 32 x 32 forms; 2: also fp32 inputs (measured slower)
 *   "c8_up_q16"     (EOVAE_C8_UP_Q16, 1)      [arithmetic] x2-upsample forward and data gradient on MX records of x / dY (csrc/
 *                                             conv_f16c8_q16up.hip, conv_f16c8_q16upg.hip: f16 + MX-FP6 products; the data gradient only where it
 *                                             has a tile per CU); 2: wherever the shape allows; 0: the 8-wave kernel's sub-pixel modes on fp32
 *                                             inputs (f16 + fp8 products)
 *   "c8_pack_gather" (EOVAE_C8_PACK_GATHER, 0) per-layer weight packs by the per-element gather kernel (bit-identical reference of the LDS-staged pack)
 *   "c8_wgrad_rows" (EOVAE_C8_WGRAD_ROWS, 1)  plain 3x3 weight gradients with record input run the row-streaming kernel
 *   "c8_w4"         (EOVAE_C8_W4, -1)         one-wave-per-SIMD conv kernel: -1 where it pays, 0 never, 1 wherever the shape allows
 *   "c8_w4_ni" (4), "c8_w4_stagger" (0), "c8_w4_skew" (-1)   experiment hooks of that kernel (csrc/conv_f16c8_w4.hip)
 *   "c8_wgrad_w4"   (EOVAE_C8_WGRAD_W4, 0)    the parked one-wave weight gradient (csrc/conv_f16c8_wgrad_w4.hip) */
/* Diagnostic (csrc/diag.hip): the shader clock the chip holds under a dense f16-MFMA load.  Every workgroup (512 threads,
 * one per CU) runs `iters` rounds of 16 MFMAs per wave and writes {shader cycles, 100-MHz wall ticks} of its loop to
 * out[2 b], out[2 b + 1]: clock = cycles / ticks x 0.1 GHz.  Touches no product tensor. */
int eovae_clock_probe(unsigned long long* out, float* sink, int blocks, int iters, void* stream);
int eovae_set_option(const char* name, int value);
int eovae_get_option(const char* name);

/* ----------------------------------------------------------------------------------------
 * Convolution = implicit GEMM on the matrix cores (csrc/conv_f32.hip)
 *   mode 0 NORMAL : stride 1, pad (KS-1)/2   nn.Conv2d 3x3/1x1      layers.py:64-66,81-87,123-126
 *                                            model.py:162,165,236,260; F.conv2d dynamic_conv.py:527-533,699-705
 *   mode 1 DOWN   : F.pad(0,1,0,1) + 3x3 stride 2                    layers.py:33-37
 *   mode 2 UP     : F.interpolate(nearest x2) + 3x3 stride 1 pad 1   layers.py:47-50
 *   mode 3 ZINS   : adjoint of DOWN w.r.t. its input (zero-insert x2, window offset -2)
 * Weights are consumed PACKED: wp[tap][KP][MP] (zero padded), produced by eovae_conv_pack_weights
 * from the PyTorch layout w[Cout][Cin][KS][KS]:
 *   flip = 0  forward operand        (k = Cin, m = Cout)
 *   flip = 1  data-gradient operand  (k = Cout, m = Cin, taps rotated 180 degrees)
 * -------------------------------------------------------------------------------------- */
int eovae_conv_packed_dims(int Cout, int Cin, int KS, int mode, int flip, int* KP, int* MP);
int eovae_conv_pack_weights(const float* w, float* wp, int Cout, int Cin, int KS, int mode,
                            int flip, float scale, void* stream);
/* y[N][Cout][Hout][Wout] = conv(x[N][Cin][Hin][Win], wp) * out_scale + bias[Cout] + residual;
 * bias / residual may be NULL.  For a data-gradient call pass (Cin, Cout) swapped together with
 * the flip-packed weights: dx = conv2d_fwd(dy, wp_flip). */
int eovae_conv2d_fwd(const float* x, const float* wp, const float* bias, const float* residual,
                     float* y, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                     int KS, int mode, int KP, int MP, float out_scale, void* stream);
/* Split-bf16 ("bf16x3") variant of the 3x3 stride-1 convolution (csrc/conv_bf16x3.hip): operands
 * carried as hi+lo bf16 pairs, 3 bf16 MFMAs per tile step, fp32 accumulate (~2^-17 per product).
 * precision 3 = bf16x3 (fp32-grade), 1 = plain bf16.  Modes 0 (NORMAL), 2 (UP), 3 (ZINS).
 * Weights are consumed split-packed: [tap][Cin/16][Cout/128][hi|lo][128][16] bf16. */
size_t eovae_conv3x3_bf16_packed_elems(int Cout, int Cin, int flip);
int eovae_conv3x3_bf16_pack(const float* w, void* wp, int Cout, int Cin, int flip, float scale,
                            void* stream);
/* subpix: "nearest x2 + 3x3" (layers.py:47-50) evaluated as four parity kernels with 2x2 support on the
 * LOW-res grid (4/9 of the MFMAs, pixel-shuffle folded into the epilogue / loader):
 *   1 = forward  (x low-res, wp = pack(expand(w), flip 0), Cout %% 128 == 0, y = [N][Cout][2Hin][2Win])
 *   2 = data grad (x = dy high-res [N][Cin][Hin][Win], wp = pack(expand(w), flip 1), y = dx [N][Cout][Hin/2][Win/2]) */
int eovae_conv3x3_bf16_fwd(const float* x, const void* wp, const float* bias, const float* residual,
                           float* y, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                           int mode, int precision, int subpix, void* stream);
/* "f16c8" variant of the same 3x3 stride-1 convolution (csrc/conv_f16c8.hip): x*w = f16(x)*f16(w) on
 * v_mfma_f32_32x32x16_f16 plus ONE block-scaled fp8 product (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 =
 * [x_lo | x_hi] . [w_hi | w_lo]) carrying both first-order corrections: 2 matrix-core units per algorithmic
 * product instead of bf16x3's 3, whole-model error 1.2e-4 vs the reference's fp32 output (bar 1e-3).
 * Replaces the same reference call sites as eovae_conv3x3_bf16_fwd (layers.py:29-50, 70-110 via F.conv2d /
 * its autograd data gradient).  Modes / subpix as eovae_conv3x3_bf16_fwd; the GEMM K channel count must be
 * a multiple of 32.  Weights: 16 KiB per (tap, 32-channel chunk, 128-co block) + a 16-byte header holding the
 * float bits of max|w| (set by the packer, read by the kernel for the fp8 block scales) + 1 KiB of scratch for the
 * packer's partial maxima (no memset / fold launches); w must be 16-byte aligned; always size the buffer with
 * eovae_conv3x3_f16c8_packed_bytes().
 * xamax (nullable): device pointer to the float bits of max|x| (eovae_absmax_bits) for inputs that may sit
 * outside f16's range (gradients): the loader pre-scales by an exact power of two, the epilogue undoes it. */
int eovae_absmax_bits(const float* x, size_t n, unsigned* out_bits, void* stream);
size_t eovae_conv3x3_f16c8_packed_bytes(int Cout, int Cin, int flip);
int eovae_conv3x3_f16c8_pack(const float* w, void* wp, int Cout, int Cin, int flip, float scale,
                             void* stream);
/* both orientations (flip 0 into wp_fwd, flip 1 into wp_bwd) from one amax pass */
int eovae_conv3x3_f16c8_pack_pair(const float* w, void* wp_fwd, void* wp_bwd, int Cout, int Cin, float scale,
                                  void* stream);
/* Every plain 3x3 layer of a model at once (two launches instead of three per layer).  descs: DEVICE array of n
 * 32-byte records { const float* w; void* wp_fwd; void* wp_bwd; int Cout; int Cin; }, buffers sized by
 * eovae_conv3x3_f16c8_packed_bytes(Cout, Cin, 0 / 1); result = n calls of eovae_conv3x3_f16c8_pack_pair. */
int eovae_conv3x3_f16c8_pack_multi(const void* descs, int n, float scale, void* stream);
int eovae_conv3x3_f16c8_fwd(const float* x, const void* wp, const float* bias, const float* residual,
                            float* y, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout,
                            int Hout, int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups,
                            void* stream);
/* as above; out_amax (nullable, pre-zeroed device word) receives the float bits of max|y| from the epilogue (atomicMax):
 * the range pre-scale of the layer that reads y without a reduction pass over y */
int eovae_conv3x3_f16c8_fwd_amax(const float* x, const void* wp, const float* bias, const float* residual,
                            float* y, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout,
                            int Hout, int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups, unsigned* out_amax,
                            void* stream);
/* as above, and the entry that understands xpacked = 2: x = MX operand records written by eovae_gn_silu_fwd_packed_mx (f16
 * plane + MX-FP6 plane, see there); the correction product [x_lo | x_hi].[w_hi | w_lo] then runs on block-scaled e2m3
 * operands (per-32-channel E8M0 scales, OCP MX) at ~1.6 x the fp8 rate.  x6 (nullable): the FP6 plane when it does not sit
 * right behind the f16 plane (NULL: x + N*Cin*H*W*2 bytes).  Same call site as eovae_conv3x3_f16c8_fwd (F.conv2d of
 * layers.py:64-66, 81-83). */
int eovae_conv3x3_f16c8_fwd_mx(const float* x, const void* x6, const void* wp, const float* bias, const float* residual,
                            float* y, const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout,
                            int Hout, int Wout, int mode, int subpix, int xpacked, double* gn_sums, int gn_groups, unsigned* out_amax,
                            void* stream);
/* which kernel the call above launches for these arguments: 1 = conv3x3_f16c8_w4_kernel (csrc/conv_f16c8_w4.hip: one wave
 * per SIMD, 4 x 4 register blocking, 16-row tiles -- plain layers that give every CU two tiles; fp32 inputs are quantised
 * to f16 + MX-FP6 in the kernel: library option c8_dgrad_mx), 2 = conv3x3_f16c8_q16_kernel (csrc/conv_f16c8_q16.hip: the
 * same walk on 16 x 16 MFMAs, every byte by LDS-DMA -- those layers on MX operand records; library option c8_q16), 0 =
 * conv3x3_f16c8_kernel (8 waves, 8-row tiles: sub-pixel / space-to-depth forms, small grids).  No launch. */
int eovae_conv3x3_f16c8_which(int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int mode, int subpix,
                              int xpacked);
/* "nearest x2 + 3x3" forward (layers.py:47-50, Upsample) on MX operand records of the LOW-res input (csrc/conv_f16c8_q16up.hip,
 * round 6): sub-pixel form (4/9 of the MFMAs), 16 x 16 MFMAs, every byte fed by LDS-DMA, both column parities of the output
 * in one wave (16-byte stores).  x is not a GroupNorm output here (the residual stream), so its records are written by
 *   eovae_records_mx_from_f32: x [N][C][HW] fp32 -> rec (N*C*HW*4 bytes: F plane [n][C/8][HW] x 16 B = 8 x f16 | S plane
 *   [n][C/32][piece][HW] x 16 B = MX-FP6 of (x - f16 x) | x), pre-scaled by the power of two that puts *xamax (float bits of
 *   max|x|, device, nullable) at 2^13..2^14; C %% 32 == 0.
 * eovae_conv3x3_f16c8_up_mx: y [N][Cout][2 Hin][2 Win] = conv + bias; wp = eovae_conv3x3_f16c8_pack(eovae_subpix_expand(w))
 *   (flip 0, 4 Cout rows); xamax = the pointer the records were written with (the scale is undone with the result); gn_sums
 *   as eovae_conv3x3_f16c8_fwd, out_amax as eovae_conv3x3_f16c8_fwd_amax.  The stride-2 layer's DATA GRADIENT is the same
 *   operator on its space-to-depth weights (eovae_down_expand(w, dgrad = 1): xrec = records of dY, Cin = the layer's Cout,
 *   Cout = the layer's Cin).  Needs Cin %% 32, Cout %% 128, Hin %% 8, Win %% 32 (..._eligible; library options c8_up_q16,
 *   c8_mx).  The products are those of the plain forward on MX records: f16(x) f16(w) + the MX-FP6 correction product. */
int eovae_conv3x3_f16c8_up_mx_eligible(int N, int Cin, int Hin, int Win, int Cout);
int eovae_records_mx_from_f32(const float* x, const unsigned* xamax, float* rec, int N, int C, int HW, void* stream);
int eovae_conv3x3_f16c8_up_mx(const float* xrec, const void* wp, const float* bias, float* y, const unsigned* xamax, int N, int Cin,
                              int Hin, int Win, int Cout, double* gn_sums, int gn_groups, unsigned* out_amax, void* stream);
/* Data gradient of the same layer (adjoint of layers.py:47-50 w.r.t. its input) on MX operand records of the HIGH-res dY
 * (csrc/conv_f16c8_q16upg.hip): the GEMM K runs over the four parity planes of dY (a lane's DMA offset is twice the low-res
 * offset, the parity a constant on the descriptor base), 4 f16 + 2 correction phases per 32 channels of a parity.
 *   dyrec = eovae_records_mx_from_f32(dY [N][Cout][2 H][2 W], dyamax);  wp = eovae_conv3x3_f16c8_pack(eovae_subpix_expand(w), flip 1);
 *   dx [N][Cin][H][W].  Needs Cout %% 32, Cin %% 128, H %% 16, W %% 32 (..._eligible; options c8_up_q16, c8_mx). */
int eovae_conv3x3_f16c8_up_dgrad_mx_eligible(int N, int Cin, int H, int W, int Cout);
int eovae_conv3x3_f16c8_up_dgrad_mx(const float* dyrec, const void* wp, float* dx, const unsigned* dyamax, int N, int Cin, int H,
                                    int W, int Cout, void* stream);
/* gn_sums (nullable; meant for the forward calls, any subpix form): [N][gn_groups][2] doubles, zero-filled by the caller; the
 * epilogue adds (sum, sum of squares) of the values it stores (bias and residual included) per image and channel
 * group, i.e. the statistics pass of the GroupNorm(gn_groups) that reads y next (layers.py:91-92, 106-108 follow
 * every conv with one) -- hand the array to eovae_gn_silu_fwd[_packed] as `sums` and that call does not read y
 * a second time.  Cout / gn_groups must be a power of two in 4..128. */
/* f16c8 weight gradient (same tiling and sub-pixel forms as eovae_conv3x3_bf16_wgrad; replaces the same
 * autograd call site of F.conv2d, layers.py:29-50, 70-110): per 32 pixels and tap 2 f16 MFMAs + 1 block-scaled
 * fp8 MFMA [dy_lo|dy_hi].[x_hi|x_lo].  dyamax as for the data gradient (NULL = dy used unscaled).  db (nullable,
 * pre-zeroed [Cout]) receives sum(dy) in every form, the sub-pixel one included. */
int eovae_conv3x3_f16c8_wgrad(const float* x, const float* dy, float* dwp, float* db,
                              const unsigned* dyamax, const unsigned* xamax, int N, int Cin, int Hin, int Win,
                              int Cout, int Hout, int Wout, int mode, int subpix, int xpacked, void* stream);
/* The same weight gradient for a PLAIN 3x3 layer (mode 0, subpix 0) written straight into dw[Cout][Cin][3][3] (PyTorch
 * layout): the launch's blocks store partial tiles into ws (eovae_conv3x3_f16c8_wgrad_dw_workspace bytes), a second
 * kernel folds them in a fixed order.  No pre-zeroed accumulator, no float atomics (bit-reproducible), no unpack
 * step.  db (nullable, pre-zeroed [Cout]) is accumulated as before. */
size_t eovae_conv3x3_f16c8_wgrad_dw_workspace(int Cin, int Cout);
int eovae_conv3x3_f16c8_wgrad_dw(const float* x, const float* dy, float* dw, float* db, const unsigned* dyamax,
                                 const unsigned* xamax, int N, int Cin, int Hin, int Win, int Cout, int Hout, int Wout,
                                 int xpacked, void* ws, void* stream);
/* xamax (nullable, fp32 x only): float bits of max|x| for activations that are not GroupNorm outputs (the residual
 * stream in front of Downsample / Upsample layers, layers.py:33-37, 47-50): x is pre-scaled by an exact power of two
 * like dy, so the f16c8 operands make no assumption about its range.  The forward entry takes the same pointer as
 * its `xamax` argument. */
/* Weight gradient of the x2-upsample layer conv3x3(nearest_x2(x)) (layers.py Upsample) in sub-pixel ROW-STREAMING form
 * (csrc/conv_f16c8_wgrad_rows.hip): x fp32 [N][Cin][H][W] is re-written once as operand records (scaled by xamax's power
 * of two when given), each output parity of dy [N][Cout][2H][2W] is a 2x2-tap weight gradient on the low-res grid, a
 * fold sums splits and parities in a fixed order into dw[Cout][Cin][3][3] (written).  db nullable (pre-zeroed, atomics).
 * eligible: Cout % 128 == 0, Cin % 64 == 0, W % 32 == 0; ws >= eovae_conv3x3_f16c8_wgrad_up_workspace bytes. */
int eovae_conv3x3_f16c8_wgrad_up_eligible(int N, int Cin, int H, int W, int Cout);
size_t eovae_conv3x3_f16c8_wgrad_up_workspace(int N, int Cin, int H, int W, int Cout);
int eovae_conv3x3_f16c8_wgrad_up(const float* x, const float* dy, float* dw, float* db, const unsigned* dyamax,
                                 const unsigned* xamax, int N, int Cin, int H, int W, int Cout, void* ws, void* stream);
/* Weight gradient of the stride-2 layer conv3x3_s2(pad(x, (0,1,0,1))) (layers.py Downsample) through the same kernel with
 * the operands' roles swapped (the hi-res x in dy's place, records of the low-res dy in x's place; 9 of the 16 sub-pixel
 * products are taps of this layer, the fold drops the rest): x fp32 [N][Cin][2H][2W], dy fp32 [N][Cout][H][W] ->
 * dw[Cout][Cin][3][3] (written, bit-reproducible).  eligible: Cin % 128 == 0, Cout % 64 == 0, W % 32 == 0 (H, W = output
 * grid).  No bias gradient (eovae_bias_grad). */
int eovae_conv3x3_f16c8_wgrad_down_eligible(int N, int Cin, int H, int W, int Cout);
size_t eovae_conv3x3_f16c8_wgrad_down_workspace(int N, int Cin, int H, int W, int Cout);
int eovae_conv3x3_f16c8_wgrad_down(const float* x, const float* dy, float* dw, const unsigned* dyamax, const unsigned* xamax,
                                   int N, int Cin, int H, int W, int Cout, void* ws, void* stream);
/* Upper bound of max|y| (float bits) from y's GroupNorm accumulators sums[n_groups][2] as eovae_conv3x3_f16c8_fwd filled
 * them: sqrt(max sum of squares).  For the range guard's power-of-two pre-scale of a layer that reads y un-normalised:
 * one 512-value reduction instead of a pass over y. */
int eovae_amax_bound_from_sums(const double* sums, int n_groups, unsigned* out_bits, void* stream);
/* xpacked = 1 (both entries above): x is not fp32 NCHW but the f16c8 OPERAND-RECORD tensor written by
 * eovae_gn_silu_fwd_packed: [N][C/8][H][W] records of 32 bytes = { 8 x f16 y | 8 x e5m2 (y - f16(y)) * 2^12 |
 * 8 x e5m2 y } for channels 8g..8g+7 -- N*C*H*W dwords, the size of the fp32 activation it replaces.  The
 * conversion to the kernels' operand format is then done once (in the HBM-bound GroupNorm-apply pass) instead of
 * once per 128-co block of the forward convolution and once per block of the weight gradient.  Forward / weight
 * gradient of plain 3x3 layers only (subpix 0, mode 0; wgrad: Cin %% 64 == 0).  Replaces the
 * `h = conv(swish(norm(h)))` pairs of layers.py:91-111 and model.py:193-196, 349-357. */
int eovae_gn_silu_fwd_packed(const float* x, const float* gamma, const float* beta, void* pa, float* stats,
                             const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                             void* stream);
/* As eovae_gn_silu_fwd_packed, but the result is written as MX OPERAND RECORDS ("format 2", xpacked = 2 of the consumers;
 * C %% 32 == 0; pa = N*C*HW dwords, the size of the fp32 activation): first half = f16 plane [N][C/8][HW] x 16 bytes (8 x f16
 * y of channels 8g..8g+7), second half = MX-FP6 plane [N][C/32][piece 0..3][HW] x 16 bytes -- per pixel and 32 channels the
 * block-scaled e2m3 operands of the forward convolution's correction product: pieces 0, 1 = (y - f16(y)), pieces 2, 3 = y; a
 * block = 6 dwords of 32 x 6 bits + its E8M0 scale byte in dword 6 (csrc/conv_f16c8_common.h).  Consumers: the forward of a
 * plain 3x3 layer (eovae_conv3x3_f16c8_fwd_mx) and its weight gradient in the f16 form of the row-streaming kernel
 * (eovae_conv3x3_f16c8_wgrad_dw); ask eovae_records_mx_ok for the layer first. */
int eovae_gn_silu_fwd_packed_mx(const float* x, const float* gamma, const float* beta, void* pa, float* stats,
                                const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                                void* stream);
/* 1 when the plain 3x3 layer (N, Cin, H, W) -> Cout takes MX operand records in its forward AND its weight gradient (shape of
 * the row-streaming kernel; library options c8_mx = 1, c8_wgrad_rows = 1, c8_wgrad_corr = 0) */
int eovae_records_mx_ok(int N, int Cin, int H, int W, int Cout);
/* we[4*Cout][Cin][3][3] = parity-major sub-pixel expansion of w[Cout][Cin][3][3]; and its adjoint
 * dw[Cout][Cin][3][3] from dwp[9][4*Cout][Cin] */
int eovae_subpix_expand(const float* w, float* we, int Cout, int Cin, void* stream);
int eovae_subpix_fold(const float* dwp, float* dw, int Cout, int Cin, void* stream);
/* Space-to-depth form of "F.pad(0,1,0,1) + 3x3 stride 2" (layers.py:33-37) on the same kernels, so no
 * MFMA is spent on inserted zeros:  conv3x3_bf16_fwd subpix 3 = forward (x high-res, wp =
 * pack(down_expand(w, 0)) as a [Cout][4*Cin] layer), subpix 4 = data gradient (x = dy low-res, wp =
 * pack(down_expand(w, 1)) as a [4*Cin][Cout] layer, y = dx high-res); conv3x3_bf16_wgrad subpix 2 =
 * weight gradient into dwp[9][Cout][4*Cin], finished by down_fold. */
int eovae_down_expand(const float* w, float* we, int Cout, int Cin, int dgrad, void* stream);
int eovae_down_fold(const float* dwp, float* dw, int Cout, int Cin, void* stream);
/* split-bf16 weight gradient: dwp[9][Cout][Cin] += dy (x) x, modes 0 / 2 of the forward layer;
 * db (nullable, pre-zeroed [Cout]) also receives the bias gradient sum(dy) (not with subpix 1) */
int eovae_conv3x3_bf16_wgrad(const float* x, const float* dy, float* dwp, float* db, int N, int Cin,
                             int Hin, int Win, int Cout, int Hout, int Wout, int mode, int precision,
                             int subpix, void* stream);
/* dwp[tap][Cout][Cin] += sum_{n,pix} dy * x (mode 0/1/2 of the FORWARD layer); caller zero-fills */
int eovae_conv2d_wgrad(const float* x, const float* dy, float* dwp, int N, int Cin, int Hin,
                       int Win, int Cout, int Hout, int Wout, int KS, int mode, void* stream);
/* dwp[tap][Cout][Cin] -> dw[Cout][Cin][KS][KS] */
/* ----------------------------------------------------------------------------------------
 * Weight gradient of the wavelength-conditioned first / last 3x3 convolutions when the thin side has at most 4
 * channels (csrc/conv_thin.hip; reference dynamic_conv.py:186-206 DynamicConv.forward, :231-246
 * DynamicConv_decoder.forward -> F.conv2d): plain fp32 FMAs, one pass over the many-channel tensor, partial sums
 * folded in a fixed order (deterministic).  dw[Cout][Cin][3][3] is WRITTEN (PyTorch layout, no unpack step);
 * db[Cout] (nullable) is written only when Cout > Cin.  x may be operand records (xpacked = 1) when Cin > Cout.
 * ws: eovae_conv3x3_thin_wgrad_workspace(...) bytes.  _eligible: 1 when the shape is taken (plain 3x3, stride 1,
 * min(Cin, Cout) <= 4, the other side >= 32 and a multiple of 4 (8 for records), W <= 256). */
size_t eovae_conv3x3_thin_wgrad_workspace(int N, int Cin, int H, int W, int Cout);
int eovae_conv3x3_thin_wgrad_eligible(int N, int Cin, int H, int W, int Cout, int xpacked);
int eovae_conv3x3_thin_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int Cin, int H, int W,
                             int Cout, int xpacked, void* ws, void* stream);
/* Forward of the last layer (at most 4 output channels, Cin a multiple of 8 and >= 32; x fp32 or operand records):
 * y = conv3x3(x, w[Cout][Cin][3][3], padding 1) + bias (nullable), fp32 FMAs, one pass over x. */
int eovae_conv3x3_thin_fwd_eligible(int N, int Cin, int H, int W, int Cout);
int eovae_conv3x3_thin_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                           int Cout, int xpacked, void* stream);

int eovae_conv_unpack_wgrad(const float* dwp, float* dw, int Cout, int Cin, int KS, void* stream);
/* db[C] = sum_{n,hw} dy[N][C][HW] */
int eovae_bias_grad(const float* dy, float* db, int N, int C, int HW, void* stream);
/* adjoint of nearest x2: dx[N][C][H][W] = 2x2 block sums of du[N][C][2H][2W] */
int eovae_upsample2x_bwd(const float* du, float* dx, int N, int C, int H, int W, void* stream);

/* ----------------------------------------------------------------------------------------
 * GroupNorm(G groups, eps, affine) [+ x*sigmoid(x)]  (csrc/groupnorm.hip)
 *   nn.GroupNorm(32, C, eps=1e-6) + swish: layers.py:61-63,78-80,91-92,106-108;
 *   model.py:159-161,193-194,295-297,349-350.  silu = 0 for AttnBlock.norm (layers.py:119-121,129)
 * stats[N*G][2] = (mean, rstd) is produced by the forward and consumed by the backward.
 * -------------------------------------------------------------------------------------- */
size_t eovae_gn_workspace_bytes(int N, int C, int HW, int G);
/* sums (nullable): [N*G][2] doubles (sum, sum of squares) of x per (image, group) already accumulated by the
 * kernel that wrote x (eovae_conv3x3_f16c8_fwd gn_sums); NULL = computed here by a pass over x */
int eovae_gn_silu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                      const double* sums, int N, int C, int HW, int G, float eps, int silu, void* ws,
                      void* stream);
/* One-pass GroupNorm backward (a persistent cooperative kernel: x and dy are read once, 12 B per element instead of 20).
 * eovae_gn_bwd_fused(0 / 1 / 2) switches it off / on / on for every eligible shape (1 leaves the shapes where the
 * two-pass kernels measured faster -- 4 channels per group, no dx_add -- to them), any other value only queries; the return value is the setting in
 * force (-1 = undecided: environment EOVAE_GN_BWD_FUSED, default on, is read at first use).
 * eovae_gn_bwd_fused_eligible: 1 when eovae_gn_silu_bwd takes that kernel for the shape (HW % 1024 == 0, team <= 2 x CUs). */
int eovae_gn_bwd_fused(int on);
/* Time-outs of the one-pass kernel's team hand-off on the current device since the last clear (0 = none, else the token
 * of the latest).  The word is host memory the kernels write through, so the call costs nothing and sees every launch the
 * caller has synchronised with; a trainer checks it once per step (a time-out also poisons that call's dgamma / dbeta
 * with NaN).  clear != 0 resets it. */
unsigned long long eovae_gn_bwd_timeouts(int clear);
/* what the wave behind the latest time-out saw: out8[0] = token | 1, [1] = start of its wait window and [2], [3] = its two
 * elapsed readings (100-MHz ticks; the bound is 2e8), [4] = shader clock counter, [5] = window restarts after a jumping
 * reading, [6] = N << 48 | C << 32 | HW of the call, [7] = 0.  0, or -1 when no one-pass backward has run on the device. */
int eovae_gn_bwd_timeout_info(unsigned long long* out8);
int eovae_gn_bwd_fused_eligible(int N, int C, int HW, int G);
/* dx = d/dx (+ dx_add if not NULL); dgamma[C], dbeta[C]; dx_amax_bits (nullable, device): receives the
 * float bits of max|dx| -- the pre-scale input of eovae_conv3x3_f16c8_fwd's data-gradient calls, produced
 * here for free instead of by a separate eovae_absmax_bits pass over dx */
int eovae_gn_silu_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                      const float* stats, const float* dx_add, float* dx, float* dgamma,
                      float* dbeta, unsigned* dx_amax_bits, int N, int C, int HW, int G, int silu,
                      void* ws, void* stream);

/* ----------------------------------------------------------------------------------------
 * Bottleneck attention (csrc/attention.hip): F.scaled_dot_product_attention, one head, d = C
 *   layers.py:133-139.  q,k,v,o: [B][C][T] (NCHW conv outputs, token index contiguous).
 * -------------------------------------------------------------------------------------- */
/* strided-batched fp32 GEMM on MFMA: C(m,n) = alpha * sum_k A(m,k) B(k,n), element strides */
int eovae_gemm_f32(const float* A, const float* B, float* C, int batch, int M, int N, int K,
                   long long sam, long long sak, long long sbk, long long sbn, long long scm,
                   long long scn, long long batA, long long batB, long long batC, float alpha,
                   void* stream);
/* split-bf16 strided-batched GEMM (csrc/gemm_bf16x3.hip): C (+)= alpha*A*B + bias[m] + res; serves
 * the 1x1 convolutions (fwd / data-grad / batch-reduced weight-grad: K extended over kb image slabs,
 * atomic = 1 when K is split over blocks) and attention */
int eovae_gemm_bf16x3(const float* A, const float* B, float* C, const float* bias, const float* res,
                      int batch, int M, int N, int K, long long sam, long long sak, long long sbk,
                      long long sbn, long long scm, long long scn, long long batA, long long batB,
                      long long batC, float alpha, int ksplit, int atomic, int precision,
                      int kb, long long kbA, long long kbB, void* stream);
int eovae_softmax_rows(float* s, long long rows, int n, void* stream);
int eovae_softmax_bwd_rows(const float* p, float* dp, long long rows, int n, void* stream);
/* p[B][T][T] receives softmax(scale * q^T k) and must be kept for the backward */
/* precision: 0 = exact fp32 MFMA GEMMs, 3 = split-bf16, 1 = bf16 */
int eovae_attention_fwd(const float* q, const float* k, const float* v, float* p, float* o, int B,
                        int C, int T, float scale, int precision, void* stream);
/* dp[B][T][T] is scratch */
int eovae_attention_bwd(const float* q, const float* k, const float* v, const float* p,
                        const float* d_o, float* dp, float* dq, float* dk, float* dv, int B, int C,
                        int T, float scale, int precision, void* stream);

/* ----------------------------------------------------------------------------------------
 * Weight generators of the dynamic convolutions (csrc/generator.hip): the post-norm transformer encoder layer
 *   nn.TransformerEncoderLayer(d 256, 4 heads, FFN 2048, GELU, dropout 0)   modules/dynamic_conv.py:86-101, 152-167
 * and the Linear layers around it (:111-113, 336-366) as small exact-fp32 kernels over the [<= 160, 256] token matrix.
 *   gen_gemm   C[b](m,n) = epi(alpha * sum_k A[b](m,k) B[b](n,k)); element strides; C rows scm apart, columns contiguous
 *              epi 0: + bias[n] (nullable) | 1: + bias, pre-activation -> aux_out (layout of C), GELU -> C
 *                  2: times GELU'(aux)     | 3: C += value
 *              ksplit > 1: K slice s -> C + s * part_stride, no epilogue (folded by gen_ln_fwd / gen_ln_bwd)
 *   gen_ln_fwd y = LayerNorm(sum_s parts[s] + bias + res) * gamma + beta; keeps xhat [rows][d] and rstd [rows]
 *   gen_ln_bwd dx of it for dy_total = dy + sum_s parts[s] (written to dysum when nparts > 0); dgamma / dbeta when given
 *   gen_colsum out[c] = sum_r x[r * ld + c]   (bias gradients)
 * -------------------------------------------------------------------------------------- */
int eovae_gen_gemm(const float* A, const float* B, float* C, const float* bias, const float* aux, float* aux_out, int batch,
                   int M, int N, int K, long long sam, long long sak, long long sbn, long long sbk, long long scm,
                   long long batA, long long batB, long long batC, float alpha, int epi, int ksplit, long long part_stride,
                   void* stream);
int eovae_gen_ln_fwd(const float* parts, int nparts, long long part_stride, const float* bias, const float* res,
                     const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int rows, int d, float eps,
                     void* stream);
int eovae_gen_ln_bwd(const float* dy, const float* parts, int nparts, long long part_stride, float* dysum, const float* xhat,
                     const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta, int rows, int d,
                     void* stream);
int eovae_gen_colsum(const float* x, float* out, int rows, int n, long long ld, void* stream);

/* ----------------------------------------------------------------------------------------
 * Latent glue (csrc/latent.hip)
 *   posterior: DiagonalGaussianDistribution            distributions.py:28-32,44-46,64-67
 *   latent_bn: rearrange 2x2 unshuffle + nn.BatchNorm2d(4Z, affine=False)
 *                                                       new_autoencoder.py:466-469,533-536
 *   latent_inv: inverse BN (running stats, eps 1e-4) + 2x2 shuffle   new_autoencoder.py:423-428,538-543
 * -------------------------------------------------------------------------------------- */
/* moments[B][2Z][HW]; eps NULL => mode; z and/or kl[B] may be NULL */
int eovae_posterior_fwd(const float* moments, const float* eps, float* z, float* kl, int B, int Z,
                        int HW, void* stream);
int eovae_posterior_bwd(const float* moments, const float* eps, const float* dz, const float* dkl,
                        float* dmoments, int B, int Z, int HW, void* stream);
/* z[B][Z][H][W] -> zp[B][4Z][H/2][W/2]; training: batch stats + running-stat update
 * (momentum, unbiased variance); save[4Z][2] = (mean, invstd) for the backward */
int eovae_latent_bn_fwd(const float* z, float* zp, float* run_mean, float* run_var, float* save,
                        int B, int Z, int H, int W, int training, float eps, float momentum,
                        void* stream);
int eovae_latent_bn_bwd(const float* zp, const float* dzp, const float* save, float* dz, int B,
                        int Z, int H, int W, int training, void* stream);
/* backward = 0: zp -> z = zp*sqrt(run_var+eps)+run_mean (spatial); backward = 1: dz -> dzp */
int eovae_latent_inv(const float* src, float* dst, const float* run_mean, const float* run_var,
                     int B, int Z, int H, int W, float eps, int backward, void* stream);

/* ----------------------------------------------------------------------------------------
 * Losses (csrc/loss.hip, csrc/msssim.hip): EOConsistencyLoss   consistency_loss.py:399-483
 *   kind 0 = F.l1_loss (:418), kind 1 = CharbonnierLoss eps (:12-21),
 *   kind 2 = F.mse_loss (the Stage-1 distillation step, weight_distill_train.py:211-228)
 *   MS-SSIM: torchmetrics MultiScaleStructuralSimilarityIndexMeasure(data_range=6,
 *            gaussian sigma 1.5 => 11 taps, 5 betas, normalize='relu')   consistency_loss.py:24-37
 * -------------------------------------------------------------------------------------- */
size_t eovae_loss_workspace_bytes(void);
int eovae_pixel_loss_fwd(const float* pred, const float* target, size_t n, int kind, float eps,
                         float* out, void* ws, void* stream);
/* gout: device scalar holding d(total)/d(loss) */
int eovae_pixel_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred,
                         size_t n, int kind, float eps, void* stream);
/* Spectral-angle loss: SAMLoss, consistency_loss.py:186-209 -- mean over pixels of 1 - cos(pred, target)
 * along the channel axis, eps 1e-8 in the denominator.  pred/target [N][C][HW]. */
int eovae_sam_loss_fwd(const float* pred, const float* target, int N, int C, int HW, float eps, float* out,
                       void* ws, void* stream);
int eovae_sam_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int N, int C,
                       int HW, float eps, void* stream);
/* Gradient-difference loss: GradientDifferenceLoss(alpha=1), consistency_loss.py:243-269.
 * pred/target [planes][H][W] (planes = N*C). */
int eovae_gdl_loss_fwd(const float* pred, const float* target, int planes, int H, int W, float* out, void* ws,
                       void* stream);
int eovae_gdl_loss_bwd(const float* pred, const float* target, const float* gout, float* dpred, int planes, int H,
                       int W, void* stream);
/* Focal frequency loss (csrc/ffl.hip): FocalFrequencyLoss(alpha, patch_factor, ave_spectrum=False,
 * log_matrix, batch_matrix=True), ffl.py:17-102 as built at consistency_loss.py:388-395.  The 2-D ortho DFT
 * of the patch differences runs as fp32 MFMA GEMMs against on-device DFT tables.  gy (eovae_ffl_gy_elems
 * floats, caller-allocated) carries conj(dL/dY) from the forward to the backward call. */
size_t eovae_ffl_workspace_bytes(int B, int C, int H, int W, int patch_factor);
size_t eovae_ffl_gy_elems(int B, int C, int H, int W, int patch_factor);
int eovae_ffl_fwd(const float* pred, const float* target, float* out, float* gy, void* ws, int B, int C, int H,
                  int W, int patch_factor, float alpha, int log_matrix, void* stream);
int eovae_ffl_bwd(const float* gy, const float* gout, float* dpred, void* ws, int B, int C, int H, int W,
                  int patch_factor, void* stream);
/* bytes for the image pyramid + per-scale statistics kept between forward and backward */
size_t eovae_msssim_workspace_bytes(int B, int C, int H, int W, int scales);
/* out[0] = mean_b prod_s v_{s,b}^{beta_s}; ws keeps pyramid + per-image factors */
int eovae_msssim_fwd(const float* pred, const float* target, float* out, int B, int C, int H, int W,
                     int scales, const float* betas, float data_range, void* ws, void* stream);
/* dpred = gout[0] * d(msssim)/d(pred), using the state the forward left in ws */
int eovae_msssim_bwd(const float* pred, const float* target, const float* gout, float* dpred, int B,
                     int C, int H, int W, int scales, const float* betas, float data_range, void* ws,
                     void* stream);

/* ----------------------------------------------------------------------------------------
 * Optimiser step over ONE flat buffer (csrc/optim.hip)
 *   torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step   new_autoencoder.py:649-653
 * -------------------------------------------------------------------------------------- */
size_t eovae_gradnorm_workspace_bytes(void);
/* norm_out[0] = ||g||_2, norm_out[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0) */
int eovae_grad_norm(const float* g, size_t n, float max_norm, float* norm_out, void* ws,
                    void* stream);
/* step = Adam step count after increment; clip (may be NULL) = norm_out of eovae_grad_norm */
/* weight_decay > 0: decoupled (torch.optim.AdamW, weight_distill_train.py:317); 0: torch.optim.Adam */
int eovae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, int step, const float* clip, float grad_scale,
                    float weight_decay, void* stream);
int eovae_scale_inplace(float* g, size_t n, const float* clip, float s, void* stream);
/* per-parameter gradient tensors -> flat gradient buffer in ONE launch (instead of one autograd
 * accumulate-add per parameter): table = nrows x {source address or 0 (zeros), destination element offset,
 * element count <= 16384}, int64, device memory */
int eovae_multi_gather(const long long* table, int nrows, float* flat, void* stream);

/* ----------------------------------------------------------------------------------------
 * encode_latents.py running statistics (csrc/stats.hip): RunningStatsButFast.update
 *   encode_latents.py:65-94.  state[5][C] = mean, var, min, max, (count replicated)
 * -------------------------------------------------------------------------------------- */
int eovae_running_stats_update(const float* z, float* state, int B, int C, int HW, void* stream);

/* ----------------------------------------------------------------------------------------
 * Batch preparation, the step before the path (csrc/batchprep.hip): clamp + per-band z-score,
 * bilinear resize (align_corners=False), D4 augmentation, in the reference's order
 *   terramesh_datamodule.py:130-339 (normalisers), :476-479 (resize), :347-369 (flips / rot90)
 * dst is [N][C][Ht][Wt] (or [Wt][Ht] when rot is odd).  mean/std NULL = no z-score; lo > hi = no clamp.
 * -------------------------------------------------------------------------------------- */
int eovae_batch_prep(const float* src, const float* mean, const float* stdv, float* dst, int N, int C,
                     int Hs, int Ws, int Ht, int Wt, float lo, float hi, int hflip, int vflip, int rot,
                     void* stream);

/* ----------------------------------------------------------------------------------------
 * EQ-VAE transforms of the train step (csrc/eqvae.hip).  Forward latent transform
 *   rot90(F.interpolate(z, size, 'bilinear', align_corners=False), k, dims=[-1,-2])
 *   new_autoencoder.py:456-460, 520-531  = eovae_batch_prep(mean = std = NULL, lo > hi, rot = (4-k)%4);
 * eovae_resize_rot_bwd is its adjoint (gradient w.r.t. z); eovae_area_rot builds the no-grad target
 *   rot90(F.interpolate(images, size, mode='area'), k)   new_autoencoder.py:612-617, 623-627.
 * rot = quarter turns in the torch.rot90(dims=[-2,-1]) sense, (Hd, Wd) = (Ht, Wt) swapped when odd.
 * -------------------------------------------------------------------------------------- */
int eovae_resize_rot_bwd(const float* dy, float* dx, int planes, int Hs, int Ws, int Ht, int Wt, int rot,
                         void* stream);
int eovae_area_rot(const float* src, float* dst, int planes, int Hs, int Ws, int Ht, int Wt, int rot,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOVAE_HIP_H */
