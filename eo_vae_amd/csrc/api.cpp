// C-ABI housekeeping: thread-local error text + version probe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void eovae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Run-time switches (kernel selection for A/B runs and tests).  Each starts from its environment variable, read ONCE at
// first use; after that only eovae_set_option changes it -- no getenv on a launch path, and a test flips a kernel
// through the ABI instead of mutating os.environ under a running launch.
#include <atomic>
#include <stdlib.h>
#include "eovae_common.h"
namespace {
struct OptDef { const char* name; const char* env; int dflt; };
const OptDef kOpts[EOVAE_OPT_COUNT] = {
    {"c8_wgrad_rows", "EOVAE_C8_WGRAD_ROWS", 1},   // row-streaming weight gradient (conv_f16c8_wgrad_rows.hip)
    {"c8_w4", "EOVAE_C8_W4", -1},                  // conv_f16c8_w4.hip: -1 when it pays, 0 never, 1 whenever the shape allows
    {"c8_w4_ni", "EOVAE_C8_W4_NI", 4},
    {"c8_w4_stagger", "EOVAE_C8_W4_STAGGER", 0},
    {"c8_w4_skew", "EOVAE_C8_W4_SKEW", -1},
    {"c8_wgrad_w4", "EOVAE_C8_WGRAD_W4", 0},       // conv_f16c8_wgrad_w4.hip (parked: slower than the 8-wave kernels)
    {"c8_pack_gather", "EOVAE_C8_PACK_GATHER", 0}, // per-layer weight packs by the per-element gather kernel (reference of the LDS-staged one)
    {"c8_mx", "EOVAE_C8_MX", 1},                   // MX-FP6 correction product (conv_f16c8_common.h); 0 = the fp8 form everywhere
    {"c8_wgrad_corr", "EOVAE_C8_WGRAD_CORR", 0},   // 1: the row-streaming weight gradients keep their fp8 correction product (r4); 0: f16 x f16 alone
    {"c8_q16", "EOVAE_C8_Q16", 1},                 // 16 x 16 MFMAs where a kernel has that form (higher clock under load): 1 (default) record inputs
                                                   // and the plain weight gradient; 2 the fp32-input data gradients too (measured 0.88-0.92 x the
                                                   // 32 x 32 kernel: an experiment that stays tested); 0: the 32 x 32 forms
    {"c8_dgrad_mx", "EOVAE_C8_DGRAD_MX", 1},       // the one-wave kernel quantises fp32 inputs (data gradients) to f16 + MX-FP6 itself; 0: f16 + fp8
    {"c8_up_q16", "EOVAE_C8_UP_Q16", 1},           // x2-upsample forward / data gradient on MX records (conv_f16c8_q16up.hip, conv_f16c8_q16upg.hip); 2: data gradient at any tile count; 0: 8-wave sub-pixel modes
};
std::atomic<int> g_opt[EOVAE_OPT_COUNT];
std::atomic<int> g_opt_init{0};
void opt_init() {
  if (g_opt_init.load(std::memory_order_acquire) == 2) return;
  int expect = 0;
  if (g_opt_init.compare_exchange_strong(expect, 1)) {
    for (int i = 0; i < EOVAE_OPT_COUNT; ++i) {
      const char* e = getenv(kOpts[i].env);
      g_opt[i].store(e && *e ? atoi(e) : kOpts[i].dflt, std::memory_order_relaxed);
    }
    g_opt_init.store(2, std::memory_order_release);
  } else {
    while (g_opt_init.load(std::memory_order_acquire) != 2) {}
  }
}
}  // namespace
int eovae_opt(int id) {
  opt_init();
  return (id >= 0 && id < EOVAE_OPT_COUNT) ? g_opt[id].load(std::memory_order_relaxed) : 0;
}

extern "C" {
const char* eovae_last_error(void) { return g_err; }
int eovae_set_option(const char* name, int value) {
  opt_init();
  for (int i = 0; name && i < EOVAE_OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) {
      g_opt[i].store(value, std::memory_order_relaxed);
      return EOVAE_OK;
    }
  eovae_set_error("eovae_set_option: unknown option '%s'", name ? name : "(null)");
  return EOVAE_ERR_ARG;
}
int eovae_get_option(const char* name) {
  opt_init();
  for (int i = 0; name && i < EOVAE_OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) return g_opt[i].load(std::memory_order_relaxed);
  eovae_set_error("eovae_get_option: unknown option '%s'", name ? name : "(null)");
  return EOVAE_ERR_ARG;
}
int eovae_abi_version(void) { return 12; }  // 12: eovae_conv3x3_f16c8_up_mx, eovae_records_mx_from_f32 (x2-upsample forward on MX records); 11: MX-FP6 correction product (eovae_gn_silu_fwd_packed_mx, eovae_conv3x3_f16c8_fwd_mx, 24 KiB weight tiles); 3: f16c8 convolution entries; 4: gn_silu_bwd reports max|dx|; 5: f16c8 operand records (xpacked); 6: GroupNorm sums from the conv epilogue; 7: f16c8 wgrad takes xamax (range guard), conv3x3_f16c8_which, gn_bwd_fused switches; 8: conv3x3_thin_wgrad; 9: eovae_set_option / eovae_get_option; 10: eovae_gen_* (weight generators)
const char* eovae_target_arch(void) { return "gfx950"; }
int eovae_build_flags(void) { return EOVAE_PARKED ? 1 : 0; }   // bit 0: the parked kernels are compiled in (eovae_common.h)
}
