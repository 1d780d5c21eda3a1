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

extern "C" {
const char* eovae_last_error(void) { return g_err; }
int eovae_abi_version(void) { return 8; }  // 3: f16c8 convolution entries; 4: gn_silu_bwd reports max|dx|; 5: f16c8 operand records (xpacked); 6: GroupNorm sums from the conv epilogue; 7: f16c8 wgrad takes xamax (range guard), conv3x3_f16c8_which, gn_bwd_fused switches; 8: conv3x3_thin_wgrad
const char* eovae_target_arch(void) { return "gfx950"; }
}
