// Version / error plumbing of the C-ABI (include/lora_amd.h).
#include <cstdarg>
#include <cstdio>

#include "lora_amd.h"

namespace lora_amd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lora_amd

extern "C" int lora_amd_abi_version(void) { return LORA_AMD_ABI_VERSION; }
extern "C" const char *lora_amd_last_error(void) { return lora_amd::g_err; }
extern "C" const char *lora_amd_target_arch(void) { return "gfx950"; }
