// tvmi_core.hip — version / error plumbing of libtvmi_kernels.so.
#include <string.h>

#include "tvmi_common.h"

namespace tvmi {
namespace {
thread_local char g_last_error[256] = "";
}

int set_error(int code, const char* what) {
  const char* hip_msg = hipGetErrorString(static_cast<hipError_t>(code));
  snprintf(g_last_error, sizeof(g_last_error), "%s (%d: %s)", what ? what : "tvmi", code,
           hip_msg ? hip_msg : "?");
  return code;
}
}  // namespace tvmi

extern "C" int tvmi_version(void) { return TVMI_ABI_VERSION; }
extern "C" const char* tvmi_arch(void) { return "gfx950"; }
extern "C" const char* tvmi_last_error(void) { return tvmi::g_last_error; }
