// Error plumbing and library-level entry points of the C ABI (include/orbslam_hip.h).
#include "common.h"

namespace orbhip {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace orbhip

extern "C" {
const char* orbhip_last_error(void) { return orbhip::g_err; }
int orbhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* orbhip_version(void) { return "orbslam_hip 0.1 (gfx950)"; }
}
