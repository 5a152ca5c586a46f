// Shared host-side helpers for the HIP library (error plumbing, small utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>

#include "../../include/orbslam_hip.h"

namespace orbhip {

void set_error(const char* fmt, ...);
}  // namespace orbhip
#include <atomic>
namespace orbhip {
extern std::atomic<int> g_default_device;   // device used by the handle-less (matcher / BA) entry points
inline int use_default_device() {
  hipError_t e = hipSetDevice(g_default_device.load());
  if (e != hipSuccess) { set_error("hipSetDevice(%d) failed: %s", g_default_device.load(), hipGetErrorString(e)); return ORBHIP_ENODEV; }
  return 0;
}

#define ORBHIP_CHECK_HIP(expr)                                                                   \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      ::orbhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (_e == hipErrorOutOfMemory) ? ORBHIP_ENOMEM : ORBHIP_ENODEV;                       \
    }                                                                                            \
  } while (0)

#define ORBHIP_REQUIRE(cond, code, msg)          \
  do {                                           \
    if (!(cond)) {                               \
      ::orbhip::set_error("%s (%s)", msg, #cond); \
      return code;                               \
    }                                            \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// growable device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    size_t want = need + need / 8;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return ORBHIP_ENOMEM; }
    bytes = want;
    // debugging aid: ORBHIP_POISON=ff (or 00) fills every new device buffer with that byte, so that a kernel reading
    // memory nobody wrote shows up as NaNs (ff) instead of depending on what the allocator handed out
    static const int poison = []() { const char* v = std::getenv("ORBHIP_POISON"); return v ? (int)std::strtol(v, nullptr, 16) : -1; }();
    if (poison >= 0) { (void)hipMemset(p, poison, want); (void)hipDeviceSynchronize(); }
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <typename T> T* as() const { return (T*)p; }
};

}  // namespace orbhip
