// Shared host-side helpers for the HIP library (error plumbing, small utilities).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>

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

// Environment switches.  The product library reads only the ones a maintainer needs (documented in INTEGRATION.md: ORBHIP_MATCH_MFMA,
// ORBHIP_BA_PERSIST, ORBHIP_BA_WG, ORBHIP_BA_LOOKAHEAD / ORBHIP_BA_LA_MAX, the ORBHIP_*_TIMING diagnostics, ORBHIP_POISON, the
// ORBHIP_KEYCAP test hook); the knobs of measured-and-rejected alternatives exist only in -DORBHIP_EXPERIMENTS builds (tools/).
#ifdef ORBHIP_EXPERIMENTS
#define ORBHIP_EXP_ENV(name) std::getenv(name)
#else
#define ORBHIP_EXP_ENV(name) ((const char*)nullptr)
#endif

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Copies between PINNED host staging and device memory are done BY A KERNEL on the call's stream (k_ws_copy, capi_common.hip: pinned
// host memory is mapped into the device's address space), not by hipMemcpyAsync.  Measured (tools/scratch: rocprofv3 --hip-trace of
// tests/test_gpu_concurrency.py): with two or three host threads issuing asynchronous copies, a hipMemcpyAsync of a few hundred
// kilobytes now and then takes 7-10 ms ON THE HOST, in two threads at once, with the GPU idle (the copy itself: 60 us) - the copy-
// engine path of the runtime serialises the threads.  It was the whole p99 of a Tracking step beside LocalBA + GlobalBA: 8.4 ms with
// hipMemcpyAsync, 0.61 ms with the kernel (and the step alone went from 0.414 to 0.386 ms: no copy-engine hand-off in the chain).
// ORBHIP_WS_COPY_KERNEL=0 in an experiments build (tools/build_experiments.sh) restores hipMemcpyAsync for the A/B (tools/trace_concurrency.sh).
int ws_copy_kernel(void* dst, const void* src, size_t bytes, hipStream_t s);
inline bool ws_copy_by_kernel() {
  static const bool on = []() { const char* e = ORBHIP_EXP_ENV("ORBHIP_WS_COPY_KERNEL"); return !(e && e[0] == '0'); }();   // (the A/B switch exists in experiments builds only)
  return on;
}
inline hipError_t ws_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {      // one side is pinned host memory
  if (ws_copy_by_kernel()) return ws_copy_kernel(dst, src, bytes, s) ? hipErrorUnknown : hipSuccess;
  return hipMemcpyAsync(dst, src, bytes, kind, s);
}

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
    // (on a non-blocking stream of its own: hipMemset on the legacy stream + a device synchronise break another host thread's
    // hipGraph capture - "operation would make the legacy stream depend on a capturing blocking stream")
    if (poison >= 0) {
      hipStream_t ps = nullptr;
      if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) == hipSuccess) {
        (void)hipMemsetAsync(p, poison, want, ps); (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps);
      }
    }
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <typename T> T* as() const { return (T*)p; }
};

// growable pinned host buffer
struct PinnedHost {
  void* p = nullptr; size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
    size_t want = need + need / 4;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; set_error("hipHostMalloc(%zu) failed", want); return ORBHIP_ENOMEM; }
    bytes = want;
    return 0;
  }
};

// Per-host-thread workspace of the host-pointer entry points (matcher / frame calls): ONE non-blocking stream, device
// buffers and pinned staging that grow and are reused across calls (slot order = request order inside a call), so a call
// costs no hipMalloc / hipFree and its copies are true asynchronous DMA; every call ends with a single stream synchronise.
// No destructor: freeing from a thread_local destructor can run after the HIP runtime has shut down.
struct ThreadWs {
  hipStream_t s = nullptr; int device = -1; int prio_want = 0, prio_cur = 0;
  std::vector<DevBuf> dev; std::vector<PinnedHost> pin;
  size_t dnext = 0, pnext = 0;
  int begin() {                                              // select the default device, (re)create the stream, rewind the slots
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
    if (int rc = use_default_device()) return rc;
    const int d = g_default_device.load();
    if (s && device != d) { (void)hipStreamDestroy(s); s = nullptr; dev.clear(); pin.clear(); }   // (buffers of the old device are leaked on purpose)
    if (s && prio_cur != prio_want) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); s = nullptr; }     // (orbhip_set_thread_priority since the last call)
    if (!s) {
      // the calling thread's stream; a thread that asked for priority (the Tracking thread: one frame's chain of ~15 short kernels
      // beside another thread's bundle adjustment) gets the device's greatest stream priority
      int lo = 0, hi = 0;
      hipError_t e = hipSuccess;
      if (prio_want && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
      else e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      if (e != hipSuccess) { s = nullptr; set_error("hipStreamCreate failed"); return ORBHIP_ENODEV; }
      device = d; prio_cur = prio_want;
    }
    dnext = pnext = 0;
    return 0;
  }
  template <typename T> T* d(size_t count, int* rc) {
    if (dnext >= dev.size()) dev.resize(dnext + 1);
    DevBuf& b = dev[dnext++];
    int r = b.ensure(std::max<size_t>(count * sizeof(T), 16));
    if (r && !*rc) *rc = r;
    return b.as<T>();
  }
  template <typename T> T* h(size_t count, int* rc) {
    if (pnext >= pin.size()) pin.resize(pnext + 1);
    PinnedHost& b = pin[pnext++];
    int r = b.ensure(std::max<size_t>(count * sizeof(T), 16));
    if (r && !*rc) *rc = r;
    return (T*)b.p;
  }
  template <typename T> T* up(const T* src, size_t count, int* rc) {          // host (any memory) -> pinned -> device, asynchronous
    T* hp = h<T>(count, rc); T* dp = d<T>(count, rc);
    if (*rc) return dp;
    if (count) {
      std::memcpy(hp, src, count * sizeof(T));
      if (copy(dp, hp, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpyAsync H2D failed"); *rc = ORBHIP_ENODEV; }
    }
    return dp;
  }
  template <typename T> T* down(const T* dp, size_t count, int* rc) {        // device -> pinned (valid after sync())
    T* hp = h<T>(count, rc);
    if (*rc) return hp;
    if (count && copy(hp, dp, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpyAsync D2H failed"); *rc = ORBHIP_ENODEV; }
    return hp;
  }
  // several host arrays in ONE pinned block and ONE asynchronous copy (a copy costs microseconds of fixed overhead, the
  // per-frame calls move a few hundred kilobytes in ten pieces): add() the pieces, commit(), then dev<T>(piece)
  struct Pack {
    struct Piece { const void* src; size_t bytes, off; };
    std::vector<Piece> pieces; size_t total = 0; uint8_t* dbase = nullptr;
    int add(const void* src, size_t bytes) {
      const size_t off = total;
      pieces.push_back({src, bytes, off});
      total = (off + bytes + 255) & ~(size_t)255;
      return (int)pieces.size() - 1;
    }
    template <typename T> T* dev(int piece) const { return piece < 0 ? nullptr : (T*)(dbase + pieces[piece].off); }
  };
  int commit(Pack& P) {
    int rc = 0;
    uint8_t* hp = h<uint8_t>(P.total, &rc); P.dbase = d<uint8_t>(P.total, &rc);
    if (rc) return rc;
    for (const Pack::Piece& q : P.pieces) if (q.bytes) std::memcpy(hp + q.off, q.src, q.bytes);
    if (P.total && copy(P.dbase, hp, P.total, hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpyAsync H2D failed"); return ORBHIP_ENODEV; }
    return 0;
  }
  hipError_t copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) { return ws_copy(dst, src, bytes, kind, s); }
  int sync() {
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("stream synchronise failed: %s", hipGetErrorString(e)); return ORBHIP_ENODEV; }
    return 0;
  }
};
ThreadWs& thread_ws();          // (capi_common.hip)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, device) and is shared by every host thread and every context:
// the largest value requested so far is kept per (function, device) under a lock and only ever RAISED - a per-thread or
// per-context cache lets a smaller request lower the limit under a thread whose cache still says "set" (ADVICE r3).
// The current device must be `device`.  (capi_common.hip)
int raise_dynamic_lds(const void* func, int device, size_t bytes);

}  // namespace orbhip
