// Error plumbing and library-level entry points of the C ABI (include/orbslam_hip.h).
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

#include <atomic>
namespace orbhip {
std::atomic<int> g_default_device{0};
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
ThreadWs& thread_ws() { static thread_local ThreadWs ws; return ws; }

// 16 bytes per thread and step; both pointers are 256-byte aligned workspace blocks, the tail goes byte by byte
__global__ void k_ws_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t bytes) {
  const size_t n16 = bytes >> 4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[(n16 << 4) + threadIdx.x] = src[(n16 << 4) + threadIdx.x];
}
int ws_copy_kernel(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (!bytes) return 0;
  const int blocks = (int)std::min<size_t>(((bytes >> 4) + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(k_ws_copy, dim3(blocks), dim3(256), 0, s, (uint8_t*)dst, (const uint8_t*)src, bytes);
  return hipGetLastError() == hipSuccess ? 0 : ORBHIP_ENODEV;
}
int raise_dynamic_lds(const void* func, int device, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  std::lock_guard<std::mutex> g(mu);
  size_t& have = granted[std::make_pair(func, device)];
  if (bytes <= have) return 0;
  ORBHIP_CHECK_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return 0;
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
int orbhip_copy_pinned_async(void* dst, const void* src, size_t bytes, void* stream) {
  if (!bytes) return 0;
  if (!dst || !src) { orbhip::set_error("orbhip_copy_pinned_async: NULL pointer"); return ORBHIP_EINVAL; }
  // (a bulk copy over the host link needs latency x bandwidth ~ 80 KB in flight: 8 workgroups of 16-byte lanes reach 25 GB/s, 20 reach
  // 55, 28 or more 57 GB/s; more than that only delays the memory traffic of the kernels the copy overlaps with - a host-fed extract
  // pipeline ran at 0.77 of its bound with 28 - 32 workgroups, 0.53 with 128, 0.48 with 512; round-5 sweeps)
  const int blocks = (int)std::min<size_t>(((bytes >> 4) + 255) / 256 + 1, (size_t)32);
  hipLaunchKernelGGL(orbhip::k_ws_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint8_t*)dst, (const uint8_t*)src, bytes);
  if (hipGetLastError() != hipSuccess) { orbhip::set_error("orbhip_copy_pinned_async: launch failed"); return ORBHIP_ENODEV; }
  return 0;
}
int orbhip_set_default_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { orbhip::set_error("device ordinal %d out of range", device); return ORBHIP_EINVAL; }
  orbhip::g_default_device.store(device);
  return 0;
}
int orbhip_get_default_device(void) { return orbhip::g_default_device.load(); }
int orbhip_set_thread_priority(int high) { orbhip::thread_ws().prio_want = high ? 1 : 0; return 0; }
}
