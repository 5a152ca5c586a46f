// Cross-workgroup hand-off latency on gfx950 (MI355X: 8 XCDs, one L2 each): what a persistent (flag-synchronised) Cholesky
// chain would pay per step instead of a kernel boundary.  Workgroup A and workgroup B ping-pong through a flag in global
// memory N times; variants:
//   0  flag only, relaxed agent-scope atomics (no fences)
//   1  flag + 8 KB payload, the HIP memory-model way: plain stores, __threadfence(), flag store / flag load, __threadfence(), plain loads
//   2  flag + 8 KB payload written and read with agent-scope relaxed atomics (sc1: no cache maintenance), s_waitcnt before the flag
// for B on the same XCD as A (workgroup ids 0 and 8) and on another XCD (ids 0 and 1).  Timed with s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o flag_hop flag_hop.hip && ./flag_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2000
#define SPIN_CAP (1 << 22)
__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_flag(const int* p, int v) {
  for (int it = 0; it < SPIN_CAP; it++) { if (ld_flag(p) >= v) return true; __builtin_amdgcn_s_sleep(1); }
  return false;
}
template <int V>
__global__ __launch_bounds__(256) void k_hop(int* flag, double* buf, int partner, unsigned long long* ticks, double* sink) {
  const int me = blockIdx.x, tid = threadIdx.x;
  if (me != 0 && me != partner) return;
  const bool A = me == 0;
  double* mine = buf + (A ? 0 : 1024), *theirs = buf + (A ? 1024 : 0);
  double acc = 0.0;
  __shared__ int s_ok;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < N; t++) {
    const int my_turn = 2 * t + (A ? 1 : 2), wait_for = 2 * t + (A ? 2 : 1);
    if (!A) {                                                  // B waits first
      if (tid == 0) s_ok = wait_flag(flag, wait_for);
      __syncthreads();
      if (!s_ok) return;
      if (V == 1) { __threadfence(); for (int u = 0; u < 4; u++) acc += theirs[tid + 256 * u]; }
      if (V == 2) { for (int u = 0; u < 4; u++) acc += __hip_atomic_load(&theirs[tid + 256 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if (V == 1) { for (int u = 0; u < 4; u++) mine[tid + 256 * u] = acc + t + u; __threadfence(); }
    if (V == 2) { for (int u = 0; u < 4; u++) __hip_atomic_store(&mine[tid + 256 * u], acc + t + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_waitcnt(0); }
    __syncthreads();
    if (tid == 0) st_flag(flag, my_turn);
    if (A) {
      if (tid == 0) s_ok = wait_flag(flag, wait_for);
      __syncthreads();
      if (!s_ok) return;
      if (V == 1) { __threadfence(); for (int u = 0; u < 4; u++) acc += theirs[tid + 256 * u]; }
      if (V == 2) { for (int u = 0; u < 4; u++) acc += __hip_atomic_load(&theirs[tid + 256 * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && A) ticks[0] = t1 - t0;
  sink[me * 256 + tid] = acc;
}
template <int V> static void run(const char* name, int partner, int* flag, double* buf, unsigned long long* ticks, double* sink) {
  hipMemset(flag, 0, 4); hipMemset(buf, 0, 2048 * 8);
  hipLaunchKernelGGL(k_hop<V>, dim3(16), dim3(256), 0, 0, flag, buf, partner, ticks, sink);
  hipDeviceSynchronize();
  unsigned long long t = 0; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  double chk = 0; hipMemcpy(&chk, sink, 8, hipMemcpyDeviceToHost);
  printf("%-44s partner wg %2d (%s XCD): %7.1f ns per one-way hop   (check %.0f)\n", name, partner, partner % 8 == 0 ? "same " : "other", t * 10.0 / (2.0 * N), chk);
}
int main() {
  int* flag; double *buf, *sink; unsigned long long* ticks;
  hipMalloc(&flag, 64); hipMalloc(&buf, 2048 * 8); hipMalloc(&sink, 16 * 256 * 8); hipMalloc(&ticks, 8);
  for (int partner : {8, 1, 4}) {
    run<0>("flag only (relaxed agent atomics)", partner, flag, buf, ticks, sink);
    run<1>("flag + 8 KB, __threadfence both sides", partner, flag, buf, ticks, sink);
    run<2>("flag + 8 KB, agent-scope atomic data accesses", partner, flag, buf, ticks, sink);
  }
  return 0;
}
