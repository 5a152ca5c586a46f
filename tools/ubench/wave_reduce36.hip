// Check of csrc/wave_reduce.h on the device: 36 values per lane, the totals against a host sum in the same order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o wave_reduce36 wave_reduce36.hip && ./wave_reduce36
#include "../../ceres_mono_orb_slam2_amd/csrc/wave_reduce.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void k(const double* in, double* out, int* slot) {
  const int lane = threadIdx.x;
  double v[36];
  for (int j = 0; j < 36; j++) v[j] = in[lane * 36 + j];
  const double t = orbhip::wave_reduce36(v, lane);
  out[lane] = t; slot[lane] = orbhip::wave_reduce36_slot(lane);
}
int main() {
  std::vector<double> h(64 * 36); srand(7);
  for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * 1e3;
  double *din, *dout; int* dslot;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dslot, 64 * 4);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dslot);
  double out[64]; int slot[64];
  hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost); hipMemcpy(slot, dslot, sizeof(slot), hipMemcpyDeviceToHost);
  int bad = 0, seen[36] = {0};
  for (int l = 0; l < 64; l++) {
    if (slot[l] < 0) continue;
    seen[slot[l]]++;
    // host sum in the kernel's order: pairs ^32, ^16, ^8, ^1, ^2, ^4
    double s[64];
    for (int i = 0; i < 64; i++) s[i] = h[i * 36 + slot[l]];
    const int order[6] = {32, 16, 8, 1, 2, 4};
    for (int o : order) { double t[64]; for (int i = 0; i < 64; i++) t[i] = s[i] + s[i ^ o]; for (int i = 0; i < 64; i++) s[i] = t[i]; }
    if (s[l] != out[l]) { bad++; printf("lane %d slot %d: %.17g vs %.17g\n", l, slot[l], out[l], s[l]); }
  }
  for (int j = 0; j < 36; j++) if (seen[j] != 1) { bad++; printf("value %d owned by %d lanes\n", j, seen[j]); }
  printf(bad ? "FAIL\n" : "wave_reduce36 OK (bit-exact against the host sum in the same order)\n");
  return bad != 0;
}
