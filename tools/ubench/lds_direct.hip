// Semantics probe for global_load_lds_dword on gfx950: where does lane l's dword land in LDS, with an instruction offset, with
// inactive lanes, with unaligned global addresses?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* g, unsigned int* out, int mode) {
  __shared__ unsigned int lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const int lane = threadIdx.x;
  if (mode == 0) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 4 * lane), (void __attribute__((address_space(3)))*)lds, 4, 0, 0);
  } else if (mode == 1) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 4 * lane), (void __attribute__((address_space(3)))*)lds, 4, 256, 0);
  } else if (mode == 2) {
    if (lane & 1) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 4 * lane), (void __attribute__((address_space(3)))*)lds, 4, 0, 0);
  } else if (mode == 3) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 4 * lane + 1), (void __attribute__((address_space(3)))*)lds, 4, 0, 0);
  } else if (mode == 4) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + 4 * lane), (void __attribute__((address_space(3)))*)(lds + 64), 4, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
// many resident workgroups with their own LDS segments: every one loads 448 dwords of its own global region and checks them
__global__ void kmany(const unsigned int* g, unsigned int* nbad) {
  extern __shared__ unsigned int dl[];
  const int lane = threadIdx.x;
  const unsigned int* src = g + (size_t)blockIdx.x * 448;
  for (int i = lane; i < 1536; i += 64) dl[i] = 0xDEADBEEFu;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#define LD(j) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + lane), (void __attribute__((address_space(3)))*)dl, 4, 256 * j, 0)   /* the immediate offset moves BOTH addresses */
  LD(0); LD(1); LD(2); LD(3); LD(4); LD(5); LD(6);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  int bad = 0;
  for (int i = lane; i < 448; i += 64) bad += dl[i] != src[i];
  for (int i = 448 + lane; i < 1536; i += 64) bad += dl[i] != 0xDEADBEEFu;
  if (bad) atomicAdd(nbad, (unsigned)bad);
}
int main() {
  {
    const int NB = 65536;
    std::vector<unsigned int> hg((size_t)NB * 448); for (size_t i = 0; i < hg.size(); i++) hg[i] = (unsigned)(i * 2654435761u);
    unsigned int *dgm, *dbad; hipMalloc(&dgm, hg.size() * 4); hipMalloc(&dbad, 4); hipMemset(dbad, 0, 4);
    hipMemcpy(dgm, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kmany, dim3(NB), dim3(64), 6144, 0, dgm, dbad);
    unsigned int nb = 0; hipMemcpy(&nb, dbad, 4, hipMemcpyDeviceToHost);
    printf("many workgroups: %u bad dwords\n", nb);
  }
  std::vector<unsigned int> h(256);
  for (int i = 0; i < 256; i++) h[i] = 0x01000000u * (i & 255) + 0x00010000u * ((i * 4 + 2) & 255) + 0x100u * ((i * 4 + 1) & 255) + ((i * 4) & 255);   // byte j of the buffer = j & 255 (except the top byte pattern)
  std::vector<unsigned char> hb(1024); for (int i = 0; i < 1024; i++) hb[i] = (unsigned char)(i ^ (i >> 8) * 0x55);
  unsigned char* dg; unsigned int* dout;
  hipMalloc(&dg, 1024); hipMalloc(&dout, 2048);
  hipMemcpy(dg, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 5; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dg, dout, mode);
    std::vector<unsigned int> o(512);
    hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    printf("mode %d:", mode);
    for (int i = 0; i < 512; i++) if (o[i] != 0xDEADBEEFu && (i < 6 || (i >= 62 && i < 70) || (i >= 126 && i < 132))) printf(" [%d]=%08x", i, o[i]);
    int n = 0; for (int i = 0; i < 512; i++) n += o[i] != 0xDEADBEEFu;
    printf("  (%d written)\n", n);
  }
  return 0;
}
