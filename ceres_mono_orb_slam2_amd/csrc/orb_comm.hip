// The ONE collective of the hot path, reachable from C / C++ (include/orbslam_hip.h, "multi-GPU"): after a batched GlobalBA every
// rank - one process per GPU, one sub-map per rank - merges the landmark updates of all ranks so that each holds the whole map
// (north_star: "a single RCCL all-gather over xGMI to merge landmark updates").  The reference is a single C++ process
// (src/MonoORBSlam.cc:52-100, System threads in src/System.cc): a multi-GPU embedding of it talks to this entry point, not to
// torch.distributed (ceres_mono_orb_slam2_amd/sharding.py is the Python twin the bench uses; tests compare the two).
//
// RCCL is NOT a link-time dependency of liborbslam_hip.so: the six functions used are looked up at the first orbhip_comm_* call
// (dlopen "librccl.so", or the path in ORBHIP_RCCL_LIB; a process that already holds an RCCL - PyTorch brings its own - gets that one).
//
// Sub-maps are ragged.  Every rank contributes ONE fixed-size slot { count, pad, cap x (X, Y, Z[, id]) } - `cap` landmarks per rank,
// agreed by the caller (the sub-map size bound) - so the merge is a single ncclAllGather, no size exchange before it; a kernel then
// packs the slots into the dense output in rank order and leaves the per-rank counts.
#include <dlfcn.h>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/orbslam_hip.h"

namespace orbhip {
namespace {
typedef struct { char internal[128]; } rccl_unique_id;       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
struct RcclApi {
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*CommCount)(rccl_comm_t, int*) = nullptr;
  int (*CommUserRank)(rccl_comm_t, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  bool ok = false;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_rccl.ok) return 0;
  const char* path = std::getenv("ORBHIP_RCCL_LIB");
  const char* names[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) { if (n && *n) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; } }
  if (!h) { set_error("RCCL is not available: %s (set ORBHIP_RCCL_LIB to librccl.so)", dlerror()); return ORBHIP_ENODEV; }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))dlsym(h, "ncclCommUserRank");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.CommCount || !a.CommUserRank || !a.AllGather || !a.GetErrorString) {
    set_error("the RCCL library found does not export the nccl* entry points"); return ORBHIP_ENODEV;
  }
  a.ok = true;
  g_rccl = a;
  return 0;
}
#define ORBHIP_CHECK_RCCL(expr)                                                                                              \
  do {                                                                                                                       \
    const int r_ = (expr);                                                                                                   \
    if (r_ != 0) { set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); return ORBHIP_ENODEV; } \
  } while (0)

// slot of a rank: [0] = count (as int64 bits), [1] = 0, then cap x W doubles (W = 3, or 4 with the id's bits in the fourth)
__global__ void k_lm_pack(const double* __restrict__ pts, const int64_t* __restrict__ ids, int n, int cap, int W, double* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { ((int64_t*)slot)[0] = n; ((int64_t*)slot)[1] = 0; }
  if (i >= cap) return;
  double* o = slot + 2 + (size_t)W * i;
  const bool live = i < n;
  o[0] = live ? pts[3 * (size_t)i] : 0.0; o[1] = live ? pts[3 * (size_t)i + 1] : 0.0; o[2] = live ? pts[3 * (size_t)i + 2] : 0.0;
  if (W == 4) ((int64_t*)o)[3] = live ? ids[i] : 0;
}
// dense output in rank order: rank r's landmarks start at the sum of the counts before it (<= 64 ranks: every thread adds them up itself)
__global__ void k_lm_unpack(const double* __restrict__ slots, int world, int cap, int W, double* __restrict__ pts_all, int64_t* __restrict__ ids_all,
                            int cap_all, int32_t* __restrict__ counts, int* __restrict__ n_all) {
  const int r = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = 2 + (size_t)W * cap;
  long long base = 0, total = 0;
  for (int q = 0; q < world; q++) { const long long c = ((const int64_t*)(slots + stride * q))[0]; if (q < r) base += c; total += c; }
  const long long cnt = ((const int64_t*)(slots + stride * r))[0];
  if (i == 0) { counts[r] = (int32_t)cnt; if (r == 0) *n_all = (int)total; }
  if (i >= cnt || base + i >= cap_all) return;
  const double* s = slots + stride * r + 2 + (size_t)W * i;
  double* o = pts_all + 3 * (size_t)(base + i);
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  if (W == 4 && ids_all) ids_all[base + i] = ((const int64_t*)s)[3];
}
}  // namespace
}  // namespace orbhip

using namespace orbhip;

struct orbhip_comm {
  rccl_comm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  bool owned = false;
  DevBuf send, recv, meta;                                       // slot of this rank, slots of all ranks, counts + total
  int32_t* h_meta = nullptr;                                     // pinned: counts[world], n_all
};

extern "C" {

int orbhip_comm_get_unique_id(uint8_t* id128) {
  ORBHIP_REQUIRE(id128, ORBHIP_EINVAL, "NULL argument");
  if (int rc = load_rccl()) return rc;
  rccl_unique_id id;
  ORBHIP_CHECK_RCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, id.internal, 128);
  return 0;
}

static int comm_finish(orbhip_comm* c, orbhip_comm** out) {
  if (hipHostMalloc((void**)&c->h_meta, sizeof(int32_t) * (size_t)(c->world + 1), hipHostMallocDefault) != hipSuccess) {
    if (c->owned && c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c; set_error("hipHostMalloc failed"); return ORBHIP_ENOMEM;
  }
  *out = c;
  return 0;
}

int orbhip_comm_create(const uint8_t* id128, int world_size, int rank, int device, orbhip_comm** out) {
  ORBHIP_REQUIRE(id128 && out && world_size >= 1 && world_size <= 64 && rank >= 0 && rank < world_size, ORBHIP_EINVAL, "bad communicator arguments (1 <= world size <= 64)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  ORBHIP_REQUIRE(device >= 0 && device < ndev, ORBHIP_EINVAL, "device ordinal out of range");
  if (int rc = load_rccl()) return rc;
  ORBHIP_CHECK_HIP(hipSetDevice(device));
  rccl_unique_id id;
  std::memcpy(id.internal, id128, 128);
  orbhip_comm* c = new orbhip_comm();
  c->world = world_size; c->rank = rank; c->device = device; c->owned = true;
  const int r = g_rccl.CommInitRank(&c->comm, world_size, id, rank);
  if (r != 0) { delete c; set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); return ORBHIP_ENODEV; }
  return comm_finish(c, out);
}

int orbhip_comm_adopt(void* nccl_comm, int device, orbhip_comm** out) {
  ORBHIP_REQUIRE(nccl_comm && out, ORBHIP_EINVAL, "NULL argument");
  if (int rc = load_rccl()) return rc;
  orbhip_comm* c = new orbhip_comm();
  c->comm = (rccl_comm_t)nccl_comm; c->device = device; c->owned = false;
  if (g_rccl.CommCount(c->comm, &c->world) != 0 || g_rccl.CommUserRank(c->comm, &c->rank) != 0 || c->world < 1 || c->world > 64) {
    delete c; set_error("not a usable ncclComm_t (1 <= ranks <= 64)"); return ORBHIP_EINVAL;
  }
  return comm_finish(c, out);
}

int orbhip_comm_info(const orbhip_comm* c, int* world_size, int* rank) {
  ORBHIP_REQUIRE(c, ORBHIP_EINVAL, "NULL communicator");
  if (world_size) *world_size = c->world;
  if (rank) *rank = c->rank;
  return 0;
}

int orbhip_comm_destroy(orbhip_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->owned && c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
  if (c->h_meta) (void)hipHostFree(c->h_meta);
  c->send.release(); c->recv.release(); c->meta.release();
  delete c;
  return 0;
}

int orbhip_allgather_landmarks(orbhip_comm* c, const double* d_pts3_local, const int64_t* d_ids_local, int n_local, int cap_per_rank,
                               double* d_pts3_all, int64_t* d_ids_all, int cap_all, int32_t* counts_out, int* n_all, void* stream) {
  ORBHIP_REQUIRE(c && c->comm, ORBHIP_EINVAL, "NULL communicator");
  ORBHIP_REQUIRE(n_local >= 0 && cap_per_rank >= 1 && cap_all >= 0 && (n_local == 0 || d_pts3_local) && d_pts3_all, ORBHIP_EINVAL, "bad argument");
  ORBHIP_REQUIRE(n_local <= cap_per_rank, ORBHIP_ECAP, "this rank holds more landmarks than the slot the ranks agreed on (cap_per_rank)");
  ORBHIP_REQUIRE((d_ids_local != nullptr) == (d_ids_all != nullptr) || n_local == 0, ORBHIP_EINVAL, "ids must be given on both sides or on neither");
  ORBHIP_CHECK_HIP(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const int W = d_ids_all ? 4 : 3;
  const size_t slot = 2 + (size_t)W * cap_per_rank;
  if (int rc = c->send.ensure(slot * sizeof(double))) return rc;
  if (int rc = c->recv.ensure(slot * sizeof(double) * (size_t)c->world)) return rc;
  if (int rc = c->meta.ensure(sizeof(int32_t) * (size_t)(c->world + 1))) return rc;
  hipLaunchKernelGGL(k_lm_pack, dim3((cap_per_rank + 255) / 256), dim3(256), 0, s, d_pts3_local, d_ids_local, n_local, cap_per_rank, W, c->send.as<double>());
  ORBHIP_CHECK_HIP(hipGetLastError());
  // THE collective: ncclAllGather of one slot per rank (ncclFloat64 = 8 in rccl.h's ncclDataType_t)
  ORBHIP_CHECK_RCCL(g_rccl.AllGather(c->send.as<double>(), c->recv.as<double>(), slot, 8, c->comm, s));
  int32_t* d_counts = c->meta.as<int32_t>();
  hipLaunchKernelGGL(k_lm_unpack, dim3((cap_per_rank + 255) / 256, c->world), dim3(256), 0, s, c->recv.as<double>(), c->world, cap_per_rank, W, d_pts3_all, d_ids_all,
                     cap_all, d_counts, (int*)(d_counts + c->world));
  ORBHIP_CHECK_HIP(hipGetLastError());
  ORBHIP_CHECK_HIP(hipMemcpyAsync(c->h_meta, d_counts, sizeof(int32_t) * (size_t)(c->world + 1), hipMemcpyDeviceToHost, s));
  ORBHIP_CHECK_HIP(hipStreamSynchronize(s));
  const int total = c->h_meta[c->world];
  if (counts_out) std::memcpy(counts_out, c->h_meta, sizeof(int32_t) * (size_t)c->world);
  if (n_all) *n_all = total;
  ORBHIP_REQUIRE(total <= cap_all, ORBHIP_ECAP, "the merged map holds more landmarks than d_pts3_all has room for (the first cap_all are written)");
  return 0;
}

}  // extern "C"
