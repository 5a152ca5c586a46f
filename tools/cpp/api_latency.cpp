// Wall latency of the per-frame host-pointer entry points through the C ABI, called from C++ (what the reference's
// Tracking thread would do): orbm_hamming_best2 2000 x 2000, orbm_search_by_projection (1800 queries, SearchByProjection
// (cur, last) shape), orbm_search_for_initialization, ba_pose_optimization.  Prints one JSON line.
//   g++ -O2 -std=c++17 -I include tools/cpp/api_latency.cpp -o /tmp/api_latency -L ceres_mono_orb_slam2_amd/lib -lorbslam_hip
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include "orbslam_hip.h"

static double lat_ms(const std::function<void()>& f, int n = 200) {
  for (int i = 0; i < 10; i++) f();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) f();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  std::mt19937 rng(7);
  const int n = 2000, W = 1241, H = 376;
  const float quota[8] = {434, 362, 302, 251, 209, 175, 145, 122};
  std::vector<float> k1(4 * n), k2(4 * n), quv(2 * n), qr(n), qang(n);
  std::vector<uint8_t> d1(32 * n), d2(32 * n);
  std::vector<int32_t> mn(n), mx(n);
  std::uniform_real_distribution<float> ux(20, W - 20), uy(20, H - 20), un(-1.5f, 1.5f), ua(0, 360);
  std::discrete_distribution<int> lv(quota, quota + 8);
  for (int i = 0; i < n; i++) {
    const int l = lv(rng);
    k1[4 * i] = ux(rng); k1[4 * i + 1] = uy(rng); k1[4 * i + 2] = (float)l; k1[4 * i + 3] = ua(rng);
    k2[4 * i] = k1[4 * i] + 3.f + un(rng); k2[4 * i + 1] = k1[4 * i + 1] - 2.f + un(rng); k2[4 * i + 2] = (float)l; k2[4 * i + 3] = k1[4 * i + 3];
    for (int b = 0; b < 32; b++) { d1[32 * i + b] = (uint8_t)rng(); d2[32 * i + b] = d1[32 * i + b]; }
    for (int f = 0; f < 20; f++) d2[32 * i + (rng() & 31)] ^= (uint8_t)(1u << (rng() & 7));      // ~20 bits apart: a true match
    quv[2 * i] = k1[4 * i] + 3.f; quv[2 * i + 1] = k1[4 * i + 1] - 2.f;
    qr[i] = 15.f * std::pow(1.2f, (float)l); qang[i] = k1[4 * i + 3]; mn[i] = l - 1; mx[i] = l + 1;
  }
  const float bounds[4] = {0, (float)W, 0, (float)H};
  std::vector<int32_t> bi(n), bd(n), sd(n), match(n), bdist(n);
  int nm = 0;
  const double t_b2 = lat_ms([&] { if (orbm_hamming_best2(d1.data(), n, d2.data(), n, nullptr, nullptr, bi.data(), bd.data(), sd.data())) { fprintf(stderr, "%s\n", orbhip_last_error()); exit(1); } });
  int hits = 0; for (int i = 0; i < n; i++) hits += bi[i] == i;
  std::vector<uint8_t> taken(n);
  const double t_sbp = lat_ms([&] {
    std::fill(taken.begin(), taken.end(), 0);
    if (orbm_search_by_projection(k2.data(), d2.data(), n, bounds, quv.data(), qr.data(), mn.data(), mx.data(), nullptr, d1.data(), nullptr, qang.data(), n,
                                  nullptr, 0.f, taken.data(), 0, 0.9f, 100, 1, match.data(), bdist.data(), &nm)) { fprintf(stderr, "%s\n", orbhip_last_error()); exit(1); }
  });
  const int nm_sbp = nm;
  std::vector<float> prev(2 * n);
  std::vector<float> k1l0 = k1, k2l0 = k2;
  for (int i = 0; i < n; i++) { k1l0[4 * i + 2] = 0; k2l0[4 * i + 2] = 0; }
  const double t_init = lat_ms([&] {
    for (int i = 0; i < n; i++) { prev[2 * i] = k1[4 * i]; prev[2 * i + 1] = k1[4 * i + 1]; }
    if (orbm_search_for_initialization(k1l0.data(), d1.data(), n, k2l0.data(), d2.data(), n, bounds, prev.data(), 100, 0.9f, 1, match.data(), &nm)) { fprintf(stderr, "%s\n", orbhip_last_error()); exit(1); }
  }, 50);
  // PoseOptimization: 2000 points in front of a KITTI camera
  std::vector<double> Xw(3 * n), uv(2 * n); std::vector<float> isg(n, 1.f); std::vector<uint8_t> outl(n);
  const double K4[4] = {718.856, 718.856, 607.1928, 185.2157};
  std::uniform_real_distribution<double> uz(4, 60);
  for (int i = 0; i < n; i++) {
    const double z = uz(rng), u = ux(rng), v = uy(rng);
    Xw[3 * i] = (u - K4[2]) / K4[0] * z; Xw[3 * i + 1] = (v - K4[3]) / K4[1] * z; Xw[3 * i + 2] = z;
    uv[2 * i] = u + un(rng) * 0.5; uv[2 * i + 1] = v + un(rng) * 0.5;
  }
  int ninl = 0;
  const double t_pose = lat_ms([&] {
    double pose[7] = {0.03, -0.02, 0.05, 0.002, -0.001, 0.003, 1.0};
    if (ba_pose_optimization(K4, pose, Xw.data(), uv.data(), isg.data(), n, outl.data(), &ninl, nullptr)) { fprintf(stderr, "%s\n", orbhip_last_error()); exit(1); }
  });
  printf("{\"caller\": \"C++ through the C ABI\", \"orbm_hamming_best2_2000x2000_ms\": %.4f, \"best2_self_hits\": %d, \"orbm_search_by_projection_ms\": %.4f, "
         "\"search_by_projection_matches\": %d, \"orbm_search_for_initialization_ms\": %.4f, \"search_for_initialization_matches\": %d, "
         "\"ba_pose_optimization_ms\": %.4f, \"pose_inliers\": %d}\n", t_b2, hits, t_sbp, nm_sbp, t_init, nm, t_pose, ninl);
  return 0;
}
