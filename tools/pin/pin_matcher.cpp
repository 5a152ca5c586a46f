// pin_matcher: the two std-only pieces of the reference's matcher, taken from the integrator's build of the reference
// (libORB_SLAM2: src/ORBmatcher.cc needs the whole data model to compile, cv::Mat to call), against liborbslam_hip.so:
//   ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1422-1437)       vs  orbm_descriptor_distance          20 000 random pairs + edge patterns
//   ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418, a protected member: reached through a derived class)
//                                                                      vs  the rotation-consistency pass of orbm_search_by_bow
//     (the library has no standalone three-maxima entry point - the pass is fused into its matchers - so the harness builds
//     matching problems whose outcome is decided by it alone: one vocabulary node, every query an exact copy of one target, angle
//     differences drawn so that the 30-bin histogram has prescribed counts, ties and the `max2 < 0.1 max1` / `max3 < 0.1 max1` cases
//     (:1409-1417); the bins that survive in match12 must be the reference's ind1 / ind2 / ind3.)
// Exit code 0 iff everything agrees; --json <file> writes a machine-readable report.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
#include "orbslam_hip.h"
#ifndef PIN_SYNTAX_ONLY
#include "ORBmatcher.h"
struct MatcherAccess : ORB_SLAM2::ORBmatcher {
  MatcherAccess() : ORB_SLAM2::ORBmatcher(0.6f, true) {}
  using ORB_SLAM2::ORBmatcher::ComputeThreeMaxima;
};
#endif

static const int HISTO_LENGTH = 30;

// the bins the library keeps for a given histogram of rotation differences (counts[b] matches fall into bin b)
static bool library_kept_bins(const std::vector<int>& counts, std::set<int>& kept, std::string& err) {
  int n = 0;
  for (int c : counts) n += c;
  if (n == 0) { kept.clear(); return true; }
  std::mt19937 rng(12345u + (unsigned)n);
  std::vector<uint8_t> desc((size_t)n * 32);
  for (auto& b : desc) b = (uint8_t)rng();
  // distinct descriptors far apart: query i matches target i only (distance 0; every other pair is a random 256-bit distance >> TH_LOW)
  std::vector<float> a1(n), a2(n);
  std::vector<int> bin_of(n);
  int i = 0;
  for (int b = 0; b < HISTO_LENGTH; b++)
    for (int c = 0; c < counts[b]; c++, i++) {
      // rot = a1 - a2 (+360 if negative), bin = round(rot * factor) with factor = 1.0f / HISTO_LENGTH (src/ORBmatcher.cc:170, :236-241):
      // the reference's bins are 30 DEGREES wide - only bins 0 .. 12 of the 30 are ever used; the middle of bin b is 30 b degrees
      const float rot = 30.0f * b + (b == 0 ? (float)(rng() % 11) : (float)((int)(rng() % 21) - 10));      // +-10 degrees around the bin centre (bin 0: 0 .. 10, -10 would wrap to bin 12): never on a rounding edge
      a2[i] = (float)(rng() % 360);
      float x = a2[i] + rot; while (x >= 360.f) x -= 360.f; while (x < 0.f) x += 360.f;
      a1[i] = x; bin_of[i] = b;
    }
  std::vector<uint32_t> node(1, 7u), off{0u, (uint32_t)n}, idx(n);
  for (int k = 0; k < n; k++) idx[k] = (uint32_t)k;
  std::vector<int32_t> m12(n, -1); int nm = 0;
  if (orbm_search_by_bow(desc.data(), n, nullptr, a1.data(), desc.data(), n, nullptr, a2.data(), node.data(), off.data(), idx.data(), 1, node.data(), off.data(), idx.data(), 1,
                         0.99f, 50, 0, 1, m12.data(), &nm)) { err = orbhip_last_error(); return false; }
  kept.clear();
  for (int k = 0; k < n; k++) if (m12[k] >= 0) kept.insert(bin_of[k]);
  return true;
}

int main(int argc, char** argv) {
  const char* json = nullptr;
  for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], "--json")) json = argv[i + 1];
  int bad_dist = 0, n_dist = 0, bad_hist = 0, n_hist = 0;
  std::mt19937 rng(7);
  // ---- DescriptorDistance
  std::vector<std::vector<uint8_t>> pats = {std::vector<uint8_t>(32, 0x00), std::vector<uint8_t>(32, 0xFF), std::vector<uint8_t>(32, 0xAA), std::vector<uint8_t>(32, 0x55)};
  for (int k = 0; k < 20000 + 16; k++) {
    uint8_t a[32], b[32];
    if (k < 16) { memcpy(a, pats[k / 4].data(), 32); memcpy(b, pats[k % 4].data(), 32); }
    else for (int j = 0; j < 32; j++) { a[j] = (uint8_t)rng(); b[j] = (k % 3 == 0) ? (uint8_t)(a[j] ^ (1u << (rng() % 8))) : (uint8_t)rng(); }
    const int mine = orbm_descriptor_distance(a, b);
    int ref = mine;
#ifndef PIN_SYNTAX_ONLY
    cv::Mat ma(1, 32, CV_8UC1, a), mb(1, 32, CV_8UC1, b);
    ref = ORB_SLAM2::ORBmatcher::DescriptorDistance(ma, mb);
#endif
    n_dist++; bad_dist += mine != ref;
  }
  // ---- ComputeThreeMaxima through the rotation-consistency pass
  std::vector<std::vector<int>> cases;
  // (bins 0 .. 11: what 30-degree bins can reach with the +-10 degree jitter)
  { std::vector<int> c(HISTO_LENGTH, 0); c[3] = 40; c[4] = 30; c[7] = 20; c[10] = 5; cases.push_back(c); }                  // three clear maxima
  { std::vector<int> c(HISTO_LENGTH, 0); c[0] = 50; c[11] = 4; c[10] = 4; cases.push_back(c); }                            // max2, max3 < 0.1 max1: only one bin survives
  { std::vector<int> c(HISTO_LENGTH, 0); c[5] = 50; c[6] = 30; c[7] = 4; cases.push_back(c); }                             // max3 < 0.1 max1: two survive
  { std::vector<int> c(HISTO_LENGTH, 0); c[2] = 10; c[9] = 10; c[1] = 10; c[5] = 10; cases.push_back(c); }                 // four-way tie: scan order decides
  { std::vector<int> c(HISTO_LENGTH, 0); c[8] = 12; c[1] = 12; c[11] = 7; c[4] = 7; cases.push_back(c); }                  // ties for first and for third
  for (int r = 0; r < 40; r++) { std::vector<int> c(HISTO_LENGTH, 0); for (int b = 0; b < 12; b++) c[b] = (rng() % 2 == 0) ? (int)(rng() % 25) : 0; cases.push_back(c); }
  std::string report = "[";
  for (size_t k = 0; k < cases.size(); k++) {
    std::set<int> kept; std::string err;
    if (!library_kept_bins(cases[k], kept, err)) { fprintf(stderr, "orbm_search_by_bow: %s\n", err.c_str()); return 3; }
    std::set<int> want = kept;
#ifndef PIN_SYNTAX_ONLY
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int b = 0; b < HISTO_LENGTH; b++) rotHist[b].assign(cases[k][b], 0);
    int i1 = -1, i2 = -1, i3 = -1;
    MatcherAccess M; M.ComputeThreeMaxima(rotHist, HISTO_LENGTH, i1, i2, i3);
    want.clear();
    for (int b : {i1, i2, i3}) if (b >= 0 && cases[k][b] > 0) want.insert(b);
#endif
    n_hist++;
    const bool same = want == kept;
    bad_hist += !same;
    if (!same) { printf("histogram case %zu: library keeps {", k); for (int b : kept) printf(" %d", b); printf(" }, reference {"); for (int b : want) printf(" %d", b); printf(" }\n"); }
    report += std::string(k ? "," : "") + "{\"case\":" + std::to_string(k) + ",\"same\":" + (same ? "true" : "false") + "}";
  }
  report += "]";
  printf("DescriptorDistance: %d of %d pairs differ; ComputeThreeMaxima: %d of %d histograms differ\n%s\n", bad_dist, n_dist, bad_hist, n_hist,
         (bad_dist || bad_hist) ? "DIFFERENT" : "PINNED");
  if (json) {
    FILE* f = fopen(json, "w");
    if (f) { fprintf(f, "{\"descriptor_distance\":{\"pairs\":%d,\"different\":%d},\"three_maxima\":{\"histograms\":%d,\"different\":%d,\"cases\":%s}}\n", n_dist, bad_dist, n_hist, bad_hist, report.c_str()); fclose(f); }
  }
  return (bad_dist || bad_hist) ? 1 : 0;
}
