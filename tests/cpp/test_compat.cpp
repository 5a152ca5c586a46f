// Exercises the reference-named C++ shims (csrc/compat/orbslam_compat.h) end to end on a GPU:
// reads a raw test vector written by the Python test, runs ORBextractor::operator(),
// ORBmatcher::{DescriptorDistance,HammingBest2} and CeresOptimizer::{PoseOptimization,OptimizeSim3}, and writes the
// results back for the Python side to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ceres_mono_orb_slam2_amd/csrc/compat/orbslam_compat.h"

using namespace ORB_SLAM2;

static std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b(n);
  if (fread(b.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: test_compat img.raw w h nfeatures out.bin\n"); return 2; }
  const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]);
  std::vector<uint8_t> img = slurp(argv[1]);
  try {
    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    MatT image(h, w, img.data(), (size_t)w), mask, desc;
    std::vector<KeyPointT> kps;
    ex(image, mask, kps, desc);
    if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8) return 3;
    FILE* f = fopen(argv[5], "wb");
    int n = (int)kps.size();
    fwrite(&n, 4, 1, f);
    fwrite(kps.data(), sizeof(KeyPointT), n, f);
    fwrite(desc.data, 32, n, f);
    // matcher: descriptors against themselves
    ORBmatcher m(0.9f, true);
    std::vector<int> bi, bd, sd;
    m.HammingBest2(desc, desc, bi, bd, sd);
    fwrite(bi.data(), 4, n, f); fwrite(bd.data(), 4, n, f); fwrite(sd.data(), 4, n, f);
    MatT r0(1, 32, desc.ptr(0), 32), r1(1, 32, desc.ptr(n > 1 ? 1 : 0), 32);
    int d01 = ORBmatcher::DescriptorDistance(r0, r1);
    fwrite(&d01, 4, 1, f);
    if (argc >= 8) {
      // optimizer shims: problems written by the Python test as raw little-endian arrays
      std::vector<uint8_t> pb = slurp(argv[6]);
      const uint8_t* c = pb.data();
      auto take = [&](void* dst, size_t bytes) { memcpy(dst, c, bytes); c += bytes; };
      int np_ = 0; take(&np_, 4);
      PoseProblem P;
      take(P.K4, 32); take(P.pose7, 56);
      P.Xw.resize(3 * np_); P.uv.resize(2 * np_); P.inv_sigma2.resize(np_);
      take(P.Xw.data(), 24 * np_); take(P.uv.data(), 16 * np_); take(P.inv_sigma2.data(), 4 * np_);
      int inl = CeresOptimizer::PoseOptimization(&P);
      fwrite(&inl, 4, 1, f); fwrite(P.pose7, 8, 7, f); fwrite(P.is_outliers_.data(), 1, np_, f);
      std::vector<uint8_t> sb = slurp(argv[7]);
      c = sb.data();
      int ns = 0; take(&ns, 4);
      Sim3Problem Q; double S12[7];
      take(Q.K1, 32); take(Q.K2, 32); take(S12, 56);
      Q.P3D2c.resize(3 * ns); Q.obs1.resize(2 * ns); Q.P3D1c.resize(3 * ns); Q.obs2.resize(2 * ns);
      Q.inv_sigma2_1.resize(ns); Q.inv_sigma2_2.resize(ns);
      take(Q.P3D2c.data(), 24 * ns); take(Q.obs1.data(), 16 * ns); take(Q.P3D1c.data(), 24 * ns); take(Q.obs2.data(), 16 * ns);
      take(Q.inv_sigma2_1.data(), 4 * ns); take(Q.inv_sigma2_2.data(), 4 * ns);
      int sinl = CeresOptimizer::OptimizeSim3(&Q, S12, 10.f, false);
      fwrite(&sinl, 4, 1, f); fwrite(S12, 8, 7, f); fwrite(Q.is_outliers_.data(), 1, ns, f);
      printf("test_compat: pose inliers %d, sim3 inliers %d\n", inl, sinl);
    }
    if (argc >= 9) {
      // widened rows: window candidates, ORBVocabulary::transform, OptimizeEssentialGraph (inputs written by the Python test)
      std::vector<uint8_t> ab = slurp(argv[8]);
      const uint8_t* c = ab.data();
      auto take = [&](void* dst, size_t bytes) { memcpy(dst, c, bytes); c += bytes; };
      FrameView F;
      F.undistort_keypoints_ = kps; F.descriptors_ = desc; F.min_x_ = 0; F.max_x_ = (float)w; F.min_y_ = 0; F.max_y_ = (float)h;
      int nq = 0; take(&nq, 4);
      std::vector<float> qxy(2 * nq), qr(nq); std::vector<int> qmn(nq), qmx(nq);
      take(qxy.data(), 8 * nq); take(qr.data(), 4 * nq); take(qmn.data(), 4 * nq); take(qmx.data(), 4 * nq);
      std::vector<std::vector<size_t>> ind;
      FrameOps::GetFeaturesInArea(F, qxy, qr, qmn, qmx, ind);
      for (int q = 0; q < nq; q++) { int m = (int)ind[q].size(); fwrite(&m, 4, 1, f); for (size_t v : ind[q]) { int vi = (int)v; fwrite(&vi, 4, 1, f); } }
      int nn = 0, L = 0, nch = 0; take(&nn, 4); take(&L, 4); take(&nch, 4);
      std::vector<uint8_t> nd(32 * (size_t)nn); std::vector<uint32_t> co(nn + 1), ch(nch); std::vector<int32_t> wi(nn); std::vector<double> wt(nn);
      take(nd.data(), nd.size()); take(co.data(), 4 * (nn + 1)); take(ch.data(), 4 * (size_t)nch); take(wi.data(), 4 * (size_t)nn); take(wt.data(), 8 * (size_t)nn);
      ORBVocabulary voc(nd.data(), co.data(), ch.data(), wi.data(), wt.data(), nn, L);
      BowVector bv; FeatureVector fv;
      voc.transform(desc, bv, fv, 2);
      int nw = (int)bv.size(); fwrite(&nw, 4, 1, f);
      for (auto& kv : bv) { fwrite(&kv.first, 4, 1, f); fwrite(&kv.second, 8, 1, f); }
      int nfv = (int)fv.size(); fwrite(&nfv, 4, 1, f);
      for (auto& kv : fv) { int m = (int)kv.second.size(); fwrite(&kv.first, 4, 1, f); fwrite(&m, 4, 1, f); fwrite(kv.second.data(), 4, m, f); }
      double self = voc.score(bv, bv); fwrite(&self, 8, 1, f);
      EssentialGraphProblem G;
      int nv = 0, ne = 0; take(&nv, 4); take(&ne, 4);
      G.Scw_datas.resize(7 * (size_t)nv); G.kf_fixed.resize(nv); G.edge_j.resize(ne); G.edge_i.resize(ne); G.edge_Sji.resize(7 * (size_t)ne);
      take(G.Scw_datas.data(), 56 * (size_t)nv); take(G.kf_fixed.data(), nv); take(G.edge_j.data(), 4 * (size_t)ne); take(G.edge_i.data(), 4 * (size_t)ne);
      take(G.edge_Sji.data(), 56 * (size_t)ne);
      CeresOptimizer::OptimizeEssentialGraph(&G);
      fwrite(G.Scw_datas.data(), 8, 7 * (size_t)nv, f);
      printf("test_compat: %d bow words, %d fv nodes, essential graph of %d keyframes\n", nw, nfv, nv);
    }
    fclose(f);
    printf("test_compat: %d keypoints, d(0,1)=%d\n", n, d01);
  } catch (const std::exception& e) {
    fprintf(stderr, "test_compat: %s\n", e.what());
    return 1;
  }
  return 0;
}
