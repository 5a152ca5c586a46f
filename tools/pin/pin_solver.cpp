// pin_solver <cases dir>: CeresOptimizer::PoseOptimization's solve (src/CeresOptimizer.cc:275-342) and the BundleAdjustment solve
// (:59-225), rebuilt from the REFERENCE'S OWN cost functors (include/CeresOptimizer.h: PoseErrorTerm :111-166, PoseGraph3dErrorTerm
// :56-109) with the reference's options (Huber sqrt(5.991), EigenQuaternionParameterization, 100 / N iterations; the sparse / dense
// linear solver choice does not change the iterates beyond rounding), against ba_pose_optimization / ba_solve of liborbslam_hip.so
// on the problems make_cases.py wrote (pose_*.bin, ba_*.bin: little-endian, layout below).  Reports iteration counts, final costs,
// pose / point differences; exit code 0 iff every problem meets the bars the in-repo tests state (iterations equal, cost 1e-9,
// poses 1e-7).
//   pose_XXX.bin : int32 n | double K4[4] | double pose7[7] | double Xw[3n] | double uv[2n] | float inv_sigma2[n]
//   ba_XXX.bin   : int32 ncam, npts, nobs, iters | double K4[4 ncam] | double poses7[7 ncam] | uint8 cam_fixed[ncam] | double pts[3 npts]
//                  | int32 obs_cam[nobs] | int32 obs_pt[nobs] | double obs_uv[2 nobs] | float obs_inv_sigma2[nobs]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "orbslam_hip.h"
#ifndef PIN_SYNTAX_ONLY
#include "CeresOptimizer.h"
#endif

template <typename T> static bool rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return n == 0 || fread(v.data(), sizeof(T), n, f) == n; }
static double maxdiff(const double* a, const double* b, size_t n) { double m = 0; for (size_t i = 0; i < n; i++) m = std::fmax(m, std::fabs(a[i] - b[i])); return m; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: pin_solver <cases dir>\n"); return 2; }
  int bad = 0, total = 0;
  for (int k = 0;; k++) {
    char name[64]; snprintf(name, sizeof(name), "/pose_%03d.bin", k);
    FILE* f = fopen((std::string(argv[1]) + name).c_str(), "rb");
    if (!f) break;
    int32_t n = 0; std::vector<double> K4, pose, X, uv; std::vector<float> isg;
    if (fread(&n, 4, 1, f) != 1 || !rd(f, K4, 4) || !rd(f, pose, 7) || !rd(f, X, 3 * (size_t)n) || !rd(f, uv, 2 * (size_t)n) || !rd(f, isg, n)) { fclose(f); return 3; }
    fclose(f);
    total++;
    std::vector<double> mine = pose; std::vector<uint8_t> outl(n); int ninl = 0; ba_summary s;
    if (ba_pose_optimization(K4.data(), mine.data(), X.data(), uv.data(), isg.data(), n, outl.data(), &ninl, &s)) { fprintf(stderr, "ba_pose_optimization: %s\n", orbhip_last_error()); return 3; }
    double dpose = -1, dcost = -1; int it_ref = -1;
#ifndef PIN_SYNTAX_ONLY
    Eigen::Matrix3d K = Eigen::Matrix3d::Identity(); K(0, 0) = K4[0]; K(1, 1) = K4[1]; K(0, 2) = K4[2]; K(1, 2) = K4[3];
    Eigen::Vector3d t(pose[0], pose[1], pose[2]); Eigen::Quaterniond q(pose[6], pose[3], pose[4], pose[5]);
    ceres::Problem problem;
    ceres::LossFunction* loss = new ceres::HuberLoss(sqrt(5.991));
    ceres::LocalParameterization* qp = new ceres::EigenQuaternionParameterization;
    for (int i = 0; i < n; i++) {
      Eigen::Matrix2d info = Eigen::Matrix2d::Identity() * isg[i];
      ceres::CostFunction* c = ORB_SLAM2::PoseErrorTerm::Create(K, Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]), Eigen::Vector3d(X[3 * i], X[3 * i + 1], X[3 * i + 2]), info);
      problem.AddResidualBlock(c, loss, t.data(), q.coeffs().data());
      problem.SetParameterization(q.coeffs().data(), qp);
    }
    ceres::Solver::Options o; o.max_num_iterations = 100; o.linear_solver_type = ceres::DENSE_QR;
    ceres::Solver::Summary sum; ceres::Solve(o, &problem, &sum);
    q.normalize();
    const double ref[7] = {t[0], t[1], t[2], q.x(), q.y(), q.z(), q.w()};
    dpose = maxdiff(ref, mine.data(), 7); dcost = std::fabs(sum.final_cost - s.final_cost) / std::fmax(sum.final_cost, 1e-300);
    it_ref = (int)sum.iterations.size() - 1;
#endif
    const bool ok = dpose <= 1e-7 && dcost <= 1e-9 && it_ref == s.iterations;
    printf("pose %d (%d observations): iterations %d / %d (here / reference), relative cost difference %.2e, pose difference %.2e -> %s\n", k, n, s.iterations, it_ref, dcost, dpose, ok ? "ok" : "DIFFERENT");
    bad += ok ? 0 : 1;
  }
  for (int k = 0;; k++) {
    char name[64]; snprintf(name, sizeof(name), "/ba_%03d.bin", k);
    FILE* f = fopen((std::string(argv[1]) + name).c_str(), "rb");
    if (!f) break;
    int32_t hd[4]; std::vector<double> K4, poses, pts, ouv; std::vector<uint8_t> fixed; std::vector<int32_t> oc, op; std::vector<float> isg;
    if (fread(hd, 4, 4, f) != 4) { fclose(f); return 3; }
    const int ncam = hd[0], npts = hd[1], nobs = hd[2], iters = hd[3];
    if (!rd(f, K4, 4 * (size_t)ncam) || !rd(f, poses, 7 * (size_t)ncam) || !rd(f, fixed, ncam) || !rd(f, pts, 3 * (size_t)npts) || !rd(f, oc, nobs) || !rd(f, op, nobs) ||
        !rd(f, ouv, 2 * (size_t)nobs) || !rd(f, isg, nobs)) { fclose(f); return 3; }
    fclose(f);
    total++;
    std::vector<double> mp = poses, mx = pts, w(nobs); std::vector<uint8_t> rob(nobs, 1);
    for (int i = 0; i < nobs; i++) w[i] = (double)isg[i];                       // information = invSigma2, un-square-rooted (src/CeresOptimizer.cc:120-123)
    ba_options o; o.max_iterations = iters; o.huber_delta = sqrt(5.991); o.fix_points = 0; o.stop_flag = nullptr;
    ba_summary s;
    if (ba_solve(K4.data(), mp.data(), fixed.data(), ncam, mx.data(), npts, oc.data(), op.data(), ouv.data(), w.data(), rob.data(), nobs, &o, &s)) { fprintf(stderr, "ba_solve: %s\n", orbhip_last_error()); return 3; }
    double dpose = -1, dpt = -1, dcost = -1; int it_ref = -1;
#ifndef PIN_SYNTAX_ONLY
    std::vector<Eigen::Vector3d> T(ncam); std::vector<Eigen::Quaterniond> Q(ncam);
    for (int c = 0; c < ncam; c++) { T[c] = Eigen::Vector3d(poses[7 * c], poses[7 * c + 1], poses[7 * c + 2]); Q[c] = Eigen::Quaterniond(poses[7 * c + 6], poses[7 * c + 3], poses[7 * c + 4], poses[7 * c + 5]); }
    std::vector<double> rx = pts;
    ceres::Problem problem;
    ceres::LossFunction* loss = new ceres::HuberLoss(sqrt(5.991));
    ceres::LocalParameterization* qp = new ceres::EigenQuaternionParameterization;
    for (int i = 0; i < nobs; i++) {
      const int c = oc[i], p = op[i];
      Eigen::Matrix3d K = Eigen::Matrix3d::Identity(); K(0, 0) = K4[4 * c]; K(1, 1) = K4[4 * c + 1]; K(0, 2) = K4[4 * c + 2]; K(1, 2) = K4[4 * c + 3];
      Eigen::Matrix2d info = Eigen::Matrix2d::Identity() * isg[i];
      ceres::CostFunction* cf = ORB_SLAM2::PoseGraph3dErrorTerm::Create(K, Eigen::Vector2d(ouv[2 * i], ouv[2 * i + 1]), info);
      problem.AddResidualBlock(cf, loss, T[c].data(), Q[c].coeffs().data(), rx.data() + 3 * p);
      problem.SetParameterization(Q[c].coeffs().data(), qp);
      if (fixed[c]) { problem.SetParameterBlockConstant(T[c].data()); problem.SetParameterBlockConstant(Q[c].coeffs().data()); }
    }
    ceres::Solver::Options so; so.max_num_iterations = iters; so.linear_solver_type = ceres::DENSE_SCHUR;
    ceres::Solver::Summary sum; ceres::Solve(so, &problem, &sum);
    std::vector<double> rp(7 * (size_t)ncam);
    for (int c = 0; c < ncam; c++) { rp[7 * c] = T[c][0]; rp[7 * c + 1] = T[c][1]; rp[7 * c + 2] = T[c][2]; rp[7 * c + 3] = Q[c].x(); rp[7 * c + 4] = Q[c].y(); rp[7 * c + 5] = Q[c].z(); rp[7 * c + 6] = Q[c].w(); }
    dpose = maxdiff(rp.data(), mp.data(), rp.size()); dpt = maxdiff(rx.data(), mx.data(), rx.size());
    dcost = std::fabs(sum.final_cost - s.final_cost) / std::fmax(sum.final_cost, 1e-300);
    it_ref = (int)sum.iterations.size() - 1;
#endif
    const bool ok = dpose <= 1e-7 && dcost <= 1e-9 && it_ref == s.iterations;
    printf("ba %d (%d cameras, %d points, %d observations): iterations %d / %d, relative cost difference %.2e, poses %.2e, points %.2e -> %s\n", k, ncam, npts, nobs,
           s.iterations, it_ref, dcost, dpose, dpt, ok ? "ok" : "DIFFERENT");
    bad += ok ? 0 : 1;
  }
  if (!total) { fprintf(stderr, "no pose_000.bin / ba_000.bin under %s (python tools/pin/make_cases.py <dir>)\n", argv[1]); return 2; }
  printf("%s: %d of %d problems within the bars\n", bad ? "DIFFERENT" : "PINNED", total - bad, total);
  return bad ? 1 : 0;
}
