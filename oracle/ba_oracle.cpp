// ============================================================================
// oracle/ba_oracle.cpp -- CPU restatement of the reference bundle-adjustment path.
//
// TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header for the rule).
//
// PARITY STATUS: "parity unpinned".  The reference delegates the solve to
// Ceres Solver (un-vendored, unpinned, < 2.2: uses LocalParameterization), which
// is absent here, so neither the reference nor Ceres can be run.  This file
// restates
//   * the cost functors PoseGraph3dErrorTerm / PoseErrorTerm
//     (include/CeresOptimizer.h:56-166),
//   * CeresOptimizer::{PoseOptimization, BundleAdjustment, LocalBundleAdjustment,
//     CheckOutlier(s)} as flattened-array functions (src/CeresOptimizer.cc:49-599),
//   * Ceres 1.14's trust-region Levenberg-Marquardt with its default options,
//     HuberLoss + Triggs corrector, EigenQuaternionParameterization and Jacobi
//     scaling,
// in fp64, single-threaded.  The linear system is solved exactly through the
// Schur complement (mathematically identical to the reference's
// SPARSE_NORMAL_CHOLESKY).  It is pinned only by analytic checks in
// tests/ (finite-difference Jacobians, zero-noise convergence, scipy
// least_squares on loss-free problems).
// ============================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>
#include <thread>
#include <functional>
#include <atomic>

namespace {

// Worker threads for the per-observation evaluation and the Schur elimination (the reference runs LocalBA with
// options.num_threads = 4, src/CeresOptimizer.cc:516).  Every sum keeps its single-thread order (a thread owns whole
// rows of the reduced system; costs and gradients are accumulated sequentially from per-observation values), so results
// are bit-identical for any thread count.
int g_ba_threads = 1;
void parallel_for(int n, const std::function<void(int, int, int)>& body) {   // body(lo, hi, tid)
  int T = std::max(1, std::min(g_ba_threads, n / 64 + 1));
  if (T == 1) { body(0, n, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(body, (int)((int64_t)n * t / T), (int)((int64_t)n * (t + 1) / T), t);
  for (auto& x : th) x.join();
}

struct Opts {
  int max_iters;          // options.max_num_iterations
  double huber_delta;     // sqrt(5.991); applied to observations whose robust flag is set
  int fix_points;         // 1 = points are constants (PoseOptimization)
  const volatile uint8_t* stop;   // StopFlagCallback (may be null)
};
struct Summary {
  double initial_cost, final_cost;
  int iterations;          // number of LM iterations attempted (iteration 0 excluded)
  int successful_steps;
  int termination;         // 0 no-convergence(max iters) 1 gradient tol 2 parameter tol 3 function tol
                           // 4 user stop 5 failure (too many invalid steps) 6 min trust-region radius
  double final_radius;
};

// Eigen's q*v for q stored [x,y,z,w]:  v + w*(2 qv x v) + qv x (2 qv x v)
inline void quat_rotate(const double q[4], const double v[3], double out[3]) {
  double uvx = 2 * (q[1] * v[2] - q[2] * v[1]);
  double uvy = 2 * (q[2] * v[0] - q[0] * v[2]);
  double uvz = 2 * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  out[1] = v[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  out[2] = v[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}
// rotation matrix of a (near-)unit quaternion [x,y,z,w] (Eigen toRotationMatrix)
inline void quat_to_R(const double q[4], double R[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
         tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// EigenQuaternionParameterization::Plus: q+ = dq (x) q, delta = half-angle vector
inline void quat_plus(const double q[4], const double d[3], double out[4]) {
  double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n > 0.0) {
    double s = std::sin(n) / n;
    double dx = s * d[0], dy = s * d[1], dz = s * d[2], dw = std::cos(n);
    // Hamilton product dq * q
    out[3] = dw * q[3] - dx * q[0] - dy * q[1] - dz * q[2];
    out[0] = dw * q[0] + dx * q[3] + dy * q[2] - dz * q[1];
    out[1] = dw * q[1] - dx * q[2] + dy * q[3] + dz * q[0];
    out[2] = dw * q[2] + dx * q[1] - dy * q[0] + dz * q[3];
  } else {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  }
}

struct Problem {
  int ncam, npts, nobs;
  const double* K4;           // per cam fx,fy,cx,cy
  const uint8_t* cam_fixed;
  const int32_t* obs_cam; const int32_t* obs_pt;
  const double* obs_uv; const double* obs_w; const uint8_t* obs_robust;
  Opts opt;
  // derived
  std::vector<int> cam_col;   // free cam -> column block index, else -1
  std::vector<int> pt_col;    // used point -> block index, else -1
  int nfc = 0, nfp = 0;
};

// residual + (optionally) jacobians of one observation at (pose7, X).
// r = w*(uv - proj);  Jc (2x6: t then half-angle delta), Jp (2x3).  Robust
// correction applied: r, J scaled by sqrt(rho'), returns rho.
inline double eval_obs(const Problem& P, int i, const double* pose7, const double* X, double r[2],
                       double* Jc, double* Jp) {
  const double* K = P.K4 + 4 * P.obs_cam[i];
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double* t = pose7; const double* q = pose7 + 3;
  double RX[3];
  quat_rotate(q, X, RX);
  double p[3] = {RX[0] + t[0], RX[1] + t[1], RX[2] + t[2]};
  // projected = K * p ; residual = obs - projected.xy / projected.z
  double u = (fx * p[0] + cx * p[2]) / p[2];
  double v = (fy * p[1] + cy * p[2]) / p[2];
  const double w = P.obs_w[i];
  r[0] = w * (P.obs_uv[2 * i] - u);
  r[1] = w * (P.obs_uv[2 * i + 1] - v);
  double s = r[0] * r[0] + r[1] * r[1];
  double rho0 = s, rho1 = 1.0;
  if (P.obs_robust[i]) {
    const double a = P.opt.huber_delta, b = a * a;
    if (s > b) {
      double rr = std::sqrt(s);
      rho0 = 2 * a * rr - b;
      rho1 = std::max(std::numeric_limits<double>::min(), a / rr);
    }
  }
  const double sq = std::sqrt(rho1);
  if (Jc || Jp) {
    const double iz = 1.0 / p[2];
    // d(u,v)/dp
    double Jpi[6] = {fx * iz, 0, -fx * p[0] * iz * iz, 0, fy * iz, -fy * p[1] * iz * iz};
    const double ws = -w * sq;    // dr/dp = -w * Jpi, then robust scaling
    if (Jc) {
      // dr/dt = -w Jpi ; dr/ddelta = -w Jpi * (-2 [RX]x) = 2 w Jpi [RX]x
      // [RX]x = [[0,-z,y],[z,0,-x],[-y,x,0]]
      double A[9] = {0, -RX[2], RX[1], RX[2], 0, -RX[0], -RX[1], RX[0], 0};
      for (int a = 0; a < 2; a++) {
        for (int c = 0; c < 3; c++) Jc[a * 6 + c] = ws * Jpi[a * 3 + c];
        for (int c = 0; c < 3; c++) {
          double acc = 0;
          for (int k = 0; k < 3; k++) acc += Jpi[a * 3 + k] * A[k * 3 + c];
          Jc[a * 6 + 3 + c] = -2.0 * ws * acc;
        }
      }
    }
    if (Jp) {
      double R[9];
      // d(q*X)/dX for Eigen's formula with a unit quaternion = R(q)
      quat_to_R(q, R);
      for (int a = 0; a < 2; a++)
        for (int c = 0; c < 3; c++) {
          double acc = 0;
          for (int k = 0; k < 3; k++) acc += Jpi[a * 3 + k] * R[k * 3 + c];
          Jp[a * 3 + c] = ws * acc;
        }
    }
  }
  r[0] *= sq; r[1] *= sq;
  return rho0;
}

double total_cost(const Problem& P, const double* poses, const double* pts) {
  std::vector<double> rho(P.nobs);
  parallel_for(P.nobs, [&](int lo, int hi, int) {
    for (int i = lo; i < hi; i++) {
      double r[2];
      rho[i] = eval_obs(P, i, poses + 7 * P.obs_cam[i], pts + 3 * P.obs_pt[i], r, nullptr, nullptr);
    }
  });
  double c = 0;
  for (int i = 0; i < P.nobs; i++) c += 0.5 * rho[i];
  return c;
}

// dense in-place Cholesky (lower) of n x n row-major A; returns false if not PD.  With worker threads (g_ba_threads > 1)
// the rows below the diagonal of a column are shared out among a team that meets at a barrier after every column; each
// entry is still one sequential dot product over k, so the factor is bit-identical for any thread count.
bool cholesky(std::vector<double>& A, int n) {
  const int T = (n >= 256) ? std::max(1, g_ba_threads) : 1;
  std::atomic<int> arrived{0}, phase{0};
  std::atomic<bool> failed{false};
  auto barrier = [&](int& local_phase) {
    if (T == 1) return;
    local_phase ^= 1;
    if (arrived.fetch_add(1) == T - 1) { arrived.store(0); phase.store(local_phase); }
    else while (phase.load() != local_phase) std::this_thread::yield();
  };
  auto team = [&](int tid) {
    int local_phase = 0;
    for (int j = 0; j < n; j++) {
      double d = A[(size_t)j * n + j];
      for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0) || !std::isfinite(d)) { failed.store(true); return; }     // (every member sees the same d)
      d = std::sqrt(d);
      const double* aj = &A[(size_t)j * n];
      for (int i = j + 1 + tid; i < n; i += T) {
        double s = A[(size_t)i * n + j];
        const double* ai = &A[(size_t)i * n];
        for (int k = 0; k < j; k++) s -= ai[k] * aj[k];
        A[(size_t)i * n + j] = s / d;
      }
      barrier(local_phase);                       // column j complete (the diagonal is written after everyone has used it)
      if (tid == 0) A[(size_t)j * n + j] = d;
    }
  };
  if (T == 1) team(0);
  else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(team, t); for (auto& x : th) x.join(); }
  return !failed.load();
}
void chol_solve(const std::vector<double>& L, int n, std::vector<double>& b) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}
inline bool inv3_sym(const double* C, double* Ci) {
  double a = C[0], b = C[1], c = C[2], d = C[4], e = C[5], f = C[8];
  double A = d * f - e * e, B = -(b * f - c * e), Cc = b * e - c * d;
  double det = a * A + b * B + c * Cc;
  if (!(det != 0.0) || !std::isfinite(det)) return false;
  double id = 1.0 / det;
  Ci[0] = A * id; Ci[1] = B * id; Ci[2] = Cc * id;
  Ci[3] = Ci[1]; Ci[4] = (a * f - c * c) * id; Ci[5] = -(a * e - b * c) * id;
  Ci[6] = Ci[2]; Ci[7] = Ci[5]; Ci[8] = (a * d - b * b) * id;
  return true;
}

// One full Ceres-style trust-region LM solve.
int lm_solve(Problem& P, double* poses, double* pts, Summary* sum) {
  const int nobs = P.nobs;
  // reduced program: free cams / points that appear in at least one observation
  P.cam_col.assign(P.ncam, -1); P.pt_col.assign(P.npts, -1);
  std::vector<uint8_t> cam_used(P.ncam, 0), pt_used(P.npts, 0);
  for (int i = 0; i < nobs; i++) { cam_used[P.obs_cam[i]] = 1; pt_used[P.obs_pt[i]] = 1; }
  P.nfc = P.nfp = 0;
  for (int c = 0; c < P.ncam; c++) if (cam_used[c] && !P.cam_fixed[c]) P.cam_col[c] = P.nfc++;
  if (!P.opt.fix_points) for (int p = 0; p < P.npts; p++) if (pt_used[p]) P.pt_col[p] = P.nfp++;
  const int nc6 = 6 * P.nfc, np3 = 3 * P.nfp, ncols = nc6 + np3;

  std::vector<double> r(2 * (size_t)nobs), Jc(12 * (size_t)nobs), Jp(6 * (size_t)nobs), rho_obs(nobs);
  std::vector<double> scale(ncols, 1.0), grad(ncols), diag(ncols), step(ncols);
  std::vector<double> cand_poses(poses, poses + 7 * (size_t)P.ncam), cand_pts(pts, pts + 3 * (size_t)P.npts);
  double radius = 1e4, decrease_factor = 2.0;
  double x_cost = 0, x_norm = 0;
  int iteration = 0, invalid_steps = 0;
  sum->successful_steps = 0; sum->termination = 0;

  auto compute_x_norm = [&]() {
    double s = 0;
    for (int c = 0; c < P.ncam; c++) if (P.cam_col[c] >= 0) for (int k = 0; k < 7; k++) s += poses[7 * c + k] * poses[7 * c + k];
    for (int p = 0; p < P.npts; p++) if (P.pt_col[p] >= 0) for (int k = 0; k < 3; k++) s += pts[3 * p + k] * pts[3 * p + k];
    return std::sqrt(s);
  };
  // evaluate cost/residuals/jacobians at x, gradient, (first time) Jacobi scaling, scale J
  auto evaluate_at_x = [&](bool first) -> double {
    x_cost = 0;
    std::fill(grad.begin(), grad.end(), 0.0);
    parallel_for(nobs, [&](int lo, int hi, int) {
      for (int i = lo; i < hi; i++) {
        int c = P.obs_cam[i], p = P.obs_pt[i];
        rho_obs[i] = eval_obs(P, i, poses + 7 * c, pts + 3 * p, &r[2 * i], P.cam_col[c] >= 0 ? &Jc[12 * (size_t)i] : nullptr,
                              P.pt_col[p] >= 0 ? &Jp[6 * (size_t)i] : nullptr);
      }
    });
    for (int i = 0; i < nobs; i++) {
      int cc = P.cam_col[P.obs_cam[i]], pc = P.pt_col[P.obs_pt[i]];
      x_cost += 0.5 * rho_obs[i];
      if (cc >= 0) for (int k = 0; k < 6; k++) grad[6 * cc + k] += Jc[12 * (size_t)i + k] * r[2 * i] + Jc[12 * (size_t)i + 6 + k] * r[2 * i + 1];
      if (pc >= 0) for (int k = 0; k < 3; k++) grad[nc6 + 3 * pc + k] += Jp[6 * (size_t)i + k] * r[2 * i] + Jp[6 * (size_t)i + 3 + k] * r[2 * i + 1];
    }
    if (first) {
      std::vector<double> n2(ncols, 0.0);
      for (int i = 0; i < nobs; i++) {
        int cc = P.cam_col[P.obs_cam[i]], pc = P.pt_col[P.obs_pt[i]];
        if (cc >= 0) for (int k = 0; k < 6; k++) n2[6 * cc + k] += Jc[12 * (size_t)i + k] * Jc[12 * (size_t)i + k] + Jc[12 * (size_t)i + 6 + k] * Jc[12 * (size_t)i + 6 + k];
        if (pc >= 0) for (int k = 0; k < 3; k++) n2[nc6 + 3 * pc + k] += Jp[6 * (size_t)i + k] * Jp[6 * (size_t)i + k] + Jp[6 * (size_t)i + 3 + k] * Jp[6 * (size_t)i + 3 + k];
      }
      for (int j = 0; j < ncols; j++) scale[j] = 1.0 / (1.0 + std::sqrt(n2[j]));
    }
    for (int i = 0; i < nobs; i++) {
      int cc = P.cam_col[P.obs_cam[i]], pc = P.pt_col[P.obs_pt[i]];
      if (cc >= 0) for (int a = 0; a < 2; a++) for (int k = 0; k < 6; k++) Jc[12 * (size_t)i + 6 * a + k] *= scale[6 * cc + k];
      if (pc >= 0) for (int a = 0; a < 2; a++) for (int k = 0; k < 3; k++) Jp[6 * (size_t)i + 3 * a + k] *= scale[nc6 + 3 * pc + k];
    }
    // gradient max norm = || x - Plus(x, -g) ||_inf   (ambient space)
    double gmax = 0;
    for (int c = 0; c < P.ncam; c++) {
      int cc = P.cam_col[c];
      if (cc < 0) continue;
      for (int k = 0; k < 3; k++) gmax = std::max(gmax, std::fabs(grad[6 * cc + k]));
      double d[3] = {-grad[6 * cc + 3], -grad[6 * cc + 4], -grad[6 * cc + 5]}, qn[4];
      quat_plus(poses + 7 * c + 3, d, qn);
      for (int k = 0; k < 4; k++) gmax = std::max(gmax, std::fabs(poses[7 * c + 3 + k] - qn[k]));
    }
    for (int j = nc6; j < ncols; j++) gmax = std::max(gmax, std::fabs(grad[j]));
    return gmax;
  };

  x_norm = compute_x_norm();
  double gmax = evaluate_at_x(true);
  sum->initial_cost = x_cost;
  bool done = false;
  if (gmax <= 1e-10) { sum->termination = 1; done = true; }
  if (!done && P.opt.stop && *P.opt.stop) { sum->termination = 4; done = true; }   // callback after iteration 0

  std::vector<double> B, C, S, rhs, Cinv, yc, yp;
  while (!done) {
    if (iteration >= P.opt.max_iters) { sum->termination = 0; break; }
    if (radius <= 1e-32) { sum->termination = 6; break; }
    iteration++;
    // ---- normal equations of the scaled Jacobian -------------------------------
    B.assign((size_t)36 * P.nfc, 0.0); C.assign((size_t)9 * P.nfp, 0.0);
    std::vector<double> gs(ncols, 0.0);
    for (int i = 0; i < nobs; i++) {
      int cc = P.cam_col[P.obs_cam[i]], pc = P.pt_col[P.obs_pt[i]];
      const double* jc = &Jc[12 * (size_t)i]; const double* jp = &Jp[6 * (size_t)i];
      if (cc >= 0) {
        for (int a = 0; a < 6; a++) {
          for (int b = 0; b < 6; b++) B[36 * (size_t)cc + 6 * a + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
          gs[6 * cc + a] += jc[a] * r[2 * i] + jc[6 + a] * r[2 * i + 1];
        }
      }
      if (pc >= 0) {
        for (int a = 0; a < 3; a++) {
          for (int b = 0; b < 3; b++) C[9 * (size_t)pc + 3 * a + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
          gs[nc6 + 3 * pc + a] += jp[a] * r[2 * i] + jp[3 + a] * r[2 * i + 1];
        }
      }
    }
    for (int c = 0; c < P.nfc; c++) for (int k = 0; k < 6; k++) diag[6 * c + k] = B[36 * (size_t)c + 7 * k];
    for (int p = 0; p < P.nfp; p++) for (int k = 0; k < 3; k++) diag[nc6 + 3 * p + k] = C[9 * (size_t)p + 4 * k];
    for (int j = 0; j < ncols; j++) diag[j] = std::min(std::max(diag[j], 1e-6), 1e32) / radius;   // = D^2
    // ---- Schur complement ---------------------------------------------------
    S.assign((size_t)nc6 * nc6, 0.0); rhs.assign(nc6, 0.0);
    for (int c = 0; c < P.nfc; c++)
      for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) S[(size_t)(6 * c + a) * nc6 + 6 * c + b] = B[36 * (size_t)c + 6 * a + b];
        S[(size_t)(6 * c + a) * nc6 + 6 * c + a] += diag[6 * c + a];
        rhs[6 * c + a] = gs[6 * c + a];
      }
    bool ok = true;
    Cinv.assign((size_t)9 * P.nfp, 0.0);
    for (int p = 0; p < P.nfp; p++) {
      double Cp[9];
      for (int k = 0; k < 9; k++) Cp[k] = C[9 * (size_t)p + k];
      for (int k = 0; k < 3; k++) Cp[4 * k] += diag[nc6 + 3 * p + k];
      if (!inv3_sym(Cp, &Cinv[9 * (size_t)p])) ok = false;
    }
    // observations grouped by point
    std::vector<std::vector<int>> by_pt(P.nfp);
    if (P.nfp) for (int i = 0; i < nobs; i++) { int pc = P.pt_col[P.obs_pt[i]]; if (pc >= 0) by_pt[pc].push_back(i); }
    // per observation: E_i = Jc^T Jp (6x3) and E_i * Cinv (6x3), at the observation's position in its point's list
    std::vector<size_t> pt_base(P.nfp + 1, 0);
    for (int p = 0; p < P.nfp; p++) pt_base[p + 1] = pt_base[p] + by_pt[p].size();
    std::vector<double> E(18 * pt_base[P.nfp], 0.0), EC(18 * pt_base[P.nfp], 0.0);
    std::vector<int> ca_flat(pt_base[P.nfp], -1);            // reduced camera column of every observation, point-major
    if (ok) parallel_for(P.nfp, [&](int plo, int phi, int) {
      for (int p = plo; p < phi; p++) {
        const double* Ci = &Cinv[9 * (size_t)p];
        const std::vector<int>& L = by_pt[p];
        double* Ep = &E[18 * pt_base[p]]; double* ECp = &EC[18 * pt_base[p]];
        for (size_t a = 0; a < L.size(); a++) {
          int i = L[a];
          int cc = P.cam_col[P.obs_cam[i]];
          ca_flat[pt_base[p] + a] = cc;
          if (cc < 0) continue;
          const double* jc = &Jc[12 * (size_t)i]; const double* jp = &Jp[6 * (size_t)i];
          for (int u = 0; u < 6; u++) for (int v = 0; v < 3; v++) Ep[18 * a + 3 * u + v] = jc[u] * jp[v] + jc[6 + u] * jp[3 + v];
          for (int u = 0; u < 6; u++) for (int v = 0; v < 3; v++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += Ep[18 * a + 3 * u + k] * Ci[3 * k + v];
            ECp[18 * a + 3 * u + v] = acc;
          }
        }
      }
    });
    // S -= sum_p E Cinv E^T, rhs -= sum_p E Cinv g_p: a thread owns a contiguous range of camera rows (tracks are runs of
    // consecutive keyframes, so it mostly meets its own points) and walks the points in ascending order: every entry is
    // summed in the single-thread order
    const int T = std::max(1, g_ba_threads);
    auto schur_rows = [&](int tid) {
      const int nfc1 = std::max(P.nfc, 1);
      for (int p = 0; p < P.nfp; p++) {
        const size_t base = pt_base[p], len = pt_base[p + 1] - base;
        const int* cap = &ca_flat[base];
        const double* Ep = &E[18 * base]; const double* ECp = &EC[18 * base];
        const double* gp = &gs[nc6 + 3 * p];
        for (size_t a = 0; a < len; a++) {
          const int ca = cap[a];
          if (ca < 0 || (int)((int64_t)ca * T / nfc1) != tid) continue;
          for (int u = 0; u < 6; u++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += ECp[18 * a + 3 * u + k] * gp[k];
            rhs[6 * ca + u] -= acc;
          }
          for (size_t b = 0; b < len; b++) {
            const int cb = cap[b];
            if (cb < 0) continue;
            for (int u = 0; u < 6; u++) for (int v = 0; v < 6; v++) {
              double acc = 0;
              for (int k = 0; k < 3; k++) acc += ECp[18 * a + 3 * u + k] * Ep[18 * b + 3 * v + k];
              S[(size_t)(6 * ca + u) * nc6 + 6 * cb + v] -= acc;
            }
          }
        }
      }
    };
    if (ok) {
      if (T == 1) schur_rows(0);
      else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(schur_rows, t); for (auto& x : th) x.join(); }
    }
    if (ok && nc6) ok = cholesky(S, nc6);
    double model_cost_change = 0;
    if (ok) {
      yc = rhs;
      if (nc6) chol_solve(S, nc6, yc);
      yp.assign(np3, 0.0);
      for (int p = 0; p < P.nfp; p++) {
        double t[3] = {gs[nc6 + 3 * p], gs[nc6 + 3 * p + 1], gs[nc6 + 3 * p + 2]};
        for (int i : by_pt[p]) {
          int cc = P.cam_col[P.obs_cam[i]];
          if (cc < 0) continue;
          const double* jc = &Jc[12 * (size_t)i]; const double* jp = &Jp[6 * (size_t)i];
          // E^T yc = Jp^T (Jc yc)
          double m0 = 0, m1 = 0;
          for (int u = 0; u < 6; u++) { m0 += jc[u] * yc[6 * cc + u]; m1 += jc[6 + u] * yc[6 * cc + u]; }
          for (int v = 0; v < 3; v++) t[v] -= jp[v] * m0 + jp[3 + v] * m1;
        }
        const double* Ci = &Cinv[9 * (size_t)p];
        for (int v = 0; v < 3; v++) yp[3 * p + v] = Ci[3 * v] * t[0] + Ci[3 * v + 1] * t[1] + Ci[3 * v + 2] * t[2];
      }
      for (int j = 0; j < nc6; j++) step[j] = -yc[j];
      for (int j = 0; j < np3; j++) step[nc6 + j] = -yp[j];
      // model_cost_change = -(J s).(r + J s / 2)
      for (int i = 0; i < nobs; i++) {
        int cc = P.cam_col[P.obs_cam[i]], pc = P.pt_col[P.obs_pt[i]];
        double m0 = 0, m1 = 0;
        if (cc >= 0) for (int u = 0; u < 6; u++) { m0 += Jc[12 * (size_t)i + u] * step[6 * cc + u]; m1 += Jc[12 * (size_t)i + 6 + u] * step[6 * cc + u]; }
        if (pc >= 0) for (int v = 0; v < 3; v++) { m0 += Jp[6 * (size_t)i + v] * step[nc6 + 3 * pc + v]; m1 += Jp[6 * (size_t)i + 3 + v] * step[nc6 + 3 * pc + v]; }
        model_cost_change -= m0 * (r[2 * i] + m0 / 2) + m1 * (r[2 * i + 1] + m1 / 2);
      }
    }
    if (!ok || !(model_cost_change > 0.0)) {
      // HandleInvalidStep
      if (++invalid_steps >= 5) { sum->termination = 5; break; }
      radius /= decrease_factor; decrease_factor *= 2;
      if (P.opt.stop && *P.opt.stop) { sum->termination = 4; break; }
      continue;
    }
    invalid_steps = 0;
    // ---- candidate = Plus(x, step * scale) -----------------------------------
    double step_norm2 = 0;
    for (int c = 0; c < P.ncam; c++) {
      int cc = P.cam_col[c];
      for (int k = 0; k < 7; k++) cand_poses[7 * c + k] = poses[7 * c + k];
      if (cc < 0) continue;
      for (int k = 0; k < 3; k++) cand_poses[7 * c + k] = poses[7 * c + k] + step[6 * cc + k] * scale[6 * cc + k];
      double d[3] = {step[6 * cc + 3] * scale[6 * cc + 3], step[6 * cc + 4] * scale[6 * cc + 4], step[6 * cc + 5] * scale[6 * cc + 5]};
      quat_plus(poses + 7 * c + 3, d, &cand_poses[7 * c + 3]);
      for (int k = 0; k < 7; k++) { double e = poses[7 * c + k] - cand_poses[7 * c + k]; step_norm2 += e * e; }
    }
    for (int p = 0; p < P.npts; p++) {
      int pc = P.pt_col[p];
      for (int k = 0; k < 3; k++) cand_pts[3 * p + k] = pts[3 * p + k];
      if (pc < 0) continue;
      for (int k = 0; k < 3; k++) {
        cand_pts[3 * p + k] = pts[3 * p + k] + step[nc6 + 3 * pc + k] * scale[nc6 + 3 * pc + k];
        double e = pts[3 * p + k] - cand_pts[3 * p + k]; step_norm2 += e * e;
      }
    }
    double cand_cost = total_cost(P, cand_poses.data(), cand_pts.data());
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached
    if (std::sqrt(step_norm2) <= 1e-8 * (x_norm + 1e-8)) { sum->termination = 2; break; }
    // FunctionToleranceReached
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum->termination = 3; break; }
    double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > 1e-3) {
      std::memcpy(poses, cand_poses.data(), sizeof(double) * 7 * P.ncam);
      std::memcpy(pts, cand_pts.data(), sizeof(double) * 3 * P.npts);
      x_norm = compute_x_norm();
      gmax = evaluate_at_x(false);
      sum->successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0;
      if (gmax <= 1e-10) { sum->termination = 1; break; }
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
    if (P.opt.stop && *P.opt.stop) { sum->termination = 4; break; }
  }
  sum->final_cost = x_cost;
  sum->iterations = iteration;
  sum->final_radius = radius;
  return 0;
}

}  // namespace

extern "C" {

struct orc_ba_opts { int max_iters; double huber_delta; int fix_points; const volatile uint8_t* stop; };
struct orc_ba_summary { double initial_cost, final_cost; int iterations, successful_steps, termination; double final_radius; };

// Generic reprojection BA (BundleAdjustment, src/CeresOptimizer.cc:59-225, flattened).
// poses7 = [tx,ty,tz,qx,qy,qz,qw] per camera (src/MatEigenConverter.cc:66-85 layout).
int orc_ba_solve(const double* K4, double* poses7, const uint8_t* cam_fixed, int ncam, double* pts3, int npts,
                 const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, const double* obs_w,
                 const uint8_t* obs_robust, int nobs, const orc_ba_opts* o, orc_ba_summary* s) {
  Problem P;
  P.ncam = ncam; P.npts = npts; P.nobs = nobs; P.K4 = K4; P.cam_fixed = cam_fixed;
  P.obs_cam = obs_cam; P.obs_pt = obs_pt; P.obs_uv = obs_uv; P.obs_w = obs_w; P.obs_robust = obs_robust;
  P.opt.max_iters = o->max_iters; P.opt.huber_delta = o->huber_delta; P.opt.fix_points = o->fix_points; P.opt.stop = o->stop;
  Summary S;
  int rc = lm_solve(P, poses7, pts3, &S);
  if (s) { s->initial_cost = S.initial_cost; s->final_cost = S.final_cost; s->iterations = S.iterations;
           s->successful_steps = S.successful_steps; s->termination = S.termination; s->final_radius = S.final_radius; }
  return rc;
}

// CheckOutlier (src/CeresOptimizer.cc:227-241): chi2 = |e|^2 * inv_sigma2 > thres.  depth (optional out) = z in camera.
int orc_check_outlier(const double* K4, const double* pose7, const double* X, const double* uv, double inv_sigma2,
                      double thres, double* depth) {
  double RX[3];
  quat_rotate(pose7 + 3, X, RX);
  double p[3] = {RX[0] + pose7[0], RX[1] + pose7[1], RX[2] + pose7[2]};
  double u = (K4[0] * p[0] + K4[2] * p[2]) / p[2], v = (K4[1] * p[1] + K4[3] * p[2]) / p[2];
  double eu = uv[0] - u, ev = uv[1] - v;
  if (depth) *depth = p[2];
  return ((eu * eu + ev * ev) * inv_sigma2 > thres) ? 1 : 0;
}

// PoseOptimization (src/CeresOptimizer.cc:275-342) on flattened arrays.
// Returns n_initial - n_bad, 0 (pose untouched) if n < 3.  pose7 is updated with
// the NORMALISED quaternion (":336").  outlier[i] set by CheckOutliers.
int orc_pose_optimization(const double* K4, double* pose7, const double* Xw, const double* uv,
                          const float* inv_sigma2, int n, uint8_t* outlier, orc_ba_summary* s) {
  if (n < 3) return 0;
  std::vector<int32_t> oc(n, 0), op(n);
  std::vector<double> w(n);
  std::vector<uint8_t> rob(n, 1);
  for (int i = 0; i < n; i++) { op[i] = i; w[i] = (double)inv_sigma2[i]; }   // F7: weight = invSigma2 itself
  uint8_t fixed = 0;
  std::vector<double> pts(Xw, Xw + 3 * (size_t)n);
  orc_ba_opts o{100, std::sqrt(5.991), 1, nullptr};
  orc_ba_solve(K4, pose7, &fixed, 1, pts.data(), n, oc.data(), op.data(), uv, w.data(), rob.data(), n, &o, s);
  int n_bad = 0;
  for (int i = 0; i < n; i++) {
    outlier[i] = (uint8_t)orc_check_outlier(K4, pose7, Xw + 3 * i, uv + 2 * i, (double)inv_sigma2[i], 5.991, nullptr);
    n_bad += outlier[i];
  }
  double* q = pose7 + 3;
  double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; k++) q[k] /= nq;
  return n - n_bad;
}

// LocalBundleAdjustment's optimisation core (src/CeresOptimizer.cc:408-571) on flattened arrays:
// pass 1 = every observation with Huber, <=5 iterations; classify (chi2 > 5.991 or depth <= 0, only for
// observations whose camera is a LOCAL keyframe, :547-565); pass 2 = the SAME problem plus every
// not-erased observation again without loss (F6), <=10 iterations; classify again.
// cam_local[c] = 1 for local keyframes; cam_fixed[c] = 1 for fixed keyframes and KF id 0.
// obs_erase[i] (out) = final to_erase membership.  Returns 1 if aborted by *stop before a solve
// (nothing written back, :509-512), else 0.
int orc_local_ba(const double* K4, double* poses7, const uint8_t* cam_fixed, const uint8_t* cam_local, int ncam,
                 double* pts3, int npts, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv,
                 const float* obs_inv_sigma2, int nobs, const volatile uint8_t* stop, int duplicate_blocks,
                 uint8_t* obs_erase, orc_ba_summary* s1, orc_ba_summary* s2) {
  std::vector<double> P0(poses7, poses7 + 7 * (size_t)ncam), X0(pts3, pts3 + 3 * (size_t)npts);
  std::vector<double> w(nobs);
  for (int i = 0; i < nobs; i++) w[i] = (double)obs_inv_sigma2[i];
  std::vector<int32_t> oc(obs_cam, obs_cam + nobs), op(obs_pt, obs_pt + nobs);
  std::vector<double> uv(obs_uv, obs_uv + 2 * (size_t)nobs);
  std::vector<uint8_t> rob(nobs, 1);
  std::vector<uint8_t> erase(nobs, 0);
  auto classify = [&]() {
    for (int i = 0; i < nobs; i++) {
      erase[i] = 0;
      int c = obs_cam[i];
      if (!cam_local[c]) continue;
      double depth;
      int out = orc_check_outlier(K4 + 4 * c, P0.data() + 7 * c, X0.data() + 3 * obs_pt[i], obs_uv + 2 * i,
                                  (double)obs_inv_sigma2[i], 5.991, &depth);
      if (out || depth <= 0) erase[i] = 1;
    }
  };
  if (stop && *stop) return 1;
  orc_ba_opts o1{5, std::sqrt(5.991), 0, stop};
  orc_ba_solve(K4, P0.data(), cam_fixed, ncam, X0.data(), npts, oc.data(), op.data(), uv.data(), w.data(), rob.data(), nobs, &o1, s1);
  classify();
  // pass 2
  if (!duplicate_blocks) { oc.clear(); op.clear(); uv.clear(); w.clear(); rob.clear(); }
  for (int i = 0; i < nobs; i++) {
    if (erase[i]) continue;
    oc.push_back(obs_cam[i]); op.push_back(obs_pt[i]);
    uv.push_back(obs_uv[2 * i]); uv.push_back(obs_uv[2 * i + 1]);
    w.push_back((double)obs_inv_sigma2[i]); rob.push_back(0);
  }
  if (stop && *stop) return 1;
  orc_ba_opts o2{10, std::sqrt(5.991), 0, stop};
  orc_ba_solve(K4, P0.data(), cam_fixed, ncam, X0.data(), npts, oc.data(), op.data(), uv.data(), w.data(), rob.data(), (int)oc.size(), &o2, s2);
  classify();
  std::memcpy(obs_erase, erase.data(), nobs);
  // write back with the 7-vector codec's normalisation (src/MatEigenConverter.cc:74-85)
  for (int c = 0; c < ncam; c++) {
    double* q = P0.data() + 7 * c + 3;
    double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; k++) q[k] /= nq;
  }
  std::memcpy(poses7, P0.data(), sizeof(double) * 7 * ncam);
  std::memcpy(pts3, X0.data(), sizeof(double) * 3 * npts);
  return 0;
}

// residual / jacobian of one observation for finite-difference tests
void orc_ba_eval_obs(const double* K4, const double* pose7, const double* X, const double* uv, double w, int robust,
                     double huber_delta, double* r2, double* Jc12, double* Jp6, double* rho) {
  Problem P; int32_t z = 0; uint8_t rb = (uint8_t)robust;
  P.K4 = K4; P.obs_cam = &z; P.obs_pt = &z; P.obs_uv = uv; P.obs_w = &w; P.obs_robust = &rb; P.opt.huber_delta = huber_delta;
  *rho = eval_obs(P, 0, pose7, X, r2, Jc12, Jp6);
}
// MatEigenConverter::Matrix4dToMatrix_7_1 (src/MatEigenConverter.cc:66-75): Tcw_7_1 = [pose.block<3,1>(0,3),
// Eigen::Quaterniond(R).coeffs()], coeffs = [x,y,z,w].  Eigen's matrix -> quaternion assignment
// (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>): trace > 0 -> w from the trace; otherwise the
// largest diagonal element selects the component computed from a square root.  T row-major 4x4.
void orc_matrix4d_to_pose7(const double* T, double* out) {
  auto M = [&](int r, int c) { return T[4 * r + c]; };
  double q[4];
  double t = M(0, 0) + M(1, 1) + M(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M(2, 1) - M(1, 2)) * t;
    q[1] = (M(0, 2) - M(2, 0)) * t;
    q[2] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M(k, j) - M(j, k)) * t;
    q[j] = (M(j, i) + M(i, j)) * t;
    q[k] = (M(k, i) + M(i, k)) * t;
  }
  out[0] = M(0, 3); out[1] = M(1, 3); out[2] = M(2, 3);
  out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}
// MatEigenConverter::Matrix_7_1_ToMatrix4d (src/MatEigenConverter.cc:77-85): Quaterniond(w,x,y,z).normalized()
// .toRotationMatrix() into an identity 4x4, translation copied.
void orc_pose7_to_matrix4d(const double* p, double* T) {
  const double n = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
  const double q[4] = {p[3] / n, p[4] / n, p[5] / n, p[6] / n};
  double R[9];
  quat_to_R(q, R);
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = p[r]; }
  T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
}

// worker threads of the evaluation / Schur elimination (results do not depend on it); returns the previous value
int orc_set_ba_threads(int n) { int o = g_ba_threads; g_ba_threads = n < 1 ? 1 : n; return o; }
void orc_quat_plus(const double* q, const double* d, double* out) { quat_plus(q, d, out); }
void orc_quat_rotate(const double* q, const double* v, double* out) { quat_rotate(q, v, out); }

}  // extern "C"
