// ============================================================================
// oracle/sim3_oracle.cpp -- CPU restatement of CeresOptimizer::OptimizeSim3.
//
// TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header for the rule).
//
// PARITY STATUS: "parity unpinned".  The reference (src/CeresOptimizer.cc:24-47,
// 601-735; include/CeresOptimizer.h:168-264) delegates to Ceres Solver and to
// Sophus::Sim3d (both un-vendored, unpinned, absent here).  This file restates
//   * Sophus' Sim(3): tangent order [upsilon(3), omega(3), sigma], storage
//     [qx,qy,qz,qw (|q|^2 = scale), tx,ty,tz] (= Sophus::Sim3d::data()), exp / log
//     with the closed-form W / W^-1 matrices, group product, inverse, action;
//   * Sim3ErrorTerm (residual = w * (pi(K * S(^-1) * P) - obs), the 2x7 "left
//     perturbation" Jacobian exactly as written in the header -- also for the
//     inverse term, where it is NOT the derivative; restated, not corrected);
//   * Sim3Parameterization::Plus (x (+) d = log(exp(x) * exp(d')), d'[6] = max(d[6], -20))
//     with the identity 7x7 ComputeJacobian;
//   * Ceres 1.14's trust-region LM (same controller as oracle/ba_oracle.cpp) on the single
//     7-parameter block, HuberLoss(sqrt(th2)) shared by all 2n residual blocks;
//   * the outlier count of :694-726, including Eigen's Quaterniond(Matrix3d) applied to the
//     SCALED matrix s*R and Eigen's unit-quaternion rotation formula applied to the result.
// It is pinned only by analytic checks in tests/test_oracle_sim3.py (scipy expm of the 4x4
// generator, group axioms, finite-difference Jacobian of the forward term, zero-noise recovery).
// ============================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct V3 { double x, y, z; };
struct M3 { double m[3][3]; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 mul(const M3& A, V3 v) {
  return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
          A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline M3 hat(V3 w) { return {{{0, -w.z, w.y}, {w.z, 0, -w.x}, {-w.y, w.x, 0}}}; }
inline M3 matmul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += A.m[i][k] * B.m[k][j];
    C.m[i][j] = s;
  }
  return C;
}
inline M3 lincomb(double a, const M3& A, double b, const M3& B, double c) {   // a*A + b*B + c*I
  M3 C;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C.m[i][j] = a * A.m[i][j] + b * B.m[i][j] + (i == j ? c : 0.0);
  return C;
}

constexpr double kEps = 1e-10;   // Sophus::Constants<double>::epsilon()

struct Quat { double x, y, z, w; };                // Eigen coefficient order
struct Sim3 { Quat q; V3 t; };                     // |q|^2 = scale

inline Quat qmul(Quat a, Quat b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline double qn2(Quat q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }

// RxSO3 action on a point: scale*p + w*(2 v x p) + v x (2 v x p)   (= |q|^2 R p)
inline V3 rxso3_act(Quat q, V3 p) {
  V3 v{q.x, q.y, q.z};
  V3 c = cross(v, p);
  c = c + c;
  return qn2(q) * p + (q.w * c + cross(v, c));
}
inline V3 sim3_act(const Sim3& S, V3 p) { return rxso3_act(S.q, p) + S.t; }
inline Sim3 sim3_mul(const Sim3& A, const Sim3& B) { return {qmul(A.q, B.q), A.t + rxso3_act(A.q, B.t)}; }
inline Sim3 sim3_inv(const Sim3& S) {
  double n2 = qn2(S.q);
  Quat qi{-S.q.x / n2, -S.q.y / n2, -S.q.z / n2, S.q.w / n2};
  V3 ti = rxso3_act(qi, S.t);
  return {qi, {-ti.x, -ti.y, -ti.z}};
}

// W(omega, sigma) = int_0^1 exp(t (sigma I + [omega]x)) dt  in closed form (Sophus details::calcW)
M3 calc_W(V3 omega, double theta, double sigma) {
  const M3 Om = hat(omega), Om2 = matmul(Om, Om);
  const double scale = std::exp(sigma);
  double A, B, C;
  if (std::fabs(sigma) < kEps) {
    C = 1.0;
    if (std::fabs(theta) < kEps) { A = 0.5; B = 1.0 / 6.0; }
    else { double t2 = theta * theta; A = (1.0 - std::cos(theta)) / t2; B = (theta - std::sin(theta)) / (t2 * theta); }
  } else {
    C = (scale - 1.0) / sigma;
    if (std::fabs(theta) < kEps) {
      double s2 = sigma * sigma;
      A = ((sigma - 1.0) * scale + 1.0) / s2;
      B = (scale * 0.5 * s2 + scale - 1.0 - sigma * scale) / (s2 * sigma);
    } else {
      double t2 = theta * theta, a = scale * std::sin(theta), b = scale * std::cos(theta), c = t2 + sigma * sigma;
      A = (a * sigma + (1.0 - b) * theta) / (theta * c);
      B = (C - ((b - 1.0) * sigma + a * theta) / c) * 1.0 / t2;
    }
  }
  return lincomb(A, Om, B, Om2, C);
}

M3 calc_Winv(V3 omega, double theta, double sigma, double scale) {
  const M3 Om = hat(omega), Om2 = matmul(Om, Om);
  const double scale_sq = scale * scale, t2 = theta * theta, st = std::sin(theta), ct = std::cos(theta);
  double a, b, c;
  if (std::fabs(sigma * sigma) < kEps) {
    c = 1.0 - 0.5 * sigma;
    a = -0.5;
    if (std::fabs(t2) < kEps) b = 1.0 / 12.0;
    else b = (theta * st + 2.0 * ct - 2.0) / (2.0 * t2 * (ct - 1.0));
  } else {
    const double scale_cu = scale_sq * scale;
    c = sigma / (scale - 1.0);
    if (std::fabs(t2) < kEps) {
      a = (-sigma * scale + scale - 1.0) / ((scale - 1.0) * (scale - 1.0));
      b = (scale_sq * sigma - 2.0 * scale_sq + scale * sigma + 2.0 * scale) / (2.0 * scale_cu - 6.0 * scale_sq + 6.0 * scale - 2.0);
    } else {
      const double ss = scale * st, sc = scale * ct;
      a = (theta * sc - theta - sigma * ss) / (theta * (scale_sq - 2.0 * sc + 1.0));
      b = -scale * (theta * ss - theta * st + sigma * sc - scale * sigma + sigma * ct - sigma) /
          (t2 * (scale_cu - 2.0 * scale * sc - scale_sq + 2.0 * sc + scale - 1.0));
    }
  }
  return lincomb(a, Om, b, Om2, c);
}

Sim3 sim3_exp(const double* a) {
  V3 ups{a[0], a[1], a[2]}, om{a[3], a[4], a[5]};
  const double sigma = a[6];
  const double t2 = om.x * om.x + om.y * om.y + om.z * om.z;
  double theta, fi, fr;
  if (t2 < kEps * kEps) {
    theta = 0.0;
    double t4 = t2 * t2;
    fi = 0.5 - t2 / 48.0 + t4 / 3840.0;
    fr = 1.0 - t2 / 8.0 + t4 / 384.0;
  } else {
    theta = std::sqrt(t2);
    fi = std::sin(0.5 * theta) / theta;
    fr = std::cos(0.5 * theta);
  }
  const double rs = std::sqrt(std::exp(sigma));
  Sim3 S;
  S.q = {rs * fi * om.x, rs * fi * om.y, rs * fi * om.z, rs * fr};
  S.t = mul(calc_W(om, theta, sigma), ups);
  return S;
}

void sim3_log(const Sim3& S, double* out) {
  const double scale = qn2(S.q), sigma = std::log(scale);
  const double inv = 1.0 / std::sqrt(scale);
  const Quat u{S.q.x * inv, S.q.y * inv, S.q.z * inv, S.q.w * inv};
  const double n2 = u.x * u.x + u.y * u.y + u.z * u.z, w = u.w;
  double f, theta;
  if (n2 < kEps * kEps) {
    f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w);
    theta = 2.0 * n2 / w;
  } else {
    const double n = std::sqrt(n2);
    const double at = (w < 0.0) ? std::atan2(-n, -w) : std::atan2(n, w);
    f = 2.0 * at / n;
    theta = f * n;
  }
  V3 om{f * u.x, f * u.y, f * u.z};
  V3 ups = mul(calc_Winv(om, theta, sigma, scale), S.t);
  out[0] = ups.x; out[1] = ups.y; out[2] = ups.z; out[3] = om.x; out[4] = om.y; out[5] = om.z; out[6] = sigma;
}

// Sim3Parameterization::Plus (src/CeresOptimizer.cc:24-41)
void sim3_plus(const double* x, const double* d, double* out) {
  double dd[7];
  for (int i = 0; i < 7; i++) dd[i] = d[i];
  dd[6] = std::max(dd[6], -20.0);
  sim3_log(sim3_mul(sim3_exp(x), sim3_exp(dd)), out);
}

struct Term { double P[3], u, v, w; int inverse; int cam; };   // cam 0 -> K1, 1 -> K2

// Sim3ErrorTerm::Evaluate (include/CeresOptimizer.h:178-236) + Huber corrector.  Returns rho.
double eval_term(const Term& T, const double* K4, const Sim3& S, const Sim3& Sinv, double huber, double* r, double* J) {
  V3 p = sim3_act(T.inverse ? Sinv : S, {T.P[0], T.P[1], T.P[2]});
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const double pr0 = fx * p.x + cx * p.z, pr1 = fy * p.y + cy * p.z, pr2 = p.z;
  double r0 = T.w * (pr0 / pr2 - T.u), r1 = T.w * (pr1 / pr2 - T.v);
  const double s = r0 * r0 + r1 * r1;
  double rho0 = s, rho1 = 1.0;
  const double b = huber * huber;
  if (s > b) { const double rr = std::sqrt(s); rho0 = 2 * huber * rr - b; rho1 = std::max(DBL_MIN, huber / rr); }
  const double sq = std::sqrt(rho1);
  if (J) {
    const double Z2 = p.z * p.z;
    const double c00 = fx / p.z, c02 = -p.x * fx / Z2, c11 = fy / p.z, c12 = -fy * p.y / Z2;
    // left = [I | -hat(p) | p]
    const double L[3][7] = {{1, 0, 0, 0, p.z, -p.y, p.x}, {0, 1, 0, -p.z, 0, p.x, p.y}, {0, 0, 1, p.y, -p.x, 0, p.z}};
    for (int j = 0; j < 7; j++) {
      J[j] = sq * T.w * (c00 * L[0][j] + c02 * L[2][j]);
      J[7 + j] = sq * T.w * (c11 * L[1][j] + c12 * L[2][j]);
    }
  }
  r[0] = sq * r0; r[1] = sq * r1;
  return rho0;
}

bool chol7_solve(double* A, double* b) {
  const int n = 7;
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
  return true;
}

// Eigen::Quaterniond(Matrix3d) (QuaternionBase::operator=(MatrixBase), Shepperd-style branch on the trace)
Quat quat_from_matrix(const M3& M) {
  double q[4];   // x y z w
  double t = M.m[0][0] + M.m[1][1] + M.m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M.m[2][1] - M.m[1][2]) * t;
    q[1] = (M.m[0][2] - M.m[2][0]) * t;
    q[2] = (M.m[1][0] - M.m[0][1]) * t;
  } else {
    int i = 0;
    if (M.m[1][1] > M.m[0][0]) i = 1;
    if (M.m[2][2] > M.m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M.m[i][i] - M.m[j][j] - M.m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M.m[k][j] - M.m[j][k]) * t;
    q[j] = (M.m[j][i] + M.m[i][j]) * t;
    q[k] = (M.m[k][i] + M.m[i][k]) * t;
  }
  return {q[0], q[1], q[2], q[3]};
}

// Sophus rotationMatrix(): normalise, then Eigen toRotationMatrix
M3 rot_matrix(Quat q) {
  const double inv = 1.0 / std::sqrt(qn2(q));
  const double x = q.x * inv, y = q.y * inv, z = q.z * inv, w = q.w * inv;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x,
               txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  return {{{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}}};
}

// CheckOutlier (src/CeresOptimizer.cc:227-241) fed with Quaterniond(s*R) as :695-709 does:
// Eigen's q*v = v + w*(2 qv x v) + qv x (2 qv x v), which assumes |q| = 1 -- restated as is.
int check_outlier_sim3(const double* K4, const Sim3& S, const double* P, double u, double v, float inv_sigma, double thres) {
  const double scale = qn2(S.q);
  M3 R = rot_matrix(S.q);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = scale * R.m[i][j];
  const Quat q = quat_from_matrix(R);
  V3 qv{q.x, q.y, q.z}, p{P[0], P[1], P[2]};
  V3 uvv = cross(qv, p);
  uvv = uvv + uvv;
  V3 c = p + (q.w * uvv + cross(qv, uvv)) + S.t;
  const double px = K4[0] * c.x + K4[2] * c.z, py = K4[1] * c.y + K4[3] * c.z, pz = c.z;
  const double eu = u - px / pz, ev = v - py / pz;
  return ((eu * eu + ev * ev) * (double)inv_sigma > thres) ? 1 : 0;
}

}  // namespace

extern "C" {

struct orc_sim3_summary { double initial_cost, final_cost; int iterations, successful_steps, termination; double final_radius; };

void orc_sim3_exp(const double* lie7, double* qt7) {
  Sim3 S = sim3_exp(lie7);
  qt7[0] = S.q.x; qt7[1] = S.q.y; qt7[2] = S.q.z; qt7[3] = S.q.w; qt7[4] = S.t.x; qt7[5] = S.t.y; qt7[6] = S.t.z;
}
void orc_sim3_log(const double* qt7, double* lie7) {
  Sim3 S{{qt7[0], qt7[1], qt7[2], qt7[3]}, {qt7[4], qt7[5], qt7[6]}};
  sim3_log(S, lie7);
}
void orc_sim3_plus(const double* x, const double* d, double* out) { sim3_plus(x, d, out); }
void orc_sim3_inverse(const double* qt7, double* out) {
  Sim3 S{{qt7[0], qt7[1], qt7[2], qt7[3]}, {qt7[4], qt7[5], qt7[6]}};
  Sim3 I = sim3_inv(S);
  out[0] = I.q.x; out[1] = I.q.y; out[2] = I.q.z; out[3] = I.q.w; out[4] = I.t.x; out[5] = I.t.y; out[6] = I.t.z;
}
void orc_sim3_act(const double* qt7, const double* p, double* out) {
  Sim3 S{{qt7[0], qt7[1], qt7[2], qt7[3]}, {qt7[4], qt7[5], qt7[6]}};
  V3 r = sim3_act(S, {p[0], p[1], p[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// residual (2) and Jacobian (2x7 row-major) of one Sim3ErrorTerm, no loss
void orc_sim3_eval_term(const double* K4, const double* lie7, const double* P, const double* uv, double w, int inverse, double* r, double* J) {
  Sim3 S = sim3_exp(lie7), Si = sim3_inv(S);
  Term T{{P[0], P[1], P[2]}, uv[0], uv[1], w, inverse, 0};
  eval_term(T, K4, S, Si, 1e300, r, J);
}

// OptimizeSim3 (src/CeresOptimizer.cc:601-735) on flattened arrays.  Correspondence i contributes
// term A: K1, obs1[i], P3D2c[i] (kf2's point in kf2's camera frame), forward S12;  term B: K2, obs2[i],
// P3D1c[i], inverse S12 -- in this order (:660-683).  s12 = Sophus::Sim3d::data() layout, updated in place with
// exp(optimised log) (:691).  fix_scale is accepted and ignored exactly as the reference ignores bFixScale.
// outlier (nullable) [n] = is_outlier_12 || is_outlier_21.  Returns n - n_bad, or 0 when that is < 10 (:731).
int orc_optimize_sim3(const double* K1, const double* K2, double* s12, const double* P3D2c, const double* obs1,
                      const float* inv_sigma2_1, const double* P3D1c, const double* obs2, const float* inv_sigma2_2, int n,
                      double th2, int fix_scale, uint8_t* outlier, orc_sim3_summary* sum) {
  (void)fix_scale;
  const double huber = std::sqrt(th2);
  double x[7];
  {
    Sim3 S0{{s12[0], s12[1], s12[2], s12[3]}, {s12[4], s12[5], s12[6]}};
    sim3_log(S0, x);
  }
  std::vector<Term> terms;
  terms.reserve(2 * (size_t)n);
  for (int i = 0; i < n; i++) {
    Term A{{P3D2c[3 * i], P3D2c[3 * i + 1], P3D2c[3 * i + 2]}, obs1[2 * i], obs1[2 * i + 1], (double)inv_sigma2_1[i], 0, 0};
    Term B{{P3D1c[3 * i], P3D1c[3 * i + 1], P3D1c[3 * i + 2]}, obs2[2 * i], obs2[2 * i + 1], (double)inv_sigma2_2[i], 1, 1};
    terms.push_back(A); terms.push_back(B);
  }
  const int nt = (int)terms.size();
  orc_sim3_summary S{};
  double radius = 1e4, dec = 2.0, x_cost = 0, x_norm = 0;
  double g[7], H[49], scale[7];
  int iteration = 0, invalid = 0;
  const int max_iters = 100;

  auto cost_at = [&](const double* xx) {
    Sim3 Sx = sim3_exp(xx), Si = sim3_inv(Sx);
    double c = 0, r[2];
    for (int i = 0; i < nt; i++) c += 0.5 * eval_term(terms[i], terms[i].cam ? K2 : K1, Sx, Si, huber, r, nullptr);
    return c;
  };
  auto evaluate = [&](bool first) -> double {
    Sim3 Sx = sim3_exp(x), Si = sim3_inv(Sx);
    x_cost = 0;
    std::fill(g, g + 7, 0.0); std::fill(H, H + 49, 0.0);
    for (int i = 0; i < nt; i++) {
      double r[2], J[14];
      x_cost += 0.5 * eval_term(terms[i], terms[i].cam ? K2 : K1, Sx, Si, huber, r, J);
      for (int a = 0; a < 7; a++) {
        g[a] += J[a] * r[0] + J[7 + a] * r[1];
        for (int b2 = 0; b2 < 7; b2++) H[a * 7 + b2] += J[a] * J[b2] + J[7 + a] * J[7 + b2];
      }
    }
    if (first) for (int a = 0; a < 7; a++) scale[a] = 1.0 / (1.0 + std::sqrt(H[a * 8]));
    double xn = 0;
    for (int k = 0; k < 7; k++) xn += x[k] * x[k];
    x_norm = std::sqrt(xn);
    double mg[7], xp[7], gmax = 0;
    for (int k = 0; k < 7; k++) mg[k] = -g[k];
    sim3_plus(x, mg, xp);
    for (int k = 0; k < 7; k++) gmax = std::max(gmax, std::fabs(x[k] - xp[k]));
    return gmax;
  };

  if (nt > 0) {
    double gmax = evaluate(true);
    S.initial_cost = x_cost;
    bool done = gmax <= 1e-10;
    if (done) S.termination = 1;
    while (!done) {
      if (iteration >= max_iters) { S.termination = 0; break; }
      if (radius <= 1e-32) { S.termination = 6; break; }
      iteration++;
      double A[49], Hs[49], gs[7], y[7];
      for (int a = 0; a < 7; a++) {
        gs[a] = g[a] * scale[a];
        for (int b2 = 0; b2 < 7; b2++) Hs[a * 7 + b2] = H[a * 7 + b2] * scale[a] * scale[b2];
      }
      std::memcpy(A, Hs, sizeof(A));
      for (int a = 0; a < 7; a++) A[a * 8] += std::min(std::max(Hs[a * 8], 1e-6), 1e32) / radius;
      std::memcpy(y, gs, sizeof(y));
      bool ok = chol7_solve(A, y);
      double mcc = 0;
      if (ok) for (int a = 0; a < 7; a++) {
        double hs = 0;
        for (int b2 = 0; b2 < 7; b2++) hs += Hs[a * 7 + b2] * (-y[b2]);
        mcc -= (-y[a]) * (gs[a] + 0.5 * hs);
      }
      if (!ok || !(mcc > 0.0)) {
        if (++invalid >= 5) { S.termination = 5; break; }
        radius /= dec; dec *= 2;
        continue;
      }
      invalid = 0;
      double d[7], cand[7];
      for (int k = 0; k < 7; k++) d[k] = (-y[k]) * scale[k];
      sim3_plus(x, d, cand);
      double sn = 0;
      for (int k = 0; k < 7; k++) { double e = x[k] - cand[k]; sn += e * e; }
      double cand_cost = cost_at(cand);
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      if (std::sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) { S.termination = 2; break; }
      const double cost_change = x_cost - cand_cost;
      if (std::fabs(cost_change) <= 1e-6 * x_cost) { S.termination = 3; break; }
      const double rel = cost_change / mcc;
      if (rel > 1e-3) {
        std::memcpy(x, cand, sizeof(x));
        gmax = evaluate(false);
        S.successful_steps++;
        radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
        dec = 2.0;
        if (gmax <= 1e-10) { S.termination = 1; break; }
      } else {
        radius /= dec; dec *= 2.0;
      }
    }
  }
  S.final_cost = x_cost; S.iterations = iteration; S.final_radius = radius;
  if (sum) *sum = S;

  const Sim3 S12 = sim3_exp(x), S21 = sim3_inv(S12);
  s12[0] = S12.q.x; s12[1] = S12.q.y; s12[2] = S12.q.z; s12[3] = S12.q.w; s12[4] = S12.t.x; s12[5] = S12.t.y; s12[6] = S12.t.z;
  const double thres = huber * huber;
  int n_bad = 0;
  for (int i = 0; i < n; i++) {
    int o12 = check_outlier_sim3(K1, S12, P3D2c + 3 * i, obs1[2 * i], obs1[2 * i + 1], inv_sigma2_1[i], thres);
    int o21 = check_outlier_sim3(K2, S21, P3D1c + 3 * i, obs2[2 * i], obs2[2 * i + 1], inv_sigma2_2[i], thres);
    if (outlier) outlier[i] = (uint8_t)(o12 | o21);
    if (o12 | o21) n_bad++;
  }
  if (n - n_bad < 10) return 0;
  return n - n_bad;
}

}  // extern "C"

// ============================================================================ OptimizeEssentialGraph (second half)
// CeresOptimizer::OptimizeEssentialGraph, the solve and the write-back arithmetic (src/CeresOptimizer.cc:737-957) with
// EssentialGraphErrorTerm (include/CeresOptimizer.h:266-330): vertices = Sim(3) tangent 7-vectors (Scw.log()) under
// Sim3Parameterization::Plus, the loop keyframe constant; every edge (j, i, Sji) contributes the 7-vector residual
// log(Sji * Si * Sj^-1) (identity information, no loss) with Jacobians J_i = Jr * Adj(Sj), J_j = -J_i where
// Jr = I + ad/2 + ad^2/12 of the residual.  Ceres 1.14 LM with default options; the normal equations are solved densely
// (SPARSE_NORMAL_CHOLESKY is the same solve up to rounding).  Sophus::Sim3d::Adj() restated: [[sR, [t]x R, -t], [0, R, 0],
// [0, 0, 1]] - pinned against expm in tests/test_oracle_essential_graph.py.
#include <vector>
namespace {

void sim3_adj(const Sim3& S, double A[49]) {
  const double scale = qn2(S.q);
  const M3 R = rot_matrix(S.q);
  const M3 T = hat(S.t);
  const M3 TR = matmul(T, R);
  for (int k = 0; k < 49; k++) A[k] = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { A[i * 7 + j] = scale * R.m[i][j]; A[i * 7 + 3 + j] = TR.m[i][j]; A[(3 + i) * 7 + 3 + j] = R.m[i][j]; }
  A[0 * 7 + 6] = -S.t.x; A[1 * 7 + 6] = -S.t.y; A[2 * 7 + 6] = -S.t.z;
  A[48] = 1.0;
}

// residual (7) and J_i (7x7 row-major); J_j = -J_i
void eg_eval_edge(const double* lie_j, const double* lie_i, const Sim3& Sji, double* r, double* Ji) {
  const Sim3 Si = sim3_exp(lie_i), Sj = sim3_exp(lie_j);
  const Sim3 E = sim3_mul(sim3_mul(Sji, Si), sim3_inv(Sj));
  sim3_log(E, r);
  if (!Ji) return;
  double ad[49];
  for (int k = 0; k < 49; k++) ad[k] = 0.0;
  const double ux = r[0], uy = r[1], uz = r[2], wx = r[3], wy = r[4], wz = r[5], sg = r[6];
  // block(0,0) = hat(omega) + sigma I ; block(0,3) = hat(upsilon) ; block(0,6) = -upsilon ; block(3,3) = hat(omega)
  const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}}, U[3][3] = {{0, -uz, uy}, {uz, 0, -ux}, {-uy, ux, 0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { ad[i * 7 + j] = W[i][j] + (i == j ? sg : 0.0); ad[i * 7 + 3 + j] = U[i][j]; ad[(3 + i) * 7 + 3 + j] = W[i][j]; }
  ad[0 * 7 + 6] = -ux; ad[1 * 7 + 6] = -uy; ad[2 * 7 + 6] = -uz;
  double ad2[49], Jr[49], Adj[49];
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) { double s = 0; for (int k = 0; k < 7; k++) s += ad[i * 7 + k] * ad[k * 7 + j]; ad2[i * 7 + j] = s; }
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) Jr[i * 7 + j] = (i == j ? 1.0 : 0.0) + 0.5 * ad[i * 7 + j] + 1.0 / 12. * ad2[i * 7 + j];
  sim3_adj(Sj, Adj);
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) { double s = 0; for (int k = 0; k < 7; k++) s += Jr[i * 7 + k] * Adj[k * 7 + j]; Ji[i * 7 + j] = s; }
}

bool dense_chol(std::vector<double>& A, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}
void dense_chol_solve(const std::vector<double>& L, int n, std::vector<double>& b) {
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * n + k] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k]; b[i] = s / L[(size_t)i * n + i]; }
}

}  // namespace

extern "C" {

void orc_sim3_adj(const double* qt7, double* A49) {
  Sim3 S{{qt7[0], qt7[1], qt7[2], qt7[3]}, {qt7[4], qt7[5], qt7[6]}};
  sim3_adj(S, A49);
}
void orc_sim3_mul(const double* a, const double* b, double* out) {
  Sim3 A{{a[0], a[1], a[2], a[3]}, {a[4], a[5], a[6]}}, B{{b[0], b[1], b[2], b[3]}, {b[4], b[5], b[6]}};
  Sim3 C = sim3_mul(A, B);
  out[0] = C.q.x; out[1] = C.q.y; out[2] = C.q.z; out[3] = C.q.w; out[4] = C.t.x; out[5] = C.t.y; out[6] = C.t.z;
}
void orc_eg_eval_edge(const double* lie_j, const double* lie_i, const double* Sji_qt7, double* r, double* Ji) {
  Sim3 S{{Sji_qt7[0], Sji_qt7[1], Sji_qt7[2], Sji_qt7[3]}, {Sji_qt7[4], Sji_qt7[5], Sji_qt7[6]}};
  eg_eval_edge(lie_j, lie_i, S, r, Ji);
}

// the solve: lie7 [n*7] in/out; kf_fixed[n]; edges (edge_j[e], edge_i[e], Sji qt7) in insertion order
int orc_optimize_essential_graph(double* lie7, const uint8_t* kf_fixed, int n, const int32_t* edge_j, const int32_t* edge_i,
                                 const double* edge_Sji, int ne, int max_iters, orc_sim3_summary* sum) {
  std::vector<int> col(n, -1);
  int nf = 0;
  for (int v = 0; v < n; v++) if (!kf_fixed[v]) col[v] = nf++;
  const int nc = 7 * nf;
  std::vector<Sim3> Sji(ne);
  for (int e = 0; e < ne; e++) Sji[e] = Sim3{{edge_Sji[7 * e], edge_Sji[7 * e + 1], edge_Sji[7 * e + 2], edge_Sji[7 * e + 3]}, {edge_Sji[7 * e + 4], edge_Sji[7 * e + 5], edge_Sji[7 * e + 6]}};
  std::vector<double> r(7 * (size_t)ne), J(49 * (size_t)ne), g(nc), scale(nc, 1.0), H, step(nc), cand(lie7, lie7 + 7 * (size_t)n);
  orc_sim3_summary S{};
  double radius = 1e4, dec = 2.0, x_cost = 0, x_norm = 0;
  int iteration = 0, invalid = 0;
  auto cost_at = [&](const double* x) {
    double c = 0, rr[7];
    for (int e = 0; e < ne; e++) { eg_eval_edge(x + 7 * edge_j[e], x + 7 * edge_i[e], Sji[e], rr, nullptr); for (int k = 0; k < 7; k++) c += 0.5 * rr[k] * rr[k]; }
    return c;
  };
  auto evaluate = [&](bool first) -> double {
    x_cost = 0;
    std::fill(g.begin(), g.end(), 0.0);
    for (int e = 0; e < ne; e++) {
      double* re = &r[7 * (size_t)e]; double* Je = &J[49 * (size_t)e];
      eg_eval_edge(lie7 + 7 * edge_j[e], lie7 + 7 * edge_i[e], Sji[e], re, Je);
      for (int k = 0; k < 7; k++) x_cost += 0.5 * re[k] * re[k];
      const int ci = col[edge_i[e]], cj = col[edge_j[e]];
      for (int a = 0; a < 7; a++) {
        double s = 0;
        for (int k = 0; k < 7; k++) s += Je[k * 7 + a] * re[k];
        if (ci >= 0) g[7 * ci + a] += s;
        if (cj >= 0) g[7 * cj + a] -= s;
      }
    }
    if (first) {
      std::vector<double> n2(nc, 0.0);
      for (int e = 0; e < ne; e++) {
        const double* Je = &J[49 * (size_t)e];
        const int ci = col[edge_i[e]], cj = col[edge_j[e]];
        for (int a = 0; a < 7; a++) {
          double s = 0;
          for (int k = 0; k < 7; k++) s += Je[k * 7 + a] * Je[k * 7 + a];
          if (ci >= 0) n2[7 * ci + a] += s;
          if (cj >= 0) n2[7 * cj + a] += s;
        }
      }
      for (int c = 0; c < nc; c++) scale[c] = 1.0 / (1.0 + std::sqrt(n2[c]));
    }
    double xn = 0, gmax = 0;
    for (int v = 0; v < n; v++) {
      if (col[v] < 0) continue;
      double mg[7], xp[7];
      for (int k = 0; k < 7; k++) { xn += lie7[7 * v + k] * lie7[7 * v + k]; mg[k] = -g[7 * col[v] + k]; }
      sim3_plus(lie7 + 7 * v, mg, xp);
      for (int k = 0; k < 7; k++) gmax = std::max(gmax, std::fabs(lie7[7 * v + k] - xp[k]));
    }
    x_norm = std::sqrt(xn);
    return gmax;
  };
  bool done = false;
  double gmax = evaluate(true);
  S.initial_cost = x_cost;
  if (gmax <= 1e-10) { S.termination = 1; done = true; }
  while (!done && nc > 0) {
    if (iteration >= max_iters) { S.termination = 0; break; }
    if (radius <= 1e-32) { S.termination = 6; break; }
    iteration++;
    // scaled normal equations
    H.assign((size_t)nc * nc, 0.0);
    std::vector<double> gs(nc);
    for (int c = 0; c < nc; c++) gs[c] = g[c] * scale[c];
    for (int e = 0; e < ne; e++) {
      const double* Je = &J[49 * (size_t)e];
      const int ci = col[edge_i[e]], cj = col[edge_j[e]];
      for (int a = 0; a < 7; a++)
        for (int b = 0; b < 7; b++) {
          double s = 0;
          for (int k = 0; k < 7; k++) s += Je[k * 7 + a] * Je[k * 7 + b];
          if (ci >= 0) H[(size_t)(7 * ci + a) * nc + 7 * ci + b] += s * scale[7 * ci + a] * scale[7 * ci + b];
          if (cj >= 0) H[(size_t)(7 * cj + a) * nc + 7 * cj + b] += s * scale[7 * cj + a] * scale[7 * cj + b];
          if (ci >= 0 && cj >= 0) {
            H[(size_t)(7 * ci + a) * nc + 7 * cj + b] -= s * scale[7 * ci + a] * scale[7 * cj + b];
            H[(size_t)(7 * cj + a) * nc + 7 * ci + b] -= s * scale[7 * cj + a] * scale[7 * ci + b];
          }
        }
    }
    std::vector<double> Hs(H);
    for (int c = 0; c < nc; c++) H[(size_t)c * nc + c] += std::min(std::max(Hs[(size_t)c * nc + c], 1e-6), 1e32) / radius;
    std::vector<double> y(gs);
    bool ok = dense_chol(H, nc);
    double mcc = 0;
    if (ok) {
      dense_chol_solve(H, nc, y);
      for (int c = 0; c < nc; c++) step[c] = -y[c];
      // model cost change = -(J s).(r + J s / 2) over the residual blocks (scaled J, scaled step)
      for (int e = 0; e < ne; e++) {
        const double* Je = &J[49 * (size_t)e]; const double* re = &r[7 * (size_t)e];
        const int ci = col[edge_i[e]], cj = col[edge_j[e]];
        for (int k = 0; k < 7; k++) {
          double m = 0;
          for (int a = 0; a < 7; a++) {
            if (ci >= 0) m += Je[k * 7 + a] * scale[7 * ci + a] * step[7 * ci + a];
            if (cj >= 0) m -= Je[k * 7 + a] * scale[7 * cj + a] * step[7 * cj + a];
          }
          mcc -= m * (re[k] + m / 2);
        }
      }
    }
    if (!ok || !(mcc > 0.0)) {
      if (++invalid >= 5) { S.termination = 5; break; }
      radius /= dec; dec *= 2;
      continue;
    }
    invalid = 0;
    double sn = 0;
    for (int v = 0; v < n; v++) {
      for (int k = 0; k < 7; k++) cand[7 * v + k] = lie7[7 * v + k];
      if (col[v] < 0) continue;
      double d[7];
      for (int k = 0; k < 7; k++) d[k] = step[7 * col[v] + k] * scale[7 * col[v] + k];
      sim3_plus(lie7 + 7 * v, d, &cand[7 * v]);
      for (int k = 0; k < 7; k++) { const double e2 = lie7[7 * v + k] - cand[7 * v + k]; sn += e2 * e2; }
    }
    double cand_cost = cost_at(cand.data());
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    if (std::sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) { S.termination = 2; break; }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * x_cost) { S.termination = 3; break; }
    const double rel = cost_change / mcc;
    if (rel > 1e-3) {
      std::memcpy(lie7, cand.data(), sizeof(double) * 7 * n);
      gmax = evaluate(false);
      S.successful_steps++;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
      dec = 2.0;
      if (gmax <= 1e-10) { S.termination = 1; break; }
    } else {
      radius /= dec; dec *= 2.0;
    }
  }
  S.final_cost = x_cost; S.iterations = iteration; S.final_radius = radius;
  if (sum) *sum = S;
  return 0;
}

// the write-back arithmetic (src/CeresOptimizer.cc:916-956): Tiw = [R | t / s] per keyframe (row-major 3x4); every map point
// P' = corrected_Swr * (Srw * P) with r = pt_ref[p] (reference keyframe index), Srw from the ORIGINAL tangents
void orc_essential_graph_correct(const double* lie_orig, const double* lie_opt, int n, double* Tiw, const int32_t* pt_ref, double* pts, int npts) {
  std::vector<Sim3> Swc(n), Scw0(n);
  for (int v = 0; v < n; v++) {
    const Sim3 S = sim3_exp(lie_opt + 7 * v);
    Swc[v] = sim3_inv(S); Scw0[v] = sim3_exp(lie_orig + 7 * v);
    const M3 R = rot_matrix(S.q);
    const double s = qn2(S.q);
    const double inv_s = 1. / s;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Tiw[12 * v + 4 * i + j] = R.m[i][j];
    Tiw[12 * v + 3] = inv_s * S.t.x; Tiw[12 * v + 7] = inv_s * S.t.y; Tiw[12 * v + 11] = inv_s * S.t.z;
  }
  for (int p = 0; p < npts; p++) {
    const int rk = pt_ref[p];
    const V3 Pc = sim3_act(Scw0[rk], {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]});
    const V3 Pw = sim3_act(Swc[rk], Pc);
    pts[3 * p] = Pw.x; pts[3 * p + 1] = Pw.y; pts[3 * p + 2] = Pw.z;
  }
}

}  // extern "C"
