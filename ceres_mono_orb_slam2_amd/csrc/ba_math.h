// fp64 geometry shared by the BA kernels and their host wrappers: Eigen-convention quaternion
// rotation, the EigenQuaternionParameterization Plus, and the robustified reprojection residual
// with analytic Jacobians (reference include/CeresOptimizer.h:56-166).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

namespace orbhip {

#define BA_HD __host__ __device__ __forceinline__

// Eigen's q*v for q = [x,y,z,w]:  v + w*(2 qv x v) + qv x (2 qv x v)
BA_HD void quat_rotate(const double* q, const double* v, double* out) {
  double uvx = 2 * (q[1] * v[2] - q[2] * v[1]);
  double uvy = 2 * (q[2] * v[0] - q[0] * v[2]);
  double uvz = 2 * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
  out[1] = v[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
  out[2] = v[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

BA_HD void quat_to_R(const double* q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
         tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// q+ = dq (x) q with dq = [sin|d|/|d| d, cos|d|]  (d = half-angle vector)
BA_HD void quat_plus(const double* q, const double* d, double* out) {
  double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n > 0.0) {
    double s = sin(n) / n;
    double dx = s * d[0], dy = s * d[1], dz = s * d[2], dw = cos(n);
    out[3] = dw * q[3] - dx * q[0] - dy * q[1] - dz * q[2];
    out[0] = dw * q[0] + dx * q[3] + dy * q[2] - dz * q[1];
    out[1] = dw * q[1] - dx * q[2] + dy * q[3] + dz * q[0];
    out[2] = dw * q[2] + dx * q[1] - dy * q[0] + dz * q[3];
  } else {
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  }
}

// r = sqrt(rho') * w * (uv - pi(K (q X + t))); returns rho (cost = rho/2).
// Jc (2x6, [t | half-angle delta]) and Jp (2x3) may be NULL.  robust: 0 = no loss, 1 = Huber(delta), 2 = Huber block + its loss-free twin.
// RX_out (may be NULL): the rotated point R X the Jacobians are built from (Jc = Q [I | -2 [RX]x], Jp = Q R).
BA_HD double reproj_eval(const double* K4, const double* pose7, const double* X, double u_obs, double v_obs,
                         double w, int robust, double huber, double* r, double* Jc, double* Jp, double* RX_out = nullptr) {
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const double* t = pose7;
  const double* q = pose7 + 3;
  double RX[3];
  quat_rotate(q, X, RX);
  if (RX_out) { RX_out[0] = RX[0]; RX_out[1] = RX[1]; RX_out[2] = RX[2]; }
  const double p0 = RX[0] + t[0], p1 = RX[1] + t[1], p2 = RX[2] + t[2];
  const double u = (fx * p0 + cx * p2) / p2;
  const double v = (fy * p1 + cy * p2) / p2;
  double r0 = w * (u_obs - u), r1 = w * (v_obs - v);
  const double s = r0 * r0 + r1 * r1;
  double rho0 = s, rho1 = 1.0;
  if (robust) {
    const double b = huber * huber;
    if (s > b) {
      const double rr = sqrt(s);
      rho0 = 2 * huber * rr - b;
      rho1 = fmax(DBL_MIN, huber / rr);
    }
  }
  // robust == 2: the observation is present TWICE, once under the loss and once without (LocalBundleAdjustment's second
  // pass re-adds every kept block on the same problem).  Both blocks share the raw residual r0 and Jacobian J0
  // and differ only by the corrector scale (sqrt(rho') vs 1), so their joint contribution to J^T J, J^T r and the cost is
  // exactly that of ONE block scaled by sqrt(rho' + 1) with cost (rho + s) / 2: half the blocks, a quarter of the Schur pairs.
  if (robust == 2) { rho0 += s; rho1 += 1.0; }
  const double sq = sqrt(rho1);
  if (Jc || Jp) {
    const double iz = 1.0 / p2;
    const double J00 = fx * iz, J02 = -fx * p0 * iz * iz, J11 = fy * iz, J12 = -fy * p1 * iz * iz;
    const double ws = -w * sq;
    if (Jc) {
      // dr/dt = ws*Jpi ; dr/ddelta = -2 ws Jpi [RX]x
      Jc[0] = ws * J00; Jc[1] = 0.0 * ws; Jc[2] = ws * J02;
      Jc[6] = 0.0 * ws; Jc[7] = ws * J11; Jc[8] = ws * J12;
      // Jpi * [RX]x, [RX]x = [[0,-z,y],[z,0,-x],[-y,x,0]]
      const double a00 = J02 * (-RX[1]), a01 = J00 * (-RX[2]) + J02 * RX[0], a02 = J00 * RX[1];
      const double a10 = J11 * RX[2] + J12 * (-RX[1]), a11 = J12 * RX[0], a12 = J11 * (-RX[0]);
      Jc[3] = -2.0 * ws * a00; Jc[4] = -2.0 * ws * a01; Jc[5] = -2.0 * ws * a02;
      Jc[9] = -2.0 * ws * a10; Jc[10] = -2.0 * ws * a11; Jc[11] = -2.0 * ws * a12;
    }
    if (Jp) {
      double R[9];
      quat_to_R(q, R);
      for (int c = 0; c < 3; c++) {
        Jp[c] = ws * (J00 * R[c] + J02 * R[6 + c]);
        Jp[3 + c] = ws * (J11 * R[3 + c] + J12 * R[6 + c]);
      }
    }
  }
  r[0] = r0 * sq; r[1] = r1 * sq;
  return rho0;
}

// CheckOutlier (src/CeresOptimizer.cc:227-241)
BA_HD int check_outlier(const double* K4, const double* pose7, const double* X, double u_obs, double v_obs,
                        double inv_sigma2, double thres, double* depth) {
  double RX[3];
  quat_rotate(pose7 + 3, X, RX);
  const double p0 = RX[0] + pose7[0], p1 = RX[1] + pose7[1], p2 = RX[2] + pose7[2];
  const double u = (K4[0] * p0 + K4[2] * p2) / p2, v = (K4[1] * p1 + K4[3] * p2) / p2;
  const double eu = u_obs - u, ev = v_obs - v;
  if (depth) *depth = p2;
  return ((eu * eu + ev * ev) * inv_sigma2 > thres) ? 1 : 0;
}

}  // namespace orbhip
