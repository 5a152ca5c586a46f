// fp64 Sim(3) arithmetic for OptimizeSim3 (reference src/CeresOptimizer.cc:24-47, 601-735;
// include/CeresOptimizer.h:168-264).  The reference leans on Sophus::Sim3d, which is absent from the
// tree; its conventions are kept: tangent = [upsilon(3), omega(3), sigma], storage qt7 =
// [qx,qy,qz,qw, tx,ty,tz] with |q|^2 = scale (Sophus::Sim3d::data()), x (+) d = log(exp(x) exp(d)).
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

namespace orbhip {

#define S3_HD __host__ __device__ __forceinline__
#define S3_EPS 1e-10

// out = a*Om + b*Om^2 + c*I applied to v, Om = [w]x :  a (w x v) + b (w x (w x v)) + c v
S3_HD void s3_poly_apply(const double* w, double a, double b, double c, const double* v, double* out) {
  const double c1x = w[1] * v[2] - w[2] * v[1], c1y = w[2] * v[0] - w[0] * v[2], c1z = w[0] * v[1] - w[1] * v[0];
  const double c2x = w[1] * c1z - w[2] * c1y, c2y = w[2] * c1x - w[0] * c1z, c2z = w[0] * c1y - w[1] * c1x;
  out[0] = a * c1x + b * c2x + c * v[0];
  out[1] = a * c1y + b * c2y + c * v[1];
  out[2] = a * c1z + b * c2z + c * v[2];
}

// q (x) p for the scaled quaternion: |q|^2 p + w (2 v x p) + v x (2 v x p)  = s R p
S3_HD void s3_rot_scale(const double* q, const double* p, double* out) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double cx = 2 * (q[1] * p[2] - q[2] * p[1]), cy = 2 * (q[2] * p[0] - q[0] * p[2]), cz = 2 * (q[0] * p[1] - q[1] * p[0]);
  out[0] = n2 * p[0] + (q[3] * cx + (q[1] * cz - q[2] * cy));
  out[1] = n2 * p[1] + (q[3] * cy + (q[2] * cx - q[0] * cz));
  out[2] = n2 * p[2] + (q[3] * cz + (q[0] * cy - q[1] * cx));
}

S3_HD void s3_act(const double* S, const double* p, double* out) {
  s3_rot_scale(S, p, out);
  out[0] += S[4]; out[1] += S[5]; out[2] += S[6];
}

S3_HD void s3_inverse(const double* S, double* out) {
  const double n2 = S[0] * S[0] + S[1] * S[1] + S[2] * S[2] + S[3] * S[3];
  out[0] = -S[0] / n2; out[1] = -S[1] / n2; out[2] = -S[2] / n2; out[3] = S[3] / n2;
  double t[3];
  s3_rot_scale(out, S + 4, t);
  out[4] = -t[0]; out[5] = -t[1]; out[6] = -t[2];
}

S3_HD void s3_mul(const double* A, const double* B, double* out) {
  double t[3];
  s3_rot_scale(A, B + 4, t);
  const double x = A[3] * B[0] + A[0] * B[3] + A[1] * B[2] - A[2] * B[1];
  const double y = A[3] * B[1] + A[1] * B[3] + A[2] * B[0] - A[0] * B[2];
  const double z = A[3] * B[2] + A[2] * B[3] + A[0] * B[1] - A[1] * B[0];
  const double w = A[3] * B[3] - A[0] * B[0] - A[1] * B[1] - A[2] * B[2];
  out[0] = x; out[1] = y; out[2] = z; out[3] = w;
  out[4] = A[4] + t[0]; out[5] = A[5] + t[1]; out[6] = A[6] + t[2];
}

// exp: tangent -> qt7.  translation = W upsilon, W = A Om + B Om^2 + C I  (closed form of int_0^1 exp(t(sigma I + Om)) dt)
S3_HD void s3_exp(const double* a, double* S) {
  const double* om = a + 3;
  const double sigma = a[6];
  const double t2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double theta, fi, fr;
  if (t2 < S3_EPS * S3_EPS) {
    theta = 0.0;
    const double t4 = t2 * t2;
    fi = 0.5 - t2 / 48.0 + t4 / 3840.0;
    fr = 1.0 - t2 / 8.0 + t4 / 384.0;
  } else {
    theta = sqrt(t2);
    fi = sin(0.5 * theta) / theta;
    fr = cos(0.5 * theta);
  }
  const double scale = exp(sigma), rs = sqrt(scale);
  S[0] = rs * fi * om[0]; S[1] = rs * fi * om[1]; S[2] = rs * fi * om[2]; S[3] = rs * fr;
  double A, B, C;
  if (fabs(sigma) < S3_EPS) {
    C = 1.0;
    if (fabs(theta) < S3_EPS) { A = 0.5; B = 1.0 / 6.0; }
    else { const double th2 = theta * theta; A = (1.0 - cos(theta)) / th2; B = (theta - sin(theta)) / (th2 * theta); }
  } else {
    C = (scale - 1.0) / sigma;
    if (fabs(theta) < S3_EPS) {
      const double s2 = sigma * sigma;
      A = ((sigma - 1.0) * scale + 1.0) / s2;
      B = (scale * 0.5 * s2 + scale - 1.0 - sigma * scale) / (s2 * sigma);
    } else {
      const double th2 = theta * theta, sa = scale * sin(theta), sb = scale * cos(theta), cc = th2 + sigma * sigma;
      A = (sa * sigma + (1.0 - sb) * theta) / (theta * cc);
      B = (C - ((sb - 1.0) * sigma + sa * theta) / cc) * 1.0 / th2;
    }
  }
  s3_poly_apply(om, A, B, C, a, S + 4);
}

// log: qt7 -> tangent
S3_HD void s3_log(const double* S, double* out) {
  const double scale = S[0] * S[0] + S[1] * S[1] + S[2] * S[2] + S[3] * S[3], sigma = log(scale);
  const double inv = 1.0 / sqrt(scale);
  const double ux = S[0] * inv, uy = S[1] * inv, uz = S[2] * inv, w = S[3] * inv;
  const double n2 = ux * ux + uy * uy + uz * uz;
  double f, theta;
  if (n2 < S3_EPS * S3_EPS) {
    f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w);
    theta = 2.0 * n2 / w;
  } else {
    const double n = sqrt(n2);
    const double at = (w < 0.0) ? atan2(-n, -w) : atan2(n, w);
    f = 2.0 * at / n;
    theta = f * n;
  }
  double om[3] = {f * ux, f * uy, f * uz};
  const double scale_sq = scale * scale, th2 = theta * theta, st = sin(theta), ct = cos(theta);
  double a, b, c;
  if (fabs(sigma * sigma) < S3_EPS) {
    c = 1.0 - 0.5 * sigma;
    a = -0.5;
    if (fabs(th2) < S3_EPS) b = 1.0 / 12.0;
    else b = (theta * st + 2.0 * ct - 2.0) / (2.0 * th2 * (ct - 1.0));
  } else {
    const double scale_cu = scale_sq * scale;
    c = sigma / (scale - 1.0);
    if (fabs(th2) < S3_EPS) {
      a = (-sigma * scale + scale - 1.0) / ((scale - 1.0) * (scale - 1.0));
      b = (scale_sq * sigma - 2.0 * scale_sq + scale * sigma + 2.0 * scale) / (2.0 * scale_cu - 6.0 * scale_sq + 6.0 * scale - 2.0);
    } else {
      const double ss = scale * st, sc = scale * ct;
      a = (theta * sc - theta - sigma * ss) / (theta * (scale_sq - 2.0 * sc + 1.0));
      b = -scale * (theta * ss - theta * st + sigma * sc - scale * sigma + sigma * ct - sigma) /
          (th2 * (scale_cu - 2.0 * scale * sc - scale_sq + 2.0 * sc + scale - 1.0));
    }
  }
  s3_poly_apply(om, a, b, c, S + 4, out);
  out[3] = om[0]; out[4] = om[1]; out[5] = om[2]; out[6] = sigma;
}

// Sim3Parameterization::Plus (src/CeresOptimizer.cc:24-41)
S3_HD void s3_plus(const double* x, const double* d, double* out) {
  double dd[7], Sx[7], Sd[7], P[7];
#pragma unroll
  for (int i = 0; i < 7; i++) dd[i] = d[i];
  dd[6] = fmax(dd[6], -20.0);
  s3_exp(x, Sx);
  s3_exp(dd, Sd);
  s3_mul(Sx, Sd, P);
  s3_log(P, out);
}

// Sim3ErrorTerm::Evaluate (include/CeresOptimizer.h:178-236) with the Huber corrector folded in.
// S is already exp(x) for the forward term or its inverse for the inverse term.  J (2x7 row-major) may be NULL.
S3_HD double s3_term_eval(const double* K4, const double* S, const double* P, double u_obs, double v_obs, double w,
                          double huber, double* r, double* J) {
  double p[3];
  s3_act(S, P, p);
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const double pr0 = fx * p[0] + cx * p[2], pr1 = fy * p[1] + cy * p[2], pr2 = p[2];
  const double r0 = w * (pr0 / pr2 - u_obs), r1 = w * (pr1 / pr2 - v_obs);
  const double s = r0 * r0 + r1 * r1;
  double rho0 = s, rho1 = 1.0;
  const double b = huber * huber;
  if (s > b) {
    const double rr = sqrt(s);
    rho0 = 2 * huber * rr - b;
    rho1 = fmax(DBL_MIN, huber / rr);
  }
  const double sq = sqrt(rho1);
  if (J) {
    const double Z2 = p[2] * p[2];
    const double c00 = fx / p[2], c02 = -p[0] * fx / Z2, c11 = fy / p[2], c12 = -fy * p[1] / Z2;
    const double k = sq * w;
    // J_camera * [I | -hat(p) | p]
    J[0] = k * c00;              J[1] = k * 0.0;              J[2] = k * c02;
    J[3] = k * (c02 * p[1]);     J[4] = k * (c00 * p[2] - c02 * p[0]);   J[5] = k * (-c00 * p[1]);
    J[6] = k * (c00 * p[0] + c02 * p[2]);
    J[7] = k * 0.0;              J[8] = k * c11;              J[9] = k * c12;
    J[10] = k * (-c11 * p[2] + c12 * p[1]);   J[11] = k * (-c12 * p[0]);   J[12] = k * (c11 * p[0]);
    J[13] = k * (c11 * p[1] + c12 * p[2]);
  }
  r[0] = sq * r0; r[1] = sq * r1;
  return rho0;
}

// :695-709 -- CheckOutlier fed with Eigen::Quaterniond(s*R) (matrix->quaternion on the SCALED matrix) and Eigen's
// unit-quaternion rotation formula; restated as the reference computes it.
S3_HD int s3_check_outlier(const double* K4, const double* S, const double* P, double u_obs, double v_obs, float inv_sigma,
                           double thres) {
  const double scale = S[0] * S[0] + S[1] * S[1] + S[2] * S[2] + S[3] * S[3];
  const double inv = 1.0 / sqrt(scale);
  const double x = S[0] * inv, y = S[1] * inv, z = S[2] * inv, w = S[3] * inv;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x,
               txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double M[3][3] = {{scale * (1 - (tyy + tzz)), scale * (txy - twz), scale * (txz + twy)},
                    {scale * (txy + twz), scale * (1 - (txx + tzz)), scale * (tyz - twx)},
                    {scale * (txz - twy), scale * (tyz + twx), scale * (1 - (txx + tyy))}};
  double q[4];
  double t = M[0][0] + M[1][1] + M[2][2];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M[2][1] - M[1][2]) * t; q[1] = (M[0][2] - M[2][0]) * t; q[2] = (M[1][0] - M[0][1]) * t;
  } else {
    int i = 0;
    if (M[1][1] > M[0][0]) i = 1;
    if (M[2][2] > M[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(M[i][i] - M[j][j] - M[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M[k][j] - M[j][k]) * t;
    q[j] = (M[j][i] + M[i][j]) * t;
    q[k] = (M[k][i] + M[i][k]) * t;
  }
  const double ux = 2 * (q[1] * P[2] - q[2] * P[1]), uy = 2 * (q[2] * P[0] - q[0] * P[2]), uz = 2 * (q[0] * P[1] - q[1] * P[0]);
  const double c0 = P[0] + (q[3] * ux + (q[1] * uz - q[2] * uy)) + S[4];
  const double c1 = P[1] + (q[3] * uy + (q[2] * ux - q[0] * uz)) + S[5];
  const double c2 = P[2] + (q[3] * uz + (q[0] * uy - q[1] * ux)) + S[6];
  const double px = K4[0] * c0 + K4[2] * c2, py = K4[1] * c1 + K4[3] * c2;
  const double eu = u_obs - px / c2, ev = v_obs - py / c2;
  return ((eu * eu + ev * ev) * (double)inv_sigma > thres) ? 1 : 0;
}


// ---- pieces of the essential-graph error term (include/CeresOptimizer.h:266-330) --------------------------------------
// Sophus::Sim3d::Adj(): [[s R, [t]x R, -t], [0, R, 0], [0, 0, 1]]  (7x7 row-major)
S3_HD void s3_adj(const double* S, double* A) {
  const double scale = S[0] * S[0] + S[1] * S[1] + S[2] * S[2] + S[3] * S[3];
  const double inv = 1.0 / sqrt(scale);
  const double x = S[0] * inv, y = S[1] * inv, z = S[2] * inv, w = S[3] * inv;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x,
               txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
  const double T[3][3] = {{0, -S[6], S[5]}, {S[6], 0, -S[4]}, {-S[5], S[4], 0}};
#pragma unroll
  for (int k = 0; k < 49; k++) A[k] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      A[i * 7 + j] = scale * R[i][j];
      A[i * 7 + 3 + j] = T[i][0] * R[0][j] + T[i][1] * R[1][j] + T[i][2] * R[2][j];
      A[(3 + i) * 7 + 3 + j] = R[i][j];
    }
  A[6] = -S[4]; A[13] = -S[5]; A[20] = -S[6];
  A[48] = 1.0;
}

// residual r = log(Sji * exp(lie_i) * exp(lie_j)^-1) and, if Ji != NULL, J_i = (I + ad/2 + ad^2/12) * Adj(exp(lie_j)); J_j = -J_i
S3_HD void s3_graph_edge(const double* lie_j, const double* lie_i, const double* Sji, double* r, double* Ji) {
  double Si[7], Sj[7], Sjinv[7], T1[7], E[7];
  s3_exp(lie_i, Si); s3_exp(lie_j, Sj);
  s3_inverse(Sj, Sjinv);
  s3_mul(Sji, Si, T1);
  s3_mul(T1, Sjinv, E);
  s3_log(E, r);
  if (!Ji) return;
  double ad[49];
#pragma unroll
  for (int k = 0; k < 49; k++) ad[k] = 0.0;
  const double ux = r[0], uy = r[1], uz = r[2], wx = r[3], wy = r[4], wz = r[5], sg = r[6];
  const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}}, U[3][3] = {{0, -uz, uy}, {uz, 0, -ux}, {-uy, ux, 0}};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { ad[i * 7 + j] = W[i][j] + (i == j ? sg : 0.0); ad[i * 7 + 3 + j] = U[i][j]; ad[(3 + i) * 7 + 3 + j] = W[i][j]; }
  ad[6] = -ux; ad[13] = -uy; ad[20] = -uz;
  double Jr[49], Adj[49];
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) {
      double s2 = 0;
      for (int k = 0; k < 7; k++) s2 += ad[i * 7 + k] * ad[k * 7 + j];
      Jr[i * 7 + j] = (i == j ? 1.0 : 0.0) + 0.5 * ad[i * 7 + j] + 1.0 / 12. * s2;
    }
  s3_adj(Sj, Adj);
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) {
      double s2 = 0;
      for (int k = 0; k < 7; k++) s2 += Jr[i * 7 + k] * Adj[k * 7 + j];
      Ji[i * 7 + j] = s2;
    }
}

}  // namespace orbhip
