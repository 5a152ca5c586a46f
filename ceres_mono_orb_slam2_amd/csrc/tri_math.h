// Per-match body of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:267-378, monocular): ray parallax, linear triangulation, depth /
// reprojection / scale gates - shared by k_triangulate (orb_frame.hip) and the device-resident CreateNewMapPoints (orb_localmap.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace orbhip {

struct TriCam { double T1[12], T2[12], Ow1[3], Ow2[3]; float K1[4], K2[4]; float ratio_factor; };

// right singular vector of the smallest singular value of a 4x4 matrix: one-sided (Hestenes) Jacobi, double
__device__ inline void null_vector4_dev(const double* A, double* x) {
  double U[4][4], V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) { U[i][j] = A[4 * i + j]; V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { alpha += U[i][p] * U[i][p]; beta += U[i][q] * U[i][q]; gamma += U[i][p] * U[i][q]; }
        if (gamma == 0.0 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double up = U[i][p], uq = U[i][q];
          U[i][p] = c * up - s * uq; U[i][q] = s * up + c * uq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  int best = 0; double bn = 1e300;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double nrm = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) nrm += U[i][j] * U[i][j];
    if (nrm < bn) { bn = nrm; best = j; }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) x[i] = best == 0 ? V[i][0] : best == 1 ? V[i][1] : best == 2 ? V[i][2] : V[i][3];
}

// one match: keypoint 1 (x1p, y1p, octave o1) of the current keyframe, keypoint 2 of the neighbour; true = accepted, X = the new point
__device__ inline bool triangulate_one(const TriCam& C, const float x1p, const float y1p, const int o1, const float x2p, const float y2p, const int o2,
                                       const float* __restrict__ level_sigma2, const float* __restrict__ scale_factors, double* __restrict__ Xout) {
  const double* T1 = C.T1; const double* T2 = C.T2;
  const float fx1 = C.K1[0], fy1 = C.K1[1], cx1 = C.K1[2], cy1 = C.K1[3], invfx1 = 1.0f / fx1, invfy1 = 1.0f / fy1;
  const float fx2 = C.K2[0], fy2 = C.K2[1], cx2 = C.K2[2], cy2 = C.K2[3], invfx2 = 1.0f / fx2, invfy2 = 1.0f / fy2;
  const double xn1[3] = {(double)((x1p - cx1) * invfx1), (double)((y1p - cy1) * invfy1), 1.0};
  const double xn2[3] = {(double)((x2p - cx2) * invfx2), (double)((y2p - cy2) * invfy2), 1.0};
  double ray1[3], ray2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    ray1[k] = T1[k] * xn1[0] + T1[4 + k] * xn1[1] + T1[8 + k] * xn1[2];
    ray2[k] = T2[k] * xn2[0] + T2[4 + k] * xn2[1] + T2[8 + k] * xn2[2];
  }
  const double dot = ray1[0] * ray2[0] + ray1[1] * ray2[1] + ray1[2] * ray2[2];
  const double n1 = sqrt(ray1[0] * ray1[0] + ray1[1] * ray1[1] + ray1[2] * ray1[2]);
  const double n2 = sqrt(ray2[0] * ray2[0] + ray2[1] * ray2[1] + ray2[2] * ray2[2]);
  const float cosPar = (float)(dot / (n1 * n2));
  const float cosStereo = cosPar + 1;
  if (!(cosPar < cosStereo && cosPar > 0 && cosPar < 0.9998)) return false;     // (:296-297) no stereo and very low parallax
  double A[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    A[j] = xn1[0] * T1[8 + j] - T1[j];
    A[4 + j] = xn1[1] * T1[8 + j] - T1[4 + j];
    A[8 + j] = xn2[0] * T2[8 + j] - T2[j];
    A[12 + j] = xn2[1] * T2[8 + j] - T2[4 + j];
  }
  double x4[4];
  null_vector4_dev(A, x4);
  if (x4[3] == 0) return false;
  const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
  const float z1 = (float)(T1[8] * X[0] + T1[9] * X[1] + T1[10] * X[2] + T1[11]);
  if (z1 <= 0) return false;
  const float z2 = (float)(T2[8] * X[0] + T2[9] * X[1] + T2[10] * X[2] + T2[11]);
  if (z2 <= 0) return false;
  const float s1 = level_sigma2[o1];
  const float x1 = (float)(T1[0] * X[0] + T1[1] * X[1] + T1[2] * X[2] + T1[3]);
  const float y1 = (float)(T1[4] * X[0] + T1[5] * X[1] + T1[6] * X[2] + T1[7]);
  const float invz1 = (float)(1.0 / z1);
  const float u1 = fx1 * x1 * invz1 + cx1, v1 = fy1 * y1 * invz1 + cy1;
  const float ex1 = u1 - x1p, ey1 = v1 - y1p;
  if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)s1) return false;
  const float s2 = level_sigma2[o2];
  const float x2 = (float)(T2[0] * X[0] + T2[1] * X[1] + T2[2] * X[2] + T2[3]);
  const float y2 = (float)(T2[4] * X[0] + T2[5] * X[1] + T2[6] * X[2] + T2[7]);
  const float invz2 = (float)(1.0 / z2);
  const float u2 = fx2 * x2 * invz2 + cx2, v2 = fy2 * y2 * invz2 + cy2;
  const float ex2 = u2 - x2p, ey2 = v2 - y2p;
  if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)s2) return false;
  const double d1x = X[0] - C.Ow1[0], d1y = X[1] - C.Ow1[1], d1z = X[2] - C.Ow1[2];
  const double d2x = X[0] - C.Ow2[0], d2y = X[1] - C.Ow2[1], d2z = X[2] - C.Ow2[2];
  const float dist1 = (float)sqrt(d1x * d1x + d1y * d1y + d1z * d1z), dist2 = (float)sqrt(d2x * d2x + d2y * d2y + d2z * d2z);
  if (dist1 == 0 || dist2 == 0) return false;
  const float ratioDist = dist2 / dist1;
  const float ratioOctave = scale_factors[o1] / scale_factors[o2];
  if (ratioDist * C.ratio_factor < ratioOctave || ratioDist > ratioOctave * C.ratio_factor) return false;
  Xout[0] = X[0]; Xout[1] = X[1]; Xout[2] = X[2];
  return true;
}

}  // namespace orbhip
