// ============================================================================
// oracle/tri_oracle.cpp -- CPU restatement of the per-match triangulation + gates of LocalMapping::CreateNewMapPoints
// (reference src/LocalMapping.cc:267-378, monocular branch): ray parallax test, linear triangulation (4x4 homogeneous
// system, null vector by SVD), positive depth in both keyframes, chi-square reprojection gates, scale consistency.
//
// TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header for the rule).
//
// PARITY STATUS: "parity unpinned".  The reference takes the null vector from Eigen::JacobiSVD<Matrix4d> (Eigen is
// absent here); any backward-stable SVD gives the same vector up to rounding (it is divided by its last component, which
// fixes the sign), so the restatement uses a one-sided (Hestenes) Jacobi SVD in double and is pinned against LAPACK
// (numpy.linalg.svd) in tests/test_oracle_tri.py.  The float / double mix of every gate follows the reference
// expression by expression.
// ============================================================================
#include <cmath>
#include <cstdint>

namespace {
// right singular vector of the smallest singular value of a 4x4 matrix (row-major), one-sided Jacobi
void null_vector4(const double A[16], double x[4]) {
  double U[4][4], V[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { U[i][j] = A[4 * i + j]; V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 4; i++) { alpha += U[i][p] * U[i][p]; beta += U[i][q] * U[i][q]; gamma += U[i][p] * U[i][q]; }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-16 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
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
  for (int j = 0; j < 4; j++) {
    double nrm = 0;
    for (int i = 0; i < 4; i++) nrm += U[i][j] * U[i][j];
    if (nrm < bn) { bn = nrm; best = j; }
  }
  for (int i = 0; i < 4; i++) x[i] = V[i][best];
}
}  // namespace

extern "C" {

void orc_null_vector4(const double* A, double* x) { null_vector4(A, x); }

// T1 / T2 = [Rcw | tcw] row-major 3x4 of the current / neighbour keyframe; K = (fx, fy, cx, cy) float, inv = 1.0f / f as
// KeyFrame stores invfx_ / invfy_; kp = n x (x, y, octave) float of the matched UNDISTORTED keypoints (already gathered by
// matched_indices_); ratio_factor = 1.5f * scale_factor_.  ok[i] = 1 and x3D[i] set when the match survives every gate.
void orc_triangulate_matches(const double* T1, const double* T2, const float* K1, const float* K2, const float* kp1, const float* kp2,
                             int n, const float* level_sigma2, const float* scale_factors, float ratio_factor, double* x3D, uint8_t* ok) {
  double Ow1[3], Ow2[3];
  for (int k = 0; k < 3; k++) {
    Ow1[k] = -(T1[k] * T1[3] + T1[4 + k] * T1[7] + T1[8 + k] * T1[11]);      // -Rcw^T tcw (KeyFrame::GetCameraCenter)
    Ow2[k] = -(T2[k] * T2[3] + T2[4 + k] * T2[7] + T2[8 + k] * T2[11]);
  }
  const float fx1 = K1[0], fy1 = K1[1], cx1 = K1[2], cy1 = K1[3], invfx1 = 1.0f / fx1, invfy1 = 1.0f / fy1;
  const float fx2 = K2[0], fy2 = K2[1], cx2 = K2[2], cy2 = K2[3], invfx2 = 1.0f / fx2, invfy2 = 1.0f / fy2;
  for (int m = 0; m < n; m++) {
    ok[m] = 0; x3D[3 * m] = x3D[3 * m + 1] = x3D[3 * m + 2] = 0.0;
    const float x1p = kp1[3 * m], y1p = kp1[3 * m + 1], x2p = kp2[3 * m], y2p = kp2[3 * m + 1];
    const int o1 = (int)kp1[3 * m + 2], o2 = (int)kp2[3 * m + 2];
    const double xn1[3] = {(double)((x1p - cx1) * invfx1), (double)((y1p - cy1) * invfy1), 1.0};
    const double xn2[3] = {(double)((x2p - cx2) * invfx2), (double)((y2p - cy2) * invfy2), 1.0};
    double ray1[3], ray2[3];
    for (int k = 0; k < 3; k++) {                       // Rwc * xn = Rcw^T * xn
      ray1[k] = T1[k] * xn1[0] + T1[4 + k] * xn1[1] + T1[8 + k] * xn1[2];
      ray2[k] = T2[k] * xn2[0] + T2[4 + k] * xn2[1] + T2[8 + k] * xn2[2];
    }
    const double dot = ray1[0] * ray2[0] + ray1[1] * ray2[1] + ray1[2] * ray2[2];
    const double n1 = std::sqrt(ray1[0] * ray1[0] + ray1[1] * ray1[1] + ray1[2] * ray1[2]);
    const double n2 = std::sqrt(ray2[0] * ray2[0] + ray2[1] * ray2[1] + ray2[2] * ray2[2]);
    const float cosPar = (float)(dot / (n1 * n2));
    const float cosStereo = cosPar + 1;
    if (!(cosPar < cosStereo && cosPar > 0 && cosPar < 0.9998)) continue;
    double A[16];
    for (int j = 0; j < 4; j++) {
      A[j] = xn1[0] * T1[8 + j] - T1[j];
      A[4 + j] = xn1[1] * T1[8 + j] - T1[4 + j];
      A[8 + j] = xn2[0] * T2[8 + j] - T2[j];
      A[12 + j] = xn2[1] * T2[8 + j] - T2[4 + j];
    }
    double x4[4];
    null_vector4(A, x4);
    if (x4[3] == 0) continue;
    const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
    const float z1 = (float)(T1[8] * X[0] + T1[9] * X[1] + T1[10] * X[2] + T1[11]);
    if (z1 <= 0) continue;
    const float z2 = (float)(T2[8] * X[0] + T2[9] * X[1] + T2[10] * X[2] + T2[11]);
    if (z2 <= 0) continue;
    const float s1 = level_sigma2[o1];
    const float x1 = (float)(T1[0] * X[0] + T1[1] * X[1] + T1[2] * X[2] + T1[3]);
    const float y1 = (float)(T1[4] * X[0] + T1[5] * X[1] + T1[6] * X[2] + T1[7]);
    const float invz1 = (float)(1.0 / z1);
    const float u1 = fx1 * x1 * invz1 + cx1, v1 = fy1 * y1 * invz1 + cy1;
    const float ex1 = u1 - x1p, ey1 = v1 - y1p;
    if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)s1) continue;
    const float s2 = level_sigma2[o2];
    const float x2 = (float)(T2[0] * X[0] + T2[1] * X[1] + T2[2] * X[2] + T2[3]);
    const float y2 = (float)(T2[4] * X[0] + T2[5] * X[1] + T2[6] * X[2] + T2[7]);
    const float invz2 = (float)(1.0 / z2);
    const float u2 = fx2 * x2 * invz2 + cx2, v2 = fy2 * y2 * invz2 + cy2;
    const float ex2 = u2 - x2p, ey2 = v2 - y2p;
    if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)s2) continue;
    const double d1x = X[0] - Ow1[0], d1y = X[1] - Ow1[1], d1z = X[2] - Ow1[2];
    const double d2x = X[0] - Ow2[0], d2y = X[1] - Ow2[1], d2z = X[2] - Ow2[2];
    const float dist1 = (float)std::sqrt(d1x * d1x + d1y * d1y + d1z * d1z), dist2 = (float)std::sqrt(d2x * d2x + d2y * d2y + d2z * d2z);
    if (dist1 == 0 || dist2 == 0) continue;
    const float ratioDist = dist2 / dist1;
    const float ratioOctave = scale_factors[o1] / scale_factors[o2];
    if (ratioDist * ratio_factor < ratioOctave || ratioDist > ratioOctave * ratio_factor) continue;
    ok[m] = 1; x3D[3 * m] = X[0]; x3D[3 * m + 1] = X[1]; x3D[3 * m + 2] = X[2];
  }
}

}  // extern "C"
