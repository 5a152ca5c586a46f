// ============================================================================
// oracle/orb_oracle.cpp -- CPU restatement of the reference ORB front-end.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (the package
// ceres_mono_orb_slam2_amd/ and its HIP library) may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY STATUS: "parity unpinned".  The reference (b51/ceres_mono_orb_slam2)
// ships no tests or golden vectors, and its pixel arithmetic lives in OpenCV
// (un-vendored, unpinned: README says 2.4.11 / 3.2), which is absent here, so the
// reference cannot be built or run.  This file restates
//   * src/ORBextractor.cc (whole file except ComputeKeyPointsOld) and
//   * the OpenCV 2.4/3.2 semantics of cv::resize(INTER_LINEAR, 8U), cv::FAST
//     (FAST-9/16 + score + 3x3 NMS), cv::GaussianBlur(7x7, sigma 2, 8U),
//     cv::fastAtan2, cvRound/cvFloor/cvCeil  
// and is pinned only against (a) the known-answer values derivable from the
// reference source (quotas, umax, scale tables, pattern table, pyramid sizes)
// and (b) independent numpy re-implementations in tests/.
//
// Canonical choices where the reference is not deterministic:
//   * octree tie-break among equal-size nodes = creation sequence (later first);
//   * BRIEF sampling coordinates use un-contracted float mul/add
//     (compile with -ffp-contract=off) and cos/sin = float(det_cos/det_sin in
//     double) as specified in det_sincos() below.
// Single-threaded, plain C++17, no dependencies.
// ============================================================================
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <vector>
#include <algorithm>
#include <utility>

#include "orb_pattern_data.h"

namespace {

const int PATCH_SIZE = 31;        // src/ORBextractor.cc:72
const int HALF_PATCH_SIZE = 15;   // :73
const int EDGE_THRESHOLD = 19;    // :74

// ---- OpenCV scalar helpers -------------------------------------
inline int cv_round(double v) { return (int)std::nearbyint(v); }   // round-half-even (default FE mode)
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

struct KeyPoint {     // mirrors cv::KeyPoint's 7 fields (28 bytes)
  float x, y, size, angle, response;
  int octave, class_id;
};

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> px;   // dense row-major, stride == w
  uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

// ---- cv::fastAtan2 (degrees), OpenCV 2.4/3.x scalar form --------
float fast_atan2_deg(float y, float x) {
  static const float k = (float)(180.0 / 3.14159265358979323846);
  static const float p1 = 0.9997878412794807f * k;
  static const float p3 = -0.3258083974640975f * k;
  static const float p5 = 0.1555786518463281f * k;
  static const float p7 = -0.04432655554792128f * k;
  const float eps = (float)2.2204460492503131e-16;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- deterministic sin/cos (project-canonical stand-in for libm cosf/sinf) --
// Spec: x (radians, 0 <= x < ~6.3) in double; k = nearest integer to x*(2/pi);
// r = (x - k*PIO2_HI) - k*PIO2_LO; sin/cos of r by the fdlibm kernel
// polynomials evaluated in Horner form WITHOUT fused multiply-add; quadrant fix
// by k&3.  Result is rounded to float by the caller.
void det_sincos(double x, double* s_out, double* c_out) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double PIO2_LO = 6.07710050650619224932e-11;   // pi/2 - PIO2_HI
  double kd = std::nearbyint(x * TWO_OVER_PI);
  int k = (int)kd;
  double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
  double z = r * r;
  // sin kernel
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = ((((S6 * z + S5) * z + S4) * z + S3) * z + S2) * z + S1;
  double sn = r + (r * z) * ps;
  // cos kernel
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = ((((C6 * z + C5) * z + C4) * z + C3) * z + C2) * z + C1;
  double cs = (1.0 - 0.5 * z) + (z * z) * pc;
  switch (k & 3) {
    case 0: *s_out = sn;  *c_out = cs;  break;
    case 1: *s_out = cs;  *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
  }
}

// ---- extractor state (src/ORBextractor.cc:410-470) --------------------------
struct Extractor {
  int nfeatures; double scaleFactor; int nlevels; int iniThFAST, minThFAST;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota, umax;
  // per-call state
  std::vector<Image> pyr, blurred;
  std::vector<std::vector<KeyPoint>> cands;       // pre-octree candidates per level (window coords)
  std::vector<std::vector<KeyPoint>> level_kps;   // post-octree, level coordinates (unscaled), with angle

  Extractor(int nf, float sf, int nl, int ini, int mn)
      : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniThFAST(ini), minThFAST(mn) {
    scale.resize(nl); sigma2.resize(nl); inv_scale.resize(nl); inv_sigma2.resize(nl);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
      scale[i] = (float)(scale[i - 1] * scaleFactor);          // float*double -> float (:421)
      sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; i++) { inv_scale[i] = 1.0f / scale[i]; inv_sigma2[i] = 1.0f / sigma2[i]; }
    quota.resize(nl);
    float factor = (float)(1.0f / scaleFactor);                 // :436
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
      quota[l] = cv_round(nDesired);
      sum += quota[l];
      nDesired *= factor;
    }
    quota[nl - 1] = std::max(nfeatures - sum, 0);
    // umax (:452-469)
    umax.assign(HALF_PATCH_SIZE + 1, 0);
    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }
};

// ---- cv::resize INTER_LINEAR, CV_8UC1 ---------------------------
void resize_linear_u8(const Image& src, Image& dst, int dw, int dh) {
  dst.w = dw; dst.h = dh; dst.px.assign((size_t)dw * dh, 0);
  const int sw = src.w, sh = src.h;
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    ialpha[2 * dx] = sat_short(cv_round(c0 * 2048));
    ialpha[2 * dx + 1] = sat_short(cv_round(c1 * 2048));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    ibeta[2 * dy] = sat_short(cv_round(c0 * 2048));
    ibeta[2 * dy + 1] = sat_short(cv_round(c1 * 2048));
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = &src.px[(size_t)sy * sw];
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = std::min(sx + 1, sw - 1);          // second tap clamped (its weight is 0 there)
      out[dx] = S[sx] * ialpha[2 * dx] + S[sx1] * ialpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
    int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
    hresize(sy0, row0);
    hresize(sy1, row1);
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = &dst.px[(size_t)dy * dw];
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- ComputePyramid (src/ORBextractor.cc:1107-1132) -------------------------
// The reflect-101 border added by copyMakeBorder is never read by the mono
// pipeline, so level images are stored without it.
void compute_pyramid(Extractor& E, const uint8_t* img, int w, int h, int stride) {
  E.pyr.assign(E.nlevels, Image());
  for (int l = 0; l < E.nlevels; l++) {
    float s = E.inv_scale[l];
    int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
    if (l == 0) {
      Image& I = E.pyr[0];
      I.w = lw; I.h = lh; I.px.resize((size_t)lw * lh);
      for (int y = 0; y < h; y++) std::memcpy(&I.px[(size_t)y * w], img + (size_t)y * stride, w);
    } else {
      resize_linear_u8(E.pyr[l - 1], E.pyr[l], lw, lh);
    }
  }
}

// ---- cv::FAST(img, kps, threshold, nms=true) on a sub-image -----
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// is-corner test at threshold t, literal definition (9 contiguous of 16, strict)
bool fast9_is_corner(const int d[16], int t) {   // d[k] = center - ring[k]
  for (int s = 0; s < 16; s++) {
    bool allb = true, alld = true;
    for (int k = 0; k < 9; k++) {
      int v = d[(s + k) & 15];
      if (!(v > t)) allb = false;     // ring darker than centre by more than t  (ring < p - t)
      if (!(v < -t)) alld = false;    // ring brighter than centre by more than t (ring > p + t)
    }
    if (allb || alld) return true;
  }
  return false;
}

// cornerScore<16>: OpenCV's min/max sweep, restated (returns the largest t for
// which the pixel is still a corner, or threshold-1.. for non-corners; only
// called on detected corners)
int fast9_score(const int d16[16], int threshold) {
  int d[25];
  for (int k = 0; k < 25; k++) d[k] = d16[k & 15];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]);
    a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

// FAST with NMS on the sub-image rows [y0,y1) x cols [x0,x1) of I.  Output in
// row-major order, coordinates relative to (x0,y0).
void fast_subimage(const Image& I, int x0, int y0, int x1, int y1, int threshold,
                   std::vector<KeyPoint>& out) {
  out.clear();
  const int cols = x1 - x0, rows = y1 - y0;
  if (cols < 7 || rows < 7) return;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> score((size_t)rows * cols, 0);
  for (int y = 3; y < rows - 3; y++)
    for (int x = 3; x < cols - 3; x++) {
      int d[16];
      int p = I.at(y0 + y, x0 + x);
      for (int k = 0; k < 16; k++) d[k] = p - I.at(y0 + y + RING_DY[k], x0 + x + RING_DX[k]);
      if (fast9_is_corner(d, threshold)) score[(size_t)y * cols + x] = (uint8_t)fast9_score(d, threshold);
    }
  // a detected corner always has score >= threshold; at threshold 0 a score of 0 is
  // possible, and OpenCV then keeps it only if > all neighbours, i.e. never (0 > 0 false).
  for (int y = 3; y < rows - 3; y++)
    for (int x = 3; x < cols - 3; x++) {
      int s = score[(size_t)y * cols + x];
      if (s == 0) {
        // distinguish "not a corner" from "corner with score 0": both are dropped by the
        // strict > test against (at least one) zero-valued frame/neighbour only when all
        // neighbours are 0 too -> 0 > 0 is false, so dropped either way.
        continue;
      }
      bool keep = true;
      for (int dy = -1; dy <= 1 && keep; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (!dx && !dy) continue;
          if (!(s > score[(size_t)(y + dy) * cols + (x + dx)])) { keep = false; break; }
        }
      if (keep) out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
    }
}

// ---- DistributeOctTree (src/ORBextractor.cc:481-763) -------------
struct Node {
  std::vector<KeyPoint> keys;
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::list<Node>::iterator lit;
  bool noMore = false;
  long seq = 0;     // creation sequence number: canonical replacement for the heap address (F9)
};

void divide_node(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(n.URx - n.ULx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(n.BRy - n.ULy) / 2);
  n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy;
  n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy;
  n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
  n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
  n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
  for (const KeyPoint& kp : n.keys) {
    if (kp.x < n1.URx) {
      if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp);
    } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
    else n4.keys.push_back(kp);
  }
  n1.noMore = n1.keys.size() == 1; n2.noMore = n2.keys.size() == 1;
  n3.noMore = n3.keys.size() == 1; n4.noMore = n4.keys.size() == 1;
}

std::vector<KeyPoint> distribute_octree(const std::vector<KeyPoint>& in, int minX, int maxX, int minY,
                                        int maxY, int N) {
  if (in.empty()) return {};   // (also covers degenerate windows of tiny pyramid levels)
  int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni < 1) nIni = 1;   // reference would index an empty vector here (portrait windows); guarded
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> L;
  std::vector<Node*> ini(nIni);
  long seq = 0;
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
    ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY;
    ni.BRx = ni.URx; ni.BRy = maxY - minY;
    ni.seq = seq++;
    L.push_back(ni);
    ini[i] = &L.back();
  }
  for (const KeyPoint& kp : in) {
    int idx = (int)(kp.x / hX);
    if (idx >= nIni) idx = nIni - 1;   // cannot happen for in-window points; guard only
    ini[idx]->keys.push_back(kp);
  }
  for (auto lit = L.begin(); lit != L.end();) {
    if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
    else if (lit->keys.empty()) lit = L.erase(lit);
    else ++lit;
  }
  bool finish = false;
  typedef std::pair<int, Node*> SP;
  auto sp_less = [](const SP& a, const SP& b) {
    if (a.first != b.first) return a.first < b.first;
    return a.second->seq < b.second->seq;
  };
  std::vector<SP> sizeAndNode;
  auto push_children = [&](Node* kids[4], int& nToExpand) {
    for (int c = 0; c < 4; c++) {
      Node& k = *kids[c];
      if (k.keys.size() > 0) {
        k.seq = seq++;
        L.push_front(k);
        if (k.keys.size() > 1) {
          nToExpand++;
          sizeAndNode.push_back(std::make_pair((int)k.keys.size(), &L.front()));
          L.front().lit = L.begin();
        }
      }
    }
  };
  while (!finish) {
    int prevSize = (int)L.size();
    auto lit = L.begin();
    int nToExpand = 0;
    sizeAndNode.clear();
    while (lit != L.end()) {
      if (lit->noMore) { ++lit; continue; }
      Node n1, n2, n3, n4;
      divide_node(*lit, n1, n2, n3, n4);
      Node* kids[4] = {&n1, &n2, &n3, &n4};
      push_children(kids, nToExpand);
      lit = L.erase(lit);
    }
    if ((int)L.size() >= N || (int)L.size() == prevSize) {
      finish = true;
    } else if (((int)L.size() + nToExpand * 3) > N) {
      while (!finish) {
        prevSize = (int)L.size();
        std::vector<SP> prev = sizeAndNode;
        sizeAndNode.clear();
        std::sort(prev.begin(), prev.end(), sp_less);
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          divide_node(*prev[j].second, n1, n2, n3, n4);
          Node* kids[4] = {&n1, &n2, &n3, &n4};
          int dummy = 0;
          push_children(kids, dummy);
          L.erase(prev[j].second->lit);
          if ((int)L.size() >= N) break;
        }
        if ((int)L.size() >= N || (int)L.size() == prevSize) finish = true;
      }
    }
  }
  std::vector<KeyPoint> res;
  res.reserve(L.size());
  for (auto& n : L) {
    const KeyPoint* best = &n.keys[0];
    float maxR = best->response;
    for (size_t k = 1; k < n.keys.size(); k++)
      if (n.keys[k].response > maxR) { best = &n.keys[k]; maxR = n.keys[k].response; }
    res.push_back(*best);
  }
  return res;
}

// ---- per-level cell loop (src/ORBextractor.cc:765-848) -----------
void level_candidates(const Extractor& E, const Image& I, std::vector<KeyPoint>& cands, int& minBX,
                      int& maxBX, int& minBY, int& maxBY) {
  cands.clear();
  const float W = 30;
  const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
  const int maxBorderX = I.w - EDGE_THRESHOLD + 3, maxBorderY = I.h - EDGE_THRESHOLD + 3;
  minBX = minBorderX; maxBX = maxBorderX; minBY = minBorderY; maxBY = maxBorderY;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols < 1 || nRows < 1) return;    // reference divides by zero here; tiny levels yield nothing
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
  std::vector<KeyPoint> cell;
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      fast_subimage(I, (int)iniX, (int)iniY, (int)maxX, (int)maxY, E.iniThFAST, cell);
      if (cell.empty()) fast_subimage(I, (int)iniX, (int)iniY, (int)maxX, (int)maxY, E.minThFAST, cell);
      for (KeyPoint kp : cell) {
        kp.x += j * wCell;
        kp.y += i * hCell;
        cands.push_back(kp);
      }
    }
  }
}

// ---- IC_Angle (src/ORBextractor.cc:77-104) ----------------------------------
float ic_angle(const Image& I, float px, float py, const std::vector<int>& umax) {
  int m_01 = 0, m_10 = 0;
  const int cy = cv_round(py), cx = cv_round(px);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * I.at(cy, cx + u);
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0, d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = I.at(cy + v, cx + u), val_minus = I.at(cy - v, cx + u);
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2_deg((float)m_01, (float)m_10);
}

// ---- cv::GaussianBlur(7x7, sigma=2, REFLECT_101), CV_8U ---------
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
  return i;
}
// which OpenCV GaussianBlur is restated: 0 = the 8-bit taps of OpenCV <= 3.4.1 (below), 1 = the 8.8 fixed-point taps of the
// "bit-exact" ufixedpoint16 path of later versions, error-diffused to sum 256 (include/orbslam_hip.h: orbx_set_opencv_variant)
static int g_blur_variant = 0;
void gauss7_taps(int taps[7]) {
  if (g_blur_variant == 1) { const int t[7] = {18, 34, 48, 56, 48, 34, 18}; for (int i = 0; i < 7; i++) taps[i] = t[i]; return; }
  // getGaussianKernel(7, 2, CV_32F): float taps normalised by the float sum; then
  // convertTo(CV_32S, 256) = cvRound(tap*256)  ->  {18,34,49,55,49,34,18}
  float cf[7]; double sum = 0;
  const double scale2X = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)std::exp(scale2X * x * x); sum += cf[i]; }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); taps[i] = cv_round((double)cf[i] * 256.0); }
}
void gaussian_blur7(const Image& src, Image& dst) {
  int k[7]; gauss7_taps(k);
  const int w = src.w, h = src.h;
  dst.w = w; dst.h = h; dst.px.assign((size_t)w * h, 0);
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int t = 0; t < 7; t++) s += k[t] * src.at(y, reflect101(x + t - 3, w));
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int t = 0; t < 7; t++) s += k[t] * tmp[(size_t)reflect101(y + t - 3, h) * w + x];
      dst.px[(size_t)y * w + x] = sat_u8((s + (1 << 15)) >> 16);
    }
}

// ---- computeOrbDescriptor (src/ORBextractor.cc:107-147) ---------------------
void orb_descriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) {
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float angle = (float)kpt.angle * factorPI;
  double sd, cd;
  det_sincos((double)angle, &sd, &cd);
  float a = (float)cd, b = (float)sd;
  const int cy = cv_round(kpt.y), cx = cv_round(kpt.x);
  const signed char* pat = ORB_BIT_PATTERN_31;
  auto get = [&](int idx) -> int {
    float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
    float fy = px * b + py * a;     // un-contracted (-ffp-contract=off)
    float fx = px * a - py * b;
    return img.at(cy + cv_round(fy), cx + cv_round(fx));
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int k = 0; k < 8; k++) {
      int t0 = get(2 * k), t1 = get(2 * k + 1);
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

// ---- operator() (src/ORBextractor.cc:1043-1105) -----------------------------
int extract(Extractor& E, const uint8_t* img, int w, int h, int stride, KeyPoint* kps, uint8_t* desc,
            int cap) {
  if (!img || w <= 0 || h <= 0) return 0;
  compute_pyramid(E, img, w, h, stride);
  E.cands.assign(E.nlevels, {});
  E.level_kps.assign(E.nlevels, {});
  E.blurred.assign(E.nlevels, Image());
  for (int l = 0; l < E.nlevels; l++) {
    int minBX, maxBX, minBY, maxBY;
    level_candidates(E, E.pyr[l], E.cands[l], minBX, maxBX, minBY, maxBY);
    std::vector<KeyPoint>& K = E.level_kps[l];
    K = distribute_octree(E.cands[l], minBX, maxBX, minBY, maxBY, E.quota[l]);
    const int scaledPatchSize = (int)(PATCH_SIZE * E.scale[l]);
    for (KeyPoint& kp : K) { kp.x += minBX; kp.y += minBY; kp.octave = l; kp.size = (float)scaledPatchSize; }
  }
  for (int l = 0; l < E.nlevels; l++)
    for (KeyPoint& kp : E.level_kps[l]) kp.angle = ic_angle(E.pyr[l], kp.x, kp.y, E.umax);
  int n = 0;
  for (int l = 0; l < E.nlevels; l++) {
    std::vector<KeyPoint>& K = E.level_kps[l];
    if (K.empty()) continue;
    gaussian_blur7(E.pyr[l], E.blurred[l]);
    for (const KeyPoint& kp0 : K) {
      if (n >= cap) return -1;
      orb_descriptor(kp0, E.blurred[l], desc + (size_t)32 * n);
      KeyPoint kp = kp0;
      if (l != 0) { float s = E.scale[l]; kp.x *= s; kp.y *= s; }
      kps[n++] = kp;
    }
  }
  return n;
}

}  // namespace

// ============================ C entry points ================================
extern "C" {
int orc_set_blur_variant(int v) { if (v != 0 && v != 1) return -1; g_blur_variant = v; return 0; }

void* orc_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orc_destroy(void* h) { delete (Extractor*)h; }

// tables: scale[nl], inv_scale[nl], sigma2[nl], inv_sigma2[nl] (float), quota[nl], umax[16] (int)
void orc_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* quota,
                int* umax) {
  Extractor& E = *(Extractor*)h;
  for (int i = 0; i < E.nlevels; i++) {
    scale[i] = E.scale[i]; inv_scale[i] = E.inv_scale[i]; sigma2[i] = E.sigma2[i];
    inv_sigma2[i] = E.inv_sigma2[i]; quota[i] = E.quota[i];
  }
  for (int i = 0; i < 16; i++) umax[i] = E.umax[i];
}

// returns number of keypoints (<= cap) or -1 if cap too small; kps = 28-byte records
int orc_extract(void* h, const uint8_t* img, int w, int hgt, int stride, void* kps, uint8_t* desc, int cap) {
  return extract(*(Extractor*)h, img, w, hgt, stride, (KeyPoint*)kps, desc, cap);
}

// introspection of the last orc_extract call (for stage-by-stage parity tests)
void orc_level_dims(void* h, int level, int* w, int* hgt) {
  Extractor& E = *(Extractor*)h; *w = E.pyr[level].w; *hgt = E.pyr[level].h;
}
void orc_level_image(void* h, int level, int blurred, uint8_t* out) {
  Extractor& E = *(Extractor*)h;
  const Image& I = blurred ? E.blurred[level] : E.pyr[level];
  if (!I.px.empty()) std::memcpy(out, I.px.data(), I.px.size());
}
int orc_level_num_candidates(void* h, int level) { return (int)((Extractor*)h)->cands[level].size(); }
// out: int32 triples (x, y, score) in detection-window coordinates, candidate order
void orc_level_candidates(void* h, int level, int* out) {
  Extractor& E = *(Extractor*)h;
  int i = 0;
  for (const KeyPoint& kp : E.cands[level]) { out[i++] = (int)kp.x; out[i++] = (int)kp.y; out[i++] = (int)kp.response; }
}
int orc_level_num_keypoints(void* h, int level) { return (int)((Extractor*)h)->level_kps[level].size(); }
void orc_level_keypoints(void* h, int level, void* out) {
  Extractor& E = *(Extractor*)h;
  if (!E.level_kps[level].empty())
    std::memcpy(out, E.level_kps[level].data(), E.level_kps[level].size() * sizeof(KeyPoint));
}

// ---- stand-alone stage functions (unit-testable against numpy) -------------
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  Image S; S.w = sw; S.h = sh; S.px.assign(src, src + (size_t)sw * sh);
  Image D; resize_linear_u8(S, D, dw, dh);
  std::memcpy(dst, D.px.data(), D.px.size());
}
void orc_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
  Image S; S.w = w; S.h = h; S.px.assign(src, src + (size_t)w * h);
  Image D; gaussian_blur7(S, D);
  std::memcpy(dst, D.px.data(), D.px.size());
}
void orc_gauss7_taps(int* taps) { gauss7_taps(taps); }
// FAST+NMS on a whole (sub)image; out = int32 triples (x,y,score); returns count (<= cap)
int orc_fast(const uint8_t* src, int w, int h, int threshold, int* out, int cap) {
  Image S; S.w = w; S.h = h; S.px.assign(src, src + (size_t)w * h);
  std::vector<KeyPoint> k; fast_subimage(S, 0, 0, w, h, threshold, k);
  int n = 0;
  for (const KeyPoint& kp : k) { if (n >= cap) break; out[3 * n] = (int)kp.x; out[3 * n + 1] = (int)kp.y; out[3 * n + 2] = (int)kp.response; n++; }
  return (int)k.size();
}
// octree on explicit candidates: in = int32 triples (x,y,score); out = int32 triples; returns count
int orc_octree(const int* in, int n, int minX, int maxX, int minY, int maxY, int N, int* out, int cap) {
  std::vector<KeyPoint> v(n);
  for (int i = 0; i < n; i++) v[i] = KeyPoint{(float)in[3 * i], (float)in[3 * i + 1], 7.f, -1.f, (float)in[3 * i + 2], 0, -1};
  std::vector<KeyPoint> r = distribute_octree(v, minX, maxX, minY, maxY, N);
  int m = 0;
  for (const KeyPoint& kp : r) { if (m >= cap) break; out[3 * m] = (int)kp.x; out[3 * m + 1] = (int)kp.y; out[3 * m + 2] = (int)kp.response; m++; }
  return (int)r.size();
}
float orc_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void orc_det_sincos(double x, double* s, double* c) { det_sincos(x, s, c); }
float orc_ic_angle(const uint8_t* img, int w, int h, int x, int y) {
  Image S; S.w = w; S.h = h; S.px.assign(img, img + (size_t)w * h);
  Extractor E(1000, 1.2f, 8, 20, 7);
  return ic_angle(S, (float)x, (float)y, E.umax);
}
void orc_brief(const uint8_t* img, int w, int h, int x, int y, float angle_deg, uint8_t* desc) {
  Image S; S.w = w; S.h = h; S.px.assign(img, img + (size_t)w * h);
  KeyPoint kp{(float)x, (float)y, 31.f, angle_deg, 0.f, 0, -1};
  orb_descriptor(kp, S, desc);
}
const signed char* orc_pattern() { return ORB_BIT_PATTERN_31; }

}  // extern "C"
