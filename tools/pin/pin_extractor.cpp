// pin_extractor <cases dir>: the reference's ORBextractor (compiled from the reference tree, unchanged, against the integrator's
// OpenCV) against orbx_extract of liborbslam_hip.so on the frames make_cases.py wrote (frame_*.pgm, binary P5), stage by stage:
//   pyramid level l  : ORBextractor::mvImagePyramid[l] (cv::resize INTER_LINEAR, src/ORBextractor.cc:1107-1132)  vs  orbx_get_level_image
//   keypoints        : count, order, every cv::KeyPoint field                                                   vs  orbx_extract
//   descriptors      : 32 bytes per keypoint (cv::GaussianBlur 7x7 sigma 2 + rotated BRIEF, :1086, :107-147)
// --json <file>: a machine-readable report - per frame the FIRST stage that differs ("pyramid level l", "keypoint count", "keypoint i",
// "descriptor i") or null.  Exit code 0 iff every stage of every frame is identical.  A pyramid difference points at cv::resize's fixed-point rounding, a
// descriptor-only difference at cv::GaussianBlur's (OpenCV <= 3.4.1: 8-bit taps {18,34,49,55,49,34,18}, what the library
// implements; OpenCV >= 3.4.2 / 4.x use a different fixed-point kernel - see INTEGRATION.md, "Pinning against the reference").
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
#include "orbslam_hip.h"
#ifndef PIN_SYNTAX_ONLY
#include "ORBextractor.h"
#endif

static bool read_pgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  int maxv = 0;
  if (fscanf(f, "P5 %d %d %d", &w, &h, &maxv) != 3 || maxv != 255) { fclose(f); return false; }
  fgetc(f);
  px.resize((size_t)w * h);
  const bool ok = fread(px.data(), 1, px.size(), f) == px.size();
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: pin_extractor <cases dir> [nfeatures]\n"); return 2; }
  const int nfeatures = (argc > 2 && argv[2][0] != '-') ? atoi(argv[2]) : 2000;
  const char* json = nullptr;
  for (int i = 2; i + 1 < argc; i++) if (!strcmp(argv[i], "--json")) json = argv[i + 1];
  std::string report = "[";
  int bad_frames = 0, frames = 0;
  for (int k = 0;; k++) {
    char name[64]; snprintf(name, sizeof(name), "/frame_%03d.pgm", k);
    std::vector<unsigned char> px; int w = 0, h = 0;
    if (!read_pgm(std::string(argv[1]) + name, px, w, h)) break;
    frames++;
    // ---- this library
    orbx_ctx* ctx = nullptr;
    if (orbx_create(nfeatures, 1.2f, 8, 20, 7, 0, &ctx)) { fprintf(stderr, "orbx_create: %s\n", orbhip_last_error()); return 3; }
    const int cap = orbx_max_keypoints(ctx);
    std::vector<orbx_keypoint> kps(cap); std::vector<unsigned char> desc((size_t)cap * 32); int n = 0;
    if (orbx_extract(ctx, px.data(), w, h, w, kps.data(), desc.data(), cap, &n)) { fprintf(stderr, "orbx_extract: %s\n", orbhip_last_error()); return 3; }
    int diff_pyr = 0, diff_kp = 0, diff_desc = 0, n_ref = -1;
    std::string first;                                                  // the first stage that differs, in pipeline order
#ifndef PIN_SYNTAX_ONLY
    // ---- the reference
    ORB_SLAM2::ORBextractor ref(nfeatures, 1.2f, 8, 20, 7);
    cv::Mat image(h, w, CV_8UC1, px.data()), rdesc;
    std::vector<cv::KeyPoint> rk;
    ref(image, cv::Mat(), rk, rdesc);
    n_ref = (int)rk.size();
    for (int l = 0; l < 8; l++) {
      int lw = 0, lh = 0;
      orbx_get_level_image(ctx, 0, l, 0, nullptr, &lw, &lh);
      std::vector<unsigned char> lv((size_t)lw * lh);
      orbx_get_level_image(ctx, 0, l, 0, lv.data(), &lw, &lh);
      const cv::Mat& m = ref.mvImagePyramid[l];
      if (m.cols != lw || m.rows != lh) { diff_pyr += lw * lh; if (first.empty()) first = "pyramid level " + std::to_string(l) + " (size)"; continue; }
      int d = 0;
      for (int y = 0; y < lh; y++) d += memcmp(m.ptr<unsigned char>(y), lv.data() + (size_t)y * lw, lw) != 0 ? 1 : 0;
      if (d) printf("  frame %d level %d: %d of %d rows differ\n", k, l, d, lh);
      if (d && first.empty()) first = "pyramid level " + std::to_string(l);
      diff_pyr += d;
    }
    if (n_ref != n) { diff_kp = abs(n_ref - n) + 1; if (first.empty()) first = "keypoint count"; }
    for (int i = 0; i < n && i < n_ref; i++) {
      const cv::KeyPoint& a = rk[i]; const orbx_keypoint& b = kps[i];
      if (a.pt.x != b.x || a.pt.y != b.y || a.size != b.size || a.angle != b.angle || a.response != b.response || a.octave != b.octave) { diff_kp++; if (first.empty()) first = "keypoint " + std::to_string(i); }
      else if (memcmp(rdesc.ptr<unsigned char>(i), desc.data() + (size_t)i * 32, 32) != 0) { diff_desc++; if (first.empty()) first = "descriptor " + std::to_string(i); }
    }
#endif
    printf("frame %d (%d x %d): %d keypoints here, %d in the reference; pyramid rows differing %d, keypoints differing %d, descriptors differing %d\n",
           k, w, h, n, n_ref, diff_pyr, diff_kp, diff_desc);
    bad_frames += (diff_pyr || diff_kp || diff_desc) ? 1 : 0;
    report += std::string(frames > 1 ? "," : "") + "{\"frame\":" + std::to_string(k) + ",\"w\":" + std::to_string(w) + ",\"h\":" + std::to_string(h) + ",\"keypoints\":" + std::to_string(n) +
              ",\"keypoints_reference\":" + std::to_string(n_ref) + ",\"pyramid_rows_differing\":" + std::to_string(diff_pyr) + ",\"keypoints_differing\":" + std::to_string(diff_kp) +
              ",\"descriptors_differing\":" + std::to_string(diff_desc) + ",\"first_difference\":" + (first.empty() ? std::string("null") : "\"" + first + "\"") + "}";
    orbx_destroy(ctx);
  }
  if (!frames) { fprintf(stderr, "no frame_000.pgm under %s (python tools/pin/make_cases.py <dir>)\n", argv[1]); return 2; }
  printf("%s: %d of %d frames identical\n", bad_frames ? "DIFFERENT" : "PINNED", frames - bad_frames, frames);
  if (json) { FILE* f = fopen(json, "w"); if (f) { fprintf(f, "{\"frames\":%d,\"identical\":%d,\"per_frame\":%s]}\n", frames, frames - bad_frames, report.c_str()); fclose(f); } }
  return bad_frames ? 1 : 0;
}
