// TEST INFRASTRUCTURE - not OpenCV.  The handful of cv:: declarations that csrc/compat/orbslam_compat.h's
// `#ifdef ORBCOMPAT_HAVE_OPENCV` branch touches, with OpenCV's public signatures (cv::Mat::create(rows, cols, type), type(), data,
// step, cv::_InputArray::getMat() / empty(), cv::_OutputArray::create() / release() / getMat(), cv::KeyPoint's public fields,
// CV_Assert, CV_8U / CV_8UC1), so that the branch a reference maintainer would build against the real library is compiled and run
// by the test suite at all (tests/test_gpu_compat_cpp.py::test_opencv_signature_branch).  Headers share their pixels like cv::Mat.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

namespace cv {
typedef unsigned char uchar;
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
class KeyPoint {
 public:
  KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  Point2f pt; float size, angle, response; int octave, class_id;
};
class Mat {
 public:
  struct MatStep { size_t v = 0; operator size_t() const { return v; } MatStep& operator=(size_t s) { v = s; return *this; } };
  Mat() {}
  Mat(int r, int c, int type, void* d, size_t s = 0) : rows(r), cols(c), data((uchar*)d), type_(type) { step = s ? s : (size_t)c; }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;
    store_ = std::make_shared<std::vector<uchar>>((size_t)r * c);
    rows = r; cols = c; type_ = type; step = (size_t)c; data = store_->data();
  }
  void release() { rows = cols = 0; data = nullptr; step = 0; store_.reset(); }
  bool empty() const { return data == nullptr || rows * cols == 0; }
  int type() const { return type_; }
  template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  MatStep step;
 private:
  int type_ = CV_8UC1;
  std::shared_ptr<std::vector<uchar>> store_;
};
class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 protected:
  Mat* m_ = nullptr;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) { m_ = &m; }
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline InputArray noArray() { static _InputArray none; return none; }
}  // namespace cv
