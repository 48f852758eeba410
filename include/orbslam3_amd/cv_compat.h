// cv_compat.h -- the few OpenCV types the ORBextractor / ORBmatcher seam uses, for builds without OpenCV.
// When <opencv2/core.hpp> is available the adaptors use the real cv:: types instead (same memory layout:
// cv::KeyPoint is the 28-byte record orbx_keypoint_t mirrors).
#ifndef ORBSLAM3_AMD_CV_COMPAT_H_
#define ORBSLAM3_AMD_CV_COMPAT_H_

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && !defined(ORBX_NO_OPENCV)
#define ORBX_HAVE_OPENCV 1
#endif
#endif

#ifdef ORBX_HAVE_OPENCV
#include <opencv2/core.hpp>
namespace orbx_cv = cv;
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace orbx_cv {
typedef unsigned char uchar;
struct Point2f {
  float x = 0, y = 0;
  Point2f() {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {  // layout of cv::KeyPoint
  Point2f pt;
  float   size = 0, angle = -1, response = 0;
  int     octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");
enum { CV_8U = 0, CV_8UC1 = 0 };
// Minimal single-channel 8-bit matrix with shared ownership (what cv::Mat is used for at this seam).
class Mat {
 public:
  int    rows = 0, cols = 0;
  size_t step = 0;
  uchar* data = nullptr;
  Mat() {}
  Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
  Mat(int r, int c, int /*type*/, void* ext, size_t stp = 0) : rows(r), cols(c), step(stp ? stp : (size_t)c), data((uchar*)ext) {}
  void create(int r, int c, int /*type*/) {
    if (r == rows && c == cols && own_) return;
    rows = r, cols = c, step = (size_t)c;
    own_ = std::shared_ptr<uchar>(new uchar[(size_t)r * c + 1], std::default_delete<uchar[]>());
    data = own_.get();
  }
  void         release() { own_.reset(), data = nullptr, rows = cols = 0, step = 0; }
  bool         empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int          type() const { return CV_8UC1; }
  uchar*       ptr(int r = 0) { return data + (size_t)r * step; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
  Mat          row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step).share(own_); }
  Mat          clone() const {
    Mat m(rows, cols, CV_8U);
    for (int r = 0; r < rows; r++) std::memcpy(m.ptr(r), ptr(r), (size_t)cols);
    return m;
  }

 private:
  std::shared_ptr<uchar> own_;
  Mat&                   share(const std::shared_ptr<uchar>& o) {
    own_ = o;
    return *this;
  }
};
}  // namespace orbx_cv
#endif
#endif
