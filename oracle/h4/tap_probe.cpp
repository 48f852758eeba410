// tap_probe.cpp -- TEST INFRASTRUCTURE (oracle/): the descriptor loop of the reference's computeOrbDescriptor
// (src/ORBextractor.cc:94-139) restated with the SAME expression shapes -- int pattern coordinates times float a / b through a
// GET_VALUE macro, cvRound as OpenCV's x86-64 inline (fast_math.hpp: _mm_cvtss_si32) -- so that it can be compiled with the
// reference's own flags (CMakeLists.txt:10-17: -O3 -march=native, -std=c++20, g++'s default -ffp-contract=fast) and with
// baseline flags, and the oracle's two tap-contraction modes can be pinned to what the compiler really emits
// (tests/test_tap_contraction.py).  Never linked into the product.
#include <emmintrin.h>

#include <cmath>

struct Point {
  int x, y;
};
typedef unsigned char uchar;
static inline int cvRound(float value) {
  __m128 t = _mm_set_ss(value);
  return _mm_cvtss_si32(t);
}
const float factorPI = (float)(3.14159265358979323846 / 180.f);

extern "C" void tap_probe_descriptor(float kpt_angle, const uchar* center, int step, const Point* pattern, uchar* desc) {
  float angle = (float)kpt_angle * factorPI;
  float a = (float)std::cos(angle), b = (float)std::sin(angle);
#define GET_VALUE(idx) center[cvRound(pattern[idx].x * b + pattern[idx].y * a) * step + cvRound(pattern[idx].x * a - pattern[idx].y * b)]
  for (int i = 0; i < 32; ++i, pattern += 16) {
    int t0, t1, val;
    t0  = GET_VALUE(0);
    t1  = GET_VALUE(1);
    val = t0 < t1;
    t0  = GET_VALUE(2);
    t1  = GET_VALUE(3);
    val |= (t0 < t1) << 1;
    t0 = GET_VALUE(4);
    t1 = GET_VALUE(5);
    val |= (t0 < t1) << 2;
    t0 = GET_VALUE(6);
    t1 = GET_VALUE(7);
    val |= (t0 < t1) << 3;
    t0 = GET_VALUE(8);
    t1 = GET_VALUE(9);
    val |= (t0 < t1) << 4;
    t0 = GET_VALUE(10);
    t1 = GET_VALUE(11);
    val |= (t0 < t1) << 5;
    t0 = GET_VALUE(12);
    t1 = GET_VALUE(13);
    val |= (t0 < t1) << 6;
    t0 = GET_VALUE(14);
    t1 = GET_VALUE(15);
    val |= (t0 < t1) << 7;
    desc[i] = (uchar)val;
  }
#undef GET_VALUE
}

// many keypoints of one image in one call (the Python loop would dominate the test otherwise)
extern "C" void tap_probe_many(const uchar* img, int step, const int* xy, const float* angles, int n, const Point* pattern, uchar* desc) {
  for (int i = 0; i < n; i++) tap_probe_descriptor(angles[i], img + xy[2 * i + 1] * step + xy[2 * i], step, pattern, desc + 32 * i);
}
