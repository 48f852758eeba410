// undistort_core.h -- cv::undistortPoints(src, dst, K, distCoeffs, noArray(), P) for one point, as Frame::UndistortKeyPoints and
// Frame::ComputeImageBounds call it (reference src/Frame.cc:886-943): OpenCV >= 4.5 cvUndistortPointsInternal
// (modules/calib3d/src/undistort.dispatch.cpp) with the default TermCriteria(MAX_ITER, 5, 0.01) -- five fixed-point iterations
// in double precision, no tilt terms, R = identity, new projection P = K.  Shared by the device kernel and the host entry
// point; every operation is an IEEE double operation in the order OpenCV performs it (compile with -ffp-contract=off).
#ifndef ORBX_UNDISTORT_CORE_H_
#define ORBX_UNDISTORT_CORE_H_

#if defined(__HIPCC__)
#define UD_HD __host__ __device__
#else
#define UD_HD
#endif

namespace orbx {

struct UndistortParams {
  double fx, fy, cx, cy;  // camera matrix K (converted from the float cv::Mat, cvConvert)
  double k[12];           // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 (ORB-SLAM3 supplies 4 or 5; the rest are 0)
  double pfx, pfy, pcx, pcy;  // new camera matrix P (= mK)
};

UD_HD inline void undistort_point(const UndistortParams& p, float xin, float yin, float* xout, float* yout) {
  const double ifx = 1. / p.fx, ify = 1. / p.fy;
  double       x = xin, y = yin;
  const double u = x, v = y;
  x = (x - p.cx) * ifx;
  y = (y - p.cy) * ify;
  // tilt compensation with the identity matrix: vecUntilt = (x, y, 1), invProj = 1. / 1
  const double invProj = 1. / 1.;
  const double x0 = x = invProj * x;
  const double y0 = y = invProj * y;
  for (int j = 0; j < 5; j++) {  // criteria.type == COUNT, maxCount == 5
    const double r2     = x * x + y * y;
    const double icdist = (1 + ((p.k[7] * r2 + p.k[6]) * r2 + p.k[5]) * r2) / (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2);
    if (icdist < 0) {  // undistortPoints.regression_14583
      x = (u - p.cx) * ifx;
      y = (v - p.cy) * ify;
      break;
    }
    const double deltaX = 2 * p.k[2] * x * y + p.k[3] * (r2 + 2 * x * x) + p.k[8] * r2 + p.k[9] * r2 * r2;
    const double deltaY = p.k[2] * (r2 + 2 * y * y) + 2 * p.k[3] * x * y + p.k[10] * r2 + p.k[11] * r2 * r2;
    x                   = (x0 - deltaX) * icdist;
    y                   = (y0 - deltaY) * icdist;
  }
  // RR = P * I: rows (pfx, 0, pcx), (0, pfy, pcy), (0, 0, 1)
  const double xx = p.pfx * x + 0. * y + p.pcx;
  const double yy = 0. * x + p.pfy * y + p.pcy;
  const double ww = 1. / (0. * x + 0. * y + 1.);
  *xout           = (float)(xx * ww);
  *yout           = (float)(yy * ww);
}

}  // namespace orbx
#endif
