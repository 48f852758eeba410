// orb_sincosf_port.h -- bit-exact port of glibc's float sine/cosine for the descriptor steering step.
//
// The reference evaluates `a = cos(angle), b = sin(angle)` with the host libm on a float argument
// (src/ORBextractor.cc:97-98), and the rotated tap coordinates are rounded to integers (:105-106),
// so the device must reproduce libm's results to the last bit.  glibc (>= 2.28, this image: 2.35)
// implements sinf/cosf as an argument reduction plus a polynomial, all in double precision, and rounds
// once to float (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c).  On x86-64
// an ifunc picks between a plain SSE2 build and a build compiled with -mfma in which every a + b*c
// is contracted into one fused multiply-add, so both evaluation orders are provided; the host probes
// which one its libm uses (orbx_create) and the kernels take that as a flag.
//
// The port covers the domain the extractor produces: x = deg * (float)(pi/180), deg in [0, 360],
// i.e. 0 <= x < 120 (glibc's "fast" reduction range).  It is checked exhaustively against the host
// libm in tests/test_sincos_port.py (every float deg in [0, 360]).
//
// Compile with -ffp-contract=off: the only fused operations must be the explicit ones below.
#ifndef ORB_SINCOSF_PORT_H_
#define ORB_SINCOSF_PORT_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORB_HD __host__ __device__ __forceinline__
#else
#define ORB_HD static inline
#endif

namespace orbport {

struct SinCosTab {
  double c0, c1, c2, c3, c4;  // cosine polynomial
  double s1, s2, s3;          // sine polynomial
};

template <bool FMA>
ORB_HD double mad(double a, double b, double c) {  // a*b + c
  if (FMA) return __builtin_fma(a, b, c);
  double p = a * b;
  return p + c;
}

// sinf_poly of glibc's s_sincosf.h: n even -> sine polynomial, n odd -> cosine polynomial.
template <bool FMA>
ORB_HD float sin_poly(double x, double x2, bool neg_cos, int n) {
  const double sgn = neg_cos ? -1.0 : 1.0;
  const double C0 = sgn * 0x1p0, C1 = sgn * -0x1.ffffffd0c621cp-2, C2 = sgn * 0x1.55553e1068f19p-5,
               C3 = sgn * -0x1.6c087e89a359dp-10, C4 = sgn * 0x1.99343027bf8c3p-16;
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = mad<FMA>(x2, S3, S2);
    double x7 = x3 * x2;
    double s  = mad<FMA>(x3, S1, x);
    return (float)mad<FMA>(x7, s1, s);
  } else {
    double x4 = x2 * x2;
    double c2 = mad<FMA>(x2, C4, C3);
    double c1 = mad<FMA>(x2, C1, C0);
    double x6 = x4 * x2;
    double c  = mad<FMA>(x4, C2, c1);
    return (float)mad<FMA>(x6, c2, c);
  }
}

ORB_HD uint32_t f2u(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  return v.u;
}

// Valid for 0 <= y < 120.  which = 0 -> sinf(y), 1 -> cosf(y).
template <bool FMA>
ORB_HD float sincos_one(float y, int which) {
  const double HPI_INV = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  const double HPI     = 0x1.921FB54442D18p0;
  double   x   = (double)y;
  uint32_t top = (f2u(y) >> 20) & 0x7ff;
  if (top < 0x3f4u) {  // abstop12(y) < abstop12(pi/4)
    double x2 = x * x;
    if (top < 0x398u) {  // |y| < 2^-12
      return which ? 1.0f : y;
    }
    return sin_poly<FMA>(x, x2, false, which);
  }
  double  r = x * HPI_INV;
  int32_t n = ((int32_t)r + 0x800000) >> 24;
  // x - n*hpi  (FMA build: fnmadd)
  double xr = FMA ? __builtin_fma(-(double)n, HPI, x) : (x - (double)n * HPI);
  double s = (((n + 1) & 2) != 0) ? -1.0 : 1.0;  // sign table {1,-1,-1,1}[n & 3]
  return sin_poly<FMA>(xr * s, xr * xr, (n & 2) != 0, which ? (n ^ 1) : n);
}

template <bool FMA>
ORB_HD void sincosf_port(float y, float* sn, float* cs) {
  *sn = sincos_one<FMA>(y, 0);
  *cs = sincos_one<FMA>(y, 1);
}

}  // namespace orbport
#endif
