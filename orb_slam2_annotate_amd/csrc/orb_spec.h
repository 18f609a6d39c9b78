// orb_spec.h -- the frozen arithmetic of the ORB front-end, shared by host and device code
// of the PRODUCT (the CPU oracle under oracle/ restates the same spec independently in C).
//
// Everything here must evaluate identically on the host (clang x86-64) and on gfx950:
// only IEEE-754 +,-,*,/ and conversions, no fused multiply-add (build with
// -ffp-contract=off), no libm transcendental.  SURVEY.md section 8(c) "canonical spec".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ORB_HD __host__ __device__ __forceinline__

namespace orbfe {

constexpr int kPatchSize = 31;      // src/ORBextractor.cc:71
constexpr int kHalfPatch = 15;      // :72
constexpr int kEdgeThreshold = 19;  // :73
constexpr int kMinBorder = kEdgeThreshold - 3;  // 16, :823
// cv::GaussianBlur(7x7, sigma 2) arithmetic variants (k_blur.hip; include/orbfe.h ORBFE_BLUR_*)
constexpr int kBlurSpecCv4 = 0;        // OpenCV >= 3.4.1 / 4.x bit-exact fixed point, taps 18 34 48 56
constexpr int kBlurSpecCv2Scalar = 1;  // OpenCV 2.4.x / 3.0-3.3 generic C++ path, taps 18 34 49 55, round half up
constexpr int kBlurSpecCv2Sse2 = 2;    // the same with the SSE2 column pass: round half to even on the first w & ~3 columns

// cvRound(float): round half to even (SSE cvtss2si semantics), used at :82,118,124-125.
ORB_HD int cv_round(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float2int_rn(v);
#else
  return (int)__builtin_nearbyintf(v);
#endif
}

// cv::fastAtan2 (degrees, [0,360)), called by IC_Angle at src/ORBextractor.cc:105.
ORB_HD float fast_atan2(float y, float x) {
  const float k = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * k;
  const float p3 = -0.3258083974640975f * k;
  const float p5 = 0.1555786518463281f * k;
  const float p7 = -0.04432655554792128f * k;
  const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
  float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
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

// cos/sin of a float radian angle, replacing libm cosf/sinf at src/ORBextractor.cc:116.
// Double-precision Cody-Waite reduction + minimax kernels, one final rounding to float.
ORB_HD void sincos_spec(float rad, float* c, float* s) {
  const double x = (double)rad;
#if defined(__HIP_DEVICE_COMPILE__)
  const double k = rint(x * 6.36619772367581382433e-01);
#else
  const double k = __builtin_nearbyint(x * 6.36619772367581382433e-01);
#endif
  double r = x - k * 1.57079632673412561417e+00;
  r = r - k * 6.07710050650619224932e-11;
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = -2.50507602534068634195e-08 + z * ps;
  ps = 2.75573137070700676789e-06 + z * ps;
  ps = -1.98412698298579493134e-04 + z * ps;
  ps = 8.33333333332248946124e-03 + z * ps;
  ps = -1.66666666666666324348e-01 + z * ps;
  const double sr = r + (r * z) * ps;
  double pc = -1.13596475577881948265e-11;
  pc = 2.08757232129817482790e-09 + z * pc;
  pc = -2.75573143513906633035e-07 + z * pc;
  pc = 2.48015872894767294178e-05 + z * pc;
  pc = -1.38888888888741095749e-03 + z * pc;
  pc = 4.16666666666666019037e-02 + z * pc;
  const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
  const int n = (int)((long long)k & 3);
  double cs, sn;
  if (n == 0) { cs = cr; sn = sr; }
  else if (n == 1) { cs = -sr; sn = cr; }
  else if (n == 2) { cs = -cr; sn = -sr; }
  else { cs = sr; sn = -cr; }
  *c = (float)cs;
  *s = (float)sn;
}

// 256-bit Hamming distance, ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1828-1844).
ORB_HD int hamming256(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __builtin_popcount(a[i] ^ b[i]);
  return d;
}

}  // namespace orbfe
