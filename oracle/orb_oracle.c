/*
 * orb_oracle.c -- CPU ORACLE (test infrastructure, see orb_oracle.h header note).
 * PARITY UNPINNED at the OpenCV boundary; constants pinned by known-answer tests.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference root).  Build with -ffp-contract=off:
 * float expressions must not be FMA-contracted (hard part 2).
 */
#include "orb_oracle.h"
#include "orb_pattern_data.h"

#include <float.h>
#include <limits.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* arithmetic primitives                                               */
/* ------------------------------------------------------------------ */

/* cvRound: SSE cvtsd2si / lrint semantics = round half to even (default FP env). */
int orc_cvround(double v) { return (int)nearbyint(v); }

static int cvfloor_f(float v) { return (int)floorf(v); }

/* cv::fastAtan2 polynomial (OpenCV 2.4.9+/3.x scalar path), called at
 * src/ORBextractor.cc:105.  All fp32, no FMA. */
float orc_fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* Project-owned sincos replacing libm cosf/sinf at src/ORBextractor.cc:116.
 * Double Cody-Waite reduction by pi/2 + fdlibm-degree kernels, rounded once
 * to float: equals the correctly rounded result except when the double value
 * sits within ~1e-16 of a float rounding tie.  Only IEEE +,-,* on doubles, so
 * the HIP device code reproduces it bit for bit.  Valid for |rad| < 1e4. */
void orc_sincos(float rad, float *c, float *s) {
  const double x = (double)rad;
  const double k = nearbyint(x * 6.36619772367581382433e-01); /* 2/pi */
  double r = x - k * 1.57079632673412561417e+00;             /* pi/2 hi (33 bits) */
  r = r - k * 6.07710050650619224932e-11;                     /* pi/2 lo */
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
  switch (n) {
    case 0: cs = cr; sn = sr; break;
    case 1: cs = -sr; sn = cr; break;
    case 2: cs = -cr; sn = -sr; break;
    default: cs = sr; sn = -cr; break;
  }
  *c = (float)cs;
  *s = (float)sn;
}

const signed char *orc_pattern(void) { return ORC_BIT_PATTERN_31; }

/* ------------------------------------------------------------------ */
/* constructor tables: src/ORBextractor.cc:415-486                     */
/* ------------------------------------------------------------------ */
void orc_extractor_init(orc_extractor *e, int nfeatures, float scaleFactor, int nlevels,
                        int iniThFAST, int minThFAST) {
  memset(e, 0, sizeof(*e));
  if (nlevels > ORC_MAX_LEVELS) nlevels = ORC_MAX_LEVELS;
  e->nfeatures = nfeatures;
  e->scaleFactor = (double)scaleFactor;
  e->nlevels = nlevels;
  e->iniThFAST = iniThFAST;
  e->minThFAST = minThFAST;
  e->mvScaleFactor[0] = 1.0f;
  e->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) { /* :424-428, float*double -> float */
    e->mvScaleFactor[i] = (float)((double)e->mvScaleFactor[i - 1] * e->scaleFactor);
    e->mvLevelSigma2[i] = e->mvScaleFactor[i] * e->mvScaleFactor[i];
  }
  for (int i = 0; i < nlevels; i++) { /* :432-436 */
    e->mvInvScaleFactor[i] = 1.0f / e->mvScaleFactor[i];
    e->mvInvLevelSigma2[i] = 1.0f / e->mvLevelSigma2[i];
  }
  /* :448-458 */
  float factor = (float)(1.0 / e->scaleFactor); /* 1.0f / double */
  float nDesired = (float)nfeatures * (1 - factor) /
                   (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    e->mnFeaturesPerLevel[level] = orc_cvround(nDesired);
    sum += e->mnFeaturesPerLevel[level];
    nDesired *= factor;
  }
  e->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  /* umax, :471-485 */
  const int HP = 15;
  int v, v0;
  int vmax = cvfloor_f(HP * sqrtf(2.f) / 2 + 1);
  int vmin = (int)ceilf(HP * sqrtf(2.f) / 2);
  const double hp2 = HP * HP;
  for (v = 0; v <= vmax; ++v) e->umax[v] = orc_cvround(sqrt(hp2 - v * v));
  for (v = HP, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
}

/* src/ORBextractor.cc:1207-1208 */
void orc_level_size(const orc_extractor *e, int W, int H, int level, int *w, int *h) {
  float scale = e->mvInvScaleFactor[level];
  *w = orc_cvround((float)W * scale);
  *h = orc_cvround((float)H * scale);
}

/* ------------------------------------------------------------------ */
/* cv::resize INTER_LINEAR, 8UC1 (OpenCV imgproc fixed-point path)      */
/* called at src/ORBextractor.cc:1219                                  */
/* ------------------------------------------------------------------ */
static short sat_short_round(float v) {
  int i = orc_cvround(v);
  if (i > 32767) i = 32767;
  if (i < -32768) i = -32768;
  return (short)i;
}

void orc_resize_linear(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                       int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int *xofs = (int *)malloc(sizeof(int) * dw);
  short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
  int *rows[2];
  rows[0] = (int *)malloc(sizeof(int) * dw);
  rows[1] = (int *)malloc(sizeof(int) * dw);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvfloor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = sat_short_round(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvfloor_f(fy);
    fy -= sy;
    short b0 = sat_short_round((1.f - fy) * 2048);
    short b1 = sat_short_round(fy * 2048);
    for (int k = 0; k < 2; k++) { /* rows clamped, weights are not */
      int r = sy + k;
      if (r < 0) r = 0;
      if (r >= sh) r = sh - 1;
      const uint8_t *S = src + (size_t)r * sstride;
      for (int dx = 0; dx < dw; dx++) {
        int sx = xofs[dx];
        int s1 = sx + 1 < sw ? S[sx + 1] : S[sx];
        rows[k][dx] = S[sx] * ialpha[2 * dx] + s1 * ialpha[2 * dx + 1];
      }
    }
    uint8_t *D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs);
  free(ialpha);
  free(rows[0]);
  free(rows[1]);
}

/* ------------------------------------------------------------------ */
/* cv::GaussianBlur 7x7 sigma=2 reflect-101, 8U fixed-point            */
/* (OpenCV >= 4 bit-exact path; kernel by error diffusion, sums to 256) */
/* called at src/ORBextractor.cc:1175                                  */
/* ------------------------------------------------------------------ */
static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * (n - 1) - i;
  }
  return i;
}

/* cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1 (src/ORBextractor.cc:1175), in the three
 * arithmetic variants the unpinned OpenCV version allows (see include/orbfe.h ORBFE_BLUR_*; all restated from OpenCV's
 * published sources, none verifiable here):
 *   spec 0  OpenCV >= 3.4.1 / 4.x fixed-point path (smooth.cpp, ufixedpoint16): bit-exact kernel 18 34 48 56 48 34 18,
 *           which sums to 256, first pass exact in 8.8, (x + 2^15) >> 16 after the second;
 *   spec 1  OpenCV 2.4.x / 3.0-3.3, filter.cpp without SIMD: getGaussianKernel(7, 2) in float converted tap by tap
 *           with convertTo(CV_32S, 256) -> 18 34 49 55 49 34 18 (sum 257), column pass FixedPtCastEx<int, uchar>(16):
 *           saturate_cast<uchar>((x + 2^15) >> 16);
 *   spec 2  the same versions with SSE2 (SymmColumnVec_32s8u): for x < width - width % 4 the column pass converts the
 *           row sums to float, multiplies by tap / 2^16, accumulates in float (centre tap first, then the symmetric
 *           pairs outward) and converts with cvtps2dq = round-half-to-EVEN, then packs with unsigned saturation; the
 *           remaining width % 4 columns go through the scalar spec-1 code. */
void orc_gaussian_blur7_spec(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int spec) {
  static const int K0[7] = {18, 34, 48, 56, 48, 34, 18};
  static const int K1[7] = {18, 34, 49, 55, 49, 34, 18};
  const int *K = spec == 0 ? K0 : K1;
  uint16_t *hs = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
  uint8_t *row = (uint8_t *)malloc((size_t)w + 6);
  for (int y = 0; y < h; y++) { /* horizontal pass into 8.8 fixed point (<= 255 * 257 = 65535) */
    const uint8_t *S = src + (size_t)y * sstride;
    memcpy(row + 3, S, w);
    for (int i = 1; i <= 3; i++) { row[3 - i] = S[reflect101(-i, w)]; row[3 + w - 1 + i] = S[reflect101(w - 1 + i, w)]; }
    uint16_t *H = hs + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      const uint8_t *r = row + x;
      H[x] = (uint16_t)(K[0] * (r[0] + r[6]) + K[1] * (r[1] + r[5]) + K[2] * (r[2] + r[4]) + K[3] * r[3]);
    }
  }
  const int simd_cols = spec == 2 ? w - (w % 4) : 0;
  const float kf[4] = {(float)K[3] * (1.f / 65536.f), (float)K[2] * (1.f / 65536.f), (float)K[1] * (1.f / 65536.f),
                       (float)K[0] * (1.f / 65536.f)}; /* kernel.convertTo(CV_32F, 1./(1 << 16)): centre, +-1, +-2, +-3 */
  for (int y = 0; y < h; y++) { /* vertical pass, 16.16 -> u8 */
    const uint16_t *r[7];
    for (int j = 0; j < 7; j++) r[j] = hs + (size_t)reflect101(y + j - 3, h) * w;
    uint8_t *D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      if (x < simd_cols) { /* float column pass, no FMA (-ffp-contract=off), default rounding mode */
        float s0 = (float)(int)r[3][x] * kf[0] + 0.f;
        for (int k = 1; k <= 3; k++) s0 = s0 + (float)((int)r[3 + k][x] + (int)r[3 - k][x]) * kf[k];
        long v = lrintf(s0); /* cvtps2dq */
        D[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); /* packs + packus */
        continue;
      }
      uint32_t acc = (uint32_t)K[0] * ((uint32_t)r[0][x] + r[6][x]) + (uint32_t)K[1] * ((uint32_t)r[1][x] + r[5][x]) +
                     (uint32_t)K[2] * ((uint32_t)r[2][x] + r[4][x]) + (uint32_t)K[3] * r[3][x];
      const uint32_t v = (acc + (1u << 15)) >> 16;
      D[x] = (uint8_t)(v > 255 ? 255 : v); /* saturate_cast: only reachable with the 257-sum taps */
    }
  }
  free(row);
  free(hs);
}
void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
  orc_gaussian_blur7_spec(src, w, h, sstride, dst, dstride, 0);
}

/* ------------------------------------------------------------------ */
/* cv::FAST (FAST-9/16) with non-max suppression                       */
/* called at src/ORBextractor.cc:874,880                               */
/* ------------------------------------------------------------------ */
static const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                   {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                   {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

static int fast_is_corner(const uint8_t *p, int stride, int t) {
  int v = p[0];
  int dark[16], bright[16];
  for (int k = 0; k < 16; k++) {
    int x = p[kCircle[k][0] + kCircle[k][1] * stride];
    dark[k] = x < v - t;
    bright[k] = x > v + t;
  }
  for (int s = 0; s < 16; s++) {
    int cd = 0, cb = 0;
    for (int j = 0; j < 9; j++) {
      cd += dark[(s + j) & 15];
      cb += bright[(s + j) & 15];
    }
    if (cd == 9 || cb == 9) return 1;
  }
  return 0;
}

/* cornerScore<16>: max(thr, max_arc min(d), max_arc min(-d)) - 1 */
int orc_fast_score_pixel(const uint8_t *p, int stride, int threshold) {
  int v = p[0], d[16];
  for (int k = 0; k < 16; k++) d[k] = v - p[kCircle[k][0] + kCircle[k][1] * stride];
  int a0 = threshold;
  for (int s = 0; s < 16; s++) {
    int a = INT_MAX;
    for (int j = 0; j < 9; j++) if (d[(s + j) & 15] < a) a = d[(s + j) & 15];
    if (a > a0) a0 = a;
  }
  int b0 = -a0;
  for (int s = 0; s < 16; s++) {
    int b = INT_MIN;
    for (int j = 0; j < 9; j++) if (d[(s + j) & 15] > b) b = d[(s + j) & 15];
    if (b < b0) b0 = b;
  }
  return -b0 - 1;
}

/* Brute-force statement of cv::FAST(...,nonmax=true): kept as the readable spec and
 * as the cross-check of the optimized orc_fast_nms below (tests/test_oracle_*.py). */
int orc_fast_nms_bruteforce(const uint8_t *img, int w, int h, int stride, int threshold, int *xs,
                            int *ys, int *scores, int cap) {
  if (w < 7 || h < 7) return 0;
  if (threshold < 0) threshold = 0;
  if (threshold > 255) threshold = 255;
  uint8_t *sc = (uint8_t *)calloc((size_t)w * h, 1);
  uint8_t *is = (uint8_t *)calloc((size_t)w * h, 1);
  for (int i = 3; i < h - 3; i++)
    for (int j = 3; j < w - 3; j++) {
      const uint8_t *p = img + (size_t)i * stride + j;
      if (fast_is_corner(p, stride, threshold)) {
        is[(size_t)i * w + j] = 1;
        sc[(size_t)i * w + j] = (uint8_t)orc_fast_score_pixel(p, stride, threshold);
      }
    }
  int n = 0;
  for (int i = 3; i < h - 3; i++)
    for (int j = 3; j < w - 3; j++) {
      if (!is[(size_t)i * w + j]) continue;
      int s = sc[(size_t)i * w + j];
      const uint8_t *r0 = sc + (size_t)(i - 1) * w + j, *r1 = sc + (size_t)i * w + j,
                    *r2 = sc + (size_t)(i + 1) * w + j;
      if (s > r1[-1] && s > r1[1] && s > r0[-1] && s > r0[0] && s > r0[1] && s > r2[-1] &&
          s > r2[0] && s > r2[1]) {
        if (n < cap) { xs[n] = j; ys[n] = i; scores[n] = s; }
        n++;
      }
    }
  free(sc);
  free(is);
  return n;
}

/* cornerScore<16> with OpenCV's early-outs (same value as orc_fast_score_pixel). */
static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[25];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    if (d[k + 3] < a) a = d[k + 3];
    if (a <= a0) continue;
    for (int j = 4; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
    int t = a < d[k] ? a : d[k];
    if (t > a0) a0 = t;
    t = a < d[k + 9] ? a : d[k + 9];
    if (t > a0) a0 = t;
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    for (int j = 3; j <= 5; j++) if (d[k + j] > b) b = d[k + j];
    if (b >= b0) continue;
    for (int j = 6; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
    int t = b > d[k] ? b : d[k];
    if (t < b0) b0 = t;
    t = b > d[k + 9] ? b : d[k + 9];
    if (t < b0) b0 = t;
  }
  return -b0 - 1;
}

/* FAST_t<16> as OpenCV runs it: threshold table quick-reject, 9-of-16 arc test,
 * 3 rolling score rows for the 3x3 non-max suppression, raster emission. */
int orc_fast_nms(const uint8_t *img, int w, int h, int stride, int threshold, int *xs, int *ys,
                 int *scores, int cap) {
  if (w < 7 || h < 7) return 0;
  const int K = 8, N = 16 + K + 1;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  if (threshold < 0) threshold = 0;
  if (threshold > 255) threshold = 255;
  uint8_t tab[512];
  for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  uint8_t *bufmem = (uint8_t *)calloc((size_t)w * 3, 1);
  int *cpmem = (int *)malloc(sizeof(int) * (size_t)(w + 1) * 3);
  uint8_t *buf[3] = {bufmem, bufmem + w, bufmem + 2 * w};
  int *cpbuf[3] = {cpmem + 1, cpmem + 1 + (w + 1), cpmem + 1 + 2 * (w + 1)};
  int n = 0;
  for (int i = 3; i < h - 2; i++) {
    const uint8_t *ptr = img + (size_t)i * stride + 3;
    uint8_t *curr = buf[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, w);
    int ncorners = 0;
    if (i < h - 3) {
      for (int j = 3; j < w - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t *t = &tab[0] - v + 255;
        int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
        d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
        d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
        d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
        d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
        d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold); break; }
            } else count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold); break; }
            } else count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = buf[(i - 4 + 3) % 3];
    const uint8_t *pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
          score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
        if (n < cap) { xs[n] = j; ys[n] = i - 1; scores[n] = score; }
        n++;
      }
    }
  }
  free(bufmem);
  free(cpmem);
  return n;
}

/* ------------------------------------------------------------------ */
/* grid stage: src/ORBextractor.cc:815-896                             */
/* ------------------------------------------------------------------ */
int orc_grid_candidates(const orc_extractor *e, const uint8_t *img, int cols, int rows, int stride,
                        float *xs, float *ys, float *resp, int cap) {
  const float W = 30;
  const int minBorderX = 19 - 3, minBorderY = minBorderX;
  const int maxBorderX = cols - 19 + 3, maxBorderY = rows - 19 + 3;
  const float width = (float)(maxBorderX - minBorderX);
  const float height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W);
  const int nRows = (int)(height / W);
  if (nCols <= 0 || nRows <= 0) return 0; /* reference would divide by zero */
  const int wCell = (int)ceilf(width / nCols);
  const int hCell = (int)ceilf(height / nRows);
  int tmpcap = (wCell + 6) * (hCell + 6);
  int *tx = (int *)malloc(sizeof(int) * tmpcap * 3), *ty = tx + tmpcap, *ts = ty + tmpcap;
  int n = 0;
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
      const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
      const uint8_t *sub = img + (size_t)y0 * stride + x0;
      int c = orc_fast_nms(sub, x1 - x0, y1 - y0, stride, e->iniThFAST, tx, ty, ts, tmpcap);
      if (c == 0) c = orc_fast_nms(sub, x1 - x0, y1 - y0, stride, e->minThFAST, tx, ty, ts, tmpcap);
      for (int k = 0; k < c; k++) {
        if (n < cap) {
          xs[n] = (float)tx[k] + (float)(j * wCell);
          ys[n] = (float)ty[k] + (float)(i * hCell);
          resp[n] = (float)ts[k];
        }
        n++;
      }
    }
  }
  free(tx);
  return n;
}

/* ------------------------------------------------------------------ */
/* DistributeOctTree + DivideNode: src/ORBextractor.cc:498-558,566-808 */
/* std::list emulated with a node pool (index = creation order); the    */
/* reference's (count, pointer) sort key becomes (count, creation order)*/
/* -- documented deviation, SURVEY.md 7 hard part 4.                    */
/* ------------------------------------------------------------------ */
typedef struct {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  int *keys; /* indices into the input arrays */
  int nkeys;
  int noMore;
  int prev, next; /* list links, -1 = none */
} onode;

typedef struct {
  onode *pool;
  int npool, cappool;
  int head, tail, size;
} olist;

static int ol_new(olist *L) {
  if (L->npool == L->cappool) {
    L->cappool = L->cappool ? L->cappool * 2 : 256;
    L->pool = (onode *)realloc(L->pool, sizeof(onode) * L->cappool);
  }
  memset(&L->pool[L->npool], 0, sizeof(onode));
  return L->npool++;
}
static void ol_push_back(olist *L, int i) {
  L->pool[i].prev = L->tail;
  L->pool[i].next = -1;
  if (L->tail >= 0) L->pool[L->tail].next = i; else L->head = i;
  L->tail = i;
  L->size++;
}
static void ol_push_front(olist *L, int i) {
  L->pool[i].next = L->head;
  L->pool[i].prev = -1;
  if (L->head >= 0) L->pool[L->head].prev = i; else L->tail = i;
  L->head = i;
  L->size++;
}
static int ol_erase(olist *L, int i) { /* returns next */
  int p = L->pool[i].prev, n = L->pool[i].next;
  if (p >= 0) L->pool[p].next = n; else L->head = n;
  if (n >= 0) L->pool[n].prev = p; else L->tail = p;
  L->size--;
  return n;
}

/* DivideNode :498-558.  Children are allocated in the pool (c[0..3]) but not linked. */
static void divide_node(olist *L, int pi, const float *xs, const float *ys, int c[4]) {
  for (int k = 0; k < 4; k++) c[k] = ol_new(L);
  onode *P = &L->pool[pi];
  const int halfX = (int)ceilf((float)(P->URx - P->ULx) / 2);
  const int halfY = (int)ceilf((float)(P->BRy - P->ULy) / 2);
  onode *n1 = &L->pool[c[0]], *n2 = &L->pool[c[1]], *n3 = &L->pool[c[2]], *n4 = &L->pool[c[3]];
  n1->ULx = P->ULx; n1->ULy = P->ULy;
  n1->URx = P->ULx + halfX; n1->URy = P->ULy;
  n1->BLx = P->ULx; n1->BLy = P->ULy + halfY;
  n1->BRx = P->ULx + halfX; n1->BRy = P->ULy + halfY;
  n2->ULx = n1->URx; n2->ULy = n1->URy;
  n2->URx = P->URx; n2->URy = P->URy;
  n2->BLx = n1->BRx; n2->BLy = n1->BRy;
  n2->BRx = P->URx; n2->BRy = P->ULy + halfY;
  n3->ULx = n1->BLx; n3->ULy = n1->BLy;
  n3->URx = n1->BRx; n3->URy = n1->BRy;
  n3->BLx = P->BLx; n3->BLy = P->BLy;
  n3->BRx = n1->BRx; n3->BRy = P->BLy;
  n4->ULx = n3->URx; n4->ULy = n3->URy;
  n4->URx = n2->BRx; n4->URy = n2->BRy;
  n4->BLx = n3->BRx; n4->BLy = n3->BRy;
  n4->BRx = P->BRx; n4->BRy = P->BRy;
  for (int k = 0; k < 4; k++) {
    L->pool[c[k]].keys = (int *)malloc(sizeof(int) * (P->nkeys > 0 ? P->nkeys : 1));
    L->pool[c[k]].nkeys = 0;
  }
  for (int i = 0; i < P->nkeys; i++) {
    int id = P->keys[i];
    onode *t;
    if (xs[id] < (float)n1->URx) t = (ys[id] < (float)n1->BRy) ? n1 : n3;
    else if (ys[id] < (float)n1->BRy) t = n2;
    else t = n4;
    t->keys[t->nkeys++] = id;
  }
  for (int k = 0; k < 4; k++)
    if (L->pool[c[k]].nkeys == 1) L->pool[c[k]].noMore = 1;
}

typedef struct { int count, seq; } szptr;
static int szptr_cmp(const void *a, const void *b) {
  const szptr *A = (const szptr *)a, *B = (const szptr *)b;
  if (A->count != B->count) return A->count < B->count ? -1 : 1;
  return A->seq < B->seq ? -1 : (A->seq > B->seq ? 1 : 0);
}

int orc_distribute_octtree(const float *xs, const float *ys, const float *resp, int n, int minX,
                           int maxX, int minY, int maxY, int N, int *out_idx, int cap) {
  const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
  const float hX = (float)(maxX - minX) / (float)nIni;
  olist L;
  memset(&L, 0, sizeof(L));
  L.head = L.tail = -1;
  if (nIni <= 0) return 0; /* reference assumes width>height (:560) */
  int *ini = (int *)malloc(sizeof(int) * nIni);
  for (int i = 0; i < nIni; i++) {
    int ni = ol_new(&L);
    onode *nd = &L.pool[ni];
    nd->ULx = (int)(hX * (float)i); nd->ULy = 0;
    nd->URx = (int)(hX * (float)(i + 1)); nd->URy = 0;
    nd->BLx = nd->ULx; nd->BLy = maxY - minY;
    nd->BRx = nd->URx; nd->BRy = maxY - minY;
    nd->keys = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    nd->nkeys = 0;
    ol_push_back(&L, ni);
    ini[i] = ni;
  }
  for (int i = 0; i < n; i++) {
    int b = (int)(xs[i] / hX);
    if (b >= nIni) b = nIni - 1; /* reference: out-of-range write (UB); clamp */
    onode *nd = &L.pool[ini[b]];
    nd->keys[nd->nkeys++] = i;
  }
  for (int lit = L.head; lit >= 0;) { /* :608-619 */
    if (L.pool[lit].nkeys == 1) { L.pool[lit].noMore = 1; lit = L.pool[lit].next; }
    else if (L.pool[lit].nkeys == 0) lit = ol_erase(&L, lit);
    else lit = L.pool[lit].next;
  }
  int bFinish = 0;
  szptr *vSize = NULL; int nSize = 0, capSize = 0;
#define VS_PUSH(cnt, sq) do { if (nSize == capSize) { capSize = capSize ? capSize * 2 : 256; \
      vSize = (szptr *)realloc(vSize, sizeof(szptr) * capSize); } \
      vSize[nSize].count = (cnt); vSize[nSize].seq = (sq); nSize++; } while (0)
  while (!bFinish) {
    int prevSize = L.size;
    int lit = L.head;
    int nToExpand = 0;
    nSize = 0;
    while (lit >= 0) {
      if (L.pool[lit].noMore) { lit = L.pool[lit].next; continue; }
      int c[4];
      divide_node(&L, lit, xs, ys, c);
      for (int k = 0; k < 4; k++) {
        if (L.pool[c[k]].nkeys > 0) {
          ol_push_front(&L, c[k]);
          if (L.pool[c[k]].nkeys > 1) { nToExpand++; VS_PUSH(L.pool[c[k]].nkeys, c[k]); }
        }
      }
      lit = ol_erase(&L, lit);
    }
    if (L.size >= N || L.size == prevSize) {
      bFinish = 1;
    } else if (L.size + nToExpand * 3 > N) {
      while (!bFinish) {
        prevSize = L.size;
        int nPrev = nSize;
        szptr *vPrev = (szptr *)malloc(sizeof(szptr) * (nPrev > 0 ? nPrev : 1));
        memcpy(vPrev, vSize, sizeof(szptr) * nPrev);
        nSize = 0;
        qsort(vPrev, nPrev, sizeof(szptr), szptr_cmp);
        for (int j = nPrev - 1; j >= 0; j--) {
          int c[4];
          int pi = vPrev[j].seq;
          divide_node(&L, pi, xs, ys, c);
          for (int k = 0; k < 4; k++) {
            if (L.pool[c[k]].nkeys > 0) {
              ol_push_front(&L, c[k]);
              if (L.pool[c[k]].nkeys > 1) VS_PUSH(L.pool[c[k]].nkeys, c[k]);
            }
          }
          ol_erase(&L, pi);
          if (L.size >= N) break;
        }
        free(vPrev);
        if (L.size >= N || L.size == prevSize) bFinish = 1;
      }
    }
  }
#undef VS_PUSH
  int nout = 0;
  for (int lit = L.head; lit >= 0; lit = L.pool[lit].next) { /* :787-805 */
    onode *nd = &L.pool[lit];
    int best = nd->keys[0];
    float maxR = resp[best];
    for (int k = 1; k < nd->nkeys; k++)
      if (resp[nd->keys[k]] > maxR) { best = nd->keys[k]; maxR = resp[best]; }
    if (nout < cap) out_idx[nout] = best;
    nout++;
  }
  for (int i = 0; i < L.npool; i++) free(L.pool[i].keys);
  free(L.pool);
  free(vSize);
  free(ini);
  return nout;
}

/* ------------------------------------------------------------------ */
/* IC_Angle: src/ORBextractor.cc:78-106                                */
/* ------------------------------------------------------------------ */
float orc_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax) {
  int m_01 = 0, m_10 = 0;
  const uint8_t *center = img + (size_t)y * stride + x;
  for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
  for (int v = 1; v <= 15; ++v) {
    int v_sum = 0;
    int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ */
/* computeOrbDescriptor: src/ORBextractor.cc:111-152                   */
/* ------------------------------------------------------------------ */
void orc_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg,
                    uint8_t desc[32]) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float angle = angle_deg * factorPI;
  float a, b;
  orc_sincos(angle, &a, &b);
  const uint8_t *center = blur + (size_t)y * stride + x;
  const signed char *pat = ORC_BIT_PATTERN_31;
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int k = 0; k < 8; k++) {
      float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1];
      float x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
      int t0 = center[orc_cvround(x0 * b + y0 * a) * stride + orc_cvround(x0 * a - y0 * b)];
      int t1 = center[orc_cvround(x1 * b + y1 * a) * stride + orc_cvround(x1 * a - y1 * b)];
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ------------------------------------------------------------------ */
/* ORBextractor::operator(): src/ORBextractor.cc:1119-1197             */
/* ------------------------------------------------------------------ */
int orc_extract(orc_extractor *e, const uint8_t *img, int W, int H, int stride, orc_keypoint *kps,
                uint8_t *desc, int capacity, int *n_out, uint8_t *pyr_out) {
  *n_out = 0;
  if (!img || W <= 0 || H <= 0) return 0; /* :1122 empty image -> silent return */
  const int nl = e->nlevels;
  int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
  uint8_t *lev[ORC_MAX_LEVELS];
  double t0 = now_s();
  /* ComputePyramid :1203-1234 (the 19-px border is never read: SURVEY.md 8) */
  for (int l = 0; l < nl; l++) {
    orc_level_size(e, W, H, l, &lw[l], &lh[l]);
    lev[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] > 0 ? (size_t)lw[l] * lh[l] : 1);
    if (l == 0) {
      for (int y = 0; y < H; y++) memcpy(lev[0] + (size_t)y * W, img + (size_t)y * stride, W);
    } else {
      orc_resize_linear(lev[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lev[l], lw[l], lh[l], lw[l]);
    }
  }
  double t1 = now_s();
  e->t_pyramid += t1 - t0;
  if (pyr_out) {
    size_t off = 0;
    for (int l = 0; l < nl; l++) { memcpy(pyr_out + off, lev[l], (size_t)lw[l] * lh[l]); off += (size_t)lw[l] * lh[l]; }
  }
  /* ComputeKeyPointsOctTree :815-922 */
  orc_keypoint *lk[ORC_MAX_LEVELS];
  int ln[ORC_MAX_LEVELS];
  int total = 0;
  for (int l = 0; l < nl; l++) {
    double ta = now_s();
    const int minBX = 16, minBY = 16, maxBX = lw[l] - 16, maxBY = lh[l] - 16;
    int cap = (lw[l] * lh[l]) / 4 + 16;
    float *xs = (float *)malloc(sizeof(float) * cap * 3), *ys = xs + cap, *rs = ys + cap;
    int nc = (lw[l] > 32 && lh[l] > 32) ? orc_grid_candidates(e, lev[l], lw[l], lh[l], lw[l], xs, ys, rs, cap) : 0;
    double tb = now_s();
    e->t_fast += tb - ta;
    int *sel = (int *)malloc(sizeof(int) * (nc + 4));
    int ns = nc > 0 ? orc_distribute_octtree(xs, ys, rs, nc, minBX, maxBX, minBY, maxBY,
                                             e->mnFeaturesPerLevel[l], sel, nc + 4) : 0;
    e->t_octree += now_s() - tb;
    lk[l] = (orc_keypoint *)malloc(sizeof(orc_keypoint) * (ns > 0 ? ns : 1));
    ln[l] = ns;
    const int scaledPatchSize = (int)(31 * e->mvScaleFactor[l]);
    for (int i = 0; i < ns; i++) {
      orc_keypoint *k = &lk[l][i];
      k->x = xs[sel[i]] + (float)minBX;
      k->y = ys[sel[i]] + (float)minBY;
      k->size = (float)scaledPatchSize;
      k->angle = -1;
      k->response = rs[sel[i]];
      k->octave = l;
      k->class_id = -1;
    }
    free(sel);
    free(xs);
    total += ns;
  }
  double t2 = now_s();
  for (int l = 0; l < nl; l++) /* :920-921 */
    for (int i = 0; i < ln[l]; i++)
      lk[l][i].angle = orc_ic_angle(lev[l], lw[l], orc_cvround(lk[l][i].x), orc_cvround(lk[l][i].y), e->umax);
  e->t_orient += now_s() - t2;
  int rc = 0;
  if (total > capacity) rc = -1;
  int offset = 0;
  for (int l = 0; l < nl && rc == 0; l++) {
    if (ln[l] == 0) continue;
    double ta = now_s();
    uint8_t *blur = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
    orc_gaussian_blur7_spec(lev[l], lw[l], lh[l], lw[l], blur, lw[l], e->blur_spec);
    double tb = now_s();
    e->t_blur += tb - ta;
    for (int i = 0; i < ln[l]; i++) {
      orc_keypoint *k = &lk[l][i];
      orc_descriptor(blur, lw[l], orc_cvround(k->x), orc_cvround(k->y), k->angle, desc + (size_t)(offset + i) * 32);
    }
    if (l != 0) {
      float scale = e->mvScaleFactor[l];
      for (int i = 0; i < ln[l]; i++) { lk[l][i].x *= scale; lk[l][i].y *= scale; }
    }
    memcpy(kps + offset, lk[l], sizeof(orc_keypoint) * ln[l]);
    offset += ln[l];
    free(blur);
    e->t_desc += now_s() - tb;
  }
  for (int l = 0; l < nl; l++) { free(lk[l]); free(lev[l]); }
  *n_out = rc == 0 ? total : 0;
  return rc;
}

/* ------------------------------------------------------------------ */
/* matcher: src/ORBmatcher.cc                                          */
/* ------------------------------------------------------------------ */
/* bench.py's matching roofline counts the reference's DescriptorDistance calls: per-thread tally */
static __thread int64_t g_distance_calls = 0;
void orc_distance_calls_reset(void) { g_distance_calls = 0; }
int64_t orc_distance_calls(void) { return g_distance_calls; }
/* work counters of the last orc_compute_stereo_matches call of this thread (bench.py reports them next to the
   distance count): row-bucket entries scanned (src/Frame.cc:573-595), SAD refinements entered (:598-652) */
static __thread int64_t g_stereo_scanned = 0, g_stereo_sad = 0;
void orc_stereo_counters(int64_t *out) { out[0] = g_stereo_scanned; out[1] = g_stereo_sad; }

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) { /* :1828-1844 */
  int dist = 0;
  g_distance_calls++;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

void orc_three_maxima(const int *hs, int L, int *ind1, int *ind2, int *ind3) { /* :1777-1821 */
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = hs[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
    else if (s > max3) { max3 = s; *ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

#define HISTO_LENGTH 30
#define TH_LOW 50
#define TH_HIGH 100

typedef struct { int *v[HISTO_LENGTH]; int n[HISTO_LENGTH]; } rothist;
static void rh_init(rothist *h, int cap) { for (int i = 0; i < HISTO_LENGTH; i++) { h->v[i] = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1)); h->n[i] = 0; } }
static void rh_free(rothist *h) { for (int i = 0; i < HISTO_LENGTH; i++) free(h->v[i]); }
static void rh_push(rothist *h, float a1, float a2, int idx) { /* :272-281 */
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  h->v[bin][h->n[bin]++] = idx;
}
static int rh_prune(rothist *h, int32_t *arr) { /* :303-322; returns removed count */
  int i1 = -1, i2 = -1, i3 = -1, removed = 0;
  orc_three_maxima(h->n, HISTO_LENGTH, &i1, &i2, &i3);
  for (int i = 0; i < HISTO_LENGTH; i++) {
    if (i == i1 || i == i2 || i == i3) continue;
    for (int j = 0; j < h->n[i]; j++) { arr[h->v[i][j]] = -1; removed++; }
  }
  return removed;
}

/* merge-walk of two ascending node lists (std::map iteration + lower_bound, :211-300) */
#define FV_WALK_BEGIN(fv1, fv2) \
  { int a_ = 0, b_ = 0; \
    while (a_ < (fv1)->n_nodes && b_ < (fv2)->n_nodes) { \
      if ((fv1)->node_ids[a_] == (fv2)->node_ids[b_]) {
#define FV_WALK_END(fv1, fv2) \
        a_++; b_++; \
      } else if ((fv1)->node_ids[a_] < (fv2)->node_ids[b_]) { \
        while (a_ < (fv1)->n_nodes && (fv1)->node_ids[a_] < (fv2)->node_ids[b_]) a_++; \
      } else { \
        while (b_ < (fv2)->n_nodes && (fv2)->node_ids[b_] < (fv1)->node_ids[a_]) b_++; \
      } } }

int orc_search_by_bow(const uint8_t *desc1, const uint8_t *has_mp1, const float *ang1, int n1,
                      const orc_featvec *fv1, const uint8_t *desc2, const float *ang2, int n2,
                      const orc_featvec *fv2, float nnratio, int check_ori, int32_t *match_f) {
  (void)n1;
  for (int i = 0; i < n2; i++) match_f[i] = -1;
  int nmatches = 0;
  rothist rh;
  rh_init(&rh, n2);
  FV_WALK_BEGIN(fv1, fv2)
    for (int iKF = fv1->offsets[a_]; iKF < fv1->offsets[a_ + 1]; iKF++) {
      const unsigned realIdxKF = fv1->indices[iKF];
      if (!has_mp1[realIdxKF]) continue;
      const uint8_t *dKF = desc1 + (size_t)realIdxKF * 32;
      int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
      for (int iF = fv2->offsets[b_]; iF < fv2->offsets[b_ + 1]; iF++) {
        const unsigned realIdxF = fv2->indices[iF];
        if (match_f[realIdxF] >= 0) continue;
        const int dist = orc_descriptor_distance(dKF, desc2 + (size_t)realIdxF * 32);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 <= TH_LOW) {
        if ((float)bestDist1 < nnratio * (float)bestDist2) {
          match_f[bestIdxF] = (int32_t)realIdxKF;
          if (check_ori) rh_push(&rh, ang1[realIdxKF], ang2[bestIdxF], bestIdxF);
          nmatches++;
        }
      }
    }
  FV_WALK_END(fv1, fv2)
  if (check_ori) nmatches -= rh_prune(&rh, match_f);
  rh_free(&rh);
  return nmatches;
}

int orc_search_by_bow_kf(const uint8_t *desc1, const uint8_t *has_mp1, const float *ang1, int n1,
                         const orc_featvec *fv1, const uint8_t *desc2, const uint8_t *has_mp2,
                         const float *ang2, int n2, const orc_featvec *fv2, float nnratio,
                         int check_ori, int32_t *match12) {
  for (int i = 0; i < n1; i++) match12[i] = -1;
  uint8_t *matched2 = (uint8_t *)calloc(n2 > 0 ? n2 : 1, 1);
  int nmatches = 0;
  rothist rh;
  rh_init(&rh, n1);
  FV_WALK_BEGIN(fv1, fv2)
    for (int i1 = fv1->offsets[a_]; i1 < fv1->offsets[a_ + 1]; i1++) {
      const unsigned idx1 = fv1->indices[i1];
      if (!has_mp1[idx1]) continue;
      const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
      int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
      for (int i2 = fv2->offsets[b_]; i2 < fv2->offsets[b_ + 1]; i2++) {
        const unsigned idx2 = fv2->indices[i2];
        if (matched2[idx2] || !has_mp2[idx2]) continue;
        int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 < TH_LOW) {
        if ((float)bestDist1 < nnratio * (float)bestDist2) {
          match12[idx1] = bestIdx2;
          matched2[bestIdx2] = 1;
          if (check_ori) rh_push(&rh, ang1[idx1], ang2[bestIdx2], (int)idx1);
          nmatches++;
        }
      }
    }
  FV_WALK_END(fv1, fv2)
  if (check_ori) nmatches -= rh_prune(&rh, match12);
  rh_free(&rh);
  free(matched2);
  return nmatches;
}

/* CheckDistEpipolarLine :156-175 */
static int check_epipolar(float x1, float y1, float x2, float y2, const float *F, float sigma2) {
  const float a = x1 * F[0] + y1 * F[3] + F[6];
  const float b = x1 * F[1] + y1 * F[4] + F[7];
  const float c = x1 * F[2] + y1 * F[5] + F[8];
  const float num = a * x2 + b * y2 + c;
  const float den = a * a + b * b;
  if (den == 0) return 0;
  const float dsqr = num * num / den;
  return (double)dsqr < 3.84 * (double)sigma2;
}

int orc_search_for_triangulation(const uint8_t *desc1, const uint8_t *has_mp1, const float *x1,
                                 const float *y1, const float *ang1, const uint8_t *stereo1, int n1,
                                 const orc_featvec *fv1, const uint8_t *desc2,
                                 const uint8_t *has_mp2, const float *x2, const float *y2,
                                 const float *ang2, const int32_t *oct2, const uint8_t *stereo2,
                                 int n2, const orc_featvec *fv2, const float *F12, float ex,
                                 float ey, const float *scale_factors2, const float *level_sigma2_2,
                                 int only_stereo, int check_ori, int32_t *match12) {
  (void)n2;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  int nmatches = 0;
  rothist rh;
  rh_init(&rh, n1);
  FV_WALK_BEGIN(fv1, fv2)
    for (int i1 = fv1->offsets[a_]; i1 < fv1->offsets[a_ + 1]; i1++) {
      const unsigned idx1 = fv1->indices[i1];
      if (has_mp1[idx1]) continue;
      const int bStereo1 = stereo1[idx1];
      if (only_stereo && !bStereo1) continue;
      const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
      int bestDist = TH_LOW, bestIdx2 = -1;
      for (int i2 = fv2->offsets[b_]; i2 < fv2->offsets[b_ + 1]; i2++) {
        const unsigned idx2 = fv2->indices[i2];
        if (has_mp2[idx2]) continue; /* vbMatched2 is never set (:774,827) */
        const int bStereo2 = stereo2[idx2];
        if (only_stereo && !bStereo2) continue;
        const int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
        if (dist > TH_LOW || dist > bestDist) continue;
        if (!bStereo1 && !bStereo2) {
          const float distex = ex - x2[idx2];
          const float distey = ey - y2[idx2];
          if (distex * distex + distey * distey < 100 * scale_factors2[oct2[idx2]]) continue;
        }
        if (check_epipolar(x1[idx1], y1[idx1], x2[idx2], y2[idx2], F12, level_sigma2_2[oct2[idx2]])) {
          bestIdx2 = (int)idx2;
          bestDist = dist;
        }
      }
      if (bestIdx2 >= 0) {
        match12[idx1] = bestIdx2;
        nmatches++;
        if (check_ori) rh_push(&rh, ang1[idx1], ang2[bestIdx2], (int)idx1);
      }
    }
  FV_WALK_END(fv1, fv2)
  if (check_ori) nmatches -= rh_prune(&rh, match12);
  rh_free(&rh);
  return nmatches;
}

/* ------------------------------------------------------------------ */
/* Frame::ComputeStereoMatches: src/Frame.cc:512-686                   */
/* ------------------------------------------------------------------ */
typedef struct { int d, i; } distidx;
static int distidx_cmp(const void *a, const void *b) {
  const distidx *A = (const distidx *)a, *B = (const distidx *)b;
  if (A->d != B->d) return A->d < B->d ? -1 : 1;
  return A->i < B->i ? -1 : (A->i > B->i ? 1 : 0);
}

int orc_compute_stereo_matches(const orc_extractor *e, int W, int H, const orc_keypoint *kpL,
                               const uint8_t *descL, int N, const orc_keypoint *kpR,
                               const uint8_t *descR, int Nr, const uint8_t *pyrL,
                               const uint8_t *pyrR, float mbf, float mb, float *uRight,
                               float *depth) {
  for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = H;
  int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
  size_t loff[ORC_MAX_LEVELS];
  size_t off = 0;
  for (int l = 0; l < e->nlevels; l++) { orc_level_size(e, W, H, l, &lw[l], &lh[l]); loff[l] = off; off += (size_t)lw[l] * lh[l]; }
  /* vRowIndices :519-539 */
  int *rowCnt = (int *)calloc(nRows, sizeof(int));
  int **rowIdx = (int **)calloc(nRows, sizeof(int *));
  for (int pass = 0; pass < 2; pass++) {
    for (int iR = 0; iR < Nr; iR++) {
      const float kpY = kpR[iR].y;
      if (kpR[iR].octave < 0 || kpR[iR].octave >= e->nlevels || !(fabsf(kpY) < 3.0e38f)) continue; /* not a keypoint of this pyramid */
      const float r = 2.0f * e->mvScaleFactor[kpR[iR].octave];
      const int maxr = (int)ceilf(kpY + r);
      const int minr = (int)floorf(kpY - r);
      for (int yi = minr; yi <= maxr; yi++) {
        if (yi < 0 || yi >= nRows) continue; /* reference: unchecked (A13) */
        if (pass == 0) rowCnt[yi]++;
        else rowIdx[yi][rowCnt[yi]++] = iR;
      }
    }
    if (pass == 0)
      for (int y = 0; y < nRows; y++) { rowIdx[y] = (int *)malloc(sizeof(int) * (rowCnt[y] > 0 ? rowCnt[y] : 1)); rowCnt[y] = 0; }
  }
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  distidx *vDistIdx = (distidx *)malloc(sizeof(distidx) * (N > 0 ? N : 1));
  int nDist = 0;
  g_stereo_scanned = 0;
  g_stereo_sad = 0;
  for (int iL = 0; iL < N; iL++) {
    const orc_keypoint *kl = &kpL[iL];
    const int levelL = kl->octave;
    const float vL = kl->y, uL = kl->x;
    /* records no extractor writes: the reference indexes its tables out of range with them; here "no stereo" */
    if (levelL < 0 || levelL >= e->nlevels || !(fabsf(uL) < 3.0e38f) || !(fabsf(vL) < 3.0e38f)) continue;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    if (rowCnt[row] == 0) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    int bestIdxR = 0;
    const uint8_t *dL = descL + (size_t)iL * 32;
    g_stereo_scanned += rowCnt[row];
    for (int iC = 0; iC < rowCnt[row]; iC++) {
      const int iR = rowIdx[row][iC];
      const orc_keypoint *kr = &kpR[iR];
      if (kr->octave < levelL - 1 || kr->octave > levelL + 1) continue;
      const float uR = kr->x;
      if (uR >= minU && uR <= maxU) {
        const int dist = orc_descriptor_distance(dL, descR + (size_t)iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = kpR[bestIdxR].x;
      const float scaleFactor = e->mvInvScaleFactor[kl->octave];
      const float scaleduL = roundf(kl->x * scaleFactor);
      const float scaledvL = roundf(kl->y * scaleFactor);
      const float scaleduR0 = roundf(uR0 * scaleFactor);
      const int w = 5, L = 5;
      const int lv = kl->octave;
      const uint8_t *IL = pyrL + loff[lv], *IR = pyrR + loff[lv];
      const int st = lw[lv];
      const int cy = (int)scaledvL, cxL = (int)scaleduL;
      int bestD = INT_MAX, bestincR = 0;
      float vDists[11];
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= (float)lw[lv]) continue;
      /* rowRange / colRange (:609-610, :626) assert 0 <= start <= end <= size: a patch that leaves the level throws in
         the reference; "no stereo" here */
      if (cy < w || cy + w >= lh[lv] || cxL < w || cxL + w >= lw[lv] || (int)scaleduR0 < L + w) continue;
      g_stereo_sad++;
      const int cL = IL[(size_t)cy * st + cxL];
      for (int incR = -L; incR <= +L; incR++) {
        const int cxR = (int)(scaleduR0 + (float)incR);
        const int cR = IR[(size_t)cy * st + cxR];
        double acc = 0; /* cv::norm(NORM_L1) on CV_32F accumulates in double */
        for (int dy = -w; dy <= w; dy++)
          for (int dx = -w; dx <= w; dx++) {
            float a = (float)IL[(size_t)(cy + dy) * st + cxL + dx] - (float)cL;
            float b = (float)IR[(size_t)(cy + dy) * st + cxR + dx] - (float)cR;
            acc += fabs((double)(a - b));
          }
        float dist = (float)acc;
        if (dist < (float)bestD) { bestD = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = e->mvScaleFactor[kl->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
        depth[iL] = mbf / disparity;
        uRight[iL] = bestuR;
        vDistIdx[nDist].d = bestD;
        vDistIdx[nDist].i = iL;
        nDist++;
      }
    }
  }
  if (nDist > 0) { /* :672-685; empty case is UB in the reference -- guarded */
    qsort(vDistIdx, nDist, sizeof(distidx), distidx_cmp);
    const float median = (float)vDistIdx[nDist / 2].d;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = nDist - 1; i >= 0; i--) {
      if ((float)vDistIdx[i].d < thDist) break;
      uRight[vDistIdx[i].i] = -1;
      depth[vDistIdx[i].i] = -1;
    }
  }
  int cnt = 0;
  for (int i = 0; i < N; i++) cnt += uRight[i] >= 0 || depth[i] >= 0;
  for (int y = 0; y < nRows; y++) free(rowIdx[y]);
  free(rowIdx);
  free(rowCnt);
  free(vDistIdx);
  return cnt;
}

/* ------------------------------------------------------------------ */
/* DBoW2 vocabulary: text loader and transform                         */
/* Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424, 1127-1259   */
/* ------------------------------------------------------------------ */
/* children lists in file order (m_nodes[pid].children.push_back(nid), :1386) */
static void vocab_link_children(orc_vocab *v) {
  const int n = v->n_nodes;
  v->child_off = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
  v->child_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 1 ? n - 1 : 1));
  for (int i = 1; i < n; i++) v->child_off[v->parent[i] + 1]++;
  for (int i = 0; i < n; i++) v->child_off[i + 1] += v->child_off[i];
  int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  memcpy(cur, v->child_off, sizeof(int32_t) * (size_t)n);
  for (int i = 1; i < n; i++) v->child_idx[cur[v->parent[i]]++] = i;
  free(cur);
}

orc_vocab *orc_vocab_load_text(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *txt = (char *)malloc((size_t)sz + 1);
  if (fread(txt, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(txt); return NULL; }
  fclose(f);
  txt[sz] = 0;
  orc_vocab *v = (orc_vocab *)calloc(1, sizeof(orc_vocab));
  char *line = txt, *end;
  /* header: k L scoring weighting (:1351-1362) */
  end = strchr(line, '\n');
  if (end) *end = 0;
  int n1 = -1, n2 = -1;
  if (sscanf(line, "%d %d %d %d", &v->k, &v->L, &n1, &n2) != 4 || v->k < 0 || v->k > 20 || v->L < 1 ||
      v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { free(txt); free(v); return NULL; }
  v->scoring = n1;
  v->weighting = n2;
  int cap = 1024, n = 1;
  v->parent = (int32_t *)malloc(sizeof(int32_t) * cap);
  v->desc = (uint8_t *)calloc((size_t)cap * 32, 1);
  v->weight = (double *)calloc(cap, sizeof(double));
  v->word_id = (int32_t *)malloc(sizeof(int32_t) * cap);
  v->parent[0] = 0;
  v->word_id[0] = -1;
  line = end ? end + 1 : txt + sz;
  while (line < txt + sz) {
    end = strchr(line, '\n');
    if (end) *end = 0;
    char *p = line;
    while (*p == ' ' || *p == '\t' || *p == '\r') p++;
    if (*p) { /* non-empty line = one node (:1374-1417) */
      if (n == cap) {
        cap *= 2;
        v->parent = (int32_t *)realloc(v->parent, sizeof(int32_t) * cap);
        v->desc = (uint8_t *)realloc(v->desc, (size_t)cap * 32);
        v->weight = (double *)realloc(v->weight, sizeof(double) * cap);
        v->word_id = (int32_t *)realloc(v->word_id, sizeof(int32_t) * cap);
      }
      char *q;
      long pid = strtol(p, &q, 10);
      long leaf = strtol(q, &q, 10);
      for (int i = 0; i < 32; i++) v->desc[(size_t)n * 32 + i] = (uint8_t)strtol(q, &q, 10);
      v->weight[n] = strtod(q, &q);
      v->parent[n] = (int32_t)pid;
      v->word_id[n] = leaf > 0 ? v->n_words++ : -1;
      n++;
    }
    line = end ? end + 1 : txt + sz;
  }
  v->n_nodes = n;
  free(txt);
  vocab_link_children(v);
  return v;
}

/* the same tree from arrays: entry i = node i+1 = one node line of the text file (:1374-1417) */
orc_vocab *orc_vocab_from_arrays(int k, int L, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                                 const uint8_t *desc, const double *weight) {
  orc_vocab *v = (orc_vocab *)calloc(1, sizeof(orc_vocab));
  const int n = n_nodes + 1;
  v->k = k; v->L = L;
  v->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  v->desc = (uint8_t *)calloc((size_t)n * 32, 1);
  v->weight = (double *)calloc((size_t)n, sizeof(double));
  v->word_id = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  v->parent[0] = 0;
  v->word_id[0] = -1;
  for (int i = 0; i < n_nodes; i++) {
    v->parent[i + 1] = parent[i];
    v->weight[i + 1] = weight[i];
    v->word_id[i + 1] = is_leaf[i] ? v->n_words++ : -1;
  }
  memcpy(v->desc + 32, desc, (size_t)n_nodes * 32);
  v->n_nodes = n;
  vocab_link_children(v);
  return v;
}

void orc_vocab_free(orc_vocab *v) {
  if (!v) return;
  free(v->parent); free(v->child_off); free(v->child_idx); free(v->desc); free(v->weight); free(v->word_id);
  free(v);
}

int orc_vocab_transform(const orc_vocab *v, const uint8_t *desc, int n, int levelsup, uint32_t *word_id,
                        double *weight, uint32_t *node_id) {
  int used = 0;
  const int nid_level = v->L - levelsup;
  for (int i = 0; i < n; i++) {
    word_id[i] = 0; weight[i] = 0; node_id[i] = 0;
    if (v->child_off[1] == v->child_off[0]) continue; /* empty vocabulary (:1134) */
    const uint8_t *fd = desc + (size_t)i * 32;
    int final_id = 0, current_level = 0;
    uint32_t nid = 0; /* root when nid_level <= 0 (:1228) */
    do {
      ++current_level;
      const int b = v->child_off[final_id], e = v->child_off[final_id + 1];
      final_id = v->child_idx[b];
      int best_d = orc_descriptor_distance(fd, v->desc + (size_t)final_id * 32); /* FORB::distance */
      for (int c = b + 1; c < e; c++) {
        const int id = v->child_idx[c];
        const int d = orc_descriptor_distance(fd, v->desc + (size_t)id * 32);
        if (d < best_d) { best_d = d; final_id = id; }
      }
      if (current_level == nid_level) nid = (uint32_t)final_id;
    } while (v->child_off[final_id + 1] > v->child_off[final_id]);
    word_id[i] = v->word_id[final_id] >= 0 ? (uint32_t)v->word_id[final_id] : 0u;
    weight[i] = v->weight[final_id];
    node_id[i] = nid;
    if (weight[i] > 0) used++;
  }
  return used;
}

/* ------------------------------------------------------------------ */
/* MapPoint::ComputeDistinctiveDescriptors: src/MapPoint.cc:269-333    */
/* ------------------------------------------------------------------ */
static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int orc_distinctive_descriptor(const uint8_t *desc, int n) {
  if (n <= 0) return -1;
  int *row = (int *)malloc(sizeof(int) * n);
  int bestMedian = INT_MAX, bestIdx = 0;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) row[j] = i == j ? 0 : orc_descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
    qsort(row, n, sizeof(int), int_cmp);
    const int median = row[(size_t)(0.5 * (n - 1))];
    if (median < bestMedian) { bestMedian = median; bestIdx = i; }
  }
  free(row);
  return bestIdx;
}

/* ------------------------------------------------------------------ */
/* cv::cvtColor to gray, 8U: src/Tracking.cc:176-262                   */
/* ------------------------------------------------------------------ */
void orc_cvt_gray(const uint8_t *src, int w, int h, int sstride, int channels, int rgb_order,
                  uint8_t *dst, int dstride) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint8_t *p = src + (size_t)y * sstride + (size_t)x * channels;
      const int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
      dst[(size_t)y * dstride + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
}

/* ------------------------------------------------------------------ */
/* Frame grid: src/Frame.cc:246-267 (AssignFeaturesToGrid), :417-427    */
/* (PosInGrid), :358-415 (GetFeaturesInArea)                            */
/* ------------------------------------------------------------------ */
static int pos_in_grid(const orc_frame *f, float x, float y, int *px, int *py) {
  const float wInv = (float)ORC_GRID_COLS / (f->mnMaxX - f->mnMinX); /* src/Frame.cc:109-110 */
  const float hInv = (float)ORC_GRID_ROWS / (f->mnMaxY - f->mnMinY);
  *px = (int)roundf((x - f->mnMinX) * wInv);
  *py = (int)roundf((y - f->mnMinY) * hInv);
  return !(*px < 0 || *px >= ORC_GRID_COLS || *py < 0 || *py >= ORC_GRID_ROWS);
}

void orc_frame_build_grid(orc_frame *f) {
  const int nc = ORC_GRID_COLS * ORC_GRID_ROWS;
  f->cell_off = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
  f->cell_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(f->N > 0 ? f->N : 1));
  int px, py;
  for (int i = 0; i < f->N; i++)
    if (pos_in_grid(f, f->x[i], f->y[i], &px, &py)) f->cell_off[px * ORC_GRID_ROWS + py + 1]++;
  for (int c = 0; c < nc; c++) f->cell_off[c + 1] += f->cell_off[c];
  int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)nc);
  memcpy(cur, f->cell_off, sizeof(int32_t) * (size_t)nc);
  for (int i = 0; i < f->N; i++)
    if (pos_in_grid(f, f->x[i], f->y[i], &px, &py)) f->cell_idx[cur[px * ORC_GRID_ROWS + py]++] = i;
  free(cur);
}
void orc_frame_free_grid(orc_frame *f) { free(f->cell_off); free(f->cell_idx); f->cell_off = NULL; f->cell_idx = NULL; }

int orc_features_in_area(const orc_frame *f, float x, float y, float r, int minLevel, int maxLevel, int32_t *out, int cap) {
  const float wInv = (float)ORC_GRID_COLS / (f->mnMaxX - f->mnMinX);
  const float hInv = (float)ORC_GRID_ROWS / (f->mnMaxY - f->mnMinY);
  int n = 0;
  int nMinCellX = (int)floorf((x - f->mnMinX - r) * wInv);
  if (nMinCellX < 0) nMinCellX = 0;
  if (nMinCellX >= ORC_GRID_COLS) return 0;
  int nMaxCellX = (int)ceilf((x - f->mnMinX + r) * wInv);
  if (nMaxCellX > ORC_GRID_COLS - 1) nMaxCellX = ORC_GRID_COLS - 1;
  if (nMaxCellX < 0) return 0;
  int nMinCellY = (int)floorf((y - f->mnMinY - r) * hInv);
  if (nMinCellY < 0) nMinCellY = 0;
  if (nMinCellY >= ORC_GRID_ROWS) return 0;
  int nMaxCellY = (int)ceilf((y - f->mnMinY + r) * hInv);
  if (nMaxCellY > ORC_GRID_ROWS - 1) nMaxCellY = ORC_GRID_ROWS - 1;
  if (nMaxCellY < 0) return 0;
  const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * ORC_GRID_ROWS + iy;
      for (int j = f->cell_off[c]; j < f->cell_off[c + 1]; j++) {
        const int id = f->cell_idx[j];
        if (bCheckLevels) {
          if (f->octave[id] < minLevel) continue;
          if (maxLevel >= 0 && f->octave[id] > maxLevel) continue;
        }
        const float distx = f->x[id] - x, disty = f->y[id] - y;
        if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = id; n++; }
      }
    }
  return n;
}

/* src/ORBmatcher.cc:140-150 */
static float radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }

int orc_search_by_projection_mappoints(orc_frame *F, const float *sf, const uint8_t *blocked0, int nMP,
                                       const uint8_t *in_view, const int32_t *level, const float *view_cos,
                                       const float *proj_x, const float *proj_y, const float *proj_xr,
                                       const uint8_t *mp_desc, const uint8_t *mp_obs_positive, float th, float nnratio,
                                       int32_t *match) {
  int nmatches = 0;
  const int bFactor = th != 1.0;
  uint8_t *blocked = (uint8_t *)malloc((size_t)(F->N > 0 ? F->N : 1));
  memcpy(blocked, blocked0, (size_t)F->N);
  for (int i = 0; i < F->N; i++) match[i] = -1;
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(F->N > 0 ? F->N : 1));
  for (int iMP = 0; iMP < nMP; iMP++) {
    if (!in_view[iMP]) continue;
    const int nPredictedLevel = level[iMP];
    float r = radius_by_viewing_cos(view_cos[iMP]);
    if (bFactor) r *= th;
    const int nc = orc_features_in_area(F, proj_x[iMP], proj_y[iMP], r * sf[nPredictedLevel], nPredictedLevel - 1,
                                        nPredictedLevel, vIdx, F->N);
    if (nc == 0) continue;
    const uint8_t *d0 = mp_desc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = 0; c < nc; c++) {
      const int idx = vIdx[c];
      if (blocked[idx]) continue;
      if (F->uRight && F->uRight[idx] > 0) {
        const float er = fabsf(proj_xr[iMP] - F->uRight[idx]);
        if (er > r * sf[nPredictedLevel]) continue;
      }
      const int dist = orc_descriptor_distance(d0, F->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F->octave[idx]; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = F->octave[idx]; bestDist2 = dist; }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      match[bestIdx] = iMP;
      blocked[bestIdx] = mp_obs_positive ? mp_obs_positive[iMP] : 1;
      nmatches++;
    }
  }
  free(vIdx);
  free(blocked);
  return nmatches;
}

int orc_search_by_projection_lastframe(orc_frame *Cur, const float *sf, float mbf, int nLast, const uint8_t *valid,
                                       const float *u, const float *v, const float *invzc, const int32_t *last_octave,
                                       const float *last_angle, const uint8_t *mp_desc, const uint8_t *obs_positive,
                                       const uint8_t *blocked_at_entry, int mode, float th, int check_ori,
                                       int32_t *match_cur) {
  int nmatches = 0;
  rothist rh;
  rh_init(&rh, nLast > Cur->N ? nLast : Cur->N);
  for (int i = 0; i < Cur->N; i++) match_cur[i] = -1;
  uint8_t *blocked = (uint8_t *)calloc((size_t)(Cur->N > 0 ? Cur->N : 1), 1);
  if (blocked_at_entry) memcpy(blocked, blocked_at_entry, (size_t)Cur->N); /* :1572-1574 on the entry state */
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Cur->N > 0 ? Cur->N : 1));
  for (int i = 0; i < nLast; i++) {
    if (!valid[i]) continue;
    const int nLastOctave = last_octave[i];
    const float radius = th * sf[nLastOctave];
    int nc;
    if (mode == 1) nc = orc_features_in_area(Cur, u[i], v[i], radius, nLastOctave, -1, vIdx, Cur->N);
    else if (mode == 2) nc = orc_features_in_area(Cur, u[i], v[i], radius, 0, nLastOctave, vIdx, Cur->N);
    else nc = orc_features_in_area(Cur, u[i], v[i], radius, nLastOctave - 1, nLastOctave + 1, vIdx, Cur->N);
    if (nc == 0) continue;
    const uint8_t *dMP = mp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < nc; c++) {
      const int i2 = vIdx[c];
      if (blocked[i2]) continue; /* mvpMapPoints[i2] set with Observations()>0 */
      if (Cur->uRight && Cur->uRight[i2] > 0) {
        const float ur = u[i] - mbf * invzc[i];
        const float er = fabsf(ur - Cur->uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = orc_descriptor_distance(dMP, Cur->desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      match_cur[bestIdx2] = i;
      blocked[bestIdx2] = obs_positive ? obs_positive[i] : 1;
      nmatches++;
      if (check_ori) rh_push(&rh, last_angle[i], Cur->angle[bestIdx2], bestIdx2);
    }
  }
  if (check_ori) nmatches -= rh_prune(&rh, match_cur);
  rh_free(&rh);
  free(vIdx);
  free(blocked);
  return nmatches;
}

int orc_search_by_projection_reloc(orc_frame *Cur, const float *sf, int n, const uint8_t *valid, const float *u,
                                   const float *v, const int32_t *level, const float *kf_angle,
                                   const uint8_t *mp_desc, const uint8_t *blocked0, float th, int orb_dist,
                                   int check_ori, int32_t *match_cur) {
  int nmatches = 0;
  rothist rh;
  rh_init(&rh, n);
  for (int i = 0; i < Cur->N; i++) match_cur[i] = -1;
  uint8_t *blocked = (uint8_t *)calloc((size_t)(Cur->N > 0 ? Cur->N : 1), 1);
  if (blocked0) memcpy(blocked, blocked0, (size_t)Cur->N);
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Cur->N > 0 ? Cur->N : 1));
  for (int i = 0; i < n; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * sf[nPredictedLevel];
    const int nc = orc_features_in_area(Cur, u[i], v[i], radius, nPredictedLevel - 1, nPredictedLevel + 1, vIdx, Cur->N);
    if (nc == 0) continue;
    const uint8_t *dMP = mp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < nc; c++) {
      const int i2 = vIdx[c];
      if (blocked[i2]) continue; /* CurrentFrame.mvpMapPoints[i2] != NULL */
      const int dist = orc_descriptor_distance(dMP, Cur->desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= orb_dist) {
      match_cur[bestIdx2] = i;
      blocked[bestIdx2] = 1;
      nmatches++;
      if (check_ori) rh_push(&rh, kf_angle[i], Cur->angle[bestIdx2], bestIdx2);
    }
  }
  if (check_ori) nmatches -= rh_prune(&rh, match_cur);
  rh_free(&rh);
  free(vIdx);
  free(blocked);
  return nmatches;
}

int orc_search_by_projection_sim3(orc_frame *KF, const float *sf, int n, const uint8_t *valid, const float *u,
                                  const float *v, const int32_t *level, const uint8_t *mp_desc,
                                  const uint8_t *matched0, float th, int32_t *match) {
  int nmatches = 0;
  for (int i = 0; i < KF->N; i++) match[i] = -1;
  uint8_t *matched = (uint8_t *)calloc((size_t)(KF->N > 0 ? KF->N : 1), 1);
  if (matched0) memcpy(matched, matched0, (size_t)KF->N);
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(KF->N > 0 ? KF->N : 1));
  for (int i = 0; i < n; i++) {
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * sf[nPredictedLevel];
    const int nc = orc_features_in_area(KF, u[i], v[i], radius, -1, -1, vIdx, KF->N); /* KeyFrame::GetFeaturesInArea */
    if (nc == 0) continue;
    const uint8_t *dMP = mp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx = -1;
    for (int c = 0; c < nc; c++) {
      const int idx = vIdx[c];
      if (matched[idx]) continue;
      const int kpLevel = KF->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = orc_descriptor_distance(dMP, KF->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_LOW) { match[bestIdx] = i; matched[bestIdx] = 1; nmatches++; }
  }
  free(vIdx);
  free(matched);
  return nmatches;
}

int orc_search_for_initialization(const orc_frame *F1, orc_frame *F2, float *prev_x, float *prev_y, int window,
                                  float nnratio, int check_ori, int32_t *match12) {
  int nmatches = 0;
  for (int i = 0; i < F1->N; i++) match12[i] = -1;
  rothist rh;
  rh_init(&rh, F1->N);
  const int n2 = F2->N > 0 ? F2->N : 1;
  int *vMatchedDistance = (int *)malloc(sizeof(int) * (size_t)n2);
  int *vnMatches21 = (int *)malloc(sizeof(int) * (size_t)n2);
  for (int i = 0; i < F2->N; i++) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n2);
  for (int i1 = 0; i1 < F1->N; i1++) {
    const int level1 = F1->octave[i1];
    if (level1 > 0) continue;
    const int nc = orc_features_in_area(F2, prev_x[i1], prev_y[i1], (float)window, level1, level1, vIdx, F2->N);
    if (nc == 0) continue;
    const uint8_t *d1 = F1->desc + (size_t)i1 * 32;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int c = 0; c < nc; c++) {
      const int i2 = vIdx[c];
      const int dist = orc_descriptor_distance(d1, F2->desc + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if ((float)bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { match12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        match12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_ori) rh_push(&rh, F1->angle[i1], F2->angle[bestIdx2], i1);
      }
    }
  }
  if (check_ori) { /* :568-592: only entries still matched are cleared and counted */
    int i1 = -1, i2 = -1, i3 = -1;
    orc_three_maxima(rh.n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int j = 0; j < rh.n[i]; j++) {
        const int idx1 = rh.v[i][j];
        if (match12[idx1] >= 0) { match12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < F1->N; i1++)
    if (match12[i1] >= 0) { prev_x[i1] = F2->x[match12[i1]]; prev_y[i1] = F2->y[match12[i1]]; }
  rh_free(&rh);
  free(vIdx);
  free(vMatchedDistance);
  free(vnMatches21);
  return nmatches;
}

void orc_fuse_search(orc_frame *KF, const float *sf, const float *inv_level_sigma2, int n, const uint8_t *valid,
                     const float *u, const float *v, const float *ur, const int32_t *level, const uint8_t *mp_desc,
                     float th, int chi2, int32_t *best_idx) {
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(KF->N > 0 ? KF->N : 1));
  for (int i = 0; i < n; i++) {
    best_idx[i] = -1;
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * sf[nPredictedLevel];
    const int nc = orc_features_in_area(KF, u[i], v[i], radius, -1, -1, vIdx, KF->N);
    const uint8_t *dMP = mp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx = -1;
    for (int c = 0; c < nc; c++) {
      const int idx = vIdx[c];
      const int kpLevel = KF->octave[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (chi2) {
        const float kpx = KF->x[idx], kpy = KF->y[idx];
        if (KF->uRight && KF->uRight[idx] >= 0) {
          const float kpr = KF->uRight[idx];
          const float ex = u[i] - kpx, ey = v[i] - kpy, er = ur[i] - kpr;
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
        } else {
          const float ex = u[i] - kpx, ey = v[i] - kpy;
          const float e2 = ex * ex + ey * ey;
          if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
        }
      }
      const int dist = orc_descriptor_distance(dMP, KF->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_LOW) best_idx[i] = bestIdx;
  }
  free(vIdx);
}

static void sim3_one_way(orc_frame *KF, const float *sf, int n, const uint8_t *valid, const float *u, const float *v,
                         const int32_t *level, const uint8_t *desc, float th, int *vnMatch) {
  int32_t *vIdx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(KF->N > 0 ? KF->N : 1));
  for (int i = 0; i < n; i++) {
    vnMatch[i] = -1;
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * sf[nPredictedLevel];
    const int nc = orc_features_in_area(KF, u[i], v[i], radius, -1, -1, vIdx, KF->N);
    const uint8_t *dMP = desc + (size_t)i * 32;
    int bestDist = INT_MAX, bestIdx = -1;
    for (int c = 0; c < nc; c++) {
      const int idx = vIdx[c];
      if (KF->octave[idx] < nPredictedLevel - 1 || KF->octave[idx] > nPredictedLevel) continue;
      const int dist = orc_descriptor_distance(dMP, KF->desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
  }
  free(vIdx);
}

int orc_search_by_sim3(orc_frame *KF1, orc_frame *KF2, const float *sf1, const float *sf2, const uint8_t *valid1,
                       const float *u1, const float *v1, const int32_t *level1, const uint8_t *desc1,
                       const uint8_t *valid2, const float *u2, const float *v2, const int32_t *level2,
                       const uint8_t *desc2, float th, int32_t *match12) {
  int *vnMatch1 = (int *)malloc(sizeof(int) * (size_t)(KF1->N > 0 ? KF1->N : 1));
  int *vnMatch2 = (int *)malloc(sizeof(int) * (size_t)(KF2->N > 0 ? KF2->N : 1));
  sim3_one_way(KF2, sf2, KF1->N, valid1, u1, v1, level1, desc1, th, vnMatch1); /* KF1 points searched in KF2 */
  sim3_one_way(KF1, sf1, KF2->N, valid2, u2, v2, level2, desc2, th, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < KF1->N; i1++) {
    match12[i1] = -1;
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nFound++; }
  }
  free(vnMatch1);
  free(vnMatch2);
  return nFound;
}

/* ------------------------------------------------------------------ */
/* cv::remap, CV_32FC1 map pair, INTER_LINEAR, BORDER_CONSTANT, 8UC1    */
/* (Examples/Stereo/stereo_euroc.cc:136-137)                            */
/* ------------------------------------------------------------------ */
static int cvround_sse(float v) { /* cvtss2si: half to even; out of range / NaN -> 0x80000000 */
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)nearbyintf(v);
}
static int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

void orc_remap_linear(const uint8_t *src, int sw, int sh, int sstride, const float *mapx, const float *mapy,
                      int map_stride, int dw, int dh, uint8_t *dst, int dstride) {
  const unsigned width1 = (unsigned)(sw - 1 > 0 ? sw - 1 : 0), height1 = (unsigned)(sh - 1 > 0 ? sh - 1 : 0);
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++) {
      const int isx = cvround_sse(mapx[(size_t)y * map_stride + x] * 32), isy = cvround_sse(mapy[(size_t)y * map_stride + x] * 32);
      const int fx = isx & 31, fy = isy & 31;
      const int sx = sat_short(isx >> 5), sy = sat_short(isy >> 5);
      const int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
      int v0, v1, v2, v3;
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t *S = src + (size_t)sy * sstride + sx;
        v0 = S[0]; v1 = S[1]; v2 = S[sstride]; v3 = S[sstride + 1];
      } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
        dst[(size_t)y * dstride + x] = 0;
        continue;
      } else { /* borderInterpolate(BORDER_CONSTANT) = -1 outside, tap replaced by the border value 0 */
        const int x0 = (sx >= 0 && sx < sw) ? sx : -1, x1 = (sx + 1 >= 0 && sx + 1 < sw) ? sx + 1 : -1;
        const int y0 = (sy >= 0 && sy < sh) ? sy : -1, y1 = (sy + 1 >= 0 && sy + 1 < sh) ? sy + 1 : -1;
        v0 = (x0 >= 0 && y0 >= 0) ? src[(size_t)y0 * sstride + x0] : 0;
        v1 = (x1 >= 0 && y0 >= 0) ? src[(size_t)y0 * sstride + x1] : 0;
        v2 = (x0 >= 0 && y1 >= 0) ? src[(size_t)y1 * sstride + x0] : 0;
        v3 = (x1 >= 0 && y1 >= 0) ? src[(size_t)y1 * sstride + x1] : 0;
      }
      const int val = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
      dst[(size_t)y * dstride + x] = (uint8_t)(val > 255 ? 255 : val);
    }
}

/* ------------------------------------------------------------------ */
/* cv::undistortPoints (cvUndistortPoints), Frame.cc:443-510, 689-713   */
/* ------------------------------------------------------------------ */
void orc_undistort_points(const float *xy, int n, const float *K4, const float *dist, int n_dist, float *out_xy) {
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n_dist && i < 8; i++) k[i] = (double)dist[i];
  const double fx = (double)K4[0], fy = (double)K4[1], cx = (double)K4[2], cy = (double)K4[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  const int iters = n_dist > 0 ? 5 : 1;
  /* RR = P * I = K: rows (fx 0 cx), (0 fy cy), (0 0 1) */
  for (int i = 0; i < n; i++) {
    double x = (double)xy[2 * i], y = (double)xy[2 * i + 1], x0, y0;
    x0 = x = (x - cx) * ifx;
    y0 = y = (y - cy) * ify;
    for (int j = 0; j < iters; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx;
    const double yy = 0.0 * x + fy * y + cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    out_xy[2 * i] = (float)(xx * ww);
    out_xy[2 * i + 1] = (float)(yy * ww);
  }
}

/* ------------------------------------------------------------------ */
/* cv::initUndistortRectifyMap(K, D, R, P[:3,:3], size, CV_32F, M1, M2) */
/* as Examples/Stereo/stereo_euroc.cc:97-98 calls it, once at start-up.  */
/* Restated from the published imgproc/undistort.cpp (scalar path):      */
/* iR = inv(P * R) by cv::invert's closed 3x3 form, the row-wise         */
/* incremental _x += ir[0] sums, the rational radial + tangential model  */
/* in double, (float) at the end.  Parity unpinned like every OpenCV     */
/* primitive here (an AVX2 build of OpenCV 4.x evaluates the columns in  */
/* lanes; not modelled).  D: 4, 5 or 8 coefficients (k1 k2 p1 p2 [k3     */
/* [k4 k5 k6]]); R == NULL: identity; P == NULL: K.  Returns 0 / -1.     */
/* ------------------------------------------------------------------ */
int orc_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P, int w,
                                   int h, float *map_x, float *map_y) {
  if (!K || w <= 0 || h <= 0 || !map_x || !map_y || !(nD == 0 || nD == 4 || nD == 5 || nD == 8) || (nD > 0 && !D)) return -1;
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double *Rm = R ? R : I3, *Ar = P ? P : K;
  double M[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += Ar[3 * i + k] * Rm[3 * k + j];
      M[3 * i + j] = acc;
    }
  /* cv::invert, n == 3: determinant by the first row, adjugate times 1/det */
  double d = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (d == 0) return -1;
  d = 1. / d;
  double ir[9];
  ir[0] = (M[4] * M[8] - M[5] * M[7]) * d;
  ir[1] = (M[2] * M[7] - M[1] * M[8]) * d;
  ir[2] = (M[1] * M[5] - M[2] * M[4]) * d;
  ir[3] = (M[5] * M[6] - M[3] * M[8]) * d;
  ir[4] = (M[0] * M[8] - M[2] * M[6]) * d;
  ir[5] = (M[2] * M[3] - M[0] * M[5]) * d;
  ir[6] = (M[3] * M[7] - M[4] * M[6]) * d;
  ir[7] = (M[1] * M[6] - M[0] * M[7]) * d;
  ir[8] = (M[0] * M[4] - M[1] * M[3]) * d;
  const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
  const double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0;
  const double k3 = nD >= 5 ? D[4] : 0, k4 = nD >= 8 ? D[5] : 0, k5 = nD >= 8 ? D[6] : 0, k6 = nD >= 8 ? D[7] : 0;
  for (int i = 0; i < h; i++) {
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
      const double ww = 1. / _w, x = _x * ww, y = _y * ww;
      const double x2 = x * x, y2 = y * y;
      const double r2 = x2 + y2, _2xy = 2 * x * y;
      const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
      const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
      const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
      map_x[(size_t)i * w + j] = (float)(fx * xd + u0);
      map_y[(size_t)i * w + j] = (float)(fy * yd + v0);
    }
  }
  return 0;
}

void orc_image_bounds(int cols, int rows, const float *K4, const float *dist, int n_dist, float *b) {
  if (n_dist > 0 && dist[0] != 0.0) {
    const float in[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    float out[8];
    orc_undistort_points(in, 4, K4, dist, n_dist, out);
    b[0] = out[4] < out[0] ? out[4] : out[0];   /* std::min(a,b): b < a ? b : a */
    b[1] = out[2] < out[6] ? out[6] : out[2];   /* std::max(a,b): a < b ? b : a */
    b[2] = out[3] < out[1] ? out[3] : out[1];
    b[3] = out[5] < out[7] ? out[7] : out[5];
  } else {
    b[0] = 0.0f; b[1] = (float)cols; b[2] = 0.0f; b[3] = (float)rows;
  }
}

void orc_stereo_from_rgbd(const float *kx, const float *ky, const float *kux, int n, const float *depth_img, int w,
                          int h, int stride_floats, float mbf, float *uRight, float *depth) {
  (void)w; (void)h;
  for (int i = 0; i < n; i++) {
    uRight[i] = -1; depth[i] = -1;
    const float d = depth_img[(size_t)(int)ky[i] * stride_floats + (int)kx[i]];
    if (d > 0) { depth[i] = d; uRight[i] = kux[i] - mbf / d; }
  }
}
