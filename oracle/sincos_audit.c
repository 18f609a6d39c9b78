/* sincos_audit.c -- quantifies the documented deviation "own sincos vs glibc cosf/sinf"
 * (hard part 2).  Walks every float32 angle in [0,360) degrees exactly as
 * computeOrbDescriptor does (rad = angle * (float)(pi/180), src/ORBextractor.cc:108,115),
 * compares orc_sincos with libm cosf/sinf, and counts the angles for which any of the 1024
 * rounded sampling offsets of the rBRIEF pattern differs.  Test infrastructure only.
 * usage: sincos_audit [stride]   (stride 1 = exhaustive, ~70.5M angles) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

int main(int argc, char **argv) {
  long stride = argc > 1 ? atol(argv[1]) : 1;
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const signed char *pat = orc_pattern();
  uint32_t lo = 0, hi;
  float f360 = 360.0f;
  memcpy(&hi, &f360, 4);
  long n = 0, valdiff = 0, geodiff = 0, crdiff = 0;
  for (uint32_t bits = lo; bits < hi; bits += (uint32_t)stride) {
    float deg;
    memcpy(&deg, &bits, 4);
    float rad = deg * factorPI;
    float c, s;
    orc_sincos(rad, &c, &s);
    float gc = cosf(rad), gs = sinf(rad);
    float rc = (float)cos((double)rad), rs = (float)sin((double)rad);
    n++;
    if (c != rc || s != rs) crdiff++;
    if (c == gc && s == gs) continue;
    valdiff++;
    int differs = 0;
    for (int k = 0; k < 512 && !differs; k++) {
      float x = (float)pat[2 * k], y = (float)pat[2 * k + 1];
      if (orc_cvround(x * s + y * c) != orc_cvround(x * gs + y * gc)) differs = 1;
      if (orc_cvround(x * c - y * s) != orc_cvround(x * gc - y * gs)) differs = 1;
    }
    geodiff += differs;
  }
  printf("{\"angles\": %ld, \"stride\": %ld, \"value_differs_from_glibc\": %ld, "
         "\"geometry_differs_from_glibc\": %ld, \"differs_from_double_rounded\": %ld}\n",
         n, stride, valdiff, geodiff, crdiff);
  return 0;
}
