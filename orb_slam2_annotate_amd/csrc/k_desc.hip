// k_desc.hip -- per-keypoint tail of ORBextractor::operator(): IC_Angle orientation
// (src/ORBextractor.cc:78-106) on the UNBLURRED level, the 256-bit steered BRIEF descriptor
// (:111-152) on the BLURRED level, and the final cv::KeyPoint record (:905-916,:1187-1195).
// One 64-lane wavefront per keypoint: the 749-pixel disc moments are lane-strided integer
// sums reduced with DPP shuffles (exact, order-free); the 256 binary tests map to
// 4 x 64 lanes and a wave ballot IS the 8 descriptor bytes (bit k of byte i = test 8i+k).
// Gathers hit L2 (a 37x37 neighbourhood per keypoint); output is 28+32 B per keypoint.
#include "kernels.h"

namespace orbfe {

__global__ __launch_bounds__(256) void k_orient_desc(OrientDescArgs a,
                                                     const LevelKp* __restrict__ levelKp,
                                                     const int32_t* __restrict__ levelCount,
                                                     const uint32_t* __restrict__ pattern,
                                                     const int32_t* __restrict__ umax,
                                                     float* __restrict__ kpOut,
                                                     uint8_t* __restrict__ descOut,
                                                     int32_t* __restrict__ nOut) {
  const int lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int f = blockIdx.y;
  if (slot >= a.kpSlotsPerFrame) return;
  const int32_t* cnt = levelCount + (size_t)f * a.nlevels;
  int l = 0, base = 0;
  for (int k = 0; k < a.nlevels; k++) {
    if (slot >= a.kpStart[k]) l = k;
  }
  for (int k = 0; k < l; k++) base += cnt[k];
  if (slot == 0 && lane == 0) {
    int tot = 0;
    for (int k = 0; k < a.nlevels; k++) tot += cnt[k];
    nOut[f] = tot;
  }
  const int i = slot - a.kpStart[l];
  if (i >= cnt[l]) return;
  const int outIdx = base + i;
  if (outIdx >= a.outCapacity) return;  // host reports ORBFE_ERR_CAPACITY from nOut
  const LevelKp kp = levelKp[(size_t)f * a.kpSlotsPerFrame + slot];
  const int x = kp.x, y = kp.y;

  // ---- IC_Angle ----
  const LevelView lv = a.pyr.lv[l];
  const uint8_t* c = lv.base + (size_t)f * lv.frameStride + (size_t)y * lv.pitch + x;
  int m10 = 0, m01 = 0;
  for (int idx = lane; idx < 31 * 31; idx += 64) {
    const int r = idx / 31;
    const int dy = r - 15, dx = idx - r * 31 - 15;
    const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
    if (adx <= umax[ady]) {
      const int I = c[dy * lv.pitch + dx];
      m10 += dx * I;
      m01 += dy * I;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m10 += __shfl_xor(m10, o, 64);
    m01 += __shfl_xor(m01, o, 64);
  }
  const float angle = fast_atan2((float)m01, (float)m10);

  // ---- steered BRIEF ----
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float ca, sb;
  sincos_spec(__fmul_rn(angle, factorPI), &ca, &sb);
  const LevelView bl = a.blur.lv[l];
  const uint8_t* cb = bl.base + (size_t)f * bl.frameStride + (size_t)y * bl.pitch + x;
  unsigned long long* dout = reinterpret_cast<unsigned long long*>(
      descOut + ((size_t)f * a.outCapacity + outIdx) * 32);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t p = pattern[lane + 64 * j];  // (x0,y0,x1,y1) as int8
    const float x0 = (float)(int8_t)(p & 0xff), y0 = (float)(int8_t)((p >> 8) & 0xff);
    const float x1 = (float)(int8_t)((p >> 16) & 0xff), y1 = (float)(int8_t)(p >> 24);
    const int r0 = cv_round(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)));
    const int c0 = cv_round(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)));
    const int r1 = cv_round(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)));
    const int c1 = cv_round(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)));
    const int t0 = cb[r0 * bl.pitch + c0];
    const int t1 = cb[r1 * bl.pitch + c1];
    const unsigned long long bits = __ballot(t0 < t1);
    if (lane == 0) dout[j] = bits;
  }
  if (lane == 0) {
    float* o = kpOut + ((size_t)f * a.outCapacity + outIdx) * 7;
    const float sc = a.scale[l];
    o[0] = __fmul_rn((float)x, sc);
    o[1] = __fmul_rn((float)y, sc);
    o[2] = a.kpSize[l];
    o[3] = angle;
    o[4] = (float)kp.score;
    reinterpret_cast<int32_t*>(o)[5] = l;
    reinterpret_cast<int32_t*>(o)[6] = -1;
  }
}

void launch_orient_desc(hipStream_t s, const OrientDescArgs& a, const LevelKp* d_levelKp,
                        const int32_t* d_levelCount, const uint32_t* d_pattern,
                        const int32_t* d_umax, int nFrames, void* d_kpOut, uint8_t* d_descOut,
                        int32_t* d_nOut) {
  if (nFrames <= 0 || a.kpSlotsPerFrame <= 0) return;
  dim3 grid((a.kpSlotsPerFrame + 3) / 4, nFrames);
  hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, s, a, d_levelKp, d_levelCount, d_pattern,
                     d_umax, (float*)d_kpOut, d_descOut, d_nOut);
}

}  // namespace orbfe
