// k_blur.hip -- cv::GaussianBlur(7x7, sigma=2, BORDER_REFLECT_101) on a u8 level
// (the `workingMat` of src/ORBextractor.cc:1169-1175), OpenCV's bit-exact fixed-point path:
// separable kernel [18,34,48,56,48,34,18]/256, horizontal pass exact in 8.8, vertical pass
// in 16.16, one rounding (x + 2^15) >> 16.  HBM-bound stencil: each 256-thread workgroup
// stages a (64+6)x(16+6) tile in LDS (one read of every source byte plus the 3-px halo),
// keeps the horizontal sums in LDS and writes 64x16 outputs as 32-bit coalesced stores.
#include "kernels.h"

namespace orbfe {

namespace {
constexpr int kBW = 64, kBH = 16, kInPitch = 72;
__device__ __forceinline__ int reflect101c(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);  // clamp only matters for never-used tile cells
}
}  // namespace

__global__ __launch_bounds__(256) void k_blur7(LevelView src, LevelViewMut dst) {
  __shared__ uint8_t tin[(kBH + 6) * kInPitch];
  __shared__ uint16_t hb[(kBH + 6) * kBW];
  const int tid = threadIdx.x;
  const int bx = blockIdx.x * kBW, by = blockIdx.y * kBH, f = blockIdx.z;
  const uint8_t* S = src.base + (size_t)f * src.frameStride;
  for (int i = tid; i < (kBH + 6) * (kBW + 6); i += 256) {
    const int ty = i / (kBW + 6), tx = i - ty * (kBW + 6);
    const int sx = reflect101c(bx - 3 + tx, src.w), sy = reflect101c(by - 3 + ty, src.h);
    tin[ty * kInPitch + tx] = S[(size_t)sy * src.pitch + sx];
  }
  __syncthreads();
  for (int i = tid; i < (kBH + 6) * kBW; i += 256) {
    const int r = i >> 6, c = i & 63;
    const uint8_t* t = &tin[r * kInPitch + c];
    hb[i] = (uint16_t)(18 * (t[0] + t[6]) + 34 * (t[1] + t[5]) + 48 * (t[2] + t[4]) + 56 * t[3]);
  }
  __syncthreads();
  const int row = tid >> 4, cg = (tid & 15) * 4;
  const int y = by + row, x = bx + cg;
  if (y >= dst.h || x >= dst.w) return;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint16_t* h = &hb[row * kBW + cg + k];
    const uint32_t acc = 18u * ((uint32_t)h[0] + h[6 * kBW]) + 34u * ((uint32_t)h[kBW] + h[5 * kBW]) +
                         48u * ((uint32_t)h[2 * kBW] + h[4 * kBW]) + 56u * (uint32_t)h[3 * kBW];
    packed |= ((acc + (1u << 15)) >> 16) << (8 * k);
  }
  *reinterpret_cast<uint32_t*>(dst.base + (size_t)f * dst.frameStride + (size_t)y * dst.pitch + x) = packed;
}

void launch_blur7(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames) {
  if (dst.w <= 0 || dst.h <= 0 || nFrames <= 0) return;
  dim3 grid((dst.w + kBW - 1) / kBW, (dst.h + kBH - 1) / kBH, nFrames);
  hipLaunchKernelGGL(k_blur7, grid, dim3(256), 0, s, src, dst);
}

}  // namespace orbfe
