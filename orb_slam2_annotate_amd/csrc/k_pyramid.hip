// k_pyramid.hip -- bilinear pyramid level (cv::resize INTER_LINEAR, src/ORBextractor.cc:1219).
// Integer fixed-point, HBM-bound: every thread produces 4 horizontally adjacent output
// pixels (one 32-bit store, coalesced 256 B per wave row); the four source taps per pixel
// come from two rows ~1.2x denser than the output, i.e. they are served by L1/L2 lines the
// neighbouring lanes touch as well.
#include "kernels.h"

namespace orbfe {

__global__ __launch_bounds__(256) void k_resize(LevelView src, LevelViewMut dst,
                                                const int32_t* __restrict__ xofs,
                                                const int16_t* __restrict__ alpha,
                                                const int32_t* __restrict__ yofs,
                                                const int16_t* __restrict__ beta) {
  const int gx = blockIdx.x * 64 + threadIdx.x;  // group of 4 output columns
  const int dy = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (dy >= dst.h) return;
  const int dx0 = gx * 4;
  if (dx0 >= dst.w) return;
  const int sy = yofs[dy];
  const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
  int r0y = sy < 0 ? 0 : (sy >= src.h ? src.h - 1 : sy);
  int r1y = sy + 1 < 0 ? 0 : (sy + 1 >= src.h ? src.h - 1 : sy + 1);
  const uint8_t* S0 = src.base + (size_t)f * src.frameStride + (size_t)r0y * src.pitch;
  const uint8_t* S1 = src.base + (size_t)f * src.frameStride + (size_t)r1y * src.pitch;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int dx = dx0 + k;
    if (dx < dst.w) {
      const int sx = xofs[dx];
      const int sx1 = sx + 1 < src.w ? sx + 1 : sx;
      const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
      const int h0 = S0[sx] * a0 + S0[sx1] * a1;
      const int h1 = S1[sx] * a0 + S1[sx1] * a1;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      packed |= (uint32_t)(v & 0xff) << (8 * k);
    }
  }
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + (size_t)dy * dst.pitch + dx0;
  *reinterpret_cast<uint32_t*>(D) = packed;  // pitch % 64 == 0, dx0 % 4 == 0: aligned, in-row
}

void launch_resize(hipStream_t s, LevelView src, LevelViewMut dst, const int32_t* d_xofs,
                   const int16_t* d_alpha, const int32_t* d_yofs, const int16_t* d_beta,
                   int nFrames) {
  dim3 block(64, 4);
  dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, nFrames);
  hipLaunchKernelGGL(k_resize, grid, block, 0, s, src, dst, d_xofs, d_alpha, d_yofs, d_beta);
}

// level 0 = copy of the caller's image (the copyMakeBorder of :1231 without the dead border)
__global__ __launch_bounds__(256) void k_copy2d(LevelView src, LevelViewMut dst) {
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int y = blockIdx.y;
  const int f = blockIdx.z;
  if (x >= dst.w) return;
  const uint8_t* S = src.base + (size_t)f * src.frameStride + (size_t)y * src.pitch;
  uint32_t p = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (x + k < dst.w) p |= (uint32_t)S[x + k] << (8 * k);
  *reinterpret_cast<uint32_t*>(dst.base + (size_t)f * dst.frameStride + (size_t)y * dst.pitch + x) = p;
}

void launch_copy2d(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames) {
  dim3 grid((dst.w + 1023) / 1024, dst.h, nFrames);
  hipLaunchKernelGGL(k_copy2d, grid, dim3(256), 0, s, src, dst);
}

}  // namespace orbfe
