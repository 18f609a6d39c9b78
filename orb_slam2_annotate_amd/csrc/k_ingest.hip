// k_ingest.hip -- the per-pixel / per-map-point loops right next to the hot path (SURVEY.md 8(f)
// ranks 3-4): colour -> gray conversion in front of the extractor (Tracking::GrabImage*,
// src/Tracking.cc:176-262) and MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:269-333).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "kernels.h"

int orbfe_set_error_(int code, const char* msg);
static int ifail(int code, const std::string& m) { return orbfe_set_error_(code, m.c_str()); }
#define IHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {

// cv::cvtColor(CV_RGB2GRAY etc.), 8U fixed point: (R*4899 + G*9617 + B*1868 + 2^13) >> 14.
// 4 output pixels per thread, one 32-bit store; 12 or 16 source bytes read as bytes (any stride).
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, int w, int h, int sstride,
                                                  size_t sFrame, int channels, int rgbOrder,
                                                  uint8_t* __restrict__ dst, int dstride, size_t dFrame) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, f = blockIdx.z;
  if (x0 >= w) return;
  const uint8_t* s = src + (size_t)f * sFrame + (size_t)y * sstride + (size_t)x0 * channels;
  uint8_t* d = dst + (size_t)f * dFrame + (size_t)y * dstride + x0;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (x0 + k < w) {
      const uint8_t* p = s + k * channels;
      const int r = rgbOrder ? p[0] : p[2], g = p[1], b = rgbOrder ? p[2] : p[0];
      packed |= (uint32_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14) << (8 * k);
    }
  }
  if (x0 + 3 < w && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(d) = packed;
  } else {
    for (int k = 0; k < 4 && x0 + k < w; k++) d[k] = (uint8_t)(packed >> (8 * k));
  }
}

// One workgroup per map point: n x n Hamming distances, per row the (size_t)(0.5*(n-1))-th order
// statistic via a 257-bin histogram, then the first row with the least median.
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets,
                                                     int32_t* __restrict__ best) {
  __shared__ int hist[4][257];
  __shared__ unsigned bestKey;  // median << 16 | row  (first minimum = smallest key)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = blockIdx.x;
  const int b = offsets[m], n = offsets[m + 1] - b;
  if (tid == 0) bestKey = 0xffffffffu;
  __syncthreads();
  if (n <= 0) { if (tid == 0) best[m] = -1; return; }
  const int k = (int)(0.5 * (n - 1));
  for (int i = wave; i < n; i += 4) {  // one wave per row
    for (int t = lane; t < 257; t += 64) hist[wave][t] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint4* pi = reinterpret_cast<const uint4*>(desc + (size_t)(b + i) * 32);
    const uint4 a0 = pi[0], a1 = pi[1];
    for (int j = lane; j < n; j += 64) {
      const uint4* pj = reinterpret_cast<const uint4*>(desc + (size_t)(b + j) * 32);
      const uint4 c0 = pj[0], c1 = pj[1];
      const int d = __popc(a0.x ^ c0.x) + __popc(a0.y ^ c0.y) + __popc(a0.z ^ c0.z) + __popc(a0.w ^ c0.w) +
                    __popc(a1.x ^ c1.x) + __popc(a1.y ^ c1.y) + __popc(a1.z ^ c1.z) + __popc(a1.w ^ c1.w);
      atomicAdd(&hist[wave][d], 1);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      int acc = 0, med = 0;
      for (int t = 0; t < 257; t++) { acc += hist[wave][t]; if (acc > k) { med = t; break; } }
      atomicMin(&bestKey, ((unsigned)med << 16) | (unsigned)i);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if (tid == 0) best[m] = (int32_t)(bestKey & 0xffffu);
}

}  // namespace

extern "C" int orbfe_cvt_gray(int device, const uint8_t* src, int width, int height, int stride, int channels,
                              int rgb_order, uint8_t* dst, int dst_stride) {
  if (!src || !dst || width <= 0 || height <= 0 || (channels != 3 && channels != 4) || stride < width * channels ||
      dst_stride < width)
    return ifail(ORBFE_ERR_INVALID, "cvt_gray: bad argument");
  IHIP(hipSetDevice(device));
  uint8_t *ds = nullptr, *dd = nullptr;
  const size_t sb = (size_t)width * channels * height, db = (size_t)width * height;
  IHIP(hipMalloc((void**)&ds, sb));
  hipError_t err = hipMalloc((void**)&dd, db);
  if (err == hipSuccess) err = hipMemcpy2D(ds, (size_t)width * channels, src, stride, (size_t)width * channels, height, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_cvt_gray, dim3((width + 1023) / 1024, height, 1), dim3(256), 0, 0, ds, width, height,
                       width * channels, 0, channels, rgb_order, dd, width, 0);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err == hipSuccess) err = hipMemcpy2D(dst, dst_stride, dd, width, width, height, hipMemcpyDeviceToHost);
  (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("cvt_gray: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

extern "C" int orbfe_cvt_gray_batch_device(int device, const uint8_t* d_src, int n_frames, int width, int height,
                                           int stride, size_t frame_stride, int channels, int rgb_order,
                                           uint8_t* d_dst, int dst_stride, size_t dst_frame_stride) {
  if (!d_src || !d_dst || n_frames < 0 || width <= 0 || height <= 0 || (channels != 3 && channels != 4) ||
      stride < width * channels || dst_stride < width)
    return ifail(ORBFE_ERR_INVALID, "cvt_gray_batch_device: bad argument");
  if (n_frames == 0) return ORBFE_OK;
  IHIP(hipSetDevice(device));
  hipLaunchKernelGGL(k_cvt_gray, dim3((width + 1023) / 1024, height, n_frames), dim3(256), 0, 0, d_src, width, height,
                     stride, frame_stride, channels, rgb_order, d_dst, dst_stride, dst_frame_stride);
  IHIP(hipGetLastError());
  IHIP(hipDeviceSynchronize());
  return ORBFE_OK;
}

extern "C" int orbfe_distinctive_descriptors(int device, const uint8_t* descriptors, const int32_t* offsets,
                                             int n_points, int32_t* best_index) {
  if (n_points < 0 || (n_points > 0 && (!descriptors || !offsets || !best_index)))
    return ifail(ORBFE_ERR_INVALID, "distinctive_descriptors: bad argument");
  if (n_points == 0) return ORBFE_OK;
  for (int i = 0; i < n_points; i++)
    if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 65535)
      return ifail(ORBFE_ERR_INVALID, "distinctive_descriptors: bad offsets (or more than 65535 observations)");
  if (offsets[0] != 0) return ifail(ORBFE_ERR_INVALID, "distinctive_descriptors: offsets[0] != 0");
  const size_t total = (size_t)offsets[n_points];
  IHIP(hipSetDevice(device));
  uint8_t* dd = nullptr; int32_t *doff = nullptr, *dbest = nullptr;
  IHIP(hipMalloc((void**)&dd, total * 32 + 32));
  hipError_t err = hipMalloc((void**)&doff, ((size_t)n_points + 1) * 4);
  if (err == hipSuccess) err = hipMalloc((void**)&dbest, (size_t)n_points * 4);
  if (err == hipSuccess && total) err = hipMemcpy(dd, descriptors, total * 32, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(doff, offsets, ((size_t)n_points + 1) * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(256), 0, 0, dd, doff, dbest);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpy(best_index, dbest, (size_t)n_points * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dd);
  if (doff) (void)hipFree(doff);
  if (dbest) (void)hipFree(dbest);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("distinctive_descriptors: ") + hipGetErrorString(err));
  return ORBFE_OK;
}
