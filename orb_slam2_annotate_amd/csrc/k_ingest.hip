// k_ingest.hip -- the per-pixel / per-map-point loops right next to the hot path (SURVEY.md 8(f)
// ranks 3-4): colour -> gray conversion in front of the extractor (Tracking::GrabImage*,
// src/Tracking.cc:176-262) and MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:269-333).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "kernels.h"
#include "orb_spec.h"

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


// ---------------------------------------------------------------------------------------------
// cv::remap(src, dst, M1 CV_32F, M2 CV_32F, INTER_LINEAR) -- the EuRoC stereo rectification in
// front of the extractor (Examples/Stereo/stereo_euroc.cc:97-98 builds the maps once, :136-137
// applies them to every frame).  The float maps are converted once, at create time, to what
// cv::remap converts them to for every tile of every frame: integer source coordinate saturated
// to int16 (packed sx | sy << 16) and a 10-bit sub-pixel phase (fy*32 + fx); the per-frame kernel
// then moves 6 map bytes + 1 source byte + 1 output byte per pixel.
// ---------------------------------------------------------------------------------------------
struct orbfe_rectifier {
  int device = 0;
  int width = 0, height = 0;  // destination size = map size
  uint32_t* xy = nullptr;     // [height][pitch]
  uint16_t* phase = nullptr;  // [height][pitch]
  int pitch = 0;              // multiple of 4
};

namespace {

__device__ __forceinline__ int cvround_sse(float v) {  // cvtss2si: out of range / NaN -> INT_MIN
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
  return __float2int_rn(v);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

__global__ __launch_bounds__(256) void k_remap_convert(const float* __restrict__ mx, const float* __restrict__ my,
                                                       int mstride, int w, int h, int pitch,
                                                       uint32_t* __restrict__ xy, uint16_t* __restrict__ phase) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= pitch) return;
  uint32_t c = 0x80008000u;  // padding columns: (-32768, -32768) -> border
  uint16_t p = 0;
  if (x < w) {
    const int isx = cvround_sse(mx[(size_t)y * mstride + x] * 32.0f), isy = cvround_sse(my[(size_t)y * mstride + x] * 32.0f);
    c = ((uint32_t)sat_short(isx >> 5) & 0xffffu) | ((uint32_t)sat_short(isy >> 5) << 16);
    p = (uint16_t)((isy & 31) * 32 + (isx & 31));
  }
  xy[(size_t)y * pitch + x] = c;
  phase[(size_t)y * pitch + x] = p;
}

// 4 output pixels per thread (one dword store); block = 64 x 4 threads = 256 x 4 pixels.  Blocks of
// one tile over all frames are adjacent in the grid, so the map tile is fetched from HBM once.
__global__ __launch_bounds__(256) void k_remap(const uint32_t* __restrict__ xy, const uint16_t* __restrict__ phase,
                                               int pitch, int w, int h, int tilesX, int nFrames,
                                               const uint8_t* __restrict__ src, int sw, int sh, int sstride,
                                               size_t sFrame, uint8_t* __restrict__ dst, int dstride, size_t dFrame) {
  const int f = blockIdx.x % nFrames, tile = blockIdx.x / nFrames;
  const int x0 = ((tile % tilesX) * 64 + (threadIdx.x & 63)) * 4, y = (tile / tilesX) * 4 + (threadIdx.x >> 6);
  if (x0 >= w || y >= h) return;
  const uint4 c4 = *reinterpret_cast<const uint4*>(xy + (size_t)y * pitch + x0);
  const uint2 p4 = *reinterpret_cast<const uint2*>(phase + (size_t)y * pitch + x0);
  const uint32_t cs[4] = {c4.x, c4.y, c4.z, c4.w};
  const uint32_t ps[4] = {p4.x & 0xffffu, p4.x >> 16, p4.y & 0xffffu, p4.y >> 16};
  const uint8_t* S0 = src + (size_t)f * sFrame;
  const unsigned width1 = (unsigned)(sw - 1 > 0 ? sw - 1 : 0), height1 = (unsigned)(sh - 1 > 0 ? sh - 1 : 0);
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int sx = (int)(int16_t)(cs[k] & 0xffffu), sy = (int)(int16_t)(cs[k] >> 16);
    const int fx = (int)(ps[k] & 31u), fy = (int)(ps[k] >> 5);
    int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
      const uint8_t* S = S0 + (size_t)sy * sstride + sx;
      v0 = S[0]; v1 = S[1]; v2 = S[sstride]; v3 = S[sstride + 1];
    } else if (!(sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
      const bool x0in = sx >= 0 && sx < sw, x1in = sx + 1 >= 0 && sx + 1 < sw;
      const bool y0in = sy >= 0 && sy < sh, y1in = sy + 1 >= 0 && sy + 1 < sh;
      if (x0in && y0in) v0 = S0[(size_t)sy * sstride + sx];
      if (x1in && y0in) v1 = S0[(size_t)sy * sstride + sx + 1];
      if (x0in && y1in) v2 = S0[(size_t)(sy + 1) * sstride + sx];
      if (x1in && y1in) v3 = S0[(size_t)(sy + 1) * sstride + sx + 1];
    }
    // sum of v*w with w = (32-fy)(32-fx)*32 ...; (32 t + 2^14) >> 15 == (t + 512) >> 10
    const int t = (v0 * (32 - fx) + v1 * fx) * (32 - fy) + (v2 * (32 - fx) + v3 * fx) * fy;
    packed |= (uint32_t)((t + 512) >> 10) << (8 * k);
  }
  uint8_t* d = dst + (size_t)f * dFrame + (size_t)y * dstride + x0;
  if (x0 + 3 < w && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(d) = packed;
  } else {
    for (int k = 0; k < 4 && x0 + k < w; k++) d[k] = (uint8_t)(packed >> (8 * k));
  }
}

int remap_launch(orbfe_rectifier* r, const uint8_t* d_src, int n_frames, int sw, int sh, int sstride, size_t sFrame,
                 uint8_t* d_dst, int dstride, size_t dFrame, hipStream_t stream) {
  const int tilesX = (r->width + 255) / 256, tilesY = (r->height + 3) / 4;
  hipLaunchKernelGGL(k_remap, dim3((unsigned)tilesX * tilesY * n_frames), dim3(256), 0, stream, r->xy, r->phase, r->pitch,
                     r->width, r->height, tilesX, n_frames, d_src, sw, sh, sstride, sFrame, d_dst, dstride, dFrame);
  IHIP(hipGetLastError());
  return ORBFE_OK;
}

}  // namespace

// internal (extractor.hip): the rectification of a sub-batch enqueued on that sub-batch's own stream
extern "C" int orbfe_remap_launch_(orbfe_rectifier* r, const uint8_t* d_src, int n_frames, int sw, int sh, int sstride,
                                   size_t sFrame, uint8_t* d_dst, int dstride, size_t dFrame, hipStream_t stream, int* w, int* h,
                                   int* device) {
  if (!r) return ifail(ORBFE_ERR_INVALID, "NULL rectifier");
  if (w) *w = r->width;
  if (h) *h = r->height;
  if (device) *device = r->device;
  if (!d_src || n_frames <= 0) return ORBFE_OK;  // query only
  return remap_launch(r, d_src, n_frames, sw, sh, sstride, sFrame, d_dst, dstride, dFrame, stream);
}

extern "C" int orbfe_rectifier_create(int device, const float* map_x, const float* map_y, int width, int height,
                                  int map_stride, orbfe_rectifier** out) {
  if (!map_x || !map_y || !out || width <= 0 || height <= 0 || map_stride < width || width > 32767 || height > 32767)
    return ifail(ORBFE_ERR_INVALID, "remap_create: bad argument");
  IHIP(hipSetDevice(device));
  orbfe_rectifier* r = new orbfe_rectifier();
  r->device = device; r->width = width; r->height = height; r->pitch = (width + 3) & ~3;
  float *dmx = nullptr, *dmy = nullptr;
  const size_t mb = (size_t)width * height * 4, n = (size_t)r->pitch * height;
  hipError_t err = hipMalloc((void**)&r->xy, n * 4);
  if (err == hipSuccess) err = hipMalloc((void**)&r->phase, n * 2);
  if (err == hipSuccess) err = hipMalloc((void**)&dmx, mb);
  if (err == hipSuccess) err = hipMalloc((void**)&dmy, mb);
  if (err == hipSuccess) err = hipMemcpy2D(dmx, (size_t)width * 4, map_x, (size_t)map_stride * 4, (size_t)width * 4, height, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy2D(dmy, (size_t)width * 4, map_y, (size_t)map_stride * 4, (size_t)width * 4, height, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_remap_convert, dim3((r->pitch + 255) / 256, height), dim3(256), 0, 0, dmx, dmy, width, width,
                       height, r->pitch, r->xy, r->phase);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (dmx) (void)hipFree(dmx);
  if (dmy) (void)hipFree(dmy);
  if (err != hipSuccess) {
    if (r->xy) (void)hipFree(r->xy);
    if (r->phase) (void)hipFree(r->phase);
    delete r;
    return ifail(err == hipErrorOutOfMemory ? ORBFE_ERR_NOMEM : ORBFE_ERR_HIP, std::string("remap_create: ") + hipGetErrorString(err));
  }
  *out = r;
  return ORBFE_OK;
}

extern "C" void orbfe_rectifier_destroy(orbfe_rectifier* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  (void)hipFree(r->xy);
  (void)hipFree(r->phase);
  delete r;
}

extern "C" int orbfe_remap(orbfe_rectifier* r, const uint8_t* src, int src_width, int src_height, int src_stride,
                           uint8_t* dst, int dst_stride) {
  if (!r || !src || !dst || src_width <= 0 || src_height <= 0 || src_stride < src_width || dst_stride < r->width)
    return ifail(ORBFE_ERR_INVALID, "remap: bad argument");
  IHIP(hipSetDevice(r->device));
  uint8_t *ds = nullptr, *dd = nullptr;
  IHIP(hipMalloc((void**)&ds, (size_t)src_width * src_height));
  hipError_t err = hipMalloc((void**)&dd, (size_t)r->pitch * r->height);
  if (err == hipSuccess) err = hipMemcpy2D(ds, src_width, src, src_stride, src_width, src_height, hipMemcpyHostToDevice);
  int rc = ORBFE_OK;
  if (err == hipSuccess) rc = remap_launch(r, ds, 1, src_width, src_height, src_width, 0, dd, r->pitch, 0, 0);
  if (err == hipSuccess && rc == ORBFE_OK) err = hipDeviceSynchronize();
  if (err == hipSuccess && rc == ORBFE_OK) err = hipMemcpy2D(dst, dst_stride, dd, r->pitch, r->width, r->height, hipMemcpyDeviceToHost);
  (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("remap: ") + hipGetErrorString(err));
  return rc;
}

extern "C" int orbfe_remap_batch_device(orbfe_rectifier* r, const uint8_t* d_src, int n_frames, int src_width,
                                        int src_height, int src_stride, size_t src_frame_stride, uint8_t* d_dst,
                                        int dst_stride, size_t dst_frame_stride) {
  if (!r || !d_src || !d_dst || n_frames < 0 || src_width <= 0 || src_height <= 0 || src_stride < src_width ||
      dst_stride < r->width)
    return ifail(ORBFE_ERR_INVALID, "remap_batch_device: bad argument");
  if (n_frames == 0) return ORBFE_OK;
  IHIP(hipSetDevice(r->device));
  const int rc = remap_launch(r, d_src, n_frames, src_width, src_height, src_stride, src_frame_stride, d_dst, dst_stride,
                              dst_frame_stride, 0);
  if (rc != ORBFE_OK) return rc;
  IHIP(hipDeviceSynchronize());
  return ORBFE_OK;
}


// ---------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints / ComputeImageBounds (cv::undistortPoints with P = K,
// src/Frame.cc:443-510) and Frame::ComputeStereoFromRGBD (src/Frame.cc:689-713): the per-keypoint
// loops between the extractor and the matchers.  Double precision, no FMA contraction (the
// library is built with -ffp-contract=off), IEEE division: bit-identical to the host arithmetic.
// ---------------------------------------------------------------------------------------------
namespace {

struct UndistortParams {
  double fx, fy, cx, cy, ifx, ify;
  double k[8];
  int iters;
};

__device__ __forceinline__ void undistort_one(const UndistortParams& p, float xin, float yin, float* xo, float* yo) {
  double x = (double)xin, y = (double)yin, x0, y0;
  x0 = x = (x - p.cx) * p.ifx;
  y0 = y = (y - p.cy) * p.ify;
  for (int j = 0; j < p.iters; j++) {
    const double r2 = x * x + y * y;
    const double icdist =
        (1 + ((p.k[7] * r2 + p.k[6]) * r2 + p.k[5]) * r2) / (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2);
    const double deltaX = 2 * p.k[2] * x * y + p.k[3] * (r2 + 2 * x * x);
    const double deltaY = p.k[2] * (r2 + 2 * y * y) + 2 * p.k[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = p.fx * x + 0.0 * y + p.cx;  // RR = P * I = K
  const double yy = 0.0 * x + p.fy * y + p.cy;
  const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort_points(UndistortParams p, const float* __restrict__ xy, int n,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  undistort_one(p, xy[2 * i], xy[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

// mvKeysUn for a device-resident extractor batch: 28-byte records copied, pt replaced
__global__ __launch_bounds__(256) void k_undistort_keypoints(UndistortParams p, const float* __restrict__ kp,
                                                             const int32_t* __restrict__ nKp, int capacity,
                                                             float* __restrict__ out) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nKp[f] || i >= capacity) return;
  const float* s = kp + ((size_t)f * capacity + i) * 7;
  float* d = out + ((size_t)f * capacity + i) * 7;
  float x, y;
  undistort_one(p, s[0], s[1], &x, &y);
  d[0] = x; d[1] = y;
#pragma unroll
  for (int k = 2; k < 7; k++) d[k] = s[k];
}

__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const float* __restrict__ kx, const float* __restrict__ ky,
                                                          const float* __restrict__ kux, int n,
                                                          const float* __restrict__ depthImg, int stride, float mbf,
                                                          float* __restrict__ uRight, float* __restrict__ depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float d = depthImg[(size_t)(int)ky[i] * stride + (int)kx[i]];
  float ur = -1.0f, dd = -1.0f;
  if (d > 0) { dd = d; ur = kux[i] - mbf / d; }
  uRight[i] = ur;
  depth[i] = dd;
}

bool undistort_params(const float* K4, const float* dist, int n_dist, UndistortParams* p) {
  if (!K4 || n_dist < 0 || n_dist > 8 || (n_dist > 0 && !dist) || !(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8))
    return false;
  p->fx = (double)K4[0]; p->fy = (double)K4[1]; p->cx = (double)K4[2]; p->cy = (double)K4[3];
  p->ifx = 1. / p->fx; p->ify = 1. / p->fy;
  for (int i = 0; i < 8; i++) p->k[i] = i < n_dist ? (double)dist[i] : 0.0;
  p->iters = n_dist > 0 ? 5 : 1;
  return true;
}

}  // namespace

// cv::initUndistortRectifyMap(K, D, R, P[:3,:3], size, CV_32F, M1, M2) as Examples/Stereo/stereo_euroc.cc:97-98 calls it,
// once at start-up: the two float maps orbfe_rectifier_create takes.  One THREAD per map row -- the reference walks a row
// with running sums (_x += ir[0], ...), so a row is a sequential chain in double precision; rows are independent.  No FMA
// contraction (the build's -ffp-contract=off), IEEE division.
namespace {
struct RectifyMapParams { double ir[9], fx, fy, u0, v0, k1, k2, p1, p2, k3, k4, k5, k6; int w, h; };
__global__ __launch_bounds__(64) void k_init_rectify_map(const RectifyMapParams p, float* __restrict__ mapX, float* __restrict__ mapY) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= p.h) return;
  double _x = __dadd_rn(__dmul_rn((double)i, p.ir[1]), p.ir[2]), _y = __dadd_rn(__dmul_rn((double)i, p.ir[4]), p.ir[5]),
         _w = __dadd_rn(__dmul_rn((double)i, p.ir[7]), p.ir[8]);
  float* mx = mapX + (size_t)i * p.w;
  float* my = mapY + (size_t)i * p.w;
  for (int j = 0; j < p.w; j++) {
    const double ww = __ddiv_rn(1.0, _w), x = __dmul_rn(_x, ww), y = __dmul_rn(_y, ww);
    const double x2 = __dmul_rn(x, x), y2 = __dmul_rn(y, y);
    const double r2 = __dadd_rn(x2, y2), _2xy = __dmul_rn(__dmul_rn(2.0, x), y);
    const double num = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(p.k3, r2), p.k2), r2), p.k1), r2));
    const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(p.k6, r2), p.k5), r2), p.k4), r2));
    const double kr = __ddiv_rn(num, den);
    const double xd = __dadd_rn(__dadd_rn(__dmul_rn(x, kr), __dmul_rn(p.p1, _2xy)), __dmul_rn(p.p2, __dadd_rn(r2, __dmul_rn(2.0, x2))));
    const double yd = __dadd_rn(__dadd_rn(__dmul_rn(y, kr), __dmul_rn(p.p1, __dadd_rn(r2, __dmul_rn(2.0, y2)))), __dmul_rn(p.p2, _2xy));
    mx[j] = (float)__dadd_rn(__dmul_rn(p.fx, xd), p.u0);
    my[j] = (float)__dadd_rn(__dmul_rn(p.fy, yd), p.v0);
    _x = __dadd_rn(_x, p.ir[0]); _y = __dadd_rn(_y, p.ir[3]); _w = __dadd_rn(_w, p.ir[6]);
  }
}
}  // namespace

extern "C" int orbfe_init_undistort_rectify_map(int device, const double* K, const double* D, int n_dist, const double* R,
                                                const double* P, int width, int height, float* map_x, float* map_y) {
  if (!K || width <= 0 || height <= 0 || !map_x || !map_y || !(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8) ||
      (n_dist > 0 && !D))
    return ifail(ORBFE_ERR_INVALID, "init_undistort_rectify_map: bad argument (K, the maps and 0 / 4 / 5 / 8 distortion coefficients are required)");
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double *Rm = R ? R : I3, *Ar = P ? P : K;
  // iR = (P * R)^-1: nine dot products and cv::invert's closed 3 x 3 form -- a dozen double operations of set-up, evaluated
  // here in the order the reference evaluates them (host: this translation unit is built with -ffp-contract=off)
  double M[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += Ar[3 * i + k] * Rm[3 * k + j];
      M[3 * i + j] = acc;
    }
  double d = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (d == 0) return ifail(ORBFE_ERR_INVALID, "init_undistort_rectify_map: P * R is singular");
  d = 1. / d;
  RectifyMapParams p;
  p.ir[0] = (M[4] * M[8] - M[5] * M[7]) * d; p.ir[1] = (M[2] * M[7] - M[1] * M[8]) * d; p.ir[2] = (M[1] * M[5] - M[2] * M[4]) * d;
  p.ir[3] = (M[5] * M[6] - M[3] * M[8]) * d; p.ir[4] = (M[0] * M[8] - M[2] * M[6]) * d; p.ir[5] = (M[2] * M[3] - M[0] * M[5]) * d;
  p.ir[6] = (M[3] * M[7] - M[4] * M[6]) * d; p.ir[7] = (M[1] * M[6] - M[0] * M[7]) * d; p.ir[8] = (M[0] * M[4] - M[1] * M[3]) * d;
  p.fx = K[0]; p.fy = K[4]; p.u0 = K[2]; p.v0 = K[5];
  p.k1 = n_dist > 0 ? D[0] : 0; p.k2 = n_dist > 1 ? D[1] : 0; p.p1 = n_dist > 2 ? D[2] : 0; p.p2 = n_dist > 3 ? D[3] : 0;
  p.k3 = n_dist >= 5 ? D[4] : 0; p.k4 = n_dist >= 8 ? D[5] : 0; p.k5 = n_dist >= 8 ? D[6] : 0; p.k6 = n_dist >= 8 ? D[7] : 0;
  p.w = width; p.h = height;
  IHIP(hipSetDevice(device));
  float* dmap = nullptr;
  const size_t n = (size_t)width * height;
  IHIP(hipMalloc((void**)&dmap, 2 * n * 4));
  hipLaunchKernelGGL(k_init_rectify_map, dim3((height + 63) / 64), dim3(64), 0, 0, p, dmap, dmap + n);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipMemcpy(map_x, dmap, n * 4, hipMemcpyDeviceToHost);
  if (err == hipSuccess) err = hipMemcpy(map_y, dmap + n, n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dmap);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("init_undistort_rectify_map: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

extern "C" int orbfe_undistort_points(int device, const float* xy, int n, const float* K4, const float* dist,
                                      int n_dist, float* out_xy) {
  UndistortParams p;
  if (n < 0 || (n > 0 && (!xy || !out_xy)) || !undistort_params(K4, dist, n_dist, &p))
    return ifail(ORBFE_ERR_INVALID, "undistort_points: bad argument");
  if (n == 0) return ORBFE_OK;
  IHIP(hipSetDevice(device));
  float *din = nullptr, *dout = nullptr;
  IHIP(hipMalloc((void**)&din, (size_t)n * 8));
  hipError_t err = hipMalloc((void**)&dout, (size_t)n * 8);
  if (err == hipSuccess) err = hipMemcpy(din, xy, (size_t)n * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256), dim3(256), 0, 0, p, din, n, dout);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpy(out_xy, dout, (size_t)n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(din);
  if (dout) (void)hipFree(dout);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("undistort_points: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

extern "C" int orbfe_undistort_keypoints_batch_device(int device, const orbfe_keypoint* d_keypoints,
                                                      const int32_t* d_n, int n_frames, int capacity,
                                                      const float* K4, const float* dist, int n_dist,
                                                      orbfe_keypoint* d_keypoints_un) {
  UndistortParams p;
  if (!d_keypoints || !d_n || !d_keypoints_un || n_frames < 0 || capacity <= 0 || !undistort_params(K4, dist, n_dist, &p))
    return ifail(ORBFE_ERR_INVALID, "undistort_keypoints_batch_device: bad argument");
  if (n_frames == 0) return ORBFE_OK;
  IHIP(hipSetDevice(device));
  hipLaunchKernelGGL(k_undistort_keypoints, dim3((capacity + 255) / 256, n_frames), dim3(256), 0, 0, p,
                     reinterpret_cast<const float*>(d_keypoints), d_n, capacity, reinterpret_cast<float*>(d_keypoints_un));
  IHIP(hipGetLastError());
  IHIP(hipDeviceSynchronize());
  return ORBFE_OK;
}

extern "C" int orbfe_compute_image_bounds(int device, int cols, int rows, const float* K4, const float* dist,
                                          int n_dist, float* bounds4) {
  if (!bounds4 || cols <= 0 || rows <= 0) return ifail(ORBFE_ERR_INVALID, "compute_image_bounds: bad argument");
  if (n_dist > 0 && dist && dist[0] != 0.0) {
    const float in[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    float out[8];
    const int rc = orbfe_undistort_points(device, in, 4, K4, dist, n_dist, out);
    if (rc != ORBFE_OK) return rc;
    bounds4[0] = out[4] < out[0] ? out[4] : out[0];  // min(mat(0,0), mat(2,0)), src/Frame.cc:496-499
    bounds4[1] = out[2] < out[6] ? out[6] : out[2];
    bounds4[2] = out[3] < out[1] ? out[3] : out[1];
    bounds4[3] = out[5] < out[7] ? out[7] : out[5];
  } else {
    bounds4[0] = 0.0f; bounds4[1] = (float)cols; bounds4[2] = 0.0f; bounds4[3] = (float)rows;
  }
  return ORBFE_OK;
}

extern "C" int orbfe_stereo_from_rgbd(int device, const float* kx, const float* ky, const float* kux, int n,
                                      const float* depth_image, int width, int height, int stride_floats, float mbf,
                                      float* u_right, float* depth) {
  if (n < 0 || width <= 0 || height <= 0 || stride_floats < width || !depth_image ||
      (n > 0 && (!kx || !ky || !kux || !u_right || !depth)))
    return ifail(ORBFE_ERR_INVALID, "stereo_from_rgbd: bad argument");
  for (int i = 0; i < n; i++)
    if (!(kx[i] >= 0 && ky[i] >= 0 && (int)kx[i] < width && (int)ky[i] < height))
      return ifail(ORBFE_ERR_INVALID, "stereo_from_rgbd: keypoint outside the depth image");
  if (n == 0) return ORBFE_OK;
  IHIP(hipSetDevice(device));
  float* buf = nullptr;
  const size_t img = (size_t)width * height;
  IHIP(hipMalloc((void**)&buf, (img + 5 * (size_t)n) * 4));
  float *dimg = buf, *dkx = buf + img, *dky = dkx + n, *dkux = dky + n, *dur = dkux + n, *ddp = dur + n;
  hipError_t err = hipMemcpy2D(dimg, (size_t)width * 4, depth_image, (size_t)stride_floats * 4, (size_t)width * 4, height, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dkx, kx, (size_t)n * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dky, ky, (size_t)n * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dkux, kux, (size_t)n * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((n + 255) / 256), dim3(256), 0, 0, dkx, dky, dkux, n, dimg, width, mbf, dur, ddp);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpy(u_right, dur, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (err == hipSuccess) err = hipMemcpy(depth, ddp, (size_t)n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (err != hipSuccess) return ifail(ORBFE_ERR_HIP, std::string("stereo_from_rgbd: ") + hipGetErrorString(err));
  return ORBFE_OK;
}
