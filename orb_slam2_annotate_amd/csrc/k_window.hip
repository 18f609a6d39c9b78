// k_window.hip -- Frame grid + window search (Frame::AssignFeaturesToGrid / PosInGrid /
// GetFeaturesInArea, src/Frame.cc:246-267, 358-427) fused with the Hamming distances the
// projection searches of ORBmatcher take over every window (src/ORBmatcher.cc:51-138, 1484-1633).
//
// Layout: the 64x48 grid of the reference (mGrid[ix][iy], a vector of ascending feature indices
// per cell) is one sorted array of keys (cell << 16 | feature index), cell = ix*48 + iy, plus
// 3073 cell offsets.  With that cell order the cells [minY..maxY] of one grid column are ONE
// contiguous span of the array, and walking the columns left to right visits the features in
// exactly the reference's scan order (ix outer, iy inner, index ascending inside a cell).
// HBM-light integer work: a 64-lane wavefront owns one query, takes a span 64 entries at a
// time, and a ballot keeps the survivors in scan order.
#include "kernels.h"
#include "match_kernels.h"

namespace orbfe {

namespace {

constexpr int GRID_COLS = 64, GRID_ROWS = 48, GRID_CELLS = GRID_COLS * GRID_ROWS;
constexpr uint32_t KEY_NONE = 0xffffffffu;

// One block per frame; keys sorted in LDS (bitonic, sortN = power of two >= n, <= 16384).
__global__ __launch_bounds__(1024) void k_grid_build(GridFrame f, int sortN, uint32_t* __restrict__ sortedKey,
                                                     int32_t* __restrict__ cellOff) {
  extern __shared__ uint32_t keys[];
  const int t = threadIdx.x;
  for (int i = t; i < sortN; i += 1024) {
    uint32_t k = KEY_NONE;
    if (i < f.n) {
      // PosInGrid, src/Frame.cc:417-427: round() half away from zero
      const int px = (int)roundf((f.x[i] - f.minX) * f.wInv);
      const int py = (int)roundf((f.y[i] - f.minY) * f.hInv);
      if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) k = ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= sortN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < sortN; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const uint32_t a = keys[i], b = keys[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = t; i < f.n; i += 1024) sortedKey[i] = keys[i];
  // cellOff[c] = first position whose key >= c << 16 (c == GRID_CELLS -> number of gridded features)
  for (int c = t; c <= GRID_CELLS; c += 1024) {
    const uint32_t want = (uint32_t)c << 16;
    int lo = 0, hi = f.n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    cellOff[c] = lo;
  }
}

// One wavefront per query, 4 per block.
__global__ __launch_bounds__(256) void k_window_search(GridFrame f, const uint32_t* __restrict__ sortedKey,
                                                       const int32_t* __restrict__ cellOff, WindowQueries q,
                                                       int32_t* __restrict__ count, uint32_t* __restrict__ cand) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= q.n) return;
  const float x = q.x[qi], y = q.y[qi], r = q.r[qi];
  const int minLevel = q.minLevel[qi], maxLevel = q.maxLevel[qi];
  // GetFeaturesInArea, src/Frame.cc:363-381
  int nMinCellX = (int)floorf((x - f.minX - r) * f.wInv);
  if (nMinCellX < 0) nMinCellX = 0;
  int nMaxCellX = (int)ceilf((x - f.minX + r) * f.wInv);
  if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
  int nMinCellY = (int)floorf((y - f.minY - r) * f.hInv);
  if (nMinCellY < 0) nMinCellY = 0;
  int nMaxCellY = (int)ceilf((y - f.minY + r) * f.hInv);
  if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
  const bool empty = nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0 ||
                     (q.active && !q.active[qi]);
  int n = 0;
  uint32_t bestKey = 0xffffffffu, bestId = 0;
  if (!empty) {
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    uint32_t qd[8];
    if (q.desc) {
      const uint4* p = reinterpret_cast<const uint4*>(q.desc + (size_t)qi * 32);
      const uint4 a = p[0], b = p[1];
      qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w; qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
    }
    const float ur = q.ur ? q.ur[qi] : 0.0f;
    const float gur = (q.best && q.gate && q.gateUr) ? q.gateUr[qi] : 0.0f;
    uint32_t* out = cand + (size_t)qi * q.K;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      const int s = cellOff[ix * GRID_ROWS + nMinCellY], e = cellOff[ix * GRID_ROWS + nMaxCellY + 1];
      for (int base = s; base < e; base += 64) {
        const int j = base + lane;
        bool ok = j < e;
        uint32_t id = 0;
        if (ok) {
          id = sortedKey[j] & 0xffffu;
          if (bCheckLevels) {
            const int o = f.octave[id];
            if (o < minLevel) ok = false;
            if (maxLevel >= 0 && o > maxLevel) ok = false;
          }
          const float distx = f.x[id] - x, disty = f.y[id] - y;
          if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
          // stereo consistency of the projection searches (src/ORBmatcher.cc:91-96, 1560-1566)
          if (ok && q.ur && f.uRight) {
            const float u2 = f.uRight[id];
            if (u2 > 0 && fabsf(ur - u2) > r) ok = false;
          }
        }
        const unsigned long long m = __ballot(ok);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (q.best) {  // wave-uniform
          if (ok) {
            bool pass = true;
            if (q.gate) {  // Fuse: src/ORBmatcher.cc:1036-1058 -- fp32 like the reference, the compare in double
              const float kpx = f.x[id], kpy = f.y[id];
              const float inv = q.invSigma2[f.octave[id]];
              const float ex = __fsub_rn(x, kpx), ey = __fsub_rn(y, kpy);
              const float u2 = f.uRight ? f.uRight[id] : -1.0f;
              if (u2 >= 0) {
                const float er = __fsub_rn(gur, u2);
                const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                if ((double)__fmul_rn(e2, inv) > 7.8) pass = false;
              } else {
                const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                if ((double)__fmul_rn(e2, inv) > 5.99) pass = false;
              }
            }
            if (pass) {
              const uint4* p = reinterpret_cast<const uint4*>(f.desc + (size_t)id * 32);
              const uint4 a = p[0], b = p[1];
              const uint32_t dist = __popc(qd[0] ^ a.x) + __popc(qd[1] ^ a.y) + __popc(qd[2] ^ a.z) + __popc(qd[3] ^ a.w) +
                                    __popc(qd[4] ^ b.x) + __popc(qd[5] ^ b.y) + __popc(qd[6] ^ b.z) + __popc(qd[7] ^ b.w);
              const uint32_t key = (dist << 16) | (uint32_t)pos;  // pos < 16384: first minimum in scan order = smallest key
              if (key < bestKey) { bestKey = key; bestId = id; }
            }
          }
        } else if (ok && pos < q.K) {
          uint32_t dist = 0;
          if (q.desc) {
            const uint4* p = reinterpret_cast<const uint4*>(f.desc + (size_t)id * 32);
            const uint4 a = p[0], b = p[1];
            dist = __popc(qd[0] ^ a.x) + __popc(qd[1] ^ a.y) + __popc(qd[2] ^ a.z) + __popc(qd[3] ^ a.w) +
                   __popc(qd[4] ^ b.x) + __popc(qd[5] ^ b.y) + __popc(qd[6] ^ b.z) + __popc(qd[7] ^ b.w);
          }
          out[pos] = (dist << 16) | id;
        }
        n += __popcll(m);
      }
    }
  }
  if (q.best) {
    uint32_t k = bestKey;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = (uint32_t)__shfl_xor((int)k, o, 64);
      k = t < k ? t : k;
    }
    // (keys are unique: they carry the scan position) the owner of the minimum writes its keypoint
    if (k == 0xffffffffu) { if (lane == 0) q.best[qi] = -1; }
    else if (bestKey == k) q.best[qi] = (int)(k >> 16) <= q.maxDist ? (int32_t)bestId : -1;
    return;
  }
  if (lane == 0) count[qi] = n;
}

}  // namespace

void launch_grid_build(hipStream_t s, const GridFrame& f, uint32_t* sortedKey, int32_t* cellOff) {
  int sortN = 64;
  while (sortN < f.n) sortN <<= 1;
  hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(1024), (size_t)sortN * 4, s, f, sortN, sortedKey, cellOff);
}

// orbfe_frame_from_device: the extractor's 28-byte cv::KeyPoint records -> the resident frame's arrays, descriptors copied
// device to device.  Thread t < n splits record t; the whole grid moves the n x 32 descriptor bytes as 16-byte pieces.
__global__ __launch_bounds__(256) void k_frame_from_records(const float* __restrict__ kp, const uint8_t* __restrict__ desc, int n,
                                                            float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ angle, int32_t* __restrict__ octave,
                                                            uint8_t* __restrict__ descOut, uint8_t* __restrict__ stereoZero) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const float* r = kp + (size_t)t * 7;
    if (x) x[t] = r[0];
    if (y) y[t] = r[1];
    angle[t] = r[3];
    octave[t] = reinterpret_cast<const int32_t*>(r)[5];
    if (stereoZero) stereoZero[t] = 0;  // monocular frame: no mvuRight
  }
  const uint4* src = reinterpret_cast<const uint4*>(desc);  // (rows of 32 bytes in buffers the library or torch allocated: 16-byte aligned)
  uint4* dst = reinterpret_cast<uint4*>(descOut);
  for (int i = t; i < 2 * n; i += gridDim.x * 256) dst[i] = src[i];
}

void launch_frame_from_records(hipStream_t s, const float* d_kp, const uint8_t* d_desc, int n, float* x, float* y, float* angle,
                               int32_t* octave, uint8_t* descOut, uint8_t* stereoZero) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_frame_from_records, dim3((n + 255) / 256), dim3(256), 0, s, d_kp, d_desc, n, x, y, angle, octave, descOut,
                     stereoZero);
}

void launch_window_search(hipStream_t s, const GridFrame& f, const uint32_t* sortedKey, const int32_t* cellOff,
                          const WindowQueries& q, int32_t* count, uint32_t* cand) {
  if (q.n <= 0) return;
  hipLaunchKernelGGL(k_window_search, dim3((q.n + 3) / 4), dim3(256), 0, s, f, sortedKey, cellOff, q, count, cand);
}

}  // namespace orbfe
