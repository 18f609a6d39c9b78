// k_fast.hip -- the FAST grid stage of ComputeKeyPointsOctTree (src/ORBextractor.cc:846-896)
// as ONE fused kernel: one 256-thread workgroup per (frame, grid cell) = per cv::FAST call
// of the reference.  The cell's pixels plus the 3-px ring halo are staged in LDS once
// (coalesced row reads from HBM/L2), then
//   A. every pixel gets the 9-of-16 contiguous-arc test at the LOWER threshold; corners are
//      appended to an LDS work queue (dense, so phase B has no idle lanes),
//   B. queue entries get the exact cv::FAST score (cornerScore<16>, S-1),
//   C. 3x3 strict non-max suppression restricted to the cell's detection rectangle
//      (the reference's NMS never sees across a cell boundary),
//   D. per-cell threshold fallback (:874-882): corners >= iniThFAST if any survive NMS,
//      else corners >= minThFAST; survivors are emitted in raster order into the cell's
//      slot range (no atomics on HBM: output position is a pure function of the input).
// Facts used: score = S-1 does not depend on the threshold, and a pixel kept by NMS at
// threshold t is exactly a pixel with score >= t that beats all 8 neighbours' scores.
#include "kernels.h"

namespace orbfe {

namespace {
constexpr int kMaxCell = 60;             // cell side bound: wCell = ceil(width/nCols) < 60
constexpr int kTilePitch = 72;           // >= kMaxCell + 6
constexpr int kTileRows = kMaxCell + 6;
constexpr int kScorePitch = 64;          // >= kMaxCell + 2
constexpr int kScoreRows = kMaxCell + 2;

// Bresenham circle of radius 3, the order cv::FAST uses (any rotation gives the same result).
__device__ constexpr int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__device__ constexpr int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// S = max over the 16 arcs of 9 contiguous ring pixels of min(v - x) resp. min(x - v).
// cv::FAST's response is S-1; the pixel is a corner at threshold t iff S > t.
__device__ __forceinline__ int fast_S(const uint8_t* c) {
  const int v = c[0];
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = v - (int)c[kRingDx[k] + kRingDy[k] * kTilePitch];
  int lo1[16], hi1[16], lo2[16], hi2[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { lo1[k] = imin(d[k], d[(k + 1) & 15]); hi1[k] = imax(d[k], d[(k + 1) & 15]); }
#pragma unroll
  for (int k = 0; k < 16; k++) { lo2[k] = imin(lo1[k], lo1[(k + 2) & 15]); hi2[k] = imax(hi1[k], hi1[(k + 2) & 15]); }
#pragma unroll
  for (int k = 0; k < 16; k++) { lo1[k] = imin(lo2[k], lo2[(k + 4) & 15]); hi1[k] = imax(hi2[k], hi2[(k + 4) & 15]); }
  int sd = -256, sb = 256;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    sd = imax(sd, imin(lo1[k], d[(k + 8) & 15]));
    sb = imin(sb, imax(hi1[k], d[(k + 8) & 15]));
  }
  return imax(sd, -sb);
}

// 9 contiguous set bits in a cyclic 16-bit mask?
__device__ __forceinline__ bool has_arc9(uint32_t m16) {
  uint32_t m = m16 | (m16 << 16);
  uint32_t r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m >> 8;
  return (r & 0xffffu) != 0;
}
}  // namespace

__global__ __launch_bounds__(256) void k_fast_cells(PyramidViews pyr,
                                                    const CellDesc* __restrict__ cells,
                                                    int nCells, int iniTh, int minTh,
                                                    Candidate* __restrict__ slots,
                                                    int slotsPerFrame,
                                                    uint16_t* __restrict__ cellCount) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kTileRows * kTilePitch];  // reused as class map
  __shared__ __attribute__((aligned(16))) uint8_t score[kScoreRows * kScorePitch];
  __shared__ uint16_t queue[kMaxCell * kMaxCell];
  __shared__ int qn, nHigh;
  __shared__ int waveTot[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cellId = blockIdx.x, f = blockIdx.y;
  const CellDesc cd = cells[cellId];
  const LevelView lv = pyr.lv[cd.level];
  const int cw = cd.w, ch = cd.h, x0 = cd.x0, y0 = cd.y0;
  const int tlo = imin(iniTh, minTh);

  if (tid == 0) { qn = 0; nHigh = 0; }
  // stage pixels [x0-3, x0+cw+2] x [y0-3, y0+ch+2] (always inside the level: x0 >= 19)
  {
    const uint8_t* img = lv.base + (size_t)f * lv.frameStride + (size_t)(y0 - 3) * lv.pitch + (x0 - 3);
    const int tw = cw + 6, th = ch + 6;
    for (int ty = wave; ty < th; ty += 4) {
      const uint8_t* row = img + (size_t)ty * lv.pitch;
      for (int tx = lane; tx < tw; tx += 64) tile[ty * kTilePitch + tx] = row[tx];
    }
    for (int i = tid; i < kScoreRows * kScorePitch / 4; i += 256) reinterpret_cast<uint32_t*>(score)[i] = 0;
  }
  __syncthreads();

  const int npix = cw * ch;
  const int stepY = 256 / cw, stepX = 256 - stepY * cw;
  const int py0 = tid / cw, px0 = tid - py0 * cw;

  // ---- A: arc test at the lower threshold ----
  {
    int px = px0, py = py0;
    for (int p = tid; p < npix; p += 256) {
      const uint8_t* c = &tile[(py + 3) * kTilePitch + px + 3];
      const int v = c[0];
      uint32_t dark = 0, bright = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int x = c[kRingDx[k] + kRingDy[k] * kTilePitch];
        dark |= (uint32_t)(x < v - tlo) << k;
        bright |= (uint32_t)(x > v + tlo) << k;
      }
      if (has_arc9(dark) || has_arc9(bright)) {
        const int q = atomicAdd(&qn, 1);
        queue[q] = (uint16_t)((py << 8) | px);
      }
      px += stepX; py += stepY;
      if (px >= cw) { px -= cw; py++; }
    }
  }
  __syncthreads();
  // ---- B: exact score of the queued corners ----
  {
    const int n = qn;
    for (int q = tid; q < n; q += 256) {
      const int e = queue[q], px = e & 255, py = e >> 8;
      const int S = fast_S(&tile[(py + 3) * kTilePitch + px + 3]);
      score[(py + 1) * kScorePitch + px + 1] = (uint8_t)(S - 1);
    }
  }
  __syncthreads();
  // ---- C: cell-local 3x3 NMS, classify against both thresholds ----
  uint8_t* cls = tile;  // the pixel tile is dead from here on
  {
    int px = px0, py = py0, high = 0;
    for (int p = tid; p < npix; p += 256) {
      const uint8_t* s = &score[(py + 1) * kScorePitch + px + 1];
      const int v = s[0];
      int c = 0;
      if (v > 0) {
        const bool keep = v > s[-1] && v > s[1] && v > s[-kScorePitch - 1] && v > s[-kScorePitch] &&
                          v > s[-kScorePitch + 1] && v > s[kScorePitch - 1] && v > s[kScorePitch] &&
                          v > s[kScorePitch + 1];
        if (keep) c = (v >= minTh ? 1 : 0) | (v >= iniTh ? 2 : 0);
      }
      cls[p] = (uint8_t)c;
      high += (c >> 1);
      px += stepX; py += stepY;
      if (px >= cw) { px -= cw; py++; }
    }
    if (high) atomicAdd(&nHigh, high);
  }
  __syncthreads();
  // ---- D: ordered emission (raster order inside the cell, :884-893) ----
  const int want = nHigh > 0 ? 2 : 1;
  Candidate* out = slots + (size_t)f * slotsPerFrame + cd.slotBase;
  int run = 0;  // identical in every thread
  {
    int px = px0, py = py0;
    for (int pbase = 0; pbase < npix; pbase += 256) {
      const int p = pbase + tid;
      const bool sel = p < npix && (cls[p] & want);
      const unsigned long long bal = __ballot(sel);
      const int inWave = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) waveTot[wave] = __popcll(bal);
      __syncthreads();
      int base = run;
      for (int w = 0; w < wave; w++) base += waveTot[w];
      run += waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
      if (sel) {
        Candidate c;
        c.xy = (uint32_t)(x0 + px - kMinBorder) | ((uint32_t)(y0 + py - kMinBorder) << 16);
        c.score = score[(py + 1) * kScorePitch + px + 1];
        out[base + inWave] = c;
      }
      __syncthreads();
      px += stepX; py += stepY;
      if (px >= cw) { px -= cw; py++; }
    }
  }
  if (tid == 0) cellCount[(size_t)f * nCells + cellId] = (uint16_t)run;
}

void launch_fast_cells(hipStream_t s, PyramidViews pyr, const CellDesc* d_cells, int nCells,
                       int nFrames, int iniTh, int minTh, Candidate* d_slots, int slotsPerFrame,
                       uint16_t* d_cellCount) {
  if (nCells <= 0 || nFrames <= 0) return;
  iniTh = iniTh < 0 ? 0 : (iniTh > 255 ? 255 : iniTh);  // cv::FAST clamps the threshold
  minTh = minTh < 0 ? 0 : (minTh > 255 ? 255 : minTh);
  hipLaunchKernelGGL(k_fast_cells, dim3(nCells, nFrames), dim3(256), 0, s, pyr, d_cells, nCells,
                     iniTh, minTh, d_slots, slotsPerFrame, d_cellCount);
}

// Ordered compaction: cells of a level in cell-row-major order, raster inside each cell.
__global__ __launch_bounds__(256) void k_gather_candidates(const CellDesc* __restrict__ cells,
                                                           const LevelGeom* __restrict__ lvg,
                                                           const Candidate* __restrict__ slots,
                                                           int slotsPerFrame,
                                                           const uint16_t* __restrict__ cellCount,
                                                           int cellsPerFrame,
                                                           Candidate* __restrict__ cand,
                                                           int32_t* __restrict__ candCount,
                                                           int32_t* __restrict__ cellPrefix,
                                                           int nlevels) {
  __shared__ int waveTot[4];
  const int l = blockIdx.x, f = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const LevelGeom g = lvg[l];
  const uint16_t* cnt = cellCount + (size_t)f * cellsPerFrame + g.cellStart;
  int32_t* pre = cellPrefix + (size_t)f * cellsPerFrame + g.cellStart;
  int run = 0;  // identical in every thread
  for (int cb = 0; cb < g.nCells; cb += 256) {
    const int c = cb + tid;
    const int v = c < g.nCells ? cnt[c] : 0;
    // wave inclusive scan
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) waveTot[wave] = x;
    __syncthreads();
    int base = run;
    for (int w = 0; w < wave; w++) base += waveTot[w];
    run += waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
    if (c < g.nCells) pre[c] = base + x - v;
    __syncthreads();
  }
  if (tid == 0) candCount[(size_t)f * nlevels + l] = run;
  const Candidate* sl = slots + (size_t)f * slotsPerFrame;
  Candidate* out = cand + (size_t)f * slotsPerFrame + g.slotStart;
  for (int c = wave; c < g.nCells; c += 4) {
    const int n = cnt[c], b = pre[c];
    const Candidate* src = sl + cells[g.cellStart + c].slotBase;
    for (int i = lane; i < n; i += 64) out[b + i] = src[i];
  }
}

void launch_gather_candidates(hipStream_t s, const CellDesc* d_cells, const LevelGeom* d_lv,
                              int nlevels, int nFrames, const Candidate* d_slots,
                              int slotsPerFrame, const uint16_t* d_cellCount, int cellsPerFrame,
                              Candidate* d_cand, int32_t* d_candCount, int32_t* d_cellPrefix) {
  if (nFrames <= 0) return;
  hipLaunchKernelGGL(k_gather_candidates, dim3(nlevels, nFrames), dim3(256), 0, s, d_cells, d_lv,
                     d_slots, slotsPerFrame, d_cellCount, cellsPerFrame, d_cand, d_candCount,
                     d_cellPrefix, nlevels);
}

}  // namespace orbfe
