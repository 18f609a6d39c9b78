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
#include <cstdlib>

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

// The same arrays by COUNTING instead of sorting (frames of up to 8192 features: every Frame / KeyFrame of the reference's
// settings).  The bitonic network above is 66 passes with a workgroup barrier each for 2048 keys -- 29 us, half of a
// host-array projection search and most of orbfe_frame_upload.  Here: cell counts with LDS atomics, one block scan over the
// 3072 cells (= cellOff), an unordered scatter of the keys into their cell's segment, and every key's rank inside its
// segment by counting the smaller keys there (segments hold ~1 feature; the order inside a cell is ascending feature
// index, as mGrid's push_back order, src/Frame.cc:246-259).  Seven barriers.  A frame that piles more than kGridRankMax
// features into one cell (synthetic clusters) would make that last step quadratic: the workgroup then runs the bitonic
// network instead -- same result.
constexpr int kGridRankMax = 64;
__global__ __launch_bounds__(1024) void k_grid_build_count(GridFrame f, int sortN, uint32_t* __restrict__ sortedKey,
                                                           int32_t* __restrict__ cellOff) {
  extern __shared__ uint32_t keys[];       // [sortN] scatter target / bitonic array
  __shared__ int s_off[GRID_CELLS + 1];    // counts, then first position of each cell
  __shared__ int s_cur[GRID_CELLS];        // scatter cursors
  __shared__ int s_wave[16];
  __shared__ int s_max;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int c = t; c < GRID_CELLS; c += 1024) { s_off[c] = 0; s_cur[c] = 0; }
  if (t == 0) s_max = 0;
  __syncthreads();
  auto key_of = [&](int i) -> uint32_t {
    // PosInGrid, src/Frame.cc:417-427: round() half away from zero
    const int px = (int)roundf((f.x[i] - f.minX) * f.wInv);
    const int py = (int)roundf((f.y[i] - f.minY) * f.hInv);
    if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) return KEY_NONE;
    return ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
  };
  for (int i = t; i < f.n; i += 1024) {
    const uint32_t k = key_of(i);
    if (k != KEY_NONE) atomicAdd(&s_off[k >> 16], 1);
  }
  __syncthreads();
  // exclusive scan over the cells: thread t owns cells 3t .. 3t+2
  const int c0 = s_off[3 * t], c1 = s_off[3 * t + 1], c2 = s_off[3 * t + 2];
  {
    int m = c0 > c1 ? c0 : c1;
    m = m > c2 ? m : c2;
    m = wave_max_i32_dpp(m);
    if (lane == 0) atomicMax(&s_max, m);
  }
  const int incl = wave_incl_scan_dpp(c0 + c1 + c2);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += s_wave[w];
  const int excl = base + incl - (c0 + c1 + c2);
  __syncthreads();  // (everyone has read its three counts)
  s_off[3 * t] = excl; s_off[3 * t + 1] = excl + c0; s_off[3 * t + 2] = excl + c0 + c1;
  cellOff[3 * t] = excl; cellOff[3 * t + 1] = excl + c0; cellOff[3 * t + 2] = excl + c0 + c1;
  if (t == 1023) { s_off[GRID_CELLS] = excl + c0 + c1 + c2; cellOff[GRID_CELLS] = excl + c0 + c1 + c2; }
  __syncthreads();
  if (s_max <= kGridRankMax) {  // block-uniform
    for (int i = t; i < f.n; i += 1024) {
      const uint32_t k = key_of(i);
      if (k != KEY_NONE) keys[s_off[k >> 16] + atomicAdd(&s_cur[k >> 16], 1)] = k;
    }
    __syncthreads();
    for (int i = t; i < f.n; i += 1024) {
      const uint32_t k = key_of(i);
      if (k == KEY_NONE) continue;
      const int s = s_off[k >> 16], e = s_off[(k >> 16) + 1];
      int rank = 0;
      for (int p = s; p < e; p++) rank += keys[p] < k ? 1 : 0;
      sortedKey[s + rank] = k;
    }
    return;
  }
  // ---- a crowded cell: the bitonic network of k_grid_build (cellOff is already written) ----
  for (int i = t; i < sortN; i += 1024) keys[i] = i < f.n ? key_of(i) : KEY_NONE;
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
}

// One wavefront per query, 4 per block.
__device__ __forceinline__ void window_search_one(const GridFrame& f, const uint32_t* __restrict__ sortedKey,
                                                  const int32_t* __restrict__ cellOff, const WindowQueries& q,
                                                  int32_t* __restrict__ count, uint32_t* __restrict__ cand, int qi, int lane) {
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
    const uint32_t k = wave_min_u32_dpp(bestKey);
    // (keys are unique: they carry the scan position) the owner of the minimum writes its keypoint
    if (k == 0xffffffffu) { if (lane == 0) q.best[qi] = -1; }
    else if (bestKey == k) q.best[qi] = (int)(k >> 16) <= q.maxDist ? (int32_t)bestId : -1;
    return;
  }
  if (lane == 0) count[qi] = n;
}

__global__ __launch_bounds__(256) void k_window_search(GridFrame f, const uint32_t* __restrict__ sortedKey,
                                                       const int32_t* __restrict__ cellOff, WindowQueries q,
                                                       int32_t* __restrict__ count, uint32_t* __restrict__ cand) {
  window_search_one(f, sortedKey, cellOff, q, count, cand, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// The window searches of ALL jobs of a call in one launch (Fuse into K key frames, one frame against K relocalisation
// candidates, the two directions of SearchBySim3): K launches of ~250 workgroups each ran one after the other on the
// call's stream, 8 us apiece.  A workgroup finds its job by its block range (scalar search) and reads the job's block.
__global__ __launch_bounds__(256) void k_window_search_multi(const WindowSearchJob* __restrict__ jobs, int nJobs) {
  int j = 0;
  for (int k = 1; k < nJobs; k++)
    if ((int)blockIdx.x >= jobs[k].blockStart) j = k;  // block-uniform
  const WindowSearchJob J = jobs[j];
  window_search_one(J.f, J.sortedKey, J.cellOff, J.q, J.count, J.cand, ((int)blockIdx.x - J.blockStart) * 4 + (threadIdx.x >> 6),
                    threadIdx.x & 63);
}

// rotation-histogram bin, src/ORBmatcher.cc:1601-1610 (C round(): half away from zero)
__device__ __forceinline__ int claim_rot_bin(float a1, float a2) {
  float rot = __fsub_rn(a1, a2);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  return bin;
}

// ---------------------------------------------------------------------------------------------------------------------
// The claim loops of SearchByProjection (map points :77-135, last frame :1572-1612, key frame :1726-1760, Sim3 :431-451) and
// SearchForInitialization (:492-545) -- see ClaimJob (match_kernels.h).  One workgroup per job.  A round lets EVERY query
// choose, in parallel, among the candidates that no query in front of it holds (according to the previous round's
// choices); rounds repeat until no choice changes.  The fixed point is the reference's sequential result: query 0 never
// depends on anyone, and once the queries in front of j have their final choices so has j -- at most nq + 1 rounds, in
// practice three or four, because two map points rarely want the same feature.
// ---------------------------------------------------------------------------------------------------------------------

// three dominant bins (ComputeThreeMaxima, src/ORBmatcher.cc:1635-1690): the 30 counts come into registers at once
__device__ __forceinline__ void claim_three_maxima(const int* hist, int* keep) {
  int h[30];
#pragma unroll
  for (int i = 0; i < 30; i++) h[i] = hist[i];
  int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    const int s = h[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
    else if (s > max3) { max3 = s; i3 = i; }
  }
  if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
  else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
  keep[0] = i1; keep[1] = i2; keep[2] = i3;
}

// BEST / RATIO forms.  A feature is hidden from query j when an EARLIER query whose match blocks (blockVal) chose it in the
// previous round.  own[2][n] hold, per feature, a stamp (round << 21 | 0x1fffff - query) written with atomicMax by every
// blocking chooser of a round: the newest round wins, inside a round the smallest query index; a round reads the array the
// previous round wrote and writes the other one, so a round is ONE barrier and nothing is ever cleared (an entry whose round
// is not the previous one is "free"; 0xffffffff = taken at entry, hidden from everyone).  What a thread needs of its FIRST
// query (nq <= 1024: its only one) stays in registers over the rounds -- flags, list length, the first eight candidates, the
// current choice -- so a round touches LDS only unless a list is longer than eight; further queries of a thread (tid + 1024,
// ...) take everything from memory every round.  The match array is built in LDS (the stamp arrays are free by then) and
// written out once.
constexpr int kClaimRoundBits = 11, kClaimJBits = 21;
constexpr uint32_t kClaimJMask = (1u << kClaimJBits) - 1u, kClaimTaken = 0xffffffffu;
// wave64 sum / max through the DPP network; every lane returns the wave's value
__device__ __forceinline__ int claim_wave_sum(int x) { return wave_sum_dpp(x); }
__device__ __forceinline__ int claim_wave_max(int x) { return wave_max_i32_dpp(x); }

// kOneJob: the (usual) single job travels as the kernel argument -- no dependent load of a block that the input copy has
// just put into HBM in front of the kernel's first memory request
template <bool kOneJob>
__global__ __launch_bounds__(1024) void k_window_claim(const ClaimJob job, const ClaimJob* __restrict__ jobs) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_hist[16][32];  // per wave: a thousand atomics on thirty shared counters would queue behind each other
  __shared__ int s_keep[3];
  __shared__ int s_changed[3];
  __shared__ int s_total, s_pruned, s_maxc;
  const ClaimJob J = kOneJob ? job : jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* own0 = reinterpret_cast<uint32_t*>(J.owner ? J.owner : s_dyn);
  uint32_t* own1 = own0 + J.n;
  uint8_t* s_oct = reinterpret_cast<uint8_t*>(s_dyn + (J.owner ? 0 : 2 * J.n));  // RATIO: octave bytes of the features (255: read HBM)
  if (tid < 512) s_hist[tid >> 5][tid & 31] = 0;
  if (tid < 3) s_changed[tid] = 0;
  if (tid == 0) { s_total = 0; s_pruned = 0; s_maxc = 0; }
  const bool has0 = tid < J.nq;
  bool act0 = false, bv0 = true;
  int nc0 = 0, c0 = -2;
  uint32_t e0[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  {
    if (has0) {
      const uint4* L4 = reinterpret_cast<const uint4*>(J.cand + (size_t)tid * J.K);  // K % 8 == 0, lists 256-byte aligned
      const uint4 ea = L4[0], eb = L4[1];
      e0[0] = ea.x; e0[1] = ea.y; e0[2] = ea.z; e0[3] = ea.w; e0[4] = eb.x; e0[5] = eb.y; e0[6] = eb.z; e0[7] = eb.w;
      act0 = !J.active || J.active[tid];
      bv0 = !J.blockVal || J.blockVal[tid];
    }
    int mc = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = J.count[j];
      mc = c > mc ? c : mc;
      if (j == tid) nc0 = c < J.K ? c : J.K;
      else J.choice[j] = -2;  // (no choice computed yet: the first round counts as a change)
    }
    for (int i = tid; i < J.n; i += 1024) {
      const uint32_t v = (J.blocked && J.blocked[i]) ? kClaimTaken : 0u;
      own0[i] = v;
      own1[i] = v;
    }
    if (J.mode == CLAIM_RATIO)
      for (int i = tid; i < J.n; i += 1024) {
        const int o = J.octave[i];
        s_oct[i] = (uint8_t)((unsigned)o < 255u ? o : 255);
      }
    __syncthreads();
    mc = claim_wave_max(mc);
    if (lane == 0 && mc) atomicMax(&s_maxc, mc);
  }
  int rounds = 0;          // rounds run so far
  uint32_t stampR = 0;     // round field of the stamps the PREVIOUS round wrote (0: nothing written yet)
  for (;; rounds++) {
    if (stampR >= (1u << kClaimRoundBits) - 2u) {  // (round 2047 is never used: its stamp of query 0 would read "taken at entry")
      // the round field is about to wrap (only chains of thousands of queries that all want the same features get here):
      // clear the arrays and write the current choices again under round 1
      __syncthreads();
      for (int i = tid; i < J.n; i += 1024) {
        if (own0[i] != kClaimTaken) own0[i] = 0u;
        if (own1[i] != kClaimTaken) own1[i] = 0u;
      }
      __syncthreads();
      uint32_t* rd = (rounds & 1) ? own1 : own0;  // the array this round is going to read
      for (int j = tid; j < J.nq; j += 1024) {
        const int c = j == tid ? c0 : J.choice[j];
        const bool bv = j == tid ? bv0 : (!J.blockVal || J.blockVal[j]);
        if (c >= 0 && bv) atomicMax(&rd[c], (1u << kClaimJBits) | (kClaimJMask - (uint32_t)j));
      }
      stampR = 1;
      __syncthreads();
    }
    const uint32_t* rd = (rounds & 1) ? own1 : own0;
    uint32_t* wr = (rounds & 1) ? own0 : own1;
    const uint32_t stampW = stampR + 1;
    const int flag = rounds % 3;
    if (tid == 0) s_changed[(rounds + 1) % 3] = 0;  // (last read two barriers ago)
    int changed = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      int c = -1;
      const bool slot0 = j == tid;
      if (slot0 ? act0 : (!J.active || J.active[j])) {
        int nc;
        if (slot0) nc = nc0;
        else { nc = J.count[j]; nc = nc < J.K ? nc : J.K; }
        // the list is read eight entries (two 16-byte requests) at a time: one memory round trip per eight candidates
        const uint4* L4 = reinterpret_cast<const uint4*>(J.cand + (size_t)j * J.K);
        int bestDist = 256, bestDist2 = 256, bestIdx = -1, secIdx = -1;
        // best and second best of eight entries, as selects (written with branches the compiler kept the four running
        // values in scratch memory behind a computed store address).  `free`: not taken at entry, not held by an earlier
        // query in the previous round, not past the end of the list
#define ORBFE_CLAIM_PROC8(EV, NVALID)                                                                                      \
  do {                                                                                                                    \
    uint32_t own_[8];                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; u++) own_[u] = u < (NVALID) ? rd[(EV)[u] & 0xffffu] : kClaimTaken;           \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                                       \
      const bool held_ = own_[u] == kClaimTaken ||                                                                        \
                         ((own_[u] >> kClaimJBits) == stampR && stampR != 0u && (int)(kClaimJMask - (own_[u] & kClaimJMask)) < j); \
      const int idx_ = (int)((EV)[u] & 0xffffu), dist_ = (int)((EV)[u] >> 16);                                            \
      const bool lt1_ = !held_ && dist_ < bestDist, lt2_ = !held_ && dist_ < bestDist2;                                   \
      bestDist2 = lt1_ ? bestDist : (lt2_ ? dist_ : bestDist2);                                                           \
      secIdx = lt1_ ? bestIdx : (lt2_ ? idx_ : secIdx);                                                                   \
      bestDist = lt1_ ? dist_ : bestDist;                                                                                 \
      bestIdx = lt1_ ? idx_ : bestIdx;                                                                                    \
    }                                                                                                                     \
  } while (0)
        if (slot0) {
          if (nc > 0) ORBFE_CLAIM_PROC8(e0, nc);
        } else if (nc > 0) {
          const uint4 ea = L4[0], eb = L4[1];
          const uint32_t ev8[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
          ORBFE_CLAIM_PROC8(ev8, nc);
        }
        for (int k0 = 8; k0 < nc; k0 += 8) {
          const uint4 ea = L4[k0 >> 2], eb = L4[(k0 >> 2) + 1];
          const uint32_t ev8[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
          ORBFE_CLAIM_PROC8(ev8, nc - k0);
        }
#undef ORBFE_CLAIM_PROC8
        if (bestIdx >= 0 && bestDist <= J.maxDist) {
          c = bestIdx;
          if (J.mode == CLAIM_RATIO && (float)bestDist > __fmul_rn(J.nnratio, (float)bestDist2)) {
            // (:124-127: the ratio only counts between two candidates of the same level)
            int bestLevel = s_oct[bestIdx], bestLevel2 = secIdx >= 0 ? (int)s_oct[secIdx] : -1;
            if (bestLevel == 255) bestLevel = J.octave[bestIdx];
            if (bestLevel2 == 255) bestLevel2 = J.octave[secIdx];
            if (bestLevel == bestLevel2) c = -1;
          }
        }
      }
      bool bv;
      if (slot0) {
        if (c != c0) { c0 = c; changed = 1; }
        bv = bv0;
      } else {
        if (c != J.choice[j]) { J.choice[j] = c; changed = 1; }
        bv = !J.blockVal || J.blockVal[j];
      }
      if (c >= 0 && bv) atomicMax(&wr[c], (stampW << kClaimJBits) | (kClaimJMask - (uint32_t)j));
    }
    if (changed) s_changed[flag] = 1;
    __syncthreads();
    stampR = stampW;
    if (!s_changed[flag]) break;  // block-uniform: a round without a change is the fixed point
  }
  // ---- the match array (in LDS, over the stamp arrays), the rotation histogram (:1614-1628) and the count ----
  int* match = reinterpret_cast<int*>(own0);
  for (int i = tid; i < J.n; i += 1024) match[i] = -1;
  __syncthreads();
  int ev = 0, bin0 = -1;
  for (int j = tid; j < J.nq; j += 1024) {
    const int c = j == tid ? c0 : J.choice[j];
    if (c < 0) continue;
    atomicMax(&match[c], j);  // a feature whose holder does not block is overwritten by the later ones
    ev++;
    if (J.checkOri) {
      const int b = claim_rot_bin(J.qAngle[j], J.fAngle[c]);  // every take is pushed (:1601-1610)
      atomicAdd(&s_hist[wave][b], 1);
      if (j == tid) bin0 = b;
    }
  }
  ev = claim_wave_sum(ev);
  if (lane == 0 && ev) atomicAdd(&s_total, ev);
  __syncthreads();
  if (J.checkOri) {
    if (tid < 30) {
      int t = 0;
#pragma unroll
      for (int w = 0; w < 16; w++) t += s_hist[w][tid];
      s_hist[0][tid] = t;
    }
    __syncthreads();
    if (tid == 0) claim_three_maxima(s_hist[0], s_keep);
    __syncthreads();
    const int k0 = s_keep[0], k1 = s_keep[1], k2 = s_keep[2];
    int pr = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = j == tid ? c0 : J.choice[j];
      if (c < 0) continue;
      const int b = j == tid ? bin0 : claim_rot_bin(J.qAngle[j], J.fAngle[c]);
      if (b == k0 || b == k1 || b == k2) continue;
      match[c] = -1;  // :1617-1625: every entry of a dropped bin clears its feature and counts
      pr++;
    }
    pr = claim_wave_sum(pr);
    if (lane == 0 && pr) atomicAdd(&s_pruned, pr);
    __syncthreads();
  }
  for (int i = tid; i < J.n; i += 1024) J.match[i] = match[i];
  if (tid == 0) {
    J.header[0] = s_maxc;
    J.header[1] = s_total - s_pruned;
    J.header[2] = rounds;
    J.header[3] = 0;
  }
}

// SearchForInitialization (:492-545).  A feature is hidden from query j only for candidates at distance >= the smallest
// distance at which an earlier query took it (vMatchedDistance, :516-517): the choosers of a feature form a linked list
// (owner = head, link = next), walked for the ones in front of j; the last taker keeps the feature (:529-533).  Called once
// per initialisation attempt: the plain form (everything re-read every round, three barriers per round).
__global__ __launch_bounds__(1024) void k_window_claim_init(const ClaimJob* __restrict__ jobs) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_hist[30];
  __shared__ int s_keep[3];
  __shared__ int s_changed[2];
  __shared__ int s_total, s_pruned, s_maxc;
  const ClaimJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  int32_t* owner = J.owner ? J.owner : s_dyn;
  constexpr int kNone = 0x7fffffff;
  if (tid < 30) s_hist[tid] = 0;
  if (tid < 2) s_changed[tid] = 0;
  if (tid == 0) { s_total = 0; s_pruned = 0; s_maxc = 0; }
  __syncthreads();
  {
    int mc = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      J.choice[j] = -2;  // (no choice computed yet: the first round counts as a change)
      const int c = J.count[j];
      mc = c > mc ? c : mc;
    }
    if (mc) atomicMax(&s_maxc, mc);
    for (int i = tid; i < J.n; i += 1024) owner[i] = -1;
  }
  __syncthreads();
  int rounds = 0;
  for (;; rounds++) {
    const int flag = rounds & 1;
    if (tid == 0) s_changed[flag ^ 1] = 0;  // (everyone read it before the rebuild barriers of the previous round)
    int changed = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      int c = -1;
      if (!J.active || J.active[j]) {
        int nc = J.count[j];
        nc = nc < J.K ? nc : J.K;
        const uint4* L4 = reinterpret_cast<const uint4*>(J.cand + (size_t)j * J.K);
        int bestDist = kNone, bestDist2 = kNone, bestIdx = -1;
        for (int k0 = 0; k0 < nc; k0 += 8) {
          const uint4 ea = L4[k0 >> 2], eb = L4[(k0 >> 2) + 1];
          const uint32_t ev8[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (k0 + u >= nc) break;
            const uint32_t e = ev8[u];
            const int i2 = (int)(e & 0xffffu), dist = (int)(e >> 16);
            int held = kNone;  // vMatchedDistance[i2] as query j sees it
            for (int i = owner[i2]; i >= 0; i = J.link[i])
              if (i < j) {
                const int ci = J.choice[i];  // (may be this round's: only a consistent "i holds i2 at d" entry is used)
                if (ci >= 0 && (ci & 0xffff) == i2) held = (ci >> 16) < held ? (ci >> 16) : held;
              }
            if (held <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
          }
        }
        if (bestIdx >= 0 && bestDist <= J.maxDist && (float)bestDist < __fmul_rn((float)bestDist2, J.nnratio))
          c = (bestDist << 16) | bestIdx;
      }
      if (c != J.choice[j]) { J.choice[j] = c; changed = 1; }
    }
    if (changed) s_changed[flag] = 1;
    __syncthreads();
    if (!s_changed[flag]) break;  // block-uniform: a round without a change is the fixed point
    for (int i = tid; i < J.n; i += 1024) owner[i] = -1;
    __syncthreads();
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = J.choice[j];
      if (c >= 0) J.link[j] = atomicExch(&owner[c & 0xffff], j);
    }
    __syncthreads();
  }
  // ---- vnMatches12, the rotation histogram (:557-563), the count, "update prev matched" (:595-600) ----
  int ev = 0;
  for (int j = tid; j < J.nq; j += 1024) {
    const int c = J.choice[j];
    int m = -1;
    if (c >= 0) {
      const int f = c & 0xffff;
      if (J.checkOri) atomicAdd(&s_hist[claim_rot_bin(J.qAngle[j], J.fAngle[f])], 1);  // every take is pushed (:540-553)
      bool holder = true;  // a later query that took the feature replaced this one (:529-533)
      for (int i = owner[f]; i >= 0; i = J.link[i])
        if (i > j) { const int ci = J.choice[i]; if (ci >= 0 && (ci & 0xffff) == f) holder = false; }
      if (holder) { m = f; ev++; }
    }
    J.match[j] = m;
  }
  if (ev) atomicAdd(&s_total, ev);
  __syncthreads();
  if (J.checkOri) {
    if (tid == 0) claim_three_maxima(s_hist, s_keep);
    __syncthreads();
    const int k0 = s_keep[0], k1 = s_keep[1], k2 = s_keep[2];
    int pr = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = J.choice[j];
      if (c < 0) continue;
      const int b = claim_rot_bin(J.qAngle[j], J.fAngle[c & 0xffff]);
      if (b == k0 || b == k1 || b == k2) continue;
      if (J.match[j] >= 0) { J.match[j] = -1; pr++; }  // :557-563: only a still-matched entry counts
    }
    if (pr) atomicAdd(&s_pruned, pr);
    __syncthreads();
  }
  if (J.prevX)
    for (int j = tid; j < J.nq; j += 1024) {
      const int m = J.match[j];
      J.prevX[j] = m >= 0 ? J.fx[m] : J.qx[j];
      J.prevY[j] = m >= 0 ? J.fy[m] : J.qy[j];
    }
  if (tid == 0) {
    J.header[0] = s_maxc;
    J.header[1] = s_total - s_pruned;
    J.header[2] = rounds;
    J.header[3] = 0;
  }
}

}  // namespace

void launch_window_claim(hipStream_t s, const ClaimJob* d_jobs, const ClaimJob* h_first, int nJobs, size_t ldsBytes, bool initForm) {
  if (nJobs <= 0) return;
  if (initForm) hipLaunchKernelGGL(k_window_claim_init, dim3(nJobs), dim3(1024), ldsBytes, s, d_jobs);
  else if (nJobs == 1 && h_first) hipLaunchKernelGGL((k_window_claim<true>), dim3(1), dim3(1024), ldsBytes, s, *h_first, d_jobs);
  else hipLaunchKernelGGL((k_window_claim<false>), dim3(nJobs), dim3(1024), ldsBytes, s, ClaimJob{}, d_jobs);
}

void launch_grid_build(hipStream_t s, const GridFrame& f, uint32_t* sortedKey, int32_t* cellOff) {
  int sortN = 64;
  while (sortN < f.n) sortN <<= 1;
  static const int kForceSort = getenv("ORBFE_GRID_SORT") ? atoi(getenv("ORBFE_GRID_SORT")) : 0;  // 1: the bitonic kernel for every frame
  if (f.n <= 8192 && !kForceSort) hipLaunchKernelGGL(k_grid_build_count, dim3(1), dim3(1024), (size_t)sortN * 4, s, f, sortN, sortedKey, cellOff);
  else hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(1024), (size_t)sortN * 4, s, f, sortN, sortedKey, cellOff);
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

void launch_window_search_multi(hipStream_t s, const WindowSearchJob* d_jobs, int nJobs, int totalBlocks) {
  if (nJobs <= 0 || totalBlocks <= 0) return;
  hipLaunchKernelGGL(k_window_search_multi, dim3(totalBlocks), dim3(256), 0, s, d_jobs, nJobs);
}

void launch_window_search(hipStream_t s, const GridFrame& f, const uint32_t* sortedKey, const int32_t* cellOff,
                          const WindowQueries& q, int32_t* count, uint32_t* cand) {
  if (q.n <= 0) return;
  hipLaunchKernelGGL(k_window_search, dim3((q.n + 3) / 4), dim3(256), 0, s, f, sortedKey, cellOff, q, count, cand);
}

}  // namespace orbfe
