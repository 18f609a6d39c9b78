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

// rotation-histogram bin, src/ORBmatcher.cc:1601-1610 (C round(): half away from zero)
__device__ __forceinline__ int claim_rot_bin(float a1, float a2) {
  float rot = __fsub_rn(a1, a2);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  return bin;
}

// The claim loops of SearchByProjection (map points :77-135, last frame :1572-1612, key frame :1726-1760, Sim3 :431-451) and
// SearchForInitialization (:492-545) -- see ClaimJob (match_kernels.h).  One workgroup per job.  A round lets EVERY query
// choose, in parallel, among the candidates that no query in front of it holds (according to the previous round's
// choices); rounds repeat until no choice changes.  The fixed point is the reference's sequential result: query 0 never
// depends on anyone, and once the queries in front of j have their final choices so has j -- at most nq + 1 rounds, in
// practice two or three, because two map points rarely want the same feature.
//   BEST / RATIO: a feature is hidden from query j when an earlier query whose match blocks (blockVal) chose it:
//                 owner[feature] = the smallest such query (atomicMin), hidden <=> owner < j.
//   INIT:         a feature is hidden from query j for candidates at distance >= the smallest distance at which an
//                 earlier query took it (vMatchedDistance, :516-517): the choosers of a feature form a linked list
//                 (owner = head, link = next), walked for the ones in front of j.
__global__ __launch_bounds__(1024) void k_window_claim(const ClaimJob* __restrict__ jobs) {
  extern __shared__ int32_t s_dyn[];
  __shared__ int s_hist[30];
  __shared__ int s_keep[3];
  __shared__ int s_changed[2];
  __shared__ int s_total, s_pruned, s_maxc, s_conflict;
  const ClaimJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  int32_t* owner = J.owner ? J.owner : s_dyn;
  constexpr int kNone = 0x7fffffff;
  const bool init = J.mode == CLAIM_INIT;
  const int ownerFree = init ? -1 : kNone;
  if (tid < 30) s_hist[tid] = 0;
  if (tid < 2) s_changed[tid] = 0;
  if (tid == 0) { s_total = 0; s_pruned = 0; s_maxc = 0; s_conflict = 0; }
  __syncthreads();
  // What a thread needs of its FIRST query (nq <= 1024: its only one) stays in registers over the rounds -- the flags, the
  // list length, the first eight candidates, the current choice: after round 0 a round of the BEST / RATIO forms touches LDS
  // only, unless a list is longer than eight.  Further queries of the thread (tid + 1024, ...) and the INIT form take
  // everything from memory every round.
  uint8_t* s_oct = reinterpret_cast<uint8_t*>(s_dyn + (J.owner ? 0 : J.n));  // RATIO: octave bytes of the features (255: read HBM)
  const bool has0 = !init && tid < J.nq;
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
      J.choice[j] = -2;  // (no choice computed yet: the first round counts as a change)
      const int c = J.count[j];
      mc = c > mc ? c : mc;
      if (j == tid) nc0 = c < J.K ? c : J.K;
    }
    if (mc) atomicMax(&s_maxc, mc);
    // (BEST / RATIO: a feature that is taken at entry is "owned by query -1", i.e. hidden from everyone)
    for (int i = tid; i < J.n; i += 1024) owner[i] = (!init && J.blocked && J.blocked[i]) ? -1 : ownerFree;
    if (J.mode == CLAIM_RATIO)
      for (int i = tid; i < J.n; i += 1024) {
        const int o = J.octave[i];
        s_oct[i] = (uint8_t)((unsigned)o < 255u ? o : 255);
      }
  }
  __syncthreads();
  int rounds = 0;
  for (;; rounds++) {
    const int flag = rounds & 1;
    if (tid == 0) s_changed[flag ^ 1] = 0;  // (everyone read it before the rebuild barriers of the previous round)
    int changed = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      int c = -1;
      const bool slot0 = has0 && j == tid;
      if (slot0 ? act0 : (!J.active || J.active[j])) {
        int nc;
        if (slot0) nc = nc0;
        else { nc = J.count[j]; nc = nc < J.K ? nc : J.K; }
        // the list is read eight entries (two 16-byte requests) at a time: one memory round trip per eight candidates instead
        // of one per candidate -- a round is a handful of dependent round trips, and most lists are shorter than eight
        const uint4* L4 = reinterpret_cast<const uint4*>(J.cand + (size_t)j * J.K);
        if (init) {
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
        } else {
          int bestDist = 256, bestDist2 = 256, bestIdx = -1, secIdx = -1;
          // best and second best of eight entries, as selects (written with branches the compiler kept the four running
          // values in scratch memory behind a computed store address)
#define ORBFE_CLAIM_PROC8(EV, NVALID)                                                                                      \
  do {                                                                                                                    \
    int own_[8];                                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 8; u++) own_[u] = u < (NVALID) ? owner[(EV)[u] & 0xffffu] : -1; /* eight lookups in flight */ \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                                       \
      const bool free_ = own_[u] >= j; /* not held by an earlier query, not taken at entry, not past the end of the list */ \
      const int idx_ = (int)((EV)[u] & 0xffffu), dist_ = (int)((EV)[u] >> 16);                                            \
      const bool lt1_ = free_ && dist_ < bestDist, lt2_ = free_ && dist_ < bestDist2;                                     \
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
      }
      if (slot0) {
        if (c != c0) { c0 = c; J.choice[j] = c; changed = 1; }
      } else if (c != J.choice[j]) { J.choice[j] = c; changed = 1; }
    }
    if (changed) s_changed[flag] = 1;
    __syncthreads();
    if (!s_changed[flag]) break;  // block-uniform: a round without a change is the fixed point
    // (a feature taken at entry keeps its -1: atomicMin with a query index never lowers it)
    for (int i = tid; i < J.n; i += 1024)
      if (init || owner[i] != -1) owner[i] = ownerFree;
    __syncthreads();
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = (has0 && j == tid) ? c0 : J.choice[j];
      if (c < 0) continue;
      if (init) J.link[j] = atomicExch(&owner[c & 0xffff], j);
      else if ((has0 && j == tid) ? bv0 : (!J.blockVal || J.blockVal[j])) atomicMin(&owner[c], j);
    }
    __syncthreads();
    if (rounds == 0 && J.mode == CLAIM_BEST) {
      // the usual case ends here: when no query's first choice is held by an earlier one, every query keeps the best of ALL
      // its candidates and a second round would change nothing.  (Not so with the ratio test, whose outcome also depends on
      // whether the SECOND best is still free.)
      int conflict = 0;
      for (int j = tid; j < J.nq; j += 1024) {
        const int c = (has0 && j == tid) ? c0 : J.choice[j];
        if (c >= 0 && owner[c] < j) conflict = 1;
      }
      if (conflict) s_conflict = 1;
      __syncthreads();
      if (!s_conflict) break;  // block-uniform
    }
  }
  // ---- the match array, the rotation histogram (ComputeThreeMaxima, :1635-1690) and the count ----
  const int nOut = init ? J.nq : J.n;
  if (!init)
    for (int i = tid; i < nOut; i += 1024) J.match[i] = -1;
  __syncthreads();
  int ev = 0;
  for (int j = tid; j < J.nq; j += 1024) {
    const int c = J.choice[j];
    int m = -1;
    if (c >= 0) {
      const int f = init ? (c & 0xffff) : c;
      if (J.checkOri) atomicAdd(&s_hist[claim_rot_bin(J.qAngle[j], J.fAngle[f])], 1);  // every take is pushed (:540-553, :1601-1610)
      if (init) {
        bool holder = true;  // a later query that took the feature replaced this one (:529-533)
        for (int i = owner[f]; i >= 0; i = J.link[i])
          if (i > j) { const int ci = J.choice[i]; if (ci >= 0 && (ci & 0xffff) == f) holder = false; }
        if (holder) { m = f; ev++; }
      } else {
        atomicMax(&J.match[c], j);  // a feature whose holder does not block is overwritten by the later ones
        ev++;
      }
    }
    if (init) J.match[j] = m;
  }
  if (ev) atomicAdd(&s_total, ev);
  __syncthreads();
  if (J.checkOri) {
    if (tid == 0) {
      int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
      for (int i = 0; i < 30; i++) {
        const int s = s_hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
        else if (s > max3) { max3 = s; i3 = i; }
      }
      if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
      else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
      s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    const int k0 = s_keep[0], k1 = s_keep[1], k2 = s_keep[2];
    int pr = 0;
    for (int j = tid; j < J.nq; j += 1024) {
      const int c = J.choice[j];
      if (c < 0) continue;
      const int f = init ? (c & 0xffff) : c;
      const int b = claim_rot_bin(J.qAngle[j], J.fAngle[f]);
      if (b == k0 || b == k1 || b == k2) continue;
      if (init) { if (J.match[j] >= 0) { J.match[j] = -1; pr++; } }   // :557-563: only a still-matched entry counts
      else { J.match[c] = -1; pr++; }                                  // :1617-1625: every entry of a dropped bin counts
    }
    if (pr) atomicAdd(&s_pruned, pr);
    __syncthreads();
  }
  if (init && J.prevX)
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

void launch_window_claim(hipStream_t s, const ClaimJob* d_jobs, int nJobs, size_t ldsBytes) {
  if (nJobs <= 0) return;
  hipLaunchKernelGGL(k_window_claim, dim3(nJobs), dim3(1024), ldsBytes, s, d_jobs);
}

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
