// k_octree.hip -- ORBextractor::DistributeOctTree (src/ORBextractor.cc:566-808) on the device:
// one 256-thread workgroup per (frame, pyramid level), node list in LDS.
//
// The reference's sequential std::list algorithm is restated in "generation" form:
//   * keys never move: a key's node is nodeOf[key]; a node's key order is always ascending
//     candidate index (root bucketing and DivideNode are stable), so "first key with the max
//     response" (:787-805) is max(response) then min(candidate index) -- one atomicMax per key;
//   * a split pass maps the node list to  reverse(children created, in creation order) ++
//     survivors in order,  which is what push_front/erase produce;
//   * the largest-first passes (:714-782) sort the expandable nodes of the previous pass by
//     (count, pointer) ascending and walk from the back; nodes created later sit EARLIER in the
//     list, so with creation order standing in for the pointer value that walk is
//     "count descending, list position ascending".  All of them are split speculatively
//     (4-way key histogram), a prefix sum over the walk order finds the node at which the
//     list reaches N (the reference's early break, :774-775).
// Everything is integer except the root bucketing (float divide, :598) -- identical to the host.
#include <cstdlib>

#include "kernels.h"

namespace orbfe {

namespace {

struct Rect { int16_t x0, x1, y0, y1; };

// exclusive scan of a[0..len) in place (LDS) by the T threads of the workgroup; returns the total to every thread
// kDpp: the wave scan in the DPP network (kernels.h) instead of six ds_bpermute steps.  The single-frame forms take it (their
// critical path is these scans: gather + octree 55 -> 50 us per KITTI frame); the throughput form keeps the LDS permutes -- with
// DPP its stage was 6 % shorter alone (0.633 -> 0.596 ms per 1024 KITTI frames) but the pipelined step 1.5 % LONGER in a same-box
// A/B (104.7 k vs 103.0 k stereo frames/s, twice): a wave that waits for the LDS crossbar leaves the VALU port to the FAST and
// blur waves next to it, a wave walking a DPP chain (with its wait states, at s_setprio 3) does not.
template <bool kDpp>
__device__ __forceinline__ int octree_wave_incl_scan(int v, int lane) {
  if constexpr (kDpp) return wave_incl_scan_dpp(v);
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  return x;
}
template <int T, bool kDpp>
__device__ int block_scan_excl(int* a, int len, int* waveTot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int run = 0;
  for (int base = 0; base < len; base += T) {
    const int i = base + tid;
    const int v = i < len ? a[i] : 0;
    const int x = octree_wave_incl_scan<kDpp>(v, lane);
    if (T == 64) {  // one wavefront: the running total lives in a register
      const int tot = __builtin_amdgcn_readlane(x, 63);
      if (i < len) a[i] = run + x - v;
      run += tot;
    } else {
      if (lane == 63) waveTot[wave] = x;
      __syncthreads();
      int b = run;
      for (int w = 0; w < wave; w++) b += waveTot[w];
#pragma unroll
      for (int w = 0; w < T / 64; w++) run += waveTot[w];
      if (i < len) a[i] = b + x - v;
      __syncthreads();
    }
  }
  if (T == 64) __syncthreads();  // orders the LDS stores before the callers' reads (a wave-level fence for 64 threads)
  return run;
}

// single-wavefront steps inside a larger workgroup (HYB form of octree_body): LDS accesses of one wave are performed
// in program order, so a compiler-level fence is all "the other lanes' stores are visible" needs -- no s_barrier
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exclusive scan of a[0..len) in place by ONE wavefront (lane = 0..63); returns the total to every lane
template <bool kDpp>
__device__ __forceinline__ int wave_scan_excl(int* a, int len, int lane) {
  int run = 0;
  for (int base = 0; base < len; base += 64) {
    const int i = base + lane;
    const int v = i < len ? a[i] : 0;
    const int x = octree_wave_incl_scan<kDpp>(v, lane);
    if (i < len) a[i] = run + x - v;
    run += __builtin_amdgcn_readlane(x, 63);
  }
  wave_sync();
  return run;
}

__device__ __forceinline__ int quadrant(const Rect r, int x, int y) {
  const int mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2), :500
  const int my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
  return (x < mx ? 0 : 1) + (y < my ? 0 : 2);      // n1,n2,n3,n4 (:535-545)
}
__device__ __forceinline__ Rect child_rect(const Rect r, int q) {
  const int mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1);
  const int my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
  Rect c;
  c.x0 = (int16_t)((q & 1) ? mx : r.x0);
  c.x1 = (int16_t)((q & 1) ? r.x1 : mx);
  c.y0 = (int16_t)((q & 2) ? my : r.y0);
  c.y1 = (int16_t)((q & 2) ? r.y1 : my);
  return c;
}

}  // namespace

// REG: the candidates of the level (n <= 256 * kRegCand) live in registers -- candidate j*256 + tid in
// slot j -- instead of being re-read from / re-written to global memory in every pass: a pass is a
// chain of dependent LDS accesses only (measured on one 640x480 frame, level 0, n = 2262: the
// histogram pass 11 k -> 3 k cycles, re-homing 6.4 k -> 2 k).  Larger candidate sets (noise images)
// take the global-memory form of the same loops.
constexpr int kRegCand = 16;

template <bool REG, int T, bool HYB = false, int RC = kRegCand>
__device__ __forceinline__ void octree_body(const OctreeArgs& a, uint8_t* smem, int* waveTot, int* sh, const int l, const int f) {
  const int tid = threadIdx.x;
  constexpr bool kDpp = REG || T == 1024;  // the single-frame forms (see octree_wave_incl_scan)
  const LevelGeom g = a.lvg[l];
  const int M = a.maxL;
  // LDS carve (M entries each)
  Rect* rect[2];
  int* cnt[2];
  rect[0] = reinterpret_cast<Rect*>(smem);
  rect[1] = rect[0] + M;
  cnt[0] = reinterpret_cast<int*>(rect[1] + M);
  cnt[1] = cnt[0] + M;
  int* child = cnt[1] + M;        // [4*M] child key counts, later child list positions
  int* scanA = child + 4 * M;     // [M] scans over processing order
  int* scanB = scanA + M;         // [M] scans over list order
  uint16_t* order = reinterpret_cast<uint16_t*>(scanB + M);  // [M] processing order -> list position
  uint16_t* rankOf = order + M;   // [M] list position -> processing rank (valid where inS)
  uint8_t* inS = reinterpret_cast<uint8_t*>(rankOf + M);      // [M]

  const int n = a.candCount[(size_t)f * a.nlevels + l];
  const Candidate* cand = a.cand + (size_t)f * a.slotsPerFrame + g.slotStart;
  uint16_t* nodeOf = a.nodeOf + (size_t)f * a.slotsPerFrame + g.slotStart;
  LevelKp* out = a.levelKp + (size_t)f * a.kpSlotsPerFrame + g.kpStart;
  int32_t* outCount = a.levelCount + (size_t)f * a.nlevels + l;
  const int N = g.quota;
  const int nIni = g.nIni;
  if (n <= 0 || nIni <= 0) { if (tid == 0) *outCount = 0; return; }

  const int bw = g.w - 2 * kMinBorder, bh = g.h - 2 * kMinBorder;  // maxX-minX, maxY-minY
  const float hX = __fdiv_rn((float)bw, (float)nIni);               // :572

  // ---- roots (:576-619) ----
  int cur = 0;
  for (int i = tid; i < nIni; i += T) scanB[i] = 0;
  __syncthreads();
  uint32_t kxy[RC];   // REG: candidate coordinates / current node of candidate j*256 + tid
  uint32_t knode[RC];
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < RC; j++) {
      const int k = j * T + tid;
      kxy[j] = 0; knode[j] = 0;
      if (k < n) {
        kxy[j] = cand[k].xy;
        int b = (int)__fdiv_rn((float)(kxy[j] & 0xffffu), hX);
        if (b >= nIni) b = nIni - 1;
        knode[j] = (uint32_t)b;
      }
      // root counts: one LDS atomic per wave and root instead of one per candidate on 1-3 addresses
      for (int b = 0; b < nIni; b++) {
        const unsigned long long m = __ballot(k < n && knode[j] == (uint32_t)b);
        if ((tid & 63) == 0 && m) atomicAdd(&scanB[b], __popcll(m));
      }
    }
  } else {
    for (int k0 = tid; k0 < n; k0 += 4 * T) {
      uint32_t xy4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) xy4[u] = k0 + u * T < n ? cand[k0 + u * T].xy : 0u;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * T;
        if (k >= n) continue;
        int b = (int)__fdiv_rn((float)(xy4[u] & 0xffffu), hX);
        if (b >= nIni) b = nIni - 1;
        nodeOf[k] = (uint16_t)b;
        atomicAdd(&scanB[b], 1);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nIni; i += T) { scanA[i] = scanB[i] > 0 ? 1 : 0; }
  __syncthreads();
  int L = block_scan_excl<T, kDpp>(scanA, nIni, waveTot);  // scanA[i] = list position of root i (if non-empty)
  for (int i = tid; i < nIni; i += T) {
    if (scanB[i] > 0) {
      Rect r;
      r.x0 = (int16_t)(int)__fmul_rn(hX, (float)i);
      r.x1 = (int16_t)(int)__fmul_rn(hX, (float)(i + 1));
      r.y0 = 0;
      r.y1 = (int16_t)bh;
      rect[cur][scanA[i]] = r;
      cnt[cur][scanA[i]] = scanB[i];
    }
  }
  __syncthreads();
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < RC; j++) knode[j] = (uint32_t)scanA[knode[j]];
  } else {
    for (int k = tid; k < n; k += T) nodeOf[k] = (uint16_t)scanA[nodeOf[k]];
  }
  __syncthreads();

  int C = 0;            // nodes at the head of the list created by the previous pass
  bool phase2 = false;
  for (;;) {
    const int prevSize = L;
    Rect* rc = rect[cur];
    int* cn = cnt[cur];
    // A. candidates of this pass
    for (int p = tid; p < L; p += T) {
      const bool c = phase2 ? (p < C && cn[p] > 1) : (cn[p] > 1);
      inS[p] = c ? 1 : 0;
      child[4 * p] = child[4 * p + 1] = child[4 * p + 2] = child[4 * p + 3] = 0;
    }
    __syncthreads();
    // B. speculative 4-way histogram of the candidates' keys (DivideNode :531-546)
    int kbin[RC];  // REG: child counter this candidate voted for in this pass (-1: its node is not split)
    if constexpr (REG) {
      // four candidates at a time: their node flags and rectangles are requested together, so a pass
      // is 4 x (one LDS round trip + 4 atomics) instead of 16 dependent round trips
#pragma unroll
      for (int j0 = 0; j0 < RC; j0 += 4) {
        uint8_t in4[4];
        Rect r4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int p = (int)knode[j0 + u];
          in4[u] = inS[p];
          r4[u] = rc[p];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = j0 + u;
          const int p = (int)knode[j];
          int bin = -1;
          if (j * T + tid < n && in4[u]) bin = 4 * p + quadrant(r4[u], (int)(kxy[j] & 0xffffu), (int)(kxy[j] >> 16));
          kbin[j] = bin;
          if (bin >= 0) atomicAdd(&child[bin], 1);
        }
      }
    } else {
      // four keys per thread and trip: their node ids and coordinates are requested together and unconditionally, so a
      // trip is ONE memory round trip instead of two dependent ones per key (node id -> is it split? -> coordinates);
      // the sweeps over the keys are most of a pass (level 0 of a 1241 x 376 frame: ~5000 keys, 20 per thread)
      for (int k0 = tid; k0 < n; k0 += 4 * T) {
        int p4[4];
        uint32_t xy4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u * T;
          p4[u] = k < n ? (int)nodeOf[k] : 0;
          xy4[u] = k < n ? cand[k].xy : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k0 + u * T < n && inS[p4[u]])
            atomicAdd(&child[4 * p4[u] + quadrant(rc[p4[u]], (int)(xy4[u] & 0xffffu), (int)(xy4[u] >> 16))], 1);
      }
    }
    __syncthreads();
    int Cn, nSurv;
    Rect* rn = rect[cur ^ 1];
    int* cnn = cnt[cur ^ 1];
    if constexpr (HYB) {
      // C-E on the node list belong to wave 0 alone: ~20 short dependent LDS steps that cost a workgroup barrier each
      // when 256 threads share them (a barrier is ~1 us with 8 workgroups of 4 waves on the CU) and only a compiler
      // fence when one wave walks the list 64 nodes at a time.  The other waves wait at the barrier before pass F; the
      // phase-2 ranking, quadratic in the candidates, stays on all waves.
      const int lane = tid & 63, wave = tid >> 6;
      if (phase2) {
        for (int p = wave; p < C; p += T / 64) {
          if (!inS[p]) continue;  // wave-uniform
          const int c = cn[p];
          int r = 0;
          for (int p0 = 0; p0 < C; p0 += 64) {
            const int p2 = p0 + lane;
            bool before = false;
            if (p2 < C && inS[p2]) { const int c2 = cn[p2]; before = (c2 > c) || (c2 == c && p2 < p); }
            r += __popcll(__ballot(before));
          }
          if (lane == 0) { order[r] = (uint16_t)p; rankOf[p] = (uint16_t)r; }
        }
        __syncthreads();
      }
      if (wave == 0) {
        // (round 4, measured and NOT kept: these loops with FOUR entries per lane and trip, loads in front of any use, so that
        // the two or three dependent LDS round trips of an entry overlap -- this section is 13 k of the 26 k cycles of a pass
        // at level 0 of a KITTI frame.  Bit-equal, and slower everywhere: stage 0.62 -> 0.78 ms per 1024 KITTI frames, 1.19 ->
        // 1.64 per 4096 VGA frames, single frame 50 -> 54 us: most passes walk lists shorter than one trip, and the
        // unrolled, predicated form issues four entries' instructions for them)
        int m;
        if (!phase2) {
          for (int p = lane; p < L; p += 64) scanA[p] = inS[p];
          wave_sync();
          m = wave_scan_excl<kDpp>(scanA, L, lane);
          for (int p = lane; p < L; p += 64)
            if (inS[p]) { order[scanA[p]] = (uint16_t)p; rankOf[p] = (uint16_t)scanA[p]; }
          wave_sync();
        } else {
          int E = 0;  // candidates of this pass (all have p < C)
          for (int p0 = 0; p0 < C; p0 += 64) E += __popcll(__ballot(p0 + lane < C && inS[p0 + lane]));
          // growth of the list per split: (#non-empty children - 1), in walk order
          for (int j = lane; j < E; j += 64) {
            const int p = order[j];
            scanA[j] = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0) - 1;
          }
          wave_sync();
          wave_scan_excl<kDpp>(scanA, E, lane);  // scanA[j] = growth before split j
          // split j happens iff the list is still < N before it: L + scanA[j] < N (early break :774)
          m = 0;
          for (int j0 = 0; j0 < E; j0 += 64) m += __popcll(__ballot(j0 + lane < E && L + scanA[j0 + lane] < N));
          for (int j = m + lane; j < E; j += 64) inS[order[j]] = 0;
          wave_sync();
        }
        // D. creation index of the children: exclusive scan of #non-empty children over the walk order
        for (int j = lane; j < m; j += 64) {
          const int p = order[j];
          scanA[j] = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0);
        }
        for (int p = lane; p < L; p += 64) scanB[p] = inS[p] ? 0 : 1;
        wave_sync();
        const int cn0 = wave_scan_excl<kDpp>(scanA, m, lane);
        const int ns0 = wave_scan_excl<kDpp>(scanB, L, lane);
        // E. write the next list: reverse(created) ++ survivors
        int expand = 0;
        for (int j = lane; j < m; j += 64) {
          const int p = order[j];
          int ci = scanA[j];
          const Rect r = rc[p];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int c = child[4 * p + q];
            if (c > 0) {
              const int np = cn0 - 1 - ci;
              rn[np] = child_rect(r, q);
              cnn[np] = c;
              child[4 * p + q] = np;
              expand += (c > 1);
              ci++;
            }
          }
        }
        for (int p = lane; p < L; p += 64)
          if (!inS[p]) {
            const int np = cn0 + scanB[p];
            rn[np] = rc[p];
            cnn[np] = cn[p];
            scanB[p] = np;
          }
        if constexpr (kDpp) expand = wave_sum_dpp(expand);
        else {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) expand += __shfl_xor(expand, o, 64);
        }
        if (lane == 0) { sh[0] = cn0; sh[1] = ns0; sh[2] = expand; }
      }
      __syncthreads();
      Cn = sh[0];
      nSurv = sh[1];
    } else {
      // C. processing order
      int m;  // nodes actually split in this pass
      if (!phase2) {
        for (int p = tid; p < L; p += T) scanA[p] = inS[p];
        __syncthreads();
        m = block_scan_excl<T, kDpp>(scanA, L, waveTot);
        for (int p = tid; p < L; p += T)
          if (inS[p]) { order[scanA[p]] = (uint16_t)p; rankOf[p] = (uint16_t)scanA[p]; }
        __syncthreads();
      } else {
        // rank by (count desc, list position asc) among the candidates (all have p < C)
        // one wavefront per candidate p (p = wave, wave+4, ...), its lanes sweep p2: a ballot counts
        // the candidates that walk before p
        int nE = 0;
        {
          const int lane = tid & 63, wave = tid >> 6;
          for (int p = wave; p < C; p += T / 64) {
            if (!inS[p]) continue;  // wave-uniform
            const int c = cn[p];
            int r = 0;
            for (int p0 = 0; p0 < C; p0 += 64) {
              const int p2 = p0 + lane;
              bool before = false;
              if (p2 < C && inS[p2]) { const int c2 = cn[p2]; before = (c2 > c) || (c2 == c && p2 < p); }
              r += __popcll(__ballot(before));
            }
            if (lane == 0) { order[r] = (uint16_t)p; rankOf[p] = (uint16_t)r; nE++; }
          }
        }
        if (tid == 0) sh[0] = 0;
        __syncthreads();
        if (nE) atomicAdd(&sh[0], nE);
        __syncthreads();
        const int E = sh[0];
        // growth of the list per split: (#non-empty children - 1), in walk order
        for (int j = tid; j < E; j += T) {
          const int p = order[j];
          scanA[j] = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0) - 1;
        }
        __syncthreads();
        block_scan_excl<T, kDpp>(scanA, E, waveTot);  // scanA[j] = growth before split j
        // split j happens iff the list is still < N before it: L + scanA[j] < N (early break :774)
        if (tid == 0) sh[1] = 0;
        __syncthreads();
        int mine = 0;
        for (int j = tid; j < E; j += T) mine += (L + scanA[j] < N) ? 1 : 0;
        if (mine) atomicAdd(&sh[1], mine);
        __syncthreads();
        m = sh[1];
        for (int j = tid; j < E; j += T)
          if (j >= m) inS[order[j]] = 0;
        __syncthreads();
      }
      // D. creation index of the children: exclusive scan of #non-empty children over the walk order
      for (int j = tid; j < m; j += T) {
        const int p = order[j];
        scanA[j] = (child[4 * p] > 0) + (child[4 * p + 1] > 0) + (child[4 * p + 2] > 0) + (child[4 * p + 3] > 0);
      }
      for (int p = tid; p < L; p += T) scanB[p] = inS[p] ? 0 : 1;
      __syncthreads();
      Cn = block_scan_excl<T, kDpp>(scanA, m, waveTot);
      nSurv = block_scan_excl<T, kDpp>(scanB, L, waveTot);
      // E. write the next list: reverse(created) ++ survivors
      if (tid == 0) sh[2] = 0;
      __syncthreads();
      int expand = 0;
      for (int j = tid; j < m; j += T) {
        const int p = order[j];
        int ci = scanA[j];
        const Rect r = rc[p];
  #pragma unroll
        for (int q = 0; q < 4; q++) {
          const int c = child[4 * p + q];
          if (c > 0) {
            const int np = Cn - 1 - ci;
            rn[np] = child_rect(r, q);
            cnn[np] = c;
            child[4 * p + q] = np;
            expand += (c > 1);
            ci++;
          }
        }
      }
      for (int p = tid; p < L; p += T)
        if (!inS[p]) {
          const int np = Cn + scanB[p];
          rn[np] = rc[p];
          cnn[np] = cn[p];
          scanB[p] = np;
        }
      if (expand) atomicAdd(&sh[2], expand);
      __syncthreads();
    }
    // F. re-home the keys
    if constexpr (REG) {
      // a split that the early break cancelled (inS cleared in C) keeps its node; the vote of pass B
      // already names the child otherwise
#pragma unroll
      for (int j0 = 0; j0 < RC; j0 += 4) {
        uint8_t in4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) in4[u] = inS[knode[j0 + u]];
        int v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = j0 + u;
          const int* srcp = (in4[u] && kbin[j] >= 0) ? &child[kbin[j]] : &scanB[knode[j]];
          v4[u] = *srcp;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if ((j0 + u) * T + tid < n) knode[j0 + u] = (uint32_t)v4[u];
      }
    } else {
      for (int k0 = tid; k0 < n; k0 += 4 * T) {  // batched like pass B
        int p4[4];
        uint32_t xy4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u * T;
          p4[u] = k < n ? (int)nodeOf[k] : 0;
          xy4[u] = k < n ? cand[k].xy : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u * T;
          if (k >= n) continue;
          const int p = p4[u];
          const int np = inS[p] ? child[4 * p + quadrant(rc[p], (int)(xy4[u] & 0xffffu), (int)(xy4[u] >> 16))] : scanB[p];
          nodeOf[k] = (uint16_t)np;
        }
      }
    }
    const int nToExpand = sh[2];
    __syncthreads();
    cur ^= 1;
    L = Cn + nSurv;
    C = Cn;
    // termination, :707-714 / :780-781
    if (L >= N || L == prevSize) break;
    if (!phase2 && L + nToExpand * 3 > N) phase2 = true;
  }

  // ---- best response per node, first key wins ties (:787-805) ----
  int* best = scanA;
  for (int p = tid; p < L; p += T) best[p] = 0;
  __syncthreads();
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < RC; j++) {
      const int k = j * T + tid;
      if (k < n)
        atomicMax(reinterpret_cast<unsigned int*>(&best[knode[j]]), (cand[k].score << 24) | (0xffffffu - (unsigned)k));
    }
  } else {
    for (int k0 = tid; k0 < n; k0 += 4 * T) {
      int p4[4];
      uint32_t sc4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + u * T;
        p4[u] = k < n ? (int)nodeOf[k] : 0;
        sc4[u] = k < n ? cand[k].score : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (k0 + u * T < n)
          atomicMax(reinterpret_cast<unsigned int*>(&best[p4[u]]), (sc4[u] << 24) | (0xffffffu - (unsigned)(k0 + u * T)));
    }
  }
  __syncthreads();
  // The keypoints leave in SPATIAL order (128-byte column strip, then row): consecutive slots are
  // what one k_orient_desc workgroup processes, and neighbours in memory share the cache lines of
  // their patches.  `rank` keeps the reference's list position, which decides the output row.
  const int nKp = L < g.kpCap ? L : g.kpCap;
  uint32_t* skey = reinterpret_cast<uint32_t*>(scanB);
  for (int p = tid; p < nKp; p += T) {
    const Candidate c = cand[0xffffffu - ((unsigned)best[p] & 0xffffffu)];
    const uint32_t x = (c.xy & 0xffffu) + kMinBorder, y = (c.xy >> 16) + kMinBorder;  // :909-910
    skey[p] = ((x >> 7) << 23) | (y << 7) | (x & 127u);  // strip (9 bits), row (16), column inside the strip (7): unique for x, y < 65536
  }
  __syncthreads();
  for (int p = tid; p < nKp; p += T) {
    const uint32_t key = skey[p];
    int pos = 0;
#pragma unroll 8
    for (int q = 0; q < nKp; q++) pos += skey[q] < key;  // keys are distinct (one keypoint per pixel)
    const Candidate c = cand[0xffffffu - ((unsigned)best[p] & 0xffffffu)];
    LevelKp o;
    o.x = (uint16_t)((c.xy & 0xffffu) + kMinBorder);
    o.y = (uint16_t)((c.xy >> 16) + kMinBorder);
    o.score = (uint16_t)c.score;
    o.rank = (uint16_t)p;
    out[pos] = o;
  }
  if (tid == 0) *outCount = nKp;
}

// Two builds of the same body: k_octree keeps the candidates in global memory (28 VGPRs, as many
// workgroups per CU as LDS allows -- the throughput form for large batches), k_octree_reg keeps them
// in registers when a level has at most 256 * kRegCand candidates (~100 VGPRs, 4 workgroups per CU,
// but a 20 % shorter critical path -- the latency form for the live-camera case of a few frames).
__global__ __launch_bounds__(256) void k_octree(OctreeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int waveTot[4];
  __shared__ int sh[4];
  // Throughput form: a 1-D grid of gridDim.x workgroups walks the (level, frame) items LEVEL-MAJOR: all frames of level
  // 0 first, so the long items start first and consecutive workgroups -- which go to consecutive XCDs -- hold equal
  // work.  Round 1 launched a (level, frame) grid: level l of every frame ran on XCD l (8 levels, 8 XCDs) and the XCD
  // with the level-0 items decided the duration: stage 1.15 -> 0.70 ms per 1024 KITTI frames, 2.05 -> 1.43 ms per 4096
  // VGA frames from the ordering alone.  With gridDim.x below the item count ($ORBFE_OCTREE_GRID workgroups per CU) the
  // grid size, not the LDS, bounds the workgroups in flight, and the rest of each CU stays free for the other streams.
  ORBFE_LATENCY_KERNEL_PRIO();
  const int nItems = a.nlevels * a.nFrames;
  for (int it = blockIdx.x; it < nItems; it += gridDim.x) {
    const int l = it / a.nFrames, f = it - l * a.nFrames;
    octree_body<false, 256, true>(a, smem, waveTot, sh, l, f);
    __syncthreads();  // the next item reuses the LDS arrays
  }
}

// Measured and dropped (round 2): octree_body<false, 64>, ONE wavefront per (frame, level) with no barrier at all.  The
// number of resident (frame, level) problems is set by their LDS node lists (6 per CU at 2000 features), not by wave
// slots, so a problem that is worked on by one wave instead of four just takes longer: stage 1.15 -> 2.97 ms per 1024
// KITTI frames, 2.05 -> 2.84 ms per 4096 VGA frames, pipeline -17 % / -7 %.  The body stays a template on the size.
__global__ __launch_bounds__(256) void k_octree_reg(OctreeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int waveTot[4];
  __shared__ int sh[4];
  const int n = a.candCount[(size_t)blockIdx.y * a.nlevels + blockIdx.x];  // block-uniform
  if (n <= 256 * kRegCand) octree_body<true, 256, true>(a, smem, waveTot, sh, blockIdx.x, blockIdx.y);
  else octree_body<false, 256, true>(a, smem, waveTot, sh, blockIdx.x, blockIdx.y);
}

// The latency form with 1024 threads and up to 8192 candidates in registers (8 per thread): level 0 of a 1241 x 376 frame
// lists ~5000 candidates -- more than 256 x 16 -- and fell back to the global-memory sweeps, two dependent HBM round trips
// per pass: 82 of the 167 us of kernel time of a single KITTI frame.
__global__ __launch_bounds__(1024) void k_octree_reg1024(OctreeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int waveTot[16];
  __shared__ int sh[4];
  if (a.gCells) {
    // ---- the ordered compaction of k_gather_candidates (k_fast.hip) for this (frame, level): cells in cell-row-major
    //      order, raster inside each cell ----
    const int l = blockIdx.x, f = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LevelGeom g = a.lvg[l];
    const uint16_t* cnt = a.gCellCount + (size_t)f * a.gCellsPerFrame + g.cellStart;
    int32_t* pre = a.gCellPrefix + (size_t)f * a.gCellsPerFrame + g.cellStart;
    int run = 0;  // identical in every thread
    for (int cb = 0; cb < g.nCells; cb += 1024) {
      const int c = cb + tid;
      const int v = c < g.nCells ? cnt[c] : 0;
      const int x = wave_incl_scan_dpp(v);
      if (lane == 63) waveTot[wave] = x;
      __syncthreads();
      int base = run;
      for (int w = 0; w < wave; w++) base += waveTot[w];
#pragma unroll
      for (int w = 0; w < 16; w++) run += waveTot[w];
      if (c < g.nCells) pre[c] = base + x - v;
      __syncthreads();
    }
    if (tid == 0) a.gCandCount[(size_t)f * a.nlevels + l] = run;
    const Candidate* sl = a.gSlots + (size_t)f * a.slotsPerFrame;
    Candidate* out = a.gCand + (size_t)f * a.slotsPerFrame + g.slotStart;
    for (int t = tid; t < 4 * g.nCells; t += 1024) {  // four threads per cell
      const int c = t >> 2, part = t & 3;
      const int n = cnt[c], b = pre[c];
      const Candidate* src = sl + a.gCells[g.cellStart + c].slotBase;
      for (int i = part; i < n; i += 4) out[b + i] = src[i];
    }
    __syncthreads();  // (the workgroup's own stores to cand / candCount are visible to it behind the barrier)
  }
  const int n = a.candCount[(size_t)blockIdx.y * a.nlevels + blockIdx.x];  // block-uniform
  // (node-list phases on wave 0, as in the other forms: shared by all sixteen waves -- HYB = false -- the stage takes 70
  // instead of 55 us per KITTI frame, 52 instead of 37 per VGA frame, same box)
  if (n <= 1024 * 8) octree_body<true, 1024, true, 8>(a, smem, waveTot, sh, blockIdx.x, blockIdx.y);
  else octree_body<false, 1024, true>(a, smem, waveTot, sh, blockIdx.x, blockIdx.y);
}

// Third build: the node list does not fit in LDS (a level asked for more than ~2 890 keypoints, e.g. 3000 features
// on a 1-level pyramid): the same generation passes with the per-workgroup arrays in a global-memory slab.  Every
// array access becomes a flat load / store / atomic and __syncthreads() orders them inside the workgroup -- slow
// next to the LDS form, but it is the rare configuration and it stays on the device (no CPU path).
__global__ __launch_bounds__(256) void k_octree_global(OctreeArgs a) {
  __shared__ int waveTot[4];
  __shared__ int sh[4];
  const int f = blockIdx.x, l = blockIdx.y;  // frames along x: one level's items spread over the XCDs
  uint8_t* slab = a.work + ((size_t)f * a.nlevels + l) * a.workStride;
  octree_body<false, 256>(a, slab, waveTot, sh, l, f);
}

size_t octree_lds_bytes(int maxL) {
  // 2 rect (8) + 2 cnt (4) + child (16) + scanA (4) + scanB (4) + order (2) + rankOf (2) + inS (1)
  return (size_t)maxL * (2 * 8 + 2 * 4 + 16 + 4 + 4 + 2 + 2 + 1) + 64;
}

namespace {
int octree_latency_threads() {
  static const int t = getenv("ORBFE_OCTREE_T") ? atoi(getenv("ORBFE_OCTREE_T")) : 1024;
  return t;
}
}  // namespace
bool octree_gathers(int nFrames, int maxL) {
  static const bool off = getenv("ORBFE_OCTREE_GATHER") && atoi(getenv("ORBFE_OCTREE_GATHER")) == 0;
  return !off && nFrames > 0 && nFrames <= 8 && octree_lds_bytes(maxL) <= kOctreeLdsLimit && octree_latency_threads() == 1024;
}

hipError_t launch_octree(hipStream_t s, const OctreeArgs& args, int nlevels, int nFrames) {
  if (nFrames <= 0) return hipSuccess;
  OctreeArgs a = args;
  a.nFrames = nFrames;
  size_t lds = octree_lds_bytes(a.maxL);
  if (lds > kOctreeLdsLimit) {  // node list in global memory
    if (!a.work || a.workStride < lds) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_octree_global, dim3(nFrames, nlevels), dim3(256), 0, s, a);
    return hipSuccess;
  }
  const bool latencyForm = nFrames <= 8;  // too few workgroups to fill the GPU anyway
  static const size_t pad = occupancy_pad_bytes("OCTREE", 0);
  if (!latencyForm && lds + pad <= 64 * 1024) lds += pad;
  static thread_local size_t configured[3] = {0, 0, 0};
  // (the 1024-thread form for whole batches, measured once: octree stage 0.63 -> 1.81 ms per 1024 KITTI frames, 1.19 -> 3.52 per
  // 4096 VGA frames, pipelines -17 % -- sixteen waves per problem leave room for two problems per CU instead of six)
  const bool wide = latencyForm && octree_latency_threads() == 1024;
  if (a.gCells && !wide) return hipErrorInvalidValue;  // (only the wide single-frame form gathers)
  const void* fn = wide ? reinterpret_cast<const void*>(k_octree_reg1024)
                        : (latencyForm ? reinterpret_cast<const void*>(k_octree_reg) : reinterpret_cast<const void*>(k_octree));
  const int which = wide ? 2 : (latencyForm ? 1 : 0);
  if (lds > 64 * 1024 && lds > configured[which]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured[which] = lds;
  }
  if (wide) hipLaunchKernelGGL(k_octree_reg1024, dim3(nlevels, nFrames), dim3(1024), lds, s, a);
  else if (latencyForm) hipLaunchKernelGGL(k_octree_reg, dim3(nlevels, nFrames), dim3(256), lds, s, a);
  else {
    // workgroups per CU of the persistent grid; 0 = one workgroup per (frame, level).  Round 2 capped it at 4; with the
    // round-3 kernels the pipelines measure the same with any cap (KITTI 105.6 k vs 104.0 k, TUM 379.9 k vs 382.9 k, EuRoC
    // 262.0 k vs 260.0 k stereo frames / frames per second for 0 vs 4) and the stage alone is 22 % faster without one
    // (0.62 vs 0.81 ms per 1024 KITTI frames): no cap by default
    static const int perCu = getenv("ORBFE_OCTREE_GRID") ? atoi(getenv("ORBFE_OCTREE_GRID")) : 0;
    unsigned grid = (unsigned)nlevels * (unsigned)nFrames;
    if (perCu > 0 && (unsigned)perCu * 256u < grid) grid = (unsigned)perCu * 256u;
    hipLaunchKernelGGL(k_octree, dim3(grid), dim3(256), lds, s, a);
  }
  return hipSuccess;
}

}  // namespace orbfe
