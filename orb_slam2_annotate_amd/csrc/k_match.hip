// k_match.hip -- 256-bit Hamming matching kernels (ORBmatcher, src/ORBmatcher.cc; stereo,
// src/Frame.cc:512-686).  Pure XOR + v_bcnt popcount work, no MFMA: one 64-lane wavefront
// owns a query descriptor (held in 8 SGPR/VGPR dwords), its lanes stride over the candidate
// descriptors (32 B each, two 16-B loads per lane) and the best/second-best bookkeeping of
// the reference's sequential scan is reproduced by an order-preserving wave reduction:
// keys are (distance << 16 | scan position), so "first minimum wins" / "last minimum wins"
// are a plain min over the wave.
#include "kernels.h"
#include "match_kernels.h"

namespace orbfe {

namespace {
struct Desc { uint32_t w[8]; };

__device__ __forceinline__ Desc load_desc(const uint8_t* base, uint32_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)idx * 32);
  const uint4 a = p[0], b = p[1];
  Desc d;
  d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w;
  d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
  return d;
}
__device__ __forceinline__ int hdist(const Desc& a, const Desc& b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a.w[i] ^ b.w[i]);
  return d;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_min_u32_dpp(v); }  // (kernels.h: DPP, not ds_bpermute)
// rotation-histogram bin, src/ORBmatcher.cc:272-279 (C round(): half away from zero)
__device__ __forceinline__ int rot_bin(float a1, float a2) {
  const float factor = 1.0f / 30;
  float rot = __fsub_rn(a1, a2);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, factor));
  if (bin == 30) bin = 0;
  return bin;
}
}  // namespace

__global__ __launch_bounds__(256) void k_hamming_pairs(const uint8_t* __restrict__ a,
                                                       const uint8_t* __restrict__ b, int n,
                                                       int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = hdist(load_desc(a, i), load_desc(b, i));
}
void launch_hamming_pairs(hipStream_t s, const uint8_t* a, const uint8_t* b, int n, int32_t* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, out);
}

__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t* __restrict__ d1, int n1,
                                                        const uint8_t* __restrict__ d2, int n2,
                                                        int32_t* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 16;
  if (j >= n2) return;
  const Desc b = load_desc(d2, j);
  for (int i = i0; i < i0 + 16 && i < n1; i++) {
    const Desc a = load_desc(d1, i);  // wave-uniform address: one broadcast load
    out[(size_t)i * n2 + j] = hdist(a, b);
  }
}
void launch_hamming_matrix(hipStream_t s, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int32_t* out) {
  if (n1 <= 0 || n2 <= 0) return;
  hipLaunchKernelGGL(k_hamming_matrix, dim3((n2 + 255) / 256, (n1 + 15) / 16), dim3(256), 0, s, d1, n1, d2, n2, out);
}

// ---------------------------------------------------------------------------------------
// SearchByBoW (src/ORBmatcher.cc:185-325 and :610-743): one wavefront per shared vocabulary
// node.  Queries inside a node are sequential (a frame feature claimed by an earlier query is
// skipped by later ones, :242-244/:664,691); different nodes never interact.
// ---------------------------------------------------------------------------------------
// one shared vocabulary node: the wave scans node-2 for every node-1 feature, in order
__device__ __forceinline__ void bow_node(const BowArgs& a, const NodePair np, int lane, uint8_t* claimed) {
  for (int i = lane; i < np.cnt2; i += 64) claimed[i] = 0;
  __syncthreads();
  for (int q = 0; q < np.cnt1; q++) {
    const uint32_t idx1 = a.indices1[np.off1 + q];
    if (a.hasMp1 && !a.hasMp1[idx1]) continue;  // wave-uniform (NULL: every feature has a MapPoint)
    const Desc d1 = load_desc(a.desc1, idx1);
    uint32_t key1 = (256u << 16) | 0xffffu;  // (bestDist1, position) -- first minimum wins
    uint32_t best2 = 256u;
    for (int p = lane; p < np.cnt2; p += 64) {
      const uint32_t idx2 = a.indices2[np.off2 + p];
      if (claimed[p]) continue;
      if (a.hasMp2 && !a.hasMp2[idx2]) continue;  // KF-KF form, :664
      const uint32_t dist = (uint32_t)hdist(d1, load_desc(a.desc2, idx2));
      const uint32_t key = (dist << 16) | (uint32_t)p;
      if (dist < (key1 >> 16)) { best2 = key1 >> 16; key1 = key; }
      else if (dist < best2) best2 = dist;
    }
    // two smallest of the multiset, first position of the minimum
    wave_top2_dpp(key1, best2);
    const uint32_t bestDist1 = key1 >> 16, pos = key1 & 0xffffu;
    const bool pass1 = a.strictLow ? (bestDist1 < 50u) : (bestDist1 <= 50u);  // :263 vs :686
    if (pass1 && (float)bestDist1 < __fmul_rn(a.nnratio, (float)best2)) {
      if (lane == 0) {
        claimed[pos] = 1;
        const uint32_t idx2 = a.indices2[np.off2 + pos];
        const int bin = rot_bin(a.angle1[(size_t)idx1 * a.angleStride], a.angle2[(size_t)idx2 * a.angleStride]);
        if (a.strictLow) {  // KF-KF: vpMatches12[idx1] = MapPoint of idx2; histogram holds idx1
          a.match[idx1] = (int32_t)idx2;
          a.bin[idx1] = (int8_t)bin;
        } else {            // KF-Frame: vpMapPointMatches[idxF] = MapPoint of idxKF; histogram holds idxF
          a.match[idx2] = (int32_t)idx1;
          a.bin[idx2] = (int8_t)bin;
        }
      }
      __syncthreads();  // single-wave block: orders the LDS claim before the next query
    }
  }
}

// The common case -- a node holds a few dozen features on either side (k = 10, level 2: ~1 % of a frame each; a skewed
// vocabulary: a few hundred) -- without a memory access inside the sequential query loop: lane p keeps candidates p,
// p + 64, ... (NC per lane: up to 64 * NC candidates) with their descriptor, index, angle and "claimed" flag in
// registers, the queries pass through the wave 64 at a time (lane q keeps query q's), a query is handed to the wave with
// v_readlane (q is the loop counter).  bow_node's chain per query was index -> descriptor -> candidate descriptors
// (three dependent round trips, ~40 us per node -- and the LARGEST node of a pair decides how long the launch takes);
// here a node is two round trips plus ~100 ALU cycles per query.  Same scan order, same claims: the key carries the
// candidate's position in the node, so the minimum over lanes and slots is the reference's first minimum.
// the wave's two smallest distances: DPP (kernels.h) in the single-call kernels, whose launch lasts as long as the queries of
// its largest node walk this reduction one after the other (129 queries: 74 -> 54 us); LDS permutes in the batch kernel of the
// device-resident pipelines, where thousands of node waves overlap and the VALU port is what they share with the extractor's
// kernels (EuRoC stereo, same box: 99.8 k with the permutes, 98.8 k with DPP)
template <bool kDpp>
__device__ __forceinline__ void bow_top2(uint32_t& key1, uint32_t& best2) {
  if constexpr (kDpp) { wave_top2_dpp(key1, best2); return; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t ok1k = (uint32_t)__shfl_xor((int)key1, o, 64);
    const uint32_t ob2 = (uint32_t)__shfl_xor((int)best2, o, 64);
    const uint32_t lo = ok1k < key1 ? ok1k : key1, hi = ok1k < key1 ? key1 : ok1k;
    const uint32_t m2 = ob2 < best2 ? ob2 : best2;
    key1 = lo;
    best2 = (hi >> 16) < m2 ? (hi >> 16) : m2;
  }
  key1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key1);
  best2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)best2);
}

template <int NC, bool kDpp>
__device__ __forceinline__ void bow_node_regs(const BowArgs& a, const NodePair np, int lane) {
  uint32_t idx2[NC];
  bool free2[NC];
  Desc D2[NC];
  float ang2[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const int p = lane + 64 * c;
    idx2[c] = 0; free2[c] = false; ang2[c] = 0.f; D2[c] = Desc{};
    if (p < np.cnt2) { idx2[c] = a.indices2[np.off2 + p]; free2[c] = !(a.hasMp2 && !a.hasMp2[idx2[c]]); }
  }
#pragma unroll
  for (int c = 0; c < NC; c++)
    if (lane + 64 * c < np.cnt2) { D2[c] = load_desc(a.desc2, idx2[c]); ang2[c] = a.angle2[(size_t)idx2[c] * a.angleStride]; }
  for (int q0 = 0; q0 < np.cnt1; q0 += 64) {  // (wave-uniform trip count)
    uint32_t idx1 = 0;
    bool ok1 = false;
    Desc D1 = {};
    float ang1 = 0.f;
    if (q0 + lane < np.cnt1) {
      idx1 = a.indices1[np.off1 + q0 + lane];
      ok1 = !(a.hasMp1 && !a.hasMp1[idx1]);
      D1 = load_desc(a.desc1, idx1);
      ang1 = a.angle1[(size_t)idx1 * a.angleStride];
    }
    const unsigned long long okMask = __ballot(ok1);
    const int nq = np.cnt1 - q0 < 64 ? np.cnt1 - q0 : 64;
    for (int q = 0; q < nq; q++) {
      if (!((okMask >> q) & 1ull)) continue;  // wave-uniform
      Desc d1;
#pragma unroll
      for (int i = 0; i < 8; i++) d1.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)D1.w[i], q);
      uint32_t key1 = (256u << 16) | 0xffffu;  // (bestDist1, position) -- first minimum wins
      uint32_t best2 = 256u;
#pragma unroll
      for (int c = 0; c < NC; c++)
        if (free2[c]) {  // ascending position inside the lane: strict < keeps the earlier of two equal distances
          const uint32_t dist = (uint32_t)hdist(d1, D2[c]);
          if (dist < (key1 >> 16)) { best2 = key1 >> 16; key1 = (dist << 16) | (uint32_t)(lane + 64 * c); }
          else if (dist < best2) best2 = dist;
        }
      bow_top2<kDpp>(key1, best2);
      const uint32_t bestDist1 = key1 >> 16, pos = key1 & 0xffffu;
      const bool pass1 = a.strictLow ? (bestDist1 < 50u) : (bestDist1 <= 50u);  // :263 vs :686
      if (pass1 && (float)bestDist1 < __fmul_rn(a.nnratio, (float)best2)) {  // wave-uniform; pos < cnt2 here
        const int pl = (int)(pos & 63u), pc = (int)(pos >> 6);                // the winner's lane and slot (wave-uniform)
        uint32_t i2 = 0;
        float a2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++)
          if (c == pc) {  // scalar branch
            if (lane == pl) free2[c] = false;  // :267 / :691
            i2 = (uint32_t)__builtin_amdgcn_readlane((int)idx2[c], pl);
            a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ang2[c]), pl));
          }
        const uint32_t i1 = (uint32_t)__builtin_amdgcn_readlane((int)idx1, q);
        const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ang1), q));
        if (lane == 0) {
          const int bin = rot_bin(a1, a2);
          if (a.strictLow) { a.match[i1] = (int32_t)i2; a.bin[i1] = (int8_t)bin; }
          else { a.match[i2] = (int32_t)i1; a.bin[i2] = (int8_t)bin; }
        }
      }
    }
  }
}

// NCMAX = candidate slots per lane the launch was built for (the host knows the largest node of the call): nodes beyond
// 64 * NCMAX candidates take the LDS-claim path
template <int NCMAX, bool kDpp = true>
__device__ __forceinline__ void bow_node_any(const BowArgs& a, const NodePair np, int lane, uint8_t* claimed) {
  if (np.cnt2 <= 64) bow_node_regs<1, kDpp>(a, np, lane);
  else if (NCMAX >= 2 && np.cnt2 <= 128) bow_node_regs<2, kDpp>(a, np, lane);
  else if (NCMAX >= 4 && np.cnt2 <= 256) bow_node_regs<4, kDpp>(a, np, lane);
  else bow_node(a, np, lane, claimed);
}

template <int NCMAX>
__global__ __launch_bounds__(64) void k_search_by_bow(BowArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t claimed[];  // per position in node 2
  bow_node_any<NCMAX>(a, a.pairs[blockIdx.x], threadIdx.x, claimed);
}

// K (frame, frame) problems in ONE launch (orbfe_search_by_bow_multi: one frame against K candidate key frames): the
// node pairs of all problems are numbered through, pairStart[k] = first pair of problem k, args[k] its operands
template <int NCMAX>
__global__ __launch_bounds__(64) void k_search_by_bow_multi(const BowArgs* __restrict__ args, const int32_t* __restrict__ pairStart, int K) {
  extern __shared__ __attribute__((aligned(16))) uint8_t claimed[];
  const int g = blockIdx.x;
  int lo = 0, hi = K;  // last k with pairStart[k] <= g (block-uniform: scalar loads)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pairStart[mid] <= g) lo = mid; else hi = mid;
  }
  const BowArgs a = args[lo];
  bow_node_any<NCMAX>(a, a.pairs[g - pairStart[lo]], threadIdx.x, claimed);
}

// Device-resident batch: pair p = SearchByBoW(KF = frame p, F = frame p+1) with the FeatureVectors
// built by k_vocab_featvec.  Block (i, p) takes the i-th node of frame p and looks the same node
// id up in frame p+1 (the merge-walk of :211-300 as a binary search over the ascending ids).
__global__ __launch_bounds__(64) void k_search_by_bow_batch(BowBatch b) {
  extern __shared__ __attribute__((aligned(16))) uint8_t claimed[];
  const int lane = threadIdx.x, i = blockIdx.x, p = blockIdx.y;
  const int n1 = b.fvCount[p], n2 = b.fvCount[p + 1];
  if (i >= n1) return;
  const size_t c = (size_t)b.capacity;
  const uint32_t* nodes1 = b.fvNodes + (size_t)p * c;
  const uint32_t* nodes2 = b.fvNodes + (size_t)(p + 1) * c;
  const int32_t* off1 = b.fvOffsets + (size_t)p * (c + 1);
  const int32_t* off2 = b.fvOffsets + (size_t)(p + 1) * (c + 1);
  const uint32_t id = nodes1[i];
  int lo = 0, hi = n2;  // lower_bound
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (nodes2[mid] < id) lo = mid + 1; else hi = mid;
  }
  if (lo >= n2 || nodes2[lo] != id) return;
  BowArgs a = {};
  const size_t st = b.frameStep > 1 ? (size_t)b.frameStep : 1;
  a.desc1 = b.desc + (size_t)p * st * c * 32;
  a.desc2 = b.desc + (size_t)(p + 1) * st * c * 32;
  a.indices1 = b.fvIndices + (size_t)p * c;
  a.indices2 = b.fvIndices + (size_t)(p + 1) * c;
  a.angle1 = b.kp + (size_t)p * st * c * 7 + 3;        // cv::KeyPoint::angle
  a.angle2 = b.kp + (size_t)(p + 1) * st * c * 7 + 3;
  a.angleStride = 7;
  a.nnratio = b.nnratio;
  a.strictLow = 0;
  a.match = b.match + (size_t)p * c;
  a.bin = b.bin + (size_t)p * c;
  const NodePair np = {off1[i], off1[i + 1] - off1[i], off2[lo], off2[lo + 1] - off2[lo]};
  bow_node_any<1, false>(a, np, lane, claimed);
}

// ---------------------------------------------------------------------------------------
// SearchForTriangulation (src/ORBmatcher.cc:754-928): queries are independent (vbMatched2 is
// never written, :774,827) -> one wavefront per KF1 feature.  A candidate replaces the best
// when dist <= bestDist, so among equal distances the LAST scanned one wins (:840).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void search_triangulation_one(const TriArgs& a, int q, int lane) {
  if (q >= a.nQueries) return;
  const TriQuery tq = a.queries[q];
  const uint32_t idx1 = tq.idx1;
  const Desc d1 = load_desc(a.desc1, idx1);
  const float x1 = a.x1[idx1], y1 = a.y1[idx1];
  const bool st1 = a.stereo1[idx1] != 0;
  // epipolar line of kp1 in image 2, CheckDistEpipolarLine :158-160
  const float* F = a.F12;
  const float la = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[0]), __fmul_rn(y1, F[3])), F[6]);
  const float lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[1]), __fmul_rn(y1, F[4])), F[7]);
  const float lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, F[2]), __fmul_rn(y1, F[5])), F[8]);
  const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
  uint32_t best = 0xffffffffu;  // (dist << 16) | (0xffff - position): min dist, last position
  for (int p = lane; p < tq.cnt2; p += 64) {
    const uint32_t idx2 = a.indices2[tq.off2 + p];
    if (a.hasMp2[idx2]) continue;
    const bool st2 = a.stereo2[idx2] != 0;
    if (a.onlyStereo && !st2) continue;
    const uint32_t dist = (uint32_t)hdist(d1, load_desc(a.desc2, idx2));
    if (dist > 50u) continue;
    const float x2 = a.x2[idx2], y2 = a.y2[idx2];
    const int oc = a.octave2[idx2];
    if (!st1 && !st2) {  // :845-854
      const float dex = __fsub_rn(a.ex, x2), dey = __fsub_rn(a.ey, y2);
      if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.0f, a.scaleFactors2[oc])) continue;
    }
    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, x2), __fmul_rn(lb, y2)), lc);
    if (den == 0) continue;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    if (!((double)dsqr < __dmul_rn(3.84, (double)a.levelSigma2_2[oc]))) continue;
    const uint32_t key = (dist << 16) | (0xffffu - (uint32_t)p);
    best = key < best ? key : best;
  }
  best = wave_min_u32(best);
  if (lane == 0) {
    if (best != 0xffffffffu) {
      const uint32_t pos = 0xffffu - (best & 0xffffu);
      const uint32_t idx2 = a.indices2[tq.off2 + pos];
      a.match[idx1] = (int32_t)idx2;
      a.bin[idx1] = (int8_t)rot_bin(a.angle1[idx1], a.angle2[idx2]);
    }
  }
}

__global__ __launch_bounds__(256) void k_search_triangulation(TriArgs a) {
  search_triangulation_one(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}
// K (key frame, neighbour) problems in ONE launch (orbfe_search_for_triangulation_multi, LocalMapping::CreateNewMapPoints:
// 20 neighbours): blockStart[k] = first workgroup of problem k (4 queries per workgroup), args[k] its operands.  The K
// launches ran one after the other on the call's stream, ~10 us apiece for a few hundred wavefronts each.
__global__ __launch_bounds__(256) void k_search_triangulation_multi(const TriArgs* __restrict__ args, const int32_t* __restrict__ blockStart,
                                                                    int K) {
  const int g = blockIdx.x;
  int lo = 0, hi = K;  // last k with blockStart[k] <= g (block-uniform: scalar loads)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blockStart[mid] <= g) lo = mid; else hi = mid;
  }
  const TriArgs a = args[lo];
  search_triangulation_one(a, (g - blockStart[lo]) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Rotation-consistency pruning shared by the three searches (:303-322, ComputeThreeMaxima
// :1777-1821): one workgroup histograms the accepted matches, keeps the three dominant bins.
__global__ __launch_bounds__(256) void k_rot_prune(int32_t* __restrict__ match, const int8_t* __restrict__ bin,
                                                   int n, int checkOri, int32_t* __restrict__ nMatches,
                                                   int batchStride) {
  match += (size_t)blockIdx.x * batchStride;  // one workgroup per match array of a batch
  bin += (size_t)blockIdx.x * batchStride;
  nMatches += blockIdx.x;
  __shared__ int hist[30];
  __shared__ int keep[3];
  __shared__ int total;
  const int tid = threadIdx.x;
  if (tid < 30) hist[tid] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < n; i += 256)
    if (match[i] >= 0) { mine++; atomicAdd(&hist[bin[i]], 1); }
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (!checkOri) { if (tid == 0) *nMatches = total; return; }
  if (tid == 0) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < 30; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
      else if (s > max3) { max3 = s; i3 = i; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
    else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
    keep[0] = i1; keep[1] = i2; keep[2] = i3;
    int kept = 0;
    for (int i = 0; i < 30; i++)
      if (i == i1 || i == i2 || i == i3) kept += hist[i];
    *nMatches = kept;
  }
  __syncthreads();
  const int k0 = keep[0], k1 = keep[1], k2 = keep[2];
  for (int i = tid; i < n; i += 256)
    if (match[i] >= 0) {
      const int b = bin[i];
      if (b != k0 && b != k1 && b != k2) match[i] = -1;
    }
}

void launch_search_by_bow(hipStream_t s, const BowArgs& a, int nPairs, int maxCnt2) {
  if (nPairs <= 0) return;
  const size_t lds = (size_t)((maxCnt2 + 15) & ~15);
  if (maxCnt2 <= 64) hipLaunchKernelGGL((k_search_by_bow<1>), dim3(nPairs), dim3(64), lds, s, a);
  else if (maxCnt2 <= 128) hipLaunchKernelGGL((k_search_by_bow<2>), dim3(nPairs), dim3(64), lds, s, a);
  else hipLaunchKernelGGL((k_search_by_bow<4>), dim3(nPairs), dim3(64), lds, s, a);
}
void launch_search_by_bow_multi(hipStream_t s, const BowArgs* d_args, const int32_t* d_pairStart, int K, int nPairsTotal, int maxCnt2) {
  if (nPairsTotal <= 0 || K <= 0) return;
  const size_t lds = (size_t)((maxCnt2 + 15) & ~15);
  if (maxCnt2 <= 64) hipLaunchKernelGGL((k_search_by_bow_multi<1>), dim3(nPairsTotal), dim3(64), lds, s, d_args, d_pairStart, K);
  else if (maxCnt2 <= 128) hipLaunchKernelGGL((k_search_by_bow_multi<2>), dim3(nPairsTotal), dim3(64), lds, s, d_args, d_pairStart, K);
  else hipLaunchKernelGGL((k_search_by_bow_multi<4>), dim3(nPairsTotal), dim3(64), lds, s, d_args, d_pairStart, K);
}
void launch_search_triangulation(hipStream_t s, const TriArgs& a) {
  if (a.nQueries <= 0) return;
  hipLaunchKernelGGL(k_search_triangulation, dim3((a.nQueries + 3) / 4), dim3(256), 0, s, a);
}
void launch_search_triangulation_multi(hipStream_t s, const TriArgs* d_args, const int32_t* d_blockStart, int K, int totalBlocks) {
  if (K <= 0 || totalBlocks <= 0) return;
  hipLaunchKernelGGL(k_search_triangulation_multi, dim3(totalBlocks), dim3(256), 0, s, d_args, d_blockStart, K);
}
void launch_rot_prune(hipStream_t s, int32_t* match, const int8_t* bin, int n, int checkOri, int32_t* nMatches) {
  hipLaunchKernelGGL(k_rot_prune, dim3(1), dim3(256), 0, s, match, bin, n, checkOri, nMatches, 0);
}
void launch_rot_prune_batch(hipStream_t s, int32_t* match, const int8_t* bin, int n, int nArrays, int checkOri, int32_t* nMatches) {
  if (nArrays <= 0) return;  // array k at match + k*n, count at nMatches[k]
  hipLaunchKernelGGL(k_rot_prune, dim3(nArrays), dim3(256), 0, s, match, bin, n, checkOri, nMatches, n);
}
void launch_search_by_bow_batch(hipStream_t s, const BowBatch& b, int nPairs, int checkOri, int32_t* d_nMatches, int maxNodes) {
  if (nPairs <= 0 || b.capacity <= 0) return;
  const size_t lds = (size_t)((b.capacity + 15) & ~15);
  // a FeatureVector has at most min(capacity, nodes of the vocabulary at that level) entries: no workgroups beyond that
  const int gx = (maxNodes > 0 && maxNodes < b.capacity) ? maxNodes : b.capacity;
  hipLaunchKernelGGL(k_search_by_bow_batch, dim3(gx, nPairs), dim3(64), lds, s, b);
  hipLaunchKernelGGL(k_rot_prune, dim3(nPairs), dim3(256), 0, s, b.match, b.bin, b.capacity, checkOri, d_nMatches,
                     b.capacity);
}

// ---------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:512-686)
// ---------------------------------------------------------------------------------------
struct StereoPair {  // per-pair operands (kept apart from the kernel argument so that stays read-only)
  const float* kpL; const uint8_t* descL; int N;
  const float* kpR; const uint8_t* descR; int Nr;
  int frameL, frameR;
  float* uRight; float* depth; int32_t* sad;
  const int32_t* rowStart; const int32_t* sortedIdx; const float4* sortedRec;
};

// ---- 16 lanes per left keypoint (one DPP row), four keypoints per wavefront ----
// Round 1 gave a whole wavefront to a left keypoint whose row band holds ~80 right keypoints of which ~9 reach the
// Hamming distance, and read the 11 x (11 x 11) SAD windows with byte loads: mostly idle lanes and waits (1.35 ms per
// 512 KITTI pairs; this form 0.85 ms).  A keypoint owns the 16 lanes of a DPP row: the candidate scan strides 16, the SAD gives patch ROW r
// to lane r (one 16-byte request for the left row, 24 bytes for the 21-pixel right strip all 11 shifts share), the
// 11 x 11 absolute differences of a shift are v_sad_u16 on packed pairs -- |(l - cL) - (r - cR)| = |(l + cR) - (r + cL)|
// with both sides in [0, 510] -- and the row sums meet in a rotate-add over the DPP row.  Same results bit for bit
// (integer sums; the fp32 tail is the same code), a quarter of the wavefronts and ~1/3 of the VALU work per keypoint.
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:1
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:2
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:4
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:8
  return v;
}
__device__ __forceinline__ int row16_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
  return v;
}
// v_perm_b32 selector: bytes (o, o+1) of the 8 bytes {hi, lo} as a packed u16 pair
constexpr uint32_t selpair(int o) { return (uint32_t)o | 0x0c00u | ((uint32_t)(o + 1) << 16) | 0x0c000000u; }

// `lv` = per-level views of both pyramids in LDS (a vector level index into the kernel argument would go through scratch)
struct StereoLds { LevelView L[kMaxLevels], R[kMaxLevels]; float scale[2 * kMaxLevels]; };  // scale: mvScaleFactor | mvInvScaleFactor

// kRec: the right keypoints' sorted 16-byte records exist (pp.sortedRec; every batch call and every host call with row
// buckets) -- a template parameter, so the candidate scan does not test a loop-invariant pointer on every trip
template <bool kRec>
__device__ __forceinline__ void stereo_row16(const StereoArgs& a, const StereoLds& lv, const StereoPair& pp, int iL, bool alive,
                                             int lane) {
  const int sub = lane & 15, rowBase = lane & 48;
  float uL = 0.f, vL = 0.f;
  int levelL = 0;
  bool record = true;  // (`alive` = the slot holds a left keypoint and gets an output; `record` = its fields are usable)
  if (alive) {
    const float* kl = pp.kpL + (size_t)iL * 7;
    uL = kl[0]; vL = kl[1];
    levelL = reinterpret_cast<const int32_t*>(kl)[5];
    // a record no extractor writes (device operands are the caller's: zeroed / stale / foreign buffers): an octave that is
    // not a pyramid level or a non-finite position is "no stereo" (-1), never an index.  The reference would index
    // mvInvScaleFactor / mvImagePyramid out of range there (src/Frame.cc:600-609)
    if ((unsigned)levelL >= (unsigned)a.pyrL.nlevels || !(fabsf(uL) < 3.0e38f) || !(fabsf(vL) < 3.0e38f)) { record = false; levelL = 0; uL = vL = 0.f; }
  }
  const int row = (int)vL;
  const float minU = __fsub_rn(uL, a.maxD), maxU = uL;  // minD = 0
  bool live = alive && record && !(maxU < 0) && row >= 0 && row < a.rows;  // vRowIndices[vL] (:560) has nRows entries
  int pBeg = 0, pEnd = live ? pp.Nr : 0;
  if (pp.rowStart && live) {
    // rows that can hold a candidate: a right keypoint of octave o covers image rows floor(y - r) .. ceil(y + r) with
    // r = 2 * scale[o], and only octaves <= levelL + 1 are accepted (:579) -- so the bucket rows within
    // ceil(2 * scale[levelL + 1]) + 2 of `row` are enough (a.bandR is that bound for the top octave; most keypoints sit
    // on the low octaves, whose band is 4 rows instead of 10: half the records to scan)
    int band = a.bandR;
    {
      const int oMax = levelL + 1 < a.pyrL.nlevels ? levelL + 1 : a.pyrL.nlevels - 1;
      const int bl = (int)ceilf(__fmul_rn(2.0f, lv.scale[oMax < 0 ? 0 : oMax])) + 2;
      band = bl < band ? bl : band;
    }
    int lo = row - band, hi = row + band;
    lo = lo < 0 ? 0 : lo;
    hi = hi > a.rows - 1 ? a.rows - 1 : hi;
    if (lo > hi) { live = false; pEnd = 0; }
    else { pBeg = pp.rowStart[lo]; pEnd = pp.rowStart[hi + 1]; }
  }
  Desc dL = {};
  if (live) dL = load_desc(pp.descL, iL);
  // ---- candidate scan (:568-594), 16 lanes striding the band's sorted records ----
  uint32_t best = 0xffffffffu;  // dist << 20 | iR
  float bestU = 0.f;
  for (int p = pBeg + sub; __any(p < pEnd); p += 16) {
    if (p < pEnd) {
      int iR, octR;
      float yR, uR;
      if (kRec) {
        const float4 rc = pp.sortedRec[p];
        uR = rc.x; yR = rc.y; octR = __float_as_int(rc.z); iR = __float_as_int(rc.w);
      } else {
        iR = pp.rowStart ? pp.sortedIdx[p] : p;
        const float* kr = pp.kpR + (size_t)iR * 7;
        octR = reinterpret_cast<const int32_t*>(kr)[5];
        yR = kr[1];
        uR = kr[0];
      }
      if ((unsigned)octR >= (unsigned)a.pyrL.nlevels || !(fabsf(yR) < 3.0e38f)) continue;  // not a record of this pyramid: never a candidate
      const float r = __fmul_rn(2.0f, lv.scale[octR]);
      const int maxr = (int)ceilf(__fadd_rn(yR, r));
      const int minr = (int)floorf(__fsub_rn(yR, r));
      const bool cand = row >= minr && row <= maxr && octR >= levelL - 1 && octR <= levelL + 1 && uR >= minU && uR <= maxU;
      if (cand) {
        const uint32_t dist = (uint32_t)hdist(dL, load_desc(pp.descR, iR));
        const uint32_t key = (dist << 20) | (uint32_t)iR;
        if (dist < 100u && key < best) { best = key; bestU = uR; }  // bestDist starts at TH_HIGH, strict <
      }
    }
  }
  const uint32_t mine = best;
  best = row16_min_u32(best);
  live = live && best != 0xffffffffu && (int)(best >> 20) < 75;  // thOrbDist = (TH_HIGH+TH_LOW)/2, :517,598
  // the winning lane of the row (keys are unique: they carry iR) hands over its uR
  const unsigned long long win = __ballot(mine == best);
  const uint32_t winRow = (uint32_t)(win >> rowBase) & 0xffffu;
  const int src = rowBase + (winRow ? __builtin_ctz(winRow) : 0);
  const float uR0 = __shfl(bestU, src, 64);
  float oU = -1.0f, oD = -1.0f;
  int oS = -1;
  // ---- SAD refinement on the left keypoint's pyramid level (:600-638) ----
  // (wave-uniform: skipped when none of the wave's four left keypoints found a candidate below thOrbDist -- the block is
  // 403 of the kernel's 920 VALU instructions per wave)
  if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
  const float sf = lv.scale[kMaxLevels + levelL];
  const float scaleduL = roundf(__fmul_rn(uL, sf));
  const float scaledvL = roundf(__fmul_rn(vL, sf));
  const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
  const LevelView L = lv.L[levelL], R = lv.R[levelL];
  const float iniu = scaleduR0 + 5 - 5;
  const float endu = scaleduR0 + 5 + 5 + 1;
  if (iniu < 0 || endu >= (float)R.w) live = false;
  const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
  // the reference takes the patches with cv::Mat::rowRange / colRange (:609-610, :626), which assert
  // 0 <= start <= end <= size: a window that leaves the level is an exception there and "no stereo" here.  Every byte
  // the SAD reads below lies inside rows cy-5 .. cy+5, columns cxL-5 .. cxL+5 (left) / cxR0-10 .. cxR0+10 (right).
  if (cy < 5 || cy + 5 >= L.h || cy + 5 >= R.h || cxL < 5 || cxL + 5 >= L.w || cxR0 < 10) live = false;
  // lane r < 11 of the row holds patch row cy-5+r: left bytes cxL-5 .. cxL+5 in ql[0..2], right bytes cxR0-10 .. cxR0+10 in
  // qr[0..5].  Keypoints of the extractor lie >= 19 px inside their level, so the 16- and 24-byte requests stay inside the
  // image row; any other caller's keypoints take byte loads of exactly the pixels the reference reads
  uint32_t ql[4] = {0u, 0u, 0u, 0u}, qr[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  if (live && sub < 11) {
    const uint8_t* pl = L.base + (size_t)pp.frameL * L.frameStride + (ptrdiff_t)(cy + sub - 5) * L.pitch + (cxL - 5);
    const uint8_t* pr = R.base + (size_t)pp.frameR * R.frameStride + (ptrdiff_t)(cy + sub - 5) * R.pitch + (cxR0 - 10);
    struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
    struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };
    const bool wide = cxL >= 5 && cxL + 10 < L.w && cxR0 >= 10 && cxR0 + 13 < R.w && cy >= 5 && cy + 5 < L.h && cy + 5 < R.h;
    if (wide) {
      const U4u a4 = *reinterpret_cast<const U4u*>(pl);
      const U4u b4 = *reinterpret_cast<const U4u*>(pr);
      const U2u c2 = *reinterpret_cast<const U2u*>(pr + 16);
      ql[0] = a4.x; ql[1] = a4.y; ql[2] = a4.z;
      qr[0] = b4.x; qr[1] = b4.y; qr[2] = b4.z; qr[3] = b4.w; qr[4] = c2.x; qr[5] = c2.y;
    } else {
#pragma unroll
      for (int k = 0; k < 11; k++) ql[k >> 2] |= (uint32_t)pl[k] << (8 * (k & 3));
#pragma unroll
      for (int k = 0; k < 21; k++) qr[k >> 2] |= (uint32_t)pr[k] << (8 * (k & 3));
    }
  }
  // centres: cL = left (cy, cxL) = row 5, byte 5; cR(inc) = right (cy, cxR0+inc) = row 5, bytes 5 .. 15
  const uint32_t cL = ((uint32_t)__shfl((int)ql[1], rowBase + 5, 64) >> 8) & 0xffu;
  const uint32_t c1 = (uint32_t)__shfl((int)qr[1], rowBase + 5, 64), c2 = (uint32_t)__shfl((int)qr[2], rowBase + 5, 64),
                 c3 = (uint32_t)__shfl((int)qr[3], rowBase + 5, 64);
  const uint32_t cLL = cL | (cL << 16);
  uint32_t A[5], B[19], C[11];
#pragma unroll
  for (int m = 0; m < 5; m++) A[m] = __builtin_amdgcn_perm(ql[(m >> 1) + 1], ql[m >> 1], selpair((2 * m) & 3));
  const uint32_t l10 = (ql[2] >> 16) & 0xffu;
#pragma unroll
  for (int t = 0; t < 19; t++) B[t] = __builtin_amdgcn_perm(qr[(t >> 2) + 1], qr[t >> 2], selpair(t & 3)) + cLL;
#pragma unroll
  for (int t = 10; t < 21; t++) C[t - 10] = ((qr[t >> 2] >> (8 * (t & 3))) & 0xffu) + cL;
  int dists[11];
#pragma unroll
  for (int s = 0; s < 11; s++) {  // s = inc + 5
    const int cb = 5 + s;          // byte of the row-5 strip that is this shift's centre
    const uint32_t cw = cb < 8 ? c1 : (cb < 12 ? c2 : c3);
    const uint32_t cR = (cw >> (8 * (cb & 3))) & 0xffu;
    const uint32_t cRR = cR | (cR << 16);
    uint32_t acc = 0;
#pragma unroll
    for (int m = 0; m < 5; m++) acc = __builtin_amdgcn_sad_u16(A[m] + cRR, B[2 * m + s], acc);
    acc = __builtin_amdgcn_sad_u16(l10 + cR, C[s], acc);
    dists[s] = row16_sum_i32(sub < 11 ? (int)acc : 0);
  }
  int bestSad = 0x7fffffff, bestinc = 0;
#pragma unroll
  for (int i = 0; i < 11; i++)
    if (dists[i] < bestSad) { bestSad = dists[i]; bestinc = i - 5; }
  if (bestinc == -5 || bestinc == 5) live = false;
  float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
  for (int i = 1; i < 10; i++)
    if (i == bestinc + 5) { d1 = (float)dists[i - 1]; d2 = (float)dists[i]; d3 = (float)dists[i + 1]; }
  // parabola sub-pixel (:644-651)
  const float deltaR = __fdiv_rn(__fsub_rn(d1, d3),
                                 __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
  if (deltaR < -1 || deltaR > 1) live = false;
  float bestuR = __fmul_rn(lv.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
  float disparity = __fsub_rn(uL, bestuR);
  if (live && disparity >= 0 && disparity < a.maxD) {
    if (disparity <= 0) {
      disparity = 0.01f;
      bestuR = (float)__dsub_rn((double)uL, 0.01);
    }
    oD = __fdiv_rn(a.mbf, disparity);
    oU = bestuR;
    oS = bestSad;
  }
  }  // any keypoint of the wave alive
  if (alive && sub == 0) { pp.uRight[iL] = oU; pp.depth[iL] = oD; pp.sad[iL] = oS; }
}

// Counting sort of the right keypoints by image row floor(y): rowStart[r] .. rowStart[r+1] index
// sortedIdx (the flat equivalent of vRowIndices, src/Frame.cc:519-539, before band expansion).
// One workgroup per stereo pair; rows <= 8191.
__global__ __launch_bounds__(256) void k_stereo_bucket(const float* __restrict__ kpBase, const int32_t* __restrict__ nArr,
                                                       int nFixed, int capacity, int rows,
                                                       int32_t* __restrict__ rowStartBase,
                                                       int32_t* __restrict__ sortedBase, float4* __restrict__ recBase) {
  extern __shared__ int cnt[];  // rows + 1
  __shared__ int waveTot[4];
  ORBFE_LATENCY_KERNEL_PRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = blockIdx.x;
  const float* kpR = nArr ? kpBase + (size_t)(2 * p + 1) * capacity * 7 : kpBase;
  int Nr = nArr ? nArr[2 * p + 1] : nFixed;
  if (Nr > capacity) Nr = capacity;
  int32_t* rowStart = rowStartBase + (size_t)p * (rows + 1);
  int32_t* sorted = sortedBase + (size_t)p * capacity;
  float4* rec = recBase ? recBase + (size_t)p * capacity : nullptr;
  for (int i = tid; i <= rows; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < Nr; i += 256) {
    int b = (int)floorf(kpR[(size_t)i * 7 + 1]);
    b = b < 0 ? 0 : (b > rows - 1 ? rows - 1 : b);
    atomicAdd(&cnt[b], 1);
  }
  __syncthreads();
  int run = 0;
  for (int base = 0; base <= rows; base += 256) {  // exclusive scan over the rows
    const int i = base + tid;
    const int v = i <= rows ? cnt[i] : 0;
    int x = v;  // (LDS permutes on purpose: part of the batch pipeline, see k_octree.hip octree_wave_incl_scan)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) waveTot[wave] = x;
    __syncthreads();
    int b = run;
    for (int w = 0; w < wave; w++) b += waveTot[w];
    run += waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
    if (i <= rows) { cnt[i] = b + x - v; rowStart[i] = b + x - v; }
    __syncthreads();
  }
  for (int i = tid; i < Nr; i += 256) {
    int b = (int)floorf(kpR[(size_t)i * 7 + 1]);
    b = b < 0 ? 0 : (b > rows - 1 ? rows - 1 : b);
    const int pos = atomicAdd(&cnt[b], 1);
    sorted[pos] = i;
    if (rec) rec[pos] = make_float4(kpR[(size_t)i * 7], kpR[(size_t)i * 7 + 1], __int_as_float(reinterpret_cast<const int32_t*>(kpR)[(size_t)i * 7 + 5]),
                                    __int_as_float(i));
  }
}

__device__ __forceinline__ void stereo_stage_views(const StereoArgs& a, StereoLds& lv) {
  const int t = threadIdx.x;
  if (t < kMaxLevels) lv.L[t] = a.pyrL.lv[t];
  else if (t < 2 * kMaxLevels) lv.R[t - kMaxLevels] = a.pyrR.lv[t - kMaxLevels];
  else if (t < 4 * kMaxLevels) lv.scale[t - 2 * kMaxLevels] = a.scaleTab[t - 2 * kMaxLevels];  // (the table holds 2 x 16 floats)
  __syncthreads();
}

// 256 threads = 16 left keypoints (16 lanes each, stereo_row16)
__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a) {
  __shared__ StereoLds lv;
  stereo_stage_views(a, lv);
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 16 + (threadIdx.x >> 4);
  StereoPair pp = {a.kpL, a.descL, a.N, a.kpR, a.descR, a.Nr, a.frameL, a.frameR, a.uRight, a.depth, a.sad,
                   a.rowStart, a.sortedIdx, a.rowStart ? a.sortedRec : nullptr};
  if (pp.sortedRec) stereo_row16<true>(a, lv, pp, iL, iL < a.N, lane);
  else stereo_row16<false>(a, lv, pp, iL, iL < a.N, lane);
}

// Batched, device-resident form: pair p = frames (2p, 2p+1) of one extractor batch; the
// keypoint counts are read from device memory (d_n of orbfe_extract_batch_device).
__global__ __launch_bounds__(256) void k_stereo_match_batch(const StereoArgs a, const StereoBatch b) {
  ORBFE_LATENCY_KERNEL_PRIO();
  __shared__ StereoLds lv;
  stereo_stage_views(a, lv);
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int p = blockIdx.y;
  int N = b.n[2 * p], Nr = b.n[2 * p + 1];
  if (N > b.capacity) N = b.capacity;
  if (Nr > b.capacity) Nr = b.capacity;
  const size_t oL = (size_t)(2 * p) * b.capacity, oR = (size_t)(2 * p + 1) * b.capacity, oO = (size_t)p * b.capacity;
  if (iL < b.capacity && iL >= N && (lane & 15) == 0) {  // slots past the left frame's keypoints: defined "no stereo" outputs
    b.uRight[oO + iL] = -1.0f; b.depth[oO + iL] = -1.0f; b.sad[oO + iL] = -1;
  }
  if (blockIdx.x * 16 >= N) return;  // block-uniform: nothing left to match
  StereoPair pp;
  pp.kpL = b.kp + oL * 7; pp.descL = b.desc + oL * 32; pp.N = N;
  pp.kpR = b.kp + oR * 7; pp.descR = b.desc + oR * 32; pp.Nr = Nr;
  pp.frameL = 2 * p; pp.frameR = 2 * p + 1;
  pp.uRight = b.uRight + oO; pp.depth = b.depth + oO; pp.sad = b.sad + oO;
  pp.rowStart = a.rowStart ? a.rowStart + (size_t)p * (a.rows + 1) : nullptr;
  pp.sortedIdx = a.rowStart ? a.sortedIdx + (size_t)p * b.capacity : nullptr;
  pp.sortedRec = (a.rowStart && a.sortedRec) ? a.sortedRec + (size_t)p * b.capacity : nullptr;
  if (pp.sortedRec) stereo_row16<true>(a, lv, pp, iL, iL < N, lane);  // block-uniform
  else stereo_row16<false>(a, lv, pp, iL, iL < N, lane);
}

// Median cut (:672-685): drop matches whose SAD >= 1.5*1.4*median, median = sorted[size/2].
// SAD <= 121*510 < 65536 -> two-level 8+8 bit radix select in one workgroup (one per pair).
__global__ __launch_bounds__(256) void k_stereo_median_cut(int N, const int32_t* __restrict__ sad,
                                                           float* __restrict__ uRight, float* __restrict__ depth,
                                                           int32_t* __restrict__ nStereo, int pairStride) {
  sad += (size_t)blockIdx.x * pairStride;
  uRight += (size_t)blockIdx.x * pairStride;
  depth += (size_t)blockIdx.x * pairStride;
  nStereo += blockIdx.x;
  __shared__ int hist[256];
  __shared__ int sel[3];  // hi bin, remaining k, total
  __shared__ int waveTot[4];
  ORBFE_LATENCY_KERNEL_PRIO();
  const int tid = threadIdx.x;
  hist[tid] = 0;
  if (tid == 0) sel[2] = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < N; i += 256) {
    const int s = sad[i];
    if (s >= 0) { atomicAdd(&hist[s >> 8], 1); mine++; }
  }
  if (mine) atomicAdd(&sel[2], mine);
  __syncthreads();
  const int total = sel[2];
  if (total == 0) { if (tid == 0) *nStereo = 0; return; }
  // the bin that holds the element of rank k: the 256 threads scan the 256 bins together (thread 0 walking them one LDS
  // round trip at a time took ~13 us of the kernel's 34)
  auto pick = [&](int k) {
    const int lane = tid & 63, wave = tid >> 6;
    const int v = hist[tid];
    int x = v;  // (LDS permutes on purpose: part of the batch pipeline, see k_octree.hip octree_wave_incl_scan)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) waveTot[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += waveTot[w];
    const int incl = base + x;
    if (v > 0 && incl - v <= k && k < incl) { sel[0] = tid; sel[1] = k - (incl - v); }
    __syncthreads();
  };
  pick(total / 2);
  const int hi = sel[0];
  __syncthreads();
  hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < N; i += 256) {
    const int s = sad[i];
    if (s >= 0 && (s >> 8) == hi) atomicAdd(&hist[s & 255], 1);
  }
  __syncthreads();
  pick(sel[1]);
  const float median = (float)((hi << 8) | sel[0]);
  __syncthreads();
  if (tid == 0) sel[1] = 0;
  __syncthreads();
  const float thDist = __fmul_rn(1.5f * 1.4f, median);
  int kept = 0;
  for (int i = tid; i < N; i += 256) {
    const int s = sad[i];
    if (s < 0) continue;
    if ((float)s < thDist) kept++;
    else { uRight[i] = -1.0f; depth[i] = -1.0f; }
  }
  if (kept) atomicAdd(&sel[1], kept);
  __syncthreads();
  if (tid == 0) *nStereo = sel[1];
}

void launch_stereo(hipStream_t s, const StereoArgs& a, int32_t* d_nStereo) {
  if (a.N <= 0) return;
  if (a.rowStart)
    hipLaunchKernelGGL(k_stereo_bucket, dim3(1), dim3(256), (size_t)(a.rows + 1) * sizeof(int), s, a.kpR,
                       (const int32_t*)nullptr, a.Nr, a.Nr, a.rows, const_cast<int32_t*>(a.rowStart),
                       const_cast<int32_t*>(a.sortedIdx), const_cast<float4*>(a.sortedRec));
  hipLaunchKernelGGL(k_stereo_match, dim3((a.N + 15) / 16), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_stereo_median_cut, dim3(1), dim3(256), 0, s, a.N, a.sad, a.uRight, a.depth, d_nStereo, 0);
}

void launch_stereo_batch(hipStream_t s, const StereoArgs& a, const StereoBatch& b, int nPairs, int32_t* d_nStereo) {
  if (nPairs <= 0 || b.capacity <= 0) return;
  if (a.rowStart)
    hipLaunchKernelGGL(k_stereo_bucket, dim3(nPairs), dim3(256), (size_t)(a.rows + 1) * sizeof(int), s, b.kp, b.n, 0,
                       b.capacity, a.rows, const_cast<int32_t*>(a.rowStart), const_cast<int32_t*>(a.sortedIdx),
                       const_cast<float4*>(a.sortedRec));
  hipLaunchKernelGGL(k_stereo_match_batch, dim3((b.capacity + 15) / 16, nPairs), dim3(256), 0, s, a, b);
  hipLaunchKernelGGL(k_stereo_median_cut, dim3(nPairs), dim3(256), 0, s, b.capacity, b.sad, b.uRight, b.depth,
                     d_nStereo, b.capacity);
}

}  // namespace orbfe
