// vocabulary.hip -- DBoW2 ORB vocabulary on the device: text loader (TemplatedVocabulary::
// loadFromTextFile, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424) and transform()
// (:1127-1194 feature loop, :1218-1259 k-ary descent with FORB::distance, FORB.cpp:81-101).
// the step immediately before SearchByBoW (src/Frame.cc:433-440).
//
// Built for the tree the reference loads -- k = 10, L = 6: 1 111 111 nodes, 53 MB of records, far beyond the L2 --
// not for a toy tree in cache (round 2 was one THREAD per descriptor chasing childOff -> childIdx -> descriptor:
// ~3 dependent scattered round trips per level and 10 separate 32-byte segments per lane and level).
//   * nodes are renumbered breadth-first: the children of a node are adjacent 48-byte records in file order, so a
//     level of the descent is ONE contiguous read (480 bytes for k = 10) with no index hop;
//   * a descriptor is descended by 16 lanes (one DPP row): lane c takes child c's record (three 16-byte requests),
//     its 256-bit Hamming distance to the query (held in 8 registers by every lane of the row), and the row takes
//     min(distance << 8 | c) by four row rotations -- the strict '<' scan of :1241-1250 keeps the FIRST minimum in
//     file order, which is the smallest key; nodes with 17..32 children take a second trip;
//   * the winner's (firstChild, nChild) -- part of the record its lane already holds -- goes to the row by
//     ds_bpermute: ONE dependent memory round trip per level, L per descriptor;
//   * each row runs U = 2 descriptors at once (independent chains, all their requests in flight together).
// Batched forms build the DBoW2::FeatureVector of every frame on the device (64-bit (node, feature) keys written by
// the transform kernel, one bitonic sort per frame in LDS).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "kernels.h"
#include "match_kernels.h"

using namespace orbfe;

int orbfe_set_error_(int code, const char* msg);
static int vfail(int code, const std::string& m) { return orbfe_set_error_(code, m.c_str()); }
#define VHIP(expr)                                                                                   \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) return vfail(ORBFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct orbfe_vocabulary {
  int device = 0;
  int k = 0, L = 0, scoring = 0, weighting = 0;
  int nNodes = 0, nWords = 0;
  hipStream_t stream = nullptr;
  VocabDevice d = {};
  // grow-only scratch of the host-array entry point
  uint8_t* d_desc = nullptr; uint32_t* d_word = nullptr; uint32_t* d_node = nullptr; double* d_weight = nullptr;
  int scratchCap = 0;
  // grow-only workspace of the batched device path
  uint32_t* w_nodes = nullptr; int32_t* w_offsets = nullptr; uint32_t* w_indices = nullptr; int32_t* w_count = nullptr;
  int8_t* w_bin = nullptr; unsigned long long* w_keys = nullptr;
  unsigned long long* fk_keys = nullptr; size_t fkCap = 0;  // key scratch of orbfe_vocabulary_featvec_batch_device
  size_t wFrames = 0; int wCap = 0;
  hipStream_t lastStream = nullptr;  // stream of the last batched call (its workspace may still be in use there)
  hipEvent_t evFv[32] = {}, evBoundary[32] = {};  // per sub-batch: FeatureVectors built / boundary pair searched
  orbfe_extractor* lastMulti = nullptr;  // extractor whose sub-batch streams ran the last per-sub-batch call
};

namespace {

__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {  // minimum over the 16 lanes of a DPP row, in every lane
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:8
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:4
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:2
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:1
  return v;
}

struct NodeRec { uint4 a, b, m; };  // descriptor (a, b), m = (firstChild, nChild | positive << 8, id, word)

__device__ __forceinline__ NodeRec load_node(const VocabNode* nodes, uint32_t pos) {
  const uint4* p = reinterpret_cast<const uint4*>(nodes + pos);
  NodeRec r;
  r.a = p[0]; r.b = p[1]; r.m = p[2];
  return r;
}

__device__ __forceinline__ uint32_t hdist(const uint4 fa, const uint4 fb, const NodeRec& r) {
  return __popc(fa.x ^ r.a.x) + __popc(fa.y ^ r.a.y) + __popc(fa.z ^ r.a.z) + __popc(fa.w ^ r.a.w) +
         __popc(fb.x ^ r.b.x) + __popc(fb.y ^ r.b.y) + __popc(fb.z ^ r.b.z) + __popc(fb.w ^ r.b.w);
}

// transform(feature, word_id, weight, nid, levelsup), :1218-1259, for descriptors [0, n[f]) of every frame f.
// Workgroup = 16 rows x U descriptors; lane c of a row owns child c (and c + 16) of the row's current nodes.
// Every lane runs every iteration (the row reductions and ds_bpermute need all source lanes): rows whose
// descriptors have reached a leaf, and slots past the frame's count, just stop issuing loads.
template <int U>
__global__ __launch_bounds__(256) void k_vocab_transform16(VocabDevice v, const uint8_t* __restrict__ desc,
                                                           const int32_t* __restrict__ nPerFrame, int nFixed, int capacity,
                                                           int frameStep, int nidLevel, uint32_t* __restrict__ word,
                                                           double* __restrict__ weight, uint32_t* __restrict__ node,
                                                           unsigned long long* __restrict__ keys) {
  const int tid = threadIdx.x, c = tid & 15, row = tid >> 4, f = blockIdx.y;
  const int fi = f * frameStep;  // input slot (descriptors, count); outputs are dense per frame
  int n = nPerFrame ? nPerFrame[fi] : nFixed;
  if (n > capacity) n = capacity;
  const int base = (blockIdx.x * 16 + row) * U;
  if (blockIdx.x * 16 * U >= n && !keys) return;  // whole workgroup past the frame's count
  const size_t fo = (size_t)f * capacity, fin = (size_t)fi * capacity;
  const int laneBase = (tid & 63) & ~15;
  uint4 fa[U], fb[U];
  uint32_t first[U], cnt[U], nid[U], leafWord[U], leafPos[U], leafPositive[U];
  bool live[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int i = base + u;
    live[u] = i < n && v.rootChildren > 0;  // an empty vocabulary leaves (0, 0, 0), :1134
    const uint4* q = reinterpret_cast<const uint4*>(desc + (fin + (size_t)(live[u] ? i : 0)) * 32);
    fa[u] = live[u] ? q[0] : make_uint4(0, 0, 0, 0);
    fb[u] = live[u] ? q[1] : make_uint4(0, 0, 0, 0);
    first[u] = 1; cnt[u] = live[u] ? v.rootChildren : 0;
    nid[u] = 0; leafWord[u] = 0; leafPos[u] = 0; leafPositive[u] = 0;
  }
  for (int level = 1;; level++) {
    bool any = false;
#pragma unroll
    for (int u = 0; u < U; u++) any |= cnt[u] > 0;
    if (!__any(any)) break;
    NodeRec r0[U] = {}, r1[U] = {};
    bool two = false;
#pragma unroll
    for (int u = 0; u < U; u++) {
      two |= cnt[u] > 16;
      if ((uint32_t)c < cnt[u]) r0[u] = load_node(v.nodes, first[u] + c);
    }
    two = __any(two);  // wave-uniform: a node with more than 16 children anywhere in the wave
    if (two) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((uint32_t)c + 16 < cnt[u]) r1[u] = load_node(v.nodes, first[u] + c + 16);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t key = 0x7fffffffu;
      if ((uint32_t)c < cnt[u]) key = (hdist(fa[u], fb[u], r0[u]) << 8) | (uint32_t)c;
      if (two && (uint32_t)c + 16 < cnt[u]) {
        const uint32_t k1 = (hdist(fa[u], fb[u], r1[u]) << 8) | (uint32_t)(c + 16);
        key = k1 < key ? k1 : key;
      }
      key = row_min_u32(key);
      const uint32_t cw = key & 0xff;            // winning child, the same in every lane of the row
      const bool hi = cw >= 16;
      const int src = (laneBase + (int)(cw & 15)) << 2;
      const uint4 m = (two && hi) ? r1[u].m : r0[u].m;
      const bool act = cnt[u] > 0;               // uniform in the row
      const uint32_t pos = first[u] + cw;
      const uint32_t nf = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m.x);
      const uint32_t nc = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m.y);
      if (level == nidLevel) {                   // uniform in the wave
        const uint32_t id = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m.z);
        if (act) nid[u] = id;
      }
      if (act && (nc & 0xff) == 0) {             // reached a leaf: isLeaf() ends the loop (:1257)
        leafWord[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)m.w);
        leafPos[u] = pos;
        leafPositive[u] = (nc >> 8) & 1;
      }
      if (act) { first[u] = nf; cnt[u] = nc & 0xff; }
    }
  }
  if (c == 0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = base + u;
      if (i >= capacity) continue;
      if (keys) keys[fo + i] = (live[u] && leafPositive[u]) ? (((unsigned long long)nid[u] << 32) | (unsigned)i) : ~0ull;  // "not stopped" (:1161)
      if (i >= n) continue;
      if (node) node[fo + i] = nid[u];
      if (word) {
        const int w = (int)leafWord[u];
        word[fo + i] = live[u] && w >= 0 ? (uint32_t)w : 0u;
        weight[fo + i] = live[u] ? v.weight[leafPos[u]] : 0.0;
      }
    }
  }
}

// Per frame: build the DBoW2::FeatureVector as CSR (node ids ascending; inside a node the feature indices
// ascending = addFeature order, FeatureVector.cpp:31-45) with one bitonic sort in LDS of the 64-bit keys
// (node << 32 | feature) the transform kernel wrote.
__global__ __launch_bounds__(256) void k_vocab_featvec(FeatVecBatch b) {
  extern __shared__ unsigned long long keys[];  // b.sortN entries
  __shared__ int waveTot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.x;
  const unsigned long long* src = b.keys + (size_t)f * b.capacity;
  for (int i = tid; i < b.sortN; i += 256) keys[i] = i < b.capacity ? src[i] : ~0ull;  // padding sorts last
  __syncthreads();
  for (int k2 = 2; k2 <= b.sortN; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < b.sortN; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a0 = keys[i], a1 = keys[ixj];
          const bool up = (i & k2) == 0;
          if ((a0 > a1) == up) { keys[i] = a1; keys[ixj] = a0; }
        }
      }
      __syncthreads();
    }
  // segment heads -> node list + offsets; indices in sorted order
  uint32_t* nodes = b.fvNodes + (size_t)f * b.capacity;
  int32_t* offs = b.fvOffsets + (size_t)f * (b.capacity + 1);
  uint32_t* idx = b.fvIndices + (size_t)f * b.capacity;
  int run = 0, used = 0;
  for (int base = 0; base < b.sortN; base += 256) {
    const int i = base + tid;
    const unsigned long long key = i < b.sortN ? keys[i] : ~0ull;
    const bool valid = key != ~0ull;
    const bool head = valid && (i == 0 || (keys[i - 1] >> 32) != (key >> 32));
    const unsigned long long bal = __ballot(head);
    const unsigned long long balV = __ballot(valid);
    if (lane == 0) waveTot[wave] = __popcll(bal);
    __syncthreads();
    int o = run + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) o += waveTot[w];
    run += waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
    if (valid) idx[i] = (uint32_t)(key & 0xffffffffu);
    if (head) { nodes[o] = (uint32_t)(key >> 32); offs[o] = i; }
    used += __popcll(balV);  // per-wave; summed below
    __syncthreads();
  }
  // number of used features = first padding position = total valid (identical for every wave after reduction)
  __shared__ int usedTot[4];
  if (lane == 0) usedTot[wave] = used;
  __syncthreads();
  if (tid == 0) {
    const int total = usedTot[0] + usedTot[1] + usedTot[2] + usedTot[3];
    offs[run] = total;
    b.fvCount[f] = run;
  }
}

}  // namespace

namespace orbfe {
constexpr int kVocabU = 2;  // descriptors per 16-lane row
void launch_vocab_transform(hipStream_t s, const VocabDevice& v, const uint8_t* desc, const int32_t* nPerFrame, int nFixed,
                            int capacity, int frameStep, int nFrames, int nidLevel, uint32_t* word, double* weight, uint32_t* node,
                            unsigned long long* keys) {
  if (nFrames <= 0 || capacity <= 0) return;
  const int per = 16 * kVocabU, span = (nPerFrame || keys) ? capacity : nFixed;
  if (span <= 0) return;
  hipLaunchKernelGGL(k_vocab_transform16<kVocabU>, dim3((span + per - 1) / per, nFrames), dim3(256), 0, s, v, desc, nPerFrame,
                     nFixed, capacity, frameStep > 1 ? frameStep : 1, nidLevel, word, weight, node, keys);
}
void launch_vocab_featvec(hipStream_t s, const VocabDevice& v, const FeatVecBatch& b, int nFrames, int nidLevel) {
  if (nFrames <= 0) return;
  launch_vocab_transform(s, v, b.desc, b.n, 0, b.capacity, b.frameStep, nFrames, nidLevel, b.word, b.weight, nullptr, b.keys);
  hipLaunchKernelGGL(k_vocab_featvec, dim3(nFrames), dim3(256), (size_t)b.sortN * 8, s, b);
}
}  // namespace orbfe

static void vocab_free_device(orbfe_vocabulary* v) {
  if (v->d.nodes) (void)hipFree((void*)v->d.nodes);
  if (v->d.weight) (void)hipFree((void*)v->d.weight);
  if (v->d_desc) (void)hipFree(v->d_desc);
  if (v->d_word) (void)hipFree(v->d_word);
  if (v->d_node) (void)hipFree(v->d_node);
  if (v->d_weight) (void)hipFree(v->d_weight);
  if (v->w_nodes) (void)hipFree(v->w_nodes);
  if (v->w_offsets) (void)hipFree(v->w_offsets);
  if (v->w_indices) (void)hipFree(v->w_indices);
  if (v->w_count) (void)hipFree(v->w_count);
  if (v->w_bin) (void)hipFree(v->w_bin);
  if (v->w_keys) (void)hipFree(v->w_keys);
  if (v->fk_keys) (void)hipFree(v->fk_keys);
  for (int i = 0; i < 32; i++) {
    if (v->evFv[i]) (void)hipEventDestroy(v->evFv[i]);
    if (v->evBoundary[i]) (void)hipEventDestroy(v->evBoundary[i]);
  }
  if (v->stream) (void)hipStreamDestroy(v->stream);
}

// Nodes 1..n of a DBoW2 tree as the text file lists them (node 0 = the root) -> BFS records on the device.
static int vocab_upload(orbfe_vocabulary* v, const int32_t* parent, const uint8_t* desc, const double* weight,
                        const int32_t* wordId) {
  const int n = v->nNodes;
  std::vector<int32_t> off((size_t)n + 1, 0), idx((size_t)(n > 1 ? n - 1 : 1), 0);
  for (int i = 1; i < n; i++) {
    if (parent[i] < 0 || parent[i] >= n) return vfail(ORBFE_ERR_INVALID, "vocabulary: parent id out of range");
    // every descent must terminate: a child's id is larger than its parent's in a well-formed file
    if (parent[i] >= i) return vfail(ORBFE_ERR_INVALID, "vocabulary: node listed before its parent");
    off[parent[i] + 1]++;
  }
  for (int i = 0; i < n; i++) {
    if (off[i + 1] > 32) return vfail(ORBFE_ERR_INVALID, "vocabulary: a node has more than 32 children");
    off[i + 1] += off[i];
  }
  std::vector<int32_t> cur(off.begin(), off.begin() + n);
  for (int i = 1; i < n; i++) idx[cur[parent[i]]++] = i;  // children in file order (:1386)
  // breadth-first positions: the children of a node become adjacent records, in file order
  std::vector<int32_t> order((size_t)n), pos((size_t)n);
  order[0] = 0;
  int filled = 1;
  for (int q = 0; q < filled; q++) {
    const int id = order[q];
    pos[id] = q;
    for (int c = off[id]; c < off[id + 1]; c++) order[filled++] = idx[c];
  }
  if (filled != n) return vfail(ORBFE_ERR_INVALID, "vocabulary: unreachable nodes");
  std::vector<VocabNode> rec((size_t)n);
  std::vector<double> wpos((size_t)n);
  for (int q = 0; q < n; q++) {
    const int id = order[q];
    VocabNode& r = rec[q];
    memcpy(r.d, desc + (size_t)id * 32, 32);
    const int nc = off[id + 1] - off[id];
    r.firstChild = nc ? (uint32_t)pos[idx[off[id]]] : 0u;  // == q's children are consecutive from here
    r.nChild = (uint32_t)nc | (weight[id] > 0 ? 0x100u : 0u);
    r.id = (uint32_t)id;
    r.word = wordId[id];
    wpos[q] = weight[id];
  }
  VHIP(hipSetDevice(v->device));
  VHIP(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
  VocabNode* dn; double* dwt;
  VHIP(hipMalloc((void**)&dn, (size_t)n * sizeof(VocabNode)));
  v->d.nodes = dn;
  VHIP(hipMalloc((void**)&dwt, (size_t)n * 8));
  v->d.weight = dwt;
  v->d.rootChildren = (uint32_t)(off[1] - off[0]);
  VHIP(hipMemcpy(dn, rec.data(), (size_t)n * sizeof(VocabNode), hipMemcpyHostToDevice));
  VHIP(hipMemcpy(dwt, wpos.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  return ORBFE_OK;
}

static int vocab_finish(orbfe_vocabulary* v, const std::vector<int32_t>& parent, const std::vector<uint8_t>& desc,
                        const std::vector<double>& weight, const std::vector<int32_t>& wordId, orbfe_vocabulary** out) {
  int rc = vocab_upload(v, parent.data(), desc.data(), weight.data(), wordId.data());
  if (rc) { vocab_free_device(v); delete v; return rc; }
  *out = v;
  return ORBFE_OK;
}

extern "C" int orbfe_vocabulary_load_text(const char* path, int device, orbfe_vocabulary** out) {
  if (!path || !out) return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: NULL argument");
  *out = nullptr;
  FILE* fp = fopen(path, "rb");
  if (!fp) return vfail(ORBFE_ERR_INVALID, std::string("vocabulary_load_text: cannot open ") + path);
  std::string txt;  // ORBvoc.txt is 145 MB: one read, then pointer parsing (no stream per line)
  {
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) txt.append(buf, got);
    fclose(fp);
  }
  char* p = &txt[0];
  char* const end = p + txt.size();  // (*end is the string's own NUL)
  auto line_end = [&](char* q) { while (q < end && *q != '\n') q++; return q; };
  if (p == end) return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: empty file");
  int k = -1, L = -1, n1 = -1, n2 = -1;
  {
    const std::string head(p, line_end(p));
    std::stringstream ss(head);
    ss >> k >> L >> n1 >> n2;
  }
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)  // :1357-1361
    return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: not a DBoW2 text vocabulary");
  std::vector<int32_t> parent(1, 0), wordId(1, -1);
  std::vector<uint8_t> desc(32, 0);
  std::vector<double> weight(1, 0.0);
  const size_t guess = txt.size() / 100 + 16;
  parent.reserve(guess); wordId.reserve(guess); weight.reserve(guess); desc.reserve(guess * 32);
  int nWords = 0;
  p = line_end(p);
  while (p < end) {
    p++;  // the '\n'
    char* le = line_end(p);
    const char* q = p;
    while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
    if (q < le) {  // see orbfe.h: empty lines are ignored
      // A node is ONE line (the reference parses each through its own getline + stringstream, :1374-1417): the line is
      // NUL-terminated in place so that strtol / strtod -- which skip '\n' as white space -- cannot run on into the next
      // node's tokens; the fields a short line lacks read as 0, like the reference's failed `>>` extractions
      *le = 0;
      char* r;
      const long pid = strtol(q, &r, 10);
      const long leaf = strtol(r, &r, 10);
      uint8_t d[32];
      for (int i = 0; i < 32; i++) d[i] = (uint8_t)strtol(r, &r, 10);
      const double w = strtod(r, &r);
      parent.push_back((int32_t)pid);
      desc.insert(desc.end(), d, d + 32);
      weight.push_back(w);
      wordId.push_back(leaf > 0 ? nWords++ : -1);
    }
    p = le;
  }
  orbfe_vocabulary* v = new (std::nothrow) orbfe_vocabulary();
  if (!v) return vfail(ORBFE_ERR_NOMEM, "out of memory");
  v->device = device; v->k = k; v->L = L; v->scoring = n1; v->weighting = n2;
  v->nNodes = (int)parent.size();
  v->nWords = nWords;
  return vocab_finish(v, parent, desc, weight, wordId, out);
}

// The same tree from arrays (what the lines of the text file hold, node i+1 = entry i): a vocabulary built or
// converted by the caller, or a synthetic one of ORBvoc's size without a 145 MB text detour.
extern "C" int orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                                       const uint8_t* is_leaf, const uint8_t* descriptors, const double* weight,
                                       int device, orbfe_vocabulary** out) {
  if (!out) return vfail(ORBFE_ERR_INVALID, "vocabulary_create: NULL argument");
  *out = nullptr;
  if (n_nodes < 0 || (n_nodes > 0 && (!parent || !is_leaf || !descriptors || !weight)))
    return vfail(ORBFE_ERR_INVALID, "vocabulary_create: NULL argument");
  if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
    return vfail(ORBFE_ERR_INVALID, "vocabulary_create: header values out of range (:1357-1361)");
  const size_t n = (size_t)n_nodes + 1;
  std::vector<int32_t> par(n, 0), wordId(n, -1);
  std::vector<uint8_t> desc(n * 32, 0);
  std::vector<double> wt(n, 0.0);
  int nWords = 0;
  for (int i = 0; i < n_nodes; i++) {
    par[i + 1] = parent[i];
    wt[i + 1] = weight[i];
    wordId[i + 1] = is_leaf[i] ? nWords++ : -1;
  }
  if (n_nodes) memcpy(desc.data() + 32, descriptors, (size_t)n_nodes * 32);
  orbfe_vocabulary* v = new (std::nothrow) orbfe_vocabulary();
  if (!v) return vfail(ORBFE_ERR_NOMEM, "out of memory");
  v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->nNodes = (int)n;
  v->nWords = nWords;
  return vocab_finish(v, par, desc, wt, wordId, out);
}

extern "C" void orbfe_vocabulary_destroy(orbfe_vocabulary* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->stream) (void)hipStreamSynchronize(v->stream);
  vocab_free_device(v);
  delete v;
}

extern "C" int orbfe_vocabulary_info(const orbfe_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (!v) return vfail(ORBFE_ERR_INVALID, "NULL vocabulary");
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->nNodes;
  if (n_words) *n_words = v->nWords;
  return ORBFE_OK;
}

extern "C" int orbfe_vocabulary_transform(orbfe_vocabulary* v, const uint8_t* descriptors, int n, int levelsup,
                                          uint32_t* word_id, double* weight, uint32_t* node_id) {
  if (!v || n < 0 || (n > 0 && (!descriptors || !word_id || !weight || !node_id)))
    return vfail(ORBFE_ERR_INVALID, "vocabulary_transform: bad argument");
  if (n == 0) return 0;
  VHIP(hipSetDevice(v->device));
  if (n > v->scratchCap) {
    VHIP(hipStreamSynchronize(v->stream));
    if (v->d_desc) { (void)hipFree(v->d_desc); (void)hipFree(v->d_word); (void)hipFree(v->d_node); (void)hipFree(v->d_weight); }
    v->d_desc = nullptr; v->d_word = nullptr; v->d_node = nullptr; v->d_weight = nullptr;
    v->scratchCap = 0;
    const int cap = n + n / 2 + 256;
    VHIP(hipMalloc((void**)&v->d_desc, (size_t)cap * 32));
    VHIP(hipMalloc((void**)&v->d_word, (size_t)cap * 4));
    VHIP(hipMalloc((void**)&v->d_node, (size_t)cap * 4));
    VHIP(hipMalloc((void**)&v->d_weight, (size_t)cap * 8));
    v->scratchCap = cap;
  }
  VHIP(hipMemcpyAsync(v->d_desc, descriptors, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
  launch_vocab_transform(v->stream, v->d, v->d_desc, nullptr, n, n, 1, 1, v->L - levelsup, v->d_word, v->d_weight, v->d_node,
                         nullptr);
  VHIP(hipGetLastError());
  VHIP(hipMemcpyAsync(word_id, v->d_word, (size_t)n * 4, hipMemcpyDeviceToHost, v->stream));
  VHIP(hipMemcpyAsync(weight, v->d_weight, (size_t)n * 8, hipMemcpyDeviceToHost, v->stream));
  VHIP(hipMemcpyAsync(node_id, v->d_node, (size_t)n * 4, hipMemcpyDeviceToHost, v->stream));
  VHIP(hipStreamSynchronize(v->stream));
  int used = 0;
  for (int i = 0; i < n; i++) used += weight[i] > 0;
  return used;
}

static int ensure_bow_workspace(orbfe_vocabulary* v, int nFrames, int capacity) {
  if ((size_t)nFrames <= v->wFrames && capacity <= v->wCap) return ORBFE_OK;
  VHIP(hipStreamSynchronize(v->stream));
  if (v->w_nodes) { (void)hipFree(v->w_nodes); (void)hipFree(v->w_offsets); (void)hipFree(v->w_indices); (void)hipFree(v->w_count); (void)hipFree(v->w_bin); (void)hipFree(v->w_keys); }
  v->w_nodes = nullptr; v->w_offsets = nullptr; v->w_indices = nullptr; v->w_count = nullptr; v->w_bin = nullptr; v->w_keys = nullptr;
  v->wFrames = 0; v->wCap = 0;
  const size_t F = (size_t)nFrames, c = (size_t)capacity;
  VHIP(hipMalloc((void**)&v->w_nodes, F * c * 4));
  VHIP(hipMalloc((void**)&v->w_offsets, F * (c + 1) * 4));
  VHIP(hipMalloc((void**)&v->w_indices, F * c * 4));
  VHIP(hipMalloc((void**)&v->w_count, F * 4));
  VHIP(hipMalloc((void**)&v->w_bin, F * c));
  VHIP(hipMalloc((void**)&v->w_keys, F * c * 8));
  v->wFrames = F;
  v->wCap = capacity;
  return ORBFE_OK;
}

static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// DBoW2::FeatureVector of every frame of a device-resident batch (Frame::ComputeBoW, src/Frame.cc:433-440)
extern "C" int orbfe_vocabulary_featvec_batch_device(orbfe_vocabulary* v, const uint8_t* d_descriptors,
                                                     const int32_t* d_n, int n_frames, int capacity, int levelsup,
                                                     uint32_t* d_fv_nodes, int32_t* d_fv_offsets,
                                                     uint32_t* d_fv_indices, int32_t* d_fv_count, uint32_t* d_word,
                                                     double* d_weight) {
  if (!v || n_frames < 0 || capacity <= 0 || !d_descriptors || !d_n || !d_fv_nodes || !d_fv_offsets ||
      !d_fv_indices || !d_fv_count || ((d_word == nullptr) != (d_weight == nullptr)))
    return vfail(ORBFE_ERR_INVALID, "vocabulary_featvec_batch_device: bad argument");
  if (n_frames == 0) return ORBFE_OK;
  const int sortN = next_pow2(capacity);
  if ((size_t)sortN * 8 > 96 * 1024) return vfail(ORBFE_ERR_INVALID, "vocabulary_featvec_batch_device: capacity > 8192");
  VHIP(hipSetDevice(v->device));
  FeatVecBatch b = {};
  b.desc = d_descriptors; b.n = d_n; b.capacity = capacity; b.sortN = sortN;
  b.word = d_word; b.weight = d_weight;
  b.fvNodes = d_fv_nodes; b.fvOffsets = d_fv_offsets; b.fvIndices = d_fv_indices; b.fvCount = d_fv_count;
  const size_t need = (size_t)n_frames * capacity;
  if (need > v->fkCap) {
    VHIP(hipStreamSynchronize(v->stream));
    if (v->fk_keys) (void)hipFree(v->fk_keys);
    v->fk_keys = nullptr; v->fkCap = 0;
    VHIP(hipMalloc((void**)&v->fk_keys, need * 8));
    v->fkCap = need;
  }
  b.keys = v->fk_keys;
  if ((size_t)sortN * 8 > 64 * 1024) {
    static thread_local bool configured = false;
    if (!configured) {
      VHIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_vocab_featvec), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      configured = true;
    }
  }
  launch_vocab_featvec(v->stream, v->d, b, n_frames, v->L - levelsup);
  VHIP(hipGetLastError());
  VHIP(hipStreamSynchronize(v->stream));
  return ORBFE_OK;
}

// Tracking::TrackReferenceKeyFrame-style matching over a device-resident batch: for t = 1..n-1,
// ComputeBoW of both frames then SearchByBoW(KF = frame t-1 with a MapPoint on every feature,
// F = frame t)  (src/Tracking.cc:836-843, src/ORBmatcher.cc:185-325).
extern "C" int orbfe_extractor_consumer_begin_(orbfe_extractor* e, hipStream_t* s);
extern "C" int orbfe_extractor_consumer_end_(orbfe_extractor* e);
extern "C" int orbfe_extractor_split_(orbfe_extractor* e, int* S, int* per, int* frames, int* lanes, hipStream_t* streams,
                                      hipEvent_t* chunkDone);
extern "C" void orbfe_extractor_stage_mark_(orbfe_extractor* e, int stage, int sub, int isEnd, hipStream_t s, int frames);

// e != NULL: enqueue on the extractor's stream, ordered behind every sub-batch of its last extract call, and
// return without waiting (orbfe_extractor_synchronize() to wait); e == NULL: the vocabulary's own stream, waits.
// step: frame t of the call is frame slot t*step of the arrays (2 = the left frames of an L,R-interleaved stereo batch)
static int bow_match_consecutive(orbfe_vocabulary* v, orbfe_extractor* e, int n_frames,
                                 const orbfe_keypoint* d_keypoints, const uint8_t* d_descriptors,
                                 const int32_t* d_n, int capacity, int levelsup, float nnratio,
                                 int check_orientation, int32_t* d_match, int32_t* d_nmatches, int step = 1) {
  if (!v || n_frames < 0 || capacity <= 0 || capacity > 65535 || !d_keypoints || !d_descriptors || !d_n || !d_match ||
      !d_nmatches)
    return vfail(ORBFE_ERR_INVALID, "bow_match_consecutive_batch_device: bad argument");
  if (n_frames < 2) return ORBFE_OK;
  const int sortN = next_pow2(capacity);
  if ((size_t)sortN * 8 > 64 * 1024) return vfail(ORBFE_ERR_INVALID, "bow_match_consecutive_batch_device: capacity > 8192");
  VHIP(hipSetDevice(v->device));
  int rc;
  if ((size_t)n_frames > v->wFrames || capacity > v->wCap || (v->lastMulti && v->lastMulti != e)) {
    // the workspace may still be in use on the streams of the previous call
    if (v->lastStream) VHIP(hipStreamSynchronize(v->lastStream));
    if (v->lastMulti && (rc = orbfe_extractor_synchronize(v->lastMulti))) return rc;
    v->lastMulti = nullptr;
  }
  rc = ensure_bow_workspace(v, n_frames, capacity);
  if (rc) return rc;
  FeatVecBatch fb = {};
  fb.desc = d_descriptors; fb.n = d_n; fb.capacity = capacity; fb.sortN = sortN; fb.frameStep = step;
  fb.fvNodes = v->w_nodes; fb.fvOffsets = v->w_offsets; fb.fvIndices = v->w_indices; fb.fvCount = v->w_count;
  fb.keys = v->w_keys;
  BowBatch bb = {};
  bb.kp = reinterpret_cast<const float*>(d_keypoints); bb.desc = d_descriptors; bb.capacity = capacity; bb.frameStep = step;
  bb.fvNodes = v->w_nodes; bb.fvOffsets = v->w_offsets; bb.fvIndices = v->w_indices; bb.fvCount = v->w_count;
  bb.nnratio = nnratio; bb.match = d_match; bb.bin = v->w_bin;
  const size_t c = (size_t)capacity;
  // nodes at level L - levelsup of a k-ary tree: at most k^level (FeatureVector entries per frame)
  int maxNodes = 1;
  for (int l = 0; l < v->L - levelsup && maxNodes < capacity; l++) maxNodes = maxNodes > capacity / (v->k > 1 ? v->k : 1) ? capacity : maxNodes * v->k;
  auto featvec_range = [&](hipStream_t st, int f0, int n) {
    FeatVecBatch r = fb;
    r.desc += (size_t)f0 * step * c * 32; r.n += (size_t)f0 * step;
    r.fvNodes += (size_t)f0 * c; r.fvOffsets += (size_t)f0 * (c + 1); r.fvIndices += (size_t)f0 * c; r.fvCount += f0;
    r.keys += (size_t)f0 * c;
    launch_vocab_featvec(st, v->d, r, n, v->L - levelsup);
  };
  auto search_range = [&](hipStream_t st, int p0, int np) -> int {
    if (np <= 0) return ORBFE_OK;
    VHIP(hipMemsetAsync(d_match + (size_t)p0 * c, 0xff, (size_t)np * c * 4, st));
    VHIP(hipMemsetAsync(v->w_bin + (size_t)p0 * c, 0, (size_t)np * c, st));
    BowBatch r = bb;
    r.kp += (size_t)p0 * step * c * 7; r.desc += (size_t)p0 * step * c * 32;
    r.fvNodes += (size_t)p0 * c; r.fvOffsets += (size_t)p0 * (c + 1); r.fvIndices += (size_t)p0 * c; r.fvCount += p0;
    r.match += (size_t)p0 * c; r.bin += (size_t)p0 * c;
    launch_search_by_bow_batch(st, r, np, check_orientation, d_nmatches + p0, maxNodes);
    return ORBFE_OK;
  };
  if (e) {
    int S = 0, per = 0, frames = 0, lanes = 0;
    hipStream_t streams[32];  // orbfe_extractor::kMaxStreams
    hipEvent_t chunkDone[32];
    if ((rc = orbfe_extractor_split_(e, &S, &per, &frames, &lanes, streams, chunkDone))) return rc;
    if (!lanes && S > 1 && frames == n_frames * step && per % step == 0 && per / step >= 2) {
      per /= step;  // frames of THIS call per sub-batch
      // Per sub-batch, on the sub-batch's own stream right behind its extraction (no join of the streams): the
      // FeatureVectors of its frames, then its consecutive pairs.  The pair that straddles two sub-batches
      // (last frame of i-1, first frame of i) runs on stream i behind an event of stream i-1's FeatureVectors, and
      // stream i-1 is made to wait for it before anything enqueued later (the next extract call) overwrites that frame.
      if (v->lastStream) { VHIP(hipStreamSynchronize(v->lastStream)); v->lastStream = nullptr; }
      v->lastMulti = e;
      for (int i = 0; i < 32; i++)
        if (!v->evFv[i]) {
          VHIP(hipEventCreateWithFlags(&v->evFv[i], hipEventDisableTiming));
          VHIP(hipEventCreateWithFlags(&v->evBoundary[i], hipEventDisableTiming));
        }
      int nSub = 0;
      for (int i = 0; i < S; i++) {
        const int f0 = i * per, n = f0 + per <= n_frames ? per : n_frames - f0;
        if (n <= 0) break;
        nSub = i + 1;
        orbfe_extractor_stage_mark_(e, ORBFE_STAGE_MATCH, i, 0, streams[i], n);
        featvec_range(streams[i], f0, n);
        VHIP(hipEventRecord(v->evFv[i], streams[i]));
      }
      for (int i = 0; i < nSub; i++) {
        const int f0 = i * per, n = f0 + per <= n_frames ? per : n_frames - f0;
        if (i > 0) {
          // the straddling pair first, on its own: the earlier stream is released as soon as this one launch is done
          // instead of after the whole sub-batch's pairs
          VHIP(hipStreamWaitEvent(streams[i], v->evFv[i - 1], 0));
          if ((rc = search_range(streams[i], f0 - 1, 1))) return rc;
          VHIP(hipEventRecord(v->evBoundary[i], streams[i]));
          VHIP(hipStreamWaitEvent(streams[i - 1], v->evBoundary[i], 0));
        }
        if ((rc = search_range(streams[i], f0, n - 1))) return rc;
        orbfe_extractor_stage_mark_(e, ORBFE_STAGE_MATCH, i, 1, streams[i], n);
        if (i > 0) VHIP(hipEventRecord(chunkDone[i], streams[i]));  // "sub-batch i done" now includes its matcher
      }
      VHIP(hipGetLastError());
      return ORBFE_OK;
    }
  }
  if (v->lastMulti) {  // back to one stream after a per-sub-batch call
    if ((rc = orbfe_extractor_synchronize(v->lastMulti))) return rc;
    v->lastMulti = nullptr;
  }
  hipStream_t s = v->stream;
  if (e) {
    if ((rc = orbfe_extractor_consumer_begin_(e, &s))) return rc;
    if (v->lastStream && v->lastStream != s) VHIP(hipStreamSynchronize(v->lastStream));
  } else if (v->lastStream && v->lastStream != s) {
    VHIP(hipStreamSynchronize(v->lastStream));
  }
  v->lastStream = s;
  featvec_range(s, 0, n_frames);
  if ((rc = search_range(s, 0, n_frames - 1))) return rc;
  VHIP(hipGetLastError());
  if (e) return orbfe_extractor_consumer_end_(e);
  VHIP(hipStreamSynchronize(s));
  return ORBFE_OK;
}

extern "C" int orbfe_bow_match_consecutive_batch_device(orbfe_vocabulary* v, int n_frames,
                                                        const orbfe_keypoint* d_keypoints,
                                                        const uint8_t* d_descriptors, const int32_t* d_n,
                                                        int capacity, int levelsup, float nnratio,
                                                        int check_orientation, int32_t* d_match,
                                                        int32_t* d_nmatches) {
  return bow_match_consecutive(v, nullptr, n_frames, d_keypoints, d_descriptors, d_n, capacity, levelsup, nnratio,
                               check_orientation, d_match, d_nmatches);
}

extern "C" int orbfe_bow_match_consecutive_batch_device_async(orbfe_vocabulary* v, orbfe_extractor* e, int n_frames,
                                                              const orbfe_keypoint* d_keypoints,
                                                              const uint8_t* d_descriptors, const int32_t* d_n,
                                                              int capacity, int levelsup, float nnratio,
                                                              int check_orientation, int32_t* d_match,
                                                              int32_t* d_nmatches) {
  if (!e) return vfail(ORBFE_ERR_INVALID, "bow_match_consecutive_batch_device_async: NULL extractor");
  return bow_match_consecutive(v, e, n_frames, d_keypoints, d_descriptors, d_n, capacity, levelsup, nnratio,
                               check_orientation, d_match, d_nmatches);
}

// The same over the LEFT frames of an (L0, R0, L1, R1, ...) stereo batch: Frame::ComputeBoW and SearchByBoW work on a
// stereo Frame's left keypoints (mvKeys / mDescriptors, src/Frame.cc:61-117); pair t-1 against pair t for t = 1..n_pairs-1.
extern "C" int orbfe_bow_match_consecutive_stereo_batch_device_async(orbfe_vocabulary* v, orbfe_extractor* e, int n_pairs,
                                                                     const orbfe_keypoint* d_keypoints,
                                                                     const uint8_t* d_descriptors, const int32_t* d_n,
                                                                     int capacity, int levelsup, float nnratio,
                                                                     int check_orientation, int32_t* d_match,
                                                                     int32_t* d_nmatches) {
  if (!e) return vfail(ORBFE_ERR_INVALID, "bow_match_consecutive_stereo_batch_device_async: NULL extractor");
  return bow_match_consecutive(v, e, n_pairs, d_keypoints, d_descriptors, d_n, capacity, levelsup, nnratio,
                               check_orientation, d_match, d_nmatches, 2);
}

// internal: device view for the batched BoW path (extractor.hip)
extern "C" int orbfe_vocabulary_device_(orbfe_vocabulary* v, VocabDevice* out, int* L, int* device) {
  if (!v) return vfail(ORBFE_ERR_INVALID, "NULL vocabulary");
  *out = v->d;
  *L = v->L;
  *device = v->device;
  return ORBFE_OK;
}
