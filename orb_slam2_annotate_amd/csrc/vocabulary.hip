// vocabulary.hip -- DBoW2 ORB vocabulary on the device: text loader (TemplatedVocabulary::
// loadFromTextFile, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424) and transform()
// (:1127-1194 feature loop, :1218-1259 k-ary descent with FORB::distance, FORB.cpp:81-101).
// the step immediately before SearchByBoW (src/Frame.cc:433-440).
//
// The tree is stored flat: node descriptors [n][32], child lists in file order as CSR, word id and
// weight per node.  transform is one THREAD per descriptor (L levels x k Hamming distances, the
// upper tree levels stay in cache); batched forms build the DBoW2::FeatureVector of every frame on
// the device (64-bit (node, feature) keys, one bitonic sort per frame in LDS).
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
  int8_t* w_bin = nullptr;
  size_t wFrames = 0; int wCap = 0;
  hipStream_t lastStream = nullptr;  // stream of the last batched call (its workspace may still be in use there)
  hipEvent_t evFv[32] = {}, evBoundary[32] = {};  // per sub-batch: FeatureVectors built / boundary pair searched
  orbfe_extractor* lastMulti = nullptr;  // extractor whose sub-batch streams ran the last per-sub-batch call
};

namespace {

__device__ __forceinline__ int hdist32(const uint32_t* a, const uint4 b0, const uint4 b1) {
  return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
         __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

// transform(feature, word_id, weight, nid, levelsup), :1218-1259
__device__ __forceinline__ void descend(const VocabDevice& v, const uint8_t* fdesc, int nidLevel, uint32_t* word,
                                        double* weight, uint32_t* node) {
  const uint4* f4 = reinterpret_cast<const uint4*>(fdesc);
  const uint4 fa = f4[0], fb = f4[1];
  const uint32_t fw[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
  int finalId = 0, level = 0;
  uint32_t nid = 0;
  do {
    ++level;
    const int b = v.childOff[finalId], e = v.childOff[finalId + 1];
    finalId = v.childIdx[b];
    const uint4* d4 = reinterpret_cast<const uint4*>(v.desc + (size_t)finalId * 32);
    int best = hdist32(fw, d4[0], d4[1]);
    for (int c = b + 1; c < e; c++) {
      const int id = v.childIdx[c];
      const uint4* q = reinterpret_cast<const uint4*>(v.desc + (size_t)id * 32);
      const int d = hdist32(fw, q[0], q[1]);
      if (d < best) { best = d; finalId = id; }  // strict: the first minimum wins
    }
    if (level == nidLevel) nid = (uint32_t)finalId;
  } while (v.childOff[finalId + 1] > v.childOff[finalId]);  // !isLeaf()
  const int w = v.wordId[finalId];
  *word = w >= 0 ? (uint32_t)w : 0u;
  *weight = v.weight[finalId];
  *node = nid;
}

__global__ __launch_bounds__(256) void k_vocab_transform(VocabDevice v, const uint8_t* __restrict__ desc, int n,
                                                         int nidLevel, uint32_t* __restrict__ word,
                                                         double* __restrict__ weight, uint32_t* __restrict__ node) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w = 0, nd = 0;
  double wt = 0;
  if (v.childOff[1] > v.childOff[0]) descend(v, desc + (size_t)i * 32, nidLevel, &w, &wt, &nd);
  word[i] = w;
  weight[i] = wt;
  node[i] = nd;
}

// Per frame: transform every keypoint descriptor, then build the DBoW2::FeatureVector as CSR
// (node ids ascending; inside a node the feature indices ascending = addFeature order,
// FeatureVector.cpp:31-45) with one bitonic sort of 64-bit keys (node << 32 | feature) in LDS.
__global__ __launch_bounds__(256) void k_vocab_featvec(VocabDevice v, FeatVecBatch b, int nidLevel) {
  extern __shared__ unsigned long long keys[];  // b.sortN entries
  __shared__ int waveTot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, f = blockIdx.x;
  int n = b.n[f];
  if (n > b.capacity) n = b.capacity;
  const uint8_t* desc = b.desc + (size_t)f * b.capacity * 32;
  const bool have = v.childOff[1] > v.childOff[0];
  for (int i = tid; i < b.sortN; i += 256) {
    unsigned long long key = ~0ull;  // padding sorts last
    if (i < n && have) {
      uint32_t w, nd;
      double wt;
      descend(v, desc + (size_t)i * 32, nidLevel, &w, &wt, &nd);
      if (b.word) { b.word[(size_t)f * b.capacity + i] = w; b.weight[(size_t)f * b.capacity + i] = wt; }
      if (wt > 0) key = ((unsigned long long)nd << 32) | (unsigned)i;  // "not stopped" (:1161)
    }
    keys[i] = key;
  }
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
void launch_vocab_featvec(hipStream_t s, const VocabDevice& v, const FeatVecBatch& b, int nFrames, int nidLevel) {
  if (nFrames <= 0) return;
  hipLaunchKernelGGL(k_vocab_featvec, dim3(nFrames), dim3(256), (size_t)b.sortN * 8, s, v, b, nidLevel);
}
}  // namespace orbfe

static void vocab_free_device(orbfe_vocabulary* v) {
  if (v->d.desc) (void)hipFree((void*)v->d.desc);
  if (v->d.childOff) (void)hipFree((void*)v->d.childOff);
  if (v->d.childIdx) (void)hipFree((void*)v->d.childIdx);
  if (v->d.wordId) (void)hipFree((void*)v->d.wordId);
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
  for (int i = 0; i < 32; i++) {
    if (v->evFv[i]) (void)hipEventDestroy(v->evFv[i]);
    if (v->evBoundary[i]) (void)hipEventDestroy(v->evBoundary[i]);
  }
  if (v->stream) (void)hipStreamDestroy(v->stream);
}

static int vocab_upload(orbfe_vocabulary* v, const std::vector<int32_t>& parent, const std::vector<uint8_t>& desc,
                        const std::vector<double>& weight, const std::vector<int32_t>& wordId) {
  const int n = v->nNodes;
  std::vector<int32_t> off((size_t)n + 1, 0), idx((size_t)(n > 1 ? n - 1 : 1), 0);
  for (int i = 1; i < n; i++) {
    if (parent[i] < 0 || parent[i] >= n) return vfail(ORBFE_ERR_INVALID, "vocabulary: parent id out of range");
    off[parent[i] + 1]++;
  }
  for (int i = 0; i < n; i++) off[i + 1] += off[i];
  std::vector<int32_t> cur(off.begin(), off.begin() + n);
  for (int i = 1; i < n; i++) idx[cur[parent[i]]++] = i;  // children in file order (:1386)
  // every descent must terminate: a child's id is larger than its parent's in a well-formed file
  for (int i = 1; i < n; i++)
    if (parent[i] >= i) return vfail(ORBFE_ERR_INVALID, "vocabulary: node listed before its parent");
  VHIP(hipSetDevice(v->device));
  VHIP(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
  uint8_t* dd; int32_t *doff, *didx, *dw; double* dwt;
  VHIP(hipMalloc((void**)&dd, (size_t)n * 32));
  VHIP(hipMalloc((void**)&doff, ((size_t)n + 1) * 4));
  VHIP(hipMalloc((void**)&didx, idx.size() * 4));
  VHIP(hipMalloc((void**)&dw, (size_t)n * 4));
  VHIP(hipMalloc((void**)&dwt, (size_t)n * 8));
  v->d.desc = dd; v->d.childOff = doff; v->d.childIdx = didx; v->d.wordId = dw; v->d.weight = dwt;
  VHIP(hipMemcpy(dd, desc.data(), (size_t)n * 32, hipMemcpyHostToDevice));
  VHIP(hipMemcpy(doff, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
  VHIP(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  VHIP(hipMemcpy(dw, wordId.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  VHIP(hipMemcpy(dwt, weight.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  return ORBFE_OK;
}

extern "C" int orbfe_vocabulary_load_text(const char* path, int device, orbfe_vocabulary** out) {
  if (!path || !out) return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: NULL argument");
  *out = nullptr;
  std::ifstream f(path);
  if (!f.is_open()) return vfail(ORBFE_ERR_INVALID, std::string("vocabulary_load_text: cannot open ") + path);
  std::string line;
  if (!std::getline(f, line)) return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: empty file");
  int k = -1, L = -1, n1 = -1, n2 = -1;
  {
    std::stringstream ss(line);
    ss >> k >> L >> n1 >> n2;
  }
  if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)  // :1357-1361
    return vfail(ORBFE_ERR_INVALID, "vocabulary_load_text: not a DBoW2 text vocabulary");
  std::vector<int32_t> parent(1, 0), wordId(1, -1);
  std::vector<uint8_t> desc(32, 0);
  std::vector<double> weight(1, 0.0);
  int nWords = 0;
  while (std::getline(f, line)) {
    if (line.find_first_not_of(" \t\r") == std::string::npos) continue;  // see orbfe.h: empty lines ignored
    std::stringstream ss(line);
    int pid = 0, leaf = 0;
    ss >> pid >> leaf;
    uint8_t d[32];
    for (int i = 0; i < 32; i++) {
      int x = 0;
      ss >> x;
      d[i] = (uint8_t)x;
    }
    double w = 0;
    ss >> w;
    parent.push_back(pid);
    desc.insert(desc.end(), d, d + 32);
    weight.push_back(w);
    wordId.push_back(leaf > 0 ? nWords++ : -1);
  }
  orbfe_vocabulary* v = new (std::nothrow) orbfe_vocabulary();
  if (!v) return vfail(ORBFE_ERR_NOMEM, "out of memory");
  v->device = device; v->k = k; v->L = L; v->scoring = n1; v->weighting = n2;
  v->nNodes = (int)parent.size();
  v->nWords = nWords;
  int rc = vocab_upload(v, parent, desc, weight, wordId);
  if (rc) { vocab_free_device(v); delete v; return rc; }
  *out = v;
  return ORBFE_OK;
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
  hipLaunchKernelGGL(k_vocab_transform, dim3((n + 255) / 256), dim3(256), 0, v->stream, v->d, v->d_desc, n,
                     v->L - levelsup, v->d_word, v->d_weight, v->d_node);
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
  if (v->w_nodes) { (void)hipFree(v->w_nodes); (void)hipFree(v->w_offsets); (void)hipFree(v->w_indices); (void)hipFree(v->w_count); (void)hipFree(v->w_bin); }
  v->w_nodes = nullptr; v->w_offsets = nullptr; v->w_indices = nullptr; v->w_count = nullptr; v->w_bin = nullptr;
  v->wFrames = 0; v->wCap = 0;
  const size_t F = (size_t)nFrames, c = (size_t)capacity;
  VHIP(hipMalloc((void**)&v->w_nodes, F * c * 4));
  VHIP(hipMalloc((void**)&v->w_offsets, F * (c + 1) * 4));
  VHIP(hipMalloc((void**)&v->w_indices, F * c * 4));
  VHIP(hipMalloc((void**)&v->w_count, F * 4));
  VHIP(hipMalloc((void**)&v->w_bin, F * c));
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
static int bow_match_consecutive(orbfe_vocabulary* v, orbfe_extractor* e, int n_frames,
                                 const orbfe_keypoint* d_keypoints, const uint8_t* d_descriptors,
                                 const int32_t* d_n, int capacity, int levelsup, float nnratio,
                                 int check_orientation, int32_t* d_match, int32_t* d_nmatches) {
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
  fb.desc = d_descriptors; fb.n = d_n; fb.capacity = capacity; fb.sortN = sortN;
  fb.fvNodes = v->w_nodes; fb.fvOffsets = v->w_offsets; fb.fvIndices = v->w_indices; fb.fvCount = v->w_count;
  BowBatch bb = {};
  bb.kp = reinterpret_cast<const float*>(d_keypoints); bb.desc = d_descriptors; bb.capacity = capacity;
  bb.fvNodes = v->w_nodes; bb.fvOffsets = v->w_offsets; bb.fvIndices = v->w_indices; bb.fvCount = v->w_count;
  bb.nnratio = nnratio; bb.match = d_match; bb.bin = v->w_bin;
  const size_t c = (size_t)capacity;
  // nodes at level L - levelsup of a k-ary tree: at most k^level (FeatureVector entries per frame)
  int maxNodes = 1;
  for (int l = 0; l < v->L - levelsup && maxNodes < capacity; l++) maxNodes = maxNodes > capacity / (v->k > 1 ? v->k : 1) ? capacity : maxNodes * v->k;
  auto featvec_range = [&](hipStream_t st, int f0, int n) {
    FeatVecBatch r = fb;
    r.desc += (size_t)f0 * c * 32; r.n += f0;
    r.fvNodes += (size_t)f0 * c; r.fvOffsets += (size_t)f0 * (c + 1); r.fvIndices += (size_t)f0 * c; r.fvCount += f0;
    launch_vocab_featvec(st, v->d, r, n, v->L - levelsup);
  };
  auto search_range = [&](hipStream_t st, int p0, int np) -> int {
    if (np <= 0) return ORBFE_OK;
    VHIP(hipMemsetAsync(d_match + (size_t)p0 * c, 0xff, (size_t)np * c * 4, st));
    VHIP(hipMemsetAsync(v->w_bin + (size_t)p0 * c, 0, (size_t)np * c, st));
    BowBatch r = bb;
    r.kp += (size_t)p0 * c * 7; r.desc += (size_t)p0 * c * 32;
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
    if (!lanes && S > 1 && frames == n_frames && per >= 2) {
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

// internal: device view for the batched BoW path (extractor.hip)
extern "C" int orbfe_vocabulary_device_(orbfe_vocabulary* v, VocabDevice* out, int* L, int* device) {
  if (!v) return vfail(ORBFE_ERR_INVALID, "NULL vocabulary");
  *out = v->d;
  *L = v->L;
  *device = v->device;
  return ORBFE_OK;
}
