// matcher.hip -- C-ABI entry points of the Hamming matchers and the stereo matcher
// (include/orbfe.h, "Matcher").  Stateless and re-entrant: every thread owns a private
// device arena + stream per device (the reference constructs stack-local ORBmatcher objects
// on three threads concurrently, src/LocalMapping.cc:261, src/LoopClosing.cc:294).
#include <hip/hip_runtime.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "kernels.h"
#include "match_kernels.h"

using namespace orbfe;

static thread_local std::string g_merr = "";
extern "C" const char* orbfe_last_error(void);  // extractor.hip owns the generic one
static int mfail(int code, const std::string& msg);

namespace {

struct Arena {
  int device = -1;
  hipStream_t stream = nullptr;
  uint8_t* base = nullptr;
  size_t cap = 0, used = 0;
  // pinned mirror of the uploaded part of the arena: up() only copies into it, flush() sends the whole
  // dirty range in ONE host-to-device copy before the first kernel of the call
  uint8_t* hmirror = nullptr;
  size_t hcap = 0, dirtyLo = 0, dirtyHi = 0;
  ~Arena() {
    if (device >= 0) {
      (void)hipSetDevice(device);
      if (hmirror) (void)hipHostFree(hmirror);
      if (base) (void)hipFree(base);
      if (stream) (void)hipStreamDestroy(stream);
    }
  }
};
thread_local std::map<int, Arena> t_arenas;

// Reserve `bytes` up front (sum of all buffers of a call), then carve.
hipError_t arena_begin(int device, size_t bytes, Arena** out) {
  hipError_t err = hipSetDevice(device);
  if (err != hipSuccess) return err;
  Arena& a = t_arenas[device];
  if (a.device < 0) {
    a.device = device;
    err = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking);
    if (err != hipSuccess) { a.device = -1; return err; }
  }
  if (bytes > a.cap) {
    if (a.base) (void)hipFree(a.base);
    a.base = nullptr;
    a.cap = 0;
    size_t want = bytes + bytes / 2 + (1u << 20);
    err = hipMalloc((void**)&a.base, want);
    if (err != hipSuccess) return err;
    a.cap = want;
  }
  a.used = 0;
  a.dirtyLo = a.dirtyHi = 0;
  *out = &a;
  return hipSuccess;
}
template <typename T>
T* carve(Arena* a, size_t n) {
  size_t off = (a->used + 255) & ~(size_t)255;
  a->used = off + n * sizeof(T);
  return reinterpret_cast<T*>(a->base + off);
}
inline size_t pad(size_t bytes) { return ((bytes + 255) & ~(size_t)255) + 256; }

// the pinned mirror covers arena offsets [0, upto); what earlier up() calls of this call staged is kept
hipError_t grow_mirror(Arena* a, size_t upto) {
  if (upto <= a->hcap) return hipSuccess;
  const size_t want = upto + upto / 2 + (1u << 16);
  uint8_t* nh = nullptr;
  hipError_t e = hipHostMalloc((void**)&nh, want, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  if (a->hmirror) {
    if (a->dirtyHi > a->dirtyLo) std::memcpy(nh + a->dirtyLo, a->hmirror + a->dirtyLo, a->dirtyHi - a->dirtyLo);
    (void)hipHostFree(a->hmirror);
  }
  a->hmirror = nh;
  a->hcap = want;
  return hipSuccess;
}
inline void mark_dirty(Arena* a, size_t off, size_t bytes) {
  if (a->dirtyHi == a->dirtyLo) { a->dirtyLo = off; a->dirtyHi = off + bytes; }
  else {
    if (off < a->dirtyLo) a->dirtyLo = off;
    if (off + bytes > a->dirtyHi) a->dirtyHi = off + bytes;
  }
}
template <typename T>
hipError_t up(Arena* a, T** d, const T* h, size_t n) {
  *d = carve<T>(a, n ? n : 1);
  if (n == 0) return hipSuccess;
  const size_t off = (size_t)(reinterpret_cast<uint8_t*>(*d) - a->base), bytes = n * sizeof(T);
  hipError_t e = grow_mirror(a, off + bytes);
  if (e != hipSuccess) return e;
  std::memcpy(a->hmirror + off, h, bytes);
  mark_dirty(a, off, bytes);
  return hipSuccess;
}
// a device array of n elements whose bytes all start as `byteValue`: filled in the mirror, so it travels with the one
// host-to-device copy of the call instead of costing a fill kernel of its own
template <typename T>
hipError_t up_fill(Arena* a, T** d, size_t n, int byteValue) {
  *d = carve<T>(a, n ? n : 1);
  const size_t off = (size_t)(reinterpret_cast<uint8_t*>(*d) - a->base), bytes = (n ? n : 1) * sizeof(T);
  hipError_t e = grow_mirror(a, off + bytes);
  if (e != hipSuccess) return e;
  std::memset(a->hmirror + off, byteValue, bytes);
  mark_dirty(a, off, bytes);
  return hipSuccess;
}
// results: ONE device-to-host copy of the arena range [first, last) into the pinned mirror (same offsets), to be read
// through mirror_of() after the stream is synchronised -- instead of one pageable copy per output array
hipError_t down_range(Arena* a, const void* first, const void* last) {
  const size_t lo = (size_t)(reinterpret_cast<const uint8_t*>(first) - a->base);
  const size_t hi = (size_t)(reinterpret_cast<const uint8_t*>(last) - a->base);
  hipError_t e = grow_mirror(a, hi);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(a->hmirror + lo, a->base + lo, hi - lo, hipMemcpyDeviceToHost, a->stream);
}
template <typename T>
const T* mirror_of(Arena* a, const T* d) {
  return reinterpret_cast<const T*>(a->hmirror + (reinterpret_cast<const uint8_t*>(d) - a->base));
}
// one H2D copy for everything up() staged since arena_begin(); call before the first kernel launch
hipError_t flush(Arena* a) {
  if (a->dirtyHi == a->dirtyLo) return hipSuccess;
  hipError_t e = hipMemcpyAsync(a->base + a->dirtyLo, a->hmirror + a->dirtyLo, a->dirtyHi - a->dirtyLo,
                                hipMemcpyHostToDevice, a->stream);
  a->dirtyLo = a->dirtyHi = 0;
  return e;
}

// merge-walk of the two ascending node-id lists (the std::map iteration + lower_bound of
// src/ORBmatcher.cc:211-300)
void shared_nodes(const orbfe_featvec* f1, const orbfe_featvec* f2, std::vector<NodePair>* out) {
  int a = 0, b = 0;
  while (a < f1->n_nodes && b < f2->n_nodes) {
    const uint32_t ia = f1->node_ids[a], ib = f2->node_ids[b];
    if (ia == ib) {
      out->push_back(NodePair{f1->offsets[a], f1->offsets[a + 1] - f1->offsets[a], f2->offsets[b],
                              f2->offsets[b + 1] - f2->offsets[b]});
      a++;
      b++;
    } else if (ia < ib) a++;
    else b++;
  }
}

// pinned staging of the calling thread (defined with the window searches below)
hipError_t staging_reserve_(size_t bytes);
uint8_t* staging_ptr_();
hipError_t staging_mark_pending_(hipStream_t s);

bool featvec_ok(const orbfe_featvec* f, int n) {
  if (!f || f->n_nodes < 0) return false;
  if (f->n_nodes == 0) return true;
  if (!f->node_ids || !f->offsets || !f->indices) return false;
  if (f->offsets[0] != 0) return false;
  for (int i = 0; i < f->n_nodes; i++) {
    if (f->offsets[i + 1] < f->offsets[i]) return false;
    if (i > 0 && f->node_ids[i] <= f->node_ids[i - 1]) return false;
  }
  const int tot = f->offsets[f->n_nodes];
  for (int i = 0; i < tot; i++)
    if (f->indices[i] >= (uint32_t)n) return false;
  return true;
}

}  // namespace

static int mfail(int code, const std::string& msg) {
  // route through the shared thread-local error text
  extern int orbfe_set_error_(int, const char*);
  return orbfe_set_error_(code, msg.c_str());
}
#define MHIP(expr)                                                                       \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return mfail(ORBFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" int orbfe_descriptor_distance(int device, const uint8_t* a, const uint8_t* b, int n, int32_t* out) {
  if (n < 0 || (n > 0 && (!a || !b || !out))) return mfail(ORBFE_ERR_INVALID, "descriptor_distance: bad argument");
  if (n == 0) return ORBFE_OK;
  Arena* ar;
  MHIP(arena_begin(device, 2 * pad((size_t)n * 32) + pad((size_t)n * 4), &ar));
  uint8_t *da, *db;
  MHIP(up(ar, &da, a, (size_t)n * 32));
  MHIP(up(ar, &db, b, (size_t)n * 32));
  int32_t* dout = carve<int32_t>(ar, n);
  MHIP(flush(ar));
  launch_hamming_pairs(ar->stream, da, db, n, dout);
  MHIP(hipGetLastError());
  MHIP(hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ar->stream));
  MHIP(hipStreamSynchronize(ar->stream));
  return ORBFE_OK;
}

extern "C" int orbfe_hamming_matrix(int device, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int32_t* out) {
  if (n1 < 0 || n2 < 0 || ((n1 > 0 && n2 > 0) && (!d1 || !d2 || !out)))
    return mfail(ORBFE_ERR_INVALID, "hamming_matrix: bad argument");
  if (n1 == 0 || n2 == 0) return ORBFE_OK;
  Arena* ar;
  MHIP(arena_begin(device, pad((size_t)n1 * 32) + pad((size_t)n2 * 32) + pad((size_t)n1 * n2 * 4), &ar));
  uint8_t *da, *db;
  MHIP(up(ar, &da, d1, (size_t)n1 * 32));
  MHIP(up(ar, &db, d2, (size_t)n2 * 32));
  int32_t* dout = carve<int32_t>(ar, (size_t)n1 * n2);
  MHIP(flush(ar));
  launch_hamming_matrix(ar->stream, da, n1, db, n2, dout);
  MHIP(hipGetLastError());
  MHIP(hipMemcpyAsync(out, dout, (size_t)n1 * n2 * 4, hipMemcpyDeviceToHost, ar->stream));
  MHIP(hipStreamSynchronize(ar->stream));
  return ORBFE_OK;
}

static int bow_common(int device, const uint8_t* desc1, const uint8_t* has_mp1, const float* angle1, int n1,
                      const orbfe_featvec* fv1, const uint8_t* desc2, const uint8_t* has_mp2, const float* angle2,
                      int n2, const orbfe_featvec* fv2, float nnratio, int check_ori, int kfkf, int32_t* match) {
  if (n1 < 0 || n2 < 0 || !match) return mfail(ORBFE_ERR_INVALID, "search_by_bow: bad argument");
  const int nOut = kfkf ? n1 : n2;
  for (int i = 0; i < nOut; i++) match[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  if (!desc1 || !has_mp1 || !angle1 || !desc2 || !angle2 || (kfkf && !has_mp2))
    return mfail(ORBFE_ERR_INVALID, "search_by_bow: NULL input");
  if (!featvec_ok(fv1, n1) || !featvec_ok(fv2, n2)) return mfail(ORBFE_ERR_INVALID, "search_by_bow: malformed FeatureVector");
  std::vector<NodePair> pairs;
  shared_nodes(fv1, fv2, &pairs);
  if (pairs.empty()) return 0;
  int maxCnt2 = 0;
  for (const NodePair& p : pairs) maxCnt2 = p.cnt2 > maxCnt2 ? p.cnt2 : maxCnt2;
  if (maxCnt2 > 65535) return mfail(ORBFE_ERR_INVALID, "search_by_bow: more than 65535 features in one node");
  const size_t t1 = fv1->offsets[fv1->n_nodes], t2 = fv2->offsets[fv2->n_nodes];
  Arena* ar;
  MHIP(arena_begin(device, pad(pairs.size() * sizeof(NodePair)) + pad((size_t)n1 * 32) + pad((size_t)n2 * 32) + 2 * pad(n1) +
                               2 * pad(n2) + 2 * pad((size_t)n1 * 4) + 2 * pad((size_t)n2 * 4) + pad(t1 * 4) + pad(t2 * 4) + 4096, &ar));
  BowArgs a = {};
  NodePair* dp;
  MHIP(up(ar, &dp, pairs.data(), pairs.size()));
  uint8_t *dd1, *dd2, *dm1, *dm2 = nullptr;
  float *da1, *da2;
  uint32_t *di1, *di2;
  MHIP(up(ar, &dd1, desc1, (size_t)n1 * 32));
  MHIP(up(ar, &dd2, desc2, (size_t)n2 * 32));
  MHIP(up(ar, &dm1, has_mp1, (size_t)n1));
  if (kfkf) MHIP(up(ar, &dm2, has_mp2, (size_t)n2));
  MHIP(up(ar, &da1, angle1, (size_t)n1));
  MHIP(up(ar, &da2, angle2, (size_t)n2));
  MHIP(up(ar, &di1, fv1->indices, t1));
  MHIP(up(ar, &di2, fv2->indices, t2));
  int32_t* dmatch;
  int8_t* dbin;
  MHIP(up_fill(ar, &dmatch, (size_t)nOut, 0xff));
  int32_t* dcount = carve<int32_t>(ar, 1);
  MHIP(up_fill(ar, &dbin, (size_t)nOut, 0));
  a.pairs = dp; a.desc1 = dd1; a.hasMp1 = dm1; a.angle1 = da1; a.indices1 = di1;
  a.desc2 = dd2; a.hasMp2 = dm2; a.angle2 = da2; a.indices2 = di2;
  a.angleStride = 1;
  a.nnratio = nnratio; a.strictLow = kfkf; a.match = dmatch; a.bin = dbin;
  MHIP(flush(ar));
  launch_search_by_bow(ar->stream, a, (int)pairs.size(), maxCnt2);
  launch_rot_prune(ar->stream, dmatch, dbin, nOut, check_ori, dcount);
  MHIP(hipGetLastError());
  MHIP(down_range(ar, dmatch, dcount + 1));  // match[nOut] and the count, contiguous in the arena
  MHIP(hipStreamSynchronize(ar->stream));
  std::memcpy(match, mirror_of(ar, dmatch), (size_t)nOut * 4);
  return *mirror_of(ar, dcount);
}

extern "C" int orbfe_search_by_bow(int device, const uint8_t* desc1, const uint8_t* has_mp1, const float* angle1,
                                   int n1, const orbfe_featvec* fv1, const uint8_t* desc2, const float* angle2,
                                   int n2, const orbfe_featvec* fv2, float nnratio, int check_orientation,
                                   int32_t* match_f) {
  return bow_common(device, desc1, has_mp1, angle1, n1, fv1, desc2, nullptr, angle2, n2, fv2, nnratio,
                    check_orientation, 0, match_f);
}
extern "C" int orbfe_search_by_bow_kf(int device, const uint8_t* desc1, const uint8_t* has_mp1, const float* angle1,
                                      int n1, const orbfe_featvec* fv1, const uint8_t* desc2,
                                      const uint8_t* has_mp2, const float* angle2, int n2,
                                      const orbfe_featvec* fv2, float nnratio, int check_orientation,
                                      int32_t* match12) {
  return bow_common(device, desc1, has_mp1, angle1, n1, fv1, desc2, has_mp2, angle2, n2, fv2, nnratio,
                    check_orientation, 1, match12);
}

extern "C" int orbfe_search_for_triangulation(int device, const uint8_t* desc1, const uint8_t* has_mp1,
                                              const float* x1, const float* y1, const float* angle1,
                                              const uint8_t* stereo1, int n1, const orbfe_featvec* fv1,
                                              const uint8_t* desc2, const uint8_t* has_mp2, const float* x2,
                                              const float* y2, const float* angle2, const int32_t* octave2,
                                              const uint8_t* stereo2, int n2, const orbfe_featvec* fv2,
                                              const float* F12, float ex, float ey, const float* scale_factors2,
                                              const float* level_sigma2_2, int n_levels2, int only_stereo,
                                              int check_orientation, int32_t* match12) {
  if (n1 < 0 || n2 < 0 || !match12) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: bad argument");
  for (int i = 0; i < n1; i++) match12[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  if (!desc1 || !has_mp1 || !x1 || !y1 || !angle1 || !stereo1 || !desc2 || !has_mp2 || !x2 || !y2 || !angle2 ||
      !octave2 || !stereo2 || !F12 || !scale_factors2 || !level_sigma2_2 || n_levels2 <= 0 || n_levels2 > ORBFE_MAX_LEVELS)
    return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: NULL input");
  if (!featvec_ok(fv1, n1) || !featvec_ok(fv2, n2)) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: malformed FeatureVector");
  for (int i = 0; i < n2; i++)
    if (octave2[i] < 0 || octave2[i] >= n_levels2) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: octave out of range");
  std::vector<NodePair> pairs;
  shared_nodes(fv1, fv2, &pairs);
  std::vector<TriQuery> queries;
  for (const NodePair& p : pairs) {
    if (p.cnt2 > 65535) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: more than 65535 features in one node");
    for (int i = 0; i < p.cnt1; i++) {
      const uint32_t idx1 = fv1->indices[p.off1 + i];
      if (has_mp1[idx1]) continue;                    // :800-803
      if (only_stereo && !stereo1[idx1]) continue;    // :807-809
      queries.push_back(TriQuery{idx1, p.off2, p.cnt2});
    }
  }
  if (queries.empty()) return 0;
  const size_t t2 = fv2->offsets[fv2->n_nodes];
  Arena* ar;
  MHIP(arena_begin(device, pad(queries.size() * sizeof(TriQuery)) + pad((size_t)n1 * 32) + pad((size_t)n2 * 32) +
                               4 * pad((size_t)n1 * 4) + 6 * pad((size_t)n2 * 4) + 2 * pad(n1) + 2 * pad(n2) + pad(t2 * 4) + 8192, &ar));
  TriArgs a = {};
  TriQuery* dq;
  MHIP(up(ar, &dq, queries.data(), queries.size()));
  uint8_t *dd1, *dd2, *ds1, *ds2, *dm2;
  float *dx1, *dy1, *da1, *dx2, *dy2, *da2, *dF, *dsf, *dsg;
  int32_t* doc2;
  uint32_t* di2;
  MHIP(up(ar, &dd1, desc1, (size_t)n1 * 32));
  MHIP(up(ar, &dd2, desc2, (size_t)n2 * 32));
  MHIP(up(ar, &ds1, stereo1, (size_t)n1));
  MHIP(up(ar, &ds2, stereo2, (size_t)n2));
  MHIP(up(ar, &dm2, has_mp2, (size_t)n2));
  MHIP(up(ar, &dx1, x1, (size_t)n1));
  MHIP(up(ar, &dy1, y1, (size_t)n1));
  MHIP(up(ar, &da1, angle1, (size_t)n1));
  MHIP(up(ar, &dx2, x2, (size_t)n2));
  MHIP(up(ar, &dy2, y2, (size_t)n2));
  MHIP(up(ar, &da2, angle2, (size_t)n2));
  MHIP(up(ar, &doc2, octave2, (size_t)n2));
  MHIP(up(ar, &di2, fv2->indices, t2));
  MHIP(up(ar, &dF, F12, (size_t)9));
  MHIP(up(ar, &dsf, scale_factors2, (size_t)n_levels2));
  MHIP(up(ar, &dsg, level_sigma2_2, (size_t)n_levels2));
  int32_t* dmatch;
  int8_t* dbin;
  MHIP(up_fill(ar, &dmatch, (size_t)n1, 0xff));
  int32_t* dcount = carve<int32_t>(ar, 1);
  MHIP(up_fill(ar, &dbin, (size_t)n1, 0));
  a.queries = dq; a.nQueries = (int)queries.size();
  a.desc1 = dd1; a.x1 = dx1; a.y1 = dy1; a.angle1 = da1; a.stereo1 = ds1;
  a.desc2 = dd2; a.hasMp2 = dm2; a.x2 = dx2; a.y2 = dy2; a.angle2 = da2; a.octave2 = doc2; a.stereo2 = ds2;
  a.indices2 = di2; a.F12 = dF; a.ex = ex; a.ey = ey; a.scaleFactors2 = dsf; a.levelSigma2_2 = dsg;
  a.onlyStereo = only_stereo; a.match = dmatch; a.bin = dbin;
  MHIP(flush(ar));
  launch_search_triangulation(ar->stream, a);
  launch_rot_prune(ar->stream, dmatch, dbin, n1, check_orientation, dcount);
  MHIP(hipGetLastError());
  MHIP(down_range(ar, dmatch, dcount + 1));  // match12[n1] and the count, contiguous in the arena
  MHIP(hipStreamSynchronize(ar->stream));
  std::memcpy(match12, mirror_of(ar, dmatch), (size_t)n1 * 4);
  return *mirror_of(ar, dcount);
}

// ---------------------------------------------------------------------------------------------
// Device-resident Frame / KeyFrame operands (round 3).  The live system matches one key frame against 10-20
// neighbours (LocalMapping::CreateNewMapPoints / SearchInNeighbors, src/LocalMapping.cc:256-315, 517-573) and one
// frame against several candidates (Tracking::Relocalization, src/Tracking.cc:1478-1498): with host-pointer operands
// every call uploaded both frames' descriptors again.  orbfe_frame_upload moves what a frame contributes to ANY search
// -- keypoint arrays, descriptors, the 64 x 48 grid (built once), the FeatureVector's index list -- to the device once;
// the handle is immutable afterwards, so any thread may use it concurrently.
// ---------------------------------------------------------------------------------------------
struct orbfe_frame {
  int device = 0, n = 0;
  std::vector<float> hx, hy, hangle, hur;   // host copies: the claim loops and chi-square gates read them
  std::vector<int32_t> hoct;
  std::vector<uint8_t> hstereo;             // mvuRight[i] >= 0
  std::vector<uint32_t> nodeIds;            // FeatureVector (host side of the merge-walk)
  std::vector<int32_t> offsets;
  std::vector<uint32_t> hindices;
  orbfe_featvec fv = {};
  bool haveFv = false;
  uint8_t* slab = nullptr;                  // one device allocation (from the slab pool)
  size_t slabCap = 0;
  hipEvent_t ready = nullptr;               // recorded behind the upload + grid build; consumers on other streams wait for it
  mutable std::atomic<bool> settled{false}; // a consumer has synchronised behind `ready`: no further waits needed
  float *dx = nullptr, *dy = nullptr, *dangle = nullptr, *dur = nullptr;
  int32_t* doct = nullptr;
  uint8_t *ddesc = nullptr, *dstereo = nullptr;
  uint32_t *dkey = nullptr, *dindices = nullptr;
  int32_t* dcell = nullptr;
  std::vector<uint8_t> hdesc;               // (the host-pointer fallbacks of a view need it)
  orbfe_frame_view view = {};               // canonical view: host copies + resident = this
};

namespace {
inline const orbfe_frame_view* canon(const orbfe_frame_view* f) { return (f && f->resident) ? &f->resident->view : f; }
}

namespace {
// Slabs and events of released frames are kept for the next upload: hipMalloc / hipFree cost tens of microseconds and
// hipFree waits for the whole device -- in a live system every key-frame insertion would stall the extractor's streams.
struct FramePool {
  std::mutex m;
  struct Slab { int device; uint8_t* p; size_t cap; };
  std::vector<Slab> slabs;
  std::vector<std::pair<int, hipEvent_t>> events;
  static constexpr size_t kKeep = 64;
  ~FramePool() {}  // (process exit: the runtime reclaims device memory; no HIP calls from static destructors)
};
FramePool g_framePool;

hipError_t slab_get(int device, size_t bytes, uint8_t** p, size_t* cap) {
  {
    std::lock_guard<std::mutex> lk(g_framePool.m);
    auto& v = g_framePool.slabs;
    int best = -1;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].device == device && v[i].cap >= bytes && v[i].cap <= 4 * bytes + (1u << 16) && (best < 0 || v[i].cap < v[(size_t)best].cap))
        best = (int)i;
    if (best >= 0) {
      *p = v[(size_t)best].p; *cap = v[(size_t)best].cap;
      v.erase(v.begin() + best);
      return hipSuccess;
    }
  }
  const size_t want = (bytes + (1u << 16) - 1) & ~(size_t)((1u << 16) - 1);  // 64 KB classes: frames of similar size share slabs
  hipError_t e = hipMalloc((void**)p, want);
  if (e == hipSuccess) *cap = want;
  return e;
}
void slab_put(int device, uint8_t* p, size_t cap) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_framePool.m);
    if (g_framePool.slabs.size() < FramePool::kKeep) { g_framePool.slabs.push_back({device, p, cap}); return; }
  }
  (void)hipFree(p);
}
hipError_t event_get(int device, hipEvent_t* e) {
  {
    std::lock_guard<std::mutex> lk(g_framePool.m);
    auto& v = g_framePool.events;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].first == device) { *e = v[i].second; v.erase(v.begin() + (long)i); return hipSuccess; }
  }
  return hipEventCreateWithFlags(e, hipEventDisableTiming);
}
void event_put(int device, hipEvent_t e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(g_framePool.m);
    if (g_framePool.events.size() < 4 * FramePool::kKeep) { g_framePool.events.push_back({device, e}); return; }
  }
  (void)hipEventDestroy(e);
}

// A search that reads a resident frame on ITS stream: ordered behind the frame's upload + grid build (which ran on the
// uploading thread's stream) by the frame's event -- the upload itself does not wait for the device.  Once any consumer
// has synchronised behind the event the frame is settled and nothing waits any more.
thread_local std::vector<const orbfe_frame*> t_unsettled;
hipError_t frame_use(Arena* ar, const orbfe_frame* f) {
  if (!f || f->settled.load(std::memory_order_acquire)) return hipSuccess;
  hipError_t e = hipStreamWaitEvent(ar->stream, f->ready, 0);
  if (e == hipSuccess) t_unsettled.push_back(f);
  return e;
}
void frames_settle() {  // call after the stream of the call has been synchronised
  for (const orbfe_frame* f : t_unsettled) f->settled.store(true, std::memory_order_release);
  t_unsettled.clear();
}
// an entry point that returns early (a HIP error between frame_use and its synchronisation) must not leave frames on the
// list: they could be released before this thread's next call settles -- and writes to -- them
struct UnsettledScope {
  UnsettledScope() { t_unsettled.clear(); }
  ~UnsettledScope() { t_unsettled.clear(); }
};
}  // namespace

extern "C" void orbfe_frame_release(orbfe_frame* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  // a frame whose own upload may still be in flight (released before any search used it): its slab must not be handed
  // to the next upload until then
  if (f->ready && !f->settled.load(std::memory_order_acquire)) (void)hipEventSynchronize(f->ready);
  for (size_t i = 0; i < t_unsettled.size();)
    if (t_unsettled[i] == f) t_unsettled.erase(t_unsettled.begin() + (long)i); else i++;
  slab_put(f->device, f->slab, f->slabCap);
  event_put(f->device, f->ready);
  delete f;
}

extern "C" const orbfe_frame_view* orbfe_frame_get_view(const orbfe_frame* f) { return f ? &f->view : nullptr; }

namespace {
// host side of a resident frame: copies of what the claim loops / gates read, the canonical view, the slab layout
struct FrameLayout { size_t oX, oY, oA, oU, oO, oK, oC, oI, oD, oS, total; };
int frame_host_init(const char* who, int device, const orbfe_frame_view* v, const orbfe_featvec* fv, orbfe_frame** outF,
                    FrameLayout* L, size_t* nIdxOut) {
  if (!v || v->n < 0 || v->n > GRID_MAX_FEATURES || !(v->max_x > v->min_x) || !(v->max_y > v->min_y) ||
      (v->n > 0 && (!v->x || !v->y || !v->octave || !v->desc)))
    return mfail(ORBFE_ERR_INVALID, std::string(who) + ": bad frame view (x, y, octave, desc and the image bounds are required)");
  const int n = v->n;
  if (fv && !featvec_ok(fv, n)) return mfail(ORBFE_ERR_INVALID, std::string(who) + ": malformed FeatureVector");
  orbfe_frame* f = new (std::nothrow) orbfe_frame();
  if (!f) return mfail(ORBFE_ERR_NOMEM, "out of memory");
  f->device = device; f->n = n;
  f->hx.assign(v->x, v->x + n); f->hy.assign(v->y, v->y + n); f->hoct.assign(v->octave, v->octave + n);
  f->hdesc.assign(v->desc, v->desc + (size_t)n * 32);
  if (v->angle) f->hangle.assign(v->angle, v->angle + n);
  if (v->u_right) f->hur.assign(v->u_right, v->u_right + n);
  f->hstereo.assign((size_t)n, 0);
  if (v->u_right) for (int i = 0; i < n; i++) f->hstereo[i] = v->u_right[i] >= 0 ? 1 : 0;
  size_t nIdx = 0;
  if (fv) {
    f->haveFv = true;
    if (fv->n_nodes > 0) {  // (an EMPTY FeatureVector may come with NULL arrays: nothing is read from them)
      f->nodeIds.assign(fv->node_ids, fv->node_ids + fv->n_nodes);
      f->offsets.assign(fv->offsets, fv->offsets + fv->n_nodes + 1);
      nIdx = (size_t)fv->offsets[fv->n_nodes];
      f->hindices.assign(fv->indices, fv->indices + nIdx);
    } else {
      f->offsets.assign(1, 0);
    }
    f->fv.n_nodes = fv->n_nodes; f->fv.node_ids = f->nodeIds.data(); f->fv.offsets = f->offsets.data(); f->fv.indices = f->hindices.data();
  }
  orbfe_frame_view& c = f->view;
  c.n = n; c.x = f->hx.data(); c.y = f->hy.data(); c.octave = f->hoct.data();
  c.angle = v->angle ? f->hangle.data() : nullptr;
  c.u_right = v->u_right ? f->hur.data() : nullptr;
  c.desc = f->hdesc.data();
  c.min_x = v->min_x; c.max_x = v->max_x; c.min_y = v->min_y; c.max_y = v->max_y;
  c.resident = f;
  // one slab: x y angle u_right | octave | key | cell | indices | desc | stereo.  The index list gets room for one index per
  // feature even when no FeatureVector comes with the upload: orbfe_frame_set_featvec may attach it later (Frame::ComputeBoW
  // runs after the constructor, src/Tracking.cc:836-843)
  const size_t N = (size_t)(n ? n : 1);
  size_t off = 0;
  auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  // what always comes from the host first (mvuRight, stereo flags, index list, then the positions), what the device can
  // supply behind it (angles, octaves, descriptors), what the grid build writes last: an upload is ONE copy of [0, oK), a
  // frame built from the extractor's records ONE copy of [0, oX) (or [0, oA) with caller-undistorted positions)
  L->oU = place(N * 4); L->oS = place(N); L->oI = place((nIdx > N ? nIdx : N) * 4); L->oX = place(N * 4); L->oY = place(N * 4);
  L->oA = place(N * 4); L->oO = place(N * 4); L->oD = place(N * 32); L->oK = place(N * 4); L->oC = place(3073 * 4);
  L->total = off;
  hipError_t err = hipSetDevice(device);
  if (err == hipSuccess) err = slab_get(device, off, &f->slab, &f->slabCap);
  if (err == hipSuccess) err = event_get(device, &f->ready);
  if (err != hipSuccess) {
    const int code = err == hipErrorOutOfMemory ? ORBFE_ERR_NOMEM : ORBFE_ERR_HIP;
    orbfe_frame_release(f);
    return mfail(code, std::string(who) + ": " + hipGetErrorString(err));
  }
  f->settled.store(false);
  uint8_t* b = f->slab;
  f->dx = (float*)(b + L->oX); f->dy = (float*)(b + L->oY); f->dangle = (float*)(b + L->oA); f->dur = (float*)(b + L->oU);
  f->doct = (int32_t*)(b + L->oO); f->dkey = (uint32_t*)(b + L->oK); f->dcell = (int32_t*)(b + L->oC); f->dindices = (uint32_t*)(b + L->oI);
  f->ddesc = b + L->oD; f->dstereo = b + L->oS;
  *outF = f;
  *nIdxOut = nIdx;
  return ORBFE_OK;
}

// grid build behind whatever filled the slab, then the ready event: NO host wait (frame_use orders the consumers)
hipError_t frame_finish(Arena* ar, orbfe_frame* f, const orbfe_frame_view* v) {
  GridFrame g{};
  g.x = f->dx; g.y = f->dy; g.octave = f->doct; g.uRight = v->u_right ? f->dur : nullptr; g.desc = f->ddesc; g.n = f->n;
  g.minX = v->min_x; g.minY = v->min_y;
  g.wInv = 64.0f / (v->max_x - v->min_x);
  g.hInv = 48.0f / (v->max_y - v->min_y);
  launch_grid_build(ar->stream, g, f->dkey, f->dcell);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipEventRecord(f->ready, ar->stream);
  return err;
}
}  // namespace

extern "C" int orbfe_frame_upload(int device, const orbfe_frame_view* v, const orbfe_featvec* fv, orbfe_frame** out) {
  if (!out) return mfail(ORBFE_ERR_INVALID, "frame_upload: NULL argument");
  *out = nullptr;
  orbfe_frame* f = nullptr;
  FrameLayout L;
  size_t nIdx = 0;
  int rc = frame_host_init("frame_upload", device, v, fv, &f, &L, &nIdx);
  if (rc != ORBFE_OK) return rc;
  const int n = f->n;
  // only what the searches read goes up: the key / cell arrays behind it are written by the grid build
  const size_t upBytes = L.oK;
  Arena* ar;
  // staged through the thread's pinned mirror: one copy up, then the grid build (Frame::AssignFeaturesToGrid, once)
  hipError_t err = arena_begin(device, 1024, &ar);
  if (err == hipSuccess) err = staging_reserve_(upBytes);
  if (err == hipSuccess) {
    uint8_t* h = staging_ptr_();
    if (n) {
      std::memcpy(h + L.oX, f->hx.data(), (size_t)n * 4); std::memcpy(h + L.oY, f->hy.data(), (size_t)n * 4);
      if (v->angle) std::memcpy(h + L.oA, f->hangle.data(), (size_t)n * 4); else std::memset(h + L.oA, 0, (size_t)n * 4);
      if (v->u_right) std::memcpy(h + L.oU, f->hur.data(), (size_t)n * 4); else std::memset(h + L.oU, 0, (size_t)n * 4);
      std::memcpy(h + L.oO, f->hoct.data(), (size_t)n * 4);
      std::memcpy(h + L.oD, f->hdesc.data(), (size_t)n * 32);
      std::memcpy(h + L.oS, f->hstereo.data(), (size_t)n);
    }
    if (nIdx) std::memcpy(h + L.oI, f->hindices.data(), nIdx * 4);
    // ONE copy (every further hipMemcpyAsync costs the host ~5 us)
    err = hipMemcpyAsync(f->slab, h, upBytes, hipMemcpyHostToDevice, ar->stream);
    if (err == hipSuccess) err = staging_mark_pending_(ar->stream);  // the next use of the staging buffer waits for these copies
  }
  if (err == hipSuccess) err = frame_finish(ar, f, v);
  if (err != hipSuccess) { orbfe_frame_release(f); return mfail(ORBFE_ERR_HIP, std::string("frame_upload: ") + hipGetErrorString(err)); }
  *out = f;
  return ORBFE_OK;
}

// Frame::Frame (src/Frame.cc:61-117) is extract -> undistort -> stereo -> grid: the keypoint records and descriptors the
// extractor produced are still in HBM when the Frame is built.  orbfe_frame_from_device makes the resident operands from
// THOSE (28-byte records -> x / y / angle / octave arrays, descriptors device to device, grid built on the device): of
// the frame's 60 bytes per keypoint only mvuRight (and, with ORBFE_FRAME_XY_FROM_VIEW, the undistorted positions) travel
// over PCIe.  `view` holds the host arrays the claim loops read (what orbfe_extract returned to the caller, after its own
// UndistortKeyPoints); view->n records are taken.
extern "C" int orbfe_frame_from_device(int device, const orbfe_keypoint* d_keypoints, const uint8_t* d_descriptors,
                                       const orbfe_frame_view* view, const orbfe_featvec* fv, int flags, orbfe_frame** out) {
  if (!out) return mfail(ORBFE_ERR_INVALID, "frame_from_device: NULL argument");
  *out = nullptr;
  if (view && view->n > 0 && (!d_keypoints || !d_descriptors)) return mfail(ORBFE_ERR_INVALID, "frame_from_device: NULL device arrays");
  orbfe_frame* f = nullptr;
  FrameLayout L;
  size_t nIdx = 0;
  int rc = frame_host_init("frame_from_device", device, view, fv, &f, &L, &nIdx);
  if (rc != ORBFE_OK) return rc;
  const int n = f->n;
  const bool xyFromView = (flags & ORBFE_FRAME_XY_FROM_VIEW) != 0;
  Arena* ar;
  hipError_t err = arena_begin(device, 1024, &ar);
  // host part: mvuRight + stereo flags + the FeatureVector's index list (+ positions), adjacent at the head of the slab:
  // ONE copy through the pinned staging
  const size_t hostBytes = xyFromView ? L.oA : L.oX;
  const bool anyHost = view->u_right || nIdx || xyFromView;
  if (err == hipSuccess && anyHost) err = staging_reserve_(hostBytes);
  if (err == hipSuccess && n && anyHost) {
    uint8_t* h = staging_ptr_();
    if (view->u_right) { std::memcpy(h + L.oU, f->hur.data(), (size_t)n * 4); std::memcpy(h + L.oS, f->hstereo.data(), (size_t)n); }
    else { std::memset(h + L.oU, 0, (size_t)n * 4); std::memset(h + L.oS, 0, (size_t)n); }
    if (nIdx) std::memcpy(h + L.oI, f->hindices.data(), nIdx * 4);
    if (xyFromView) { std::memcpy(h + L.oX, f->hx.data(), (size_t)n * 4); std::memcpy(h + L.oY, f->hy.data(), (size_t)n * 4); }
    err = hipMemcpyAsync(f->slab, h, hostBytes, hipMemcpyHostToDevice, ar->stream);
    if (err == hipSuccess) err = staging_mark_pending_(ar->stream);
  }
  if (err == hipSuccess && n) {
    launch_frame_from_records(ar->stream, reinterpret_cast<const float*>(d_keypoints), d_descriptors, n, xyFromView ? nullptr : f->dx,
                              xyFromView ? nullptr : f->dy, f->dangle, f->doct, f->ddesc, anyHost ? nullptr : f->dstereo);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = frame_finish(ar, f, view);
  if (err != hipSuccess) { orbfe_frame_release(f); return mfail(ORBFE_ERR_HIP, std::string("frame_from_device: ") + hipGetErrorString(err)); }
  *out = f;
  return ORBFE_OK;
}

// implemented in extractor.hip: device pointers of frame `frame` of the handle's own output block (the host-buffer calls)
extern "C" int orbfe_extractor_output_device_(orbfe_extractor* e, int frame, const orbfe_keypoint** d_kp, const uint8_t** d_desc,
                                              int* n, int* device);

extern "C" int orbfe_frame_from_extractor(orbfe_extractor* e, int frame, const orbfe_frame_view* view, const orbfe_featvec* fv,
                                          int flags, orbfe_frame** out) {
  if (!out) return mfail(ORBFE_ERR_INVALID, "frame_from_extractor: NULL argument");
  *out = nullptr;
  const orbfe_keypoint* dkp = nullptr;
  const uint8_t* ddesc = nullptr;
  int n = 0, device = 0;
  int rc = orbfe_extractor_output_device_(e, frame, &dkp, &ddesc, &n, &device);
  if (rc != ORBFE_OK) return rc;
  if (!view || view->n > n) return mfail(ORBFE_ERR_INVALID, "frame_from_extractor: the view holds more keypoints than the extractor produced for this frame");
  return orbfe_frame_from_device(device, dkp, ddesc, view, fv, flags, out);
}

// Frame::ComputeBoW runs after the constructor (src/Tracking.cc:836-843, src/Frame.cc:433-440): attach the FeatureVector
// to a frame that was made resident without one.  Call it before the handle is shared with other threads.
extern "C" int orbfe_frame_set_featvec(orbfe_frame* f, const orbfe_featvec* fv) {
  if (!f || !fv) return mfail(ORBFE_ERR_INVALID, "frame_set_featvec: NULL argument");
  if (!featvec_ok(fv, f->n)) return mfail(ORBFE_ERR_INVALID, "frame_set_featvec: malformed FeatureVector");
  const size_t nIdx = fv->n_nodes > 0 ? (size_t)fv->offsets[fv->n_nodes] : 0;
  if (nIdx > (size_t)(f->n ? f->n : 1)) return mfail(ORBFE_ERR_INVALID, "frame_set_featvec: more indices than features");
  if (fv->n_nodes > 0) {
    f->nodeIds.assign(fv->node_ids, fv->node_ids + fv->n_nodes);
    f->offsets.assign(fv->offsets, fv->offsets + fv->n_nodes + 1);
    f->hindices.assign(fv->indices, fv->indices + nIdx);
  } else {
    f->nodeIds.clear(); f->hindices.clear(); f->offsets.assign(1, 0);
  }
  f->fv.n_nodes = fv->n_nodes; f->fv.node_ids = f->nodeIds.data(); f->fv.offsets = f->offsets.data(); f->fv.indices = f->hindices.data();
  f->haveFv = true;
  if (nIdx == 0) return ORBFE_OK;
  Arena* ar;
  hipError_t err = arena_begin(f->device, 1024, &ar);
  if (err == hipSuccess) err = staging_reserve_(nIdx * 4);
  if (err == hipSuccess) {
    std::memcpy(staging_ptr_(), f->hindices.data(), nIdx * 4);
    err = hipMemcpyAsync(f->dindices, staging_ptr_(), nIdx * 4, hipMemcpyHostToDevice, ar->stream);
  }
  if (err == hipSuccess) err = hipStreamSynchronize(ar->stream);  // (the handle may be in use on other streams afterwards)
  if (err != hipSuccess) return mfail(ORBFE_ERR_HIP, std::string("frame_set_featvec: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

// ---- the FeatureVector searches on resident frames: per call only the shared-node list, the map-point masks and
//      the result arrays travel ----
static int bow_resident(const orbfe_frame* k1, const uint8_t* has_mp1, const orbfe_frame* k2, const uint8_t* has_mp2,
                        float nnratio, int check_ori, int kfkf, int32_t* match) {
  UnsettledScope unsettledScope;
  if (!k1 || !k2 || !match) return mfail(ORBFE_ERR_INVALID, "search_by_bow_resident: NULL argument");
  if (k1->device != k2->device) return mfail(ORBFE_ERR_INVALID, "search_by_bow_resident: frames on different devices");
  const int n1 = k1->n, n2 = k2->n, nOut = kfkf ? n1 : n2;
  for (int i = 0; i < nOut; i++) match[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  if (!k1->haveFv || !k2->haveFv || k1->hangle.empty() || k2->hangle.empty() || !has_mp1 || (kfkf && !has_mp2))
    return mfail(ORBFE_ERR_INVALID, "search_by_bow_resident: the frames were uploaded without FeatureVector / angles, or a mask is NULL");
  std::vector<NodePair> pairs;
  shared_nodes(&k1->fv, &k2->fv, &pairs);
  if (pairs.empty()) return 0;
  int maxCnt2 = 0;
  for (const NodePair& p : pairs) maxCnt2 = p.cnt2 > maxCnt2 ? p.cnt2 : maxCnt2;
  if (maxCnt2 > 65535) return mfail(ORBFE_ERR_INVALID, "search_by_bow: more than 65535 features in one node");
  Arena* ar;
  MHIP(arena_begin(k1->device, pad(pairs.size() * sizeof(NodePair)) + pad(n1) + pad(n2) + 2 * pad((size_t)nOut * 4) + 4096, &ar));
  MHIP(frame_use(ar, k1));
  MHIP(frame_use(ar, k2));
  NodePair* dp;
  uint8_t *dm1, *dm2 = nullptr;
  MHIP(up(ar, &dp, pairs.data(), pairs.size()));
  MHIP(up(ar, &dm1, has_mp1, (size_t)n1));
  if (kfkf) MHIP(up(ar, &dm2, has_mp2, (size_t)n2));
  int32_t* dmatch;
  int8_t* dbin;
  MHIP(up_fill(ar, &dmatch, (size_t)nOut, 0xff));
  int32_t* dcount = carve<int32_t>(ar, 1);
  MHIP(up_fill(ar, &dbin, (size_t)nOut, 0));
  BowArgs a = {};
  a.pairs = dp; a.desc1 = k1->ddesc; a.hasMp1 = dm1; a.angle1 = k1->dangle; a.indices1 = k1->dindices;
  a.desc2 = k2->ddesc; a.hasMp2 = dm2; a.angle2 = k2->dangle; a.indices2 = k2->dindices;
  a.angleStride = 1;
  a.nnratio = nnratio; a.strictLow = kfkf; a.match = dmatch; a.bin = dbin;
  MHIP(flush(ar));
  launch_search_by_bow(ar->stream, a, (int)pairs.size(), maxCnt2);
  launch_rot_prune(ar->stream, dmatch, dbin, nOut, check_ori, dcount);
  MHIP(hipGetLastError());
  MHIP(down_range(ar, dmatch, dcount + 1));
  MHIP(hipStreamSynchronize(ar->stream));
  frames_settle();
  std::memcpy(match, mirror_of(ar, dmatch), (size_t)nOut * 4);
  return *mirror_of(ar, dcount);
}

extern "C" int orbfe_search_by_bow_resident(const orbfe_frame* kf, const uint8_t* has_mp_kf, const orbfe_frame* f,
                                            float nnratio, int check_orientation, int32_t* match_f) {
  return bow_resident(kf, has_mp_kf, f, nullptr, nnratio, check_orientation, 0, match_f);
}
extern "C" int orbfe_search_by_bow_kf_resident(const orbfe_frame* kf1, const uint8_t* has_mp1, const orbfe_frame* kf2,
                                               const uint8_t* has_mp2, float nnratio, int check_orientation,
                                               int32_t* match12) {
  return bow_resident(kf1, has_mp1, kf2, has_mp2, nnratio, check_orientation, 1, match12);
}

// ORBmatcher::SearchByBoW of ONE frame / key frame against K candidate key frames in one call: Tracking::Relocalization
// runs SearchByBoW(pKF_k, mCurrentFrame, ...) over every candidate (src/Tracking.cc:1478-1498), LoopClosing::ComputeSim3
// SearchByBoW(mpCurrentKF, pKF_k, ...) (src/LoopClosing.cc:294-321).  A single resident call is a round trip of ~0.16 ms
// whatever it computes; here the K shared-node lists and masks go up in ONE copy, K + 1 launches run back to back on one
// stream, and the K match arrays with their counts come back in ONE copy.
//   kfkf = 0: (KF_k, F):   key frame k on the `1` side (mask has_mp_k[k]), `one` = the frame; match [k * one->n + i2]
//   kfkf = 1: (KF, KF_k):  `one` = the current key frame on the `1` side (mask has_mp_one), candidate k on the `2` side
//                          (mask has_mp_k[k]); match [k * one->n + i1]
static int bow_multi(const orbfe_frame* one, const uint8_t* has_mp_one, int K, const orbfe_frame* const* many,
                     const uint8_t* const* has_mp_k, float nnratio, int check_ori, int kfkf, int32_t* match,
                     int32_t* n_matches) {
  UnsettledScope unsettledScope;
  if (!one || K < 0 || (K > 0 && (!many || !has_mp_k || !match || !n_matches)))
    return mfail(ORBFE_ERR_INVALID, "search_by_bow_multi: bad argument");
  const int nOut = one->n;
  for (size_t i = 0; i < (size_t)K * nOut; i++) match[i] = -1;
  for (int k = 0; k < K; k++) n_matches[k] = 0;
  if (K == 0 || nOut == 0) return ORBFE_OK;
  if (!one->haveFv || one->hangle.empty() || (kfkf && !has_mp_one))
    return mfail(ORBFE_ERR_INVALID, "search_by_bow_multi: frame uploaded without FeatureVector / angles, or NULL mask");
  std::vector<std::vector<NodePair>> pairs((size_t)K);
  std::vector<int> maxCnt2((size_t)K, 0);
  size_t bytes = pad((size_t)nOut) + 2 * pad((size_t)K * nOut * 4) + pad((size_t)K * 4) + pad((size_t)K * sizeof(BowArgs)) + pad((size_t)K * 4) + 8192;
  for (int k = 0; k < K; k++) {
    const orbfe_frame* c = many[k];
    if (!c || (c->n > 0 && !has_mp_k[k])) return mfail(ORBFE_ERR_INVALID, "search_by_bow_multi: NULL candidate or mask");
    if (c->device != one->device) return mfail(ORBFE_ERR_INVALID, "search_by_bow_multi: frames on different devices");
    if (c->n == 0) continue;
    if (!c->haveFv || c->hangle.empty())
      return mfail(ORBFE_ERR_INVALID, "search_by_bow_multi: candidate uploaded without FeatureVector / angles");
    if (kfkf) shared_nodes(&one->fv, &c->fv, &pairs[k]); else shared_nodes(&c->fv, &one->fv, &pairs[k]);
    for (const NodePair& p : pairs[k]) maxCnt2[k] = p.cnt2 > maxCnt2[k] ? p.cnt2 : maxCnt2[k];
    if (maxCnt2[k] > 65535) return mfail(ORBFE_ERR_INVALID, "search_by_bow: more than 65535 features in one node");
    bytes += pad(pairs[k].size() * sizeof(NodePair)) + pad((size_t)c->n);
  }
  Arena* ar;
  MHIP(arena_begin(one->device, bytes, &ar));
  MHIP(frame_use(ar, one));
  for (int k = 0; k < K; k++) MHIP(frame_use(ar, many[k]));
  std::vector<NodePair*> dp((size_t)K, nullptr);
  std::vector<uint8_t*> dmk((size_t)K, nullptr);
  uint8_t* dmOne = nullptr;
  if (kfkf) MHIP(up(ar, &dmOne, has_mp_one, (size_t)nOut));
  for (int k = 0; k < K; k++) {
    if (pairs[k].empty()) continue;
    MHIP(up(ar, &dp[k], pairs[k].data(), pairs[k].size()));
    MHIP(up(ar, &dmk[k], has_mp_k[k], (size_t)many[k]->n));
  }
  int32_t* dmatch;
  int8_t* dbin;
  MHIP(up_fill(ar, &dmatch, (size_t)K * nOut, 0xff));  // match arrays and counts adjacent: one copy back
  int32_t* dcount;
  MHIP(up_fill(ar, &dcount, (size_t)K, 0));
  MHIP(up_fill(ar, &dbin, (size_t)K * nOut, 0));
  // the K problems as ONE launch: operands of problem k in hargs[k], its node pairs numbered from pairStart[k]
  std::vector<BowArgs> hargs;
  std::vector<int32_t> pairStart;
  int total = 0, maxAll = 0;
  for (int k = 0; k < K; k++) {
    if (pairs[k].empty()) continue;
    const orbfe_frame* c = many[k];
    const orbfe_frame *f1 = kfkf ? one : c, *f2 = kfkf ? c : one;
    BowArgs a = {};
    a.pairs = dp[k];
    a.desc1 = f1->ddesc; a.hasMp1 = kfkf ? dmOne : dmk[k]; a.angle1 = f1->dangle; a.indices1 = f1->dindices;
    a.desc2 = f2->ddesc; a.hasMp2 = kfkf ? dmk[k] : nullptr; a.angle2 = f2->dangle; a.indices2 = f2->dindices;
    a.angleStride = 1;
    a.nnratio = nnratio; a.strictLow = kfkf; a.match = dmatch + (size_t)k * nOut; a.bin = dbin + (size_t)k * nOut;
    hargs.push_back(a);
    pairStart.push_back(total);
    total += (int)pairs[k].size();
    maxAll = maxCnt2[k] > maxAll ? maxCnt2[k] : maxAll;
  }
  BowArgs* dargs = nullptr;
  int32_t* dstart = nullptr;
  if (!hargs.empty()) {
    MHIP(up(ar, &dargs, hargs.data(), hargs.size()));
    MHIP(up(ar, &dstart, pairStart.data(), pairStart.size()));
  }
  MHIP(flush(ar));
  launch_search_by_bow_multi(ar->stream, dargs, dstart, (int)hargs.size(), total, maxAll);
  launch_rot_prune_batch(ar->stream, dmatch, dbin, nOut, K, check_ori, dcount);  // all K histograms in one launch
  MHIP(hipGetLastError());
  MHIP(down_range(ar, dmatch, dcount + K));
  MHIP(hipStreamSynchronize(ar->stream));
  frames_settle();
  std::memcpy(match, mirror_of(ar, dmatch), (size_t)K * nOut * 4);
  std::memcpy(n_matches, mirror_of(ar, dcount), (size_t)K * 4);
  return ORBFE_OK;
}

extern "C" int orbfe_search_by_bow_multi(int n_keyframes, const orbfe_frame* const* kf, const uint8_t* const* has_mp_kf,
                                         const orbfe_frame* f, float nnratio, int check_orientation, int32_t* match_f,
                                         int32_t* n_matches) {
  return bow_multi(f, nullptr, n_keyframes, kf, has_mp_kf, nnratio, check_orientation, 0, match_f, n_matches);
}
extern "C" int orbfe_search_by_bow_kf_multi(const orbfe_frame* kf1, const uint8_t* has_mp1, int n_keyframes,
                                            const orbfe_frame* const* kf2, const uint8_t* const* has_mp2, float nnratio,
                                            int check_orientation, int32_t* match12, int32_t* n_matches) {
  return bow_multi(kf1, has_mp1, n_keyframes, kf2, has_mp2, nnratio, check_orientation, 1, match12, n_matches);
}

// ORBmatcher::SearchForTriangulation of ONE key frame against K neighbours (LocalMapping::CreateNewMapPoints,
// src/LocalMapping.cc:256-315: the same mpCurrentKeyFrame, a new F12 per neighbour): one upload of the K query lists,
// masks and matrices, 2 K launches on one stream, one download of the K match arrays.
extern "C" int orbfe_search_for_triangulation_multi(const orbfe_frame* kf1, const uint8_t* has_mp1, int n_neighbours,
                                                    const orbfe_frame* const* kf2, const uint8_t* const* has_mp2,
                                                    const float* F12, const float* ex, const float* ey,
                                                    const float* scale_factors2, const float* level_sigma2_2,
                                                    int n_levels2, int only_stereo, int check_orientation,
                                                    int32_t* match12, int32_t* n_matches) {
  UnsettledScope unsettledScope;
  if (!kf1 || n_neighbours < 0 || (n_neighbours > 0 && (!kf2 || !has_mp2 || !F12 || !ex || !ey || !match12 || !n_matches)) ||
      !scale_factors2 || !level_sigma2_2 || n_levels2 <= 0 || n_levels2 > ORBFE_MAX_LEVELS)
    return mfail(ORBFE_ERR_INVALID, "search_for_triangulation_multi: bad argument");
  const int n1 = kf1->n, K = n_neighbours;
  for (size_t i = 0; i < (size_t)K * n1; i++) match12[i] = -1;
  for (int k = 0; k < K; k++) n_matches[k] = 0;
  if (K == 0 || n1 == 0) return ORBFE_OK;
  if (!has_mp1 || !kf1->haveFv || kf1->hangle.empty())
    return mfail(ORBFE_ERR_INVALID, "search_for_triangulation_multi: key frame uploaded without FeatureVector / angles, or NULL mask");
  // host: the shared nodes and the query list of every neighbour (:787-811)
  std::vector<std::vector<TriQuery>> queries((size_t)K);
  size_t bytes = pad((size_t)K * 9 * 4) + 2 * pad((size_t)n_levels2 * 4) + 2 * pad((size_t)K * n1 * 4) + pad((size_t)K * 4) + 8192 +
                 pad((size_t)K * sizeof(TriArgs)) + pad((size_t)K * 4);
  std::vector<NodePair> pairs;
  for (int k = 0; k < K; k++) {
    const orbfe_frame* k2 = kf2[k];
    if (!k2 || (!has_mp2[k] && k2->n > 0)) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation_multi: NULL neighbour");
    if (k2->device != kf1->device) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation_multi: frames on different devices");
    if (k2->n > 0 && (!k2->haveFv || k2->hangle.empty()))
      return mfail(ORBFE_ERR_INVALID, "search_for_triangulation_multi: neighbour uploaded without FeatureVector / angles");
    for (int i = 0; i < k2->n; i++)
      if (k2->hoct[i] < 0 || k2->hoct[i] >= n_levels2) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: octave out of range");
    pairs.clear();
    if (k2->n > 0) shared_nodes(&kf1->fv, &k2->fv, &pairs);
    for (const NodePair& p : pairs) {
      if (p.cnt2 > 65535) return mfail(ORBFE_ERR_INVALID, "search_for_triangulation: more than 65535 features in one node");
      for (int i = 0; i < p.cnt1; i++) {
        const uint32_t idx1 = kf1->hindices[p.off1 + i];
        if (has_mp1[idx1]) continue;                         // :800-803
        if (only_stereo && !kf1->hstereo[idx1]) continue;    // :807-809
        queries[k].push_back(TriQuery{idx1, p.off2, p.cnt2});
      }
    }
    bytes += pad(queries[k].size() * sizeof(TriQuery)) + pad((size_t)k2->n);
  }
  Arena* ar;
  MHIP(arena_begin(kf1->device, bytes, &ar));
  MHIP(frame_use(ar, kf1));
  for (int k = 0; k < K; k++) MHIP(frame_use(ar, kf2[k]));
  float *dF, *dsf, *dsg;
  MHIP(up(ar, &dF, F12, (size_t)K * 9));
  MHIP(up(ar, &dsf, scale_factors2, (size_t)n_levels2));
  MHIP(up(ar, &dsg, level_sigma2_2, (size_t)n_levels2));
  std::vector<TriQuery*> dq((size_t)K, nullptr);
  std::vector<uint8_t*> dm2((size_t)K, nullptr);
  for (int k = 0; k < K; k++) {
    if (queries[k].empty()) continue;
    MHIP(up(ar, &dq[k], queries[k].data(), queries[k].size()));
    MHIP(up(ar, &dm2[k], has_mp2[k], (size_t)kf2[k]->n));
  }
  int32_t* dmatch;
  int8_t* dbin;
  MHIP(up_fill(ar, &dmatch, (size_t)K * n1, 0xff));   // match arrays and counts adjacent: one copy back
  int32_t* dcount;
  MHIP(up_fill(ar, &dcount, (size_t)K, 0));
  MHIP(up_fill(ar, &dbin, (size_t)K * n1, 0));
  // the K problems run as ONE launch: their argument blocks and first workgroups travel with the inputs
  std::vector<TriArgs> targs;
  std::vector<int32_t> blockStart;
  int totalBlocks = 0;
  for (int k = 0; k < K; k++) {
    if (queries[k].empty()) continue;
    const orbfe_frame* k2 = kf2[k];
    TriArgs a = {};
    a.queries = dq[k]; a.nQueries = (int)queries[k].size();
    a.desc1 = kf1->ddesc; a.x1 = kf1->dx; a.y1 = kf1->dy; a.angle1 = kf1->dangle; a.stereo1 = kf1->dstereo;
    a.desc2 = k2->ddesc; a.hasMp2 = dm2[k]; a.x2 = k2->dx; a.y2 = k2->dy; a.angle2 = k2->dangle; a.octave2 = k2->doct;
    a.stereo2 = k2->dstereo; a.indices2 = k2->dindices;
    a.F12 = dF + (size_t)k * 9; a.ex = ex[k]; a.ey = ey[k]; a.scaleFactors2 = dsf; a.levelSigma2_2 = dsg;
    a.onlyStereo = only_stereo; a.match = dmatch + (size_t)k * n1; a.bin = dbin + (size_t)k * n1;
    targs.push_back(a);
    blockStart.push_back(totalBlocks);
    totalBlocks += (a.nQueries + 3) / 4;
  }
  TriArgs* dargs = nullptr;
  int32_t* dstart = nullptr;
  if (!targs.empty()) {
    MHIP(up(ar, &dargs, targs.data(), targs.size()));
    MHIP(up(ar, &dstart, blockStart.data(), blockStart.size()));
  }
  MHIP(flush(ar));
  if (targs.size() == 1) launch_search_triangulation(ar->stream, targs[0]);
  else if (!targs.empty()) launch_search_triangulation_multi(ar->stream, dargs, dstart, (int)targs.size(), totalBlocks);
  launch_rot_prune_batch(ar->stream, dmatch, dbin, n1, K, check_orientation, dcount);  // all K histograms in one launch
  MHIP(hipGetLastError());
  MHIP(down_range(ar, dmatch, dcount + K));
  MHIP(hipStreamSynchronize(ar->stream));
  frames_settle();
  std::memcpy(match12, mirror_of(ar, dmatch), (size_t)K * n1 * 4);
  std::memcpy(n_matches, mirror_of(ar, dcount), (size_t)K * 4);
  return ORBFE_OK;
}

// implemented in extractor.hip (needs the handle internals)
extern "C" int orbfe_stereo_views_(orbfe_extractor* e, int frame, PyramidViews* pv, float* scale, float* invScale,
                                   int* nlevels, int* device, const float** d_scaleTab);

extern "C" int orbfe_compute_stereo_matches(orbfe_extractor* left, int frameL, orbfe_extractor* right, int frameR,
                                            const orbfe_keypoint* kpL, const uint8_t* descL, int N,
                                            const orbfe_keypoint* kpR, const uint8_t* descR, int Nr, float mbf,
                                            float mb, float* uRight, float* depth) {
  if (!left || !right || N < 0 || Nr < 0 || (N > 0 && (!kpL || !descL || !uRight || !depth)) ||
      (Nr > 0 && (!kpR || !descR)))
    return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: bad argument");
  for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
  if (N == 0 || Nr == 0) return 0;
  if (Nr >= (1 << 20)) return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: too many right keypoints");
  StereoArgs a = {};
  int nlL = 0, nlR = 0, devL = 0, devR = 0;
  float scL[kMaxLevels], iscL[kMaxLevels], scR[kMaxLevels], iscR[kMaxLevels];
  const float* dTabR = nullptr;
  int rc;
  if ((rc = orbfe_stereo_views_(left, frameL, &a.pyrL, scL, iscL, &nlL, &devL, &a.scaleTab))) return rc;
  if ((rc = orbfe_stereo_views_(right, frameR, &a.pyrR, scR, iscR, &nlR, &devR, &dTabR))) return rc;
  if (nlL != nlR || devL != devR) return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: extractors differ");
  for (int l = 0; l < nlL; l++)
    if (a.pyrL.lv[l].w != a.pyrR.lv[l].w || a.pyrL.lv[l].h != a.pyrR.lv[l].h)
      return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: left/right pyramids differ in size");
  // host operands are checked here (the reference indexes mvInvScaleFactor / mvImagePyramid / vRowIndices with them
  // unchecked, src/Frame.cc:537-538,560,600-610); the device-operand form cannot look and answers "no stereo" instead
  const float W0 = (float)a.pyrL.lv[0].w, H0 = (float)a.pyrL.lv[0].h;
  for (int side = 0; side < 2; side++) {
    const orbfe_keypoint* kp = side ? kpR : kpL;
    for (int i = 0, n = side ? Nr : N; i < n; i++) {
      if (kp[i].octave < 0 || kp[i].octave >= nlL) return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: octave out of range");
      if (!(kp[i].x >= 0.f && kp[i].x < W0 && kp[i].y >= 0.f && kp[i].y < H0))  // (false for NaN too)
        return mfail(ORBFE_ERR_INVALID, "compute_stereo_matches: keypoint outside the image (or not finite)");
    }
  }
  Arena* ar;
  const int rows = a.pyrL.lv[0].h;
  MHIP(arena_begin(devL, pad((size_t)N * 60) + pad((size_t)Nr * 60) + 3 * pad((size_t)N * 4) + pad((size_t)Nr * 4) +
                             pad((size_t)(rows + 1) * 4) + 4096, &ar));
  float *dkl, *dkr;
  uint8_t *ddl, *ddr;
  MHIP(up(ar, &dkl, reinterpret_cast<const float*>(kpL), (size_t)N * 7));
  MHIP(up(ar, &dkr, reinterpret_cast<const float*>(kpR), (size_t)Nr * 7));
  MHIP(up(ar, &ddl, descL, (size_t)N * 32));
  MHIP(up(ar, &ddr, descR, (size_t)Nr * 32));
  a.kpL = dkl; a.descL = ddl; a.N = N; a.kpR = dkr; a.descR = ddr; a.Nr = Nr;
  a.frameL = frameL; a.frameR = frameR;
  a.mbf = mbf;
  a.maxD = mbf / mb;  // minZ = mb, maxD = mbf/minZ (:542-544)
  a.uRight = carve<float>(ar, N);
  a.depth = carve<float>(ar, N);
  int32_t* dcount = carve<int32_t>(ar, 1);  // (right behind the two outputs: one copy brings all three back)
  a.sad = carve<int32_t>(ar, N);
  if (rows + 1 <= 8192) {  // row index of the right keypoints (k_stereo_bucket)
    a.rowStart = carve<int32_t>(ar, (size_t)rows + 1);
    a.sortedIdx = carve<int32_t>(ar, (size_t)Nr);
    a.rows = rows;
    a.bandR = (int)std::ceil(2.0f * scL[nlL - 1]) + 2;
  }
  MHIP(flush(ar));
  launch_stereo(ar->stream, a, dcount);
  MHIP(hipGetLastError());
  MHIP(down_range(ar, a.uRight, dcount + 1));
  MHIP(hipStreamSynchronize(ar->stream));
  std::memcpy(uRight, mirror_of(ar, a.uRight), (size_t)N * 4);
  std::memcpy(depth, mirror_of(ar, a.depth), (size_t)N * 4);
  return *mirror_of(ar, dcount);
}

// ---------------------------------------------------------------------------------------------
// Frame grid, GetFeaturesInArea and the two tracking-thread projection searches
// (SURVEY.md 8(f) rank 1).  The device gathers every window and takes the Hamming distances
// (k_window.hip); the claim logic below is the reference's sequential loop over the compact
// per-query candidate lists -- it depends on the order of the map points and stays on the host.
// ---------------------------------------------------------------------------------------------
namespace {

struct WindowResult {
  std::vector<int32_t> count;
  std::vector<uint32_t> cand;  // [nq * K] (dist << 16 | feature index) in scan order
  int K = 0;
};

bool frame_ok(const orbfe_frame_view* f) {
  if (!f || f->n < 0 || f->n > GRID_MAX_FEATURES) return false;
  if (!(f->max_x > f->min_x) || !(f->max_y > f->min_y)) return false;
  if (f->n > 0 && (!f->x || !f->y || !f->octave)) return false;
  return true;
}

// Pinned staging of the calling thread: every input array of a call is packed into it and travels in
// ONE host-to-device copy, counts + candidate lists come back in ONE copy (13 + 2 small transfers of
// ~7 us each were most of a call before).
struct Staging {
  uint8_t* h = nullptr;
  size_t cap = 0;
  hipEvent_t pending = nullptr;  // an asynchronous copy OUT of the buffer that nobody waited for (orbfe_frame_upload): the
  bool isPending = false;        // next use of the buffer waits for it first
  ~Staging() {
    if (h) (void)hipHostFree(h);
    if (pending) (void)hipEventDestroy(pending);
  }
};
thread_local Staging t_staging;

hipError_t staging_reserve(size_t bytes);
}  // namespace
namespace {
hipError_t staging_reserve_(size_t bytes) { return staging_reserve(bytes); }
uint8_t* staging_ptr_() { return t_staging.h; }
hipError_t staging_mark_pending_(hipStream_t s) {
  if (!t_staging.pending) {
    hipError_t e = hipEventCreateWithFlags(&t_staging.pending, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipEventRecord(t_staging.pending, s);
  if (e == hipSuccess) t_staging.isPending = true;
  return e;
}
hipError_t staging_reserve(size_t bytes) {
  if (t_staging.isPending) {  // (normally long done: the copy took microseconds, the caller's next call comes later)
    hipError_t e = hipEventSynchronize(t_staging.pending);
    if (e != hipSuccess) return e;
    t_staging.isPending = false;
  }
  if (bytes <= t_staging.cap) return hipSuccess;
  if (t_staging.h) (void)hipHostFree(t_staging.h);
  t_staging.h = nullptr;
  t_staging.cap = 0;
  const size_t want = bytes + bytes / 2 + (1u << 16);
  hipError_t e = hipHostMalloc((void**)&t_staging.h, want, hipHostMallocDefault);
  if (e == hipSuccess) t_staging.cap = want;
  return e;
}

// One window search = one frame + one set of query windows.  Several jobs of a call (Fuse against K neighbour key frames,
// the two directions of SearchBySim3) share ONE upload, one group of launches, one download and one synchronisation.
struct WindowJob {
  const orbfe_frame_view* f;
  int nq;
  const float *qx, *qy, *qr;
  const int32_t *qmin, *qmax;
  const uint8_t* qactive;
  const float* qur;
  const uint8_t* qdesc;  // jobs that pass the SAME pointer share one device copy
  WindowResult* res;
  // BEST mode (bestOut != NULL): no candidate lists come back -- the device keeps the first minimum of every window, behind
  // Fuse's chi-square gate when gate != 0 (gur / invSigma2), and writes the keypoint or -1 (WindowQueries::best)
  int32_t* bestOut = nullptr;
  const float* gur = nullptr;
  const float* invSigma2 = nullptr;
  int nLevels = 0, gate = 0, maxDist = 256;
  // CLAIM mode (claim != NULL): the lists stay on the device and k_window_claim replays the reference's claim loop over
  // them (match_kernels.h: ClaimJob); the match array, the count and -- SearchForInitialization -- the updated previous
  // positions come back
  struct ClaimSpec* claim = nullptr;
};
struct ClaimSpec {
  int mode = CLAIM_BEST, maxDist = 100, checkOri = 0;
  float nnratio = 0.0f;
  const uint8_t* blocked = nullptr;   // [f->n] features taken at entry (NULL: none)
  const uint8_t* blockVal = nullptr;  // [nq] does query i's match hide its feature (NULL: always)
  const float* qAngle = nullptr;      // [nq] (checkOri)
  int32_t* match = nullptr;           // out: [f->n] (BEST / RATIO) or [nq] (INIT)
  int32_t* nMatches = nullptr;        // out
  float* prevX = nullptr;             // INIT, out [nq]: the matched feature's position, else the query's own
  float* prevY = nullptr;
  int rounds = 0;                     // out: rounds the fixed point took
};

// Upload frames (unless resident: keypoint arrays, descriptors and grid are on the device already) + queries, build the
// grids, search every window; grows K until every list fits.
thread_local int t_lastClaimRounds = 0;
int window_search_multi(int device, WindowJob* jobs, int nJobs, int K0) {
  UnsettledScope unsettledScope;
  t_lastClaimRounds = 0;
  int K = ((K0 < 8 ? 8 : K0) + 7) & ~7;  // (k_window_claim reads the lists eight entries at a time)
  struct Lay { size_t oX, oY, oOct, oUr, oDesc, oQx, oQy, oQr, oQmin, oQmax, oQact, oQur, oQdesc, oOut, oGur, oSig, oAng, oBlk, oBval, oQang, oScr;
               bool withDesc, withUr, res; int frameOf; };
  std::vector<Lay> lay((size_t)nJobs);
  size_t off = 0;
  int nClaim = 0;
  bool claimInit = false;
  auto place = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  for (int j = 0; j < nJobs; j++) {
    const WindowJob& J = jobs[j];
    Lay& L = lay[j];
    const size_t n = (size_t)J.f->n, q = (size_t)J.nq;
    L.res = J.f->resident != nullptr;
    if (L.res && J.f->resident->device != device) return mfail(ORBFE_ERR_INVALID, "resident frame lives on another device");
    L.withDesc = J.f->desc && J.qdesc;
    L.withUr = J.qur && J.f->u_right;
    L.oX = L.oY = L.oOct = L.oUr = L.oDesc = L.oAng = L.oBlk = L.oBval = L.oQang = L.oScr = 0;
    L.frameOf = j;  // several jobs on the SAME host-array frame (one frame against K candidates): one upload, one grid
    if (!L.res)
      for (int k = 0; k < j; k++)
        if (!lay[k].res && jobs[k].f == J.f && lay[k].withDesc == L.withDesc) { L.frameOf = lay[k].frameOf; break; }
    if (!L.res && L.frameOf != j) {
      const Lay& F = lay[L.frameOf];
      L.oX = F.oX; L.oY = F.oY; L.oOct = F.oOct; L.oUr = F.oUr; L.oDesc = F.oDesc;
    } else if (!L.res) {
      L.oX = place(n * 4); L.oY = place(n * 4); L.oOct = place(n * 4);
      L.oUr = J.f->u_right ? place(n * 4) : 0;
      L.oDesc = L.withDesc ? place(n * 32) : 0;
    }
    L.oQx = place(q * 4); L.oQy = place(q * 4); L.oQr = place(q * 4); L.oQmin = place(q * 4); L.oQmax = place(q * 4);
    L.oQact = J.qactive ? place(q) : 0;
    L.oQur = L.withUr ? place(q * 4) : 0;
    L.oGur = (J.bestOut && J.gate && J.gur) ? place(q * 4) : 0;
    L.oSig = (J.bestOut && J.gate) ? place((size_t)J.nLevels * 4) : 0;
    L.oQdesc = 0;
    if (L.withDesc) {
      int shared = -1;
      for (int k = 0; k < j; k++)
        if (lay[k].withDesc && jobs[k].qdesc == J.qdesc && jobs[k].nq == J.nq) { shared = k; break; }
      L.oQdesc = shared >= 0 ? lay[shared].oQdesc : place(q * 32);
    }
    if (J.claim) {
      if (J.nq > 0x1fffff) return mfail(ORBFE_ERR_INVALID, "more than 2097151 points in one projection search");
      if (nClaim && claimInit != (J.claim->mode == CLAIM_INIT)) return mfail(ORBFE_ERR_INVALID, "mixed claim forms in one call");
      claimInit = J.claim->mode == CLAIM_INIT;
      nClaim++;
      if (J.claim->checkOri && !L.res) L.oAng = place(n * 4);  // (a resident frame has its angles on the device)
      if (J.claim->blocked) L.oBlk = place(n);
      if (J.claim->blockVal) L.oBval = place(q);
      if (J.claim->checkOri) L.oQang = place(q * 4);
    }
  }
  const size_t oClaim = nClaim ? place((size_t)nClaim * sizeof(ClaimJob)) : 0;
  const size_t oWs = nJobs > 1 ? place((size_t)nJobs * sizeof(WindowSearchJob)) : 0;
  const size_t inBytes = off ? off : 256;
  constexpr size_t kOwnerLds = 60 * 1024;  // the claim kernel's per-feature arrays live in its LDS up to this size (~6800 features)
  for (;;) {
    Arena* ar;
    size_t outBytes = 0, gridBytes = 0, scrBytes = 0, claimLds = 0;
    for (int j = 0; j < nJobs; j++) {
      lay[j].oOut = outBytes;
      const size_t q = (size_t)jobs[j].nq, n = (size_t)jobs[j].f->n;
      if (jobs[j].claim) {
        // comes back: header | match | (INIT: previous positions).  Stays: counts | lists | choice | link | (owner)
        const bool ini = jobs[j].claim->mode == CLAIM_INIT;
        outBytes += pad(16 + (ini ? q : n) * 4 + (ini ? 2 * q * 4 : 0));
        lay[j].oScr = scrBytes;
        // the claim kernel's dynamic LDS: the per-feature owner array(s) -- in HBM when they do not fit -- + the features'
        // octave bytes (RATIO)
        const size_t ownBytes = (ini ? 1 : 2) * n * 4, octBytes = jobs[j].claim->mode == CLAIM_RATIO ? n : 0;
        const bool ownInLds = ownBytes + octBytes <= kOwnerLds;
        scrBytes += pad(q * 4) + pad(q * (size_t)K * 4) + 2 * pad(q * 4) + (ownInLds ? 0 : pad(ownBytes));
        const size_t lds = (ownInLds ? ownBytes : 0) + octBytes;
        if (lds > claimLds) claimLds = lds;
      } else {
        outBytes += pad(q * 4) + (jobs[j].bestOut ? 0 : pad(q * (size_t)K * 4));
      }
      if (!lay[j].res && lay[j].frameOf == j) gridBytes += pad(n * 4) + pad(3073 * 4);
    }
    MHIP(arena_begin(device, pad(inBytes) + gridBytes + outBytes + scrBytes + 2048, &ar));
    for (int j = 0; j < nJobs; j++)
      if (lay[j].res) MHIP(frame_use(ar, jobs[j].f->resident));
    MHIP(staging_reserve(inBytes > outBytes ? inBytes : outBytes));
    uint8_t* h = t_staging.h;
    for (int j = 0; j < nJobs; j++) {
      const WindowJob& J = jobs[j];
      const Lay& L = lay[j];
      const size_t n = (size_t)J.f->n, q = (size_t)J.nq;
      if (!L.res && L.frameOf == j && n) {
        std::memcpy(h + L.oX, J.f->x, n * 4); std::memcpy(h + L.oY, J.f->y, n * 4); std::memcpy(h + L.oOct, J.f->octave, n * 4);
        if (J.f->u_right) std::memcpy(h + L.oUr, J.f->u_right, n * 4);
        if (L.withDesc) std::memcpy(h + L.oDesc, J.f->desc, n * 32);
      }
      if (q) {
        std::memcpy(h + L.oQx, J.qx, q * 4); std::memcpy(h + L.oQy, J.qy, q * 4); std::memcpy(h + L.oQr, J.qr, q * 4);
        std::memcpy(h + L.oQmin, J.qmin, q * 4); std::memcpy(h + L.oQmax, J.qmax, q * 4);
        if (J.qactive) std::memcpy(h + L.oQact, J.qactive, q);
        if (L.withUr) std::memcpy(h + L.oQur, J.qur, q * 4);
        if (L.withDesc) std::memcpy(h + L.oQdesc, J.qdesc, q * 32);
        if (J.bestOut && J.gate && J.gur) std::memcpy(h + L.oGur, J.gur, q * 4);
        if (J.bestOut && J.gate) std::memcpy(h + L.oSig, J.invSigma2, (size_t)J.nLevels * 4);
      }
      if (J.claim) {
        if (L.oAng && n) std::memcpy(h + L.oAng, J.f->angle, n * 4);
        if (J.claim->blocked && n) std::memcpy(h + L.oBlk, J.claim->blocked, n);
        if (J.claim->blockVal && q) std::memcpy(h + L.oBval, J.claim->blockVal, q);
        if (J.claim->checkOri && q) std::memcpy(h + L.oQang, J.claim->qAngle, q * 4);
      }
    }
    uint8_t* din = carve<uint8_t>(ar, inBytes);
    uint8_t* dgrid = carve<uint8_t>(ar, gridBytes ? gridBytes : 1);
    uint8_t* dout = carve<uint8_t>(ar, outBytes ? outBytes : 1);
    uint8_t* dscr = carve<uint8_t>(ar, scrBytes ? scrBytes : 1);
    // the claim jobs' argument blocks travel with the inputs (their device addresses are known once the arena is carved)
    for (int j = 0, c = 0; j < nJobs; j++) {
      const WindowJob& J = jobs[j];
      if (!J.claim) continue;
      const Lay& L = lay[j];
      const size_t q = (size_t)J.nq, n = (size_t)J.f->n;
      const bool ini = J.claim->mode == CLAIM_INIT;
      ClaimJob cj{};
      uint8_t* sc = dscr + L.oScr;
      cj.count = reinterpret_cast<const int32_t*>(sc); sc += pad(q * 4);
      cj.cand = reinterpret_cast<const uint32_t*>(sc); sc += pad(q * (size_t)K * 4);
      cj.choice = reinterpret_cast<int32_t*>(sc); sc += pad(q * 4);
      cj.link = reinterpret_cast<int32_t*>(sc); sc += pad(q * 4);
      const size_t ownBytes = (ini ? 1 : 2) * n * 4, octBytes = J.claim->mode == CLAIM_RATIO ? n : 0;
      cj.owner = ownBytes + octBytes > kOwnerLds ? reinterpret_cast<int32_t*>(sc) : nullptr;
      cj.K = K; cj.nq = J.nq; cj.n = J.f->n;
      cj.active = J.qactive ? din + L.oQact : nullptr;
      cj.blocked = J.claim->blocked ? din + L.oBlk : nullptr;
      cj.blockVal = J.claim->blockVal ? din + L.oBval : nullptr;
      const orbfe_frame* R = L.res ? J.f->resident : nullptr;
      cj.octave = R ? R->doct : reinterpret_cast<const int32_t*>(din + L.oOct);
      cj.qAngle = J.claim->checkOri ? reinterpret_cast<const float*>(din + L.oQang) : nullptr;
      cj.fAngle = J.claim->checkOri ? (R ? R->dangle : reinterpret_cast<const float*>(din + L.oAng)) : nullptr;
      cj.fx = R ? R->dx : reinterpret_cast<const float*>(din + L.oX);
      cj.fy = R ? R->dy : reinterpret_cast<const float*>(din + L.oY);
      cj.qx = reinterpret_cast<const float*>(din + L.oQx); cj.qy = reinterpret_cast<const float*>(din + L.oQy);
      cj.mode = J.claim->mode; cj.maxDist = J.claim->maxDist; cj.checkOri = J.claim->checkOri; cj.nnratio = J.claim->nnratio;
      cj.header = reinterpret_cast<int32_t*>(dout + L.oOut);
      cj.match = cj.header + 4;
      cj.prevX = ini ? reinterpret_cast<float*>(cj.match + q) : nullptr;
      cj.prevY = ini ? cj.prevX + q : nullptr;
      std::memcpy(h + oClaim + (size_t)c * sizeof(ClaimJob), &cj, sizeof(ClaimJob));
      c++;
    }
    // the search jobs: every device address is known once the arena is carved, so the job blocks travel with the inputs too
    size_t goff = 0;
    std::vector<const uint32_t*> keyOf((size_t)nJobs, nullptr);
    std::vector<const int32_t*> cellOf((size_t)nJobs, nullptr);
    std::vector<WindowSearchJob> wsj((size_t)nJobs);
    std::vector<char> buildsGrid((size_t)nJobs, 0);
    int totalBlocks = 0;
    for (int j = 0; j < nJobs; j++) {
      const WindowJob& J = jobs[j];
      const Lay& L = lay[j];
      const size_t n = (size_t)J.f->n, q = (size_t)J.nq;
      GridFrame g{};
      const uint32_t* dkey;
      const int32_t* dcell;
      if (L.res) {
        const orbfe_frame* R = J.f->resident;
        g.x = R->dx; g.y = R->dy; g.octave = R->doct; g.uRight = J.f->u_right ? R->dur : nullptr;
        g.desc = L.withDesc ? R->ddesc : nullptr;
        dkey = R->dkey; dcell = R->dcell;
      } else {
        g.x = reinterpret_cast<const float*>(din + L.oX); g.y = reinterpret_cast<const float*>(din + L.oY);
        g.octave = reinterpret_cast<const int32_t*>(din + L.oOct);
        g.uRight = J.f->u_right ? reinterpret_cast<const float*>(din + L.oUr) : nullptr;
        g.desc = L.withDesc ? din + L.oDesc : nullptr;
      }
      g.n = J.f->n;
      g.minX = J.f->min_x; g.minY = J.f->min_y;
      g.wInv = 64.0f / (J.f->max_x - J.f->min_x);  // src/Frame.cc:109-110 (FRAME_GRID_COLS / ROWS)
      g.hInv = 48.0f / (J.f->max_y - J.f->min_y);
      if (!L.res && L.frameOf != j) {
        dkey = keyOf[L.frameOf]; dcell = cellOf[L.frameOf];
      } else if (!L.res) {
        uint32_t* k = reinterpret_cast<uint32_t*>(dgrid + goff);
        goff += pad(n * 4);
        int32_t* c = reinterpret_cast<int32_t*>(dgrid + goff);
        goff += pad(3073 * 4);
        buildsGrid[j] = 1;
        dkey = k; dcell = c;
        keyOf[j] = k; cellOf[j] = c;
      }
      WindowQueries wq{};
      wq.x = reinterpret_cast<const float*>(din + L.oQx); wq.y = reinterpret_cast<const float*>(din + L.oQy);
      wq.r = reinterpret_cast<const float*>(din + L.oQr);
      wq.minLevel = reinterpret_cast<const int32_t*>(din + L.oQmin); wq.maxLevel = reinterpret_cast<const int32_t*>(din + L.oQmax);
      wq.active = J.qactive ? din + L.oQact : nullptr;
      wq.ur = L.withUr ? reinterpret_cast<const float*>(din + L.oQur) : nullptr;
      wq.desc = L.withDesc ? din + L.oQdesc : nullptr;
      wq.n = J.nq; wq.K = K;
      if (J.bestOut) {
        wq.best = reinterpret_cast<int32_t*>(dout + L.oOut);  // (the job's output block holds the nq keypoint indices)
        wq.gate = J.gate; wq.maxDist = J.maxDist;
        wq.gateUr = (J.gate && J.gur) ? reinterpret_cast<const float*>(din + L.oGur) : nullptr;
        wq.invSigma2 = J.gate ? reinterpret_cast<const float*>(din + L.oSig) : nullptr;
      }
      // counts and candidate lists of a job are adjacent, the jobs' blocks too: one copy back
      int32_t* dcount = reinterpret_cast<int32_t*>(dout + L.oOut);
      uint32_t* dcand = reinterpret_cast<uint32_t*>(dout + L.oOut + pad(q * 4));
      if (J.claim) {  // the lists of a claim job stay in the scratch area
        dcount = reinterpret_cast<int32_t*>(dscr + L.oScr);
        dcand = reinterpret_cast<uint32_t*>(dscr + L.oScr + pad(q * 4));
      }
      wsj[j] = WindowSearchJob{g, dkey, dcell, wq, dcount, dcand, totalBlocks};
      totalBlocks += (J.nq + 3) / 4;
    }
    if (nJobs > 1) std::memcpy(h + oWs, wsj.data(), (size_t)nJobs * sizeof(WindowSearchJob));
    MHIP(hipMemcpyAsync(din, h, inBytes, hipMemcpyHostToDevice, ar->stream));
    for (int j = 0; j < nJobs; j++)
      if (buildsGrid[j]) {
        launch_grid_build(ar->stream, wsj[j].f, const_cast<uint32_t*>(wsj[j].sortedKey), const_cast<int32_t*>(wsj[j].cellOff));
        MHIP(hipGetLastError());
      }
    if (nJobs > 1) {  // ONE launch for the window searches of all jobs
      launch_window_search_multi(ar->stream, reinterpret_cast<const WindowSearchJob*>(din + oWs), nJobs, totalBlocks);
    } else {
      launch_window_search(ar->stream, wsj[0].f, wsj[0].sortedKey, wsj[0].cellOff, wsj[0].q, wsj[0].count, wsj[0].cand);
    }
    MHIP(hipGetLastError());
    if (nClaim) {
      launch_window_claim(ar->stream, reinterpret_cast<const ClaimJob*>(din + oClaim), reinterpret_cast<const ClaimJob*>(h + oClaim), nClaim,
                          claimLds, claimInit);
      MHIP(hipGetLastError());
    }
    if (outBytes) MHIP(hipMemcpyAsync(h, dout, outBytes, hipMemcpyDeviceToHost, ar->stream));
    MHIP(hipStreamSynchronize(ar->stream));
    frames_settle();
    int mx = 0;
    for (int j = 0; j < nJobs; j++) {
      const size_t q = (size_t)jobs[j].nq;
      if (jobs[j].bestOut) {
        if (q) std::memcpy(jobs[j].bestOut, h + lay[j].oOut, q * 4);
        continue;
      }
      if (jobs[j].claim) {
        ClaimSpec* C = jobs[j].claim;
        const int32_t* hd = reinterpret_cast<const int32_t*>(h + lay[j].oOut);
        mx = hd[0] > mx ? hd[0] : mx;
        if (hd[0] > K) continue;  // truncated lists: the call searches again with room for the largest
        const bool ini = C->mode == CLAIM_INIT;
        const size_t nOut = ini ? q : (size_t)jobs[j].f->n;
        if (nOut) std::memcpy(C->match, hd + 4, nOut * 4);
        *C->nMatches = hd[1];
        C->rounds = hd[2];
        t_lastClaimRounds = hd[2] + 1 > t_lastClaimRounds ? hd[2] + 1 : t_lastClaimRounds;
        if (ini && q) { std::memcpy(C->prevX, hd + 4 + q, q * 4); std::memcpy(C->prevY, hd + 4 + 2 * q, q * 4); }
        continue;
      }
      WindowResult* res = jobs[j].res;
      res->count.assign(q, 0);
      res->cand.resize(q * (size_t)K);
      res->K = K;
      if (q) {
        std::memcpy(res->count.data(), h + lay[j].oOut, q * 4);
        std::memcpy(res->cand.data(), h + lay[j].oOut + pad(q * 4), q * (size_t)K * 4);
      }
      for (size_t i = 0; i < q; i++) mx = res->count[i] > mx ? res->count[i] : mx;
    }
    if (mx <= K) return ORBFE_OK;
    K = (mx + 7) & ~7;  // a window held more features than the list: search again with room for the largest
  }
}

int window_search(int device, const orbfe_frame_view* f, int nq, const float* qx, const float* qy, const float* qr,
                  const int32_t* qmin, const int32_t* qmax, const uint8_t* qactive, const float* qur,
                  const uint8_t* qdesc, int K0, WindowResult* res) {
  WindowJob j{f, nq, qx, qy, qr, qmin, qmax, qactive, qur, qdesc, res};
  return window_search_multi(device, &j, 1, K0);
}

}  // namespace

extern "C" int orbfe_debug_last_claim_rounds(void) { return t_lastClaimRounds; }

extern "C" int orbfe_features_in_area(int device, const orbfe_frame_view* frame, int n_queries, const float* x,
                                      const float* y, const float* r, const int32_t* min_level,
                                      const int32_t* max_level, int capacity, int32_t* count, int32_t* indices) {
  frame = canon(frame);
  if (!frame_ok(frame) || n_queries < 0 || capacity < 0 ||
      (n_queries > 0 && (!x || !y || !r || !min_level || !max_level || !count || (capacity > 0 && !indices))))
    return mfail(ORBFE_ERR_INVALID, "features_in_area: bad argument");
  if (n_queries == 0) return ORBFE_OK;
  WindowResult res;
  const int rc = window_search(device, frame, n_queries, x, y, r, min_level, max_level, nullptr, nullptr, nullptr,
                               capacity, &res);
  if (rc != ORBFE_OK) return rc;
  bool over = false;
  for (int i = 0; i < n_queries; i++) {
    count[i] = res.count[i];
    const int m = res.count[i] < capacity ? res.count[i] : capacity;
    if (res.count[i] > capacity) over = true;
    for (int c = 0; c < m; c++) indices[(size_t)i * capacity + c] = (int32_t)(res.cand[(size_t)i * res.K + c] & 0xffffu);
  }
  if (over) return mfail(ORBFE_ERR_CAPACITY, "features_in_area: a window holds more features than capacity (counts are exact)");
  return ORBFE_OK;
}

extern "C" int orbfe_search_by_projection(int device, const orbfe_frame_view* F, const float* scale_factors,
                                          int n_levels, const uint8_t* blocked, int n_mp, const uint8_t* in_view,
                                          const int32_t* level, const float* view_cos, const float* proj_x,
                                          const float* proj_y, const float* proj_xr, const uint8_t* mp_desc,
                                          const uint8_t* mp_obs_positive, float th, float nnratio, int32_t* match,
                                          int32_t* n_matches) {
  F = canon(F);
  if (!frame_ok(F) || !scale_factors || n_levels <= 0 || n_mp < 0 || !n_matches || (F->n > 0 && (!match || !F->desc)) ||
      (n_mp > 0 && (!in_view || !level || !view_cos || !proj_x || !proj_y || !mp_desc)) || (F->u_right && n_mp > 0 && !proj_xr))
    return mfail(ORBFE_ERR_INVALID, "search_by_projection: bad argument");
  for (int i = 0; i < n_mp; i++)
    if (in_view[i] && (level[i] < 0 || level[i] >= n_levels))
      return mfail(ORBFE_ERR_INVALID, "search_by_projection: predicted level outside the pyramid");
  for (int i = 0; i < F->n; i++) match[i] = -1;
  *n_matches = 0;
  if (n_mp == 0 || F->n == 0) return ORBFE_OK;
  const bool bFactor = th != 1.0;
  std::vector<float> qr(n_mp);
  std::vector<int32_t> qmin(n_mp), qmax(n_mp);
  for (int i = 0; i < n_mp; i++) {
    float r = view_cos[i] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, src/ORBmatcher.cc:140-146
    if (bFactor) r *= th;
    const int lv = in_view[i] ? level[i] : 0;
    qr[i] = r * scale_factors[lv];
    qmin[i] = lv - 1;
    qmax[i] = lv;
  }
  // the claim loop (:77-135: best / second best among the features no earlier map point holds, level + ratio test) runs on
  // the device behind the window search (k_window_claim); the match array comes back
  ClaimSpec C;
  C.mode = CLAIM_RATIO; C.maxDist = 100 /* TH_HIGH */; C.nnratio = nnratio;
  C.blocked = blocked; C.blockVal = mp_obs_positive;
  C.match = match; C.nMatches = n_matches;
  WindowJob job{F, n_mp, proj_x, proj_y, qr.data(), qmin.data(), qmax.data(), in_view, proj_xr, mp_desc, nullptr};
  job.claim = &C;
  return window_search_multi(device, &job, 1, 32);
}

extern "C" int orbfe_search_by_projection_last_frame(int device, const orbfe_frame_view* Cur, const float* scale_factors,
                                                     int n_levels, float mbf, int n_last, const uint8_t* valid,
                                                     const float* u, const float* v, const float* invzc,
                                                     const int32_t* last_octave, const float* last_angle,
                                                     const uint8_t* mp_desc, const uint8_t* obs_positive,
                                                     const uint8_t* blocked, int mode,
                                                     float th, int check_orientation, int32_t* match_cur,
                                                     int32_t* n_matches) {
  Cur = canon(Cur);
  if (!frame_ok(Cur) || !scale_factors || n_levels <= 0 || n_last < 0 || !n_matches || mode < 0 || mode > 2 ||
      (Cur->n > 0 && (!match_cur || !Cur->desc)) || (check_orientation && Cur->n > 0 && !Cur->angle) ||
      (n_last > 0 && (!valid || !u || !v || !last_octave || !mp_desc || (check_orientation && !last_angle))) ||
      (Cur->u_right && n_last > 0 && !invzc))
    return mfail(ORBFE_ERR_INVALID, "search_by_projection_last_frame: bad argument");
  for (int i = 0; i < n_last; i++)
    if (valid[i] && (last_octave[i] < 0 || last_octave[i] >= n_levels))
      return mfail(ORBFE_ERR_INVALID, "search_by_projection_last_frame: octave outside the pyramid");
  for (int i = 0; i < Cur->n; i++) match_cur[i] = -1;
  *n_matches = 0;
  if (n_last == 0 || Cur->n == 0) return ORBFE_OK;
  std::vector<float> qr(n_last), qur;
  std::vector<int32_t> qmin(n_last), qmax(n_last);
  if (Cur->u_right) qur.resize(n_last);
  for (int i = 0; i < n_last; i++) {
    const int o = valid[i] ? last_octave[i] : 0;
    qr[i] = th * scale_factors[o];  // src/ORBmatcher.cc:1543
    if (mode == 1) { qmin[i] = o; qmax[i] = -1; }            // bForward:  GetFeaturesInArea(u,v,radius,nLastOctave)
    else if (mode == 2) { qmin[i] = 0; qmax[i] = o; }        // bBackward: (u,v,radius,0,nLastOctave)
    else { qmin[i] = o - 1; qmax[i] = o + 1; }
    if (Cur->u_right) qur[i] = u[i] - mbf * invzc[i];        // :1562
  }
  // claim loop :1572-1612 + rotation histogram :1614-1628 on the device (k_window_claim)
  ClaimSpec C;
  C.mode = CLAIM_BEST; C.maxDist = 100 /* TH_HIGH */; C.checkOri = check_orientation ? 1 : 0;
  C.blocked = blocked;        // mvpMapPoints[i2] set with Observations() > 0 at entry (:1572-1574)
  C.blockVal = obs_positive;
  C.qAngle = last_angle;
  C.match = match_cur; C.nMatches = n_matches;
  WindowJob job{Cur, n_last, u, v, qr.data(), qmin.data(), qmax.data(), valid, Cur->u_right ? qur.data() : nullptr, mp_desc, nullptr};
  job.claim = &C;
  return window_search_multi(device, &job, 1, 32);
}

namespace {

// radius = th * scale_factors[level] and the level window [level + lo, level + hi] of each query
// (hi_open: no level filter at all, KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:611-650)
int level_queries(const char* who, int n, const uint8_t* valid, const int32_t* level, const float* sf, int n_levels,
                  float th, int lo, int hi, bool no_filter, std::vector<float>* qr, std::vector<int32_t>* qmin,
                  std::vector<int32_t>* qmax) {
  qr->resize(n); qmin->resize(n); qmax->resize(n);
  for (int i = 0; i < n; i++) {
    int lv = 0;
    if (valid[i]) {
      lv = level[i];
      if (lv < 0 || lv >= n_levels) return mfail(ORBFE_ERR_INVALID, std::string(who) + ": level outside the pyramid");
    }
    (*qr)[i] = th * sf[lv];
    (*qmin)[i] = no_filter ? -1 : lv + lo;
    (*qmax)[i] = no_filter ? -1 : lv + hi;
  }
  return ORBFE_OK;
}

}  // namespace

extern "C" int orbfe_search_by_projection_keyframe(int device, const orbfe_frame_view* Cur, const float* scale_factors,
                                                   int n_levels, const uint8_t* blocked, int n, const uint8_t* valid,
                                                   const float* u, const float* v, const int32_t* level,
                                                   const float* kf_angle, const uint8_t* mp_desc, float th,
                                                   int orb_dist, int check_orientation, int32_t* match_cur,
                                                   int32_t* n_matches) {
  Cur = canon(Cur);
  if (!frame_ok(Cur) || !scale_factors || n_levels <= 0 || n < 0 || !n_matches ||
      (Cur->n > 0 && (!match_cur || !Cur->desc)) || (check_orientation && Cur->n > 0 && !Cur->angle) ||
      (n > 0 && (!valid || !u || !v || !level || !mp_desc || (check_orientation && !kf_angle))))
    return mfail(ORBFE_ERR_INVALID, "search_by_projection_keyframe: bad argument");
  std::vector<float> qr;
  std::vector<int32_t> qmin, qmax;
  int rc = level_queries("search_by_projection_keyframe", n, valid, level, scale_factors, n_levels, th, -1, +1, false,
                         &qr, &qmin, &qmax);
  if (rc != ORBFE_OK) return rc;
  for (int i = 0; i < Cur->n; i++) match_cur[i] = -1;
  *n_matches = 0;
  if (n == 0 || Cur->n == 0) return ORBFE_OK;
  ClaimSpec C;  // claim loop :1726-1760 on the device
  C.mode = CLAIM_BEST; C.maxDist = orb_dist; C.checkOri = check_orientation ? 1 : 0;
  C.blocked = blocked; C.qAngle = kf_angle;
  C.match = match_cur; C.nMatches = n_matches;
  WindowJob job{Cur, n, u, v, qr.data(), qmin.data(), qmax.data(), valid, nullptr, mp_desc, nullptr};
  job.claim = &C;
  return window_search_multi(device, &job, 1, 32);
}

// The same search of ONE current frame against the projected map points of K candidate key frames in one call
// (Tracking::Relocalization, src/Tracking.cc:1577,1595: matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i],
// sFound, 10 | 3, 100 | 64) per candidate whose PnP pose survived).  Candidate k brings n[k] points: valid[k] / u[k] / v[k]
// / level[k] / kf_angle[k] / mp_desc[k] / blocked[k] (may be NULL) are its arrays; th[k] / orb_dist[k] its window and
// distance bound.  The K window searches share ONE upload of the queries, one launch group and one download
// (window_search_multi); a resident Cur uploads nothing of the frame.  match_cur [k * Cur->n + i2], n_matches [k].
extern "C" int orbfe_search_by_projection_keyframe_multi(int device, const orbfe_frame_view* Cur, const float* scale_factors,
                                                         int n_levels, int n_candidates, const uint8_t* const* blocked,
                                                         const int32_t* n, const uint8_t* const* valid, const float* const* u,
                                                         const float* const* v, const int32_t* const* level,
                                                         const float* const* kf_angle, const uint8_t* const* mp_desc,
                                                         const float* th, const int32_t* orb_dist, int check_orientation,
                                                         int32_t* match_cur, int32_t* n_matches) {
  Cur = canon(Cur);
  const int K = n_candidates;
  if (!frame_ok(Cur) || !scale_factors || n_levels <= 0 || K < 0 ||
      (K > 0 && (!n || !valid || !u || !v || !level || !mp_desc || !th || !orb_dist || !n_matches || (check_orientation && !kf_angle))) ||
      (Cur->n > 0 && K > 0 && (!match_cur || !Cur->desc)) || (check_orientation && Cur->n > 0 && !Cur->angle))
    return mfail(ORBFE_ERR_INVALID, "search_by_projection_keyframe_multi: bad argument");
  std::vector<std::vector<float>> qr((size_t)K);
  std::vector<std::vector<int32_t>> qmin((size_t)K), qmax((size_t)K);
  std::vector<ClaimSpec> claims((size_t)K);
  std::vector<WindowJob> wj;
  for (int k = 0; k < K; k++) {
    if (n[k] < 0 || (n[k] > 0 && (!valid[k] || !u[k] || !v[k] || !level[k] || !mp_desc[k] || (check_orientation && !kf_angle[k]))))
      return mfail(ORBFE_ERR_INVALID, "search_by_projection_keyframe_multi: NULL array of a candidate");
    int rc = level_queries("search_by_projection_keyframe_multi", n[k], valid[k], level[k], scale_factors, n_levels, th[k], -1, +1,
                           false, &qr[k], &qmin[k], &qmax[k]);
    if (rc != ORBFE_OK) return rc;
    for (int i = 0; i < Cur->n; i++) match_cur[(size_t)k * Cur->n + i] = -1;
    n_matches[k] = 0;
    if (n[k] == 0 || Cur->n == 0) continue;
    ClaimSpec& C = claims[k];
    C.mode = CLAIM_BEST; C.maxDist = orb_dist[k]; C.checkOri = check_orientation ? 1 : 0;
    C.blocked = blocked ? blocked[k] : nullptr;
    C.qAngle = check_orientation ? kf_angle[k] : nullptr;
    C.match = match_cur + (size_t)k * Cur->n; C.nMatches = &n_matches[k];
    WindowJob job{Cur, n[k], u[k], v[k], qr[k].data(), qmin[k].data(), qmax[k].data(), valid[k], nullptr, mp_desc[k], nullptr};
    job.claim = &C;
    wj.push_back(job);
  }
  if (wj.empty()) return ORBFE_OK;
  // one claim workgroup per candidate behind the K window searches: one upload, one launch group, one download
  return window_search_multi(device, wj.data(), (int)wj.size(), 32);
}

extern "C" int orbfe_search_by_projection_sim3(int device, const orbfe_frame_view* KF, const float* scale_factors,
                                               int n_levels, const uint8_t* matched, int n, const uint8_t* valid,
                                               const float* u, const float* v, const int32_t* level,
                                               const uint8_t* mp_desc, float th, int32_t* match, int32_t* n_matches) {
  KF = canon(KF);
  if (!frame_ok(KF) || !scale_factors || n_levels <= 0 || n < 0 || !n_matches || (KF->n > 0 && (!match || !KF->desc)) ||
      (n > 0 && (!valid || !u || !v || !level || !mp_desc)))
    return mfail(ORBFE_ERR_INVALID, "search_by_projection_sim3: bad argument");
  std::vector<float> qr;
  std::vector<int32_t> qmin, qmax;
  // the reference gathers the window without a level filter and then keeps octaves [level-1, level]
  // (src/ORBmatcher.cc:410-428): the same set, in the same order, as filtering inside the window
  int rc = level_queries("search_by_projection_sim3", n, valid, level, scale_factors, n_levels, th, -1, 0, false, &qr,
                         &qmin, &qmax);
  if (rc != ORBFE_OK) return rc;
  for (int i = 0; i < KF->n; i++) match[i] = -1;
  *n_matches = 0;
  if (n == 0 || KF->n == 0) return ORBFE_OK;
  ClaimSpec C;  // claim loop :431-451 on the device
  C.mode = CLAIM_BEST; C.maxDist = 50 /* TH_LOW */;
  C.blocked = matched;
  C.match = match; C.nMatches = n_matches;
  WindowJob job{KF, n, u, v, qr.data(), qmin.data(), qmax.data(), valid, nullptr, mp_desc, nullptr};
  job.claim = &C;
  return window_search_multi(device, &job, 1, 32);
}

extern "C" int orbfe_search_for_initialization(int device, const orbfe_frame_view* F1, const orbfe_frame_view* F2,
                                               float* prev_x, float* prev_y, int window_size, float nnratio,
                                               int check_orientation, int32_t* match12, int32_t* n_matches) {
  F1 = canon(F1);
  F2 = canon(F2);
  if (!frame_ok(F1) || !frame_ok(F2) || !n_matches || window_size < 0 ||
      (F1->n > 0 && (!match12 || !prev_x || !prev_y || !F1->desc)) || (F2->n > 0 && !F2->desc) ||
      (check_orientation && ((F1->n > 0 && !F1->angle) || (F2->n > 0 && !F2->angle))))
    return mfail(ORBFE_ERR_INVALID, "search_for_initialization: bad argument");
  const int n1 = F1->n;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  *n_matches = 0;
  if (n1 == 0 || F2->n == 0) return ORBFE_OK;
  std::vector<float> qr(n1, (float)window_size);
  std::vector<int32_t> qlv(n1, 0);
  std::vector<uint8_t> active(n1);
  for (int i = 0; i < n1; i++) active[i] = F1->octave[i] > 0 ? 0 : 1;  // only level-0 keypoints (:482-484)
  for (int i = 0; i < n1; i++) qlv[i] = active[i] ? F1->octave[i] : 0;
  // the claim loop (:492-545: vMatchedDistance, the nnratio test, a later better match replaces the earlier one), the rotation
  // histogram (:557-563) and "update prev matched" (:595-600) run on the device (k_window_claim, CLAIM_INIT)
  ClaimSpec C;
  C.mode = CLAIM_INIT; C.maxDist = 50 /* TH_LOW */; C.nnratio = nnratio; C.checkOri = check_orientation ? 1 : 0;
  C.qAngle = F1->angle;
  C.match = match12; C.nMatches = n_matches;
  C.prevX = prev_x; C.prevY = prev_y;
  WindowJob job{F2, n1, prev_x, prev_y, qr.data(), qlv.data(), qlv.data(), active.data(), nullptr, F1->desc, nullptr};
  job.claim = &C;
  return window_search_multi(device, &job, 1, 128);
}

namespace {
// best keypoint of every window with octave in [level-1, level] (first minimum in scan order), the
// inner search shared by Fuse x2 and SearchBySim3; gate = chi-square test of Fuse (src/ORBmatcher.cc:1029-1060).
// Several (key frame, projected points) jobs of one call run as ONE window_search_multi group.
struct BestJob {
  const orbfe_frame_view* KF;
  const float* sf; const float* inv_level_sigma2;
  int n;
  const uint8_t* valid; const float *u, *v, *ur; const int32_t* level; const uint8_t* desc;
  int32_t* best;
};
int window_best_multi(const char* who, int device, BestJob* jobs, int nJobs, int n_levels, float th, bool gate, int max_dist) {
  std::vector<std::vector<float>> qr((size_t)nJobs);
  std::vector<std::vector<int32_t>> qmin((size_t)nJobs), qmax((size_t)nJobs);
  std::vector<WindowJob> wj;
  for (int j = 0; j < nJobs; j++) {
    BestJob& J = jobs[j];
    int rc = level_queries(who, J.n, J.valid, J.level, J.sf, n_levels, th, -1, 0, false, &qr[j], &qmin[j], &qmax[j]);
    if (rc != ORBFE_OK) return rc;
    for (int i = 0; i < J.n; i++) J.best[i] = -1;
    if (J.n == 0 || J.KF->n == 0) continue;
    // The whole inner search runs on the device (round 4): the window scan, the octave filter, Fuse's chi-square gate
    // (src/ORBmatcher.cc:1036-1058) and the first minimum with its distance bound -- a point's best keypoint depends on no
    // other point, so there is no claim loop to replay: 4 bytes per point come back instead of a 32-entry candidate list
    WindowJob w{J.KF, J.n, J.u, J.v, qr[j].data(), qmin[j].data(), qmax[j].data(), J.valid, nullptr, J.desc, nullptr};
    w.bestOut = J.best;
    w.gate = gate ? 1 : 0;
    w.gur = (gate && J.KF->u_right) ? J.ur : nullptr;
    w.invSigma2 = J.inv_level_sigma2;
    w.nLevels = n_levels;
    w.maxDist = max_dist;
    wj.push_back(w);
  }
  if (wj.empty()) return ORBFE_OK;
  return window_search_multi(device, wj.data(), (int)wj.size(), 8);
}
int window_best(const char* who, int device, const orbfe_frame_view* KF, const float* sf, int n_levels,
                const float* inv_level_sigma2, int n, const uint8_t* valid, const float* u, const float* v,
                const float* ur, const int32_t* level, const uint8_t* desc, float th, bool gate, int max_dist,
                int32_t* best) {
  BestJob j{KF, sf, inv_level_sigma2, n, valid, u, v, ur, level, desc, best};
  return window_best_multi(who, device, &j, 1, n_levels, th, gate, max_dist);
}
}  // namespace

// The per-point search of ORBmatcher::Fuse for the SAME map points against K key frames in one call: what
// LocalMapping::SearchInNeighbors does neighbour by neighbour (src/LocalMapping.cc:542-549: matcher.Fuse(pKFi,
// vpMapPointMatches) for every target key frame).  Arrays are [k * n + i]; mp_desc [n * 32] is shared.
extern "C" int orbfe_fuse_search_multi(int device, int n_keyframes, const orbfe_frame_view* const* KF,
                                       const float* scale_factors, const float* inv_level_sigma2, int n_levels, int n,
                                       const uint8_t* valid, const float* u, const float* v, const float* ur,
                                       const int32_t* level, const uint8_t* mp_desc, float th, int chi2_gate,
                                       int32_t* best_idx) {
  if (n_keyframes < 0 || n < 0 || !scale_factors || n_levels <= 0 || (n_keyframes > 0 && !KF) ||
      (n_keyframes > 0 && n > 0 && (!valid || !u || !v || !level || !mp_desc || !best_idx)) || (chi2_gate && !inv_level_sigma2))
    return mfail(ORBFE_ERR_INVALID, "fuse_search_multi: bad argument");
  std::vector<BestJob> jobs;
  for (int k = 0; k < n_keyframes; k++) {
    const orbfe_frame_view* f = canon(KF[k]);
    if (!frame_ok(f) || (f->n > 0 && !f->desc) || (chi2_gate && f->u_right && n > 0 && !ur))
      return mfail(ORBFE_ERR_INVALID, "fuse_search_multi: bad key frame view");
    for (int i = 0; i < f->n; i++)
      if (chi2_gate && (f->octave[i] < 0 || f->octave[i] >= n_levels))
        return mfail(ORBFE_ERR_INVALID, "fuse_search_multi: keypoint octave outside the pyramid");
    const size_t o = (size_t)k * n;
    jobs.push_back(BestJob{f, scale_factors, inv_level_sigma2, n, valid + o, u + o, v + o, ur ? ur + o : nullptr, level + o,
                           mp_desc, best_idx + o});
  }
  if (jobs.empty()) return ORBFE_OK;
  return window_best_multi("fuse_search_multi", device, jobs.data(), (int)jobs.size(), n_levels, th, chi2_gate != 0, 50 /* TH_LOW */);
}

extern "C" int orbfe_fuse_search(int device, const orbfe_frame_view* KF, const float* scale_factors,
                                 const float* inv_level_sigma2, int n_levels, int n, const uint8_t* valid,
                                 const float* u, const float* v, const float* ur, const int32_t* level,
                                 const uint8_t* mp_desc, float th, int chi2_gate, int32_t* best_idx) {
  KF = canon(KF);
  if (!frame_ok(KF) || !scale_factors || n_levels <= 0 || n < 0 || (KF->n > 0 && !KF->desc) ||
      (n > 0 && (!valid || !u || !v || !level || !mp_desc || !best_idx)) ||
      (chi2_gate && (!inv_level_sigma2 || (KF->u_right && n > 0 && !ur))))
    return mfail(ORBFE_ERR_INVALID, "fuse_search: bad argument");
  for (int i = 0; i < KF->n; i++)
    if (chi2_gate && (KF->octave[i] < 0 || KF->octave[i] >= n_levels))
      return mfail(ORBFE_ERR_INVALID, "fuse_search: keypoint octave outside the pyramid");
  return window_best("fuse_search", device, KF, scale_factors, n_levels, inv_level_sigma2, n, valid, u, v, ur, level,
                     mp_desc, th, chi2_gate != 0, 50 /* TH_LOW */, best_idx);
}

extern "C" int orbfe_search_by_sim3(int device, const orbfe_frame_view* KF1, const orbfe_frame_view* KF2,
                                    const float* scale_factors1, const float* scale_factors2, int n_levels,
                                    const uint8_t* valid1, const float* u1, const float* v1, const int32_t* level1,
                                    const uint8_t* desc1, const uint8_t* valid2, const float* u2, const float* v2,
                                    const int32_t* level2, const uint8_t* desc2, float th, int32_t* match12,
                                    int32_t* n_found) {
  KF1 = canon(KF1);
  KF2 = canon(KF2);
  if (!frame_ok(KF1) || !frame_ok(KF2) || !scale_factors1 || !scale_factors2 || n_levels <= 0 || !n_found ||
      (KF1->n > 0 && (!valid1 || !u1 || !v1 || !level1 || !desc1 || !match12 || !KF1->desc)) ||
      (KF2->n > 0 && (!valid2 || !u2 || !v2 || !level2 || !desc2 || !KF2->desc)))
    return mfail(ORBFE_ERR_INVALID, "search_by_sim3: bad argument");
  std::vector<int32_t> m1(KF1->n ? KF1->n : 1), m2(KF2->n ? KF2->n : 1);
  // both directions (src/ORBmatcher.cc:1190-1275 and :1277-1358) as one upload / launch group / download
  BestJob jobs[2] = {{KF2, scale_factors2, nullptr, KF1->n, valid1, u1, v1, nullptr, level1, desc1, m1.data()},
                     {KF1, scale_factors1, nullptr, KF2->n, valid2, u2, v2, nullptr, level2, desc2, m2.data()}};
  int rc = window_best_multi("search_by_sim3", device, jobs, 2, n_levels, th, false, 100 /* TH_HIGH */);
  if (rc != ORBFE_OK) return rc;
  int nFound = 0;
  for (int i1 = 0; i1 < KF1->n; i1++) {
    match12[i1] = -1;
    const int idx2 = m1[i1];
    if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
  }
  *n_found = nFound;
  return ORBFE_OK;
}
