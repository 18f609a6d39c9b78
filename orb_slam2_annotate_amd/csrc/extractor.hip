// extractor.hip -- the extractor handle, its HBM workspace and the batch pipeline behind the
// C-ABI of include/orbfe.h (replacing ORBextractor::operator(), src/ORBextractor.cc:1119-1197).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbfe.h"
#include "kernels.h"
#include "match_kernels.h"
#include "octree_host.h"
#include "orb_params.h"
#include "orb_pattern.inc"

using namespace orbfe;

static thread_local std::string g_err = "";
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return fail(ORBFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

extern "C" const char* orbfe_last_error(void) { return g_err.c_str(); }
int orbfe_set_error_(int code, const char* msg) { return fail(code, msg); }  // used by matcher.hip, vocabulary.hip
extern "C" int orbfe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static_assert(sizeof(orbfe_keypoint) == 28, "cv::KeyPoint layout");

struct orbfe_extractor {
  ExtractorTables tab;
  int device = 0;
  hipStream_t stream = nullptr;              // stream 0: owns the timing events and the copies
  static constexpr int kMaxStreams = 32;
  hipStream_t extra[kMaxStreams - 1] = {};   // sub-batch streams 1..31
  int nStreams = 1;                          // >1: sub-batches of one call run concurrently
  int subsReady = 0;                         // sub-batches whose stream and events exist (ensure_subs)
  // cross-stream ordering without host synchronisation: evChunkDone[i] marks the end of sub-batch i of the
  // last extract call (consumers on `stream` wait for it); evConsumerDone marks the end of the last kernel
  // on `stream` that READS the workspace / the caller's outputs of all sub-batches (batched stereo matcher),
  // and every sub-batch stream of the next extract call waits for it before it overwrites them
  // lane schedule (laneMode): every sub-batch uses the same three streams -- P = extra[0] (pyramid), V = extra[1]
  // (FAST, blur: the VALU-bound kernels), T = stream (gather, octree, orientation + descriptors: the latency-bound
  // tail) -- so the sub-batches form a software pipeline: while V works on sub-batch k, P already builds the pyramid
  // of k+1 and T finishes k-1.  One VALU-bound kernel is in flight at any time instead of whatever the phases of
  // independent sub-batch streams happen to overlap.  evPyr/evFast/evBlur order the lanes inside a sub-batch,
  // evTail[k] (end of sub-batch k on T) holds back the next call's pyramid of the same workspace slice.
  bool laneMode = false;
  hipEvent_t evPyr[kMaxStreams] = {}, evFast[kMaxStreams] = {}, evBlur[kMaxStreams] = {}, evTail[kMaxStreams] = {};
  bool tailPending[kMaxStreams] = {};
  hipEvent_t evChunkDone[kMaxStreams] = {};
  hipEvent_t evConsumerDone = nullptr;
  int chunksPending = 0;                     // sub-batch streams 1..chunksPending-1 carry an unrecorded-for-consumer event
  bool consumerPending = false;
  int lastSplitFrames = -1, lastSplitStreams = -1;
  int lastPer = 0, lastS = 1;     // frames per sub-batch and sub-batch count of the last run_pipeline call
  bool lastLanes = false;
  double stageFrames[ORBFE_STAGE_COUNT] = {};
  // stage timing: a ring of event pairs per stage so that asynchronous calls can stay in flight
  static constexpr int kEvRing = 8;
  static constexpr int kEvSubs = kMaxStreams;  // one event pair per (call slot, sub-batch stream, stage)
  hipEvent_t evA[kEvRing][kEvSubs][ORBFE_STAGE_COUNT] = {}, evB[kEvRing][kEvSubs][ORBFE_STAGE_COUNT] = {};
  bool evUsed[kEvRing][kEvSubs][ORBFE_STAGE_COUNT] = {};
  int evLaunches[kEvRing][kEvSubs][ORBFE_STAGE_COUNT] = {};
  int evFrames[kEvRing][kEvSubs][ORBFE_STAGE_COUNT] = {};
  int evSlot = 0;
  unsigned stageMask = 0;
  bool hostOctree = false;  // debug cross-check only (orbfe_extractor_debug_host_octree)
  // GaussianBlur inside the FAST kernel (k_fast_cells<.., true>) instead of the separate k_blur7 launch.  Measured
  // (r02, same box, 4096 VGA frames): fused 8.41 ms vs 5.28 + 3.26 ms for the two launches, pipeline 285.9 k vs
  // 291.8 k frames/s -- FAST already runs at the VALU issue ceiling, so the blur's instructions cost their full price
  // inside it; off by default, selectable ($ORBFE_FUSED=1 / orbfe_extractor_set_fused) and parity-tested
  bool fused = false;
  // FAST threshold order (k_fast_cells<.., kLowFirst>): 0 = auto (picked per call from the fallback rate the last
  // finished launch measured), 1 = iniThFAST first + per-cell fallback, 2 = one attempt at the lower threshold
  int fastMode = 0;
  bool fastLowFirst = false;                 // current choice in auto mode
  static constexpr int kStatSlots = 64;       // counters per sub-batch (k_fast_cells spreads its sampled reports over them)
  unsigned int* d_fastStat = nullptr;        // per sub-batch: sampled count of cells that needed minThFAST
  unsigned int* h_fastStat = nullptr;        // pinned copy
  hipEvent_t evStat[kMaxStreams] = {};
  bool statPending[kMaxStreams] = {};
  double statCells[kMaxStreams] = {};
  bool copyUnaligned = false;  // debug: repack caller-owned frames with an odd stride first (round-1 behaviour)
  int blurSpec = kBlurSpecCv4;  // GaussianBlur arithmetic, orbfe_extractor_set_blur_spec / $ORBFE_BLUR_SPEC
  int octreeMaxL = 0;
  double stageMs[ORBFE_STAGE_COUNT] = {};
  int64_t stageLaunches[ORBFE_STAGE_COUNT] = {};

  FrameGeom geom;
  int capFrames = 0;  // frames the workspace is sized for
  // constant device tables
  float4* d_patternF = nullptr;
  uint4* d_momentTab = nullptr;
  uint8_t* d_hostIn = nullptr;      // input slab of the small host-batch path (frames at the caller's pitch)
  size_t hostInBytes = 0;
  DescTile* d_descTiles = nullptr;  // tile form of the orientation + descriptor stage (k_desc_tiles.hip)
  int nDescTiles = 0;
  int knockoutChunks = 0;
  int descTilesMode = -1;           // -1: $ORBFE_DESC_TILES (default 0 = the per-keypoint form); 0 / 1 forced by orbfe_extractor_set_desc_tiles
  int32_t* d_umax = nullptr;
  CellDesc* d_cells = nullptr;
  LevelGeom* d_lvgeom = nullptr;
  int32_t* d_xofs[kMaxLevels] = {};
  int16_t* d_alpha[kMaxLevels] = {};
  int32_t* d_yofs[kMaxLevels] = {};
  int16_t* d_beta[kMaxLevels] = {};
  uint32_t* d_colrec[kMaxLevels] = {};
  uint32_t* d_rowrec[kMaxLevels] = {};
  ChainTile* d_chainTiles = nullptr;    // the one-launch pyramid of the single-frame form (k_pyramid_chain): tiles + what each needs
  int nChainTiles = 0, chainBufA = 0, chainBufB = 0, chainMaxW = 0, chainMaxH = 0;
  bool chainOk = false;
  bool pyrChain = false;                // use it for calls of <= 8 frames ($ORBFE_PYR_CHAIN, orbfe_extractor_set_pyramid_chain): off,
                                        // it measured no faster than the seven launches
  int32_t* d_tileGx[kMaxLevels] = {};   // ownership tables of the fused blur + resize kernel (ResizeTables::tileGx / tileDy)
  int32_t* d_tileDy[kMaxLevels] = {};
  bool pyrBlur = true;                  // blur level l and write level l+1 from the same staged tiles ($ORBFE_PYRBLUR,
                                        // orbfe_extractor_set_pyramid_blur)
  // per-batch workspace
  uint8_t* d_pyr = nullptr;
  uint8_t* d_blur = nullptr;
  Candidate* d_slots = nullptr;
  Candidate* d_cand = nullptr;
  uint16_t* d_cellCount = nullptr;
  int32_t* d_cellPrefix = nullptr;
  int32_t* d_candCount = nullptr;
  uint16_t* d_nodeOf = nullptr;
  uint8_t* d_octreeWork = nullptr;  // node lists of k_octree_global (only when they do not fit in LDS)
  size_t octreeWorkStride = 0;
  float* d_scaleTab = nullptr;     // mvScaleFactor[16] + mvInvScaleFactor[16]
  float* d_frameStereo = nullptr;  // orbfe_extract_stereo_frame: mvuRight | mvDepth | survivors of the pair
  size_t frameStereoCap = 0;
  int32_t* d_stereoSad = nullptr;  // scratch of the batched stereo matcher
  int32_t* d_stereoRowStart = nullptr;
  int32_t* d_stereoSorted = nullptr;
  float4* d_stereoRec = nullptr;   // (uR, yR, octave, index) of the right keypoints in row order
  size_t stereoSadCap = 0, stereoRowCap = 0;
  LevelKp* d_levelKp = nullptr;
  int32_t* d_levelCount = nullptr;
  // device-side outputs used by the host-buffer API
  // outputs of the host-buffer API: ONE block [keypoints | descriptors | counts], so a small batch
  // comes back in a single D2H copy through the pinned staging buffer
  uint8_t* d_outBlock = nullptr;
  size_t outBlockBytes = 0;      // bytes from the block start to the end of the counts of the last call
  orbfe_keypoint* d_kpOut = nullptr;
  uint8_t* d_descOut = nullptr;
  int32_t* d_nOut = nullptr;
  // the handle's own output block still holds the records of the last HOST-buffer call (orbfe_extract / small
  // orbfe_extract_batch): orbfe_frame_from_extractor builds resident frames from it without the features travelling again
  int outLastFrames = 0, outLastCapacity = 0;
  std::vector<int32_t> outLastCount;
  uint8_t* h_outStage = nullptr;  // pinned
  size_t outStageBytes = 0;
  int outCap = 0;
  // pinned-host pipelined path (orbfe_extract_batch_pipelined): two input slabs + two output blocks in HBM,
  // one copy stream per direction, events per slot
  hipStream_t sH2D = nullptr, sD2H = nullptr;
  uint8_t* d_pipeIn[2] = {nullptr, nullptr};
  uint8_t* d_pipeOut[2] = {nullptr, nullptr};
  size_t pipeInBytes = 0, pipeOutBytes = 0;
  hipEvent_t evIn[2] = {}, evComp[2] = {}, evOutDone[2] = {};
  // host staging
  std::vector<int32_t> h_candCount, h_levelCount;
  std::vector<Candidate> h_cand;
  std::vector<LevelKp> h_levelKp;
  // description of the last call (for mvImagePyramid-style reads)
  PyramidViews lastPyr = {};
  PyramidViews lastBlur = {};
  int lastFrames = 0;
  int lastFrameBase = 0;  // index, in the caller's batch, of the first frame the retained pyramid belongs to (pipelined host path: its LAST chunk)
  bool haveLast = false;
};

namespace {

template <typename T>
int dalloc(T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  return ORBFE_OK;
}
template <typename T>
void dfree(T** p) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
}

// HIP streams of destroyed handles are kept and handed to the next handle that asks for the same role (sub-batch i, the
// two copy streams).  Two reasons.  (1) hipStreamCreate / Destroy are not free, and a process that makes one handle per
// workload or per camera re-creates the same streams again and again.  (2) Round 4 measured that the ORDER in which a process
// creates its streams matters: a HIP stream is bound to one of the GPU_MAX_HW_QUEUES hardware queues when it is created, and
// a library that creates streams of its own in between -- RCCL, when torch.distributed is initialised -- shifts that
// binding for every stream created after it: the 8-stream KITTI pipeline ran 11 % slower with the communicator created
// first (104.0 k -> 92.5 k stereo frames/s; bench.py LazyDist, tools/dist_ab.sh).  A recycled stream keeps its queue.
namespace {
struct StreamPool {
  std::mutex m;
  struct Item { int device, role; hipStream_t s; };
  std::vector<Item> items;
};
StreamPool g_streamPool;
constexpr int kRoleH2D = 1000, kRoleD2H = 1001;
hipError_t stream_get(int device, int role, hipStream_t* s) {
  {
    std::lock_guard<std::mutex> lk(g_streamPool.m);
    auto& v = g_streamPool.items;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].device == device && v[i].role == role) { *s = v[i].s; v.erase(v.begin() + (long)i); return hipSuccess; }
  }
  // $ORBFE_STREAM_PRIORITY = high | low: the handle's streams in a priority class of their own (its own set of hardware
  // queues in the HIP runtime), so that streams other libraries create at normal priority cannot shift their binding
  static const int kPrioEnv = [] {
    const char* v = getenv("ORBFE_STREAM_PRIORITY");
    return !v ? 0 : (std::string(v) == "high" ? 1 : (std::string(v) == "low" ? 2 : 0));
  }();
  if (kPrioEnv) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      return hipStreamCreateWithPriority(s, hipStreamNonBlocking, kPrioEnv == 1 ? greatest : least);
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void stream_put(int device, int role, hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  {
    std::lock_guard<std::mutex> lk(g_streamPool.m);
    if (g_streamPool.items.size() < 256) { g_streamPool.items.push_back({device, role, s}); return; }
  }
  (void)hipStreamDestroy(s);
}
}  // namespace

// Streams and events of sub-batches [e->subsReady, n) are created on first use: a handle for one live camera (1 stream)
// costs 1 stream + 136 events to create instead of 32 streams + ~4 300 events.
int ensure_subs(orbfe_extractor* e, int n) {
  if (n > orbfe_extractor::kMaxStreams) n = orbfe_extractor::kMaxStreams;
  // $ORBFE_STREAM_SKEW = k (experiment, tools/skew_ab.sh): k idle streams are created in front of the handle's first extra
  // stream, shifting the hardware-queue binding of everything created after them -- what a library like RCCL does by accident
  static const int kSkew = getenv("ORBFE_STREAM_SKEW") ? atoi(getenv("ORBFE_STREAM_SKEW")) : 0;
  static bool skewed = false;
  if (kSkew > 0 && !skewed && n > 1) {
    skewed = true;
    for (int k = 0; k < kSkew; k++) { hipStream_t dummy; (void)hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking); }
  }
  for (int i = e->subsReady; i < n; i++) {
    if (i > 0 && !e->extra[i - 1]) HIPCHK(stream_get(e->device, i, &e->extra[i - 1]));
    HIPCHK(hipEventCreateWithFlags(&e->evStat[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evChunkDone[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evPyr[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evFast[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evBlur[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evTail[i], hipEventDisableTiming));
    for (int r = 0; r < orbfe_extractor::kEvRing; r++)
      for (int st = 0; st < ORBFE_STAGE_COUNT; st++) {
        HIPCHK(hipEventCreate(&e->evA[r][i][st]));
        HIPCHK(hipEventCreate(&e->evB[r][i][st]));
      }
    e->subsReady = i + 1;
  }
  return ORBFE_OK;
}

void free_geometry(orbfe_extractor* e) {
  dfree(&e->d_cells);
  dfree(&e->d_lvgeom);
  dfree(&e->d_descTiles);
  e->nDescTiles = 0;
  dfree(&e->d_chainTiles);
  e->nChainTiles = 0;
  e->chainOk = false;
  for (int l = 0; l < kMaxLevels; l++) { dfree(&e->d_xofs[l]); dfree(&e->d_alpha[l]); dfree(&e->d_yofs[l]); dfree(&e->d_beta[l]); dfree(&e->d_colrec[l]); dfree(&e->d_rowrec[l]); dfree(&e->d_tileGx[l]); dfree(&e->d_tileDy[l]); }
}
void free_workspace(orbfe_extractor* e) {
  dfree(&e->d_pyr); dfree(&e->d_blur); dfree(&e->d_slots); dfree(&e->d_cand);
  dfree(&e->d_cellCount); dfree(&e->d_cellPrefix); dfree(&e->d_candCount); dfree(&e->d_nodeOf);
  dfree(&e->d_levelKp); dfree(&e->d_levelCount); dfree(&e->d_octreeWork);
  e->octreeWorkStride = 0;
  e->capFrames = 0;
}
void free_outputs(orbfe_extractor* e) {
  dfree(&e->d_outBlock);
  e->d_kpOut = nullptr; e->d_descOut = nullptr; e->d_nOut = nullptr;
  e->outCap = 0;
  e->outLastFrames = 0;
}

int ensure_geometry(orbfe_extractor* e, int W, int H) {
  if (e->geom.W == W && e->geom.H == H && e->d_lvgeom) return ORBFE_OK;
  if (W > 32767 || H > 32767)  // signed 16-bit node rectangles in k_octree (unsigned 16-bit coordinates elsewhere)
    return fail(ORBFE_ERR_INVALID, "images larger than 32767 x 32767 are not supported");
  free_geometry(e);
  free_workspace(e);
  e->haveLast = false;
  e->geom.build(e->tab, W, H);
  const FrameGeom& g = e->geom;
  for (const CellDesc& c : g.cells)
    if (c.w > 60 || c.h > 60) return fail(ORBFE_ERR_INVALID, "FAST grid cell larger than 60 px");
  for (int l = 0; l < g.nlevels; l++)
    if (g.lv[l].w < 1 || g.lv[l].h < 1) return fail(ORBFE_ERR_INVALID, "image too small for the pyramid");
  int maxL = 4;
  for (int l = 0; l < g.nlevels; l++) {
    if (g.lv[l].kpCap > maxL) maxL = g.lv[l].kpCap;
    if (g.lv[l].nIni > maxL) maxL = g.lv[l].nIni;
    if (g.lv[l].slotCount >= (1 << 24)) return fail(ORBFE_ERR_INVALID, "level too large for the octree kernel");
  }
  maxL = (maxL + 3) & ~3;
  if (maxL > 65532)  // 16-bit node positions (nodeOf, order, rank)
    return fail(ORBFE_ERR_INVALID, "nfeatures too large: more than 65532 keypoints on one pyramid level");
  e->octreeMaxL = maxL;  // octree_lds_bytes(maxL) > kOctreeLdsLimit: node lists in global memory (ensure_workspace)
  int rc;
  if ((rc = dalloc(&e->d_cells, g.cells.size()))) return rc;
  if (!g.cells.empty()) HIPCHK(hipMemcpy(e->d_cells, g.cells.data(), g.cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
  if ((rc = dalloc(&e->d_lvgeom, (size_t)kMaxLevels))) return rc;
  HIPCHK(hipMemcpy(e->d_lvgeom, g.lv, sizeof(LevelGeom) * kMaxLevels, hipMemcpyHostToDevice));
  {
    const std::vector<DescTile> tiles = build_desc_tiles(g.lv, g.nlevels);
    e->nDescTiles = (int)tiles.size();
    if (!tiles.empty()) {
      if ((rc = dalloc(&e->d_descTiles, tiles.size()))) return rc;
      HIPCHK(hipMemcpy(e->d_descTiles, tiles.data(), tiles.size() * sizeof(DescTile), hipMemcpyHostToDevice));
    }
  }
  {
    // k_pyramid_chain: 32 x 32 tiles of levels 1 .. n-1, each with the rectangles it needs of the levels below it
    // (walking the cv::resize tables back to level 0); usable while the two LDS rectangle buffers fit
    std::vector<ChainTile> tiles;
    size_t bufA = 0, bufB = 0;
    int maxW = 0, maxH = 0;
    bool ok = g.nlevels > 1;
    for (int l = g.nlevels - 1; l >= 1 && ok; l--) {  // the longest chains first
      const int TS = l >= 4 ? 16 : 32;  // (deep levels: smaller tiles, i.e. smaller rectangles to recompute -- 16 measured best of 32 / 16 / 8)
      for (int y0 = 0; y0 < g.lv[l].h && ok; y0 += TS)
        for (int x0 = 0; x0 < g.lv[l].w; x0 += TS) {
          ChainTile t = {};
          t.level = l;
          int rx0 = x0, ry0 = y0, rx1 = std::min(x0 + TS, g.lv[l].w) - 1, ry1 = std::min(y0 + TS, g.lv[l].h) - 1;  // inclusive
          for (int k = l; k >= 0; k--) {
            const int w = rx1 - rx0 + 1, h = ry1 - ry0 + 1;
            if (w > 255 || h > 255) { ok = false; break; }
            t.r[k] = ChainRect{(int16_t)rx0, (int16_t)ry0, (int16_t)w, (int16_t)h};
            const size_t bytes = (size_t)((w + 3) & ~3) * h;
            if (k & 1) bufB = std::max(bufB, bytes); else bufA = std::max(bufA, bytes);
            if (k > 0) { maxW = std::max(maxW, w); maxH = std::max(maxH, h); }  // (table entries of the level-k output rectangle)
            if (k == 0) break;
            const ResizeTables& z = g.rz[k];
            const int Wp = g.lv[k - 1].w, Hp = g.lv[k - 1].h;
            auto clampr = [&](int v) { return v < 0 ? 0 : (v >= Hp ? Hp - 1 : v); };
            const int sx0 = z.xofs[rx0], sx1 = std::min(z.xofs[rx1] + 1, Wp - 1);
            const int sy0 = clampr(z.yofs[ry0]), sy1 = clampr(z.yofs[ry1] + 1);
            rx0 = sx0; rx1 = std::max(sx1, sx0); ry0 = sy0; ry1 = std::max(sy1, sy0);
          }
          if (!ok) break;
          tiles.push_back(t);
        }
    }
    bufA = (bufA + 15) & ~(size_t)15;
    bufB = (bufB + 15) & ~(size_t)15;
    if (ok && bufA + bufB + (size_t)(maxW + maxH) * 8 > (size_t)60 * 1024) ok = false;
    e->chainOk = false;
    if (ok && !tiles.empty()) {
      if ((rc = dalloc(&e->d_chainTiles, tiles.size()))) return rc;
      HIPCHK(hipMemcpy(e->d_chainTiles, tiles.data(), tiles.size() * sizeof(ChainTile), hipMemcpyHostToDevice));
      e->nChainTiles = (int)tiles.size();
      e->chainBufA = (int)bufA; e->chainBufB = (int)bufB; e->chainMaxW = maxW; e->chainMaxH = maxH;
      e->chainOk = true;
    }
  }
  for (int l = 1; l < g.nlevels; l++) {
    const ResizeTables& t = g.rz[l];
    if ((rc = dalloc(&e->d_xofs[l], t.xofs.size()))) return rc;
    if ((rc = dalloc(&e->d_alpha[l], t.alpha.size()))) return rc;
    if ((rc = dalloc(&e->d_yofs[l], t.yofs.size()))) return rc;
    if ((rc = dalloc(&e->d_beta[l], t.beta.size()))) return rc;
    HIPCHK(hipMemcpy(e->d_xofs[l], t.xofs.data(), t.xofs.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_alpha[l], t.alpha.data(), t.alpha.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_yofs[l], t.yofs.data(), t.yofs.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_beta[l], t.beta.data(), t.beta.size() * 2, hipMemcpyHostToDevice));
    if (!t.colrec.empty()) {
      if ((rc = dalloc(&e->d_colrec[l], t.colrec.size()))) return rc;
      if ((rc = dalloc(&e->d_rowrec[l], t.rowrec.size()))) return rc;
      HIPCHK(hipMemcpy(e->d_colrec[l], t.colrec.data(), t.colrec.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(e->d_rowrec[l], t.rowrec.data(), t.rowrec.size() * 4, hipMemcpyHostToDevice));
      if (!t.tileGx.empty()) {
        if ((rc = dalloc(&e->d_tileGx[l], t.tileGx.size()))) return rc;
        if ((rc = dalloc(&e->d_tileDy[l], t.tileDy.size()))) return rc;
        HIPCHK(hipMemcpy(e->d_tileGx[l], t.tileGx.data(), t.tileGx.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_tileDy[l], t.tileDy.data(), t.tileDy.size() * 4, hipMemcpyHostToDevice));
      }
    }
  }
  return ORBFE_OK;
}

int ensure_workspace(orbfe_extractor* e, int nFrames) {
  if (nFrames <= e->capFrames) return ORBFE_OK;
  free_workspace(e);
  const FrameGeom& g = e->geom;
  const size_t B = (size_t)nFrames;
  int rc;
  if ((rc = dalloc(&e->d_pyr, B * g.pyrBytes))) return rc;
  if ((rc = dalloc(&e->d_blur, B * g.pyrBytes))) return rc;
  if ((rc = dalloc(&e->d_slots, B * (size_t)g.totalSlots))) return rc;
  if ((rc = dalloc(&e->d_cand, B * (size_t)g.totalSlots))) return rc;
  if ((rc = dalloc(&e->d_cellCount, B * g.cells.size()))) return rc;
  if ((rc = dalloc(&e->d_cellPrefix, B * g.cells.size()))) return rc;
  if ((rc = dalloc(&e->d_candCount, B * (size_t)g.nlevels))) return rc;
  if ((rc = dalloc(&e->d_nodeOf, B * (size_t)g.totalSlots))) return rc;
  if ((rc = dalloc(&e->d_levelKp, B * (size_t)g.totalKpCap))) return rc;
  if ((rc = dalloc(&e->d_levelCount, B * (size_t)g.nlevels))) return rc;
  if (octree_lds_bytes(e->octreeMaxL) > kOctreeLdsLimit) {
    e->octreeWorkStride = (octree_lds_bytes(e->octreeMaxL) + 255) & ~(size_t)255;
    if ((rc = dalloc(&e->d_octreeWork, B * (size_t)g.nlevels * e->octreeWorkStride))) return rc;
  }
  e->capFrames = nFrames;
  return ORBFE_OK;
}

int ensure_outputs(orbfe_extractor* e, int nFrames, int capacity) {
  const int need = nFrames * capacity;
  if (need == e->outCap && e->d_nOut) return ORBFE_OK;  // exact layout: a small batch stays one compact block
  free_outputs(e);
  int rc;
  const size_t kpBytes = ((size_t)need * sizeof(orbfe_keypoint) + 255) & ~(size_t)255;
  const size_t descBytes = ((size_t)need * 32 + 255) & ~(size_t)255;
  const size_t cntBytes = (size_t)(nFrames > 4096 ? nFrames : 4096) * 4;
  if ((rc = dalloc(&e->d_outBlock, kpBytes + descBytes + cntBytes))) return rc;
  e->d_kpOut = reinterpret_cast<orbfe_keypoint*>(e->d_outBlock);
  e->d_descOut = e->d_outBlock + kpBytes;
  e->d_nOut = reinterpret_cast<int32_t*>(e->d_outBlock + kpBytes + descBytes);
  e->outCap = need;
  return ORBFE_OK;
}

// Stage timing with HIP events on the handle's own stream.  Events are only RECORDED inside
// the pipeline (no host sync); resolve_stage_times() reads them after the call's final
// stream synchronisation, so profiling does not perturb the timed region.
struct StageTimer {
  orbfe_extractor* e;
  int stage, sub;
  hipStream_t s;
  bool on;
  // sub = index of the sub-batch (its stream is s); every sub-batch of a call is timed on its own stream
  StageTimer(orbfe_extractor* e_, int st, int n, int frames, int sub_, hipStream_t s_)
      : e(e_), stage(st), sub(sub_), s(s_),
        on(sub_ >= 0 && sub_ < orbfe_extractor::kEvSubs && ((e_->stageMask >> st) & 1u)) {
    if (!on) return;
    (void)hipEventRecord(e->evA[e->evSlot][sub][stage], s);
    e->evLaunches[e->evSlot][sub][stage] = n;
    e->evFrames[e->evSlot][sub][stage] = frames;
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(e->evB[e->evSlot][sub][stage], s);
    e->evUsed[e->evSlot][sub][stage] = true;
  }
};
void resolve_slot(orbfe_extractor* e, int slot) {
  for (int sub = 0; sub < orbfe_extractor::kEvSubs; sub++)
    for (int st = 0; st < ORBFE_STAGE_COUNT; st++) {
      if (!e->evUsed[slot][sub][st]) continue;
      e->evUsed[slot][sub][st] = false;
      float ms = 0;
      if (hipEventSynchronize(e->evB[slot][sub][st]) == hipSuccess &&
          hipEventElapsedTime(&ms, e->evA[slot][sub][st], e->evB[slot][sub][st]) == hipSuccess) {
        e->stageMs[st] += ms;
        e->stageLaunches[st] += e->evLaunches[slot][sub][st];
        e->stageFrames[st] += e->evFrames[slot][sub][st];
      }
    }
}
void resolve_stage_times(orbfe_extractor* e) {
  for (int slot = 0; slot < orbfe_extractor::kEvRing; slot++) resolve_slot(e, slot);
}
// advance to the next ring slot before a call records into it (waits for a 16-calls-old one)
void next_event_slot(orbfe_extractor* e) {
  if (!e->stageMask) return;
  e->evSlot = (e->evSlot + 1) % orbfe_extractor::kEvRing;
  resolve_slot(e, e->evSlot);
}

// Debug cross-check only: DistributeOctTree on the host (octree_host.cpp) with a D2H/H2D round trip.
int run_host_octree(orbfe_extractor* e, int nFrames) {
  const FrameGeom& g = e->geom;
  hipStream_t s = e->stream;
  {
    const int nl = g.nlevels;
    e->h_candCount.resize((size_t)nFrames * nl);
    HIPCHK(hipMemcpyAsync(e->h_candCount.data(), e->d_candCount, sizeof(int32_t) * nFrames * nl, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<size_t> off((size_t)nFrames * nl + 1, 0);
    for (size_t i = 0; i < (size_t)nFrames * nl; i++) off[i + 1] = off[i] + (size_t)e->h_candCount[i];
    e->h_cand.resize(off.back() + 1);
    for (int f = 0; f < nFrames; f++)
      for (int l = 0; l < nl; l++) {
        const size_t i = (size_t)f * nl + l;
        if (e->h_candCount[i] > 0)
          HIPCHK(hipMemcpyAsync(e->h_cand.data() + off[i], e->d_cand + (size_t)f * g.totalSlots + g.lv[l].slotStart,
                                sizeof(Candidate) * e->h_candCount[i], hipMemcpyDeviceToHost, s));
      }
    HIPCHK(hipStreamSynchronize(s));
    e->h_levelKp.assign((size_t)nFrames * g.totalKpCap, LevelKp{0, 0, 0, 0});
    e->h_levelCount.assign((size_t)nFrames * nl, 0);
    std::atomic<int> next{0};
    const int nTasks = nFrames * nl;
    auto worker = [&]() {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= nTasks) break;
        const int f = i / nl, l = i % nl;
        const LevelGeom& lg = g.lv[l];
        const int n = distribute_octree_host(e->h_cand.data() + off[i], e->h_candCount[i], kMinBorder,
                                             lg.w - kEdgeThreshold + 3, kMinBorder, lg.h - kEdgeThreshold + 3,
                                             lg.quota, e->h_levelKp.data() + (size_t)f * g.totalKpCap + lg.kpStart, lg.kpCap);
        e->h_levelCount[i] = n > lg.kpCap ? lg.kpCap : n;
      }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int nThreads = (int)(hw ? hw : 4);
    if (nThreads > 16) nThreads = 16;
    if (nThreads > nTasks) nThreads = nTasks;
    std::vector<std::thread> pool;
    for (int i = 1; i < nThreads; i++) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    HIPCHK(hipMemcpyAsync(e->d_levelKp, e->h_levelKp.data(), sizeof(LevelKp) * e->h_levelKp.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->d_levelCount, e->h_levelCount.data(), sizeof(int32_t) * e->h_levelCount.size(), hipMemcpyHostToDevice, s));
  }
  return ORBFE_OK;
}

// The device pipeline for frames [f0, f0+nFrames) of a call, enqueued on stream `s`.
// $ORBFE_KNOCKOUT = bit mask of stages whose launches are SKIPPED once the handle has run $ORBFE_KNOCKOUT_AFTER (default 4)
// chunks in full (1 pyramid + blur, 2 FAST, 4 gather + octree, 8 orientation + descriptors, 16 stereo matcher): a timing
// experiment -- how much does a stage add to the PIPELINED step? (tools/knockout.sh; the skipped stage's outputs are
// the previous step's, so everything downstream still runs on real data; results unchecked, ORBFE_BENCH_NO_CHECK)
static int knockout_mask(orbfe_extractor* e, bool count) {
  static const int kMask = getenv("ORBFE_KNOCKOUT") ? atoi(getenv("ORBFE_KNOCKOUT")) : 0;
  static const int kAfter = getenv("ORBFE_KNOCKOUT_AFTER") ? atoi(getenv("ORBFE_KNOCKOUT_AFTER")) : 4;
  if (!kMask) return 0;
  if (count) e->knockoutChunks++;
  return e->knockoutChunks > kAfter ? kMask : 0;
}

// level0: view of the call's input frames in HBM (frame 0 of the call).
int run_chunk(orbfe_extractor* e, hipStream_t s, int sub, LevelView level0, int f0, int nFrames,
              orbfe_keypoint* d_kp, uint8_t* d_desc, int capacity, int32_t* d_nOut, PyramidViews* pyrOut,
              PyramidViews* blurOut, hipStream_t sV = nullptr, hipStream_t sT = nullptr) {
  // lanes: pyramid on s (= P), FAST + blur on sV, gather / octree / orientation + descriptors on sT; one stream
  // for everything when sV is NULL
  const bool lanes = sV != nullptr;
  if (!lanes) { sV = s; sT = s; }
  const int ko = knockout_mask(e, true);
  const FrameGeom& g = e->geom;
  const size_t F = (size_t)f0;
  const int nCells = g.nFastCells;
  const bool fused = e->fused && g.fusedBlur && e->blurSpec == kBlurSpecCv4;  // the fused phase E implements spec 0 only
  PyramidViews pyr = {}, blur = {}, pyr0 = {}, blur0 = {};
  pyr.nlevels = blur.nlevels = pyr0.nlevels = blur0.nlevels = g.nlevels;
  pyr0.lv[0] = level0;
  for (int l = 0; l < g.nlevels; l++) {
    if (l > 0) pyr0.lv[l] = LevelView{e->d_pyr + g.lv[l].off, g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
    blur0.lv[l] = LevelView{e->d_blur + g.lv[l].off, g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
    pyr.lv[l] = pyr0.lv[l];
    pyr.lv[l].base += F * pyr0.lv[l].frameStride;
    blur.lv[l] = blur0.lv[l];
    blur.lv[l].base += F * blur0.lv[l].frameStride;
  }
  if (e->copyUnaligned && ((level0.pitch & 3) || (level0.frameStride & 3) || (reinterpret_cast<uintptr_t>(level0.base) & 3))) {
    // caller-owned frames that are not dword-aligned (e.g. a tight 1241-byte stride).  Round 1 copied them into the
    // handle's own 64-B pitched level-0 slab first (an extra read + write of every input pixel); since round 2 the
    // four consumers of level 0 (resize, FAST, blur, orientation) read byte-aligned requests in place and the copy
    // is only kept behind $ORBFE_COPY_UNALIGNED=1 as a cross-check
    LevelView own{e->d_pyr + g.lv[0].off, g.pyrBytes, g.lv[0].pitch, g.lv[0].w, g.lv[0].h};
    LevelView src = level0;
    src.base += F * level0.frameStride;
    LevelViewMut dst{e->d_pyr + g.lv[0].off + F * g.pyrBytes, g.pyrBytes, g.lv[0].pitch, g.lv[0].w, g.lv[0].h};
    launch_copy2d(s, src, dst, nFrames);
    pyr0.lv[0] = own;
    pyr.lv[0] = own;
    pyr.lv[0].base += F * own.frameStride;
  }
  if (pyrOut) { *pyrOut = pyr0; *blurOut = blur0; }
  Candidate* slots = e->d_slots + F * g.totalSlots;
  Candidate* cand = e->d_cand + F * g.totalSlots;
  uint16_t* cellCount = e->d_cellCount + F * nCells;
  int32_t* cellPrefix = e->d_cellPrefix + F * nCells;
  int32_t* candCount = e->d_candCount + F * g.nlevels;
  LevelKp* levelKp = e->d_levelKp + F * g.totalKpCap;
  int32_t* levelCount = e->d_levelCount + F * g.nlevels;
  // pyrBlur: level l is blurred and level l+1 written by ONE kernel per level (the staged tile serves both), so the
  // separate blur launch below is skipped; needs the packed resize tables and the unfused FAST kernel
  // (not for the few-frame launches of a live camera: there the 8 dependent blur + resize launches are a longer critical
  // path than 7 small resize launches followed by one blur launch -- 0.195 vs 0.167 ms per single frame)
  const bool pyrBlur = e->pyrBlur && !fused && !lanes && nFrames > 8;
  // a few frames (the live camera): the whole pyramid in ONE launch -- every tile recomputes its chain from level 0 in LDS
  // (k_pyramid_chain) -- instead of n-1 dependent launches of a few microseconds each
  const bool chain = e->pyrChain && e->chainOk && !pyrBlur && !fused && nFrames <= 8 && !(ko & 1);
  if (chain) {
    StageTimer t(e, ORBFE_STAGE_PYRAMID, 1, nFrames, sub, s);
    PyrChainArgs ca = {};
    ca.l0 = pyr.lv[0];
    for (int l = 0; l < g.nlevels; l++) { ca.w[l] = g.lv[l].w; ca.h[l] = g.lv[l].h; }
    for (int l = 1; l < g.nlevels; l++) {
      ca.lv[l] = LevelViewMut{const_cast<uint8_t*>(pyr.lv[l].base), g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
      ca.xofs[l] = e->d_xofs[l]; ca.alpha[l] = e->d_alpha[l]; ca.yofs[l] = e->d_yofs[l]; ca.beta[l] = e->d_beta[l];
    }
    ca.tiles = e->d_chainTiles;
    ca.bufA = e->chainBufA; ca.bufB = e->chainBufB; ca.maxW = e->chainMaxW; ca.maxH = e->chainMaxH;
    launch_pyramid_chain(s, ca, e->nChainTiles, nFrames);
  } else
  {  // ComputePyramid, :1203-1234
    StageTimer t(e, ORBFE_STAGE_PYRAMID, pyrBlur ? g.nlevels : g.nlevels - 1, nFrames, sub, s);
    for (int l = 1; l <= g.nlevels && !(ko & 1); l++) {
      if (pyrBlur) {
        LevelViewMut bdst{const_cast<uint8_t*>(blur.lv[l - 1].base), g.pyrBytes, g.lv[l - 1].pitch, g.lv[l - 1].w, g.lv[l - 1].h};
        if (l < g.nlevels && e->d_tileGx[l]) {
          LevelViewMut next{const_cast<uint8_t*>(pyr.lv[l].base), g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
          launch_blur7_resize(s, pyr.lv[l - 1], bdst, next, e->d_colrec[l], e->d_rowrec[l], e->d_tileGx[l], e->d_tileDy[l],
                              nFrames, e->blurSpec);
          continue;
        }
        launch_blur7(s, pyr.lv[l - 1], bdst, nFrames, e->blurSpec);
      }
      if (l == g.nlevels) break;
      LevelViewMut dst{const_cast<uint8_t*>(pyr.lv[l].base), g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
      launch_resize(s, pyr.lv[l - 1], dst, e->d_xofs[l], e->d_alpha[l], e->d_yofs[l], e->d_beta[l], e->d_colrec[l],
                    e->d_rowrec[l], nFrames);
    }
  }
  // GaussianBlur of every level, :1169-1175 -- the levels are independent: one launch.  It only needs
  // the pyramid, so odd sub-batches run it BEFORE the FAST stage: neighbouring streams are then in
  // different phases (VALU-bound blur/FAST next to latency-bound octree/descriptors) instead of in step.
  if (lanes) {
    HIPCHK(hipEventRecord(e->evPyr[sub], s));
    HIPCHK(hipStreamWaitEvent(sV, e->evPyr[sub], 0));
  }
  auto do_blur = [&]() {
    StageTimer t(e, ORBFE_STAGE_BLUR, 1, nFrames, sub, sV);
    LevelViewMut dsts[kMaxLevels];
    for (int l = 0; l < g.nlevels; l++)
      dsts[l] = LevelViewMut{const_cast<uint8_t*>(blur.lv[l].base), g.pyrBytes, g.lv[l].pitch, g.lv[l].w, g.lv[l].h};
    launch_blur7_levels(sV, pyr.lv, dsts, g.nlevels, nFrames, e->blurSpec);
  };
  const bool blurFirst = !lanes && !fused && !pyrBlur && (sub & 1) != 0;  // measured +1.7 % frames/s (A/B on one box, 4 runs each)
  if (blurFirst) do_blur();
  {  // FAST grid stage, :846-896; fused: the same wavefronts also write the blurred level (:1169-1175)
    // threshold order: results are identical either way; auto follows the fallback rate of the launches that have
    // finished (hysteresis 0.40 / 0.50), so the choice lags the content by a call or two -- speed only
    if (e->fastMode == 0) {
      for (int i = 0; i < orbfe_extractor::kMaxStreams; i++)
        if (e->statPending[i] && hipEventQuery(e->evStat[i]) == hipSuccess) {
          e->statPending[i] = false;
          double hits = 0;
          for (int k = 0; k < orbfe_extractor::kStatSlots; k++) hits += e->h_fastStat[i * orbfe_extractor::kStatSlots + k];
          const double rate = e->statCells[i] > 0 ? 8.0 * hits / e->statCells[i] : 0.0;  // every 8th cell reports
          if (rate > 0.50) e->fastLowFirst = true;
          else if (rate < 0.40) e->fastLowFirst = false;
        }
      (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
    }
    const bool lowFirst = e->fastMode == 2 || (e->fastMode == 0 && e->fastLowFirst);
    unsigned int* stat = nullptr;
    // (not for the few-frame launches of a live camera: the two tiny copies and the event cost ~10 us of latency there)
    if (e->fastMode == 0 && nFrames > 8 && sub >= 0 && sub < orbfe_extractor::kMaxStreams && !e->statPending[sub]) {
      stat = e->d_fastStat + sub * orbfe_extractor::kStatSlots;
      HIPCHK(hipMemsetAsync(stat, 0, sizeof(unsigned int) * orbfe_extractor::kStatSlots, sV));
    }
    if (!(ko & 2)) {
      StageTimer t(e, ORBFE_STAGE_FAST, 1, nFrames, sub, sV);
      launch_fast_cells(sV, pyr, e->d_cells, nCells, nFrames, e->tab.iniThFAST, e->tab.minThFAST, slots, g.totalSlots,
                        cellCount, g.maxCellW, g.maxCellH, fused ? &blur : nullptr, (int)g.cells.size(), lowFirst, stat);
    }
    if (stat) {
      HIPCHK(hipMemcpyAsync(e->h_fastStat + sub * orbfe_extractor::kStatSlots, stat, sizeof(unsigned int) * orbfe_extractor::kStatSlots,
                            hipMemcpyDeviceToHost, sV));
      HIPCHK(hipEventRecord(e->evStat[sub], sV));
      e->statPending[sub] = true;
      e->statCells[sub] = (double)nCells * nFrames;
    }
  }
  if (lanes) {  // the tail lane starts on the candidates while the VALU lane goes on with the blur
    HIPCHK(hipEventRecord(e->evFast[sub], sV));
    HIPCHK(hipStreamWaitEvent(sT, e->evFast[sub], 0));
    if (!fused && !pyrBlur) do_blur();
    HIPCHK(hipEventRecord(e->evBlur[sub], sV));
  }
  if (ko & 4) {
  } else if (!e->hostOctree) {  // candidate ordering + DistributeOctTree, :566-808, one workgroup per (frame, level)
    // (round 4, measured and not kept: the candidate ordering INSIDE the octree kernel, in front of each (frame, level) item --
    // one launch and one global round trip fewer per chain; KITTI 100.6 k vs 101.4 k, TUM 361.9 k vs 360.1 k stereo frames /
    // frames per second for separate vs fused in a same-box A/B: the launch is not what the chain waits for)
    StageTimer t(e, ORBFE_STAGE_OCTREE, 2, nFrames, sub, sT);
    const bool octGathers = octree_gathers(nFrames, e->octreeMaxL);  // a few frames: the octree workgroups gather their own level
    if (!octGathers)
      launch_gather_candidates(sT, e->d_cells, e->d_lvgeom, g.nlevels, nFrames, slots, g.totalSlots, cellCount,
                               nCells, cand, candCount, cellPrefix);
    OctreeArgs oa = {};
    if (octGathers) {
      oa.gCells = e->d_cells; oa.gSlots = slots; oa.gCellCount = cellCount; oa.gCellsPerFrame = nCells; oa.gCellPrefix = cellPrefix;
      oa.gCand = cand; oa.gCandCount = candCount;
    }
    oa.cand = cand;
    oa.slotsPerFrame = g.totalSlots;
    oa.candCount = candCount;
    oa.lvg = e->d_lvgeom;
    oa.nlevels = g.nlevels;
    oa.nodeOf = e->d_nodeOf + F * g.totalSlots;
    oa.levelKp = levelKp;
    oa.levelCount = levelCount;
    oa.kpSlotsPerFrame = g.totalKpCap;
    oa.maxL = e->octreeMaxL;
    oa.work = e->d_octreeWork ? e->d_octreeWork + F * (size_t)g.nlevels * e->octreeWorkStride : nullptr;
    oa.workStride = e->octreeWorkStride;
    HIPCHK(launch_octree(sT, oa, g.nlevels, nFrames));
  } else {
    launch_gather_candidates(s, e->d_cells, e->d_lvgeom, g.nlevels, nFrames, slots, g.totalSlots, cellCount,
                             nCells, cand, candCount, cellPrefix);
    int rc = run_host_octree(e, nFrames);  // single-stream debug path: f0 == 0
    if (rc) return rc;
  }
  if (!lanes && !blurFirst && !fused && !pyrBlur) do_blur();
  if (lanes) HIPCHK(hipStreamWaitEvent(sT, e->evBlur[sub], 0));

  if (!(ko & 8)) {  // computeOrientation + computeDescriptors + output records
    StageTimer t(e, ORBFE_STAGE_ORIENT_DESC, 1, nFrames, sub, sT);
    OrientDescArgs a = {};
    a.pyr = pyr;
    a.blur = blur;
    a.nlevels = g.nlevels;
    a.kpSlotsPerFrame = g.totalKpCap;
    a.outCapacity = capacity;
    for (int l = 0; l < g.nlevels; l++) {
      a.kpStart[l] = g.lv[l].kpStart;
      a.kpCap[l] = g.lv[l].kpCap;
      a.scale[l] = e->tab.scale[l];
      a.kpSize[l] = (float)(int)(kPatchSize * e->tab.scale[l]);  // :905
    }
    // Tile form (k_desc_tiles.hip, round 3): a third of the HBM / L2 traffic of the per-keypoint gathers, but 1.6x their
    // VALU instructions (every tile pays its staging and list scan for ~19 keypoints) -- on this instruction-bound
    // pipeline it measures 2 % slower, so it is an option (orbfe_extractor_set_desc_tiles / $ORBFE_DESC_TILES=1), not
    // the default; DESIGN.md 4 has the counters.
    static const int kTilesEnv = getenv("ORBFE_DESC_TILES") ? atoi(getenv("ORBFE_DESC_TILES")) : 0;
    const int tilesMode = e->descTilesMode >= 0 ? e->descTilesMode : kTilesEnv;
    if (tilesMode && e->nDescTiles > 0)
      launch_orient_desc_tiles(sT, a, e->d_descTiles, e->nDescTiles, levelKp, levelCount, e->d_patternF, e->d_momentTab,
                               e->d_umax, nFrames, d_kp + F * capacity, d_desc + F * (size_t)capacity * 32, d_nOut + F, e->lastS);
    else
      launch_orient_desc(sT, a, levelKp, levelCount, e->d_patternF, e->d_momentTab, e->d_umax, nFrames,
                         d_kp + F * capacity, d_desc + F * (size_t)capacity * 32, d_nOut + F, e->lastS);
  }
  if (lanes) {
    HIPCHK(hipEventRecord(e->evTail[sub], sT));
    e->tailPending[sub] = true;
  }
  HIPCHK(hipGetLastError());
  return ORBFE_OK;
}

// One call = up to nStreams sub-batches of consecutive frames, each on its own HIP stream with
// its own slice of the workspace: the VALU-bound kernels (FAST, blur) of one sub-batch overlap the
// gather/latency-bound ones (orientation+descriptor, octree, resize) of the other.
// `pre`: work enqueued on a sub-batch's stream in FRONT of its kernels (frames [f0, f0 + n) of the batch), e.g. the
// rectification that produces those frames -- stream order is the only synchronisation it needs, and what the previous
// call left on that stream (its kernels and matchers, which read the same frame slots) is behind it by construction.
struct PreChunk {
  int (*fn)(void* ctx, hipStream_t s, int f0, int n) = nullptr;
  void* ctx = nullptr;
};

int run_pipeline(orbfe_extractor* e, LevelView level0, int nFrames, orbfe_keypoint* d_kp,
                 uint8_t* d_desc, int capacity, int32_t* d_nOut, const hipEvent_t* waitFor = nullptr, int nWait = 0,
                 const PreChunk* pre = nullptr) {
  int S = e->hostOctree ? 1 : e->nStreams;
  if (S > nFrames) S = nFrames;
  if (S < 1) S = 1;
  if (e->lastSplitFrames != nFrames || e->lastSplitStreams != (e->laneMode ? -S : S)) {
    // a different split maps frames to different streams: drain everything first
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < orbfe_extractor::kMaxStreams - 1; i++)
      if (e->extra[i]) HIPCHK(hipStreamSynchronize(e->extra[i]));
    e->lastSplitFrames = nFrames;
    e->lastSplitStreams = e->laneMode ? -S : S;
  }
  int per = (nFrames + S - 1) / S;
  if ((nFrames & 1) == 0 && (per & 1)) per++;  // stereo batches (L0,R0,L1,R1,...): never split a pair across sub-batches
  const bool lanes = e->laneMode && !e->hostOctree && S >= 2;
  e->lastPer = per;
  e->lastS = S;
  e->lastLanes = lanes;
  if (lanes) {
    hipStream_t sP = e->extra[0], sV = e->extra[1], sT = e->stream;
    if (e->consumerPending) HIPCHK(hipStreamWaitEvent(sP, e->evConsumerDone, 0));  // a matcher still reads the last pyramid
    for (int k = 0; k < nWait; k++) {  // e.g. the H2D copy of this chunk (read by P, V and T), the D2H of its output block (T)
      HIPCHK(hipStreamWaitEvent(sP, waitFor[k], 0));
      HIPCHK(hipStreamWaitEvent(sT, waitFor[k], 0));
    }
    for (int i = 0; i < S; i++) {
      const int f0 = i * per;
      const int n = f0 + per <= nFrames ? per : nFrames - f0;
      if (n <= 0) break;
      // slice i of the workspace is free once the previous call's tail lane has finished with it
      if (e->tailPending[i]) HIPCHK(hipStreamWaitEvent(sP, e->evTail[i], 0));
      if (pre && pre->fn) {  // (slice i's readers of the previous call: behind evTail[i] / evConsumerDone, waited for above)
        StageTimer t(e, ORBFE_STAGE_H2D, 2, n, i, sP);  // "h2d" = the ingest stage of a call: here the rectification
        int rcp = pre->fn(pre->ctx, sP, f0, n);
        if (rcp) return rcp;
      }
      int rc = run_chunk(e, sP, i, level0, f0, n, d_kp, d_desc, capacity, d_nOut, i == 0 ? &e->lastPyr : nullptr,
                         i == 0 ? &e->lastBlur : nullptr, sV, sT);
      if (rc) return rc;
    }
    e->consumerPending = false;
    e->chunksPending = 1;  // every sub-batch ends on `stream`: consumers there are ordered behind all of them
    e->lastFrames = nFrames;
    e->lastFrameBase = 0;
    e->haveLast = true;
    return ORBFE_OK;
  }
  for (int i = 0; i < orbfe_extractor::kMaxStreams; i++) e->tailPending[i] = false;  // (drained above if the mode changed)
  for (int i = 0; i < S; i++) {
    const int f0 = i * per;
    const int n = f0 + per <= nFrames ? per : nFrames - f0;
    if (n <= 0) break;
    hipStream_t s = i == 0 ? e->stream : e->extra[i - 1];
    if (i > 0 && e->consumerPending) HIPCHK(hipStreamWaitEvent(s, e->evConsumerDone, 0));
    for (int k = 0; k < nWait; k++) HIPCHK(hipStreamWaitEvent(s, waitFor[k], 0));  // e.g. the H2D copy of this chunk
    if (pre && pre->fn) {
      StageTimer t(e, ORBFE_STAGE_H2D, 2, n, i, s);  // "h2d" = the ingest stage of a call: here the rectification
      int rcp = pre->fn(pre->ctx, s, f0, n);
      if (rcp) return rcp;
    }
    int rc = run_chunk(e, s, i, level0, f0, n, d_kp, d_desc, capacity, d_nOut, i == 0 ? &e->lastPyr : nullptr,
                       i == 0 ? &e->lastBlur : nullptr);
    if (rc) return rc;
    if (i > 0) HIPCHK(hipEventRecord(e->evChunkDone[i], s));
  }
  e->consumerPending = false;  // stream 0 is ordered behind the consumer by itself
  e->chunksPending = S;
  e->lastFrames = nFrames;
  e->lastFrameBase = 0;
  e->haveLast = true;
  return ORBFE_OK;
}

int sync_all(orbfe_extractor* e) {
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < orbfe_extractor::kMaxStreams - 1; i++)
    if (e->extra[i]) HIPCHK(hipStreamSynchronize(e->extra[i]));
  return ORBFE_OK;
}

}  // namespace

extern "C" int orbfe_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                                      int minThFAST, int device, orbfe_extractor** out) {
  if (!out) return fail(ORBFE_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (nfeatures < 0 || nlevels < 1 || nlevels > ORBFE_MAX_LEVELS || !(scaleFactor > 1.0f))
    return fail(ORBFE_ERR_INVALID, "bad extractor parameters");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(ORBFE_ERR_HIP, "no such HIP device");
  HIPCHK(hipSetDevice(device));
  orbfe_extractor* e = new (std::nothrow) orbfe_extractor();
  if (!e) return fail(ORBFE_ERR_NOMEM, "out of memory");
  e->device = device;
  e->tab.init(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
  hipError_t err = stream_get(device, 0, &e->stream);
  if (const char* env = getenv("ORBFE_COPY_UNALIGNED")) e->copyUnaligned = atoi(env) != 0;
  if (const char* env = getenv("ORBFE_FAST_MODE")) {
    const std::string v(env);
    e->fastMode = v == "high" ? 1 : (v == "low" ? 2 : 0);
  }
  if (err == hipSuccess) err = hipMalloc((void**)&e->d_fastStat, sizeof(unsigned int) * orbfe_extractor::kMaxStreams * orbfe_extractor::kStatSlots);
  if (err == hipSuccess)
    err = hipHostMalloc((void**)&e->h_fastStat, sizeof(unsigned int) * orbfe_extractor::kMaxStreams * orbfe_extractor::kStatSlots, hipHostMallocDefault);
  if (const char* env = getenv("ORBFE_LANES")) e->laneMode = atoi(env) != 0;
  if (const char* env = getenv("ORBFE_FUSED")) e->fused = atoi(env) != 0;
  if (const char* env = getenv("ORBFE_PYRBLUR")) e->pyrBlur = atoi(env) != 0;
  if (const char* env = getenv("ORBFE_PYR_CHAIN")) e->pyrChain = atoi(env) != 0;
  if (const char* env = getenv("ORBFE_BLUR_SPEC")) {
    const int v = atoi(env);
    if (v >= 0 && v <= 2) e->blurSpec = v;
  }
  if (const char* env = getenv("ORBFE_STREAMS")) {
    int v = atoi(env);
    if (v >= 1 && v <= orbfe_extractor::kMaxStreams) e->nStreams = v;
  }
  if (err == hipSuccess) err = hipEventCreateWithFlags(&e->evConsumerDone, hipEventDisableTiming);
  if (err == hipSuccess && ensure_subs(e, e->nStreams > 3 ? e->nStreams : 3) != ORBFE_OK) err = hipErrorUnknown;  // 3: the lane schedule's P / V / T
  float patF[1024];
  // test i = (x0, y0, x1, y1) in the table; uploaded as (x0, x1, y0, y1) so that k_orient_desc rotates the
  // two points of a test in one packed-fp32 operation per product
  for (int i = 0; i < 256; i++) {
    patF[4 * i + 0] = (float)kOrbBitPattern31[4 * i + 0];
    patF[4 * i + 1] = (float)kOrbBitPattern31[4 * i + 2];
    patF[4 * i + 2] = (float)kOrbBitPattern31[4 * i + 1];
    patF[4 * i + 3] = (float)kOrbBitPattern31[4 * i + 3];
  }
  uint8_t momTab[1024];
  build_moment_table(momTab);
  if (err == hipSuccess) err = hipMalloc((void**)&e->d_patternF, sizeof(patF));
  if (err == hipSuccess) err = hipMemcpy(e->d_patternF, patF, sizeof(patF), hipMemcpyHostToDevice);
  float scTab[2 * kMaxLevels] = {};
  for (int i = 0; i < e->tab.nlevels; i++) { scTab[i] = e->tab.scale[i]; scTab[kMaxLevels + i] = e->tab.invScale[i]; }
  if (err == hipSuccess) err = hipMalloc((void**)&e->d_scaleTab, sizeof(scTab));
  if (err == hipSuccess) err = hipMemcpy(e->d_scaleTab, scTab, sizeof(scTab), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMalloc((void**)&e->d_momentTab, sizeof(momTab));
  if (err == hipSuccess) err = hipMemcpy(e->d_momentTab, momTab, sizeof(momTab), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMalloc((void**)&e->d_umax, 16 * sizeof(int32_t));
  if (err == hipSuccess) err = hipMemcpy(e->d_umax, e->tab.umax, 16 * sizeof(int32_t), hipMemcpyHostToDevice);
  if (err != hipSuccess) {
    orbfe_extractor_destroy(e);
    return fail(ORBFE_ERR_HIP, std::string("extractor_create: ") + hipGetErrorString(err));
  }
  *out = e;
  return ORBFE_OK;
}

extern "C" void orbfe_extractor_destroy(orbfe_extractor* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (int i = 0; i < orbfe_extractor::kMaxStreams - 1; i++)
    if (e->extra[i]) stream_put(e->device, i + 1, e->extra[i]);
  free_geometry(e);
  free_workspace(e);
  free_outputs(e);
  if (e->h_outStage) (void)hipHostFree(e->h_outStage);
  if (e->h_fastStat) (void)hipHostFree(e->h_fastStat);
  dfree(&e->d_fastStat);
  for (int i = 0; i < orbfe_extractor::kMaxStreams; i++)
    if (e->evStat[i]) (void)hipEventDestroy(e->evStat[i]);
  for (int i = 0; i < 2; i++) {
    dfree(&e->d_pipeIn[i]);
    dfree(&e->d_pipeOut[i]);
    if (e->evIn[i]) (void)hipEventDestroy(e->evIn[i]);
    if (e->evComp[i]) (void)hipEventDestroy(e->evComp[i]);
    if (e->evOutDone[i]) (void)hipEventDestroy(e->evOutDone[i]);
  }
  stream_put(e->device, kRoleH2D, e->sH2D);
  stream_put(e->device, kRoleD2H, e->sD2H);
  dfree(&e->d_patternF);
  dfree(&e->d_frameStereo);
  dfree(&e->d_stereoSad);
  dfree(&e->d_scaleTab);
  dfree(&e->d_stereoRowStart);
  dfree(&e->d_stereoSorted);
  dfree(&e->d_stereoRec);
  dfree(&e->d_momentTab);
  dfree(&e->d_hostIn);
  dfree(&e->d_umax);
  for (int r = 0; r < orbfe_extractor::kEvRing; r++)
    for (int u = 0; u < orbfe_extractor::kEvSubs; u++)
      for (int i = 0; i < ORBFE_STAGE_COUNT; i++) {
        if (e->evA[r][u][i]) (void)hipEventDestroy(e->evA[r][u][i]);
        if (e->evB[r][u][i]) (void)hipEventDestroy(e->evB[r][u][i]);
      }
  for (int i = 0; i < orbfe_extractor::kMaxStreams; i++) {
    if (e->evChunkDone[i]) (void)hipEventDestroy(e->evChunkDone[i]);
    if (e->evPyr[i]) (void)hipEventDestroy(e->evPyr[i]);
    if (e->evFast[i]) (void)hipEventDestroy(e->evFast[i]);
    if (e->evBlur[i]) (void)hipEventDestroy(e->evBlur[i]);
    if (e->evTail[i]) (void)hipEventDestroy(e->evTail[i]);
  }
  if (e->evConsumerDone) (void)hipEventDestroy(e->evConsumerDone);
  stream_put(e->device, 0, e->stream);
  delete e;
}

extern "C" int orbfe_extractor_get_levels(const orbfe_extractor* e) { return e ? e->tab.nlevels : 0; }
extern "C" float orbfe_extractor_get_scale_factor(const orbfe_extractor* e) { return e ? (float)e->tab.scaleFactor : 0.f; }
#define GETTER(name, field)                                                       \
  extern "C" int name(const orbfe_extractor* e, float* out) {                     \
    if (!e || !out) return fail(ORBFE_ERR_INVALID, #name ": NULL argument");      \
    for (int i = 0; i < e->tab.nlevels; i++) out[i] = e->tab.field[i];            \
    return ORBFE_OK;                                                              \
  }
GETTER(orbfe_extractor_get_scale_factors, scale)
GETTER(orbfe_extractor_get_inverse_scale_factors, invScale)
GETTER(orbfe_extractor_get_scale_sigma_squares, sigma2)
GETTER(orbfe_extractor_get_inverse_scale_sigma_squares, invSigma2)
extern "C" int orbfe_extractor_get_features_per_level(const orbfe_extractor* e, int32_t* out) {
  if (!e || !out) return fail(ORBFE_ERR_INVALID, "NULL argument");
  for (int i = 0; i < e->tab.nlevels; i++) out[i] = e->tab.quota[i];
  return ORBFE_OK;
}
extern "C" int orbfe_extractor_get_umax(const orbfe_extractor* e, int32_t* out16) {
  if (!e || !out16) return fail(ORBFE_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 16; i++) out16[i] = e->tab.umax[i];
  return ORBFE_OK;
}
extern "C" int orbfe_extractor_max_keypoints(const orbfe_extractor* e) {
  if (!e) return 0;
  // quota + 3 per level, or 4 root children per level on very wide images (nIni <= 16 assumed)
  int n = 0;
  for (int l = 0; l < e->tab.nlevels; l++) n += (e->tab.quota[l] + 3 > 64 ? e->tab.quota[l] + 3 : 64);
  return n;
}
extern "C" int orbfe_extractor_max_keypoints_for(const orbfe_extractor* e, int width, int height) {
  if (!e || width <= 0 || height <= 0) return 0;
  if (e->geom.W == width && e->geom.H == height && e->geom.totalKpCap > 0) return e->geom.totalKpCap;  // current geometry
  FrameGeom g;  // the exact bound of the pipeline for this image size: per level max(quota + 3, 4 * nIni)
  g.build(e->tab, width, height);
  return g.totalKpCap;
}
extern "C" int orbfe_extractor_level_size(const orbfe_extractor* e, int width, int height, int level, int* w, int* h) {
  if (!e || !w || !h || level < 0 || level >= e->tab.nlevels) return fail(ORBFE_ERR_INVALID, "bad argument");
  const float s = e->tab.invScale[level];
  *w = cv_round((float)width * s);
  *h = cv_round((float)height * s);
  return ORBFE_OK;
}

extern "C" int orbfe_extract_batch_device_async(orbfe_extractor* e, const uint8_t* d_images, int n_frames,
                                                int width, int height, int stride, size_t frame_stride,
                                                orbfe_keypoint* d_keypoints, uint8_t* d_descriptors,
                                                int capacity, int32_t* d_n_out) {
  if (!e || !d_keypoints || !d_descriptors || !d_n_out || capacity <= 0 || n_frames < 0)
    return fail(ORBFE_ERR_INVALID, "extract_batch_device: bad argument");
  if (n_frames == 0) return ORBFE_OK;
  if (!d_images || width <= 0 || height <= 0 || stride < width)
    return fail(ORBFE_ERR_INVALID, "extract_batch_device: bad image");
  if (frame_stride < (size_t)stride * (size_t)height)  // rows are staged up to the PITCH (16-byte requests): see orbfe.h
    return fail(ORBFE_ERR_INVALID, "extract_batch_device: frame_stride < stride * height (every frame must be readable for stride * height bytes)");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if (e->geom.W != width || e->geom.H != height || n_frames > e->capFrames) {
    int rcs = sync_all(e);  // the workspace is about to be re-allocated
    if (rcs) return rcs;
  }
  e->outLastFrames = 0;  // "the frame this handle produced last" is no longer the one in its own output block
  if ((rc = ensure_geometry(e, width, height))) return rc;
  if ((rc = ensure_workspace(e, n_frames))) return rc;
  next_event_slot(e);
  LevelView l0{d_images, frame_stride, stride, width, height};
  return run_pipeline(e, l0, n_frames, d_keypoints, d_descriptors, capacity, d_n_out);
}

// Examples/Stereo/stereo_euroc.cc:136-137 (cv::remap of the left and of the right image) in front of the two ExtractORB
// calls of the stereo Frame constructor (src/Frame.cc:78-81), for a device-resident batch of raw pairs: sub-batch by
// sub-batch, on the sub-batch's own stream -- rectified frames are written interleaved (L0, R0, L1, R1, ...), the layout
// orbfe_stereo_match_batch_device reads.
extern "C" int orbfe_remap_launch_(orbfe_rectifier* r, const uint8_t* d_src, int n_frames, int sw, int sh, int sstride,
                                   size_t sFrame, uint8_t* d_dst, int dstride, size_t dFrame, hipStream_t stream, int* w, int* h,
                                   int* device);
namespace {
struct RectifyCtx {
  orbfe_rectifier *rl, *rr;
  const uint8_t *rawL, *rawR;
  int sw, sh, sstride;
  size_t sFrame;
  uint8_t* rect;
  int w, h;
};
int rectify_chunk(void* c, hipStream_t s, int f0, int n) {
  const RectifyCtx* x = static_cast<const RectifyCtx*>(c);
  if ((f0 & 1) || (n & 1)) return fail(ORBFE_ERR_INVALID, "extract_stereo_rectified: a sub-batch splits a stereo pair");
  const size_t fr = (size_t)x->w * x->h;
  const int p0 = f0 / 2, np = n / 2;
  int rc = orbfe_remap_launch_(x->rl, x->rawL + (size_t)p0 * x->sFrame, np, x->sw, x->sh, x->sstride, x->sFrame,
                               x->rect + (size_t)f0 * fr, x->w, 2 * fr, s, nullptr, nullptr, nullptr);
  if (rc) return rc;
  return orbfe_remap_launch_(x->rr, x->rawR + (size_t)p0 * x->sFrame, np, x->sw, x->sh, x->sstride, x->sFrame,
                             x->rect + (size_t)(f0 + 1) * fr, x->w, 2 * fr, s, nullptr, nullptr, nullptr);
}
}  // namespace

extern "C" int orbfe_extract_stereo_rectified_batch_device_async(
    orbfe_extractor* e, orbfe_rectifier* rect_left, orbfe_rectifier* rect_right, const uint8_t* d_raw_left,
    const uint8_t* d_raw_right, int n_pairs, int src_width, int src_height, int src_stride, size_t src_frame_stride,
    uint8_t* d_rectified, orbfe_keypoint* d_keypoints, uint8_t* d_descriptors, int capacity, int32_t* d_n_out) {
  if (!e || !rect_left || !rect_right || !d_keypoints || !d_descriptors || !d_n_out || capacity <= 0 || n_pairs < 0)
    return fail(ORBFE_ERR_INVALID, "extract_stereo_rectified: bad argument");
  if (n_pairs == 0) return ORBFE_OK;
  if (!d_raw_left || !d_raw_right || !d_rectified || src_width <= 0 || src_height <= 0 || src_stride < src_width ||
      src_frame_stride < (size_t)src_stride * (src_height - 1) + src_width)
    return fail(ORBFE_ERR_INVALID, "extract_stereo_rectified: bad image");
  int wl, hl, dl, wr, hr, dr, rc;
  if ((rc = orbfe_remap_launch_(rect_left, nullptr, 0, 0, 0, 0, 0, nullptr, 0, 0, nullptr, &wl, &hl, &dl))) return rc;
  if ((rc = orbfe_remap_launch_(rect_right, nullptr, 0, 0, 0, 0, 0, nullptr, 0, 0, nullptr, &wr, &hr, &dr))) return rc;
  if (wl != wr || hl != hr || dl != e->device || dr != e->device)
    return fail(ORBFE_ERR_INVALID, "extract_stereo_rectified: the two rectifiers differ in size or device from each other / the extractor");
  HIPCHK(hipSetDevice(e->device));
  const int n_frames = 2 * n_pairs;
  if (e->geom.W != wl || e->geom.H != hl || n_frames > e->capFrames) {
    int rcs = sync_all(e);  // the workspace is about to be re-allocated
    if (rcs) return rcs;
  }
  if ((rc = ensure_geometry(e, wl, hl))) return rc;
  if ((rc = ensure_workspace(e, n_frames))) return rc;
  next_event_slot(e);
  RectifyCtx ctx{rect_left, rect_right, d_raw_left, d_raw_right, src_width, src_height, src_stride, src_frame_stride,
                 d_rectified, wl, hl};
  PreChunk pre;
  pre.fn = rectify_chunk;
  pre.ctx = &ctx;
  LevelView l0{d_rectified, (size_t)wl * hl, wl, wl, hl};
  return run_pipeline(e, l0, n_frames, d_keypoints, d_descriptors, capacity, d_n_out, nullptr, 0, &pre);
}

extern "C" int orbfe_extractor_synchronize(orbfe_extractor* e) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  resolve_stage_times(e);
  return ORBFE_OK;
}

extern "C" int orbfe_extract_batch_device(orbfe_extractor* e, const uint8_t* d_images, int n_frames,
                                          int width, int height, int stride, size_t frame_stride,
                                          orbfe_keypoint* d_keypoints, uint8_t* d_descriptors,
                                          int capacity, int32_t* d_n_out) {
  int rc = orbfe_extract_batch_device_async(e, d_images, n_frames, width, height, stride, frame_stride,
                                            d_keypoints, d_descriptors, capacity, d_n_out);
  if (rc) return rc;
  return orbfe_extractor_synchronize(e);
}

extern "C" int orbfe_extract_batch_pipelined(orbfe_extractor* e, const uint8_t* images, int n_frames, int width, int height,
                                             int stride, size_t frame_stride, orbfe_keypoint* keypoints,
                                             uint8_t* descriptors, int capacity, int* n_out, int chunk_frames);

extern "C" int orbfe_extract_batch(orbfe_extractor* e, const uint8_t* images, int n_frames, int width,
                                   int height, int stride, size_t frame_stride,
                                   orbfe_keypoint* keypoints, uint8_t* descriptors, int capacity,
                                   int* n_out) {
  if (!e || !n_out || n_frames < 0) return fail(ORBFE_ERR_INVALID, "extract_batch: bad argument");
  for (int f = 0; f < n_frames; f++) n_out[f] = 0;
  if (n_frames == 0) return ORBFE_OK;
  if (!images || width <= 0 || height <= 0) return ORBFE_OK;  // empty image: silent return (:1122)
  if (!keypoints || !descriptors || capacity <= 0 || stride < width)
    return fail(ORBFE_ERR_INVALID, "extract_batch: bad output buffers");
  bool pinnedIn = false;
  {
    hipPointerAttribute_t at;
    pinnedIn = hipPointerGetAttributes(&at, images) == hipSuccess && at.type != hipMemoryTypeUnregistered;
    (void)hipGetLastError();
  }
  if (!e->hostOctree && frame_stride >= (size_t)stride * (height - 1) + width && (pinnedIn ? n_frames >= 32 : n_frames >= 512))
    // a real batch: chunked H2D / kernels / D2H overlapped on separate streams instead of upload-all, compute,
    // download-all.  Pageable buffers would have to be page-locked for the call, which costs milliseconds (measured
    // 5-9 ms for 128 VGA frames) and only pays for very large batches; pinned ones (orbfe_host_alloc) always pay.
    return orbfe_extract_batch_pipelined(e, images, n_frames, width, height, stride, frame_stride, keypoints, descriptors,
                                         capacity, n_out, 0);
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = sync_all(e))) return rc;  // an earlier asynchronous call may still use the workspace
  e->outLastFrames = 0;
  if ((rc = ensure_geometry(e, width, height))) return rc;
  if ((rc = ensure_workspace(e, n_frames))) return rc;
  if ((rc = ensure_outputs(e, n_frames, capacity))) return rc;
  const FrameGeom& g = e->geom;
  next_event_slot(e);
  LevelView l0{e->d_pyr + g.lv[0].off, g.pyrBytes, g.lv[0].pitch, width, height};
  const bool tight = frame_stride == (size_t)stride * height || n_frames == 1;
  {
    StageTimer t(e, ORBFE_STAGE_H2D, 0, 0, 0, e->stream);
    if (tight && stride != g.lv[0].pitch) {
      // ONE linear copy of the frames as they lie in host memory into an input slab of the caller's pitch, which the
      // kernels read in place at any stride (like caller-owned device frames).  A 2-D copy of rows that are not a
      // multiple of the DMA granule degenerates into one descriptor per row: 3.0 ms for ONE 1241 x 376 frame (0.15 GB/s)
      // against 0.1 ms this way -- the live-camera latency of a KITTI frame was that copy.
      const size_t bytes = (size_t)(n_frames - 1) * frame_stride + (size_t)(height - 1) * stride + width;
      // the slab is readable for stride * height bytes per frame: the kernels stage rows up to the pitch, not the width
      const size_t slab = (size_t)(n_frames - 1) * frame_stride + (size_t)stride * height + 64;
      if (slab > e->hostInBytes) {
        // dalloc frees the old slab first: until the new one exists nothing may still point at it (the last result's
        // level-0 view lives in it: get_pyramid_level / compute_stereo_matches of the PREVIOUS call would read freed memory)
        e->hostInBytes = 0;
        e->haveLast = false;
        if ((rc = dalloc(&e->d_hostIn, slab))) return rc;
        e->hostInBytes = slab;
      }
      HIPCHK(hipMemcpyAsync(e->d_hostIn, images, bytes, hipMemcpyHostToDevice, e->stream));
      l0 = LevelView{e->d_hostIn, frame_stride, stride, width, height};
    } else {
      // level 0 lives at the head of the per-frame pyramid slab (pitch-aligned copy)
      for (int f = 0; f < n_frames; f++)
        HIPCHK(hipMemcpy2DAsync(e->d_pyr + (size_t)f * g.pyrBytes + g.lv[0].off, g.lv[0].pitch,
                                images + (size_t)f * frame_stride, stride, width, height,
                                hipMemcpyHostToDevice, e->stream));
    }
  }
  // (the sub-batch streams read the uploaded frames.  Round 4 measured a single frame -- everything on e->stream, in order --
  // without this wait and without the one behind the kernels: 0.206 / 0.168 ms per KITTI / VGA frame against 0.205 / 0.158
  // with them, same box, three alternations: the waits are not what a frame's 60 us outside its kernels is made of)
  HIPCHK(hipStreamSynchronize(e->stream));
  if ((rc = run_pipeline(e, l0, n_frames, e->d_kpOut, e->d_descOut, capacity, e->d_nOut))) return rc;
  if ((rc = sync_all(e))) return rc;
  {
    StageTimer t(e, ORBFE_STAGE_D2H, 0, 0, 0, e->stream);
    bool overflow = false;
    const size_t blockBytes = (size_t)(reinterpret_cast<uint8_t*>(e->d_nOut) - e->d_outBlock) + sizeof(int32_t) * (size_t)n_frames;
    if (blockBytes <= (size_t)512 * 1024) {
      // small batch (the live-camera case): everything in ONE copy into pinned memory, then scattered
      // by the CPU -- one DMA + one synchronisation instead of 1 + 2*n_frames copies and two syncs
      if (e->outStageBytes < blockBytes) {
        if (e->h_outStage) (void)hipHostFree(e->h_outStage);
        e->h_outStage = nullptr;
        e->outStageBytes = 0;
        HIPCHK(hipHostMalloc((void**)&e->h_outStage, (size_t)512 * 1024, hipHostMallocDefault));
        e->outStageBytes = (size_t)512 * 1024;
      }
      HIPCHK(hipMemcpyAsync(e->h_outStage, e->d_outBlock, blockBytes, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      const uint8_t* hk = e->h_outStage;
      const uint8_t* hd = e->h_outStage + (e->d_descOut - e->d_outBlock);
      const int32_t* hc = reinterpret_cast<const int32_t*>(e->h_outStage + (reinterpret_cast<uint8_t*>(e->d_nOut) - e->d_outBlock));
      for (int f = 0; f < n_frames; f++) {
        int n = hc[f];
        if (n > capacity) { overflow = true; n = capacity; }
        n_out[f] = n;
        if (n > 0) {
          std::memcpy(keypoints + (size_t)f * capacity, hk + (size_t)f * capacity * sizeof(orbfe_keypoint), sizeof(orbfe_keypoint) * (size_t)n);
          std::memcpy(descriptors + (size_t)f * capacity * 32, hd + (size_t)f * capacity * 32, (size_t)n * 32);
        }
      }
    } else {
      std::vector<int32_t> cnt(n_frames);
      HIPCHK(hipMemcpyAsync(cnt.data(), e->d_nOut, sizeof(int32_t) * n_frames, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int f = 0; f < n_frames; f++) {
        int n = cnt[f];
        if (n > capacity) { overflow = true; n = capacity; }
        n_out[f] = n;
        if (n > 0) {
          HIPCHK(hipMemcpyAsync(keypoints + (size_t)f * capacity, e->d_kpOut + (size_t)f * capacity,
                                sizeof(orbfe_keypoint) * n, hipMemcpyDeviceToHost, e->stream));
          HIPCHK(hipMemcpyAsync(descriptors + (size_t)f * capacity * 32, e->d_descOut + (size_t)f * capacity * 32,
                                (size_t)n * 32, hipMemcpyDeviceToHost, e->stream));
        }
      }
      HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (overflow) return fail(ORBFE_ERR_CAPACITY, "keypoint capacity too small");
  }
  e->outLastCount.assign(n_out, n_out + n_frames);
  e->outLastCapacity = capacity;
  e->outLastFrames = n_frames;
  resolve_stage_times(e);
  return ORBFE_OK;
}

// internal (matcher.hip, orbfe_frame_from_extractor): device records of frame `frame` of the last host-buffer call
extern "C" int orbfe_extractor_output_device_(orbfe_extractor* e, int frame, const orbfe_keypoint** d_kp, const uint8_t** d_desc,
                                              int* n, int* device) {
  if (!e || !d_kp || !d_desc || !n || !device) return fail(ORBFE_ERR_INVALID, "frame_from_extractor: NULL argument");
  if (e->outLastFrames <= 0)
    return fail(ORBFE_ERR_INVALID, "frame_from_extractor: the handle holds no output block (the last call was not orbfe_extract / a small "
                                   "orbfe_extract_batch; device-batch callers own their outputs: orbfe_frame_from_device)");
  if (frame < 0 || frame >= e->outLastFrames) return fail(ORBFE_ERR_INVALID, "frame_from_extractor: frame out of range");
  *d_kp = e->d_kpOut + (size_t)frame * e->outLastCapacity;
  *d_desc = e->d_descOut + (size_t)frame * e->outLastCapacity * 32;
  *n = e->outLastCount[(size_t)frame];
  *device = e->device;
  return ORBFE_OK;
}

// ---- pinned host memory + the pipelined host-batch path (end-to-end form of operator()) ----
extern "C" int orbfe_host_alloc(void** p, size_t bytes) {
  if (!p) return fail(ORBFE_ERR_INVALID, "host_alloc: NULL argument");
  *p = nullptr;
  HIPCHK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return ORBFE_OK;
}
extern "C" void orbfe_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

namespace {
// page-locks [p, p+bytes) for the duration of a call when the caller did not allocate it with orbfe_host_alloc
struct ScopedPin {
  void* p = nullptr;
  int pin(const void* ptr, size_t bytes) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type != hipMemoryTypeUnregistered) return ORBFE_OK;  // already pinned
    (void)hipGetLastError();
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      return ORBFE_OK;  // pageable memory still works (the runtime stages it), only slower
    }
    p = const_cast<void*>(ptr);
    return ORBFE_OK;
  }
  ~ScopedPin() { if (p) (void)hipHostUnregister(p); }
};
}  // namespace

// ORBextractor::operator() for a host batch, end to end: images in host memory -> keypoints and descriptors in host
// memory (src/ORBextractor.cc:1119-1197 per frame).  The batch is cut into chunks; the H2D copy of chunk k+1, the
// kernels of chunk k and the D2H copy of chunk k-1 run at the same time on three sets of streams (two input slabs and
// two output blocks in HBM, ordered by events; no host wait inside the loop).  Outputs use the layout of
// orbfe_extract_batch: frame f at keypoints + f*capacity, descriptors + f*capacity*32; rows past n_out[f] are
// unspecified.  Host buffers from orbfe_host_alloc (pinned) transfer at PCIe speed; others are page-locked for the
// duration of the call.
extern "C" int orbfe_extract_batch_pipelined(orbfe_extractor* e, const uint8_t* images, int n_frames, int width, int height,
                                             int stride, size_t frame_stride, orbfe_keypoint* keypoints,
                                             uint8_t* descriptors, int capacity, int* n_out, int chunk_frames) {
  if (!e || !n_out || n_frames < 0) return fail(ORBFE_ERR_INVALID, "extract_batch_pipelined: bad argument");
  for (int f = 0; f < n_frames; f++) n_out[f] = 0;
  if (n_frames == 0) return ORBFE_OK;
  if (!images || width <= 0 || height <= 0) return ORBFE_OK;  // empty image: silent return (:1122)
  if (!keypoints || !descriptors || capacity <= 0 || stride < width || frame_stride < (size_t)stride * (height - 1) + width)
    return fail(ORBFE_ERR_INVALID, "extract_batch_pipelined: bad buffers");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  if ((rc = sync_all(e))) return rc;
  if ((rc = ensure_geometry(e, width, height))) return rc;
  int C = chunk_frames > 0 ? chunk_frames : 256;
  if (C > n_frames) C = n_frames;
  if ((rc = ensure_workspace(e, C))) return rc;
  if (!e->sH2D) {
    HIPCHK(stream_get(e->device, kRoleH2D, &e->sH2D));
    HIPCHK(stream_get(e->device, kRoleD2H, &e->sD2H));
    for (int i = 0; i < 2; i++) {
      HIPCHK(hipEventCreateWithFlags(&e->evIn[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->evComp[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->evOutDone[i], hipEventDisableTiming));
    }
  }
  // device input slab: tightly packed host frames are uploaded as they are (ONE linear copy per chunk, row pitch =
  // the caller's stride, which the kernels read in place whatever its alignment; only $ORBFE_COPY_UNALIGNED=1 repacks
  // an odd pitch in run_chunk) -- a 2-D copy of 1241-byte rows
  // degenerates into one DMA descriptor per row and ran at 0.1 GB/s; only batches with gaps between the frames
  // take the per-frame 2-D form into a 64-byte pitch
  const bool tight = frame_stride == (size_t)stride * height;
  const int pitch = tight ? stride : (width + 63) & ~63;
  const size_t inBytes = (size_t)C * pitch * height;
  const size_t kpB = ((size_t)C * capacity * sizeof(orbfe_keypoint) + 255) & ~(size_t)255;
  const size_t deB = ((size_t)C * capacity * 32 + 255) & ~(size_t)255;
  const size_t outBytes = kpB + deB + (size_t)C * 4;
  if (inBytes > e->pipeInBytes || outBytes > e->pipeOutBytes) {
    HIPCHK(hipStreamSynchronize(e->sH2D));
    HIPCHK(hipStreamSynchronize(e->sD2H));
    for (int i = 0; i < 2; i++) {
      if ((rc = dalloc(&e->d_pipeIn[i], inBytes))) return rc;
      if ((rc = dalloc(&e->d_pipeOut[i], outBytes))) return rc;
    }
    e->pipeInBytes = inBytes;
    e->pipeOutBytes = outBytes;
  }
  ScopedPin pinIn, pinKp, pinDesc;
  pinIn.pin(images, frame_stride * (size_t)(n_frames - 1) + (size_t)stride * (height - 1) + width);
  pinKp.pin(keypoints, (size_t)n_frames * capacity * sizeof(orbfe_keypoint));
  pinDesc.pin(descriptors, (size_t)n_frames * capacity * 32);
  int32_t* h_cnt = nullptr;  // counts come back through a small pinned block of their own
  HIPCHK(hipHostMalloc((void**)&h_cnt, sizeof(int32_t) * (size_t)n_frames, hipHostMallocDefault));
  const int nChunks = (n_frames + C - 1) / C;
  int status = ORBFE_OK;
  for (int k = 0; k < nChunks && status == ORBFE_OK; k++) {
    const int slot = k & 1;
    const int f0 = k * C, n = f0 + C <= n_frames ? C : n_frames - f0;
    auto H = [&](hipError_t err) { if (err != hipSuccess && status == ORBFE_OK) status = fail(ORBFE_ERR_HIP, hipGetErrorString(err)); };
    // input slab `slot` is free once the kernels of chunk k-2 are done
    if (k >= 2) H(hipStreamWaitEvent(e->sH2D, e->evComp[slot], 0));
    if (tight) {
      // (up to the last pixel of the chunk's last frame: the caller's array may end there)
      H(hipMemcpyAsync(e->d_pipeIn[slot], images + (size_t)f0 * frame_stride,
                       (size_t)(n - 1) * frame_stride + (size_t)(height - 1) * stride + width, hipMemcpyHostToDevice, e->sH2D));
    } else {
      for (int f = 0; f < n; f++)
        H(hipMemcpy2DAsync(e->d_pipeIn[slot] + (size_t)f * pitch * height, pitch, images + (size_t)(f0 + f) * frame_stride, stride,
                           width, height, hipMemcpyHostToDevice, e->sH2D));
    }
    H(hipEventRecord(e->evIn[slot], e->sH2D));
    // kernels: wait for the upload, and for the D2H of chunk k-2 before its output block is overwritten
    hipEvent_t waits[2] = {e->evIn[slot], e->evOutDone[slot]};
    orbfe_keypoint* d_kp = reinterpret_cast<orbfe_keypoint*>(e->d_pipeOut[slot]);
    uint8_t* d_de = e->d_pipeOut[slot] + kpB;
    int32_t* d_n = reinterpret_cast<int32_t*>(e->d_pipeOut[slot] + kpB + deB);
    next_event_slot(e);
    LevelView l0{e->d_pipeIn[slot], (size_t)pitch * height, pitch, width, height};
    if (status == ORBFE_OK) {
      rc = run_pipeline(e, l0, n, d_kp, d_de, capacity, d_n, waits, k >= 2 ? 2 : 1);
      if (rc) { status = rc; break; }
      e->lastFrameBase = f0;  // the pyramid / blurred levels / candidates that stay behind are those of frames [f0, f0 + n)
    }
    for (int i = 1; i < e->chunksPending; i++) H(hipStreamWaitEvent(e->stream, e->evChunkDone[i], 0));  // join the sub-batches
    H(hipEventRecord(e->evComp[slot], e->stream));
    // results of this chunk straight into the caller's arrays
    H(hipStreamWaitEvent(e->sD2H, e->evComp[slot], 0));
    H(hipMemcpyAsync(keypoints + (size_t)f0 * capacity, d_kp, (size_t)n * capacity * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, e->sD2H));
    H(hipMemcpyAsync(descriptors + (size_t)f0 * capacity * 32, d_de, (size_t)n * capacity * 32, hipMemcpyDeviceToHost, e->sD2H));
    H(hipMemcpyAsync(h_cnt + f0, d_n, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, e->sD2H));
    H(hipEventRecord(e->evOutDone[slot], e->sD2H));
  }
  hipError_t e1 = hipStreamSynchronize(e->sD2H);
  int rc2 = sync_all(e);
  hipError_t e2 = hipStreamSynchronize(e->sH2D);
  bool overflow = false;
  if (status == ORBFE_OK && e1 == hipSuccess && e2 == hipSuccess && rc2 == ORBFE_OK)
    for (int f = 0; f < n_frames; f++) {
      int n = h_cnt[f];
      if (n > capacity) { overflow = true; n = capacity; }
      n_out[f] = n;
    }
  (void)hipHostFree(h_cnt);
  resolve_stage_times(e);
  if (status != ORBFE_OK) return status;
  if (e1 != hipSuccess) return fail(ORBFE_ERR_HIP, hipGetErrorString(e1));
  if (e2 != hipSuccess) return fail(ORBFE_ERR_HIP, hipGetErrorString(e2));
  if (rc2) return rc2;
  if (overflow) return fail(ORBFE_ERR_CAPACITY, "keypoint capacity too small");
  return ORBFE_OK;
}

extern "C" int orbfe_extract(orbfe_extractor* e, const uint8_t* image, int width, int height, int stride,
                             orbfe_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out) {
  return orbfe_extract_batch(e, image, 1, width, height, stride, (size_t)stride * (size_t)(height > 0 ? height : 0),
                             keypoints, descriptors, capacity, n_out);
}

// frame index of the LAST extract call -> index into what the handle still holds.  The chunked (pipelined) host path keeps
// the intermediate data of its last chunk only: frames before it fail loudly instead of silently answering with another
// frame's pyramid.
static int retained_frame(orbfe_extractor* e, int frame, int* local) {
  if (!e->haveLast) return fail(ORBFE_ERR_INVALID, "no extract call yet");
  const int f = frame - e->lastFrameBase;
  if (frame < 0 || f >= e->lastFrames) return fail(ORBFE_ERR_INVALID, "frame out of range");
  if (f < 0)
    return fail(ORBFE_ERR_INVALID, "pyramid of frame " + std::to_string(frame) + " not retained: the pipelined host path keeps the last chunk only (frames " +
                                       std::to_string(e->lastFrameBase) + ".." + std::to_string(e->lastFrameBase + e->lastFrames - 1) + ")");
  *local = f;
  return ORBFE_OK;
}

static int copy_level_out(orbfe_extractor* e, const PyramidViews& pv, int frame, int level, uint8_t* dst, int dst_stride) {
  if (!e || !dst) return fail(ORBFE_ERR_INVALID, "NULL argument");
  { int rcf = retained_frame(e, frame, &frame); if (rcf) return rcf; }
  if (level < 0 || level >= pv.nlevels)
    return fail(ORBFE_ERR_INVALID, "frame/level out of range");
  const LevelView& v = pv.lv[level];
  if (dst_stride < v.w) return fail(ORBFE_ERR_INVALID, "dst_stride too small");
  HIPCHK(hipSetDevice(e->device));
  { int rcs = sync_all(e); if (rcs) return rcs; }
  HIPCHK(hipMemcpy2DAsync(dst, dst_stride, v.base + (size_t)frame * v.frameStride, v.pitch, v.w, v.h,
                          hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return ORBFE_OK;
}
extern "C" int orbfe_extractor_get_pyramid_level(orbfe_extractor* e, int frame, int level, uint8_t* dst, int dst_stride) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  return copy_level_out(e, e->lastPyr, frame, level, dst, dst_stride);
}
extern "C" int orbfe_extractor_debug_blurred_level(orbfe_extractor* e, int frame, int level, uint8_t* dst, int dst_stride) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  return copy_level_out(e, e->lastBlur, frame, level, dst, dst_stride);
}
extern "C" int orbfe_extractor_pyramid_level_device(orbfe_extractor* e, int frame, int level, const uint8_t** d_ptr,
                                                    int* pitch, int* w, int* h) {
  if (!e || !d_ptr || !pitch || !w || !h) return fail(ORBFE_ERR_INVALID, "NULL argument");
  { int rcf = retained_frame(e, frame, &frame); if (rcf) return rcf; }
  if (level < 0 || level >= e->lastPyr.nlevels) return fail(ORBFE_ERR_INVALID, "frame/level out of range");
  const LevelView& v = e->lastPyr.lv[level];
  *d_ptr = v.base + (size_t)frame * v.frameStride;
  *pitch = v.pitch;
  *w = v.w;
  *h = v.h;
  return ORBFE_OK;
}

extern "C" int orbfe_extractor_debug_candidates(orbfe_extractor* e, int frame, int level, float* xs, float* ys,
                                                float* resp, int cap) {
  if (!e || !xs || !ys || !resp) return fail(ORBFE_ERR_INVALID, "NULL argument");
  { int rcf = retained_frame(e, frame, &frame); if (rcf) return rcf; }
  if (level < 0 || level >= e->geom.nlevels) return fail(ORBFE_ERR_INVALID, "frame/level out of range");
  HIPCHK(hipSetDevice(e->device));
  { int rcs = sync_all(e); if (rcs) return rcs; }
  int32_t n = 0;
  HIPCHK(hipMemcpy(&n, e->d_candCount + (size_t)frame * e->geom.nlevels + level, 4, hipMemcpyDeviceToHost));
  std::vector<Candidate> c((size_t)(n > 0 ? n : 1));
  if (n > 0)
    HIPCHK(hipMemcpy(c.data(), e->d_cand + (size_t)frame * e->geom.totalSlots + e->geom.lv[level].slotStart,
                     sizeof(Candidate) * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap; i++) {
    xs[i] = (float)(c[i].xy & 0xffffu);
    ys[i] = (float)(c[i].xy >> 16);
    resp[i] = (float)c[i].score;
  }
  return n;
}

extern "C" int orbfe_extractor_profile(orbfe_extractor* e, int stage_mask) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  { int rcs = sync_all(e); if (rcs) return rcs; }
  resolve_stage_times(e);
  e->stageMask = stage_mask < 0 ? 0xffffffffu : (unsigned)stage_mask;
  for (int i = 0; i < ORBFE_STAGE_COUNT; i++) { e->stageMs[i] = 0; e->stageLaunches[i] = 0; e->stageFrames[i] = 0; }
  return ORBFE_OK;
}
extern "C" int orbfe_extractor_profile_get(orbfe_extractor* e, double* ms_out, int64_t* launches_out,
                                           double* frames_out) {
  if (!e || !ms_out || !launches_out) return fail(ORBFE_ERR_INVALID, "NULL argument");
  for (int i = 0; i < ORBFE_STAGE_COUNT; i++) {
    ms_out[i] = e->stageMs[i];
    launches_out[i] = e->stageLaunches[i];
    if (frames_out) frames_out[i] = e->stageFrames[i];
  }
  return ORBFE_OK;
}
extern "C" const char* orbfe_stage_name(int stage) {
  static const char* names[ORBFE_STAGE_COUNT] = {"h2d", "pyramid", "fast", "octree", "blur", "orient_desc", "d2h", "match"};
  return (stage >= 0 && stage < ORBFE_STAGE_COUNT) ? names[stage] : "?";
}

// ---- standalone primitives on host buffers ----
extern "C" int orbfe_resize_linear(int device, const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                   int dw, int dh, int dstride) {
  if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || sstride < sw || dstride < dw)
    return fail(ORBFE_ERR_INVALID, "resize: bad argument");
  HIPCHK(hipSetDevice(device));
  ResizeTables t;
  build_resize_tables(sw, sh, dw, dh, &t);
  const int sp = (sw + 63) & ~63, dp = (dw + 63) & ~63;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  int32_t *d_xofs = nullptr, *d_yofs = nullptr;
  int16_t *d_alpha = nullptr, *d_beta = nullptr;
  uint32_t *d_colrec = nullptr, *d_rowrec = nullptr;
  int rc = ORBFE_OK;
  auto cleanup = [&]() {
    dfree(&d_src); dfree(&d_dst); dfree(&d_xofs); dfree(&d_yofs); dfree(&d_alpha); dfree(&d_beta);
    dfree(&d_colrec); dfree(&d_rowrec);
  };
  if ((rc = dalloc(&d_src, (size_t)sp * sh)) || (rc = dalloc(&d_dst, (size_t)dp * dh)) ||
      (rc = dalloc(&d_xofs, t.xofs.size())) || (rc = dalloc(&d_yofs, t.yofs.size())) ||
      (rc = dalloc(&d_alpha, t.alpha.size())) || (rc = dalloc(&d_beta, t.beta.size()))) { cleanup(); return rc; }
  hipError_t err = hipMemcpy2D(d_src, sp, src, sstride, sw, sh, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(d_xofs, t.xofs.data(), t.xofs.size() * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(d_yofs, t.yofs.data(), t.yofs.size() * 4, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(d_alpha, t.alpha.data(), t.alpha.size() * 2, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(d_beta, t.beta.data(), t.beta.size() * 2, hipMemcpyHostToDevice);
  if (err == hipSuccess && !t.colrec.empty()) {
    if ((rc = dalloc(&d_colrec, t.colrec.size())) || (rc = dalloc(&d_rowrec, t.rowrec.size()))) { cleanup(); return rc; }
    err = hipMemcpy(d_colrec, t.colrec.data(), t.colrec.size() * 4, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_rowrec, t.rowrec.data(), t.rowrec.size() * 4, hipMemcpyHostToDevice);
  }
  if (err == hipSuccess) {
    launch_resize(nullptr, LevelView{d_src, 0, sp, sw, sh}, LevelViewMut{d_dst, 0, dp, dw, dh}, d_xofs, d_alpha, d_yofs, d_beta,
                  d_colrec, d_rowrec, 1);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err == hipSuccess) err = hipMemcpy2D(dst, dstride, d_dst, dp, dw, dh, hipMemcpyDeviceToHost);
  cleanup();
  if (err != hipSuccess) return fail(ORBFE_ERR_HIP, std::string("resize: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

extern "C" int orbfe_gaussian_blur7_spec(int device, int spec, const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
extern "C" int orbfe_gaussian_blur7(int device, const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  return orbfe_gaussian_blur7_spec(device, kBlurSpecCv4, src, w, h, sstride, dst, dstride);
}
extern "C" int orbfe_gaussian_blur7_spec(int device, int spec, const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  if (!src || !dst || w <= 0 || h <= 0 || sstride < w || dstride < w || spec < 0 || spec > 2) return fail(ORBFE_ERR_INVALID, "blur: bad argument");
  HIPCHK(hipSetDevice(device));
  const int p = (w + 63) & ~63;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  int rc;
  if ((rc = dalloc(&d_src, (size_t)p * h))) return rc;
  if ((rc = dalloc(&d_dst, (size_t)p * h))) { dfree(&d_src); return rc; }
  hipError_t err = hipMemcpy2D(d_src, p, src, sstride, w, h, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    launch_blur7(nullptr, LevelView{d_src, 0, p, w, h}, LevelViewMut{d_dst, 0, p, w, h}, 1, spec);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err == hipSuccess) err = hipMemcpy2D(dst, dstride, d_dst, p, w, h, hipMemcpyDeviceToHost);
  dfree(&d_src);
  dfree(&d_dst);
  if (err != hipSuccess) return fail(ORBFE_ERR_HIP, std::string("blur: ") + hipGetErrorString(err));
  return ORBFE_OK;
}

// internal (matcher.hip): pyramid views + scale tables of the last extract call
extern "C" int orbfe_stereo_views_(orbfe_extractor* e, int frame, PyramidViews* pv, float* scale, float* invScale,
                                   int* nlevels, int* device, const float** d_scaleTab) {
  if (!e->haveLast) return fail(ORBFE_ERR_INVALID, "compute_stereo_matches: extractor has no pyramid yet");
  int local = 0;
  { int rcf = retained_frame(e, frame, &local); if (rcf) return rcf; }
  HIPCHK(hipSetDevice(e->device));
  { int rcs = sync_all(e); if (rcs) return rcs; }
  *pv = e->lastPyr;
  for (int l = 0; l < pv->nlevels; l++)  // the caller indexes the views with ITS frame number
    pv->lv[l].base -= (size_t)e->lastFrameBase * pv->lv[l].frameStride;
  for (int l = 0; l < e->tab.nlevels; l++) { scale[l] = e->tab.scale[l]; invScale[l] = e->tab.invScale[l]; }
  *nlevels = e->tab.nlevels;
  *device = e->device;
  *d_scaleTab = e->d_scaleTab;
  return ORBFE_OK;
}

// internal (vocabulary.hip): a consumer of the last extract call's outputs that runs on the handle's own
// stream.  begin: stream 0 waits (on the device) for every sub-batch stream of that call and is returned;
// end: marks the consumer's last kernel so that the next extract call's sub-batch streams wait for it.
extern "C" int orbfe_extractor_consumer_begin_(orbfe_extractor* e, hipStream_t* s) {
  if (!e || !s) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  for (int i = 1; i < e->chunksPending; i++) HIPCHK(hipStreamWaitEvent(e->stream, e->evChunkDone[i], 0));
  *s = e->stream;
  if ((e->stageMask >> ORBFE_STAGE_MATCH) & 1u) {  // same ring slot as the extract call it follows, sub-batch 0
    if (e->evUsed[e->evSlot][0][ORBFE_STAGE_MATCH]) {  // a second matcher behind the same call: one interval for both
      e->evLaunches[e->evSlot][0][ORBFE_STAGE_MATCH]++;
    } else {
      HIPCHK(hipEventRecord(e->evA[e->evSlot][0][ORBFE_STAGE_MATCH], e->stream));
      e->evLaunches[e->evSlot][0][ORBFE_STAGE_MATCH] = 1;
      e->evFrames[e->evSlot][0][ORBFE_STAGE_MATCH] = e->lastFrames;
    }
  }
  return ORBFE_OK;
}
extern "C" int orbfe_extractor_consumer_end_(orbfe_extractor* e) {
  if ((e->stageMask >> ORBFE_STAGE_MATCH) & 1u) {
    HIPCHK(hipEventRecord(e->evB[e->evSlot][0][ORBFE_STAGE_MATCH], e->stream));
    e->evUsed[e->evSlot][0][ORBFE_STAGE_MATCH] = true;
  }
  HIPCHK(hipEventRecord(e->evConsumerDone, e->stream));
  e->consumerPending = true;
  return ORBFE_OK;
}

// internal (vocabulary.hip): how the last extract call was split, so that a batched matcher can work per sub-batch on
// the sub-batch's own stream.  streams / chunkDone: kMaxStreams entries.
extern "C" int orbfe_extractor_split_(orbfe_extractor* e, int* S, int* per, int* frames, int* lanes, hipStream_t* streams,
                                      hipEvent_t* chunkDone) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  *S = e->haveLast ? e->lastS : 0;
  *per = e->lastPer;
  *frames = e->lastFrames;
  *lanes = e->lastLanes ? 1 : 0;
  for (int i = 0; i < orbfe_extractor::kMaxStreams; i++) {
    streams[i] = i == 0 ? e->stream : e->extra[i - 1];
    chunkDone[i] = e->evChunkDone[i];
  }
  return ORBFE_OK;
}
// internal: stage-timer marks for work another translation unit enqueues on a sub-batch stream
extern "C" void orbfe_extractor_stage_mark_(orbfe_extractor* e, int stage, int sub, int isEnd, hipStream_t s, int frames) {
  if (!e || sub < 0 || sub >= orbfe_extractor::kEvSubs || !((e->stageMask >> stage) & 1u)) return;
  if (!isEnd) {
    // a second matcher behind the same sub-batch on the same stream (stereo, then BoW): one interval from the first
    // one's start to the last one's end
    if (e->evUsed[e->evSlot][sub][stage]) { e->evLaunches[e->evSlot][sub][stage]++; return; }
    (void)hipEventRecord(e->evA[e->evSlot][sub][stage], s);
    e->evLaunches[e->evSlot][sub][stage] = 1;
    e->evFrames[e->evSlot][sub][stage] = frames;
  } else {
    (void)hipEventRecord(e->evB[e->evSlot][sub][stage], s);
    e->evUsed[e->evSlot][sub][stage] = true;
  }
}

// Debug: route DistributeOctTree through the host implementation (cross-check of k_octree).
extern "C" int orbfe_extractor_debug_host_octree(orbfe_extractor* e, int enable) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  e->hostOctree = enable != 0;
  return ORBFE_OK;
}

// FAST threshold order: 0 = auto, 1 = iniThFAST first with per-cell fallback, 2 = one attempt at the lower threshold.
extern "C" int orbfe_extractor_set_fast_mode(orbfe_extractor* e, int mode) {
  if (!e || mode < 0 || mode > 2) return fail(ORBFE_ERR_INVALID, "set_fast_mode: 0 (auto), 1 (high first) or 2 (low first)");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->fastMode = mode;
  for (int i = 0; i < orbfe_extractor::kMaxStreams; i++) e->statPending[i] = false;
  return ORBFE_OK;
}

// Which OpenCV's GaussianBlur arithmetic the extractor reproduces (include/orbfe.h, ORBFE_BLUR_*).
extern "C" int orbfe_extractor_set_blur_spec(orbfe_extractor* e, int spec) {
  if (!e || spec < 0 || spec > 2) return fail(ORBFE_ERR_INVALID, "set_blur_spec: 0 (OpenCV >= 3.4.1/4.x), 1 (2.4/3.x scalar) or 2 (2.4/3.x SSE2)");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->blurSpec = spec;
  return ORBFE_OK;
}

// GaussianBlur inside the FAST kernel (default) or as the separate k_blur7 launch (A/B and cross-check).
extern "C" int orbfe_extractor_set_fused(orbfe_extractor* e, int enable) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->fused = enable != 0;
  return ORBFE_OK;
}

// ComputePyramid and the per-level GaussianBlur as ONE kernel per level (1, the default; $ORBFE_PYRBLUR) or as the
// separate resize and blur launches (0).  Identical results.
extern "C" int orbfe_extractor_set_pyramid_blur(orbfe_extractor* e, int enable) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->pyrBlur = enable != 0;
  return ORBFE_OK;
}

// The pyramid of a call of <= 8 frames in ONE launch (k_pyramid_chain) instead of n-1 dependent resize launches.  Identical
// results; off by default (measured: no faster).
extern "C" int orbfe_extractor_set_pyramid_chain(orbfe_extractor* e, int enable) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->pyrChain = enable != 0;
  return ORBFE_OK;
}

// Order of the blur kernel's two passes (process-wide, k_blur.hip); the blurred bytes are the same either way.
extern "C" int orbfe_set_blur_pass_order(int order) {
  if (order >= 0) set_blur_pass_order(order);
  return blur_pass_order();
}

// Schedule of the sub-batches of a call: 0 = one independent stream per sub-batch, 1 = three lanes shared by all
// sub-batches (pyramid | FAST + blur | gather + octree + orientation/descriptors) as a software pipeline.
extern "C" int orbfe_extractor_set_desc_tiles(orbfe_extractor* e, int enable) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  e->descTilesMode = enable < 0 ? -1 : (enable ? 1 : 0);
  return ORBFE_OK;
}

extern "C" int orbfe_extractor_set_schedule(orbfe_extractor* e, int lanes) {
  if (!e) return fail(ORBFE_ERR_INVALID, "NULL handle");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  e->laneMode = lanes != 0;
  return ORBFE_OK;
}

// Number of sub-batch streams one call is split over (1..32; default 1, or $ORBFE_STREAMS).
extern "C" int orbfe_extractor_set_streams(orbfe_extractor* e, int n) {
  if (!e || n < 1 || n > orbfe_extractor::kMaxStreams) return fail(ORBFE_ERR_INVALID, "set_streams: 1..32");
  HIPCHK(hipSetDevice(e->device));
  int rc = sync_all(e);
  if (rc) return rc;
  if ((rc = ensure_subs(e, n))) return rc;
  e->nStreams = n;
  return ORBFE_OK;
}

// Frame::ComputeStereoMatches for every stereo pair of a device-resident batch (see orbfe.h).
extern "C" int orbfe_stereo_match_batch_device(orbfe_extractor* e, int n_pairs, const orbfe_keypoint* d_keypoints,
                                               const uint8_t* d_descriptors, const int32_t* d_n, int capacity,
                                               float mbf, float mb, float* d_uRight, float* d_depth,
                                               int32_t* d_n_stereo) {
  if (!e || n_pairs < 0 || capacity <= 0 || !d_keypoints || !d_descriptors || !d_n || !d_uRight || !d_depth ||
      !d_n_stereo)
    return fail(ORBFE_ERR_INVALID, "stereo_match_batch_device: bad argument");
  if (n_pairs == 0) return ORBFE_OK;
  if (!e->haveLast || e->lastFrames < 2 * n_pairs)
    return fail(ORBFE_ERR_INVALID, "stereo_match_batch_device: the last extract call holds fewer than 2*n_pairs frames");
  if (capacity >= (1 << 20)) return fail(ORBFE_ERR_INVALID, "stereo_match_batch_device: capacity too large");
  HIPCHK(hipSetDevice(e->device));
  const size_t need = (size_t)n_pairs * capacity;
  const int rows = e->lastPyr.lv[0].h;
  const size_t needRows = (size_t)n_pairs * (rows + 1);
  if (need > e->stereoSadCap || needRows > e->stereoRowCap) {
    int rcs = sync_all(e);
    if (rcs) return rcs;
    int rc = dalloc(&e->d_stereoSad, need);
    if (!rc) rc = dalloc(&e->d_stereoSorted, need);
    if (!rc) rc = dalloc(&e->d_stereoRec, need);
    if (!rc) rc = dalloc(&e->d_stereoRowStart, needRows);
    if (rc) return rc;
    e->stereoSadCap = need;
    e->stereoRowCap = needRows;
  }
  StereoArgs a = {};
  a.pyrL = e->lastPyr;
  a.pyrR = e->lastPyr;
  a.scaleTab = e->d_scaleTab;
  a.mbf = mbf;
  a.maxD = mbf / mb;  // minZ = mb, maxD = mbf/minZ (src/Frame.cc:542-544)
  if (rows + 1 <= 8192) {  // row index of the right keypoints (k_stereo_bucket)
    a.rowStart = e->d_stereoRowStart;
    a.sortedIdx = e->d_stereoSorted;
    a.sortedRec = e->d_stereoRec;
    a.rows = rows;
    a.bandR = (int)std::ceil(2.0f * e->tab.scale[e->tab.nlevels - 1]) + 2;
  }
  StereoBatch b = {};
  b.kp = reinterpret_cast<const float*>(d_keypoints);
  b.desc = d_descriptors;
  b.n = d_n;
  b.capacity = capacity;
  b.uRight = d_uRight;
  b.depth = d_depth;
  b.sad = e->d_stereoSad;
  // pairs [p0, p0+np) on stream st: every operand moved to the first pair by pointer arithmetic
  auto launch_range = [&](hipStream_t st, int p0, int np, int sub) {
    StereoArgs aa = a;
    StereoBatch bb = b;
    for (int l = 0; l < aa.pyrL.nlevels; l++) {
      aa.pyrL.lv[l].base += (size_t)(2 * p0) * aa.pyrL.lv[l].frameStride;
      aa.pyrR.lv[l].base += (size_t)(2 * p0) * aa.pyrR.lv[l].frameStride;
    }
    if (aa.rowStart) { aa.rowStart += (size_t)p0 * (rows + 1); aa.sortedIdx += (size_t)p0 * capacity; aa.sortedRec += (size_t)p0 * capacity; }
    bb.kp += (size_t)(2 * p0) * capacity * 7;
    bb.desc += (size_t)(2 * p0) * capacity * 32;
    bb.n += 2 * p0;
    bb.uRight += (size_t)p0 * capacity;
    bb.depth += (size_t)p0 * capacity;
    bb.sad += (size_t)p0 * capacity;
    if (knockout_mask(e, false) & 16) return;
    StageTimer t(e, ORBFE_STAGE_MATCH, 1, 2 * np, sub, st);
    launch_stereo_batch(st, aa, bb, np, d_n_stereo + p0);
  };
  const int S = e->lastS, per = e->lastPer;
  if (!e->lastLanes && S > 1 && (per & 1) == 0 && 2 * n_pairs == e->lastFrames) {
    // The pairs of a sub-batch are matched on that sub-batch's own stream, right behind its extraction: no join of
    // the sub-batch streams, and the (latency-bound) matcher of one sub-batch overlaps the kernels of the others.
    // The next extract call's sub-batch i is enqueued on the same stream, i.e. behind this matcher, by itself.
    for (int i = 0; i < S; i++) {
      const int f0 = i * per;
      const int n = f0 + per <= e->lastFrames ? per : e->lastFrames - f0;
      if (n <= 0) break;
      launch_range(i == 0 ? e->stream : e->extra[i - 1], f0 / 2, n / 2, i);
      if (i > 0) HIPCHK(hipEventRecord(e->evChunkDone[i], e->extra[i - 1]));  // "sub-batch i done" now includes its matcher
    }
    HIPCHK(hipGetLastError());
    return ORBFE_OK;
  }
  // one launch on stream 0 behind all sub-batches (joined ON THE DEVICE, no host synchronisation)
  { hipStream_t s0; int rc = orbfe_extractor_consumer_begin_(e, &s0); if (rc) return rc; }
  {
    const unsigned keep = e->stageMask;
    e->stageMask &= ~(1u << ORBFE_STAGE_MATCH);  // consumer_begin_/end_ time this form
    launch_range(e->stream, 0, n_pairs, 0);
    e->stageMask = keep;
  }
  HIPCHK(hipGetLastError());
  // the next extract call's sub-batch streams overwrite the pyramid slabs and the caller's keypoint /
  // descriptor / count buffers this matcher is still reading: they wait for this event (run_pipeline)
  { int rc = orbfe_extractor_consumer_end_(e); if (rc) return rc; }
  return ORBFE_OK;  // asynchronous on the handle's streams: orbfe_extractor_synchronize() to wait
}

// The stereo Frame constructor's front end in ONE call (src/Frame.cc:78-96: ExtractORB on two threads, join,
// ComputeStereoMatches): both eyes go through this handle as a two-frame batch -- one upload, every kernel of the
// single-frame chain over two frames, the stereo matcher on the records that are still in HBM -- and keypoints,
// descriptors, mvuRight and mvDepth come back in one download.  Two handles on two threads + orbfe_compute_stereo_matches
// (which takes the keypoints it has just downloaded up again) cost 0.42-0.51 ms per KITTI pair; this call ~0.3 ms.
extern "C" int orbfe_extract_stereo_frame(orbfe_extractor* e, const uint8_t* left, const uint8_t* right, int width, int height,
                                          int stride, orbfe_keypoint* kpL, uint8_t* descL, int* nL, orbfe_keypoint* kpR,
                                          uint8_t* descR, int* nR, int capacity, float mbf, float mb, float* uRight,
                                          float* depth) {
  if (!e || !nL || !nR) return fail(ORBFE_ERR_INVALID, "extract_stereo_frame: bad argument");
  *nL = *nR = 0;
  if (!left || !right || width <= 0 || height <= 0) return ORBFE_OK;  // empty image: silent return (:1122)
  if (!kpL || !descL || !kpR || !descR || !uRight || !depth || capacity <= 0 || stride < width || capacity >= (1 << 20))
    return fail(ORBFE_ERR_INVALID, "extract_stereo_frame: bad output buffers");
  HIPCHK(hipSetDevice(e->device));
  int rc;
  e->outLastFrames = 0;
  if ((rc = ensure_geometry(e, width, height))) return rc;
  if ((rc = ensure_workspace(e, 2))) return rc;
  if ((rc = ensure_outputs(e, 2, capacity))) return rc;
  if (e->frameStereoCap < (size_t)capacity) {
    if ((rc = sync_all(e))) return rc;
    e->frameStereoCap = 0;
    if ((rc = dalloc(&e->d_frameStereo, 2 * (size_t)capacity + 16))) return rc;
    e->frameStereoCap = (size_t)capacity;
  }
  next_event_slot(e);
  // both images into the input slab at the caller's pitch (the kernels read rows in place at any stride)
  const size_t frameStride = ((size_t)stride * height + 255) & ~(size_t)255;
  const size_t slab = 2 * frameStride + 64;
  if (slab > e->hostInBytes) {
    e->hostInBytes = 0;
    e->haveLast = false;
    if ((rc = dalloc(&e->d_hostIn, slab))) return rc;
    e->hostInBytes = slab;
  }
  const size_t bytes = (size_t)(height - 1) * stride + width;
  {
    StageTimer t(e, ORBFE_STAGE_H2D, 0, 0, 0, e->stream);
    HIPCHK(hipMemcpyAsync(e->d_hostIn, left, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_hostIn + frameStride, right, bytes, hipMemcpyHostToDevice, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));  // (as orbfe_extract_batch: sub-batch streams would read the uploaded frames)
  const LevelView l0{e->d_hostIn, frameStride, stride, width, height};
  if ((rc = run_pipeline(e, l0, 2, e->d_kpOut, e->d_descOut, capacity, e->d_nOut))) return rc;
  float* d_ur = e->d_frameStereo;
  float* d_dp = e->d_frameStereo + capacity;
  int32_t* d_ns = reinterpret_cast<int32_t*>(e->d_frameStereo + 2 * (size_t)capacity);
  if ((rc = orbfe_stereo_match_batch_device(e, 1, e->d_kpOut, e->d_descOut, e->d_nOut, capacity, mbf, mb, d_ur, d_dp, d_ns)))
    return rc;
  if (e->lastS > 1 || e->lastLanes) { if ((rc = sync_all(e))) return rc; }  // (a pair is one sub-batch: everything is on e->stream)
  {
    StageTimer t(e, ORBFE_STAGE_D2H, 0, 0, 0, e->stream);
    const size_t blockBytes = (size_t)(reinterpret_cast<uint8_t*>(e->d_nOut) - e->d_outBlock) + sizeof(int32_t) * 2;
    const size_t stBytes = 2 * (size_t)capacity * sizeof(float);
    const size_t need = blockBytes + stBytes;
    if (e->outStageBytes < need) {
      if (e->h_outStage) (void)hipHostFree(e->h_outStage);
      e->h_outStage = nullptr;
      e->outStageBytes = 0;
      const size_t want = need > (size_t)512 * 1024 ? need : (size_t)512 * 1024;
      HIPCHK(hipHostMalloc((void**)&e->h_outStage, want, hipHostMallocDefault));
      e->outStageBytes = want;
    }
    HIPCHK(hipMemcpyAsync(e->h_outStage, e->d_outBlock, blockBytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_outStage + blockBytes, e->d_frameStereo, stBytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const uint8_t* hk = e->h_outStage;
    const uint8_t* hd = e->h_outStage + (e->d_descOut - e->d_outBlock);
    const int32_t* hc = reinterpret_cast<const int32_t*>(e->h_outStage + (reinterpret_cast<uint8_t*>(e->d_nOut) - e->d_outBlock));
    bool overflow = false;
    int n[2];
    orbfe_keypoint* kk[2] = {kpL, kpR};
    uint8_t* dd[2] = {descL, descR};
    for (int f = 0; f < 2; f++) {
      n[f] = hc[f];
      if (n[f] > capacity) { overflow = true; n[f] = capacity; }
      if (n[f] > 0) {
        std::memcpy(kk[f], hk + (size_t)f * capacity * sizeof(orbfe_keypoint), sizeof(orbfe_keypoint) * (size_t)n[f]);
        std::memcpy(dd[f], hd + (size_t)f * capacity * 32, (size_t)n[f] * 32);
      }
    }
    *nL = n[0];
    *nR = n[1];
    if (n[0] > 0) {
      std::memcpy(uRight, e->h_outStage + blockBytes, sizeof(float) * (size_t)n[0]);
      std::memcpy(depth, e->h_outStage + blockBytes + (size_t)capacity * sizeof(float), sizeof(float) * (size_t)n[0]);
    }
    e->outLastCount.assign(n, n + 2);
    e->outLastCapacity = capacity;
    e->outLastFrames = 2;
    if (overflow) return fail(ORBFE_ERR_CAPACITY, "keypoint capacity too small");
  }
  resolve_stage_times(e);
  return ORBFE_OK;
}

// ---- host-logic debug entry points (no GPU needed; CPU tests compare them with the oracle) ----

// DistributeOctTree of the library's host implementation (octree_host.cpp) on flat arrays:
// candidates (x,y relative to minBorder, response) in emission order -> selected (x,y level
// coordinates, response) in list order.  Returns the count or a negative status.
extern "C" int orbfe_debug_octree_host(const uint16_t* xs, const uint16_t* ys, const uint8_t* resp, int n, int minX,
                                       int maxX, int minY, int maxY, int N, uint16_t* out_x, uint16_t* out_y,
                                       uint8_t* out_resp, int cap) {
  if (n < 0 || cap < 0 || (n > 0 && (!xs || !ys || !resp)) || (cap > 0 && (!out_x || !out_y || !out_resp)))
    return fail(ORBFE_ERR_INVALID, "debug_octree_host: bad argument");
  std::vector<Candidate> cand((size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { cand[i].xy = (uint32_t)xs[i] | ((uint32_t)ys[i] << 16); cand[i].score = resp[i]; }
  std::vector<LevelKp> out((size_t)(cap > 0 ? cap : 1));
  const int k = distribute_octree_host(cand.data(), n, minX, maxX, minY, maxY, N, out.data(), cap);
  for (int i = 0; i < k && i < cap; i++) { out_x[i] = out[i].x; out_y[i] = out[i].y; out_resp[i] = (uint8_t)out[i].score; }
  return k;
}

// Geometry tables the pipeline derives on the host for a WxH input: per level (w, h, nCols, nRows,
// wCell, hCell, nCells, quota, nIni) as 9 ints, and the FAST grid cells as (level, x0, y0, w, h).
// Returns the number of cells (cells may be NULL to query the count).
extern "C" int orbfe_debug_geometry(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                                    int width, int height, int32_t* levels9, float* tables4, int16_t* cells5,
                                    int cell_cap) {
  if (width <= 0 || height <= 0 || !levels9 || nlevels < 1 || nlevels > ORBFE_MAX_LEVELS)
    return fail(ORBFE_ERR_INVALID, "debug_geometry: bad argument");
  ExtractorTables tab;
  tab.init(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
  if (tables4)
    for (int l = 0; l < tab.nlevels; l++) {
      tables4[l * 4 + 0] = tab.scale[l]; tables4[l * 4 + 1] = tab.invScale[l];
      tables4[l * 4 + 2] = tab.sigma2[l]; tables4[l * 4 + 3] = tab.invSigma2[l];
    }
  FrameGeom g;
  g.build(tab, width, height);
  for (int l = 0; l < g.nlevels; l++) {
    const LevelGeom& v = g.lv[l];
    const int32_t row[9] = {v.w, v.h, v.nCols, v.nRows, v.wCell, v.hCell, v.nCells, v.quota, v.nIni};
    for (int k = 0; k < 9; k++) levels9[l * 9 + k] = row[k];
  }
  if (cells5)
    for (int i = 0; i < g.nFastCells && i < cell_cap; i++) {
      const CellDesc& c = g.cells[i];
      cells5[i * 5 + 0] = c.level; cells5[i * 5 + 1] = c.x0; cells5[i * 5 + 2] = c.y0; cells5[i * 5 + 3] = c.w; cells5[i * 5 + 4] = c.h;
    }
  return g.nFastCells;
}

// Ownership tables of the fused blur + resize kernel (ResizeTables::tileGx / tileDy) with the window start of every
// 4-column group and the upper source row of every output row, for the host-logic test.  n_tiles_x / n_tiles_y come
// back 0 when the fused kernel is not used for this pair of sizes.
extern "C" int orbfe_debug_resize_tiles(int sw, int sh, int dw, int dh, int32_t* tile_gx, int* n_tiles_x, int32_t* tile_dy,
                                        int* n_tiles_y, int32_t* group_start, int32_t* row_upper) {
  if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || !tile_gx || !n_tiles_x || !tile_dy || !n_tiles_y || !group_start || !row_upper)
    return fail(ORBFE_ERR_INVALID, "debug_resize_tiles: bad argument");
  ResizeTables t;
  build_resize_tables(sw, sh, dw, dh, &t);
  *n_tiles_x = *n_tiles_y = 0;
  if (t.tileGx.empty()) return ORBFE_OK;
  *n_tiles_x = (int)t.tileGx.size() - 1;
  *n_tiles_y = (int)t.tileDy.size() - 1;
  std::memcpy(tile_gx, t.tileGx.data(), t.tileGx.size() * 4);   // caller: (sw + 63) / 64 + 1 entries
  std::memcpy(tile_dy, t.tileDy.data(), t.tileDy.size() * 4);   // caller: (sh + 63) / 64 + 1 entries
  const int ngx = (dw + 3) / 4;
  for (int g = 0; g < ngx; g++) group_start[g] = (int32_t)t.colrec[12 * (size_t)g + 8];
  for (int dy = 0; dy < dh; dy++) row_upper[dy] = (int32_t)t.rowrec[4 * (size_t)dy];
  return ORBFE_OK;
}

// cv::resize coefficient tables (xofs, alpha pairs, yofs, beta pairs) the resize kernel uses.
extern "C" int orbfe_debug_resize_tables(int sw, int sh, int dw, int dh, int32_t* xofs, int16_t* alpha, int32_t* yofs,
                                         int16_t* beta) {
  if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || !xofs || !alpha || !yofs || !beta)
    return fail(ORBFE_ERR_INVALID, "debug_resize_tables: bad argument");
  ResizeTables t;
  build_resize_tables(sw, sh, dw, dh, &t);
  std::memcpy(xofs, t.xofs.data(), t.xofs.size() * 4);
  std::memcpy(alpha, t.alpha.data(), t.alpha.size() * 2);
  std::memcpy(yofs, t.yofs.data(), t.yofs.size() * 4);
  std::memcpy(beta, t.beta.data(), t.beta.size() * 2);
  return ORBFE_OK;
}
