// kernels.h -- launch wrappers of the hand-written gfx950 kernels (one .hip file per stage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "orb_params.h"

namespace orbfe {

// n / d for a run-time invariant d with the host-built multiplier M = floor(2^32 / d): the estimate
// umulhi(n, M) is q or q-1 for EVERY 32-bit n, one compare fixes it (block-uniform operands: scalar ALU).
#if defined(__HIPCC__)
// ---- wave64 scans and reductions in the DPP network (row shifts / rotations inside the 16-lane rows, then the two row
// broadcasts; checked on gfx950 by tools/ubench/dpp_scan.hip).  __shfl_up / __shfl_xor compile to ds_bpermute_b32 -- a trip
// through the LDS crossbar, ~100 cycles of latency per step of a dependent six-step chain; a DPP step is a VALU operand
// modifier.  Round 4 found 216 ds_bpermute in k_octree and 273 in the matcher kernels: every scan and min-reduction of the
// latency-bound kernels went that way. ----
__device__ __forceinline__ int wave_incl_scan_dpp(int x) {  // inclusive prefix sum over the 64 lanes
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return x;
}
__device__ __forceinline__ int wave_sum_dpp(int x) { return __builtin_amdgcn_readlane(wave_incl_scan_dpp(x), 63); }  // -> every lane (scalar)
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {  // -> every lane (scalar)
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:1
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:2
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:4
  t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:8: every lane = its row's minimum
  t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;      // rows 1, 3 += rows 0, 2
  t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;      // rows 2, 3 += rows 0..1
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_max_i32_dpp(int v) {  // -> every lane (scalar)
  int t;
  t = __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
  return __builtin_amdgcn_readlane(v, 63);
}
// The two smallest distances of the wave and the position of the first minimum (SearchByBoW's best / second best): every lane
// brings key1 = (distance << 16 | position) of its own minimum and best2 = its own second-smallest distance; merging two
// disjoint sets keeps the smaller key and, as second, the smallest of (the other key's distance, the two seconds).
__device__ __forceinline__ void wave_top2_dpp(uint32_t& key1, uint32_t& best2) {  // -> every lane (scalar)
#define ORBFE_TOP2_STEP(CTRL, RMASK, IDK, IDB)                                                                   \
  {                                                                                                             \
    const uint32_t ok_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDK), (int)key1, CTRL, RMASK, 0xf, false);  \
    const uint32_t ob_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(IDB), (int)best2, CTRL, RMASK, 0xf, false); \
    const uint32_t lo_ = ok_ < key1 ? ok_ : key1, hi_ = ok_ < key1 ? key1 : ok_;                                 \
    const uint32_t m2_ = ob_ < best2 ? ob_ : best2;                                                              \
    key1 = lo_;                                                                                                  \
    best2 = (hi_ >> 16) < m2_ ? (hi_ >> 16) : m2_;                                                               \
  }
  // rotations inside a row: after step k a lane holds the top two of 2^k consecutive lanes of its row (disjoint halves)
  ORBFE_TOP2_STEP(0x121, 0xf, key1, best2)
  ORBFE_TOP2_STEP(0x122, 0xf, key1, best2)
  ORBFE_TOP2_STEP(0x124, 0xf, key1, best2)
  ORBFE_TOP2_STEP(0x128, 0xf, key1, best2)
  // row broadcasts: the lanes that receive nothing merge with the identity (no candidate: distance 0xffff)
  ORBFE_TOP2_STEP(0x142, 0xa, 0xffffffffu, 0xffffu)
  ORBFE_TOP2_STEP(0x143, 0xc, 0xffffffffu, 0xffffu)
#undef ORBFE_TOP2_STEP
  key1 = (uint32_t)__builtin_amdgcn_readlane((int)key1, 63);
  best2 = (uint32_t)__builtin_amdgcn_readlane((int)best2, 63);
}

__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t M) {
  uint32_t q = __umulhi(n, M);
  if (n - q * d >= d) q++;
  return q;
}
#endif
// Wave issue priority of the latency-bound kernels (round 4).  The pipeline co-runs VALU-bound kernels (FAST, blur: 5-8
// waves per SIMD, 0.55-0.87 VALU busy) with kernels that mostly wait -- k_orient_desc with ONE wave per SIMD, the octree, the
// stereo matcher.  At equal priority the lone waiting wave queues for issue behind every busy wave each time it wakes up, its
// kernel runs several times longer than alone (k_orient_desc: 170 us alone, 600-900 us in the pipeline) and holds its stream's
// chain and hardware queue for that long.  s_setprio 3 lets such a wave issue as soon as it is ready: it has few instructions to
// issue, so the busy kernels lose little.  -DORBFE_SETPRIO=0 builds without (same-box A/B in DESIGN.md 4).
#ifndef ORBFE_SETPRIO
#define ORBFE_SETPRIO 1
#endif
#if ORBFE_SETPRIO
#define ORBFE_LATENCY_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define ORBFE_LATENCY_KERNEL_PRIO() ((void)0)
#endif
inline uint32_t udiv_magic_multiplier(uint32_t d) {
  return d <= 1 ? 0xffffffffu : (uint32_t)((1ULL << 32) / d);  // d = 1: umulhi gives n-1, the compare adds the 1
}


// Occupancy caps for the latency-bound kernels: unused dynamic LDS per workgroup, in KB, so that at
// most 160 / (own + pad) workgroups of the kernel are resident on a CU and the VALU-bound kernels of the other
// sub-batch streams keep wave slots.  Defaults are the measured optimum; $ORBFE_PAD_<NAME> overrides (A/B sweeps).
size_t occupancy_pad_bytes(const char* name, int default_kb);

// A pyramid level of a batch of frames in HBM: frame f, row y starts at
// base + f*frameStride + y*pitch.  Row-major u8, pitch is a multiple of 64 for owned levels.
struct LevelView {
  const uint8_t* base;
  size_t frameStride;
  int pitch;
  int w, h;
};
struct LevelViewMut {
  uint8_t* base;
  size_t frameStride;
  int pitch;
  int w, h;
};

struct PyramidViews {
  LevelView lv[kMaxLevels];
  int nlevels;
};

// Candidate corner as the FAST grid stage emits it (src/ORBextractor.cc:884-893):
// xy = x | y<<16 relative to (minBorderX, minBorderY), score = cv::FAST response.
struct Candidate {
  uint32_t xy;
  uint32_t score;
};

// Keypoint selected by the octree, level coordinates (already + minBorder, :909-916).
struct LevelKp {
  uint16_t x, y;
  uint16_t score;  // FAST response (0..255)
  uint16_t rank;   // position in the reference's output order of the level (the slot order is spatial)
};

// ---- pyramid (ComputePyramid, :1203-1234) ----
// ---- the whole pyramid of a few frames in ONE launch (single-frame form, k_pyramid_chain) ----
// A workgroup owns one 32 x 32 tile of one level l >= 1 and produces it from LEVEL 0: it stages the level-0 rectangle its
// tile depends on and walks the chain level 1, 2, ... l in LDS, recomputing the (slightly larger) rectangle of every
// intermediate level -- the same fixed-point bilinear step from the same inputs, so every pixel equals the one the
// level-by-level kernels write.  Seven dependent launches of a few microseconds each (+ the gaps between them) become one.
struct ChainRect { int16_t x0, y0, w, h; };
struct ChainTile { int32_t level; ChainRect r[kMaxLevels]; };  // r[k], k = 0 .. level: what the tile needs of level k; r[level] = the tile
struct PyrChainArgs {
  LevelView l0;                       // level 0 (the caller's pitch)
  LevelViewMut lv[kMaxLevels];        // destinations, levels 1 .. nlevels-1
  const int32_t* xofs[kMaxLevels]; const int16_t* alpha[kMaxLevels];   // cv::resize tables of level k-1 -> k
  const int32_t* yofs[kMaxLevels]; const int16_t* beta[kMaxLevels];
  int w[kMaxLevels], h[kMaxLevels];   // level sizes
  const ChainTile* tiles;
  int bufA, bufB, maxW, maxH;         // LDS carve: bytes of the two rectangle buffers, longest rectangle side (table entries)
};
size_t pyramid_chain_lds_bytes(const PyrChainArgs& a);
void launch_pyramid_chain(hipStream_t s, const PyrChainArgs& a, int nTiles, int nFrames);
void launch_resize(hipStream_t s, LevelView src, LevelViewMut dst, const int32_t* d_xofs,
                   const int16_t* d_alpha, const int32_t* d_yofs, const int16_t* d_beta,
                   const uint32_t* d_colrec, const uint32_t* d_rowrec, int nFrames);
void launch_copy2d(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames);

// ---- FAST grid stage (:846-896) ----
void launch_fast_cells(hipStream_t s, PyramidViews pyr, const CellDesc* d_cells, int nCells,
                       int nFrames, int iniTh, int minTh, Candidate* d_slots, int slotsPerFrame,
                       uint16_t* d_cellCount, int maxCellW, int maxCellH, const PyramidViews* blurOut = nullptr,
                       int nCellsAll = 0, bool lowFirst = false, unsigned int* d_fallbackStat = nullptr);
// ordered compaction of the per-cell slots of each (frame, level) into d_cand
void launch_gather_candidates(hipStream_t s, const CellDesc* d_cells, const LevelGeom* d_lv,
                              int nlevels, int nFrames, const Candidate* d_slots,
                              int slotsPerFrame, const uint16_t* d_cellCount, int cellsPerFrame,
                              Candidate* d_cand, int32_t* d_candCount, int32_t* d_cellPrefix);

// ---- DistributeOctTree on the device (:566-808) ----
struct OctreeArgs {
  const Candidate* cand;      // ordered candidates, [frame][slotStart(level) + k]
  int slotsPerFrame;
  const int32_t* candCount;   // [frame][level]
  const LevelGeom* lvg;
  int nlevels;
  int nFrames;                // set by launch_octree
  uint16_t* nodeOf;           // scratch, same indexing as cand
  LevelKp* levelKp;           // out, [frame][kpStart(level) + i]
  int32_t* levelCount;        // out, [frame][level]
  int kpSlotsPerFrame;
  int maxL;                   // node capacity (>= every level's kpCap and nIni, multiple of 4)
  uint8_t* work;              // global-memory node lists, one slab of workStride bytes per (frame, level);
  size_t workStride;          //   only read when octree_lds_bytes(maxL) > kOctreeLdsLimit
  // the single-frame form gathers the level's candidates itself (octree_gathers(): k_gather_candidates's work at the head of
  // the (frame, level) workgroup -- one launch and its gap less in a chain that is nothing but launches): inputs of the gather
  const CellDesc* gCells; const Candidate* gSlots; const uint16_t* gCellCount; int gCellsPerFrame; int32_t* gCellPrefix;
  Candidate* gCand; int32_t* gCandCount;   // = cand / candCount, writable
};
bool octree_gathers(int nFrames, int maxL);  // launch_octree(..., nFrames) will run the form that gathers: fill the g* fields, skip launch_gather_candidates
constexpr size_t kOctreeLdsLimit = 150 * 1024;  // beyond it the node list lives in global memory (k_octree_global)
size_t octree_lds_bytes(int maxL);
hipError_t launch_octree(hipStream_t s, const OctreeArgs& a, int nlevels, int nFrames);

// ---- blur (GaussianBlur 7x7 sigma 2 reflect-101, :1169-1175) ----
void launch_blur7(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames, int spec = kBlurSpecCv4);
void launch_blur7_resize(hipStream_t s, LevelView src, LevelViewMut dst, LevelViewMut next, const uint32_t* d_colrec,
                         const uint32_t* d_rowrec, const int32_t* d_tileGx, const int32_t* d_tileDy, int nFrames, int spec);
void launch_blur7_levels(hipStream_t s, const LevelView* src, const LevelViewMut* dst, int nlevels, int nFrames,
                         int spec = kBlurSpecCv4);
int blur_pass_order();              // 1: horizontal pass on bytes first (default), 0: vertical packed-16 first; same bytes
void set_blur_pass_order(int order);

// ---- orientation + descriptor + final keypoint record (:78-152, :905-916, :1187-1195) ----
struct OrientDescArgs {
  PyramidViews pyr;   // unblurred levels (IC_Angle)
  PyramidViews blur;  // blurred levels (rBRIEF)
  int kpStart[kMaxLevels];
  int kpCap[kMaxLevels];
  float scale[kMaxLevels];
  float kpSize[kMaxLevels];
  int nlevels;
  int kpSlotsPerFrame;   // totalKpCap
  int outCapacity;       // caller's per-frame capacity
};
void launch_orient_desc(hipStream_t s, const OrientDescArgs& a, const LevelKp* d_levelKp,
                        const int32_t* d_levelCount, const float4* d_patternF, const uint4* d_momentTab,
                        const int32_t* d_umax, int nFrames, void* d_kpOut, uint8_t* d_descOut,
                        int32_t* d_nOut, int concurrentLaunches = 1 /* sub-batch streams of the call */);
void build_moment_table(uint8_t* tab /* 1024 bytes */);
// tile form (k_desc_tiles.hip): one workgroup per 128 x 128 tile of a level, level and blurred level staged in LDS once
struct DescTile {  // core origin (x0, y0), staged window origin (cx0, ry0) and rows, in level coordinates; 32-bit fields:
  int32_t level, x0, y0, cx0, ry0, nrows, first /* the frame's first tile */, pad;  // read with scalar loads
};
std::vector<DescTile> build_desc_tiles(const LevelGeom* lv, int nlevels);
void launch_orient_desc_tiles(hipStream_t s, const OrientDescArgs& a, const DescTile* d_tiles, int tilesPerFrame,
                              const LevelKp* d_levelKp, const int32_t* d_levelCount, const float4* d_patternF,
                              const uint4* d_momentTab, const int32_t* d_umax, int nFrames, void* d_kpOut,
                              uint8_t* d_descOut, int32_t* d_nOut, int concurrentLaunches = 1);

// ---- matching ----
void launch_hamming_pairs(hipStream_t s, const uint8_t* a, const uint8_t* b, int n, int32_t* out);
void launch_hamming_matrix(hipStream_t s, const uint8_t* d1, int n1, const uint8_t* d2, int n2,
                           int32_t* out);

}  // namespace orbfe
