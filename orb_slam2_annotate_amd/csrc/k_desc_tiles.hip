// k_desc_tiles.hip -- the per-keypoint tail of ORBextractor::operator() -- IC_Angle orientation
// (src/ORBextractor.cc:78-106) on the UNBLURRED level, the 256-bit steered BRIEF descriptor (:111-152) on the BLURRED
// level, the final cv::KeyPoint record (:905-916, :1187-1195) -- in TILE form (round 3; the throughput form).
//
// k_orient_desc (k_desc.hip) gathers per keypoint: 31 rows x 32 bytes of the level for the moments and 37 rows x 48
// bytes of the blurred level for the descriptor, i.e. ~90 cache-line requests per keypoint for 1261 useful bytes; the
// octree spreads the keypoints evenly, so from level 1 up every pixel of a level lies in 2-5 patches and the same
// lines are requested again and again (round 2: 60 L1->L2 requests and 2.1 KB of HBM traffic per keypoint, the L1
// stalled on its request queue 56 % of the time, the waves waiting 59 %).
//
// Here a workgroup owns a 128 x 128-pixel TILE of one pyramid level of one frame.  It stages the tile plus an 18-px
// rim (164 rows x 176 bytes, 28.9 KB of LDS) ONCE with full 16-byte requests on consecutive addresses, and every
// keypoint whose centre lies in the tile takes its moments from that LDS image; the blurred tile -- requested into
// registers BEFORE the moments are evaluated, so its latency hides behind them -- then replaces it, and the descriptors
// are sampled from LDS.  Per frame the level data is requested 2 x 1.6 times (the rims), always whole lines, instead
// of ~8 times in 32/48-byte pieces; rims shared by neighbouring tiles meet in the XCD's L2 (all tiles of a frame run on
// one XCD).  There is no dependent memory chain per keypoint any more: a tile is three load rounds (keypoint list +
// level tile, blurred tile, -) whatever the number of keypoints in it.
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace orbfe {

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };  // 16-byte load at any alignment

constexpr int kTW = 128, kTH = 128;      // tile core
constexpr int kRim = 18;                 // rows above / below; the steered pattern stays inside radius sqrt(13^2+13^2) < 18.5
constexpr int kRimL = 24;                // columns left of the core (a multiple of 8 >= 18)
constexpr int kPitch = 176;              // LDS row pitch: 24 + 128 + 18 = 170 -> 11 x 16 bytes
constexpr int kParts = kPitch / 16;
constexpr int kRows = kTH + 2 * kRim;    // 164
constexpr int kStageIters = (kRows * kParts + 255) / 256;  // 16-byte pieces per thread (8)
constexpr int kMaxList = 64;             // keypoints of a tile handled per pass (more: further passes)

__device__ __forceinline__ int wave_sum(int x) {  // DPP wave64 sum, total read from lane 63 into a scalar register
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(x, 63);
}

// 16-byte pieces of the tile: piece q = tid + 256 k -> (row q / 11, part q % 11); voff[k] = row * pitch + 16 * part.
// Rows <= h-2 and piece starts <= w-1: every byte read lies inside the frame's rows (a piece may run into the next
// row, never past row h-1) whatever the pitch is -- also for caller-owned level-0 frames with stride == width.
// Pieces outside that are not requested; their registers (and LDS bytes) hold garbage nobody reads.
__device__ __forceinline__ void tile_request(const uint8_t* base /* block-uniform: frame, row ry0, column cx0 */, int pitch,
                                             int nrows, int colsLeft /* w - cx0 */, int tid, uint4 (&r)[kStageIters]) {
#pragma unroll
  for (int k = 0; k < kStageIters; k++) {
    const int q = tid + 256 * k;
    const int row = (q * 745) >> 13, part = q - kParts * row;  // q / 11 for q < 2048
    if (row < nrows && 16 * part < colsLeft) {
      const U4u v = *reinterpret_cast<const U4u*>(base + (uint32_t)(row * pitch + 16 * part));
      r[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
}
__device__ __forceinline__ void tile_store(uint32_t* tile, int tid, const uint4 (&r)[kStageIters]) {
#pragma unroll
  for (int k = 0; k < kStageIters; k++) {
    const int q = tid + 256 * k;
    if (q < kRows * kParts) *reinterpret_cast<uint4*>(&tile[4 * q]) = r[k];
  }
}
}  // namespace

// A workgroup walks its share of the items as a software pipeline -- at every moment one load round is in flight:
//   scan the item's keypoint list (records requested one item ahead)     | level tile of THIS item arriving (requested one item ahead)
//   store the level tile to LDS, request the BLURRED tile                |
//   moments of the tile's keypoints from LDS, angle + sincos per wave    | blurred tile arriving
//   store the blurred tile to LDS, request the NEXT item's level tile and list records
//   descriptors from LDS, output records                                 | next item's level tile arriving
__global__ __launch_bounds__(256) void k_orient_desc_tiles(OrientDescArgs a, const DescTile* __restrict__ tiles,
                                                           int tilesPerFrame, int nFrames, uint32_t tilesMagic,
                                                           const LevelKp* __restrict__ levelKp,
                                                           const int32_t* __restrict__ levelCount,
                                                           const float4* __restrict__ patternF,
                                                           const uint4* __restrict__ momentTab,
                                                           const int32_t* __restrict__ umax, float* __restrict__ kpOut,
                                                           uint8_t* __restrict__ descOut, int32_t* __restrict__ nOut,
                                                           int ablate /* timing experiments only ($ORBFE_DESC_TILES_ABLATE) */) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tile[kRows * kPitch / 4];
  __shared__ uint32_t s_kxy[kMaxList];      // x | y << 16
  __shared__ uint32_t s_ksr[kMaxList];      // score | rank << 16
  __shared__ int s_m10[kMaxList], s_m01[kMaxList];
  __shared__ float s_angle[kMaxList], s_cos[kMaxList], s_sin[kMaxList];
  __shared__ int s_waveCnt[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // lane roles of the moment pass: (disc row, left / right half)
  const int mrow = lane >> 1, mhalf = lane & 1, mdy = mrow - 15;
  const bool mactive = mrow < 31;
  uint4 wt = make_uint4(0, 0, 0, 0), mk = make_uint4(0, 0, 0, 0);
  if (mactive) {
    const int d = umax[mdy < 0 ? -mdy : mdy];
    wt = momentTab[(d * 2 + mhalf) * 2];
    mk = momentTab[(d * 2 + mhalf) * 2 + 1];
  }
  const int moff = mrow * kPitch + 16 * mhalf - 15 * kPitch - 15;  // lane's byte offset from the keypoint's centre
  float4 P[4];
#pragma unroll
  for (int j = 0; j < 4; j++) P[j] = patternF[lane + 64 * j];  // (x0, x1, y0, y1) of test lane + 64 j

  // XCD-aware work mapping (as k_orient_desc): workgroups are dealt round-robin over the 8 XCDs, so those with
  // blockIdx % 8 == x walk the contiguous chunk x of the items (item = frame * tilesPerFrame + tile): the tiles of a
  // frame run on ONE XCD, neighbouring tiles at the same time, and its L2 serves the rims they share.
  const unsigned totalWork = (unsigned)tilesPerFrame * (unsigned)nFrames;
  const unsigned chunkW = (totalWork + 7u) >> 3;
  const unsigned perXcd = gridDim.x >> 3;
  const unsigned chunk0 = (blockIdx.x & 7u) * chunkW;
  unsigned work = chunk0 + (blockIdx.x >> 3);
  unsigned chunkEnd = chunk0 + chunkW;
  if (chunkEnd > totalWork) chunkEnd = totalWork;
  if (work >= chunkEnd) return;

  uint4 stage[kStageIters];
  uint2 pre0, pre1;  // list records tid, tid + 256 of the item ahead
  // level tile + the first 512 list records of item w (everything derived from w is block-uniform: scalar registers)
  auto request_item = [&](unsigned w) {
    const int f = (int)udiv_magic(w, (uint32_t)tilesPerFrame, tilesMagic);
    const DescTile tl = tiles[(int)(w - (unsigned)f * (unsigned)tilesPerFrame)];
    const LevelView lv = a.pyr.lv[tl.level];
    tile_request(lv.base + (size_t)f * lv.frameStride + (size_t)tl.ry0 * lv.pitch + tl.cx0, lv.pitch,
                 (ablate & 16) ? 0 : tl.nrows, lv.w - tl.cx0, tid, stage);
    const int count = levelCount[(size_t)f * a.nlevels + tl.level];
    const uint2* list = reinterpret_cast<const uint2*>(levelKp + (size_t)f * a.kpSlotsPerFrame + a.kpStart[tl.level]);
    if (tid < count) pre0 = list[tid];
    if (tid + 256 < count) pre1 = list[tid + 256];
  };
  request_item(work);

  for (;;) {
    const int f = (int)udiv_magic(work, (uint32_t)tilesPerFrame, tilesMagic);
    const DescTile tl = tiles[(int)(work - (unsigned)f * (unsigned)tilesPerFrame)];
    const int l = tl.level, cx0 = tl.cx0, ry0 = tl.ry0;
    const int32_t* cnt = levelCount + (size_t)f * a.nlevels;
    const int count = cnt[l];
    int outBase = 0;
    for (int k = 0; k < l; k++) outBase += cnt[k];
    if (tl.first && tid == 0) {  // the frame's first tile reports the keypoint count
      int tot = 0;
      for (int k = 0; k < a.nlevels; k++) tot += cnt[k];
      nOut[f] = tot;  // the host reports ORBFE_ERR_CAPACITY when this exceeds the capacity
    }
    const uint2* list = reinterpret_cast<const uint2*>(levelKp + (size_t)f * a.kpSlotsPerFrame + a.kpStart[l]);
    const unsigned nextWork = work + perXcd;
    const bool haveNext = nextWork < chunkEnd;  // block-uniform

    for (int win = 0, total = 1; win < total; win += kMaxList) {  // one pass unless a tile holds > kMaxList keypoints
      if (win > 0) request_item(work);  // (rare) further passes stage the level tile again
      // ---- the keypoints of this tile, in list order (ordered compaction: deterministic windows) ----
      int running = 0;
      for (int i0 = 0; i0 < count; i0 += 256) {
        const int i = i0 + tid;
        uint2 rec = make_uint2(0, 0);
        if (i0 == 0) rec = pre0;
        else if (i0 == 256) rec = pre1;
        else if (i < count) rec = list[i];
        const int x = (int)(rec.x & 0xffffu), y = (int)(rec.x >> 16);
        const bool match = i < count && (unsigned)(x - tl.x0) < (unsigned)kTW && (unsigned)(y - tl.y0) < (unsigned)kTH;
        const unsigned long long bal = __ballot(match);
        if (lane == 0) s_waveCnt[wave] = __popcll(bal);
        __syncthreads();
        int ord = running + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++) ord += s_waveCnt[w];
        running += s_waveCnt[0] + s_waveCnt[1] + s_waveCnt[2] + s_waveCnt[3];
        if (match && ord >= win && ord < win + kMaxList) { s_kxy[ord - win] = rec.x; s_ksr[ord - win] = rec.y; }
        __syncthreads();
      }
      total = running;
      int n = total - win;
      if (n > kMaxList) n = kMaxList;
      const bool lastWin = win + kMaxList >= total;
      if (n <= 0) {  // block-uniform: an empty tile costs its list scan only
        if (haveNext) request_item(nextWork);
        break;
      }
      tile_store(s_tile, tid, stage);
      __syncthreads();
      // ---- the blurred tile: on its way while the moments are evaluated ----
      {
        const LevelView bl = a.blur.lv[l];
        tile_request(bl.base + (size_t)f * bl.frameStride + (size_t)ry0 * bl.pitch + cx0, bl.pitch,
                     (ablate & 8) ? 0 : tl.nrows, bl.w - cx0, tid, stage);
      }

      // ---- 1. intensity-centroid moments over the 749-pixel disc, one wavefront per keypoint, two keypoints in flight ----
      const int origin = -ry0 * kPitch - cx0;  // byte offset of level pixel (0, 0) in the tile
      for (int j0 = wave; j0 < ((ablate & 1) ? 0 : n); j0 += 8) {
        int sW[2], sI[2];
        uint32_t d[2][5], mis[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int j = j0 + 4 * u < n ? j0 + 4 * u : j0;
          const uint32_t xy = __builtin_amdgcn_readfirstlane(s_kxy[j]);
          const int off = (int)(xy >> 16) * kPitch + (int)(xy & 0xffffu) + origin + (mactive ? moff : 0);
          const uint32_t* p = &s_tile[off >> 2];
          mis[u] = (uint32_t)(off & 3);
#pragma unroll
          for (int k = 0; k < 5; k++) d[u][k] = p[k];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const uint32_t q0 = __builtin_amdgcn_alignbyte(d[u][1], d[u][0], mis[u]), q1 = __builtin_amdgcn_alignbyte(d[u][2], d[u][1], mis[u]);
          const uint32_t q2 = __builtin_amdgcn_alignbyte(d[u][3], d[u][2], mis[u]), q3 = __builtin_amdgcn_alignbyte(d[u][4], d[u][3], mis[u]);
          unsigned w = __builtin_amdgcn_udot4(q0, wt.x, 0u, false);
          w = __builtin_amdgcn_udot4(q1, wt.y, w, false);
          w = __builtin_amdgcn_udot4(q2, wt.z, w, false);
          w = __builtin_amdgcn_udot4(q3, wt.w, w, false);
          unsigned sm = __builtin_amdgcn_udot4(q0, mk.x, 0u, false);
          sm = __builtin_amdgcn_udot4(q1, mk.y, sm, false);
          sm = __builtin_amdgcn_udot4(q2, mk.z, sm, false);
          sm = __builtin_amdgcn_udot4(q3, mk.w, sm, false);
          sW[u] = (int)w - 16 * (int)sm;  // sum(dx * I); lanes 62, 63 have zero weights and masks
          sI[u] = mdy * (int)sm;          // sum(dy * I)
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int m10 = wave_sum(sW[u]), m01 = wave_sum(sI[u]);
          if (lane == 0 && j0 + 4 * u < n) { s_m10[j0 + 4 * u] = m10; s_m01[j0 + 4 * u] = m01; }
        }
      }
      // ---- 2. angle and rotation: lane i of wave w takes keypoint w + 4 i -- the moments this wave itself just wrote,
      //         so no workgroup barrier in between, and the four waves evaluate their sincos side by side ----
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const int j = wave + 4 * lane;
        if (j < n) {
          // (ablation bits 1 / 2, timing experiments: no moments / no atan2 + sincos -- the lists still get DEFINED values, so
          // the descriptor loop samples inside the tile and downstream kernels read real records)
          float angle = 0.f, ca = 1.f, sb = 0.f;
          if (!(ablate & 3)) {
            angle = fast_atan2((float)s_m01[j], (float)s_m10[j]);
            const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
            sincos_spec(__fmul_rn(angle, factorPI), &ca, &sb);
          } else if (!(ablate & 2)) {  // bit 1 only: the arithmetic on a fixed moment pair
            angle = fast_atan2(1.0f, 1.0f);
            const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
            sincos_spec(__fmul_rn(angle, factorPI), &ca, &sb);
          }
          s_angle[j] = angle;
          s_cos[j] = ca;
          s_sin[j] = sb;
        }
      }
      __syncthreads();  // moments done everywhere: the level tile is dead
      tile_store(s_tile, tid, stage);  // the blurred tile replaces it
      __syncthreads();
      if (lastWin && haveNext) request_item(nextWork);  // next item's level tile + list records: in flight during the descriptors

      // ---- 3. steered BRIEF from the LDS tile + output records, two keypoints in flight ----
      const uint8_t* tbytes = reinterpret_cast<const uint8_t*>(s_tile);
      const float sc = a.scale[l], ksz = a.kpSize[l];
      // (ablation bit 4 removes the 512 samples of a keypoint, NOT the loop: it also writes the cv::KeyPoint records that
      // k_stereo_match_batch / the BoW matcher read behind this kernel -- round 3's three memory-access faults were this
      // loop skipped as a whole, the records left at the caller's zeros and the stereo SAD reading rows -5.. of level 0)
      for (int j0 = wave; j0 < n; j0 += 8) {
        uint32_t xy[2], sr[2];
        float ca[2], sb[2];
        int v0[2][4], v1[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int j = j0 + 4 * u < n ? j0 + 4 * u : j0;
          xy[u] = __builtin_amdgcn_readfirstlane(s_kxy[j]);
          sr[u] = __builtin_amdgcn_readfirstlane(s_ksr[j]);
          ca[u] = s_cos[j];
          sb[u] = s_sin[j];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          // cvRound by the magic-number add: for |v| < 2^22, float(v + 1.5*2^23) has the bit pattern 0x4B400000 + RNE(v)
          // (round-half-even, like cvRound).  The row keeps its low 24 bits (0x400000 + row: v_mad_u32_u24), the column its
          // full pattern; both biases are folded into the base offset.
          constexpr float kMagic = 12582912.0f;
          constexpr uint32_t kBias = 0x400000u * (uint32_t)kPitch + 0x4B400000u;
          const uint32_t baseK = (uint32_t)((int)(xy[u] >> 16) * kPitch + (int)(xy[u] & 0xffffu) + origin) - kBias;
          // row = x*b + y*a, col = x*a - y*b (src/ORBextractor.cc:123-125 with a = cos, b = sin), every product and sum
          // rounded on its own (no FMA); P[t] = (x0, x1, y0, y1) of test t
#pragma unroll
          for (int t4 = 0; t4 < 4; t4++) {
            const float r0 = __fadd_rn(__fadd_rn(__fmul_rn(P[t4].x, sb[u]), __fmul_rn(P[t4].z, ca[u])), kMagic);
            const float r1 = __fadd_rn(__fadd_rn(__fmul_rn(P[t4].y, sb[u]), __fmul_rn(P[t4].w, ca[u])), kMagic);
            const float c0 = __fadd_rn(__fsub_rn(__fmul_rn(P[t4].x, ca[u]), __fmul_rn(P[t4].z, sb[u])), kMagic);
            const float c1 = __fadd_rn(__fsub_rn(__fmul_rn(P[t4].y, ca[u]), __fmul_rn(P[t4].w, sb[u])), kMagic);
            v0[u][t4] = (ablate & 4) ? 0 : tbytes[__umul24(__float_as_uint(r0), (uint32_t)kPitch) + __float_as_uint(c0) + baseK];
            v1[u][t4] = (ablate & 4) ? 0 : tbytes[__umul24(__float_as_uint(r1), (uint32_t)kPitch) + __float_as_uint(c1) + baseK];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int outIdx = outBase + (int)(sr[u] >> 16);
          if (j0 + 4 * u >= n || outIdx >= a.outCapacity) continue;  // wave-uniform
          unsigned long long bits = 0;
#pragma unroll
          for (int t4 = 0; t4 < 4; t4++) {
            const unsigned long long bt = __ballot(v0[u][t4] < v1[u][t4]);
            if (lane == t4) bits = bt;
          }
          // the four ballots are the 32 descriptor bytes: lanes 0..3 store one 8-byte word each; lanes 0..6 store the
          // seven words of the cv::KeyPoint record
          unsigned long long* dout = reinterpret_cast<unsigned long long*>(descOut + ((size_t)f * a.outCapacity + outIdx) * 32);
          if (lane < 4) dout[lane] = bits;
          const int kx = (int)(xy[u] & 0xffffu), ky = (int)(xy[u] >> 16);
          uint32_t w = 0xffffffffu;  // class_id = -1 (lane 6)
          if (lane == 0) w = __float_as_uint(__fmul_rn((float)kx, sc));
          if (lane == 1) w = __float_as_uint(__fmul_rn((float)ky, sc));
          if (lane == 2) w = __float_as_uint(ksz);
          if (lane == 3) w = __float_as_uint(s_angle[j0 + 4 * u]);
          if (lane == 4) w = __float_as_uint((float)(sr[u] & 0xffffu));
          if (lane == 5) w = (uint32_t)l;
          if (lane < 7) reinterpret_cast<uint32_t*>(kpOut + ((size_t)f * a.outCapacity + outIdx) * 7)[lane] = w;
        }
      }
      __syncthreads();  // the next pass / item reuses the tile and the lists
    }
    if (!haveNext) break;
    work = nextWork;
  }
}

// Tiles of a frame: every 128 x 128 cell of every level that can hold a keypoint centre (19 <= x <= w-20, 19 <= y <= h-20,
// src/ORBextractor.cc:823-834), large levels first.
std::vector<DescTile> build_desc_tiles(const LevelGeom* lv, int nlevels) {
  std::vector<DescTile> out;
  for (int l = 0; l < nlevels; l++) {
    const int w = lv[l].w, h = lv[l].h;
    if (w < 39 || h < 39) continue;
    for (int ty = 19 / kTH; ty <= (h - 20) / kTH; ty++)
      for (int tx = 19 / kTW; tx <= (w - 20) / kTW; tx++) {
        DescTile t;
        t.level = l; t.x0 = (tx * kTW); t.y0 = (ty * kTH);
        t.cx0 = (t.x0 - kRimL > 0 ? t.x0 - kRimL : 0);
        t.ry0 = (t.y0 - kRim > 0 ? t.y0 - kRim : 0);
        int ry1 = t.y0 + kTH - 1 + kRim;
        if (ry1 > h - 2) ry1 = h - 2;
        t.nrows = (ry1 - t.ry0 + 1);
        t.first = (out.empty() ? 1 : 0);
        t.pad = 0;
        out.push_back(t);
      }
  }
  return out;
}

void launch_orient_desc_tiles(hipStream_t s, const OrientDescArgs& a, const DescTile* d_tiles, int tilesPerFrame,
                              const LevelKp* d_levelKp, const int32_t* d_levelCount, const float4* d_patternF,
                              const uint4* d_momentTab, const int32_t* d_umax, int nFrames, void* d_kpOut,
                              uint8_t* d_descOut, int32_t* d_nOut, int concurrentLaunches) {
  if (nFrames <= 0 || tilesPerFrame <= 0) return;
  const unsigned total = (unsigned)tilesPerFrame * (unsigned)nFrames;
  const unsigned full = (total + 7u) / 8u * 8u;
  // persistent grid: as many workgroups as fit on the chip at once (4 per CU by LDS), each walking its XCD's items as a
  // software pipeline; $ORBFE_DESC_TILES_GRID = workgroups per CU
  static const int kGridEnv = [] {
    if (getenv("ORBFE_DESC_TILES_GRID")) return atoi(getenv("ORBFE_DESC_TILES_GRID"));
    int perCu = 0;  // what the registers and the LDS of the kernel allow: more would queue behind the resident ones
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_orient_desc_tiles, 256, 0) != hipSuccess || perCu < 1) perCu = 3;
    return perCu;
  }();
  static const int kAblate = getenv("ORBFE_DESC_TILES_ABLATE") ? atoi(getenv("ORBFE_DESC_TILES_ABLATE")) : 0;
  unsigned grid = full;
  if (kGridEnv > 0 && (unsigned)kGridEnv * 256u < full) grid = (unsigned)kGridEnv * 256u;
  (void)concurrentLaunches;
  hipLaunchKernelGGL(k_orient_desc_tiles, dim3(grid), dim3(256), 0, s, a, d_tiles, tilesPerFrame, nFrames,
                     udiv_magic_multiplier((uint32_t)tilesPerFrame), d_levelKp, d_levelCount, d_patternF, d_momentTab, d_umax,
                     (float*)d_kpOut, d_descOut, d_nOut, kAblate);
}

}  // namespace orbfe
