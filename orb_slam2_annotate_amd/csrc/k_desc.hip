// k_desc.hip -- per-keypoint tail of ORBextractor::operator(): IC_Angle orientation
// (src/ORBextractor.cc:78-106) on the UNBLURRED level, the 256-bit steered BRIEF descriptor
// (:111-152) on the BLURRED level, and the final cv::KeyPoint record (:905-916,:1187-1195).
//
// A 256-thread workgroup owns 64 keypoint slots, 16 per wavefront; each wavefront takes its slots through three
// phases on its own (no workgroup barrier):
//   1. moments, one wavefront per keypoint: lane = (disc row, left/right half); each lane loads
//      its 16 pixels as aligned dwords and folds them with v_dot4_u32_u8 against per-row weight
//      bytes ((dx+16) inside the disc, 0 outside) and mask bytes -> m10 = sum(w*I) - 16*sum(I),
//      m01 = dy*sum(I); integer, order-free, DPP reduction across the 62 lanes;
//   2. one LANE per keypoint of the wave: cv::fastAtan2 and the double-precision sincos_spec, so the
//      transcendental part is not replicated across 64 lanes;
//   3. descriptors, one wavefront per keypoint: the 37x48-byte blurred patch is staged in LDS with
//      16-byte requests, lane l evaluates tests l, l+64, l+128, l+192 on it (rotation in plain
//      fp32 mul/add, no FMA); a wave ballot IS 8 descriptor bytes.
#include <cstdlib>

#include "kernels.h"

namespace orbfe {

namespace {
struct __attribute__((aligned(4))) U4 { uint32_t x, y, z, w; };  // 16-byte load at 4-byte alignment
// blurred patch of one keypoint staged in LDS: rows y-18..y+18, 48 bytes from the 4-byte aligned
// column ws <= x-18 (the steered pattern stays inside radius sqrt(13^2+13^2) < 18.5)
constexpr int kPatchRows = 37, kPatchDw = 12;

// wave64 sum in the DPP network (row shifts, then the two row broadcasts); the total is read from lane 63
// into a scalar register -- no LDS permutes, no per-step address arithmetic
__device__ __forceinline__ int wave_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(x, 63);
}
// LDS written by some lanes of a wave, read by others of the SAME wave: order and visibility without a workgroup barrier
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

// kKpPerBlock = 64 for throughput (16 keypoints per wavefront, 16 workgroups per VGA frame), 16 for
// launches of a few frames (4 per wavefront, 4x the workgroups: the serial chain per wave is 4x shorter).
// kUM / kUD: keypoints whose row loads (moments) / patch loads (descriptors) a wave keeps in flight at once (4 / 2;
// 8 / 4 measured no faster, see launch_orient_desc).
// kAbl: the $ORBFE_ORIENT_ABLATE hooks are compiled in (tools/ablate_orient.sh).  As a run-time argument alone the
// sampling switch (bit 4) put TWO scalar branches around every one of the 16 samples a lane takes per keypoint -- 32 taken
// or skipped branches and ~80 SALU instructions per keypoint in a kernel whose cost IS its per-wave chain -- so the
// product build instantiates kAbl = false, where `ablate` is the constant 0.
template <int kKpPerBlock, int kUM, int kUD, bool kAbl>
__global__ __launch_bounds__(256) void k_orient_desc(OrientDescArgs a,
                                                     const LevelKp* __restrict__ levelKp,
                                                     const int32_t* __restrict__ levelCount,
                                                     const float4* __restrict__ patternF,
                                                     const uint4* __restrict__ momentTab,
                                                     const int32_t* __restrict__ umax,
                                                     float* __restrict__ kpOut,
                                                     uint8_t* __restrict__ descOut,
                                                     int32_t* __restrict__ nOut, int blocksPerFrame,
                                                     int nFrames, uint32_t blocksMagic,
                                                     int ablateArg /* 0; $ORBFE_ORIENT_ABLATE, timing experiments: 1 no moment loads, 2 no patch loads, 4 no sampling */,
                                                     int interleave) {
  const int ablate = kAbl ? ablateArg : 0;
  __shared__ int s_m10[kKpPerBlock], s_m01[kKpPerBlock];
  __shared__ int s_x[kKpPerBlock], s_y[kKpPerBlock], s_level[kKpPerBlock], s_out[kKpPerBlock];
  __shared__ unsigned s_score[kKpPerBlock];
  __shared__ float s_angle[kKpPerBlock], s_cos[kKpPerBlock], s_sin[kKpPerBlock];
  __shared__ __attribute__((aligned(16))) uint32_t s_patch[4 * kUD * kPatchRows * kPatchDw];  // per wave: kUD patches

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ORBFE_LATENCY_KERNEL_PRIO();
  // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so the workgroups with
  // blockIdx % 8 == x walk the contiguous chunk x of the work items (work = frame * blocksPerFrame + slot chunk):
  // all workgroups of a frame run on ONE XCD, back to back, and that L2 fetches every line of the frame's pyramid
  // once however many keypoint patches overlap it.
  // PERSISTENT form (gridDim.x < 8 * chunkW): a workgroup takes every (gridDim.x / 8)-th item of its chunk.  The
  // number of workgroups in flight -- what decides how many frames' patches compete for an L2 -- is then the grid
  // size, not an LDS reservation, and the LDS and wave slots the kernel does not use stay free for the kernels of the
  // other streams (round 1 capped the occupancy with 23 KB of dead LDS per workgroup: 4 x 40 KB = the whole CU).
  const unsigned totalWork = (unsigned)blocksPerFrame * (unsigned)nFrames;
  const unsigned chunkW = (totalWork + 7u) >> 3;
  const unsigned perXcd = gridDim.x >> 3;
  for (unsigned itemInChunk = blockIdx.x >> 3; itemInChunk < chunkW; itemInChunk += perXcd) {
  const unsigned work = (blockIdx.x & 7u) * chunkW + itemInChunk;
  if (work >= totalWork) break;  // block-uniform
  const int f = (int)udiv_magic(work, (uint32_t)blocksPerFrame, blocksMagic);  // work / blocksPerFrame
  const int slot0 = (int)(work - (unsigned)f * (unsigned)blocksPerFrame) * kKpPerBlock;
  const int32_t* cnt = levelCount + (size_t)f * a.nlevels;

  // ---- slot -> (level, index, output row).  A wave owns the kKpPerBlock / 4 consecutive slots it also takes through
  //      phases 1-3, and its first lanes resolve them: nothing in the kernel crosses a wave, so there is no workgroup
  //      barrier -- the four waves of a workgroup drift apart and one's transcendental chain (phase 2) overlaps the
  //      others' memory phases (round 2 resolved all slots and all angles on wave 0 between __syncthreads: three waves
  //      waited through both serial sections) ----
  constexpr int kKpPerWave = kKpPerBlock / 4;
  const int myKp = wave * kKpPerWave + lane;  // the keypoint this lane resolves (phase 0) and rotates (phase 2) when lane < kKpPerWave
  if (lane < kKpPerWave) {
    // slot order = spatial order (k_octree: 128-byte column strip, then row).  INTERLEAVED over the waves (round 4): wave w
    // takes slots w, w + 4, w + 8, ... so the four waves of the workgroup work on four NEIGHBOURING keypoints at the same
    // time -- their patches overlap in the same cache lines, and the CU's 32 KB L1 (one patch pair pulls ~11 KB of lines
    // through it) serves the second to fourth request for a line.  With 16 consecutive slots per wave a wave came back to a
    // neighbourhood only after the other three had pushed it out of L1.  ($ORBFE_ORIENT_INTERLEAVE=0: the old order)
    const int slot = slot0 + (interleave ? wave + 4 * lane : myKp);
    int out = -1, l = 0;
    if (slot < a.kpSlotsPerFrame) {
      int base = 0;
      for (int k = 0; k < a.nlevels; k++) {
        if (slot >= a.kpStart[k]) { l = k; }
      }
      for (int k = 0; k < l; k++) base += cnt[k];
      const int i = slot - a.kpStart[l];
      if (i < cnt[l]) {
        const LevelKp kp = levelKp[(size_t)f * a.kpSlotsPerFrame + slot];
        if (base + (int)kp.rank < a.outCapacity) out = base + (int)kp.rank;
        s_x[myKp] = kp.x;
        s_y[myKp] = kp.y;
        s_score[myKp] = kp.score;
      }
    }
    s_level[myKp] = l;
    s_out[myKp] = out;
    if (slot == 0) {
      int tot = 0;
      for (int k = 0; k < a.nlevels; k++) tot += cnt[k];
      nOut[f] = tot;  // the host reports ORBFE_ERR_CAPACITY when this exceeds the capacity
    }
  }
  wave_sync_lds();

  // ---- 1. intensity-centroid moments over the 749-pixel disc ----
  {
    const int row = lane >> 1, half = lane & 1;  // rows 0..30 <-> dy = -15..15 (lanes 62,63 idle)
    const int dy = row - 15;
    const int ady = dy < 0 ? -dy : dy;
    const bool active = row < 31;
    uint4 wt = make_uint4(0, 0, 0, 0), mk = make_uint4(0, 0, 0, 0);
    if (active) {
      const int d = umax[ady];
      wt = momentTab[(d * 2 + half) * 2];
      mk = momentTab[(d * 2 + half) * 2 + 1];
    }
    // 16 keypoints per wave, 4 at a time: the 4 x 5 row loads are issued back to back so one
    // memory latency covers four keypoints (the kernel is latency-bound, not VALU-bound)
    for (int j0 = wave * kKpPerWave; j0 < wave * kKpPerWave + kKpPerWave; j0 += kUM) {
      uint32_t dw[kUM][4];
      int kout[kUM];  // per-keypoint values are wave-uniform: kept in scalar registers (readfirstlane)
#pragma unroll
      for (int u = 0; u < kUM; u++) {
        const int j = j0 + u;
#pragma unroll
        for (int k = 0; k < 4; k++) dw[u][k] = 0;
        kout[u] = __builtin_amdgcn_readfirstlane(s_out[j]);
        if (kout[u] >= 0) {
          const int kl = __builtin_amdgcn_readfirstlane(s_level[j]);
          const int kx = __builtin_amdgcn_readfirstlane(s_x[j]), ky = __builtin_amdgcn_readfirstlane(s_y[j]);
          const LevelView lv = a.pyr.lv[kl];
          // scalar patch origin (row y-15, column x-15); the lane adds its 32-bit (row, half) offset
          const uint8_t* origin = lv.base + (size_t)f * lv.frameStride + (size_t)(ky - 15) * lv.pitch + (kx - 15);
          if (active && !(ablate & 1)) {
            const uint8_t* p = origin + (uint32_t)(row * lv.pitch + (half ? 16 : 0));
            {
              // one byte-aligned 16-byte request per (row, half) whatever the level's pitch and the keypoint's column
              // (global_load_dwordx4 takes any address on gfx950, profiles/r02_unaligned.txt): no fifth dword, no
              // v_alignbyte -- stage -5 % in a same-box A/B; columns x-15 .. x+16 are inside the row for every keypoint
              // (19 <= x <= w-20)
              struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
              const U4u q = *reinterpret_cast<const U4u*>(p);
              dw[u][0] = q.x; dw[u][1] = q.y; dw[u][2] = q.z; dw[u][3] = q.w;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUM; u++) {
        const int j = j0 + u;
        if (kout[u] < 0) continue;  // scalar branch
        const uint32_t q0 = dw[u][0], q1 = dw[u][1], q2 = dw[u][2], q3 = dw[u][3];
        unsigned w = __builtin_amdgcn_udot4(q0, wt.x, 0u, false);
        w = __builtin_amdgcn_udot4(q1, wt.y, w, false);
        w = __builtin_amdgcn_udot4(q2, wt.z, w, false);
        w = __builtin_amdgcn_udot4(q3, wt.w, w, false);
        unsigned sm = __builtin_amdgcn_udot4(q0, mk.x, 0u, false);
        sm = __builtin_amdgcn_udot4(q1, mk.y, sm, false);
        sm = __builtin_amdgcn_udot4(q2, mk.z, sm, false);
        sm = __builtin_amdgcn_udot4(q3, mk.w, sm, false);
        const int sW = (int)w - 16 * (int)sm;  // sum(dx * I); inactive lanes have zero weights
        const int sI = dy * (int)sm;           // sum(dy * I)
        const int m10 = wave_sum(sW), m01 = wave_sum(sI);
        if (lane == 0) { s_m10[j] = m10; s_m01[j] = m01; }
      }
    }
  }
  wave_sync_lds();

  // ---- 2. angle and rotation, one lane per keypoint of the wave ----
  if (lane < kKpPerWave && s_out[myKp] >= 0) {
    const float angle = fast_atan2((float)s_m01[myKp], (float)s_m10[myKp]);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float ca, sb;
    sincos_spec(__fmul_rn(angle, factorPI), &ca, &sb);
    s_angle[myKp] = angle;
    s_cos[myKp] = ca;
    s_sin[myKp] = sb;
  }
  wave_sync_lds();

  // ---- 3. steered BRIEF + output record ----
  {
    float4 P[4];
#pragma unroll
    for (int j = 0; j < 4; j++) P[j] = patternF[lane + 64 * j];  // (x0,x1,y0,y1) of test lane+64j
    // two keypoints per iteration.  Their blurred patches are staged in LDS with 16-byte requests
    // (3 lanes per row: 111 lanes = 2 instructions per keypoint, each patch line touched once) and
    // the 512 samples are LDS byte reads; sampling global memory directly cost one L1 line lookup
    // per lane per sample (8 x 64 per keypoint) and was the slowest part of the kernel.
    uint32_t* myPatch = &s_patch[wave * kUD * kPatchRows * kPatchDw];
    constexpr int kKpPerWave3 = kKpPerBlock / 4;
    const int jBeg = wave * kKpPerWave3, jEnd = jBeg + kKpPerWave3;
    // (round 4, measured and NOT kept -- -DORBFE_DESC_PIPELINE builds it: requesting the patch rows of the NEXT kUD keypoints as
    // soon as the current ones are in LDS, so that their round trip runs under the 512 samples, the ballots and the stores.
    // Same registers, same LDS; exclusive stage 1.37 -> 1.42 ms per 1024 KITTI frames, pipeline unchanged (100.2 k vs 100.2 k)
    // in a same-box A/B: the wave is not waiting for THESE loads -- with the loads removed altogether the stage still takes
    // 0.88 ms, tools/ablate_orient.sh)
    U4 stage[kUD][2];
    int colOffN[kUD], koutN[kUD];  // per-keypoint values are wave-uniform: scalar registers, scalar branches
    auto request = [&](int j0) {
#pragma unroll
      for (int u = 0; u < kUD; u++) {
        const int j = j0 + u;
        colOffN[u] = 0;
        koutN[u] = __builtin_amdgcn_readfirstlane(s_out[j]);
        if (koutN[u] < 0) continue;
        const int kl = __builtin_amdgcn_readfirstlane(s_level[j]);
        const int kx = __builtin_amdgcn_readfirstlane(s_x[j]), ky = __builtin_amdgcn_readfirstlane(s_y[j]);
        const LevelView bl = a.blur.lv[kl];
        int ws = (kx - 18) & ~3;
        if (ws > bl.pitch - 4 * kPatchDw) ws = bl.pitch - 4 * kPatchDw;  // stay inside the row pitch
        colOffN[u] = kx - ws;
        const uint8_t* pb = bl.base + (size_t)f * bl.frameStride + (size_t)(ky - 18) * bl.pitch + ws;  // scalar
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int idx = lane + 64 * h;
          const int row = (idx * 43) >> 7, part = idx - 3 * row;  // idx / 3 for idx < 128
          if (idx < 3 * kPatchRows && !(ablate & 2)) stage[u][h] = *reinterpret_cast<const U4*>(pb + (uint32_t)(row * bl.pitch + 16 * part));
          else stage[u][h] = U4{0u, 0u, 0u, 0u};
        }
      }
    };
#ifdef ORBFE_DESC_PIPELINE
    request(jBeg);
#endif
    for (int j0 = jBeg; j0 < jEnd; j0 += kUD) {
#ifndef ORBFE_DESC_PIPELINE
      request(j0);
#endif
      int t0v[kUD][4], t1v[kUD][4];
      int colOff[kUD], kout[kUD];
#pragma unroll
      for (int u = 0; u < kUD; u++) {
        colOff[u] = colOffN[u];
        kout[u] = koutN[u];
        if (kout[u] < 0) continue;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int idx = lane + 64 * h;
          if (idx < 3 * kPatchRows)
            *reinterpret_cast<uint4*>(&myPatch[u * kPatchRows * kPatchDw + 4 * idx]) =
                make_uint4(stage[u][h].x, stage[u][h].y, stage[u][h].z, stage[u][h].w);
        }
      }
#ifdef ORBFE_DESC_PIPELINE
      if (j0 + kUD < jEnd) request(j0 + kUD);  // (wave-uniform) in flight during everything below
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int u = 0; u < kUD; u++) {
        const int j = j0 + u;
        if (kout[u] < 0) continue;
        const float ca = s_cos[j], sb = s_sin[j];
        const uint8_t* pbytes = reinterpret_cast<const uint8_t*>(s_patch);
        // cvRound by the magic-number add: for |v| < 2^22, float(v + 1.5*2^23) has the bit pattern
        // 0x4B400000 + RNE(v) (round-half-even, like cvRound; 1.5*2^23 is even).  Row and column are left
        // biased and the bias of row*48 + col (0x4B400000 * 49 mod 2^32) is folded into the base offset.
        constexpr float kMagic = 12582912.0f;
        constexpr uint32_t kBias = 0x4B400000u * 49u;
        const uint32_t baseK = (uint32_t)((wave * kUD + u) * kPatchRows * kPatchDw * 4 + 18 * 4 * kPatchDw + colOff[u]) - kBias;
        // row = x*b + y*a, col = x*a - y*b (src/ORBextractor.cc:123-125 with a = cos, b = sin; no FMA), P[t] = (x0, x1, y0, y1).
        // Plain fp32 v_mul / v_add: they issue at 2.4 cycles per wave, the packed v_pk_mul_f32 / v_pk_add_f32 of round 2 at
        // 7.8 for two values (profiles/r02_valu_rate.txt, r03_valu_rate2.txt) -- fewer instructions, more VALU time
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float x0 = P[t].x, x1 = P[t].y, y0 = P[t].z, y1 = P[t].w;
          const float r0 = __fadd_rn(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)), kMagic);
          const float r1 = __fadd_rn(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)), kMagic);
          const float c0 = __fadd_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)), kMagic);
          const float c1 = __fadd_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)), kMagic);
          const uint32_t ir0 = __float_as_uint(r0), ir1 = __float_as_uint(r1);
          const uint32_t ic0 = __float_as_uint(c0), ic1 = __float_as_uint(c1);
          t0v[u][t] = (ablate & 4) ? (int)(ir0 & 255u) : pbytes[baseK + ir0 * (4u * kPatchDw) + ic0];
          t1v[u][t] = (ablate & 4) ? (int)(ic1 & 255u) : pbytes[baseK + ir1 * (4u * kPatchDw) + ic1];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < kUD; u++) {
        const int j = j0 + u;
        const int outIdx = kout[u];
        if (outIdx < 0) continue;
        unsigned long long* dout = reinterpret_cast<unsigned long long*>(
            descOut + ((size_t)f * a.outCapacity + outIdx) * 32);
        // the four ballots are the 32 descriptor bytes: lane t stores ballot t (ONE 8-byte store instruction of 4 lanes
        // instead of four single-lane ones), lanes 0..6 store the seven words of the cv::KeyPoint record (one
        // instruction instead of seven) -- the kernel's memory instructions per keypoint drop from 14 to 5
        // (branch-free: `if (lane == t) bits = bt` / `if (lane == k) w = ...` compiled into twelve exec-mask branches per
        // keypoint around LDS reads the compiler will not speculate; the ballots go to their lanes through selects on loop-invariant lane masks, the record
        // words are read / computed by every lane -- broadcast LDS reads -- and selected)
        uint32_t bitsLo = 0, bitsHi = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const unsigned long long bt = __ballot(t0v[u][t] < t1v[u][t]);
          bitsLo = lane == t ? (uint32_t)bt : bitsLo;
          bitsHi = lane == t ? (uint32_t)(bt >> 32) : bitsHi;
        }
        if (lane < 4) dout[lane] = (unsigned long long)bitsLo | ((unsigned long long)bitsHi << 32);
        {
          const int l = __builtin_amdgcn_readfirstlane(s_level[j]);
          const float sc = a.scale[l];
          const uint32_t wx = __float_as_uint(__fmul_rn((float)s_x[j], sc)), wy = __float_as_uint(__fmul_rn((float)s_y[j], sc));
          const uint32_t wa = __float_as_uint(s_angle[j]), ws = __float_as_uint((float)s_score[j]);
          const uint32_t wz = __float_as_uint(a.kpSize[l]);
          uint32_t w = 0xffffffffu;  // class_id = -1 (lane 6)
          w = lane == 5 ? (uint32_t)l : w;
          w = lane == 4 ? ws : w;
          w = lane == 3 ? wa : w;
          w = lane == 2 ? wz : w;
          w = lane == 1 ? wy : w;
          w = lane == 0 ? wx : w;
          if (lane < 7) reinterpret_cast<uint32_t*>(kpOut + ((size_t)f * a.outCapacity + outIdx) * 7)[lane] = w;
        }
      }
    }
  }
  wave_sync_lds();  // the next item reuses the wave's part of the slot arrays
  }  // work items of this workgroup
}

// keypoints whose moment-row / patch loads a wave keeps in flight at once (template kUM / kUD): 4 / 2.  A/B builds override
// (tools/ab_build.sh um2 -DORBFE_ORIENT_UM=2 -DORBFE_ORIENT_UD=1): measured at the end of round 4, (4, 2) / (2, 1) / (1, 1) give
// 107.0 k / 107.2 k / 106.9 k KITTI stereo frames/s and 1.28 / 1.26 / 1.27 ms exclusive -- the kernel does not wait for its
// loads; see DESIGN.md 4 for the forms with MORE loads in flight (slower in the pipeline)
#ifndef ORBFE_ORIENT_UM
#define ORBFE_ORIENT_UM 4
#endif
#ifndef ORBFE_ORIENT_UD
#define ORBFE_ORIENT_UD 2
#endif

void launch_orient_desc(hipStream_t s, const OrientDescArgs& a, const LevelKp* d_levelKp,
                        const int32_t* d_levelCount, const float4* d_patternF, const uint4* d_momentTab,
                        const int32_t* d_umax, int nFrames, void* d_kpOut, uint8_t* d_descOut,
                        int32_t* d_nOut, int concurrentLaunches) {
  if (nFrames <= 0 || a.kpSlotsPerFrame <= 0) return;
  const bool latencyForm = nFrames <= 8;
  // keypoints per workgroup: 16 for a few frames; 64 for batches ($ORBFE_ORIENT_KPB = 128 | 256: measured, not faster).  Phase 2 --
  // fastAtan2 and the double-precision sincos, ~600 instructions -- runs one LANE per keypoint of the wave: with 16 keypoints
  // per wave three quarters of its lanes idle, 64 per wave (256 per workgroup) fill them and save 15 % of the kernel's VALU
  // instructions -- but the wave's serial chain is four times as long: KITTI 100.2 k (64) / 98.4 k (128) / 95.9 k (256), TUM
  // 362.7 k / 363.9 k / 355.2 k in a same-box A/B (round 4).  Instruction count is not what this kernel waits for.
  static const int kKpbEnv = getenv("ORBFE_ORIENT_KPB") ? atoi(getenv("ORBFE_ORIENT_KPB")) : 64;
  const int kpb = latencyForm ? 16 : (kKpbEnv == 256 ? 256 : (kKpbEnv == 128 ? 128 : 64));
  const int blocksPerFrame = (a.kpSlotsPerFrame + kpb - 1) / kpb;
  const unsigned total = (unsigned)blocksPerFrame * (unsigned)nFrames;
  const uint32_t blocksMagic = udiv_magic_multiplier((uint32_t)blocksPerFrame);
  // Occupancy cap for the throughput form.  All workgroups of a frame run on one XCD; with fewer frames in flight per
  // XCD their patches stay in its 4 MB L2 (HBM fetch 1.17 -> 0.63 GB per 256 frames between 9 and 4 workgroups per CU).
  // Round 1 set the cap with 23 KB of unused dynamic LDS per workgroup ($ORBFE_PAD_ORIENT, still honoured); round 2 sets
  // it with the grid: a few workgroups per CU that loop over the work items, so the cap costs no LDS and the other
  // streams' kernels can share the CUs.  What counts is the number in flight over all the sub-batch streams of the call:
  // 16 per CU divided by the streams, between 2 and 4 (measured per 4096 VGA frames on 8 streams: 2 per CU 323 k
  // frames/s, 3: 314 k, 4: 312 k, LDS cap: 298 k; the stage ALONE is fastest at 4); $ORBFE_ORIENT_GRID overrides, 0 = no cap.
  static const size_t kPad = occupancy_pad_bytes("ORIENT", 0);
  static const int kGridEnv = getenv("ORBFE_ORIENT_GRID") ? atoi(getenv("ORBFE_ORIENT_GRID")) : -1;
  int kGridPerCu = kGridEnv;
  if (kGridPerCu < 0) {
    kGridPerCu = 16 / (concurrentLaunches < 1 ? 1 : concurrentLaunches);
    kGridPerCu = kGridPerCu < 2 ? 2 : (kGridPerCu > 4 ? 4 : kGridPerCu);
    // frames with many keypoints on a large pyramid (1241 x 376 / 2000 features: 32 workgroups and 2.9 MB of pyramid +
    // blurred levels per frame) want ONE frame's workgroups per XCD in flight: 1 per CU measured 100.1 k vs 94.7 k
    // stereo frames/s on 8 streams, while 640 x 480 / 1000 features lose 5 % that way (same-box A/B)
    if (a.kpSlotsPerFrame >= 24 * 64 && concurrentLaunches >= 8) kGridPerCu = 1;
    // a launch that has the device to itself (one stream): no cap -- round 3, with the barrier-free kernel: 1.37 vs 1.59 ms
    // per 1024 KITTI frames, 2.89 vs 3.45 per 4096 TUM frames, 1.76 vs 1.97 per 2048 EuRoC frames against 4 per CU
    if (concurrentLaunches <= 1) kGridPerCu = 0;
  }
  // Round 2 measured the wider form <64, 8, 4> (twice the row / patch loads in flight per wave, 106 VGPRs, same 4
  // workgroups per CU): stage 3.36 vs 3.38 ms per 4096 VGA frames, pipeline unchanged; round 4 again, with <64, 16, 8> too, in
  // the one-workgroup-per-CU configuration of the KITTI pipeline: 100.9 k (4, 2) / 99.5 k (8, 4) / 91.6 k (16, 8) stereo
  // frames per second -- the registers and LDS the wide forms hold cost the co-running kernels more than the loads in
  // flight gain.  <.., 4, 2> stays.
  const unsigned full = (total + 7u) / 8u * 8u;
  static const int kAblate = getenv("ORBFE_ORIENT_ABLATE") ? atoi(getenv("ORBFE_ORIENT_ABLATE")) : 0;
  static const int kInterleave = getenv("ORBFE_ORIENT_INTERLEAVE") ? atoi(getenv("ORBFE_ORIENT_INTERLEAVE")) : 1;
  if (latencyForm) {
    if (kAblate)
      hipLaunchKernelGGL((k_orient_desc<16, ORBFE_ORIENT_UM, ORBFE_ORIENT_UD, true>), dim3(full), dim3(256), 0, s, a, d_levelKp, d_levelCount,
                         d_patternF, d_momentTab, d_umax, (float*)d_kpOut, d_descOut, d_nOut, blocksPerFrame, nFrames, blocksMagic, kAblate, kInterleave);
    else
      hipLaunchKernelGGL((k_orient_desc<16, ORBFE_ORIENT_UM, ORBFE_ORIENT_UD, false>), dim3(full), dim3(256), 0, s, a, d_levelKp, d_levelCount,
                         d_patternF, d_momentTab, d_umax, (float*)d_kpOut, d_descOut, d_nOut, blocksPerFrame, nFrames, blocksMagic, 0, kInterleave);
  } else {
    unsigned grid = full;
    if (kGridPerCu > 0 && (unsigned)kGridPerCu * 256u < full) grid = (unsigned)kGridPerCu * 256u;  // 256 CUs, multiple of 8
#define ORBFE_LAUNCH_ORIENT(KPB, ABL)                                                                                      \
  hipLaunchKernelGGL((k_orient_desc<KPB, ORBFE_ORIENT_UM, ORBFE_ORIENT_UD, ABL>), dim3(grid), dim3(256), kPad, s, a, d_levelKp, d_levelCount, d_patternF, \
                     d_momentTab, d_umax, (float*)d_kpOut, d_descOut, d_nOut, blocksPerFrame, nFrames, blocksMagic, kAblate, kInterleave)
    if (kAblate) ORBFE_LAUNCH_ORIENT(64, true);  // (the ablation build exists for the default workgroup size only)
    else if (kpb == 256) ORBFE_LAUNCH_ORIENT(256, false);
    else if (kpb == 128) ORBFE_LAUNCH_ORIENT(128, false);
    else ORBFE_LAUNCH_ORIENT(64, false);
#undef ORBFE_LAUNCH_ORIENT
  }
}

// Weight/mask bytes of a 16-pixel half row of the orientation disc, for every half-width d:
// entry ((d*2 + half)*2 + kind): kind 0 = weights (dx + 16 inside the disc, else 0), kind 1 = mask.
// half 0 covers dx = -15..0, half 1 covers dx = +1..+16 (dx = 16 is padding).
void build_moment_table(uint8_t* tab /* 16*2*2*16 bytes */) {
  for (int d = 0; d < 16; d++)
    for (int half = 0; half < 2; half++)
      for (int i = 0; i < 16; i++) {
        const int dx = half ? i + 1 : i - 15;
        const int adx = dx < 0 ? -dx : dx;
        const bool in = adx <= d && adx <= 15;
        tab[((d * 2 + half) * 2 + 0) * 16 + i] = (uint8_t)(in ? dx + 16 : 0);
        tab[((d * 2 + half) * 2 + 1) * 16 + i] = (uint8_t)(in ? 1 : 0);
      }
}

}  // namespace orbfe
