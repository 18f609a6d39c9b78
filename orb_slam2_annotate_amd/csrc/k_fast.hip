// k_fast.hip -- the FAST grid stage of ComputeKeyPointsOctTree (src/ORBextractor.cc:846-896)
// as ONE fused kernel: one wavefront (64-thread workgroup) per (frame, grid cell) = per cv::FAST call
// of the reference.  The cell's pixels plus the 3-px ring halo are staged in LDS once
// (16-byte requests from HBM/L2), then
//   A. a cheap necessary test (two adjacent cardinal ring points) in packed 16-bit arithmetic builds an
//      ordered work list of ~19 % of the pixels,
//   B. the listed pixels get the exact cv::FAST response (cornerScore<16>, S-1); a pixel is a corner at
//      threshold t iff S > t; the list is compacted to the corners,
//   C. 3x3 strict non-max suppression restricted to the cell's detection rectangle
//      (the reference's NMS never sees across a cell boundary),
//   D. per-cell threshold fallback (:874-882): corners >= iniThFAST if any survive NMS,
//      else corners >= minThFAST; survivors are emitted in raster order into the cell's
//      slot range (no atomics on HBM: output position is a pure function of the input).
// Facts used: score = S-1 does not depend on the threshold, and a pixel kept by NMS at
// threshold t is exactly a pixel with score >= t that beats all 8 neighbours' scores.
// VALU-issue-bound; it wants resident waves, hence the small LDS footprint.
//
// Fused form (kBlur): the staged tile is exactly the +-3 neighbourhood cv::GaussianBlur(7x7) needs for the
// cell's detection rectangle, so after D the same wavefront writes the BLURRED rectangle (phase E, the
// arithmetic of k_blur.hip: packed-u16 vertical pass, v_dot2 horizontal pass) -- the level is read from HBM
// once for FAST and blur together, and the blur's load latency hides behind the FAST work of the other
// resident waves.  The detection rectangles tile [19, w-19) x [19, h-19); the 19-px frame around them is
// covered by blur-only cells (no FAST phases, reflect-101 staging) appended to the cell table by
// FrameGeom::build, so one launch writes the whole blurred level.
#include <cstdlib>

#include "kernels.h"

namespace orbfe {

namespace {
// cell side bound: wCell = ceil(width/nCols) < 60, so py fits 6 bits and px 8 bits of a list entry
// LDS row pitch in dwords is a template parameter of the kernel: 12 (cells up to 40 px wide: every VGA-class
// level), 16 (up to 56) or 24 -- a 12-dword pitch needs 5.4 KB per cell instead of 8.9 KB, i.e. up to 29
// instead of 17 resident wavefronts per CU (the kernel loses 30 % with 12 instead of 17).

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// 65536 / d + 1 for d = 1..32 (exact i / d for i < 4096 by multiply-shift) -- a table instead of a
// run-time division per workgroup
__constant__ uint32_t kInv16[33] = {0,     65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554,
                                    5958,  5462,  5042,  4682,  4370,  4097,  3856,  3641, 3450, 3277, 3121,
                                    2979,  2850,  2731,  2622,  2521,  2428,  2341,  2260, 2185, 2115, 2049};

// wave64 inclusive scan in the DPP network (checked on gfx950 by tools/ubench/dpp_scan.hip)
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return x;
}
struct __attribute__((aligned(4))) U4 { uint32_t x, y, z, w; };  // 16-byte load at 4-byte alignment
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// Bresenham circle of radius 3 in cv::FAST's order.
[[maybe_unused]] constexpr int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
[[maybe_unused]] constexpr int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// v_perm_b32 selector that builds the packed u16 pair (byte[o], byte[o+2]) of the 8 bytes {hi,lo}
constexpr uint32_t sel2(int o) { return (uint32_t)o | 0x0c00u | ((uint32_t)(o + 2) << 16) | 0x0c000000u; }

// ---- blur arithmetic shared with k_blur.hip ----
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_uu(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t k, uint32_t c) {
  return __builtin_amdgcn_udot2(as_u2(a), as_u2(k), c, false);
}
constexpr uint32_t pk(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }
__device__ __forceinline__ int reflect101f(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);  // clamp only matters for never-used tile cells
}
struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };  // 4-byte store at any byte address
struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };  // 16-byte load at any byte address
}  // namespace

// packed pair (lo, hi) of two bytes; 3-input packed max / min of u16 values < 0x7c00 through the f16 instructions
__device__ __forceinline__ uint32_t pk2(uint8_t lo, uint8_t hi) {
  typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
  u16x2_ v = {(unsigned short)lo, (unsigned short)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// 32-bit 3-input forms for the per-pixel exact score (one pixel per lane)
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// One WAVEFRONT (64-thread workgroup) per (frame, grid cell).  A cell is ~31x32 pixels: 256 four-pixel
// groups, ~190 work-list entries -- a 64-lane unit packs those 99 % full where a 256-thread group
// ran its last pass 3/4 empty, and nothing crosses a wave any more: no barriers, no wave totals.
// The tile lives in LDS with the cell's first pixel at byte column 4, so a group of 4
// horizontally adjacent pixels reads aligned dwords.
//   A. cheap necessary test, packed 16-bit, 4 pixels per lane: a 9-of-16 arc always covers two
//      ADJACENT cardinal ring points (0,4,8,12), so a corner needs (c0|c8)&(c4|c12) in one
//      polarity.  ~19 % of pixels pass; they are appended to an LDS work list IN RASTER ORDER
//      (ballot prefix).
//   B. exact cv::FAST response S-1 for the listed pixels, TWO pixels per lane (dense lanes) in the halves of
//      a register, sliding min/max windows on the raw ring values with gfx950's 3-input packed
//      v_pk_maximum3_f16 / v_pk_minimum3_f16 (bit patterns 0..255 are ordered as f16 exactly as they are as integers);
//      the list shrinks in place to the corners.
//   C. cell-local 3x3 strict NMS of the corners (one lane per corner) + threshold classes.
//   D. per-cell 20->7 fallback, then ordered compaction of the survivors = emission order.
// Dynamic LDS: tile [tileRows][24 dw] | score [scoreRows][24 dw] | queue [queueLen] u16, sized by
// the largest cell of the frame geometry.
// kLowFirst: ONE attempt at min(iniThFAST, minThFAST) that classifies the survivors for both thresholds (round 1's
// form) instead of iniThFAST first + per-cell fallback.  Same results; which is faster depends on the image: on
// textured frames few cells need the fallback and the first form lists far fewer pixels for the exact score (-15 %);
// on sparse, low-contrast frames most cells fall back and pay the pre-test twice (+12 %).  The extractor picks per
// call from the fallback rate the previous call measured (fallbackStat; extractor.hip).
template <int kPitchDw, bool kBlur, bool kLowFirst>
__global__ __launch_bounds__(64) void k_fast_cells(PyramidViews pyr,
                                                   const CellDesc* __restrict__ cells,
                                                   int nCells, int nFrames, int iniTh, int minTh,
                                                   Candidate* __restrict__ slots,
                                                   int slotsPerFrame,
                                                   uint16_t* __restrict__ cellCount,
                                                   int tileRows, int scoreRows, uint32_t cellsMagic,
                                                   PyramidViews blurOut, int nFastCells,
                                                   unsigned int* __restrict__ fallbackStat,
                                                   int cutoff /* 0; $ORBFE_FAST_CUTOFF: stop after staging (1) / A (2) / B (3) / C (4) -- per-phase instruction counts */) {
  extern __shared__ uint32_t lds[];
  uint32_t* tile = lds;                                   // pixels: origin (x0-4, y0-3)
  uint32_t* score = lds + tileRows * kPitchDw;            // FAST responses: origin (x0-4, y0-1)
  uint16_t* queue = reinterpret_cast<uint16_t*>(score + scoreRows * kPitchDw);  // py<<8 | px, bits 14/15 = NMS classes

  const int lane = threadIdx.x;
  // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so
  // block b takes work item (b % 8) * chunk + b / 8 -- every XCD walks a contiguous run of
  // (frame, cell) items and neighbouring cells, which share halo rows, meet in the same L2.
  const unsigned chunk = gridDim.x >> 3;
  const unsigned work = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  if (work >= (unsigned)nCells * (unsigned)nFrames) return;
  const int f = (int)udiv_magic(work, (uint32_t)nCells, cellsMagic);  // work / nCells
  const int cellId = (int)(work - (unsigned)f * (unsigned)nCells);
  const CellDesc cd = cells[cellId];
  const LevelView lv = pyr.lv[cd.level];
  const int cw = cd.w, ch = cd.h, x0 = cd.x0, y0 = cd.y0;
  const int ngx = (cw + 3) >> 2;          // 4-pixel groups per row
  const int tdw = ngx + 2;                // tile dwords per row
  // exact i / d for i < 4096 by multiply-shift (divisors <= 17)
  const uint32_t invT = kInv16[tdw];
  const unsigned long long ltMask = (1ull << lane) - 1ull;
  constexpr int P = kPitchDw * 4;         // LDS row pitch in bytes

  // ---- stage the tile: rows y0-3 .. y0+ch+2, bytes x0-4 .. x0+4*ngx+3 (inside the level) ----
  const bool blurOnly = kBlur && (cd.flags & kCellBlurOnly) != 0;
  {
    const uint8_t* lvb = lv.base + (size_t)f * lv.frameStride;
    const int th = ch + 6;
    if (kBlur && (cd.flags & kCellColReflect)) {
      // frame cell whose columns leave the level: bytes with BORDER_REFLECT_101 in both directions
      for (int i = lane; i < th * tdw; i += 64) {
        const int ty = (int)(((uint32_t)i * invT) >> 16), tx = i - ty * tdw;
        const uint8_t* row = lvb + (size_t)reflect101f(y0 - 3 + ty, lv.h) * lv.pitch;
        const int c = x0 - 4 + 4 * tx;
        tile[ty * kPitchDw + tx] = (uint32_t)row[reflect101f(c, lv.w)] | ((uint32_t)row[reflect101f(c + 1, lv.w)] << 8) |
                                   ((uint32_t)row[reflect101f(c + 2, lv.w)] << 16) | ((uint32_t)row[reflect101f(c + 3, lv.w)] << 24);
      }
    } else {
      const bool rr = kBlur && (cd.flags & kCellRowReflect) != 0;  // frame cell at the top / bottom: reflected row index
      {
        // a lane moves 16 bytes: (row, part) -> ONE byte-aligned 16-byte request -> one ds_write_b128.  The tile starts
        // at x0-4, which is dword-aligned for no cell in particular; global_load_dwordx4 takes any address on gfx950 at
        // full bandwidth (profiles/r02_unaligned.txt), so there is no byte shifting (round 1 loaded 5 aligned dwords
        // and shifted them with 4 v_alignbyte: the same stage measured 1 % slower) and caller-owned frames with an odd
        // stride (KITTI: 1241) take the same path as the handle's own 64-byte pitched levels
        const int parts = (tdw + 3) >> 2;
        const uint32_t invP = kInv16[parts];
        const int rowBytes = lv.pitch - (x0 - 4);  // bytes from the tile's first column to the end of the row
        // two pieces per lane and trip, both requested before either is stored: a ~31 x 32 cell is 114 pieces, i.e. ONE
        // memory round trip at the head of the wave instead of two dependent ones
        for (int i0 = lane; i0 < th * parts; i0 += 128) {
          const uint8_t* pp[2];
          int dsti[2], partv[2];
          bool ok[2], wide[2];
          uint4 o[2];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int i = i0 + 64 * u;
            ok[u] = i < th * parts;
            const int ty = (int)(((uint32_t)i * invP) >> 16), part = i - ty * parts;
            const int sy = rr ? reflect101f(y0 - 3 + ty, lv.h) : y0 - 3 + ty;
            pp[u] = lvb + (size_t)sy * lv.pitch + (x0 - 4) + 16 * part;
            dsti[u] = ty * kPitchDw + 4 * part;
            partv[u] = part;
            wide[u] = ok[u] && 16 * part + 16 <= rowBytes;
            o[u] = make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
          for (int u = 0; u < 2; u++)
            if (wide[u]) {
              const U4u q = *reinterpret_cast<const U4u*>(pp[u]);
              o[u] = make_uint4(q.x, q.y, q.z, q.w);
            }
#pragma unroll
          for (int u = 0; u < 2; u++)
            if (ok[u] && !wide[u]) {  // the last piece of a right-edge cell: stay inside the row (and the caller's buffer)
              uint32_t d[4];
#pragma unroll
              for (int k = 0; k < 4; k++) {
                d[k] = 0;
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
                  if (16 * partv[u] + 4 * k + bb < rowBytes) d[k] |= (uint32_t)pp[u][4 * k + bb] << (8 * bb);
              }
              o[u] = make_uint4(d[0], d[1], d[2], d[3]);
            }
#pragma unroll
          for (int u = 0; u < 2; u++)
            if (ok[u]) *reinterpret_cast<uint4*>(&tile[dsti[u]]) = o[u];
        }
      }
    }
    if (!blurOnly) {
      uint4* sz = reinterpret_cast<uint4*>(score);  // row pitch 96 B: (ch+2)*6 aligned 16-byte stores
      for (int i = lane; i < (ch + 2) * (kPitchDw / 4); i += 64) sz[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  if (cutoff == 1 && !blurOnly) {  // (profiling cut-off: the cell reports no candidates, so the later kernels see a consistent state)
    if (lane == 0) cellCount[(size_t)f * nFastCells + cellId] = 0;
    return;
  }

  if (!blurOnly) {  // wave-uniform: the FAST phases A-D of a detection cell
  // The reference calls cv::FAST(cell, iniThFAST) and only if that returns nothing cv::FAST(cell, minThFAST)
  // (:874-882).  A cell is one wavefront, so the same order costs nothing here: phases A-C run at iniThFAST
  // first -- its pre-test passes far fewer pixels than the one at minThFAST (7 is inside the image noise) -- and
  // the wave-uniform fallback repeats them at minThFAST for the cells that kept no corner.
  int nc = 0;      // wave-uniform: corners in the list after phase B
  int tcur = kLowFirst ? (iniTh < minTh ? iniTh : minTh) : iniTh;
  int useHigh = 1, fellBack = 0;  // wave-uniform
  for (int attempt = 0; attempt < 2; attempt++) {
  // ---- A: cardinal-pair test at the current threshold; ordered work list.  A lane owns 8 adjacent
  //      pixels (two tile dwords); list positions come from a DPP inclusive scan of the lane counts ----
  int nq = 0;  // wave-uniform
  {
    const s16x2 T = {(short)tcur, (short)tcur};
    // the quantised pre-test is selective only while t >> 2 is well above the quantisation step (t = 7 would list 70 % of
    // the pixels): thresholds below 16 take the exact packed-16 form
    [[maybe_unused]] const bool quantA = tcur >= 16 && tcur <= 255;          // wave-uniform
    [[maybe_unused]] const uint32_t biasT6 = 0x80808080u - (uint32_t)(tcur >> 2) * 0x01010101u;
    const int ng8 = (cw + 7) >> 3;          // 8-pixel groups per row
    const int ngroups8 = ng8 * ch;
    const uint32_t invG8 = kInv16[ng8];
    for (int g0 = 0; g0 < ngroups8; g0 += 64) {
      const int g = g0 + lane;
      uint32_t pass = 0;
      int gy = 0, gx = 0;
      if (g < ngroups8) {
        gy = (int)(((uint32_t)g * invG8) >> 16);
        gx = g - gy * ng8;
        const uint32_t* mid = &tile[(gy + 3) * kPitchDw + 2 * gx];
        const uint32_t* upp = &tile[gy * kPitchDw + 2 * gx + 1];
        const uint32_t* dnp = &tile[(gy + 6) * kPitchDw + 2 * gx + 1];
        const uint32_t m[4] = {mid[0], mid[1], mid[2], mid[3]};
        const uint32_t ups[2] = {upp[0], upp[1]}, dns[2] = {dnp[0], dnp[1]};
#ifndef ORBFE_FAST_A_EXACT
        if (quantA) {
          // QUANTISED form of the same necessary test, four pixels per 32-bit lane-op in the full-rate instruction class
          // (v_add / v_sub / v_or / v_bitop3: 2.6 cycles, against 4.3 for every packed-16 op and v_perm,
          // profiles/r03_valu_rate2.txt).  Pixels become 6-bit values x6 = x >> 2 in 8-bit fields, t6 = t >> 2:
          //   x > c + t  =>  x6 >= c6 + t6   <=>  bit 7 of the field  x6 + (128 - t6 - c6)   (field in [2, 191]: no carry)
          //   x < c - t  =>  x6 <= c6 - t6   <=>  bit 7 of the field  (c6 + 128 - t6) - x6   (field in [2, 191]: no borrow)
          // (floor(a + b) >= floor(a) + floor(b), floor(a - b) <= floor(a) - floor(b)).  Being a weaker necessary condition it
          // lists a few more pixels (+2 % at t = 20 on the bench frames) and can never lose one; phase B decides exactly.
          uint32_t q[4], qu[2], qd[2];
#pragma unroll
          for (int k = 0; k < 4; k++) q[k] = (m[k] >> 2) & 0x3f3f3f3fu;
#pragma unroll
          for (int k = 0; k < 2; k++) { qu[k] = (ups[k] >> 2) & 0x3f3f3f3fu; qd[k] = (dns[k] >> 2) & 0x3f3f3f3fu; }
          uint32_t fl[2];
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const uint32_t c6 = q[hh + 1];
            const uint32_t e6 = __builtin_amdgcn_alignbyte(q[hh + 2], q[hh + 1], 3);  // bytes x+3 .. x+6
            const uint32_t w6 = __builtin_amdgcn_alignbyte(q[hh + 1], q[hh], 1);      // bytes x-3 .. x
            const uint32_t kb = biasT6 - c6, kd = biasT6 + c6;
            const uint32_t bS = qd[hh] + kb, bN = qu[hh] + kb, bE = e6 + kb, bW = w6 + kb;
            const uint32_t dS = kd - qd[hh], dN = kd - qu[hh], dE = kd - e6, dW = kd - w6;
            fl[hh] = ((bS | bN) & (bE | bW)) | ((dS | dN) & (dE | dW));
          }
          // bit 7 of the eight bytes -> pass bits 0..7: dword 0's flags move to bits 3, 11, 19, 27, one multiply lines all
          // eight up in the top byte (partial products on distinct bits: no carries)
          const uint32_t g = ((fl[0] & 0x80808080u) >> 4) | (fl[1] & 0x80808080u);
          pass = (g * 0x00204081u) >> 24;
        } else
#endif
        {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {    // the two dwords of the 8-pixel group
          const uint32_t m0 = m[hh], m1 = m[hh + 1], m2 = m[hh + 2], up = ups[hh], dn = dns[hh];
          uint32_t p[2];
#pragma unroll
          for (int st = 0; st < 2; st++) {  // stream 0: pixels (0,2); stream 1: pixels (1,3)
            const s16x2 c = as_s2(__builtin_amdgcn_perm(m1, m0, sel2(4 + st)));
            const s16x2 rS = as_s2(__builtin_amdgcn_perm(dn, dn, sel2(st)));      // ring 0  (0,+3)
            const s16x2 rE = as_s2(__builtin_amdgcn_perm(m2, m1, sel2(3 + st)));  // ring 4  (+3,0)
            const s16x2 rN = as_s2(__builtin_amdgcn_perm(up, up, sel2(st)));      // ring 8  (0,-3)
            const s16x2 rW = as_s2(__builtin_amdgcn_perm(m1, m0, sel2(1 + st)));  // ring 12 (-3,0)
            // both axes hold a dark point <=> mD = max(min(S,N), min(E,W)) < v-t; bright: mB = min(max, max) > v+t;
            // either <=> max(v - mD, mB - v) > t <=> sign bit of t - max(...)
            const s16x2 mD = pk_max(pk_min(rS, rN), pk_min(rE, rW));
            const s16x2 mB = pk_min(pk_max(rS, rN), pk_max(rE, rW));
            p[st] = as_u(T - pk_max(c - mD, mB - c));
          }
          // the four sign bits (pixels 0..3 = p0.lo, p1.lo, p0.hi, p1.hi) in three steps instead of six shifts / masks
          // per stream: one v_perm lines the four sign-carrying bytes up, the multiply moves bits 7, 15, 23, 31 to
          // 28..31 (the partial products land on distinct bits: no carries)
          const uint32_t sg = __builtin_amdgcn_perm(p[1], p[0], 0x07030501u) & 0x80808080u;
          pass |= ((sg * 0x00204081u) >> 28) << (4 * hh);
        }
        }
        const int valid = cw - 8 * gx;
        if (valid < 8) pass &= (1u << valid) - 1u;
      }
      const int c = __popc(pass);  // 0..8
      const int incl = wave_incl_scan(c);
      int q = nq + incl - c;
      nq += __builtin_amdgcn_readlane(incl, 63);
      const uint32_t e0 = (uint32_t)((gy << 8) | (8 * gx));
      while (pass) {               // raster order inside the lane: ascending bit = ascending x
        const int j = __builtin_ctz(pass);
        queue[q++] = (uint16_t)(e0 + j);
        pass &= pass - 1u;
      }
    }
  }
  __syncthreads();
  if (cutoff == 2) { if (lane == 0) cellCount[(size_t)f * nFastCells + cellId] = 0; return; }

  // ---- B: exact response of the listed pixels (cornerScore<16>: S-1, corner iff S > t); the list is
  //      compacted in place to the corners (order kept: a pass writes no further than it has read) ----
  nc = 0;
#ifndef ORBFE_FAST_B_SCALAR
  {
    // TWO listed pixels per lane (entries 2i, 2i+1 of the pass: neighbours in raster order), packed in the halves of a
    // register: the sliding windows are gfx950's 3-input packed
    // v_pk_maximum3_f16 / v_pk_minimum3_f16 applied to the BIT PATTERNS 0..255 -- positive f16 denormals, whose order as
    // floats is their order as integers; the instructions return one of their inputs unchanged (checked for all pairs
    // < 0x7c00 by tools/ubench/valu_rate2.hip, profiles/r03_valu_rate2.txt).  80 window instructions per TWO pixels
    // instead of per pixel (the 2-input v_pk_max_i16 form costs what the 32-bit v_max3_i32 form does).
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
    uint8_t* sbytes = reinterpret_cast<uint8_t*>(score);
    for (int q0 = 0; q0 < nq; q0 += 128) {
      const int qa = q0 + 2 * lane;
      const bool okA = qa < nq, okB = qa + 1 < nq;
      // an idle half reads the tile's first ring (in-bounds garbage, never a corner: the ok flags gate it)
      const uint32_t pair = okA ? *reinterpret_cast<const uint32_t*>(&queue[qa]) : 0u;  // qa is even, queueLen is even
      const int ea = (int)(pair & 0xffffu), eb = okB ? (int)(pair >> 16) : 0;
      const uint8_t* ca = tb + ((ea >> 8) + 3) * P + 4 + (ea & 255);
      const uint8_t* cb = tb + ((eb >> 8) + 3) * P + 4 + (eb & 255);
      // (ds_read_u8 x 2 + one v_perm per pair: the d16 loads that write one half of a register do NOT keep the other half
      // on this part -- SRAM ECC -- which is why the compiler does not select them either)
      const uint32_t v2 = pk2(ca[0], cb[0]);
      uint32_t r[16];
#pragma unroll
      for (int k = 0; k < 16; k++) r[k] = pk2(ca[kRingDx[k] + kRingDy[k] * P], cb[kRingDx[k] + kRingDy[k] * P]);
      uint32_t mx3[16], mn3[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        mx3[k] = pk_max3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
        mn3[k] = pk_min3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
      }
      uint32_t mx9[16], mn9[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        mx9[k] = pk_max3(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
        mn9[k] = pk_min3(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
      }
      uint32_t darkest = pk_min3(mx9[0], mx9[1], mx9[2]), brightest = pk_max3(mn9[0], mn9[1], mn9[2]);
#pragma unroll
      for (int k = 3; k < 15; k += 2) { darkest = pk_min3(darkest, mx9[k], mx9[k + 1]); brightest = pk_max3(brightest, mn9[k], mn9[k + 1]); }
      darkest = pk_min3(darkest, mx9[15], mx9[15]);
      brightest = pk_max3(brightest, mn9[15], mn9[15]);
      // S = max(v - darkest, brightest - v) per half (differences in [-255, 255]: signed 16-bit)
      const s16x2 S2 = pk_max(as_s2(v2) - as_s2(darkest), as_s2(brightest) - as_s2(v2));
      const int Sa = S2.x, Sb = S2.y;
      const bool cornerA = okA && Sa > tcur, cornerB = okB && Sb > tcur;
      const unsigned long long balA = __ballot(cornerA), balB = __ballot(cornerB);
      const int pos = nc + __popcll(balA & ltMask) + __popcll(balB & ltMask);
      if (cornerA) {
        sbytes[((ea >> 8) + 1) * P + 4 + (ea & 255)] = (uint8_t)(Sa - 1);
        queue[pos] = (uint16_t)ea;
      }
      if (cornerB) {
        sbytes[((eb >> 8) + 1) * P + 4 + (eb & 255)] = (uint8_t)(Sb - 1);
        queue[pos + (cornerA ? 1 : 0)] = (uint16_t)eb;
      }
      nc += __popcll(balA) + __popcll(balB);
    }
  }
#else
  {
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
    uint8_t* sbytes = reinterpret_cast<uint8_t*>(score);
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      bool isCorner = false;
      int e = 0, S = 0;
      if (q < nq) {
        e = queue[q];
        const int px = e & 255, py = e >> 8;
        const uint8_t* c = tb + (py + 3) * P + 4 + px;
        const int v = c[0];
        // cornerScore on the raw ring values: min over an arc of (v - r) is v - max over the arc of r, so the
        // sliding windows run on r itself and the centre enters twice at the end instead of 16 times up front:
        //   S = max( v - min_k max9_k(r),  max_k min9_k(r) - v )
        int r[16];
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = (int)c[kRingDx[k] + kRingDy[k] * P];
        int mx3[16], mn3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          mx3[k] = max3i(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
          mn3[k] = min3i(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
        }
        int mx9[16], mn9[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          mx9[k] = max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
          mn9[k] = min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
        }
        int darkest = min3i(mx9[0], mx9[1], mx9[2]), brightest = max3i(mn9[0], mn9[1], mn9[2]);
#pragma unroll
        for (int k = 3; k < 15; k += 2) { darkest = min3i(darkest, mx9[k], mx9[k + 1]); brightest = max3i(brightest, mn9[k], mn9[k + 1]); }
        darkest = min(darkest, mx9[15]);
        brightest = max(brightest, mn9[15]);
        S = max(v - darkest, brightest - v);
        isCorner = S > tcur;
      }
      const unsigned long long bal = __ballot(isCorner);
      if (isCorner) {
        sbytes[((e >> 8) + 1) * P + 4 + (e & 255)] = (uint8_t)(S - 1);
        queue[nc + __popcll(bal & ltMask)] = (uint16_t)e;
      }
      nc += __popcll(bal);
    }
  }
#endif
  __syncthreads();
  if (cutoff == 3) { if (lane == 0) cellCount[(size_t)f * nFastCells + cellId] = 0; return; }

  // ---- C: cell-local 3x3 strict NMS, one lane per corner.  Two-attempt form: survivors get bit 15 of the list entry
  //      (every listed corner has S > tcur, i.e. response S-1 >= tcur: what cv::FAST(cell, tcur) keeps after NMS).
  //      kLowFirst: classes in bits 14 (>= minThFAST) and 15 (>= iniThFAST), the fallback is decided afterwards ----
  const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(score);
  int anyKept = 0;
  for (int q = lane; q < nc; q += 64) {
    const int e = queue[q];
    const uint8_t* c = sbytes + ((e >> 8) + 1) * P + 4 + (e & 255);
    const int v = c[0];
    const int nb = max3i(max3i(c[-P - 1], c[-P], c[-P + 1]), max3i(c[-1], c[1], c[P - 1]), max(c[P], c[P + 1]));
    int flags;
    if (kLowFirst) flags = v > nb ? (((v >= minTh) << 14) | ((v >= iniTh) << 15)) : 0;
    else flags = (v > nb) << 15;
    queue[q] = (uint16_t)(e | flags);
    anyKept |= flags >> 15;
  }
  const int found = __ballot(anyKept) != 0ull;
  __syncthreads();
  if (kLowFirst) {  // per-cell threshold fallback (:874-882): corners >= iniThFAST if any survived NMS, else >= minThFAST
    useHigh = found;
    fellBack = !found;
    break;
  }
  if (found || attempt == 1) break;
  tcur = minTh;  // :880: nothing survived at iniThFAST
  fellBack = 1;
  }  // attempt
  // cells that needed minThFAST -- a scheduling hint only, so every 8th work item reports, spread over 64 counters
  // (one counter for all cells serialised 2 M same-address atomics per launch: 24 ms instead of 4)
  if (fallbackStat && fellBack && lane == 0 && (work & 7u) == 0) atomicAdd(&fallbackStat[(work >> 3) & 63u], 1u);
  if (cutoff == 4) { if (lane == 0) cellCount[(size_t)f * nFastCells + cellId] = 0; return; }

  // ---- D: ordered compaction of the survivors (the list is in raster order, :884-893) ----
  Candidate* out = slots + (size_t)f * slotsPerFrame + cd.slotBase;
  const int bit = (kLowFirst && !useHigh) ? 0x4000 : 0x8000;
  const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(score);
  int run = 0;  // wave-uniform
  for (int q0 = 0; q0 < nc; q0 += 64) {
    const int q = q0 + lane;
    const int e = q < nc ? queue[q] : 0;
    const bool sel = (e & bit) != 0;
    const unsigned long long bal = __ballot(sel);
    const int o = run + __popcll(bal & ltMask);
    run += __popcll(bal);
    if (sel) {
      const int px = e & 255, py = (e >> 8) & 63;
      Candidate cnd;
      cnd.xy = (uint32_t)(x0 + px - kMinBorder) | ((uint32_t)(y0 + py - kMinBorder) << 16);
      cnd.score = sbytes[(py + 1) * P + 4 + px];
      out[o] = cnd;
    }
  }
  if (lane == 0) cellCount[(size_t)f * nFastCells + cellId] = (uint16_t)run;
  }  // !blurOnly

  if (kBlur) {
    // ---- E: GaussianBlur 7x7 of the cell's own rectangle from the staged tile (arithmetic of k_blur.hip:
    //      [18,34,48,56,48,34,18]/256 twice, one rounding (x + 2^15) >> 16).  The vertical sums reuse the
    //      score + queue area, which is dead after D ----
    __syncthreads();
    uint2* vbuf = reinterpret_cast<uint2*>(score);  // [4*rowBlocks][tdw] entries of 4 u16
    const int rowBlocks = (ch + 3) >> 2;
    for (int i = lane; i < rowBlocks * tdw; i += 64) {
      const int rb = (int)(((uint32_t)i * invT) >> 16), tj = i - rb * tdw;
      const uint32_t* tp = &tile[(4 * rb) * kPitchDw + tj];
      u16x2 te[10], to[10];  // even bytes (0,2) and odd bytes (1,3) of each source dword
#pragma unroll
      for (int j = 0; j < 10; j++) {
        const uint32_t r = tp[j * kPitchDw];   // rows past the tile (last, partial block): in-bounds garbage, never stored
        te[j] = as_u2(r & 0x00ff00ffu);
        to[j] = as_u2(__builtin_amdgcn_perm(r, r, 0x0c030c01u));
      }
      const u16x2 k18 = {18, 18}, k34 = {34, 34}, k48 = {48, 48}, k56 = {56, 56};
      uint2* vo = &vbuf[(4 * rb) * tdw + tj];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        u16x2 a = (te[r] + te[r + 6]) * k18;
        a = (te[r + 1] + te[r + 5]) * k34 + a;
        a = (te[r + 2] + te[r + 4]) * k48 + a;
        const uint32_t A = as_uu(te[r + 3] * k56 + a);   // (v0, v2)
        u16x2 b = (to[r] + to[r + 6]) * k18;
        b = (to[r + 1] + to[r + 5]) * k34 + b;
        b = (to[r + 2] + to[r + 4]) * k48 + b;
        const uint32_t B = as_uu(to[r + 3] * k56 + b);   // (v1, v3)
        uint2 o;
        o.x = __builtin_amdgcn_perm(B, A, 0x05040100u);  // (v0, v1)
        o.y = __builtin_amdgcn_perm(B, A, 0x07060302u);  // (v2, v3)
        vo[r * tdw] = o;
      }
    }
    __syncthreads();
    const LevelView bv = blurOut.lv[cd.level];
    uint8_t* D = const_cast<uint8_t*>(bv.base) + (size_t)f * bv.frameStride + (size_t)y0 * bv.pitch + x0;
    const uint32_t invN = kInv16[ngx];
    for (int i = lane; i < ch * ngx; i += 64) {
      const int r = (int)(((uint32_t)i * invN) >> 16), g = i - r * ngx;
      const uint2* vp = &vbuf[r * tdw + g];
      const uint2 e0 = vp[0], e1 = vp[1], e2 = vp[2];
      // d_k = (v'[2k], v'[2k+1]) with v' indexed from tile column 4*g; output pixel j of the group = tile column 4*g+4+j
      const uint32_t d0 = e0.x, d1 = e0.y, d2 = e1.x, d3 = e1.y, d4 = e2.x, d5 = e2.y;
      const uint32_t R = 1u << 15;
      uint32_t o0 = dot2(d0, pk(0, 18), R);   // taps v'1..v'7
      o0 = dot2(d1, pk(34, 48), o0);
      o0 = dot2(d2, pk(56, 48), o0);
      o0 = dot2(d3, pk(34, 18), o0);
      uint32_t o1 = dot2(d1, pk(18, 34), R);  // taps v'2..v'8
      o1 = dot2(d2, pk(48, 56), o1);
      o1 = dot2(d3, pk(48, 34), o1);
      o1 = dot2(d4, pk(18, 0), o1);
      uint32_t o2 = dot2(d1, pk(0, 18), R);   // taps v'3..v'9
      o2 = dot2(d2, pk(34, 48), o2);
      o2 = dot2(d3, pk(56, 48), o2);
      o2 = dot2(d4, pk(34, 18), o2);
      uint32_t o3 = dot2(d2, pk(18, 34), R);  // taps v'4..v'10
      o3 = dot2(d3, pk(48, 56), o3);
      o3 = dot2(d4, pk(48, 34), o3);
      o3 = dot2(d5, pk(18, 0), o3);
      // the rounded sums are < 2^24: byte 2 of each is the result; two v_perm gather them
      const uint32_t lo = __builtin_amdgcn_perm(o1, o0, 0x0c0c0602u);   // (o0.b2, o1.b2, 0, 0)
      const uint32_t hi = __builtin_amdgcn_perm(o3, o2, 0x06020c0cu);   // (0, 0, o2.b2, o3.b2)
      const uint32_t v = lo | hi;
      uint8_t* d = D + (uint32_t)r * (uint32_t)bv.pitch + 4u * (uint32_t)g;
      const int valid = cw - 4 * g;
      if (valid >= 4) {
        U1u st = {v};
        *reinterpret_cast<U1u*>(d) = st;  // one dword store at whatever byte alignment x0 has
      } else {  // the rectangle's last, partial group: bytes (the next cell owns the rest of the dword)
        d[0] = (uint8_t)v;
        if (valid > 1) d[1] = (uint8_t)(v >> 8);
        if (valid > 2) d[2] = (uint8_t)(v >> 16);
      }
    }
  }
}

void launch_fast_cells(hipStream_t s, PyramidViews pyr, const CellDesc* d_cells, int nCells,
                       int nFrames, int iniTh, int minTh, Candidate* d_slots, int slotsPerFrame,
                       uint16_t* d_cellCount, int maxCellW, int maxCellH, const PyramidViews* blurOut,
                       int nCellsAll, bool lowFirst, unsigned int* d_fallbackStat) {
  // blurOut != NULL: the fused FAST+blur form over all nCellsAll cells (FAST cells first, then blur-only frame
  // cells); NULL: FAST only over the nCells detection cells
  const int nWork = blurOut ? nCellsAll : nCells;
  if (nWork <= 0 || nFrames <= 0) return;
  iniTh = iniTh < 0 ? 0 : (iniTh > 255 ? 255 : iniTh);  // cv::FAST clamps the threshold
  minTh = minTh < 0 ? 0 : (minTh > 255 ? 255 : minTh);
  const unsigned total = (unsigned)nWork * (unsigned)nFrames;
  const int tileRows = maxCellH + 6, scoreRows = maxCellH + 2;
  const int queueLen = (((maxCellW + 3) & ~3) * maxCellH + 1) & ~1;
  // smallest pitch that holds a tile row (4 * ceil(tdw / 4) staged dwords, tdw = ceil(w/4) + 2) and a score row
  const int tdwMax = ((maxCellW + 3) >> 2) + 2;
  const int need = ((tdwMax + 3) >> 2) * 4;
  const int pitch = (need <= 12 && maxCellW <= 40) ? 12 : (need <= 16 && maxCellW <= 56) ? 16 : 24;
  size_t work2 = (size_t)scoreRows * pitch * 4 + (size_t)queueLen * 2;  // score + queue
  if (blurOut) {  // phase E keeps the vertical sums there: 4*ceil(h/4) rows of tdw 8-byte entries
    const size_t vb = (size_t)((maxCellH + 3) & ~3) * tdwMax * 8;
    if (vb > work2) work2 = vb;
  }
  const size_t ldsBytes = (size_t)tileRows * pitch * 4 + work2;
  const dim3 grid((total + 7u) / 8u * 8u);
  const uint32_t magic = udiv_magic_multiplier((uint32_t)nWork);
  PyramidViews bo = blurOut ? *blurOut : PyramidViews{};
  static const int kCutoff = getenv("ORBFE_FAST_CUTOFF") ? atoi(getenv("ORBFE_FAST_CUTOFF")) : 0;
#define ORBFE_LAUNCH_FAST(P, B, L)                                                                                     \
  hipLaunchKernelGGL((k_fast_cells<P, B, L>), grid, dim3(64), ldsBytes, s, pyr, d_cells, nWork, nFrames, iniTh, minTh, \
                     d_slots, slotsPerFrame, d_cellCount, tileRows, scoreRows, magic, bo, nCells, d_fallbackStat, kCutoff)
#define ORBFE_LAUNCH_FAST_P(B, L)            \
  do {                                       \
    if (pitch == 12) ORBFE_LAUNCH_FAST(12, B, L);      \
    else if (pitch == 16) ORBFE_LAUNCH_FAST(16, B, L); \
    else ORBFE_LAUNCH_FAST(24, B, L);                  \
  } while (0)
  if (blurOut) {
    if (lowFirst) ORBFE_LAUNCH_FAST_P(true, true);
    else ORBFE_LAUNCH_FAST_P(true, false);
  } else {
    if (lowFirst) ORBFE_LAUNCH_FAST_P(false, true);
    else ORBFE_LAUNCH_FAST_P(false, false);
  }
#undef ORBFE_LAUNCH_FAST_P
#undef ORBFE_LAUNCH_FAST
}

// Ordered compaction: cells of a level in cell-row-major order, raster inside each cell.
__global__ __launch_bounds__(256) void k_gather_candidates(const CellDesc* __restrict__ cells,
                                                           const LevelGeom* __restrict__ lvg,
                                                           const Candidate* __restrict__ slots,
                                                           int slotsPerFrame,
                                                           const uint16_t* __restrict__ cellCount,
                                                           int cellsPerFrame,
                                                           Candidate* __restrict__ cand,
                                                           int32_t* __restrict__ candCount,
                                                           int32_t* __restrict__ cellPrefix,
                                                           int nlevels) {
  __shared__ int waveTot[4];
  ORBFE_LATENCY_KERNEL_PRIO();
  // grid (nFrames, nlevels): consecutive workgroups -- which go to consecutive XCDs -- are consecutive frames of one
  // level.  With the level in blockIdx.x, level l of every frame ran on XCD l (8 levels, 8 XCDs) and the XCD with
  // the level-0 items decided the duration.
  const int f = blockIdx.x, l = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const LevelGeom g = lvg[l];
  const uint16_t* cnt = cellCount + (size_t)f * cellsPerFrame + g.cellStart;
  int32_t* pre = cellPrefix + (size_t)f * cellsPerFrame + g.cellStart;
  int run = 0;  // identical in every thread
  for (int cb = 0; cb < g.nCells; cb += 256) {
    const int c = cb + tid;
    const int v = c < g.nCells ? cnt[c] : 0;
    // wave inclusive scan
    // wave inclusive scan (LDS permutes: this kernel runs next to the VALU-bound ones, see k_octree.hip octree_wave_incl_scan)
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) waveTot[wave] = x;
    __syncthreads();
    int base = run;
    for (int w = 0; w < wave; w++) base += waveTot[w];
    run += waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
    if (c < g.nCells) pre[c] = base + x - v;
    __syncthreads();
  }
  if (tid == 0) candCount[(size_t)f * nlevels + l] = run;
  // four threads per cell: most cells hold a handful of candidates, so the copies run in parallel
  // (and each thread's chain of dependent copies is short) instead of a walk over the cells
  // (pre[] of other threads is visible: every scan round ends with a barrier)
  const Candidate* sl = slots + (size_t)f * slotsPerFrame;
  Candidate* out = cand + (size_t)f * slotsPerFrame + g.slotStart;
  for (int t = tid; t < 4 * g.nCells; t += 256) {
    const int c = t >> 2, part = t & 3;
    const int n = cnt[c], b = pre[c];
    const Candidate* src = sl + cells[g.cellStart + c].slotBase;
    for (int i = part; i < n; i += 4) out[b + i] = src[i];
  }
}

void launch_gather_candidates(hipStream_t s, const CellDesc* d_cells, const LevelGeom* d_lv,
                              int nlevels, int nFrames, const Candidate* d_slots,
                              int slotsPerFrame, const uint16_t* d_cellCount, int cellsPerFrame,
                              Candidate* d_cand, int32_t* d_candCount, int32_t* d_cellPrefix) {
  if (nFrames <= 0) return;
  hipLaunchKernelGGL(k_gather_candidates, dim3(nFrames, nlevels), dim3(256), 0, s, d_cells, d_lv,
                     d_slots, slotsPerFrame, d_cellCount, cellsPerFrame, d_cand, d_candCount,
                     d_cellPrefix, nlevels);
}

}  // namespace orbfe
