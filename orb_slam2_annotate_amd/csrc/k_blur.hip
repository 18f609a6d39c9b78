// k_blur.hip -- cv::GaussianBlur(7x7, sigma=2, BORDER_REFLECT_101) on a u8 level
// (the `workingMat` of src/ORBextractor.cc:1169-1175).  The arithmetic is a SPEC (orb_spec.h, kBlurSpec*),
// selectable per extractor because the reference does not pin its OpenCV version:
//   SPEC 0 (default)  OpenCV >= 3.4.1 / 4.x bit-exact fixed-point path: separable kernel
//                     [18,34,48,56,48,34,18]/256 (sums to 256), one rounding (x + 2^15) >> 16;
//   SPEC 1            OpenCV 2.4.x / 3.0-3.3 generic C++ path: every tap rounded on its own,
//                     [18,34,49,55,49,34,18]/256 (sums to 257), saturate_cast<uchar>((x + 2^15) >> 16);
//   SPEC 2            the same versions built with SSE2 (x86 desktop builds): the column pass of the first
//                     width & ~3 columns runs in float and ends in cvtps2dq, i.e. round-half-to-EVEN of x / 2^16
//                     (the float sums are exact below 256), the last width & 3 columns take the SPEC 1 form.
// First pass exact in 8.8 (255 * 257 = 65535 still fits the packed u16), second pass in 16.16.
// Both passes are exact integer sums, so their order is free:
//   1. a 64x64 output tile stages (64+6) x (64+8) source bytes in LDS as aligned dwords
//      (reflect-101 resolved while loading);
//   2. VERTICAL pass, packed 16-bit: a thread takes 4 adjacent columns of 4 rows (the byte -> u16
//      split of the 10 source dwords is shared), two pixels per VALU lane-op (v_pk_add_u16 /
//      v_pk_mad_u16), result kept in LDS as natural-order u16; row blocks below the level's last
//      row are skipped;
//   3. HORIZONTAL pass on u16 pairs with v_dot2_u32_u16 (two taps per instruction, 32-bit
//      accumulate), rounding, one 32-bit coalesced store per 4 pixels.
// Every source byte is read once per tile (+ halo), every output written once; measured: 61 % VALU
// issue, the rest is waiting for the tile's own staging loads.
// HF = true (round 4, second half) swaps the passes -- both are exact integer sums, so the bytes are the same:
//   2'. HORIZONTAL pass on the staged BYTES with v_dot4_u32_u8 (four taps per instruction: 10 per 4 pixels; the 8.8
//       row sums <= 255 * 257 fit 16 bits), a thread takes 4 columns of a ROW PAIR and leaves one dword per column in
//       LDS = (h[2p][c], h[2p+1][c]);
//   3'. VERTICAL pass with v_dot2_u32_u16 on those row pairs (two taps per instruction, 32-bit accumulate), a thread
//       owns 4 columns x 4 rows: five 16-byte LDS reads, 64 dot2, rounding, one 32-bit store per row.
//   ~160 instead of 216 VALU instructions per wave for the two passes (ISA, dynamic: no byte -> u16 split, no repacking).
#include <atomic>
#include <cstdlib>

#include "kernels.h"

namespace orbfe {

namespace {
// tile shapes (output columns x rows): 64 x 64 (round 1) and 128 x 32 -- see launch_blur7_levels

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t k, uint32_t c) {
  return __builtin_amdgcn_udot2(as_u2(a), as_u2(k), c, false);
}
__device__ __forceinline__ int reflect101c(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);  // clamp only matters for never-used tile cells
}
constexpr uint32_t pk(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }
}  // namespace

// All pyramid levels of all frames in ONE launch (the levels are independent): the grid is the
// concatenation of every level's tiles, tileStart[l] = first work item of level l.
struct BlurBatch {
  LevelView src[kMaxLevels];
  LevelViewMut dst[kMaxLevels];
  int tilesX[kMaxLevels], tilesY[kMaxLevels];
  unsigned tileStart[kMaxLevels + 1];  // in units of (tile, frame) work items
  uint32_t perFrameMagic[kMaxLevels], tilesXMagic[kMaxLevels];  // udiv_magic multipliers
  int nlevels, nFrames;
  // RESIZE form (one level per launch): the tile that is staged for the blur also produces the part of the NEXT
  // pyramid level whose bilinear taps start inside it (ResizeTables::tileGx / tileDy give the ownership)
  LevelViewMut next;
  const uint4* colrec; const uint4* rowrec; const int32_t* tileGx; const int32_t* tileDy;
  int ablate;  // 0; $ORBFE_BLUR_ABLATE, timing experiments (tools/ablate_blur.sh): 1 no blur stores, 2 no next-level stores, 4 no staging loads, 8 no resize at all
};

// ABL: the $ORBFE_BLUR_ABLATE hooks are compiled in (tools/ablate_blur.sh); the product build has `ablate` = constant 0 -- as a
// run-time field the store switch cost a compare, three scalar instructions and a branch per trip of the resize loop (cf. k_desc.hip)
template <int SPEC, int kBW, int kBH, bool RESIZE, bool HF, bool ABL = false>
__global__ __launch_bounds__(256) void k_blur7(const BlurBatch bb) {
  const int ablate = ABL ? bb.ablate : 0;
  constexpr int kVDW = (kBW + 8) / 4;  // dword columns the blur needs: bx-4 .. bx+kBW+3
  // RESIZE: one more staged dword per row -- an 8-byte tap window that starts in the tile's last column ends at bx+kBW+6
  constexpr int kTDW = kVDW + (RESIZE ? 2 : 0);  // (19 dwords are needed; 20 keep the rows 16-byte aligned for b128 staging)
  constexpr int kTH = kBH + 6;         // rows by-3 .. by+kBH+2
  constexpr int kSR = 256 / kTDW;      // tile rows one staging step of the workgroup covers (14 / 7)
  constexpr int kSteps = (kTH + kSR - 1) / kSR;
  constexpr int kGX = kBW / 4;         // 4-pixel output groups per tile row (16 / 32)
  constexpr int kHR = 256 / kGX;       // output rows one step of the horizontal pass covers (16 / 8)
  constexpr uint32_t K0 = 18, K1 = 34, K2 = SPEC == 0 ? 48 : 49, K3 = SPEC == 0 ? 56 : 55;
  constexpr int kVR = 5;                             // output rows per vertical-pass item (see phase 2)
  constexpr int kVB = (kBH + kVR - 1) / kVR;         // row blocks of a full tile: 13 / 7 -- the last one runs past the tile
  __shared__ __attribute__((aligned(16))) uint32_t tin[(kVB * kVR + 6) * kTDW];  // source bytes: kTH staged rows (+ the last block's overhang)
  constexpr int kPairs = (kTH + 1) / 2;            // HF: row pairs of the staged tile (35 / 19)
  __shared__ uint2 vbuf[HF ? 1 : kVB * kVR * kVDW];  // vertical sums, 4 u16 per entry
  __shared__ __attribute__((aligned(16))) uint32_t hpair[HF ? kPairs * kBW : 4];  // HF: horizontal sums, (row 2p, row 2p+1) of a column per dword
  __shared__ uint4 s_col[RESIZE ? 48 : 1];         // RESIZE: column records of the groups this tile owns
  __shared__ uint4 s_row[RESIZE ? 80 : 1];         //         and row records of the output rows it owns (<= 65 at scale >= 1)
  __shared__ int s_g0, s_nG, s_d0, s_d1;
  const int tid = threadIdx.x;
  // XCD-aware work mapping (speed only): block b -> work item (b % 8) * chunk + b / 8, so the
  // tiles one XCD's L2 sees are a contiguous raster run and share their halo rows there
  const unsigned chunk = gridDim.x >> 3;
  const unsigned work = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  if (work >= bb.tileStart[RESIZE ? 1 : bb.nlevels]) return;
  int l = 0;
  if (!RESIZE)  // (the fused form is launched per level: l = 0 is a constant and its fields sit at fixed argument offsets)
    for (int k = 1; k < bb.nlevels; k++)
      if (work >= bb.tileStart[k]) l = k;   // block-uniform (scalar) search
  const LevelView src = bb.src[l];
  const LevelViewMut dst = bb.dst[l];
  const int tilesX = bb.tilesX[l];
  const unsigned perFrame = (unsigned)tilesX * (unsigned)bb.tilesY[l];
  const unsigned w0 = work - bb.tileStart[l];
  const int f = (int)udiv_magic(w0, perFrame, bb.perFrameMagic[l]);          // w0 / perFrame
  const unsigned rem = w0 - (unsigned)f * perFrame;
  const int tyI = (int)udiv_magic(rem, (uint32_t)tilesX, bb.tilesXMagic[l]);  // rem / tilesX
  const int bx = (int)(rem - (unsigned)tyI * (unsigned)tilesX) * kBW, by = tyI * kBH;
  const int rowsValid = dst.h - by < kBH ? dst.h - by : kBH;  // output rows of this tile inside the level
  // the vertical pass computes kVR-row blocks: 5 rows make 13 x 18 = 234 (64 x 64 tile) / 7 x 34 = 238 (128 x 32) items, ONE
  // trip of the 256 threads (4-row blocks were 288 / 272 items: wave 0 went round twice and the other three waited for it
  // at the barrier)
  const int rowBlocks = (rowsValid + kVR - 1) / kVR;
  const int stageRows = rowsValid + 6;                        // <= kTH
  const uint8_t* S = src.base + (size_t)f * src.frameStride;
  // every staging load is ONE dword at whatever byte address the level and the tile give: global_load_dword takes
  // unaligned addresses on gfx950 (profiles/r02_unaligned.txt), so a caller-owned level 0 with an odd stride
  // (KITTI: 1241) runs the same paths as the handle's own 64-byte pitched levels
  struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };
  // ---- 1. stage.  The fused form issues ALL its memory requests before it waits for any: the two 16-byte pieces of the
  //      tile first (their addresses need nothing but the kernel arguments), then the ownership tables (scalar), then the
  //      column / row records -- every load unconditional on a clamped, always valid index, the LDS writes predicated
  //      afterwards.  (Written as guarded load + store pairs the compiler waited for each request in turn: five
  //      dependent round trips at the head of a workgroup whose arithmetic takes less than one.) ----
  const bool interior = bx >= 4 && bx - 4 + 4 * kTDW <= src.w && by >= 3 && by + kBH + 3 <= src.h;  // block-uniform
  const bool wideCols = RESIZE && bx >= 4 && bx - 4 + 4 * kTDW <= src.w;  // tile away from the left / right edges (block-uniform)
  struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
  constexpr int kParts = kTDW / 4;  // 5 x 16 bytes per tile row
  constexpr int kPieces = (kTH * kParts + 255) / 256;
  U4u q[kPieces] = {};
  if (wideCols && !(ablate & 4)) {
    // 16-byte requests (any byte address, profiles/r02_unaligned.txt) and ds_write_b128, 2 staging instructions per thread
    // instead of 6 -- the memory instructions of a wave, not the bytes, are what the texture addresser meters; top /
    // bottom tiles reflect the row
    const uint8_t* T0 = S + (bx - 4);  // block-uniform (scalar)
#pragma unroll
    for (int k = 0; k < kPieces; k++) {
      const int i = tid + 256 * k;
      int row = (int)((uint32_t)i / (uint32_t)kParts);
      const int part = i - row * kParts;
      row = row < stageRows ? row : 0;  // (threads past the tile re-read its first row; they store nothing)
      const int sy = interior ? by - 3 + row : reflect101c(by - 3 + row, src.h);
      q[k] = *reinterpret_cast<const U4u*>(T0 + ((uint32_t)sy * (uint32_t)src.pitch + 16u * (uint32_t)part));
    }
  }
  if (RESIZE) {  // ownership of the next level (scalar loads) and the owned groups' column / row records, for phase 1b
    const int txI = (int)(rem - (unsigned)tyI * (unsigned)tilesX);
    const int g0 = bb.tileGx[txI], nG = bb.tileGx[txI + 1] - g0;
    const int d0 = bb.tileDy[tyI], d1 = bb.tileDy[tyI + 1];
    if (tid == 0) { s_g0 = g0; s_nG = nG; s_d0 = d0; s_d1 = d1; }  // d1 - d0 <= 80 (build_resize_tables)
    const int nColRec = 3 * ((bb.next.w + 3) >> 2);  // records of the level: 3 per 4-pixel group, one per output row
    int ci = 3 * g0 + (tid - (256 - 48)), ri = d0 + tid;
    ci = ci < 0 ? 0 : (ci < nColRec ? ci : nColRec - 1);
    ri = ri < bb.next.h ? ri : bb.next.h - 1;
    const uint4 rc = bb.colrec[ci], rw = bb.rowrec[ri];
    if (tid >= 256 - 48 && tid - (256 - 48) < 3 * nG) s_col[tid - (256 - 48)] = rc;
    if (tid < d1 - d0 && tid < 80) s_row[tid] = rw;
  }
  if (wideCols) {
#pragma unroll
    for (int k = 0; k < kPieces; k++) {
      const int i = tid + 256 * k;
      const int row = (int)((uint32_t)i / (uint32_t)kParts), part = i - row * kParts;
      if (row < stageRows) *reinterpret_cast<uint4*>(&tin[row * kTDW + 4 * part]) = make_uint4(q[k].x, q[k].y, q[k].z, q[k].w);
    }
  } else if (tid < kSR * kTDW) {
    const int ty0 = (int)((uint32_t)tid / (uint32_t)kTDW), tj = tid - ty0 * kTDW;
    if (interior) {
      // interior tile (block-uniform): no reflection, every dword is an aligned in-row load
      const uint8_t* T0 = S + (size_t)(by - 3) * src.pitch + (bx - 4);     // block-uniform (scalar) tile origin
      const uint32_t o0 = (uint32_t)ty0 * (uint32_t)src.pitch + 4u * (uint32_t)tj;  // 32-bit lane offset
#pragma unroll
      for (int k = 0; k < kSteps; k++)
        if ((k + 1) * kSR <= kTH || ty0 + kSR * k < kTH)
          tin[(ty0 + kSR * k) * kTDW + tj] = reinterpret_cast<const U1u*>(T0 + (o0 + (uint32_t)(kSR * k) * (uint32_t)src.pitch))->x;
    } else if (bx >= 4 && bx - 4 + 4 * kTDW <= src.w) {
      // top / bottom tile away from the left and right edges: every dword is still an aligned in-row
      // load, only the row index is reflected
      const uint32_t c0 = (uint32_t)(bx - 4) + 4u * (uint32_t)tj;
      for (int ty = ty0; ty < stageRows; ty += kSR) {
        const int sy = reflect101c(by - 3 + ty, src.h);
        tin[ty * kTDW + tj] = reinterpret_cast<const U1u*>(S + ((uint32_t)sy * (uint32_t)src.pitch + c0))->x;
      }
    } else {
      const int c = bx - 4 + 4 * tj;
      const bool inRow = c >= 0 && c + 3 < src.w;
      int cx[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cx[k] = reflect101c(c + k, src.w);
      for (int ty = ty0; ty < stageRows; ty += kSR) {
        const int sy = reflect101c(by - 3 + ty, src.h);
        const uint8_t* row = S + (size_t)sy * src.pitch;
        uint32_t v;
        if (inRow) {
          // one dword at whatever byte address the row and column give (odd caller strides included):
          // global_load_dword takes unaligned addresses on gfx950 (profiles/r02_unaligned.txt)
          v = reinterpret_cast<const U1u*>(row + c)->x;
        } else {
          v = (uint32_t)row[cx[0]] | ((uint32_t)row[cx[1]] << 8) | ((uint32_t)row[cx[2]] << 16) | ((uint32_t)row[cx[3]] << 24);
        }
        tin[ty * kTDW + tj] = v;
      }
    }
  }
  __syncthreads();
  if (RESIZE) {
    // ---- 1b. the next pyramid level from the staged tile (cv::resize INTER_LINEAR, the arithmetic of k_resize_flat:
    //      host-built column / row records, v_perm + v_dot2 horizontally, two v_mul_hi vertically).  The tile owns the
    //      column groups [g0, g0+nG) x output rows [d0, d1) whose taps start inside it (at most 16 groups per 64 source
    //      columns, scale >= 1; ~14 x 54 at 1.2); the items are numbered flat, row-major, so the 256 threads stay busy
    //      (a (group, row slot) thread grid ran 73 % full).  The taps are read from LDS: the level is fetched from
    //      memory once for blur and resize together ----
    // Thread = (column group g, row slot): g and everything that depends on it alone -- the group's column record (4 v_perm
    // selectors, 4 coefficient pairs), its window in the tile, the store column -- are fixed for the thread; a trip of the loop
    // moves rowsPerPass = 256 / nG output rows down (round 3 numbered the items flat, i -> (i / nG, i % nG), and paid the
    // division, three 16-byte record reads and three 64-bit multiply-adds of index arithmetic per item: 75 VALU per 4 pixels,
    // now 58).  256 - nG * rowsPerPass < nG threads idle (nG = 13, 14: 9, 4).
    const int nG = s_nG, nRows = s_d1 - s_d0;
    uint8_t* Nf = bb.next.base + (size_t)f * bb.next.frameStride;
    if (nG > 0 && nRows > 0 && !(ablate & 8)) {
      const uint32_t invG = 65536u / (uint32_t)nG + 1u;      // exact t / nG for t < 4096 (nG <= 16)
      const int rowsPerPass = (int)((256u * invG) >> 16);   // 256 / nG (block-uniform)
      int r = (int)(((uint32_t)tid * invG) >> 16);
      const int g = tid - r * nG;
      if (r < rowsPerPass) {
        const uint4 s4 = s_col[3 * g], a4 = s_col[3 * g + 1];
        const int wofs = (int)s_col[3 * g + 2].x - (bx - 4);  // window start, tile-local byte column (>= 4)
        const uint32_t mis = (uint32_t)wofs & 3u;
        const uint32_t* tw = &tin[wofs >> 2];
        const uint32_t sel[4] = {s4.x, s4.y, s4.z, s4.w}, al[4] = {a4.x, a4.y, a4.z, a4.w};
        // block-uniform frame base (scalar) + 32-bit lane offset, stepped by rowsPerPass rows per trip
        uint32_t off = (uint32_t)(s_d0 + r) * (uint32_t)bb.next.pitch + 4u * (uint32_t)(s_g0 + g);  // owned levels: pitch % 64 == 0, in-row
        const uint32_t offStep = (uint32_t)rowsPerPass * (uint32_t)bb.next.pitch;
        for (; r < nRows; r += rowsPerPass, off += offStep) {
          const uint4 rr = s_row[r];  // source rows r0, r1 (clamped), b0 << 16, b1 << 16
          const uint32_t* pa = tw + __mul24((int)rr.x - (by - 3), kTDW);  // (rows of the tile: 24-bit multiply, not the 64-bit mad)
          const uint32_t* pb = tw + __mul24((int)rr.y - (by - 3), kTDW);
          const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
          const uint32_t loA = __builtin_amdgcn_alignbyte(a1, a0, mis), hiA = __builtin_amdgcn_alignbyte(a2, a1, mis);
          const uint32_t loB = __builtin_amdgcn_alignbyte(b1, b0, mis), hiB = __builtin_amdgcn_alignbyte(b2, b1, mis);
          uint32_t packed = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t hA = __builtin_amdgcn_udot2(as_u2(__builtin_amdgcn_perm(hiA, loA, sel[k])), as_u2(al[k]), 0u, false) >> 4;
            const uint32_t hB = __builtin_amdgcn_udot2(as_u2(__builtin_amdgcn_perm(hiB, loB, sel[k])), as_u2(al[k]), 0u, false) >> 4;
            const uint32_t v = (__umulhi(hA, rr.z) + __umulhi(hB, rr.w) + 2u) >> 2;
            packed |= v << (8 * k);
          }
          if (!(ablate & 2) || packed == 0x12345678u) *reinterpret_cast<uint32_t*>(Nf + off) = packed;
        }
      }
    }
  }
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + (size_t)by * dst.pitch + bx;
  // rounding of four 16.16 sums (the four columns of a group in one row) and their bytes as one dword
  auto finish = [&](uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, const bool simdCols) -> uint32_t {
    if (SPEC != 0) {
      if (SPEC == 2 && simdCols) {  // round-half-to-even of x / 2^16: + 0x7fff + (bit 16 of x)
        o0 += 0x7fffu + ((o0 >> 16) & 1u); o1 += 0x7fffu + ((o1 >> 16) & 1u);
        o2 += 0x7fffu + ((o2 >> 16) & 1u); o3 += 0x7fffu + ((o3 >> 16) & 1u);
      }
      // the taps sum to 257: 255 * 257 * 257 + 2^15 reaches 257 << 16 -- saturate_cast<uchar>
      o0 = min(o0, 0xffffffu); o1 = min(o1, 0xffffffu); o2 = min(o2, 0xffffffu); o3 = min(o3, 0xffffffu);
    }
    // the rounded sums are < 2^24: byte 2 of each is the result; two v_perm gather them
    const uint32_t lo = __builtin_amdgcn_perm(o1, o0, 0x0c0c0602u);   // (o0.b2, o1.b2, 0, 0)
    const uint32_t hi = __builtin_amdgcn_perm(o3, o2, 0x06020c0cu);   // (0, 0, o2.b2, o3.b2)
    return lo | hi;
  };
  if constexpr (HF) {
    // ---- 2'. horizontal pass on bytes: item = (row pair p, 4-pixel group g); output x of the tile takes the staged bytes
    //          x+1 .. x+7, i.e. the group's three dwords d0 d1 d2 hold every tap of its four outputs ----
    constexpr uint32_t W0a = K0 << 8 | K1 << 16 | K2 << 24, W0b = K3 | K2 << 8 | K1 << 16 | K0 << 24;       // o0: d0 bytes 1-3, d1 bytes 0-3
    constexpr uint32_t W1a = K0 << 16 | K1 << 24, W1b = K2 | K3 << 8 | K2 << 16 | K1 << 24, W1c = K0;       // o1: d0 2-3, d1, d2 byte 0
    constexpr uint32_t W2a = K0 << 24, W2b = K1 | K2 << 8 | K3 << 16 | K2 << 24, W2c = K1 | K0 << 8;       // o2: d0 3, d1, d2 0-1
    constexpr uint32_t W3b = K0 | K1 << 8 | K2 << 16 | K3 << 24, W3c = K2 | K1 << 8 | K0 << 16;            // o3: d1, d2 0-2
    const int nPairs = (stageRows + 1) >> 1;  // (an odd last row pairs with a stale one; the sums of rows past the staged ones are never read)
    for (int i = tid; i < nPairs * kGX; i += 256) {
      const int p = (int)((uint32_t)i / (uint32_t)kGX), g = i - p * kGX;
      const uint32_t* tp = &tin[(2 * p) * kTDW + g];
      uint32_t hs[2][4];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const uint32_t d0 = tp[r * kTDW], d1 = tp[r * kTDW + 1], d2 = tp[r * kTDW + 2];
        hs[r][0] = __builtin_amdgcn_udot4(d1, W0b, __builtin_amdgcn_udot4(d0, W0a, 0u, false), false);
        hs[r][1] = __builtin_amdgcn_udot4(d2, W1c, __builtin_amdgcn_udot4(d1, W1b, __builtin_amdgcn_udot4(d0, W1a, 0u, false), false), false);
        hs[r][2] = __builtin_amdgcn_udot4(d2, W2c, __builtin_amdgcn_udot4(d1, W2b, __builtin_amdgcn_udot4(d0, W2a, 0u, false), false), false);
        hs[r][3] = __builtin_amdgcn_udot4(d2, W3c, __builtin_amdgcn_udot4(d1, W3b, 0u, false), false);
      }
      *reinterpret_cast<uint4*>(&hpair[p * kBW + 4 * g]) =
          make_uint4(hs[0][0] | hs[1][0] << 16, hs[0][1] | hs[1][1] << 16, hs[0][2] | hs[1][2] << 16, hs[0][3] | hs[1][3] << 16);
    }
    __syncthreads();
    // ---- 3'. vertical pass on the row pairs: item = (4-row block rb, group g); output row t of the tile takes the staged rows
    //          t .. t+6: for even t the pairs t/2 .. t/2+3 weighted (K0,K1) (K2,K3) (K2,K1) (K0,0), for odd t the pairs
    //          (t-1)/2 .. with (0,K0) (K1,K2) (K3,K2) (K1,K0) ----
    const int rowBlocks4 = (rowsValid + 3) >> 2;   // rowBlocks4 * kGX <= 256: one item per thread
    const int rb = (int)((uint32_t)tid / (uint32_t)kGX), g = tid - rb * kGX;
    if (rb < rowBlocks4 && bx + 4 * g < dst.w) {
      const uint4* hp = reinterpret_cast<const uint4*>(&hpair[(2 * rb) * kBW + 4 * g]);
      uint32_t q[5][4];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const uint4 v = hp[k * (kBW / 4)];
        q[k][0] = v.x; q[k][1] = v.y; q[k][2] = v.z; q[k][3] = v.w;
      }
      const bool simd = SPEC == 2 && bx + 4 * g < (dst.w & ~3);
      const uint32_t R = (SPEC == 2 && simd) ? 0u : (1u << 15);
      uint32_t hv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k0 = r >> 1;
        uint32_t o[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          uint32_t a;
          if ((r & 1) == 0) {
            a = dot2(q[k0][c], pk(K0, K1), R);
            a = dot2(q[k0 + 1][c], pk(K2, K3), a);
            a = dot2(q[k0 + 2][c], pk(K2, K1), a);
            a = dot2(q[k0 + 3][c], pk(K0, 0), a);
          } else {
            a = dot2(q[k0][c], pk(0, K0), R);
            a = dot2(q[k0 + 1][c], pk(K1, K2), a);
            a = dot2(q[k0 + 2][c], pk(K3, K2), a);
            a = dot2(q[k0 + 3][c], pk(K1, K0), a);
          }
          o[c] = a;
        }
        hv[r] = finish(o[0], o[1], o[2], o[3], simd);
      }
      const uint32_t off = (uint32_t)(4 * rb) * (uint32_t)dst.pitch + 4u * (uint32_t)g;
      const int nr = rowsValid - 4 * rb;  // >= 1 rows of the block lie inside the level
      if (ablate & 1) {  // (block-uniform timing switch of tools/ablate_blur.sh: the sums stay live, nothing is stored)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (hv[r] == 0x12345678u && r < nr) *reinterpret_cast<uint32_t*>(D + (off + (uint32_t)r * (uint32_t)dst.pitch)) = hv[r];
      } else if (nr >= 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t*>(D + (off + (uint32_t)r * (uint32_t)dst.pitch)) = hv[r];
      } else {
#pragma unroll
        for (int r = 0; r < 3; r++)
          if (r < nr) *reinterpret_cast<uint32_t*>(D + (off + (uint32_t)r * (uint32_t)dst.pitch)) = hv[r];
      }
    }
  } else {
  // ---- 2. vertical pass: a thread owns 4 adjacent columns x kVR output rows; the kVR + 6 source dwords are
  //         split into u16 pairs once and shared by the rows; 8.8 sums, two pixels per lane-op ----
  for (int i = tid; i < rowBlocks * kVDW; i += 256) {
    const int rb = (int)((uint32_t)i / (uint32_t)kVDW), tj = i - rb * kVDW;
    const uint32_t* tp = &tin[(kVR * rb) * kTDW + tj];
    u16x2 te[kVR + 6], to[kVR + 6];  // even bytes (0,2) and odd bytes (1,3) of each source dword
#pragma unroll
    for (int j = 0; j < kVR + 6; j++) {
      const uint32_t r = tp[j * kTDW];  // (rows below the staged ones -- the last block of a tile -- hold stale bytes; their sums are never read)
      te[j] = as_u2(r & 0x00ff00ffu);
      to[j] = as_u2(__builtin_amdgcn_perm(r, r, 0x0c030c01u));
    }
    const u16x2 k18 = {(unsigned short)K0, (unsigned short)K0}, k34 = {(unsigned short)K1, (unsigned short)K1},
                k48 = {(unsigned short)K2, (unsigned short)K2}, k56 = {(unsigned short)K3, (unsigned short)K3};
    uint2* vo = &vbuf[(kVR * rb) * kVDW + tj];
#pragma unroll
    for (int r = 0; r < kVR; r++) {
      u16x2 a = (te[r] + te[r + 6]) * k18;
      a = (te[r + 1] + te[r + 5]) * k34 + a;
      a = (te[r + 2] + te[r + 4]) * k48 + a;
      const uint32_t A = as_u(te[r + 3] * k56 + a);   // (v0, v2)
      u16x2 b = (to[r] + to[r + 6]) * k18;
      b = (to[r + 1] + to[r + 5]) * k34 + b;
      b = (to[r + 2] + to[r + 4]) * k48 + b;
      const uint32_t B = as_u(to[r + 3] * k56 + b);   // (v1, v3)
      uint2 o;
      o.x = __builtin_amdgcn_perm(B, A, 0x05040100u);  // (v0, v1)
      o.y = __builtin_amdgcn_perm(B, A, 0x07060302u);  // (v2, v3)
      vo[r * kVDW] = o;
    }
  }
  __syncthreads();
  // ---- 3. horizontal pass on u16 pairs: out[x] = (sum_i K[i] * v[x+4+i-3] + 2^15) >> 16 ----
  // thread = (row mod kHR, column group): the tile origin D is block-uniform (scalar), the thread keeps a
  // 32-bit byte offset and an LDS pointer and steps both by kHR rows per iteration
  // one 4-pixel group from three consecutive vbuf entries (d_k = (v'[2k], v'[2k+1]) with v' indexed from the group's column)
  auto hgroup = [&](const uint2 e0, const uint2 e1, const uint2 e2, const bool simdCols) -> uint32_t {
    const uint32_t d0 = e0.x, d1 = e0.y, d2 = e1.x, d3 = e1.y, d4 = e2.x, d5 = e2.y;
    // SPEC 2, SIMD columns: no bias here, round-half-even below
    const uint32_t R = (SPEC == 2 && simdCols) ? 0u : (1u << 15);
    uint32_t o0 = dot2(d0, pk(0, K0), R);   // taps v'1..v'7
    o0 = dot2(d1, pk(K1, K2), o0);
    o0 = dot2(d2, pk(K3, K2), o0);
    o0 = dot2(d3, pk(K1, K0), o0);
    uint32_t o1 = dot2(d1, pk(K0, K1), R);  // taps v'2..v'8
    o1 = dot2(d2, pk(K2, K3), o1);
    o1 = dot2(d3, pk(K2, K1), o1);
    o1 = dot2(d4, pk(K0, 0), o1);
    uint32_t o2 = dot2(d1, pk(0, K0), R);   // taps v'3..v'9
    o2 = dot2(d2, pk(K1, K2), o2);
    o2 = dot2(d3, pk(K3, K2), o2);
    o2 = dot2(d4, pk(K1, K0), o2);
    uint32_t o3 = dot2(d2, pk(K0, K1), R);  // taps v'4..v'10
    o3 = dot2(d3, pk(K2, K3), o3);
    o3 = dot2(d4, pk(K2, K1), o3);
    o3 = dot2(d5, pk(K0, 0), o3);
    return finish(o0, o1, o2, o3, simdCols);
  };
  {
    // (measured and dropped: 16 pixels per thread from three ds_read_b128 with one 16-byte store -- KITTI +0.8 %, TUM -1.9 %,
    // EuRoC -0.5 % in a same-box A/B: noise)
    const int gx = tid % kGX;
    uint32_t off = (uint32_t)(tid / kGX) * (uint32_t)dst.pitch + 4u * (uint32_t)gx;
    const uint2* vp = &vbuf[(tid / kGX) * kVDW + gx];
    const bool colIn = bx + 4 * gx < dst.w;
    for (int row = tid / kGX; row < rowsValid; row += kHR, off += (uint32_t)kHR * (uint32_t)dst.pitch, vp += kHR * kVDW) {
      if (!colIn) continue;
      const uint32_t hv = hgroup(vp[0], vp[1], vp[2], SPEC == 2 && bx + 4 * gx < (dst.w & ~3));
      if (!(ablate & 1) || hv == 0x12345678u) *reinterpret_cast<uint32_t*>(D + off) = hv;
    }
  }
  }
}

// Order of the two separable passes (process-wide; the bytes do not depend on it): 1 = horizontal on bytes, then vertical
// on row pairs (HF, the default), 0 = vertical packed-16, then horizontal on u16 pairs (rounds 1-3).
// $ORBFE_BLUR_HFIRST / orbfe_set_blur_pass_order.
static std::atomic<int> g_blurPassOrder{-1};
int blur_pass_order() {
  int v = g_blurPassOrder.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* env = getenv("ORBFE_BLUR_HFIRST");
    v = env ? (atoi(env) != 0) : 1;
    g_blurPassOrder.store(v, std::memory_order_relaxed);
  }
  return v;
}
void set_blur_pass_order(int order) { g_blurPassOrder.store(order != 0, std::memory_order_relaxed); }

namespace {
template <int kBW, int kBH>
void launch_blur_tiles(hipStream_t s, const LevelView* src, const LevelViewMut* dst, int nlevels, int nFrames, int spec) {
  BlurBatch bb = {};
  bb.nlevels = nlevels;
  bb.nFrames = nFrames;
  unsigned total = 0;
  for (int l = 0; l < nlevels; l++) {
    bb.src[l] = src[l];
    bb.dst[l] = dst[l];
    bb.tilesX[l] = (dst[l].w + kBW - 1) / kBW;
    bb.tilesY[l] = (dst[l].h + kBH - 1) / kBH;
    bb.tileStart[l] = total;
    const unsigned perFrame = (unsigned)bb.tilesX[l] * (unsigned)bb.tilesY[l];
    bb.perFrameMagic[l] = udiv_magic_multiplier(perFrame);
    bb.tilesXMagic[l] = udiv_magic_multiplier((uint32_t)bb.tilesX[l]);
    total += (unsigned)bb.tilesX[l] * (unsigned)bb.tilesY[l] * (unsigned)nFrames;
  }
  bb.tileStart[nlevels] = total;
  if (total == 0) return;
  const dim3 grid((total + 7u) / 8u * 8u);
  static const size_t pad = occupancy_pad_bytes("BLUR", 0);
  if (blur_pass_order() != 0) {
    if (spec == kBlurSpecCv2Scalar) hipLaunchKernelGGL((k_blur7<1, kBW, kBH, false, true>), grid, dim3(256), pad, s, bb);
    else if (spec == kBlurSpecCv2Sse2) hipLaunchKernelGGL((k_blur7<2, kBW, kBH, false, true>), grid, dim3(256), pad, s, bb);
    else hipLaunchKernelGGL((k_blur7<0, kBW, kBH, false, true>), grid, dim3(256), pad, s, bb);
  } else {
    if (spec == kBlurSpecCv2Scalar) hipLaunchKernelGGL((k_blur7<1, kBW, kBH, false, false>), grid, dim3(256), pad, s, bb);
    else if (spec == kBlurSpecCv2Sse2) hipLaunchKernelGGL((k_blur7<2, kBW, kBH, false, false>), grid, dim3(256), pad, s, bb);
    else hipLaunchKernelGGL((k_blur7<0, kBW, kBH, false, false>), grid, dim3(256), pad, s, bb);
  }
}
}  // namespace

void launch_blur7_levels(hipStream_t s, const LevelView* src, const LevelViewMut* dst, int nlevels, int nFrames, int spec) {
  if (nlevels <= 0 || nFrames <= 0) return;
  static const int tile = getenv("ORBFE_BLUR_TILE") ? atoi(getenv("ORBFE_BLUR_TILE")) : 0;
  if (tile == 1) launch_blur_tiles<128, 32>(s, src, dst, nlevels, nFrames, spec);
  else launch_blur_tiles<64, 64>(s, src, dst, nlevels, nFrames, spec);
}

// One level: blur it and write the next pyramid level from the same staged tiles (see k_blur7<.., RESIZE = true>).
void launch_blur7_resize(hipStream_t s, LevelView src, LevelViewMut dst, LevelViewMut next, const uint32_t* d_colrec,
                         const uint32_t* d_rowrec, const int32_t* d_tileGx, const int32_t* d_tileDy, int nFrames, int spec) {
  if (dst.w <= 0 || dst.h <= 0 || nFrames <= 0) return;
  BlurBatch bb = {};
  bb.nlevels = 1;
  bb.nFrames = nFrames;
  bb.src[0] = src;
  bb.dst[0] = dst;
  bb.tilesX[0] = (dst.w + 63) / 64;
  bb.tilesY[0] = (dst.h + 63) / 64;
  bb.tileStart[0] = 0;
  const unsigned perFrame = (unsigned)bb.tilesX[0] * (unsigned)bb.tilesY[0];
  bb.perFrameMagic[0] = udiv_magic_multiplier(perFrame);
  bb.tilesXMagic[0] = udiv_magic_multiplier((uint32_t)bb.tilesX[0]);
  const unsigned total = perFrame * (unsigned)nFrames;
  bb.tileStart[1] = total;
  bb.next = next;
  bb.colrec = reinterpret_cast<const uint4*>(d_colrec);
  bb.rowrec = reinterpret_cast<const uint4*>(d_rowrec);
  bb.tileGx = d_tileGx;
  bb.tileDy = d_tileDy;
  static const int kAblate = getenv("ORBFE_BLUR_ABLATE") ? atoi(getenv("ORBFE_BLUR_ABLATE")) : 0;
  bb.ablate = kAblate;
  const dim3 grid((total + 7u) / 8u * 8u);
  if (kAblate) {  // (the ablation build exists for the default arithmetic only)
    if (blur_pass_order() != 0) hipLaunchKernelGGL((k_blur7<0, 64, 64, true, true, true>), grid, dim3(256), 0, s, bb);
    else hipLaunchKernelGGL((k_blur7<0, 64, 64, true, false, true>), grid, dim3(256), 0, s, bb);
  } else if (blur_pass_order() != 0) {
    if (spec == kBlurSpecCv2Scalar) hipLaunchKernelGGL((k_blur7<1, 64, 64, true, true>), grid, dim3(256), 0, s, bb);
    else if (spec == kBlurSpecCv2Sse2) hipLaunchKernelGGL((k_blur7<2, 64, 64, true, true>), grid, dim3(256), 0, s, bb);
    else hipLaunchKernelGGL((k_blur7<0, 64, 64, true, true>), grid, dim3(256), 0, s, bb);
  } else {
    if (spec == kBlurSpecCv2Scalar) hipLaunchKernelGGL((k_blur7<1, 64, 64, true, false>), grid, dim3(256), 0, s, bb);
    else if (spec == kBlurSpecCv2Sse2) hipLaunchKernelGGL((k_blur7<2, 64, 64, true, false>), grid, dim3(256), 0, s, bb);
    else hipLaunchKernelGGL((k_blur7<0, 64, 64, true, false>), grid, dim3(256), 0, s, bb);
  }
}

void launch_blur7(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames, int spec) {
  if (dst.w <= 0 || dst.h <= 0) return;
  launch_blur7_levels(s, &src, &dst, 1, nFrames, spec);
}

}  // namespace orbfe
