// k_pyramid.hip -- bilinear pyramid level (cv::resize INTER_LINEAR, src/ORBextractor.cc:1219).
// Integer fixed-point with host-built coefficient tables.  k_resize_flat is the production kernel
// (4 output columns x 8 output rows per thread, flat row-block-major thread numbering);
// k_resize keeps the older strip form for inputs without packed tables, k_resize_generic serves
// unaligned pitches and scale factors above 2, k_copy2d realigns caller-owned level-0 images.
#include <cstdlib>
#include <cstring>
#include <string>

#include "kernels.h"

namespace orbfe {

size_t occupancy_pad_bytes(const char* name, int default_kb) {
  const std::string key = std::string("ORBFE_PAD_") + name;
  const char* env = getenv(key.c_str());
  int kb = env ? atoi(env) : default_kb;
  if (kb < 0) kb = 0;
  if (kb > 60) kb = 60;  // stays below the 64 KB default limit of dynamic LDS
  return (size_t)kb * 1024;
}

namespace {
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int kRowsPerThread = 8;

// horizontal interpolation of one source row for the thread's 4 output columns:
// 8 source bytes starting at column `sxb` (two or three aligned dword loads), then per column
// one v_perm_b32 (bytes S[sx], S[sx+1] as a u16 pair) and one v_dot2_u32_u16 with (a0, a1).
struct HRow { int h[4]; };
__device__ __forceinline__ HRow hrow(const uint8_t* rowp, uint32_t mis, const uint32_t (&sel)[4],
                                     const uint32_t (&al)[4]) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(rowp);  // 4-byte aligned
  const uint32_t d0 = p[0], d1 = p[1];
  const uint32_t d2 = mis ? p[2] : 0u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, mis);
  const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, mis);
  HRow r;
#pragma unroll
  for (int k = 0; k < 4; k++)
    r.h[k] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel[k])),
                                         __builtin_bit_cast(u16x2, al[k]), 0u, false) >> 4;
  return r;
}
}  // namespace

// Each thread produces 4 adjacent output columns x 8 output rows, walking down the source rows
// so that every horizontally interpolated source row is computed once and reused by the two
// output rows that blend it.  Requires pitch % 4 == 0 (owned levels always; caller-owned level 0
// falls back to k_resize_generic otherwise) and scale <= 2 (the 4 columns' taps span <= 8 bytes).
__global__ __launch_bounds__(256) void k_resize(LevelView src, LevelViewMut dst,
                                                const int32_t* __restrict__ xofs,
                                                const int16_t* __restrict__ alpha,
                                                const int32_t* __restrict__ yofs,
                                                const int16_t* __restrict__ beta) {
  const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  // the row strip is wave-uniform: keep it (and everything derived from it) in scalar registers
  const int dyBase = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + (threadIdx.x >> 6)) * kRowsPerThread);
  const int f = blockIdx.z;
  if (dx0 >= dst.w || dyBase >= dst.h) return;
  // per-thread column setup
  int sx[4];
  uint32_t al[4], sel[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int dx = dx0 + k < dst.w ? dx0 + k : dst.w - 1;
    sx[k] = xofs[dx];
    al[k] = (uint32_t)(uint16_t)alpha[2 * dx] | ((uint32_t)(uint16_t)alpha[2 * dx + 1] << 16);
  }
  int sxb = sx[0];
  if (sxb > src.w - 8) sxb = src.w - 8;  // keep the 8-byte window inside the row
  if (sxb < 0) sxb = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t o = (uint32_t)(sx[k] - sxb);
    const uint32_t o1 = o + 1 < 8 ? o + 1 : o;  // tap sx+1 beyond the window only when its weight is 0
    sel[k] = o | 0x0c00u | (o1 << 16) | 0x0c000000u;
  }
  const uint8_t* S = src.base + (size_t)f * src.frameStride + sxb;
  const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(S) & 3);
  S -= mis;
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + dx0;

  // two interpolated source rows stay in registers (X, Y); every output row reuses whichever of
  // them it can (scalar row indices -> scalar branches, no register shuffling)
  int idxX = -1, idxY = -1;
  HRow X = {}, Y = {};
#pragma unroll 1
  for (int r = 0; r < kRowsPerThread; r++) {
    const int dy = dyBase + r;
    if (dy >= dst.h) break;
    const int sy = yofs[dy];
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    const int r0 = sy < 0 ? 0 : (sy >= src.h ? src.h - 1 : sy);
    const int r1 = sy + 1 < 0 ? 0 : (sy + 1 >= src.h ? src.h - 1 : sy + 1);
    bool aIsX;
    if (r0 == idxY) {
      if (r1 != idxX && r1 != idxY) { X = hrow(S + (size_t)r1 * src.pitch, mis, sel, al); idxX = r1; }
      aIsX = false;
    } else {
      if (r0 != idxX) { X = hrow(S + (size_t)r0 * src.pitch, mis, sel, al); idxX = r0; }
      if (r1 != idxX && r1 != idxY) { Y = hrow(S + (size_t)r1 * src.pitch, mis, sel, al); idxY = r1; }
      aIsX = true;
    }
    // (A, B) = rows (r0, r1)
    const bool bIsX = (r1 == idxX);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int hA = aIsX ? X.h[k] : Y.h[k];
      const int hB = bIsX ? X.h[k] : Y.h[k];
      const int v = (((b0 * hA) >> 16) + ((b1 * hB) >> 16) + 2) >> 2;
      packed |= (uint32_t)(v & 0xff) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(D + (size_t)dy * dst.pitch) = packed;  // pitch % 64 == 0: aligned, in-row
  }
}

// Flat form: one thread = 4 adjacent output columns x kFlatRows consecutive output rows, threads
// numbered row-block-major over the level, so a wavefront is 64 consecutive column groups wherever
// the rows break -- lanes stay busy on levels whose width is not a multiple of 256 pixels (257,
// 309, 370 ... lost up to half of the lanes of the strip form above).  All per-column / per-row
// coefficients come packed from host-built tables (ResizeTables::colrec / rowrec); all row windows
// are requested before the first is used, so one memory round trip serves 16 output pixels; an
// interpolated source row is reused by the next output row when the whole wave agrees (scalar
// branch, no per-lane control flow); (b*h)>>16 is one v_mul_hi_u32 against b<<16.
#ifndef ORBFE_FLAT_ROWS
#define ORBFE_FLAT_ROWS 8
#endif
constexpr int kFlatRows = ORBFE_FLAT_ROWS;
#ifndef ORBFE_RESIZE_WAVES
#define ORBFE_RESIZE_WAVES 1
#endif
// the rows of one thread: rr[r] = row record of output row dy0 + r (vector or scalar registers, see the kernel)
template <bool UNALIGNED>
__device__ __forceinline__ void resize_flat_rows(const LevelView& src, const LevelViewMut& dst, int f, int gx, int dy0,
                                                 const uint4 s4, const uint4 a4, int sxb, const uint4 (&rr)[kFlatRows]) {
  const uint32_t sel[4] = {s4.x, s4.y, s4.z, s4.w}, al[4] = {a4.x, a4.y, a4.z, a4.w};
  // UNALIGNED: the source rows start at arbitrary byte addresses (a caller-owned level 0 with an odd stride): the
  // 8-byte window is read with two byte-aligned dword loads (fine on gfx950, profiles/r02_unaligned.txt), no third
  // load and no byte shift; otherwise the window is cut out of aligned dwords by v_alignbyte
  const uint8_t* S = src.base + (size_t)f * src.frameStride + sxb;
  const uint32_t mis = UNALIGNED ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(S) & 3);
  S -= mis;
  struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };
  // Output row r blends source rows (rr[r].x, rr[r].y); at pyramid scales rr[r].x is usually
  // rr[r-1].y.  When that holds for every lane of the wave (a scalar condition: the lanes of a wave
  // share their rows except where a wave straddles two row blocks) the interpolated row is reused
  // instead of being fetched and interpolated again.
  bool reuse[kFlatRows];
  reuse[0] = false;
#pragma unroll
  for (int r = 1; r < kFlatRows; r++) reuse[r] = __all(rr[r].x == rr[r - 1].y) != 0;
  uint32_t w[2 * kFlatRows][3];  // raw 12-byte windows of the source rows still needed
#pragma unroll
  for (int r = 0; r < kFlatRows; r++) {
    if (!reuse[r]) {
      const uint8_t* qa = S + (size_t)rr[r].x * src.pitch;
      if (UNALIGNED) {
        w[2 * r][0] = reinterpret_cast<const U1u*>(qa)->x; w[2 * r][1] = reinterpret_cast<const U1u*>(qa + 4)->x; w[2 * r][2] = 0u;
      } else {
        const uint32_t* pa = reinterpret_cast<const uint32_t*>(qa);
        w[2 * r][0] = pa[0]; w[2 * r][1] = pa[1]; w[2 * r][2] = mis ? pa[2] : 0u;
      }
    }
    const uint8_t* qb = S + (size_t)rr[r].y * src.pitch;
    if (UNALIGNED) {
      w[2 * r + 1][0] = reinterpret_cast<const U1u*>(qb)->x; w[2 * r + 1][1] = reinterpret_cast<const U1u*>(qb + 4)->x; w[2 * r + 1][2] = 0u;
    } else {
      const uint32_t* pb = reinterpret_cast<const uint32_t*>(qb);
      w[2 * r + 1][0] = pb[0]; w[2 * r + 1][1] = pb[1]; w[2 * r + 1][2] = mis ? pb[2] : 0u;
    }
  }
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + 4 * gx;
  uint32_t hA[4], hB[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < kFlatRows; r++) {
    if (reuse[r]) {
#pragma unroll
      for (int k = 0; k < 4; k++) hA[k] = hB[k];
    } else {
      const uint32_t lo = __builtin_amdgcn_alignbyte(w[2 * r][1], w[2 * r][0], mis);
      const uint32_t hi = __builtin_amdgcn_alignbyte(w[2 * r][2], w[2 * r][1], mis);
#pragma unroll
      for (int k = 0; k < 4; k++)
        hA[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel[k])),
                                       __builtin_bit_cast(u16x2, al[k]), 0u, false) >> 4;
    }
    {
      const uint32_t lo = __builtin_amdgcn_alignbyte(w[2 * r + 1][1], w[2 * r + 1][0], mis);
      const uint32_t hi = __builtin_amdgcn_alignbyte(w[2 * r + 1][2], w[2 * r + 1][1], mis);
#pragma unroll
      for (int k = 0; k < 4; k++)
        hB[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel[k])),
                                       __builtin_bit_cast(u16x2, al[k]), 0u, false) >> 4;
    }
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t v = (__umulhi(hA[k], rr[r].z) + __umulhi(hB[k], rr[r].w) + 2u) >> 2;
      packed |= v << (8 * k);
    }
    if (dy0 + r < dst.h) *reinterpret_cast<uint32_t*>(D + (size_t)(dy0 + r) * dst.pitch) = packed;
  }
}

template <bool UNALIGNED>
__global__ __launch_bounds__(256, ORBFE_RESIZE_WAVES) void k_resize_flat(LevelView src, LevelViewMut dst,
                                                     const uint4* __restrict__ colrec,
                                                     const uint4* __restrict__ rowrec, int ngx,
                                                     uint32_t magic, int total) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (gid >= total) return;
  const int rb = (int)__umulhi((uint32_t)gid, magic);  // gid / ngx, exact for gid * ngx < 2^32
  const int gx = gid - rb * ngx;
  const uint4 s4 = colrec[3 * gx], a4 = colrec[3 * gx + 1];
  const int sxb = (int)colrec[3 * gx + 2].x;
  // The 8 row records are the same for every lane unless the wave straddles two row blocks (one wave in ~5 on a VGA
  // level): then they come through the scalar cache into SGPRs -- 8 s_load_dwordx4 instead of 8 vector 16-byte loads
  // per lane, which were 40 % of the kernel's traffic through the texture addresser (the unit it saturates first:
  // halving the occupancy did not slow it, profiles/r02_occupancy_pad_sweep.txt).
  const int rb0 = __builtin_amdgcn_readfirstlane(rb);
  uint4 rr[kFlatRows];
  if (__all(rb == rb0)) {
    const int dyU = rb0 * kFlatRows;
#pragma unroll
    for (int r = 0; r < kFlatRows; r++) rr[r] = rowrec[dyU + r < dst.h ? dyU + r : dst.h - 1];
    resize_flat_rows<UNALIGNED>(src, dst, f, gx, dyU, s4, a4, sxb, rr);
  } else {
    const int dy0 = rb * kFlatRows;
#pragma unroll
    for (int r = 0; r < kFlatRows; r++) rr[r] = rowrec[dy0 + r < dst.h ? dy0 + r : dst.h - 1];
    resize_flat_rows<UNALIGNED>(src, dst, f, gx, dy0, s4, a4, sxb, rr);
  }
}

// Generic path (any pitch/alignment/scale): 4 output pixels per thread, byte loads.
__global__ __launch_bounds__(256) void k_resize_generic(LevelView src, LevelViewMut dst,
                                                        const int32_t* __restrict__ xofs,
                                                        const int16_t* __restrict__ alpha,
                                                        const int32_t* __restrict__ yofs,
                                                        const int16_t* __restrict__ beta) {
  const int gx = blockIdx.x * 64 + threadIdx.x;  // group of 4 output columns
  const int dy = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (dy >= dst.h) return;
  const int dx0 = gx * 4;
  if (dx0 >= dst.w) return;
  const int sy = yofs[dy];
  const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
  int r0y = sy < 0 ? 0 : (sy >= src.h ? src.h - 1 : sy);
  int r1y = sy + 1 < 0 ? 0 : (sy + 1 >= src.h ? src.h - 1 : sy + 1);
  const uint8_t* S0 = src.base + (size_t)f * src.frameStride + (size_t)r0y * src.pitch;
  const uint8_t* S1 = src.base + (size_t)f * src.frameStride + (size_t)r1y * src.pitch;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int dx = dx0 + k;
    if (dx < dst.w) {
      const int sx = xofs[dx];
      const int sx1 = sx + 1 < src.w ? sx + 1 : sx;
      const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
      const int h0 = S0[sx] * a0 + S0[sx1] * a1;
      const int h1 = S1[sx] * a0 + S1[sx1] * a1;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      packed |= (uint32_t)(v & 0xff) << (8 * k);
    }
  }
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + (size_t)dy * dst.pitch + dx0;
  *reinterpret_cast<uint32_t*>(D) = packed;  // pitch % 64 == 0, dx0 % 4 == 0: aligned, in-row
}

void launch_resize(hipStream_t s, LevelView src, LevelViewMut dst, const int32_t* d_xofs,
                   const int16_t* d_alpha, const int32_t* d_yofs, const int16_t* d_beta,
                   const uint32_t* d_colrec, const uint32_t* d_rowrec, int nFrames) {
  const bool aligned = (src.pitch & 3) == 0 && (src.frameStride & 3) == 0 && (reinterpret_cast<uintptr_t>(src.base) & 3) == 0;
  const bool shape = src.w >= 8 && (long long)src.w <= 2LL * dst.w;
  const bool fast = aligned && shape;
  const int ngx = (dst.w + 3) / 4;
  const long long total = (long long)ngx * ((dst.h + kFlatRows - 1) / kFlatRows);
  if (shape && d_colrec && d_rowrec && total * ngx < (1LL << 32) && total > 0) {
    const uint32_t magic = (uint32_t)((1ULL << 32) / (uint32_t)ngx) + 1u;
    const dim3 grid((unsigned)((total + 255) / 256), nFrames);
    static const size_t pad = occupancy_pad_bytes("RESIZE", 0);
    // byte-aligned dword loads for every level since the end of round 2 (same-box A/B: pipeline +1-3 %, two
    // v_alignbyte and a load per source row fewer); <false> is kept for A/B builds (-DORBFE_RESIZE_ALIGNED_LOADS)
#ifdef ORBFE_RESIZE_ALIGNED_LOADS
    if (aligned)
#else
    if (false)
#endif
      hipLaunchKernelGGL(k_resize_flat<false>, grid, dim3(256), pad, s, src, dst, reinterpret_cast<const uint4*>(d_colrec),
                         reinterpret_cast<const uint4*>(d_rowrec), ngx, magic, (int)total);
    else  // caller-owned level 0 at an odd stride: byte-aligned dword loads
      hipLaunchKernelGGL(k_resize_flat<true>, grid, dim3(256), pad, s, src, dst, reinterpret_cast<const uint4*>(d_colrec),
                         reinterpret_cast<const uint4*>(d_rowrec), ngx, magic, (int)total);
  } else if (fast) {
    dim3 grid((dst.w + 255) / 256, (dst.h + 4 * kRowsPerThread - 1) / (4 * kRowsPerThread), nFrames);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, src, dst, d_xofs, d_alpha, d_yofs, d_beta);
  } else {
    dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, nFrames);
    hipLaunchKernelGGL(k_resize_generic, grid, dim3(64, 4), 0, s, src, dst, d_xofs, d_alpha, d_yofs, d_beta);
  }
}

// level 0 = copy of the caller's image (the copyMakeBorder of :1231 without the dead border) into the
// pitch-aligned slab; used when the caller's rows are not 4-byte aligned (e.g. KITTI's 1241-byte rows),
// so that every later kernel reads aligned dwords.  A thread moves 16 bytes of one row: five aligned
// source dwords -> v_alignbyte -> one 16-byte store; threads are numbered row-major over (row, chunk).
__global__ __launch_bounds__(256) void k_copy2d(LevelView src, LevelViewMut dst, int chunks, uint32_t magic, int total) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (gid >= total) return;
  const int y = (int)__umulhi((uint32_t)gid, magic);  // gid / chunks
  const int x = (gid - y * chunks) * 16;
  const uint8_t* S = src.base + (size_t)f * src.frameStride + (size_t)y * src.pitch + x;
  uint8_t* D = dst.base + (size_t)f * dst.frameStride + (size_t)y * dst.pitch + x;  // 16-byte aligned: pitch % 64 == 0
  const int left = dst.w - x;  // bytes of this row still to copy (> 0)
  uint4 o;
  if (left >= 16 + 3) {  // the five dwords stay inside the source row
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(S) & 3);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(S - mis);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = mis ? p[4] : 0u;
    o.x = __builtin_amdgcn_alignbyte(d1, d0, mis);
    o.y = __builtin_amdgcn_alignbyte(d2, d1, mis);
    o.z = __builtin_amdgcn_alignbyte(d3, d2, mis);
    o.w = __builtin_amdgcn_alignbyte(d4, d3, mis);
  } else {  // row tail: bytes, zero beyond the row
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    for (int k = 0; k < 16 && k < left; k++) v[k >> 2] |= (uint32_t)S[k] << (8 * (k & 3));
    o = make_uint4(v[0], v[1], v[2], v[3]);
  }
  *reinterpret_cast<uint4*>(D) = o;
}

void launch_copy2d(hipStream_t s, LevelView src, LevelViewMut dst, int nFrames) {
  if (dst.w <= 0 || dst.h <= 0 || nFrames <= 0) return;
  const int chunks = (dst.w + 15) / 16;
  const int total = chunks * dst.h;
  const uint32_t magic = (uint32_t)((1ULL << 32) / (uint32_t)chunks) + 1u;  // exact for gid * chunks < 2^32
  hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((total + 255) / 256), nFrames), dim3(256), 0, s, src, dst, chunks, magic, total);
}


// ---- k_pyramid_chain: see kernels.h (PyrChainArgs) ----
// i / d for i < 2^24, 1 <= d < 256 by multiply-high (2^32 / d + 1: the error term i * (m * d - 2^32) stays below 2^32)
__device__ __forceinline__ uint32_t chain_div(uint32_t i, uint32_t d, uint32_t magic) { return d == 1u ? i : __umulhi(i, magic); }
__device__ __forceinline__ uint32_t chain_magic(uint32_t d) { return d <= 1u ? 0u : (uint32_t)(0x100000000ull / d) + 1u; }
__global__ __launch_bounds__(256) void k_pyramid_chain(const PyrChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const ChainTile T = a.tiles[blockIdx.x];  // block-uniform (scalar loads)
  const int f = blockIdx.y, tid = threadIdx.x;
  uint8_t* bufIn = lds;
  uint8_t* bufOut = lds + a.bufA;
  uint2* colrec = reinterpret_cast<uint2*>(lds + a.bufA + a.bufB);  // per column of the output rectangle: taps | weights
  uint2* rowrec = colrec + a.maxW;                                   // per row
  // ---- the level-0 rectangle, 4 bytes per request where the row allows ----
  ChainRect rIn = T.r[0];
  int pitchIn = (rIn.w + 3) & ~3;
  {
    const uint8_t* S = a.l0.base + (size_t)f * a.l0.frameStride + (size_t)rIn.y0 * a.l0.pitch + rIn.x0;
    const int dwPerRow = pitchIn >> 2;
    const uint32_t inv = chain_magic((uint32_t)dwPerRow);
    struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };
    for (int i = tid; i < rIn.h * dwPerRow; i += 256) {
      const int y = (int)chain_div((uint32_t)i, (uint32_t)dwPerRow, inv), c = i - y * dwPerRow;
      const uint8_t* p = S + (size_t)y * a.l0.pitch + 4 * c;
      uint32_t v;
      if (rIn.x0 + 4 * c + 3 < a.w[0]) v = reinterpret_cast<const U1u*>(p)->x;
      else {  // the last dword of a row at the right image border: stay inside the row
        v = 0;
        for (int b = 0; b < 4; b++)
          if (rIn.x0 + 4 * c + b < a.w[0]) v |= (uint32_t)p[b] << (8 * b);
      }
      *reinterpret_cast<uint32_t*>(bufIn + y * pitchIn + 4 * c) = v;
    }
  }
  // ---- level k from level k-1, k = 1 .. T.level (cv::resize INTER_LINEAR in 11-bit fixed point, the arithmetic of
  //      k_resize_generic: rows and the second tap clamped at use) ----
  for (int k = 1; k <= T.level; k++) {
    const ChainRect rOut = T.r[k];
    const int Wp = a.w[k - 1], Hp = a.h[k - 1];
    const int32_t* xo = a.xofs[k];
    const int16_t* al = a.alpha[k];
    const int32_t* yo = a.yofs[k];
    const int16_t* be = a.beta[k];
    for (int x = tid; x < rOut.w; x += 256) {
      const int dx = rOut.x0 + x;
      const int sx = xo[dx];
      const int sx1 = sx + 1 < Wp ? sx + 1 : sx;
      colrec[x] = make_uint2((uint32_t)(sx - rIn.x0) | ((uint32_t)(sx1 - rIn.x0) << 16),
                             (uint32_t)(uint16_t)al[2 * dx] | ((uint32_t)(uint16_t)al[2 * dx + 1] << 16));
    }
    for (int y = tid; y < rOut.h; y += 256) {
      const int dy = rOut.y0 + y;
      const int sy = yo[dy];
      const int r0 = sy < 0 ? 0 : (sy >= Hp ? Hp - 1 : sy);
      const int r1 = sy + 1 < 0 ? 0 : (sy + 1 >= Hp ? Hp - 1 : sy + 1);
      rowrec[y] = make_uint2((uint32_t)(r0 - rIn.y0) | ((uint32_t)(r1 - rIn.y0) << 16),
                             (uint32_t)(uint16_t)be[2 * dy] | ((uint32_t)(uint16_t)be[2 * dy + 1] << 16));
    }
    __syncthreads();  // (also: the previous level's stores to bufIn)
    const int pitchOut = (rOut.w + 3) & ~3;
    const uint32_t invW = chain_magic((uint32_t)rOut.w);
    for (int i = tid; i < rOut.w * rOut.h; i += 256) {
      const int y = (int)chain_div((uint32_t)i, (uint32_t)rOut.w, invW), x = i - y * rOut.w;
      const uint2 c = colrec[x], r = rowrec[y];
      const uint8_t* S0 = bufIn + (r.x & 0xffffu) * pitchIn;
      const uint8_t* S1 = bufIn + (r.x >> 16) * pitchIn;
      const int o0 = (int)(c.x & 0xffffu), o1 = (int)(c.x >> 16);
      const int a0 = (int)(c.y & 0xffffu), a1 = (int)(c.y >> 16);
      const int b0 = (int)(r.y & 0xffffu), b1 = (int)(r.y >> 16);
      const int h0 = S0[o0] * a0 + S0[o1] * a1;
      const int h1 = S1[o0] * a0 + S1[o1] * a1;
      bufOut[y * pitchOut + x] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
    __syncthreads();
    uint8_t* t = bufIn; bufIn = bufOut; bufOut = t;
    rIn = rOut;
    pitchIn = pitchOut;
  }
  // ---- the tile itself (owned levels: pitch % 64 == 0, tile origin % 32 == 0) ----
  const LevelViewMut D = a.lv[T.level];
  uint8_t* out = D.base + (size_t)f * D.frameStride + (size_t)rIn.y0 * D.pitch + rIn.x0;
  const int dwPerRow = pitchIn >> 2;  // <= 8
  for (int i = tid; i < rIn.h * dwPerRow; i += 256) {
    const int y = i / dwPerRow, c = i - y * dwPerRow;
    const uint32_t v = *reinterpret_cast<const uint32_t*>(bufIn + y * pitchIn + 4 * c);
    uint8_t* p = out + (size_t)y * D.pitch + 4 * c;
    if (4 * c + 3 < rIn.w) *reinterpret_cast<uint32_t*>(p) = v;
    else
      for (int b = 0; 4 * c + b < rIn.w; b++) p[b] = (uint8_t)(v >> (8 * b));
  }
}

size_t pyramid_chain_lds_bytes(const PyrChainArgs& a) { return (size_t)a.bufA + a.bufB + (size_t)(a.maxW + a.maxH) * sizeof(uint2); }

void launch_pyramid_chain(hipStream_t s, const PyrChainArgs& a, int nTiles, int nFrames) {
  if (nTiles <= 0 || nFrames <= 0) return;
  hipLaunchKernelGGL(k_pyramid_chain, dim3(nTiles, nFrames), dim3(256), pyramid_chain_lds_bytes(a), s, a);
}

}  // namespace orbfe
