// Micro-benchmark (gfx950), second set: issue rate of the candidate instructions for a leaner FAST kernel
// (round 3) and the SEMANTICS of v_pk_maximum3_f16 / v_pk_minimum3_f16 on small unsigned integers read as
// f16 bit patterns (0 .. 1023 = f16 denormals, whose order as floats is their order as integers).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate2 tools/ubench/valu_rate2.hip
//   tools/ubench/valu_rate2 > profiles/rNN_valu_rate2.txt
//
// Method as valu_rate.hip: 256 CUs x 8 workgroups of 256 threads, 16 independent accumulators, HIP events.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N_IT 4096
#define N_ACC 16

#define OPS(X)                                                                \
  X(V_MAX_I32, "v_max_i32 %0, %0, %1")                                        \
  X(V_MIN_U32, "v_min_u32 %0, %0, %1")                                        \
  X(V_AND_B32, "v_and_b32 %0, %0, %1")                                        \
  X(V_OR_B32, "v_or_b32 %0, %0, %1")                                          \
  X(V_SUB_U32, "v_sub_u32 %0, %0, %1")                                        \
  X(V_LSHLREV, "v_lshlrev_b32 %0, 1, %0")                                     \
  X(V_MAX_U16, "v_max_u16 %0, %0, %1")                                        \
  X(V_SUB_U16, "v_sub_u16 %0, %0, %1")                                        \
  X(V_MAX3_I32, "v_max3_i32 %0, %0, %1, %2")                                  \
  X(V_MED3_I32, "v_med3_i32 %0, %0, %1, %2")                                  \
  X(V_MAX3_U16, "v_max3_u16 %0, %0, %1, %2")                                  \
  X(V_PK_MAX_I16, "v_pk_max_i16 %0, %0, %1")                                  \
  X(V_PK_MAX_U16, "v_pk_max_u16 %0, %0, %1")                                  \
  X(V_PK_SUB_I16, "v_pk_sub_i16 %0, %0, %1")                                  \
  X(V_PK_ADD_U16, "v_pk_add_u16 %0, %0, %1")                                  \
  X(V_PK_MAX_F16, "v_pk_max_f16 %0, %0, %1")                                  \
  X(V_PK_MAXIMUM3_F16, "v_pk_maximum3_f16 %0, %0, %1, %2")                    \
  X(V_PK_MINIMUM3_F16, "v_pk_minimum3_f16 %0, %0, %1, %2")                    \
  X(V_MAXIMUM3_F32, "v_maximum3_f32 %0, %0, %1, %2")                          \
  X(V_AND_OR_B32, "v_and_or_b32 %0, %0, %1, %2")                              \
  X(V_OR3_B32, "v_or3_b32 %0, %0, %1, %2")                                    \
  X(V_LSHL_OR_B32, "v_lshl_or_b32 %0, %0, 1, %2")                             \
  X(V_ADD3_U32, "v_add3_u32 %0, %0, %1, %2")                                  \
  X(V_BFE_U32, "v_bfe_u32 %0, %0, 1, 31")                                     \
  X(V_BFI_B32, "v_bfi_b32 %0, %0, %1, %2")                                    \
  X(V_SAD_U8, "v_sad_u8 %0, %0, %1, %2")                                      \
  X(V_MOV_DPP, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")   \
  X(V_ADD_DPP, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
  X(V_MAX_SDWA, "v_max_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2") \
  X(V_CVT_F32_UBYTE0, "v_cvt_f32_ubyte0 %0, %0")                               \
  X(V_MOV_B32, "v_mov_b32 %0, %1")                                            \
  X(V_NOT_B32, "v_not_b32 %0, %0")                                            \
  X(V_CNDMASK, "v_cndmask_b32 %0, %0, %1, vcc")                               \
  X(V_CMP_GT_U32, "v_cmp_gt_u32 vcc, %0, %1")                                 \
  X(V_LSHLREV_7, "v_lshlrev_b32 %0, 7, %1")                                   \
  X(V_LSHLREV_V, "v_lshlrev_b32 %0, %1, %0")                                  \
  X(V_LSHRREV_V, "v_lshrrev_b32 %0, %1, %0")                                  \
  X(V_ASHRREV, "v_ashrrev_i32 %0, 3, %0")                                     \
  X(V_ADD_U16, "v_add_u16 %0, %0, %1")                                        \
  X(V_MIN_U16, "v_min_u16 %0, %0, %1")                                        \
  X(V_MAX_I16, "v_max_i16 %0, %0, %1")                                        \
  X(V_MUL_LO_U16, "v_mul_lo_u16 %0, %0, %1")                                  \
  X(V_LSHRREV_B16, "v_lshrrev_b16 %0, 1, %0")                                 \
  X(V_MAD_U16, "v_mad_u16 %0, %0, %1, %2")                                    \
  X(V_ADD_F32, "v_add_f32 %0, %0, %1")                                        \
  X(V_MUL_F32, "v_mul_f32 %0, %0, %1")                                        \
  X(V_MAX_F32, "v_max_f32 %0, %0, %1")                                        \
  X(V_MIN_F32, "v_min_f32 %0, %0, %1")                                        \
  X(V_FMAC_F32, "v_fmac_f32 %0, %1, %2")                                      \
  X(V_MAX_F16, "v_max_f16 %0, %0, %1")                                        \
  X(V_CVT_F32_U32, "v_cvt_f32_u32 %0, %0")                                    \
  X(V_CVT_U32_F32, "v_cvt_u32_f32 %0, %0")                                    \
  X(V_RNDNE_F32, "v_rndne_f32 %0, %0")                                        \
  X(V_LSHL_ADD_U32, "v_lshl_add_u32 %0, %0, 1, %2")                           \
  X(V_ADD_LSHL_U32, "v_add_lshl_u32 %0, %0, %1, 1")                           \
  X(V_XAD_U32, "v_xad_u32 %0, %0, %1, %2")                                    \
  X(V_FFBL_B32, "v_ffbl_b32 %0, %0")                                          \
  X(V_BFREV_B32, "v_bfrev_b32 %0, %0")                                        \
  X(V_DOT2C_I32_I16, "v_dot2c_i32_i16 %0, %1, %2")                            \
  X(V_DOT4C_I32_I8, "v_dot4c_i32_i8 %0, %1, %2")                              \
  X(V_PK_MUL_LO_U16, "v_pk_mul_lo_u16 %0, %0, %1")                            \
  X(V_PK_LSHRREV_B16, "v_pk_lshrrev_b16 %0, 1, %0")                           \
  X(V_PK_ASHRREV_I16, "v_pk_ashrrev_i16 %0, 15, %0")                          \
  X(V_BITOP3_B32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")                  \
  X(V_SAD_U32, "v_sad_u32 %0, %0, %1, %2")                                    \
  X(V_ADD_CO_U32, "v_add_co_u32 %0, vcc, %0, %1")                             \
  X(V_READLANE_LIKE_BPERMUTE, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")

enum Op {
#define X(name, text) name,
  OPS(X)
#undef X
      N_OPS
};
static const char* kOpText[N_OPS] = {
#define X(name, text) text,
    OPS(X)
#undef X
};

template <int OP>
__device__ __forceinline__ void body(uint32_t (&a)[N_ACC], uint32_t b, uint32_t c) {
#pragma unroll
  for (int i = 0; i < N_ACC; i++) {
#define X(name, text) \
  if (OP == name) asm volatile(text : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    OPS(X)
#undef X
  }
}

// 64-bit accumulators / register pairs: v_mad_u64_u32 (the compiler's choice for a 32-bit a*b+c whose operands may
// exceed 24 bits), v_lshl_add_u64 (64-bit pointer + offset), v_pk_mul_f32 / v_pk_add_f32 (what the SLP vectoriser makes
// of two independent fp32 operations)
enum Op64 { MAD_U64_U32, LSHL_ADD_U64, PK_MUL_F32, PK_ADD_F32, N_OPS64 };
static const char* kOp64Text[N_OPS64] = {"v_mad_u64_u32 %0, vcc, %1, %2, %0", "v_lshl_add_u64 %0, %0, 1, %3", "v_pk_mul_f32 %0, %0, %3",
                                         "v_pk_add_f32 %0, %0, %3"};
template <int OP>
__global__ __launch_bounds__(256) void k_rate64(uint64_t* out, uint32_t seed) {
  uint64_t a[N_ACC];
#pragma unroll
  for (int i = 0; i < N_ACC; i++) a[i] = (uint64_t)seed * (threadIdx.x + i + 1);
  const uint32_t b = seed ^ 0x00120123u, c = 0x00020c00u + seed;
  const uint64_t d = ((uint64_t)0x3f800000u << 32) | 0x3f800001u;
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int i = 0; i < N_ACC; i++) {
      if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c), "v"(d) : "vcc");
      if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %3" : "+v"(a[i]) : "v"(b), "v"(c), "v"(d) : "vcc");
      if (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %3" : "+v"(a[i]) : "v"(b), "v"(c), "v"(d) : "vcc");
      if (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %3" : "+v"(a[i]) : "v"(b), "v"(c), "v"(d) : "vcc");
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < N_ACC; i++) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run64(int nCU) {
  const int blocks = nCU * 8;
  uint64_t* d;
  hipMalloc(&d, 8ull * 256 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate64<OP>), dim3(blocks), dim3(256), 0, 0, d, 7u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate64<OP>), dim3(blocks), dim3(256), 0, 0, d, 7u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)N_IT * N_ACC * 4.0 * blocks;
  printf("%-100s %8.3f ms  %7.1f G wave-instr/s chip = %.2f cyc/wave-instr/SIMD @2.4GHz\n", kOp64Text[OP], ms, winst / ms / 1e6,
         (double)nCU * 4 * 2.4e9 / (winst / (ms * 1e-3)));
  hipFree(d);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t seed) {
  uint32_t a[N_ACC];
#pragma unroll
  for (int i = 0; i < N_ACC; i++) a[i] = seed * (threadIdx.x + i + 1);
  uint32_t b = seed ^ 0x00120123u, c = 0x00020c00u + seed;
  for (int it = 0; it < N_IT; it++) body<OP>(a, b, c);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < N_ACC; i++) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(int nCU) {
  const int wavesPerSimd = 8, blocks = nCU * wavesPerSimd;
  uint32_t* d;
  hipMalloc(&d, 4ull * 256 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, d, 7u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), 0, 0, d, 7u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)N_IT * N_ACC * 4.0 * blocks;
  printf("%-100s %8.3f ms  %7.1f G wave-instr/s chip = %.2f cyc/wave-instr/SIMD @2.4GHz\n", kOpText[OP], ms, winst / ms / 1e6,
         (double)nCU * 4 * 2.4e9 / (winst / (ms * 1e-3)));
  hipFree(d);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int OP>
struct RunAll {
  static void go(int nCU) {
    run<OP>(nCU);
    RunAll<OP + 1>::go(nCU);
  }
};
template <>
struct RunAll<N_OPS> {
  static void go(int) {}
};

// semantics: every triple (x, y, z) of a sample set, both halves, against integer max / min
__global__ void k_sem(const uint32_t* __restrict__ in, uint32_t* __restrict__ outMax, uint32_t* __restrict__ outMin, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  uint32_t r, s;
  asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(s) : "v"(x), "v"(y), "v"(z));
  outMax[i] = r;
  outMin[i] = s;
}

static int semantics(uint32_t limit) {
  const int n = 1 << 20;
  std::vector<uint32_t> h(3 * n), rmax(n), rmin(n);
  uint64_t st = 0x9E3779B97F4A7C15ull ^ limit;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 20); };
  for (auto& v : h) v = (rnd() % limit) | ((rnd() % limit) << 16);
  for (int k = 0; k < 3 * 4096; k++) h[k] = (k % limit) | (((k * 7) % limit) << 16);  // runs incl. 0 and equal values
  uint32_t *d, *dmax, *dmin;
  hipMalloc(&d, 12ull * n); hipMalloc(&dmax, 4ull * n); hipMalloc(&dmin, 4ull * n);
  hipMemcpy(d, h.data(), 12ull * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, d, dmax, dmin, n);
  hipMemcpy(rmax.data(), dmax, 4ull * n, hipMemcpyDeviceToHost);
  hipMemcpy(rmin.data(), dmin, 4ull * n, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    uint32_t emax = 0, emin = 0;
    for (int hh = 0; hh < 2; hh++) {
      const uint32_t a = (h[3 * i] >> (16 * hh)) & 0xffff, b = (h[3 * i + 1] >> (16 * hh)) & 0xffff, c = (h[3 * i + 2] >> (16 * hh)) & 0xffff;
      const uint32_t mx = a > b ? (a > c ? a : c) : (b > c ? b : c), mn = a < b ? (a < c ? a : c) : (b < c ? b : c);
      emax |= mx << (16 * hh);
      emin |= mn << (16 * hh);
    }
    if (emax != rmax[i] || emin != rmin[i]) {
      if (bad < 4) printf("  MISMATCH x=%08x y=%08x z=%08x max=%08x (want %08x) min=%08x (want %08x)\n", h[3 * i], h[3 * i + 1], h[3 * i + 2], rmax[i], emax, rmin[i], emin);
      bad++;
    }
  }
  printf("# v_pk_maximum3_f16 / v_pk_minimum3_f16 on u16 pairs < %u read as f16 bit patterns: %d of %d triples differ from integer max / min\n",
         limit, bad, n);
  hipFree(d); hipFree(dmax); hipFree(dmin);
  return bad;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int nCU = p.multiProcessorCount;
  printf("# %s  CUs=%d  (8 waves per SIMD, full EXEC, N_IT=%d x %d instr per wave)\n", p.name, nCU, N_IT, N_ACC);
  int bad = semantics(256) + semantics(1024) + semantics(0x7c00);
  RunAll<0>::go(nCU);
  run64<MAD_U64_U32>(nCU);
  run64<LSHL_ADD_U64>(nCU);
  run64<PK_MUL_F32>(nCU);
  run64<PK_ADD_F32>(nCU);
  return bad ? 1 : 0;
}
