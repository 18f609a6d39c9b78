// Micro-benchmark (gfx950): VALU issue rate per SIMD as a function of resident waves per SIMD, for the
// plain ops the ORB kernels are made of, with a full and a half EXEC mask.  It settles the VALU ceiling
// bench.py prices `roofline.valu_issue` against (VERDICT r01 "weak" item 2): the micro-architecture
// guide's row "v_fma_f32 (wave64): 2 cyc (SIMD-32); one wave alone: 4".
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
//   tools/ubench/valu_rate > profiles/rNN_valu_rate.txt
//
// Method: grid = 256 CUs x k workgroups of 256 threads (4 waves: one per SIMD), k = waves per SIMD; no
// LDS, < 32 VGPRs, so all k*256 workgroups are resident at once.  Every wave runs N_IT x 16 independent
// instructions of ONE op (16 accumulators: a 4-cycle dependent latency can never be the limit).
// Two clocks: HIP events around the launch (chip-wide wave-instr/s) and s_memtime inside each wave
// (shader cycles per instruction as the wave sees them; x 1/k = the SIMD's issue interval).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define N_IT 8192
#define N_ACC 16
typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { ADD_U32, FMA_F32, MIN3_I32, PERM_B32, PK_FMA_F32, BCNT, XOR_B32, PK_MIN_I16,
          MUL_HI_U32, MUL_LO_U32, MUL_U32_U24, MAD_U32_U24, DOT2_U32_U16, DOT4_U32_U8, ALIGNBYTE, SAD_U16, PK_MAD_U16, LSHRREV, N_OPS };
static const char* kOpName[N_OPS] = {"v_add_u32", "v_fma_f32", "v_min3_i32", "v_perm_b32", "v_pk_fma_f32",
                                      "v_bcnt_u32_b32", "v_xor_b32", "v_pk_min_i16", "v_mul_hi_u32", "v_mul_lo_u32",
                                      "v_mul_u32_u24", "v_mad_u32_u24", "v_dot2_u32_u16", "v_dot4_u32_u8", "v_alignbyte_b32",
                                      "v_sad_u16", "v_pk_mad_u16", "v_lshrrev_b32"};

template <int OP>
__device__ __forceinline__ void body(uint32_t (&a)[N_ACC], uint32_t b, uint32_t c) {
#pragma unroll
  for (int i = 0; i < N_ACC; i++) {
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == MIN3_I32) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
    if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == DOT2_U32_U16) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == DOT4_U32_U8) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == SAD_U16) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
  }
}
template <>
__device__ __forceinline__ void body<PK_FMA_F32>(uint32_t (&a)[N_ACC], uint32_t b, uint32_t c) {
  // 8 register PAIRS, issued twice = 16 instructions per iteration, like the others
  f2 bb = {__builtin_bit_cast(float, b), __builtin_bit_cast(float, b)};
  f2 cc = {__builtin_bit_cast(float, c), __builtin_bit_cast(float, c)};
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int i = 0; i < N_ACC; i += 2) {
      f2 x = {__builtin_bit_cast(float, a[i]), __builtin_bit_cast(float, a[i + 1])};
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(bb), "v"(cc));
      a[i] = __builtin_bit_cast(uint32_t, x.x);
      a[i + 1] = __builtin_bit_cast(uint32_t, x.y);
    }
}

// HALF = 1: lanes 32..63 of every wave are inactive for the whole loop (EXEC = 0x00000000ffffffff)
template <int OP, int HALF>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint64_t* cycles, uint32_t seed) {
  uint32_t a[N_ACC];
#pragma unroll
  for (int i = 0; i < N_ACC; i++) a[i] = seed * (threadIdx.x + i + 1);
  uint32_t b = seed ^ 0x3f800123u, c = 0x0c020c00u + seed;
  uint64_t t0 = 0, t1 = 0;
  if (!HALF || (threadIdx.x & 63) < 32) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; it++) body<OP>(a, b, c);
    t1 = __builtin_amdgcn_s_memtime();
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < N_ACC; i++) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int HALF>
void run(int wavesPerSimd, int nCU) {
  const int blocks = nCU * wavesPerSimd;
  uint32_t* d;
  uint64_t* dc;
  hipMalloc(&d, 4ull * 256 * blocks);
  hipMalloc(&dc, 8ull * 4 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<OP, HALF>), dim3(blocks), dim3(256), 0, 0, d, dc, 7u);  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<OP, HALF>), dim3(blocks), dim3(256), 0, 0, d, dc, 7u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> cyc(4ull * blocks);
  hipMemcpy(cyc.data(), dc, 8ull * 4 * blocks, hipMemcpyDeviceToHost);
  std::sort(cyc.begin(), cyc.end());
  const double instrPerWave = (double)N_IT * N_ACC;
  const double medCyc = (double)cyc[cyc.size() / 2] / instrPerWave;  // s_memtime ticks per instruction, one wave
  const double winst = instrPerWave * 4.0 * blocks;
  // s_memtime ticks are shader cycles (MI355X_MICROARCH.md, cycle-constants table): ticks/instr of one wave
  // divided by the resident waves per SIMD = the SIMD's issue interval, independent of the 2.4 GHz assumption
  printf("%-15s exec=%-4s waves/SIMD=%d  %8.3f ms  %7.1f G wave-instr/s chip  = %.2f cyc/wave-instr/SIMD @2.4GHz"
         "  (per-wave s_memtime ticks/instr %.3f)\n",
         kOpName[OP], HALF ? "half" : "full", wavesPerSimd, ms, winst / ms / 1e6,
         (double)nCU * 4 * 2.4e9 / (winst / (ms * 1e-3)), medCyc);
  hipFree(d);
  hipFree(dc);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int OP>
void sweep(int nCU) {
  for (int w : {1, 2, 4, 8}) run<OP, 0>(w, nCU);
  for (int w : {1, 2, 4, 8}) run<OP, 1>(w, nCU);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int nCU = p.multiProcessorCount;
  printf("# %s  CUs=%d  clockRate=%d kHz  (cycles below assume 2.4 GHz; N_IT=%d x %d instr per wave)\n", p.name, nCU,
         p.clockRate, N_IT, N_ACC);
  sweep<ADD_U32>(nCU);
  sweep<FMA_F32>(nCU);
  sweep<PK_FMA_F32>(nCU);
  sweep<MIN3_I32>(nCU);
  sweep<PERM_B32>(nCU);
  sweep<BCNT>(nCU);
  sweep<XOR_B32>(nCU);
  sweep<PK_MIN_I16>(nCU);
  // round 2, second half: the integer multiply / dot / shift family the resize and blur arithmetic is built from
  // (full EXEC, 4 waves per SIMD only)
  run<MUL_HI_U32, 0>(4, nCU);
  run<MUL_LO_U32, 0>(4, nCU);
  run<MUL_U32_U24, 0>(4, nCU);
  run<MAD_U32_U24, 0>(4, nCU);
  run<DOT2_U32_U16, 0>(4, nCU);
  run<DOT4_U32_U8, 0>(4, nCU);
  run<ALIGNBYTE, 0>(4, nCU);
  run<SAD_U16, 0>(4, nCU);
  run<PK_MAD_U16, 0>(4, nCU);
  run<LSHRREV, 0>(4, nCU);
  return 0;
}
