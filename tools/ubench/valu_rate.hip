// Micro-benchmark: issue rate of the VALU ops the FAST kernel is built from (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
#define N_IT 2048
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
  uint32_t b = seed ^ 0x12345;
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, b)));
      if (OP == 1) { int x = (int)a[i], y = (int)b; a[i] = (uint32_t)(x < y ? x : y); }
      if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c020c00u + i);
      if (OP == 3) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, a[i]) - __builtin_bit_cast(s2, b));
      if (OP == 4) { int x = (int)a[i], y = (int)b, z = (int)a[(i + 1) & 7]; int m = x < y ? x : y; a[i] = (uint32_t)(m < z ? m : z); }
      if (OP == 5) a[i] = a[i] * b;
      if (OP == 6) a[i] = (uint32_t)__mul24((int)a[i], (int)b);
      if (OP == 7) { unsigned long long t = (unsigned long long)a[i] * b + a[(i + 1) & 7]; a[i] = (uint32_t)(t >> 7); }
      if (OP == 8) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);
      if (OP == 9) a[i] = __builtin_amdgcn_alignbyte(a[i], b, a[i] & 3);
      asm volatile("" : "+v"(a[i]));
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char* name) {
  uint32_t* d;
  hipMalloc(&d, 4 * 256 * 2048);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 7u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 7u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double winst = 2048.0 * 4 * N_IT * 8;  // wave-instructions
  printf("%-14s %.3f ms  %.1f G wave-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms,
         winst / ms / 1e6, 1024 * 2.4e9 / (winst / (ms * 1e-3)));
  hipFree(d);
}
int main() {
  run<1>("v_min_i32");
  run<0>("v_pk_min_i16");
  run<2>("v_perm_b32");
  run<3>("v_pk_sub_i16");
  run<4>("v_min3_i32");
  run<5>("v_mul_lo_u32");
  run<6>("v_mul_i32_i24");
  run<7>("v_mad_u64_u32");
  run<8>("v_dot4_u32_u8");
  run<9>("v_alignbyte");
  return 0;
}
