// Checks the DPP wave64 inclusive scan used by k_fast.hip against a serial sum (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return x;
}
__global__ void k(const int* in, int* out, int* tot) {
  const int v = in[threadIdx.x];
  const int s = wave_incl_scan(v);
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *tot = __builtin_amdgcn_readlane(s, 63);
}
int main() {
  int h[64], o[64], *di, *d_o, *dt, t;
  hipMalloc(&di, 256); hipMalloc(&d_o, 256); hipMalloc(&dt, 4);
  int bad = 0;
  for (int trial = 0; trial < 50; trial++) {
    for (int i = 0; i < 64; i++) h[i] = rand() % 9;
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, d_o, dt);
    hipMemcpy(o, d_o, 256, hipMemcpyDeviceToHost);
    hipMemcpy(&t, dt, 4, hipMemcpyDeviceToHost);
    int run = 0;
    for (int i = 0; i < 64; i++) { run += h[i]; if (o[i] != run) bad++; }
    if (t != run) bad++;
  }
  printf("dpp_scan mismatches: %d\n", bad);
  return bad != 0;
}
