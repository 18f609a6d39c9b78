// Micro-benchmark (gfx950): do byte-unaligned 16-byte global loads and 4-byte global stores work, and what do
// they cost?  (KITTI's 1241-byte image stride makes every level-0 row start at an odd address.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/unaligned tools/ubench/unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };

// every lane copies 16 bytes from src + off + 16*i to dst + 16*i (dst aligned)
__global__ __launch_bounds__(256) void k_load(const uint8_t* __restrict__ src, uint4* __restrict__ dst, size_t n16, int off) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const U4u q = *reinterpret_cast<const U4u*>(src + off + 16 * i);
    dst[i] = make_uint4(q.x, q.y, q.z, q.w);
  }
}
// every lane stores 4 bytes to dst + off + 4*i
__global__ __launch_bounds__(256) void k_store(const uint32_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n4, int off) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    U1u v = {src[i]};
    *reinterpret_cast<U1u*>(dst + off + 4 * i) = v;
  }
}

int main() {
  const size_t N = 512ull << 20;
  uint8_t *a, *b;
  hipMalloc(&a, N + 64);
  hipMalloc(&b, N + 64);
  std::vector<uint8_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131 + 7);
  for (size_t o = 0; o < N; o += h.size()) hipMemcpy(a + o, h.data(), h.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int off : {0, 1, 2, 3, 4, 8, 13}) {
    const size_t n16 = (N - 64) / 16;
    hipLaunchKernelGGL(k_load, dim3(256 * 16), dim3(256), 0, 0, a, (uint4*)b, n16, off);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_load, dim3(256 * 16), dim3(256), 0, 0, a, (uint4*)b, n16, off);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint8_t> chk(4096);
    hipMemcpy(chk.data(), b + (1 << 20), 4096, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 0; i < 4096; i++) ok &= chk[i] == h[((1 << 20) + i + off) % h.size()];
    printf("load16  off=%2d  %7.3f ms  %7.1f GB/s (read+write)  %s\n", off, ms, 2.0 * n16 * 16 / ms / 1e6, ok ? "correct" : "WRONG");
  }
  for (int off : {0, 1, 2, 3}) {
    const size_t n4 = (N - 64) / 4;
    hipMemset(b, 0, N);
    hipLaunchKernelGGL(k_store, dim3(256 * 16), dim3(256), 0, 0, (const uint32_t*)a, b, n4, off);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_store, dim3(256 * 16), dim3(256), 0, 0, (const uint32_t*)a, b, n4, off);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint8_t> chk(4096);
    hipMemcpy(chk.data(), b + (1 << 20) + off, 4096, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 0; i < 4096; i++) ok &= chk[i] == h[((1 << 20) + i) % h.size()];
    printf("store4  off=%2d  %7.3f ms  %7.1f GB/s (read+write)  %s\n", off, ms, 2.0 * n4 * 4 / ms / 1e6, ok ? "correct" : "WRONG");
  }
  return 0;
}
