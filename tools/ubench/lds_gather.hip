// Micro-benchmark (gfx950): the ring gather of k_fast_cells phase B from a ~5 KB LDS tile (48-byte row pitch), one
// listed pixel per lane at raster-ordered positions of 13 % density -- how should a lane fetch its 16 ring bytes + centre?
//   A. 17 ds_read_u8 (the kernel's form)
//   B. 2 unaligned ds_read_b32 (rows -3 / +3: 3 bytes each) + 5 unaligned ds_read_b64 (rows -2 .. +2: x-3 .. x+4)
//   C. 2 unaligned ds_read_b32 + 10 ds_read_u8 + centre (rows +-3 as dwords, the rest as bytes)
// Every variant folds what it read into one word so that nothing is optimised away; the figure is time per pass of a
// wave, 8 waves per SIMD resident (LDS padded to the kernel's 5.4 KB per wave).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_gather tools/ubench/lds_gather.hip && tools/ubench/lds_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int P = 48, ROWS = 38, TILE = P * ROWS;  // bytes
constexpr int N_IT = 2048;
struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };
struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };

template <int V>
__global__ __launch_bounds__(64) void k(const uint16_t* __restrict__ pos, uint32_t* __restrict__ out, int nPos) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[5440];
  for (int i = threadIdx.x; i < 5440; i += 64) tile[i] = (uint8_t)(i * 37 + blockIdx.x);
  __syncthreads();
  uint32_t acc = 0;
  int q = threadIdx.x;
  for (int it = 0; it < N_IT; it++) {
    const int e = pos[q];  // py << 8 | px, the centre at tile byte (py + 3) * P + 4 + px
    q += 64;
    if (q >= nPos) q -= nPos;
    const uint8_t* c = tile + ((e >> 8) + 3) * P + 4 + (e & 255);
    if (V == 0) {
      constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
      constexpr int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
      uint32_t s = c[0];
#pragma unroll
      for (int k = 0; k < 16; k++) s += (uint32_t)c[dx[k] + dy[k] * P] << (k & 15);
      acc += s;
    } else if (V == 1) {
      const uint32_t a = reinterpret_cast<const U1u*>(c - 3 * P - 1)->x, b = reinterpret_cast<const U1u*>(c + 3 * P - 1)->x;
      uint32_t s = a ^ (b << 3);
#pragma unroll
      for (int r = -2; r <= 2; r++) {
        const U2u w = *reinterpret_cast<const U2u*>(c + r * P - 3);
        s += w.x + (w.y << (r + 3));
      }
      acc += s;
    } else {
      const uint32_t a = reinterpret_cast<const U1u*>(c - 3 * P - 1)->x, b = reinterpret_cast<const U1u*>(c + 3 * P - 1)->x;
      uint32_t s = a ^ (b << 3) ^ c[0];
      constexpr int dx[10] = {2, 3, 3, 3, 2, -2, -3, -3, -3, -2};
      constexpr int dy[10] = {2, 1, 0, -1, -2, -2, -1, 0, 1, 2};
#pragma unroll
      for (int k = 0; k < 10; k++) s += (uint32_t)c[dx[k] + dy[k] * P] << k;
      acc += s;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int V>
void run(const char* name, const uint16_t* d_pos, int nPos, int nCU) {
  const int blocks = nCU * 32;  // 8 waves per SIMD
  uint32_t* d;
  hipMalloc(&d, 4ull * 64 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(64), 0, 0, d_pos, d, nPos);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(64), 0, 0, d_pos, d, nPos);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per CU: 32 waves x N_IT passes in ms -> cycles per wave-pass per CU at 2.4 GHz
  printf("%-58s %8.3f ms  = %.1f CU-cycles per wave-pass (32 waves per CU resident)\n", name, ms, ms * 1e-3 * 2.4e9 / (32.0 * N_IT));
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int nCU = p.multiProcessorCount;
  // listed pixels of a 32 x 32 cell in raster order, 13 % density
  std::vector<uint16_t> pos;
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int y = 0; y < 32; y++)
    for (int x = 0; x < 32; x++) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      if ((st >> 33) % 100 < 13) pos.push_back((uint16_t)((y << 8) | x));
    }
  uint16_t* d_pos;
  hipMalloc(&d_pos, pos.size() * 2);
  hipMemcpy(d_pos, pos.data(), pos.size() * 2, hipMemcpyHostToDevice);
  printf("# %s  CUs=%d  %zu listed pixels per cell, %d passes per wave\n", p.name, nCU, pos.size(), N_IT);
  run<0>("A: 17 x ds_read_u8", d_pos, (int)pos.size(), nCU);
  run<1>("B: 2 x ds_read_b32 + 5 x ds_read_b64, unaligned", d_pos, (int)pos.size(), nCU);
  run<2>("C: 2 x ds_read_b32 unaligned + 11 x ds_read_u8", d_pos, (int)pos.size(), nCU);
  return 0;
}
