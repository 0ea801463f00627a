// What ds_read_b64_tr_b16 returns: a [32 rows][16 cols] fp16 image with value = 100 row + col;
// lane l = 16 g + 4 r + p passes the address of row 4 g + r, columns 4 p .. 4 p + 3 (8 bytes).
// Expected (MI355X guide): lane 16 g + c receives column c of rows 4 g .. 4 g + 3.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tr_probe tools/micro/tr_read_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using h4 = __attribute__((ext_vector_type(4))) _Float16;
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 img[32 * 16];
  for (int i = threadIdx.x; i < 512; i += 64) img[i] = (_Float16)(100 * (i >> 4) + (i & 15));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, r = (l >> 2) & 3, p = l & 3;
  typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
  const s4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      reinterpret_cast<__attribute__((address_space(3))) s4*>(
          (__attribute__((address_space(3))) _Float16*)(img) + (4 * g + r) * 16 + 4 * p));
  const h4 v = __builtin_bit_cast(h4, raw);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      printf(" %5.0f", h[l * 4 + e]);
      if (h[l * 4 + e] != 100 * (4 * (l >> 4) + e) + (l & 15)) ++bad;
    }
    printf("\n");
  }
  printf("%s\n", bad ? "UNEXPECTED mapping" : "as expected: lane 16g+c <- rows 4g..4g+3 of column c");
  return 0;
}
