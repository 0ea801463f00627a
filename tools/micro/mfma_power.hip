// What the matrix pipes sustain at the package power cap: pure MFMA loops on every SIMD (2 waves
// per SIMD, 8 / 32 independent accumulators, pseudo-random fp16 operands, no memory traffic),
// ~1.5 s per variant.  Prints executed TFLOP/s and the shader clock seen in-kernel (s_memtime
// against the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/micro/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ h8 rnd8(unsigned s) {
  h8 r;
  for (int e = 0; e < 8; ++e) {
    s = s * 1664525u + 1013904223u;
    r[e] = (_Float16)(((int)(s >> 9) % 2048 - 1024) * (1.0f / 8192.0f));
  }
  return r;
}

// MODE 0: v_mfma_f32_32x32x16_f16, 8 accumulators (the GEMM's wave tile)
// MODE 1: v_mfma_f32_16x16x32_f16, 32 accumulators (same output tile)
// ZERO: all-zero operands (how much of the power is data toggling)
template <int MODE, bool ZERO>
__global__ void __launch_bounds__(512) k(long long* out, int iters, unsigned seed) {
  const unsigned t = threadIdx.x + blockIdx.x * 512u + seed;
  h8 a[4], an[4], b[2];
  for (int i = 0; i < 4; ++i) {
    a[i] = ZERO ? h8{0, 0, 0, 0, 0, 0, 0, 0} : rnd8(t * 7u + i);
    an[i] = -a[i];
  }
  for (int j = 0; j < 2; ++j) b[j] = ZERO ? h8{0, 0, 0, 0, 0, 0, 0, 0} : rnd8(t * 13u + 100 + j);
  const long long c0 = (long long)__builtin_readcyclecounter();
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (MODE == 0) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((it & 1) ? an[i] : a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      // same flops per iteration: 3 x 32 MFMAs of 16x16x32 (K 32) = 2 x [3 x 8 of 32x32x16]
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((it & 1) ? an[i & 3] : a[i & 3], b[j & 1], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  }
  const long long c1 = (long long)__builtin_readcyclecounter();
  const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  if (s == 123.456f) out[2] = 1;
}

template <int MODE, bool ZERO>
void run(const char* name) {
  long long* d;
  hipMalloc(&d, 32);
  // flops per iteration and wave: MODE 0: 24 x 32*32*16*2; MODE 1: 96 x 16*16*32*2
  const double fl_iter = MODE == 0 ? 24.0 * 32768.0 : 96.0 * 16384.0;
  const int iters = MODE == 0 ? 40000 : 20000;
  double best = 0, ghz = 0;
  auto t_begin = std::chrono::steady_clock::now();
  int launches = 0;
  double last_ms = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < 1.5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ZERO>), dim3(256), dim3(512), 0, 0, d, iters, 17u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    last_ms = ms; ++launches;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  best = fl_iter * iters * 8 * 256 / (last_ms * 1e-3) / 1e12;
  ghz = (double)h[0] / ((double)h[1] * 10.0);
  printf("%-52s %7.1f TF/s executed (last of %d launches, %.2f ms), shader clock %.2f GHz, %.1f clocks per %s\n",
         name, best, launches, last_ms, ghz, (double)h[0] / iters / (MODE == 0 ? 24.0 : 96.0) / 2.0,
         MODE == 0 ? "32x32x16 (2 waves)" : "16x16x32 (2 waves)");
  hipFree(d);
}

int main() {
  run<0, false>("32x32x16 f16, 8 accumulators, random operands");
  run<1, false>("16x16x32 f16, 32 accumulators, random operands");
  run<0, true>("32x32x16 f16, 8 accumulators, zero operands");
  run<1, true>("16x16x32 f16, 32 accumulators, zero operands");
  return 0;
}
