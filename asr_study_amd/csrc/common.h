// Shared helpers for the gfx950 kernels.  CDNA4 only: 64-wide wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "../../include/asr_hip.h"

#define ASR_WAVE 64

void asr_set_error(const char* fmt, ...);

#define ASR_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      asr_set_error(__VA_ARGS__);                \
      return ASR_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define ASR_CHECK_HIP(expr)                                                    \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) {                                                   \
      asr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                    __FILE__, __LINE__);                                       \
      return ASR_ERR_LAUNCH;                                                   \
    }                                                                          \
  } while (0)

#define ASR_CHECK_LAUNCH() ASR_CHECK_HIP(hipGetLastError())

static inline size_t asr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#ifdef __HIPCC__
__device__ __forceinline__ float asr_neg_inf() { return -__builtin_huge_valf(); }

// log(exp(a)+exp(b)), safe for -inf operands.
__device__ __forceinline__ float asr_lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == asr_neg_inf()) return m;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float asr_lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == asr_neg_inf()) return m;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__device__ __forceinline__ float asr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double asr_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float asr_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
