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
// The power-of-two pre-scale of the packed split-fp16 planes (asr_pack_hl; gemm.hip's
// pow2_scale): 2^(9 - e) with max = f * 2^e, f in [0.5, 1), i.e. max * scale in [2^8, 2^9);
// 1 for a null pointer or a zero maximum.
__device__ __forceinline__ float asr_pow2_scale(const float* absmax) {
  if (absmax == nullptr) return 1.f;
  const unsigned b = __float_as_uint(*absmax);
  const int e = (int)((b >> 23) & 0xff) - 126;
  if ((b & 0x7fffffffu) == 0u) return 1.f;
  int k = 9 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return __uint_as_float((unsigned)(127 + k) << 23);
}
// The LSTM's `activation` hyper-parameter (core/layers.py:452, :463: g = act(z_c), h = o act(c);
// asr_lstm_args.activation / asr_lstm_ln_args.activation, Keras-1.2.2 names): 0 tanh, 1 relu,
// 2 sigmoid, 3 hard_sigmoid, 4 linear, 5 softsign, 6 softplus.  Only the VARIANT kernels
// (lstm_*_kernel_hv, lstm_ln.hip) read it; the default kernels are tanh.  asr_act_slope =
// act'(x) written in terms of y = act(x), which is all the BPTT kernels keep of the candidate.
__device__ __forceinline__ float asr_act_apply(int id, float x) {
  switch (id) {
    case 1: return fmaxf(x, 0.f);
    case 2: return __fdividef(1.f, 1.f + __expf(-fminf(fmaxf(x, -80.f), 80.f)));
    case 3: return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f);
    case 4: return x;
    case 5: return __fdividef(x, 1.f + fabsf(x));
    case 6: return x > 20.f ? x : log1pf(__expf(fminf(x, 20.f)));
    default: return tanhf(x);
  }
}
__device__ __forceinline__ float asr_act_slope(int id, float y) {
  switch (id) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y * (1.f - y);
    case 3: return (y > 0.f && y < 1.f) ? 0.2f : 0.f;
    case 4: return 1.f;
    case 5: { const float a = 1.f - fabsf(y); return a * a; }
    case 6: return 1.f - __expf(-y);
    default: return 1.f - y * y;
  }
}
#endif
