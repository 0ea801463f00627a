// K12 counter-based random numbers for the training-time noise of the path -- gfx950.
//
// Replaces the TF-1.3 / Keras-1.2.2 random ops behind core/models.py:250-251 (GaussianNoise),
// :257-258 (input Dropout), :265-266 (variational dropout masks B_W / B_U of every LSTM,
// drawn once per batch) and core/layers_utils.py:34-42 (zoneout keep masks).  TF's own
// stream cannot be replayed, so parity is defined on OUR stream: Philox-4x32-10 (Salmon et
// al. 2011, the generator TF itself uses), key = the 64-bit seed, counter = (block index,
// stream id, step, 0); element 4*b + j of a tensor is word j of block b.  The generator is
// stateless: the same (seed, stream, step) reproduces the same tensor on any launch geometry,
// and oracle/rng.py restates it in NumPy (bit-exact masks; noise to fp32 round-off).
//   uniform u = (word >> 8) * 2^-24 in [0, 1)
//   keep mask = u >= p ? scale : 0            (scale = 1/(1-p): inverted dropout; 1: zoneout)
//   normal    = sqrt(-2 ln u1) * cos / sin (2 pi u2), u1 = (w0 + 1) 2^-32, u2 = w1 2^-32
// HBM-bound: one 16-byte store per thread and block of four values.
#include "common.h"

namespace {

struct Philox { unsigned c[4]; };

__device__ __forceinline__ Philox philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox p;
  p.c[0] = c0; p.c[1] = c1; p.c[2] = c2; p.c[3] = c3;
  return p;
}

// mode 0: keep mask; 1: out = in * keep mask (mask also stored if mask_out); 2: out = in +
// sigma * normal (in may be NULL: pure noise); 3: raw 32-bit words (tests)
__global__ void __launch_bounds__(256)
random_kernel(int mode, float* __restrict__ out, const float* __restrict__ in,
              float* __restrict__ mask_out, long long n, float p, float scale, unsigned k0,
              unsigned k1, unsigned stream_id, unsigned step) {
  const long long nb = (n + 3) / 4;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nb;
       b += (long long)gridDim.x * blockDim.x) {
    const Philox r = philox4x32_10((unsigned)b, stream_id, step, (unsigned)(b >> 32), k0, k1);
    float v[4];
    if (mode == 2) {
      // two Box-Muller pairs per block
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)r.c[2 * h] + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
        const float u2 = (float)r.c[2 * h + 1] * 2.3283064365386963e-10f;            // [0, 1]
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        v[2 * h] = rad * cs;
        v[2 * h + 1] = rad * sn;
      }
    } else if (mode == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(r.c[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (float)(r.c[j] >> 8) * 5.9604644775390625e-08f;               // 2^-24
        v[j] = u >= p ? scale : 0.f;
      }
    }
    const long long e0 = 4 * b;
    if (e0 + 3 < n) {
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (mode == 1 || (mode == 2 && in)) {
        const float4 x = *reinterpret_cast<const float4*>(in + e0);
        if (mode == 1) {
          if (mask_out) *reinterpret_cast<float4*>(mask_out + e0) = o;
          o = make_float4(x.x * o.x, x.y * o.y, x.z * o.z, x.w * o.w);
        } else {
          o = make_float4(x.x + p * o.x, x.y + p * o.y, x.z + p * o.z, x.w + p * o.w);
        }
      } else if (mode == 2) {
        o = make_float4(p * o.x, p * o.y, p * o.z, p * o.w);
      }
      *reinterpret_cast<float4*>(out + e0) = o;
    } else {
      for (int j = 0; j < 4 && e0 + j < n; ++j) {
        float o = v[j];
        if (mode == 1) {
          if (mask_out) mask_out[e0 + j] = o;
          o *= in[e0 + j];
        } else if (mode == 2) {
          o = (in ? in[e0 + j] : 0.f) + p * o;
        }
        out[e0 + j] = o;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
mul_kernel(long long n, const float* __restrict__ x, const float* __restrict__ y,
           float* __restrict__ out) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = x[i] * y[i];
}

int launch(int mode, float* out, const float* in, float* mask_out, int64_t n, float p, float scale,
           uint64_t seed, uint32_t stream_id, uint32_t step, hipStream_t stream) {
  ASR_CHECK_ARG(out && n > 0, "random: bad arguments");
  ASR_CHECK_ARG(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in) |
                  reinterpret_cast<uintptr_t>(mask_out)) & 15) == 0, "random: 16-byte alignment");
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(random_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, mode, out, in,
                     mask_out, (long long)n, p, scale, (unsigned)(seed & 0xffffffffu),
                     (unsigned)(seed >> 32), stream_id, step);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

}  // namespace

extern "C" int asr_dropout_masks(float* out, int64_t n, float p, float scale, uint64_t seed,
                                 uint32_t stream_id, uint32_t step, asr_stream_t stream) {
  ASR_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_masks: p must be in [0, 1)");
  return launch(0, out, nullptr, nullptr, n, p, scale, seed, stream_id, step, (hipStream_t)stream);
}

extern "C" int asr_dropout_apply(const float* in, float* out, float* mask_out, int64_t n, float p,
                                 float scale, uint64_t seed, uint32_t stream_id, uint32_t step,
                                 asr_stream_t stream) {
  ASR_CHECK_ARG(in && p >= 0.f && p < 1.f, "dropout_apply: bad arguments");
  return launch(1, out, in, mask_out, n, p, scale, seed, stream_id, step, (hipStream_t)stream);
}

extern "C" int asr_gaussian_noise(const float* in, float* out, int64_t n, float sigma,
                                  uint64_t seed, uint32_t stream_id, uint32_t step,
                                  asr_stream_t stream) {
  return launch(2, out, in, nullptr, n, sigma, 0.f, seed, stream_id, step, (hipStream_t)stream);
}

extern "C" int asr_random_words(unsigned* out, int64_t n, uint64_t seed, uint32_t stream_id,
                                uint32_t step, asr_stream_t stream) {
  return launch(3, reinterpret_cast<float*>(out), nullptr, nullptr, n, 0.f, 0.f, seed, stream_id,
                step, (hipStream_t)stream);
}

extern "C" int asr_mul(int64_t n, const float* x, const float* y, float* out,
                       asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(x && y && out && n > 0, "mul: bad arguments");
  ASR_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, "mul: 16-byte alignment");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (long long)n, x, y,
                     out);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}
