// K11 optimiser: global-norm clip + Adam / SGD-momentum over one flat fp32 buffer.
//
// Replaces Keras 1.2.2 Adam/SGD(clipnorm=...) (train.py:133-137) and the l2
// weight regularisers (core/models.py:263-264,279: their gradient 2*l2*w is part
// of the clipped gradient, their value l2*sum(w^2) part of the reported loss).
// HBM-bound: 28 B/param for Adam (read g,p,m,v; write p,m,v), 16-byte accesses,
// one pass for the norm (float64 accumulation, fixed-order two-level reduce ->
// deterministic) and one for the update; the clip scale is read from device
// memory so there is no host round trip between the two.
#include "common.h"

namespace {

constexpr int kNormBlocks = 1024;

__device__ __forceinline__ float seg_l2(const asr_segment* seg, int n_seg, int64_t idx) {
  // segments are sorted by offset and cover the buffer; binary search
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid].offset <= idx) lo = mid; else hi = mid - 1;
  }
  return seg[lo].l2;
}

// A workgroup sums one contiguous chunk (16-byte loads, two groups in flight per thread); the
// segment of an element is found by walking forward from the chunk's first one -- the
// per-element binary search with 4-byte loads ran at 1 TB/s, a quarter of the Adam pass that
// moves 3.5x the bytes.
__global__ void __launch_bounds__(256)
norm_partial_kernel(const float* __restrict__ p, const float* __restrict__ g, int64_t n,
                    const asr_segment* __restrict__ seg, int n_seg,
                    double* __restrict__ partial, int vec) {
  double s_g = 0.0, s_w = 0.0;
  const int64_t chunk = (((n + gridDim.x - 1) / gridDim.x) + 2047) / 2048 * 2048;
  const int64_t b0 = (int64_t)blockIdx.x * chunk;
  const int64_t b1 = b0 + chunk < n ? b0 + chunk : n;
  int cur = 0;
  {                                             // segment of the thread's first element
    int lo = 0, hi = n_seg - 1;
    const int64_t first = b0 + 4 * threadIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg[mid].offset <= first) lo = mid; else hi = mid - 1;
    }
    cur = lo;
  }
  float l2 = seg[cur].l2;
  int64_t next = cur + 1 < n_seg ? seg[cur + 1].offset : INT64_MAX;
  auto add = [&](int64_t idx, float w, float gr) {
    while (idx >= next) {
      ++cur;
      l2 = seg[cur].l2;
      next = cur + 1 < n_seg ? seg[cur + 1].offset : INT64_MAX;
    }
    const float gg = gr + 2.f * l2 * w;
    s_g += (double)gg * gg;
    s_w += (double)l2 * w * w;
  };
  for (int64_t i = b0 + 4 * threadIdx.x; i < b1; i += 2048) {
    const int64_t j = i + 1024;
    if (vec && j + 3 < b1) {                    // two whole groups, 16-byte aligned
      const float4 w0 = *reinterpret_cast<const float4*>(p + i);
      const float4 g0 = *reinterpret_cast<const float4*>(g + i);
      const float4 w1 = *reinterpret_cast<const float4*>(p + j);
      const float4 g1 = *reinterpret_cast<const float4*>(g + j);
      add(i, w0.x, g0.x); add(i + 1, w0.y, g0.y); add(i + 2, w0.z, g0.z); add(i + 3, w0.w, g0.w);
      add(j, w1.x, g1.x); add(j + 1, w1.y, g1.y); add(j + 2, w1.z, g1.z); add(j + 3, w1.w, g1.w);
    } else {
      for (int64_t k = i; k < i + 4 && k < b1; ++k) add(k, p[k], g[k]);
      for (int64_t k = j; k < j + 4 && k < b1; ++k) add(k, p[k], g[k]);
    }
  }
  s_g = asr_wave_sum_d(s_g);
  s_w = asr_wave_sum_d(s_w);
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s_g; sh[1][w] = s_w; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    partial[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__global__ void __launch_bounds__(256)
norm_final_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  double s_g = 0.0, s_w = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    s_g += partial[2 * i];
    s_w += partial[2 * i + 1];
  }
  s_g = asr_wave_sum_d(s_g);
  s_w = asr_wave_sum_d(s_w);
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s_g; sh[1][w] = s_w; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = sqrt(sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
    out[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__device__ __forceinline__ float clip_scale(const double* norm, float clipnorm) {
  if (!(clipnorm > 0.f) || norm == nullptr) return 1.f;
  const double nrm = norm[0];
  return nrm >= (double)clipnorm ? (float)((double)clipnorm / nrm) : 1.f;
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, const asr_segment* __restrict__ seg, int n_seg,
            const double* __restrict__ norm, float clipnorm, float lr_t, float b1, float b2,
            float eps) {
  if (norm != nullptr && norm[0] < 0.0) return;      // step vetoed (asr_optim_guard): no update
  const float sc = clip_scale(norm, clipnorm);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float l2 = seg_l2(seg, n_seg, i);
    const float w = p[i];
    const float gg = (g[i] + 2.f * l2 * w) * sc;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = w - lr_t * mm / (sqrtf(vv) + eps);
  }
}

__global__ void __launch_bounds__(256)
sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ vel,
           int64_t n, const asr_segment* __restrict__ seg, int n_seg,
           const double* __restrict__ norm, float clipnorm, float lr, float mu) {
  if (norm != nullptr && norm[0] < 0.0) return;      // step vetoed (asr_optim_guard): no update
  const float sc = clip_scale(norm, clipnorm);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float l2 = seg_l2(seg, n_seg, i);
    const float w = p[i];
    const float gg = (g[i] + 2.f * l2 * w) * sc;
    const float vn = mu * vel[i] - lr * gg;
    vel[i] = vn;
    p[i] = w + vn;
  }
}

// Vetoes the optimiser step that follows when a device-side fault flag is set: the norm (never
// negative otherwise) becomes -1 and the update kernels return without touching anything.
__global__ void optim_guard_kernel(double* __restrict__ norm, const int* __restrict__ a,
                                   const int* __restrict__ b) {
  if ((a && *a != 0) || (b && *b != 0)) norm[0] = -1.0;
}

// flags of this process' recurrent workspaces as 0.0 / 1.0 floats: the two slots behind the
// flat gradient buffer, so that the gradient all-reduce (a sum) makes a timeout of ANY rank
// visible to EVERY rank's guard
__global__ void timeout_flags_kernel(const int* __restrict__ a, const int* __restrict__ b,
                                     float* __restrict__ out2) {
  out2[0] = (a && *a != 0) ? 1.f : 0.f;
  out2[1] = (b && *b != 0) ? 1.f : 0.f;
}

// Holds `blocks` workgroups (256 threads, lds_bytes of LDS each) on the device for `ticks`
// of the 100 MHz wall clock: a stand-in for a foreign kernel (an RCCL collective) that
// takes compute units away from the persistent recurrent kernels -- tests only.
__global__ void __launch_bounds__(256)
occupy_kernel(long long ticks, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {
    acc += (unsigned)threadIdx.x;
    __builtin_amdgcn_s_sleep(8);
  }
  if (sink && acc == 0xFFFFFFFFu) { lds[threadIdx.x] = 1.f; *sink = acc + (unsigned)lds[0]; }
}

int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t asr_optim_workspace_bytes(int64_t n) {
  (void)n;
  return (size_t)kNormBlocks * 2 * sizeof(double);
}

extern "C" int asr_grad_norm(const float* params, const float* grads, int64_t n,
                             const asr_segment* segments_dev, int n_seg, double* norm_out,
                             void* workspace, size_t ws_bytes, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(params && grads && segments_dev && norm_out && workspace && n > 0 && n_seg > 0,
                "grad_norm: bad arguments");
  if (ws_bytes < asr_optim_workspace_bytes(n)) {
    asr_set_error("grad_norm: workspace too small");
    return ASR_ERR_WORKSPACE;
  }
  int blocks = grid_for(n);
  if (blocks > kNormBlocks) blocks = kNormBlocks;
  double* partial = reinterpret_cast<double*>(workspace);
  const int vec = ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads)) & 15) == 0;
  hipLaunchKernelGGL(norm_partial_kernel, dim3(blocks), dim3(256), 0, stream, params, grads, n,
                     segments_dev, n_seg, partial, vec);
  ASR_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, stream, partial, blocks,
                     norm_out);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" int asr_optim_guard(double* norm_dev, const int* flag_a, const int* flag_b,
                               asr_stream_t stream_) {
  ASR_CHECK_ARG(norm_dev, "optim_guard: null norm");
  hipLaunchKernelGGL(optim_guard_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, norm_dev,
                     flag_a, flag_b);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" int asr_timeout_flags(const int* flag_a, const int* flag_b, float* out2,
                                 asr_stream_t stream_) {
  ASR_CHECK_ARG(out2, "timeout_flags: null output");
  hipLaunchKernelGGL(timeout_flags_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, flag_a,
                     flag_b, out2);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" int asr_debug_occupy(int blocks, int lds_bytes, double seconds, asr_stream_t stream_) {
  ASR_CHECK_ARG(blocks > 0 && blocks <= 4096 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 &&
                seconds >= 0.0 && seconds <= 5.0, "debug_occupy: blocks in [1, 4096], LDS <= 160 KB, "
                "at most 5 s");
  if (lds_bytes > 64 * 1024)
    ASR_CHECK_HIP(hipFuncSetAttribute((const void*)occupy_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes,
                     (hipStream_t)stream_, (long long)(seconds * 1e8), (unsigned*)nullptr);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" int asr_adam_step(float* params, const float* grads, float* m, float* v, int64_t n,
                             const asr_segment* segments_dev, int n_seg,
                             const double* norm_dev, float clipnorm, float lr, float beta1,
                             float beta2, float eps, int step, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(params && grads && m && v && segments_dev && n > 0 && n_seg > 0 && step >= 1,
                "adam: bad arguments");
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) /
                      (1.0 - pow((double)beta1, step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, stream, params, grads, m, v,
                     n, segments_dev, n_seg, norm_dev, clipnorm, (float)lr_t, beta1, beta2, eps);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" int asr_sgd_step(float* params, const float* grads, float* vel, int64_t n,
                            const asr_segment* segments_dev, int n_seg,
                            const double* norm_dev, float clipnorm, float lr, float momentum,
                            asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(params && grads && vel && segments_dev && n > 0 && n_seg > 0,
                "sgd: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, params, grads, vel, n,
                     segments_dev, n_seg, norm_dev, clipnorm, lr, momentum);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

namespace {
__global__ void __launch_bounds__(256)
axpby_kernel(int64_t n, float a, const float* __restrict__ x, float b,
             const float* __restrict__ y, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 p = reinterpret_cast<const float4*>(x)[i];
    const float4 q = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(out)[i] =
        make_float4(a * p.x + b * q.x, a * p.y + b * q.y, a * p.z + b * q.z, a * p.w + b * q.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a * x[i] + b * y[i];
}
}  // namespace

extern "C" int asr_axpby(int64_t n, float a, const float* x, float b, const float* y, float* out,
                         asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(x && y && out && n > 0, "axpby: bad arguments");
  ASR_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                  reinterpret_cast<uintptr_t>(out)) & 15) == 0, "axpby: 16-byte alignment");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, a, x, b, y, out);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}
