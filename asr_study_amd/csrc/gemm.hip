// K4/K6 fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) + column sum -- gfx950.
//
// Replaces the hoisted input->4H gate matmul K.dot(x*B_W, W) (core/layers.py:439),
// TimeDistributed(Dense) (core/models.py:278-279) and their gradients.  fp32 in /
// fp32 accumulate MFMA is bit-identical to an fmaf chain, so the 1e-4 parity
// budget is untouched (there is no xf32/TF32 path on gfx950).
//
// Tiling: 128x128x16 block tile, 256 threads = 4 waves in a 2x2 grid, each wave
// owns a 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  Both
// operands are staged K-MAJOR in LDS (As[k][m], Bs[k][n], row pad 4) so that the
// MFMA fragment reads (lane l -> [k = 2kk + l/32][l%32]) are conflict-free
// ds_read_b32; whichever of the four storage orders the caller has is transposed
// on the way in (16-byte global loads; ds_write_b128 for mn-contiguous sources,
// 2-way (free) ds_write_b32 for k-contiguous ones).  Global->register prefetch of
// tile k+1 overlaps the 32 MFMAs of tile k; one barrier per K step.
// Variational-dropout masks (core/models.py:265-266) are applied in the loader
// (A side) or the epilogue (C side), bias in the epilogue.  Split-K writes
// per-split partials and reduces them in a fixed order (deterministic).
#include "common.h"
#include <atomic>
#include <mutex>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 128, BN = 128, PAD = 4;
constexpr int LDS_LD = BM + PAD;   // BM == BN

struct TileSrc {
  const float* p;      // base pointer
  int ld;              // leading dimension of the storage
  int mn_contig;       // 1: storage is (K, MN) row-major; 0: (MN, K) row-major
  int mn_total, k_total;
  const float* scale;  // optional mask over storage (row % period, col)
  int period, scale_ld;
  int vec_ok;          // 16-byte aligned rows
  int scale_vec;       // mask rows 16-byte aligned as well
};

// row % period; the period is n_pad (a multiple of 16, in practice a power of two)
__device__ __forceinline__ int mod_period(int row, int period) {
  return (period & (period - 1)) == 0 ? (row & (period - 1)) : row % period;
}

// Loads the (BK x 128) tile starting at (k0, mn0) into BK/8 float4 registers.
template <int BK>
__device__ __forceinline__ void tile_load(const TileSrc& s, int k0, int mn0, int k_end,
                                          float4 (&r)[BK / 8]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < BK / 8; ++it) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (s.mn_contig) {
      const int kk = (tid >> 5) + 8 * it;
      const int mn = mn0 + 4 * (tid & 31);
      const int k = k0 + kk;
      if (k < k_end) {
        const float* src = s.p + (size_t)k * s.ld + mn;
        if (s.vec_ok && mn + 3 < s.mn_total) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (mn + e < s.mn_total) v[e] = src[e];
        }
        if (s.scale) {
          const float* sc = s.scale + (size_t)mod_period(k, s.period) * s.scale_ld + mn;
          if (s.scale_vec && mn + 3 < s.mn_total) {
            const float4 q = *reinterpret_cast<const float4*>(sc);
            v[0] *= q.x; v[1] *= q.y; v[2] *= q.z; v[3] *= q.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (mn + e < s.mn_total) v[e] *= sc[e];
          }
        }
      }
    } else {
      // BK/4 float4 per row; thread -> (row, k-quad)
      constexpr int QPR = BK / 4;                 // k-quads per row
      const int e = tid + 256 * it;
      const int mn = mn0 + e / QPR;
      const int k = k0 + 4 * (e % QPR);
      if (mn < s.mn_total) {
        const float* src = s.p + (size_t)mn * s.ld + k;
        if (s.vec_ok && k + 3 < k_end) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < k_end) v[e] = src[e];
        }
        if (s.scale) {
          const float* sc = s.scale + (size_t)mod_period(mn, s.period) * s.scale_ld + k;
          if (s.scale_vec && k + 3 < k_end) {
            const float4 q = *reinterpret_cast<const float4*>(sc);
            v[0] *= q.x; v[1] *= q.y; v[2] *= q.z; v[3] *= q.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < k_end) v[e] *= sc[e];
          }
        }
      }
    }
    r[it] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int BK>
__device__ __forceinline__ void tile_store(int mn_contig, const float4 (&r)[BK / 8],
                                           float (*S)[LDS_LD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < BK / 8; ++it) {
    if (mn_contig) {
      const int kk = (tid >> 5) + 8 * it;
      *reinterpret_cast<float4*>(&S[kk][4 * (tid & 31)]) = r[it];
    } else {
      constexpr int QPR = BK / 4;
      const int e = tid + 256 * it;
      const int mn = e / QPR;
      const int kb = 4 * (e % QPR);
      S[kb + 0][mn] = r[it].x;
      S[kb + 1][mn] = r[it].y;
      S[kb + 2][mn] = r[it].z;
      S[kb + 3][mn] = r[it].w;
    }
  }
}

struct Epilogue {
  float* C; int ldc;
  float alpha, beta;
  const float* bias;
  const float* c_scale; int c_period, c_ld;
  float* partial;      // split-K: raw accumulators go here ([split][M][N])
  float clamp_hi;      // > 0 (segmented gemm_hlx only): C = min(max(., 0), clamp_hi)
};

template <int BK>
__global__ void __launch_bounds__(256)
gemm_f32_mfma_kernel(TileSrc A, TileSrc B, int M, int N, int K, int k_per_split,
                     Epilogue ep) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > K) k_end = K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[BK / 8], rb[BK / 8];
  const int nk = (k_end - k_begin + BK - 1) / BK;
  if (nk > 0) {
    tile_load<BK>(A, k_begin, m0, k_end, ra);
    tile_load<BK>(B, k_begin, n0, k_end, rb);
    tile_store<BK>(A.mn_contig, ra, As[0]);
    tile_store<BK>(B.mn_contig, rb, Bs[0]);
  }
  __syncthreads();
  const int lrow = lane >> 5, lcol = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      tile_load<BK>(A, k_begin + (kt + 1) * BK, m0, k_end, ra);
      tile_load<BK>(B, k_begin + (kt + 1) * BK, n0, k_end, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[cur][2 * kk + lrow][wm * 64 + i * 32 + lcol];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[cur][2 * kk + lrow][wn * 64 + j * 32 + lcol];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      tile_store<BK>(A.mn_contig, ra, As[cur ^ 1]);
      tile_store<BK>(B.mn_contig, rb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31,
  //      row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lcol;
      if (col >= N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lrow;
        if (row >= M) continue;
        float v = acc[i][j][e];
        if (ep.partial) {
          ep.partial[((size_t)blockIdx.z * M + row) * N + col] = v;
        } else {
          v *= ep.alpha;
          float* dst = ep.C + (size_t)row * ep.ldc + col;
          if (ep.bias) v += ep.bias[col];
          if (ep.c_scale) v *= ep.c_scale[(size_t)mod_period(row, ep.c_period) * ep.c_ld + col];
          if (ep.beta != 0.f) v += ep.beta * *dst;
          *dst = v;
        }
      }
    }
}

__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N,
                          Epilogue ep) {
  const size_t total = (size_t)M * N;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / N), col = (int)(e % N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += partial[(size_t)s * total + e];
    v *= ep.alpha;
    float* dst = ep.C + (size_t)row * ep.ldc + col;
    if (ep.bias) v += ep.bias[col];
    if (ep.c_scale) v *= ep.c_scale[(size_t)mod_period(row, ep.c_period) * ep.c_ld + col];
    if (ep.beta != 0.f) v += ep.beta * *dst;
    *dst = v;
  }
}

// The same on whole 16-byte quads (N, ldc multiples of 4, 16-byte aligned arrays: every weight
// gradient of the step).  The scalar form keeps 4 bytes per load in flight and sat at 2.2 TB/s.
__global__ void __launch_bounds__(256)
gemm_splitk_reduce4_kernel(const float* __restrict__ partial, int splits, int M, int N,
                           Epilogue ep) {
  const size_t total4 = (size_t)M * N / 4;
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4;
       q += (size_t)gridDim.x * blockDim.x) {
    const size_t e = q * 4;
    const int row = (int)(e / N), col = (int)(e % N);
    float4 v = p4[q];
    for (int s = 1; s < splits; ++s) {            // (fixed order: deterministic)
      const float4 t = p4[(size_t)s * total4 + q];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x *= ep.alpha; v.y *= ep.alpha; v.z *= ep.alpha; v.w *= ep.alpha;
    if (ep.bias) {
      v.x += ep.bias[col]; v.y += ep.bias[col + 1]; v.z += ep.bias[col + 2]; v.w += ep.bias[col + 3];
    }
    if (ep.c_scale) {
      const float* m = ep.c_scale + (size_t)mod_period(row, ep.c_period) * ep.c_ld + col;
      v.x *= m[0]; v.y *= m[1]; v.z *= m[2]; v.w *= m[3];
    }
    float4* dst = reinterpret_cast<float4*>(ep.C + (size_t)row * ep.ldc + col);
    if (ep.beta != 0.f) {
      const float4 o = *dst;
      v.x += ep.beta * o.x; v.y += ep.beta * o.y; v.z += ep.beta * o.z; v.w += ep.beta * o.w;
    }
    *dst = v;
  }
}

// picks the quad form where the layout allows it
static void launch_splitk_reduce(const float* partial, int splits, int M, int N, const Epilogue& ep,
                                 unsigned shm, hipStream_t stream) {
  const size_t total = (size_t)M * N;
  const bool quads = (N & 3) == 0 && (ep.ldc & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(ep.C) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
  const size_t items = quads ? total / 4 : total;
  int blocks = (int)((items + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (quads)
    hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, dim3(blocks), dim3(256), shm, stream, partial,
                       splits, M, N, ep);
  else
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), shm, stream, partial,
                       splits, M, N, ep);
}

// ===========================================================================
// Split-fp16 GEMM: every fp32 operand element x is staged in LDS as
// hi = fp16(x*s), lo = fp16(x*s - hi) (s = per-tensor power of two that maps max|x|
// into [2^8, 2^9): 22 mantissa bits for elements within 2^-10 of the maximum, an
// absolute resolution of 2^-32 of the maximum below that -- fp16 subnormals are
// kept) and the product is accumulated in fp32 as hi*hi + hi*lo + lo*hi with three
// v_mfma_f32_32x32x16_f16 per K=16 slab instead of eight v_mfma_f32_32x32x2_f32:
// 16x the MFMA rate for 3x the MFMAs.  The dropped lo*lo term is 2^-22 relative.
// Tile 128x128x32, 4 waves (2x2, 64x64 each), both operands K-contiguous in LDS
// ([mn][32 + 8 pad] halfs -> conflict-free ds_read_b128), register prefetch of the
// next tile + double-buffered LDS, one barrier per K=32 step.
using hx8 = __attribute__((ext_vector_type(8))) _Float16;
using hx4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr int HBK = 32;
constexpr int HLD = HBK + 8;                 // halfs per LDS row (80 bytes)

__device__ __forceinline__ float pow2_scale(const float* absmax) {
  // 2^(9 - e) with max = f * 2^e, f in [0.5, 1): max*scale in [2^8, 2^9)
  if (absmax == nullptr) return 1.f;
  const unsigned b = __float_as_uint(*absmax);
  int e = (int)((b >> 23) & 0xff) - 126;
  if ((b & 0x7fffffffu) == 0u) return 1.f;
  int k = 9 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return __uint_as_float((unsigned)(127 + k) << 23);
}

// 16 fp32 values of one thread's share of a tile -> (hi, lo) halfs, [4 rows][4 k]
struct Frag16 { float v[4][4]; };   // v[r][c]: r = mn offset (0..3), c = k offset (0..3)

__device__ __forceinline__ void h_tile_load(const TileSrc& s, int k0, int mn0, int k_end,
                                            Frag16 (&f)[1]) {
  // mn-contiguous source: thread owns k rows 4kq..4kq+3 x mn 4mq..4mq+3
  // k-contiguous source:  thread owns mn rows r0 + {0,32,64,96}... see below
  const int tid = threadIdx.x;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) f[0].v[r][c] = 0.f;
  if (s.mn_contig) {
    const int kq = tid >> 5, mq = tid & 31;
    const int mn = mn0 + 4 * mq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + 4 * kq + c;
      if (k < k_end) {
        const float* src = s.p + (size_t)k * s.ld + mn;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (s.vec_ok && mn + 3 < s.mn_total) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (mn + e < s.mn_total) t[e] = src[e];
        }
        if (s.scale) {
          const float* sc = s.scale + (size_t)mod_period(k, s.period) * s.scale_ld + mn;
          if (s.scale_vec && mn + 3 < s.mn_total) {
            const float4 q = *reinterpret_cast<const float4*>(sc);
            t[0] *= q.x; t[1] *= q.y; t[2] *= q.z; t[3] *= q.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (mn + e < s.mn_total) t[e] *= sc[e];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) f[0].v[r][c] = t[r];
      }
    }
  } else {
    // 128 rows x 8 k-quads = 1024 float4; thread takes rows (tid>>3) + 32*r, quad tid&7
    const int kq = tid & 7;
    const int k = k0 + 4 * kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mn = mn0 + (tid >> 3) + 32 * r;
      if (mn < s.mn_total) {
        const float* src = s.p + (size_t)mn * s.ld + k;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (s.vec_ok && k + 3 < k_end) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < k_end) t[e] = src[e];
        }
        if (s.scale) {
          const float* sc = s.scale + (size_t)mod_period(mn, s.period) * s.scale_ld + k;
          if (s.scale_vec && k + 3 < k_end) {
            const float4 q = *reinterpret_cast<const float4*>(sc);
            t[0] *= q.x; t[1] *= q.y; t[2] *= q.z; t[3] *= q.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < k_end) t[e] *= sc[e];
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) f[0].v[r][c] = t[c];
      }
    }
  }
}

// tile row (0..31) of a thread for a k-contiguous operand: the two rows that share a
// 16-lane ds_write_b64 group are 4 apart (80-byte rows -> banks 16 apart: no overlap)
__device__ __forceinline__ int kc_row() {
  const int rr = threadIdx.x >> 3;
  return (rr & ~7) | ((rr & 1) << 2) | ((rr >> 1) & 3);
}

__device__ __forceinline__ void h_tile_store(int mn_contig, const Frag16 (&f)[1], float scale,
                                             _Float16 (*Shi)[HLD], _Float16 (*Slo)[HLD],
                                             int kq = -1) {
  const int tid = threadIdx.x;
  const int krow = kq < 0 ? (tid >> 3) : kc_row();   // fast path passes kq >= 0
  if (kq < 0) kq = tid >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    hx4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = f[0].v[r][c] * scale;
      const _Float16 h = (_Float16)x;
      hi[c] = h;
      lo[c] = (_Float16)(x - (float)h);
    }
    int row, col;
    if (mn_contig) { row = 4 * (tid & 31) + r; col = 4 * kq; }
    else { row = krow + 32 * r; col = 4 * (tid & 7); }
    *reinterpret_cast<hx4*>(&Shi[row][col]) = hi;
    *reinterpret_cast<hx4*>(&Slo[row][col]) = lo;
  }
}

struct HScales { const float* a_absmax; const float* b_absmax; };

__global__ void __launch_bounds__(256)
gemm_f16x2_kernel(TileSrc A, TileSrc B, int M, int N, int K, int k_per_split, Epilogue ep,
                  HScales hs) {
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  // [buf 2][A_hi, A_lo, B_hi, B_lo][128][HLD]
  auto tile = [&](int buf, int which) {
    return reinterpret_cast<_Float16 (*)[HLD]>(hsm + ((size_t)(buf * 4 + which) * 128) * HLD);
  };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > K) k_end = K;
  const float sa = pow2_scale(hs.a_absmax), sb = pow2_scale(hs.b_absmax);

  f32x16 am[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) am[i][j][e] = 0.f;

  Frag16 fa[1], fb[1];
  const int nk = (k_end - k_begin + HBK - 1) / HBK;
  if (nk > 0) {
    h_tile_load(A, k_begin, m0, k_end, fa);
    h_tile_load(B, k_begin, n0, k_end, fb);
    h_tile_store(A.mn_contig, fa, sa, tile(0, 0), tile(0, 1));
    h_tile_store(B.mn_contig, fb, sb, tile(0, 2), tile(0, 3));
  }
  __syncthreads();
  const int lrow = lane & 31, lk = 8 * (lane >> 5);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      h_tile_load(A, k_begin + (kt + 1) * HBK, m0, k_end, fa);
      h_tile_load(B, k_begin + (kt + 1) * HBK, n0, k_end, fb);
    }
    _Float16 (*Ah)[HLD] = tile(cur, 0);
    _Float16 (*Al)[HLD] = tile(cur, 1);
    _Float16 (*Bh)[HLD] = tile(cur, 2);
    _Float16 (*Bl)[HLD] = tile(cur, 3);
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      hx8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const hx8*>(&Ah[wm * 64 + i * 32 + lrow][16 * ks + lk]);
        al[i] = *reinterpret_cast<const hx8*>(&Al[wm * 64 + i * 32 + lrow][16 * ks + lk]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const hx8*>(&Bh[wn * 64 + j * 32 + lrow][16 * ks + lk]);
        bl[j] = *reinterpret_cast<const hx8*>(&Bl[wn * 64 + j * 32 + lrow][16 * ks + lk]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], am[i][j], 0, 0, 0);
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], am[i][j], 0, 0, 0);
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], am[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) {
      h_tile_store(A.mn_contig, fa, sa, tile(cur ^ 1, 0), tile(cur ^ 1, 1));
      h_tile_store(B.mn_contig, fb, sb, tile(cur ^ 1, 2), tile(cur ^ 1, 3));
    }
    __syncthreads();
  }
  const float unscale = 1.f / (sa * sb);
  const int lcol = lane & 31, lhalf = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lcol;
      if (col >= N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        if (row >= M) continue;
        float v = am[i][j][e] * unscale;
        if (ep.partial) {
          ep.partial[((size_t)blockIdx.z * M + row) * N + col] = v;
        } else {
          v *= ep.alpha;
          float* dst = ep.C + (size_t)row * ep.ldc + col;
          if (ep.bias) v += ep.bias[col];
          if (ep.c_scale) v *= ep.c_scale[(size_t)mod_period(row, ep.c_period) * ep.c_ld + col];
          if (ep.beta != 0.f) v += ep.beta * *dst;
          *dst = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------
// Fast path of the split-fp16 GEMM: the same tile / MFMA / LDS scheme with a
// BRANCH-FREE K loop.  All global reads are 16-byte buffer loads through a
// descriptor whose extent is the operand's; rows or K positions outside the problem
// get an out-of-range offset and come back as zeros, so the loop body is straight
// line code the compiler can software-pipeline (the generic kernel's per-element
// bound checks cost ~370 branches and ~110 waits per K slab).  Workgroups are mapped
// XCD-aware (see tile_of_block).  Preconditions (checked by the host, which otherwise
// falls back to the generic kernel): 16-byte aligned rows, K % 4 == 0, an
// mn-contiguous operand has mn % 4 == 0, extents < 4 GiB, mask period a power of two.
using u32x4g = __attribute__((ext_vector_type(4))) unsigned;
constexpr unsigned kOob = 0xFFFFFFF0u;

// blockIdx.x -> (row tile, column tile, K split).  The dispatcher deals consecutive
// workgroup ids round-robin over the 8 XCDs (id % 8), each with its own 4 MB L2, so
// every XCD gets a CONTIGUOUS range of an ordering in which the ~64 workgroups it runs
// at a time form a compact (8 row tiles x 8 column tiles) block of one K split: their
// A / B panels are fetched into that L2 once and shared.
struct TileId { int tm, tn, z; };
__device__ __forceinline__ TileId tile_of_id(int id, int TM, int TN, int splits) {
  constexpr int kXcd = 8, GM = 8;
  const int total = TM * TN * splits;
  const int q = total / kXcd, r = total % kXcd;       // bijective for any total
  const int xcd = id % kXcd, idx = id / kXcd;
  const int g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tiles = TM * TN;
  TileId t;
  t.z = g / tiles;
  const int rem = g - t.z * tiles;
  const int band = rem / (GM * TN);                   // band of GM row tiles
  const int in_band = rem - band * (GM * TN);
  const int rows_here = (TM - band * GM) < GM ? (TM - band * GM) : GM;
  t.tm = band * GM + in_band % rows_here;
  t.tn = in_band / rows_here;
  return t;
}
__device__ __forceinline__ TileId tile_of_block(int TM, int TN, int splits) {
  return tile_of_id((int)blockIdx.x, TM, TN, splits);
}

struct FastSrc {
  const float* p; int ld, mn_total;
  unsigned extent;                 // bytes addressable from p
  const float* scale; int pmask, scale_ld; unsigned scale_extent;
};

// k quad (0..7) of a thread for an mn-contiguous operand.  Rotating it with the column
// quad makes the 16 lanes that share a ds_write_b64 cycle hit 16 different bank pairs
// when they store their transposed 4x4 blocks ([mn][k] rows are 80 bytes apart, so
// unrotated lanes 4 rows apart collide 8-way); global reads stay 32-byte contiguous
// per lane pair and whole lines per workgroup.
__device__ __forceinline__ int mn_kquad() {
  const int tid = threadIdx.x;
  return ((tid >> 5) + ((tid & 31) >> 1)) & 7;
}

template <bool MN, bool MASK>
struct FastLoader {
  __amdgpu_buffer_rsrc_t rsrc, mrsrc;
  unsigned off[4], moff[4];        // byte offsets of this thread's four loads at k_begin
  unsigned step;                   // bytes per K slab
  int k0, k_end;                   // this thread's first K index, slab range end
  int pmask, scale_ld, mn;

  __device__ __forceinline__ void init(const FastSrc& s, int mn0, int k_begin, int kend) {
    const int tid = threadIdx.x;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p), 0, s.extent, 0x00020000);
    if (MASK)
      mrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.scale), 0, s.scale_extent,
                                                0x00020000);
    k_end = kend;
    pmask = s.pmask; scale_ld = s.scale_ld;
    if (MN) {        // storage (K, MN): thread = k rows 4*kq+c, columns 4*(tid&31)..+3
      k0 = k_begin + 4 * mn_kquad();
      mn = mn0 + 4 * (tid & 31);
      const bool ok = mn + 3 < s.mn_total;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        off[c] = ok ? (unsigned)(((size_t)(k0 + c) * s.ld + mn) * 4) : kOob;
      step = (unsigned)(HBK * s.ld * 4);
    } else {         // storage (MN, K): thread = rows kc_row()+32r, k quad tid&7
      k0 = k_begin + 4 * (tid & 7);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mn0 + kc_row() + 32 * r;
        off[r] = row < s.mn_total ? (unsigned)(((size_t)row * s.ld + k0) * 4) : kOob;
        if (MASK) moff[r] = (unsigned)(((size_t)(row & s.pmask) * s.scale_ld + k0) * 4);
      }
      step = (unsigned)(HBK * 4);
    }
  }

  // Issues the loads only; the mask (if any) lands in fm and is applied by apply_mask()
  // right before the values are needed, so nothing here waits on memory.
  __device__ __forceinline__ void load(int kt, Frag16 (&f)[1], Frag16 (&fm)[1]) const {
    float4 q[4], m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kt * HBK + (MN ? i : 0);
      const bool ok = k < k_end && off[i] != kOob;
      const unsigned o = ok ? off[i] + (unsigned)kt * step : kOob;
      q[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
      if (MASK) {
        const unsigned mraw = MN ? (unsigned)(((size_t)(k & pmask) * scale_ld + mn) * 4)
                                 : moff[i] + (unsigned)kt * step;
        const unsigned mo = ok ? mraw : kOob;
        m[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, mo, 0, 0));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MN) {      // load i = k row c: element e belongs to tile row e
        f[0].v[0][i] = q[i].x; f[0].v[1][i] = q[i].y; f[0].v[2][i] = q[i].z; f[0].v[3][i] = q[i].w;
        if (MASK) {
          fm[0].v[0][i] = m[i].x; fm[0].v[1][i] = m[i].y; fm[0].v[2][i] = m[i].z;
          fm[0].v[3][i] = m[i].w;
        }
      } else {       // load i = tile row r: elements are 4 consecutive k
        f[0].v[i][0] = q[i].x; f[0].v[i][1] = q[i].y; f[0].v[i][2] = q[i].z; f[0].v[i][3] = q[i].w;
        if (MASK) {
          fm[0].v[i][0] = m[i].x; fm[0].v[i][1] = m[i].y; fm[0].v[i][2] = m[i].z;
          fm[0].v[i][3] = m[i].w;
        }
      }
    }
  }
  __device__ __forceinline__ static void apply_mask(Frag16 (&f)[1], const Frag16 (&fm)[1]) {
    if (!MASK) return;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) f[0].v[r][c] *= fm[0].v[r][c];
  }
};

// NBUF = 2: double-buffered LDS (80 KB: 2 workgroups per CU, one barrier per slab);
// NBUF = 1: one LDS buffer (40 KB: 4 workgroups per CU, two barriers per slab).
template <bool AMN, bool BMN, bool MASK, int NBUF>
__global__ void __launch_bounds__(256)
gemm_f16x2_fast_kernel(FastSrc A, FastSrc B, int M, int N, int K, int k_per_split, int splits,
                       Epilogue ep, HScales hs) {
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  // [buf 2][A_hi, A_lo, B_hi, B_lo][128][HLD]
  auto tile = [&](int buf, int which) {
    return reinterpret_cast<_Float16 (*)[HLD]>(hsm + ((size_t)(buf * 4 + which) * 128) * HLD);
  };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileId tb = tile_of_block((M + BM - 1) / BM, (N + BN - 1) / BN, splits);
  const int m0 = tb.tm * BM, n0 = tb.tn * BN;
  const int k_begin = tb.z * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > K) k_end = K;
  const float sa = pow2_scale(hs.a_absmax), sb = pow2_scale(hs.b_absmax);

  f32x16 am[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) am[i][j][e] = 0.f;

  const int kq_rot = mn_kquad();
  FastLoader<AMN, MASK> la;
  FastLoader<BMN, false> lb;
  la.init(A, m0, k_begin, k_end);
  lb.init(B, n0, k_begin, k_end);
  // Two register sets: while slab kt is multiplied out of LDS, slab kt+1 (loaded one
  // iteration ago) is converted into the other LDS buffer and the loads of slab kt+2
  // are already in flight -- a full iteration of latency cover.
  Frag16 fa0[1], fb0[1], fa1[1], fb1[1], fm0[1], fm1[1], fmb[1];
  const int nk = (k_end - k_begin + HBK - 1) / HBK;
  la.load(0, fa0, fm0);
  lb.load(0, fb0, fmb);
  la.load(1, fa1, fm1);
  lb.load(1, fb1, fmb);
  la.apply_mask(fa0, fm0);
  h_tile_store(AMN, fa0, sa, tile(0, 0), tile(0, 1), kq_rot);
  h_tile_store(BMN, fb0, sb, tile(0, 2), tile(0, 3), kq_rot);
  __syncthreads();
  const int lrow = lane & 31, lk = 8 * (lane >> 5);
  auto slab = [&](int kt, Frag16 (&fl_a)[1], Frag16 (&fl_m)[1], Frag16 (&fl_b)[1],
                  Frag16 (&fs_a)[1], Frag16 (&fs_m)[1], Frag16 (&fs_b)[1]) {
    const int cur = NBUF == 2 ? (kt & 1) : 0;
    const int nxt = NBUF == 2 ? (cur ^ 1) : 0;
    // past the last slab every offset is out of range: those loads return zeros, unused
    la.load(kt + 2, fl_a, fl_m);
    lb.load(kt + 2, fl_b, fmb);
    _Float16 (*Ah)[HLD] = tile(cur, 0);
    _Float16 (*Al)[HLD] = tile(cur, 1);
    _Float16 (*Bh)[HLD] = tile(cur, 2);
    _Float16 (*Bl)[HLD] = tile(cur, 3);
#pragma unroll
    for (int ks = 0; ks < HBK / 16; ++ks) {
      hx8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const hx8*>(&Ah[wm * 64 + i * 32 + lrow][16 * ks + lk]);
        al[i] = *reinterpret_cast<const hx8*>(&Al[wm * 64 + i * 32 + lrow][16 * ks + lk]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const hx8*>(&Bh[wn * 64 + j * 32 + lrow][16 * ks + lk]);
        bl[j] = *reinterpret_cast<const hx8*>(&Bl[wn * 64 + j * 32 + lrow][16 * ks + lk]);
      }
      // term-major order: consecutive MFMAs go to four different accumulators, so none
      // waits for the previous one's result (the small cross terms first)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], am[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], am[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], am[i][j], 0, 0, 0);
    }
    la.apply_mask(fs_a, fs_m);
    if (NBUF == 1) __syncthreads();          // every wave has read the slab
    h_tile_store(AMN, fs_a, sa, tile(nxt, 0), tile(nxt, 1), kq_rot);
    h_tile_store(BMN, fs_b, sb, tile(nxt, 2), tile(nxt, 3), kq_rot);
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    slab(kt, fa0, fm0, fb0, fa1, fm1, fb1);     // loads slab kt+2 -> set 0, stores set 1
    if (kt + 1 < nk) slab(kt + 1, fa1, fm1, fb1, fa0, fm0, fb0);
  }
  const float unscale = 1.f / (sa * sb);
  const int lcol = lane & 31, lhalf = lane >> 5;
  if (m0 + BM <= M && n0 + BN <= N) {
    // interior tile: no per-element bound checks, so the 16 loads of a fragment (mask,
    // old C) are issued back to back before the first is needed
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + lcol;
        const int row0 = m0 + wm * 64 + i * 32 + 4 * lhalf;
        if (ep.partial) {
          float* dst = ep.partial + ((size_t)tb.z * M + row0) * N + col;
#pragma unroll
          for (int e = 0; e < 16; ++e)
            dst[(size_t)((e & 3) + 8 * (e >> 2)) * N] = am[i][j][e] * unscale;
          continue;
        }
        float* dst = ep.C + (size_t)row0 * ep.ldc + col;
        const float bias = ep.bias ? ep.bias[col] : 0.f;
        float old[16], msk[16];
        const bool use_old = ep.beta != 0.f;
        const bool use_msk = ep.c_scale != nullptr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          old[e] = use_old ? dst[(size_t)dr * ep.ldc] : 0.f;
          msk[e] = use_msk ? ep.c_scale[(size_t)mod_period(row0 + dr, ep.c_period) * ep.c_ld + col]
                           : 1.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          const float v = (am[i][j][e] * unscale * ep.alpha + bias) * msk[e];
          dst[(size_t)dr * ep.ldc] = use_old ? v + ep.beta * old[e] : v;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lcol;
      if (col >= N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        if (row >= M) continue;
        float v = am[i][j][e] * unscale;
        if (ep.partial) {
          ep.partial[((size_t)tb.z * M + row) * N + col] = v;
        } else {
          v *= ep.alpha;
          float* dst = ep.C + (size_t)row * ep.ldc + col;
          if (ep.bias) v += ep.bias[col];
          if (ep.c_scale) v *= ep.c_scale[(size_t)mod_period(row, ep.c_period) * ep.c_ld + col];
          if (ep.beta != 0.f) v += ep.beta * *dst;
          *dst = v;
        }
      }
    }
}

// ===========================================================================
// Packed split-fp16 operands ("hl" planes): convert once, multiply many times.
//
// gemm_f16x2_fast_kernel splits every fp32 element into hi/lo halfs INSIDE its K loop
// (~140 VALU per 24 MFMAs and slab), and a tile of an operand is re-read -- and re-split --
// by every workgroup of its row / column band (a dz element ~20 times per step).  Here the
// split happens once per tensor: pack_hl_kernel writes two fp16 planes hi = fp16(x*s),
// lo = fp16(x*s - hi) (s = the per-tensor power of two of pow2_scale; variational-dropout
// masks are multiplied in BEFORE the split, in fp32, so the GEMM needs no mask path) with the
// source's columns ("r" planes: (rows, ldk_r)) or rows ("c" planes: (cols, ldk_c)) contiguous:
//   x, y (.) B_U, dz: "r" planes only.  x@W and dz@W^T reduce over their columns (row-major
//     form of the GEMM); the weight gradients x^T dz and h^T dz reduce over their ROWS and read
//     the same planes transposed out of LDS (k_major form, HlLoaderT) -- the second
//     orientation was a third of the pack passes' bytes;
//   W: both ("c" = the B operand of x@W, "r" = the B operand of dz@W^T).
// A plane row holds ldk reduction indices (ldk % 32 == 0, zero padded).  The two planes are ONE
// array, interleaved in groups of 16 indices: a row is 2 ldk halfs,
//   half [32 g, 32 g + 16)      = hi[16 g .. 16 g + 15]
//   half [32 g + 16, 32 g + 32) = lo[16 g .. 16 g + 15]
// so the 32-deep K slab of a row -- both planes -- is ONE 128-byte line.  (With separate
// planes a slab touched half of a line in each, the other halves belonging to the next slab,
// by which time the 32 KB L1 had long turned over: every line crossed L2 -> L1 twice.)  A
// reduction offset is a multiple of 16 (a group).  gemm_hlx_kernel is then a
// plain fp16 GEMM with three MFMAs per fragment pair and fp32 accumulation: 16-byte global
// loads -> ds_write_b128 -> ds_read_b128 (ds_read_b64_tr_b16 in the k_major form) ->
// v_mfma_f32_16x16x32_f16, no VALU in the K loop besides addresses.
struct HlSrc {
  const _Float16* p;      // interleaved planes, offset to (first row, first reduction group)
  int ld;                 // plane columns per row: a row is 2 ld halfs (hi and lo)
  int rows;               // MN extent (plane rows in the row-major form, columns in k_major)
  unsigned extent;        // bytes addressable from p
};
// Segmented reduction range of the row-major A operand (asr_gemm_hl_args.a_seg_k): the K range
// is the concatenation of segments of seg_k indices (a multiple of 32: whole slabs), segment i
// read from the planes shifted by a byte offset -- the implicit im2col over time of
// asr_conv2d_*: tap dt of a filter reads the activation planes dt frames further on.  The
// loader adds adj[slab / slabs_per_segment] (= the segment's byte shift minus the bytes the
// running reduction offset has advanced by then) to its offset; slab / sps is a multiply-shift
// with a magic the host verified over the launch's slab range.  magic = 0: no segments.
struct HlSeg {
  unsigned magic;
  unsigned batch;          // k_major only: > 1 = a batch of GEMMs sharing B (asr_gemm_hl_args.batch):
  unsigned adj[16];        //   adj[b] = byte offset of A_b; else the segments' offset corrections
};
__device__ __forceinline__ size_t hl_index(int row, int k, int ld) {      // half index of hi
  return (size_t)row * (2 * (size_t)ld) + (size_t)(k >> 4) * 32 + (k & 15);
}

// 64 x 64 source tile per workgroup; thread (ty = tid >> 4, tx = tid & 15) owns the 4 x 4
// block rows 4 ty.., columns 4 tx..  Both orientations leave through LDS images laid out in
// OUTPUT order -- per output row 4 groups of (16 hi, 16 lo) = 256 contiguous bytes -- so that a
// lane stores 16 bytes and 16 lanes a whole 256-byte run.  (Storing the 8-byte pieces a thread
// holds wrote half of every 64 bytes per instruction: the interleaved layout then cost the
// pack 12 %.)  The transposed image is written with its 8-byte granules XOR-swizzled by the
// column quad (16 lanes of one store = 16 columns, same granule: 16-way conflict otherwise;
// VERDICT r2 weak 7); an odd swizzle swaps the halves of a 16-byte chunk, undone at the read.
__global__ void __launch_bounds__(256)
pack_hl_kernel(const float* __restrict__ src, int rows, int cols, int ld,
               const float* __restrict__ mask, int mask_period, int mask_ld,
               const float* __restrict__ absmax, float* __restrict__ scale_out,
               _Float16* __restrict__ r_hl, int ldk_r, _Float16* __restrict__ c_hl, int ldk_c,
               const float* __restrict__ mask2, _Float16* __restrict__ r2_hl) {
  // one 16 KB image per output that is asked for (dynamic LDS: a row-only pack keeps 10
  // workgroups on a CU).  mask2 / r2_hl: a SECOND set of row planes of the same source under
  // another mask (the two directions' variational-dropout masks of a BiLSTM input): the source
  // is read once for both (asr_pack_args.mask2).
  extern __shared__ __attribute__((aligned(16))) _Float16 pack_lds[];
  typedef _Float16 (*Image)[128];
  Image rimg = reinterpret_cast<Image>(pack_lds);
  Image cimg = reinterpret_cast<Image>(pack_lds + (r_hl ? 64 * 128 : 0));
  Image rimg2 = reinterpret_cast<Image>(pack_lds + ((r_hl ? 1 : 0) + (c_hl ? 1 : 0)) * 64 * 128);
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float s = pow2_scale(absmax);
  if (scale_out && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *scale_out = s;
  const int c = c0 + 4 * tx;
  hx4 hi[4], lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 4 * ty + i;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* q = src + (size_t)r * ld + c;
      if (c + 3 < cols) {
        const float4 t = *reinterpret_cast<const float4*>(q);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = q[e];
      }
      if (r2_hl) {                              // second row image: the source under mask2
        float w[4] = {v[0], v[1], v[2], v[3]};
        const float* m2 = mask2 + (size_t)mod_period(r, mask_period) * mask_ld + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < cols) w[e] *= m2[e];
        hx4 h2, l2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = w[e] * s;
          const _Float16 h = (_Float16)x;
          h2[e] = h;
          l2[e] = (_Float16)(x - (float)h);
        }
        _Float16* q = &rimg2[4 * ty + i][(tx >> 2) * 32 + (tx & 3) * 4];
        *reinterpret_cast<hx4*>(q) = h2;
        *reinterpret_cast<hx4*>(q + 16) = l2;
      }
      if (mask) {
        // (n_pad is a multiple of 16, not necessarily a power of two: 48, 80, 96 ...)
        const float* m = mask + (size_t)mod_period(r, mask_period) * mask_ld + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] *= m[e];
      }
    } else if (r2_hl) {
      _Float16* q = &rimg2[4 * ty + i][(tx >> 2) * 32 + (tx & 3) * 4];
      *reinterpret_cast<hx4*>(q) = hx4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      *reinterpret_cast<hx4*>(q + 16) = hx4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * s;
      const _Float16 h = (_Float16)x;
      hi[i][e] = h;
      lo[i][e] = (_Float16)(x - (float)h);
    }
    if (r_hl) {                               // row image: group tx >> 2, halfs 4 (tx & 3) ..
      _Float16* q = &rimg[4 * ty + i][(tx >> 2) * 32 + (tx & 3) * 4];
      *reinterpret_cast<hx4*>(q) = hi[i];
      *reinterpret_cast<hx4*>(q + 16) = lo[i];
    }
  }
  if (c_hl) {
    // column image: granule (8 bytes) of rows 4 ty .. of column 4 tx + e, swizzled by tx
    const int gr = (ty >> 2) * 8 + (ty & 3);
    const int f = ((2 * tx) & 14) | (tx >> 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hx4 a, b;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = hi[i][e]; b[i] = lo[i][e]; }
      *reinterpret_cast<hx4*>(&cimg[4 * tx + e][(gr ^ f) * 4]) = a;
      *reinterpret_cast<hx4*>(&cimg[4 * tx + e][((gr + 4) ^ f) * 4]) = b;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;             // 64 output rows x 16 chunks of 16 bytes
    const int rr = idx >> 4, j = idx & 15;
    const int k0 = (j >> 2) * 16;               // first reduction index of the chunk's group
    if (r_hl && r0 + rr < rows && c0 + k0 < ldk_r)       // (columns in [cols, ldk_r): zeros)
      *reinterpret_cast<hx8*>(r_hl + hl_index(r0 + rr, c0 + k0, ldk_r) + (j & 3) * 8) =
          *reinterpret_cast<const hx8*>(&rimg[rr][j * 8]);
    if (r2_hl && r0 + rr < rows && c0 + k0 < ldk_r)
      *reinterpret_cast<hx8*>(r2_hl + hl_index(r0 + rr, c0 + k0, ldk_r) + (j & 3) * 8) =
          *reinterpret_cast<const hx8*>(&rimg2[rr][j * 8]);
    if (c_hl && c0 + rr < cols && r0 + k0 < ldk_c) {     // (rows in [rows, ldk_c): zeros)
      const int f = ((2 * (rr >> 2)) & 14) | (rr >> 5);
      hx8 v = *reinterpret_cast<const hx8*>(&cimg[rr][(j ^ (f >> 1)) * 8]);
      if (f & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
      *reinterpret_cast<hx8*>(c_hl + hl_index(c0 + rr, r0 + k0, ldk_c) + (j & 3) * 8) = v;
    }
  }
}

// Row planes only, streaming (r6): the form of the pack the training step runs on its critical
// path (a layer input under one or both directions' masks; y (.) B_U).  pack_hl_kernel stages 64 x
// 64 tiles through LDS because it may have to transpose; a row-only pack has nothing to
// transpose: a thread takes FOUR consecutive columns of one row (one 16-byte load, coalesced over
// the wave), the four lanes of a quad -- 16 columns, one (16 hi, 16 lo) group -- exchange their
// 8-byte pieces with DPP quad permutes so that every lane stores 16 contiguous bytes (lane 0 / 1:
// the hi halfs, lane 2 / 3: the lo halfs), and a wave writes 1 KB runs.  No LDS, no barrier.
// Same arithmetic as pack_hl_kernel: bit-identical planes.  Needs cols % 16 == 0 == ldk_r - cols,
// 16-byte aligned rows of source and masks.
template <int CTRL>
__device__ __forceinline__ unsigned quad_from(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__global__ void __launch_bounds__(256)
pack_rows_kernel(const float* __restrict__ src, long long total4, int c4, int ld,
                 const float* __restrict__ mask, const float* __restrict__ mask2, int mask_period,
                 int mask_ld, const float* __restrict__ absmax, float* __restrict__ scale_out,
                 _Float16* __restrict__ r_hl, _Float16* __restrict__ r2_hl, int ldk) {
  const float s = pow2_scale(absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const int q = threadIdx.x & 3;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4;
       idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / c4), c = 4 * (int)(idx - (long long)r * c4);
    const float4 t = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
    const float v[4] = {t.x, t.y, t.z, t.w};
    const size_t mo = (size_t)mod_period(r, mask_period) * mask_ld + c;
    auto emit = [&](const float* m, _Float16* out) {
      float w[4] = {v[0], v[1], v[2], v[3]};
      if (m) {
        const float4 mm = *reinterpret_cast<const float4*>(m + mo);
        w[0] *= mm.x; w[1] *= mm.y; w[2] *= mm.z; w[3] *= mm.w;
      }
      hx4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = w[e] * s;
        const _Float16 hh = (_Float16)x;
        h[e] = hh;
        l[e] = (_Float16)(x - (float)hh);
      }
      const unsigned long long hb = __builtin_bit_cast(unsigned long long, h);
      const unsigned long long lb = __builtin_bit_cast(unsigned long long, l);
      const unsigned h0 = (unsigned)hb, h1 = (unsigned)(hb >> 32);
      const unsigned l0 = (unsigned)lb, l1 = (unsigned)(lb >> 32);
      // lane q of the quad stores: 0 -> hi of lanes 0, 1; 1 -> hi of lanes 2, 3; 2 -> lo of
      // lanes 0, 1; 3 -> lo of lanes 2, 3   (quad_perm [0,2,0,2] = 0x88, [1,3,1,3] = 0xDD)
      const bool hi_lane = q < 2;
      u32x4g o;
      // (every permute is executed by ALL lanes and the pieces selected afterwards: a permute
      // inside a divergent branch would read lanes that are switched off)
      const unsigned ha0 = quad_from<0x88>(h0), ha1 = quad_from<0x88>(h1);
      const unsigned hb0 = quad_from<0xDD>(h0), hb1 = quad_from<0xDD>(h1);
      const unsigned la0 = quad_from<0x88>(l0), la1 = quad_from<0x88>(l1);
      const unsigned lb0 = quad_from<0xDD>(l0), lb1 = quad_from<0xDD>(l1);
      o[0] = hi_lane ? ha0 : la0;
      o[1] = hi_lane ? ha1 : la1;
      o[2] = hi_lane ? hb0 : lb0;
      o[3] = hi_lane ? hb1 : lb1;
      *reinterpret_cast<u32x4g*>(out + hl_index(r, c & ~15, ldk) + 8 * q) = o;
    };
    emit(mask, r_hl);
    if (r2_hl) emit(mask2, r2_hl);
  }
}

// ---------------------------------------------------------------------------
// The packed-plane kernel.  256 x 256 x 32 tile by default (outputs of at least 256 x 256).
// With 128 x 128 tiles both split-fp16 GEMMs sit at ~250 TF/s algorithmic whatever the K loop
// costs: a workgroup moves 32 KB from L2 per 1 MFLOP (32 flop/B), i.e. ~7.5 TB/s at that
// rate, with an L2 hit rate of ~72 % -- the loop is fed at the L2 / fabric rate.  A 256 x 256
// tile halves the bytes per flop.  512 threads = 8 waves (2 x 4, 128 x 64 each = 8 x 4 MFMA
// tiles of 16 x 16, 128 accumulator registers); LDS image per plane [256 rows][32 halfs]
// (64-byte rows; 4 planes x 2 buffers = 128 KB) with the 16-byte chunk index XOR-swizzled by a
// permutation of (row >> 2) & 3, which makes both the ds_write_b128 of the staging pass and the
// ds_read_b128 of the 16 x 32 fragments bank-conflict free (lane groups of
// MI355X_MICROARCH.md "LDS"; checked exhaustively when the layout was chosen).
//
// The kernel runs at the 1400 W package power cap (tools/clock_probe.py): what it sustains is
// set by energy per flop, not by issue slots.  v_mfma_f32_16x16x32_f16 sustains 2.0 PFLOP/s on
// random operands at the cap (shader clock 2.04 GHz) where v_mfma_f32_32x32x16_f16 sustains
// 1.66 (1.80 GHz) -- tools/micro/mfma_power.hip -- so the tile is built from the 16 x 16 shape.

__device__ __forceinline__ int hl256_slot(int row, int kc) {      // half index in a plane
  return row * 32 + ((kc ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 3);    // 0, 2, 3, 1
}

template <int NT, bool SEG = false>
struct HlLoaderX {
  // One operand's share of a K slab: (NT / 2) rows x 128 bytes (the row's line: hi and lo of 32
  // reduction indices) = 4 NT 16-byte chunks; thread t takes chunks t + NT i, i < 4: row
  // (t >> 3) + (NT / 8) i, chunk t & 7 of the line -- a wave instruction reads 8 whole lines.
  // Chunk c of a line is reduction chunk 2 (c >> 2) + (c & 1) of plane (c >> 1) & 1.  One
  // offset register: the four rows are a uniform stride apart (scalar offset operand, which
  // the descriptor's range check does not see: rows past the operand and the reduction tail
  // are compares that send the lane's offset out of range -> zeros).
  __amdgpu_buffer_rsrc_t rs;
  unsigned off;            // bytes: (row0 + t >> 3, first reduction group) + 16 (t & 7)
  unsigned row_step;       // bytes between the thread's consecutive chunks: NT / 8 rows
  int kq;                  // first reduction index of the thread's chunk within a slab
  int depth;               // reduction indices from the first slab on
  int rows_left;           // operand rows from the thread's first one on
  int g0;                  // index of the first slab (kb / 32): segments count from k = 0
  const HlSeg* sg;         // segmented reduction range (kernel argument), or its magic is 0
  __device__ __forceinline__ void init(const HlSrc& s, int row0, int kb, int ke,
                                       const HlSeg* seg = nullptr) {
    const int tid = threadIdx.x;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(s.p), 0, s.extent, 0x00020000);
    const int c = tid & 7;
    kq = 16 * (c >> 2) + 8 * (c & 1);
    depth = ke - kb;
    rows_left = s.rows - row0 - (tid >> 3);
    off = (unsigned)((hl_index(row0 + (tid >> 3), kb, s.ld) + 8 * c) * 2);
    row_step = (unsigned)(NT / 8) * (unsigned)s.ld * 4u;
    g0 = kb / HBK;
    sg = seg;
  }
  // (the address arithmetic apart from the loads: VALU at the head of an MFMA phase is slow)
  __device__ __forceinline__ void offsets(int kt, unsigned (&o)[4]) const {
    unsigned base = off + (unsigned)(kt * HBK * 4);
    if constexpr (SEG)                         // uniform: scalar multiply-shift + one scalar load
      base += sg->adj[((unsigned)(g0 + kt) * sg->magic) >> 20];
    const unsigned ok = kt * HBK + kq < depth ? base : kOob;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = i * (NT / 8) < rows_left ? ok : kOob;
  }
  __device__ __forceinline__ void issue(const unsigned (&o)[4], u32x4g (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, o[i], i * row_step, 0);
  }
  __device__ __forceinline__ void load(int kt, u32x4g (&v)[4]) const {
    unsigned o[4];
    offsets(kt, o);
    issue(o, v);
  }
  __device__ __forceinline__ static void store(const u32x4g (&v)[4], _Float16* Shi, _Float16* Slo) {
    const int tid = threadIdx.x;
    const int c = tid & 7;
    _Float16* dst = ((c >> 1) & 1 ? Slo : Shi) + hl256_slot(tid >> 3, 2 * (c >> 2) + (c & 1));
#pragma unroll
    for (int i = 0; i < 4; ++i)       // + NT / 8 rows: the same swizzle (NT / 8 is a multiple of 16)
      *reinterpret_cast<u32x4g*>(dst + i * (NT / 8) * 32) = v[i];
  }
};

// K-MAJOR operands (asr_gemm_hl_args.k_major): the weight gradients x^T dz and h^T dz reduce over
// the ROWS of their operands, i.e. they can read the same (rows, cols) planes as x@W and
// dz@W^T if the fragments are transposed on the way out of LDS -- ds_read_b64_tr_b16 -- and the
// second orientation of x, y and dz is never packed (a third of the pack passes' bytes).
// A slab is 32 plane rows (reduction indices) x TM2 columns: per row TM2 / 16 groups of
// (16 hi, 16 lo) = 4 TM2 contiguous bytes; a wave's load instruction reads one whole row.
// LDS image per operand: one SUBTILE per (column group, plane): [32 k][16 columns] halfs,
// 32-byte rows, 1 KB, the layout ds_read_b64_tr_b16 reads conflict-free (guide T10): lane
// l = 16 g + c of a read at byte 8 l receives column c of rows 4 g .. 4 g + 3.  Two reads (rows
// 0..15, rows 16..31) give a lane the reduction indices {4 g .., 16 + 4 g ..}: a permutation of
// the MFMA's k order that both operands share, so the products are unchanged.  Subtiles are
// 1056 bytes apart: the 8 lanes of a staging ds_write_b128 group hold 4 subtiles x 2 halves of
// one row, and 32-byte steps spread them over all banks.
constexpr int kSubtile = 1056;                       // bytes
template <int NT>
struct HlLoaderT {
  static constexpr int CPR = NT / 8;                 // 16-byte chunks per slab row (tile / 4)
  __amdgpu_buffer_rsrc_t rs;
  unsigned off;            // bytes: (first row + t / CPR, tile's first column group) + 16 (t % CPR)
  unsigned row_step;       // bytes between the thread's consecutive chunks: 8 rows
  unsigned slab_step;      // bytes per slab: 32 rows
  int rows_left;           // reduction rows from the thread's first one on (slab 0)
  __device__ __forceinline__ void init(const HlSrc& s, int col0, int kb, int ke) {
    const int tid = threadIdx.x;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(s.p), 0, s.extent, 0x00020000);
    const unsigned pitch = (unsigned)s.ld * 4u;      // bytes per plane row (hi + lo)
    const int r = tid / CPR, c = tid % CPR;
    rows_left = ke - kb - r;
    // (column groups past the operand's M / N extent: out of range -> zeros; the last group
    // may hold columns past it, which feed only output rows / columns that are never stored)
    const bool col_ok = col0 + 16 * (c >> 2) < ((s.rows + 15) & ~15);
    off = col_ok ? (unsigned)(kb + r) * pitch + (unsigned)col0 * 4u + 16u * c : kOob;
    row_step = 8u * pitch;
    slab_step = 32u * pitch;
  }
  __device__ __forceinline__ void offsets(int kt, unsigned (&o)[4]) const {
    const unsigned ok = off == kOob ? kOob : off + (unsigned)kt * slab_step;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = kt * HBK + 8 * i < rows_left ? ok : kOob;
  }
  __device__ __forceinline__ void issue(const unsigned (&o)[4], u32x4g (&v)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, o[i], i * row_step, 0);
  }
  __device__ __forceinline__ void load(int kt, u32x4g (&v)[4]) const {
    unsigned o[4];
    offsets(kt, o);
    issue(o, v);
  }
  // chunk c of row r: column group c >> 2, plane (c >> 1) & 1 -> subtile c >> 1, its half c & 1
  __device__ __forceinline__ static void store(const u32x4g (&v)[4], char* image) {
    const int tid = threadIdx.x;
    const int r = tid / CPR, c = tid % CPR;
    char* dst = image + (c >> 1) * kSubtile + r * 32 + (c & 1) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4g*>(dst + i * 256) = v[i];
  }
  // the 16 (columns of group `grp`) x 32 (k) fragment of one plane as the MFMA wants it
  __device__ __forceinline__ static hx8 fragment(const char* image, int grp, int plane, int lane) {
    typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
    const char* q = image + (grp * 2 + plane) * kSubtile + lane * 8;
    const s4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4*)(q));
    const s4 t2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4*)(q + 512));
    const hx4 a = __builtin_bit_cast(hx4, t1), b = __builtin_bit_cast(hx4, t2);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// The epilogue of one output tile of gemm_hlx_kernel / gemm_hlp_kernel: lane = (row frow of a
// 16 x 16 tile, its columns 4 fk .. 4 fk + 3); z = the split (partial sums) the tile belongs to.
template <int MI, int NJ, bool SEG>
__device__ __forceinline__ void hlx_epilogue(const f32x4 (&am)[2 * MI][2 * NJ], const Epilogue ep,
                                             int M, int N, int m0, int n0, int z, float unscale,
                                             int wm, int wn, int frow, int fk, int TM2, int TN2) {
  constexpr int RB = 2 * MI, CB = 2 * NJ;
  const int row_w = m0 + wm * (32 * MI) + frow;      // + 16 i
  const int col_w = n0 + wn * (32 * NJ) + 4 * fk;    // + 16 j
  const bool interior = m0 + TM2 <= M && n0 + TN2 <= N;
  // 16-byte stores need 16-byte aligned rows
  const bool vec_c = (ep.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(ep.C) & 15) == 0;
  // The epilogue is 1/5 of a K = 1024 tile if it is written element by element with its
  // options tested inside (hipcc branches around every optional load and waits for it): the
  // variants are separated OUTSIDE the element loops, the interior ones are straight-line code.
  if (interior && ep.partial && (N & 3) == 0) {                 // split-K partial sums
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        float* dst = ep.partial + ((size_t)z * M + row_w + 16 * i) * N + col_w + 16 * j;
        *reinterpret_cast<f32x4*>(dst) = am[i][j] * unscale;
      }
    return;
  }
  const bool use_old = ep.beta != 0.f;
  const bool use_msk = ep.c_scale != nullptr;
  if (interior && !ep.partial && vec_c && !use_old && !use_msk) {      // C = alpha A B^T + bias
    const float sc = unscale * ep.alpha;
    f32x4 bias[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ep.bias)
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[j][e] = ep.bias[col_w + 16 * j + e];
    }
    if constexpr (SEG) {
      if (ep.clamp_hi > 0.f) {          // the convolution's clipped ReLU, fused
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            float* dst = ep.C + (size_t)(row_w + 16 * i) * ep.ldc + col_w + 16 * j;
            f32x4 v = am[i][j] * sc + bias[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), ep.clamp_hi);
            *reinterpret_cast<f32x4*>(dst) = v;
          }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        float* dst = ep.C + (size_t)(row_w + 16 * i) * ep.ldc + col_w + 16 * j;
        *reinterpret_cast<f32x4*>(dst) = am[i][j] * sc + bias[j];
      }
    return;
  }
  const bool vec_m = !use_msk || ((ep.c_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(ep.c_scale) & 15) == 0);
  if (interior && !ep.partial && vec_c && vec_m) {   // with old C and / or a row mask: loads first
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = row_w + 16 * i;
      f32x4 old[CB], msk[CB];
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        old[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        msk[j] = f32x4{1.f, 1.f, 1.f, 1.f};
      }
      if (use_old) {
#pragma unroll
        for (int j = 0; j < CB; ++j)
          old[j] = *reinterpret_cast<const f32x4*>(ep.C + (size_t)row * ep.ldc + col_w + 16 * j);
      }
      if (use_msk) {
        const float* mrow = ep.c_scale + (size_t)mod_period(row, ep.c_period) * ep.c_ld;
#pragma unroll
        for (int j = 0; j < CB; ++j)
          msk[j] = *reinterpret_cast<const f32x4*>(mrow + col_w + 16 * j);
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ep.bias)
#pragma unroll
          for (int e = 0; e < 4; ++e) bias[e] = ep.bias[col_w + 16 * j + e];
        float* dst = ep.C + (size_t)row * ep.ldc + col_w + 16 * j;
        *reinterpret_cast<f32x4*>(dst) = (am[i][j] * (unscale * ep.alpha) + bias) * msk[j] + ep.beta * old[j];
      }
    }
    return;
  }
  // edge tiles (and unaligned outputs): per-element bound checks
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int row = row_w + 16 * i;
      if (row >= M) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = col_w + 16 * j + e;
        if (col >= N) continue;
        if (ep.partial) {
          ep.partial[((size_t)z * M + row) * N + col] = am[i][j][e] * unscale;
          continue;
        }
        float* dst = ep.C + (size_t)row * ep.ldc + col;
        float v = am[i][j][e] * unscale * ep.alpha + (ep.bias ? ep.bias[col] : 0.f);
        if (use_msk) v *= ep.c_scale[(size_t)mod_period(row, ep.c_period) * ep.c_ld + col];
        if (use_old) v += ep.beta * *dst;
        if constexpr (SEG) {
          if (ep.clamp_hi > 0.f) v = fminf(fmaxf(v, 0.f), ep.clamp_hi);
        }
        *dst = v;
      }
    }
}

// WN waves across the columns (2 rows of waves); a wave owns 32 MI x 32 NJ of the tile as
// RB x CB = 2 MI x 2 NJ MFMA tiles of 16 x 16: the square tile is 64 MI = 32 NJ WN wide and
// every thread stages four 16-byte chunks per operand.
//   <4, 4, 2>: 256 x 256, 512 threads, 128 KB LDS -- the main kernel;
//   <2, 2, 2>: 128 x 128, 256 threads,  64 KB LDS -- small outputs, and the one that fits on a
//              CU BESIDE a recurrent workgroup (96 KB + 64 KB of LDS, 2 + 1 waves per SIMD).
// The MFMA takes the fragment of B (16 columns of C) as its first operand and the fragment of A
// (16 rows of C) as its second: a lane then holds FOUR CONSECUTIVE COLUMNS of one row of C
// (row = lane & 15, columns 4 (lane >> 4) ..), and the epilogue is 16-byte stores.
//
// Debug (asr_gemm_hl_profile): shader clocks per phase of the K loop, summed over the slabs of
// workgroup 0 of a launch, per wave: 0 fragment reads, 1 barrier before the MFMAs, 2 MFMAs of
// the first half of the rows (with the staging traffic), 3 MFMAs of the second half, 4 barrier
// behind the MFMAs, 5 the prologue; 6 = the K loop in ticks of the 100 MHz real-time counter.
__device__ int g_hl_prof_on = 0;
__device__ long long g_hl_prof[8][8];

template <int WN, int MI, int NJ, bool KM, bool SEG = false>
__global__ void __launch_bounds__(128 * WN)
gemm_hlx_kernel(HlSrc A, HlSrc B, int M, int N, int K, int k_per_split, int splits,
                Epilogue ep, const float* __restrict__ a_scale,
                const float* __restrict__ b_scale, HlSeg seg) {
  constexpr int NT = 128 * WN, TM2 = 64 * MI, TN2 = 32 * NJ * WN;
  constexpr int RB = 2 * MI, CB = 2 * NJ;           // 16 x 16 tiles per wave: rows x columns
  static_assert(TM2 == TN2 && TM2 * 8 == 4 * NT, "square tile, four chunks per thread and operand");
  // KM: the operands are K-major planes (rows = reduction index), HlLoaderT; else HlLoaderX
  // SEG: A's reduction range is segmented (HlSeg; the conv2d entry points) -- its own
  // instantiation, so that the plain kernels' K loop is untouched
  using Loader = std::conditional_t<KM, HlLoaderT<NT>, HlLoaderX<NT>>;
  using LoaderA = std::conditional_t<KM, HlLoaderT<NT>, HlLoaderX<NT, SEG>>;
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  constexpr int kOpBytes = (TM2 / 16) * 2 * kSubtile;          // KM: one operand's slab image
  auto opimg = [&](int buf, int which) {
    return reinterpret_cast<char*>(hsm) + (size_t)(buf * 2 + which) * kOpBytes;
  };
  // halfs per plane: + 64 bytes, so that a row's hi and lo chunks (one 8-lane group of the
  // staging ds_write_b128) fall into different banks
  constexpr int kPlane = TM2 * 32 + 32;
  auto plane = [&](int buf, int which) { return hsm + (size_t)(buf * 4 + which) * kPlane; };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;          // 2 x WN waves, (32 MI) x (32 NJ) each
  // (K-major batches: z = split * batch + b, so that the split-K partials of all batch members
  // form ONE ([split][batch * M][N]) array and the plain reduce kernel folds them)
  const int nbatch = (KM && seg.batch > 1u) ? (int)seg.batch : 1;
  const TileId tb = tile_of_block((M + TM2 - 1) / TM2, (N + TN2 - 1) / TN2, splits * nbatch);
  const int m0 = tb.tm * TM2, n0 = tb.tn * TN2;
  const int zb = tb.z % nbatch;
  const int k_begin = (tb.z / nbatch) * k_per_split;
  int k_end = k_begin + k_per_split;
  if (k_end > K) k_end = K;
  const float sa = a_scale ? *a_scale : 1.f, sb = b_scale ? *b_scale : 1.f;
  if constexpr (KM) {
    if (nbatch > 1) {       // member zb: A shifted by its byte offset, C by zb x M rows
      const unsigned o = seg.adj[zb];
      A.p = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(A.p) + o);
      A.extent -= o;
      ep.C += (size_t)zb * M * ep.ldc;
    }
  }

  f32x4 am[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) am[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool prof = g_hl_prof_on != 0 && blockIdx.x == 0;
  long long pt[6] = {0, 0, 0, 0, 0, 0};
  long long plast = prof ? (long long)__builtin_readcyclecounter() : 0;
  auto stamp = [&](int i) {
    if (prof) {
      const long long now = (long long)__builtin_readcyclecounter();
      pt[i] += now - plast;
      plast = now;
    }
  };
  LoaderA la;
  Loader lb;
  if constexpr (KM) la.init(A, m0, k_begin, k_end);
  else la.init(A, m0, k_begin, k_end, &seg);
  lb.init(B, n0, k_begin, k_end);
  // One register set, two slabs ahead: at the top of step kt the registers hold slab kt+1
  // (issued a whole step earlier, so it has landed); it is written to the LDS buffer the
  // barrier at the end of step kt-1 released, and the loads of slab kt+2 are issued at once --
  // they fly across this step's 96 MFMAs per wave AND its barriers (plain buffer loads, no
  // vmcnt wait at the barrier).  The scheduling fence keeps the compiler from sinking the
  // loads to their use, which would expose the whole L2/HBM latency every step.
  u32x4g av[4], bv[4];
  const int nk = (k_end - k_begin + HBK - 1) / HBK;
  la.load(0, av);
  lb.load(0, bv);
  if constexpr (KM) {
    Loader::store(av, opimg(0, 0));
    Loader::store(bv, opimg(0, 1));
  } else {
    Loader::store(av, plane(0, 0), plane(0, 1));
    Loader::store(bv, plane(0, 2), plane(0, 3));
  }
  la.load(1, av);
  lb.load(1, bv);
  __syncthreads();
  const int frow = lane & 15, fk = lane >> 4;       // fragment: row of its 16, 16-byte K chunk
  // Ping-pong: a step is four phases -- read the fragments of half of the wave's rows (and, the
  // first time, of all its columns), 12 MI NJ MFMAs on them (with, in the first half, the LDS
  // writes of the next slab and the loads of the one after interleaved), and again for the
  // other rows -- each closed by a workgroup barrier.  The second row of
  // waves (wm = 1; waves w and w+4 share a SIMD) runs ONE BARRIER LATE, so on every SIMD one
  // wave multiplies while the other reads: the matrix pipe no longer idles through the LDS
  // round trips of two waves in lockstep.  Buffer hazards with the one-phase lag: a slab's
  // buffer is last read three phases (>= one barrier) before either row rewrites it, and is
  // first read three phases after the later row wrote it.
  if (wm == 1) __builtin_amdgcn_s_barrier();
  stamp(5);
  const long long real0 = prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const _Float16* Ah = plane(cur, 0);
    const _Float16* Al = plane(cur, 1);
    const _Float16* Bh = plane(cur, 2);
    const _Float16* Bl = plane(cur, 3);
    hx8 fah[MI], fal[MI], fbh[CB], fbl[CB];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if constexpr (KM) {
        if (hf == 0) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            fbh[j] = Loader::fragment(opimg(cur, 1), wn * CB + j, 0, lane);
            fbl[j] = Loader::fragment(opimg(cur, 1), wn * CB + j, 1, lane);
          }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          fah[i] = Loader::fragment(opimg(cur, 0), wm * RB + hf * MI + i, 0, lane);
          fal[i] = Loader::fragment(opimg(cur, 0), wm * RB + hf * MI + i, 1, lane);
        }
      } else {
      if (hf == 0) {
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int slot = hl256_slot(wn * (32 * NJ) + j * 16 + frow, fk);
          fbh[j] = *reinterpret_cast<const hx8*>(Bh + slot);
          fbl[j] = *reinterpret_cast<const hx8*>(Bl + slot);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int slot = hl256_slot(wm * (32 * MI) + (hf * MI + i) * 16 + frow, fk);
        fah[i] = *reinterpret_cast<const hx8*>(Ah + slot);
        fal[i] = *reinterpret_cast<const hx8*>(Al + slot);
      }
      }
      unsigned oa[4], ob[4];
      if (hf == 0) {
        la.offsets(kt + 2, oa);
        lb.offsets(kt + 2, ob);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(oa[i]), "+v"(ob[i]));
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(0);
      __syncthreads();
      stamp(1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (hf == 0) {
        // the staging traffic rides in the issue gaps of this phase's MFMAs (16 cycles of pipe
        // per MFMA, ~4 of issue): the 8 LDS writes of slab kt+1 first -- each frees its
        // registers -- then the 8 loads of slab kt+2 into them; the read phases stay bare.
        // Past the last slab every offset is out of range: those loads return zeros, unused.
        if constexpr (KM) {
          Loader::store(av, opimg(nxt, 0));
          Loader::store(bv, opimg(nxt, 1));
        } else {
          Loader::store(av, plane(nxt, 0), plane(nxt, 1));
          Loader::store(bv, plane(nxt, 2), plane(nxt, 3));
        }
        la.issue(oa, av);
        lb.issue(ob, bv);
      }
      // term-major order: consecutive MFMAs go to MI CB different accumulators
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fal[i], am[hf * MI + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[j], fah[i], am[hf * MI + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fah[i], am[hf * MI + i][j], 0, 0, 0);
      if (hf == 0) {
        constexpr int NM = 3 * MI * CB;                    // MFMAs of the phase: 48 or 24
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM >= 48 ? 2 : 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                  // DS write
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM >= 48 ? 4 : 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // VMEM read
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(hf == 0 ? 2 : 3);
      __syncthreads();
      stamp(4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if (prof && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) g_hl_prof[wave][i] = pt[i];
    g_hl_prof[wave][6] = (long long)__builtin_amdgcn_s_memrealtime() - real0;
  }
  // ---- epilogue
  hlx_epilogue<MI, NJ, SEG>(am, ep, M, N, m0, n0, tb.z, 1.f / (sa * sb), wm, wn, frow, fk, TM2, TN2);
}


// PERSISTENT row-major form (gemm_hlp_kernel; round 6).  The 256 x 256 tile of the plain kernel
// pays ~17 us per output tile around its K loop -- the dispatch of a new workgroup onto a CU whose
// LDS and registers the previous one filled, a cold prologue (two dependent L2 / HBM round
// trips), 256 KB of epilogue stores that must drain before the workgroup may end -- which is
// 19 % of a K = 1024 tile (x@W: 73 us of K loop) and 5 % of a K = 4096 one (dX); measured as
// 1.447 / 1.245 ms for the two shapes of equal flops.  Here ONE workgroup per CU walks tiles:
// the K loop never stops -- the loads two slabs ahead run on into the NEXT tile's first slabs
// (a load cursor of its own: per-tile buffer descriptor and row bounds in SGPRs, the thread's
// offsets are tile-invariant), the epilogue of a finished tile is issued at the top of the next
// tile's first step, i.e. behind the closing barrier, where its VALU work and stores overlap the
// MFMA phase of the other wave row of the ping-pong, and the accumulators are cleared there.
// Products and their order within a tile are those of gemm_hlx_kernel: results are BIT-IDENTICAL
// (tests/test_gpu_gemm_hl.py).  Tiles are handed out dynamically, per XCD, in the order of
// tile_of_block (ctl[x] = next index of XCD x's contiguous range: L2 locality as before, and a
// CU that starts late -- a kernel of another stream still on it -- simply takes fewer tiles);
// the id of the next tile is fetched by thread 0 at a tile's first step, posted in LDS at the
// second and read by all at the third (needs K >= 4 slabs).  ctl[8] counts finished workgroups
// (atomicInc wraps it to 0); the last one clears ctl[0..7]: a control block is reusable by the
// next launch without a memset.
template <int WN, int MI, int NJ>
__global__ void __launch_bounds__(128 * WN)
gemm_hlp_kernel(HlSrc A, HlSrc B, int M, int N, int K, Epilogue ep,
                const float* __restrict__ a_scale, const float* __restrict__ b_scale,
                unsigned* __restrict__ ctl) {
  constexpr int NT = 128 * WN, TM2 = 64 * MI, TN2 = 32 * NJ * WN;
  constexpr int RB = 2 * MI, CB = 2 * NJ;
  static_assert(TM2 == TN2 && TM2 * 8 == 4 * NT, "square tile, four chunks per thread and operand");
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];
  constexpr int kPlane = TM2 * 32 + 32;
  auto plane = [&](int buf, int which) { return hsm + (size_t)(buf * 4 + which) * kPlane; };
  int* mbox = reinterpret_cast<int*>(hsm + (size_t)8 * kPlane);     // two tile-id slots
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int TMc = (M + TM2 - 1) / TM2, TNc = (N + TN2 - 1) / TN2, total = TMc * TNc;
  const int xcd = (int)(blockIdx.x % 8u);
  const int nk = (K + HBK - 1) / HBK;
  const float sa = a_scale ? *a_scale : 1.f, sb = b_scale ? *b_scale : 1.f;
  const float unscale = 1.f / (sa * sb);

  if (tid == 0) mbox[0] = (int)(atomicAdd(&ctl[xcd], 1u) * 8u) + xcd;
  __syncthreads();
  int vt = __builtin_amdgcn_readfirstlane(mbox[0]);          // the tile being multiplied
  int v_nxt = total;                                           // the one after it (known from kt = 2)
  if (vt < total) {
    // ---- load cursor: tile lv, slab lkt of it; everything uniform except the two offsets
    const int c8 = tid & 7, r8 = tid >> 3;
    const int kq = 16 * (c8 >> 2) + 8 * (c8 & 1);
    const unsigned offA = (unsigned)r8 * (unsigned)A.ld * 4u + 16u * c8;
    const unsigned offB = (unsigned)r8 * (unsigned)B.ld * 4u + 16u * c8;
    const unsigned stepA = (unsigned)(NT / 8) * (unsigned)A.ld * 4u;
    const unsigned stepB = (unsigned)(NT / 8) * (unsigned)B.ld * 4u;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int ra_left = 0, rb_left = 0, lkt = 0;
    auto setup = [&](int v) {
      if (v < total) {
        const TileId t = tile_of_id(v, TMc, TNc, 1);
        const unsigned ba = (unsigned)(t.tm * TM2) * (unsigned)A.ld * 4u;
        const unsigned bb = (unsigned)(t.tn * TN2) * (unsigned)B.ld * 4u;
        rsA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(A.p)) + ba, 0, A.extent - ba, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(B.p)) + bb, 0, B.extent - bb, 0x00020000);
        ra_left = M - t.tm * TM2;
        rb_left = N - t.tn * TN2;
      } else {                                   // past the last tile: every offset out of range
        ra_left = 0;
        rb_left = 0;
      }
    };
    auto offsets = [&](unsigned (&oa)[4], unsigned (&ob)[4]) {
      const bool kin = lkt * HBK + kq < K;
      const unsigned a0 = kin ? offA + (unsigned)(lkt * HBK * 4) : kOob;
      const unsigned b0 = kin ? offB + (unsigned)(lkt * HBK * 4) : kOob;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        oa[i] = r8 + i * (NT / 8) < ra_left ? a0 : kOob;
        ob[i] = r8 + i * (NT / 8) < rb_left ? b0 : kOob;
      }
    };
    auto issue = [&](const unsigned (&oa)[4], const unsigned (&ob)[4], u32x4g (&av)[4],
                     u32x4g (&bv)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        av[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa[i], i * stepA, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        bv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ob[i], i * stepB, 0);
    };
    auto advance = [&]() {                       // to the next slab (of the next tile)
      if (++lkt == nk) {
        lkt = 0;
        setup(v_nxt);
      }
    };
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A.p), 0, A.extent, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(B.p), 0, B.extent, 0x00020000);
    setup(vt);

    f32x4 am[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) am[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4g av[4], bv[4];
    {
      unsigned oa[4], ob[4];
      offsets(oa, ob);
      issue(oa, ob, av, bv);
      advance();
      HlLoaderX<NT>::store(av, plane(0, 0), plane(0, 1));
      HlLoaderX<NT>::store(bv, plane(0, 2), plane(0, 3));
      offsets(oa, ob);
      issue(oa, ob, av, bv);
      advance();
    }
    __syncthreads();
    const int frow = lane & 15, fk = lane >> 4;
    TileId tc = tile_of_id(vt, TMc, TNc, 1);
    int m0 = tc.tm * TM2, n0 = tc.tn * TN2;
    unsigned pending = 0u;                        // thread 0: the fetched index of the next tile
    int ti = 0;                                   // tiles this workgroup has started
    int kt = 0;
    int cur = 0;
    if (wm == 1) __builtin_amdgcn_s_barrier();    // the second wave row runs one barrier late
    for (;;) {
      // ---- the id of the next tile: fetched, posted, read (one step each)
      if (kt == 0) {
        if (tid == 0) pending = atomicAdd(&ctl[xcd], 1u);
      } else if (kt == 1) {
        if (tid == 0) mbox[(ti + 1) & 1] = (int)(pending * 8u) + xcd;
      } else if (kt == 2) {
        v_nxt = __builtin_amdgcn_readfirstlane(mbox[(ti + 1) & 1]);
      }
      const int nxt = cur ^ 1;
      const _Float16* Ah = plane(cur, 0);
      const _Float16* Al = plane(cur, 1);
      const _Float16* Bh = plane(cur, 2);
      const _Float16* Bl = plane(cur, 3);
      hx8 fah[MI], fal[MI], fbh[CB], fbl[CB];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (hf == 0) {
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            const int slot = hl256_slot(wn * (32 * NJ) + j * 16 + frow, fk);
            fbh[j] = *reinterpret_cast<const hx8*>(Bh + slot);
            fbl[j] = *reinterpret_cast<const hx8*>(Bl + slot);
          }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int slot = hl256_slot(wm * (32 * MI) + (hf * MI + i) * 16 + frow, fk);
          fah[i] = *reinterpret_cast<const hx8*>(Ah + slot);
          fal[i] = *reinterpret_cast<const hx8*>(Al + slot);
        }
        unsigned oa[4], ob[4];
        if (hf == 0) {
          offsets(oa, ob);
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(oa[i]), "+v"(ob[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (hf == 0) {
          HlLoaderX<NT>::store(av, plane(nxt, 0), plane(nxt, 1));
          HlLoaderX<NT>::store(bv, plane(nxt, 2), plane(nxt, 3));
          issue(oa, ob, av, bv);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fal[i], am[hf * MI + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl[j], fah[i], am[hf * MI + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j)
            am[hf * MI + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[j], fah[i], am[hf * MI + i][j], 0, 0, 0);
        if (hf == 0) {
          constexpr int NM = 3 * MI * CB;
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM >= 48 ? 2 : 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                  // DS write
          }
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM >= 48 ? 4 : 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // VMEM read
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (hf == 0) advance();                  // (scalar: the cursor's next slab / tile)
      }
      cur = nxt;
      if (++kt == nk) {
        // ---- a finished tile: its epilogue here, behind the closing barrier -- the other wave
        // row's MFMA phase runs meanwhile -- and the accumulators start the next tile at zero
        // (its lane coordinates are made opaque HERE: the address arithmetic they feed must not
        // be hoisted out of the K loop, where every register is taken)
        int fr = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(fr), "+v"(fq));
        hlx_epilogue<MI, NJ, false>(am, ep, M, N, m0, n0, 0, unscale, wm, wn, fr, fq, TM2, TN2);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int j = 0; j < CB; ++j) am[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        kt = 0;
        ++ti;
        vt = v_nxt;
        if (vt >= total) break;
        tc = tile_of_id(vt, TMc, TNc, 1);
        m0 = tc.tm * TM2;
        n0 = tc.tn * TN2;
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  }
  // ---- the last workgroup to finish re-arms the control block
  if (tid == 0) {
    const unsigned done = atomicInc(&ctl[8], gridDim.x - 1u);
    if (done == gridDim.x - 1u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicExch(&ctl[i], 0u);
    }
  }
}


// max |x| of a flat tensor -> out[0] (float).  Two launches: per-block maxima via
// atomicMax on the float bits (all non-negative, so integer order == float order).
__global__ void __launch_bounds__(256)
absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  const size_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    const float4 q = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
  // one atomic per workgroup (atomics to one address serialise in the L2)
  __shared__ float wmax[4];
  m = asr_wave_max(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));
  }
}

// Column sums (bias gradients): out[n] = beta*out[n] + sum_m X[m][n].  HBM-bound:
// X is read exactly once with 256-byte coalesced rows.  Stage 1: grid (N/64, RS);
// each 256-thread block owns 64 columns x one slice of rows (4 row phases, float64
// accumulators, LDS combine) and writes one float64 partial per column; stage 2
// adds the RS partials in a fixed order (deterministic).
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ X, int M, int N, int ldx, int rows_per_slice,
                      double* __restrict__ partial) {
  __shared__ double part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int m0 = blockIdx.y * rows_per_slice;
  int m1 = m0 + rows_per_slice;
  if (m1 > M) m1 = M;
  double s0 = 0.0, s1 = 0.0;
  if (col < N) {
    int m = m0 + ry;
    for (; m + 4 < m1; m += 8) {
      s0 += (double)X[(size_t)m * ldx + col];
      s1 += (double)X[(size_t)(m + 4) * ldx + col];
    }
    for (; m < m1; m += 4) s0 += (double)X[(size_t)m * ldx + col];
  }
  part[ry][cx] = s0 + s1;
  __syncthreads();
  if (ry == 0 && col < N)
    partial[(size_t)blockIdx.y * N + col] = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
}

// 64 columns per workgroup; the four waves take every fourth slice (independent loads,
// four accumulators each) and meet in LDS.
__global__ void __launch_bounds__(256)
colsum_final_kernel(const double* __restrict__ partial, int slices, int N,
                    float* __restrict__ out, float beta) {
  __shared__ double part[4][64];
  const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  if (col < N) {
    int s = sg;
    for (; s + 12 < slices; s += 16) {
      t0 += partial[(size_t)s * N + col];
      t1 += partial[(size_t)(s + 4) * N + col];
      t2 += partial[(size_t)(s + 8) * N + col];
      t3 += partial[(size_t)(s + 12) * N + col];
    }
    for (; s < slices; s += 4) t0 += partial[(size_t)s * N + col];
  }
  part[sg][cx] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (sg == 0 && col < N) {
    const double t = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
    out[col] = (beta != 0.f ? beta * out[col] : 0.f) + (float)t;
  }
}

int colsum_slices(int M) {
  int rs = (M + 127) / 128;
  if (rs > 256) rs = 256;
  if (rs < 1) rs = 1;
  return rs;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Dynamic-LDS "ballast" for the small helper kernels that run on the weight-gradient
// stream beside a BPTT kernel: a recurrent workgroup reserves 96 KB of its CU's 160 KB,
// so a helper asking for 80 KB can never be placed on the same CU and steal issue slots
// from its one-wave-per-SIMD critical path.
static size_t lds_ballast(const void* kernel) {
  const size_t bytes = 80 * 1024;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}

extern "C" size_t asr_gemm_workspace_bytes(const asr_gemm_args* a) {
  if (!a || a->split_k <= 1) return 0;
  return asr_align_up((size_t)a->split_k * a->M * a->N * sizeof(float), 256);
}

extern "C" int asr_gemm(const asr_gemm_args* a, void* workspace, size_t ws_bytes,
                        asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(a && a->A && a->B && a->C, "gemm: null pointer");
  ASR_CHECK_ARG(a->M > 0 && a->N > 0 && a->K >= 0, "gemm: bad shape %d %d %d", a->M, a->N, a->K);
  TileSrc A, B;
  A.p = a->A; A.ld = a->lda; A.mn_contig = a->trans_a ? 1 : 0;
  A.mn_total = a->M; A.k_total = a->K;
  A.scale = a->a_scale; A.period = a->a_scale_period > 0 ? a->a_scale_period : 1;
  A.scale_ld = a->a_scale_ld;
  A.scale_vec = a->a_scale && (a->a_scale_ld % 4 == 0) && aligned16(a->a_scale);
  A.vec_ok = (a->lda % 4 == 0) && aligned16(a->A);
  B.p = a->B; B.ld = a->ldb; B.mn_contig = a->trans_b ? 0 : 1;
  B.mn_total = a->N; B.k_total = a->K;
  B.scale = nullptr; B.period = 1; B.scale_ld = 0; B.scale_vec = 0;
  B.vec_ok = (a->ldb % 4 == 0) && aligned16(a->B);
  ASR_CHECK_ARG(a->lda >= (a->trans_a ? a->M : a->K), "gemm: lda too small");
  ASR_CHECK_ARG(a->ldb >= (a->trans_b ? a->K : a->N), "gemm: ldb too small");
  ASR_CHECK_ARG(a->ldc >= a->N, "gemm: ldc too small");
  constexpr int BK = 16;                       // K slab of the exact-fp32 kernel
  int splits = a->split_k > 1 ? a->split_k : 1;
  int k_per_split = (a->K + splits - 1) / splits;
  k_per_split = (k_per_split + BK - 1) / BK * BK;
  if (k_per_split < BK) k_per_split = BK;
  // vector loads along K need 4-aligned split starts: BK multiple guarantees it
  while (splits > 1 && (size_t)(splits - 1) * k_per_split >= (size_t)a->K) --splits;
  Epilogue ep;
  ep.C = a->C; ep.ldc = a->ldc; ep.alpha = a->alpha; ep.beta = a->beta; ep.bias = a->bias;
  ep.c_scale = a->c_scale; ep.c_period = a->c_scale_period > 0 ? a->c_scale_period : 1;
  ep.c_ld = a->c_scale_ld; ep.partial = nullptr; ep.clamp_hi = 0.f;
  if (splits > 1) {
    const size_t need = (size_t)splits * a->M * a->N * sizeof(float);
    if (!workspace || ws_bytes < need) {
      asr_set_error("gemm: split-K workspace %zu < %zu bytes", ws_bytes, need);
      return ASR_ERR_WORKSPACE;
    }
    ep.partial = reinterpret_cast<float*>(workspace);
  }
  dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM, splits);
  const int tiles_mn = (int)(grid.x * grid.y);
  static const int prec_env = [] { const char* v = getenv("ASR_GEMM_PREC"); return v ? atoi(v) : 1; }();
  const int prec = a->precision == 0 ? 0 : (a->precision == 1 ? 1 : prec_env);
  if (prec == 1 && a->K >= 32) {
    // split-fp16 path: K slabs of 32
    int kps = (a->K + splits - 1) / splits;
    kps = (kps + HBK - 1) / HBK * HBK;
    int sp = splits;
    while (sp > 1 && (size_t)(sp - 1) * kps >= (size_t)a->K) --sp;
    grid.z = sp;
    const size_t shm = (size_t)2 * 4 * 128 * HLD * sizeof(_Float16);
    static bool attr_done = false;
    if (!attr_done) {
      ASR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f16x2_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      attr_done = true;
    }
    HScales hs{a->a_absmax, a->b_absmax};
    // fast (branch-free) kernel when the geometry allows 16-byte buffer loads throughout
    auto extent = [](const TileSrc& t) {
      return t.mn_contig ? ((size_t)(t.k_total - 1) * t.ld + t.mn_total) * 4
                         : ((size_t)(t.mn_total - 1) * t.ld + t.k_total) * 4;
    };
    const size_t lim = ((size_t)1 << 32) - ((size_t)1 << 20);
    const bool pow2 = (A.period & (A.period - 1)) == 0;
    const size_t mask_ext = A.scale ? ((size_t)(A.period - 1) * A.scale_ld +
                                       (A.mn_contig ? A.mn_total : A.k_total)) * 4 : 0;
    const bool fast = A.vec_ok && B.vec_ok && a->K % 4 == 0 &&
                      (!A.mn_contig || a->M % 4 == 0) && (!B.mn_contig || a->N % 4 == 0) &&
                      extent(A) < lim && extent(B) < lim &&
                      (!A.scale || (A.scale_vec && pow2 && mask_ext < lim));
    if (fast) {
      FastSrc fa_, fb_;
      fa_.p = A.p; fa_.ld = A.ld; fa_.mn_total = A.mn_total; fa_.extent = (unsigned)extent(A);
      fa_.scale = A.scale; fa_.pmask = A.period - 1; fa_.scale_ld = A.scale_ld;
      fa_.scale_extent = (unsigned)mask_ext;
      fb_.p = B.p; fb_.ld = B.ld; fb_.mn_total = B.mn_total; fb_.extent = (unsigned)extent(B);
      fb_.scale = nullptr; fb_.pmask = 0; fb_.scale_ld = 0; fb_.scale_extent = 0;
      const int total = tiles_mn * sp;
      const int key = (A.mn_contig ? 4 : 0) | (B.mn_contig ? 2 : 0) | (A.scale ? 1 : 0);
#define ASR_FAST_CASE(KEY, AMN, BMN, MASK)                                                    \
      case KEY: {                                                                              \
        static bool done = false;                                                              \
        if (!done) {                                                                           \
          ASR_CHECK_HIP(hipFuncSetAttribute(                                                   \
              (const void*)gemm_f16x2_fast_kernel<AMN, BMN, MASK, 2>,                          \
              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                          \
          done = true;                                                                         \
        }                                                                                      \
        hipLaunchKernelGGL((gemm_f16x2_fast_kernel<AMN, BMN, MASK, 2>), dim3(total),           \
                           dim3(256), shm, stream, fa_, fb_, a->M, a->N, a->K, kps, sp, ep,    \
                           hs);                                                                \
      } break;
      switch (key) {
        ASR_FAST_CASE(0, false, false, false)
        ASR_FAST_CASE(1, false, false, true)
        ASR_FAST_CASE(2, false, true, false)
        ASR_FAST_CASE(3, false, true, true)
        ASR_FAST_CASE(4, true, false, false)
        ASR_FAST_CASE(5, true, false, true)
        ASR_FAST_CASE(6, true, true, false)
        ASR_FAST_CASE(7, true, true, true)
      }
#undef ASR_FAST_CASE
    } else {
      hipLaunchKernelGGL(gemm_f16x2_kernel, grid, dim3(256), shm, stream, A, B, a->M, a->N, a->K,
                         kps, ep, hs);
    }
    splits = sp;
  } else
    hipLaunchKernelGGL(gemm_f32_mfma_kernel<16>, grid, dim3(256), 0, stream, A, B, a->M, a->N,
                       a->K, k_per_split, ep);
  ASR_CHECK_LAUNCH();
  if (splits > 1) {
    Epilogue ep2 = ep;
    ep2.partial = nullptr;
    lds_ballast((const void*)gemm_splitk_reduce4_kernel);
    launch_splitk_reduce(reinterpret_cast<const float*>(workspace), splits, a->M, a->N, ep2,
                         (unsigned)lds_ballast((const void*)gemm_splitk_reduce_kernel), stream);
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}

extern "C" int asr_pack_hl(const asr_pack_args* a, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(a && a->src && a->rows > 0 && a->cols > 0 && a->ld >= a->cols,
                "pack_hl: bad source");
  ASR_CHECK_ARG(aligned16(a->src) && a->ld % 4 == 0, "pack_hl: source rows must be 16-byte aligned");
  ASR_CHECK_ARG(a->r_hl || a->c_hl, "pack_hl: need the planes of at least one orientation");
  if (a->r_hl)
    ASR_CHECK_ARG(a->ldk_r % 32 == 0 && a->ldk_r >= a->cols && a->ldk_r < a->cols + 32 &&
                  aligned16(a->r_hl), "pack_hl: bad row-plane geometry");
  if (a->c_hl)
    ASR_CHECK_ARG(a->ldk_c % 32 == 0 && a->ldk_c >= a->rows && a->ldk_c < a->rows + 32 &&
                  aligned16(a->c_hl), "pack_hl: bad column-plane geometry");
  if (a->mask)
    ASR_CHECK_ARG(a->mask_period > 0 && a->mask_ld >= a->cols,
                  "pack_hl: a mask needs a positive row period and mask_ld >= cols");
  if (a->r2_hl)
    ASR_CHECK_ARG(a->r_hl && a->mask && a->mask2 && aligned16(a->r2_hl),
                  "pack_hl: the second row planes need r_hl, mask and mask2 (same period / ld)");
  // row planes only, whole groups, aligned rows: the streaming kernel (ASR_PACK_ROWS=0: the tiled one)
  {
    const char* pr = getenv("ASR_PACK_ROWS");
    const bool rows_ok = a->r_hl && !a->c_hl && a->cols % 16 == 0 && a->ldk_r == a->cols &&
                         a->ld % 4 == 0 && aligned16(a->src) &&
                         (!a->mask || (a->mask_ld % 4 == 0 && aligned16(a->mask))) &&
                         (!a->r2_hl || aligned16(a->mask2)) && !(pr && *pr == '0');
    if (rows_ok) {
      const long long total4 = (long long)a->rows * (a->cols / 4);
      long long blocks = (total4 + 255) / 256;
      if (blocks > 256 * 32) blocks = 256 * 32;
      hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a->src, total4,
                         a->cols / 4, a->ld, a->mask, a->r2_hl ? a->mask2 : nullptr,
                         a->mask ? a->mask_period : 1, a->mask_ld, a->absmax, a->scale_out,
                         reinterpret_cast<_Float16*>(a->r_hl), reinterpret_cast<_Float16*>(a->r2_hl),
                         a->ldk_r);
      ASR_CHECK_LAUNCH();
      return ASR_OK;
    }
  }
  dim3 grid((a->cols + 63) / 64, (a->rows + 63) / 64);
  const size_t pack_shm = (size_t)((a->r_hl ? 1 : 0) + (a->c_hl ? 1 : 0) + (a->r2_hl ? 1 : 0)) *
                          64 * 128 * sizeof(_Float16);
  hipLaunchKernelGGL(pack_hl_kernel, grid, dim3(256), pack_shm, stream, a->src, a->rows, a->cols, a->ld,
                     a->mask, a->mask ? a->mask_period : 1, a->mask_ld, a->absmax, a->scale_out,
                     reinterpret_cast<_Float16*>(a->r_hl), a->ldk_r,
                     reinterpret_cast<_Float16*>(a->c_hl), a->ldk_c,
                     a->r2_hl ? a->mask2 : nullptr, reinterpret_cast<_Float16*>(a->r2_hl));
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

// Debug: enable != 0 arms the phase profile of the next asr_gemm_hl launches;
// out64 (may be NULL) receives the 8 waves x 8 phases of the last armed launch's workgroup 0.
extern "C" int asr_gemm_hl_profile(int enable, long long* out48, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_HIP(hipStreamSynchronize(stream));
  if (out48) ASR_CHECK_HIP(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_hl_prof), sizeof(long long) * 64));
  ASR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hl_prof_on), &enable, sizeof(int)));
  return ASR_OK;
}


// ---- host side of the persistent form: control blocks (9 words each, zero between launches:
// the kernel re-arms its block itself), handed out round-robin from a ring per device so that
// launches in flight on different streams never share one
static bool hlp_enabled() {
  const char* v = getenv("ASR_GEMM_PERSIST");
  return !(v && *v == '0');
}
static int hlp_num_cu() {
  static int ncu[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    ncu[dev] = prop.multiProcessorCount;
  }
  return ncu[dev];
}
// (a __device__ array: zero at module load, one instance per device, no allocation -- the library
// still owns no device memory it had to ask for)
constexpr int kHlpRing = 256, kHlpWords = 16;
__device__ unsigned g_hlp_ctl[kHlpRing * kHlpWords];
static unsigned* hlp_control_block() {
  static unsigned* ring[16] = {nullptr};
  static std::atomic<unsigned> next[16];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!ring[dev]) {
    std::lock_guard<std::mutex> lock(mu);
    if (!ring[dev]) {
      void* p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_hlp_ctl)) != hipSuccess || !p) return nullptr;
      ring[dev] = reinterpret_cast<unsigned*>(p);
    }
  }
  return ring[dev] + (size_t)(next[dev].fetch_add(1u) % kHlpRing) * kHlpWords;
}

static int hl_splits(const asr_gemm_hl_args* a, int* kps_out) {
  int splits = a->split_k > 1 ? a->split_k : 1;
  int kps = (a->K + splits - 1) / splits;
  kps = (kps + HBK - 1) / HBK * HBK;
  while (splits > 1 && (size_t)(splits - 1) * kps >= (size_t)a->K) --splits;
  if (kps_out) *kps_out = kps;
  return splits;
}

extern "C" size_t asr_gemm_hl_workspace_bytes(const asr_gemm_hl_args* a) {
  if (!a) return 0;
  const int splits = hl_splits(a, nullptr);
  if (splits <= 1) return 0;
  const size_t nbatch = a->batch > 1 ? a->batch : 1;
  return asr_align_up((size_t)splits * nbatch * a->M * a->N * sizeof(float), 256);
}

extern "C" int asr_gemm_hl(const asr_gemm_hl_args* a, void* workspace, size_t ws_bytes,
                           asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(a && a->a_hl && a->b_hl && a->C, "gemm_hl: null pointer");
  ASR_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && (a->k_major || a->K % 8 == 0),
                "gemm_hl: bad shape %d %d %d (K must be a multiple of 8)", a->M, a->N, a->K);
  const bool km = a->k_major != 0;
  if (km)
    ASR_CHECK_ARG(a->lda >= a->M && a->ldb >= a->N && a->lda % 16 == 0 && a->ldb % 16 == 0 &&
                  a->ldc >= a->N, "gemm_hl (k_major): bad leading dimensions");
  else
  ASR_CHECK_ARG((a->lda >= a->K || a->a_seg_k > 0) && a->ldb >= a->K && a->lda % 16 == 0 &&
                a->ldb % 16 == 0 && a->ldc >= a->N,
                "gemm_hl: bad leading dimensions (multiples of 16: whole (hi, lo) groups)");
  ASR_CHECK_ARG((reinterpret_cast<uintptr_t>(a->a_hl) & 63) == 0 &&
                (reinterpret_cast<uintptr_t>(a->b_hl) & 63) == 0,
                "gemm_hl: planes must start at a reduction group (64-byte aligned)");
  // bytes from the first row's first group to the end of the last row's last group
  // (k_major: K plane rows of 4 ld bytes, the last one up to the operand's last column group)
  const size_t ext_a = km ? ((size_t)(a->K - 1) * a->lda + (size_t)((a->M + 15) / 16) * 16) * 4
                          : ((size_t)(a->M - 1) * 2 * a->lda + (size_t)((a->K + 15) / 16) * 32) * 2;
  const size_t ext_b = km ? ((size_t)(a->K - 1) * a->ldb + (size_t)((a->N + 15) / 16) * 16) * 4
                          : ((size_t)(a->N - 1) * 2 * a->ldb + (size_t)((a->K + 15) / 16) * 32) * 2;
  // (32-bit offsets; a tile's rows past the operand are computed before they are masked)
  const size_t lim = ((size_t)1 << 32) - ((size_t)1 << 20);
  ASR_CHECK_ARG(ext_a + (size_t)1024 * a->lda * 4 < lim && ext_b + (size_t)1024 * a->ldb * 4 < lim,
                "gemm_hl: operand larger than 4 GiB");
  int kps = 0;
  const int splits = hl_splits(a, &kps);
  Epilogue ep;
  ep.C = a->C; ep.ldc = a->ldc; ep.alpha = a->alpha; ep.beta = a->beta; ep.bias = a->bias;
  ep.c_scale = a->c_scale; ep.c_period = a->c_scale_period > 0 ? a->c_scale_period : 1;
  ep.c_ld = a->c_scale_ld; ep.partial = nullptr; ep.clamp_hi = 0.f;
  if (splits > 1) {
    const size_t need = (size_t)splits * (a->batch > 1 ? a->batch : 1) * a->M * a->N * sizeof(float);
    if (!workspace || ws_bytes < need) {
      asr_set_error("gemm_hl: split-K workspace %zu < %zu bytes", ws_bytes, need);
      return ASR_ERR_WORKSPACE;
    }
    ep.partial = reinterpret_cast<float*>(workspace);
  }
  if (a->clamp_hi > 0.f) {
    ASR_CHECK_ARG(a->a_seg_k > 0 && a->beta == 0.f && !a->c_scale && splits == 1,
                  "gemm_hl: clamp_hi needs the segmented form, beta = 0, no mask, no split");
    ep.clamp_hi = a->clamp_hi;
  }
  HlSeg seg;
  seg.magic = 0u;
  seg.batch = 0u;
  for (int i = 0; i < 16; ++i) seg.adj[i] = 0u;
  size_t ext_a_seg = ext_a;
  const int nbatch = a->batch > 1 ? a->batch : 1;
  if (nbatch > 1) {
    // batch of k_major GEMMs sharing B: C_b = A_b^T B, A_b = a_hl shifted by a_batch_row[b] rows
    ASR_CHECK_ARG(km && nbatch <= 16 && a->a_seg_k == 0 && a->beta == 0.f && !a->bias && !a->c_scale,
                  "gemm_hl: a batch needs the k_major form, <= 16 members, no beta / bias / mask");
    long long max_row = 0;
    for (int b = 0; b < nbatch; ++b) {
      ASR_CHECK_ARG(a->a_batch_row[b] >= 0, "gemm_hl: negative batch row shift");
      if (a->a_batch_row[b] > max_row) max_row = a->a_batch_row[b];
      seg.adj[b] = (unsigned)((unsigned long long)a->a_batch_row[b] * (unsigned long long)a->lda * 4ull);
    }
    seg.batch = (unsigned)nbatch;
    ext_a_seg = ext_a + (size_t)max_row * a->lda * 4;
    ASR_CHECK_ARG(ext_a_seg + (size_t)1024 * a->lda * 4 < lim, "gemm_hl: operand larger than 4 GiB");
  }
  if (a->a_seg_k > 0) {
    // A's reduction range = n segments of a_seg_k indices, segment i shifted by a_seg_row[i]
    // plane rows (>= 0: the caller points a_hl at the lowest row any segment reads)
    ASR_CHECK_ARG(!km && a->a_seg_k % HBK == 0 && a->K % a->a_seg_k == 0 &&
                  a->K / a->a_seg_k <= 16 && a->lda >= a->a_seg_k,
                  "gemm_hl: segments need the row-major form, a_seg_k %% 32 == 0, K = n * a_seg_k, n <= 16");
    const int nseg = a->K / a->a_seg_k, sps = a->a_seg_k / HBK;
    seg.magic = (1u << 20) / (unsigned)sps + 1u;
    for (int g = 0; g < a->K / HBK; ++g)
      ASR_CHECK_ARG((int)(((unsigned)g * seg.magic) >> 20) == g / sps, "gemm_hl: segment magic");
    long long max_row = 0;
    for (int i = 0; i < nseg; ++i) {
      ASR_CHECK_ARG(a->a_seg_row[i] >= 0, "gemm_hl: negative segment row shift");
      if (a->a_seg_row[i] > max_row) max_row = a->a_seg_row[i];
      // bytes: the segment's row shift minus what the running reduction offset has advanced
      const long long adj = a->a_seg_row[i] * (long long)a->lda * 4 - (long long)i * a->a_seg_k * 4;
      seg.adj[i] = (unsigned)(unsigned long long)adj;
    }
    ext_a_seg = ((size_t)(a->M - 1 + max_row) * 2 * a->lda + (size_t)(a->a_seg_k / 16) * 32) * 2;
    ASR_CHECK_ARG(ext_a_seg + (size_t)1024 * a->lda * 4 < lim, "gemm_hl: operand larger than 4 GiB");
  }
  HlSrc A, B;
  A.p = reinterpret_cast<const _Float16*>(a->a_hl);
  A.ld = a->lda; A.rows = a->M; A.extent = (unsigned)ext_a_seg;
  B.p = reinterpret_cast<const _Float16*>(a->b_hl);
  B.ld = a->ldb; B.rows = a->N; B.extent = (unsigned)ext_b;
  // tile: 256 x 256 (512 threads, 128 KB LDS) for outputs of at least that size, else 128 x 128
  // (256 threads, 64 KB; asr_gemm_hl_args.tile = 128 forces it)
  const bool big = a->tile != 128 && a->M >= 256 && a->N >= 256;
  const int tl = big ? 256 : 128;
  const size_t shm = km ? (size_t)2 * 2 * (tl / 16) * 2 * kSubtile
                        : (size_t)2 * 4 * (tl * 32 + 32) * sizeof(_Float16);
  const int total = ((a->M + tl - 1) / tl) * ((a->N + tl - 1) / tl) * splits * nbatch;
  const bool segd = seg.magic != 0u;
  // kernel variants: [tile 256 / 128][row-major / k-major / row-major with a segmented A]
  typedef void (*hlx_t)(HlSrc, HlSrc, int, int, int, int, int, Epilogue, const float*, const float*,
                        HlSeg);
  static const hlx_t kern[2][3] = {
      {gemm_hlx_kernel<4, 4, 2, false>, gemm_hlx_kernel<4, 4, 2, true>,
       gemm_hlx_kernel<4, 4, 2, false, true>},
      {gemm_hlx_kernel<2, 2, 2, false>, gemm_hlx_kernel<2, 2, 2, true>,
       gemm_hlx_kernel<2, 2, 2, false, true>}};
  static bool attr_done[2][3] = {{false, false, false}, {false, false, false}};
  const int vi = big ? 0 : 1, vj = km ? 1 : (segd ? 2 : 0);
  if (!attr_done[vi][vj]) {
    ASR_CHECK_HIP(hipFuncSetAttribute((const void*)kern[vi][vj],
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_done[vi][vj] = true;
  }
  // the persistent form (gemm_hlp_kernel): plain row-major launches of the big tile with at
  // least two tiles per CU and five slabs per tile; ASR_GEMM_PERSIST=0 keeps the plain kernel
  // (bit-identical results: the switch changes time only)
  unsigned* ctl = nullptr;
  int grid_p = 0;
  if (!km && !segd && splits == 1 && nbatch == 1 && big && a->K >= 5 * HBK && hlp_enabled()) {
    const int ncu = hlp_num_cu();
    grid_p = ncu / 8 * 8;
    if (grid_p >= 8 && total >= 2 * grid_p) ctl = hlp_control_block();
  }
  if (ctl) {
    const size_t shm_p = shm + 16;
    static bool attr_p = false;
    if (!attr_p) {
      ASR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_hlp_kernel<4, 4, 2>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_p));
      attr_p = true;
    }
    hipLaunchKernelGGL((gemm_hlp_kernel<4, 4, 2>), dim3(grid_p), dim3(512), shm_p, stream, A, B,
                       a->M, a->N, a->K, ep, a->a_scale, a->b_scale, ctl);
  } else {
    hipLaunchKernelGGL(kern[vi][vj], dim3(total), dim3(big ? 512 : 256), shm, stream, A, B, a->M,
                       a->N, a->K, kps, splits, ep, a->a_scale, a->b_scale, seg);
  }
  ASR_CHECK_LAUNCH();
  if (splits > 1) {
    Epilogue ep2 = ep;
    ep2.partial = nullptr;
    lds_ballast((const void*)gemm_splitk_reduce4_kernel);
    launch_splitk_reduce(reinterpret_cast<const float*>(workspace), splits, a->M * nbatch, a->N,
                         ep2, (unsigned)lds_ballast((const void*)gemm_splitk_reduce_kernel), stream);
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}

extern "C" int asr_absmax(const float* x, int64_t n, float* out, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(x && out && n > 0, "absmax: bad arguments");
  ASR_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "absmax: x must be 16-byte aligned");
  ASR_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), stream));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, (size_t)n,
                     reinterpret_cast<unsigned*>(out));
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

extern "C" size_t asr_colsum_workspace_bytes(int M, int N) {
  return asr_align_up((size_t)colsum_slices(M) * N * sizeof(double), 256);
}

extern "C" int asr_colsum(const float* X, int M, int N, int ldx, float* out, float beta,
                          void* workspace, size_t ws_bytes, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(X && out && workspace && M > 0 && N > 0 && ldx >= N, "colsum: bad arguments");
  if (ws_bytes < asr_colsum_workspace_bytes(M, N)) {
    asr_set_error("colsum: workspace too small");
    return ASR_ERR_WORKSPACE;
  }
  const int rs = colsum_slices(M);
  const int rows_per_slice = (M + rs - 1) / rs;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, rs), dim3(256),
                     lds_ballast((const void*)colsum_partial_kernel), stream, X, M, N,
                     ldx, rows_per_slice, partial);
  ASR_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64), dim3(256),
                     lds_ballast((const void*)colsum_final_kernel), stream, partial,
                     rs, N, out, beta);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}
