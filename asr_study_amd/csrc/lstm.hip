// K5 recurrent LSTM sequence kernels (forward + BPTT), both directions of one
// Bidirectional layer per call -- gfx950.
//
// Replaces core/layers.py:432-469 (LSTM.step, iterated T times by Keras K.rnn in a
// tf.while_loop, once per direction) and its tf.gradients.  The input projection
// x@W+b is hoisted out of the loop (gemm.hip); what remains per step is
//     z = zx_t + (h_{t-1} (.) B_U) @ U ;  i,f,o = hard_sigmoid ; g = tanh
//     c = f*c + i*g ; h = o*tanh(c)
// i.e. a (16 x H)x(H x 4H) product per batch tile that cannot start before the
// previous step has finished: a latency problem, not a throughput one.
//
// Design (CDNA4):
//  * A layer is a set of independent CHAINS (direction, 16-row batch tile).  A
//    chain is split over workgroups by OUTPUT columns; each wave keeps its slice
//    of U stationary in VGPRs as the MFMA A-operand for the whole sequence
//    (v_mfma_f32_16x16x4_f32: exact fp32, C/D layout row = 4*(lane>>4)+reg,
//    col = lane&15, so with columns ordered unit*4+gate every lane ends up with
//    the four gates of ONE (unit, sample) -> gate math is lane-local, no LDS).
//  * K is split over the KS waves of a workgroup (short dependent MFMA chain),
//    partial sums meet in LDS (one barrier per step).
//  * Workgroups of a chain exchange h_t through 8-byte {value, step-tag} granules
//    written with ONE relaxed agent-scope (sc1) store each and polled with relaxed
//    agent-scope loads: the data is its own flag, no fences, placement independent
//    (MI355X guide, Guideline 16 / R2).  Two slots (step parity) suffice because a
//    producer can only be one step ahead of its slowest consumer.
//  * Every spin is bounded by the wall clock; a give-up is recorded in the
//    workspace status word and the kernel runs to completion without polling.
//  * mode 1 (one launch per time step, state re-read from the slabs) is the
//    always-safe fallback with the same arithmetic.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned long long u64;

struct LstmParams {
  int T, n_pad, H, NB;
  int KS, R, Kpad;
  int s_begin, s_count;
  int chain_begin;
  const float* U;
  const float* mask_u;
  const float* zx;
  float* y;
  float* cell;
  float* gates;
  const float* dy;
  float* dz;
  float* dc_state;
  u64* gran;
  int* status;
};

constexpr long long kSpinTicks = 60LL * 1000 * 1000;   // 0.6 s of the 100 MHz wall clock

__device__ __forceinline__ float hard_sigmoid(float x) {
  return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f);
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = (e^{2x}-1)/(e^{2x}+1); |abs err| ~ 1e-7, saturates cleanly.
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return __fdividef(e - 1.f, e + 1.f);
}

// Polls R granules (this lane's K slice) until every tag equals `want`.
template <int NK>
__device__ __forceinline__ void gather_granules(const u64* src, int R, int kvalid,
                                                unsigned want, float (&hv)[NK], bool& dead,
                                                int* status) {
  long long t0 = 0;
  bool timing = false;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      if (kk < R && kk < kvalid) {
        const u64 x = __hip_atomic_load(src + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hv[kk] = __uint_as_float((unsigned)x);
        ok &= ((unsigned)(x >> 32) == want);
      } else {
        hv[kk] = 0.f;
      }
    }
    if (__all(ok ? 1 : 0) || dead) return;
    if (!timing) { t0 = wall_clock64(); timing = true; }
    else if (wall_clock64() - t0 > kSpinTicks) {
      dead = true;
      if ((threadIdx.x & 63) == 0) atomicExch(status, 1);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ---------------------------------------------------------------------------
// forward.  work unit = (chain, group of 4 hidden units); block = 64*KS threads.
template <int NK>
__global__ void __launch_bounds__(NK >= 64 ? 256 : NK >= 32 ? 512 : 1024)
lstm_fwd_kernel(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float4 red[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int UG = H >> 2;
  const int ug = blockIdx.x % UG;
  const int chain = p.chain_begin + blockIdx.x / UG;
  const int dir = chain / p.NB, bt = chain % p.NB;
  const int n = bt * 16 + nl;
  const int u = 4 * ug + g;
  const int R = p.R;
  const int kbase = (w * 4 + g) * R;
  const int kvalid = H - kbase;                    // k = kbase+kk valid iff kk < kvalid
  const int KS = p.KS;

  float uf[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    uf[kk] = (kk < R && kk < kvalid)
                 ? p.U[((size_t)(dir * H + kbase + kk)) * H4 + 16 * ug + nl]
                 : 0.f;
  }
  float mask = 1.f;
  if (p.mask_u) mask = p.mask_u[((size_t)dir * p.n_pad + n) * H + u];
  float c = 0.f;
  bool dead = false;
  u64* gch = p.gran + (size_t)chain * 2 * 16 * p.Kpad;
  const int s_end = p.s_begin + p.s_count;

  float4 zx_next = make_float4(0.f, 0.f, 0.f, 0.f);
  if (w == 0) {
    const int t0 = dir == 0 ? p.s_begin : p.T - 1 - p.s_begin;
    zx_next = *reinterpret_cast<const float4*>(
        p.zx + (((size_t)t0 * p.n_pad + n) * 2 + dir) * H4 + 4 * u);
  }
  for (int s = p.s_begin; s < s_end; ++s) {
    const int t = dir == 0 ? s : p.T - 1 - s;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const float4 zx4 = zx_next;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      float hv[NK];
      if (s == p.s_begin) {
        // (re)start of a launch: state comes from the slabs
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          float v = 0.f;
          if (kk < R && kk < kvalid) {
            const int k = kbase + kk;
            v = p.y[((size_t)tp * p.n_pad + n) * H2 + dir * H + k];
            if (p.mask_u) v *= p.mask_u[((size_t)dir * p.n_pad + n) * H + k];
          }
          hv[kk] = v;
        }
        if (w == 0) c = p.cell[(((size_t)tp * p.n_pad + n) * 2 + dir) * H + u];
      } else {
        gather_granules<NK>(gch + (size_t)((s - 1) & 1) * 16 * p.Kpad + (size_t)nl * p.Kpad + kbase,
                            R, kvalid, (unsigned)s, hv, dead, p.status);
      }
      // prefetch next step's input projection behind the MFMA chain
      if (w == 0 && s + 1 < s_end) {
        const int tn = dir == 0 ? t + 1 : t - 1;
        zx_next = *reinterpret_cast<const float4*>(
            p.zx + (((size_t)tn * p.n_pad + n) * 2 + dir) * H4 + 4 * u);
      }
#pragma unroll
      for (int kk = 0; kk < NK; kk += 2) {
        if (kk < R) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk], hv[kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk + 1], hv[kk + 1], acc1, 0, 0, 0);
        }
      }
    } else if (w == 0 && s + 1 < s_end) {
      const int tn = dir == 0 ? t + 1 : t - 1;
      zx_next = *reinterpret_cast<const float4*>(
          p.zx + (((size_t)tn * p.n_pad + n) * 2 + dir) * H4 + 4 * u);
    }
    f32x4 a = acc0 + acc1;
    if (KS > 1) {
      if (w > 0) red[(w - 1) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
      __syncthreads();
      if (w == 0) {
        for (int ww = 0; ww < KS - 1; ++ww) {
          const float4 r = red[ww * 64 + lane];
          a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
        }
      }
    }
    if (w == 0) {
      const float gi = hard_sigmoid(a[0] + zx4.x);
      const float gf = hard_sigmoid(a[1] + zx4.y);
      const float gg = fast_tanh(a[2] + zx4.z);
      const float go = hard_sigmoid(a[3] + zx4.w);
      c = gf * c + gi * gg;
      const float h = go * fast_tanh(c);
      if (s + 1 < p.T) {
        const u64 val = ((u64)(unsigned)(s + 1) << 32) | (u64)__float_as_uint(h * mask);
        __hip_atomic_store(gch + (size_t)(s & 1) * 16 * p.Kpad + (size_t)nl * p.Kpad + u, val,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const size_t row = (size_t)t * p.n_pad + n;
      p.y[row * H2 + dir * H + u] = h;
      p.cell[(row * 2 + dir) * H + u] = c;
      *reinterpret_cast<float4*>(p.gates + (row * 2 + dir) * H4 + 4 * u) =
          make_float4(gi, gf, gg, go);
    }
    if (KS > 1) __syncthreads();   // red[] is reused next step
  }
}

// ---------------------------------------------------------------------------
// backward (BPTT).  work unit = (chain, group of 16 hidden units); the K axis is
// the 4H gate-gradient vector dz of the step processed before.
template <int NK>
__global__ void __launch_bounds__(NK >= 64 ? 256 : NK >= 32 ? 512 : 1024)
lstm_bwd_kernel(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float4 red[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int OG = (H + 15) >> 4;
  const int og = blockIdx.x % OG;
  const int chain = p.chain_begin + blockIdx.x / OG;
  const int dir = chain / p.NB, bt = chain % p.NB;
  const int n = bt * 16 + nl;
  const int u0 = 16 * og + 4 * g;               // this lane owns units u0..u0+3
  const bool uvalid = u0 < H;                   // H % 4 == 0: all four or none
  const int R = p.R;
  const int jbase = (w * 4 + g) * R;
  const int jvalid = H4 - jbase;
  const int KS = p.KS;

  float uf[NK];
  {
    const int krow = 16 * og + nl;              // A row i = lane & 15
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      uf[kk] = (kk < R && kk < jvalid && krow < H)
                   ? p.U[((size_t)(dir * H + krow)) * H4 + jbase + kk]
                   : 0.f;
    }
  }
  float4 mask = make_float4(1.f, 1.f, 1.f, 1.f);
  if (p.mask_u && uvalid)
    mask = *reinterpret_cast<const float4*>(p.mask_u + ((size_t)dir * p.n_pad + n) * H + u0);
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  bool dead = false;
  u64* gch = p.gran + (size_t)chain * 2 * 16 * p.Kpad;
  const int s_end = p.s_begin + p.s_count;

  if (w == 0 && p.s_begin > 0 && uvalid) {
    const float4 d4 = *reinterpret_cast<const float4*>(
        p.dc_state + ((size_t)dir * p.n_pad + n) * H + u0);
    dc[0] = d4.x; dc[1] = d4.y; dc[2] = d4.z; dc[3] = d4.w;
  }

  for (int s = p.s_begin; s < s_end; ++s) {
    // reverse of the forward processing order
    const int t = dir == 0 ? p.T - 1 - s : s;
    const int tq = dir == 0 ? t + 1 : t - 1;    // step processed just before (s-1)
    const int tc = dir == 0 ? t - 1 : t + 1;    // forward-order predecessor (c_{prev})
    const bool has_cprev = (s + 1 < p.T);
    // ---- issue this step's independent loads first (wave 0)
    float4 dy4 = make_float4(0.f, 0.f, 0.f, 0.f), c4 = dy4, cp4 = dy4;
    float4 gt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gt[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w == 0 && uvalid) {
      const size_t row = (size_t)t * p.n_pad + n;
      dy4 = *reinterpret_cast<const float4*>(p.dy + row * H2 + dir * H + u0);
      c4 = *reinterpret_cast<const float4*>(p.cell + (row * 2 + dir) * H + u0);
      if (has_cprev)
        cp4 = *reinterpret_cast<const float4*>(
            p.cell + (((size_t)tc * p.n_pad + n) * 2 + dir) * H + u0);
      const float4* gp = reinterpret_cast<const float4*>(p.gates + (row * 2 + dir) * H4 + 4 * u0);
#pragma unroll
      for (int r = 0; r < 4; ++r) gt[r] = gp[r];
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      float hv[NK];
      if (s == p.s_begin) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          hv[kk] = (kk < R && kk < jvalid)
                       ? p.dz[(((size_t)tq * p.n_pad + n) * 2 + dir) * H4 + jbase + kk]
                       : 0.f;
        }
      } else {
        gather_granules<NK>(gch + (size_t)((s - 1) & 1) * 16 * p.Kpad + (size_t)nl * p.Kpad + jbase,
                            R, jvalid, (unsigned)s, hv, dead, p.status);
      }
#pragma unroll
      for (int kk = 0; kk < NK; kk += 2) {
        if (kk < R) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk], hv[kk], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk + 1], hv[kk + 1], acc1, 0, 0, 0);
        }
      }
    }
    f32x4 a = acc0 + acc1;
    if (KS > 1) {
      if (w > 0) red[(w - 1) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
      __syncthreads();
      if (w == 0) {
        for (int ww = 0; ww < KS - 1; ++ww) {
          const float4 r = red[ww * 64 + lane];
          a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
        }
      }
    }
    if (w == 0 && uvalid) {
      const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
      const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
      const float cpv[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
      const float mv[4] = {mask.x, mask.y, mask.z, mask.w};
      const size_t row = (size_t)t * p.n_pad + n;
      float4* dzp = reinterpret_cast<float4*>(p.dz + (row * 2 + dir) * H4 + 4 * u0);
      u64* gdst = gch + (size_t)(s & 1) * 16 * p.Kpad + (size_t)nl * p.Kpad + 4 * u0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gi = gt[r].x, gf = gt[r].y, gg = gt[r].z, go = gt[r].w;
        const float dh = dyv[r] + mv[r] * a[r];
        const float tch = fast_tanh(cv[r]);
        const float d_o = dh * tch;
        const float dcc = dc[r] + dh * go * (1.f - tch * tch);
        const float d_i = dcc * gg, d_g = dcc * gi, d_f = dcc * cpv[r];
        dc[r] = dcc * gf;
        const float zi = d_i * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
        const float zf = d_f * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
        const float zg = d_g * (1.f - gg * gg);
        const float zo = d_o * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
        if (s + 1 < p.T) {
          const u64 tag = (u64)(unsigned)(s + 1) << 32;
          __hip_atomic_store(gdst + 4 * r + 0, tag | (u64)__float_as_uint(zi), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gdst + 4 * r + 1, tag | (u64)__float_as_uint(zf), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gdst + 4 * r + 2, tag | (u64)__float_as_uint(zg), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gdst + 4 * r + 3, tag | (u64)__float_as_uint(zo), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        dzp[r] = make_float4(zi, zf, zg, zo);
      }
    }
    if (KS > 1) __syncthreads();
  }
  if (w == 0 && uvalid && p.dc_state) {
    *reinterpret_cast<float4*>(p.dc_state + ((size_t)dir * p.n_pad + n) * H + u0) =
        make_float4(dc[0], dc[1], dc[2], dc[3]);
  }
}

// ---------------------------------------------------------------------------
struct Plan {
  int KS, R, NK, Kpad;
  int units;           // workgroups per chain
  int chains_per_launch;
};

int even_up(int x) { return (x + 1) & ~1; }

typedef void (*kern_t)(LstmParams);

kern_t pick_kernel(bool bwd, int NK) {
  if (!bwd) {
    switch (NK) {
      case 8: return lstm_fwd_kernel<8>;
      case 16: return lstm_fwd_kernel<16>;
      case 32: return lstm_fwd_kernel<32>;
      default: return lstm_fwd_kernel<64>;
    }
  }
  switch (NK) {
    case 8: return lstm_bwd_kernel<8>;
    case 16: return lstm_bwd_kernel<16>;
    case 32: return lstm_bwd_kernel<32>;
    default: return lstm_bwd_kernel<64>;
  }
}

int nk_for(int R) { return R <= 8 ? 8 : R <= 16 ? 16 : R <= 32 ? 32 : 64; }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Chooses the K split so that every workgroup of every chain is co-resident.
int make_plan(const asr_lstm_args* a, bool bwd, Plan* out) {
  const int H = a->H;
  const int Ktot = bwd ? 4 * H : H;
  const int units = bwd ? (H + 15) / 16 : H / 4;
  const int chains = 2 * (a->n_pad / 16);
  int dev = 0, num_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ASR_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ASR_ERR_LAUNCH;
  num_cu = prop.multiProcessorCount;
  const int target_r = env_int(bwd ? "ASR_LSTM_BWD_R" : "ASR_LSTM_FWD_R", bwd ? 32 : 16);
  const int forced_ks = env_int(bwd ? "ASR_LSTM_BWD_KS" : "ASR_LSTM_FWD_KS", 0);
  Plan best; best.KS = 0;
  for (int KS = 16; KS >= 1; KS >>= 1) {
    if (forced_ks && KS != forced_ks) continue;
    const int R = even_up((Ktot + 4 * KS - 1) / (4 * KS));
    if (R > 64) continue;                       // does not fit the register file
    if (R < 2) continue;
    const int NK = nk_for(R);
    if (NK >= 64 && KS > 4) continue;           // launch bounds 256 / 512 / 1024
    if (NK >= 32 && KS > 8) continue;
    if (!forced_ks && R < target_r && KS > 1) {
      // finer than requested: only take it if nothing coarser is feasible
      const int Rc = even_up((Ktot + 2 * KS - 1) / (2 * KS));   // R at KS/2
      if (Rc <= 64) continue;
    }
    int occ = 0;
    const size_t shm = (size_t)(KS > 1 ? (KS - 1) : 0) * 64 * sizeof(float4);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)pick_kernel(bwd, NK),
                                                     64 * KS, shm) != hipSuccess)
      return ASR_ERR_LAUNCH;
    // the occupancy API can over-report by one block per CU (guide): keep a margin
    int per_cu = occ > 1 ? occ - 1 : occ;
    const long cap = (long)per_cu * num_cu;
    if (cap < units) continue;                  // even one chain would not fit
    Plan pl;
    pl.KS = KS; pl.R = R; pl.NK = NK; pl.Kpad = 4 * KS * R; pl.units = units;
    long cpl = cap / units;
    if (cpl > chains) cpl = chains;
    pl.chains_per_launch = (int)cpl;
    if (best.KS == 0 || pl.chains_per_launch > best.chains_per_launch) best = pl;
    if (pl.chains_per_launch == chains) { best = pl; break; }
  }
  if (best.KS == 0) {
    asr_set_error("lstm: no co-resident launch plan for H=%d n_pad=%d (CUs=%d)", H, a->n_pad,
                  num_cu);
    return ASR_ERR_RESIDENCY;
  }
  *out = best;
  return ASR_OK;
}

constexpr size_t kStatusBytes = 256;

size_t gran_bytes(const asr_lstm_args* a, bool bwd) {
  const int Ktot = bwd ? 4 * a->H : a->H;
  const size_t kpad_max = (size_t)Ktot + 4 * 16 * 2 + 64;   // Kpad <= K + 4*KS*2
  const size_t chains = (size_t)2 * (a->n_pad / 16);
  return asr_align_up(chains * 2 * 16 * kpad_max * sizeof(u64), 256);
}

int run(const asr_lstm_args* a, bool bwd, void* workspace, size_t ws_bytes,
        hipStream_t stream) {
  ASR_CHECK_ARG(a && a->U && workspace, "lstm: null pointer");
  ASR_CHECK_ARG(a->T > 0 && a->n_pad > 0 && a->n_pad % 16 == 0 && a->H >= 4 && a->H % 4 == 0,
                "lstm: need n_pad %% 16 == 0 and H %% 4 == 0 (T=%d n_pad=%d H=%d)", a->T,
                a->n_pad, a->H);
  if (!bwd) ASR_CHECK_ARG(a->zx && a->y && a->cell && a->gates, "lstm fwd: null slab");
  else ASR_CHECK_ARG(a->dy && a->dz && a->cell && a->gates, "lstm bwd: null slab");
  const size_t need = asr_lstm_workspace_bytes(a, bwd ? 1 : 0);
  if (ws_bytes < need) {
    asr_set_error("lstm: workspace %zu < %zu bytes", ws_bytes, need);
    return ASR_ERR_WORKSPACE;
  }
  Plan pl;
  const int rc = make_plan(a, bwd, &pl);
  if (rc != ASR_OK) return rc;
  char* ws = reinterpret_cast<char*>(workspace);
  const size_t gb = gran_bytes(a, bwd);
  LstmParams p;
  p.T = a->T; p.n_pad = a->n_pad; p.H = a->H; p.NB = a->n_pad / 16;
  p.KS = pl.KS; p.R = pl.R; p.Kpad = pl.Kpad;
  p.U = a->U; p.mask_u = a->mask_u; p.zx = a->zx; p.y = a->y; p.cell = a->cell;
  p.gates = a->gates; p.dy = a->dy; p.dz = a->dz;
  p.status = reinterpret_cast<int*>(ws);
  p.gran = reinterpret_cast<u64*>(ws + kStatusBytes);
  p.dc_state = reinterpret_cast<float*>(ws + kStatusBytes + gb);
  ASR_CHECK_HIP(hipMemsetAsync(ws, 0, kStatusBytes + gb, stream));
  const int chains = 2 * p.NB;
  const size_t shm = (size_t)(pl.KS > 1 ? pl.KS - 1 : 0) * 64 * sizeof(float4);
  kern_t k = pick_kernel(bwd, pl.NK);
  const bool stepwise = a->mode == 1;
  const int steps_per_launch = stepwise ? 1 : a->T;
  for (int cb = 0; cb < chains; cb += pl.chains_per_launch) {
    const int nch = (chains - cb) < pl.chains_per_launch ? (chains - cb) : pl.chains_per_launch;
    for (int s0 = 0; s0 < a->T; s0 += steps_per_launch) {
      p.chain_begin = cb;
      p.s_begin = s0;
      p.s_count = (a->T - s0) < steps_per_launch ? (a->T - s0) : steps_per_launch;
      hipLaunchKernelGGL(k, dim3(nch * pl.units), dim3(64 * pl.KS), shm, stream, p);
      ASR_CHECK_LAUNCH();
    }
  }
  return ASR_OK;
}

}  // namespace

extern "C" size_t asr_lstm_workspace_bytes(const asr_lstm_args* a, int backward) {
  if (!a || a->n_pad <= 0 || a->H <= 0) return 0;
  return kStatusBytes + gran_bytes(a, backward != 0) +
         asr_align_up((size_t)2 * a->n_pad * a->H * sizeof(float), 256);
}

extern "C" int asr_lstm_seq_fwd(const asr_lstm_args* a, void* workspace, size_t ws_bytes,
                                asr_stream_t stream) {
  return run(a, false, workspace, ws_bytes, (hipStream_t)stream);
}

extern "C" int asr_lstm_seq_bwd(const asr_lstm_args* a, void* workspace, size_t ws_bytes,
                                asr_stream_t stream) {
  return run(a, true, workspace, ws_bytes, (hipStream_t)stream);
}

// Synchronises `stream`, then reports whether a persistent kernel that used this
// workspace abandoned a bounded spin.
extern "C" int asr_lstm_status(const void* workspace, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int st = 0;
  ASR_CHECK_HIP(hipMemcpyAsync(&st, workspace, sizeof(int), hipMemcpyDeviceToHost, stream));
  ASR_CHECK_HIP(hipStreamSynchronize(stream));
  if (st != 0) {
    asr_set_error("lstm: persistent kernel timed out waiting for a peer workgroup");
    return ASR_ERR_TIMEOUT;
  }
  return ASR_OK;
}

extern "C" int asr_lstm_plan(const asr_lstm_args* a, int backward, int* ks, int* r,
                             int* blocks, int* chains_per_launch) {
  Plan pl;
  const int rc = make_plan(a, backward != 0, &pl);
  if (rc != ASR_OK) return rc;
  if (ks) *ks = pl.KS;
  if (r) *r = pl.R;
  if (blocks) *blocks = pl.units * pl.chains_per_launch;
  if (chains_per_launch) *chains_per_launch = pl.chains_per_launch;
  return ASR_OK;
}
