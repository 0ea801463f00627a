// K7 CTC loss + gradient, K8 greedy decode -- gfx950.
//
// Replaces core/ctc_utils.py:60-70 (tf.nn.ctc_loss) and :42
// (tf.nn.ctc_greedy_decoder) of the reference.  Algorithm: Graves 2006 eq. 6-16
// in log space, blank = C-1, beta excluding the emission at t (TF convention),
// see oracle/ctc.py for the restatement this is tested against.
//
// Mapping to CDNA4.  The alpha (and beta) recursion is a 999-deep dependent
// chain per utterance, so the design minimises the latency of ONE step rather
// than bytes moved:
//   * one 64-lane wave per (utterance, direction); lane k owns the state PAIR
//     (blank before label k, label k), so the only cross-lane traffic per step is
//     ONE DPP wave-shift of a single register (no LDS, no barrier);
//   * alpha and beta chains of all utterances run concurrently (2N waves);
//   * no log-softmax slab: a fully parallel kernel leaves ONE float per frame (the row's
//     log-sum-exp) and an emission is (logit - lse) * log2(e) where it is used; the per-step
//     emission gathers are software-pipelined several steps ahead of the dependent chain;
//   * alpha~ / beta~ stay on chip: the chains keep their rows in registers and write only a
//     checkpoint every 16 frames (4 MB at 64 x 999 frames instead of 65 MB of rows);
//   * the gradient kernel is fully parallel over (t, n): one wave per frame re-derives its
//     alpha~_t / beta~_t row from the nearest checkpoints (<= 16 recursion steps), class bins
//     in LDS -- HBM traffic of the three kernels: logits read, lse + checkpoints, gradient
//     written (algorithmic: 2 T N C 4 B);
//   * float32 log-space values of magnitude T*ln(C) ~ 3300 would only resolve
//     posteriors to ~1e-3 (TF's float kernel has that property); here every 4
//     steps the row is re-centred on its maximum (one wave-max; the removed amount
//     is accumulated in float64 for the loss), so alpha~/beta~ stay O(10), and the
//     gradient kernel normalises each frame's posteriors by their own sum (= Z
//     exactly), which cancels the drift accumulated along the recursion.
#include "common.h"

namespace {

constexpr float kNegInf = -__builtin_huge_valf();

// ---- DPP wave shifts (gfx9 encodings: wave_shr:1 = 0x138, wave_shl:1 = 0x130).
__device__ __forceinline__ float wave_shift_up(float v, float fill) {
  // lane i receives lane i-1; lane 0 receives `fill`.
  int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v),
                                      0x138, 0xf, 0xf, false);
  return __int_as_float(r);
}
__device__ __forceinline__ float wave_shift_down(float v, float fill) {
  // lane i receives lane i+1; lane 63 receives `fill`.
  int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v),
                                      0x130, 0xf, 0xf, false);
  return __int_as_float(r);
}

// The recursions run in BASE-2 log space: log-probabilities are stored as log2 p, so a
// log-sum-exp is max + v_log_f32(v_exp_f32(..) + ..) on the native base-2 transcendentals
// (no scaling multiplies, and the sum lies in [1, 3], so no denormal fix-up either).
constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2 = 0.6931471805599453;
// Two transcendentals and no branch: log2(2^a + 2^b) = max + log2(1 + 2^(min - max)); with both
// arguments -inf the difference is taken against a finite floor (-inf - floor = -inf, 2^-inf = 0)
// and the sum max + 0 is -inf again.
__device__ __forceinline__ float lse2_b2(float a, float b) {
  const float m = fmaxf(a, b);
  const float d = fminf(a, b) - fmaxf(m, -3.0e38f);
  return m + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(d));
}

// Three-way sum with three transcendentals: the largest term is 1, the other two are the median
// and the minimum (v_max3 / v_med3 / v_min3).
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  const float f = fmaxf(m, -3.0e38f);
  const float md = __builtin_amdgcn_fmed3f(a, b, c) - f;
  const float mn = fminf(fminf(a, b), c) - f;
  return m + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(md) + __builtin_amdgcn_exp2f(mn));
}

// ---------------------------------------------------------------------------
// log-sum-exp of every (t, n) row of the logits (natural log): the only thing the other
// kernels need of the softmax -- an emission is (logit - lse) * log2(e), computed where it is
// used, so no log-softmax slab is ever written.  One wave per row.
__global__ void __launch_bounds__(256)
ctc_lse_kernel(const float* __restrict__ logits, float* __restrict__ lse, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + (size_t)row * C;
  float m = kNegInf;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = asr_wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(x[c] - m);
  s = asr_wave_sum(s);
  if (lane == 0) lse[row] = m + __logf(s);
}

// alpha~ / beta~ rows are kept only every kCk frames (checkpoints, 2 x T/kCk x N rows of 512 B:
// 4 MB at cfg3 instead of 65 MB); the gradient kernel, which is parallel over the frames,
// re-derives the row of its frame from the nearest checkpoints (at most kCk recursion steps per
// frame, tens of microseconds for the whole batch).
constexpr int kCk = 16;

// ---- one step of the recursions on a lane's PPL state pairs (sb: blank before label q, sl:
// label q); eb / el: base-2 log emissions of the blank / of this lane's labels at the frame
// the step consumes.
template <int PPL>
struct CtcLane {
  int lab[PPL];
  bool valid[PPL];     // label state exists (q < L)
  bool diffp[PPL];     // label q differs from label q-1 (skip transition legal)
  __device__ __forceinline__ void init(const int* __restrict__ labels, int n, int l_max, int L,
                                       int C, int lane) {
    const int blank = C - 1;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const int q = lane * PPL + p;
      valid[p] = q < L;
      lab[p] = valid[p] ? labels[(size_t)n * l_max + q] : blank;
      const int prev = (q >= 1 && q - 1 < L) ? labels[(size_t)n * l_max + q - 1] : -1;
      diffp[p] = valid[p] && (q == 0 || lab[p] != prev);
      if (lab[p] < 0 || lab[p] >= C) lab[p] = blank;
    }
  }
  // alpha_t from alpha_{t-1} with the emissions of frame t
  __device__ __forceinline__ void alpha_step(float (&sb)[PPL], float (&sl)[PPL], float eb,
                                             const float (&el)[PPL]) const {
    // label state of the previous pair (lane-local, or lane-1's last pair)
    const float lprev = wave_shift_up(sl[PPL - 1], kNegInf);
    float nsb[PPL], nsl[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const float lp1 = (p == 0) ? lprev : sl[p - 1];
      // blank state: from itself or the previous label.  The label state adds itself to the
      // same sum when the skip from the previous label is legal: with several pairs per lane
      // the step is issue-bound and re-uses that sum (4 transcendentals per pair instead of
      // 5); with one pair per lane it is latency-bound and the two sums run side by side.
      const float x = lse2_b2(sb[p], lp1);
      nsb[p] = x + eb;
      const float e = valid[p] ? el[p] : kNegInf;
      if (PPL == 1) nsl[p] = lse3_b2(sl[p], sb[p], diffp[p] ? lp1 : kNegInf) + e;
      else nsl[p] = lse2_b2(sl[p], diffp[p] ? x : sb[p]) + e;
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) { sb[p] = nsb[p]; sl[p] = nsl[p]; }
  }
  // beta_t from beta_{t+1} with the emissions of frame t+1 (beta excludes the emission at t)
  __device__ __forceinline__ void beta_step(float (&sb)[PPL], float (&sl)[PPL], float eb,
                                            const float (&el)[PPL]) const {
    float xb[PPL], xl[PPL], y[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      xb[p] = sb[p] + eb;
      xl[p] = valid[p] ? sl[p] + el[p] : kNegInf;
      // what the PREVIOUS pair's label state may move into from this pair
      y[p] = lse2_b2(xb[p], diffp[p] ? xl[p] : kNegInf);
    }
    const float ynext = wave_shift_down(y[0], kNegInf);
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const float yn = (p == PPL - 1) ? ynext : y[p + 1];
      sb[p] = lse2_b2(xb[p], xl[p]);
      sl[p] = valid[p] ? lse2_b2(xl[p], yn) : kNegInf;
    }
  }
  __device__ __forceinline__ static void recentre(float (&sb)[PPL], float (&sl)[PPL], double* off) {
    float m = kNegInf;
#pragma unroll
    for (int p = 0; p < PPL; ++p) m = fmaxf(m, fmaxf(sb[p], sl[p]));
    m = asr_wave_max(m);
    if (m > kNegInf) {
#pragma unroll
      for (int p = 0; p < PPL; ++p) { sb[p] -= m; sl[p] -= m; }
      if (off) *off += (double)m;
    }
  }
};

// ---------------------------------------------------------------------------
// alpha / beta chains.  PPL = state pairs per lane (labels up to 64*PPL-1).
template <int PPL>
__global__ void __launch_bounds__(64)
ctc_alpha_beta_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                      const int* __restrict__ labels,
                      const int* __restrict__ label_len, const int* __restrict__ seq_len,
                      int T, int N, int n_pad, int C, int l_max,
                      float* __restrict__ alpha_ck, float* __restrict__ beta_ck,
                      double* __restrict__ logz, float* __restrict__ loss, int do_beta) {
  // frames per prefetch group: the emissions of the NEXT group are loaded while this one
  // is processed, so the group must outlast one L2-miss latency (~1-2 us); rows are
  // re-centred once per group
  constexpr int UNR = PPL == 1 ? 16 : (PPL == 2 ? 8 : 4);
  const int lane = threadIdx.x;
  const int n = do_beta ? (blockIdx.x >> 1) : blockIdx.x;
  const int dir = do_beta ? (blockIdx.x & 1) : 0;
  const int blank = C - 1;
  const int L = label_len[n];
  int Tn = seq_len[n];
  Tn = Tn < 1 ? 1 : (Tn > T ? T : Tn);
  const int SP = 2 * 64 * PPL;
  __shared__ float fin[2 * 64 * PPL];
  CtcLane<PPL> ln;
  ln.init(labels, n, l_max, L, C, lane);

  float sb[PPL], sl[PPL];   // blank-state / label-state log values
#pragma unroll
  for (int p = 0; p < PPL; ++p) { sb[p] = kNegInf; sl[p] = kNegInf; }
  double off = 0.0;        // amount removed from the log-space row so far

  const size_t row_stride = (size_t)n_pad * C;
  const float* lg_n = logits + (size_t)n * C;
  const float* lse_n = lse + n;
  const int ngroups = (Tn + UNR - 1) / UNR;
  // raw logits of the blank / this lane's labels and the row's lse of one group of frames
  // (frame f(u) given by the caller); emission = (logit - lse) * log2(e)
  float eb[UNR], el[UNR][PPL];
  auto load_frame = [&](int t, bool virt, float& b, float (&l)[PPL]) {
    const float* r = lg_n + (size_t)t * row_stride;
    const float ls = lse_n[(size_t)t * n_pad];
    const float vb = (r[blank] - ls) * kLog2e;
    b = virt ? 0.f : vb;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const float v = (r[ln.lab[p]] - ls) * kLog2e;
      l[p] = virt ? 0.f : v;
    }
  };

  if (dir == 0) {
    // ----- alpha: virtual state before t=0 is "blank 0 with probability 1".
    if (lane == 0) sb[0] = 0.f;
    auto load_group = [&](int g, float (&b)[UNR], float (&l)[UNR][PPL]) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int t = g * UNR + u; t = t < Tn ? t : Tn - 1;
        load_frame(t, false, b[u], l[u]);
      }
    };
    load_group(0, eb, el);
    for (int g = 0; g < ngroups; ++g) {
      float nb[UNR], nl[UNR][PPL];
      load_group(g + 1 < ngroups ? g + 1 : g, nb, nl);
      CtcLane<PPL>::recentre(sb, sl, &off);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = g * UNR + u;
        if (t < Tn) {
          ln.alpha_step(sb, sl, eb[u], el[u]);
          if (do_beta && (t % kCk) == 0) {       // checkpoint (only the gradient needs them)
            float2* out = reinterpret_cast<float2*>(alpha_ck + ((size_t)(t / kCk) * N + n) * SP) +
                          lane * PPL;
#pragma unroll
            for (int p = 0; p < PPL; ++p) out[p] = make_float2(sb[p], sl[p]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        eb[u] = nb[u];
#pragma unroll
        for (int p = 0; p < PPL; ++p) el[u][p] = nl[u][p];
      }
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      fin[2 * (lane * PPL + p)] = sb[p];
      fin[2 * (lane * PPL + p) + 1] = sl[p];
    }
    __syncthreads();
    if (lane == 0) {
      const float e1 = fin[2 * L];                          // final blank
      const float e2 = L > 0 ? fin[2 * (L - 1) + 1] : kNegInf;  // last label
      const double lz = (double)lse2_b2(e1, e2) + off;     // log2 p(l | x)
      logz[n] = lz;
      loss[n] = (float)(-lz * kLn2);
    }
  } else {
    // ----- beta (excludes the emission at t).  Virtual frame Tn has emission 0
    // and state "final blank" = 0.
    {
      const int qL = L;   // pair index of the final blank
      if (lane == qL / PPL) {
#pragma unroll
        for (int p = 0; p < PPL; ++p) if (p == qL % PPL) sb[p] = 0.f;
      }
    }
    // group g covers t = Tn-1-g*UNR-u ; emissions come from frame t+1.
    auto load_group = [&](int g, float (&b)[UNR], float (&l)[UNR][PPL]) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int t = Tn - 1 - (g * UNR + u); t = t < 0 ? 0 : t;
        const bool virt = (t + 1 >= Tn);
        load_frame(virt ? t : t + 1, virt, b[u], l[u]);
      }
    };
    load_group(0, eb, el);
    for (int g = 0; g < ngroups; ++g) {
      float nb[UNR], nl[UNR][PPL];
      load_group(g + 1 < ngroups ? g + 1 : g, nb, nl);
      CtcLane<PPL>::recentre(sb, sl, nullptr);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int t = Tn - 1 - (g * UNR + u);
        if (t >= 0) {
          ln.beta_step(sb, sl, eb[u], el[u]);
          if ((t % kCk) == 0) {
            float2* out = reinterpret_cast<float2*>(beta_ck + ((size_t)(t / kCk) * N + n) * SP) +
                          lane * PPL;
#pragma unroll
            for (int p = 0; p < PPL; ++p) out[p] = make_float2(sb[p], sl[p]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        eb[u] = nb[u];
#pragma unroll
        for (int p = 0; p < PPL; ++p) el[u][p] = nl[u][p];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// gradient: one wave per (t, n).  The wave re-derives alpha~_t from the checkpoint at
// t0 = kCk floor(t / kCk) (t - t0 forward steps) and beta~_t from the checkpoint at t0 + kCk,
// or from the virtual frame Tn when that lies past the utterance (at most kCk backward steps),
// then writes softmax - posterior.
template <int PPL>
__global__ void __launch_bounds__(256)
ctc_grad_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                const float* __restrict__ alpha_ck, const float* __restrict__ beta_ck,
                const double* __restrict__ logz, const int* __restrict__ labels,
                const int* __restrict__ label_len, const int* __restrict__ seq_len,
                int T, int N, int n_pad, int C, int l_max, float scale,
                float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) float bins_all[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const long pair = (long)blockIdx.x * 4 + w;
  const bool in_range = pair < (long)T * n_pad;
  const int t = in_range ? (int)(pair / n_pad) : 0;
  const int n = in_range ? (int)(pair % n_pad) : 0;
  float* bins = bins_all + (size_t)w * C;
  const int blank = C - 1;
  const int SP = 2 * 64 * PPL;
  bool active = in_range && n < N;
  int L = 0, Tn = 1;
  if (active) {
    Tn = seq_len[n];
    Tn = Tn < 1 ? 1 : (Tn > T ? T : Tn);
    active = t < Tn;
    L = label_len[n];
    const double lz = logz[n];
    if (!(lz > -1.0e300)) active = false;    // infeasible target: zero gradient
  }
  for (int c = lane; c < C; c += 64) bins[c] = 0.f;
  __syncthreads();
  float bsum = 0.f;
  // log-posteriors alpha~ + beta~ of this lane's states; normalised by THEIR sum
  // over all states of the frame (= Z exactly), which cancels the rounding drift
  // accumulated along the 999-step recursions and needs no offsets.
  float vb[PPL], vl[PPL];
  float vmax = kNegInf;
#pragma unroll
  for (int p = 0; p < PPL; ++p) { vb[p] = kNegInf; vl[p] = kNegInf; }
  const size_t row_stride = (size_t)n_pad * C;
  if (active) {                                  // (wave-uniform: one wave = one frame)
    CtcLane<PPL> ln;
    ln.init(labels, n, l_max, L, C, lane);
    const float* lg_n = logits + (size_t)n * C;
    const float* lse_n = lse + n;
    auto emis = [&](int f, float& b, float (&l)[PPL]) {
      const float* r = lg_n + (size_t)f * row_stride;
      const float ls = lse_n[(size_t)f * n_pad];
      b = (r[blank] - ls) * kLog2e;
#pragma unroll
      for (int p = 0; p < PPL; ++p) l[p] = (r[ln.lab[p]] - ls) * kLog2e;
    };
    const int t0 = t - (t % kCk);
    // ---- alpha~_t
    float ab[PPL], al[PPL];
    {
      const float2* a2 = reinterpret_cast<const float2*>(alpha_ck + ((size_t)(t0 / kCk) * N + n) * SP) +
                         lane * PPL;
#pragma unroll
      for (int p = 0; p < PPL; ++p) { const float2 a = a2[p]; ab[p] = a.x; al[p] = a.y; }
      for (int f = t0 + 1; f <= t; ++f) {
        float eb, el[PPL];
        emis(f, eb, el);
        ln.alpha_step(ab, al, eb, el);
      }
    }
    // ---- beta~_t
    float bb[PPL], bl[PPL];
    {
      const int t1 = t0 + kCk;
      int tau;                                   // the row (bb, bl) holds
      if ((t % kCk) == 0 || t1 <= Tn - 1) {
        const int tc = (t % kCk) == 0 ? t : t1;
        const float2* b2 = reinterpret_cast<const float2*>(beta_ck + ((size_t)(tc / kCk) * N + n) * SP) +
                           lane * PPL;
#pragma unroll
        for (int p = 0; p < PPL; ++p) { const float2 v2 = b2[p]; bb[p] = v2.x; bl[p] = v2.y; }
        tau = tc;
      } else {                                   // start from the virtual frame Tn
#pragma unroll
        for (int p = 0; p < PPL; ++p) { bb[p] = kNegInf; bl[p] = kNegInf; }
        if (lane == L / PPL) {
#pragma unroll
          for (int p = 0; p < PPL; ++p) if (p == L % PPL) bb[p] = 0.f;
        }
        tau = Tn;
      }
      for (int f = tau - 1; f >= t; --f) {       // beta_f from beta_{f+1}, emissions of f + 1
        float eb = 0.f, el[PPL];
#pragma unroll
        for (int p = 0; p < PPL; ++p) el[p] = 0.f;
        if (f + 1 < Tn) emis(f + 1, eb, el);
        ln.beta_step(bb, bl, eb, el);
      }
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const int q = lane * PPL + p;
      if (q <= L) {
        vb[p] = ab[p] + bb[p];
        if (q < L) vl[p] = al[p] + bl[p];
        vmax = fmaxf(vmax, fmaxf(vb[p], vl[p]));
      }
    }
  }
  vmax = asr_wave_max(vmax);
  float zsum = 0.f;
  if (active && vmax > kNegInf) {
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      vb[p] = vb[p] > kNegInf ? __builtin_amdgcn_exp2f(vb[p] - vmax) : 0.f;
      vl[p] = vl[p] > kNegInf ? __builtin_amdgcn_exp2f(vl[p] - vmax) : 0.f;
      zsum += vb[p] + vl[p];
    }
  }
  zsum = asr_wave_sum(zsum);
  if (!(zsum > 0.f)) active = false;
  if (active) {
    const float inv = 1.f / zsum;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      const int q = lane * PPL + p;
      if (q <= L) {
        bsum += vb[p] * inv;
        if (q < L && vl[p] > 0.f) {
          int lab = labels[(size_t)n * l_max + q];
          lab = (lab < 0 || lab >= C) ? blank : lab;
          atomicAdd(&bins[lab], vl[p] * inv);
        }
      }
    }
  }
  bsum = asr_wave_sum(bsum);
  __syncthreads();
  if (lane == 0 && active) bins[blank] += bsum;
  __syncthreads();
  if (in_range) {
    float* g = grad + ((size_t)t * n_pad + n) * C;
    const float* x = logits + ((size_t)t * n_pad + n) * C;
    const float ls = active ? lse[(size_t)t * n_pad + n] : 0.f;
    for (int c = lane; c < C; c += 64) {
      g[c] = active ? scale * (__expf(x[c] - ls) - bins[c]) : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------
// greedy decode: one 256-thread block per utterance.
__global__ void __launch_bounds__(256)
ctc_greedy_kernel(const float* __restrict__ logits, const int* __restrict__ seq_len,
                  int T, int N, int n_pad, int C, int* __restrict__ decoded,
                  int* __restrict__ decoded_len) {
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int blank = C - 1;
  int Tn = seq_len[n];
  Tn = Tn < 0 ? 0 : (Tn > T ? T : Tn);
  __shared__ int s_best[256 + 1];
  __shared__ int s_scan[256];
  __shared__ int s_base;
  if (tid == 0) { s_base = 0; s_best[0] = -1; }
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    int best = -1;
    if (t < Tn) {
      const float* r = logits + ((size_t)t * n_pad + n) * C;
      float bv = r[0]; best = 0;
      for (int c = 1; c < C; ++c) { const float v = r[c]; if (v > bv) { bv = v; best = c; } }
    }
    // s_best[0] carries the previous chunk's last argmax
    s_best[tid + 1] = best;
    __syncthreads();
    const int prev = s_best[tid];
    const int keep = (t < Tn && best != blank && best != prev) ? 1 : 0;
    s_scan[tid] = keep;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {          // Hillis-Steele inclusive scan
      int v = (tid >= o) ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const int base = s_base;
    if (keep) decoded[(size_t)n * T + base + s_scan[tid] - 1] = best;
    __syncthreads();
    if (tid == 255) { s_base = base + s_scan[255]; s_best[0] = s_best[256]; }
    __syncthreads();
  }
  const int total = s_base;
  for (int i = total + tid; i < T; i += 256) decoded[(size_t)n * T + i] = -1;
  if (tid == 0) decoded_len[n] = total;
}

int pick_ppl(int l_max) {
  const int pairs = l_max + 1;
  if (pairs <= 64) return 1;
  if (pairs <= 128) return 2;
  if (pairs <= 256) return 4;
  if (pairs <= 512) return 8;
  return 0;
}

}  // namespace

extern "C" size_t asr_ctc_workspace_bytes(int T, int N, int n_pad, int C, int l_max) {
  const int ppl = pick_ppl(l_max);
  if (ppl == 0 || T <= 0 || N <= 0) return 0;
  (void)C;
  const size_t sp = (size_t)2 * 64 * ppl;
  const int n_pad16 = n_pad > N ? n_pad : N;
  const size_t nck = (size_t)(T + kCk - 1) / kCk;
  // alpha + beta checkpoints, logZ (float64), one lse per frame
  return asr_align_up(nck * N * sp * sizeof(float), 256) * 2 +
         asr_align_up((size_t)N * sizeof(double), 256) +
         asr_align_up((size_t)T * n_pad16 * sizeof(float), 256);
}

extern "C" int asr_ctc_loss_grad(const float* logits, const int* labels,
                                 const int* label_len, const int* seq_len, int T,
                                 int N, int n_pad, int C, int l_max,
                                 float grad_scale, float* loss, float* grad,
                                 void* workspace, size_t ws_bytes,
                                 asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(logits && labels && label_len && seq_len && loss, "ctc: null pointer");
  // the gradient kernel re-reads the logits of neighbouring frames while other waves write grad
  ASR_CHECK_ARG(grad != logits, "ctc: grad must not alias logits");
  ASR_CHECK_ARG(T > 0 && N > 0 && n_pad >= N && C >= 2 && l_max >= 1,
                "ctc: bad shape T=%d N=%d n_pad=%d C=%d l_max=%d", T, N, n_pad, C, l_max);
  const int ppl = pick_ppl(l_max);
  ASR_CHECK_ARG(ppl != 0, "ctc: l_max=%d unsupported (max 511)", l_max);
  const size_t need = asr_ctc_workspace_bytes(T, N, n_pad, C, l_max);
  if (!workspace || ws_bytes < need) {
    asr_set_error("ctc: workspace %zu < %zu bytes", ws_bytes, need);
    return ASR_ERR_WORKSPACE;
  }
  const size_t sp = (size_t)2 * 64 * ppl;
  const size_t nck = (size_t)(T + kCk - 1) / kCk;
  const size_t ck_bytes = asr_align_up(nck * N * sp * sizeof(float), 256);
  const size_t lz_bytes = asr_align_up((size_t)N * sizeof(double), 256);
  char* wsb = reinterpret_cast<char*>(workspace);
  float* alpha_ck = reinterpret_cast<float*>(wsb);
  float* beta_ck = reinterpret_cast<float*>(wsb + ck_bytes);
  double* logz = reinterpret_cast<double*>(wsb + 2 * ck_bytes);
  float* lse = reinterpret_cast<float*>(wsb + 2 * ck_bytes + lz_bytes);
  const int rows = T * n_pad;
  hipLaunchKernelGGL(ctc_lse_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, logits, lse,
                     rows, C);
  ASR_CHECK_LAUNCH();
  const int do_beta = grad ? 1 : 0;
  const dim3 grid_ab(do_beta ? 2 * N : N);
#define LAUNCH_AB(P)                                                                   \
  hipLaunchKernelGGL(ctc_alpha_beta_kernel<P>, grid_ab, dim3(64), 0, stream, logits,   \
                     lse, labels, label_len, seq_len, T, N, n_pad, C, l_max, alpha_ck, \
                     beta_ck, logz, loss, do_beta)
  switch (ppl) {
    case 1: LAUNCH_AB(1); break;
    case 2: LAUNCH_AB(2); break;
    case 4: LAUNCH_AB(4); break;
    default: LAUNCH_AB(8); break;
  }
#undef LAUNCH_AB
  ASR_CHECK_LAUNCH();
  if (grad) {
    const long pairs = (long)T * n_pad;
    const dim3 grid_g((unsigned)((pairs + 3) / 4));
    const size_t shm = (size_t)4 * C * sizeof(float);
#define LAUNCH_G(P)                                                                     \
  hipLaunchKernelGGL(ctc_grad_kernel<P>, grid_g, dim3(256), shm, stream, logits, lse,   \
                     alpha_ck, beta_ck, logz, labels, label_len, seq_len, T, N, n_pad,  \
                     C, l_max, grad_scale, grad)
    switch (ppl) {
      case 1: LAUNCH_G(1); break;
      case 2: LAUNCH_G(2); break;
      case 4: LAUNCH_G(4); break;
      default: LAUNCH_G(8); break;
    }
#undef LAUNCH_G
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}

extern "C" int asr_ctc_greedy(const float* logits, const int* seq_len, int T, int N,
                              int n_pad, int C, int* decoded, int* decoded_len,
                              asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(logits && seq_len && decoded && decoded_len, "greedy: null pointer");
  ASR_CHECK_ARG(T > 0 && N > 0 && n_pad >= N && C >= 2, "greedy: bad shape");
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(N), dim3(256), 0, stream, logits, seq_len,
                     T, N, n_pad, C, decoded, decoded_len);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}
