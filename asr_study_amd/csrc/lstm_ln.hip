// Layer-normalised LSTM cell (core/layers.py:407-436, 460-462; core/layers_utils.py:16-19)
// -- the generic, row-per-workgroup path of the recurrent layer.
//
// Layer normalisation needs the statistics of a whole (sample, direction) row -- 4H gate
// pre-activations of h@U, 4H of x@W, H cell values -- at every step.  The persistent
// kernels of lstm.hip split a row over 16-32 workgroups, so each statistic would cost
// another cross-workgroup hand-off per step.  This optional cell (default off in the
// reference, flagged "returning a lot of nan" there) therefore runs step by step instead:
//   forward  step: uh = (h_prev (.) B_U) @ U   (asr_gemm, 16-64 rows)  ->  cell kernel,
//   backward step: cell kernel  ->  dh_prev = (duh @ U^T) (.) B_U      (asr_gemm),
// with ONE workgroup per (sample, direction) row, so every normalisation is a reduction
// inside a workgroup.  About 3 launches per step: several times slower than the
// persistent kernels, but complete: LN combines freely with multiplicative integration
// and zoneout.  All 4H axes are unit-major / gate-minor like everywhere else.
#include "common.h"

namespace {

constexpr int kT = 256;           // threads per row workgroup
constexpr int kMaxU = 2;          // units per thread: H <= 512
constexpr float kLnEps = 1e-5f;

// per-direction parameter block (floats): alpha, beta1, beta2, bias (4H each), then
// gain/bias of LN(h@U), LN(x@W) (4H each) and of LN(c) (H each)  = 34 H
struct Offs { int alpha, beta1, beta2, bias, gu, bu, gw, bw, gc, bc, total; };
__host__ __device__ inline Offs offs(int H) {
  Offs o;
  o.alpha = 0; o.beta1 = 4 * H; o.beta2 = 8 * H; o.bias = 12 * H;
  o.gu = 16 * H; o.bu = 20 * H; o.gw = 24 * H; o.bw = 28 * H; o.gc = 32 * H; o.bc = 33 * H;
  o.total = 34 * H;
  return o;
}

__device__ __forceinline__ float hsig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hsig_grad_from_y(float y) { return (y > 0.f && y < 1.f) ? 0.2f : 0.f; }

// sums of up to NV values over the workgroup; results broadcast to every thread
template <int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = asr_wave_sum(v[i]);
  __syncthreads();                       // lds reuse across calls
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[w * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (lds[i] + lds[NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float sumsq4(float4 a, float m) {
  const float x = a.x - m, y = a.y - m, z = a.z - m, w = a.w - m;
  return (x * x + y * y) + (z * z + w * w);
}
__device__ __forceinline__ float4 norm4(float4 a, float m, float r) {
  return make_float4((a.x - m) * r, (a.y - m) * r, (a.z - m) * r, (a.w - m) * r);
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {   // a*b + c
  return make_float4(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z, a.w * b.w + c.w);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}

struct CellArgs {
  int T, n_pad, H, has_mi, first;       // first: no previous step (h_prev = c_prev = 0)
  int act;                              // activation id (common.h asr_act_apply)
  int t[2], tp[2];                      // frame of this step / of the previous step, per direction
  const float* cellp;                   // (2, 34H)
  const float* zone_c; const float* zone_h;
  const float* wx; float* uh; float* y; float* cell; float* gates;
  const float* dy; float* duh; float* dwx; float* dparams;
  const float* dh_rec;                  // (n_pad, 2, H) from the previous BPTT step's GEMM
  float* dc_state; float* dhz_state;    // (n_pad, 2, H)
};

// mean and 1/sqrt(var + eps) of a row of n values spread as float4 per (thread, m)
__device__ __forceinline__ void row_stats4(const float4 (&v)[kMaxU], const bool (&ok)[kMaxU],
                                           int n, float* lds, float& mean, float& rstd) {
  float s[1] = {0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) if (ok[m]) s[0] += sum4(v[m]);
  block_sums<1>(s, lds);
  mean = s[0] / n;
  float q[1] = {0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) if (ok[m]) q[0] += sumsq4(v[m], mean);
  block_sums<1>(q, lds);
  rstd = rsqrtf(q[0] / n + kLnEps);
}
__device__ __forceinline__ void row_stats1(const float (&v)[kMaxU], const bool (&ok)[kMaxU], int n,
                                           float* lds, float& mean, float& rstd) {
  float s[1] = {0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) if (ok[m]) s[0] += v[m];
  block_sums<1>(s, lds);
  mean = s[0] / n;
  float q[1] = {0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) if (ok[m]) q[0] += (v[m] - mean) * (v[m] - mean);
  block_sums<1>(q, lds);
  rstd = rsqrtf(q[0] / n + kLnEps);
}

__global__ void __launch_bounds__(kT)
cell_ln_fwd_kernel(CellArgs a) {
  __shared__ float lds[16];
  const int n = blockIdx.x, d = blockIdx.y, H = a.H, H4 = 4 * H;
  const Offs o = offs(H);
  const float* P = a.cellp + (size_t)d * o.total;
  const int t = a.t[d], tp = a.tp[d];
  const size_t r4 = (((size_t)t * a.n_pad + n) * 2 + d) * H4;
  const size_t r1 = (((size_t)t * a.n_pad + n) * 2 + d) * H;
  float4 uh[kMaxU], wx[kMaxU];
  bool ok[kMaxU];
  int un[kMaxU];
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    un[m] = threadIdx.x + kT * m;
    ok[m] = un[m] < H;
    uh[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    wx[m] = uh[m];
    if (ok[m]) {
      if (!a.first) uh[m] = ld4(a.uh + r4 + 4 * un[m]);
      else st4(a.uh + r4 + 4 * un[m], uh[m]);           // h_prev = 0: h@U = 0
      wx[m] = ld4(a.wx + r4 + 4 * un[m]);
    }
  }
  float mu_u, rs_u, mu_w, rs_w;
  row_stats4(uh, ok, H4, lds, mu_u, rs_u);
  row_stats4(wx, ok, H4, lds, mu_w, rs_w);
  float cnew[kMaxU], hprev[kMaxU], kh[kMaxU];
  float4 g4[kMaxU];
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    cnew[m] = 0.f; hprev[m] = 0.f; kh[m] = 1.f; g4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok[m]) continue;
    const int j = 4 * un[m];
    const float4 unn = fma4(norm4(uh[m], mu_u, rs_u), ld4(P + o.gu + j), ld4(P + o.bu + j));
    const float4 wnn = fma4(norm4(wx[m], mu_w, rs_w), ld4(P + o.gw + j), ld4(P + o.bw + j));
    float4 z;
    if (a.has_mi) {
      const float4 al = ld4(P + o.alpha + j), b1 = ld4(P + o.beta1 + j), b2 = ld4(P + o.beta2 + j);
      z = add4(add4(mul4(mul4(al, wnn), unn), mul4(b1, unn)), add4(mul4(b2, wnn), ld4(P + o.bias + j)));
    } else {
      z = add4(add4(wnn, unn), ld4(P + o.bias + j));
    }
    const float gi = hsig(z.x), gf = hsig(z.y), gg = asr_act_apply(a.act, z.z), go = hsig(z.w);
    g4[m] = make_float4(gi, gf, gg, go);
    float cprev = 0.f;
    if (!a.first) {
      cprev = a.cell[(((size_t)tp * a.n_pad + n) * 2 + d) * H + un[m]];
      hprev[m] = a.y[((size_t)tp * a.n_pad + n) * 2 * H + d * H + un[m]];
    }
    float c = gf * cprev + gi * gg;
    if (a.zone_c) c = cprev + a.zone_c[((size_t)t * 2 + d) * H + un[m]] * (c - cprev);
    if (a.zone_h) kh[m] = a.zone_h[((size_t)t * 2 + d) * H + un[m]];
    cnew[m] = c;
  }
  float mu_c, rs_c;
  row_stats1(cnew, ok, H, lds, mu_c, rs_c);
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    if (!ok[m]) continue;
    const float cn = (cnew[m] - mu_c) * rs_c * P[o.gc + un[m]] + P[o.bc + un[m]];
    float h = g4[m].w * asr_act_apply(a.act, cn);
    h = hprev[m] + kh[m] * (h - hprev[m]);
    a.y[((size_t)t * a.n_pad + n) * 2 * H + d * H + un[m]] = h;
    a.cell[r1 + un[m]] = cnew[m];
    st4(a.gates + r4 + 4 * un[m], g4[m]);
  }
}

__global__ void __launch_bounds__(kT)
cell_ln_bwd_kernel(CellArgs a) {
  __shared__ float lds[32];
  const int n = blockIdx.x, d = blockIdx.y, H = a.H, H4 = 4 * H;
  const Offs o = offs(H);
  const float* P = a.cellp + (size_t)d * o.total;
  float* G = a.dparams + ((size_t)n * 2 + d) * o.total;
  const int t = a.t[d], tp = a.tp[d];          // tp: the frame processed BEFORE t in forward order
  const size_t r4 = (((size_t)t * a.n_pad + n) * 2 + d) * H4;
  const size_t sH = ((size_t)n * 2 + d) * H;
  float4 uh[kMaxU], wx[kMaxU], g4[kMaxU];
  float cval[kMaxU], cprev[kMaxU];
  bool ok[kMaxU];
  int un[kMaxU];
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    un[m] = threadIdx.x + kT * m;
    ok[m] = un[m] < H;
    uh[m] = make_float4(0.f, 0.f, 0.f, 0.f); wx[m] = uh[m]; g4[m] = uh[m];
    cval[m] = 0.f; cprev[m] = 0.f;
    if (ok[m]) {
      uh[m] = ld4(a.uh + r4 + 4 * un[m]);
      wx[m] = ld4(a.wx + r4 + 4 * un[m]);
      g4[m] = ld4(a.gates + r4 + 4 * un[m]);
      cval[m] = a.cell[(((size_t)t * a.n_pad + n) * 2 + d) * H + un[m]];
      if (tp >= 0) cprev[m] = a.cell[(((size_t)tp * a.n_pad + n) * 2 + d) * H + un[m]];
    }
  }
  float mu_u, rs_u, mu_w, rs_w, mu_c, rs_c;
  row_stats4(uh, ok, H4, lds, mu_u, rs_u);
  row_stats4(wx, ok, H4, lds, mu_w, rs_w);
  row_stats1(cval, ok, H, lds, mu_c, rs_c);
  // ---- output side: h = h_prev + kh (o tanh(LN(c)) - h_prev)
  float dcn[kMaxU], chat[kMaxU], d_o[kMaxU], s2[2] = {0.f, 0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    dcn[m] = 0.f; chat[m] = 0.f; d_o[m] = 0.f;
    if (!ok[m]) continue;
    float dh = a.dy[((size_t)t * a.n_pad + n) * 2 * H + d * H + un[m]] + a.dh_rec[sH + un[m]] +
               a.dhz_state[sH + un[m]];
    const float kh = a.zone_h ? a.zone_h[((size_t)t * 2 + d) * H + un[m]] : 1.f;
    a.dhz_state[sH + un[m]] = (1.f - kh) * dh;
    dh *= kh;
    chat[m] = (cval[m] - mu_c) * rs_c;
    const float cn = chat[m] * P[o.gc + un[m]] + P[o.bc + un[m]];
    const float tc = asr_act_apply(a.act, cn);
    d_o[m] = dh * tc;
    dcn[m] = dh * g4[m].w * asr_act_slope(a.act, tc);
    G[o.gc + un[m]] += dcn[m] * chat[m];
    G[o.bc + un[m]] += dcn[m];
    const float g = dcn[m] * P[o.gc + un[m]];
    s2[0] += g; s2[1] += g * chat[m];
  }
  block_sums<2>(s2, lds);
  // ---- cell: c = c_prev + kc (f c_prev + i g - c_prev)
  float4 dun[kMaxU], dwn[kMaxU], uhat[kMaxU], what[kMaxU];
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    dun[m] = make_float4(0.f, 0.f, 0.f, 0.f); dwn[m] = dun[m]; uhat[m] = dun[m]; what[m] = dun[m];
    if (!ok[m]) continue;
    const int j = 4 * un[m];
    const float g = dcn[m] * P[o.gc + un[m]];
    float dc = a.dc_state[sH + un[m]] + rs_c * (g - s2[0] / H - chat[m] * (s2[1] / H));
    const float kc = a.zone_c ? a.zone_c[((size_t)t * 2 + d) * H + un[m]] : 1.f;
    const float dcz = (1.f - kc) * dc;
    dc *= kc;
    const float gi = g4[m].x, gf = g4[m].y, gg = g4[m].z, go = g4[m].w;
    a.dc_state[sH + un[m]] = dc * gf + dcz;
    float4 dz;
    dz.x = dc * gg * hsig_grad_from_y(gi);
    dz.y = dc * cprev[m] * hsig_grad_from_y(gf);
    dz.z = dc * gi * asr_act_slope(a.act, gg);
    dz.w = d_o[m] * hsig_grad_from_y(go);
    uhat[m] = norm4(uh[m], mu_u, rs_u);
    what[m] = norm4(wx[m], mu_w, rs_w);
    const float4 gu = ld4(P + o.gu + j), gw = ld4(P + o.gw + j);
    if (a.has_mi) {
      const float4 unn = fma4(uhat[m], gu, ld4(P + o.bu + j));
      const float4 wnn = fma4(what[m], gw, ld4(P + o.bw + j));
      const float4 al = ld4(P + o.alpha + j), b1 = ld4(P + o.beta1 + j), b2 = ld4(P + o.beta2 + j);
      st4(G + o.alpha + j, add4(ld4(G + o.alpha + j), mul4(mul4(dz, wnn), unn)));
      st4(G + o.beta1 + j, add4(ld4(G + o.beta1 + j), mul4(dz, unn)));
      st4(G + o.beta2 + j, add4(ld4(G + o.beta2 + j), mul4(dz, wnn)));
      dun[m] = mul4(dz, fma4(al, wnn, b1));
      dwn[m] = mul4(dz, fma4(al, unn, b2));
    } else {
      dun[m] = dz; dwn[m] = dz;
    }
    st4(G + o.bias + j, add4(ld4(G + o.bias + j), dz));
    st4(G + o.gu + j, add4(ld4(G + o.gu + j), mul4(dun[m], uhat[m])));
    st4(G + o.bu + j, add4(ld4(G + o.bu + j), dun[m]));
    st4(G + o.gw + j, add4(ld4(G + o.gw + j), mul4(dwn[m], what[m])));
    st4(G + o.bw + j, add4(ld4(G + o.bw + j), dwn[m]));
    const float4 qu = mul4(dun[m], gu), qw = mul4(dwn[m], gw);
    s4[0] += sum4(qu); s4[1] += dot4(qu, uhat[m]);
    s4[2] += sum4(qw); s4[3] += dot4(qw, what[m]);
  }
  block_sums<4>(s4, lds);
#pragma unroll
  for (int m = 0; m < kMaxU; ++m) {
    if (!ok[m]) continue;
    const int j = 4 * un[m];
    const float4 gu = ld4(P + o.gu + j), gw = ld4(P + o.gw + j);
    const float4 qu = mul4(dun[m], gu), qw = mul4(dwn[m], gw);
    const float m1u = s4[0] / H4, m2u = s4[1] / H4, m1w = s4[2] / H4, m2w = s4[3] / H4;
    float4 du, dw;
    du.x = rs_u * (qu.x - m1u - uhat[m].x * m2u); du.y = rs_u * (qu.y - m1u - uhat[m].y * m2u);
    du.z = rs_u * (qu.z - m1u - uhat[m].z * m2u); du.w = rs_u * (qu.w - m1u - uhat[m].w * m2u);
    dw.x = rs_w * (qw.x - m1w - what[m].x * m2w); dw.y = rs_w * (qw.y - m1w - what[m].y * m2w);
    dw.z = rs_w * (qw.z - m1w - what[m].z * m2w); dw.w = rs_w * (qw.w - m1w - what[m].w * m2w);
    st4(a.duh + r4 + j, du);
    st4(a.dwx + r4 + j, dw);
  }
}

size_t state_bytes(const asr_lstm_ln_args* a) {
  return asr_align_up((size_t)a->n_pad * 2 * a->H * sizeof(float), 256);
}

}  // namespace

extern "C" size_t asr_lstm_ln_workspace_bytes(const asr_lstm_ln_args* a) {
  if (!a || a->n_pad <= 0 || a->H <= 0) return 0;
  return 3 * state_bytes(a);          // dh_rec, dc, dhz
}

static int check_common(const asr_lstm_ln_args* a) {
  ASR_CHECK_ARG(a && a->U && a->cellp && a->wx && a->uh && a->y && a->cell && a->gates,
                "lstm_ln: null pointer");
  ASR_CHECK_ARG(a->T > 0 && a->n_pad > 0 && a->n_pad % 16 == 0 && a->H >= 4 && a->H % 4 == 0 &&
                    a->H <= kT * kMaxU,
                "lstm_ln: need n_pad %% 16 == 0, H %% 4 == 0, H <= %d (T=%d n_pad=%d H=%d)",
                kT * kMaxU, a->T, a->n_pad, a->H);
  ASR_CHECK_ARG(a->activation >= 0 && a->activation <= 6,
                "lstm_ln: activation id %d out of range (0..6, asr_activation)", a->activation);
  return ASR_OK;
}

static void fill_cell(const asr_lstm_ln_args* a, CellArgs* c) {
  c->T = a->T; c->n_pad = a->n_pad; c->H = a->H; c->has_mi = a->has_mi;
  c->act = a->activation;
  c->cellp = a->cellp; c->zone_c = a->zone_c; c->zone_h = a->zone_h;
  c->wx = a->wx; c->uh = a->uh; c->y = a->y; c->cell = a->cell; c->gates = a->gates;
  c->dy = a->dy; c->duh = a->duh; c->dwx = a->dwx; c->dparams = a->dparams;
  c->dh_rec = nullptr; c->dc_state = nullptr; c->dhz_state = nullptr;
}

extern "C" int asr_lstm_ln_seq_fwd(const asr_lstm_ln_args* a, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = check_common(a);
  if (rc != ASR_OK) return rc;
  const int T = a->T, n_pad = a->n_pad, H = a->H;
  CellArgs c;
  fill_cell(a, &c);
  for (int s = 0; s < T; ++s) {
    c.first = s == 0;
    for (int d = 0; d < 2; ++d) {
      c.t[d] = d == 0 ? s : T - 1 - s;
      c.tp[d] = d == 0 ? s - 1 : T - s;
      if (s == 0) continue;
      // uh[t] = (h_prev (.) B_U) @ U_d
      asr_gemm_args g = {};
      g.M = n_pad; g.N = 4 * H; g.K = H;
      g.A = a->y + (size_t)c.tp[d] * n_pad * 2 * H + (size_t)d * H; g.lda = 2 * H;
      g.B = a->U + (size_t)d * H * 4 * H; g.ldb = 4 * H;
      g.C = a->uh + (((size_t)c.t[d] * n_pad) * 2 + d) * 4 * H; g.ldc = 8 * H;
      g.alpha = 1.f; g.beta = 0.f; g.precision = -1;
      if (a->mask_u) {
        g.a_scale = a->mask_u + (size_t)d * n_pad * H; g.a_scale_period = n_pad; g.a_scale_ld = H;
      }
      const int grc = asr_gemm(&g, nullptr, 0, stream_);
      if (grc != ASR_OK) return grc;
    }
    hipLaunchKernelGGL(cell_ln_fwd_kernel, dim3(n_pad, 2), dim3(kT), 0, stream, c);
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}

extern "C" int asr_lstm_ln_seq_bwd(const asr_lstm_ln_args* a, void* workspace, size_t ws_bytes,
                                   asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = check_common(a);
  if (rc != ASR_OK) return rc;
  ASR_CHECK_ARG(a->dy && a->duh && a->dwx && a->dparams && workspace, "lstm_ln bwd: null pointer");
  const size_t need = asr_lstm_ln_workspace_bytes(a);
  if (ws_bytes < need) {
    asr_set_error("lstm_ln: workspace %zu < %zu bytes", ws_bytes, need);
    return ASR_ERR_WORKSPACE;
  }
  const int T = a->T, n_pad = a->n_pad, H = a->H;
  const Offs o = offs(H);
  char* ws = reinterpret_cast<char*>(workspace);
  const size_t sb = state_bytes(a);
  float* dh_rec = reinterpret_cast<float*>(ws);
  ASR_CHECK_HIP(hipMemsetAsync(ws, 0, 3 * sb, stream));
  ASR_CHECK_HIP(hipMemsetAsync(a->dparams, 0, (size_t)n_pad * 2 * o.total * sizeof(float), stream));
  CellArgs c;
  fill_cell(a, &c);
  c.dh_rec = dh_rec;
  c.dc_state = reinterpret_cast<float*>(ws + sb);
  c.dhz_state = reinterpret_cast<float*>(ws + 2 * sb);
  c.first = 0;
  for (int s = 0; s < T; ++s) {                 // BPTT: the forward order reversed
    for (int d = 0; d < 2; ++d) {
      const int fs = T - 1 - s;                 // forward step being undone
      c.t[d] = d == 0 ? fs : T - 1 - fs;
      c.tp[d] = fs == 0 ? -1 : (d == 0 ? fs - 1 : T - fs);
    }
    hipLaunchKernelGGL(cell_ln_bwd_kernel, dim3(n_pad, 2), dim3(kT), 0, stream, c);
    ASR_CHECK_LAUNCH();
    if (s + 1 == T) break;
    for (int d = 0; d < 2; ++d) {               // dh_prev = (duh @ U_d^T) (.) B_U
      asr_gemm_args g = {};
      g.M = n_pad; g.N = H; g.K = 4 * H; g.trans_b = 1;
      g.A = a->duh + (((size_t)c.t[d] * n_pad) * 2 + d) * 4 * H; g.lda = 8 * H;
      g.B = a->U + (size_t)d * H * 4 * H; g.ldb = 4 * H;
      g.C = dh_rec + (size_t)d * H; g.ldc = 2 * H;
      g.alpha = 1.f; g.beta = 0.f; g.precision = 0;      // exact fp32: tiny, unscaled gradients
      if (a->mask_u) {
        g.c_scale = a->mask_u + (size_t)d * n_pad * H; g.c_scale_period = n_pad; g.c_scale_ld = H;
      }
      const int grc = asr_gemm(&g, nullptr, 0, stream_);
      if (grc != ASR_OK) return grc;
    }
  }
  return ASR_OK;
}
