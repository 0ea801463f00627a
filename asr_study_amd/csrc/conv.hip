// K13  2-D convolution front-end (BASELINE.json configs[2]: "2 conv front-end") -- gfx950.
//
// NO REFERENCE COUNTERPART: the reference lists Deep Speech 2 as TODO (README.md:118) and its
// deep_speech factory (core/models.py:148-214) has no convolution.  The op is Keras-1.2.2
// Convolution2D(border_mode='same', dim_ordering='tf') + the reference's clipped ReLU
// (core/models.py:116) on the (N, T, F, C) view of the time-major feature slab; see
// include/asr_hip.h (K13) for the contract.
//
// How the convolution becomes the packed split-fp16 GEMM of K4 without an im2col buffer:
//   * over FREQUENCY it is folded into one banded matrix per time tap,
//       band[dt][(fi, ci)][(fo, co)] = W[dt][fi - sf fo + pf][ci][co]   (0 outside the filter),
//     rebuilt from W every step by conv_band_kernel (a few MB) and packed into planes;
//   * over TIME it is the SEGMENTED reduction range of asr_gemm_hl: output row (t', n) of tap dt
//     reads the packed input planes of frame st t' + dt - pt -- the same planes, dt frames on.
//     conv_pack_kernel writes the input planes in a "phase" layout (frames t = st s + r of phase
//     r are consecutive slots s, zero slots where t is outside the slab: the 'same' padding), so
//     that every tap is a constant row shift of the planes for any time stride;
//   * the weight gradient is one K-major GEMM per tap on the SAME planes (x^T dz reduces over
//     the plane rows), giving the gradient of the band, which conv_band_reduce_kernel folds
//     back onto the filter taps.
// The band is ~kf / F_in dense (21 of 40, 41 of 80 here).  Where the channel counts allow whole
// slabs at the cuts (C_in % 32 == 0: the second layer) the GEMMs run on FREQUENCY BLOCKS -- the
// forward GEMM on 256-column blocks of output frequencies with the input frequencies their
// filters reach (a column range of the same planes, a shorter reduction), the input-gradient
// and weight-gradient GEMMs on tile-aligned row blocks of the band -- and the block launches of a
// pass, each half a round of the chip, go out side by side on two or three streams (ForkJoin).  The
// matrix pipes then execute 1.3 x (blocks) to 2 x (first layer) the algorithmic flops of the
// convolution, at the rate of the tuned 256 x 256 kernel (bench.py reports the ALGORITHMIC
// rate).  The clipped ReLU is applied in the epilogue of the forward GEMM (only y is written; the
// backward mask 0 < z < clip reads the same from y); the bias gradient comes from column sums
// the dz pack keeps in registers.  All kernels besides the GEMMs are HBM-bound element-wise
// passes over the activation slabs.
#include "common.h"

namespace {

typedef _Float16 hx8 __attribute__((ext_vector_type(8)));

struct Geo {
  int T_in, n_pad, F_in, C_in, C_out, kt, kf, st, sf;
  int T_out, F_out, pt, pf;       // 'same' padding before (TensorFlow's rule)
  int Ki, Ko, Ki_p, Ko_p;         // row widths of x / z and their plane widths (multiples of 32)
  int qmin, S;                    // phase layout of the x planes: slots s - qmin in [0, S)
  int padb, pada;                 // zero frames before / behind the dz planes
  long long M;                    // output rows T_out * n_pad
  // Frequency blocks: the band of a time tap is kf / F_in dense.  The FORWARD GEMM cuts its
  // output columns into 256-wide blocks of output frequencies (whole tiles); block b (outputs
  // [fo0, fo0 + nfo)) only reads the input frequencies [fi0, fi0 + nfi) its filters reach -- a
  // GEMM on a column range of the same planes with a shorter reduction (second layer of the
  // front-end: 3 blocks reading 26 / 33 / 17 of 40 input frequencies).  A block alone is half a
  // round of the chip (125 workgroups at 64 x 10 s): the blocks' launches go out on separate
  // streams (fork_join below).  nblk = 1: the whole band.
  int nblk;
  int b_fo0[4], b_nfo[4], b_fi0[4], b_nfi[4];
  // The input gradient and the weight gradient cut the band by ROWS instead (tile-aligned
  // blocks of input frequencies, each with the output frequencies that reach it): dx blocks do
  // not overlap (nothing is accumulated between blocks), and the weight gradient's result is
  // the band itself, so what counts there is the number of 256 x 256 tiles covering its
  // nonzeros (second layer: 5 row blocks x 2 tiles instead of 5 x 3).  nwblk = 1: whole band.
  int nwblk;
  int w_fo0[8], w_nfo[8], w_fi0[8], w_nfi[8];
};

__host__ __device__ inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

bool make_geo(const asr_conv2d_args* a, Geo* g) {
  if (!a || a->T_in <= 0 || a->n_pad <= 0 || a->n_pad % 16 || a->F_in <= 0 || a->C_in <= 0 ||
      a->C_out <= 0 || a->kt <= 0 || a->kt > 16 || a->kf <= 0 || a->st <= 0 || a->sf <= 0)
    return false;
  g->T_in = a->T_in; g->n_pad = a->n_pad; g->F_in = a->F_in; g->C_in = a->C_in;
  g->C_out = a->C_out; g->kt = a->kt; g->kf = a->kf; g->st = a->st; g->sf = a->sf;
  g->T_out = (a->T_in + a->st - 1) / a->st;
  g->F_out = (a->F_in + a->sf - 1) / a->sf;
  int tot = (g->T_out - 1) * a->st + a->kt - a->T_in;
  g->pt = (tot > 0 ? tot : 0) / 2;
  tot = (g->F_out - 1) * a->sf + a->kf - a->F_in;
  g->pf = (tot > 0 ? tot : 0) / 2;
  g->Ki = a->F_in * a->C_in; g->Ko = g->F_out * a->C_out;
  if (g->Ki % 4 || g->Ko % 4) return false;        // 16-byte rows (pack, GEMM epilogue)
  g->Ki_p = (g->Ki + 31) / 32 * 32; g->Ko_p = (g->Ko + 31) / 32 * 32;
  g->qmin = floordiv(0 - g->pt, a->st);
  const int qmax = floordiv(a->kt - 1 - g->pt, a->st);
  g->S = g->T_out + qmax - g->qmin;
  g->padb = a->kt - 1 - g->pt; g->pada = g->pt;
  if (g->padb < 0) g->padb = 0;
  g->M = (long long)g->T_out * a->n_pad;
  // column blocks (forward): whole 256-column tiles of output frequencies, each with the input
  // frequencies its filters reach.  Cuts need whole slabs of the reduction: C_in % 32 == 0.
  // Adopted when they lower (column tiles) x (reduction length) summed over the blocks.
  g->nblk = 1;
  g->b_fo0[0] = 0; g->b_nfo[0] = g->F_out; g->b_fi0[0] = 0; g->b_nfi[0] = a->F_in;
  if (a->C_in % 32 == 0 && a->C_out <= 256 && 256 % a->C_out == 0) {
    int cb = 256 / a->C_out;
    while ((g->F_out + cb - 1) / cb > 4) cb *= 2;
    const int nb = (g->F_out + cb - 1) / cb;
    int cost = 0, fi0[4], nfi[4];
    for (int b = 0; b < nb; ++b) {
      const int fo0 = b * cb, nfo = fo0 + cb < g->F_out ? cb : g->F_out - fo0;
      int lo = a->sf * fo0 - g->pf, hi = a->sf * (fo0 + nfo - 1) - g->pf + a->kf;
      if (lo < 0) lo = 0;
      if (hi > a->F_in) hi = a->F_in;
      if (hi <= lo) hi = lo + 1;
      fi0[b] = lo; nfi[b] = hi - lo;
      cost += ((nfo * a->C_out + 255) / 256) * nfi[b];
    }
    if (nb > 1 && cost < ((g->Ko + 255) / 256) * a->F_in) {
      g->nblk = nb;
      for (int b = 0; b < nb; ++b) {
        g->b_fo0[b] = b * cb;
        g->b_nfo[b] = b * cb + cb < g->F_out ? cb : g->F_out - b * cb;
        g->b_fi0[b] = fi0[b]; g->b_nfi[b] = nfi[b];
      }
    }
  }
  // row blocks (dgrad, weight gradient): fpb input frequencies = one 256-row tile of the band
  g->nwblk = 1;
  g->w_fo0[0] = 0; g->w_nfo[0] = g->F_out; g->w_fi0[0] = 0; g->w_nfi[0] = a->F_in;
  if (a->C_in % 32 == 0 && a->C_out % 32 == 0 && 256 % a->C_in == 0) {
    int fpb = 256 / a->C_in;
    while ((a->F_in + fpb - 1) / fpb > 8) fpb *= 2;
    const int nb = (a->F_in + fpb - 1) / fpb;
    int tiles = 0, fo0[8], nfo[8];
    for (int b = 0; b < nb; ++b) {
      const int fi0 = b * fpb, fi1 = (fi0 + fpb < a->F_in ? fi0 + fpb : a->F_in) - 1;
      // fo reaches fi when 0 <= fi - sf fo + pf < kf
      int lo = -floordiv(-(fi0 + g->pf - (a->kf - 1)), a->sf), hi = floordiv(fi1 + g->pf, a->sf);
      if (lo < 0) lo = 0;
      if (hi > g->F_out - 1) hi = g->F_out - 1;
      if (hi < lo) { hi = lo; }
      fo0[b] = lo; nfo[b] = hi - lo + 1;
      tiles += (((fi1 - fi0 + 1) * a->C_in + 255) / 256) * ((nfo[b] * a->C_out + 255) / 256);
    }
    if (nb > 1 && tiles < ((g->Ki + 255) / 256) * ((g->Ko + 255) / 256)) {
      g->nwblk = nb;
      for (int b = 0; b < nb; ++b) {
        g->w_fi0[b] = b * fpb;
        g->w_nfi[b] = (b * fpb + fpb < a->F_in ? fpb : a->F_in - b * fpb);
        g->w_fo0[b] = fo0[b]; g->w_nfo[b] = nfo[b];
      }
    }
  }
  return true;
}

// Block b as a convolution geometry of its own for the band kernels: F_in' = nfi, F_out' = nfo and
// the frequency padding that makes band'[fi'][fo'] = W[fi' - sf fo' + pf'] the block of the full
// band (pf' = pf + fi0 - sf fo0).  Time / phase fields and n_pad are the layer's.
Geo rect_geo(const Geo& g, int fi0, int nfi, int fo0, int nfo) {
  Geo q = g;
  q.F_in = nfi; q.F_out = nfo;
  q.pf = g.pf + fi0 - g.sf * fo0;
  q.Ki = q.F_in * g.C_in; q.Ko = q.F_out * g.C_out;
  q.Ki_p = (q.Ki + 31) / 32 * 32; q.Ko_p = (q.Ko + 31) / 32 * 32;
  return q;
}
Geo block_geo(const Geo& g, int b) { return rect_geo(g, g.b_fi0[b], g.b_nfi[b], g.b_fo0[b], g.b_nfo[b]); }
Geo wblock_geo(const Geo& g, int b) { return rect_geo(g, g.w_fi0[b], g.w_nfi[b], g.w_fo0[b], g.w_nfo[b]); }

// plane row of tap dt's first operand row (output frame 0, sample 0) in the phase layout
inline long long tap_row(const Geo& g, int dt) {
  const int q = floordiv(dt - g.pt, g.st), r = (dt - g.pt) - q * g.st;
  return ((long long)r * g.S + (q - g.qmin)) * g.n_pad;
}

// threads of the dz pack that share a column group: each keeps the column sums of its rows
// (bias gradient) in registers -> a (kPartRows, Ko_p) slab of partial sums, no fp32 copy of dz
constexpr int kPartRows = 2048;

struct Ws {                       // byte offsets into the caller's workspace
  size_t scal, xpl, dzpl, colpart, band_f, band_dg, bandf_pl, banddg_pl, bias_band, dband, gemm, colsum, end;
  size_t gemm_bytes, colsum_bytes;
};

int wgrad_splits(const Geo& g) {
  // kt K-major GEMMs (one per time tap, (Ki x Ko) outputs) in one batched launch: split the
  // long row reduction until the launch fills the chip ONCE -- a workgroup of the 256 x 256
  // kernel has a CU to itself (256 slots), the 128 x 128 kernel runs two per CU; a launch of
  // 330 workgroups was two rounds, the second 29 % full.  More splits also add partial-sum
  // traffic (a 64-way split per tap wrote and re-read 210 MB of partials per tap).
  const int tl = (g.Ki >= 256 && g.Ko >= 256) ? 256 : 128;
  const int slots = tl == 256 ? 256 : 512;
  const int tiles = ((g.Ki + tl - 1) / tl) * ((g.Ko + tl - 1) / tl) * g.kt;   // all taps: one launch
  long long s = slots / tiles;
  if (s > 64) s = 64;
  if (s > g.M / 256) s = g.M / 256;
  return s < 1 ? 1 : (int)s;
}

Ws make_ws(const Geo& g) {
  Ws w;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += asr_align_up(bytes, 256); return at; };
  w.scal = take(64 * sizeof(float));
  w.xpl = take((size_t)g.st * g.S * g.n_pad * g.Ki_p * 4);
  w.dzpl = take((size_t)(g.padb + g.T_out + g.pada) * g.n_pad * g.Ko_p * 4);
  w.colpart = take((size_t)kPartRows * g.Ko_p * 4);
  w.band_f = take((size_t)g.kt * g.Ki_p * g.Ko * 4);
  w.band_dg = take((size_t)g.Ki * g.kt * g.Ko_p * 4);
  // (the blocks' planes lie one behind the other, each 256-byte aligned)
  w.bandf_pl = take((size_t)g.Ko * g.kt * g.Ki_p * 4 + 8 * 256);
  w.banddg_pl = take((size_t)g.Ki * g.kt * g.Ko_p * 4 + 8 * 256);
  w.bias_band = take((size_t)g.Ko * 4);
  w.dband = take((size_t)g.kt * g.Ki * g.Ko * 4);
  w.gemm_bytes = 0;                       // split-K partials of the largest block's launch
  for (int b = 0; b < g.nwblk; ++b) {
    const Geo gb = wblock_geo(g, b);
    const size_t need = (size_t)wgrad_splits(gb) * g.kt * gb.Ki * gb.Ko * sizeof(float);
    if (need > w.gemm_bytes) w.gemm_bytes = need;
  }
  w.gemm_bytes = asr_align_up(w.gemm_bytes, 256);
  w.gemm = take(w.gemm_bytes);
  w.colsum_bytes = asr_colsum_workspace_bytes(kPartRows, g.Ko);
  w.colsum = take(w.colsum_bytes);
  w.end = o;
  return w;
}

// the power-of-two pre-scale of asr_pack_hl (gemm.hip, pow2_scale): max |x| -> [2^8, 2^9)
__device__ __forceinline__ float pow2_scale_of(const float* absmax) {
  if (absmax == nullptr) return 1.f;
  const unsigned b = __float_as_uint(*absmax);
  int e = (int)((b >> 23) & 0xff) - 126;
  if ((b & 0x7fffffffu) == 0u) return 1.f;
  int k = 9 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return __uint_as_float((unsigned)(127 + k) << 23);
}

// ---- band[dt][(fi,ci)][(fo,co)] from W, in the two arrangements the GEMMs read:
//   band_f  (kt * Ki_p rows, Ko cols)      row dt Ki_p + (fi,ci)    -> B of the forward GEMM
//   band_dg (Ki rows, kt * Ko_p cols)      col dt Ko_p + (fo,co)    -> B of the dgrad GEMM
// plus bias_band[(fo,co)] = bias[co].  One thread per (dt, k, j) of band_f (pad rows -> 0).
__global__ void __launch_bounds__(256)
conv_band_kernel(Geo g, const float* __restrict__ W, const float* __restrict__ bias,
                 float* __restrict__ band_f, float* __restrict__ band_dg,
                 float* __restrict__ bias_band) {
  const size_t total = (size_t)g.kt * g.Ki_p * g.Ko;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx % g.Ko);
    const size_t rk = idx / g.Ko;
    const int k = (int)(rk % g.Ki_p), dt = (int)(rk / g.Ki_p);
    float v = 0.f;
    if (k < g.Ki) {
      const int fi = k / g.C_in, ci = k % g.C_in, fo = j / g.C_out, co = j % g.C_out;
      const int df = fi - g.sf * fo + g.pf;
      if (df >= 0 && df < g.kf) v = W[(((size_t)dt * g.kf + df) * g.C_in + ci) * g.C_out + co];
      band_dg[(size_t)k * g.kt * g.Ko_p + (size_t)dt * g.Ko_p + j] = v;
    }
    band_f[idx] = v;
    if (bias_band && idx < (size_t)g.Ko) bias_band[idx] = bias ? bias[idx % g.C_out] : 0.f;
  }
}

// ---- activation slab -> interleaved split-fp16 planes (the r-plane format of asr_pack_hl: a
// row is K_p / 16 groups of [16 hi][16 lo]); one thread per (plane row, group of 16).
//   MODE 0: x planes in the phase layout (row = (phase r, slot s, sample n) <- frame
//           st (s + qmin) + r, zeros outside the slab);
//   MODE 1: dz planes, dz = dy (.) act'(z), rows (padb + t', n) with zero frames before and
//           behind.  Thread (slot, group) walks the rows slot, slot + kPartRows, ... of its
//           column group and leaves the column sums of what it packed in colpart[slot]
//           (the bias gradient = their sum, folded over fo; fixed order: deterministic).
template <int MODE>
__global__ void __launch_bounds__(256)
conv_pack_kernel(Geo g, const float* __restrict__ src, const float* __restrict__ zpre,
                 float clip, const float* __restrict__ absmax, float* __restrict__ scale_out,
                 _Float16* __restrict__ planes, float* __restrict__ colpart) {
  const int K = MODE == 0 ? g.Ki : g.Ko, K_p = MODE == 0 ? g.Ki_p : g.Ko_p;
  const int groups = K_p / 16;
  const long long frames = MODE == 0 ? (long long)g.st * g.S : (long long)g.padb + g.T_out + g.pada;
  const size_t rows = (size_t)frames * g.n_pad;
  const float s = pow2_scale_of(absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // MODE 0: one (row, group) item per step of a grid-stride loop; MODE 1: the group is the
  // thread's for good (the launch has exactly groups * kPartRows threads)
  const size_t total = MODE == 0 ? rows * groups : rows;
  const size_t step = MODE == 0 ? (size_t)gridDim.x * blockDim.x : (size_t)kPartRows;
  float csum[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) csum[e] = 0.f;
  for (size_t idx = MODE == 0 ? gtid : gtid / groups; idx < total; idx += step) {
    const int grp = (int)((MODE == 0 ? idx : gtid) % groups);
    const size_t row = MODE == 0 ? idx / groups : idx;
    const int n = (int)(row % g.n_pad);
    const long long fr = (long long)(row / g.n_pad);
    long long t;                                   // source frame, or out of range
    if (MODE == 0) {
      const int r = (int)(fr / g.S), sl = (int)(fr % g.S);
      t = (long long)g.st * (sl + g.qmin) + r;
      if (t >= g.T_in) t = -1;
    } else {
      t = fr - g.padb;
      if (t >= g.T_out) t = -1;
    }
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    const int c0 = grp * 16;
    if (t >= 0 && c0 < K) {
      const size_t base = ((size_t)t * g.n_pad + n) * K + c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (c0 + 4 * q < K) {                      // K % 4 == 0: whole quads
          const float4 a = *reinterpret_cast<const float4*>(src + base + 4 * q);
          v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
          if (MODE == 1 && clip > 0.f) {
            const float4 zz = *reinterpret_cast<const float4*>(zpre + base + 4 * q);
            if (!(zz.x > 0.f && zz.x < clip)) v[4 * q] = 0.f;
            if (!(zz.y > 0.f && zz.y < clip)) v[4 * q + 1] = 0.f;
            if (!(zz.z > 0.f && zz.z < clip)) v[4 * q + 2] = 0.f;
            if (!(zz.w > 0.f && zz.w < clip)) v[4 * q + 3] = 0.f;
          }
        }
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) csum[e] += v[e];
    }
    hx8 hi0, hi1, lo0, lo1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x0 = v[e] * s, x1 = v[8 + e] * s;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      hi0[e] = h0; hi1[e] = h1;
      lo0[e] = (_Float16)(x0 - (float)h0); lo1[e] = (_Float16)(x1 - (float)h1);
    }
    _Float16* dst = planes + (row * (size_t)(2 * K_p) + (size_t)grp * 32);
    *reinterpret_cast<hx8*>(dst) = hi0;
    *reinterpret_cast<hx8*>(dst + 8) = hi1;
    *reinterpret_cast<hx8*>(dst + 16) = lo0;
    *reinterpret_cast<hx8*>(dst + 24) = lo1;
  }
  if (MODE == 1 && gtid < (size_t)kPartRows * groups) {
    float4* out = reinterpret_cast<float4*>(colpart + (gtid / groups) * K_p + (gtid % groups) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      out[q] = make_float4(csum[4 * q], csum[4 * q + 1], csum[4 * q + 2], csum[4 * q + 3]);
  }
}

// y = min(max(z, 0), clip)
__global__ void __launch_bounds__(256)
conv_act_kernel(const float* __restrict__ z, float* __restrict__ y, size_t n4, float clip) {
  const float4* z4 = reinterpret_cast<const float4*>(z);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 v = z4[i];
    v.x = fminf(fmaxf(v.x, 0.f), clip); v.y = fminf(fmaxf(v.y, 0.f), clip);
    v.z = fminf(fmaxf(v.z, 0.f), clip); v.w = fminf(fmaxf(v.w, 0.f), clip);
    y4[i] = v;
  }
}

// dW[dt][df][ci][co] = sum_fo dband[dt][(sf fo + df - pf, ci)][(fo, co)] (float64 accumulation,
// fixed order); db[co] = sum_fo colsum[(fo, co)].  One thread per filter element.
// (g: the geometry of ONE frequency block -- block_geo; accumulate: add to dW (blocks 1..);
// db / colsum cover all n_fo_total output frequencies and are folded by the first call)
__global__ void __launch_bounds__(256)
conv_band_reduce_kernel(Geo g, const float* __restrict__ dband, const float* __restrict__ colsum,
                        float* __restrict__ dW, float* __restrict__ db, int accumulate,
                        int n_fo_total) {
  const size_t total = (size_t)g.kt * g.kf * g.C_in * g.C_out;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) {
    const int co = (int)(idx % g.C_out);
    size_t r = idx / g.C_out;
    const int ci = (int)(r % g.C_in); r /= g.C_in;
    const int df = (int)(r % g.kf), dt = (int)(r / g.kf);
    double acc = 0.0;
    for (int fo = 0; fo < g.F_out; ++fo) {
      const int fi = g.sf * fo + df - g.pf;
      if (fi < 0 || fi >= g.F_in) continue;
      acc += (double)dband[((size_t)dt * g.Ki + (size_t)fi * g.C_in + ci) * g.Ko +
                           (size_t)fo * g.C_out + co];
    }
    dW[idx] = (accumulate ? dW[idx] : 0.f) + (float)acc;
  }
  if (db && idx < (size_t)g.C_out) {
    double acc = 0.0;
    for (int fo = 0; fo < n_fo_total; ++fo) acc += (double)colsum[(size_t)fo * g.C_out + idx];
    db[idx] = (float)acc;
  }
}

int grid_for(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 16384) b = 16384;
  return b < 1 ? 1 : (int)b;
}

bool ws_ok(const Ws& w, void* workspace, size_t ws_bytes) {
  if (workspace && ws_bytes >= w.end && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0) return true;
  asr_set_error("conv2d: workspace %zu < %zu bytes (or not 256-byte aligned)", ws_bytes, w.end);
  return false;
}

// W -> band matrices of one block -> packed planes (either arrangement) at byte offset pl_off of
// the plane region, bias_band.  first: also max|W| and the planes' scale (the same for every
// block: later blocks leave the slots alone -- a GEMM of an earlier block may be reading them).
int build_band(const Geo& g, const asr_conv2d_args* a, const Ws& w, char* ws, bool need_f,
               bool need_dg, size_t pl_off, bool first, hipStream_t stream) {
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  float* band_f = reinterpret_cast<float*>(ws + w.band_f);
  float* band_dg = reinterpret_cast<float*>(ws + w.band_dg);
  ASR_CHECK_HIP(hipMemsetAsync(band_dg, 0, (size_t)g.Ki * g.kt * g.Ko_p * 4, stream));
  hipLaunchKernelGGL(conv_band_kernel, dim3(grid_for((size_t)g.kt * g.Ki_p * g.Ko)), dim3(256), 0,
                     stream, g, a->W, a->bias, band_f, band_dg,
                     first ? reinterpret_cast<float*>(ws + w.bias_band) : (float*)nullptr);
  ASR_CHECK_LAUNCH();
  int rc;
  if (first) {
    rc = asr_absmax(a->W, (int64_t)g.kt * g.kf * g.C_in * g.C_out, scal + 2, stream);
    if (rc) return rc;
  }
  asr_pack_args p;
  if (need_f) {
    p = asr_pack_args{};
    p.src = band_f; p.rows = g.kt * g.Ki_p; p.cols = g.Ko; p.ld = g.Ko;
    p.absmax = scal + 2; p.scale_out = first ? scal + 10 : nullptr;
    p.c_hl = ws + w.bandf_pl + pl_off; p.ldk_c = g.kt * g.Ki_p;
    rc = asr_pack_hl(&p, stream);
    if (rc) return rc;
  }
  if (need_dg) {
    p = asr_pack_args{};
    p.src = band_dg; p.rows = g.Ki; p.cols = g.kt * g.Ko_p; p.ld = g.kt * g.Ko_p;
    p.absmax = scal + 2; p.scale_out = first ? scal + 11 : nullptr;
    p.r_hl = ws + w.banddg_pl + pl_off; p.ldk_r = g.kt * g.Ko_p;
    rc = asr_pack_hl(&p, stream);
    if (rc) return rc;
  }
  return ASR_OK;
}

// The block GEMMs of one pass are independent and each fills only part of the chip (125
// workgroups of the 256 x 256 kernel per block at 64 x 10 s): they are issued on up to three
// streams (as many as fill the chip), longest first onto the less loaded one, -- the caller's and two of the library's own -- between a fork event (everything
// enqueued so far: packs, band builds) and joins (the caller's stream waits for the others).
constexpr int kSideStreams = 2;
hipStream_t side_stream(int i) {
  static hipStream_t streams[16][kSideStreams] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!streams[dev][i] &&
      hipStreamCreateWithFlags(&streams[dev][i], hipStreamNonBlocking) != hipSuccess)
    streams[dev][i] = nullptr;
  return streams[dev][i];
}

struct ForkJoin {
  hipStream_t main;
  hipStream_t lanes[1 + kSideStreams];
  bool used[1 + kSideStreams];
  double load[1 + kSideStreams];
  int n;
  // after everything enqueued on `stream` so far.  wgs = workgroups of one job: lanes beyond
  // what fills the 256 CUs only let a small job grab CUs ahead of a long one (measured: the
  // 128-column forward block started first and the longest block ran last on half the chip)
  int fork(hipStream_t stream, int n_jobs, long long wgs) {
    main = stream; n = 1; lanes[0] = stream; used[0] = true; load[0] = 0.0;
    int want = wgs > 0 ? (int)(256 / wgs) : 1;
    if (want > n_jobs) want = n_jobs;
    for (int i = 0; i < kSideStreams && n < want; ++i) {
      hipStream_t s = side_stream(i);
      if (s) { lanes[n] = s; used[n] = false; load[n] = 0.0; ++n; }
    }
    if (n == 1) return ASR_OK;
    hipEvent_t ev;
    ASR_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ASR_CHECK_HIP(hipEventRecord(ev, main));
    for (int i = 1; i < n; ++i) ASR_CHECK_HIP(hipStreamWaitEvent(lanes[i], ev, 0));
    ASR_CHECK_HIP(hipEventDestroy(ev));
    return ASR_OK;
  }
  // the lane with the least work so far (call with the jobs in descending cost order)
  hipStream_t lane(double cost) {
    int best = 0;
    for (int i = 1; i < n; ++i) if (load[i] < load[best]) best = i;
    load[best] += cost; used[best] = true;
    return lanes[best];
  }
  int join() {
    for (int i = 1; i < n; ++i) {
      if (!used[i]) continue;
      hipEvent_t ev;
      ASR_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      ASR_CHECK_HIP(hipEventRecord(ev, lanes[i]));
      ASR_CHECK_HIP(hipStreamWaitEvent(main, ev, 0));
      ASR_CHECK_HIP(hipEventDestroy(ev));
    }
    return ASR_OK;
  }
};

// block indices in descending order of (column tiles x reduction length)
void by_cost(int n, const double* cost, int* order) {
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) {
      const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
    }
}

int pack_x(const Geo& g, const asr_conv2d_args* a, const Ws& w, char* ws, hipStream_t stream) {
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  const float* amax = a->x_absmax;          // the caller's bound (e.g. the clip of the layer
  if (!amax) {                               // below), else a pass over x
    const int rc = asr_absmax(a->x, (int64_t)g.T_in * g.n_pad * g.Ki, scal + 0, stream);
    if (rc) return rc;
    amax = scal + 0;
  }
  const size_t items = (size_t)g.st * g.S * g.n_pad * (g.Ki_p / 16);
  hipLaunchKernelGGL(conv_pack_kernel<0>, dim3(grid_for(items)), dim3(256), 0, stream, g, a->x,
                     (const float*)nullptr, 0.f, amax, scal + 8,
                     reinterpret_cast<_Float16*>(ws + w.xpl), (float*)nullptr);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

int pack_dz(const Geo& g, const asr_conv2d_args* a, const Ws& w, char* ws, hipStream_t stream) {
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  const int rc = asr_absmax(a->dy, (int64_t)g.M * g.Ko, scal + 1, stream);   // |dz| <= |dy|
  if (rc) return rc;
  const unsigned blocks = (unsigned)((size_t)kPartRows * (g.Ko_p / 16) / 256);   // exact
  hipLaunchKernelGGL(conv_pack_kernel<1>, dim3(blocks), dim3(256), 0, stream, g, a->dy,
                     a->z, a->clip, scal + 1, scal + 9,
                     reinterpret_cast<_Float16*>(ws + w.dzpl),
                     reinterpret_cast<float*>(ws + w.colpart));
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}

}  // namespace

extern "C" int asr_conv2d_out_shape(const asr_conv2d_args* a, int* T_out, int* F_out) {
  Geo g;
  ASR_CHECK_ARG(make_geo(a, &g), "conv2d: bad geometry (n_pad %% 16, kt <= 16, F*C %% 4 == 0)");
  if (T_out) *T_out = g.T_out;
  if (F_out) *F_out = g.F_out;
  return ASR_OK;
}

extern "C" size_t asr_conv2d_workspace_bytes(const asr_conv2d_args* a) {
  Geo g;
  if (!make_geo(a, &g)) return 0;
  return make_ws(g).end;
}

extern "C" int asr_conv2d_fwd(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                              asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Geo g;
  ASR_CHECK_ARG(make_geo(a, &g), "conv2d: bad geometry (n_pad %% 16, kt <= 16, F*C %% 4 == 0)");
  ASR_CHECK_ARG(a->x && a->W && (a->z || (a->y && a->clip > 0.f)) && (a->y || a->clip <= 0.f),
                "conv2d fwd: null pointer");
  // z == NULL (or y): the clipped ReLU leaves the GEMM epilogue, only y is written
  const bool fused = a->clip > 0.f && (a->z == nullptr || a->z == a->y);
  float* zout = fused ? a->y : a->z;
  const Ws w = make_ws(g);
  if (!ws_ok(w, workspace, ws_bytes)) return ASR_ERR_WORKSPACE;
  char* ws = reinterpret_cast<char*>(workspace);
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  int rc;
  if (!a->reuse_x) { rc = pack_x(g, a, w, ws, stream); if (rc) return rc; }
  // block b: output columns [fo0 C_out, +Ko_b) of z from input columns [fi0 C_in, +Ki_b) of the
  // x planes (the whole band when nblk == 1; then Ki_b may be padded up to Ki_p: zero columns).
  // All bands first (the caller's stream), then the block GEMMs side by side.
  size_t pl_off[4];
  size_t off = 0;
  for (int b = 0; b < g.nblk; ++b) {
    const Geo gb = block_geo(g, b);
    pl_off[b] = off;
    rc = build_band(gb, a, w, ws, true, false, off, b == 0, stream);
    if (rc) return rc;
    off += asr_align_up((size_t)gb.Ko * g.kt * gb.Ki_p * 4, 256);
  }
  double cost[4];
  int order[4];
  for (int b = 0; b < g.nblk; ++b)
    cost[b] = (double)((g.b_nfo[b] * g.C_out + 255) / 256) * g.b_nfi[b] *
              (g.b_nfo[b] * g.C_out >= 256 ? 1.0 : 0.5);      // (a 128-wide block: half tiles)
  by_cost(g.nblk, cost, order);
  ForkJoin fj;
  rc = fj.fork(stream, g.nblk, ((g.M + 255) / 256) * ((block_geo(g, 0).Ko + 255) / 256));
  if (rc) return rc;
  for (int ob = 0; ob < g.nblk; ++ob) {
    const int b = order[ob];
    const Geo gb = block_geo(g, b);
    const int kseg = g.nblk == 1 ? g.Ki_p : gb.Ki;
    asr_gemm_hl_args h = {};
    h.M = (int)g.M; h.N = gb.Ko; h.K = g.kt * kseg;
    h.a_hl = ws + w.xpl + (size_t)(g.b_fi0[b] * g.C_in / 16) * 64; h.lda = g.Ki_p;
    h.b_hl = ws + w.bandf_pl + pl_off[b]; h.ldb = g.kt * gb.Ki_p;
    h.a_scale = scal + 8; h.b_scale = scal + 10;
    h.C = zout + (size_t)g.b_fo0[b] * g.C_out; h.ldc = g.Ko; h.alpha = 1.f; h.beta = 0.f;
    h.bias = reinterpret_cast<float*>(ws + w.bias_band);
    h.clamp_hi = fused ? a->clip : 0.f;
    h.a_seg_k = kseg;
    for (int dt = 0; dt < g.kt; ++dt) h.a_seg_row[dt] = tap_row(g, dt);
    rc = asr_gemm_hl(&h, nullptr, 0, fj.lane(cost[b]));
    if (rc) return rc;
  }
  rc = fj.join();
  if (rc) return rc;
  if (fused) {
  } else if (a->clip > 0.f) {
    const size_t n4 = (size_t)g.M * g.Ko / 4;
    hipLaunchKernelGGL(conv_act_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, a->z, a->y, n4,
                       a->clip);
    ASR_CHECK_LAUNCH();
  } else if (a->y && a->y != a->z) {
    ASR_CHECK_HIP(hipMemcpyAsync(a->y, a->z, (size_t)g.M * g.Ko * 4, hipMemcpyDeviceToDevice, stream));
  }
  return ASR_OK;
}

extern "C" int asr_conv2d_dgrad(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                                asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Geo g;
  ASR_CHECK_ARG(make_geo(a, &g), "conv2d: bad geometry (n_pad %% 16, kt <= 16, F*C %% 4 == 0)");
  ASR_CHECK_ARG(a->W && a->z && a->dy && a->dx, "conv2d dgrad: null pointer");
  // the time-strided layer of the front-end is its first: nothing trainable lies upstream
  ASR_CHECK_ARG(a->st == 1, "conv2d dgrad: time stride %d is not implemented (only the first "
                "layer of the front-end is strided in time, and its input is data)", a->st);
  const Ws w = make_ws(g);
  if (!ws_ok(w, workspace, ws_bytes)) return ASR_ERR_WORKSPACE;
  char* ws = reinterpret_cast<char*>(workspace);
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  int rc;
  if (!a->reuse_dz) { rc = pack_dz(g, a, w, ws, stream); if (rc) return rc; }
  // dx[(t, n)] = sum_dt dz[(t - dt + pt, n)] band[dt]^T: tap dt reads the dz planes
  // (kt - 1 - dt) frames below their first (zero-padded) frame.  Row blocks: block b writes
  // the input frequencies [fi0, fi0 + nfi) of dx from the output frequencies that reach them;
  // all bands first (the caller's stream), then the block GEMMs side by side.
  size_t pl_off[8];
  size_t off = 0;
  for (int b = 0; b < g.nwblk; ++b) {
    const Geo gb = wblock_geo(g, b);
    pl_off[b] = off;
    rc = build_band(gb, a, w, ws, false, true, off, b == 0, stream);
    if (rc) return rc;
    off += asr_align_up((size_t)gb.Ki * g.kt * gb.Ko_p * 4, 256);
  }
  double cost[8];
  int order[8];
  for (int b = 0; b < g.nwblk; ++b)
    cost[b] = (double)((g.w_nfi[b] * g.C_in + 255) / 256) * g.w_nfo[b];
  by_cost(g.nwblk, cost, order);
  ForkJoin fj;
  rc = fj.fork(stream, g.nwblk, ((g.M + 255) / 256) * ((wblock_geo(g, 0).Ki + 255) / 256));
  if (rc) return rc;
  for (int ob = 0; ob < g.nwblk; ++ob) {
    const int b = order[ob];
    const Geo gb = wblock_geo(g, b);
    const int kseg = g.nwblk == 1 ? g.Ko_p : gb.Ko;
    asr_gemm_hl_args h = {};
    h.M = (int)g.M; h.N = gb.Ki; h.K = g.kt * kseg;
    h.a_hl = ws + w.dzpl + (size_t)(g.w_fo0[b] * g.C_out / 16) * 64; h.lda = g.Ko_p;
    h.b_hl = ws + w.banddg_pl + pl_off[b]; h.ldb = g.kt * gb.Ko_p;
    h.a_scale = scal + 9; h.b_scale = scal + 11;
    h.C = a->dx + (size_t)g.w_fi0[b] * g.C_in; h.ldc = g.Ki; h.alpha = 1.f; h.beta = 0.f;
    h.a_seg_k = kseg;
    for (int dt = 0; dt < g.kt; ++dt)
      h.a_seg_row[dt] = (long long)(g.padb - (g.kt - 1 - g.pt) + (g.kt - 1 - dt)) * g.n_pad;
    rc = asr_gemm_hl(&h, nullptr, 0, fj.lane(cost[b]));
    if (rc) return rc;
  }
  return fj.join();
}

extern "C" int asr_conv2d_wgrad(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                                asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Geo g;
  ASR_CHECK_ARG(make_geo(a, &g), "conv2d: bad geometry (n_pad %% 16, kt <= 16, F*C %% 4 == 0)");
  ASR_CHECK_ARG(a->x && a->z && a->dy && a->dW, "conv2d wgrad: null pointer");
  const Ws w = make_ws(g);
  if (!ws_ok(w, workspace, ws_bytes)) return ASR_ERR_WORKSPACE;
  char* ws = reinterpret_cast<char*>(workspace);
  float* scal = reinterpret_cast<float*>(ws + w.scal);
  int rc;
  if (!a->reuse_x) { rc = pack_x(g, a, w, ws, stream); if (rc) return rc; }
  if (!a->reuse_dz) { rc = pack_dz(g, a, w, ws, stream); if (rc) return rc; }
  float* dband = reinterpret_cast<float*>(ws + w.dband);
  float* cs = nullptr;
  if (a->db) {
    // column sums of dz: the pack left kPartRows partial rows; folded over fo by the reduce kernel
    cs = reinterpret_cast<float*>(ws + w.band_f);          // (free during wgrad: Ko floats)
    rc = asr_colsum(reinterpret_cast<float*>(ws + w.colpart), kPartRows, g.Ko, g.Ko_p, cs, 0.f,
                    ws + w.colsum, w.colsum_bytes, stream);
    if (rc) return rc;
  }
  const size_t items = (size_t)g.kt * g.kf * g.C_in * g.C_out;
  for (int b = 0; b < g.nwblk; ++b) {
    // d band_b[dt] (Ki_b x Ko_b) = X_dt[:, block's input columns]^T dZ[:, block's output columns]:
    // both operands reduce over their plane ROWS (k_major); the kt taps share dZ and differ in
    // the row shift of the x planes only: ONE batched launch per block
    const Geo gb = wblock_geo(g, b);
    asr_gemm_hl_args h = {};
    h.M = gb.Ki; h.N = gb.Ko; h.K = (int)g.M;
    h.a_hl = ws + w.xpl + (size_t)(g.w_fi0[b] * g.C_in / 16) * 64; h.lda = g.Ki_p;
    h.b_hl = ws + w.dzpl + (size_t)g.padb * g.n_pad * g.Ko_p * 4 +
             (size_t)(g.w_fo0[b] * g.C_out / 16) * 64;
    h.ldb = g.Ko_p;
    h.a_scale = scal + 8; h.b_scale = scal + 9;
    h.C = dband; h.ldc = gb.Ko; h.alpha = 1.f; h.beta = 0.f;
    h.split_k = wgrad_splits(gb);
    h.k_major = 1;
    h.batch = g.kt;
    for (int dt = 0; dt < g.kt; ++dt) h.a_batch_row[dt] = tap_row(g, dt);
    rc = asr_gemm_hl(&h, ws + w.gemm, w.gemm_bytes, stream);
    if (rc) return rc;
    // fold the band gradient back onto the filter taps (block 0 writes, the others add; the
    // bias gradient -- column sums over ALL output frequencies -- with block 0)
    hipLaunchKernelGGL(conv_band_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                       stream, gb, dband, cs, a->dW, b == 0 ? a->db : (float*)nullptr, b > 0 ? 1 : 0,
                       g.F_out);
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}
