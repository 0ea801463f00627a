// K1-K3 MFCC / log-mel front-end -- gfx950.
//
// Replaces preprocessing/audio.py (FBank/MFCC/LogFbank._call, _postprocessing,
// _standarize) and preprocessing/audio_utils.py (preemphasis, framesig, powspec,
// delta) of the reference; tested against oracle/frontend.py, which is pinned
// bit-for-bit on the reference's own NumPy code.
//
// Kernel 1 (fe_frames_kernel): one 64-lane wave per frame.  Pre-emphasis, zero
//   padded framing and the Hamming window are applied while the 400 samples are
//   read (coalesced); the 512-point real FFT is a 256-point complex radix-4
//   Stockham FFT (4 stages x one butterfly per lane) in LDS plus the real-FFT
//   split; power spectrum, frame energy, the (sparse) triangular mel filters,
//   log and the 13-column DCT (lifter folded into the table) all stay in LDS /
//   registers.  Only T x (13 | nfilt[+1]) base features are written.  This per-frame chain
//   runs in FLOAT64 (tables included): the reference computes in float64, and a float32
//   FFT leaves ~1e-7 x the frame's largest bin as absolute error in every bin, which under
//   a nearly empty narrow mel filter became 3.6e-4 after log and standardisation; in float64
//   the standardised features agree with the reference to ~1e-5 (test bar: 1e-4).  The
//   kernel is ~0.3 ms of a 50 ms step, so the fp64 rate is irrelevant.
// Kernel 2 (fe_finalize_kernel), float64 too (the base features stay float64 in the
//   workspace; only the standardised output is float32): one workgroup per utterance.  Deltas and
//   delta-deltas (edge-replicated, audio_utils.py:153-173), stride / context
//   stacking, per-column mean / population-std (float64 accumulators, wave
//   shuffles + LDS) and the normalised write into the time-major (T, N, F) slab,
//   zero-filled past the utterance (pad_sequences 'post').
#include "common.h"

namespace {

constexpr int NFFT = 512;
constexpr int NBINS = NFFT / 2 + 1;     // 257
constexpr int FRAMES_PER_WAVE = 4;
constexpr int WAVES = 4;
constexpr int FRAMES_PER_BLOCK = FRAMES_PER_WAVE * WAVES;
constexpr double kF64Eps = 2.220446049250313e-16;   // np.finfo(float).eps
constexpr int kMelLd = 128;             // filters per row of the transposed mel table
constexpr int kMelPasses = kMelLd / 64;
// workspace tail: twiddles (385 double2) + the transposed compact mel table (<= 257 rows)
constexpr size_t kTwBytes = 6400;                                // 385 double2 -> multiple of 256
constexpr size_t kMelTBytes = (size_t)NBINS * kMelLd * sizeof(double);

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Tables of one call: the FFT twiddles, and the mel matrix in the arrangement the frame
// kernel walks: mel_t[i][jf] = mel[jf][lo_jf + i] for i < hi_jf - lo_jf, else 0 (rows up to the
// widest filter; columns up to kMelLd).  One workgroup.
__global__ void __launch_bounds__(256)
fe_prepare_kernel(int num_filt, const double* __restrict__ mel, const int* __restrict__ mel_range,
                  double2* __restrict__ tw_tab, double* __restrict__ mel_t) {
  __shared__ int s_w[256];
  const int tid = threadIdx.x;
  for (int m = tid; m <= 384; m += 256) {
    double s, c;
    sincospi(-(double)m / 256.0, &s, &c);         // angle = -2 pi m / 512
    tw_tab[m] = make_double2(c, s);
  }
  int wmax = 0;
  for (int jf = tid; jf < num_filt; jf += 256) {
    const int wd = mel_range[2 * jf + 1] - mel_range[2 * jf];
    wmax = wd > wmax ? wd : wmax;
  }
  s_w[tid] = wmax;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_w[tid] = s_w[tid] > s_w[tid + st] ? s_w[tid] : s_w[tid + st];
    __syncthreads();
  }
  wmax = s_w[0] < NBINS ? s_w[0] : NBINS;
  for (int e = tid; e < wmax * kMelLd; e += 256) {
    const int i = e / kMelLd, jf = e % kMelLd;
    double v = 0.0;
    if (jf < num_filt) {
      const int lo = mel_range[2 * jf], hi = mel_range[2 * jf + 1];
      if (lo + i < hi && lo + i < NBINS) v = mel[(size_t)jf * NBINS + lo + i];
    }
    mel_t[e] = v;
  }
}

__global__ void __launch_bounds__(256)
fe_frames_kernel(asr_frontend_cfg cfg, const float* __restrict__ audio,
                 const int* __restrict__ offsets, const int* __restrict__ lengths,
                 const double* __restrict__ window, const double* __restrict__ mel,
                 const int* __restrict__ mel_range, const double* __restrict__ dct,
                 double* __restrict__ base, int max_frames, int fb,
                 const double2* __restrict__ tw_tab, const double* __restrict__ mel_t) {
  __shared__ double2 tw[384 + 1];                // e^{-2 pi i m / 512}, m = 0..384
  __shared__ double2 buf[WAVES][2][256];         // ping-pong complex buffers
  __shared__ double pspec[WAVES][NBINS + 3];
  __shared__ double lmel[WAVES][128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int utt = blockIdx.y;
  const int len = lengths[utt];
  const int off = offsets[utt];
  int nframes = 1;
  if (len > cfg.frame_len)
    nframes = 1 + (len - cfg.frame_len + cfg.frame_step - 1) / cfg.frame_step;
  const int f_block = blockIdx.x * FRAMES_PER_BLOCK;
  if (f_block >= nframes) return;                 // uniform per block

  for (int m = tid; m <= 384; m += 256) tw[m] = tw_tab[m];    // (fe_prepare_kernel)
  // this lane's mel filters (lane, lane + 64): first bin, width, and the width the wave
  // walks (its widest filter): the same for every frame
  int mlo[kMelPasses], mwave[kMelPasses];
#pragma unroll
  for (int ps = 0; ps < kMelPasses; ++ps) {
    const int jf = lane + 64 * ps;
    const bool has = jf < cfg.num_filt;
    mlo[ps] = has ? mel_range[2 * jf] : 0;
    const int width = has ? mel_range[2 * jf + 1] - mlo[ps] : 0;
    mwave[ps] = __builtin_amdgcn_readfirstlane((int)asr_wave_max((float)width));
  }
  __syncthreads();

  double* real_in = reinterpret_cast<double*>(&buf[w][0][0]);   // 512 reals == 256 complex
  // The samples of a frame (and their predecessors, for the pre-emphasis) are fetched into
  // registers one frame AHEAD: the loads of frame fi + 1 fly under the FFT of frame fi.  (Loaded
  // where they were used, a frame waited for 8 dependent round trips to HBM / L2: 11 us per
  // frame and wave, most of the kernel.)
  float xs[NFFT / 64], xq[NFFT / 64];
  auto fetch = [&](int f) {
    const int s0 = f * cfg.frame_step;
    const bool live = f < nframes;
#pragma unroll
    for (int q = 0; q < NFFT / 64; ++q) {
      const int i = lane + 64 * q, idx = s0 + i;
      const bool in = live && i < cfg.frame_len && idx < len;
      xs[q] = in ? audio[off + idx] : 0.f;
      xq[q] = (in && idx > 0) ? audio[off + idx - 1] : 0.f;
    }
  };
  fetch(f_block + w * FRAMES_PER_WAVE);
  for (int fi = 0; fi < FRAMES_PER_WAVE; ++fi) {
    const int f = f_block + w * FRAMES_PER_WAVE + fi;
    const bool live = f < nframes;
    // ---- pre-emphasis + window (zero padded to 512)
    const int s0 = f * cfg.frame_step;
#pragma unroll
    for (int q = 0; q < NFFT / 64; ++q) {
      const int i = lane + 64 * q, idx = s0 + i;
      double v = 0.0;
      if (live && i < cfg.frame_len && idx < len) {
        const double x = (double)xs[q], xp = (double)xq[q];
        v = (idx > 0 ? x - cfg.pre_emph * xp : x) * window[i];
      }
      real_in[i] = v;
    }
    if (fi + 1 < FRAMES_PER_WAVE) fetch(f + 1);
    __syncthreads();
    // ---- 256-point complex FFT, radix-4 Stockham, Ns = 1, 4, 16, 64
    int cur = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int Ns = 1 << (2 * s);
      const double2* in = buf[w][cur];
      double2* out = buf[w][cur ^ 1];
      const int j = lane;
      const int k = j & (Ns - 1);
      double2 v0 = in[j], v1 = in[j + 64], v2 = in[j + 128], v3 = in[j + 192];
      if (s > 0) {
        const int tstep = (128 / Ns) * k;        // index into tw (512-based): 2*(64/Ns)*k
        v1 = cmul(v1, tw[tstep]);
        v2 = cmul(v2, tw[2 * tstep]);
        v3 = cmul(v3, tw[3 * tstep]);
      }
      const double2 t0 = make_double2(v0.x + v2.x, v0.y + v2.y);
      const double2 t1 = make_double2(v0.x - v2.x, v0.y - v2.y);
      const double2 t2 = make_double2(v1.x + v3.x, v1.y + v3.y);
      const double2 d = make_double2(v1.x - v3.x, v1.y - v3.y);
      const double2 t3 = make_double2(d.y, -d.x);  // -i * (v1 - v3)
      const int base_idx = ((j - k) << 2) + k;   // (j / Ns) * Ns * 4 + k
      out[base_idx] = make_double2(t0.x + t2.x, t0.y + t2.y);
      out[base_idx + Ns] = make_double2(t1.x + t3.x, t1.y + t3.y);
      out[base_idx + 2 * Ns] = make_double2(t0.x - t2.x, t0.y - t2.y);
      out[base_idx + 3 * Ns] = make_double2(t1.x - t3.x, t1.y - t3.y);
      cur ^= 1;
      __syncthreads();
    }
    // ---- real-FFT split + power spectrum / nfft, frame energy
    const double2* Z = buf[w][cur];
    double esum = 0.0;
    for (int k = lane; k < NBINS; k += 64) {
      const double2 zk = Z[k & 255];
      const double2 zc = Z[(256 - k) & 255];
      const double2 zn = make_double2(zc.x, -zc.y);                 // conj
      const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y + zn.y));
      const double2 dd = make_double2(zk.x - zn.x, zk.y - zn.y);
      const double2 o = make_double2(0.5 * dd.y, -0.5 * dd.x);      // -i/2 * (zk - zn)
      const double2 wo = cmul(tw[k], o);
      const double xr = e.x + wo.x, xi = e.y + wo.y;
      const double p = (xr * xr + xi * xi) * (1.0 / NFFT);
      pspec[w][k] = p;
      esum += p;
    }
    esum = asr_wave_sum_d(esum);
    if (esum == 0.0) esum = kF64Eps;
    __syncthreads();
    // ---- mel filterbank (triangles are sparse: only [lo, hi) bins) + log
    // filter jf = sum over its bins lo .. hi-1, in that order; the weights come from the
    // transposed compact table mel_t[i][jf] = mel[jf][lo + i] (0 from the filter's width on:
    // adding x * 0 changes nothing), so that the loop is uniform over the wave, its loads
    // coalesced and independent of each other (the per-lane `for k in [lo, hi)` over the dense
    // rows was a chain of dependent L1 round trips: the slowest part of the kernel)
#pragma unroll
    for (int ps = 0; ps < kMelPasses; ++ps) {
      const int jf = lane + 64 * ps;
      if (64 * ps >= cfg.num_filt) break;
      double acc = 0.0;
      const double* wt = mel_t + jf;
#pragma unroll 4
      for (int i = 0; i < mwave[ps]; ++i) {
        int k = mlo[ps] + i;
        k = k < NBINS ? k : NBINS - 1;
        acc += pspec[w][k] * wt[(size_t)i * kMelLd];
      }
      if (jf < cfg.num_filt) {
        if (acc == 0.0) acc = kF64Eps;
        lmel[w][jf] = log(acc);
      }
    }
    __syncthreads();
    // ---- write base features
    if (live) {
      double* row = base + ((size_t)utt * max_frames + f) * fb;
      const double le = log(esum + cfg.eps);
      if (cfg.kind == 0) {
        for (int c = lane; c < cfg.num_cep; c += 64) {
          double acc = 0.0;
          for (int jf = 0; jf < cfg.num_filt; ++jf)
            acc += lmel[w][jf] * dct[jf * cfg.num_cep + c];
          if (c == 0 && cfg.append_energy) acc = le;
          row[c] = acc;
        }
      } else {
        for (int c = lane; c < cfg.num_filt; c += 64) row[c] = lmel[w][c];
        if (cfg.append_energy && lane == 0) row[cfg.num_filt] = le;
      }
    }
    __syncthreads();
  }
}

// full[t][0:fb] = base, [fb:2fb] = delta, [2fb:3fb] = delta-delta
__device__ __forceinline__ double delta_at(const double* __restrict__ x, int ld, int T, int t,
                                           int c) {
  // sum_{n=-2..2} n * x[clamp(t+n)] / 10, same association order as the reference
  double acc = 0.0;
#pragma unroll
  for (int n = -2; n <= 2; ++n) {
    int tt = t + n;
    tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
    acc += (double)n * x[(size_t)tt * ld + c];
  }
  return acc / 10.0;
}

// Workgroup (utterance, block of CX output columns): 1024 / CX row walkers per column.  (One
// workgroup per utterance walked 62 rows per thread and pass on 64 of the 256 CUs.)  CX = 16
// for wide outputs (64-byte runs of the slab rows: 80 log-mel columns 141 -> 39 us), 8 for
// narrow ones (more workgroups: 39 MFCC columns 69 -> 36 us).
constexpr int kFinalizeThreads = 1024;

template <int CX>
__global__ void __launch_bounds__(1024)
fe_finalize_kernel(asr_frontend_cfg cfg, const int* __restrict__ lengths,
                   double* __restrict__ full, int max_frames, int fb, int ffull,
                   float* __restrict__ out, int t_out, int n_pad, int f_out,
                   int* __restrict__ out_frames) {
  constexpr int TY = kFinalizeThreads / CX;
  const int utt = blockIdx.x;
  const int c0 = blockIdx.y * CX;
  const int tid = threadIdx.x;
  const int cx = tid % CX, ty = tid / CX;
  const int len = lengths[utt];
  int T = 1;
  if (len > cfg.frame_len) T = 1 + (len - cfg.frame_len + cfg.frame_step - 1) / cfg.frame_step;
  if (T > max_frames) T = max_frames;
  double* x = full + (size_t)utt * max_frames * ffull;
  const int col = c0 + cx;                    // this thread's output column
  const int cfull = col % ffull;              // ... is column cfull of the (base | d | dd) rows
  // ---- deltas of the columns this workgroup reads (global scratch, visible block-wide after
  // the barriers).  A delta-delta column needs its delta column, which another workgroup may
  // own: this one computes it as well (workgroups that share a column write equal values).
  if (cfg.d) {
    const bool mine = col < f_out && cfull >= fb;
    const int dcol = cfull >= 2 * fb ? cfull - fb : cfull;        // the delta column involved
    if (mine)
      for (int t = ty; t < T; t += TY) x[(size_t)t * ffull + dcol] = delta_at(x, ffull, T, t, dcol - fb);
    __syncthreads();
    if (mine && cfull >= 2 * fb)
      for (int t = ty; t < T; t += TY)
        x[(size_t)t * ffull + cfull] = delta_at(x + fb, ffull, T, t, cfull - 2 * fb);
    __syncthreads();
  }
  const int stride = cfg.stride < 1 ? 1 : cfg.stride;
  const int Ts = (T + stride - 1) / stride;       // frames after feats[::stride]
  auto get = [&](int ts) -> double {
    const int cslot = col / ffull;
    const int src = ts + cslot - cfg.num_context;
    if (src < 0 || src >= Ts) return 0.0;
    return x[(size_t)(src * stride) * ffull + cfull];
  };
  // ---- column statistics: thread (cx, ty) walks t = ty, ty+TY, ... of column c0 + cx
  __shared__ double s_sum[TY][CX];
  __shared__ double s_sq[TY][CX];
  __shared__ double s_mx[TY][CX];
  __shared__ double s_mean[CX];
  __shared__ double s_inv[CX];
  double sum = 0.0, mn = 1e300, mx = -1e300;
  if (col < f_out)
    for (int ts = ty; ts < Ts; ts += TY) {
      const double v = get(ts);
      sum += v;
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
  s_sum[ty][cx] = sum;
  s_sq[ty][cx] = mn;            // (s_sq doubles as scratch for the column minimum ...)
  s_mx[ty][cx] = mx;
  __syncthreads();
  double mean = 0.0;
  if (col < f_out) {
    double tot = 0.0, lo = 1e300, hi = -1e300;
    for (int i = 0; i < TY; ++i) {
      tot += s_sum[i][cx];
      lo = s_sq[i][cx] < lo ? s_sq[i][cx] : lo;
      hi = s_mx[i][cx] > hi ? s_mx[i][cx] : hi;
    }
    // a constant column (an empty mel filter: log(eps) in every frame) has exactly its
    // value as mean -- NumPy's pairwise sum of equal terms is exact there -- so that it
    // standardises to exactly 0
    mean = lo == hi ? lo : tot / Ts;
  }
  __syncthreads();              // (... before s_sq is reused for the squares)
  double sq = 0.0;
  if (col < f_out)
    for (int ts = ty; ts < Ts; ts += TY) {
      const double dlt = get(ts) - mean;
      sq += dlt * dlt;
    }
  s_sq[ty][cx] = sq;
  __syncthreads();
  if (ty == 0 && col < f_out) {
    double tot = 0.0;
    for (int i = 0; i < TY; ++i) tot += s_sq[i][cx];
    const double var = tot / Ts;
    s_mean[cx] = cfg.mean_norm ? mean : 0.0;
    // (without mean_norm the std is still taken about the true mean -- np.std -- only the
    // shift is skipped)
    const double sd = sqrt(var);
    s_inv[cx] = cfg.var_norm ? sd + cfg.eps : 1.0;     // the divisor (audio.py:70-75)
  }
  __syncthreads();
  if (col < f_out) {
    const double mu = s_mean[cx], den = s_inv[cx];
    for (int ts = ty; ts < t_out; ts += TY) {
      float v = 0.f;
      if (ts < Ts) v = (float)((get(ts) - mu) / den);
      out[((size_t)ts * n_pad + utt) * f_out + col] = v;
    }
  }
  if (tid == 0 && blockIdx.y == 0 && out_frames) out_frames[utt] = Ts < t_out ? Ts : t_out;
}

__global__ void fe_zero_pad_rows_kernel(float* __restrict__ out, int t_out, int n_utt,
                                        int n_pad, int f_out) {
  // rows n in [n_utt, n_pad) of every time step are batch padding: zero them.
  const int padn = n_pad - n_utt;
  const size_t total = (size_t)t_out * padn * f_out;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % f_out);
    const size_t r = e / f_out;
    const int n = n_utt + (int)(r % padn);
    const int t = (int)(r / padn);
    out[((size_t)t * n_pad + n) * f_out + c] = 0.f;
  }
}

int base_cols(const asr_frontend_cfg* cfg) {
  return cfg->kind == 0 ? cfg->num_cep : cfg->num_filt + (cfg->append_energy ? 1 : 0);
}

}  // namespace

extern "C" int asr_frontend_num_frames(int samples, int frame_len, int frame_step) {
  if (samples <= frame_len) return 1;
  return 1 + (samples - frame_len + frame_step - 1) / frame_step;
}

extern "C" int asr_frontend_num_feats(const asr_frontend_cfg* cfg) {
  return base_cols(cfg) * (1 + (cfg->d ? 1 : 0) + ((cfg->d && cfg->dd) ? 1 : 0));
}

extern "C" size_t asr_frontend_workspace_bytes(const asr_frontend_cfg* cfg, int n_utt,
                                               int max_frames) {
  return asr_align_up((size_t)n_utt * max_frames * asr_frontend_num_feats(cfg) *
                          sizeof(double), 256) + kTwBytes + kMelTBytes;
}

extern "C" int asr_frontend_features(const asr_frontend_cfg* cfg, const float* audio,
                                     const int* offsets, const int* lengths,
                                     const int* host_lengths, int n_utt, int n_pad,
                                     const double* window, const double* mel,
                                     const int* mel_range, const double* dct,
                                     float* out, int t_out, int* out_frames,
                                     void* workspace, size_t ws_bytes,
                                     asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_ARG(cfg && audio && offsets && lengths && host_lengths && window && mel &&
                    mel_range && out && workspace, "frontend: null pointer");
  ASR_CHECK_ARG(cfg->nfft == NFFT, "frontend: nfft must be 512 (got %d)", cfg->nfft);
  ASR_CHECK_ARG(cfg->frame_len > 0 && cfg->frame_len <= NFFT && cfg->frame_step > 0,
                "frontend: bad frame_len/frame_step");
  ASR_CHECK_ARG(cfg->num_filt > 0 && cfg->num_filt <= 128, "frontend: num_filt > 128");
  ASR_CHECK_ARG(cfg->kind == 1 || (cfg->num_cep > 0 && cfg->num_cep <= 64 && dct),
                "frontend: bad num_cep / missing dct");
  ASR_CHECK_ARG(n_utt > 0 && n_pad >= n_utt && t_out > 0, "frontend: bad batch shape");
  int max_frames = 1;
  for (int i = 0; i < n_utt; ++i) {
    ASR_CHECK_ARG(host_lengths[i] > 1, "frontend: utterance %d has < 2 samples", i);
    const int nf = asr_frontend_num_frames(host_lengths[i], cfg->frame_len, cfg->frame_step);
    if (nf > max_frames) max_frames = nf;
  }
  const int fb = base_cols(cfg);
  const int ffull = asr_frontend_num_feats(cfg);
  const int f_out = ffull * (2 * cfg->num_context + 1);
  const size_t need = asr_frontend_workspace_bytes(cfg, n_utt, max_frames);
  if (ws_bytes < need) {
    asr_set_error("frontend: workspace %zu < %zu bytes", ws_bytes, need);
    return ASR_ERR_WORKSPACE;
  }
  double* full = reinterpret_cast<double*>(workspace);
  char* tail = reinterpret_cast<char*>(workspace) + (need - kTwBytes - kMelTBytes);
  double2* tw_tab = reinterpret_cast<double2*>(tail);
  double* mel_t = reinterpret_cast<double*>(tail + kTwBytes);
  hipLaunchKernelGGL(fe_prepare_kernel, dim3(1), dim3(256), 0, stream, cfg->num_filt, mel,
                     mel_range, tw_tab, mel_t);
  ASR_CHECK_LAUNCH();
  dim3 grid((max_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK, n_utt);
  hipLaunchKernelGGL(fe_frames_kernel, grid, dim3(256), 0, stream, *cfg, audio, offsets,
                     lengths, window, mel, mel_range, dct, full, max_frames, ffull, tw_tab, mel_t);
  ASR_CHECK_LAUNCH();
  if (f_out >= 64)
    hipLaunchKernelGGL(fe_finalize_kernel<16>, dim3(n_utt, (f_out + 15) / 16),
                       dim3(kFinalizeThreads), 0, stream, *cfg, lengths, full, max_frames, fb,
                       ffull, out, t_out, n_pad, f_out, out_frames);
  else
    hipLaunchKernelGGL(fe_finalize_kernel<8>, dim3(n_utt, (f_out + 7) / 8),
                       dim3(kFinalizeThreads), 0, stream, *cfg, lengths, full, max_frames, fb,
                       ffull, out, t_out, n_pad, f_out, out_frames);
  ASR_CHECK_LAUNCH();
  if (n_pad > n_utt) {
    hipLaunchKernelGGL(fe_zero_pad_rows_kernel, dim3(256), dim3(256), 0, stream, out, t_out,
                       n_utt, n_pad, f_out);
    ASR_CHECK_LAUNCH();
  }
  return ASR_OK;
}
