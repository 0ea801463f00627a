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
// Design (CDNA4), second iteration (the first one let every wave gather its own
// K slice as 8-byte {value,tag} granules: 8 MB of fabric reads per step at cfg2,
// 11 us/step; measured, see profiles/):
//  * A layer is a set of independent CHAINS (direction, 16-row batch tile).  A
//    chain is split over 1024-thread workgroups (one per CU) by hidden units; each
//    wave keeps its slice of U stationary in VGPRs as the MFMA A-operand for the
//    whole sequence (v_mfma_f32_16x16x4_f32: exact fp32; C/D layout row =
//    4*(lane>>4)+reg, col = lane&15, so with gate columns ordered unit*4+gate a
//    lane ends up with the four gates of ONE (unit, sample): gate math is
//    lane-local).
//  * Forward: workgroups exchange h_t (H x 16 words per chain and step).  The
//    whole workgroup gathers the chain's h ONCE into LDS (one 16-byte
//    agent-scope load per thread), every wave reads its MFMA B-operand from LDS,
//    K is split over 4 waves and reduced through LDS.
//  * Backward: a workgroup owns 16 units = 64 gate columns j.  It multiplies its
//    own dz_J (local) with U[:, J] for ALL H outputs and publishes the partial
//    dh tiles; each consumer sums the partials addressed to its units.  Exchange
//    volume is H x 16 words per producer -- the same as forward, instead of the
//    4H-wide dz vector.
//  * Hand-off protocol: every exchanged fp32 word carries the step tag in its
//    mantissa LSB (value perturbed by <= 1 ulp = 6e-8 relative); words are written
//    with agent-scope (sc1, write-through) stores and polled with agent-scope
//    16-byte loads; a word is its own flag, no fences, no dependence on
//    workgroup placement (MI355X guide, Guideline 16 / R2 "the data IS the flag").
//    Two slots (step parity) suffice: a producer is at most one step ahead of its
//    slowest consumer, so a slot holds either step s or s-2, which differ in bit
//    (s>>1)&1.  The buffer is memset to 0xFF (tag 1) before each launch.
//  * Every spin is bounded by the wall clock; a give-up is recorded in the
//    workspace status word and the kernel finishes without polling.
//  * mode 1 launches one step per kernel (exchange through the same buffer, made
//    visible by the kernel boundary, no polling): the always-safe fallback with
//    bit-identical arithmetic.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kWaves = 16;                 // 1024-thread workgroups
constexpr int kThreads = kWaves * 64;
constexpr int kMaxR = 32;                  // MFMA steps per wave and tile
constexpr int kSc1 = 16;                   // buffer-op cache policy: SC1 (agent scope)

struct LstmParams {
  int T, n_pad, H, NB;
  int KS, UGW, R;          // fwd: K split, unit groups per WG, MFMAs per wave
  int P;                   // workgroups per chain
  int s_begin, s_count;
  int chain_begin;
  int poll;                // 1: persistent (poll tags); 0: one step per launch
  const float* U;
  const float* mask_u;
  const float* zx;
  float* y;
  float* cell;
  float* gates;
  const float* dy;
  float* dz;
  float* dc_state;
  unsigned* xbuf;          // exchange buffer (words)
  long long xchain_words;  // words per chain (2 slots)
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
__device__ __forceinline__ unsigned tag_word(float v, unsigned tag) {
  return (__float_as_uint(v) & ~1u) | tag;
}

// Polls one 16-byte group of exchanged words until all four carry `tag`.
__device__ __forceinline__ u32x4 poll_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off,
                                           unsigned tag, int poll, bool& dead, int* status) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, kSc1);
  if (!poll || dead) return v;
  long long t0 = 0;
  bool timing = false;
  while (((v[0] & 1u) != tag) | ((v[1] & 1u) != tag) | ((v[2] & 1u) != tag) |
         ((v[3] & 1u) != tag)) {
    if (!timing) { t0 = wall_clock64(); timing = true; }
    else if (wall_clock64() - t0 > kSpinTicks) {
      dead = true;
      atomicExch(status, 1);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, kSc1);
  }
  return v;
}

// ---------------------------------------------------------------------------
// forward.  WG = UGW unit groups (4 units each) x KS K-splits.
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int UG = H >> 2;
  const int KS = p.KS, R = p.R;
  const int wg = blockIdx.x % p.P;
  const int chain = p.chain_begin + blockIdx.x / p.P;
  const int dir = chain / p.NB, bt = chain % p.NB;
  const int ugl = w / KS, kq = w % KS;
  const int ug = wg * p.UGW + ugl;
  const bool ug_ok = ug < UG;
  const int n = bt * 16 + nl;
  const int u = 4 * ug + g;
  const int kbase = (kq * 4 + g) * R;            // this lane's K slice [kbase, kbase+R)
  const int HS = H + 4;                          // LDS row stride of the h tile
  float* hbuf = lds;                             // [16][HS]
  float4* red = reinterpret_cast<float4*>(lds + 16 * HS);   // [UGW][KS-1][64]

  float uf[kMaxR];
#pragma unroll
  for (int kk = 0; kk < kMaxR; ++kk) {
    const int k = kbase + kk;
    uf[kk] = (ug_ok && kk < R && k < H) ? p.U[((size_t)(dir * H + k)) * H4 + 16 * ug + nl] : 0.f;
  }
  const bool owner = ug_ok && kq == 0;           // this wave finishes the cell update
  float mask = 1.f;
  if (owner && p.mask_u) mask = p.mask_u[((size_t)dir * p.n_pad + n) * H + u];
  float c = 0.f;
  bool dead = false;
  unsigned* xch = p.xbuf + (size_t)chain * p.xchain_words;    // [2][16][H]
  const int slot_words = 16 * H;
  const int s_end = p.s_begin + p.s_count;

  if (owner && p.s_begin > 0) {
    const int tpp = dir == 0 ? p.s_begin - 1 : p.T - p.s_begin;
    c = p.cell[(((size_t)tpp * p.n_pad + n) * 2 + dir) * H + u];
  }
  // input projection rows are streamed from HBM two steps ahead of their use
  auto load_zx = [&](int ss) -> float4 {
    if (!owner || ss >= s_end) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int tt = dir == 0 ? ss : p.T - 1 - ss;
    return *reinterpret_cast<const float4*>(
        p.zx + (((size_t)tt * p.n_pad + n) * 2 + dir) * H4 + 4 * u);
  };
  float4 zx_n1 = load_zx(p.s_begin);
  float4 zx_n2 = load_zx(p.s_begin + 1);
  for (int s = p.s_begin; s < s_end; ++s) {
    const int t = dir == 0 ? s : p.T - 1 - s;
    const float4 zx4 = zx_n1;
    zx_n1 = zx_n2;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      // ---- gather h_{s-1} (already masked by the producer) into LDS, once per WG
      const unsigned tag = (unsigned)((s - 1) >> 1) & 1u;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)((s - 1) & 1) * slot_words, 0, slot_words * 4, 0x00020000);
      for (int e = 4 * tid; e < slot_words; e += 4 * kThreads) {
        const u32x4 v = poll_b128(rsrc, (unsigned)e * 4u, tag, p.poll, dead, p.status);
        const int row = e / H, col = e - row * H;
        *reinterpret_cast<float4*>(hbuf + row * HS + col) =
            make_float4(__uint_as_float(v[0] & ~1u), __uint_as_float(v[1] & ~1u),
                        __uint_as_float(v[2] & ~1u), __uint_as_float(v[3] & ~1u));
      }
      __syncthreads();
      zx_n2 = load_zx(s + 2);
      if (ug_ok) {
        const float* hrow = hbuf + nl * HS + kbase;
#pragma unroll
        for (int kk = 0; kk < kMaxR; kk += 4) {
          if (kk < R) {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kbase + kk < H) hv = *reinterpret_cast<const float4*>(hrow + kk);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk], hv.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk + 1], hv.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk + 2], hv.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[kk + 3], hv.w, acc1, 0, 0, 0);
          }
        }
      }
    } else {
      zx_n2 = load_zx(s + 2);
    }
    f32x4 a = acc0 + acc1;
    if (KS > 1) {
      if (kq > 0) red[(ugl * (KS - 1) + kq - 1) * 64 + lane] = make_float4(a[0], a[1], a[2], a[3]);
      __syncthreads();
      if (kq == 0) {
        for (int q = 0; q < KS - 1; ++q) {
          const float4 r = red[(ugl * (KS - 1) + q) * 64 + lane];
          a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
        }
      }
    }
    if (owner) {
      const float gi = hard_sigmoid(a[0] + zx4.x);
      const float gf = hard_sigmoid(a[1] + zx4.y);
      const float gg = fast_tanh(a[2] + zx4.z);
      const float go = hard_sigmoid(a[3] + zx4.w);
      c = gf * c + gi * gg;
      const float h = go * fast_tanh(c);
      if (s + 1 < p.T) {
        // lanes nl, nl+16, nl+32, nl+48 hold units 4ug..4ug+3 of sample nl: collect
        // them in lane nl so the hand-off is ONE 16-byte write-through store per row
        const unsigned wtag = (unsigned)(s >> 1) & 1u;
        const unsigned w0 = tag_word(h * mask, wtag);
        u32x4 o;
        o[0] = w0;
        o[1] = (unsigned)__shfl_down((int)w0, 16, 64);
        o[2] = (unsigned)__shfl_down((int)w0, 32, 64);
        o[3] = (unsigned)__shfl_down((int)w0, 48, 64);
        if (lane < 16) {
          __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
              xch + (size_t)(s & 1) * slot_words, 0, slot_words * 4, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(o, wr, (unsigned)(nl * H + 4 * ug) * 4u, 0, kSc1);
        }
      }
      const size_t row = (size_t)t * p.n_pad + n;
      p.y[row * H2 + dir * H + u] = h;
      p.cell[(row * 2 + dir) * H + u] = c;
      *reinterpret_cast<float4*>(p.gates + (row * 2 + dir) * H4 + 4 * u) =
          make_float4(gi, gf, gg, go);
    }
  }
}

// ---------------------------------------------------------------------------
// backward (BPTT).  WG `cw` of a chain owns units [16 cw, 16 cw + 16) = gate
// columns j in [64 cw, 64 cw + 64).  TPW = output tiles (16 units) per wave.
template <int TPW>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int P = p.P;                              // = ceil(H / 16)
  const int cw = blockIdx.x % P;
  const int chain = p.chain_begin + blockIdx.x / P;
  const int dir = chain / p.NB, bt = chain % p.NB;
  constexpr int DZS = 68;                         // LDS row stride of the dz tile
  float* dzl = lds;                               // [16 n][DZS] own gate gradients
  float* part = lds + 16 * DZS;                   // [P][256] gathered partial dh

  // stationary A fragments: rows k = 16*mt + (lane&15), cols j = 64*cw + 16*g + kk
  float uf[TPW][16];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int mt = w + kWaves * i;
    const int krow = 16 * mt + nl;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int j = 64 * cw + 16 * g + kk;
      uf[i][kk] = (mt < P && krow < H && j < H4) ? p.U[((size_t)(dir * H + krow)) * H4 + j] : 0.f;
    }
  }
  // cell-backward threads: tid < 256 -> (n = tid/16, ul = tid%16)
  const bool cellthr = tid < 256;
  const int cn = bt * 16 + (tid >> 4);
  const int cu = 16 * cw + (tid & 15);
  const bool cvalid = cellthr && cu < H;
  float cmask = 1.f;
  if (cvalid && p.mask_u) cmask = p.mask_u[((size_t)dir * p.n_pad + cn) * H + cu];
  float dc = 0.f;
  if (cvalid && p.s_begin > 0) dc = p.dc_state[((size_t)dir * p.n_pad + cn) * H + cu];
  bool dead = false;
  unsigned* xch = p.xbuf + (size_t)chain * p.xchain_words;   // [2][P cons][P prod][256]
  const size_t slot_words = (size_t)P * P * 256;
  const int s_end = p.s_begin + p.s_count;

  for (int s = p.s_begin; s < s_end; ++s) {
    const int t = dir == 0 ? p.T - 1 - s : s;     // reverse of the forward order
    const int tc = dir == 0 ? t - 1 : t + 1;      // forward-order predecessor
    const bool has_cprev = (s + 1 < p.T);
    // ---- independent loads of this step (cell-backward threads)
    float dyv = 0.f, cv = 0.f, cpv = 0.f;
    float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
      const size_t row = (size_t)t * p.n_pad + cn;
      dyv = p.dy[row * H2 + dir * H + cu];
      cv = p.cell[(row * 2 + dir) * H + cu];
      if (has_cprev) cpv = p.cell[(((size_t)tc * p.n_pad + cn) * 2 + dir) * H + cu];
      gt = *reinterpret_cast<const float4*>(p.gates + (row * 2 + dir) * H4 + 4 * cu);
    }
    float dh_rec = 0.f;
    if (s > 0) {
      // ---- gather the partial dh tiles addressed to this WG: [P prod][16 n][16 u]
      const unsigned tag = (unsigned)((s - 1) >> 1) & 1u;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)((s - 1) & 1) * slot_words + (size_t)cw * P * 256, 0, P * 256 * 4,
          0x00020000);
      for (int e = 4 * tid; e < P * 256; e += 4 * kThreads) {
        const u32x4 v = poll_b128(rsrc, (unsigned)e * 4u, tag, p.poll, dead, p.status);
        *reinterpret_cast<float4*>(part + e) =
            make_float4(__uint_as_float(v[0] & ~1u), __uint_as_float(v[1] & ~1u),
                        __uint_as_float(v[2] & ~1u), __uint_as_float(v[3] & ~1u));
      }
      __syncthreads();
      if (cellthr) {
        // element (n, ul) of every producer tile sits at [n*16 + ul] == tid
        for (int pr = 0; pr < P; ++pr) dh_rec += part[pr * 256 + tid];
      }
    }
    // ---- cell backward for own units -> dz (LDS for the MFMA, global for the GEMMs)
    if (cellthr) {
      float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cvalid) {
        const float gi = gt.x, gf = gt.y, gg = gt.z, go = gt.w;
        const float dh = dyv + cmask * dh_rec;
        const float tch = fast_tanh(cv);
        const float d_o = dh * tch;
        const float dcc = dc + dh * go * (1.f - tch * tch);
        const float d_i = dcc * gg, d_g = dcc * gi, d_f = dcc * cpv;
        dc = dcc * gf;
        z4.x = d_i * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
        z4.y = d_f * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
        z4.z = d_g * (1.f - gg * gg);
        z4.w = d_o * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
        *reinterpret_cast<float4*>(p.dz + (((size_t)t * p.n_pad + cn) * 2 + dir) * H4 + 4 * cu) = z4;
      }
      *reinterpret_cast<float4*>(dzl + (tid >> 4) * DZS + 4 * (tid & 15)) = z4;
    }
    __syncthreads();
    // ---- partial dh_{prev}[k] = sum_{j in J} U[k][j] dz[j] for ALL k; publish per tile
    if (s + 1 < p.T) {
      float bv[16];
      {
        const float* drow = dzl + nl * DZS + 16 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 d4 = *reinterpret_cast<const float4*>(drow + 4 * q);
          bv[4 * q] = d4.x; bv[4 * q + 1] = d4.y; bv[4 * q + 2] = d4.z; bv[4 * q + 3] = d4.w;
        }
      }
      const unsigned wtag = (unsigned)(s >> 1) & 1u;
      __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)(s & 1) * slot_words, 0, (unsigned)(slot_words * 4), 0x00020000);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int mt = w + kWaves * i;
        if (mt < P) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 16; kk += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i][kk], bv[kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[i][kk + 1], bv[kk + 1], acc1, 0, 0, 0);
          }
          const f32x4 a = acc0 + acc1;
          // lane (g, nl) holds units 4g..4g+3 of consumer tile mt for sample nl
          u32x4 o;
          o[0] = tag_word(a[0], wtag); o[1] = tag_word(a[1], wtag);
          o[2] = tag_word(a[2], wtag); o[3] = tag_word(a[3], wtag);
          const unsigned off = (unsigned)((((size_t)mt * P + cw) * 256 + nl * 16 + 4 * g) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(o, wr, off, 0, kSc1);
        }
      }
    }
  }
  if (cvalid && p.dc_state) p.dc_state[((size_t)dir * p.n_pad + cn) * H + cu] = dc;
}

// ---------------------------------------------------------------------------
struct Plan {
  int KS, UGW, R, P, TPW;
  size_t shm;
  size_t xchain_words;
  int chains_per_launch;
};

int even_up4(int x) { return (x + 3) & ~3; }

typedef void (*kern_t)(LstmParams);

kern_t pick_bwd(int tpw) {
  switch (tpw) {
    case 1: return lstm_bwd_kernel<1>;
    case 2: return lstm_bwd_kernel<2>;
    case 3: return lstm_bwd_kernel<3>;
    default: return lstm_bwd_kernel<4>;
  }
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

int make_plan(const asr_lstm_args* a, bool bwd, Plan* out) {
  const int H = a->H;
  const int chains = 2 * (a->n_pad / 16);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ASR_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ASR_ERR_LAUNCH;
  const int num_cu = prop.multiProcessorCount;
  Plan pl;
  kern_t k;
  if (!bwd) {
    // K split so that R (MFMAs per wave) <= 32, preferring R near 16
    int KS = env_int("ASR_LSTM_FWD_KS", 0);
    if (KS != 1 && KS != 2 && KS != 4 && KS != 8 && KS != 16) {
      KS = 4;
      while (KS < 16 && even_up4((H + 4 * KS - 1) / (4 * KS)) > kMaxR) KS *= 2;
      while (KS > 1 && even_up4((H + 2 * KS - 1) / (2 * KS)) <= 16) KS /= 2;
    }
    pl.KS = KS;
    pl.UGW = kWaves / KS;
    pl.R = even_up4((H + 4 * KS - 1) / (4 * KS));
    if (pl.R > kMaxR) {
      asr_set_error("lstm fwd: H=%d too large for the register-resident U slice", H);
      return ASR_ERR_INVALID;
    }
    const int UG = H / 4;
    pl.P = (UG + pl.UGW - 1) / pl.UGW;
    pl.TPW = 0;
    pl.shm = (size_t)16 * (H + 4) * 4 + (size_t)pl.UGW * (KS > 1 ? KS - 1 : 0) * 64 * 16;
    pl.xchain_words = (size_t)2 * 16 * H;
    k = lstm_fwd_kernel;
  } else {
    pl.KS = 1; pl.UGW = 0; pl.R = 16;
    pl.P = (H + 15) / 16;
    pl.TPW = (pl.P + kWaves - 1) / kWaves;
    if (pl.TPW > 4) {
      asr_set_error("lstm bwd: H=%d too large (max 1024)", H);
      return ASR_ERR_INVALID;
    }
    pl.shm = (size_t)(16 * 68 + pl.P * 256) * 4;
    pl.xchain_words = (size_t)2 * pl.P * pl.P * 256;
    k = pick_bwd(pl.TPW);
  }
  if (pl.shm > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)pl.shm) != hipSuccess) {
      asr_set_error("lstm: cannot reserve %zu bytes of LDS", pl.shm);
      return ASR_ERR_LAUNCH;
    }
  }
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, kThreads, pl.shm) !=
      hipSuccess)
    return ASR_ERR_LAUNCH;
  if (occ < 1) {
    asr_set_error("lstm: kernel does not fit a CU (LDS %zu B)", pl.shm);
    return ASR_ERR_RESIDENCY;
  }
  // 1024-thread workgroups: count ONE per CU (the occupancy API may over-report)
  const long cap = (long)num_cu;
  if (cap < pl.P) {
    asr_set_error("lstm: a chain needs %d co-resident workgroups, device has %d CUs", pl.P,
                  num_cu);
    return ASR_ERR_RESIDENCY;
  }
  long cpl = cap / pl.P;
  if (cpl > chains) cpl = chains;
  pl.chains_per_launch = (int)cpl;
  *out = pl;
  return ASR_OK;
}

constexpr size_t kStatusBytes = 256;

size_t xbuf_bytes(const asr_lstm_args* a, bool bwd) {
  const size_t chains = (size_t)2 * (a->n_pad / 16);
  const size_t P = (a->H + 15) / 16;
  const size_t words = bwd ? (size_t)2 * P * P * 256 : (size_t)2 * 16 * a->H;
  return asr_align_up(chains * words * 4, 256);
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
  const size_t xb = xbuf_bytes(a, bwd);
  LstmParams p;
  p.T = a->T; p.n_pad = a->n_pad; p.H = a->H; p.NB = a->n_pad / 16;
  p.KS = pl.KS; p.UGW = pl.UGW; p.R = pl.R; p.P = pl.P;
  p.U = a->U; p.mask_u = a->mask_u; p.zx = a->zx; p.y = a->y; p.cell = a->cell;
  p.gates = a->gates; p.dy = a->dy; p.dz = a->dz;
  p.status = reinterpret_cast<int*>(ws);
  p.xbuf = reinterpret_cast<unsigned*>(ws + kStatusBytes);
  p.xchain_words = (long long)pl.xchain_words;
  p.dc_state = reinterpret_cast<float*>(ws + kStatusBytes + xb);
  ASR_CHECK_HIP(hipMemsetAsync(ws, 0, kStatusBytes, stream));
  ASR_CHECK_HIP(hipMemsetAsync(ws + kStatusBytes, 0xFF, xb, stream));
  const int chains = 2 * p.NB;
  kern_t k = bwd ? pick_bwd(pl.TPW) : (kern_t)lstm_fwd_kernel;
  const bool stepwise = a->mode == 1;
  p.poll = stepwise ? 0 : 1;
  const int steps_per_launch = stepwise ? 1 : a->T;
  for (int s0 = 0; s0 < a->T; s0 += steps_per_launch) {
    for (int cb = 0; cb < chains; cb += pl.chains_per_launch) {
      const int nch = (chains - cb) < pl.chains_per_launch ? (chains - cb) : pl.chains_per_launch;
      p.chain_begin = cb;
      p.s_begin = s0;
      p.s_count = (a->T - s0) < steps_per_launch ? (a->T - s0) : steps_per_launch;
      hipLaunchKernelGGL(k, dim3(nch * pl.P), dim3(kThreads), pl.shm, stream, p);
      ASR_CHECK_LAUNCH();
    }
  }
  return ASR_OK;
}

}  // namespace

extern "C" size_t asr_lstm_workspace_bytes(const asr_lstm_args* a, int backward) {
  if (!a || a->n_pad <= 0 || a->H <= 0) return 0;
  return kStatusBytes + xbuf_bytes(a, backward != 0) +
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
  if (blocks) *blocks = pl.P * pl.chains_per_launch;
  if (chains_per_launch) *chains_per_launch = pl.chains_per_launch;
  return ASR_OK;
}
