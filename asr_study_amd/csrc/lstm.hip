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
// Design (CDNA4; measured history in DESIGN.md, from 11.4 / 19.5 us per step in the first
// version to 1.5 / 1.6 now at H = 256, 1.8 / 2.1 at H = 512):
//  * A layer is a set of independent CHAINS (direction, 16-row batch tile).  A
//    chain is split over 256-thread workgroups by hidden units (16 per WG); the
//    four waves of a WG sit one per SIMD (one MFMA pipe each) and keep their slice
//    of U stationary in VGPRs as the MFMA A-operand for the whole sequence.  C/D
//    layout row = 4*(lane>>4)+reg, col = lane&15: with gate columns ordered
//    unit*4+gate a lane ends up with the four gates of ONE (unit, sample), so the
//    gate math is lane-local.
//  * Arithmetic (default, ASR_LSTM_PREC=1): every fp32 operand is split into fp16
//    hi + lo (22 mantissa bits) and a product is three v_mfma_f32_16x16x32_f16 with
//    fp32 accumulation (error ~2^-22) instead of eight exact fp32 MFMAs; BPTT scales
//    each batch column by its own power of two first.  ASR_LSTM_PREC=0 selects the
//    same kernel structure on exact v_mfma_f32_16x16x4_f32 (the EXACT instantiations of
//    fwd_body_x / bwd_body_c: plain cell, H = 256 / 512, persistent mode).
//  * Forward: workgroups exchange h_t, split and packed by the PRODUCER (one word =
//    fp16 hi << 16 | fp16 lo).  At H = 256 / 512 (fwd_body_x) K is split over the four
//    waves: a wave multiplies all 64 gate columns of the WG with the quarter of h it
//    gathered itself (registers -> MFMA B operand, no LDS staging), and the partial
//    gate tiles meet in LDS; narrower layers (fwd_body_h) stage h in LDS once and
//    every wave reads its B operand from there.  One barrier per step either way.
//  * Backward: a workgroup owns 16 units = 64 gate columns j.  It multiplies its
//    own dz_J (local) with U[:, J] for ALL H outputs and publishes the partial dh
//    tiles; a consumer lane gathers the 16-byte group of its (sample, unit quad) from
//    every producer, adds them in registers and across four lanes with DPP quad
//    permutes.  Exchange volume is H x 16 words per producer, not the 4H-wide dz
//    (bwd_body_h, bwd_body_x).  At H = 512 that exchange is 1 MB per chain-step and was
//    written through to HBM: bwd_body_c splits a chain in two dimensions instead (4 OT
//    unit blocks x 4 sample quarters, the whole batch per chain), a workgroup reduces
//    over its 64 units in registers and exchanges 64 x 16 dh words with the 4 OT - 1
//    others of its sample quarter.
//  * Kernel choice is asr_lstm_plan's (make_plan): _x / _c for the plain cell at
//    H = 256 / 512 in persistent mode, _h / _hv otherwise (any H <= 512, variants,
//    stepwise); ASR_LSTM_GENERIC=1 forces _h (the tests compare the two).
//  * A step's first poll is preceded by a short nap (ASR_LSTM_PREPOLL_F/_B): a poll
//    that reaches the L2 before the producers' stores costs a whole extra round trip.
//  * The optional cell variants (multiplicative integration, zoneout) are the VAR
//    template paths, compiled into their own kernels (lstm_*_kernel_hv); layer
//    normalisation lives in lstm_ln.hip.
//  * Hand-off protocol: every exchanged fp32 word carries the step tag in its
//    mantissa LSB (value perturbed by <= 1 ulp; the consumer clears the bit); a
//    16-byte group is its own flag -- no fences, no flags, placement independent
//    (MI355X guide, Guideline 16 / R2 "the data IS the flag").  Two slots (step
//    parity) suffice: a producer is at most one step ahead of its slowest
//    consumer, so a slot holds step s or s-2, which differ in bit (s>>1)&1 (absolute
//    step number, so a sequence may be continued by a later launch).  The buffer is
//    memset to 0xFF (tag 1) before step 0.  Transport: agent-scope
//    (sc1, write-through) stores + sc1 loads.  If -- and only if -- every
//    workgroup of a chain reports the same XCC id at kernel start, the chain
//    switches to plain stores + L1-bypassing (nt) loads served by that XCD's L2;
//    the choice changes speed only, never results.
//  * Every spin is bounded by the wall clock; a give-up is recorded in the
//    workspace status word and the kernel finishes without polling.
//  * mode 1 launches one step per kernel (same buffer, visibility from the kernel
//    boundary, no polling): the always-safe fallback with identical arithmetic.
#include "common.h"
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kThreads = 256;              // 4 waves, one per SIMD
constexpr int kSc1 = 16;                   // buffer-op cache policy: SC1 (agent scope)
constexpr int kNt = 2;                     // buffer-op cache policy: NT (bypass L1)

struct LstmParams {
  int T, n_pad, H, NB;
  int R;                   // fwd: MFMA steps per wave (= ceil4(H/4))
  int P;                   // workgroups per chain
  int nch;                 // chains in this launch
  int s_begin, s_count;
  int chain_begin;
  int poll;                // 1: persistent (poll tags); 0: one step per launch
  int allow_fast;          // may use the same-XCD transport
  int dbg;                 // ablation switches (ASR_LSTM_DBG), 0 in production
  int prepoll;             // 64-clock naps before a step's first poll (see gather_groups)
  int repoll;              // 64-clock naps between poll rounds
  int xstride;             // fwd: bytes between consecutive unit-group tiles in a slot
  long long spin;          // bound of every spin, ticks of the 100 MHz wall clock
  const float* U;
  const float* mask_u;
  const float* zx;
  float* y;
  float* cell;
  float* gates;
  const float* dy;
  float* dz;
  float* dc_state;
  unsigned* dz_absmax;     // optional: max |dz| as float bits (atomicMax)
  // optional cell variants (core/layers.py:432-469); all NULL on the default path
  const float* mi;         // (2, 4, 4H): alpha, beta1, beta2, bias per direction
  float* uh;               // (T, n_pad, 2, 4H) h_prev @ U (fwd writes, BPTT reads)
  const float* zone_c;     // (T, 2, H) zoneout coefficient of the cell state, per frame
  const float* zone_h;     // (T, 2, H) ... of the hidden state
  const float* wx;         // BPTT + mi: x @ W of the forward pass (no bias)
  float* dwx;              // BPTT + mi: d / d (x @ W); dz then holds d / d (h_prev @ U)
  float* dmi;              // BPTT + mi: (NB, 2, 4, 4H) per-batch-tile sums of the parameter
                           //            gradients d alpha, d beta1, d beta2, d bias
  float* db_part;          // BPTT, optional: (NB, 2, 4H) per-batch-tile sums of dz over the
                           //            tile's samples and all steps (bias-gradient partials)
  unsigned* xbuf;          // exchange buffer (words)
  long long xchain_words;  // words per chain (2 slots)
  int* xcc;                // [chains][P] XCC id + 1 of every workgroup
  int* status;             // [0] timeout flag, [1] chains on the fast transport
};

constexpr long long kSpinTicks = 60LL * 1000 * 1000;   // default bound: 0.6 s (100 MHz wall clock)

// Workspace layout: [sticky block][status block][XCC table][exchange buffer][dc_state].
// status[0] is the timeout flag of the LAST call (the library clears it at the start of
// every sequence); the first int of the sticky block in front of it is set together with
// it and cleared only by asr_lstm_status, so a host that checks once per training step
// still sees a timeout of any of the step's calls.
constexpr int kStickyInts = 64;                        // 256 bytes
__device__ __forceinline__ void mark_timeout(int* status) {
  atomicExch(status, 1);
  atomicExch(status - kStickyInts, 1);
}

// Debug (ASR_LSTM_DBG & 32): shader-clock ticks per phase of a step, accumulated over the
// steps of a launch by the four waves of workgroup 0 of the launch's first chain; six phases
// per wave at status + 16 ints (asr_lstm_profile).  `on` is wave-uniform.
struct StepProf {
  bool on;
  long long pt[6], last;
  __device__ __forceinline__ void init(bool enable) {
    on = enable;
#pragma unroll
    for (int i = 0; i < 6; ++i) pt[i] = 0;
    last = on ? (long long)__builtin_readcyclecounter() : 0;
  }
  __device__ __forceinline__ void stamp(int i) {
    if (on) {
      const long long now = (long long)__builtin_readcyclecounter();
      pt[i] += now - last;
      last = now;
    }
  }
  __device__ __forceinline__ void flush(int* status, int w) const {
    if (on && (threadIdx.x & 63) == 0) {
      long long* out = reinterpret_cast<long long*>(status + 16) + 6 * w;
#pragma unroll
      for (int i = 0; i < 6; ++i) out[i] = pt[i];
    }
  }
};

__device__ __forceinline__ float hard_sigmoid(float x) {
  return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f);
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = (e^{2x}-1)/(e^{2x}+1); |abs err| ~ 1e-7, saturates cleanly.
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return __fdividef(e - 1.f, e + 1.f);
}
// the same with v_rcp_f32 instead of the IEEE division sequence (1 ulp of the quotient)
__device__ __forceinline__ float fast_tanh_rcp(float x) {
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return (e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ unsigned tag_word(float v, unsigned tag) {
  return (__float_as_uint(v) & ~1u) | tag;
}
__device__ __forceinline__ bool tags_ok(const u32x4& v, unsigned tag) {
  return ((v[0] & 1u) == tag) & ((v[1] & 1u) == tag) & ((v[2] & 1u) == tag) &
         ((v[3] & 1u) == tag);
}
template <bool FAST>
__device__ __forceinline__ u32x4 xload(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, FAST ? kNt : kSc1);
}
template <bool FAST>
__device__ __forceinline__ void xstore(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, FAST ? 0 : kSc1);
}

// max over the 16 lanes of a DPP row (quad swaps, then half-row and row mirrors): four
// VALU ops with a DPP modifier instead of four LDS-crossbar shuffles
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(
                   __float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));   // [1,0,3,2]
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(
                   __float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));   // [2,3,0,1]
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(
                   __float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));  // half mirror
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(
                   __float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));  // row mirror
  return v;
}

// quad-lane exchanges (DPP quad_perm [1,0,3,2] and [2,3,0,1])
__device__ __forceinline__ float quad_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1,
                                                    0xF, 0xF, false));
}
__device__ __forceinline__ float quad_swap2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E,
                                                    0xF, 0xF, false));
}

// ---- split-fp16 arithmetic for the recurrent products ------------------------
// x = hi + lo/2048 with hi = fp16(x), lo = fp16((x - hi) * 2048): 22 mantissa bits.
// x*y ~= hi_x*hi_y + (hi_x*lo_y + lo_x*hi_y)/2048 (the lo*lo term is 2^-22 relative),
// three v_mfma_f32_16x16x32_f16 (fp32 accumulate) instead of eight fp32 MFMAs.
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr float kLoScale = 2048.f;

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * kLoScale);
}

// Loads NL 16-byte groups (byte offsets off[i]) and re-polls the stale ones until
// every word carries `tag`.
template <bool FAST, int NL>
__device__ __forceinline__ void gather_groups(__amdgpu_buffer_rsrc_t rsrc,
                                              const unsigned (&off)[NL], const bool (&use)[NL],
                                              unsigned tag, int poll, bool& dead, int* status,
                                              u32x4 (&v)[NL], int nosleep = 0, int prepoll = 0,
                                              int repoll = 1, long long spin = kSpinTicks) {
  // A poll that reaches the L2 before the producers' stores costs a whole extra round
  // trip, and a step waits for the SLOWEST of its waves: napping a little before the
  // first poll trades a small fixed delay for far fewer second rounds.
  if (poll) for (int i = 0; i < prepoll; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
  for (int i = 0; i < NL; ++i)
    if (use[i]) v[i] = xload<FAST>(rsrc, off[i]);
  if (!poll || dead) return;
  long long t0 = 0;
  bool timing = false;
  for (;;) {
    bool all_ok = true;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (use[i] && !tags_ok(v[i], tag)) all_ok = false;
    if (all_ok) return;
    if (!timing) { t0 = wall_clock64(); timing = true; }
    else if (wall_clock64() - t0 > spin) {
      dead = true;
      mark_timeout(status);
      return;
    }
    if (!nosleep) for (int i = 0; i < repoll; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (use[i] && !tags_ok(v[i], tag)) v[i] = xload<FAST>(rsrc, off[i]);
  }
}

// Decides the transport of this workgroup's chain: true iff all P workgroups of the
// chain run on the same XCD (they all read the same table, so they all agree).
__device__ bool chain_on_one_xcd(const LstmParams& p, int chain, int wg, int* lds_i) {
  if (!p.poll || !p.allow_fast) return false;
  int* tab = p.xcc + (size_t)chain * p.P;
  const int tid = threadIdx.x;
  if (tid == 0) {
    // HW_REG_XCC_ID = 20, bits [3:0]
    const int id = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf);
    __hip_atomic_store(tab + wg, id + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool ok = true;
  for (int i = tid; i < p.P; i += kThreads) {
    int v = 0;
    const long long t0 = wall_clock64();
    for (;;) {
      v = __hip_atomic_load(tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0) break;
      if (wall_clock64() - t0 > p.spin) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    lds_i[i] = ok ? v : -1;
  }
  __syncthreads();
  bool same = lds_i[0] > 0;
  for (int i = 1; i < p.P; ++i) same = same && (lds_i[i] == lds_i[0]);
  __syncthreads();
  if (same && tid == 0 && wg == 0) atomicAdd(p.status + 1, 1);
  return same;
}

// blockIdx -> (chain slot, workgroup).  Workgroups of one chain use block ids that
// are congruent mod 8, which the dispatcher is observed to place on one XCD.
__device__ __forceinline__ bool map_block(const LstmParams& p, int& chain_local, int& wg) {
  const int xslot = blockIdx.x & 7;
  const int i = blockIdx.x >> 3;
  wg = i % p.P;
  chain_local = (i / p.P) * 8 + xslot;
  return chain_local < p.nch;
}

// ---------------------------------------------------------------------------
// forward, split-fp16 MFMA variant.  NKK = number of K=32 MFMA steps (H <= 32*NKK).
template <int NKK, bool FAST, bool VAR>
__device__ __forceinline__ void fwd_body_h(const LstmParams& p, int chain, int wg, float* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int UG = H >> 2;
  const int dir = chain / p.NB, bt = chain % p.NB;
  const int ug = wg * 4 + w;
  const bool ug_ok = ug < UG;
  const int n = bt * 16 + nl;
  const int u = 4 * ug + g;
  constexpr int KP = 32 * NKK;                    // padded K
  constexpr int HS = KP + 8;                      // LDS row stride (halfs)
  _Float16* hb = reinterpret_cast<_Float16*>(lds);   // [2 slots][hi|lo][16][HS]
  constexpr int tile_halfs = 16 * HS;

  // stationary A fragments: column i = lane&15 of the gate tile, k = 32kk + 8g + e
  h8 ufh[NKK], ufl[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * kk + 8 * g + e;
      const float x = (ug_ok && k < H) ? p.U[((size_t)(dir * H + k)) * H4 + 16 * ug + nl] : 0.f;
      _Float16 hi, lo;
      split_f16(x, hi, lo);
      ufh[kk][e] = hi; ufl[kk][e] = lo;
    }
  }
  float mask = 1.f;
  if (ug_ok && p.mask_u) mask = p.mask_u[((size_t)dir * p.n_pad + n) * H + u];
  float c = 0.f;
  float hprev = 0.f;                                // VAR: this lane's own previous h
  float4 mi_a = make_float4(0.f, 0.f, 0.f, 0.f), mi_b1 = mi_a, mi_b2 = mi_a, mi_b = mi_a;
  const bool has_mi = VAR && p.mi != nullptr;
  if (has_mi && ug_ok) {
    const float* m = p.mi + (size_t)dir * 4 * H4 + 4 * u;
    mi_a = *reinterpret_cast<const float4*>(m);
    mi_b1 = *reinterpret_cast<const float4*>(m + H4);
    mi_b2 = *reinterpret_cast<const float4*>(m + 2 * H4);
    mi_b = *reinterpret_cast<const float4*>(m + 3 * H4);
  }
  bool dead = false;
  unsigned* xch = p.xbuf + (size_t)chain * p.xchain_words;    // [2][UG][16][4]
  const int slot_words = UG * (p.xstride / 4);
  const int s_end = p.s_begin + p.s_count;
  if (ug_ok && p.s_begin > 0) {
    const int tpp = dir == 0 ? p.s_begin - 1 : p.T - p.s_begin;
    c = p.cell[(((size_t)tpp * p.n_pad + n) * 2 + dir) * H + u];
    if (VAR) hprev = p.y[((size_t)tpp * p.n_pad + n) * H2 + dir * H + u];
  }
  for (int e = tid; e < 4 * tile_halfs; e += kThreads) hb[e] = (_Float16)0.f;
  __syncthreads();
  auto load_zx = [&](int ss) -> float4 {
    if (!ug_ok || ss >= s_end) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int tt = dir == 0 ? ss : p.T - 1 - ss;
    return *reinterpret_cast<const float4*>(
        p.zx + (((size_t)tt * p.n_pad + n) * 2 + dir) * H4 + 4 * u);
  };
  float4 zx_next = load_zx(p.s_begin);
  auto load_zone = [&](const float* z, int ss) -> float {
    if (!VAR || z == nullptr || !ug_ok || ss >= s_end) return 1.f;
    const int tt = dir == 0 ? ss : p.T - 1 - ss;
    return z[((size_t)tt * 2 + dir) * H + u];
  };
  float kc_next = load_zone(p.zone_c, p.s_begin), kh_next = load_zone(p.zone_h, p.s_begin);
  constexpr int NL = (KP * 4 + kThreads - 1) / kThreads;       // 16-B groups per thread
  const bool prof = (p.dbg & 32) && wg == 0 && chain == p.chain_begin && lane == 0;
  long long pt[4] = {0, 0, 0, 0}, tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
  for (int s = p.s_begin; s < s_end; ++s) {
    if (prof) tk0 = wall_clock64();
    const int t = dir == 0 ? s : p.T - 1 - s;
    const float4 zx4 = zx_next;
    const float kc = kc_next, kh = kh_next;
    f32x4 am0 = {0.f, 0.f, 0.f, 0.f}, am1 = am0, ac0 = am0, ac1 = am0;
    if (s > 0) {
      _Float16* th = hb + (size_t)(s & 1) * 2 * tile_halfs;      // hi tile, lo tile follows
      _Float16* tl = th + tile_halfs;
      const unsigned tag = (unsigned)((s - 1) >> 1) & 1u;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)((s - 1) & 1) * slot_words, 0, slot_words * 4, 0x00020000);
      unsigned off[NL];
      bool use[NL];
      u32x4 v[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int grp = tid + i * kThreads;
        use[i] = grp < UG * 16;
        off[i] = (unsigned)((grp >> 4) * p.xstride + (grp & 15) * 16);
      }
      gather_groups<FAST, NL>(rsrc, off, use, tag, p.poll, dead, p.status, v, p.dbg & 64,
                              p.prepoll, p.repoll, p.spin);
      if (prof) tk1 = wall_clock64();
      zx_next = load_zx(s + 1);
      kc_next = load_zone(p.zone_c, s + 1); kh_next = load_zone(p.zone_h, s + 1);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if (use[i]) {
          const int grp = tid + i * kThreads;
          const int gu = grp >> 4, gn = grp & 15;
          // exchanged word = fp16 hi << 16 | fp16 lo (split once, by the producer);
          // the tag sits in lo's LSB and is cleared so that zeros stay exact zeros
          const unsigned a0 = v[i][0] & ~1u, a1 = v[i][1] & ~1u;
          const unsigned a2 = v[i][2] & ~1u, a3 = v[i][3] & ~1u;
          uint2 hi2, lo2;
          hi2.x = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
          hi2.y = __builtin_amdgcn_perm(a3, a2, 0x07060302u);
          lo2.x = __builtin_amdgcn_perm(a1, a0, 0x05040100u);
          lo2.y = __builtin_amdgcn_perm(a3, a2, 0x05040100u);
          *reinterpret_cast<uint2*>(th + gn * HS + 4 * gu) = hi2;
          *reinterpret_cast<uint2*>(tl + gn * HS + 4 * gu) = lo2;
        }
      }
      __syncthreads();
      if (prof) tk2 = wall_clock64();
      if (ug_ok) {
        const _Float16* rh = th + nl * HS + 8 * g;
        const _Float16* rl = tl + nl * HS + 8 * g;
        h8 bh[NKK], bl[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          bh[kk] = *reinterpret_cast<const h8*>(rh + 32 * kk);
          bl[kk] = *reinterpret_cast<const h8*>(rl + 32 * kk);
        }
#pragma unroll
        for (int kk = 0; kk < NKK; kk += 2) {
          am0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[kk], bh[kk], am0, 0, 0, 0);
          ac0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[kk], bl[kk], ac0, 0, 0, 0);
          ac1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufl[kk], bh[kk], ac1, 0, 0, 0);
          if (kk + 1 < NKK) {
            am1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[kk + 1], bh[kk + 1], am1, 0, 0, 0);
            ac0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[kk + 1], bl[kk + 1], ac0, 0, 0, 0);
            ac1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufl[kk + 1], bh[kk + 1], ac1, 0, 0, 0);
          }
        }
      }
    } else {
      zx_next = load_zx(s + 1);
      kc_next = load_zone(p.zone_c, s + 1); kh_next = load_zone(p.zone_h, s + 1);
    }
    const f32x4 a = (am0 + am1) + (ac0 + ac1) * (1.f / kLoScale);
    if (prof) { asm volatile("" :: "v"(a[0])); tk3 = wall_clock64(); }
    if (ug_ok) {
      float z0, z1, z2, z3;
      if (has_mi) {       // z = alpha * Wx * Uh + beta1 * Uh + beta2 * Wx + b (layers.py:441-443)
        z0 = mi_a.x * zx4.x * a[0] + mi_b1.x * a[0] + mi_b2.x * zx4.x + mi_b.x;
        z1 = mi_a.y * zx4.y * a[1] + mi_b1.y * a[1] + mi_b2.y * zx4.y + mi_b.y;
        z2 = mi_a.z * zx4.z * a[2] + mi_b1.z * a[2] + mi_b2.z * zx4.z + mi_b.z;
        z3 = mi_a.w * zx4.w * a[3] + mi_b1.w * a[3] + mi_b2.w * zx4.w + mi_b.w;
      } else {
        z0 = a[0] + zx4.x; z1 = a[1] + zx4.y; z2 = a[2] + zx4.z; z3 = a[3] + zx4.w;
      }
      const float gi = hard_sigmoid(z0);
      const float gf = hard_sigmoid(z1);
      const float gg = fast_tanh(z2);
      const float go = hard_sigmoid(z3);
      float cn = gf * c + gi * gg;
      if (VAR) cn = c + kc * (cn - c);              // zoneout of the cell state (:457-459)
      c = cn;
      float h = go * fast_tanh(c);
      if (VAR) { h = hprev + kh * (h - hprev); hprev = h; }   // ... of the hidden state
      if (s + 1 < p.T) {
        const unsigned wtag = (unsigned)(s >> 1) & 1u;
        _Float16 ph, pl;
        split_f16(h * mask, ph, pl);
        const unsigned w0 = ((((unsigned)__builtin_bit_cast(unsigned short, ph) << 16) |
                              (unsigned)__builtin_bit_cast(unsigned short, pl)) & ~1u) | wtag;
        u32x4 o;
        o[0] = w0;
        o[1] = (unsigned)__shfl_down((int)w0, 16, 64);
        o[2] = (unsigned)__shfl_down((int)w0, 32, 64);
        o[3] = (unsigned)__shfl_down((int)w0, 48, 64);
        if (lane < 16) {
          __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
              xch + (size_t)(s & 1) * slot_words, 0, slot_words * 4, 0x00020000);
          xstore<FAST>(o, wr, (unsigned)(ug * p.xstride + nl * 16));
        }
      }
      const size_t row = (size_t)t * p.n_pad + n;
      p.y[row * H2 + dir * H + u] = h;
      p.cell[(row * 2 + dir) * H + u] = c;
      *reinterpret_cast<float4*>(p.gates + (row * 2 + dir) * H4 + 4 * u) =
          make_float4(gi, gf, gg, go);
      if (VAR && p.uh)
        *reinterpret_cast<float4*>(p.uh + (row * 2 + dir) * H4 + 4 * u) =
            make_float4(a[0], a[1], a[2], a[3]);
    }
    if (prof && s > 0) {
      const long long tk4 = wall_clock64();
      pt[0] += tk1 - tk0; pt[1] += tk2 - tk1; pt[2] += tk3 - tk2; pt[3] += tk4 - tk3;
    }
  }
  if (prof) {
    long long* out = reinterpret_cast<long long*>(p.status + 16) + 6 * w;
    for (int i = 0; i < 4; ++i) out[i] = pt[i];
  }
}

template <int NKK>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel_h(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int chain_local, wg;
  if (!map_block(p, chain_local, wg)) return;
  const int chain = p.chain_begin + chain_local;
  const bool fast = chain_on_one_xcd(p, chain, wg, reinterpret_cast<int*>(lds));
  if (fast) fwd_body_h<NKK, true, false>(p, chain, wg, lds);
  else fwd_body_h<NKK, false, false>(p, chain, wg, lds);
}

// the cell variants (mi / zoneout) live in their own kernels so that their extra
// registers never touch the allocation of the default ones
template <int NKK>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel_hv(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int chain_local, wg;
  if (!map_block(p, chain_local, wg)) return;
  const int chain = p.chain_begin + chain_local;
  const bool fast = chain_on_one_xcd(p, chain, wg, reinterpret_cast<int*>(lds));
  if (fast) fwd_body_h<NKK, true, true>(p, chain, wg, lds);
  else fwd_body_h<NKK, false, true>(p, chain, wg, lds);
}

template <int NL>
__device__ __forceinline__ bool all_tagged(const u32x4 (&v)[NL], unsigned flip) {
  unsigned x = 0u;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    x |= ((v[i][0] ^ flip) | (v[i][1] ^ flip)) | ((v[i][2] ^ flip) | (v[i][3] ^ flip));
  return (x & 1u) == 0u;
}

// ---- arithmetic shared by the K-split forward kernels (fwd_body_k / fwd_body_k2), written
// with contraction off and explicit FMAs so that both round identically: which of the two
// processed a batch row is then invisible in the result, bit for bit.
__device__ __forceinline__ f32x4 combine_split(const f32x4& am, const f32x4& ac0,
                                               const f32x4& ac1) {
#pragma clang fp contract(off)
  const f32x4 t = ac0 + ac1;
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(t[e], 1.f / kLoScale, am[e]);
  return r;
}
__device__ __forceinline__ float hard_sigmoid_nc(float x) {
#pragma clang fp contract(off)
  return fminf(fmaxf(__builtin_fmaf(0.2f, x, 0.5f), 0.f), 1.f);
}
__device__ __forceinline__ float tanh_nc(float x) {
#pragma clang fp contract(off)
  const float xc = fminf(fmaxf(x, -15.f), 15.f);
  const float e = __expf(2.f * xc);
  return __fdividef(e - 1.f, e + 1.f);
}
struct CellFwd { float gi, gf, gg, go, c, h, hm; };
// a: recurrent contribution h_prev @ U of the four gates; zx4: x @ W + b; hm = h * mask is
// what the next step multiplies with U
__device__ __forceinline__ CellFwd cell_forward(const f32x4& a, const float4& zx4, float c_prev,
                                                float mask) {
#pragma clang fp contract(off)
  CellFwd o;
  o.gi = hard_sigmoid_nc(a[0] + zx4.x);
  o.gf = hard_sigmoid_nc(a[1] + zx4.y);
  o.gg = tanh_nc(a[2] + zx4.z);
  o.go = hard_sigmoid_nc(a[3] + zx4.w);
  o.c = __builtin_fmaf(o.gf, c_prev, o.gi * o.gg);
  o.h = o.go * tanh_nc(o.c);
  o.hm = o.h * mask;
  return o;
}
// exchanged word: fp16 hi << 16 | fp16 lo with the step tag in the LSB
__device__ __forceinline__ unsigned packed_word(float hm, unsigned tag) {
  _Float16 ph, pl;
  split_f16(hm, ph, pl);
  return ((((unsigned)__builtin_bit_cast(unsigned short, ph) << 16) |
           (unsigned)__builtin_bit_cast(unsigned short, pl)) & ~1u) | tag;
}

// ---------------------------------------------------------------------------
// forward, split-fp16, K split over the waves, third generation (plain cell, H = 256 / 512,
// persistent mode): the default forward kernel.  NT = 2: two batch tiles per workgroup as
// fwd_body_k2; NT = 1: one tile, gather issued right after the publish.  Same changes as in
// bwd_body_x: MFMAs as inline asm with the stationary U fragments in AGPRs and the results in
// VGPRs (no v_accvgpr traffic), one OR-reduction + compare per lane as tag test, gathered
// words used with their tag bit (the LSB of the fp16 `lo` half: 2^-22 relative), no branch
// around a vector-memory instruction, gather offsets as immediates of one base register.
// Arithmetic of a (sample, unit) is the same for NT = 1 and 2, sliced or whole, either
// transport.
template <int NKW> struct FwdMfma;
template <> struct FwdMfma<2> {
  // TWO unit groups at once: am_j = sum_kk Uh_j[kk] Bh[kk] ; ac_j = sum_kk (Uh_j[kk] Bl[kk] +
  // Ul_j[kk] Bh[kk]), each chain's terms in the order kk = 0, 1, .. -- every result bit as if a
  // group ran alone -- but the four chains interleaved so that an accumulator is reused three
  // MFMAs (48 cycles of pipe) later at the earliest: a lone group's 2 chains wait on the ~40
  // cycles of MFMA latency at every step (the K-slice phase measured 1450 cycles for 768 of
  // pipe).
  static __device__ __forceinline__ void run2(f32x4& am0, f32x4& ac0, f32x4& am1, f32x4& ac1,
                                              const f32x4 (&uh0)[2], const f32x4 (&ul0)[2],
                                              const f32x4 (&uh1)[2], const f32x4 (&ul1)[2],
                                              const h8 (&bh)[2], const h8 (&bl)[2]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %4, %14, 0\n\t"      // ac0 += uh0[0] bl[0]
        "v_mfma_f32_16x16x32_f16 %3, %8, %14, 0\n\t"      // ac1 += uh1[0] bl[0]
        "v_mfma_f32_16x16x32_f16 %0, %4, %12, 0\n\t"      // am0 += uh0[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %1, %6, %12, %1\n\t"      // ac0 += ul0[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %3, %10, %12, %3\n\t"      // ac1 += ul1[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %2, %8, %12, 0\n\t"      // am1 += uh1[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %1, %5, %15, %1\n\t"      // ac0 += uh0[1] bl[1]
        "v_mfma_f32_16x16x32_f16 %3, %9, %15, %3\n\t"      // ac1 += uh1[1] bl[1]
        "v_mfma_f32_16x16x32_f16 %0, %5, %13, %0\n\t"      // am0 += uh0[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %1, %7, %13, %1\n\t"      // ac0 += ul0[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %3, %11, %13, %3\n\t"      // ac1 += ul1[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %2, %9, %13, %2\n\t"      // am1 += uh1[1] bh[1]
        "s_nop 11"
        : "=&v"(am0), "=&v"(ac0), "=&v"(am1), "=&v"(ac1)
        : "a"(uh0[0]), "a"(uh0[1]), "a"(ul0[0]), "a"(ul0[1]), "a"(uh1[0]), "a"(uh1[1]), "a"(ul1[0]), "a"(ul1[1]), "v"(bh[0]), "v"(bh[1]), "v"(bl[0]), "v"(bl[1]));
  }
};
template <> struct FwdMfma<4> {
  // TWO unit groups at once: am_j = sum_kk Uh_j[kk] Bh[kk] ; ac_j = sum_kk (Uh_j[kk] Bl[kk] +
  // Ul_j[kk] Bh[kk]), each chain's terms in the order kk = 0, 1, .. -- every result bit as if a
  // group ran alone -- but the four chains interleaved so that an accumulator is reused three
  // MFMAs (48 cycles of pipe) later at the earliest: a lone group's 2 chains wait on the ~40
  // cycles of MFMA latency at every step (the K-slice phase measured 1450 cycles for 768 of
  // pipe).
  static __device__ __forceinline__ void run2(f32x4& am0, f32x4& ac0, f32x4& am1, f32x4& ac1,
                                              const f32x4 (&uh0)[4], const f32x4 (&ul0)[4],
                                              const f32x4 (&uh1)[4], const f32x4 (&ul1)[4],
                                              const h8 (&bh)[4], const h8 (&bl)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %4, %24, 0\n\t"      // ac0 += uh0[0] bl[0]
        "v_mfma_f32_16x16x32_f16 %3, %12, %24, 0\n\t"      // ac1 += uh1[0] bl[0]
        "v_mfma_f32_16x16x32_f16 %0, %4, %20, 0\n\t"      // am0 += uh0[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %1, %8, %20, %1\n\t"      // ac0 += ul0[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %3, %16, %20, %3\n\t"      // ac1 += ul1[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %2, %12, %20, 0\n\t"      // am1 += uh1[0] bh[0]
        "v_mfma_f32_16x16x32_f16 %1, %5, %25, %1\n\t"      // ac0 += uh0[1] bl[1]
        "v_mfma_f32_16x16x32_f16 %3, %13, %25, %3\n\t"      // ac1 += uh1[1] bl[1]
        "v_mfma_f32_16x16x32_f16 %0, %5, %21, %0\n\t"      // am0 += uh0[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %1, %9, %21, %1\n\t"      // ac0 += ul0[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %3, %17, %21, %3\n\t"      // ac1 += ul1[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %2, %13, %21, %2\n\t"      // am1 += uh1[1] bh[1]
        "v_mfma_f32_16x16x32_f16 %1, %6, %26, %1\n\t"      // ac0 += uh0[2] bl[2]
        "v_mfma_f32_16x16x32_f16 %3, %14, %26, %3\n\t"      // ac1 += uh1[2] bl[2]
        "v_mfma_f32_16x16x32_f16 %0, %6, %22, %0\n\t"      // am0 += uh0[2] bh[2]
        "v_mfma_f32_16x16x32_f16 %1, %10, %22, %1\n\t"      // ac0 += ul0[2] bh[2]
        "v_mfma_f32_16x16x32_f16 %3, %18, %22, %3\n\t"      // ac1 += ul1[2] bh[2]
        "v_mfma_f32_16x16x32_f16 %2, %14, %22, %2\n\t"      // am1 += uh1[2] bh[2]
        "v_mfma_f32_16x16x32_f16 %1, %7, %27, %1\n\t"      // ac0 += uh0[3] bl[3]
        "v_mfma_f32_16x16x32_f16 %3, %15, %27, %3\n\t"      // ac1 += uh1[3] bl[3]
        "v_mfma_f32_16x16x32_f16 %0, %7, %23, %0\n\t"      // am0 += uh0[3] bh[3]
        "v_mfma_f32_16x16x32_f16 %1, %11, %23, %1\n\t"      // ac0 += ul0[3] bh[3]
        "v_mfma_f32_16x16x32_f16 %3, %19, %23, %3\n\t"      // ac1 += ul1[3] bh[3]
        "v_mfma_f32_16x16x32_f16 %2, %15, %23, %2\n\t"      // am1 += uh1[3] bh[3]
        "s_nop 11"
        : "=&v"(am0), "=&v"(ac0), "=&v"(am1), "=&v"(ac1)
        : "a"(uh0[0]), "a"(uh0[1]), "a"(uh0[2]), "a"(uh0[3]), "a"(ul0[0]), "a"(ul0[1]), "a"(ul0[2]), "a"(ul0[3]), "a"(uh1[0]), "a"(uh1[1]), "a"(uh1[2]), "a"(uh1[3]), "a"(ul1[0]), "a"(ul1[1]), "a"(ul1[2]), "a"(ul1[3]), "v"(bh[0]), "v"(bh[1]), "v"(bh[2]), "v"(bh[3]), "v"(bl[0]), "v"(bl[1]), "v"(bl[2]), "v"(bl[3]));
  }
};

template <int NKW, bool FAST, bool EXACT>
__device__ __forceinline__ void fwd_body_x(const LstmParams& p, int unit, int wg, float* lds) {
  // Requires H == 128 * NKW (every lane's gather groups and units exist)
  constexpr int NT = 1;                            // batch tiles per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int UG = H >> 2;
  const int dir = unit / p.NB, bt0 = unit % p.NB;
  const int ug = wg * 4 + w;                       // the unit group this wave FINISHES
  const int u = 4 * ug + g;
  const int kbase = 32 * NKW * w;                  // first unit of this wave's K slice
  f32x4* part = reinterpret_cast<f32x4*>(lds);     // [2 bufs][4 waves][4 gate tiles][64 lanes]

  // EXACT: the products on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate, bitwise an fmaf
  // chain).  MFMA m = (kk, half, e) of a gate tile takes from lane (g, nl) the fp32 word e of
  // its gathered group (kk, half), i.e. k-index g <-> unit kbase + 32 kk + 8 g + 4 half + e: the
  // exchange layout and the gather are those of the split path, the words are plain tagged fp32.
  constexpr int NM = EXACT ? 8 * NKW : 1;          // fp32 MFMAs per gate tile
  float uf[4][NM];                                 // EXACT: one A-fragment register each (AGPRs)
  if constexpr (EXACT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ugj = wg * 4 + j;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int k = kbase + 32 * (m >> 3) + 8 * g + (m & 7);      // (m & 7) = 4 half + e
        uf[j][m] = p.U[((size_t)(dir * H + k)) * H4 + 16 * ugj + nl];
        asm volatile("" : "+a"(uf[j][m]));         // AGPR-class from here on
      }
    }
  }
  f32x4 ufh[4][NKW], ufl[4][NKW];                  // bit patterns of 8 halfs each (AGPRs)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (EXACT) break;
    const int ugj = wg * 4 + j;
#pragma unroll
    for (int kk = 0; kk < NKW; ++kk) {
      h8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kbase + 32 * kk + 8 * g + e;
        _Float16 hi, lo;
        split_f16(p.U[((size_t)(dir * H + k)) * H4 + 16 * ugj + nl], hi, lo);
        hv[e] = hi; lv[e] = lo;
      }
      ufh[j][kk] = __builtin_bit_cast(f32x4, hv);
      ufl[j][kk] = __builtin_bit_cast(f32x4, lv);
      asm volatile("" : "+a"(ufh[j][kk]), "+a"(ufl[j][kk]));   // AGPR-class from here on
    }
  }
  const int slot_words = UG * (p.xstride / 4);
  const int s_end = p.s_begin + p.s_count;
  int n[NT];
  float mask[NT], c[NT];
  unsigned* xch[NT];
#pragma unroll
  for (int x = 0; x < NT; ++x) {
    const int bt = bt0 + x;
    n[x] = bt * 16 + nl;
    mask[x] = p.mask_u ? p.mask_u[((size_t)dir * p.n_pad + n[x]) * H + u] : 1.f;
    c[x] = 0.f;
    xch[x] = p.xbuf + (size_t)(dir * p.NB + bt) * p.xchain_words;
    if (p.s_begin > 0) {
      const int tpp = dir == 0 ? p.s_begin - 1 : p.T - p.s_begin;
      c[x] = p.cell[(((size_t)tpp * p.n_pad + n[x]) * 2 + dir) * H + u];
    }
  }
  auto load_zx = [&](int x, int ss) -> float4 {
    const int sc = ss < s_end ? ss : s_end - 1;    // past the end: a valid, unused row
    const int tt = dir == 0 ? sc : p.T - 1 - sc;
    return *reinterpret_cast<const float4*>(
        p.zx + (((size_t)tt * p.n_pad + n[x]) * 2 + dir) * H4 + 4 * u);
  };
  float4 zx_next[NT];
#pragma unroll
  for (int x = 0; x < NT; ++x) zx_next[x] = load_zx(x, p.s_begin);
  // group i = (kk, half): units kbase + 32 kk + 8 g + 4 half .. +3 of sample nl; consecutive
  // groups are xstride bytes apart ((kk, half) -> unit group + 2 kk' + half with kk' = 4 kk)
  constexpr int NL = 2 * NKW;
  const unsigned goff = (unsigned)(((kbase + 8 * g) / 4) * p.xstride + nl * 16);
  const unsigned gstep = (unsigned)p.xstride;      // between the two halves of a kk
  bool dead = false;
  StepProf prof;
  prof.init(false);
  u32x4 v[NT][NL];
  // the exchange slot holding h of step `ss` of tile x
  auto slot = [&](int x, int ss) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(xch[x] + (size_t)(ss & 1) * slot_words, 0,
                                             slot_words * 4, 0x00020000);
  };
  auto load_groups = [&](int x, const __amdgpu_buffer_rsrc_t& rsrc) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      v[x][i] = __builtin_amdgcn_raw_buffer_load_b128(
          rsrc, goff, (unsigned)((8 * (i >> 1) + (i & 1))) * gstep, FAST ? kNt : kSc1);
  };
  auto issue = [&](int x, int ss) {
    for (int i = 0; i < p.prepoll; ++i) __builtin_amdgcn_s_sleep(1);
    load_groups(x, slot(x, ss));
  };
  auto await = [&](int x, int ss, unsigned tag) {
    const unsigned flip = 0u - tag;
    bool stale = !all_tagged<NL>(v[x], flip);
    if (__builtin_amdgcn_ballot_w64(stale) == 0ull) return;
    if (!p.poll || dead) return;
    const __amdgpu_buffer_rsrc_t rsrc = slot(x, ss);
    const long long t0 = wall_clock64();
    bool gave_up = false;
    while (stale) {
#ifdef POLLCOUNT
      if (prof.on) prof.pt[0] += 1000000;
#endif
      for (int i = 0; i < p.repoll; ++i) __builtin_amdgcn_s_sleep(1);
      load_groups(x, rsrc);
      stale = !all_tagged<NL>(v[x], flip);
      if (stale && wall_clock64() - t0 > p.spin) { gave_up = true; break; }
    }
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0ull) {
      dead = true;
      if (gave_up) mark_timeout(p.status);
    }
  };
  // cell update of tile x at step s from the recurrent contribution `a`; publishes h
  auto finish_step = [&](int x, int s, const f32x4& a, const float4& zx4) {
    const int t = dir == 0 ? s : p.T - 1 - s;
    const CellFwd o = cell_forward(a, zx4, c[x], mask[x]);
    c[x] = o.c;
    const unsigned w0 = EXACT ? tag_word(o.hm, (unsigned)(s >> 1) & 1u)
                              : packed_word(o.hm, (unsigned)(s >> 1) & 1u);
    // (the last step's word is published too: nobody reads it, and no branch is needed)
    __builtin_amdgcn_raw_buffer_store_b32(w0, slot(x, s),
                                          (unsigned)(ug * p.xstride + nl * 16 + g * 4), 0,
                                          FAST ? 0 : kSc1);
    const size_t row = (size_t)t * p.n_pad + n[x];
    p.y[row * H2 + dir * H + u] = o.h;
    p.cell[(row * 2 + dir) * H + u] = c[x];
    *reinterpret_cast<float4*>(p.gates + (row * 2 + dir) * H4 + 4 * u) =
        make_float4(o.gi, o.gf, o.gg, o.go);
  };
  // one phase = one step (s >= 1) of tile x
  auto phase = [&](auto xc, int s) {
    constexpr int x = decltype(xc)::value;
    const float4 zx4 = zx_next[x];
    prof.stamp(0);
    await(x, s - 1, (unsigned)((s - 1) >> 1) & 1u);
    prof.stamp(1);
    zx_next[x] = load_zx(x, s + 1);
    // two LDS buffers by step parity (the one barrier per step keeps the waves at most one
    // step apart)
    const int buf = s & 1;
    f32x4* mine = part + ((size_t)buf * 4 + w) * 4 * 64;
    if constexpr (EXACT) {
      // the gathered fp32 words ARE the B operands (tag bit left in: <= 1 ulp); the four gate
      // tiles' accumulator chains are interleaved (32 cycles of pipe per MFMA, 40 of latency)
      f32x4 acc[4];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float bw = __uint_as_float(v[x][m >> 2][m & 3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (m == 0)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[j]) : "a"(uf[j][0]), "v"(bw));
          else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(uf[j][m]), "v"(bw));
        }
      }
      asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[j * 64 + lane] = acc[j];
    } else {
    // exchanged word = fp16 hi << 16 | fp16 lo (tag = LSB of lo, left in place)
    h8 bh[NKW], bl[NKW];
#pragma unroll
    for (int kk = 0; kk < NKW; ++kk) {
      const u32x4 q0 = v[x][2 * kk], q1 = v[x][2 * kk + 1];
      u32x4 hi, lo;
      hi[0] = __builtin_amdgcn_perm(q0[1], q0[0], 0x07060302u);
      hi[1] = __builtin_amdgcn_perm(q0[3], q0[2], 0x07060302u);
      hi[2] = __builtin_amdgcn_perm(q1[1], q1[0], 0x07060302u);
      hi[3] = __builtin_amdgcn_perm(q1[3], q1[2], 0x07060302u);
      lo[0] = __builtin_amdgcn_perm(q0[1], q0[0], 0x05040100u);
      lo[1] = __builtin_amdgcn_perm(q0[3], q0[2], 0x05040100u);
      lo[2] = __builtin_amdgcn_perm(q1[1], q1[0], 0x05040100u);
      lo[3] = __builtin_amdgcn_perm(q1[3], q1[2], 0x05040100u);
      bh[kk] = __builtin_bit_cast(h8, hi);
      bl[kk] = __builtin_bit_cast(h8, lo);
    }
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      f32x4 am0, ac0, am1, ac1;
      FwdMfma<NKW>::run2(am0, ac0, am1, ac1, ufh[j], ufl[j], ufh[j + 1], ufl[j + 1], bh, bl);
      f32x4 r0, r1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r0[e] = __builtin_fmaf(ac0[e], 1.f / kLoScale, am0[e]);
        r1[e] = __builtin_fmaf(ac1[e], 1.f / kLoScale, am1[e]);
      }
      mine[j * 64 + lane] = r0;
      mine[(j + 1) * 64 + lane] = r1;
    }
    }
    prof.stamp(2);
    __syncthreads();
    prof.stamp(3);
    const f32x4* all = part + (size_t)buf * 4 * 4 * 64 + (size_t)w * 64 + lane;
    const f32x4 a = (all[0 * 4 * 64] + all[1 * 4 * 64]) + (all[2 * 4 * 64] + all[3 * 4 * 64]);
    finish_step(x, s, a, zx4);
    prof.stamp(4);
    issue(x, s);                                   // this tile's h of step s, for step s + 1
    prof.stamp(5);
  };
  using T0 = std::integral_constant<int, 0>;
  int s = p.s_begin;
  if (s == 0) {
    // step 0: h_prev = 0, nothing to gather
#pragma unroll
    for (int x = 0; x < NT; ++x) {
      const float4 zx4 = zx_next[x];
      zx_next[x] = load_zx(x, 1);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      finish_step(x, 0, zero, zx4);
    }
    s = 1;
  }
  prof.init((p.dbg & 32) && wg == 0 && unit == p.chain_begin);
  if (s < s_end) {
    issue(0, s - 1);
    for (; s < s_end; ++s) phase(T0{}, s);
  }
  prof.flush(p.status, w);
}

template <int NKW, bool EXACT>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel_x(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit_local, wg;
  if (!map_block(p, unit_local, wg)) return;
  const int unit = p.chain_begin + unit_local;
  const bool fast = chain_on_one_xcd(p, unit, wg, reinterpret_cast<int*>(lds));
  if (fast) fwd_body_x<NKW, true, EXACT>(p, unit, wg, lds);
  else fwd_body_x<NKW, false, EXACT>(p, unit, wg, lds);
}

// ---------------------------------------------------------------------------
// forward, ONE utterance (predict.py:73-93: the reference decodes one file per call).  A
// 16-row MFMA tile would be 15/16 padding and would exchange 16 x H words per step for one
// useful row, so this kernel has no tile: a chain is one direction, a workgroup owns 16
// units = 64 gate columns, thread (kq = tid >> 6, c = tid & 63) keeps the H/4 entries
// U[kq H/4 .., 64 wg + c] in registers and multiplies them with its quarter of h in plain
// fp32 FMAs (EXACT fp32: no split), the four partial sums of a column meet in LDS, threads
// 0..15 finish one unit each.  The exchange is H words per step (tag in the LSB as
// everywhere), gathered by H/4 lanes with one 16-byte load each: the step is the bare
// hand-off latency plus ~0.25 us of arithmetic.  Only row 0 of the slabs is read / written.
template <int KQ /* H / 4 */, bool FAST>
__device__ __forceinline__ void fwd_body_n1(const LstmParams& p, int dir, int wg, float* lds) {
  const int tid = threadIdx.x;
  const int kq = tid >> 6, c = tid & 63;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  float* hs = lds;                       // [H] h_{t-1}
  float* part = lds + H;                 // [4][64] partial gate sums
  float u[KQ];
#pragma unroll
  for (int i = 0; i < KQ; ++i)
    u[i] = p.U[((size_t)(dir * H + kq * KQ + i)) * H4 + 64 * wg + c];
  const int unit = 16 * wg + (tid & 15);
  const float mask = p.mask_u ? p.mask_u[((size_t)dir * p.n_pad) * H + unit] : 1.f;
  float cst = 0.f;
  const int s_end = p.s_begin + p.s_count;
  if (p.s_begin > 0 && tid < 16) {
    const int tpp = dir == 0 ? p.s_begin - 1 : p.T - p.s_begin;
    cst = p.cell[(((size_t)tpp * p.n_pad) * 2 + dir) * H + unit];
  }
  unsigned* xch = p.xbuf + (size_t)dir * p.xchain_words;      // [2 slots][H]
  bool dead = false;
  for (int s = p.s_begin; s < s_end; ++s) {
    const int t = dir == 0 ? s : p.T - 1 - s;
    float4 zx4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 16)
      zx4 = *reinterpret_cast<const float4*>(p.zx + (((size_t)t * p.n_pad) * 2 + dir) * H4 + 4 * unit);
    if (s > 0) {
      if (tid < H / 4) {
        const unsigned flip = 0u - ((unsigned)((s - 1) >> 1) & 1u);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            xch + (size_t)((s - 1) & 1) * H, 0, H * 4, 0x00020000);
        for (int i = 0; i < p.prepoll; ++i) __builtin_amdgcn_s_sleep(1);
        u32x4 v = xload<FAST>(rsrc, (unsigned)tid * 16);
        if (p.poll && !dead) {
          long long t0 = 0;
          bool timing = false;
          while ((((v[0] ^ flip) | (v[1] ^ flip)) | ((v[2] ^ flip) | (v[3] ^ flip))) & 1u) {
            if (!timing) { t0 = wall_clock64(); timing = true; }
            else if (wall_clock64() - t0 > p.spin) { dead = true; mark_timeout(p.status); break; }
            __builtin_amdgcn_s_sleep(1);
            v = xload<FAST>(rsrc, (unsigned)tid * 16);
          }
        }
        *reinterpret_cast<float4*>(hs + 4 * tid) =
            make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                        __uint_as_float(v[3]));
      }
      __syncthreads();
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; i += 4) {
        const float4 h4 = *reinterpret_cast<const float4*>(hs + kq * KQ + i);
        acc0 = __builtin_fmaf(h4.x, u[i], acc0);
        acc1 = __builtin_fmaf(h4.y, u[i + 1], acc1);
        acc0 = __builtin_fmaf(h4.z, u[i + 2], acc0);
        acc1 = __builtin_fmaf(h4.w, u[i + 3], acc1);
      }
      part[kq * 64 + c] = acc0 + acc1;
      __syncthreads();
    }
    if (tid < 16) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      if (s > 0) {
#pragma unroll
        for (int gidx = 0; gidx < 4; ++gidx) {
          const int col = 4 * tid + gidx;
          a[gidx] = (part[col] + part[64 + col]) + (part[128 + col] + part[192 + col]);
        }
      }
      const CellFwd o = cell_forward(a, zx4, cst, mask);
      cst = o.c;
      __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)(s & 1) * H, 0, H * 4, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b32(tag_word(o.hm, (unsigned)(s >> 1) & 1u), wr,
                                            (unsigned)unit * 4, 0, FAST ? 0 : kSc1);
      const size_t row = (size_t)t * p.n_pad;
      p.y[row * H2 + dir * H + unit] = o.h;
      p.cell[(row * 2 + dir) * H + unit] = cst;
      *reinterpret_cast<float4*>(p.gates + (row * 2 + dir) * H4 + 4 * unit) =
          make_float4(o.gi, o.gf, o.gg, o.go);
    }
    // hs / part are rewritten only after the next step's gather, which the 16 finishing
    // threads reach after reading part; the gather's barrier orders the rest
    __syncthreads();
  }
}

template <int KQ>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel_n1(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // blockIdx -> (direction, workgroup): both directions' workgroups use block ids
  // congruent mod 8 each, as map_block does for chains
  int dir, wg;
  if (!map_block(p, dir, wg)) return;
  dir += p.chain_begin;
  const bool fast = chain_on_one_xcd(p, dir, wg, reinterpret_cast<int*>(lds));
  __syncthreads();
  if (fast) fwd_body_n1<KQ, true>(p, dir, wg, lds);
  else fwd_body_n1<KQ, false>(p, dir, wg, lds);
}

// am = Uh0*Bh0 + Uh1*Bh1 ; ac = Uh0*Bl0 + Ul0*Bh0 + Uh1*Bl1 + Ul1*Bh1 (K = 2 x 32), i.e.
// U^T-slice x dz tile = am + ac / 2048.  U fragments "a" (AGPR), B fragments and results "v".
// s_nop 1: a VALU-written B operand needs 2 wait states before an MFMA reads it; trailing
// s_nop: an MFMA's D needs its pass count + 4 states before a VALU reads it (hipcc pads one).
__device__ __forceinline__ void mfma_hl_tile(f32x4& am, f32x4& ac, const f32x4& uh0,
                                             const f32x4& ul0, const f32x4& uh1,
                                             const f32x4& ul1, const h8& bh0, const h8& bl0,
                                             const h8& bh1, const h8& bl1) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %2, %6, 0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %2, %7, 0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %3, %6, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %4, %9, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
      "s_nop 11"
      : "=&v"(am), "=&v"(ac)
      : "a"(uh0), "a"(ul0), "a"(uh1), "a"(ul1), "v"(bh0), "v"(bl0), "v"(bh1), "v"(bl1));
}

// Sum over the 16 samples of a batch tile of every thread's float4 (thread = (sample tid>>4,
// unit tid&15)) -> dst[64 gate columns of this workgroup], in a fixed order (deterministic).
__device__ __forceinline__ void tile_gate_sums(float4 gsum, float* lds, float* dst,
                                               bool accumulate, int ncols = 64) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float v[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {           // over the wave's four samples (lane >> 4)
    v[k] += __shfl_xor(v[k], 16);
    v[k] += __shfl_xor(v[k], 32);
  }
  __syncthreads();
  if (lane < 16) *reinterpret_cast<float4*>(lds + (w * 16 + lane) * 4) =
      make_float4(v[0], v[1], v[2], v[3]);
  __syncthreads();
  if (tid < ncols) {                      // column tid = unit (tid >> 2), gate (tid & 3)
    const float t = ((lds[tid] + lds[64 + tid]) + lds[128 + tid]) + lds[192 + tid];
    dst[tid] = (accumulate ? dst[tid] : 0.f) + t;
  }
  __syncthreads();
}

// backward, split-fp16 MFMA variant.  The gate gradients dz span many orders of
// magnitude, so each batch column n is scaled by its own power of two (max |dz|
// over the WG's 64 columns -> [2^8, 2^9)) before the fp16 split and the partial
// sums are unscaled exactly afterwards.
template <int TPW, bool FAST, bool VAR>
__device__ __forceinline__ void bwd_body_h(const LstmParams& p, int chain, int cw, float* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int P = p.P;
  const int dir = chain / p.NB, bt = chain % p.NB;
  constexpr int DZH = 72;                         // LDS row stride of the dz tiles (halfs)
  // dz tiles for the MFMA stage, double-buffered by step parity: with no barrier between
  // gather and cell math, a wave may write step s+1's tile while another still multiplies
  // step s's (the one barrier per step keeps them at most one step apart)
  constexpr int kTileFloats = 16 + (2 * 16 * DZH) / 2;        // sinv + hi + lo, in floats
  float* sinv0 = lds;

  h8 ufh[TPW][2], ufl[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int mt = w + 4 * i;
    const int krow = 16 * mt + nl;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 64 * cw + 32 * kk + 8 * g + e;
        const float x = (mt < P && krow < H && j < H4)
                            ? p.U[((size_t)(dir * H + krow)) * H4 + j] : 0.f;
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        ufh[i][kk][e] = hi; ufl[i][kk][e] = lo;
      }
    }
  }
  const int cn = bt * 16 + (tid >> 4);
  const int cu = 16 * cw + (tid & 15);
  const bool cvalid = cu < H;
  float cmask = 1.f;
  if (cvalid && p.mask_u) cmask = p.mask_u[((size_t)dir * p.n_pad + cn) * H + cu];
  float dc = 0.f;
  float dhz = 0.f;                                 // VAR: (1 - k_h) dh carried to the next step
  float zmax = 0.f;
  float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);   // sum over steps of this (sample, unit)'s dz
  const bool has_mi = VAR && p.mi != nullptr;
  float4 mi_a = make_float4(0.f, 0.f, 0.f, 0.f), mi_b1 = mi_a, mi_b2 = mi_a;
  float4 g_a = mi_a, g_b1 = mi_a, g_b2 = mi_a, g_b = mi_a;     // parameter-gradient sums
  if (has_mi && cvalid) {
    const float* m = p.mi + (size_t)dir * 4 * H4 + 4 * cu;
    mi_a = *reinterpret_cast<const float4*>(m);
    mi_b1 = *reinterpret_cast<const float4*>(m + H4);
    mi_b2 = *reinterpret_cast<const float4*>(m + 2 * H4);
  }
  if (cvalid && p.s_begin > 0) {
    dc = p.dc_state[((size_t)dir * p.n_pad + cn) * H + cu];
    if (VAR) dhz = p.dc_state[((size_t)(2 + dir) * p.n_pad + cn) * H + cu];
  }
  bool dead = false;
  unsigned* xch = p.xbuf + (size_t)chain * p.xchain_words;
  const size_t slot_words = (size_t)P * P * 256;
  const int s_end = p.s_begin + p.s_count;
  constexpr int NL = TPW;
  const bool prof = (p.dbg & 32) && cw == 0 && chain == p.chain_begin && lane == 0;
  long long pt[4] = {0, 0, 0, 0}, tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;

  float nx_dy = 0.f, nx_c = 0.f, nx_cp = 0.f, nx_kc = 1.f, nx_kh = 1.f;
  float4 nx_g = make_float4(0.f, 0.f, 0.f, 0.f), nx_uh = nx_g, nx_wx = nx_g;
  auto load_slabs = [&](int ss) {
    nx_dy = 0.f; nx_c = 0.f; nx_cp = 0.f; nx_g = make_float4(0.f, 0.f, 0.f, 0.f);
    nx_kc = 1.f; nx_kh = 1.f; nx_uh = nx_g; nx_wx = nx_g;
    if (!cvalid || ss >= s_end) return;
    const int tt = dir == 0 ? p.T - 1 - ss : ss;
    const int tcc = dir == 0 ? tt - 1 : tt + 1;
    const size_t row = (size_t)tt * p.n_pad + cn;
    nx_dy = p.dy[row * H2 + dir * H + cu];
    nx_c = p.cell[(row * 2 + dir) * H + cu];
    if (ss + 1 < p.T) nx_cp = p.cell[(((size_t)tcc * p.n_pad + cn) * 2 + dir) * H + cu];
    nx_g = *reinterpret_cast<const float4*>(p.gates + (row * 2 + dir) * H4 + 4 * cu);
    if (VAR) {
      if (p.zone_c) nx_kc = p.zone_c[((size_t)tt * 2 + dir) * H + cu];
      if (p.zone_h) nx_kh = p.zone_h[((size_t)tt * 2 + dir) * H + cu];
      if (has_mi) {
        nx_uh = *reinterpret_cast<const float4*>(p.uh + (row * 2 + dir) * H4 + 4 * cu);
        nx_wx = *reinterpret_cast<const float4*>(p.wx + (row * 2 + dir) * H4 + 4 * cu);
      }
    }
  };
  load_slabs(p.s_begin);

  for (int s = p.s_begin; s < s_end; ++s) {
    if (prof) tk0 = wall_clock64();
    float* sinv = sinv0 + (size_t)(s & 1) * kTileFloats;      // [16] 1/scale per batch column
    _Float16* dzh = reinterpret_cast<_Float16*>(sinv + 16);   // [16][DZH] hi
    _Float16* dzl = dzh + 16 * DZH;                           // [16][DZH] lo
    const int t = dir == 0 ? p.T - 1 - s : s;
    const float dyv = nx_dy, cv = nx_c, cpv = nx_cp, kc = nx_kc, kh = nx_kh;
    const float4 gt = nx_g, uh4 = nx_uh, wx4 = nx_wx;
    float dh_rec = 0.f;
    if (s > 0) {
      const unsigned tag = (unsigned)((s - 1) >> 1) & 1u;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)((s - 1) & 1) * slot_words + (size_t)cw * P * 256, 0, P * 256 * 4,
          0x00020000);
      // Lane (sample s4 = lane>>4 of this wave's four, unit quad q = (lane>>2)&3, sub =
      // lane&3) gathers the 16-byte group (sample, quad) from the producers sub*TPW+i:
      // the four loads are summed in registers and the four `sub` lanes with two DPP quad
      // permutes -- every lane then holds the complete dh of its (sample, quad) and picks
      // its own unit.  No LDS round trip and no barrier between gather and cell math.
      unsigned off[NL];
      bool use[NL];
      u32x4 v[NL];
      const int sub = lane & 3;
      const int grp_in_tile = (4 * w + (lane >> 4)) * 4 + ((lane >> 2) & 3);   // 16-B groups
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int pr = sub * TPW + i;
        use[i] = pr < P;
        off[i] = (unsigned)((pr * 64 + grp_in_tile) * 16);
      }
      gather_groups<FAST, NL>(rsrc, off, use, tag, p.poll, dead, p.status, v, p.dbg & 64,
                              p.prepoll, p.repoll, p.spin);
      if (prof) tk1 = wall_clock64();
      load_slabs(s + 1);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if (use[i]) {
          acc.x += __uint_as_float(v[i][0] & ~1u); acc.y += __uint_as_float(v[i][1] & ~1u);
          acc.z += __uint_as_float(v[i][2] & ~1u); acc.w += __uint_as_float(v[i][3] & ~1u);
        }
      }
      acc.x += quad_swap1(acc.x); acc.y += quad_swap1(acc.y);
      acc.z += quad_swap1(acc.z); acc.w += quad_swap1(acc.w);
      acc.x += quad_swap2(acc.x); acc.y += quad_swap2(acc.y);
      acc.z += quad_swap2(acc.z); acc.w += quad_swap2(acc.w);
      if (prof) tk2 = wall_clock64();
      dh_rec = sub == 0 ? acc.x : sub == 1 ? acc.y : sub == 2 ? acc.z : acc.w;
    } else {
      load_slabs(s + 1);
    }
    {
      float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cvalid) {
        const float gi = gt.x, gf = gt.y, gg = gt.z, go = gt.w;
        float dh = dyv + cmask * dh_rec;
        if (VAR) {                      // h = h_prev + k_h (h~ - h_prev)
          dh += dhz;
          dhz = (1.f - kh) * dh;
          dh *= kh;
        }
        const float tch = fast_tanh(cv);
        const float d_o = dh * tch;
        float dcc = dc + dh * go * (1.f - tch * tch);
        float dcz = 0.f;
        if (VAR) {                      // c = c_prev + k_c (c~ - c_prev)
          dcz = (1.f - kc) * dcc;
          dcc *= kc;
        }
        const float d_i = dcc * gg, d_g = dcc * gi, d_f = dcc * cpv;
        dc = dcc * gf + dcz;
        z4.x = d_i * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
        z4.y = d_f * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
        z4.z = d_g * (1.f - gg * gg);
        z4.w = d_o * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
        gsum.x += z4.x; gsum.y += z4.y; gsum.z += z4.z; gsum.w += z4.w;
        const size_t zoff = (((size_t)t * p.n_pad + cn) * 2 + dir) * H4 + 4 * cu;
        if (has_mi) {
          // z = alpha Wx Uh + beta1 Uh + beta2 Wx + b: the recurrent product sees
          // dz (alpha Wx + beta1), the input projection dz (alpha Uh + beta2)
          g_a.x += z4.x * wx4.x * uh4.x; g_a.y += z4.y * wx4.y * uh4.y;
          g_a.z += z4.z * wx4.z * uh4.z; g_a.w += z4.w * wx4.w * uh4.w;
          g_b1.x += z4.x * uh4.x; g_b1.y += z4.y * uh4.y; g_b1.z += z4.z * uh4.z; g_b1.w += z4.w * uh4.w;
          g_b2.x += z4.x * wx4.x; g_b2.y += z4.y * wx4.y; g_b2.z += z4.z * wx4.z; g_b2.w += z4.w * wx4.w;
          g_b.x += z4.x; g_b.y += z4.y; g_b.z += z4.z; g_b.w += z4.w;
          const float4 dwx = make_float4(z4.x * (mi_a.x * uh4.x + mi_b2.x), z4.y * (mi_a.y * uh4.y + mi_b2.y),
                                         z4.z * (mi_a.z * uh4.z + mi_b2.z), z4.w * (mi_a.w * uh4.w + mi_b2.w));
          *reinterpret_cast<float4*>(p.dwx + zoff) = dwx;
          zmax = fmaxf(zmax, fmaxf(fmaxf(fabsf(dwx.x), fabsf(dwx.y)), fmaxf(fabsf(dwx.z), fabsf(dwx.w))));
          z4.x *= mi_a.x * wx4.x + mi_b1.x; z4.y *= mi_a.y * wx4.y + mi_b1.y;
          z4.z *= mi_a.z * wx4.z + mi_b1.z; z4.w *= mi_a.w * wx4.w + mi_b1.w;
        }
        *reinterpret_cast<float4*>(p.dz + zoff) = z4;
      }
      // power-of-two scale of this batch column: max over its 16 threads (one DPP row)
      float m = fmaxf(fmaxf(fabsf(z4.x), fabsf(z4.y)), fmaxf(fabsf(z4.z), fabsf(z4.w)));
      zmax = fmaxf(zmax, m);
      m = row16_max(m);
      int ex = 0;
      if (m > 0.f) (void)frexpf(m, &ex); else ex = 9;
      ex = ex < -100 ? -100 : ex;                 // keep 2^(9-ex) finite for denormal maxima
      const float sc = ldexpf(1.f, 9 - ex);
      if ((tid & 15) == 0) sinv[tid >> 4] = ldexpf(1.f, ex - 9);
      h4 hi4, lo4;
      {
        _Float16 a, b;
        split_f16(z4.x * sc, a, b); hi4[0] = a; lo4[0] = b;
        split_f16(z4.y * sc, a, b); hi4[1] = a; lo4[1] = b;
        split_f16(z4.z * sc, a, b); hi4[2] = a; lo4[2] = b;
        split_f16(z4.w * sc, a, b); hi4[3] = a; lo4[3] = b;
      }
      *reinterpret_cast<h4*>(dzh + (tid >> 4) * DZH + 4 * (tid & 15)) = hi4;
      *reinterpret_cast<h4*>(dzl + (tid >> 4) * DZH + 4 * (tid & 15)) = lo4;
    }
    __syncthreads();
    if (prof) tk3 = wall_clock64();
    if (s + 1 < p.T) {
      h8 bh[2], bl[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bh[kk] = *reinterpret_cast<const h8*>(dzh + nl * DZH + 32 * kk + 8 * g);
        bl[kk] = *reinterpret_cast<const h8*>(dzl + nl * DZH + 32 * kk + 8 * g);
      }
      const float us = sinv[nl];
      const unsigned wtag = (unsigned)(s >> 1) & 1u;
      __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)(s & 1) * slot_words, 0, (unsigned)(slot_words * 4), 0x00020000);
      u32x4 o[TPW];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int mt = w + 4 * i;
        f32x4 am = {0.f, 0.f, 0.f, 0.f}, ac0 = am, ac1 = am;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[i][kk], bh[kk], am, 0, 0, 0);
          ac0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufh[i][kk], bl[kk], ac0, 0, 0, 0);
          ac1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ufl[i][kk], bh[kk], ac1, 0, 0, 0);
        }
        const f32x4 a = (am + (ac0 + ac1) * (1.f / kLoScale)) * us;
        o[i][0] = tag_word(a[0], wtag); o[i][1] = tag_word(a[1], wtag);
        o[i][2] = tag_word(a[2], wtag); o[i][3] = tag_word(a[3], wtag);
        const unsigned off = mt < P
            ? (unsigned)((((size_t)mt * P + cw) * 256 + nl * 16 + 4 * g) * 4)
            : 0xFFFFFFF0u;
        xstore<FAST>(o[i], wr, off);
      }
    }
    if (prof && s > 0) {
      const long long tk4 = wall_clock64();
      pt[0] += tk1 - tk0; pt[1] += tk2 - tk1; pt[2] += tk3 - tk2; pt[3] += tk4 - tk3;
    }
  }
  if (prof) {
    long long* out = reinterpret_cast<long long*>(p.status + 16) + 6 * w;
    for (int i = 0; i < 4; ++i) out[i] = pt[i];
  }
  if (cvalid && p.dc_state) {
    p.dc_state[((size_t)dir * p.n_pad + cn) * H + cu] = dc;
    if (VAR) p.dc_state[((size_t)(2 + dir) * p.n_pad + cn) * H + cu] = dhz;
  }
  if (has_mi && p.dmi) {
    // sums over this tile's 16 samples (LDS float atomics, once per launch), then this
    // workgroup's slice of the (NB, 2, 4, 4H) partial-gradient array (+= across slices)
    __syncthreads();
    float* accum = lds;                              // [4 params][64 gate columns]
    accum[tid] = 0.f;
    __syncthreads();
    if (cvalid) {
      const float4 gs[4] = {g_a, g_b1, g_b2, g_b};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* a = accum + k * 64 + 4 * (tid & 15);
        atomicAdd(a + 0, gs[k].x); atomicAdd(a + 1, gs[k].y);
        atomicAdd(a + 2, gs[k].z); atomicAdd(a + 3, gs[k].w);
      }
    }
    __syncthreads();
    const int k = tid >> 6, jcol = 64 * cw + (tid & 63);
    if (jcol < H4) {
      float* dst = p.dmi + (((size_t)bt * 2 + dir) * 4 + k) * H4 + jcol;
      *dst = (p.s_begin > 0 ? *dst : 0.f) + accum[tid];
    }
  }
  if (p.db_part) {
    const int left = H4 - 64 * cw;
    tile_gate_sums(gsum, lds, p.db_part + ((size_t)bt * 2 + dir) * H4 + 64 * cw, p.s_begin > 0,
                   left < 64 ? left : 64);
  }
  if (p.dz_absmax) {
    zmax = asr_wave_max(zmax);
    if (lane == 0 && zmax > 0.f) atomicMax(p.dz_absmax, __float_as_uint(zmax));
  }
}

template <int TPW>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel_h(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int chain_local, cw;
  if (!map_block(p, chain_local, cw)) return;
  const int chain = p.chain_begin + chain_local;
  const bool fast = chain_on_one_xcd(p, chain, cw, reinterpret_cast<int*>(lds));
  if (fast) bwd_body_h<TPW, true, false>(p, chain, cw, lds);
  else bwd_body_h<TPW, false, false>(p, chain, cw, lds);
}

template <int TPW>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel_hv(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int chain_local, cw;
  if (!map_block(p, chain_local, cw)) return;
  const int chain = p.chain_begin + chain_local;
  const bool fast = chain_on_one_xcd(p, chain, cw, reinterpret_cast<int*>(lds));
  if (fast) bwd_body_h<TPW, true, true>(p, chain, cw, lds);
  else bwd_body_h<TPW, false, true>(p, chain, cw, lds);
}

// ---------------------------------------------------------------------------
// backward, split-fp16, third generation (plain cell, H = 256 / 512, persistent mode): the
// default BPTT kernel.  NT = 2: two batch tiles per workgroup as bwd_body_h2 (a tile's gather
// is in flight during the other tile's phase); NT = 1: one tile, gather issued right after the
// publish.  What changed against bwd_body_h / _h2 (static count of one step at H = 512: about
// 1900 instructions -> about 700):
//  * the MFMAs are inline asm with the U^T fragments as AGPR operands and the results in
//    VGPRs.  With the builtin hipcc keeps the 128 stationary fragment registers of TPW = 8 in
//    VGPRs, accumulates into AGPRs and, short of VGPRs, parks the gathered words in AGPRs too:
//    ~640 v_accvgpr_read/write per step, every one on the critical path of a wave that is
//    alone on its SIMD.  (Wait states the compiler cannot pad are inside the asm string.)
//  * tag test = OR-reduction of (word ^ -tag) and ONE compare per lane, one ballot per wave,
//    instead of a compare per word whose lane masks met in ~75 dependent scalar operations;
//  * the gathered words are added WITH their tag bit (<= 1 ulp, as clearing it was);
//  * no branch around any vector-memory instruction in the steady loop, gather offsets as
//    immediates of one base register;
//  * the bias gradient (sum over samples and steps of dz) is accumulated in registers and
//    leaves as per-tile partial sums (LstmParams::db_part): no pass over the dz slab after it.
// Arithmetic: products and summation order of a (sample, unit) are the same for NT = 1 and 2,
// for sliced and whole sequences and for both transports.
template <int TPW, bool FAST>
__device__ __forceinline__ void bwd_body_x(const LstmParams& p, int unit, int cw, float* lds) {
  constexpr int NT = 1;                            // batch tiles per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int P = p.P;                               // == 4 * TPW here
  const int dir = unit / p.NB, bt0 = unit % p.NB;
  constexpr int DZH = 72;                         // LDS row stride of the dz tiles (halfs)
  constexpr int kTileFloats = 16 + (2 * 16 * DZH) / 2;        // sinv + hi + lo, in floats
  // two dz tile buffers by step parity (the one barrier per step keeps the waves at most one
  // step apart)

  f32x4 ufh[TPW][2], ufl[TPW][2];                  // bit patterns of 8 halfs each (AGPRs)
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int krow = 16 * (w + 4 * i) + nl;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 64 * cw + 32 * kk + 8 * g + e;
        _Float16 hi, lo;
        split_f16(p.U[((size_t)(dir * H + krow)) * H4 + j], hi, lo);
        hv[e] = hi; lv[e] = lo;
      }
      ufh[i][kk] = __builtin_bit_cast(f32x4, hv);
      ufl[i][kk] = __builtin_bit_cast(f32x4, lv);
      // from here on the fragments are AGPR-class values (defined by an asm "a" operand),
      // so the MFMA statements read them in place instead of copying them in per use
      asm volatile("" : "+a"(ufh[i][kk]), "+a"(ufl[i][kk]));
    }
  }
  const int cu = 16 * cw + (tid & 15);
  const int s_end = p.s_begin + p.s_count;
  const size_t slot_words = (size_t)P * P * 256;
  int cn[NT];
  float cmask[NT], dc[NT];
  float4 gsum[NT];
  unsigned* xch[NT];
#pragma unroll
  for (int x = 0; x < NT; ++x) {
    const int bt = bt0 + x;
    cn[x] = bt * 16 + (tid >> 4);
    cmask[x] = p.mask_u ? p.mask_u[((size_t)dir * p.n_pad + cn[x]) * H + cu] : 1.f;
    dc[x] = p.s_begin > 0 ? p.dc_state[((size_t)dir * p.n_pad + cn[x]) * H + cu] : 0.f;
    xch[x] = p.xbuf + (size_t)(dir * p.NB + bt) * p.xchain_words;
    gsum[x] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float zmax = 0.f;
  bool dead = false;
  StepProf prof;
  prof.init(false);

  // slab values of the NEXT step of each tile, prefetched one step ahead
  float nx_dy[NT], nx_c[NT], nx_cp[NT];
  float4 nx_g[NT];
  auto load_slabs = [&](int x, int ss) {
    const int sc = ss < s_end ? ss : s_end - 1;    // past the end: a valid, unused row
    const int tt = dir == 0 ? p.T - 1 - sc : sc;
    const bool has_prev = sc + 1 < p.T;            // the sequence's first frame has c_prev = 0
    const int tcc = has_prev ? (dir == 0 ? tt - 1 : tt + 1) : tt;
    const size_t row = (size_t)tt * p.n_pad + cn[x];
    nx_dy[x] = p.dy[row * H2 + dir * H + cu];
    nx_c[x] = p.cell[(row * 2 + dir) * H + cu];
    const float cp = p.cell[(((size_t)tcc * p.n_pad + cn[x]) * 2 + dir) * H + cu];
    nx_cp[x] = has_prev ? cp : 0.f;
    nx_g[x] = *reinterpret_cast<const float4*>(p.gates + (row * 2 + dir) * H4 + 4 * cu);
  };
#pragma unroll
  for (int x = 0; x < NT; ++x) load_slabs(x, p.s_begin);

  // Lane (sample = lane>>4 of this wave's four, unit quad = (lane>>2)&3, sub = lane&3)
  // gathers the 16-byte group (sample, quad) of the partial dh tiles of producers
  // sub*TPW+i (1 KB apart: immediates of one offset register); summed in registers, then
  // over the four `sub` lanes with DPP quad permutes.
  constexpr int NL = TPW;
  const int sub = lane & 3;
  const unsigned goff = (unsigned)(((sub * TPW) * 64 + (4 * w + (lane >> 4)) * 4 +
                                    ((lane >> 2) & 3)) * 16);
  u32x4 v[NT][NL];
  // this workgroup's region of the slot that holds the partial tiles of step `ss`, tile x
  auto rslot = [&](int x, int ss) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(
        xch[x] + (size_t)(ss & 1) * slot_words + (size_t)cw * P * 256, 0, P * 256 * 4, 0x00020000);
  };
  auto load_groups = [&](int x, const __amdgpu_buffer_rsrc_t& rsrc) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      v[x][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, i * 1024, FAST ? kNt : kSc1);
  };
  auto issue = [&](int x, int ss) {
    for (int i = 0; i < p.prepoll; ++i) __builtin_amdgcn_s_sleep(1);
    load_groups(x, rslot(x, ss));
  };
  // waits until every gathered word of tile x carries `tag` (re-reading the stale lanes' groups)
  auto await = [&](int x, int ss, unsigned tag) {
    const unsigned flip = 0u - tag;
    bool stale = !all_tagged<NL>(v[x], flip);
    if (__builtin_amdgcn_ballot_w64(stale) == 0ull) return;
    if (!p.poll || dead) return;
    const __amdgpu_buffer_rsrc_t rsrc = rslot(x, ss);
    const long long t0 = wall_clock64();
    bool gave_up = false;
    while (stale) {
      for (int i = 0; i < p.repoll; ++i) __builtin_amdgcn_s_sleep(1);
      load_groups(x, rsrc);
      stale = !all_tagged<NL>(v[x], flip);
      if (stale && wall_clock64() - t0 > p.spin) { gave_up = true; break; }
    }
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0ull) {
      dead = true;
      if (gave_up) mark_timeout(p.status);
    }
  };

  // everything of one step of tile x after its recurrent gradient dh_rec is known: cell
  // gradient, dz slab + LDS tile, barrier, partial dh tiles = U^T-slice x dz, publish.
  // ISSUE: whether the gather of this tile's partial tiles of step os is issued on the way.
  auto tail = [&](auto xc, auto issue_c, int s, float dh_rec, int os) {
    constexpr int x = decltype(xc)::value;
    constexpr bool ISSUE = decltype(issue_c)::value;
    float* sinv = lds + (size_t)(s & 1) * kTileFloats;        // [16] 1/scale
    _Float16* dzh = reinterpret_cast<_Float16*>(sinv + 16);   // [16][DZH] hi
    _Float16* dzl = dzh + 16 * DZH;                           // [16][DZH] lo
    const int t = dir == 0 ? p.T - 1 - s : s;
    {
      const float4 gt = nx_g[x];
      const float dyv = nx_dy[x], cv = nx_c[x], cpv = nx_cp[x];
      load_slabs(x, s + 1);
      const float gi = gt.x, gf = gt.y, gg = gt.z, go = gt.w;
      const float dh = dyv + cmask[x] * dh_rec;
      const float tch = fast_tanh(cv);
      const float d_o = dh * tch;
      const float dcc = dc[x] + dh * go * (1.f - tch * tch);
      const float d_i = dcc * gg, d_g = dcc * gi, d_f = dcc * cpv;
      dc[x] = dcc * gf;
      float4 z4;
      z4.x = d_i * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
      z4.y = d_f * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
      z4.z = d_g * (1.f - gg * gg);
      z4.w = d_o * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
      *reinterpret_cast<float4*>(p.dz + (((size_t)t * p.n_pad + cn[x]) * 2 + dir) * H4 + 4 * cu) = z4;
      gsum[x].x += z4.x; gsum[x].y += z4.y; gsum[x].z += z4.z; gsum[x].w += z4.w;
      // power-of-two scale of this batch column: max over its 16 threads (one DPP row)
      float m = fmaxf(fmaxf(fabsf(z4.x), fabsf(z4.y)), fmaxf(fabsf(z4.z), fabsf(z4.w)));
      zmax = fmaxf(zmax, m);
      m = row16_max(m);
      int ex = 0;
      if (m > 0.f) (void)frexpf(m, &ex); else ex = 9;
      ex = ex < -100 ? -100 : ex;                 // keep 2^(9-ex) finite for denormal maxima
      const float sc = ldexpf(1.f, 9 - ex);
      if ((tid & 15) == 0) sinv[tid >> 4] = ldexpf(1.f, ex - 9);
      h4 hi4, lo4;
      _Float16 a, b;
      split_f16(z4.x * sc, a, b); hi4[0] = a; lo4[0] = b;
      split_f16(z4.y * sc, a, b); hi4[1] = a; lo4[1] = b;
      split_f16(z4.z * sc, a, b); hi4[2] = a; lo4[2] = b;
      split_f16(z4.w * sc, a, b); hi4[3] = a; lo4[3] = b;
      *reinterpret_cast<h4*>(dzh + (tid >> 4) * DZH + 4 * (tid & 15)) = hi4;
      *reinterpret_cast<h4*>(dzl + (tid >> 4) * DZH + 4 * (tid & 15)) = lo4;
    }
    prof.stamp(2);
    __syncthreads();
    prof.stamp(3);
    {
      h8 bh[2], bl[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bh[kk] = *reinterpret_cast<const h8*>(dzh + nl * DZH + 32 * kk + 8 * g);
        bl[kk] = *reinterpret_cast<const h8*>(dzl + nl * DZH + 32 * kk + 8 * g);
      }
      const float us = sinv[nl];
      const float usl = us * (1.f / kLoScale);
      const unsigned wtag = (unsigned)(s >> 1) & 1u;
      // (the last step's tiles are published too: nobody reads them, and no branch is needed)
      const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch[x] + (size_t)(s & 1) * slot_words, 0, (unsigned)(slot_words * 4), 0x00020000);
      const unsigned soff = (unsigned)((((size_t)w * P + cw) * 256 + nl * 16 + 4 * g) * 4);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        f32x4 am, ac;
        mfma_hl_tile(am, ac, ufh[i][0], ufl[i][0], ufh[i][1], ufl[i][1], bh[0], bl[0], bh[1],
                     bl[1]);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = tag_word(__builtin_fmaf(ac[e], usl, am[e] * us), wtag);
        // partial tile of the output units 16 (w + 4 i) ..: 4 P KB apart
        __builtin_amdgcn_raw_buffer_store_b128(o, wr, soff, i * (4 * P * 1024), FAST ? 0 : kSc1);
      }
    }
    prof.stamp(4);
    if (ISSUE) issue(x, os);
    prof.stamp(5);
  };
  // one phase = one step (s >= 1) of tile x: finish its gather, reduce, then `tail`.
  auto phase = [&](auto xc, int s) {
    constexpr int x = decltype(xc)::value;
    prof.stamp(0);
    await(x, s - 1, (unsigned)((s - 1) >> 1) & 1u);
    prof.stamp(1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      acc.x += __uint_as_float(v[x][i][0]); acc.y += __uint_as_float(v[x][i][1]);
      acc.z += __uint_as_float(v[x][i][2]); acc.w += __uint_as_float(v[x][i][3]);
    }
    acc.x += quad_swap1(acc.x); acc.y += quad_swap1(acc.y);
    acc.z += quad_swap1(acc.z); acc.w += quad_swap1(acc.w);
    acc.x += quad_swap2(acc.x); acc.y += quad_swap2(acc.y);
    acc.z += quad_swap2(acc.z); acc.w += quad_swap2(acc.w);
    const float dh_rec = sub == 0 ? acc.x : sub == 1 ? acc.y : sub == 2 ? acc.z : acc.w;
    // the gather issued on the way: this tile's partial tiles of step s (after the last step a
    // harmless unused read)
    tail(xc, std::true_type{}, s, dh_rec, s);
  };
  using T0 = std::integral_constant<int, 0>;
  int s = p.s_begin;
  if (s == 0) {
    // step 0: no recurrent gradient yet, nothing to gather
    tail(T0{}, std::false_type{}, 0, 0.f, 0);
    s = 1;
  }
  // (first phase peeled so that every gather the loop waits for was issued by the same
  // code sequence)
  prof.init((p.dbg & 32) && cw == 0 && unit == p.chain_begin);
  if (s < s_end) {
    issue(0, s - 1);
    for (; s < s_end; ++s) phase(T0{}, s);
  }
  prof.flush(p.status, w);
#pragma unroll
  for (int x = 0; x < NT; ++x)
    p.dc_state[((size_t)dir * p.n_pad + cn[x]) * H + cu] = dc[x];
  if (p.db_part) {
#pragma unroll
    for (int x = 0; x < NT; ++x)
      tile_gate_sums(gsum[x], lds, p.db_part + ((size_t)(bt0 + x) * 2 + dir) * H4 + 64 * cw,
                     p.s_begin > 0);
  }
  if (p.dz_absmax) {
    zmax = asr_wave_max(zmax);
    if (lane == 0 && zmax > 0.f) atomicMax(p.dz_absmax, __float_as_uint(zmax));
  }
}

template <int TPW>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel_x(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit_local, cw;
  if (!map_block(p, unit_local, cw)) return;
  const int unit = p.chain_begin + unit_local;
  const bool fast = chain_on_one_xcd(p, unit, cw, reinterpret_cast<int*>(lds));
  if (fast) bwd_body_x<TPW, true>(p, unit, cw, lds);
  else bwd_body_x<TPW, false>(p, unit, cw, lds);
}

// ---------------------------------------------------------------------------
// backward, split-fp16, fourth generation: TWO-DIMENSIONAL split of dh = dz @ U^T (plain cell,
// H = 256 / 512, persistent and stepwise mode).
//
// bwd_body_x splits the 4H-long reduction over all P = H/16 workgroups of a chain: every
// workgroup publishes a partial dh tile for ALL H outputs (16 x H words: 32 KB at H = 512) and
// every consumer adds P partials.  The exchange volume of a chain-step is P^2 KB -- 1 MB at
// H = 512, 8 GB per layer, all of it written through to HBM (PMC: 8.65 GB of WRITE_SIZE per
// launch, 3.9 x the algorithmic bytes of the kernel; the write stream, not the MFMAs or the
// hand-off latency, set the 2.85 us step against the forward kernel's 1.89).
//
// Here workgroup (a, b), a < PA = H/64, b < 4, owns the reduction slice KA = gate columns of the
// 64 units [64 a, 64 a + 64) AND the output block OB = units [OT*64 b, OT*64 (b + 1)), OT =
// H/256: it multiplies dz[:, KA] (16 x 256) with U[OB, KA]^T and publishes the partial dh of ITS
// block only (16 x 64 OT words: 8 KB at H = 512).  The published volume is PA partial sums
// instead of P (4 x less: 2 MB per step at cfg3); the price is that the four workgroups (a, 0..3)
// each need dz[:, KA], i.e. each runs the gate-gradient arithmetic of the same 64 units (four
// (sample, unit) pairs per thread instead of one).  Everything of that arithmetic that does not
// depend on the recurrent gradient (tanh(c), the activation slopes) is computed BEFORE the
// step's gather is awaited, so it overlaps the hand-off.
//  * thread (n = tid >> 4, q = tid & 15) owns sample n, units 64 a + 4 q .. + 3; a gathered
//    16-byte group IS its four recurrent gradients (no cross-lane reduction);
//  * wave w multiplies output tiles w + 4 i (i < OT) over the 8 K-steps of the slice; the dz tile
//    (hi / lo halfs, 16 x 256, per-sample power-of-two scale) lives in LDS, B fragments in VGPRs,
//    the U^T fragments in 64 OT AGPRs for the whole sequence;
//  * a workgroup's progress depends on its peers only through TWO hops (its producers'
//    producers are all workgroups), so it can be two steps ahead of a consumer: FOUR exchange
//    slots (step & 3) and the tag = bit 2 of the absolute step;
//  * the dz slab row of sample n is written by workgroup b = n & 3, the bias-gradient partials of
//    unit 4 q + b by workgroup b, dc_state / max|dz| by b = 0.
// Arithmetic per (sample, unit): the same products as bwd_body_x, summed in a different order
// (PA partials of 256 columns instead of P of 64), per-sample scale over 256 columns.
__device__ __forceinline__ void mfma3_first(f32x4& am, f32x4& a1, f32x4& a2, const f32x4& uh,
                                            const f32x4& ul, const h8& bh, const h8& bl) {
  asm volatile(
      "v_mfma_f32_16x16x32_f16 %0, %3, %5, 0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %3, %6, 0\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %4, %5, 0"
      : "=&v"(am), "=&v"(a1), "=&v"(a2)
      : "a"(uh), "a"(ul), "v"(bh), "v"(bl));
}
__device__ __forceinline__ void mfma3_acc(f32x4& am, f32x4& a1, f32x4& a2, const f32x4& uh,
                                          const f32x4& ul, const h8& bh, const h8& bl) {
  asm volatile(
      "v_mfma_f32_16x16x32_f16 %0, %3, %5, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %3, %6, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %4, %5, %2"
      : "+v"(am), "+v"(a1), "+v"(a2)
      : "a"(uh), "a"(ul), "v"(bh), "v"(bl));
}
// an MFMA's D needs its pass count + 4 wait states before a VALU may read it
__device__ __forceinline__ void mfma_settle(f32x4& am, f32x4& a1, f32x4& a2) {
  asm volatile("s_nop 13" : "+v"(am), "+v"(a1), "+v"(a2));
}

template <int OT, bool FAST, bool EXACT>
__device__ __forceinline__ void bwd_body_c(const LstmParams& p, int unit, int cw, float* lds) {
  constexpr int PA = 4 * OT;                       // reduction slices (H / 64)
  constexpr int KS = 8;                            // K-steps of 32 columns per slice
  constexpr int NTILE = 4 * OT;                    // output tiles of a workgroup
  constexpr int DZS = 264;                         // LDS row stride of the dz tile (halfs)
  constexpr int DZF = 260;                         // EXACT: row stride of the fp32 dz tile (floats)
  constexpr int kBufFloats = EXACT ? 16 * DZF : 16 + (2 * 16 * DZS) / 2;   // fp32 tile | sinv + hi + lo
  constexpr int kSlotWords = 4 * NTILE * PA * 256;            // one exchange slot of a chain
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int a = cw % PA, b = cw / PA;
  const int dir = unit / p.NB, bt = unit % p.NB;

  // EXACT: products on v_mfma_f32_16x16x4_f32.  MFMA m = (c4, e) of an output tile takes from
  // lane (g, nl) the fp32 word e of its 16-byte LDS read c4 (columns 16 c4 + 4 g .. + 3 of sample
  // nl), i.e. k-index g <-> column 16 c4 + 4 g + e of the slice: the dz tile stays plain fp32 in
  // LDS (no per-sample scale, no split), the U^T fragments are one fp32 register per MFMA.
  constexpr int NM = EXACT ? 64 : 1;               // fp32 MFMAs per output tile
  float uf[OT][NM];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      const int orow = 64 * OT * b + 16 * (w + 4 * i) + nl;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int j = 256 * a + 16 * (m >> 2) + 4 * g + (m & 3);
        uf[i][m] = p.U[((size_t)(dir * H + orow)) * H4 + j];
        asm volatile("" : "+a"(uf[i][m]));         // AGPR-class from here on
      }
    }
  }
  f32x4 ufh[OT][KS], ufl[OT][KS];                  // bit patterns of 8 halfs each (AGPRs)
#pragma unroll
  for (int i = 0; i < OT; ++i) {
    if constexpr (EXACT) break;
    const int orow = 64 * OT * b + 16 * (w + 4 * i) + nl;     // output unit of this lane's A row
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      h8 hv, lv;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 256 * a + 32 * kk + 8 * g + e;
        _Float16 hi, lo;
        split_f16(p.U[((size_t)(dir * H + orow)) * H4 + j], hi, lo);
        hv[e] = hi; lv[e] = lo;
      }
      ufh[i][kk] = __builtin_bit_cast(f32x4, hv);
      ufl[i][kk] = __builtin_bit_cast(f32x4, lv);
      asm volatile("" : "+a"(ufh[i][kk]), "+a"(ufl[i][kk]));   // AGPR-class from here on
    }
  }
  const int n = tid >> 4, q = tid & 15;
  const int cn = bt * 16 + n;                      // slab row (sample) of this thread
  const int u0 = 64 * a + 4 * q;                   // its first unit
  const int s_end = p.s_begin + p.s_count;
  unsigned* xch = p.xbuf + (size_t)(dir * p.NB + bt) * p.xchain_words;
  f32x4 cmask = {1.f, 1.f, 1.f, 1.f};
  if (p.mask_u)
    cmask = *reinterpret_cast<const f32x4*>(p.mask_u + ((size_t)dir * p.n_pad + cn) * H + u0);
  f32x4 dc = {0.f, 0.f, 0.f, 0.f};
  if (p.s_begin > 0)
    dc = *reinterpret_cast<const f32x4*>(p.dc_state + ((size_t)dir * p.n_pad + cn) * H + u0);
  // (waited for HERE, once: left pending, the first use inside the loop would be a vmcnt(0) in
  // every iteration -- the compiler cannot know on which entry path they have landed)
  asm volatile("" : "+v"(cmask), "+v"(dc));
  float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);   // bias-gradient partials of unit u0 + b
  float zmax = 0.f;
  bool dead = false;
  StepProf prof;
  prof.init(false);

  // Slab values of the next TWO steps (sets A / B, used alternately, so that no register holding
  // a value still in flight is ever copied -- a copy is waited for on the spot): a step's values
  // are loaded two steps ahead, into the set the loading step has just consumed.
  struct Slabs { f32x4 dy, c, cp, g[4]; float hp; };    // hp: 1 if the step has a previous frame
  Slabs SA, SB;
  // Slab accesses as buffer operations: the frame part of an address is wave-uniform (scalar
  // arithmetic beside the VALU stream, folded into the resource's base), the (sample, unit) part
  // is a per-lane constant -- no vector address arithmetic in the step.
  const unsigned vo_dy = (unsigned)(((size_t)cn * H2 + dir * H + u0) * 4);
  const unsigned vo_c = (unsigned)((((size_t)cn * 2 + dir) * H + u0) * 4);
  const unsigned vo_g = (unsigned)((((size_t)cn * 2 + dir) * H4 + 4 * u0) * 4);
  const unsigned vo_z = (n & 3) == b ? vo_g : 0xC0000000u;   // dz rows: owner lanes only
  // Frame bases of the step being LOADED (ld_*) and of the step being STORED (st_dz) as running
  // pointers: one scalar 64-bit add per slab and step.
  const long long fstep = dir == 0 ? -1 : 1;       // frame increment of a BPTT step
  const size_t fr_dy = (size_t)p.n_pad * H2, fr_g = (size_t)p.n_pad * 2 * H4;
  auto rs = [&](const float* base, size_t frame_floats) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                             (unsigned)(frame_floats * 4), 0x00020000);
  };
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned vo, int imm) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, imm, 0));
  };
  const int tt0 = dir == 0 ? p.T - 1 - p.s_begin : p.s_begin;      // frame of the first step
  const float* ld_dy = p.dy + (size_t)tt0 * fr_dy;
  const float* ld_c = p.cell + (size_t)tt0 * fr_dy;                 // (cell rows are 2 H wide too)
  const float* ld_g = p.gates + (size_t)tt0 * fr_g;
  float* st_dz = p.dz + (size_t)tt0 * fr_g;
  // loads the slab values of step ss (the frame the ld_* bases point at), then advances them
  auto load_slabs = [&](int ss, Slabs& S) {
    const bool has_prev = ss + 1 < p.T;            // the sequence's first frame has c_prev = 0
    S.dy = ld4(rs(ld_dy, fr_dy), vo_dy, 0);
    S.c = ld4(rs(ld_c, fr_dy), vo_c, 0);
    // (no select on the fresh load: a step without a previous frame reads a valid row and
    // multiplies it by hp = 0 when the value is USED)
    S.cp = ld4(rs(has_prev ? ld_c + fstep * (long long)fr_dy : ld_c, fr_dy), vo_c, 0);
    S.hp = has_prev ? 1.f : 0.f;
    const __amdgpu_buffer_rsrc_t rg = rs(ld_g, fr_g);
#pragma unroll
    for (int j = 0; j < 4; ++j) S.g[j] = ld4(rg, vo_g, 16 * j);
    if (ss + 1 < s_end) {                          // (past the launch's end: stay on a valid frame)
      ld_dy += fstep * (long long)fr_dy;
      ld_c += fstep * (long long)fr_dy;
      ld_g += fstep * (long long)fr_g;
    }
  };

  // gather: the 16-byte group (sample n, units 4 q ..) of the partial tiles of the PA producers
  // (a', a / OT), 1 KB apart (immediates of one offset register)
  constexpr int NL = PA;
  const unsigned goff = (unsigned)(((((a / OT) * NTILE + 4 * (a % OT) + (q >> 2)) * PA) * 256 +
                                    n * 16 + (q & 3) * 4) * 4);
  u32x4 v[NL];
  auto rslot = [&](int ss) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(xch + (size_t)(ss & 3) * kSlotWords, 0,
                                             kSlotWords * 4, 0x00020000);
  };
  auto load_groups = [&](const __amdgpu_buffer_rsrc_t& rsrc) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff, i * 1024, FAST ? kNt : kSc1);
  };
  auto issue = [&](int ss) {
    for (int i = 0; i < p.prepoll; ++i) __builtin_amdgcn_s_sleep(1);
    load_groups(rslot(ss));
  };
  auto await = [&](int ss) {
    const unsigned flip = 0u - ((unsigned)(ss >> 2) & 1u);
    bool stale = !all_tagged<NL>(v, flip);
    if (__builtin_amdgcn_ballot_w64(stale) == 0ull) return;
    if (!p.poll || dead) return;
    const __amdgpu_buffer_rsrc_t rsrc = rslot(ss);
    const long long t0 = wall_clock64();
    bool gave_up = false;
    while (stale) {
#ifdef POLLCOUNT
      if (prof.on) prof.pt[0] += 1000000;
#endif
      for (int i = 0; i < p.repoll; ++i) __builtin_amdgcn_s_sleep(1);
      load_groups(rsrc);
      stale = !all_tagged<NL>(v, flip);
      if (stale && wall_clock64() - t0 > p.spin) { gave_up = true; break; }
    }
    if (__builtin_amdgcn_ballot_w64(gave_up) != 0ull) {
      dead = true;
      if (gave_up) mark_timeout(p.status);
    }
  };

  // factors of one step that do not depend on the recurrent gradient (computed while the
  // gather is in flight): dz_o = dh A_o ; dcc = dc + dh B ; dz_{i,f,g} = dcc C_{i,f,g} ; dc' = dcc gf
  struct Pre { f32x4 dy, Ao, B, Ci, Cf, Cg, gf; };
  auto precompute = [&](Slabs& S) -> Pre {
#pragma clang fp contract(off)
    // The loaded registers pass through here untouched until NOW: whatever regrouping of them
    // the compiler wants (operand pairs of packed instructions) happens behind this point, a
    // whole step after the loads, not right behind them (where it would be waited for).
    asm volatile("" : "+v"(S.dy), "+v"(S.c), "+v"(S.cp), "+v"(S.g[0]), "+v"(S.g[1]), "+v"(S.g[2]),
                 "+v"(S.g[3]));
    Pre r;
    r.dy = S.dy;
    const float cc[4] = {S.c[0], S.c[1], S.c[2], S.c[3]};
    const float cp[4] = {S.cp[0] * S.hp, S.cp[1] * S.hp, S.cp[2] * S.hp, S.cp[3] * S.hp};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gi = S.g[j][0], gf = S.g[j][1], gg = S.g[j][2], go = S.g[j][3];
      const float tch = fast_tanh_rcp(cc[j]);
      // (explicit FMAs, contraction off: the two copies of a phase in the unrolled loop must
      // round identically, or a step's result would depend on which of them processed it)
      r.Ao[j] = tch * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
      r.B[j] = go * __builtin_fmaf(-tch, tch, 1.f);
      r.Ci[j] = gg * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
      r.Cf[j] = cp[j] * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
      r.Cg[j] = gi * __builtin_fmaf(-gg, gg, 1.f);
      r.gf[j] = gf;
    }
    // keep all of it AHEAD of the await (the compiler would sink it to its uses behind the
    // polling loop, i.e. onto the critical path)
    asm volatile("" : "+v"(r.dy), "+v"(r.Ao), "+v"(r.B), "+v"(r.Ci), "+v"(r.Cf), "+v"(r.Cg),
                 "+v"(r.gf));
    return r;
  };

  // everything of one step after its recurrent gradient is known: gate gradients of the four
  // units, dz slab + LDS tile, barrier, partial dh tiles of this block, publish, next gather
  auto tail = [&](int s, const Pre& pre, const float4& dh_rec, bool do_issue, Slabs& S) {
#pragma clang fp contract(off)
    float* sinv = lds + (size_t)(s & 1) * kBufFloats;         // [16] 1 / scale
    _Float16* dzh = reinterpret_cast<_Float16*>(sinv + 16);   // [16][DZS] hi
    _Float16* dzl = dzh + 16 * DZS;                           // [16][DZS] lo
    float z[4][4];
    {
      const float dr[4] = {dh_rec.x, dh_rec.y, dh_rec.z, dh_rec.w};
      const float cm[4] = {cmask[0], cmask[1], cmask[2], cmask[3]};
      float dcv[4] = {dc[0], dc[1], dc[2], dc[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dh = __builtin_fmaf(cm[j], dr[j], pre.dy[j]);
        const float dcc = __builtin_fmaf(dh, pre.B[j], dcv[j]);
        z[j][0] = dcc * pre.Ci[j];
        z[j][1] = dcc * pre.Cf[j];
        z[j][2] = dcc * pre.Cg[j];
        z[j][3] = dh * pre.Ao[j];
        dcv[j] = dcc * pre.gf[j];
      }
      dc = f32x4{dcv[0], dcv[1], dcv[2], dcv[3]};
    }
    if constexpr (EXACT) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(z[j][0]), fabsf(z[j][1])), fmaxf(fabsf(z[j][2]), fabsf(z[j][3]))));
      zmax = fmaxf(zmax, m);
      float* row = lds + (size_t)(s & 1) * kBufFloats + n * DZF + 16 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(row + 4 * j) = f32x4{z[j][0], z[j][1], z[j][2], z[j][3]};
    } else {
    // power-of-two scale of this sample's 256 columns: max over its 16 threads (one DPP row)
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(z[j][0]), fabsf(z[j][1])), fmaxf(fabsf(z[j][2]), fabsf(z[j][3]))));
    zmax = fmaxf(zmax, m);
    m = row16_max(m);
    int ex = 0;
    if (m > 0.f) (void)frexpf(m, &ex); else ex = 9;
    ex = ex < -100 ? -100 : ex;                    // keep 2^(9-ex) finite for denormal maxima
    const float sc = ldexpf(1.f, 9 - ex);
    if (q == 0) sinv[n] = ldexpf(1.f, ex - 9);
    {
      h8 hi8[2], lo8[2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 x, y;
          split_f16(z[j][e] * sc, x, y);
          hi8[j >> 1][4 * (j & 1) + e] = x;
          lo8[j >> 1][4 * (j & 1) + e] = y;
        }
      // column c = 16 q + 8 half + e of the tile row lives at half 128 half + 8 q + e: the 8
      // lanes of a ds_write_b128 group then cover 128 contiguous bytes, and the fragment reads
      // below (lane = sample nl, chunk g) hit 16 different 16-byte bank groups (the
      // column-major order 16 q + 8 half had both two-way conflicted: 49 % of the LDS cycles)
      _Float16* rh = dzh + n * DZS + 8 * q;
      _Float16* rl = dzl + n * DZS + 8 * q;
      *reinterpret_cast<h8*>(rh) = hi8[0];
      *reinterpret_cast<h8*>(rh + 128) = hi8[1];
      *reinterpret_cast<h8*>(rl) = lo8[0];
      *reinterpret_cast<h8*>(rl + 128) = lo8[1];
    }
    }
    // (b is uniform: scalar branches, no indexed access)
    if (b == 0) { gsum.x += z[0][0]; gsum.y += z[0][1]; gsum.z += z[0][2]; gsum.w += z[0][3]; }
    else if (b == 1) { gsum.x += z[1][0]; gsum.y += z[1][1]; gsum.z += z[1][2]; gsum.w += z[1][3]; }
    else if (b == 2) { gsum.x += z[2][0]; gsum.y += z[2][1]; gsum.z += z[2][2]; gsum.w += z[2][3]; }
    else { gsum.x += z[3][0]; gsum.y += z[3][1]; gsum.z += z[3][2]; gsum.w += z[3][3]; }
    // off the dependent path (the other waves are still on their way to the barrier), and ahead
    // of the publish and the gather in the CU's in-order memory queue only by a whole MFMA phase:
    // this workgroup's rows of the dz slab, and the slab values of step s + 2 into the set this
    // step has just consumed
    {
      // (no branch around them, so that the compiler can count them in its waits: the lanes
      // that do not own the row store beyond the resource's range, which the hardware drops)
      const __amdgpu_buffer_rsrc_t rz = rs(st_dz, fr_g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 zz = {z[j][0], z[j][1], z[j][2], z[j][3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zz), rz, vo_z, 16 * j, 0);
      }
    }
    st_dz += fstep * (long long)fr_g;
    load_slabs(s + 2, S);
    prof.stamp(2);
    __syncthreads();
    prof.stamp(3);
    if constexpr (EXACT) {
      f32x4 bq[16];                                // this lane's 16 x 4 columns of sample nl
      const float* trow = lds + (size_t)(s & 1) * kBufFloats + nl * DZF + 4 * g;
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) bq[c4] = *reinterpret_cast<const f32x4*>(trow + 16 * c4);
      const unsigned wtag = (unsigned)(s >> 2) & 1u;
      const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)(s & 3) * kSlotWords, 0, kSlotWords * 4, 0x00020000);
      const unsigned soff = (unsigned)((((b * NTILE + w) * PA + a) * 256 + nl * 16 + 4 * g) * 4);
      f32x4 acc[OT];
      // (the OT tiles' accumulator chains interleaved: 32 cycles of pipe per MFMA, 40 of latency)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float bw = bq[m >> 2][m & 3];
#pragma unroll
        for (int i = 0; i < OT; ++i) {
          if (m == 0)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[i]) : "a"(uf[i][0]), "v"(bw));
          else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(uf[i][m]), "v"(bw));
        }
      }
      if constexpr (OT == 2) asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
      else asm volatile("s_nop 15" : "+v"(acc[0]));
#pragma unroll
      for (int i = 0; i < OT; ++i) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tag_word(acc[i][e], wtag);
        __builtin_amdgcn_raw_buffer_store_b128(o, wr, soff, i * (4 * PA * 1024), FAST ? 0 : kSc1);
      }
    } else
    {
      h8 bh[KS], bl[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        // columns 32 kk + 8 g .. + 7 = (q = 2 kk + (g >> 1), half = g & 1)
        bh[kk] = *reinterpret_cast<const h8*>(dzh + nl * DZS + 128 * (g & 1) + 16 * kk + 8 * (g >> 1));
        bl[kk] = *reinterpret_cast<const h8*>(dzl + nl * DZS + 128 * (g & 1) + 16 * kk + 8 * (g >> 1));
      }
      const float us = sinv[nl];
      const float usl = us * (1.f / kLoScale);
      const unsigned wtag = (unsigned)(s >> 2) & 1u;
      // (the last step's tiles are published too: nobody reads them, and no branch is needed)
      const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
          xch + (size_t)(s & 3) * kSlotWords, 0, kSlotWords * 4, 0x00020000);
      const unsigned soff = (unsigned)((((b * NTILE + w) * PA + a) * 256 + nl * 16 + 4 * g) * 4);
      auto finish = [&](int i, f32x4& am, f32x4& a1, f32x4& a2) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = tag_word(__builtin_fmaf(a1[e] + a2[e], usl, am[e] * us), wtag);
        // output tile w + 4 i of this block: 4 PA KB further on
        __builtin_amdgcn_raw_buffer_store_b128(o, wr, soff, i * (4 * PA * 1024), FAST ? 0 : kSc1);
      };
      f32x4 am[OT], a1[OT], a2[OT];
#pragma unroll
      for (int i = 0; i < OT; ++i) {
        mfma3_first(am[i], a1[i], a2[i], ufh[i][0], ufl[i][0], bh[0], bl[0]);
#pragma unroll
        for (int kk = 1; kk < KS; ++kk) {
          mfma3_acc(am[i], a1[i], a2[i], ufh[i][kk], ufl[i][kk], bh[kk], bl[kk]);
          // the previous tile's results have left the pipe by now: combine and publish them in
          // the issue slots between this tile's MFMAs
          if (i > 0 && kk == KS / 2) {
            // (pins the reads behind this point of the MFMA stream: >= 12 MFMAs after the last
            // write of these accumulators)
            asm volatile("s_nop 3" : "+v"(am[i - 1]), "+v"(a1[i - 1]), "+v"(a2[i - 1]));
            finish(i - 1, am[i - 1], a1[i - 1], a2[i - 1]);
          }
        }
      }
      mfma_settle(am[OT - 1], a1[OT - 1], a2[OT - 1]);
      finish(OT - 1, am[OT - 1], a1[OT - 1], a2[OT - 1]);
    }
    prof.stamp(4);
    if (do_issue) issue(s);
  };

  // one step s >= 1 on slab set S
  auto phase = [&](int s, Slabs& S) {
    prof.stamp(5);
    const Pre pre = precompute(S);                 // overlaps the hand-off
    prof.stamp(0);
    await(s - 1);
    prof.stamp(1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NL; ++i) {                 // (the tag bit stays in: <= 1 ulp)
      acc.x += __uint_as_float(v[i][0]); acc.y += __uint_as_float(v[i][1]);
      acc.z += __uint_as_float(v[i][2]); acc.w += __uint_as_float(v[i][3]);
    }
    tail(s, pre, acc, true, S);                    // (after the last step a harmless unused read)
  };
  // Both ways into the loop leave the vector-memory queue as the loop body does: the gather
  // is the youngest operation, both slab sets are older (the waits the compiler counts for the
  // loop body are the worst case over every path into it).
  int s = p.s_begin;
  if (s == 0) {
    // step 0: no recurrent gradient yet, nothing to gather before it
    load_slabs(0, SA);
    load_slabs(1, SB);
    const Pre pre = precompute(SA);
    tail(0, pre, make_float4(0.f, 0.f, 0.f, 0.f), true, SA);   // (reloads SA with step 2)
    s = 1;
  } else {
    load_slabs(s, SB);
    load_slabs(s + 1, SA);
    issue(s - 1);                                  // continuing a sequence
  }
  prof.init((p.dbg & 32) && cw == 0 && unit == p.chain_begin);
  for (; s + 1 < s_end; s += 2) {
    phase(s, SB);
    phase(s + 1, SA);
  }
  if (s < s_end) phase(s, SB);
  prof.flush(p.status, w);
  if (b == 0)
    *reinterpret_cast<f32x4*>(p.dc_state + ((size_t)dir * p.n_pad + cn) * H + u0) = dc;
  if (p.db_part) {
    // sum over the 16 samples of every thread's float4 (unit u0 + b), fixed order
    float vs[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // over the wave's four samples (lane >> 4)
      vs[k] += __shfl_xor(vs[k], 16);
      vs[k] += __shfl_xor(vs[k], 32);
    }
    __syncthreads();
    if (lane < 16) *reinterpret_cast<float4*>(lds + (w * 16 + lane) * 4) =
        make_float4(vs[0], vs[1], vs[2], vs[3]);
    __syncthreads();
    if (tid < 64) {                                // (unit quad tid >> 2, gate tid & 3)
      const float tsum = ((lds[tid] + lds[64 + tid]) + lds[128 + tid]) + lds[192 + tid];
      float* dst = p.db_part + ((size_t)bt * 2 + dir) * H4 + 256 * a + 16 * (tid >> 2) + 4 * b +
                   (tid & 3);
      *dst = (p.s_begin > 0 ? *dst : 0.f) + tsum;
    }
    __syncthreads();
  }
  if (p.dz_absmax && b == 0) {
    zmax = asr_wave_max(zmax);
    if (lane == 0 && zmax > 0.f) atomicMax(p.dz_absmax, __float_as_uint(zmax));
  }
}

template <int OT, bool EXACT>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel_c(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit_local, cw;
  if (!map_block(p, unit_local, cw)) return;
  const int unit = p.chain_begin + unit_local;
  const bool fast = chain_on_one_xcd(p, unit, cw, reinterpret_cast<int*>(lds));
  if (fast) bwd_body_c<OT, true, EXACT>(p, unit, cw, lds);
  else bwd_body_c<OT, false, EXACT>(p, unit, cw, lds);
}

// ---------------------------------------------------------------------------
struct Plan {
  int R, P, TPW, NKK, prec;
  int n1;                  // 1: single-utterance forward kernel (2 chains = 2 directions)
  int form_c;              // 1: BPTT with the two-dimensional split (lstm_bwd_kernel_c)
  size_t shm;
  size_t xchain_words;
  int chains_per_launch;
};

int up4(int x) { return (x + 3) & ~3; }

typedef void (*kern_t)(LstmParams);

kern_t pick_fwd_h(int nkk) {
  switch (nkk) {
    case 4: return lstm_fwd_kernel_h<4>;
    case 8: return lstm_fwd_kernel_h<8>;
    default: return lstm_fwd_kernel_h<16>;
  }
}
kern_t pick_fwd_hv(int nkk) {
  switch (nkk) {
    case 4: return lstm_fwd_kernel_hv<4>;
    case 8: return lstm_fwd_kernel_hv<8>;
    default: return lstm_fwd_kernel_hv<16>;
  }
}
kern_t pick_bwd_hv(int tpw) {
  switch (tpw) {
    case 1: return lstm_bwd_kernel_hv<1>;
    case 2: return lstm_bwd_kernel_hv<2>;
    case 4: return lstm_bwd_kernel_hv<4>;
    default: return lstm_bwd_kernel_hv<8>;
  }
}
kern_t pick_bwd_h(int tpw) {
  switch (tpw) {
    case 1: return lstm_bwd_kernel_h<1>;
    case 2: return lstm_bwd_kernel_h<2>;
    case 4: return lstm_bwd_kernel_h<4>;
    default: return lstm_bwd_kernel_h<8>;
  }
}
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

int fwd_xstride() {
  int v = env_int("ASR_LSTM_XSTRIDE", 256);
  if (v < 256 || (v & 255)) v = 256;
  return v;
}

int make_plan(const asr_lstm_args* a, bool bwd, Plan* out, kern_t* kout) {
  const int H = a->H;
  const int chains = 2 * (a->n_pad / 16);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ASR_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ASR_ERR_LAUNCH;
  const int num_cu = prop.multiProcessorCount;
  Plan pl;
  kern_t k;
  pl.P = (H + 15) / 16;
  // recurrent-product arithmetic: 1 = split-fp16 MFMA (22-bit mantissa, default),
  // 0 = exact fp32 MFMA
  pl.prec = env_int("ASR_LSTM_PREC", 1) ? 1 : 0;
  pl.NKK = 0;
  pl.n1 = 0;
  pl.form_c = 0;
  if (!bwd && a->n_valid == 1 && a->mode == 0 && (H == 256 || H == 512) &&
      !(a->mi || a->zone_c || a->zone_h || a->uh) && env_int("ASR_LSTM_N1", 1)) {
    // one utterance: the tile-free exact-fp32 kernel (fwd_body_n1); 2 chains = 2 directions
    pl.R = 0; pl.TPW = 0; pl.NKK = 0; pl.n1 = 1;
    pl.shm = (size_t)(H + 256) * 4;
    pl.xchain_words = (size_t)2 * H;
    k = H == 256 ? lstm_fwd_kernel_n1<64> : lstm_fwd_kernel_n1<128>;
  } else if (!bwd) {
    pl.R = up4((H + 3) / 4);
    if (pl.R > 128) {
      asr_set_error("lstm fwd: H=%d too large for the register-resident U slice (max 512)", H);
      return ASR_ERR_INVALID;
    }
    pl.TPW = 0;
    k = nullptr;
    if (pl.prec == 0) {
      // exact fp32: the structure of the split-fp16 kernel on v_mfma_f32_16x16x4_f32
      // (fwd_body_x<.., EXACT>) -- built for the widths it is benchmarked and compared at
      if (!(a->mode == 0 && (H == 256 || H == 512) && !(a->mi || a->zone_c || a->zone_h || a->uh))) {
        asr_set_error("lstm fwd: ASR_LSTM_PREC=0 (exact fp32 MFMA) exists for the plain cell at "
                      "H = 256 / 512 in persistent mode; H=%d mode=%d", H, a->mode);
        return ASR_ERR_INVALID;
      }
      pl.xchain_words = (size_t)2 * (H / 4) * (size_t)(fwd_xstride() / 4);
      pl.shm = (size_t)2 * 4 * 4 * 64 * 16;
      k = H == 256 ? lstm_fwd_kernel_x<2, true> : lstm_fwd_kernel_x<4, true>;
    }
    if (pl.prec == 1) {
      pl.xchain_words = (size_t)2 * (H / 4) * (size_t)(fwd_xstride() / 4);
      const int nkk = (H + 31) / 32;
      pl.NKK = nkk <= 4 ? 4 : nkk <= 8 ? 8 : 16;
      const bool variants = a->mi || a->zone_c || a->zone_h || a->uh;
      // ASR_LSTM_GENERIC=1: the any-H kernels (lstm_*_kernel_h) also where the specialised
      // ones apply (tests compare the two)
      const bool generic = env_int("ASR_LSTM_GENERIC", 0) != 0;
      if (!variants && !generic && a->mode == 0 && (H == 256 || H == 512)) {
        // plain cell, persistent mode, H = 128 NKW: K split over the waves, U fragments in
        // AGPRs (fwd_body_x)
        pl.shm = (size_t)2 * 4 * 4 * 64 * 16;
        k = H == 256 ? lstm_fwd_kernel_x<2, false> : lstm_fwd_kernel_x<4, false>;
      } else {
        // any H, the cell variants, stepwise mode: h staged in LDS once per step (fwd_body_h)
        pl.shm = (size_t)4 * 16 * (32 * pl.NKK + 8) * 2;
        k = variants ? pick_fwd_hv(pl.NKK) : pick_fwd_h(pl.NKK);
      }
    }
  } else {
    pl.R = 16;
    const int tpw = (pl.P + 3) / 4;
    if (tpw > 8) {
      asr_set_error("lstm bwd: H=%d too large (max 512)", H);
      return ASR_ERR_INVALID;
    }
    pl.TPW = tpw <= 1 ? 1 : tpw <= 2 ? 2 : tpw <= 4 ? 4 : 8;
    pl.xchain_words = (size_t)2 * pl.P * pl.P * 256;
    k = nullptr;
    if (pl.prec == 0) {
      // exact fp32: the two-dimensional split on fp32 MFMAs (bwd_body_c<.., EXACT>)
      if (!(a->mode == 0 && (H == 256 || H == 512) && !(a->mi || a->zone_c || a->zone_h))) {
        asr_set_error("lstm bwd: ASR_LSTM_PREC=0 (exact fp32 MFMA) exists for the plain cell at "
                      "H = 256 / 512 in persistent mode; H=%d mode=%d", H, a->mode);
        return ASR_ERR_INVALID;
      }
      pl.form_c = 1;
      pl.shm = (size_t)2 * 16 * 260 * 4;
      pl.xchain_words = (size_t)4 * 4 * (H / 64) * (H / 64) * 256;
      k = H == 256 ? lstm_bwd_kernel_c<1, true> : lstm_bwd_kernel_c<2, true>;
    }
    if (pl.prec == 1) {
      pl.shm = 2 * ((size_t)16 * 4 + (size_t)2 * 16 * 72 * 2);
      const bool variants = a->mi || a->zone_c || a->zone_h;
      const bool generic = env_int("ASR_LSTM_GENERIC", 0) != 0;
      const bool wide = !variants && !generic && a->mode == 0 && (H == 256 || H == 512);
      // ASR_LSTM_BWD_2D: 1 = the two-dimensional split (bwd_body_c), 0 = bwd_body_x; default:
      // from H = 512 on, where the partial-tile exchange of bwd_body_x is 1 MB per chain-step
      // (measured at H = 256: 1.89 us per step against 1.60)
      const bool form_c = wide && env_int("ASR_LSTM_BWD_2D", H >= 512 ? 1 : 0) != 0;
      if (form_c) {
        pl.form_c = 1;
        pl.shm = 2 * ((size_t)16 * 4 + (size_t)2 * 16 * 264 * 2);
        pl.xchain_words = (size_t)4 * 4 * (H / 64) * (H / 64) * 256;
        k = H == 256 ? lstm_bwd_kernel_c<1, false> : lstm_bwd_kernel_c<2, false>;
      } else if (wide) {
        k = H == 256 ? lstm_bwd_kernel_x<4> : lstm_bwd_kernel_x<8>;
      } else {
        k = variants ? pick_bwd_hv(pl.TPW) : pick_bwd_h(pl.TPW);
      }
    }
  }
  if (pl.shm < (size_t)pl.P * 4 + 16) pl.shm = (size_t)pl.P * 4 + 16;
  // Own the CU: a latency-bound workgroup must not share its SIMDs / LDS pipe with the
  // GEMM workgroups (80 KB LDS each) that the host overlaps on a second stream, so it
  // reserves enough LDS that none of those fits beside it (ASR_LSTM_EXCL=0 disables).
  // (asr_lstm_args.lds_reserve_kb = 80 lets exactly TWO recurrent workgroups share a CU and
  // still keeps every 80 KB GEMM workgroup out: the caller then confines the launch to half
  // of the CUs with a CU-masked stream and leaves the other half to the GEMM streams)
  static const int excl = env_int("ASR_LSTM_EXCL", 1);
  const size_t reserve = (size_t)(a->lds_reserve_kb > 0 ? a->lds_reserve_kb : 96) * 1024;
  if (excl && pl.shm < reserve) pl.shm = reserve;
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
  // Residency: count ONE workgroup per CU (the occupancy API may over-report, and one
  // wave per SIMD is what the design wants anyway).
  const long cap = (long)num_cu;
  if (cap < (long)pl.P) {
    asr_set_error("lstm: a chain needs %d co-resident workgroups, device has %d CUs", pl.P,
                  num_cu);
    return ASR_ERR_RESIDENCY;
  }
  long cpl = cap / pl.P;
  if (cpl > chains) cpl = chains;
  pl.chains_per_launch = (int)cpl;
  *out = pl;
  if (kout) *kout = k;
  return ASR_OK;
}

constexpr size_t kStatusBytes = 256;

// workspace preparation of a launch that starts a sequence: 16-byte words [0, n_zero) to 0,
// [n_zero, n_total) to all ones, *absmax (optional) to 0
__global__ void __launch_bounds__(256)
lstm_prepare_kernel(uint4* __restrict__ ws, long long n_zero, long long n_total,
                    unsigned* __restrict__ absmax) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u), f = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_total;
       i += (long long)gridDim.x * 256)
    ws[i] = i < n_zero ? z : f;
  if (absmax && blockIdx.x == 0 && threadIdx.x == 0) *absmax = 0u;
}
constexpr size_t kStickyBytes = kStickyInts * sizeof(int);

size_t xbuf_bytes(const asr_lstm_args* a, bool bwd) {
  const size_t chains = (size_t)2 * (a->n_pad / 16);
  const size_t P = (a->H + 15) / 16;
  const size_t words = bwd ? (size_t)2 * P * P * 256
                           : (size_t)2 * (a->H / 4) * (size_t)(fwd_xstride() / 4);
  return asr_align_up(chains * words * 4, 256);
}
size_t xcc_bytes(const asr_lstm_args* a) {
  const size_t chains = (size_t)2 * (a->n_pad / 16);
  const size_t P = (a->H + 15) / 16;
  return asr_align_up(chains * P * sizeof(int), 256);
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
  kern_t k;
  const int rc = make_plan(a, bwd, &pl, &k);
  if (rc != ASR_OK) return rc;
  char* ws = reinterpret_cast<char*>(workspace) + kStickyBytes;    // sticky block first
  const size_t xb = xbuf_bytes(a, bwd);
  const size_t cb_ = xcc_bytes(a);
  LstmParams p;
  p.T = a->T; p.n_pad = a->n_pad; p.H = a->H; p.NB = a->n_pad / 16;
  p.R = pl.R; p.P = pl.P;
  p.U = a->U; p.mask_u = a->mask_u; p.zx = a->zx; p.y = a->y; p.cell = a->cell;
  p.gates = a->gates; p.dy = a->dy; p.dz = a->dz;
  p.dz_absmax = bwd ? reinterpret_cast<unsigned*>(a->dz_absmax) : nullptr;
  p.mi = a->mi; p.uh = a->uh; p.zone_c = a->zone_c; p.zone_h = a->zone_h;
  p.wx = a->wx; p.dwx = a->dwx; p.dmi = a->dmi;
  p.db_part = bwd ? a->db_part : nullptr;
  if (a->mi) {
    ASR_CHECK_ARG(a->uh, "lstm: mi needs the uh slab");
    if (bwd) ASR_CHECK_ARG(a->wx && a->dwx && a->dmi, "lstm bwd: mi needs wx, dwx and dmi");
    ASR_CHECK_ARG(env_int("ASR_LSTM_PREC", 1) == 1 && a->mode == 0,
                  "lstm: the cell variants run on the split-fp16 persistent kernels only");
  }
  if (a->zone_c || a->zone_h)
    ASR_CHECK_ARG(env_int("ASR_LSTM_PREC", 1) == 1 && a->mode == 0,
                  "lstm: the cell variants run on the split-fp16 persistent kernels only");
  p.status = reinterpret_cast<int*>(ws);
  p.xcc = reinterpret_cast<int*>(ws + kStatusBytes);
  p.xbuf = reinterpret_cast<unsigned*>(ws + kStatusBytes + cb_);
  p.xchain_words = (long long)pl.xchain_words;
  p.dc_state = reinterpret_cast<float*>(ws + kStatusBytes + cb_ + xb);
  // step range of this call: the whole sequence, or a slice that continues a previous
  // call on the same workspace (state: exchange slots, cell slab / dc_state)
  int r_begin = 0, r_end = a->T;
  if (a->step_count > 0) {
    ASR_CHECK_ARG(a->step_begin >= 0 && a->step_begin + a->step_count <= a->T,
                  "lstm: step range [%d, +%d) outside T=%d", a->step_begin, a->step_count, a->T);
    r_begin = a->step_begin;
    r_end = a->step_begin + a->step_count;
  }
  if (r_begin == 0) {
    // status words + XCC table to 0, exchange slots to 0xFF, the gate-gradient maximum to 0:
    // ONE launch (three hipMemsetAsync were three fill kernels with a launch gap each, 30 per
    // training step)
    const size_t zero_b = kStatusBytes + cb_;
    const int blocks = (int)((zero_b + xb) / 16 / 256 + 1) < 512 ? (int)((zero_b + xb) / 16 / 256 + 1) : 512;
    hipLaunchKernelGGL(lstm_prepare_kernel, dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<uint4*>(ws), (long long)(zero_b / 16),
                       (long long)((zero_b + xb) / 16), reinterpret_cast<unsigned*>(p.dz_absmax));
    ASR_CHECK_LAUNCH();
  }
  const int chains = pl.n1 ? 2 : 2 * p.NB;
  const bool stepwise = a->mode == 1;
  p.poll = stepwise ? 0 : 1;
  p.allow_fast = env_int("ASR_LSTM_FAST", 1);
  p.dbg = env_int("ASR_LSTM_DBG", 0);
  // measured optimum on MI355X (tools/sweep_poll.sh, tools/sweep_r2c.sh): forward 8-16 naps
  // (~0.4 us; flat in that range), BPTT 4
  p.prepoll = bwd ? env_int("ASR_LSTM_PREPOLL_B", pl.form_c ? 2 : 4)
                  : env_int("ASR_LSTM_PREPOLL_F", pl.P <= 16 ? 12 : 16);
  p.repoll = bwd ? env_int("ASR_LSTM_REPOLL_B", 1) : env_int("ASR_LSTM_REPOLL_F", 1);
  p.xstride = fwd_xstride();
  // ASR_LSTM_SPIN_MS: bound of a persistent kernel's spins in milliseconds (default 600)
  p.spin = (long long)env_int("ASR_LSTM_SPIN_MS", 600) * 100000LL;
  const int steps_per_launch = stepwise ? 1 : (r_end - r_begin);
  for (int s0 = r_begin; s0 < r_end; s0 += steps_per_launch) {
    // the XCC table is rebuilt by every persistent launch (placement may differ)
    if (!stepwise && s0 > 0) ASR_CHECK_HIP(hipMemsetAsync(ws + kStatusBytes, 0, cb_, stream));
    for (int cb = 0; cb < chains; cb += pl.chains_per_launch) {
      const int nch = (chains - cb) < pl.chains_per_launch ? (chains - cb) : pl.chains_per_launch;
      p.chain_begin = cb;
      p.nch = nch;
      p.s_begin = s0;
      p.s_count = (r_end - s0) < steps_per_launch ? (r_end - s0) : steps_per_launch;
      const int groups = (nch + 7) / 8;
      hipLaunchKernelGGL(k, dim3(groups * 8 * pl.P), dim3(kThreads), pl.shm, stream, p);
      ASR_CHECK_LAUNCH();
    }
  }
  return ASR_OK;
}

}  // namespace

extern "C" size_t asr_lstm_workspace_bytes(const asr_lstm_args* a, int backward) {
  if (!a || a->n_pad <= 0 || a->H <= 0) return 0;
  return kStickyBytes + kStatusBytes + xcc_bytes(a) + xbuf_bytes(a, backward != 0) +
         asr_align_up((size_t)4 * a->n_pad * a->H * sizeof(float), 256);
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
  int st[kStickyInts + 1];
  ASR_CHECK_HIP(hipMemcpyAsync(st, workspace, sizeof(st), hipMemcpyDeviceToHost, stream));
  ASR_CHECK_HIP(hipStreamSynchronize(stream));
  if (st[0] != 0 || st[kStickyInts] != 0) {
    if (st[0] != 0) {                      // sticky flag: reported once
      ASR_CHECK_HIP(hipMemsetAsync(const_cast<void*>(workspace), 0, sizeof(int), stream));
      ASR_CHECK_HIP(hipStreamSynchronize(stream));
    }
    asr_set_error("lstm: persistent kernel timed out waiting for a peer workgroup");
    return ASR_ERR_TIMEOUT;
  }
  return ASR_OK;
}

extern "C" int asr_lstm_plan(const asr_lstm_args* a, int backward, int* ks, int* r,
                             int* blocks, int* chains_per_launch) {
  Plan pl;
  const int rc = make_plan(a, backward != 0, &pl, nullptr);
  if (rc != ASR_OK) return rc;
  if (ks) *ks = 1;
  if (r) *r = backward ? pl.TPW * 16 : pl.R;
  if (blocks) *blocks = pl.P * pl.chains_per_launch;
  if (chains_per_launch) *chains_per_launch = pl.chains_per_launch;
  return ASR_OK;
}

// Debug (ASR_LSTM_DBG & 32): per-phase ticks of workgroup 0 of the first chain, 4 waves x 6
// phases, accumulated over the steps of the last call (shader clocks in the current kernels:
// 0 pre-gather arithmetic, 1 waiting for the gather, 2 arithmetic behind it up to the
// barrier, 3 barrier, 4 products + publish, 5 issuing the next gather).
extern "C" int asr_lstm_profile(const void* workspace, asr_stream_t stream_, long long* out24) {
  hipStream_t stream = (hipStream_t)stream_;
  ASR_CHECK_HIP(hipMemcpyAsync(out24, reinterpret_cast<const char*>(workspace) + kStickyBytes + 64,
                               24 * sizeof(long long), hipMemcpyDeviceToHost, stream));
  ASR_CHECK_HIP(hipStreamSynchronize(stream));
  return ASR_OK;
}

// Synchronises `stream`; returns how many chains of the last call on this workspace
// used the same-XCD (L2) transport, or a negative asr_status.
extern "C" int asr_lstm_fast_chains(const void* workspace, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int st[2] = {0, 0};
  ASR_CHECK_HIP(hipMemcpyAsync(st, reinterpret_cast<const char*>(workspace) + kStickyBytes,
                               sizeof(st), hipMemcpyDeviceToHost, stream));
  ASR_CHECK_HIP(hipStreamSynchronize(stream));
  return st[1];
}
