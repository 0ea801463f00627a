// lstm_common.h -- what the recurrent LSTM kernels of lstm_fwd.hip / lstm_bwd.hip and their
// host side (lstm.hip: plan, launch, C ABI) share: the kernel parameter block, the hand-off
// primitives (tagged 16-byte groups, bounded polls), the step profiler, the workgroup -> chain
// mapping.  Design notes: the header comment of lstm.hip.
#pragma once
#include "common.h"
#include <type_traits>

namespace asr_lstm {
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
  long long* trace;        // debug (ASR_LSTM_DBG & 128): [block][wave][16 steps][2] ticks of the
  int trace_s0;            // 100 MHz clock at data arrival / publish, steps trace_s0 .. + 15
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
  _Float16* dz_hl;         // BPTT, optional (bwd_body_c): dz as packed planes instead of the fp32
  const float* dz_bound;   //   slab, pre-scaled by asr_pow2_scale(*dz_bound), which is written
  float* dz_scale_out;     //   to *dz_scale_out (asr_lstm_args.dz_hl)
  // optional cell variants (core/layers.py:432-469); all NULL on the default path
  int act;                 // activation id of the cell candidate / output (variant kernels)
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
}  // namespace asr_lstm

namespace {

using asr_lstm::LstmParams;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kThreads = 256;              // 4 waves, one per SIMD
constexpr int kSc1 = 16;                   // buffer-op cache policy: SC1 (agent scope)
constexpr int kNt = 2;                     // buffer-op cache policy: NT (bypass L1)



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
// (the `activation` hyper-parameter of the variant kernels: asr_act_apply / asr_act_slope, common.h)
__device__ __forceinline__ float act_apply(int id, float x) {
  return id == 0 ? fast_tanh(x) : asr_act_apply(id, x);
}
__device__ __forceinline__ float act_slope(int id, float y) { return asr_act_slope(id, y); }
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
__device__ inline bool chain_on_one_xcd(const LstmParams& p, int chain, int wg, int* lds_i) {
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


template <int NL>
__device__ __forceinline__ bool all_tagged(const u32x4 (&v)[NL], unsigned flip) {
  unsigned x = 0u;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    x |= ((v[i][0] ^ flip) | (v[i][1] ^ flip)) | ((v[i][2] ^ flip) | (v[i][3] ^ flip));
  return (x & 1u) == 0u;
}

}  // namespace

// The kernels are picked by the plan (lstm.hip, make_plan) through these host functions; each
// returns a __global__ function of its translation unit as a launchable pointer.
typedef void (*asr_lstm_kern_t)(asr_lstm::LstmParams);
asr_lstm_kern_t asr_lstm_pick_fwd_h(int nkk, bool variants);      // any H <= 512, stepwise mode
asr_lstm_kern_t asr_lstm_pick_fwd_x(int H, bool exact, bool slab = false, bool eight = false);   // plain cell, H = 256 / 512
asr_lstm_kern_t asr_lstm_pick_fwd_n1(int H);                      // one utterance, H = 256 / 512
asr_lstm_kern_t asr_lstm_pick_bwd_h(int tpw, bool variants);
asr_lstm_kern_t asr_lstm_pick_bwd_x(int H);                       // unit split, H = 256 / 512
asr_lstm_kern_t asr_lstm_pick_bwd_c(int H, bool exact, bool compact = false, bool planes = false);   // two-dimensional split
