// Forward recurrent LSTM kernels (design notes: header of lstm.hip; shared pieces:
// lstm_common.h): lstm_fwd_kernel_h / _hv (any H, cell variants, stepwise mode),
// lstm_fwd_kernel_x (plain cell, H = 256 / 512: K split over the four waves, U fragments in
// AGPRs; EXACT = fp32 MFMA), lstm_fwd_kernel_n1 (one utterance).
#include "lstm_common.h"

namespace {

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
      const float gg = VAR ? act_apply(p.act, z2) : fast_tanh(z2);
      const float go = hard_sigmoid(z3);
      float cn = gf * c + gi * gg;
      if (VAR) cn = c + kc * (cn - c);              // zoneout of the cell state (:457-459)
      c = cn;
      float h = go * (VAR ? act_apply(p.act, c) : fast_tanh(c));
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

// r6: ONE K step (32 units) of TWO gate tiles -- the six MFMAs of FwdMfma::run2's kk-th group,
// same order per accumulator -- with, in the issue slots between them (an MFMA holds the pipe 16
// cycles, a v_perm_b32 issues in 4), the UNPACK of the next K step's gathered words (NEXT): the
// 32 v_perm_b32 of a step used to run ahead of the first MFMA, ~160 clocks of the dependent
// chain of a wave that is alone on its SIMD.  FIRST: the accumulators start at 0.  No trailing
// wait states (the caller reads the results >= 6 MFMAs later, or pads itself).
template <bool FIRST, bool NEXT>
__device__ __forceinline__ void fwd_pair_step(f32x4& am0, f32x4& ac0, f32x4& am1, f32x4& ac1,
                                              const f32x4& uh0, const f32x4& ul0, const f32x4& uh1,
                                              const f32x4& ul1, const h8& bh, const h8& bl,
                                              const u32x4& q0, const u32x4& q1, h8& nbh, h8& nbl) {
  if constexpr (NEXT) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    const unsigned selh = 0x07060302u, sell = 0x05040100u;
    if constexpr (FIRST) {
      asm volatile(
          "s_nop 1\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %12, %17, 0\n\t"     // ac0  = uh0 bl
          "v_perm_b32 %4, %19, %18, %26\n\t"
          "v_perm_b32 %5, %21, %20, %26\n\t"
          "v_mfma_f32_16x16x32_f16 %3, %14, %17, 0\n\t"     // ac1  = uh1 bl
          "v_perm_b32 %6, %23, %22, %26\n\t"
          "v_perm_b32 %7, %25, %24, %26\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %12, %16, 0\n\t"     // am0  = uh0 bh
          "v_perm_b32 %8, %19, %18, %27\n\t"
          "v_perm_b32 %9, %21, %20, %27\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %13, %16, %1\n\t"    // ac0 += ul0 bh
          "v_perm_b32 %10, %23, %22, %27\n\t"
          "v_perm_b32 %11, %25, %24, %27\n\t"
          "v_mfma_f32_16x16x32_f16 %3, %15, %16, %3\n\t"    // ac1 += ul1 bh
          "v_mfma_f32_16x16x32_f16 %2, %14, %16, 0"         // am1  = uh1 bh
          : "=&v"(am0), "=&v"(ac0), "=&v"(am1), "=&v"(ac1), "=&v"(h0), "=&v"(h1), "=&v"(h2),
            "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
          : "a"(uh0), "a"(ul0), "a"(uh1), "a"(ul1), "v"(bh), "v"(bl), "v"(q0[0]), "v"(q0[1]),
            "v"(q0[2]), "v"(q0[3]), "v"(q1[0]), "v"(q1[1]), "v"(q1[2]), "v"(q1[3]), "s"(selh),
            "s"(sell));
    } else {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %1, %12, %17, %1\n\t"    // ac0 += uh0 bl
          "v_perm_b32 %4, %19, %18, %26\n\t"
          "v_perm_b32 %5, %21, %20, %26\n\t"
          "v_mfma_f32_16x16x32_f16 %3, %14, %17, %3\n\t"    // ac1 += uh1 bl
          "v_perm_b32 %6, %23, %22, %26\n\t"
          "v_perm_b32 %7, %25, %24, %26\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %12, %16, %0\n\t"    // am0 += uh0 bh
          "v_perm_b32 %8, %19, %18, %27\n\t"
          "v_perm_b32 %9, %21, %20, %27\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %13, %16, %1\n\t"    // ac0 += ul0 bh
          "v_perm_b32 %10, %23, %22, %27\n\t"
          "v_perm_b32 %11, %25, %24, %27\n\t"
          "v_mfma_f32_16x16x32_f16 %3, %15, %16, %3\n\t"    // ac1 += ul1 bh
          "v_mfma_f32_16x16x32_f16 %2, %14, %16, %2"        // am1 += uh1 bh
          : "+v"(am0), "+v"(ac0), "+v"(am1), "+v"(ac1), "=&v"(h0), "=&v"(h1), "=&v"(h2),
            "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
          : "a"(uh0), "a"(ul0), "a"(uh1), "a"(ul1), "v"(bh), "v"(bl), "v"(q0[0]), "v"(q0[1]),
            "v"(q0[2]), "v"(q0[3]), "v"(q1[0]), "v"(q1[1]), "v"(q1[2]), "v"(q1[3]), "s"(selh),
            "s"(sell));
    }
    nbh = __builtin_bit_cast(h8, u32x4{h0, h1, h2, h3});
    nbl = __builtin_bit_cast(h8, u32x4{l0, l1, l2, l3});
  } else if constexpr (FIRST) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %4, %9, 0\n\t"         // ac0  = uh0 bl
        "v_mfma_f32_16x16x32_f16 %3, %6, %9, 0\n\t"         // ac1  = uh1 bl
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, 0\n\t"         // am0  = uh0 bh
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"        // ac0 += ul0 bh
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n\t"        // ac1 += ul1 bh
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, 0"             // am1  = uh1 bh
        : "=&v"(am0), "=&v"(ac0), "=&v"(am1), "=&v"(ac1)
        : "a"(uh0), "a"(ul0), "a"(uh1), "a"(ul1), "v"(bh), "v"(bl));
  } else {
    asm volatile(
        "v_mfma_f32_16x16x32_f16 %1, %4, %9, %1\n\t"        // ac0 += uh0 bl
        "v_mfma_f32_16x16x32_f16 %3, %6, %9, %3\n\t"        // ac1 += uh1 bl
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"        // am0 += uh0 bh
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"        // ac0 += ul0 bh
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n\t"        // ac1 += ul1 bh
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2"            // am1 += uh1 bh
        : "+v"(am0), "+v"(ac0), "+v"(am1), "+v"(ac1)
        : "a"(uh0), "a"(ul0), "a"(uh1), "a"(ul1), "v"(bh), "v"(bl));
  }
}

// PROGRESSIVE step (fwd_body_x<.., SLAB>): the 32-deep slabs of a wave's K slice are polled
// SEPARATELY.  A slab whose two producer workgroups have published is multiplied as soon as the
// slabs before it are done, the others are polled again -- their loads only: a finished slab's
// offset is sent out of the descriptor's range, which costs no memory traffic and keeps every
// load unconditional (rule (3) of DESIGN.md 5).  No nap before the first poll: a poll that
// comes back partly stale has still delivered work, and a wave that missed one slab re-reads
// 2 KB, not its whole slice, so the four waves reach the step's barrier closer together.
// The slabs are multiplied IN ORDER into the same accumulator chains as the single-gather form
// (per slab and gate tile: ac += uh bl, am += uh bh, ac += ul bh), so the result is bit-identical
// to it whatever the arrival order.  One slab = twelve MFMAs on eight accumulators, an
// accumulator is reused four MFMAs (64 cycles of pipe) later at the earliest.
template <bool FIRST>
__device__ __forceinline__ void fwd_slab_mfma(f32x4 (&am)[4], f32x4 (&ac)[4], const f32x4& uh0,
                                              const f32x4& uh1, const f32x4& uh2, const f32x4& uh3,
                                              const f32x4& ul0, const f32x4& ul1, const f32x4& ul2,
                                              const f32x4& ul3, const h8& bh, const h8& bl) {
  if constexpr (FIRST) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %8, %17, 0\n\t"        // ac_j  = uh_j bl
        "v_mfma_f32_16x16x32_f16 %5, %9, %17, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %10, %17, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %11, %17, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %8, %16, 0\n\t"        // am_j  = uh_j bh
        "v_mfma_f32_16x16x32_f16 %1, %9, %16, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %10, %16, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %11, %16, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %12, %16, %4\n\t"      // ac_j += ul_j bh
        "v_mfma_f32_16x16x32_f16 %5, %13, %16, %5\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %14, %16, %6\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %15, %16, %7\n\t"
        "s_nop 11"
        : "=&v"(am[0]), "=&v"(am[1]), "=&v"(am[2]), "=&v"(am[3]), "=&v"(ac[0]), "=&v"(ac[1]),
          "=&v"(ac[2]), "=&v"(ac[3])
        : "a"(uh0), "a"(uh1), "a"(uh2), "a"(uh3), "a"(ul0), "a"(ul1), "a"(ul2), "a"(ul3), "v"(bh),
          "v"(bl));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %8, %17, %4\n\t"       // ac_j += uh_j bl
        "v_mfma_f32_16x16x32_f16 %5, %9, %17, %5\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %10, %17, %6\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %11, %17, %7\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t"       // am_j += uh_j bh
        "v_mfma_f32_16x16x32_f16 %1, %9, %16, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %10, %16, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %11, %16, %3\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %12, %16, %4\n\t"      // ac_j += ul_j bh
        "v_mfma_f32_16x16x32_f16 %5, %13, %16, %5\n\t"
        "v_mfma_f32_16x16x32_f16 %6, %14, %16, %6\n\t"
        "v_mfma_f32_16x16x32_f16 %7, %15, %16, %7\n\t"
        "s_nop 11"
        : "+v"(am[0]), "+v"(am[1]), "+v"(am[2]), "+v"(am[3]), "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]),
          "+v"(ac[3])
        : "a"(uh0), "a"(uh1), "a"(uh2), "a"(uh3), "a"(ul0), "a"(ul1), "a"(ul2), "a"(ul3), "v"(bh),
          "v"(bl));
  }
}

// the same for a workgroup of TWO gate tiles (NJ = 2, eight units per workgroup): six MFMAs per
// slab on four accumulators, an accumulator reused four MFMAs (64 cycles of pipe) later
template <bool FIRST>
__device__ __forceinline__ void fwd_slab_mfma2(f32x4 (&am)[2], f32x4 (&ac)[2], const f32x4& uh0,
                                               const f32x4& uh1, const f32x4& ul0, const f32x4& ul1,
                                               const h8& bh, const h8& bl) {
  if constexpr (FIRST) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %4, %9, 0\n\t"        // ac_j  = uh_j bl
        "v_mfma_f32_16x16x32_f16 %3, %5, %9, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, 0\n\t"        // am_j  = uh_j bh
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, 0\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\t"       // ac_j += ul_j bh
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n\t"
        "s_nop 11"
        : "=&v"(am[0]), "=&v"(am[1]), "=&v"(ac[0]), "=&v"(ac[1])
        : "a"(uh0), "a"(uh1), "a"(ul0), "a"(ul1), "v"(bh), "v"(bl));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %4, %9, %2\n\t"       // ac_j += uh_j bl
        "v_mfma_f32_16x16x32_f16 %3, %5, %9, %3\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"       // am_j += uh_j bh
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\t"       // ac_j += ul_j bh
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3\n\t"
        "s_nop 11"
        : "+v"(am[0]), "+v"(am[1]), "+v"(ac[0]), "+v"(ac[1])
        : "a"(uh0), "a"(uh1), "a"(ul0), "a"(ul1), "v"(bh), "v"(bl));
  }
}

// NJ = gate tiles (of 4 units) per workgroup: 4 = sixteen units (H/16 workgroups per chain, the
// default); 2 = EIGHT units (H/8 workgroups per chain; asr_lstm_plan picks it where the layer
// then still leaves half of the CUs free -- cfg2's 4 chains: 128 of 256 instead of 64): half
// the MFMAs, partial tiles and LDS traffic on the step's critical chain, the same gather.
// Waves w < NJ finish a tile each; a (sample, unit)'s products and their summation order (K
// split over the FOUR waves) do not depend on NJ: results are bit-identical.
template <int NKW, bool FAST, bool EXACT, bool SLAB = false, int NJ = 4>
__device__ __forceinline__ void fwd_body_x(const LstmParams& p, int unit, int wg, float* lds) {
  // Requires H == 128 * NKW (every lane's gather groups and units exist)
  static_assert(NJ == 4 || (NJ == 2 && !EXACT), "gate tiles per workgroup");
  constexpr int NT = 1;                            // batch tiles per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, nl = lane & 15;
  const int H = p.H, H4 = 4 * H, H2 = 2 * H;
  const int UG = H >> 2;
  const int dir = unit / p.NB, bt0 = unit % p.NB;
  const bool fin = w < NJ;                         // this wave finishes a gate tile
  const int ug = wg * NJ + (fin ? w : 0);          // the unit group this wave FINISHES
  const int u = 4 * ug + g;
  const int kbase = 32 * NKW * w;                  // first unit of this wave's K slice
  f32x4* part = reinterpret_cast<f32x4*>(lds);     // [2 bufs][4 waves][NJ gate tiles][64 lanes]

  // EXACT: the products on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate, bitwise an fmaf
  // chain).  MFMA m = (kk, half, e) of a gate tile takes from lane (g, nl) the fp32 word e of
  // its gathered group (kk, half), i.e. k-index g <-> unit kbase + 32 kk + 8 g + 4 half + e: the
  // exchange layout and the gather are those of the split path, the words are plain tagged fp32.
  constexpr int NM = EXACT ? 8 * NKW : 1;          // fp32 MFMAs per gate tile
  float uf[NJ][NM];                                // EXACT: one A-fragment register each (AGPRs)
  if constexpr (EXACT) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ugj = wg * NJ + j;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int k = kbase + 32 * (m >> 3) + 8 * g + (m & 7);      // (m & 7) = 4 half + e
        uf[j][m] = p.U[((size_t)(dir * H + k)) * H4 + 16 * ugj + nl];
        asm volatile("" : "+a"(uf[j][m]));         // AGPR-class from here on
      }
    }
  }
  f32x4 ufh[NJ][NKW], ufl[NJ][NKW];                // bit patterns of 8 halfs each (AGPRs)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if constexpr (EXACT) break;
    const int ugj = wg * NJ + j;
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
    for (int i = 0; i < NL; ++i) {
      v[x][i] = __builtin_amdgcn_raw_buffer_load_b128(
          rsrc, goff, (unsigned)((8 * (i >> 1) + (i & 1))) * gstep, FAST ? kNt : kSc1);
    }
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
    if (!fin) return;                              // (wave-uniform: waves NJ .. 3 only multiply)
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
    const bool tr = p.trace && lane == 0 && (unsigned)(s - p.trace_s0) < 16u;
    if (tr) p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2] = wall_clock64();
    zx_next[x] = load_zx(x, s + 1);
    // two LDS buffers by step parity (the one barrier per step keeps the waves at most one
    // step apart)
    const int buf = s & 1;
    f32x4* mine = part + ((size_t)buf * 4 + w) * NJ * 64;
    if constexpr (EXACT) {
      // the gathered fp32 words ARE the B operands (tag bit left in: <= 1 ulp); the four gate
      // tiles' accumulator chains are interleaved (32 cycles of pipe per MFMA, 40 of latency)
      f32x4 acc[NJ];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float bw = __uint_as_float(v[x][m >> 2][m & 3]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (m == 0)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[j]) : "a"(uf[j][0]), "v"(bw));
          else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(uf[j][m]), "v"(bw));
        }
      }
      if constexpr (NJ == 4)
        asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
      for (int j = 0; j < NJ; ++j) mine[j * 64 + lane] = acc[j];
    } else {
    // exchanged word = fp16 hi << 16 | fp16 lo (tag = LSB of lo, left in place)
    h8 bh[NKW], bl[NKW];
    auto unpack = [&](int kk) {
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
    };
    if constexpr (NKW == 4 && NJ == 4) {
      // r6 (H = 512): K step by K step.  The first pair of gate tiles unpacks the NEXT K step's
      // words between its MFMAs (fwd_pair_step<.., NEXT>); the second pair combines and stores
      // the first pair's partial tiles between its own K steps.  Per accumulator the products
      // are added in FwdMfma<4>::run2's order: the partial tiles are bit-identical to it.
      const u32x4 none = {0u, 0u, 0u, 0u};
      h8 nob, nol;
      unpack(0);
      f32x4 am0, ac0, am1, ac1, bm0, bc0, bm1, bc1;
      fwd_pair_step<true, true>(am0, ac0, am1, ac1, ufh[0][0], ufl[0][0], ufh[1][0], ufl[1][0],
                                bh[0], bl[0], v[x][2], v[x][3], bh[1], bl[1]);
      fwd_pair_step<false, true>(am0, ac0, am1, ac1, ufh[0][1], ufl[0][1], ufh[1][1], ufl[1][1],
                                 bh[1], bl[1], v[x][4], v[x][5], bh[2], bl[2]);
      fwd_pair_step<false, true>(am0, ac0, am1, ac1, ufh[0][2], ufl[0][2], ufh[1][2], ufl[1][2],
                                 bh[2], bl[2], v[x][6], v[x][7], bh[3], bl[3]);
      fwd_pair_step<false, false>(am0, ac0, am1, ac1, ufh[0][3], ufl[0][3], ufh[1][3], ufl[1][3],
                                  bh[3], bl[3], none, none, nob, nol);
      fwd_pair_step<true, false>(bm0, bc0, bm1, bc1, ufh[2][0], ufl[2][0], ufh[3][0], ufl[3][0],
                                 bh[0], bl[0], none, none, nob, nol);
      // (the first pair left the pipe six MFMAs ago)
      f32x4 r0, r1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r0[e] = __builtin_fmaf(ac0[e], 1.f / kLoScale, am0[e]);
        r1[e] = __builtin_fmaf(ac1[e], 1.f / kLoScale, am1[e]);
      }
      fwd_pair_step<false, false>(bm0, bc0, bm1, bc1, ufh[2][1], ufl[2][1], ufh[3][1], ufl[3][1],
                                  bh[1], bl[1], none, none, nob, nol);
      mine[0 * 64 + lane] = r0;
      mine[1 * 64 + lane] = r1;
      fwd_pair_step<false, false>(bm0, bc0, bm1, bc1, ufh[2][2], ufl[2][2], ufh[3][2], ufl[3][2],
                                  bh[2], bl[2], none, none, nob, nol);
      fwd_pair_step<false, false>(bm0, bc0, bm1, bc1, ufh[2][3], ufl[2][3], ufh[3][3], ufl[3][3],
                                  bh[3], bl[3], none, none, nob, nol);
      asm volatile("s_nop 11" : "+v"(bm0), "+v"(bc0), "+v"(bm1), "+v"(bc1));
      f32x4 r2, r3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r2[e] = __builtin_fmaf(bc0[e], 1.f / kLoScale, bm0[e]);
        r3[e] = __builtin_fmaf(bc1[e], 1.f / kLoScale, bm1[e]);
      }
      mine[2 * 64 + lane] = r2;
      mine[3 * 64 + lane] = r3;
    } else {
#pragma unroll
    for (int kk = 0; kk < NKW; ++kk) unpack(kk);
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
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
    }
    prof.stamp(2);
    __syncthreads();
    prof.stamp(3);
    const f32x4* all = part + (size_t)buf * 4 * NJ * 64 + (size_t)(fin ? w : 0) * 64 + lane;
    const f32x4 a = (all[0 * NJ * 64] + all[1 * NJ * 64]) + (all[2 * NJ * 64] + all[3 * NJ * 64]);
    finish_step(x, s, a, zx4);
    if (tr) p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2 + 1] = wall_clock64();
    prof.stamp(4);
    issue(x, s);                                   // this tile's h of step s, for step s + 1
    prof.stamp(5);
  };
  auto phase_prog = [&](int s) {
    const float4 zx4 = zx_next[0];
    prof.stamp(0);
    const unsigned flip = 0u - ((unsigned)((s - 1) >> 1) & 1u);
    const __amdgpu_buffer_rsrc_t rsrc = slot(0, s - 1);
    f32x4 am[NJ], ac[NJ];
    unsigned pend = (1u << NKW) - 1u;              // wave-uniform: slabs not yet multiplied
    long long t0 = 0;
    int round = 0;
    for (int i = 0; i < p.prepoll; ++i) __builtin_amdgcn_s_sleep(1);
    while (pend != 0u) {
#pragma unroll
      for (int kk = 0; kk < NKW; ++kk) {
        const unsigned o = (pend >> kk) & 1u ? goff : 0xFFFFFFF0u;
        v[0][2 * kk] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrc, o, (unsigned)(8 * kk) * gstep, FAST ? kNt : kSc1);
        v[0][2 * kk + 1] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrc, o, (unsigned)(8 * kk + 1) * gstep, FAST ? kNt : kSc1);
      }
      bool blocked = false;                        // a slab before this one is still missing
#pragma unroll
      for (int kk = 0; kk < NKW; ++kk) {
        if (!((pend >> kk) & 1u) || blocked) continue;
        const u32x4 q0 = v[0][2 * kk], q1 = v[0][2 * kk + 1];
        const unsigned x = (((q0[0] ^ flip) | (q0[1] ^ flip)) | ((q0[2] ^ flip) | (q0[3] ^ flip))) |
                           (((q1[0] ^ flip) | (q1[1] ^ flip)) | ((q1[2] ^ flip) | (q1[3] ^ flip)));
        if (__builtin_amdgcn_ballot_w64((x & 1u) != 0u) != 0ull && p.poll && !dead) {
          blocked = true;
          continue;
        }
        u32x4 hi, lo;
        hi[0] = __builtin_amdgcn_perm(q0[1], q0[0], 0x07060302u);
        hi[1] = __builtin_amdgcn_perm(q0[3], q0[2], 0x07060302u);
        hi[2] = __builtin_amdgcn_perm(q1[1], q1[0], 0x07060302u);
        hi[3] = __builtin_amdgcn_perm(q1[3], q1[2], 0x07060302u);
        lo[0] = __builtin_amdgcn_perm(q0[1], q0[0], 0x05040100u);
        lo[1] = __builtin_amdgcn_perm(q0[3], q0[2], 0x05040100u);
        lo[2] = __builtin_amdgcn_perm(q1[1], q1[0], 0x05040100u);
        lo[3] = __builtin_amdgcn_perm(q1[3], q1[2], 0x05040100u);
        if constexpr (NJ == 2) {
          if (kk == 0)
            fwd_slab_mfma2<true>(am, ac, ufh[0][kk], ufh[1][kk], ufl[0][kk], ufl[1][kk],
                                 __builtin_bit_cast(h8, hi), __builtin_bit_cast(h8, lo));
          else
            fwd_slab_mfma2<false>(am, ac, ufh[0][kk], ufh[1][kk], ufl[0][kk], ufl[1][kk],
                                  __builtin_bit_cast(h8, hi), __builtin_bit_cast(h8, lo));
        } else {
        if (kk == 0)
          fwd_slab_mfma<true>(am, ac, ufh[0][kk], ufh[1][kk], ufh[2][kk], ufh[3][kk], ufl[0][kk],
                              ufl[1][kk], ufl[2][kk], ufl[3][kk], __builtin_bit_cast(h8, hi),
                              __builtin_bit_cast(h8, lo));
        else
          fwd_slab_mfma<false>(am, ac, ufh[0][kk], ufh[1][kk], ufh[2][kk], ufh[3][kk], ufl[0][kk],
                               ufl[1][kk], ufl[2][kk], ufl[3][kk], __builtin_bit_cast(h8, hi),
                               __builtin_bit_cast(h8, lo));
        }
        pend &= ~(1u << kk);
      }
      if (pend != 0u) {
        ++round;
        if (round == 2) t0 = wall_clock64();
        if (round > 2 && wall_clock64() - t0 > p.spin) {
          dead = true;                             // give up: finish without polling
          mark_timeout(p.status);
        }
        for (int i = 0; i < p.repoll; ++i) __builtin_amdgcn_s_sleep(1);
      }
    }
    prof.stamp(1);
    const bool tr = p.trace && lane == 0 && (unsigned)(s - p.trace_s0) < 16u;
    if (tr) p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2] = wall_clock64();
    zx_next[0] = load_zx(0, s + 1);
    const int buf = s & 1;
    f32x4* mine = part + ((size_t)buf * 4 + w) * NJ * 64;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(ac[j][e], 1.f / kLoScale, am[j][e]);
      mine[j * 64 + lane] = t;
    }
    prof.stamp(2);
    __syncthreads();
    prof.stamp(3);
    const f32x4* all = part + (size_t)buf * 4 * NJ * 64 + (size_t)(fin ? w : 0) * 64 + lane;
    const f32x4 a = (all[0 * NJ * 64] + all[1 * NJ * 64]) + (all[2 * NJ * 64] + all[3 * NJ * 64]);
    finish_step(0, s, a, zx4);
    if (tr) p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2 + 1] = wall_clock64();
    prof.stamp(4);
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
  if constexpr (SLAB && !EXACT) {
    for (; s < s_end; ++s) phase_prog(s);
  } else {
    if (s < s_end) {
      issue(0, s - 1);
      for (; s < s_end; ++s) phase(T0{}, s);
    }
  }
  prof.flush(p.status, w);
}

template <int NKW, bool EXACT, bool SLAB = false, int NJ = 4>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel_x(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit_local, wg;
  if (!map_block(p, unit_local, wg)) return;
  const int unit = p.chain_begin + unit_local;
  const bool fast = chain_on_one_xcd(p, unit, wg, reinterpret_cast<int*>(lds));
  if (fast) fwd_body_x<NKW, true, EXACT, SLAB, NJ>(p, unit, wg, lds);
  else fwd_body_x<NKW, false, EXACT, SLAB, NJ>(p, unit, wg, lds);
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


}  // namespace

#define ASR_KERN(f) static_cast<asr_lstm_kern_t>(f)
asr_lstm_kern_t asr_lstm_pick_fwd_h(int nkk, bool variants) {
  switch (nkk) {
    case 4: return variants ? ASR_KERN(lstm_fwd_kernel_hv<4>) : ASR_KERN(lstm_fwd_kernel_h<4>);
    case 8: return variants ? ASR_KERN(lstm_fwd_kernel_hv<8>) : ASR_KERN(lstm_fwd_kernel_h<8>);
    default: return variants ? ASR_KERN(lstm_fwd_kernel_hv<16>) : ASR_KERN(lstm_fwd_kernel_h<16>);
  }
}
asr_lstm_kern_t asr_lstm_pick_fwd_x(int H, bool exact, bool slab, bool eight) {
  if (eight && !exact) {          // eight units per workgroup (NJ = 2)
    if (H == 256) return slab ? ASR_KERN((lstm_fwd_kernel_x<2, false, true, 2>))
                              : ASR_KERN((lstm_fwd_kernel_x<2, false, false, 2>));
    return slab ? ASR_KERN((lstm_fwd_kernel_x<4, false, true, 2>))
                : ASR_KERN((lstm_fwd_kernel_x<4, false, false, 2>));
  }
  if (slab && !exact)
    return H == 256 ? ASR_KERN((lstm_fwd_kernel_x<2, false, true>))
                    : ASR_KERN((lstm_fwd_kernel_x<4, false, true>));
  if (H == 256) return exact ? ASR_KERN((lstm_fwd_kernel_x<2, true>)) : ASR_KERN((lstm_fwd_kernel_x<2, false>));
  return exact ? ASR_KERN((lstm_fwd_kernel_x<4, true>)) : ASR_KERN((lstm_fwd_kernel_x<4, false>));
}
asr_lstm_kern_t asr_lstm_pick_fwd_n1(int H) {
  return H == 256 ? ASR_KERN(lstm_fwd_kernel_n1<64>) : ASR_KERN(lstm_fwd_kernel_n1<128>);
}
