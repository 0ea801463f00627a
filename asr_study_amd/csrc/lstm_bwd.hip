// Backward (BPTT) recurrent LSTM kernels (design notes: header of lstm.hip; shared pieces:
// lstm_common.h): lstm_bwd_kernel_h / _hv (any H, cell variants, stepwise mode),
// lstm_bwd_kernel_x (plain cell, H = 256 / 512: unit split, partial dh tiles exchanged),
// lstm_bwd_kernel_c (two-dimensional split; the default at H = 512 and the EXACT kernel).
#include "lstm_common.h"

namespace {

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

// Two output tiles (a, b) against the same dz tile, their MFMAs alternating: per accumulator the
// products are added in the order of mfma_hl_tile (am = Uh0 Bh0 + Uh1 Bh1; ac = Uh0 Bl0 + Ul0 Bh0
// + Uh1 Bl1 + Ul1 Bh1), so the results are bit-identical to two calls of it; an accumulator is
// reused two MFMAs (32 cycles of pipe) later at the earliest.  Two halves (K step 0 / 1), so that
// the caller can put the previous pair's combine-and-publish between them; no trailing wait
// states: the caller reads the results behind the next pair's MFMAs (or pads itself).
template <int HALF>
__device__ __forceinline__ void mfma_hl_tile2(f32x4& ama, f32x4& aca, f32x4& amb, f32x4& acb,
                                              const f32x4& uh0a, const f32x4& ul0a,
                                              const f32x4& uh1a, const f32x4& ul1a,
                                              const f32x4& uh0b, const f32x4& ul0b,
                                              const f32x4& uh1b, const f32x4& ul1b, const h8& bh0,
                                              const h8& bl0, const h8& bh1, const h8& bl1) {
  if constexpr (HALF == 0) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, 0\n\t"       // ama  = uh0a bh0
        "v_mfma_f32_16x16x32_f16 %1, %4, %9, 0\n\t"       // aca  = uh0a bl0
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, 0\n\t"       // amb  = uh0b bh0
        "v_mfma_f32_16x16x32_f16 %3, %6, %9, 0\n\t"       // acb  = uh0b bl0
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"      // aca += ul0a bh0
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3"            // acb += ul0b bh0
        : "=&v"(ama), "=&v"(aca), "=&v"(amb), "=&v"(acb)
        : "a"(uh0a), "a"(ul0a), "a"(uh0b), "a"(ul0b), "v"(bh0), "v"(bl0));
  } else {
    asm volatile(
        "v_mfma_f32_16x16x32_f16 %0, %4, %8, %0\n\t"      // ama += uh1a bh1
        "v_mfma_f32_16x16x32_f16 %2, %6, %8, %2\n\t"      // amb += uh1b bh1
        "v_mfma_f32_16x16x32_f16 %1, %4, %9, %1\n\t"      // aca += uh1a bl1
        "v_mfma_f32_16x16x32_f16 %3, %6, %9, %3\n\t"      // acb += uh1b bl1
        "v_mfma_f32_16x16x32_f16 %1, %5, %8, %1\n\t"      // aca += ul1a bh1
        "v_mfma_f32_16x16x32_f16 %3, %7, %8, %3"            // acb += ul1b bh1
        : "+v"(ama), "+v"(aca), "+v"(amb), "+v"(acb)
        : "a"(uh1a), "a"(ul1a), "a"(uh1b), "a"(ul1b), "v"(bh1), "v"(bl1));
  }
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
        const float tch = VAR ? act_apply(p.act, cv) : fast_tanh(cv);
        const float d_o = dh * tch;
        float dcc = dc + dh * go * (VAR ? act_slope(p.act, tch) : (1.f - tch * tch));
        float dcz = 0.f;
        if (VAR) {                      // c = c_prev + k_c (c~ - c_prev)
          dcz = (1.f - kc) * dcc;
          dcc *= kc;
        }
        const float d_i = dcc * gg, d_g = dcc * gi, d_f = dcc * cpv;
        dc = dcc * gf + dcz;
        z4.x = d_i * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
        z4.y = d_f * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
        z4.z = d_g * (VAR ? act_slope(p.act, gg) : (1.f - gg * gg));
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

  // factors of one step that do not depend on the recurrent gradient (r6: computed from the
  // slab values prefetched a step ahead WHILE the gather is in flight, as bwd_body_c does; they
  // used to sit behind the await, ~200 clocks of the step's dependent chain):
  //   dz_o = dh A_o ; dcc = dc + dh B ; dz_{i,f,g} = dcc C_{i,f,g} ; dc' = dcc gf
  struct Pre { float dy, Ao, B, Ci, Cf, Cg, gf; };
  auto precompute = [&](int x) -> Pre {
#pragma clang fp contract(off)
    const float4 gt = nx_g[x];
    const float gi = gt.x, gf = gt.y, gg = gt.z, go = gt.w;
    const float tch = fast_tanh(nx_c[x]);
    Pre r;
    r.dy = nx_dy[x];
    r.Ao = tch * ((go > 0.f && go < 1.f) ? 0.2f : 0.f);
    r.B = go * __builtin_fmaf(-tch, tch, 1.f);
    r.Ci = gg * ((gi > 0.f && gi < 1.f) ? 0.2f : 0.f);
    r.Cf = nx_cp[x] * ((gf > 0.f && gf < 1.f) ? 0.2f : 0.f);
    r.Cg = gi * __builtin_fmaf(-gg, gg, 1.f);
    r.gf = gf;
    // keep all of it AHEAD of the await (the compiler would sink it to its uses behind the
    // polling loop, i.e. back onto the critical path)
    asm volatile("" : "+v"(r.dy), "+v"(r.Ao), "+v"(r.B), "+v"(r.Ci), "+v"(r.Cf), "+v"(r.Cg),
                 "+v"(r.gf));
    return r;
  };

  // everything of one step of tile x after its recurrent gradient dh_rec is known: cell
  // gradient, dz slab + LDS tile, barrier, partial dh tiles = U^T-slice x dz, publish.
  // ISSUE: whether the gather of this tile's partial tiles of step os is issued on the way.
  auto tail = [&](auto xc, auto issue_c, int s, const Pre& pre, float dh_rec, int os) {
#pragma clang fp contract(off)
    constexpr int x = decltype(xc)::value;
    constexpr bool ISSUE = decltype(issue_c)::value;
    float* sinv = lds + (size_t)(s & 1) * kTileFloats;        // [16] 1/scale
    _Float16* dzh = reinterpret_cast<_Float16*>(sinv + 16);   // [16][DZH] hi
    _Float16* dzl = dzh + 16 * DZH;                           // [16][DZH] lo
    const int t = dir == 0 ? p.T - 1 - s : s;
    {
      load_slabs(x, s + 1);
      const float dh = __builtin_fmaf(cmask[x], dh_rec, pre.dy);
      const float dcc = __builtin_fmaf(dh, pre.B, dc[x]);
      dc[x] = dcc * pre.gf;
      float4 z4;
      z4.x = dcc * pre.Ci;
      z4.y = dcc * pre.Cf;
      z4.z = dcc * pre.Cg;
      z4.w = dh * pre.Ao;
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
      // r6: the output tiles go through the pipe in PAIRS whose MFMAs alternate (twelve MFMAs on
      // four accumulators: an accumulator is reused two MFMAs later at the earliest, where one
      // tile's `ac` chain of four had them back to back), and a pair is combined and published in
      // the issue slots of the NEXT pair's MFMAs.  Per accumulator the products are added in the
      // order of mfma_hl_tile: the published words are unchanged.
      static_assert(TPW % 2 == 0, "tiles in pairs");
      auto finish = [&](int i, const f32x4& am, const f32x4& ac) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = tag_word(__builtin_fmaf(ac[e], usl, am[e] * us), wtag);
        // partial tile of the output units 16 (w + 4 i) ..: 4 P KB apart
        __builtin_amdgcn_raw_buffer_store_b128(o, wr, soff, i * (4 * P * 1024), FAST ? 0 : kSc1);
        // A 16-byte store reads its data registers over several cycles, and hipcc pads that
        // hazard only for stores WITHOUT a register in the soffset field: with one, the next
        // tile's combine may overwrite `o` at once -- measured on gfx950 as the second dword of
        // lanes 12..15 of each row going out stale in about one wave per launch.  Two wait
        // states behind every publish.
        asm volatile("s_nop 1" : "+v"(o));
      };
      f32x4 am[TPW], ac[TPW];
#pragma unroll
      for (int i = 0; i < TPW; i += 2) {
        mfma_hl_tile2<0>(am[i], ac[i], am[i + 1], ac[i + 1], ufh[i][0], ufl[i][0], ufh[i][1],
                         ufl[i][1], ufh[i + 1][0], ufl[i + 1][0], ufh[i + 1][1], ufl[i + 1][1],
                         bh[0], bl[0], bh[1], bl[1]);
        // (the previous pair left the pipe six MFMAs ago: >= 96 cycles)
        if (i > 0) finish(i - 2, am[i - 2], ac[i - 2]);
        mfma_hl_tile2<1>(am[i], ac[i], am[i + 1], ac[i + 1], ufh[i][0], ufl[i][0], ufh[i][1],
                         ufl[i][1], ufh[i + 1][0], ufl[i + 1][0], ufh[i + 1][1], ufl[i + 1][1],
                         bh[0], bl[0], bh[1], bl[1]);
        if (i > 0) finish(i - 1, am[i - 1], ac[i - 1]);
      }
      asm volatile("s_nop 11" : "+v"(am[TPW - 2]), "+v"(ac[TPW - 2]), "+v"(am[TPW - 1]),
                   "+v"(ac[TPW - 1]));
      finish(TPW - 2, am[TPW - 2], ac[TPW - 2]);
      finish(TPW - 1, am[TPW - 1], ac[TPW - 1]);
    }
    prof.stamp(4);
    if (ISSUE) issue(x, os);
    prof.stamp(5);
  };
  // one phase = one step (s >= 1) of tile x: finish its gather, reduce, then `tail`.
  auto phase = [&](auto xc, int s) {
    constexpr int x = decltype(xc)::value;
    const Pre pre = precompute(x);
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
    tail(xc, std::true_type{}, s, pre, dh_rec, s);
  };
  using T0 = std::integral_constant<int, 0>;
  int s = p.s_begin;
  if (s == 0) {
    // step 0: no recurrent gradient yet, nothing to gather
    tail(T0{}, std::false_type{}, 0, precompute(0), 0.f, 0);
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

//
// NBLK = 2 (the COMPACT form, asr_lstm_args.compact): two output blocks instead of four, i.e. a
// workgroup owns H/2 outputs (TW = 4 OT / NBLK output tiles per wave, 256 AGPRs of U^T at
// H = 512) and a chain is H/32 workgroups -- the layer then occupies HALF as many CUs (128 of 256
// at cfg3) at twice the MFMAs per step and workgroup; the gate-gradient arithmetic is duplicated
// twice instead of four times, the gather and the published volume of a chain-step are
// unchanged.  It exists so that the weight-gradient GEMMs of the layer above can run on the other
// half of the chip beside the BPTT of this one (engine.backward).  Same exchange layout
// ([output tile][slice a][256 words]), same products; a (sample, unit)'s partial sums are added
// in the same order, so the gate gradients are bit-identical to NBLK = 4.
//
// PL (asr_lstm_args.dz_hl): the gate gradients leave as the packed planes of asr_pack_hl instead
// of the fp32 slab.  A thread's 16 values (4 units x 4 gates) are one (16 hi, 16 lo) group of
// the plane row at the byte offset of its fp32 values: the same four 16-byte stores, other
// contents -- and they ARE the LDS tile of the dz @ U^T products (hi = fp16(x s), lo = fp16(x s -
// hi), s = the power of two of *dz_bound, known before the pass), so the per-sample scale (a DPP
// row maximum, frexp / ldexp, the 1 / scale word in LDS) drops out of the dependent chain.
template <int OT, bool FAST, bool EXACT, int NBLK = 4, bool PL = false>
__device__ __forceinline__ void bwd_body_c(const LstmParams& p, int unit, int cw, float* lds) {
  static_assert(!(PL && EXACT), "planes are a split-fp16 format");
  constexpr int PA = 4 * OT;                       // reduction slices (H / 64)
  constexpr int KS = 8;                            // K-steps of 32 columns per slice
  constexpr int NTILE = 16 * OT / NBLK;            // output tiles of a workgroup
  constexpr int TW = NTILE / 4;                    // ... of a wave
  constexpr int NG = 4 / NBLK;                     // units per thread whose bias partials it keeps
  static_assert(NBLK == 4 || NBLK == 2, "output blocks per chain");
  constexpr int DZS = 264;                         // LDS row stride of the dz tile (halfs)
  constexpr int DZF = 260;                         // EXACT: row stride of the fp32 dz tile (floats)
  constexpr int kBufFloats = EXACT ? 16 * DZF : 16 + (2 * 16 * DZS) / 2;   // fp32 tile | sinv + hi + lo
  constexpr int kSlotWords = NBLK * NTILE * PA * 256;         // one exchange slot of a chain
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
  float uf[TW][NM];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const int orow = 16 * NTILE * b + 16 * (w + 4 * i) + nl;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int j = 256 * a + 16 * (m >> 2) + 4 * g + (m & 3);
        uf[i][m] = p.U[((size_t)(dir * H + orow)) * H4 + j];
        asm volatile("" : "+a"(uf[i][m]));         // AGPR-class from here on
      }
    }
  }
  f32x4 ufh[TW][KS], ufl[TW][KS];                  // bit patterns of 8 halfs each (AGPRs)
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    if constexpr (EXACT) break;
    const int orow = 16 * NTILE * b + 16 * (w + 4 * i) + nl;  // output unit of this lane's A row
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
  float4 gsum[NG];                                 // bias-gradient partials of units u0 + b + NBLK k
#pragma unroll
  for (int k = 0; k < NG; ++k) gsum[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float zmax = 0.f;
  bool dead = false;
  StepProf prof;
  prof.init(false);
  float sc_g = 1.f, us_g = 1.f;                    // PL: the planes' scale and its inverse
  if constexpr (PL) {
    sc_g = asr_pow2_scale(p.dz_bound);
    us_g = 1.f / sc_g;
    if (p.dz_scale_out && blockIdx.x == 0 && tid == 0) *p.dz_scale_out = sc_g;
  }

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
  const unsigned vo_z = (n & (NBLK - 1)) == b ? vo_g : 0xC0000000u;   // dz rows: owner lanes only
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
  // (PL: the plane row of a (frame, sample) starts at the byte offset of its fp32 row)
  float* st_dz = (PL ? reinterpret_cast<float*>(p.dz_hl) : p.dz) + (size_t)tt0 * fr_g;
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
  // (a', block of these units), 1 KB apart (immediates of one offset register); the slot is
  // indexed by the GLOBAL output tile 4 a + (q >> 2) = block * NTILE + tile of the block
  constexpr int NL = PA;
  const unsigned goff = (unsigned)((((4 * a + (q >> 2)) * PA) * 256 + n * 16 + (q & 3) * 4) * 4);
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
    h8 pl_hi[2], pl_lo[2];                         // PL: the thread's plane group
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
    } else if constexpr (PL) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(z[j][0]), fabsf(z[j][1])), fmaxf(fabsf(z[j][2]), fabsf(z[j][3]))));
      zmax = fmaxf(zmax, m);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = z[j][e] * sc_g;
          const _Float16 h = (_Float16)x;
          pl_hi[j >> 1][4 * (j & 1) + e] = h;
          pl_lo[j >> 1][4 * (j & 1) + e] = (_Float16)(x - (float)h);
        }
      _Float16* rh = dzh + n * DZS + 8 * q;       // (tile layout: see the scaled form below)
      _Float16* rl = dzl + n * DZS + 8 * q;
      *reinterpret_cast<h8*>(rh) = pl_hi[0];
      *reinterpret_cast<h8*>(rh + 128) = pl_hi[1];
      *reinterpret_cast<h8*>(rl) = pl_lo[0];
      *reinterpret_cast<h8*>(rl + 128) = pl_lo[1];
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
    if constexpr (NBLK == 4) {
      if (b == 0) { gsum[0].x += z[0][0]; gsum[0].y += z[0][1]; gsum[0].z += z[0][2]; gsum[0].w += z[0][3]; }
      else if (b == 1) { gsum[0].x += z[1][0]; gsum[0].y += z[1][1]; gsum[0].z += z[1][2]; gsum[0].w += z[1][3]; }
      else if (b == 2) { gsum[0].x += z[2][0]; gsum[0].y += z[2][1]; gsum[0].z += z[2][2]; gsum[0].w += z[2][3]; }
      else { gsum[0].x += z[3][0]; gsum[0].y += z[3][1]; gsum[0].z += z[3][2]; gsum[0].w += z[3][3]; }
    } else {
      if (b == 0) {
        gsum[0].x += z[0][0]; gsum[0].y += z[0][1]; gsum[0].z += z[0][2]; gsum[0].w += z[0][3];
        gsum[1].x += z[2][0]; gsum[1].y += z[2][1]; gsum[1].z += z[2][2]; gsum[1].w += z[2][3];
      } else {
        gsum[0].x += z[1][0]; gsum[0].y += z[1][1]; gsum[0].z += z[1][2]; gsum[0].w += z[1][3];
        gsum[1].x += z[3][0]; gsum[1].y += z[3][1]; gsum[1].z += z[3][2]; gsum[1].w += z[3][3];
      }
    }
    // off the dependent path (the other waves are still on their way to the barrier), and ahead
    // of the publish and the gather in the CU's in-order memory queue only by a whole MFMA phase:
    // this workgroup's rows of the dz slab, and the slab values of step s + 2 into the set this
    // step has just consumed
    {
      // (no branch around them, so that the compiler can count them in its waits: the lanes
      // that do not own the row store beyond the resource's range, which the hardware drops)
      // NT (streaming) stores: written once, read by a later kernel -- without the hint the dirty
      // dz lines displace the exchange slots from L2 and those are written back to HBM step after
      // step (r6, PMC WRITE_SIZE per cfg3 layer 2.18 -> 1.59 GB, isolated step 2.07 -> 1.98 us;
      // the same hint on the slab LOADS removes the rest of the excess -- 1.05 GB = dz alone -- but
      // costs 0.3 us per step)
      const __amdgpu_buffer_rsrc_t rz = rs(st_dz, fr_g);
      if constexpr (PL) {
        // [16 hi][16 lo] of reduction indices 16 q' .. (q' = the thread's group in the row)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pl_hi[0]), rz, vo_z, 0, kNt);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pl_hi[1]), rz, vo_z, 16, kNt);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pl_lo[0]), rz, vo_z, 32, kNt);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pl_lo[1]), rz, vo_z, 48, kNt);
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 zz = {z[j][0], z[j][1], z[j][2], z[j][3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zz), rz, vo_z, 16 * j, kNt);
      }
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
      static_assert(!EXACT || NBLK == 4, "the exact kernel exists in the four-block form");
      f32x4 acc[TW];
      // (the OT tiles' accumulator chains interleaved: 32 cycles of pipe per MFMA, 40 of latency)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float bw = bq[m >> 2][m & 3];
#pragma unroll
        for (int i = 0; i < TW; ++i) {
          if (m == 0)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[i]) : "a"(uf[i][0]), "v"(bw));
          else
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(uf[i][m]), "v"(bw));
        }
      }
      if constexpr (OT == 2) asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
      else asm volatile("s_nop 15" : "+v"(acc[0]));
#pragma unroll
      for (int i = 0; i < TW; ++i) {
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
      // (PL: the tile's lo halfs are NOT scaled by kLoScale -- the planes' convention -- while the
      // U^T fragments' are: hi x lo joins hi x hi, lo x hi keeps the 1 / 2048)
      const float us = PL ? us_g : sinv[nl];
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
          o[e] = PL ? tag_word(__builtin_fmaf(a2[e], usl, (am[e] + a1[e]) * us), wtag)
                    : tag_word(__builtin_fmaf(a1[e] + a2[e], usl, am[e] * us), wtag);
        // output tile w + 4 i of this block: 4 PA KB further on
        __builtin_amdgcn_raw_buffer_store_b128(o, wr, soff, i * (4 * PA * 1024), FAST ? 0 : kSc1);
      };
      f32x4 am[TW], a1[TW], a2[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) {
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
      mfma_settle(am[TW - 1], a1[TW - 1], a2[TW - 1]);
      finish(TW - 1, am[TW - 1], a1[TW - 1], a2[TW - 1]);
    }
    if (p.trace && lane == 0 && (unsigned)(s - p.trace_s0) < 16u)
      p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2 + 1] = wall_clock64();
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
    if (p.trace && lane == 0 && (unsigned)(s - p.trace_s0) < 16u)
      p.trace[(((size_t)blockIdx.x * 4 + w) * 16 + (s - p.trace_s0)) * 2] = wall_clock64();
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
    // sum over the 16 samples of every thread's float4 (unit u0 + b + NBLK k), fixed order
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
      float vs[4] = {gsum[kg].x, gsum[kg].y, gsum[kg].z, gsum[kg].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {                // over the wave's four samples (lane >> 4)
        vs[k] += __shfl_xor(vs[k], 16);
        vs[k] += __shfl_xor(vs[k], 32);
      }
      __syncthreads();
      if (lane < 16) *reinterpret_cast<float4*>(lds + (w * 16 + lane) * 4) =
          make_float4(vs[0], vs[1], vs[2], vs[3]);
      __syncthreads();
      if (tid < 64) {                              // (unit quad tid >> 2, gate tid & 3)
        const float tsum = ((lds[tid] + lds[64 + tid]) + lds[128 + tid]) + lds[192 + tid];
        float* dst = p.db_part + ((size_t)bt * 2 + dir) * H4 + 256 * a + 16 * (tid >> 2) +
                     4 * (b + NBLK * kg) + (tid & 3);
        *dst = (p.s_begin > 0 ? *dst : 0.f) + tsum;
      }
      __syncthreads();
    }
  }
  if (p.dz_absmax && b == 0) {
    zmax = asr_wave_max(zmax);
    if (lane == 0 && zmax > 0.f) atomicMax(p.dz_absmax, __float_as_uint(zmax));
  }
}

template <int OT, bool EXACT, int NBLK = 4, bool PL = false>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel_c(LstmParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit_local, cw;
  if (!map_block(p, unit_local, cw)) return;
  const int unit = p.chain_begin + unit_local;
  const bool fast = chain_on_one_xcd(p, unit, cw, reinterpret_cast<int*>(lds));
  if (fast) bwd_body_c<OT, true, EXACT, NBLK, PL>(p, unit, cw, lds);
  else bwd_body_c<OT, false, EXACT, NBLK, PL>(p, unit, cw, lds);
}


}  // namespace

#define ASR_KERN(f) static_cast<asr_lstm_kern_t>(f)
asr_lstm_kern_t asr_lstm_pick_bwd_h(int tpw, bool variants) {
  switch (tpw) {
    case 1: return variants ? ASR_KERN(lstm_bwd_kernel_hv<1>) : ASR_KERN(lstm_bwd_kernel_h<1>);
    case 2: return variants ? ASR_KERN(lstm_bwd_kernel_hv<2>) : ASR_KERN(lstm_bwd_kernel_h<2>);
    case 4: return variants ? ASR_KERN(lstm_bwd_kernel_hv<4>) : ASR_KERN(lstm_bwd_kernel_h<4>);
    default: return variants ? ASR_KERN(lstm_bwd_kernel_hv<8>) : ASR_KERN(lstm_bwd_kernel_h<8>);
  }
}
asr_lstm_kern_t asr_lstm_pick_bwd_x(int H) {
  return H == 256 ? ASR_KERN(lstm_bwd_kernel_x<4>) : ASR_KERN(lstm_bwd_kernel_x<8>);
}
asr_lstm_kern_t asr_lstm_pick_bwd_c(int H, bool exact, bool compact, bool planes) {
  if (planes && !exact) {
    if (compact)
      return H == 256 ? ASR_KERN((lstm_bwd_kernel_c<1, false, 2, true>))
                      : ASR_KERN((lstm_bwd_kernel_c<2, false, 2, true>));
    return H == 256 ? ASR_KERN((lstm_bwd_kernel_c<1, false, 4, true>))
                    : ASR_KERN((lstm_bwd_kernel_c<2, false, 4, true>));
  }
  if (compact && !exact)
    return H == 256 ? ASR_KERN((lstm_bwd_kernel_c<1, false, 2>)) : ASR_KERN((lstm_bwd_kernel_c<2, false, 2>));
  if (H == 256) return exact ? ASR_KERN((lstm_bwd_kernel_c<1, true>)) : ASR_KERN((lstm_bwd_kernel_c<1, false>));
  return exact ? ASR_KERN((lstm_bwd_kernel_c<2, true>)) : ASR_KERN((lstm_bwd_kernel_c<2, false>));
}
