// K9 CTC prefix beam search on the device -- gfx950.
//
// Replaces core/ctc_utils.py:48-50 (tf.nn.ctc_beam_search_decoder, top_paths=1) on logits that
// never leave HBM: eval.py / predict.py decode with width 400 (utils/core_utils.py:70-71)
// without the D2H copy of the (T, n_pad, C) slab the host decoder (decode_host.cpp) needs.
// Same algorithm, same arithmetic (double) and the same tie-breaking as decode_host.cpp; what
// changes is the shape of the loop (tests/beam_device_model.py is a line-by-line CPU model of
// it, checked against the oracle and the host decoder):
//
//  * one workgroup per utterance; the frame's <= W branches live in LDS, nodes outside the beam
//    keep only their child block in HBM.  Node ids are allocated in blocks of C-1 when a prefix
//    first expands, so the creation order TensorFlow's BeamComparer breaks ties with IS the id;
//  * a frame's branches are the previous frame's beam in its sorted order, so they are never
//    sorted; their new totals are ranked once (rank sort over LDS broadcasts);
//  * the beam is a SORTED array held in the registers of wave 0 for the whole frame, worst
//    first, place j in lane j % 64 of slot j / 64: the heap bottom is place 0, an insertion is
//    one ballot-count of the worse entries plus a one-lane shift (DPP wave_shl:1) of exactly
//    those -- slot 0 alone for a child that only just beats the bottom;
//  * a branch's turn evaluates its C-1 children in the lanes of wave 0: children that fail
//    against the current bottom are decided at once (the bottom only rises), the lowest-label
//    candidate is inserted, then the rest is re-evaluated (an insertion can evict a sibling
//    that was active).  The turn loop ends at the first branch whose old total does not beat
//    the bottom of a full beam (the branches are sorted by it).
#include "common.h"

#include <limits.h>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxC = 64;               // classes (blank included): one wave handles a frame

struct Rec {                            // per node, HBM
  int children;                         // child block or -1 (never expanded)
  int turn;                             // branch index in the current frame or -1 (not in beam)
};

struct BeamParams {
  const float* logits;
  const int* seq_len;
  int T, n_pad, C, W, merge;
  int* decoded;
  int* decoded_len;
  float* score;
  char* ws;
  size_t ws_per_utt;
  int max_blocks;
};

__device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }

__device__ __forceinline__ double lse(double a, double b) {
  if (a == neg_inf()) return b;
  if (b == neg_inf()) return a;
  const double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double readlane_d(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                          __builtin_amdgcn_readlane(__double2loint(x), l));
}
// lane i <- lane i+1, lane 63 <- carry
template <bool DPP>
__device__ __forceinline__ int shl1_i(int x, int carry) {
  if (DPP) return __builtin_amdgcn_update_dpp(carry, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  const int t = __shfl_down(x, 1, 64);
  return (threadIdx.x & 63) == 63 ? carry : t;
}
template <bool DPP>
__device__ __forceinline__ double shl1_d(double x, double carry) {
  return __hiloint2double(shl1_i<DPP>(__double2hiint(x), __double2hiint(carry)),
                          shl1_i<DPP>(__double2loint(x), __double2loint(carry)));
}

size_t lds_bytes(int W) { return (size_t)W * 100 + 72 * 8 + 64 * 4 + 16 * 4; }

template <int E, bool DPP>
__global__ __launch_bounds__(kThreads) void ctc_beam_kernel(BeamParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = p.W, C = p.C, K = p.C - 1, blank = p.C - 1;
  double* b_ob = reinterpret_cast<double*>(smem);       // branch table, by turn
  double* b_ol = b_ob + W;
  double* b_ot = b_ol + W;
  double* b_ot0 = b_ot + W;                              // old total before any reset
  double* b_nb = b_ot0 + W;
  double* b_nl = b_nb + W;
  double* b_nt = b_nl + W;
  double* h_nt = b_nt + W;                               // the beam between the phases
  double* inp = h_nt + W;                                // [72]
  int* b_node = reinterpret_cast<int*>(inp + 72);
  int* b_par = b_node + W;
  int* b_kidhead = b_par + W;                            // first child that is a branch itself
  int* b_child = b_kidhead + W;                          // child block or -1
  int* b_kidnext = b_child + W;                          // (next such sibling + 1) | label << 12
  int* b_evicted = b_kidnext + W;
  int* h_ord = b_evicted + W;
  int* h_tag = h_ord + W;                                // >= 0: branch turn; else -(parent + 2)
  int* s_misc = h_tag + W;                               // nb, nblocks, hn

  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // head of the utterance's workspace: 100 MHz ticks spent in phases B, C, D, E, then the
  // numbers of expanding turns, insertions and child blocks (asr_ctc_beam_device_counters)
  long long* counters = reinterpret_cast<long long*>(p.ws + (size_t)n * p.ws_per_utt);
  long long tk_b = 0, tk_c = 0, tk_d = 0, tk_e = 0, n_turn = 0, n_ins = 0;
  Rec* rec = reinterpret_cast<Rec*>(p.ws + (size_t)n * p.ws_per_utt + 64);
  int* block_parent = reinterpret_cast<int*>(rec + (1 + (size_t)p.max_blocks * K));
  int Tn = p.seq_len[n];
  Tn = Tn < 0 ? 0 : (Tn > p.T ? p.T : Tn);

  if (tid == 0) {
    // the root: total 0, blank path 0, no label path (decode_host.cpp: beam_one)
    b_node[0] = 0; b_par[0] = -1;
    b_ob[0] = 0.0; b_ol[0] = neg_inf(); b_ot[0] = 0.0; b_ot0[0] = 0.0;
    b_evicted[0] = 0; b_kidhead[0] = -1;
    rec[0] = Rec{-1, 0};
    s_misc[0] = 1; s_misc[1] = 0; s_misc[2] = 1;
  }
  float xnext = (tid < C && Tn > 0) ? p.logits[(size_t)n * C + tid] : 0.f;
  __syncthreads();

  for (int t = 0; t < Tn; ++t) {
    // ---- the frame's scores: max-subtracted logits, no log-softmax
    if (wave == 0) {
      const float x = lane < C ? xnext : asr_neg_inf();
      const float mx = asr_wave_max(x);
      if (lane < C) inp[lane] = (double)x - (double)mx;
    }
    if (t + 1 < Tn && tid < C) xnext = p.logits[((size_t)(t + 1) * p.n_pad + n) * C + tid];
    __syncthreads();
    const int nb = s_misc[0];
    long long tk0 = wall_clock64();

    // ---- phase B: every branch takes the frame (label path fed from the parent if the
    //      parent is still in the beam)
    for (int q = tid; q < nb; q += kThreads) {
      const int node = b_node[q], par = b_par[q];
      b_child[q] = rec[node].children;
      int pt = -1;
      double nl = b_ol[q];
      if (par >= 0) {
        const int label = (node - 1) % K;
        pt = rec[par].turn;
        if (pt >= 0) {
          const int plabel = par == 0 ? -1 : (par - 1) % K;
          nl = lse(nl, label == plabel ? b_ob[pt] : b_ot[pt]);
          b_kidnext[q] = (atomicExch(&b_kidhead[pt], q) + 1) | (label << 12);
        }
        nl += inp[label];
      }
      const double nbk = b_ot[q] + inp[blank];
      b_nb[q] = nbk;
      b_nl[q] = nl;
      b_nt[q] = lse(nbk, nl);
    }
    __syncthreads();
    { const long long x = wall_clock64(); tk_b += x - tk0; tk0 = x; }

    // ---- phase C: rank the new totals -> the beam array (total descending, id ascending)
    for (int q = tid; q < nb; q += kThreads) {
      const double nt = b_nt[q];
      const int node = b_node[q];
      int r = 0;
      for (int j0 = 0; j0 < nb; j0 += 8) {                 // 16 LDS broadcasts in flight
        double o[8];
        int d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u < nb ? j0 + u : q;          // past the end: itself, counts 0
          o[u] = b_nt[j];
          d[u] = b_node[j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) r += ((o[u] > nt) | ((o[u] == nt) & (d[u] < node))) ? 1 : 0;
      }
      h_nt[r] = nt; h_ord[r] = node; h_tag[r] = q;
    }
    __syncthreads();
    { const long long x = wall_clock64(); tk_c += x - tk0; tk0 = x; }

    // ---- phase D: the branches' turns, wave 0, beam in registers
    if (wave == 0) {
      // WORST first and always W entries: place j = slot * 64 + lane holds the beam's
      // (W - 1 - j)-th best; empty places are (-inf, no id), worse than anything, so place 0 is
      // the heap bottom (or -inf while the beam is not full: "better than the bottom" is then the
      // candidate test of a beam with room) and every insertion is "drop place 0, move the worse
      // entries down one place, put the new one behind them" -- for a child that only just beats
      // the bottom that touches slot 0 alone
      double nt[E];
      int ord[E], tag[E];
      int hn = nb, nblk = s_misc[1];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        // (selects, not branches, wherever a lane condition picks a place's value: divergent
        // control flow around the register arrays makes the compiler copy them whole)
        const int j = e * 64 + lane, rk = W - 1 - j;
        const bool real = j < W && rk < nb;
        const int rks = real ? rk : 0;
        const double a = h_nt[rks];
        const int b = h_ord[rks], c = h_tag[rks];
        nt[e] = real ? a : (j >= W ? __builtin_huge_val() : neg_inf());
        ord[e] = real ? b : (j >= W ? -1 : INT_MAX);
        tag[e] = real ? c : -1;
      }
      const double inp_c = lane < K ? inp[lane] : 0.0;
      // the turn's own fields, 64 turns at a time in the lanes (only the old total can change
      // while the frame runs -- a reset -- so that one is read from LDS at the turn)
      double c_ot0 = 0.0, c_ob = 0.0;
      int c_node = 0, c_child = -1, c_kid = -1, c_label = -1;
      for (int r = 0; r < nb; ++r) {
        if ((r & 63) == 0) {
          const int rr = r + lane < nb ? r + lane : nb - 1;
          c_ot0 = b_ot0[rr]; c_ob = b_ob[rr]; c_node = b_node[rr]; c_child = b_child[rr];
          c_kid = b_kidhead[rr];
          c_label = c_node == 0 ? -1 : (c_node - 1) % K;
        }
        double theta = readlane_d(nt[0], 0);
        if (!(readlane_d(c_ot0, r & 63) > theta)) break;
        const double ot = b_ot[r];
        int k = readlane_i(c_kid, r & 63);
        int w = k >= 0 ? b_kidnext[k] : 0;
        if (!(ot > theta)) continue;
        const double ob = readlane_d(c_ob, r & 63);
        const int node = readlane_i(c_node, r & 63);
        int blk = readlane_i(c_child, r & 63);
        if (blk < 0) {                                     // first expansion: a block of ids
          blk = nblk++;
          if (lane == 0) { block_parent[blk] = node; rec[node].children = blk; }
          if (lane < K) rec[1 + (size_t)blk * K + lane] = Rec{-1, -1};
        }
        ++n_turn;
        const int base = 1 + blk * K;
        const int blabel = readlane_i(c_label, r & 63);
        int q = -1;                                        // the child if it is a branch itself
        while (k >= 0) {
          w = __builtin_amdgcn_readfirstlane(w);
          q = lane == (w >> 12) ? k : q;
          k = (w & 0xfff) - 1;
          if (k >= 0) w = b_kidnext[k];
        }
        bool active = false;
        if (__ballot(q >= 0) != 0ull) active = q >= 0 && b_evicted[q] == 0;
        const double prev = lane == blabel ? ob : ot;
        const double v = (lane < K && prev != neg_inf()) ? inp_c + prev : neg_inf();
        unsigned long long undecided = (1ull << K) - 1ull;
        while (undecided) {
          theta = readlane_d(nt[0], 0);
          const bool mine = (undecided >> lane) & 1ull;
          const bool cand = mine & !active & (v > theta);
          const bool rej = mine & !active & !cand;
          // TF resets a rejected child's OLD probabilities too: if it is a branch of this
          // frame (evicted a moment ago) it must not expand when its turn comes
          if (rej & (q >= 0)) { b_ob[q] = neg_inf(); b_ol[q] = neg_inf(); b_ot[q] = neg_inf(); }
          const unsigned long long m = __ballot(cand);
          undecided &= ~__ballot(rej);
          if (!m) break;
          const int cs = __ffsll((long long)m) - 1;
          undecided &= ~((2ull << cs) - 1ull);             // cs inserted, lower labels decided
          const double vs = readlane_d(v, cs);
          const int id = base + cs, ntag = -(node + 2);
          const int rb = readlane_i(tag[0], 0);            // place 0 leaves the beam
          if (rb >= 0) {
            if (lane == 0) b_evicted[rb] = 1;
            active = active && q != rb;                    // a sibling of cs: offered again
          }
          hn += rb == -1 ? 1 : 0;                          // it was an empty place
          ++n_ins;
          // entries worse than the new one are a prefix of the places
          int cw = __popcll(__ballot((nt[0] < vs) | ((nt[0] == vs) & (ord[0] > id))));
          if (E > 1 && cw == 64) {
            bool more = true;
#pragma unroll
            for (int e = 1; e < E; ++e) {
              if (more) {
                const int c = __popcll(__ballot((nt[e] < vs) | ((nt[e] == vs) & (ord[e] > id))));
                cw += c;
                more = c == 64;
              }
            }
          }
          const int pos = cw - 1;                          // its place once place 0 is gone
          {
            // (lane 63 moves only if pos >= 64: its carry from slot 1 is patched in below)
            const double pn = shl1_d<DPP>(nt[0], nt[0]);
            const int po = shl1_i<DPP>(ord[0], ord[0]);
            const int pg = shl1_i<DPP>(tag[0], tag[0]);
            const bool mv = lane < pos, at = lane == pos;
            nt[0] = mv ? pn : (at ? vs : nt[0]);
            ord[0] = mv ? po : (at ? id : ord[0]);
            tag[0] = mv ? pg : (at ? ntag : tag[0]);
          }
          if (E > 1 && pos >= 64) {
            {
              const double cn = readlane_d(nt[E > 1 ? 1 : 0], 0);
              const int co = readlane_i(ord[E > 1 ? 1 : 0], 0), ct = readlane_i(tag[E > 1 ? 1 : 0], 0);
              const bool l63 = lane == 63;
              nt[0] = l63 ? cn : nt[0];
              ord[0] = l63 ? co : ord[0];
              tag[0] = l63 ? ct : tag[0];
            }
#pragma unroll
            for (int e = 1; e < E; ++e) {
              if (e * 64 <= pos) {
                const int j = e * 64 + lane;
                const bool last = e + 1 >= E;
                const double pn = shl1_d<DPP>(nt[e], last ? 0.0 : readlane_d(nt[last ? e : e + 1], 0));
                const int po = shl1_i<DPP>(ord[e], last ? 0 : readlane_i(ord[last ? e : e + 1], 0));
                const int pg = shl1_i<DPP>(tag[e], last ? 0 : readlane_i(tag[last ? e : e + 1], 0));
                const bool mv = j < pos, at = j == pos;
                nt[e] = mv ? pn : (at ? vs : nt[e]);
                ord[e] = mv ? po : (at ? id : ord[e]);
                tag[e] = mv ? pg : (at ? ntag : tag[e]);
              }
            }
          }
        }
      }
      wave_sync();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int rk = W - 1 - (e * 64 + lane);
        if (tag[e] != -1) { h_nt[rk] = nt[e]; h_ord[rk] = ord[e]; h_tag[rk] = tag[e]; }
      }
      if (lane == 0) { s_misc[1] = nblk; s_misc[2] = hn; }
    }
    __syncthreads();
    { const long long x = wall_clock64(); tk_d += x - tk0; tk0 = x; }

    // ---- phase E: the beam is the next frame's branch table, in its sorted order
    const int hn = s_misc[2];
    double e_nt[4], e_nb[4], e_nl[4];
    int e_node[4], e_par[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = tid + k * kThreads;
      if (j < hn) {
        const int g = h_tag[j];
        e_nt[k] = h_nt[j];
        e_node[k] = h_ord[j];
        e_par[k] = g >= 0 ? b_par[g] : -(g + 2);
        e_nb[k] = g >= 0 ? b_nb[g] : neg_inf();
        e_nl[k] = g >= 0 ? b_nl[g] : e_nt[k];
      }
    }
    for (int q = tid; q < nb; q += kThreads) rec[b_node[q]].turn = -1;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = tid + k * kThreads;
      if (j < hn) {
        b_node[j] = e_node[k]; b_par[j] = e_par[k];
        b_ob[j] = e_nb[k]; b_ol[j] = e_nl[k]; b_ot[j] = e_nt[k]; b_ot0[j] = e_nt[k];
        b_evicted[j] = 0; b_kidhead[j] = -1;
        rec[e_node[k]].turn = j;
      }
    }
    if (tid == 0) s_misc[0] = hn;
    __syncthreads();
    tk_e += wall_clock64() - tk0;
  }

  // ---- the best leaf's label sequence (merge_repeated collapses consecutive repeats)
  int* out = p.decoded + (size_t)n * p.T;
  __shared__ int s_len;
  if (tid == 0) {
    const int best = b_node[0];
    int L = 0, prev = -1;
    for (int c = best; c != 0; c = block_parent[(c - 1) / K]) {
      const int label = (c - 1) % K;
      if (!p.merge || label != prev) ++L;
      prev = label;
    }
    int w = L;
    prev = -1;
    for (int c = best; c != 0; c = block_parent[(c - 1) / K]) {
      const int label = (c - 1) % K;
      if (!p.merge || label != prev) out[--w] = label;
      prev = label;
    }
    p.decoded_len[n] = L;
    if (p.score) p.score[n] = (float)b_ot[0];
    s_len = L;
    counters[0] = tk_b; counters[1] = tk_c; counters[2] = tk_d; counters[3] = tk_e;
    counters[4] = n_turn; counters[5] = n_ins; counters[6] = s_misc[1];
  }
  __syncthreads();
  for (int i = s_len + tid; i < p.T; i += kThreads) out[i] = -1;
}

typedef void (*beam_kern_t)(BeamParams);

beam_kern_t pick(int W) {           // (the lane shift of the sorted beam: DPP wave_shr:1)
  if (W <= 128) return ctc_beam_kernel<2, true>;
  if (W <= 448) return ctc_beam_kernel<7, true>;
  return ctc_beam_kernel<16, true>;
}

size_t ws_per_utt(int T, int C, int W, int* max_blocks) {
  // a node expands for the first time at most once, a frame has at most W branches
  const size_t blocks = (size_t)(T > 0 ? T : 1) * (size_t)W;
  *max_blocks = (int)blocks;
  return asr_align_up(64 + (1 + blocks * (size_t)(C - 1)) * sizeof(Rec) + blocks * sizeof(int),
                      256);
}

}  // namespace

extern "C" size_t asr_ctc_beam_device_workspace_bytes(int T, int N, int C, int beam_width) {
  if (T < 0 || N <= 0 || C < 2 || beam_width < 1) return 0;
  int mb;
  return ws_per_utt(T, C, beam_width, &mb) * (size_t)N;
}

extern "C" int asr_ctc_beam_device_counters(const void* workspace, int T, int N, int C,
                                            int beam_width, int utterance, long long* out7,
                                            asr_stream_t stream) {
  ASR_CHECK_ARG(workspace && out7 && utterance >= 0 && utterance < N, "beam counters: bad args");
  int mb;
  const size_t per = ws_per_utt(T, C, beam_width, &mb);
  ASR_CHECK_HIP(hipMemcpyAsync(out7, reinterpret_cast<const char*>(workspace) + per * utterance,
                               7 * sizeof(long long), hipMemcpyDeviceToHost,
                               (hipStream_t)stream));
  ASR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return ASR_OK;
}

extern "C" int asr_ctc_beam_device(const float* logits, const int* seq_len, int T, int N, int n_pad,
                                   int C, int beam_width, int merge_repeated, int* decoded,
                                   int* decoded_len, float* log_score, void* workspace,
                                   size_t ws_bytes, asr_stream_t stream) {
  ASR_CHECK_ARG(logits && seq_len && decoded && decoded_len && workspace, "beam: null pointer");
  ASR_CHECK_ARG(T > 0 && N > 0 && n_pad >= N && C >= 2, "beam: bad shape");
  ASR_CHECK_ARG(C <= kMaxC, "beam: at most %d classes", kMaxC);
  ASR_CHECK_ARG(beam_width >= 1 && beam_width <= 1024, "beam: width must be 1 .. 1024");
  ASR_CHECK_ARG((size_t)T * (size_t)beam_width * (size_t)(C - 1) < (size_t)INT_MAX,
                "beam: T * width * labels overflows the node ids");
  BeamParams p;
  p.logits = logits; p.seq_len = seq_len; p.T = T; p.n_pad = n_pad; p.C = C; p.W = beam_width;
  p.merge = merge_repeated ? 1 : 0;
  p.decoded = decoded; p.decoded_len = decoded_len; p.score = log_score;
  p.ws = reinterpret_cast<char*>(workspace);
  p.ws_per_utt = ws_per_utt(T, C, beam_width, &p.max_blocks);
  ASR_CHECK_ARG(ws_bytes >= p.ws_per_utt * (size_t)N, "beam: workspace too small");
  beam_kern_t k = pick(beam_width);
  const size_t shm = lds_bytes(beam_width);
  if (shm > 64 * 1024)
    ASR_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)shm));
  hipLaunchKernelGGL(k, dim3(N), dim3(kThreads), shm, (hipStream_t)stream, p);
  ASR_CHECK_LAUNCH();
  return ASR_OK;
}
