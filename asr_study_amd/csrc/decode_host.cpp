// K9 CTC prefix beam search and K10 edit distance -- host side of libasr_hip.so.
//
// Replaces core/ctc_utils.py:48-50 (tf.nn.ctc_beam_search_decoder, top_paths=1,
// merge_repeated=True by default) and core/metrics.py:8 (tf.edit_distance,
// normalize=True).  Beam search is branchy, tiny (28 classes, <= 400 beams) and
// data dependent: it runs on the host cores over logits the GPU produced, one
// utterance per thread (a device version is a later round's row).
//
// Algorithm (TensorFlow's CTCBeamSearchDecoder, restated from its published
// description): a prefix tree of beam entries, each with (blank, label, total)
// log-probabilities for the previous and the current frame; scores use the frame's
// max-subtracted logits (no log-softmax); existing entries are extended first
// (label path fed from the parent when the parent is still in the beam), then every
// entry that could still beat the beam's worst member grows children; a candidate
// enters a full beam only if strictly better than the current bottom, which is
// evicted.  merge_repeated collapses consecutive identical labels of the emitted
// path.  Arithmetic is double here (TF uses float): only exact ties can differ.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

namespace {

const double kLogZero = -std::numeric_limits<double>::infinity();

inline double lse(double a, double b) {
  if (a == kLogZero) return b;
  if (b == kLogZero) return a;
  const double m = a > b ? a : b;
  return m + std::log(std::exp(a - m) + std::exp(b - m));
}

struct Node {
  int parent, label;
  int children;          // index into the child table (C-1 slots), -1 = never expanded
  int order;             // creation rank AS IF all C-1 children were created when their
                         // parent first expanded (the tie-break of the eviction order)
  double ob, ol, ot, nb, nl, nt;
};

// The beam as a binary min-heap ordered worst-first: lowest total, newest entry first
// (the order TensorFlow's TopN with its BeamComparer evicts in).  An entry's total does
// not change while it sits in the heap, so a plain heap (no allocation per insert, unlike
// a std::set) gives exactly the same bottom element.
struct Beam {
  const std::vector<Node>* nodes;
  std::vector<int> heap;
  bool worse(int a, int b) const {              // a is evicted before b
    const double ta = (*nodes)[a].nt, tb = (*nodes)[b].nt;
    if (ta != tb) return ta < tb;
    return (*nodes)[a].order > (*nodes)[b].order;
  }
  size_t size() const { return heap.size(); }
  int bottom() const { return heap[0]; }
  void push(int v) {
    size_t i = heap.size();
    heap.push_back(v);
    while (i > 0) {
      const size_t par = (i - 1) / 2;
      if (!worse(heap[i], heap[par])) break;
      std::swap(heap[i], heap[par]);
      i = par;
    }
  }
  void pop() {
    heap[0] = heap.back();
    heap.pop_back();
    size_t i = 0;
    const size_t n = heap.size();
    for (;;) {
      size_t l = 2 * i + 1, r = l + 1, m = i;
      if (l < n && worse(heap[l], heap[m])) m = l;
      if (r < n && worse(heap[r], heap[m])) m = r;
      if (m == i) break;
      std::swap(heap[i], heap[m]);
      i = m;
    }
  }
};

void beam_one(const float* logits, size_t row_stride, int T, int C, int beam_width,
              bool merge_repeated, std::vector<int>* out, float* score) {
  const int blank = C - 1;
  std::vector<Node> nodes;
  std::vector<int> child_table;                  // (C-1) slots per expanded node, -1 = absent
  nodes.reserve((size_t)beam_width * 64 + 16);
  nodes.push_back(Node{-1, -1, -1, 0, kLogZero, kLogZero, kLogZero, 0.0, kLogZero, 0.0});
  int next_order = 1;
  std::vector<int> order_base;                   // per child-table block
  std::vector<int> leaves(1, 0), branches;
  std::vector<double> inp(C);
  Beam beam;
  beam.nodes = &nodes;
  for (int t = 0; t < T; ++t) {
    const float* x = logits + (size_t)t * row_stride;
    double mx = x[0];
    for (int c = 1; c < C; ++c) mx = std::max(mx, (double)x[c]);
    for (int c = 0; c < C; ++c) inp[c] = (double)x[c] - mx;
    branches = leaves;
    std::sort(branches.begin(), branches.end(), [&](int a, int b) {
      if (nodes[a].nt != nodes[b].nt) return nodes[a].nt > nodes[b].nt;
      return nodes[a].order < nodes[b].order;
    });
    for (int b : branches) {
      Node& e = nodes[b];
      e.ob = e.nb; e.ol = e.nl; e.ot = e.nt;
    }
    for (int b : branches) {
      Node& e = nodes[b];
      if (e.parent >= 0) {
        const Node& par = nodes[e.parent];
        if (par.nt != kLogZero) {
          const double prev = (e.label == par.label) ? par.ob : par.ot;
          e.nl = lse(e.nl, prev);
        }
        e.nl += inp[e.label];
      }
      e.nb = e.ot + inp[blank];
      e.nt = lse(e.nb, e.nl);
    }
    beam.heap.clear();
    for (int b : branches) beam.push(b);
    auto is_candidate = [&](double total) {
      return total > kLogZero &&
             ((int)beam.size() < beam_width || total > nodes[beam.bottom()].nt);
    };
    for (int b : branches) {
      if (!is_candidate(nodes[b].ot)) continue;
      if (nodes[b].children < 0) {               // children are created when they first
        nodes[b].children = (int)child_table.size();      // enter the beam, not before
        child_table.insert(child_table.end(), (size_t)(C - 1), -1);
        order_base.resize(child_table.size() / (C - 1), 0);
        order_base[nodes[b].children / (C - 1)] = next_order;
        next_order += C - 1;
      }
      const int slots = nodes[b].children;
      const double b_ob = nodes[b].ob, b_ot = nodes[b].ot;
      const int b_label = nodes[b].label;
      for (int c = 0; c < C - 1; ++c) {          // label ids 0 .. C-2 (blank is C-1)
        const int idx = child_table[slots + c];
        if (idx >= 0 && nodes[idx].nt != kLogZero) continue;   // already in the beam
        const double prev = (c == b_label) ? b_ob : b_ot;
        const double nl = prev == kLogZero ? kLogZero : inp[c] + prev;
        if (!is_candidate(nl)) {
          // TF resets the rejected child's OLD probabilities too: if that child is itself
          // a branch of this frame (evicted a moment ago), it must not expand later on
          if (idx >= 0) nodes[idx].ob = nodes[idx].ol = nodes[idx].ot = kLogZero;
          continue;
        }
        int id = idx;
        if (id < 0) {
          id = (int)nodes.size();
          nodes.push_back(Node{b, c, -1, order_base[slots / (C - 1)] + c, kLogZero, kLogZero,
                               kLogZero, kLogZero, kLogZero, kLogZero});
          child_table[slots + c] = id;
        }
        nodes[id].nb = kLogZero;
        nodes[id].nl = nl;
        nodes[id].nt = nl;
        if ((int)beam.size() == beam_width) {
          const int bottom = beam.bottom();
          beam.pop();
          nodes[bottom].nb = nodes[bottom].nl = nodes[bottom].nt = kLogZero;
        }
        beam.push(id);
      }
    }
    leaves = beam.heap;
  }
  int best = leaves[0];
  for (int b : leaves) {
    if (nodes[b].nt > nodes[best].nt ||
        (nodes[b].nt == nodes[best].nt && nodes[b].order < nodes[best].order))
      best = b;
  }
  out->clear();
  int prev = -1;
  for (int c = best; nodes[c].parent >= 0; c = nodes[c].parent) {
    if (!merge_repeated || nodes[c].label != prev) out->push_back(nodes[c].label);
    prev = nodes[c].label;
  }
  std::reverse(out->begin(), out->end());
  if (score) *score = (float)nodes[best].nt;
}

int levenshtein(const int* a, int la, const int* b, int lb) {
  std::vector<int> prev(lb + 1), cur(lb + 1);
  for (int j = 0; j <= lb; ++j) prev[j] = j;
  for (int i = 1; i <= la; ++i) {
    cur[0] = i;
    for (int j = 1; j <= lb; ++j) {
      const int sub = prev[j - 1] + (a[i - 1] != b[j - 1] ? 1 : 0);
      cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), sub);
    }
    std::swap(prev, cur);
  }
  return prev[lb];
}

}  // namespace

extern "C" int asr_ctc_beam_search_host(const float* logits_host, const int* seq_len_host,
                                        int T, int N, int n_pad, int C, int beam_width,
                                        int merge_repeated, int* decoded, int* decoded_len,
                                        float* log_score) {
  ASR_CHECK_ARG(logits_host && seq_len_host && decoded && decoded_len, "beam: null pointer");
  ASR_CHECK_ARG(T > 0 && N > 0 && n_pad >= N && C >= 2 && beam_width >= 1, "beam: bad shape");
  const size_t row_stride = (size_t)n_pad * C;
  unsigned hw = std::thread::hardware_concurrency();
  int nthreads = (int)std::min<unsigned>(hw ? hw : 1, (unsigned)N);
  if (nthreads < 1) nthreads = 1;
  auto work = [&](int tid) {
    std::vector<int> path;
    for (int n = tid; n < N; n += nthreads) {
      int Tn = seq_len_host[n];
      Tn = Tn < 0 ? 0 : (Tn > T ? T : Tn);
      float sc = 0.f;
      beam_one(logits_host + (size_t)n * C, row_stride, Tn, C, beam_width, merge_repeated != 0,
               &path, &sc);
      const int L = (int)path.size();
      for (int i = 0; i < T; ++i) decoded[(size_t)n * T + i] = i < L ? path[i] : -1;
      decoded_len[n] = L;
      if (log_score) log_score[n] = sc;
    }
  };
  if (nthreads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nthreads; ++i) pool.emplace_back(work, i);
    for (auto& th : pool) th.join();
  }
  return ASR_OK;
}

extern "C" int asr_edit_distance_host(const int* hyp, const int* hyp_len, int hyp_ld,
                                      const int* truth, const int* truth_len, int truth_ld,
                                      int N, float* out_normalized) {
  ASR_CHECK_ARG(hyp && hyp_len && truth && truth_len && out_normalized && N > 0,
                "edit_distance: bad arguments");
  for (int n = 0; n < N; ++n) {
    const int la = hyp_len[n], lb = truth_len[n];
    const int d = levenshtein(hyp + (size_t)n * hyp_ld, la, truth + (size_t)n * truth_ld, lb);
    if (lb == 0) out_normalized[n] = d == 0 ? 0.f : std::numeric_limits<float>::infinity();
    else out_normalized[n] = (float)d / (float)lb;
  }
  return ASR_OK;
}
