// Operation-level entry points of the C ABI (host code): each names one role on the
// training path -- feature batch, gate projection forward / data gradient / weight
// gradient, clipped optimiser step, beam search, edit distance, all-reduce -- and is
// expressed on the general kernels' entry points (asr_frontend_features, asr_gemm,
// asr_colsum, asr_grad_norm + asr_*_step, ...).  No kernels here.
#include "common.h"

// ---------------------------------------------------------------- K1-K3 front-end
extern "C" int asr_frontend_mfcc_batch(const asr_frontend_cfg* cfg, const float* audio,
                                       const int* offsets, const int* lengths,
                                       const int* host_lengths, int n_utt, int n_pad,
                                       const double* window, const double* mel,
                                       const int* mel_range, const double* dct, float* out,
                                       int t_out, int* out_frames, void* workspace,
                                       size_t ws_bytes, asr_stream_t stream) {
  ASR_CHECK_ARG(cfg && cfg->kind == 0, "asr_frontend_mfcc_batch: cfg->kind must be 0 (MFCC)");
  return asr_frontend_features(cfg, audio, offsets, lengths, host_lengths, n_utt, n_pad, window,
                               mel, mel_range, dct, out, t_out, out_frames, workspace, ws_bytes,
                               stream);
}

extern "C" int asr_frontend_logfbank_batch(const asr_frontend_cfg* cfg, const float* audio,
                                           const int* offsets, const int* lengths,
                                           const int* host_lengths, int n_utt, int n_pad,
                                           const double* window, const double* mel,
                                           const int* mel_range, float* out, int t_out,
                                           int* out_frames, void* workspace, size_t ws_bytes,
                                           asr_stream_t stream) {
  ASR_CHECK_ARG(cfg && cfg->kind == 1,
                "asr_frontend_logfbank_batch: cfg->kind must be 1 (LogFbank)");
  return asr_frontend_features(cfg, audio, offsets, lengths, host_lengths, n_utt, n_pad, window,
                               mel, mel_range, nullptr, out, t_out, out_frames, workspace,
                               ws_bytes, stream);
}

// ---------------------------------------------------------------- K4/K6 gate GEMMs
static int check_gate(const asr_gate_gemm_args* g, const char* who) {
  ASR_CHECK_ARG(g, "%s: null args", who);
  ASR_CHECK_ARG(g->rows > 0 && g->in_dim > 0 && g->gate_dim > 0, "%s: empty problem", who);
  ASR_CHECK_ARG(g->W && g->ldw >= g->gate_dim, "%s: W / ldw", who);
  ASR_CHECK_ARG(!g->mask_w || (g->n_pad > 0 && g->rows % g->n_pad == 0),
                "%s: a mask needs rows to be whole frames of n_pad samples", who);
  return ASR_OK;
}

static void base_gemm(asr_gemm_args* a, const asr_gate_gemm_args* g) {
  *a = asr_gemm_args{};
  a->alpha = 1.0f;
  a->precision = g->precision;
}

static void fwd_args(asr_gemm_args* a, const asr_gate_gemm_args* g) {
  base_gemm(a, g);
  a->M = g->rows; a->N = g->gate_dim; a->K = g->in_dim;
  a->A = g->x; a->lda = g->ldx;
  a->B = g->W; a->ldb = g->ldw;
  a->C = g->zx; a->ldc = g->ldz;
  a->bias = g->bias;
  a->a_scale = g->mask_w; a->a_scale_period = g->n_pad; a->a_scale_ld = g->in_dim;
}

static void dgrad_args(asr_gemm_args* a, const asr_gate_gemm_args* g) {
  base_gemm(a, g);
  a->M = g->rows; a->N = g->in_dim; a->K = g->gate_dim;
  a->A = g->dz; a->lda = g->ldz;
  a->B = g->W; a->ldb = g->ldw; a->trans_b = 1;
  a->C = g->dx; a->ldc = g->ldx;
  a->beta = g->dx_beta;
  a->c_scale = g->mask_w; a->c_scale_period = g->n_pad; a->c_scale_ld = g->in_dim;
  a->a_absmax = g->dz_absmax;
}

static void wgrad_args(asr_gemm_args* a, const asr_gate_gemm_args* g) {
  base_gemm(a, g);
  a->M = g->in_dim; a->N = g->gate_dim; a->K = g->rows;
  a->A = g->x; a->lda = g->ldx; a->trans_a = 1;
  a->B = g->dz; a->ldb = g->ldz;
  a->C = g->dW; a->ldc = g->ldw;
  a->a_scale = g->mask_w; a->a_scale_period = g->n_pad; a->a_scale_ld = g->in_dim;
  a->split_k = g->split_k;
  a->b_absmax = g->dz_absmax;
}

extern "C" size_t asr_gemm_gate_workspace_bytes(const asr_gate_gemm_args* g, int role) {
  if (!g) return 0;
  asr_gemm_args a;
  if (role == 0) {
    fwd_args(&a, g);
    return asr_gemm_workspace_bytes(&a);
  }
  if (role == 1) {
    dgrad_args(&a, g);
    return asr_gemm_workspace_bytes(&a);
  }
  wgrad_args(&a, g);
  size_t gemm_ws = asr_align_up(asr_gemm_workspace_bytes(&a), 256);
  return gemm_ws + (g->db ? asr_colsum_workspace_bytes(g->rows, g->gate_dim) : 0);
}

extern "C" int asr_gemm_gate_fwd(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                                 asr_stream_t stream) {
  int rc = check_gate(g, "asr_gemm_gate_fwd");
  if (rc) return rc;
  ASR_CHECK_ARG(g->x && g->zx, "asr_gemm_gate_fwd: x / zx");
  asr_gemm_args a;
  fwd_args(&a, g);
  return asr_gemm(&a, workspace, ws_bytes, stream);
}

extern "C" int asr_gemm_gate_dgrad(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                                   asr_stream_t stream) {
  int rc = check_gate(g, "asr_gemm_gate_dgrad");
  if (rc) return rc;
  ASR_CHECK_ARG(g->dz && g->dx, "asr_gemm_gate_dgrad: dz / dx");
  asr_gemm_args a;
  dgrad_args(&a, g);
  return asr_gemm(&a, workspace, ws_bytes, stream);
}

extern "C" int asr_gemm_gate_wgrad(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                                   asr_stream_t stream) {
  int rc = check_gate(g, "asr_gemm_gate_wgrad");
  if (rc) return rc;
  ASR_CHECK_ARG(g->x && g->dz && g->dW, "asr_gemm_gate_wgrad: x / dz / dW");
  asr_gemm_args a;
  wgrad_args(&a, g);
  size_t gemm_ws = asr_align_up(asr_gemm_workspace_bytes(&a), 256);
  size_t cs_ws = g->db ? asr_colsum_workspace_bytes(g->rows, g->gate_dim) : 0;
  if (ws_bytes < gemm_ws + cs_ws) {
    asr_set_error("asr_gemm_gate_wgrad: workspace %zu < %zu", ws_bytes, gemm_ws + cs_ws);
    return ASR_ERR_WORKSPACE;
  }
  rc = asr_gemm(&a, workspace, gemm_ws, stream);
  if (rc || !g->db) return rc;
  return asr_colsum(g->dz, g->rows, g->gate_dim, g->ldz, g->db, 0.0f,
                    (char*)workspace + gemm_ws, cs_ws, stream);
}

// ---------------------------------------------------------------- K9 / K10 host ops
extern "C" int asr_ctc_beam(const float* logits_host, const int* seq_len_host, int T, int N,
                            int n_pad, int C, int beam_width, int merge_repeated, int* decoded,
                            int* decoded_len, float* log_score) {
  return asr_ctc_beam_search_host(logits_host, seq_len_host, T, N, n_pad, C, beam_width,
                                  merge_repeated, decoded, decoded_len, log_score);
}

extern "C" int asr_edit_distance(const int* hyp, const int* hyp_len, int hyp_ld, const int* truth,
                                 const int* truth_len, int truth_ld, int N,
                                 float* out_normalized) {
  return asr_edit_distance_host(hyp, hyp_len, hyp_ld, truth, truth_len, truth_ld, N,
                                out_normalized);
}

// ---------------------------------------------------------------- K11 clipped steps
extern "C" int asr_clip_adam_step(float* params, const float* grads, float* m, float* v,
                                  int64_t n, const asr_segment* segments_dev, int n_seg,
                                  double* norm_out, float clipnorm, float lr, float beta1,
                                  float beta2, float eps, int step, void* workspace,
                                  size_t ws_bytes, asr_stream_t stream) {
  int rc = asr_grad_norm(params, grads, n, segments_dev, n_seg, norm_out, workspace, ws_bytes,
                         stream);
  if (rc) return rc;
  return asr_adam_step(params, grads, m, v, n, segments_dev, n_seg, norm_out, clipnorm, lr, beta1,
                       beta2, eps, step, stream);
}

extern "C" int asr_clip_sgd_step(float* params, const float* grads, float* vel, int64_t n,
                                 const asr_segment* segments_dev, int n_seg, double* norm_out,
                                 float clipnorm, float lr, float momentum, void* workspace,
                                 size_t ws_bytes, asr_stream_t stream) {
  int rc = asr_grad_norm(params, grads, n, segments_dev, n_seg, norm_out, workspace, ws_bytes,
                         stream);
  if (rc) return rc;
  return asr_sgd_step(params, grads, vel, n, segments_dev, n_seg, norm_out, clipnorm, lr, momentum,
                      stream);
}

// ---------------------------------------------------------------- C1
extern "C" int asr_comm_allreduce(asr_comm_t comm, float* buf, int64_t n, asr_stream_t stream) {
  return asr_comm_allreduce_sum(comm, buf, n, stream);
}
