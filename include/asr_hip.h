/* asr_hip.h -- C ABI of libasr_hip.so: the MI355X (gfx950) kernels behind the
 * asr-study acoustic-training hot path (MFCC/log-mel -> BiLSTM stack -> CTC).
 *
 * The reference (igormq/asr-study) has NO native/FFI boundary: every op on this
 * path is a Keras-1.2.2 / TensorFlow-1.3.0 / NumPy call made from Python.  Each
 * entry point below therefore cites the reference call site whose arithmetic it
 * replaces (paths relative to the reference checkout); INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; device pointers are raw addresses (from
 *    torch.Tensor.data_ptr() on the Python side); no torch types anywhere.
 *  - the CALLER owns all memory.  Scratch space is sized by the matching
 *    *_workspace_bytes() query and passed in; the library never allocates
 *    persistent device memory.
 *  - all work is enqueued on the given hipStream_t (passed as void*); no entry
 *    point synchronises the device unless its comment says so.
 *  - every function returns 0 on success or a negative asr_status; the message
 *    is available from asr_last_error() (thread-local).
 *  - activations are TIME-MAJOR slabs (T, Npad, feat) float32, Npad = batch
 *    rounded up to a multiple of 16 (rows n >= N are padding and carry zeros).
 *  - fused gate layout: every (.., 4H) gate axis is ordered unit-major,
 *    gate-minor: column = unit*4 + gate, gate in {0:i, 1:f, 2:c, 3:o}
 *    (Keras' consume_less='gpu' layout is gate-major: column = gate*H + unit;
 *    core/layers.py:447-450.  The Python host converts at the checkpoint edge.)
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* asr_stream_t; /* hipStream_t */

enum asr_status {
  ASR_OK = 0,
  ASR_ERR_INVALID = -1,   /* bad argument / unsupported shape            */
  ASR_ERR_WORKSPACE = -2, /* workspace too small                         */
  ASR_ERR_LAUNCH = -3,    /* HIP launch or runtime error                 */
  ASR_ERR_TIMEOUT = -4,   /* persistent kernel gave up on a bounded spin */
  ASR_ERR_RESIDENCY = -5  /* persistent grid would not be co-resident    */
};

const char* asr_last_error(void);
/* ABI version of THIS header: bumped whenever an argument struct grows or changes meaning (no
 * struct carries a size field).  A caller checks asr_version() == ASR_HIP_ABI_VERSION right
 * after loading the library (asr_study_amd/_lib.py does) and refuses a mismatch.
 * 100: rounds 1-4.  105: asr_lstm_args +compact +activation +fwd_units, asr_pack_args +mask2
 * +r2_hl, asr_lstm_ln_args +activation.  106: asr_lstm_args +dz_hl +dz_bound +dz_scale_out. */
#define ASR_HIP_ABI_VERSION 106
int asr_version(void);
/* Device facts the host needs for sizing persistent grids (CU count etc). */
int asr_device_info(int* num_cus, int* lds_bytes_per_cu, char* arch, int arch_len);

/* ------------------------------------------------------------------------ */
/* K1-K3  Front-end.  Replaces preprocessing/audio.py:223-253 (FBank._call:  */
/* preemphasis, framing, Hamming, |rFFT|^2/nfft, energy, mel filterbank),    */
/* :339-367 (MFCC._call: log, DCT-II ortho, lifter, c0<-log energy, deltas), */
/* :404-442 (LogFbank._call), :77-150 (_postprocessing: stride / context)    */
/* and :70-75 (_standarize) together with preprocessing/audio_utils.py       */
/* :17-50,:98-120,:143-173.                                                  */
/* ------------------------------------------------------------------------ */
typedef struct asr_frontend_cfg {
  int kind;          /* 0 = MFCC, 1 = LogFbank                               */
  int frame_len;     /* round_half_up(win_len*fs)   (400)                    */
  int frame_step;    /* round_half_up(win_step*fs)  (160)                    */
  int nfft;          /* 512 (must be 512 in this build)                      */
  int num_filt;      /* 40 / 80 (<= 128)                                     */
  int num_cep;       /* 13 (MFCC only, <= 64)                                */
  int append_energy; /* MFCC: c0 <- log(energy+eps); LogFbank: extra column  */
  int d, dd;         /* append deltas / delta-deltas                         */
  int stride;        /* keep every stride-th frame                           */
  int num_context;   /* +-context frame stacking (zero rows off the edges)   */
  int mean_norm, var_norm;
  int reserved;      /* (keeps the doubles 8-byte aligned)                   */
  double pre_emph;   /* 0.97 -- float64: the per-frame chain runs in float64 */
  double eps;        /* 1e-8                                                 */
} asr_frontend_cfg;

/* Number of frames for a signal of `samples` samples (audio_utils.py:29-32). */
int asr_frontend_num_frames(int samples, int frame_len, int frame_step);
/* Feature columns produced per frame BEFORE context stacking ((1+d+dd)*base). */
int asr_frontend_num_feats(const asr_frontend_cfg* cfg);
size_t asr_frontend_workspace_bytes(const asr_frontend_cfg* cfg, int n_utt,
                                    int max_frames);
/* audio: concatenated float32 samples; offsets[i]/lengths[i] (device int32)
 * delimit utterance i.  window (frame_len), mel (num_filt x (nfft/2+1), dense
 * row-major, from audio.py:255-277 computed on the host in float64), mel_range
 * (num_filt x {first, last+1} non-zero bin of each filter, device int32), dct
 * (num_filt x num_cep: scipy DCT-II 'ortho' with the lifter folded in; MFCC
 * only) are FLOAT64 device tables (the reference computes in float64; see
 * csrc/frontend.hip).  out is the float32 (T_out, n_pad, F_out) time-major
 * slab (row stride n_pad*F_out), zero-filled past each utterance's frames
 * (pad_sequences(padding='post'), datasets/dataset_generator.py:227);
 * out_frames[i] (device int32) receives the frame count after striding.
 * host_lengths mirrors `lengths` on the host (grid sizing; no device sync). */
int asr_frontend_features(const asr_frontend_cfg* cfg, const float* audio,
                          const int* offsets, const int* lengths,
                          const int* host_lengths, int n_utt, int n_pad,
                          const double* window, const double* mel,
                          const int* mel_range, const double* dct, float* out,
                          int t_out,
                          int* out_frames, void* workspace, size_t ws_bytes,
                          asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K4/K6  fp32 MFMA GEMM.  Replaces K.dot(x*B_W, W) (core/layers.py:439,     */
/* hoisted out of the time loop), TimeDistributed(Dense) (core/models.py     */
/* :278-279) and their tf.gradients.                                         */
/*   C[M,N] = c_scale (.) (alpha * (opA(A) (.) a_scale) @ opB(B) + bias[N])  */
/*            + beta * C.    trans_a: A is stored (K,M) row-major; trans_b: B */
/*   is stored (N,K) row-major.  a_scale / c_scale: optional (period, M or K */
/*   / N) variational-dropout masks indexed by (row % period); pass NULL.    */
/* split_k > 1 needs workspace of split_k*M*N floats (deterministic reduce). */
/* ------------------------------------------------------------------------ */
typedef struct asr_gemm_args {
  int M, N, K;
  int trans_a, trans_b;
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  float alpha, beta;
  const float* bias;       /* (N) added to every row, or NULL                */
  const float* a_scale;    /* (period x Kdim-or-Mdim) mask on A rows or NULL */
  int a_scale_period;      /* row index modulo this selects the mask row     */
  int a_scale_ld;
  const float* c_scale;    /* (period x N) mask on C rows, or NULL           */
  int c_scale_period;
  int c_scale_ld;
  int split_k;             /* 0/1 = none                                     */
  /* arithmetic: 0 = exact fp32 MFMA; 1 = split-fp16 MFMA (hi + lo/2048, 22-bit   */
  /* mantissa, fp32 accumulate); -1 = library default (env ASR_GEMM_PREC, 1).    */
  int precision;
  /* split-fp16 only: device floats holding max|A| / max|B| (asr_absmax) used to */
  /* pick a power-of-two pre-scale so tiny gradients stay in fp16 range; NULL=1. */
  const float* a_absmax;
  const float* b_absmax;
} asr_gemm_args;
size_t asr_gemm_workspace_bytes(const asr_gemm_args* a);
int asr_gemm(const asr_gemm_args* a, void* workspace, size_t ws_bytes,
             asr_stream_t stream);
/* ------------------------------------------------------------------------ */
/* K4/K6 with operands converted ONCE: packed split-fp16 planes.  asr_pack_hl  */
/* turns an fp32 matrix (rows, cols) -- optionally times a variational-dropout */
/* mask[(row % period), col] (core/models.py:265-266), applied in fp32 before   */
/* the split -- into two fp16 planes hi = fp16(x*s), lo = fp16(x*s - hi) with   */
/* s = the power of two that maps *absmax into [2^8, 2^9) (1 if absmax is       */
/* NULL), in either or both orientations:                                       */
/*   r planes (rows, ldk_r): the reduction index of the later GEMM = columns,   */
/*   c planes (cols, ldk_c): the reduction index = rows (weight gradients);     */
/* ldk_* = the extent rounded up to a multiple of 32 halfs, zero padded.  The   */
/* scale is written to *scale_out (device float).  asr_gemm_hl then computes    */
/*   C[M,N] = alpha/(sa*sb) * (Ah Bh^T + Ah Bl^T + Al Bh^T) (+bias) (*c_scale)  */
/*            + beta*C                                                          */
/* from A planes (M, lda) and B planes (N, ldb), reduction index contiguous in  */
/* both, fp32 accumulation (2^-22 relative per product: the arithmetic of       */
/* asr_gemm precision 1 without its per-tile conversions).  A sub-matrix is a   */
/* pointer offset (64-byte aligned: a whole (hi, lo) group) with the same       */
/* leading dimension; K % 8 == 0; lda % 16 == 0 and ldb % 16 == 0.              */
/* ------------------------------------------------------------------------ */
typedef struct asr_pack_args {
  const float* src; int rows, cols, ld;
  const float* mask; int mask_period, mask_ld;   /* or NULL; row period (any n_pad)   */
  const float* absmax;                           /* device float, or NULL (scale 1)   */
  float* scale_out;                              /* device float, or NULL             */
  void* r_hl; int ldk_r;                         /* (rows, 2 ldk_r) halfs, or NULL    */
  void* c_hl; int ldk_c;                         /* (cols, 2 ldk_c) halfs, or NULL    */
  /* optional: a SECOND set of row planes r2_hl (geometry of r_hl) of the same source under   */
  /* mask2 (period / ld of mask) -- the two directions' dropout masks of a BiLSTM input, the  */
  /* source read once; both sets share the scale.  NULL = one set.                            */
  const float* mask2; void* r2_hl;
} asr_pack_args;
int asr_pack_hl(const asr_pack_args* a, asr_stream_t stream);
/* Debug: arm (enable != 0) / read the K-loop phase profile of asr_gemm_hl's workgroup 0:      */
/* out64 = 8 waves x 8 phases of shader clocks (fragment reads, barrier, MFMAs first half,    */
/* MFMAs second half, barrier, prologue, the K loop in 100 MHz real-time ticks, unused);      */
/* NULL to only arm / disarm.                                                                 */
int asr_gemm_hl_profile(int enable, long long* out64, asr_stream_t stream);
typedef struct asr_gemm_hl_args {
  int M, N, K;
  const void* a_hl; int lda;                     /* interleaved planes; reduction     */
  const void* b_hl; int ldb;                     /* indices per row (a row = 2 ld halfs) */
  const float* a_scale; const float* b_scale;    /* device floats (asr_pack_hl) or NULL */
  float* C; int ldc;
  float alpha, beta;
  const float* bias;                             /* (N) or NULL                       */
  const float* c_scale; int c_scale_period; int c_scale_ld;   /* mask on C rows or NULL */
  int split_k;                                   /* 0/1 = none                        */
  int tile;                                      /* 0 = library default (256 x 256 for  */
                                                 /* outputs that large), 128 = the      */
                                                 /* 128 x 128 kernel: 4 waves, 80 KB    */
                                                 /* LDS, co-resident with a recurrent   */
                                                 /* workgroup that reserves no LDS      */
  int k_major;                                   /* != 0: C = A^T B from planes whose   */
                                                 /* ROWS are the reduction index: a_hl   */
                                                 /* (K rows, lda >= M columns), b_hl (K  */
                                                 /* rows, ldb >= N columns) -- x^T dz,   */
                                                 /* h^T dz on the planes x@W / dz@W^T    */
                                                 /* use (no second orientation packed);  */
                                                 /* a sub-matrix = pointer to (first row,*/
                                                 /* first column group of 16)            */
  /* Segmented reduction range of A (row-major form only; 0 = none): K = n * a_seg_k, n <= 16,  */
  /* a_seg_k % 32 == 0; reduction indices [i a_seg_k, (i+1) a_seg_k) are read from columns      */
  /* [0, a_seg_k) of the planes a_seg_row[i] (>= 0) rows further down: C = sum_i A_i B_i^T with */
  /* A_i = a_hl shifted by a_seg_row[i] rows -- the implicit im2col over time of asr_conv2d_*   */
  /* (tap i of a filter reads the activation slab i frames on).  lda >= a_seg_k.                 */
  int a_seg_k;
  long long a_seg_row[16];
  /* Batch (k_major form only; 0 / 1 = none): `batch` <= 16 GEMMs that share B in one launch,  */
  /* C_b = A_b^T B with A_b = a_hl shifted by a_batch_row[b] (>= 0) plane rows and C_b = C +   */
  /* b * M * ldc (the members' outputs are consecutive row blocks); no beta / bias / mask.     */
  /* The per-tap weight gradients of asr_conv2d_wgrad.  split_k applies to every member; the   */
  /* workspace is split_k * batch * M * N floats.                                              */
  int batch;
  long long a_batch_row[16];
  /* > 0 (segmented form only; beta = 0, no mask): C = min(max(alpha A B^T + bias, 0), clamp_hi) */
  /* -- the clipped ReLU of asr_conv2d_fwd applied where the tile is still in registers.         */
  float clamp_hi;
} asr_gemm_hl_args;
size_t asr_gemm_hl_workspace_bytes(const asr_gemm_hl_args* a);
int asr_gemm_hl(const asr_gemm_hl_args* a, void* workspace, size_t ws_bytes,
                asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K13  2-D convolution front-end of BASELINE.json configs[2] ("DeepSpeech2-   */
/* style 5xBiLSTM(512) + 2 conv front-end").  NO REFERENCE COUNTERPART: the    */
/* reference lists Deep Speech 2 as TODO (README.md:118) and its deep_speech   */
/* factory (core/models.py:148-214) is dead code without convolutions; the op  */
/* is defined here as Keras-1.2.2 Convolution2D(nb_filter, nb_row, nb_col,     */
/* subsample=(st, sf), border_mode='same', dim_ordering='tf') on the           */
/* (N, T, F, C) view of the feature slab, followed by a clipped ReLU           */
/* min(max(z, 0), clip) (the reference's clipped_relu, core/models.py:116).    */
/* 'same' = TensorFlow's rule: out = ceil(in / stride), pad_total =            */
/* max((out - 1) stride + k - in, 0), pad_before = pad_total / 2.              */
/*                                                                              */
/* Layout: activations are time-major slabs (T, n_pad, F*C), channel minor     */
/* (feature index f*C + c) -- Reshape((T, F*C)) of the Keras tensor is the     */
/* identity; W is (kt, kf, C_in, C_out) row-major (Keras 'tf' kernel), bias    */
/* (C_out).  The convolution over FREQUENCY is folded into a banded matrix     */
/* per time tap (band[dt][(fi,ci)][(fo,co)] = W[dt][fi - sf fo + pf][ci][co]), */
/* the convolution over TIME is the segmented reduction of asr_gemm_hl: tap dt */
/* reads the packed activation planes dt frames further on (implicit im2col);  */
/* the products run on the packed split-fp16 MFMA kernel of K4.  kt <= 16.     */
/*   fwd  : z = conv(x, W) + b   (kept for the backward pass),  y = act(z)     */
/*   dgrad: dx = conv^T(dy (.) act'(z), W)                                      */
/*   wgrad: dW = x (*) (dy (.) act'(z)),  db = sum over rows and fo            */
/* The workspace keeps the packed planes of x (written by fwd) and of           */
/* dz = dy (.) act'(z) (written by whichever of dgrad / wgrad runs first);     */
/* reuse_x / reuse_dz = 1 tell a later call on the SAME workspace to skip the  */
/* pack (the host runs fwd ... dgrad, wgrad per layer with its own workspace). */
/* ------------------------------------------------------------------------ */
typedef struct asr_conv2d_args {
  int T_in, n_pad, F_in, C_in;   /* input slab (T_in, n_pad, F_in*C_in)              */
  int C_out, kt, kf, st, sf;     /* filter and strides (time, frequency)            */
  float clip;                    /* clipped-ReLU ceiling (20); <= 0: linear output  */
  const float* x;                /* fwd, wgrad                                      */
  const float* W;                /* (kt, kf, C_in, C_out)                           */
  const float* bias;             /* (C_out)                                         */
  float* z;                      /* (T_out, n_pad, F_out*C_out): fwd out, bwd in.   */
                                 /* fwd with clip > 0: NULL (or == y) fuses the     */
                                 /* clipped ReLU into the GEMM epilogue -- only y   */
                                 /* is written; bwd then takes y here (the mask     */
                                 /* 0 < z < clip reads the same from y)             */
  float* y;                      /* fwd out (may be NULL when clip <= 0: y = z)     */
  const float* dy;               /* dgrad / wgrad in, shape of z                    */
  float* dx;                     /* dgrad out, shape of x                           */
  float* dW;                     /* wgrad out, shape of W                           */
  float* db;                     /* wgrad out (C_out)                               */
  int reuse_x, reuse_dz;
  const float* x_absmax;         /* device float >= max|x| (the previous layer's    */
                                 /* clip), or NULL: measured with asr_absmax        */
} asr_conv2d_args;
/* Streams: the block GEMMs of one forward / dgrad call may be issued on two streams of the   */
/* library's own beside `stream` (forked behind everything enqueued on it, joined before the   */
/* call returns): to the caller the call is ordered on `stream` like any other.              */
/* T_out = ceil(T_in / st), F_out = ceil(F_in / sf). */
int asr_conv2d_out_shape(const asr_conv2d_args* a, int* T_out, int* F_out);
size_t asr_conv2d_workspace_bytes(const asr_conv2d_args* a);
int asr_conv2d_fwd(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                   asr_stream_t stream);
int asr_conv2d_dgrad(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                     asr_stream_t stream);
int asr_conv2d_wgrad(const asr_conv2d_args* a, void* workspace, size_t ws_bytes,
                     asr_stream_t stream);

/* out[0] = max |x[i]| over a 16-byte aligned flat tensor (HBM-bound).         */
int asr_absmax(const float* x, int64_t n, float* out, asr_stream_t stream);
/* out[n] = beta*out[n] + sum_m X[m, n]  (bias gradients; X read once,       */
/* float64 accumulation, fixed-order two-stage reduce).                      */
size_t asr_colsum_workspace_bytes(int M, int N);
int asr_colsum(const float* X, int M, int N, int ldx, float* out, float beta,
               void* workspace, size_t ws_bytes, asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K5  Recurrent LSTM sequence kernels (both directions of one Bidirectional */
/* layer per call).  Replace core/layers.py:432-469 (LSTM.step) iterated by  */
/* Keras K.rnn, and its BPTT.  Gate math: i,f,o = hard_sigmoid = clip(.2x+.5, */
/* 0,1); c = f*c + i*tanh(z_c); h = o*tanh(c); h0 = c0 = 0; the reverse      */
/* direction walks the PADDED slab from T-1 down to 0 (no Masking).          */
/* ------------------------------------------------------------------------ */
typedef struct asr_lstm_args {
  int T, n_pad, H;     /* n_pad % 16 == 0, H % 4 == 0                        */
  int mode;            /* 0 = persistent (default), 1 = one launch per step  */
  const float* U;      /* (2, H, 4H)  [dir][k][unit*4+gate]                  */
  const float* mask_u; /* (2, n_pad, H) variational mask B_U or NULL         */
  /* forward: zx (T, n_pad, 2, 4H) = x@W+b precomputed; outputs y (T, n_pad, */
  /* 2H) = [h_fwd | h_bwd], cell (T, n_pad, 2, H), gates (T, n_pad, 2, 4H)   */
  /* post-activation (kept for BPTT).                                        */
  const float* zx;
  float* y;
  float* cell;
  float* gates;
  /* backward: dy (T, n_pad, 2H) in; dz (T, n_pad, 2, 4H) out.               */
  const float* dy;
  float* dz;
  /* backward, optional: receives max |dz| (one device float; used as the     */
  /* split-fp16 GEMM pre-scale of the gradient operand), or NULL.             */
  float* dz_absmax;
  /* Optional step range: process recurrence steps [step_begin, step_begin +  */
  /* step_count) only (step s is frame s of the forward direction and frame   */
  /* T-1-s of the backward one; BPTT walks them in the opposite order).  A    */
  /* sequence is processed by calls with consecutive ranges starting at 0 on  */
  /* the SAME workspace and stream order; step_count = 0 means all T steps.   */
  /* Lets the caller overlap the GEMMs that feed / consume the outer frames   */
  /* with the recurrence over the inner ones.                                 */
  int step_begin, step_count;
  /* Optional cell variants of the reference's LSTM override (core/layers.py:432-469);  */
  /* all NULL = the plain cell.  Frames are indexed by t (slab row), not by step.       */
  /* mi: multiplicative integration z = alpha*Wx*Uh + beta1*Uh + beta2*Wx + b           */
  /*     (:441-443): (2, 4, 4H) = per direction alpha, beta1, beta2, b (unit-major,     */
  /*     gate-minor); zx must then be x@W WITHOUT the bias.  uh (T, n_pad, 2, 4H)       */
  /*     receives h_prev@U in the forward call and is read back by the backward call,   */
  /*     which also needs wx (= the forward zx) and writes dwx (T, n_pad, 2, 4H) =      */
  /*     d/d(x@W) -- dz then holds d/d(h_prev@U) -- and dmi (n_pad/16, 2, 4, 4H), the   */
  /*     per-batch-tile sums of d alpha, d beta1, d beta2, d b (sum over axis 0).       */
  /* zone_c / zone_h: zoneout (:457-467), new = prev + k*(candidate - prev) with the    */
  /*     coefficient k (T, 2, H): the per-frame keep mask in training, 1 - level at     */
  /*     test time.                                                                     */
  const float* mi;
  float* uh;
  const float* zone_c;
  const float* zone_h;
  const float* wx;
  float* dwx;
  float* dmi;
  /* backward, optional: db_part (n_pad/16, 2, 4H) receives, per batch tile, the sum of dz    */
  /* over the tile's samples and over all steps (+= across the slices of a sequence): the     */
  /* bias gradient is its sum over axis 0, so no pass over the dz slab is needed for it.      */
  /* With mi set the same sums are dmi[:, :, 3].                                              */
  float* db_part;
  /* forward, optional: number of real rows of the batch (0 = all n_pad).  n_valid == 1    */
  /* (predict.py decodes one utterance per call) selects a tile-free exact-fp32 kernel     */
  /* that reads and writes ROW 0 of the slabs only; the padding rows are left untouched.   */
  int n_valid;
  /* LDS a recurrent workgroup reserves, in KB (0 = 96: alone on its CU beside 80 KB GEMM    */
  /* workgroups).  80 = two recurrent workgroups per CU, for launches on a stream confined   */
  /* to half of the CUs (asr_stream_create_cu_mask): the recurrence is latency-bound, so two */
  /* chains interleave on one CU while the other half of the chip runs GEMMs.                */
  int lds_reserve_kb;
  /* backward, optional: 1 = the COMPACT launch geometry of the plain cell at H = 256 / 512   */
  /* in persistent mode: a chain is H/32 workgroups instead of H/16 (each owns twice the      */
  /* outputs), so a layer occupies half as many CUs (128 of 256 at H = 512, 64 utterances) at */
  /* a longer step; the caller runs the weight-gradient GEMMs of the layer above on the CUs   */
  /* left free.  Gate gradients are bit-identical to the default geometry.  Ignored where the */
  /* compact kernel does not exist (other H, cell variants, stepwise mode, forward).          */
  int compact;
  /* The reference LSTM's `activation` hyper-parameter (core/layers.py:452, :463: cell         */
  /* candidate g = act(z_c), output h = o * act(c); brsmv1 passes it on, core/models.py:220):  */
  /* 0 tanh (default), 1 relu, 2 sigmoid, 3 hard_sigmoid, 4 linear, 5 softsign, 6 softplus     */
  /* (Keras-1.2.2 names).  Anything but 0 runs on the variant kernels (as mi / zoneout do).    */
  int activation;
  /* forward, optional: units per workgroup of the wide kernels -- 0 = the library decides     */
  /* (eight at H = 256 where the layer then leaves half of the CUs free), 8 / 16 = force.      */
  /* Slices of one sequence may differ (same exchange layout): the host runs the slice that    */
  /* has GEMMs beside it on the geometry that occupies fewer CUs.  Results do not depend on it. */
  int fwd_units;
  /* backward, optional: dz_hl != NULL makes BPTT write the gate gradients as the PACKED PLANES  */
  /* of asr_pack_hl ((T n_pad) rows x 8H reduction indices: row = frame * n_pad + sample, the    */
  /* r-plane layout, ld = 8H) INSTEAD of the fp32 slab dz (which may then be NULL): a thread's   */
  /* 16 gate gradients are exactly one (16 hi, 16 lo) group at the byte offset of its fp32       */
  /* values, so the kernel stores the same 64 bytes either way and the separate pack pass over   */
  /* dz (2.1 GB per layer at H = 512, 64 utterances) disappears.  The planes' power-of-two       */
  /* scale must be known BEFORE the pass: it is asr_pack_hl's scale of *dz_bound (a device       */
  /* float >= the max |dz| this call will produce, e.g. 64 x the previous training step's        */
  /* maximum -- asr_lstm_dz_guard maintains it), written to *dz_scale_out for asr_gemm_hl.  The  */
  /* kernel then also splits dz with THAT scale for its own dz @ U^T products (instead of a      */
  /* per-sample scale): same 2^-22 product accuracy while max |dz| stays within                  */
  /* [2^-11, 2^7] x bound.  Exists on the two-dimensional-split kernels only                    */
  /* (asr_lstm_dz_hl_supported); dz_absmax and db_part work as before.                          */
  void* dz_hl;
  const float* dz_bound;
  float* dz_scale_out;
} asr_lstm_args;
/* 1 if asr_lstm_seq_bwd would honour a->dz_hl for this geometry / mode / arithmetic, else 0.   */
int asr_lstm_dz_hl_supported(const asr_lstm_args* a);
/* Keeps a layer's *dz_bound in step with the measured *dz_absmax of the pass just enqueued     */
/* (device-side, no synchronisation): with M = max * scale(bound), the bound is kept while M    */
/* stays in [2^-1, 2^6) (so that identical inputs see identical scales); otherwise it becomes    */
/* 64 * max when the maximum shrank and sqrt(bound * 64 * max) when it grew (a spike is usually  */
/* gone a step later; persistent growth settles within three steps).  planes_used != 0: the     */
/* pass wrote planes with this bound -- if M                                                    */
/* left [2^-7, 2^15] (fp16 overflow of the hi plane at 2^16, precision loss below) the sticky   */
/* timeout flag of `bwd_workspace` is raised, i.e. the step is vetoed and re-run exactly like   */
/* a step whose persistent kernel gave up (asr_lstm_status).                                    */
int asr_lstm_dz_guard(const float* dz_absmax, float* dz_bound, int planes_used,
                      void* bwd_workspace, asr_stream_t stream);
size_t asr_lstm_workspace_bytes(const asr_lstm_args* a, int backward);
int asr_lstm_seq_fwd(const asr_lstm_args* a, void* workspace, size_t ws_bytes,
                     asr_stream_t stream);
int asr_lstm_seq_bwd(const asr_lstm_args* a, void* workspace, size_t ws_bytes,
                     asr_stream_t stream);
/* Synchronises `stream`, then returns 0 or ASR_ERR_TIMEOUT if a persistent
 * kernel that used this workspace abandoned a bounded spin (results invalid)
 * in ANY call since the previous asr_lstm_status on it: the first int of the
 * workspace is a sticky flag that only this function clears (the workspace's
 * first 256 bytes must therefore be zero before its first use).  A host that
 * never wants to synchronise may instead copy that int back with its own
 * asynchronous readback once per training step.                             */
int asr_lstm_status(const void* workspace, asr_stream_t stream);
/* Launch plan the library would use (diagnostics / DESIGN.md numbers).       */
int asr_lstm_plan(const asr_lstm_args* a, int backward, int* k_split,
                  int* k_per_lane, int* blocks, int* chains_per_launch);
/* Synchronises `stream`; number of chains of the last call on this workspace
 * that ran on the same-XCD (L2) transport (diagnostic), or a negative status. */
int asr_lstm_fast_chains(const void* workspace, asr_stream_t stream);
/* Debug: hand-off timeline of the last forward launch made under ASR_LSTM_DBG & 128:       */
/* out[((block*4 + wave)*16 + step)*2 + {0 data arrived, 1 published}], 100 MHz ticks.      */
int asr_lstm_trace(long long* out, size_t n_words, asr_stream_t stream);
/* Debug (env ASR_LSTM_DBG & 32): per-phase shader-clock ticks of workgroup 0 of the
 * last call's first chain, summed over its steps: 4 waves x 6 phases {arithmetic
 * before the gather, waiting for it, arithmetic behind it, barrier, products +
 * publish, issuing the next gather}.                                           */
int asr_lstm_profile(const void* workspace, asr_stream_t stream, long long* out24);

/* ------------------------------------------------------------------------ */
/* K7  CTC loss + gradient.  Replaces core/ctc_utils.py:60-70 ->             */
/* tf.nn.ctc_loss (blank = C-1, internal softmax, ctc_merge_repeated=True).  */
/* logits/grad: (T, n_pad, C) time-major.  labels (N, l_max) int32 padded,   */
/* label_len/seq_len (N) int32, all on the device.  loss[n] = -log p(l|x);   */
/* grad = grad_scale * d loss_n / d logits, 0 for t >= seq_len[n] and for    */
/* padding rows n >= N.  An infeasible target yields loss = +inf, grad = 0   */
/* (the host raises the TF error before launching).  grad may be NULL (loss  */
/* only) and must NOT alias logits (the gradient kernel re-reads the logits  */
/* of neighbouring frames while other waves write grad): ASR_ERR_INVALID.    */
/* ------------------------------------------------------------------------ */
size_t asr_ctc_workspace_bytes(int T, int N, int n_pad, int C, int l_max);
int asr_ctc_loss_grad(const float* logits, const int* labels,
                      const int* label_len, const int* seq_len, int T, int N,
                      int n_pad, int C, int l_max, float grad_scale,
                      float* loss, float* grad, void* workspace,
                      size_t ws_bytes, asr_stream_t stream);

/* K8  Greedy decode.  Replaces core/ctc_utils.py:42 ->                      */
/* tf.nn.ctc_greedy_decoder (first-max argmax, merge repeats, drop blank).   */
/* decoded (N, T) int32 padded with -1; decoded_len (N).                     */
int asr_ctc_greedy(const float* logits, const int* seq_len, int T, int N,
                   int n_pad, int C, int* decoded, int* decoded_len,
                   asr_stream_t stream);

/* K9  Beam search (host side of the library; logits already on the host).   */
/* Replaces core/ctc_utils.py:48-50 -> tf.nn.ctc_beam_search_decoder         */
/* (top_paths=1, merge_repeated as given).  logits (T, n_pad, C) HOST float. */
int asr_ctc_beam_search_host(const float* logits_host, const int* seq_len_host,
                             int T, int N, int n_pad, int C, int beam_width,
                             int merge_repeated, int* decoded, int* decoded_len,
                             float* log_score);

/* K9  Beam search on the device (same algorithm, arithmetic and tie-breaking as the host    */
/* form; logits (T, n_pad, C) and seq_len are DEVICE pointers, as asr_ctc_greedy takes them;  */
/* decoded (N, T) int32 padded with -1, decoded_len (N), log_score (N) or NULL: device).     */
/* One workgroup per utterance; C <= 64, beam_width <= 1024.  The workspace holds the prefix  */
/* tree (its size is the hard bound T * beam_width child blocks per utterance; no init).      */
size_t asr_ctc_beam_device_workspace_bytes(int T, int N, int C, int beam_width);
int asr_ctc_beam_device(const float* logits, const int* seq_len, int T, int N, int n_pad, int C,
                        int beam_width, int merge_repeated, int* decoded, int* decoded_len,
                        float* log_score, void* workspace, size_t ws_bytes, asr_stream_t stream);

/* Work counters of one utterance of the LAST asr_ctc_beam_device call on this workspace     */
/* (synchronises the stream): out7 = 100 MHz ticks in the four phases of a frame (branch      */
/* update, ranking, turns, hand-over), expanding turns, insertions, child blocks allocated.   */
int asr_ctc_beam_device_counters(const void* workspace, int T, int N, int C, int beam_width,
                                 int utterance, long long* out7, asr_stream_t stream);

/* K10 Edit distance (host).  Replaces core/metrics.py:8 -> tf.edit_distance */
/* (normalize=True).  Ragged inputs as (N, max) padded + lengths.            */
int asr_edit_distance_host(const int* hyp, const int* hyp_len, int hyp_ld,
                           const int* truth, const int* truth_len, int truth_ld,
                           int N, float* out_normalized);

/* ------------------------------------------------------------------------ */
/* K11 Optimiser.  Replaces Keras Adam/SGD with clipnorm (train.py:133-137)  */
/* and the l2 regularisers (core/models.py:263-264,279).  params/grads/m/v   */
/* are flat float32 buffers of n elements; segments (n_seg x {int64 offset,  */
/* int64 len, float l2, float pad}) give the per-tensor l2 coefficient.      */
/* Step 1 writes norm_out[0] = ||g + 2*l2*p||_2 and norm_out[1] = sum l2*p^2 */
/* (both float64, device).  Step 2 applies the update using norm_out[0]      */
/* without a host round trip.                                                */
/* ------------------------------------------------------------------------ */
typedef struct asr_segment {
  int64_t offset;
  int64_t len;
  float l2;
  float reserved;
} asr_segment;
size_t asr_optim_workspace_bytes(int64_t n);
int asr_grad_norm(const float* params, const float* grads, int64_t n,
                  const asr_segment* segments_dev, int n_seg, double* norm_out,
                  void* workspace, size_t ws_bytes, asr_stream_t stream);
int asr_adam_step(float* params, const float* grads, float* m, float* v,
                  int64_t n, const asr_segment* segments_dev, int n_seg,
                  const double* norm_dev, float clipnorm, float lr, float beta1,
                  float beta2, float eps, int step, asr_stream_t stream);
int asr_sgd_step(float* params, const float* grads, float* vel, int64_t n,
                 const asr_segment* segments_dev, int n_seg,
                 const double* norm_dev, float clipnorm, float lr,
                 float momentum, asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K5-LN  Layer-normalised LSTM cell (layer_norm option of the reference's    */
/* override: core/layers.py:407-436, 460-462; core/layers_utils.py:16-19):    */
/* LN over the 4H row of h_prev@U, of x@W, and over the H row of the cell     */
/* state that feeds the output (the carried c stays un-normalised); combines  */
/* with multiplicative integration and zoneout.  Generic path: one workgroup  */
/* per (sample, direction) row and step (see csrc/lstm_ln.hip), several times */
/* slower than the persistent kernels.  cellp (2, 34H) per direction: alpha,  */
/* beta1, beta2, b (4H each; alpha/beta ignored unless has_mi), gain/bias of  */
/* LN(h@U) and of LN(x@W) (4H each), gain/bias of LN(c) (H each).  wx = x@W   */
/* WITHOUT bias.  Backward writes duh / dwx = d/d(raw h@U) / d/d(raw x@W) and */
/* dparams (n_pad, 2, 34H): per-row sums of the cellp gradients (sum axis 0). */
/* ------------------------------------------------------------------------ */
typedef struct asr_lstm_ln_args {
  int T, n_pad, H;       /* n_pad % 16 == 0, H % 4 == 0, H <= 512              */
  int has_mi;
  const float* U;        /* (2, H, 4H)                                         */
  const float* mask_u;   /* (2, n_pad, H) or NULL                              */
  const float* cellp;    /* (2, 34H)                                           */
  const float* zone_c;   /* (T, 2, H) or NULL                                  */
  const float* zone_h;
  const float* wx;       /* (T, n_pad, 2, 4H)                                  */
  float* uh;             /* (T, n_pad, 2, 4H) raw h_prev@U: fwd out, bwd in     */
  float* y;              /* (T, n_pad, 2H)                                     */
  float* cell;           /* (T, n_pad, 2, H)                                   */
  float* gates;          /* (T, n_pad, 2, 4H)                                  */
  const float* dy;       /* backward: (T, n_pad, 2H)                           */
  float* duh;            /* backward out                                       */
  float* dwx;            /* backward out                                       */
  float* dparams;        /* backward out                                       */
  int activation;        /* as asr_lstm_args.activation (0 = tanh)             */
} asr_lstm_ln_args;
size_t asr_lstm_ln_workspace_bytes(const asr_lstm_ln_args* a);
int asr_lstm_ln_seq_fwd(const asr_lstm_ln_args* a, asr_stream_t stream);
int asr_lstm_ln_seq_bwd(const asr_lstm_ln_args* a, void* workspace, size_t ws_bytes,
                        asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* C1  Gradient all-reduce (RCCL over xGMI).                                   */
/* One communicator per process / GPU: rank 0 calls asr_comm_unique_id and     */
/* ships the ASR_COMM_ID_BYTES to the other ranks by any side channel; every   */
/* rank then calls asr_comm_init with the HIP device already selected.         */
/* asr_comm_allreduce_sum sums n floats over the ranks in place, enqueued on    */
/* the given stream.  librccl is resolved with dlopen at first use.  This IS    */
/* the collective path of the shipped Python host (parallel.CapiComm: gradient  */
/* all-reduce, parameter broadcast, metric sums); torch.distributed only ships  */
/* the id bytes.                                                                */
/* ------------------------------------------------------------------------ */
#define ASR_COMM_ID_BYTES 128
typedef void* asr_comm_t;
int asr_comm_unique_id(void* id_out);
int asr_comm_init(const void* id, int rank, int world, asr_comm_t* comm_out);
int asr_comm_allreduce_sum(asr_comm_t comm, float* buf, int64_t n, asr_stream_t stream);
int asr_comm_destroy(asr_comm_t comm);

/* out[i] = a * x[i] + b * y[i] (out may alias x or y).  The residual connection of    */
/* brsmv1(residual=...): keras merge([new_o, o], mode='sum' | 'ave'),                 */
/* core/models.py:273-276, and its gradient accumulation.                            */
int asr_axpby(int64_t n, float a, const float* x, float b, const float* y, float* out,
              asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Operation-level entry points: one per role on the training path, expressed */
/* on the general entry points above (csrc/roles.cpp; no further kernels).    */
/* ------------------------------------------------------------------------ */
/* K1-K3 by feature class: preprocessing/audio.py:309-367 (MFCC; cfg->kind    */
/* must be 0) and :394-442 (LogFbank; cfg->kind must be 1, no DCT table).     */
/* Arguments as asr_frontend_features.                                        */
int asr_frontend_mfcc_batch(const asr_frontend_cfg* cfg, const float* audio,
                            const int* offsets, const int* lengths,
                            const int* host_lengths, int n_utt, int n_pad,
                            const double* window, const double* mel,
                            const int* mel_range, const double* dct, float* out,
                            int t_out, int* out_frames, void* workspace,
                            size_t ws_bytes, asr_stream_t stream);
int asr_frontend_logfbank_batch(const asr_frontend_cfg* cfg, const float* audio,
                                const int* offsets, const int* lengths,
                                const int* host_lengths, int n_utt, int n_pad,
                                const double* window, const double* mel,
                                const int* mel_range, float* out, int t_out,
                                int* out_frames, void* workspace, size_t ws_bytes,
                                asr_stream_t stream);

/* K4/K6 the three GEMMs of one LSTM layer's input projection                 */
/* (core/layers.py:439, K.dot(x * B_W[0], self.W), hoisted out of the time    */
/* loop) over `rows` slab rows (whole frames of n_pad samples when a mask is  */
/* given; mask row = slab row % n_pad):                                       */
/*   fwd  : zx = (x (.) B_W) @ W + bias                                       */
/*   dgrad: dx = dx_beta * dx + (dz @ W^T) (.) B_W                            */
/*   wgrad: dW = (x (.) B_W)^T @ dz ;  db = column sums of dz (if db != NULL) */
/* gate_dim is the number of gate columns this call covers (4H for one        */
/* direction -- each direction of a Bidirectional layer has its own B_W --    */
/* or 8H for both when mask_w is NULL); ldw / ldz are the row strides of W /  */
/* dW and of zx / dz (8H in the engine's fused two-direction layout).         */
typedef struct asr_gate_gemm_args {
  int rows, n_pad, in_dim, gate_dim;
  const float* x;  int ldx;   /* (rows, in_dim)                               */
  const float* W;  int ldw;   /* (in_dim, gate_dim)                           */
  const float* bias;          /* (gate_dim) or NULL (fwd)                     */
  const float* mask_w;        /* (n_pad, in_dim) variational mask or NULL     */
  float* zx;       int ldz;   /* fwd out (rows, gate_dim)                     */
  const float* dz;            /* dgrad / wgrad in, row stride ldz             */
  const float* dz_absmax;     /* device max|dz| (split-fp16 pre-scale) / NULL */
  float* dx;                  /* dgrad out (rows, in_dim), row stride ldx     */
  float dx_beta;              /* 0, or 1 to add the second direction's term   */
  float* dW;                  /* wgrad out (in_dim, gate_dim), row stride ldw */
  float* db;                  /* wgrad out (gate_dim) or NULL                 */
  int split_k;                /* wgrad: split the long `rows` reduction       */
  int precision;              /* as asr_gemm_args.precision (-1 = default)    */
} asr_gate_gemm_args;
/* role: 0 = fwd, 1 = dgrad, 2 = wgrad. */
size_t asr_gemm_gate_workspace_bytes(const asr_gate_gemm_args* g, int role);
int asr_gemm_gate_fwd(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                      asr_stream_t stream);
int asr_gemm_gate_dgrad(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                        asr_stream_t stream);
int asr_gemm_gate_wgrad(const asr_gate_gemm_args* g, void* workspace, size_t ws_bytes,
                        asr_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K12 counter-based random numbers for the training-time noise of the path:  */
/* replaces the TF random ops behind GaussianNoise (core/models.py:250-251),  */
/* the input Dropout (:257-258), the variational dropout masks B_W / B_U      */
/* (:265-266; one mask per batch, inverted scaling) and the zoneout keep      */
/* masks (core/layers_utils.py:34-42).  Philox-4x32-10; key = seed, counter = */
/* (block, stream_id, step, block >> 32); element 4b+j = word j of block b.   */
/* Stateless: (seed, stream_id, step) fully determine the tensor.             */
/*   asr_dropout_masks : out[i] = u_i >= p ? scale : 0, u = (word >> 8) 2^-24 */
/*   asr_dropout_apply : out = in * that mask (mask_out receives it, or NULL) */
/*   asr_gaussian_noise: out = in + sigma * N(0,1) (Box-Muller; in may be     */
/*                       NULL = pure noise, or equal to out)                  */
/*   asr_random_words  : the raw 32-bit words (tests)                         */
/*   asr_mul           : out = x * y elementwise (mask applied to a gradient) */
/* All pointers 16-byte aligned device memory.                                */
/* ------------------------------------------------------------------------ */
int asr_dropout_masks(float* out, int64_t n, float p, float scale, uint64_t seed,
                      uint32_t stream_id, uint32_t step, asr_stream_t stream);
int asr_dropout_apply(const float* in, float* out, float* mask_out, int64_t n, float p,
                      float scale, uint64_t seed, uint32_t stream_id, uint32_t step,
                      asr_stream_t stream);
int asr_gaussian_noise(const float* in, float* out, int64_t n, float sigma, uint64_t seed,
                       uint32_t stream_id, uint32_t step, asr_stream_t stream);
int asr_random_words(unsigned* out, int64_t n, uint64_t seed, uint32_t stream_id, uint32_t step,
                     asr_stream_t stream);
int asr_mul(int64_t n, const float* x, const float* y, float* out, asr_stream_t stream);

/* A stream whose kernels run only on the compute units set in `mask` (bit i of word i/32 =  */
/* CU i; hipExtStreamCreateWithCUMask): partitions the chip between the latency-bound        */
/* recurrences and the GEMMs that run beside them.  Destroy with asr_stream_destroy.         */
int asr_stream_create_cu_mask(const uint32_t* mask, int words, asr_stream_t* stream_out);
int asr_stream_destroy(asr_stream_t stream);

/* K9 / K10 under their operation names (same arguments as the *_host forms).  */
int asr_ctc_beam(const float* logits_host, const int* seq_len_host, int T, int N,
                 int n_pad, int C, int beam_width, int merge_repeated, int* decoded,
                 int* decoded_len, float* log_score);
int asr_edit_distance(const int* hyp, const int* hyp_len, int hyp_ld,
                      const int* truth, const int* truth_len, int truth_ld, int N,
                      float* out_normalized);

/* Device-side veto of the NEXT update (no host round trip): if *flag_a or *flag_b (device    */
/* ints, either may be NULL) is non-zero -- e.g. the sticky timeout words at the head of the   */
/* recurrent kernels' workspaces -- norm_dev[0] becomes -1 and asr_adam_step / asr_sgd_step     */
/* given that norm return without touching parameters or state.  Enqueue it between            */
/* asr_grad_norm and the step; the host sees the flag at its next (lagged) check, switches to   */
/* the stepwise recurrent kernels (mode 1) and goes on with intact weights.                     */
int asr_optim_guard(double* norm_dev, const int* flag_a, const int* flag_b, asr_stream_t stream);
/* Data parallel, the veto has to be collective (a rank that skips an update the others apply   */
/* diverges): out2[0] / out2[1] <- 1.0f where *flag_a / *flag_b is set, else 0.0f.  The host     */
/* keeps out2 directly behind the flat gradient buffer, so the gradient all-reduce (C1, a sum)  */
/* carries the flags of every rank to every rank, and then passes out2, out2 + 1 to              */
/* asr_optim_guard (which tests for any non-zero bit pattern).                                   */
int asr_timeout_flags(const int* flag_a, const int* flag_b, float* out2, asr_stream_t stream);
/* Test / diagnosis only: occupies `blocks` workgroups of 256 threads with lds_bytes of LDS     */
/* each for `seconds` (<= 5) of wall clock on `stream` -- the stand-in for a foreign kernel (an  */
/* RCCL collective) that takes compute units while a persistent recurrence needs them all.       */
int asr_debug_occupy(int blocks, int lds_bytes, double seconds, asr_stream_t stream);

/* K11 one call per optimiser step: global gradient norm (written to norm_out  */
/* as asr_grad_norm does) followed by the clipped Adam / SGD update, both      */
/* enqueued on `stream`; workspace from asr_optim_workspace_bytes(n).          */
int asr_clip_adam_step(float* params, const float* grads, float* m, float* v,
                       int64_t n, const asr_segment* segments_dev, int n_seg,
                       double* norm_out, float clipnorm, float lr, float beta1,
                       float beta2, float eps, int step, void* workspace,
                       size_t ws_bytes, asr_stream_t stream);
int asr_clip_sgd_step(float* params, const float* grads, float* vel, int64_t n,
                      const asr_segment* segments_dev, int n_seg, double* norm_out,
                      float clipnorm, float lr, float momentum, void* workspace,
                      size_t ws_bytes, asr_stream_t stream);

/* C1 asr_comm_allreduce_sum under the name the scope table uses.             */
int asr_comm_allreduce(asr_comm_t comm, float* buf, int64_t n, asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
