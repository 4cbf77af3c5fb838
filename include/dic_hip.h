/*
 * dic_hip.h -- C ABI of the MI355X (gfx950) hot path of deep-interpolation-clustering.
 *
 * One shared object (libdic_hip.so), extern "C", plain pointers and sizes only.
 * Conventions shared by every compute entry point:
 *   - returns DIC_OK (0) or a negative dic_status code; never throws, never allocates;
 *   - every pointer is a DEVICE pointer owned by the caller and borrowed for the call;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *   - `workspace` buffers are caller-allocated scratch; query their size first;
 *   - kernels hold no global mutable state and are re-entrant across streams.
 *
 * The reference has no FFI: its boundary is the Python module surface.  Each entry
 * point below names the reference lines (relative to the upstream repository) whose
 * arithmetic it replaces; INTEGRATION.md shows the ctypes stub that binds them
 * behind the upstream module names.
 *
 * Stacked input layout (reference: interpolation_layer.py:25-30, dataloader.py:54-69):
 *   x is (B, 4C, T) contiguous f32; planes [value*mask | mask | time(h) | hold-out].
 *   `lengths` (B,C) int32 is optional: when given, row (b,c) is a PREFIX of
 *   lengths[b,c] valid slots (what p0/the trainers always produce) and the mask
 *   plane is not read; when NULL the mask plane is honoured slot by slot.
 * Packed ragged layout (dataset-resident fast path):
 *   row_off (B*C+1) int64 offsets into packed `t_pk`/`v_pk` f32 arrays, rows stored
 *   back to back in (b,c) order; max_len = longest row.
 */
#ifndef DIC_HIP_H
#define DIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIC_ABI_VERSION 1

typedef enum dic_status {
    DIC_OK = 0,
    DIC_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size                      */
    DIC_ERR_UNSUPPORTED = -2,   /* shape outside the compiled limits (see each entry)   */
    DIC_ERR_WORKSPACE = -3,     /* workspace too small                                  */
    DIC_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after the launch     */
} dic_status;

typedef void* dic_stream_t;    /* hipStream_t */

/* Limits compiled into the kernels. */
#define DIC_MAX_CHANNELS 16    /* C  (C*C <= one 256-thread workgroup) */
#define DIC_MAX_REFPOINTS 64   /* R  */
#define DIC_MAX_CLUSTERS 32    /* K  */
#define DIC_LATENT_MAX_DIM 256 /* D: latent rows are handled 4 columns per lane by one wave */

int dic_version(void);
const char* dic_status_string(int status);
const char* dic_last_error_string(void);   /* detail of the last failure on this thread */

/* ------------------------------------------------------------------ k1: SCI (+CCI) ------
 * Replaces SingleChannelInterp.forward (interpolation_layer.py:31-86) and, when
 * cci_kernel != NULL, CrossChannelInterp.forward fused as its epilogue
 * (interpolation_layer.py:99-127).
 *   out   (B,R,3C): cci_kernel==NULL -> [y | w | y_trans]; else [smooth | exp(w) | y_trans-smooth]
 *   saved (B,7,C,R) or NULL: y,w,y_trans,Eu1,Exu1,Eu10,Exu10 -- what the backward needs
 *          (Eu = E_s[u], Exu = E_s[x u] under the softmax weights, u=(t-ref)^2).
 * A channel with no valid observation yields NaN y/y_trans and w=-inf, as upstream.
 * ref_grid (R) f32 = the reference time grid, i.e. torch.linspace(0, hours, R) (interpolation_layer.py:41),
 * passed as data so the kernel sees bit-identical grid values.
 * Limits: C<=16, R<=64, one encounter (2*C*T f32 + results) must fit 64 KB of LDS. */
int dic_sci_cci_fwd(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                    const float* sci_kernel, const float* cci_kernel,
                    float* out, float* saved, dic_stream_t stream);

int dic_sci_cci_fwd_ragged(const float* t_pk, const float* v_pk, const int64_t* row_off, int max_len,
                           int B, int C, int R, const float* ref_grid,
                           const float* sci_kernel, const float* cci_kernel,
                           float* out, float* saved, dic_stream_t stream);

/* The same forward reading a RAGGED ENCOUNTER STORE in place (the device-resident form of the feed_data array
 * dataloader.py:54-79 builds): for every (encounter, channel) row only the observed samples, packed back to back --
 *   t_pk, v_pk   packed time stamps (h) and values (ob * mask, rescaled): row g of the store at [row_off[g], row_off[g+1]);
 *   hold_pk      optional packed hold-out flags (plane 3, p0_data_process.py:95-117): the model input is v * hold (a denoising
 *                step, pretrain_trainer.py:139-141);
 *   row_off      (N*C + 1) int64; the packed arrays must keep >= 64 readable elements behind the last row;
 *   enc_idx      (B) int32 store encounter of each batch row (a shuffled batch is read in place: no gather, no padded copy), or
 *                NULL: the batch is encounters 0..B-1;
 *   lengths      (B,C) int32 row lengths of the BATCH rows (= the store's, gathered), or NULL: taken from row_off;
 *   T            the padded width the dense path would have (bounds every row; sizes the LDS rows).
 * times_sorted != 0: the caller certifies that every row's time stamps are non-decreasing (ragged.RaggedStore checks it once when the cohort is packed):
 * the soft-max shift min_t (t - ref)^2 is then found by bisection instead of a pass over the row -- the same value, bit for bit.
 * out (B,R,3C) f32 and / or xenc (R,B,xw) bf16 packed rows as dic_sci_cci_fwd_packed; saved as above.  Same arithmetic, same results
 * as the dense entry points on the same samples. */
int dic_sci_cci_fwd_store(const float* t_pk, const float* v_pk, const uint8_t* hold_pk, const int64_t* row_off, const int32_t* enc_idx,
                          const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid, const float* sci_kernel,
                          const float* cci_kernel, float* out, float* saved, void* xenc, int xw, int times_sorted, dic_stream_t stream);

/* Backward of the above wrt the two parameters only (the inputs carry no gradient upstream).
 *   grad_out (B,R,3C) cotangent of `out`; saved from the forward;
 *   grad_sci_kernel (C), grad_cci_kernel (C,C; ignored when cci_kernel==NULL): OVERWRITTEN.
 * Deterministic two-stage reduction through `workspace`. */
size_t dic_sci_cci_bwd_workspace(int B, int C, int R);
int dic_sci_cci_bwd(const float* grad_out, const float* saved, const float* sci_kernel,
                    const float* cci_kernel, int B, int C, int R,
                    float* grad_sci_kernel, float* grad_cci_kernel,
                    void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* Stand-alone CrossChannelInterp on an arbitrary (B,R,3C) tensor s=[y|w|y_trans]
 * (interpolation_layer.py:99-127) and its backward wrt both s and the kernel. */
int dic_cci_fwd(const float* s, const float* cci_kernel, int B, int C, int R, float* out,
                dic_stream_t stream);
size_t dic_cci_bwd_workspace(int B, int C, int R);
int dic_cci_bwd(const float* grad_out, const float* s, const float* cci_kernel, int B, int C, int R,
                float* grad_s, float* grad_cci_kernel,
                void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* The fused pair feeding the encoder LSTM without a layout pass: the forward additionally (out may then be NULL) writes
 * xenc (R,B,xw) bf16 = rows [cci(sci(x)) (3C) | 1 | 0 ...] -- time-major, padded to the MFMA k-step, the constant one carrying
 * the LSTM bias (dic_lstm_fwd_proj) -- and the backward takes the encoder's input gradient in that same layout. */
int dic_sci_cci_fwd_packed(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                           const float* sci_kernel, const float* cci_kernel, float* out, float* saved, void* xenc, int xw,
                           dic_stream_t stream);
int dic_sci_cci_bwd_packed(const void* grad_packed, int xw, const float* saved, const float* sci_kernel, const float* cci_kernel,
                           int B, int C, int R, float* grad_sci_kernel, float* grad_cci_kernel, void* workspace,
                           size_t workspace_bytes, dic_stream_t stream);

/* ------------------------------------------------------------------ k2: RBF de-interpolation
 * Replaces RBF.forward minus compress_fc (rbf.py:57-108, gaussian rbf.py:129-131).
 *   v (B,C,R) = compress_fc output;  y (B,C,T) OVERWRITTEN (0 in masked slots).
 * Limits: C<=16, R<=64.  * v_time_major != 0: v (and grad_v) are laid out (R,B,C) -- the row order TimeDistributed(CompressFC) produces them in on the
 * decoder's (R,B,.) output -- instead of (B,C,R): no transposing copy between the FC head and this kernel.
  * norm (optional, saved for the backward) = 1 / (sum_r phi + 1e-10) per valid slot.  prefix_only != 0 (with lengths): only the first
 * n slots of each row of y / norm (dic_masked_sse_bwd: of grad_rec) are written -- the training step never reads the padding, which
 * is half of the 96-slot rows at ~50 observations per channel; with 0 the padding is written as zeros, as upstream's `* mask` does.
 */
int dic_rbf_fwd(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                const float* rbf_kernel, const float* v, int v_time_major, float* y, float* norm /* (B,C,T) or NULL: saved for the
                backward */, int prefix_only, dic_stream_t stream);

/* grad_y (B,C,T) -> grad_v (B,C,R), grad_rbf_kernel (C); both OVERWRITTEN.  y, norm: forward outputs.
 * Masks are binary upstream; a non-zero mask value is treated as 1. */
size_t dic_rbf_bwd_workspace(int B, int C, int T, int R);
int dic_rbf_bwd(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                const float* rbf_kernel, const float* v, int v_time_major, const float* y, const float* norm,
                const float* grad_y, float* grad_v, float* grad_rbf_kernel,
                void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* k2 with the reconstruction loss riding along (RBF.forward + Net.rec_loss, rbf.py:57-108 + clustering_interp.py:197-203), for the
 * training step, where the reconstruction is consumed by the loss alone:
 *   dic_rbf_fwd_loss: as dic_rbf_fwd, and out2[0] = sum over the valid slots of (y - ob)^2, out2[1] = #valid slots (what
 *     dic_masked_sse_fwd returns for a binary mask) from the values the kernel holds in registers -- no second pass over y and ob;
 *   dic_rbf_bwd_loss: as dic_rbf_bwd with dL/dy = grad_loss[0] * 2 (y - ob) / sse_count[1] on the valid slots formed on the fly from ob
 *     (prefix masks: lengths required) -- the (B,C,T) gradient of the reconstruction is never written or read.
 * sse_count = out2 after the caller's all-reduce over ranks; grad_loss (1) = dL/d(out2[0] / out2[1]); all device pointers. */
size_t dic_rbf_fwd_loss_workspace(int B, int C, int T, int R);
int dic_rbf_fwd_loss(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel,
                     const float* v, int v_time_major, const float* ob, float* y, float* norm, int prefix_only, float* out2,
                     void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_rbf_bwd_loss(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel,
                     const float* v, int v_time_major, const float* y, const float* norm, const float* ob, const float* sse_count,
                     const float* grad_loss, float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes,
                     dic_stream_t stream);

/* k2 reading the ragged encounter store in place (see dic_sci_cci_fwd_store for t_pk / v_pk / row_off / enc_idx / lengths; lengths is
 * required here): the time stamps come from t_pk and -- for the fused reconstruction loss -- the observations from v_pk, so the
 * padded (B,4C,T) input and the (B,C,T) observation tensor do not exist on this path.  y, norm (and grad_y when given) stay (B,C,T).
 *   dic_rbf_fwd_store: with_loss != 0 = dic_rbf_fwd_loss (out2, workspace of dic_rbf_fwd_loss_workspace bytes), else dic_rbf_fwd;
 *   dic_rbf_bwd_store: grad_y == NULL = dic_rbf_bwd_loss (sse_count, grad_loss required), else dic_rbf_bwd. */
int dic_rbf_fwd_store(const float* t_pk, const float* v_pk, const int64_t* row_off, const int32_t* enc_idx, const int32_t* lengths,
                      int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel, const float* v, int v_time_major,
                      int with_loss, float* y, float* norm, int prefix_only, float* out2, void* workspace, size_t workspace_bytes,
                      dic_stream_t stream);
int dic_rbf_bwd_store(const float* t_pk, const float* v_pk, const int64_t* row_off, const int32_t* enc_idx, const int32_t* lengths,
                      int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel, const float* v, int v_time_major,
                      const float* y, const float* norm, const float* grad_y, const float* sse_count, const float* grad_loss,
                      float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* Net.rec_loss (clustering_interp.py:197-203): out2[0]=sum((rec*m-ob*m)^2), out2[1]=#{m==1}.
 * mask (B,C,T) may be NULL when lengths is given.  The loss is out2[0]/out2[1]. */
size_t dic_masked_sse_workspace(int B, int C, int T);
int dic_masked_sse_fwd(const float* ob, const float* rec, const float* mask, const int32_t* lengths,
                       int B, int C, int T, float* out2,
                       void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* grad_rec = grad_loss[0] * 2*m*(rec*m-ob*m)/sse_count[1]   (all device pointers). */
int dic_masked_sse_bwd(const float* ob, const float* rec, const float* mask, const int32_t* lengths,
                       int B, int C, int T, const float* sse_count, const float* grad_loss,
                       float* grad_rec, int prefix_only, dic_stream_t stream);

/* ------------------------------------------------------------------ k3: DEC ---------------
 * Replaces ClusterAssignment.forward (dec.py:49-63).
 *   z (B,D), centers (K,D) -> q (B,K); tsaved (B,K) or NULL = 1/(1+d2/alpha) for the backward;
 *   colsum (K) or NULL = sum_i q_ij (the batch-level f_j of dec.py:73; all-reduce it when sharded).
 * Limits: D%4==0, D<=256, K<=32. */
size_t dic_dec_fwd_workspace(int B, int D, int K);
int dic_dec_fwd(const float* z, const float* centers, int B, int D, int K, float alpha,
                float* q, float* tsaved, float* colsum,
                void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* target_distribution (dec.py:66-76) given the (global) column sums. */
int dic_dec_target(const float* q, const float* colsum, int B, int K, float* p, dic_stream_t stream);
/* Backward of dic_dec_fwd: grad_q (B,K) -> grad_z (B,D), grad_centers (K,D); OVERWRITTEN. */
size_t dic_dec_bwd_workspace(int B, int D, int K);
int dic_dec_bwd(const float* z, const float* centers, const float* q, const float* tsaved,
                const float* grad_q, int B, int D, int K, float alpha,
                float* grad_z, float* grad_centers,
                void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* Fused Net.kl_loss (clustering_interp.py:205-207) forward + d/dq:
 *   kl_out[0] = sum_ij p (log p - log q) / batch_div ; grad_q (B,K) or NULL = -gscale*p/(batch_div*q). */
size_t dic_dec_kl_workspace(int B, int K);
int dic_dec_kl(const float* q, const float* p, int B, int K, float batch_div, float gscale,
               float* kl_out, float* grad_q,
               void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* ------------------------------------------------------------------ k4: k-means -----------
 * Replaces the Lloyd E+M step scikit-learn runs for KMeans.fit/predict at
 * clustering_trainer.py:75-82, p2_clustering_optK.py:260-389, p4_clustering_final.py:159-174
 * (sklearn 1.7.2 _k_means_lloyd.pyx:23-218, _k_means_common.pyx:167-311).
 * `n_runs` independent problems on the SAME data share one launch (restarts / n_init).
 *   X (N,D) f32 (already mean-centred by the caller, as sklearn does), xnorm (N) = ||x||^2,
 *   centers (n_runs,K,D) IN/OUT: replaced by the updated centres,
 *   labels  (n_runs,N) int32 IN/OUT: previous labels in (use -1 before the first call),
 *   status  (n_runs,DIC_KM_STATUS_WORDS) f32 IN/OUT, zero it before the first iteration:
 *     [0] done flag (1 = converged: this and later calls are no-ops for the run)
 *     [1] iterations executed   [2] last center_shift_tot   [3] last #labels changed
 *     [4] strict-convergence flag [5] relocations performed [6] tol (IN) [7] max_iter (IN)
 * One call = one Lloyd iteration for every run that is not done: assign by
 * argmin_j(||c_j||^2 - 2 x.c_j) (first minimum wins), accumulate sums/counts, relocate empty
 * clusters, average, centre shift, convergence test (labels unchanged -> strict; else
 * shift_tot <= tol).  Limits: D%4==0, D<=256, K<=32. */
#define DIC_KM_STATUS_WORDS 8
size_t dic_kmeans_workspace(int N, int D, int K, int n_runs);
int dic_kmeans_lloyd_iter(const float* X, const float* xnorm, int N, int D, int K, int n_runs,
                          float* centers, int32_t* labels, float* status,
                          void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* The same iteration with the POINTS SHARDED over ranks (one process per GPU): each rank runs
 * dic_kmeans_lloyd_partial on its rows, the caller sums `stats` over ranks (one RCCL all-reduce of
 * n_runs * dic_kmeans_stats_words(D, K) doubles), and dic_kmeans_lloyd_finish completes the iteration identically on
 * every rank.  stats[run] = [ sums (K*D) | counts (K) | #labels changed (1) ], f64.  An empty cluster cannot be
 * relocated from partial data (the farthest point may live on another rank): finish then sets status[0] = 2 and leaves
 * the centres untouched; the caller re-runs that restart on unsharded data with dic_kmeans_lloyd_iter. */
size_t dic_kmeans_stats_words(int D, int K);
int dic_kmeans_lloyd_partial(const float* X, const float* xnorm, int N, int D, int K, int n_runs, const float* centers,
                             int32_t* labels, const float* status, double* stats, void* workspace, size_t workspace_bytes,
                             dic_stream_t stream);
int dic_kmeans_lloyd_finish(int D, int K, int n_runs, const double* stats, float* centers, float* status, dic_stream_t stream);
/* E-step only (KMeans.predict; also the final E-step after a tol stop):
 *   labels (n_runs,N) OVERWRITTEN; mindist (n_runs,N) or NULL = exact ||x-c_label||^2;
 *   inertia (n_runs) or NULL = sum of mindist (deterministic, f64 accumulation, f32 result). */
int dic_kmeans_predict(const float* X, int N, int D, int K, int n_runs, const float* centers,
                       int32_t* labels, float* mindist, float* inertia,
                       void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* k-means++ helper (sklearn _kmeans.py:218-243 inner step), batched over restarts: L candidate rows
 * `cand` (int64 indices into X) in groups of `group` consecutive candidates per restart;
 * closest (L/group, N) = current closest squared distances of each restart (+inf for the first centre);
 * dist_out (L,N) = min(closest[l/group], ||x_i - X[cand_l]||^2), distances evaluated in f64 like
 * sklearn's float32 up-cast path; pot_out (L) f64 = sum_i dist_out[l,i]. */
size_t dic_kmeans_pp_workspace(int N, int L);
int dic_kmeans_pp_candidates(const float* X, int N, int D, const int64_t* cand, int L, int group,
                             const float* closest, float* dist_out, double* pot_out,
                             void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* The same for the rows [row_lo, row_hi) only (points sharded over ranks): X, closest and dist_out keep their full
 * (., N) shapes, only those columns are read / written, pot_out = the partial potential of the range (the caller sums dist_out
 * and pot_out over ranks).  Candidate rows are still addressed in the full X. */
int dic_kmeans_pp_candidates_rows(const float* X, int N, int D, int row_lo, int row_hi, const int64_t* cand, int L, int group,
                                  const float* closest, float* dist_out, double* pot_out,
                                  void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* out (n_rows, n) f64 = inclusive running sums along the rows of x (n_rows, n) f32, in f64: the stable_cumsum of scikit-learn's k-means++ sampling
 * (sklearn/cluster/_kmeans.py:218-243, as called from clustering_trainer.py:75-82 and p2_clustering_optK.py:260-389); deterministic summation order. */
int dic_cumsum_f64(const float* x, int n_rows, int n, double* out, dic_stream_t stream);

/* ------------------------------------------------------------------ bi-LSTM recurrence -----
 * Sequential half of one bidirectional torch.nn.LSTM layer with hidden size H = 128 (EncoderRNN /
 * DecoderRNN, clustering_interp.py:14-41; gate order i,f,g,o).  The caller (lstm.py) issues the
 * time-parallel GEMMs; bf16 tensors are passed as void*:
 *   gx    (R,B,2,4,H) bf16  X.W_ih^T + b_ih + b_hh per direction (0 = forward, 1 = reverse)
 *   whh   (2,4H,H)    bf16  recurrent weights;   whh_t (2,H,4H) bf16 their per-direction transposes
 *   h0,c0 (2,B,H) f32 or NULL;   out (R,B,2H) bf16;   hn,cn (2,B,H) f32
 *   gates (R,Bp,2,4,H) bf16 post-activation gates and cs (R,Bp,2,H) bf16 cell states (the forward carries c in f32), Bp = B rounded up to
 *   64: opaque, lane-native layout exchanged only between these two calls (both NULL for inference).
 * Backward: dout (R,B,2H) bf16 / dhn, dcn (2,B,H) f32 (each may be NULL) -> dgx (R,B,2,4,H) bf16
 * pre-activation gate gradients (the caller turns them into dX, dW_ih, dW_hh by GEMMs), dh0, dc0, and
 * dbias (2,4H) f32 or NULL = sum of dG over steps and batch rows (needs dic_lstm_bwd_workspace(B) bytes).
 * state_batch_major != 0: h0, c0, hn, cn and their gradients are laid out (B,2,H) instead of nn.LSTM's (2,B,H), so that
 * hn viewed as (B,2H) is the concatenated latent [h_fwd | h_rev] (clustering_interp.py:139) and feeds the decoder with no copy. */
int dic_lstm_fwd(const void* gx, int gx_lane_native, const void* whh, const float* h0, const float* c0, int R, int B, int H,
                 void* out, void* out_relu, float* hn, float* cn, void* gates, void* cs, int state_batch_major, int write_boundary,
                 dic_stream_t stream);
/* gx_lane_native == 2: the same with eight waves per workgroup (two per SIMD, 16 hidden units each; bit-identical results).
 * gx_lane_native != 0 (B a multiple of 64): gx is not row-major but in the opaque form dic_row_proj(..., lane_native_batch = B) writes --
 * the order of the recurrence kernel's MFMA accumulators, 1-KiB pieces per wave access -- and is read straight into registers a step
 * ahead (no LDS staging of the gx tile, one barrier per step). */
/* write_boundary != 0: `out` points at time slot 1 of an (R+2,B,2H) buffer (dic_lstm_dw's out_ext); the kernel also writes h0 (bf16;
 * zeros without one) into slot 0 [:, :H] and slot R+1 [:, H:], the recurrent inputs of the first forward / reverse step. */
/* out_relu (R,B,2H) bf16 or NULL: a second copy of the output with relu applied -- what DecoderRNN.forward reads of the encoder's
 * output (clustering_interp.py:38-41).  It leaves through the LDS tile of h the next step multiplies, as whole 256-B row halves;
 * the element-wise ReLU pass over (R,B,2H) disappears (its backward: dic_lstm_bwd's dout_of_relu). */
/* Same recurrence with the input projection computed in-kernel, for narrow inputs (the encoder's 3C channels):
 * G_t = x_t.wih^T + h_{t-1}.whh^T with x (R,B,I) bf16 and wih (2,4H,I) bf16, I == 32 (zero-pad narrower inputs; fold
 * the bias in as a constant-one input column whose weights are b_ih + b_hh).  Everything else as dic_lstm_fwd; the
 * backward is dic_lstm_bwd unchanged. */
int dic_lstm_fwd_proj(const void* x, const void* wih, const void* whh, const float* h0, const float* c0, int R, int B, int H,
                      int I, void* out, void* out_relu, float* hn, float* cn, void* gates, void* cs, int state_batch_major, int write_boundary,
                      int eight_waves, dic_stream_t stream);
/* eight_waves != 0: 512-thread workgroups, two waves per SIMD with 16 hidden units each (same results bit for bit): this variant's
 * per-step chain, not its bandwidth, bounds it, and the second wave on a SIMD fills the first one's gaps. */
size_t dic_lstm_bwd_workspace(int B);
int dic_lstm_bwd(const void* whh_t, const void* gates, const void* cs, const float* c0, const void* dout,
                 const float* dhn, const float* dcn, int R, int B, int H, void* dgx, float* dh0, float* dc0,
                 float* dbias, void* workspace, size_t workspace_bytes, int state_batch_major, int dout_of_relu, dic_stream_t stream);
/* dout_of_relu != 0: the consumer of `out` was relu(out) (the decoder rectifies the encoder output, clustering_interp.py:38-41) and
 * dout is the gradient of THAT: the kernel passes it where h_t > 0, read off the sign of tanh(c_t) it computes anyway (o_t is a
 * sigmoid) -- the element-wise ReLU backward pass over (R,B,2H) disappears. */

/* The same recurrence for the two regimes the 64-row bf16 kernels above do not serve (csrc/dic_lstm32.hip): dtype =
 * DIC_DTYPE_F32 -- every tensor f32, the recurrent product on v_mfma_f32_32x32x2_f32 (exact f32: the configuration of the 1e-5
 * parity tests, which round 1 left on MIOpen's nn.LSTM) -- and dtype = DIC_DTYPE_BF16 for small batches (the reference's own
 * B = 256: one 32-row tile per workgroup instead of two; round 4: one 16-row tile up to 2048 rows, v_mfma_f32_16x16x32_bf16, same results bit
 * for bit).  Element type T of gx, whh, out, gates, cs, dout, dgx follows dtype -- except that with DIC_DTYPE_F32X3 (round 6: eight waves per 32-row tile,
 * any batch size; gate non-linearities on the transcendental unit, ~3e-7 relative) dgx leaves as SPLIT PLANES: (2, R*B, 2*4H) bf16, plane 0 = bf16(dG),
 * plane 1 = bf16(dG - plane 0) -- the f32 tensor's bytes in the form dic_gemm_tn_planes / dic_gemm_nt_planes multiply without converting;
 * gates (R,Bp,2,4,H) and cs (R+1,Bp,2,H), Bp = B rounded up to 32, are an opaque lane-native layout exchanged between the two
 * calls of one (dtype, B) -- the 16- and 32-row kernels order it differently, and which pair runs is a function of B and the process environment
 * (DIC_REC_SIXTEEN, DIC_REC16_MAX) alone -- (time slot R of cs carries c0, so the backward takes no c0).  whh (2,4H,H); the backward takes either that (read transposed
 * once at start-up) or the transposed copy (2,H,4H) with whh_is_transposed != 0.
 * state_flags of the forward: bit 0 = the state tensors (h0, c0, hn, cn) are batch-major (B,2,H); bit 1 (round 3) = `out` is time slots
 * 1..R of an (R+2,B,2H) buffer and the kernel also writes h0 (zeros without one) into slot 0 [:, :H] / slot R+1 [:, H:] -- every step's
 * recurrent input for the weight-gradient kernels, as dic_lstm_fwd's `boundary` does (two fill / copy launches per LSTM less). */
#define DIC_DTYPE_F32 0
#define DIC_DTYPE_BF16 1
#define DIC_DTYPE_F32X3 2   /* dic_lstm_rec_fwd / _bwd only: f32 tensors, the recurrent products as three-term bf16 splits (hi.hi + lo.hi + hi.lo) on the bf16 matrix cores */
int dic_lstm_rec_fwd(int dtype, const void* gx, const void* whh, const float* h0, const float* c0, int R, int B, int H, void* out,
                     float* hn, float* cn, void* gates, void* cs, int state_flags, dic_stream_t stream);
/* dic_lstm_rec_fwd_proj (bf16, batches up to 4096): the same 32-row forward recurrence with the encoder's input projection inside the kernel -- x (R,B,I)
 * bf16 packed rows [3C features | 1 | 0...] as dic_sci_cci_fwd_packed writes them (I = 32 or 64), wih (2*4H, I) bf16 from dic_lstm_pack(bias_col = 1):
 * the (R,B,8H) gx tensor and the projection launch of nn.LSTM (clustering_interp.py:22) do not exist.  Other arguments as dic_lstm_rec_fwd. */
int dic_lstm_rec_fwd_proj(const void* x, const void* wih, const void* whh, const float* h0, const float* c0, int R, int B, int H, int I, void* out,
                          float* hn, float* cn, void* gates, void* cs, int state_flags, dic_stream_t stream);

/* dic_lstm_rec_fwd_proj_x3 (round 6): the x3 (DIC_DTYPE_F32X3) forward recurrence -- eight waves per 32-row tile, any batch size -- with the NARROW input
 * projection inside the kernel: x (R,B,ldx) f32 rows [features | 1 | 0...], wih (2*4H, ldx) f32 rows [W_ih | b_ih + b_hh | 0...] (dic_lstm_pack(DIC_DTYPE_F32,
 * bias_col = 1)), ldx a multiple of 4 up to 32; both are split into bf16 hi + lo on their way into LDS / registers and the product is hi.hi + lo.hi + hi.lo
 * like the recurrent one.  The f32 gx tensor of nn.LSTM's input projection (clustering_interp.py:22; 3.2 GB at 32 768 encounters) does not exist.  Other
 * arguments as dic_lstm_rec_fwd with f32 tensors. */
int dic_lstm_rec_fwd_proj_x3(const float* x, const float* wih, int ldx, const float* whh, const float* h0, const float* c0, int R, int B, int H, float* out,
                             float* hn, float* cn, float* gates, float* cs, int state_flags, dic_stream_t stream);

/* dic_lstm_fwd_xproj (bf16, H = 128, I = 256: the decoder; clustering_interp.py:30-41): the forward recurrence with the input projection x W_ih^T + b inside
 * the kernel -- gx (R,B,8H) is never formed.  x (R,B,256) bf16 raw rows (relu_x != 0: rectified on load, the F.relu between encoder and decoder), wih (2,4H,256),
 * whh (2,4H,H), bias (2,4H) bf16 as dic_lstm_pack writes them.  gates (R,Bp,2,4,H) / cs (R,Bp,2,H), Bp = B rounded up to 64, leave in the lane-native order
 * dic_lstm_bwd reads (c0 goes to the backward separately, as with dic_lstm_fwd).  out_r: optional (R,B,2H) relu(out), as dic_lstm_fwd's.  state_flags as
 * dic_lstm_rec_fwd. */
int dic_lstm_fwd_xproj(const void* x, const void* wih, const void* whh, const void* bias, const float* h0, const float* c0, int R, int B, int H, int I,
                       void* out, void* out_r, float* hn, float* cn, void* gates, void* cs, int state_flags, int relu_x, dic_stream_t stream);

size_t dic_lstm_rec_bwd_workspace(int B);
int dic_lstm_rec_bwd(int dtype, const void* whh, int whh_is_transposed, const void* gates, const void* cs, const void* dout,
                     const float* dhn, const float* dcn, int R, int B, int H, void* dgx, float* dh0, float* dc0, float* dbias,
                     void* workspace, size_t workspace_bytes, int state_batch_major, int dout_of_relu, dic_stream_t stream);

/* dic_lstm_dx_tile: the decoder LSTM's input gradient dX = dG . W_ih (the backward of nn.LSTM's input projection, clustering_interp.py:29-41; dg (N, 1024)
 * bf16 gate gradients of both directions as dic_lstm_bwd writes them) on 256 x 256 macro-tiles with nothing resident -- one persistent 8-wave workgroup per CU, dG slabs
 * (HBM) through a 3-slot and W_ih^T slabs (L2) through a 2-slot all-LDS-DMA ring, 16-B stores straight from the accumulators: every CU takes in its
 * row share of dG once.  w_ih_t (in_features = 256, gate_columns = 1024) bf16 = W_ih TRANSPOSED (k contiguous, like the rows of dg); dx (N, 256)
 * bf16 OVERWRITTEN.  N >= 256.  Replaces the last library GEMM of the bf16 step (nn.LSTM's backward: clustering_interp.py:29-41). */
int dic_lstm_dx_tile(const void* dg, const void* w_ih_t, int64_t N, int gate_columns, int in_features, void* dx, dic_stream_t stream);
/* dic_lstm_dx_tile_x3 (round 6): the same product for the f32 step on split products -- dX (N, 256) F32 = dG . W_ih with dG as the two bf16 planes
 * dic_lstm_rec_bwd(DIC_DTYPE_F32X3) writes (hi at dg_hi, lo dg_plane elements behind it, (N, 1024) each) and W_ih^T (256, 1024) split the same way by the
 * caller (hi at w_ih_t_hi, lo wt_plane elements behind it); every product hi.hi + lo.hi + hi.lo.  Same macro-tiles and LDS-DMA rings, a slab = both planes of
 * a 32-deep k range.  N >= 256.  Replaces dic_gemm_nt on the f32 operand (nn.LSTM's backward, clustering_interp.py:29-41). */
int dic_lstm_dx_tile_x3(const void* dg_hi, long dg_plane, const void* w_ih_t_hi, long wt_plane, int64_t N, int gate_columns, int in_features, float* dx,
                        dic_stream_t stream);

/* ------------------------------------------------------------------ bi-LSTM parameters --------
 * The eight f32 parameters of one bidirectional nn.LSTM layer (clustering_interp.py:22,35: weight_ih_l0, weight_hh_l0,
 * bias_ih_l0, bias_hh_l0, then the same four with the _reverse suffix) are passed as a HOST array of 8 device pointers in
 * that order.
 *   dic_lstm_pack: (dtype = DIC_DTYPE_BF16 or DIC_DTYPE_F32 selects the OUTPUT element type) -> wih (2*4H, Ip) bf16 [columns [0,I) = W_ih; column I = b_ih + b_hh when bias_col (dic_lstm_fwd_proj's
 *     constant-one input column); rest 0], whh (2,4H,H) bf16, whh_t (2,H,4H) bf16 or NULL, bias (2*4H) bf16 = b_ih + b_hh
 *     or NULL, wih_t (Ip, 2*4H) = wih transposed (dic_lstm_dx_tile's operand) or NULL.  One launch replaces the stack / add / cast / pad / transpose sequence of torch ops.
 *   dic_lstm_dw (encoder, packed input width Ip == 32): weight gradients from ONE pass over the gate gradients,
 *     dW_hh[d] = sum_t dG_t[d]^T h_prev_t[d] (h_{t-1} for d = 0, h_{t+1} for d = 1) and dW_ih[d] = sum_t dG_t[d]^T x_t[:, :I]
 *     (x (R,B,Ip) bf16).  out_ext (R+2,B,2H) bf16 = the layer's own output in time slots 1..R (pass &out_ext[1] to
 *     dic_lstm_fwd*), slot 0 [:, :H] = h0 of the forward direction and slot R+1 [:, H:] = h0 of the reverse direction (zeros
 *     without an h0): every step's recurrent input is then the row B above / below its dG row.  MFMA with transposed LDS
 *     reads.  With wih (2*4H, Ip) bf16 (dic_lstm_pack's) and dx_parts (2, R*B, Ip) bf16 given (both or neither), the same pass also
 *     writes the per-direction input gradients dx_parts[d] = dG[d] . W_ih[d] (columns [0, I), I <= 19; the caller adds the two
 *     directions) -- the third GEMM that used to re-read dG.  Deterministic two-stage reduction of the weight gradients, written
 *     (accumulate = 0) or added (accumulate = 1) straight into the
 *     parameter gradients `grads` (host array of 8 device pointers, same order; the bias entries are not touched).
 *   dic_lstm_dw_wide (decoder, input width I == 256 = the rectified encoder outputs, x (R,B,256) bf16): the same one-pass weight
 *     gradients -- four workgroups (direction x half of the gate rows) per chunk of rows, placed on one XCD so that the x / h tiles they
 *     share are served by its L2 -- instead of the three split-K library products that each re-read dG (decoder dW 0.90 -> see DESIGN.md).
 *   dic_lstm_unpack_grads: staging tensors dw_ih (2*4H, ldw) / dw_hh (2,4H,H) / dbias (2*4H) f32 (each may be NULL) ->
 *     the parameter gradients (dbias goes to bias_ih AND bias_hh). */
int dic_lstm_pack(int dtype, const float* const* params, int H, int I, int Ip, int bias_col, void* wih, void* whh, void* whh_t, void* bias,
                  void* wih_t, dic_stream_t stream);
size_t dic_lstm_dw_workspace(int R, int B);
int dic_lstm_dw(const void* dgx, const void* out_ext, const void* x, const void* wih, void* dx_parts, int R, int B, int H, int I, int Ip,
                float* const* grads, int accumulate, void* workspace, size_t workspace_bytes, dic_stream_t stream);
size_t dic_lstm_dw_wide_workspace(int R, int B);
int dic_lstm_dw_wide(const void* dgx, const void* out_ext, const void* x, int x_relu, int R, int B, int H, int I, float* const* grads, int accumulate,
                     void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* dic_lstm_dw_x3 (round 6): the same weight gradients for the f32 step on split products (DIC_DTYPE_F32X3) -- dW_ih[d] = dG[d]^T . x, dW_hh[d] = dG[d]^T . h_prev[d],
 * every product hi.hi + lo.hi + hi.lo -- from ONE pass over the gate gradients in the form dic_lstm_rec_bwd(DIC_DTYPE_F32X3) writes them: two bf16 planes, hi at
 * dg_hi and lo dg_plane elements behind it, (R*B, 2*4H) each, streamed by LDS-DMA as they lie.  out_ext ((R+2)*B, 2H) f32 and x (R*B, ldx) f32 as the other f32
 * kernels take them (x_relu != 0: the LSTM ran on relu(x)).  I = 256 (the decoder; ldx = 256) or I <= ldx <= 32 with ldx a multiple of 4 (the encoder's rows
 * [features | 1 | 0...]; only columns [0, I) reach weight_ih).  R*B >= 16.  Narrow input, wih (2*4H, ldx) f32 (dic_lstm_pack) and dx_parts (4, R*B, ldx) f32 given:
 * four partial input gradients -- one per (direction, half of its gate rows) -- come out of the same pass (their sum is dX = dG . W_ih).  Replaces dic_gemm_tn / dic_gemm_nt on the f32 tensors (nn.LSTM's
 * backward, clustering_interp.py:14-41). */
size_t dic_lstm_dw_x3_workspace(int R, int B, int I);
int dic_lstm_dw_x3(const void* dg_hi, long dg_plane, const float* out_ext, const float* x, int ldx, int x_relu, const float* wih, float* dx_parts, int R, int B, int H,
                   int I, float* const* grads, int accumulate, void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_lstm_unpack_grads(const float* dw_ih, int ldw, const float* dw_hh, const float* dbias, int H, int I, float* const* grads,
                          int accumulate, dic_stream_t stream);

/* ------------------------------------------------------------------ row-wise projection, 256 inputs ---
 * out (N,Nout) bf16 = x (N,256) . w (Nout,256)^T + bias (Nout) (bias bf16 or NULL), Nout a multiple of 256: the decoder LSTM's input
 * projection gx over all N = R*B rows (nn.LSTM inside DecoderRNN, clustering_interp.py:47-59) with the weights resident in
 * registers -- the library GEMM for this K = 256 shape cannot overlap its short main loop with its epilogue. */
int dic_row_proj(const void* x, const void* w, const void* bias, int64_t N, int in_features, int out_features, void* out, int lane_native_batch,
                 int relu_input, dic_stream_t stream);
/* relu_input != 0 (and x_relu of dic_lstm_dw_wide): x is the RAW output of the encoder LSTM and the product is taken of relu(x) -- the F.relu
 * between encoder and decoder (clustering_interp.py:38-41) applied to the operand on its way through the kernel, so that no rectified copy of
 * the encoder output exists in HBM. */
/* lane_native_batch = B > 0 (Nout = 1024 = 2 directions x 4 gates x 128 units, N = R*B, B a multiple of 64): out is written in the
 * lane-native form dic_lstm_fwd(gx_lane_native = 1) reads (same bytes, same values, another order); 0: row-major (N,Nout). */
/* The same for CompressFC's first layer Linear(256, 128) (rbf.py:111-125) in front of its training-mode BatchNorm1d: out (N,128) bf16 and
 * sums (2*128 + 1) f64 = [column sums of out | of out^2 | N] -- what dic_bn_colstats would compute in a second pass over out. */
size_t dic_row_proj_stats_workspace(int64_t N, int out_features);
int dic_row_proj_stats(const void* x, const void* w, const void* bias, int64_t N, int in_features, int out_features, void* out, double* sums,
                       void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* ------------------------------------------------------------------ CompressFC first layer, backward ---
 * Linear(256, 128) over all N = B*R decoder rows (rbf.py:111-125, first layer; TimeDistributed utils.py:202-224): from ONE pass over
 * the rows, dx (N,256) bf16 = dz . W (or NULL) and dw (128,256) f32 = dz^T . x, for dz (N,128), x (N,256), w (128,256) bf16 -- instead of
 * two library GEMMs that each read dz.  MFMA with transposed LDS reads for the weight gradient; deterministic two-stage reduction. */
size_t dic_fc_bwd_workspace(int64_t N, int in_features, int out_features);
int dic_fc_bwd(const void* dz, const void* x, const void* w, int64_t N, int in_features, int out_features, void* dx, float* dw,
               void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* The same with dz formed inside the kernel instead of read: the layer's own output z (N,128) bf16 goes through the backward of the
 * tail behind it -- BatchNorm1d(128, training) -> ReLU -> Dropout -> Linear(128, C) (rbf.py:114-124) -- on its way into LDS, with the
 * arithmetic of dic_bnhead_bwd_input (bit-identical dz): dv (N,C) f32 the gradient of the tail's output; mean, rstd, gamma, beta (128),
 * w2 (C,128) f32; sum_da, sum_dax (128) the column sums of dic_bnhead_bwd_reduce (summed over ranks); count (1) the global row count
 * of the batch moments (dic_bn_moments); relu / drop_p / rng as in dic_bnhead_bwd_input.  Compiled for C = 6. */
int dic_fc_bwd_bnhead(const void* z, const float* dv, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2,
                      const float* sum_da, const float* sum_dax, const float* count, int C, int relu, float drop_p, const uint64_t* rng,
                      const void* x, const void* w, int64_t N, int in_features, int out_features, void* dx, float* dw,
                      void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* ------------------------------------------------------------------ CompressFC output layer ---
 * Linear(128, C) over all N = B*R decoder rows (rbf.py:111-125, last layer; TimeDistributed utils.py:202-224)
 * for small C (1..8, 12, 16): h (N,128) bf16, W (C,128) f32, b (C) f32 -> v (N,C) f32; backward:
 * dv (N,C) f32 -> dh (N,128) bf16, dW (C,128), db (C). */
int dic_head_fwd(const void* h, const float* W, const float* b, int64_t N, int K, int C, float* v, dic_stream_t stream);
size_t dic_head_bwd_workspace(int64_t N, int K, int C);
int dic_head_bwd(const void* h, const float* W, const float* dv, int64_t N, int K, int C, void* dh, float* dW, float* db,
                 void* workspace, size_t workspace_bytes, dic_stream_t stream);

/* ------------------------------------------------------------- CompressFC tail, fused ---------
 * BatchNorm1d(128) -> ReLU -> Linear(128, C) (rbf.py:116-123 with dropout p = 0 or eval mode) over N rows
 * without materialising the hidden activation.  z (N,128) bf16 = output of the first Linear; mean, rstd,
 * gamma, beta (128) f32; W (C,128), b (C) f32; C in 1..8.
 *   dic_bn_colstats:       sums[0:128] = sum_rows z, sums[128:256] = sum_rows z^2, sums[256] = N (f64, 257 entries; the caller
 *                          all-reduces the record first when the batch is sharded over ranks)
 *   dic_bn_moments:        that record -> mean, rstd = 1/sqrt(biased var + eps) (128 f32 each), count (1 f32), and -- when
 *                          running_mean / running_var are given -- nn.BatchNorm1d's running statistics (momentum < 0 =
 *                          cumulative average) and num_batches_tracked += 1, all in one launch
 *   dic_bnhead_fwd:        v (N,C) f32 = b + W relu(gamma (z - mean) rstd + beta)
 *   dic_bnhead_bwd_reduce: sums[(2+C)*128 + C] f32 = sum da | sum da*xhat | dW (C,128) | db (C), where
 *                          da = (dv W) 1[h > 0]; the first two rows are d beta and d gamma
 *   dic_bnhead_bwd_input:  dz (N,128) bf16 = gamma rstd (da - sum_da*inv_n - xhat sum_dax*inv_n); inv_n = 1 /
 *                          global row count in training mode (count != NULL: read from that device scalar instead); pass
 *                          zero sums for eval-mode BatchNorm.
 * relu = 0 drops the ReLU: BatchNorm1d(128) -> Linear(128, C), the tail of the auxiliary / fake-detection heads
 * (clustering_interp.py:43-87).  drop_p > 0 applies nn.Dropout(p) between the activation and the Linear (rbf.py:120): the
 * keep mask is a hash of (rng[0] = seed, rng[1] = call counter, element index) -- rng is a 2-word device buffer that the
 * caller keeps unchanged between a forward and its backward calls -- so no mask tensor exists. */
size_t dic_bn_colstats_workspace(int64_t N, int K);
int dic_bn_colstats(const void* z, int64_t N, int K, double* sums, void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_bn_moments(const double* sums, int K, float eps, float momentum, float* running_mean, float* running_var,
                   int64_t* num_batches_tracked, float* mean, float* rstd, float* count, dic_stream_t stream);
int dic_bnhead_fwd(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                   const float* b, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* v, dic_stream_t stream);
size_t dic_bnhead_bwd_workspace(int64_t N, int K, int C);
int dic_bnhead_bwd_reduce(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                          const float* dv, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* sums,
                          void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_bnhead_bwd_input(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                         const float* dv, const float* sum_da, const float* sum_dax, double inv_n, const float* count, int64_t N, int K, int C, int relu,
                         float drop_p, const uint64_t* rng, void* dz, dic_stream_t stream);
/* The same four kernels on an f32 (N,128) pre-activation -- the f32 step ('x3' products: ops.mfma_linear writes z in f32): every value of the
 * BatchNorm -> ReLU -> Dropout -> Linear tail stays f32, no torch BatchNorm kernels, no library GEMM for the C-wide head. */
int dic_bn_colstats_f32(const float* z, int64_t N, int K, double* sums, void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_bnhead_fwd_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                       const float* b, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* v, dic_stream_t stream);
int dic_bnhead_bwd_reduce_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                              const float* dv, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* sums,
                              void* workspace, size_t workspace_bytes, dic_stream_t stream);
int dic_bnhead_bwd_input_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                             const float* dv, const float* sum_da, const float* sum_dax, double inv_n, const float* count, int64_t N, int K, int C, int relu,
                             float drop_p, const uint64_t* rng, float* dz, dic_stream_t stream);


/* ------------------------------------------------- K-sweep statistics (p2, internal_eval) ------
 * One pass over all N^2 point pairs, nothing n x n materialised.  Replaces sklearn pairwise_distances per
 * cluster (p2_clustering_optK.py:334-351, gap-statistic inertia) and the distance work inside
 * silhouette_score / DunnIndex (internal_eval.py:37-126).
 * X (N,D) f32 with rows SORTED by cluster: cluster k = rows seg[k] .. seg[k+1]-1, seg (K+1) int32 on the
 * device, seg[0] = 0, seg[K] = N; D % 4 == 0; K <= 64.
 *   S (N,K) f32:     S[i][k] = sum over j in cluster k of ||x_i - x_j||   (f32 direct-difference distances)
 *   Dmin (N,K) f32:  Dmin[i][k] = min over j in cluster k of ||x_i - x_j|| (+inf for an empty cluster) [may be NULL]
 *   own_max (N):     farthest point of i's own cluster (0 for singletons)                              [may be NULL] */
int dic_cluster_pairdist(const float* X, const int32_t* seg, int N, int D, int K, float* S, float* Dmin, float* own_max,
                         dic_stream_t stream);

/* The same pass restricted to the pairs INSIDE a cluster -- all the gap statistic's inertia of a reference set needs (p2_clustering_optK.py:
 * 334-351: np.mean / np.sum of pairwise_distances(X[a == c]) per cluster): sum_c n_c^2 pairs instead of N^2.
 *   S_own (N) OVERWRITTEN: S_own[i] = sum over the points j of i's own cluster of ||x_i - x_j||  (rows sorted by cluster, as above). */
int dic_cluster_intra_sums(const float* X, const int32_t* seg, int N, int D, int K, float* S_own, dic_stream_t stream);
/* The per-cluster TOTALS of that pass on the matrix cores (round 6):  totals[c] (K) f64 OVERWRITTEN = sum over the ordered pairs (i, j) of cluster c of
 * ||x_i - x_j|| = np.sum(pairwise_distances(X[a == c])) (p2_clustering_optK.py:334-351) -- all the inertia of a gap-statistic reference set needs.  d^2 =
 * ||x_i||^2 + ||x_j||^2 - 2 x_i . x_j with x taken relative to centres[c] (K, D) f32 (any point near the cluster: its centroid) as ONE bf16 MFMA inner product of
 * augmented rows, every coordinate as hi + lo bf16 with the three leading partial products (the norms enter exactly): within 2e-7 of the f64 sum on the golden
 * clusterings.  X (N, D) f32 at row stride ldx, rows sorted by cluster as above, D <= 256, D % 4 == 0.  tiles (ntiles, 4) int32 on the device, sorted by
 * cluster: (first row of I, first row of J, end row of the cluster, cluster) for every pair of 256-row blocks I <= J of one cluster (block b of cluster c starts
 * at seg[c] + 256 b).  Only these pairs are visited (sum_c n_c^2 / 2 of the N^2).  Deterministic.  workspace: dic_cluster_intra_totals_workspace(N, K) bytes. */
/* The ROW SUMS of the all-pairs pass on the same machine (round 6):  S (N, K) f32 OVERWRITTEN, S[i][c] = sum over the points j of cluster c of ||x_i - x_j||
 * -- dic_cluster_pairdist's S without Dmin / own_max: what silhouette_score needs (internal_eval.py:112-123), 12 instead of 67 ms on 75 000 x 256.  Points
 * relative to ONE centre (1, D) f32 (their mean).  tiles (ntiles, 4) int32 on the device: (first row of a 256-row block I, first row of a 256-row block J of
 * cluster c, end row of c, output slot), sorted by (I, c, J); workgroup b of min(ntiles, 256) takes tiles [b per, (b + 1) per), per = ceil(ntiles / workgroups),
 * and a slot is a maximal run of one (I, c) inside one such range (slots numbered in list order).  group_start (blocks of I x K + 1) int32: the first slot of
 * every (I, c), in that order.  Deterministic.  workspace: dic_cluster_pair_rowsums_workspace(N, n_slots) bytes.  (Host side: cluster_stats._row_tile_list.) */
size_t dic_cluster_pair_rowsums_workspace(int64_t N, int n_slots);
int dic_cluster_pair_rowsums(const float* X, long ldx, const float* centre, int64_t N, int D, int K, const int32_t* tiles, int ntiles, const int32_t* group_start,
                             int n_slots, float* S, void* workspace, size_t workspace_bytes, dic_stream_t stream);
size_t dic_cluster_intra_totals_workspace(int64_t N, int K);
int dic_cluster_intra_totals(const float* X, long ldx, const int32_t* seg, const float* centres, int64_t N, int D, int K, const int32_t* tiles, int ntiles,
                             double* totals, void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* out (K, D) f64 = per-cluster sums of the rows of values (N, D) f64 at row stride ldv, labels (N) int64 in [0, K) (others are skipped): the centroids of
 * scikit-learn's calinski_harabasz_score / davies_bouldin_score and the per-cluster distance sums of p2's gap statistic (internal_eval.py:112-147,
 * p2_clustering_optK.py:334-351).  Deterministic (no atomics, fixed order).  workspace: dic_segment_sum_workspace(D, K) bytes.  K <= 64. */
size_t dic_segment_sum_workspace(int D, int K);
int dic_segment_sum_f64(const double* values, long ldv, const int64_t* labels, int N, int D, int K, double* out, void* workspace, size_t workspace_bytes,
                        dic_stream_t stream);

/* ------------------------------------------------------------- optimisation step tail ---------
 * clip_grad_norm_ scaling + torch.optim.Adam(amsgrad=True, L2 weight decay) (pretrain_trainer.py:228-229, utils.py:83)
 * over one flat f32 bucket of n elements: p, g (scaled in place by *grad_scale, NULL = 1), exp_avg m, exp_avg_sq v,
 * max_exp_avg_sq vmax.  `step` = device pointer to the already incremented step count t (f32); `grad_scale` = device
 * pointer to the clip coefficient min(1, max_norm / (||g|| + 1e-6)); `active` = per-element byte mask or NULL (all):
 * elements with 0 are left untouched, as torch.optim skips parameters whose .grad is None.  `hyper` = device pointer to
 * [lr, beta1, beta2, eps, weight_decay] overriding the scalar arguments, or NULL: read on the device so that one captured hipGraph
 * of the step follows a learning-rate scheduler. */
int dic_adam_amsgrad_step(float* p, float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, const float* step, const float* grad_scale, const unsigned char* active, const float* hyper,
                          dic_stream_t stream);
/* torch.nn.utils.clip_grad_norm_'s scalars for the flat gradient bucket (pretrain_trainer.py:228): out2[0] = ||g||_2,
 * out2[1] = min(1, max_norm / (||g||_2 + 1e-6)) -- the `grad_scale` of dic_adam_amsgrad_step.  Deterministic two-stage sum. */
size_t dic_grad_norm_workspace(int64_t n);
int dic_grad_norm_clip(const float* g, int64_t n, float max_norm, float* out2, void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* dst[j][0..n[j]) += src[j][0..n[j]) for j < count (host arrays of device pointers / sizes) in one launch per 16 entries: the small
 * parameter gradients of the hot-path kernels added into their slots of the flat gradient bucket (what autograd's AccumulateGrad does
 * with one launch each between loss.backward() and the optimizer, pretrain_trainer.py:223-231). */
int dic_accumulate_many(const float* const* src, float* const* dst, const int* n, int count, dic_stream_t stream);

/* ---- row-streaming MFMA products (csrc/dic_gemm.hip) for the dense layers in the shapes no specialised kernel covers, and for the f32 step.
 * Replaces the library GEMMs behind nn.LSTM's input projections / nn.Linear (clustering_interp.py:14-41, rbf.py:111-125) and their
 * autograd gradients.  in_dtype DIC_DTYPE_BF16: one bf16 MFMA per product.  DIC_DTYPE_F32: operands split on the fly into bf16 hi + lo
 * pieces and multiplied as hi.hi + lo.hi + hi.lo in f32 accumulators ("bf16x3": products good to ~2^-17, at 5x the exact-f32 MFMA rate).
 *   dic_gemm_nt:  Y[m][n] = sum_k act(A[m][k]) W[n][k] (+ bias[n]);  A (M,K) rows at stride lda, W (N,K) rows at stride ldw (both of
 *                 in_dtype), Y (M,N) of out_dtype at stride ldy; relu_a != 0: act = max(., 0).  K, lda, ldw multiples of 4 (f32) / 8 (bf16)
 *                 elements, operands 16-B aligned.  An input gradient dX = dY.W is the same call with W handed over transposed.
 *   dic_gemm_tn:  D[n][k] (+)= sum_m A[m][n] X[m][k] for k < kcols;  A (M,N) at stride lda, X (M,K) at stride ldx (both in_dtype), D f32
 *                 (N,kcols) at stride ldd; the M rows are cut into chunks whose f32 partial products go through `workspace`
 *                 (dic_gemm_tn_workspace bytes) and are summed in a fixed order in f64: deterministic.  accumulate != 0: added to D. */
int dic_gemm_nt(int in_dtype, int out_dtype, const void* A, long lda, const void* W, long ldw, const float* bias, long M, int N, int K,
                void* Y, long ldy, int relu_a, dic_stream_t stream);
/* dic_x3_row_proj: out (N, out_features) f32 = act(x (N,256) f32) . w (out_features,256)^T + bias, every product a three-term bf16 split, for the
 * 256-input projections of the x3 f32 step (the decoder LSTM's input projection, out_features = 1024; CompressFC's Linear(256,128)): weights split once and
 * resident in registers, x tiles streamed through LDS (csrc/dic_gemm.hip).  out_features: 128 or a multiple of 256; relu_input != 0: act = max(., 0). */
int dic_x3_row_proj(const float* x, const float* w, const float* bias, int64_t N, int in_features, int out_features, float* out, int relu_input,
                    dic_stream_t stream);
size_t dic_gemm_tn_workspace(long M, int N, int K, int K2);
/* X2 (M,K2) at stride ldx2 (or NULL, K2 = 0): a second right-hand operand multiplied in the SAME pass over A, D2 (N,K2) f32 at stride ldd2 (+)= A^T.X2
 * -- dW_ih = dG^T.x and dW_hh = dG^T.h_prev read the gate gradients once.  relu_x != 0: the first product runs on max(X, 0). */
int dic_gemm_tn(int in_dtype, const void* A, long lda, const void* X, long ldx, long M, int N, int K, float* D, long ldd, int kcols,
                const void* X2, long ldx2, int K2, float* D2, long ldd2, int accumulate, int relu_x, void* workspace, size_t workspace_bytes, dic_stream_t stream);
/* Split-plane A operands (round 6).  The x3 recurrence backward (dic_lstm_rec_bwd, DIC_DTYPE_F32X3) writes the gate gradients -- f32 in nn.LSTM's backward,
 * clustering_interp.py:14-41 -- as TWO bf16 planes, hi = bf16(x) at A_hi and lo = bf16(x - hi) a_plane elements behind it (x = hi + lo to 2^-17, the same
 * bytes as f32); these take them as they lie -- no conversion of A in the loop -- against f32 W / X (split on the way in) and produce what dic_gemm_nt /
 * dic_gemm_tn produce for the f32 tensor hi + lo.  lda in bf16 elements; K (nt) / N (tn) and lda multiples of 8. */
int dic_gemm_nt_planes(const void* A_hi, long a_plane, long lda, const float* W, long ldw, const float* bias, long M, int N, int K, float* Y, long ldy,
                       dic_stream_t stream);
int dic_gemm_tn_planes(const void* A_hi, long a_plane, long lda, const float* X, long ldx, long M, int N, int K, float* D, long ldd, int kcols,
                       const float* X2, long ldx2, int K2, float* D2, long ldd2, int accumulate, int relu_x, void* workspace, size_t workspace_bytes, dic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIC_HIP_H */
