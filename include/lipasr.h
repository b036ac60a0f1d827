/*
 * lipasr.h -- C ABI of liblipasr.so: the MI355X (gfx950) hot path of
 * fmazilu/ASR-using-robust-NN behind plain pointers and sizes.
 *
 * The reference has no FFI: its hot path sits behind Python duck-typed protocols
 * (Keras Callback / Constraint, ART estimator / attack, librosa calls).  Every
 * entry point below cites the reference interface (file:line, relative to
 * "/root/reference/Voice digit recogniton/") whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * lipasr_version(): 500 = round 5.  ABI history: 300 (round 3) -> round 4 added lipasr_flag_signal / lipasr_flag_wait and
 * lipasr_debug_chain_head without a bump -> 500: lipasr_flag_wait reports and keeps waiting (see its comment), plus the
 * round-5 entry points marked "(round 5)" below (lipasr_gemm_f16x2, lipasr_mlp_set_fuse_bn / _set_cu_budget / _exchange_errors,
 * lipasr_debug_launch_count).
 *
 * Conventions
 *   - every function returns int: 0 = LIPASR_OK, negative = LIPASR_E*; nothing
 *     throws across the ABI; lipasr_last_error() returns a thread-local message.
 *   - all array pointers are DEVICE pointers owned by the caller unless the
 *     parameter is documented "host".  The library allocates only per-handle /
 *     per-plan workspaces (tables, scratch) at create/plan time, never inside a
 *     launch function, so every launch function may be captured in a HIP graph.
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered; no
 *     launch function synchronises the device.
 *   - a handle / plan is re-entrant across handles, NOT thread-safe per handle:
 *     one handle per GPU per process (one process per GPU under data parallel).
 *     The K3 entry points (projections, norms, lipasr_sv_clip) share the handle's 4 MiB
 *     scratch: issue them on ONE stream per handle.  Classifier and MFCC plans own their
 *     workspaces, so different plans may run on different streams.
 *   - Dense kernels use the Keras layout W(in, out), y = x @ W, row-major.
 */
#ifndef LIPASR_H
#define LIPASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIPASR_OK            0
#define LIPASR_EINVAL       -1   /* bad argument (null pointer, size, unsupported shape) */
#define LIPASR_ENOMEM       -2   /* device allocation failed */
#define LIPASR_EHIP         -3   /* a HIP runtime call failed (message has hipGetErrorString) */
#define LIPASR_EUNSUPPORTED -4   /* valid request outside what the kernels implement */
#define LIPASR_ESTATE       -5   /* call order violated (e.g. mfcc before mfcc_plan) */

#define LIPASR_MAX_LAYERS   16
#define LIPASR_N_MFCC       20   /* librosa.feature.mfcc default n_mfcc */
#define LIPASR_N_MELS       128
#define LIPASR_N_FFT        2048
#define LIPASR_HOP          512
#define LIPASR_SR           22050 /* librosa.load default sr */

typedef struct lipasr_ctx* lipasr_handle_t;
typedef struct lipasr_mlp* lipasr_mlp_t;
typedef struct lipasr_mfcc* lipasr_mfcc_t;
typedef void* lipasr_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------ core */
int lipasr_version(void);
const char* lipasr_last_error(void);
/* device: HIP device ordinal.  Allocates the handle's scratch workspace. */
int lipasr_create(int device, lipasr_handle_t* out);
int lipasr_destroy(lipasr_handle_t h);

/* HIP-event timer on an explicit stream (bench.py times the stream the kernels run on). */
int lipasr_timer_create(lipasr_handle_t h, int* timer_id);
int lipasr_timer_start(lipasr_handle_t h, int timer_id, lipasr_stream_t stream);
int lipasr_timer_stop(lipasr_handle_t h, int timer_id, lipasr_stream_t stream);
/* synchronises on the stop event; host float out */
int lipasr_timer_elapsed_ms(lipasr_handle_t h, int timer_id, float* ms_host);

/* Device-side ordering between two streams: `flag` is an int in device memory (start at 0, raise monotonically).
 * lipasr_flag_signal stores `value` when the stream reaches it; lipasr_flag_wait holds its stream until *flag >= value (a
 * one-wavefront kernel that polls).  After timeout_ms it REPORTS (*err = 1) and keeps waiting: the ordering is never given up
 * because a peer was slow (round 4 let the stream go on there).  After 4 x timeout_ms it gives up (*err = 2) so that a signal
 * that can never come drains the queue instead of hanging it; a wait that finds *err == 2 when its own timeout passes leaves at
 * once.  `err` is an int the kernel writes with system scope: device memory, or pinned host memory the caller reads without
 * synchronising (what lipasr/pipeline.py does at the top of every step).  Cheaper on the waiting stream than hipEventRecord +
 * hipStreamWaitEvent.  The two streams must be able to run concurrently (e.g. disjoint CU masks, or spare wave slots: the
 * waiting wavefront holds one); under a tool that serialises kernels across streams (rocprofv3 --pmc) use events. */
int lipasr_flag_signal(lipasr_handle_t h, int* flag, int value, lipasr_stream_t stream);
int lipasr_flag_wait(lipasr_handle_t h, const int* flag, int value, int timeout_ms, int* err, lipasr_stream_t stream);

/* HIP-graph capture of any sequence of launch functions issued on `stream`. */
int lipasr_graph_begin(lipasr_handle_t h, lipasr_stream_t stream);
int lipasr_graph_end(lipasr_handle_t h, lipasr_stream_t stream, int* graph_id);
int lipasr_graph_launch(lipasr_handle_t h, int graph_id, lipasr_stream_t stream);
int lipasr_graph_destroy(lipasr_handle_t h, int graph_id);

/* A HIP stream restricted to the CUs whose bits are set in cu_mask (HOST array of n_words 32-bit words, bit i of
 * word w = CU 32 w + i; hipExtStreamCreateWithCUMask).  The end-to-end step runs the MFCC kernels and the
 * classifier's many short dependent kernels on two streams; confining the MFCC stream to part of the chip keeps the
 * rest free for the latency-bound chain (DESIGN.md, step level).  Destroy with lipasr_stream_destroy. */
int lipasr_stream_create_masked(lipasr_handle_t h, const uint32_t* cu_mask, int n_words, lipasr_stream_t* out);
int lipasr_stream_destroy(lipasr_handle_t h, lipasr_stream_t stream);

/* ------------------------------------------------------------------ K3: Lipschitz projections
 * Ws: HOST array of n_layers DEVICE pointers to Dense kernels W_l (rows[l] x cols[l], row-major,
 * rows = in, cols = out).  Scalars come back in DEVICE memory (no host sync). */

/* sigma_max(W) by power iteration on W^T W  -- replaces np.linalg.norm(w, ord=2) at
 * Constraints.py:24, extract_features_construct_dataset.py:158, train_constraints.py:58.
 * v_state: device [cols] warm-start vector (in/out); warm=0 starts from a fixed positive vector.
 * iters: number of (W v, W^T u) round trips before the final W v.  clamp_nonneg!=0 evaluates
 * max(W,0) (the matrix norm_constraint projects, Constraints.py:23). */
int lipasr_sigma_max(lipasr_handle_t h, const float* W, int rows, int cols, float* v_state, int warm,
                     int iters, int clamp_nonneg, float* sigma_out, lipasr_stream_t stream);

/* norm_constraint.on_batch_end (Constraints.py:27-33) with get_projection (:22-25) for every
 * layer: W <- max(W,0) * rho^(1/m) / (sigma_max(max(W,0)) + 2.22e-16), m = n_layers (:15-20).
 * v_state: device [sum(cols)] (layer l's vector at offset sum(cols[0..l-1])).
 * sigmas_out: device [n_layers], the pre-scaling sigma_max of each clamped kernel. */
int lipasr_project_per_layer(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols,
                             int n_layers, float rho, float* v_state, int warm, int iters,
                             float* sigmas_out, lipasr_stream_t stream);

/* simple_norm_constraint.on_batch_end (Constraints.py:171-189) with get_projection (:158-169):
 * visits order[0..n_order-1] (HOST array of layer indices, duplicates allowed; the caller builds
 * it from affected_layers_indices exactly as :173-189 iterates), each visited kernel
 * W <- W * (rho / (||W_m^T...W_1^T||_2 + 2.22e-16))^(1/m) with the product norm re-evaluated
 * after every visit.  The product norm is computed ONCE on the device (chain of skinny GEMMs +
 * Gram eigenvalue) and advanced in closed form, which is what the sequential re-evaluation
 * amounts to.  norms_out: device [n_order+1]: product norm before each visit
 * and after the last.  Requires cols[n_layers-1] <= 32 (the class dimension). */
int lipasr_project_product(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols,
                           int n_layers, float rho, const int* order, int n_order,
                           float* norms_out, lipasr_stream_t stream);

/* ||W_m^T ... W_1^T||_2 -- get_lipschitz_constrained's numerator
 * (extract_features_construct_dataset.py:188-194). sigma_out: device [1]. */
int lipasr_product_norm(lipasr_handle_t h, const float* const* Ws, const int* rows, const int* cols,
                        int n_layers, float* sigma_out, lipasr_stream_t stream);

/* customConstraint.__call__ (Constraints.py:43-46): W <- max(W,0) * rho / (||max(W,0)||_F + eps).
 * tf.norm(w, ord=2) with axis=None is the Frobenius norm. */
int lipasr_frobenius_project(lipasr_handle_t h, float* W, size_t n, float rho, lipasr_stream_t stream);

/* max_j sqrt(var_j)/gamma_j -- one BatchNorm's correction factor
 * (extract_features_construct_dataset.py:181-184). out: device [1]. */
int lipasr_bn_correction(lipasr_handle_t h, const float* gamma, const float* var, int n, float* out,
                         lipasr_stream_t stream);

/* norm_constraint_FISTA's singular-value steps (Constraints.py:78-79 `svd(T)` for the constraint
 * read-out, :86-88 `svd(Yt / gam, full_matrices=False); clip(s1, 0, rho); dot(u1 * s1, v1)`):
 * X is device [R][n] row-major with R <= 32 (the class dimension).  out (device [R][n], may alias
 * X, may be NULL) receives U min(S, hi) V^T; svals_out (device [R], may be NULL) the singular
 * values in descending order.  Thin SVD through the fp64 Gram matrix X X^T (Jacobi). */
int lipasr_sv_clip(lipasr_handle_t h, const float* X, int R, int n, float hi, float* out, float* svals_out,
                   lipasr_stream_t stream);

/* ------------------------------------------------------------------ K4: sign step
 * ART FastGradientMethod / ProjectedGradientDescent update (attacks.py:506-510, 657-661),
 * norm=inf, no clip_values: x_adv <- x0 + clip(x_adv + alpha*sign(g) - x0, -eps, +eps), in place.
 * NaN gradients count as 0.  eps = +inf gives the plain FGSM step. */
int lipasr_sign_step(lipasr_handle_t h, float* x_adv, const float* x0, const float* g, size_t n,
                     float alpha, float eps, lipasr_stream_t stream);

/* ------------------------------------------------------------------ A2: StandardScaler
 * sklearn StandardScaler().fit_transform (train_constraints.py:28-31, attacks.py:61-63):
 * per-feature mean and population std over N rows (accumulated in fp64), scale 1 for constant
 * features.  mean_out/scale_out: device double [F]. */
int lipasr_scaler_fit(lipasr_handle_t h, const float* x, int n_rows, int n_feat, double* mean_out,
                      double* scale_out, lipasr_stream_t stream);
int lipasr_scaler_apply(lipasr_handle_t h, const float* x, int n_rows, int n_feat, const double* mean,
                        const double* scale, float* out, lipasr_stream_t stream);

/* ------------------------------------------------------------------ K2: fp32 MFMA GEMM (exact fp32 fma chains)
 * C[M,N] = op(A)[M,K] * op(B)[K,N]; transA=0: A is [M][lda]; transA=1: A is [K][lda] (A^T stored);
 * transB=0: B is [K][ldb]; transB=1: B is [N][ldb]. */
int lipasr_gemm_f32(lipasr_handle_t h, int transA, int transB, int M, int N, int K, const float* A,
                    int lda, const float* B, int ldb, float* C, int ldc, lipasr_stream_t stream);

/* (round 5) The same product in the classifier's third arithmetic mode (lipasr_mlp_set_compute(plan, 2)): every operand value,
 * multiplied by scale_a / scale_b (powers of two that bring it inside fp16's range: |x scale| <= 65504, and best >= 2^-3),
 * is split into two fp16 planes and three of the four cross terms run on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
 * 2^-21 per product where the exact mode's fma chain has 2^-24, at a quarter of the matrix time.  Large well-aligned problems
 * (K a multiple of 32, leading dimensions multiples of 4, 16-byte aligned bases) take the LDS-DMA ring kernel. */
int lipasr_gemm_f16x2(lipasr_handle_t h, int transA, int transB, int M, int N, int K, const float* A, int lda,
                      const float* B, int ldb, float* C, int ldc, float scale_a, float scale_b, lipasr_stream_t stream);

/* ------------------------------------------------------------------ K2/K5: the dense classifier plan
 * get_model() of train_constraints.py:63-88 / train_google_dataset.py:49-74 as data:
 * n_layers Dense layers, widths[0..n_layers]; hidden layers are Dense(relu) [-> BatchNorm]
 * [-> Dropout(rate)], the last is Dense(softmax).  bn/dropout entries for the last layer are
 * ignored.  nonneg[l] != 0 puts Keras NonNeg on kernel l.
 *
 * Flat buffers (caller-owned, fp32):
 *   params / grads / adam_m / adam_v : n_params floats, per layer [W | b | gamma | beta], every
 *       segment start aligned to 4 floats (pad floats are zero and stay zero).  `grads` is the
 *       data-parallel all-reduce buffer.
 *   bnstate : n_state floats, per BN layer [moving_mean | moving_var].
 */
#define LIPASR_SEG_W      0
#define LIPASR_SEG_B      1
#define LIPASR_SEG_GAMMA  2
#define LIPASR_SEG_BETA   3
#define LIPASR_SEG_MMEAN  4  /* in bnstate */
#define LIPASR_SEG_MVAR   5  /* in bnstate */

int lipasr_mlp_create(lipasr_handle_t h, int n_layers, const int* widths, const int* bn,
                      const float* dropout, const int* nonneg, int max_batch, lipasr_mlp_t* out);
int lipasr_mlp_destroy(lipasr_mlp_t m);
int lipasr_mlp_sizes(lipasr_mlp_t m, size_t* n_params, size_t* n_state);
/* offset (in floats) and element count of one segment; count 0 if the layer has no such segment */
int lipasr_mlp_segment(lipasr_mlp_t m, int layer, int kind, size_t* offset, size_t* count);

/* dropout_mode: 0 = off, 1 = Philox mask from (seed, *step_dev, layer, element), 2 = masks given:
 * dropout_masks is a HOST array of n_layers DEVICE pointers (or NULL entries) to [batch][width]
 * multipliers (0 or 1/(1-rate)). */
typedef struct lipasr_dropout_cfg {
  int mode;
  uint64_t seed;
  const int* step_dev;              /* device int, may be NULL (counts as 0) */
  const float* const* masks;        /* host array of device pointers, mode 2 */
} lipasr_dropout_cfg;

/* One training-mode forward + loss + backward: Keras train_step up to the optimizer
 * (train_constraints.py:94-105 with get_model :63-88): Dense/ReLU, BatchNorm with batch statistics
 * (momentum .99, eps 1e-3; moving stats updated in bnstate), inverted dropout, softmax +
 * categorical cross-entropy from logits, gradient (p - y) * inv_batch at the logits.
 * inv_batch = 1/global batch (data parallel: the all-reduce SUMS per-replica grads).
 * Outputs: grads (flat, overwritten), loss_rows [batch] (-sum y log p per row), correct_rows
 * [batch] (1.0 where argmax p == argmax y), probs [batch][classes] (may be NULL). */
int lipasr_mlp_train_fwd_bwd(lipasr_mlp_t m, const float* params, float* bnstate, const float* x,
                             const float* y_onehot, int batch, float inv_batch,
                             const lipasr_dropout_cfg* dropout, float* grads, float* loss_rows,
                             float* correct_rows, float* probs, lipasr_stream_t stream);

/* Data-parallel form of lipasr_mlp_train_fwd_bwd (no reference counterpart: train_constraints.py:91-105 is one
 * process): the same kernels in two calls, so that the gradient all-reduce of everything but the first layer's kernel
 * runs while that kernel's gradient -- 56 % of the bytes and the last thing a backward pass can start -- is still being
 * computed.  _head: forward, loss, the whole dX chain and every gradient except [dW_0 | db_0]; _dw0: those two.
 * lipasr_mlp_grad_split: the first `late_floats` floats of `grads` belong to _dw0, the rest is final after _head. */
int lipasr_mlp_train_fwd_bwd_head(lipasr_mlp_t m, const float* params, float* bnstate, const float* x,
                                  const float* y_onehot, int batch, float inv_batch,
                                  const lipasr_dropout_cfg* dropout, float* grads, float* loss_rows,
                                  float* correct_rows, float* probs, lipasr_stream_t stream);
int lipasr_mlp_train_dw0(lipasr_mlp_t m, const float* x, int batch, float* grads, lipasr_stream_t stream);
int lipasr_mlp_grad_split(lipasr_mlp_t m, size_t* late_floats);

/* Synchronized BatchNorm under data parallelism (opt-in, for runs that must reproduce the single-device statistics of
 * train_constraints.py:68-83 exactly; the default is per-replica statistics): lipasr_mlp_train_fwd_bwd cut into
 * segments that end right after each GEMM whose epilogue leaves BatchNorm column partial sums -- sums of a and a^2 in the
 * forward pass, of g and g xhat in the backward pass -- i.e. before the kernel that consumes them.  The caller runs
 * segment 0 .. n-1 in order and, after segment s, SUM-all-reduces the first lipasr_mlp_train_segment_exchange(m, batch, s)
 * floats of `part` across the ranks (0 floats after the last).  part: caller-owned device buffer of
 * lipasr_mlp_part_floats(m) floats; stat_batch: rows of the GLOBAL batch (the statistics' denominator);
 * stat_grad_scale: 1 / world (dgamma / dbeta are computed from the already-global sums and are summed again by the
 * gradient all-reduce).  The reference model has 5 BatchNorm layers: 11 segments, 10 exchanges of <= 64 kB. */
int lipasr_mlp_train_segments(lipasr_mlp_t m, int* n_segments);
int lipasr_mlp_train_segment_exchange(lipasr_mlp_t m, int batch, int seg, size_t* floats);
int lipasr_mlp_part_floats(lipasr_mlp_t m, size_t* floats);
int lipasr_mlp_train_segment(lipasr_mlp_t m, int seg, const float* params, float* bnstate, const float* x,
                             const float* y_onehot, int batch, float inv_batch,
                             const lipasr_dropout_cfg* dropout, float* grads, float* loss_rows,
                             float* correct_rows, float* probs, float* part, int stat_batch,
                             float stat_grad_scale, lipasr_stream_t stream);

/* K5: Keras Adam (optimizer='adam', train_constraints.py:94) then NonNeg (:67-85) in one launch
 * over the flat buffers: g' = g*grad_scale; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 * w -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps); then w = w*[w>=0] on NonNeg kernels.
 * step_dev: device int holding the number of updates already applied; t = *step_dev + 1 and the
 * counter is incremented in-stream after the update. */
int lipasr_mlp_adam_nonneg(lipasr_mlp_t m, float* params, const float* grads, float* adam_m,
                           float* adam_v, int* step_dev, float lr, float beta1, float beta2, float eps,
                           float grad_scale, lipasr_stream_t stream);

/* Projections over the plan's kernels inside `params` (same semantics as the generic entry points). */
/* One optimizer step as train_constraints.py:94-105 runs it -- Adam, NonNeg, then the
 * simple_norm_constraint callback -- as ONE call: lipasr_mlp_adam_nonneg followed by
 * lipasr_mlp_project_product, with the step counter advanced inside the projection's own
 * single-workgroup kernel instead of a separate launch.  Same results as the two calls. */
int lipasr_mlp_adam_project_product(lipasr_mlp_t m, float* params, const float* grads, float* adam_m,
                                    float* adam_v, int* step_dev, float lr, float beta1, float beta2,
                                    float eps, float grad_scale, float rho, const int* order,
                                    int n_order, float* norms_out, lipasr_stream_t stream);
int lipasr_mlp_project_product(lipasr_mlp_t m, float* params, float rho, const int* order, int n_order,
                               float* norms_out, lipasr_stream_t stream);
int lipasr_mlp_project_per_layer(lipasr_mlp_t m, float* params, float rho, float* v_state, int warm,
                                 int iters, float* sigmas_out, lipasr_stream_t stream);
int lipasr_mlp_product_norm(lipasr_mlp_t m, const float* params, float* sigma_out, lipasr_stream_t stream);

/* Arithmetic of the plan's GEMMs (forward, dX, dW; training and inference).  mode 0 (default): exact
 * fp32 on v_mfma_f32_32x32x2_f32 -- the parity path (logits within 1e-3 of the reference-precision
 * oracle).  mode 1: operands rounded to bf16 (round-to-nearest-even) at the matrix instruction,
 * fp32 accumulation, on v_mfma_f32_32x32x16_bf16 -- BASELINE config 2's "bf16"; parameters,
 * activations, statistics, the loss, Adam and the projections stay fp32.  Takes effect from the
 * next launch; re-capture HIP graphs after changing it. */
int lipasr_mlp_set_compute(lipasr_mlp_t m, int mode);  /* 0 exact fp32, 1 bf16 operands, 2 (round 5) fp16 two-plane split: see lipasr_gemm_f16x2 */

/* Kernel choice of the training pass's forward and dX GEMMs: from `lds_min_tiles` 64x64 output tiles on, the LDS-tiled
 * kernel instead of the 32x32 register-fragment one (0 = the built-in 224, about one tile per CU of a whole MI355X).  A
 * pipeline that confines the classifier to part of the chip lowers it (lipasr/pipeline.py: 128 on 160 CUs).  The two
 * kernels split K differently, so results agree to fp32 rounding, not bit for bit.  Takes effect from the next launch;
 * re-capture HIP graphs after changing it. */
int lipasr_mlp_set_gemm_tiles(lipasr_mlp_t m, int lds_min_tiles);

/* (round 5) Training-mode BatchNorm (train_constraints.py:68 ff.: BatchNormalization() after every hidden Dense) INSIDE the GEMM
 * that produces its input: the row tiles of a column block exchange their column partial sums through memory during the launch
 * (8-byte {tag, value} granules) and every tile normalises the values it still holds, forward and backward -- the ten
 * bn_apply_* launches of a step and their round trips go.  Bitwise reproducible (fixed summation order, no float atomics), equal
 * to the launch chain up to the order of the partial sums (1e-6).  mode 1 (default): wherever the launch's whole grid can be
 * resident on the CUs the plan may use and the batch has at most 64 row tiles; mode 0: the launch chain everywhere (the parity
 * reference).  Not used with synchronized BatchNorm (lipasr_mlp_train_segment).  The residency argument assumes that the plan's
 * stream has its CUs to itself while a launch runs (one process per GPU, the library's contract): two processes that run large
 * fused launches on ONE GPU at the same time can each hold slots the other waits for -- the exchanges then give up after 2 s and
 * report through lipasr_mlp_exchange_errors; such a set-up wants mode 0. */
int lipasr_mlp_set_fuse_bn(lipasr_mlp_t m, int mode);
/* (round 5) How many CUs the stream this plan is launched on may use (a CU-masked stream: lipasr_stream_create_masked);
 * 0 = all of the device (default).  The exchange above spins until its column block's workgroups have all published, so the
 * library must know how many can be resident: a plan run on a masked stream WITHOUT this call may stall (each exchange gives up
 * after 2 s and sets the error count below; it never hangs the queue). */
int lipasr_mlp_set_cu_budget(lipasr_mlp_t m, int n_cus);
/* (round 5) errors_host (HOST int): non-zero if an exchange gave up since the last call (then the results of that step are
 * not valid); synchronises with the device and clears the word. */
int lipasr_mlp_exchange_errors(lipasr_mlp_t m, int* errors_host);

/* model.predict (train_constraints.py:109, attacks.py:344): inference mode (BN moving statistics,
 * no dropout).  probs and/or logits [batch][classes], either may be NULL. */
int lipasr_mlp_predict(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x,
                       int batch, float* probs, float* logits, lipasr_stream_t stream);

/* ART TensorFlowV2Classifier.loss_gradient in inference mode: dx = d mean_b CE(f(x_b), y_b) / dx. */
int lipasr_mlp_input_grad(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x,
                          const float* y_onehot, int batch, float* dx, lipasr_stream_t stream);

/* ART TensorFlowV2Classifier.class_gradient and the vector-Jacobian product the optimisation-based
 * attacks are built on (SaliencyMapMethod, CarliniL2Method, CarliniLInfMethod call sites at
 * attacks.py:538-645): dx[b] = sum_c v[b][c] * d out_c(x_b) / dx in inference mode, where out is the
 * model output -- the softmax probabilities (on_logits = 0; what ART differentiates for a Keras model
 * that ends in softmax) or the logits (on_logits = 1).  v: device [batch][classes] (a one-hot row gives
 * one class gradient).  probs_out (device [batch][classes], may be NULL) receives softmax(f(x)). */
int lipasr_mlp_output_vjp(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x,
                          const float* v, int on_logits, int batch, float* probs_out, float* dx,
                          lipasr_stream_t stream);

/* One fused FGSM/PGD iteration (attacks.py:506-510, 657-661): inference forward at x_adv, CE
 * gradient, backward to the input, and the K4 sign step applied in place on x_adv inside the last
 * backward GEMM's epilogue (dx never reaches memory). */
int lipasr_mlp_attack_step(lipasr_mlp_t m, const float* params, const float* bnstate, float* x_adv,
                           const float* x0, const float* y_onehot, int batch, float alpha, float eps,
                           lipasr_stream_t stream);

/* ART y=None: labels := one-hot argmax of the estimator's own prediction. */
int lipasr_mlp_own_labels(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x,
                          int batch, float* y_onehot_out, lipasr_stream_t stream);

/* ------------------------------------------------------------------ K1: MFCC
 * extract_features / compute_mfcc_all_files (extract_features_construct_dataset.py:24-39,144-150):
 * librosa.load(mono=True) resampling sr_in -> 22050 Hz (resampy kaiser_best), then
 * librosa.feature.mfcc defaults (reflect-padded STFT 2048/512 periodic Hann, power, 128 Slaney
 * mels, 10 log10 with amin 1e-10 and top_db 80 per clip, DCT-II ortho, 20 coefficients), frame
 * axis truncated / zero-padded to utterance_length, flattened coefficient-major
 * (index = coeff*utterance_length + frame).
 *
 * lipasr_mfcc_plan builds the tables for (sr_in, n_samp) and allocates intermediates for
 * batch_max clips; it must precede the launch functions (not capturable itself). */
int lipasr_mfcc_plan(lipasr_handle_t h, int sr_in, int n_samp, int batch_max);
/* Same with an explicit window: librosa.feature.mfcc(y, sr, n_fft=n_fft, win_length=n_fft,
 * hop_length=hop).  (2048, 512) is the plan above (LDS Stockham FFT).  Any 32 <= n_fft <= 510 with
 * 1 <= hop <= n_fft selects the short-window path, where the windowed real DFT is an fp32 MFMA
 * contraction -- the Speaker-recognition features (Speaker recognition/
 * extract_features_construct_dataset.py:224-226: win_length=441, n_fft=441, hop_length=220 on
 * 1-s windows at 22 050 Hz -> 20 x 101 = 2020).  Other values: LIPASR_EUNSUPPORTED. */
int lipasr_mfcc_plan_ex(lipasr_handle_t h, int sr_in, int n_samp, int batch_max, int n_fft, int hop);
/* resampled length int(ceil(n_samp*22050/sr_in)) and frame count 1 + n_y/hop for a plan */
int lipasr_mfcc_dims(lipasr_handle_t h, int* n_y, int* n_frames);

/* wav: [batch][n_samp] float32 mono in [-1,1).  out: [batch][20*utterance_length].
 * affine_mean / affine_scale: optional device double [20*utterance_length]; when given the output
 * is (mfcc - mean)/scale (the precomputed StandardScaler of A2 fused into the last kernel). */
int lipasr_mfcc_f32(lipasr_handle_t h, const float* wav, int batch, int utterance_length,
                    const double* affine_mean, const double* affine_scale, float* out,
                    lipasr_stream_t stream);
/* stage 1 only: y [batch][n_y] (librosa.load output) */
int lipasr_resample_f32(lipasr_handle_t h, const float* wav, int batch, float* y, lipasr_stream_t stream);
/* stages 2..: from an already-resampled 22050 Hz signal y [batch][n_y] (the audio-noise attacks add
 * noise here, attacks.py:108-114, 264-267). */
int lipasr_mfcc_from_22k(lipasr_handle_t h, const float* y, int batch, int n_y, int utterance_length,
                         const double* affine_mean, const double* affine_scale, float* out,
                         lipasr_stream_t stream);

/* 16-bit PCM input (the corpus is 16-bit PCM wav, extract_features_construct_dataset.py:27: librosa.load
 * decodes to float32 by a 2^-15 scale) and clips of DIFFERENT lengths in one launch
 * (compute_mfcc_all_files, :144-150, loops files of any length): pcm is [batch][n_samp] int16 where
 * n_samp is the plan's sample count; n_valid (device int [batch], may be NULL = n_samp everywhere) gives
 * the samples of each row that belong to the clip.  Clip u is processed exactly as a plan of
 * n_valid[u] samples would process it alone (resampled length, frame count, reflect padding and the
 * top_db maximum follow its own length; frames past its end are the zero columns of :33-37).  Same
 * bits as lipasr_mfcc_f32 on pcm * 2^-15.  Needs the 2048/512 path with the 441/320 or 441/160 resampler
 * (16 kHz / 8 kHz input); otherwise LIPASR_EUNSUPPORTED.  Rows whose length is a multiple of 4 samples go
 * through the same three kernels as float32 batches (the resampler reads int16 and cuts each row at its
 * clip's end); other row lengths through the fused resample -> STFT kernel. */
int lipasr_mfcc_i16(lipasr_handle_t h, const int16_t* pcm, const int* n_valid, int batch,
                    int utterance_length, const double* affine_mean, const double* affine_scale,
                    float* out, lipasr_stream_t stream);

/* MFCC plans as objects: each owns its tables and intermediates, so several extractors (a training
 * pipeline's and a validation pass's, or two streams) coexist on one handle without re-planning.
 * The handle-level entry points above operate on the handle's default plan.
 * sample_format: 0 = float32 in [-1, 1), 1 = int16 PCM.  n_valid as in lipasr_mfcc_i16. */
int lipasr_mfcc_create(lipasr_handle_t h, int sr_in, int n_samp_max, int batch_max, int n_fft, int hop,
                       lipasr_mfcc_t* out);
int lipasr_mfcc_destroy(lipasr_mfcc_t p);
/* fused (may be NULL): 1 when the plan runs resampling and STFT as ONE kernel (the resampled signal stays in LDS) */
int lipasr_mfcc_plan_dims(lipasr_mfcc_t p, int* n_y, int* n_frames, int* fused);
int lipasr_mfcc_extract(lipasr_mfcc_t p, const void* wav, int sample_format, const int* n_valid, int batch,
                        int utterance_length, const double* affine_mean, const double* affine_scale,
                        float* out, lipasr_stream_t stream);
int lipasr_mfcc_plan_resample(lipasr_mfcc_t p, const float* wav, int batch, float* y, lipasr_stream_t stream);
int lipasr_mfcc_plan_from_22k(lipasr_mfcc_t p, const float* y, int batch, int n_y, int utterance_length,
                              const double* affine_mean, const double* affine_scale, float* out,
                              lipasr_stream_t stream);

/* Per-kernel HIP-event timing of the next `max_calls` extractions -- lipasr_mfcc_f32 calls, or
 * lipasr_resample_f32 + lipasr_mfcc_from_22k pairs -- recorded on the stream the kernels run on.
 * _end synchronises and returns the average milliseconds of {resample, stft_mel, dct}
 * (host float[3]) and the number of extractions measured (host int).  On the fused path there is no
 * resampling kernel: slot 0 is 0 and slot 1 is the fused resample + STFT + mel kernel. */
int lipasr_mfcc_profile_begin(lipasr_handle_t h, int max_calls);
int lipasr_mfcc_profile_end(lipasr_handle_t h, float* avg_ms3, int* n_calls);
int lipasr_mfcc_plan_profile_begin(lipasr_mfcc_t p, int max_calls);
int lipasr_mfcc_plan_profile_end(lipasr_mfcc_t p, float* avg_ms3, int* n_calls);
/* knobs of one plan: keys 0 and 1 as lipasr_debug_set; key 2: value != 0 makes the plan run the fused resample -> STFT
 * kernel for every batch (1.7x the algorithmic HBM bytes instead of 4.6x, but about 1.7x the time of the three-kernel path
 * on a whole MI355X: DESIGN.md section 3); by default the fused kernel runs only where the three-kernel path cannot read
 * the input (int16 or per-clip lengths in rows that are not a multiple of 4 samples long).
 * Stage-mask bits of key 0 (2048/512 plans): 256 = the Stockham FFT kernel (stft_mel2_kernel) instead of the block-DFT
 * kernel on the matrix pipe (stft_bdft_kernel, the default since round 4): its parity reference; 64 = the round-2 kernel.
 * key 3: frames per workgroup of the block-DFT kernel (a multiple of 4 in [4, 4096]; default 44 = one workgroup per 1-s clip).
 * key 4: value != 0 lets that kernel apply the top_db floor and the DCT itself when one workgroup covers a whole clip
 * (<= 64 frames, utterance_length <= 64): no dct_kernel launch, bit-identical features; off by default (slower on batches
 * that are not cache-warm: DESIGN.md section 3).
 * Replaces the arithmetic of librosa.feature.mfcc's STFT, extract_features_construct_dataset.py:30. */
int lipasr_mfcc_plan_set(lipasr_mfcc_t p, int key, int value);

/* A12 audio-domain noise on device, Philox RNG (attacks.py:73-86, 145-183, 222-245), in place on
 * y [batch][n]:  mode 0: y + N(0, p0)               (add_white_noise, sigma = p0)
 *                mode 1: impulse mixture, p = p0, alpha = p1 (add_noise / mixtgauss)
 *                mode 2: white noise at target SNR p0 dB per clip (add_white_noise_with_snr) */
int lipasr_add_noise_f32(lipasr_handle_t h, float* y, int batch, int n, int mode, float p0, float p1,
                         uint64_t seed, lipasr_stream_t stream);

/* Knobs of the handle's default MFCC plan.  key 0: stage mask for profiling (bit0 skip the FFT passes, bit1 skip the mel
 * reduction -- both give wrong results and exist to time the remaining stages; bit2 selects the VALU resampler instead of
 * the MFMA one; bit7 (128) selects the three-kernel path -- resample, STFT, DCT with the resampled signal in HBM -- instead
 * of the fused resample -> STFT kernel: the parity reference of the fused kernel).
 * key 1: number of workgroups the persistent resampler aims for = the CUs its stream may use (default 256; a
 * pipeline that runs the MFCC on a CU-masked stream sets it to the size of the mask).  Kept in the handle. */
int lipasr_debug_set(lipasr_handle_t h, int key, int value);

/* Profiling knob: GEMM kernel choice. bits 0-1: 0 = automatic, 1 = split-K register kernel only, 2 = LDS-tiled kernel
 * wherever it is legal.  bit 2 (4): lipasr_mlp_train_fwd_bwd launches the first layer's weight gradient on its own
 * (as the data-parallel head / dw0 pair does) instead of inside the grouped launch.  bit 3 (8): the grouped weight-gradient
 * launch on 32x32 register-fragment tiles (round 2) instead of 64x64 LDS tiles.  Round 5 (arithmetic mode 2 only): bit 4 (16) the
 * XCD-aware tile order, bit 5 (32) no LDS-DMA ring kernels, bit 6 (64) the weight gradients on 64x64 ring tiles, bit 7 (128) on
 * 128x128 tiles that split per fragment (no split pass), bit 8 (256) no 128x64 exchange tiles, bit 9 (512) no loader-wavefront instance of the 64x64 exchange tile.  Same results in
 * every setting (to the rounding of a different summation order where the tile changes). */
int lipasr_debug_gemm_mode(int mode);

/* Test hook: how many launches since the library was loaded took kernel family `kind` -- 0: forward / input-gradient GEMMs on
 * 128x64 exchange tiles, 1: grouped weight-gradient launches with 128x128 split-pass tiles; -1 for an unknown kind.  (The choice
 * depends on the plan's CU budget; tests that mean to cover those kernels check that they really ran.) */
long lipasr_debug_launch_count(int kind);

/* Profiling knob: how many of the leading (small) steps of the product chain W_m^T ... W_1^T run as ONE launch
 * (chain_head_kernel, fp32 matrix instructions): -1 = automatic (the first two steps, when their panels have <= 256 columns
 * in multiples of 16 and there are at most 16 classes; never the last step), 0 = none (one launch per step, rounds 1-3),
 * n = 2 or 3 = at most n.
 * The fused steps associate their sums differently: the product agrees with the per-step launches to ~1e-7. */
int lipasr_debug_chain_head(int n);

/* Host-only (no GPU needed): copies one constant table, exactly as the kernels read it, into `out`
 * and returns its element count (negative = error); out may be NULL to query the size.
 * which: 0 Hann[2048]; 1 DCT[20*128]; 2 dense mel filter bank[128*1025]; 3 polyphase resampling taps
 * [up*taps] for sr_in -> 22050; 4 {up, down, taps, left}; 5 per-phase input offsets [up]. */
int lipasr_debug_table(int which, int sr_in, float* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* LIPASR_H */
