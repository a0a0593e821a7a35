/*
 * hbo.h -- C ABI of libhbo: MI355X-native (gfx950) implementation of HyperBO's GP hot path.
 *
 * The reference (google-research/hyperbo) has no FFI: the path sits behind plain Python
 * callables traced by JAX.  Each entry point below names the reference callable it replaces
 * (file:line relative to the reference repo root); hyperbo_amd/ binds them with ctypes and
 * re-exposes the reference's Python signatures.
 *
 * Conventions
 *   - all host arrays are C-contiguous row-major, element type = hbo_model.dtype
 *     (HBO_F32 -> float, HBO_F64 -> double); gradients and NLL values are always double.
 *   - every function returns an int status: 0 ok, <0 usage/runtime error, >0 numerical
 *     (HBO_NOT_PD: the jittered Gram matrix was not positive definite; outputs are NaN-filled,
 *     mirroring jax.scipy.linalg.cholesky which yields NaNs instead of raising).
 *   - never aborts, never throws across the ABI; hbo_last_error(ctx) returns a message.
 *   - a ctx is not re-entrant; use one ctx per GPU / per thread.  Calls are synchronous.
 *   - hyper-parameters in hbo_model are ALREADY WARPED (softplus etc. stay in Python so that
 *     arbitrary warp_func dicts keep working: hyperbo/basics/params_utils.py:97-111).
 */
#ifndef HBO_H_
#define HBO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBO_OK 0
#define HBO_ERR_ARG (-1)
#define HBO_ERR_HIP (-2)
#define HBO_ERR_NODEV (-3)
#define HBO_ERR_UNSUPPORTED (-4)
#define HBO_ERR_COMM (-5)
#define HBO_NOT_PD 1

/* closed registries: hyperbo/bo_utils/const.py:22-50 */
enum hbo_kernel_id { HBO_KERNEL_SE = 0, HBO_KERNEL_MATERN32 = 1, HBO_KERNEL_MATERN52 = 2, HBO_KERNEL_DOT = 3 };
enum hbo_mean_id { HBO_MEAN_ZERO = 0, HBO_MEAN_CONSTANT = 1, HBO_MEAN_LINEAR = 2, HBO_MEAN_LINEAR_MLP = 3 };
enum hbo_dtype { HBO_F32 = 0, HBO_F64 = 1 };
enum hbo_acq_id { HBO_ACQ_EI = 0, HBO_ACQ_PI = 1, HBO_ACQ_UCB = 2 };

#define HBO_MAX_MLP_LAYERS 8
#define HBO_MAX_FEATURE_DIM 256

typedef struct hbo_ctx hbo_ctx;
typedef struct hbo_dataset hbo_dataset; /* device-resident sub-datasets (x, y) */
typedef struct hbo_cache hbo_cache;     /* device-resident GPCache: x, chol, chol^-1, kinvy */

/* Model descriptor: kernel (hyperbo/gp_utils/kernel.py:63-183), mean (gp_utils/mean.py:54-79),
 * MLP basis (gp_utils/basis_functions.py:24-36, flax Dense: y = x @ kernel + bias, tanh on every
 * layer) and the jitter eps of hyperbo/basics/linalg.py:42,78 (default 1e-6). */
typedef struct hbo_model {
  int32_t kernel_id;        /* hbo_kernel_id */
  int32_t mean_id;          /* hbo_mean_id */
  int32_t dtype;            /* hbo_dtype */
  int32_t input_dim;        /* D */
  int32_t kernel_uses_mlp;  /* 1: base kernel evaluated on MLP features (kernel.py:148-183) */
  int32_t n_layers;         /* MLP depth (0 if unused) */
  int32_t features[HBO_MAX_MLP_LAYERS]; /* params.config['mlp_features'] */
  int32_t n_lengthscale;    /* 1 (broadcast) or the kernel's feature dimension */
  int32_t reserved0;
  double eps;
  double signal_variance;
  double noise_variance;
  double constant;          /* mean.constant */
  double dot_prod_sigma;
  double dot_prod_bias;
  double linear_bias;       /* linear_mean['bias'] */
  const void* lengthscale;                       /* [n_lengthscale] */
  const void* mlp_kernel[HBO_MAX_MLP_LAYERS];    /* [in, out] row-major */
  const void* mlp_bias[HBO_MAX_MLP_LAYERS];      /* [out] */
  const void* linear_kernel;                     /* [Fin] (linear_mean['kernel'][:,0]) */
} hbo_model;

/* Flat layout (in doubles) of the gradient w.r.t. the WARPED parameters written by hbo_nll / hbo_objective.
 * Offsets of absent parameters are -1.  total = number of doubles. */
typedef struct hbo_grad_layout {
  int32_t lengthscale;      /* n_lengthscale entries */
  int32_t signal_variance;
  int32_t noise_variance;
  int32_t constant;
  int32_t dot_prod_sigma;
  int32_t dot_prod_bias;
  int32_t linear_kernel;    /* Fin entries */
  int32_t linear_bias;
  int32_t mlp_kernel[HBO_MAX_MLP_LAYERS]; /* in*out entries each, row-major [in,out] */
  int32_t mlp_bias[HBO_MAX_MLP_LAYERS];
  int32_t total;
} hbo_grad_layout;

typedef struct hbo_task {
  const void* x; /* [n, input_dim] */
  const void* y; /* [n, m] */
  int64_t n;
  int32_t m;
  int32_t reserved0;
} hbo_task;

/* ---- context ------------------------------------------------------------------------- */
int hbo_ctx_create(int device, hbo_ctx** out);
int hbo_ctx_destroy(hbo_ctx* ctx);
const char* hbo_last_error(hbo_ctx* ctx); /* ctx may be NULL: last error of ctx-less calls */
const char* hbo_version(void);
int hbo_device_count(void);
/* name_out (nullable, cap bytes): marketing / arch name of the device; cus / mem_bytes (nullable): compute units, device memory */
int hbo_device_info(int device, char* name_out, int32_t cap, int32_t* cus, int64_t* mem_bytes);

int hbo_grad_layout_of(const hbo_model* model, hbo_grad_layout* out);

/* ---- kernel.py:33-58 cov_func(params, vx1, vx2=None, diag=False) ------------------------ */
/* out: [n1,n2] (x2 may be NULL -> x2 = x1), or [n1] when diag != 0 (requires x2 == NULL). */
int hbo_gram(hbo_ctx* ctx, const hbo_model* model, const void* x1, int64_t n1, const void* x2,
             int64_t n2, int diag, void* out);
/* ---- mean.py:34-49 mean_func(params, vx) -> [n,1] -------------------------------------- */
int hbo_mean(hbo_ctx* ctx, const hbo_model* model, const void* x, int64_t n, void* out);

/* ---- objectives.py:109-210 neg_log_marginal_likelihood + its jax.value_and_grad
 *      (gp.py:134, lbfgs.py:238) over device-resident sub-datasets ------------------------ */
int hbo_dataset_create(hbo_ctx* ctx, int dtype, int input_dim, const hbo_task* tasks, int n_tasks,
                       hbo_dataset** out);
int hbo_dataset_free(hbo_ctx* ctx, hbo_dataset* ds);
/* A random sub-sample of a resident dataset, drawn by the caller, gathered on the device: what every Adam step of
 * infer_parameters does with sub_sample_dataset_iterator (hyperbo/gp_utils/gp.py:101-111, hyperbo/basics/data_utils.py:72-100;
 * the reference indexes device arrays with jax.random.permutation).  Task k of `src` IN THE ORDER src HOLDS THEM (largest n first,
 * stable in the order of hbo_dataset_create's `tasks`) keeps counts[k] rows: rows idx[off_k + r], r < counts[k], off_k = sum of the
 * non-negative counts before k -- or all of its rows, in order, when counts[k] < 0 (no indices consumed).  Only the indices travel
 * (4 bytes per kept row); the new dataset is independent of `src` afterwards. */
int hbo_dataset_subsample(hbo_ctx* ctx, const hbo_dataset* src, const int64_t* counts, const int32_t* idx, hbo_dataset** out);
/* nll_sum: sum over the tasks of this dataset of the per-task Cholesky NLL (objectives.py:144-156,
 * incl. the (m,m)+scalar broadcast quirk for m>1); the caller divides by the number of tasks
 * (objectives.py:192-195) -- a sum so that task shards on different GPUs can be all-reduced.
 * nll_per_task (nullable): [n_tasks].  grad_sum (nullable): [layout.total] sum over tasks of
 * d nll_task / d warped-parameter.  Returns HBO_NOT_PD if any task failed (its values are NaN). */
int hbo_nll(hbo_ctx* ctx, const hbo_model* model, hbo_dataset* ds, double* nll_sum,
            double* nll_per_task, double* grad_sum);

/* ---- objectives.py:29-106 multivariate_normal_divergence (+ jax.value_and_grad) over the ALIGNED
 *      sub-datasets of `ds`: distance between N(mean_a y, cov_a y) of the m aligned columns and the GP prior
 *      N(mean_func(x), cov_func(x,x) + noise I) -- no jitter, model->eps is ignored.
 *        HBO_OBJ_NLL  = hbo_nll
 *        HBO_OBJ_EKL  utils.py:84-148 kl_multivariate_normal(partial=True, eps=0, weight=1)  ('ekl' / 'kl')
 *        HBO_OBJ_EUC  utils.py:151-173 euclidean_multivariate_normal(mean_weight=cov_weight=1) ('euc')
 *      Same output convention as hbo_nll (sums over tasks; the caller divides by the task count,
 *      objectives.py:98-101).  Any number m of aligned columns: up to 127 ride through the factorisation as augmented rows, beyond
 *      that the data rows go through the explicit inverse (objective.hip: extra_rows). */
enum hbo_objective_id { HBO_OBJ_NLL = 0, HBO_OBJ_EKL = 1, HBO_OBJ_EUC = 2 };
int hbo_objective(hbo_ctx* ctx, const hbo_model* model, hbo_dataset* ds, int objective, double* value_sum,
                  double* value_per_task, double* grad_sum);

/* ---- linalg.py:72-110 solve_gp_linear_system -> GPCache(chol, kinvy) (gp.py:540-560) ----- */
int hbo_factor(hbo_ctx* ctx, const hbo_model* model, const void* x, int64_t n, const void* y,
               int32_t m, hbo_cache** out);
/* chol_out [n,n] (lower, zeros above the diagonal), kinvy_out [n,m], y_minus_mu_out [n,m];
 * any may be NULL. */
int hbo_cache_export(hbo_ctx* ctx, hbo_cache* cache, void* chol_out, void* kinvy_out,
                     void* y_minus_mu_out);
int hbo_cache_free(hbo_ctx* ctx, hbo_cache* cache);
/* O(N^2) in-place append of n_new observations (x_new [n_new, D], y_new [n_new, m]) to a cache built with the
 * SAME hyper-parameters -- what GP.update_sub_dataset(is_append=True) + setup_predictor recompute from
 * scratch in the reference (gp.py:426-452,540-560; "One can potentially support rank-1 updates", gp.py:284).
 * HBO_ERR_UNSUPPORTED: padded capacity exhausted, re-factorise with hbo_factor. */
int hbo_cache_append(hbo_ctx* ctx, const hbo_model* model, hbo_cache* cache, const void* x_new, int64_t n_new,
                     const void* y_new);

/* ---- gp.py:242-305 predict ------------------------------------------------------------- */
/* cache == NULL -> prior branch (gp.py:275-282).  mu_out [M,1]; var_out [M,1] or [M,M] if
 * full_cov.  No noise / unbiased scaling here (that is GP.predict, gp.py:607-619). */
int hbo_predict(hbo_ctx* ctx, const hbo_model* model, hbo_cache* cache, const void* xq, int64_t M,
                int full_cov, void* mu_out, void* var_out);
/* ---- acfun.py:51-142 acquisition on top of GP.predict(full_cov=False) ------------------- */
/* var' = (var + add_noise) * scale (gp.py:607-619); EI/PI: param = target; UCB: param = beta. */
int hbo_acq(hbo_ctx* ctx, const hbo_model* model, hbo_cache* cache, const void* xq, int64_t M,
            int acq_id, double param, double add_noise, double scale, void* out);

/* ---- S hyper-parameter samples of ONE model family as one batch: what acquisition functions do on an HGP
 *      (hyperbo/bo_utils/acfun.py:72-82 loops model.predict over model.params.samples, gp.py:666-682; the jax counterpart is a
 *      vmap over the draws).  models[s]: sample s (same dtype, covariance, mean and MLP architecture for all); x [n,D], y [n,m]
 *      the observations every sample conditions on; the S Gram matrices are built, factorised and inverted as ONE batch, the S
 *      posteriors + acquisition epilogues queue up on the device; out [S,M] (model dtype), row s = hbo_acq of sample s with
 *      params[s] / add_noise[s] (the caller averages: acfun.py:82).  Rows of samples whose Gram matrix is not PD are NaN
 *      (HBO_NOT_PD). */
int hbo_acq_samples(hbo_ctx* ctx, const hbo_model* models, int32_t S, const void* x, int64_t n, const void* y, int32_t m,
                    const void* xq, int64_t M, int acq_id, const double* params, const double* add_noise, double scale, void* out);

/* ---- d acquisition / d x_query: the gradient jaxopt.ScipyBoundedMinimize(L-BFGS-B) takes of
 *      f(x) = -ac_func(model, key, x[None]) in bayesopt() (hyperbo/bo_utils/bayesopt.py:116-125).
 *      Queries are independent rows: out [M,1] (model dtype) as hbo_acq, grad_out [M, input_dim] doubles.
 *      Matern kernels: a query at zero distance from a training point contributes 0 (linalg.py:183-188). */
int hbo_acq_grad(hbo_ctx* ctx, const hbo_model* model, hbo_cache* cache, const void* xq, int64_t M,
                 int acq_id, double param, double add_noise, double scale, void* out, double* grad_out);

/* ---- dense building blocks (linalg.py:29-33 solve_linear_system); host in/out ----------- */
/* a: [n,n] SPD (only the lower triangle is read).  chol_out: lower factor (zeros above diag);
 * inv_out (nullable): full symmetric a^-1;  b/x_out (nullable): [n,m] solve a x = b. */
int hbo_spd_solve(hbo_ctx* ctx, int dtype, const void* a, int64_t n, const void* b, int32_t m,
                  void* chol_out, void* inv_out, void* x_out, double* logdet_half);
/* linalg.py:139-145, inverse_spdmatrix_vector_product(cached_cholesky=...): x = L^-T L^-1 b for a lower factor the caller holds
 * as an array (chol_lower: [n,n] row-major, only the lower triangle is read; b / x_out: [n,m]).  Two substitution sweeps on the
 * device; nothing is factorised. */
int hbo_chol_solve(hbo_ctx* ctx, int dtype, const void* chol_lower, int64_t n, const void* b, int32_t m, void* x_out);

/* ---- profiling: per-stage device time of the last hbo_nll / hbo_factor / hbo_acq call,
 *      measured with HIP events on the stream the kernels were launched on ---------------- */
#define HBO_MAX_PROFILE_STAGES 32
int hbo_profile_enable(hbo_ctx* ctx, int level); /* 0 off, 1 per stage, 2 per launch; -1: only the launches of the
                                                    bulk trailing update ("syrk_bulk", the roofline kernel of bench.py) */
/* names: array of HBO_MAX_PROFILE_STAGES char[32]; ms: total ms; launches: count */
int hbo_profile_get(hbo_ctx* ctx, char names[][32], double* ms, int32_t* launches, int32_t* n);

/* Options (integers by name).  Unknown names are an error.  The placement / overlap knobs of the launch schedules that the
 * measurement tools vary live behind hbo_tune (include/hbo_tune.h); they are not part of this boundary.
 *   potrf_group     0..16 128-wide panels per trailing update, K = 128*group (0 = auto: 3 up to 96 blocks, then 4, 8 from 256
 *                         blocks on and for fp32 factorisations on the bf16 matrix cores)
 *   lookahead       0..2  panel chain on its own stream one group ahead of the bulk update, inverse overlapped.  1 (default): where
 *                         it pays -- more than 4 blocks and (>= 18 blocks or tasks x blocks >= 80); below that everything runs
 *                         on one stream in order (5-13 % faster there).  0 = never, 2 = whenever there is more than one block
 *   small_nblk      int   matrices up to this many 128-blocks use 64x64 tiles in the inverse and in K^-1 = W^T W (default -1: auto --
 *                         48 blocks in fp64, 32 in fp32)
 *   pool_cap_mb     >=0   device buffers of freed datasets / caches are parked for the next one of the same shape (GP.train()
 *                         re-creates its sub-sampled batch every step); at most this many MB stay parked (default: a quarter
 *                         of the device memory, at most 49152; 0 = off)
 *   post_chunk      128..65536 posterior / acquisition: query candidates per pass (cross-Gram workspace = npad x post_chunk
 *                         elements whatever M; two workspaces alternate so that the Gram build of a chunk runs beside the
 *                         triangular product of the previous one)
 *   bf16x3          0/1   fp32 only: the GEMM-shaped work -- trailing updates of the factorisation, the products of the inverse
 *                         and K^-1 = W^T W (above small_nblk blocks), the posterior product V = L^-1 Kxq -- runs on the bf16 matrix cores
 *                         from exact three-way splits of both operands (six bf16 MFMAs per fp32 product, fp32 accumulate:
 *                         fp32-class accuracy at 1.3-1.5x the fp32-MFMA rate).  Default 1; 0 = fp32 MFMA.  The posterior product of
 *                         the stationary covariances goes one step further (hbo_tune post_f16x2, default on): two-way fp16 splits of
 *                         operands scaled by powers of two, three fp16 MFMAs per product (2^-22 per product: as close to fp64 as
 *                         the fp32-MFMA product) -- cfg 3's EI 97 -> 59 ms; and so do the factorisation's trailing updates, the inverse's
 *                         upper levels and K^-1 = W^T W of those covariances (hbo_tune chol_f16x2, default on: cfg 3's factor 21.5 -> 17.6 ms) */
int hbo_set_option(hbo_ctx* ctx, const char* name, int64_t value);

/* ---- multi-GPU: one process per GPU; sum-all-reduce of [nll, grads] over RCCL (xGMI) ------ */
#define HBO_UNIQUE_ID_BYTES 128
int hbo_comm_unique_id(void* out128);
int hbo_comm_init(hbo_ctx* ctx, int rank, int nranks, const void* unique_id128);
int hbo_comm_allreduce_sum(hbo_ctx* ctx, double* buf, int32_t count);   /* host buffer: up, all-reduce, down (tests, small ad-hoc sums) */
int hbo_comm_destroy(hbo_ctx* ctx);
/* The task-sharded objective (objectives.py:181-195 is an independent sum over sub-datasets; `ds` = this rank's shard, may be
 * NULL / empty: a rank beyond the task count contributes zeros).  Like hbo_objective, but value_sum / count / grad_sum are the sums
 * over ALL ranks of the communicator bound with hbo_comm_init: every rank reduces its own tasks' [value, count, gradient] on the
 * device (same summation order as hbo_objective's host loop), ONE ncclAllReduce runs in place on the context's stream (RCCL over
 * xGMI, no host hop) and the result comes back in one copy.  timing (nullable, 2 doubles): [0] ms of device time this rank spent
 * on its own shard before the collective, [1] us from there to the end of the all-reduce (HIP events on the context's stream).
 * Without a communicator the sums are the local ones.  Returns HBO_NOT_PD when the reduced value is NaN. */
int hbo_objective_sharded(hbo_ctx* ctx, const hbo_model* model, hbo_dataset* ds, int objective, double* value_sum,
                          double* count, double* grad_sum, double* timing);

#ifdef __cplusplus
}
#endif
#endif /* HBO_H_ */
