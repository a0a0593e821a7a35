// Host-side internals shared by the translation units of the C ABI (api.hip: context, options, datasets, dense entry points;
// objective.hip: hbo_objective / hbo_objective_sharded; cache.hip: hbo_factor, row append, posterior, acquisition): model
// validation and upload, the MLP feature pipeline, the per-task device buffers and the descriptors the kernels read.
#pragma once
#include "ctx.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "runtime.h"
#include "sched.h"

// ---- model ---------------------------------------------------------------------------------
static int feature_dim(const hbo_model* m) {
  return m->kernel_uses_mlp ? m->features[m->n_layers - 1] : m->input_dim;
}
static int mean_feature_dim(const hbo_model* m) {
  if (m->mean_id == HBO_MEAN_LINEAR) return m->input_dim;
  if (m->mean_id == HBO_MEAN_LINEAR_MLP) return m->features[m->n_layers - 1];
  return 0;
}
static bool needs_mlp(const hbo_model* m) { return m->kernel_uses_mlp || m->mean_id == HBO_MEAN_LINEAR_MLP; }

static int validate_model(hbo_ctx* c, const hbo_model* m) {
  if (!m) return fail(c, HBO_ERR_ARG, "model is null");
  if (m->dtype != HBO_F32 && m->dtype != HBO_F64) return fail(c, HBO_ERR_ARG, "bad dtype");
  if (m->kernel_id < 0 || m->kernel_id > HBO_KERNEL_DOT) return fail(c, HBO_ERR_ARG, "bad kernel_id");
  if (m->mean_id < 0 || m->mean_id > HBO_MEAN_LINEAR_MLP) return fail(c, HBO_ERR_ARG, "bad mean_id");
  if (m->input_dim <= 0 || m->input_dim > HBO_MAX_FEATURE_DIM) return fail(c, HBO_ERR_ARG, "bad input_dim");
  if (needs_mlp(m)) {
    if (m->n_layers <= 0 || m->n_layers > HBO_MAX_MLP_LAYERS) return fail(c, HBO_ERR_ARG, "bad n_layers");
    for (int l = 0; l < m->n_layers; ++l) {
      if (m->features[l] <= 0 || m->features[l] > HBO_MAX_FEATURE_DIM) return fail(c, HBO_ERR_ARG, "bad mlp feature size");
      if (!m->mlp_kernel[l] || !m->mlp_bias[l]) return fail(c, HBO_ERR_ARG, "mlp parameters missing");
    }
  }
  const int fd = feature_dim(m);
  if (m->kernel_id != HBO_KERNEL_DOT) {
    if (!m->lengthscale) return fail(c, HBO_ERR_ARG, "lengthscale missing");
    if (m->n_lengthscale != 1 && m->n_lengthscale != fd)
      return fail(c, HBO_ERR_ARG, "lengthscale must have 1 or feature-dim entries");
  }
  if (mean_feature_dim(m) > 0 && !m->linear_kernel) return fail(c, HBO_ERR_ARG, "linear_mean kernel missing");
  return HBO_OK;
}

static double host_elem(const void* p, int dtype, int64_t i) {
  return dtype == HBO_F64 ? ((const double*)p)[i] : (double)((const float*)p)[i];
}

// max_i A_ii of Gram + (noise + jitter) I for a stationary covariance (k(x, x) = signal variance): what the fp32 factorisation's
// f16x2 products scale the factor's entries by (ctx.h: chol_diag_bound); 0 = unknown (dot-product kernel, parameters not finite)
static inline double chol_diag_bound_of(const hbo_model* m) {
  if (m->kernel_id == HBO_KERNEL_DOT) return 0.0;
  const double b = (double)m->signal_variance + (double)m->noise_variance + (double)m->eps;
  return (b > 0 && b < 1e30) ? b : 0.0;
}
struct CholBoundScope {   // valid from run_potrf to the last product of the inverse / K^-1 of the same matrices
  hbo_ctx* c;
  CholBoundScope(hbo_ctx* ctx, double b) : c(ctx) { c->chol_diag_bound = b; }
  ~CholBoundScope() { c->chol_diag_bound = 0; }
};
// the device-side form of an (already warped) model
static void fill_model_dev(ModelDev& h, const hbo_model* m) {
  memset(&h, 0, sizeof h);
  h.kernel_id = m->kernel_id; h.mean_id = m->mean_id; h.fdim = feature_dim(m);
  h.n_ls = (m->kernel_id == HBO_KERNEL_DOT) ? 0 : m->n_lengthscale;
  h.sv = m->signal_variance; h.noise = m->noise_variance; h.eps = m->eps; h.constant = m->constant;
  h.dot_sigma = m->dot_prod_sigma; h.dot_bias = m->dot_prod_bias; h.linear_bias = m->linear_bias;
  if (m->kernel_id == HBO_KERNEL_DOT) { if (h.dot_sigma == 0) h.dot_sigma = 1; }
  else { h.dot_sigma = 1; }
  for (int d = 0; d < h.fdim; ++d) {
    double ls = 1.0;
    if (m->kernel_id != HBO_KERNEL_DOT) ls = host_elem(m->lengthscale, m->dtype, m->n_lengthscale == 1 ? 0 : d);
    h.inv_ls[d] = 1.0 / ls;
  }
  const int fm = mean_feature_dim(m);
  for (int d = 0; d < fm; ++d) h.lin_w[d] = host_elem(m->linear_kernel, m->dtype, d);
}
// fills ctx->h_model, uploads it and the MLP weights
static int upload_model(hbo_ctx* c, const hbo_model* m) {
  int rc = validate_model(c, m);
  if (rc) return rc;
  // the pinned copy may still be read by the previous upload (calls that return without waiting for the stream)
  HIPCHK(c, hipEventSynchronize(c->ev_upload));
  ModelDev& h = *c->h_model;
  fill_model_dev(h, m);
  HIPCHK(c, hipMemcpyAsync(c->d_model, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_upload, c->stream));
  if (needs_mlp(m)) {
    int fin = m->input_dim;
    for (int l = 0; l < m->n_layers; ++l) {
      const size_t wb = (size_t)fin * m->features[l] * esize(m->dtype), bb = (size_t)m->features[l] * esize(m->dtype);
      if (c->mlp_w_bytes[l] < wb) { if (c->d_mlp_w[l]) hipFree(c->d_mlp_w[l]); HIPCHK(c, hbo_malloc(c, &c->d_mlp_w[l], wb)); c->mlp_w_bytes[l] = wb; }
      if (c->mlp_b_bytes[l] < bb) { if (c->d_mlp_b[l]) hipFree(c->d_mlp_b[l]); HIPCHK(c, hbo_malloc(c, &c->d_mlp_b[l], bb)); c->mlp_b_bytes[l] = bb; }
      HIPCHK(c, hipMemcpyAsync(c->d_mlp_w[l], m->mlp_kernel[l], wb, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->d_mlp_b[l], m->mlp_bias[l], bb, hipMemcpyHostToDevice, c->stream));
      fin = m->features[l];
    }
  }
  // the MLP weights come from the caller's pageable memory: make sure the copies have consumed them
  if (needs_mlp(m)) HIPCHK(c, hipStreamSynchronize(c->stream));
  return HBO_OK;
}

// ---- feature pipeline ----------------------------------------------------------------------
// Computes the MLP activations of x (n x D, device) into acts[l] (allocated by the caller: n x f_l)
// (w / b: the layers' device weights -- the context's own copy of the last uploaded model unless a caller keeps several)
static void run_mlp(hbo_ctx* c, const hbo_model* m, const void* x, int64_t n, void* const* acts, void* const* w = nullptr, void* const* b = nullptr) {
  const void* in = x;
  int fin = m->input_dim;
  if (!w) { w = c->d_mlp_w; b = c->d_mlp_b; }
  for (int l = 0; l < m->n_layers; ++l) {
    launch_dense_tanh(m->dtype, in, w[l], b[l], acts[l], n, fin, m->features[l], c->stream);
    in = acts[l];
    fin = m->features[l];
  }
}

struct FeatBuf {   // device activations of one input matrix
  std::vector<void*> acts; std::vector<size_t> bytes;
  ~FeatBuf() { for (void* p : acts) if (p) hipFree(p); }
  int ensure(hbo_ctx* c, const hbo_model* m, int64_t n) {
    acts.resize(HBO_MAX_MLP_LAYERS, nullptr); bytes.resize(HBO_MAX_MLP_LAYERS, 0);
    for (int l = 0; l < m->n_layers; ++l) {
      const size_t need = (size_t)std::max<int64_t>(n, 1) * m->features[l] * esize(m->dtype);
      if (bytes[l] < need) { if (acts[l]) hipFree(acts[l]); acts[l] = nullptr; HIPCHK(c, hbo_malloc(c, &acts[l], need)); bytes[l] = need; }
    }
    return HBO_OK;
  }
};

static void* pinned_stage(hbo_ctx* c, size_t bytes) {
  if (c->hp_stage_bytes < bytes) {
    if (c->hp_stage) { hipDeviceSynchronize(); hipHostFree(c->hp_stage); c->hp_stage = nullptr; c->hp_stage_bytes = 0; }
    const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
    if (hipHostMalloc(&c->hp_stage, want, hipHostMallocDefault) != hipSuccess) { c->hp_stage = nullptr; return nullptr; }
    c->hp_stage_bytes = want;
  }
  return c->hp_stage;
}

// ---- datasets ----------------------------------------------------------------------------
struct TaskHost {
  int64_t n = 0; int m = 0; int npad = 0, nblk = 0; int64_t ld = 0;
  bool owns_inputs = true;   // false: X / ysum / ydiv point into the dataset's single input block
  void* X = nullptr; void* ysum = nullptr;
  void* ydiv = nullptr;   // (m+1) x n rows for the divergence objectives: (y_a - mean_a y)/sqrt(m), then -mean_a y
  void* A = nullptr; void* W = nullptr; void* S = nullptr; void* wscr = nullptr; void* svec = nullptr; int svec_cols = 0; bool svec_shared = false;   // svec_shared: a slice of hbo_dataset::d_svec
  double* dmu = nullptr; double* fnorm = nullptr;
  double* dF = nullptr; double* dtmp = nullptr; size_t dF_elems = 0;   // MLP backward workspaces
  FeatBuf feat;
};
struct hbo_dataset {
  int dtype = 0, D = 0, ntasks = 0, max_nblk = 0;
  std::vector<TaskHost*> tasks;
  std::vector<TaskDesc> h_desc;
  TaskDesc* d_desc = nullptr;
  void* d_inputs = nullptr;   // x, column sums of y and divergence rows of every task (one upload)
  void* d_svec = nullptr;     // the tasks' alpha vectors in one block (one memset instead of one per task: a fresh batch of 64
                              // tasks per Adam step paid 64 fill kernels)
  // results of one evaluation, one device block = one copy back: [value T][gradient T x out_stride][info T (int)]
  double* d_pack = nullptr; size_t pack_bytes = 0;
  int* d_info = nullptr;
  double* d_nll = nullptr;
  std::vector<TaskDesc> h_desc_dev;   // what d_desc holds
  double* d_partials = nullptr; size_t partials_bytes = 0;
  double* d_gradout = nullptr;
  double* d_mlpgrad = nullptr; size_t mlpgrad_elems = 0;
  MlpTaskDev* d_mlp = nullptr; std::vector<MlpTaskDev> h_mlp_dev;   // per-task pointers of the batched MLP passes (what d_mlp holds)
  bool has_S = false;
};

static void free_task(hbo_ctx* c, TaskHost* t) {
  if (!t) return;
  if (t->owns_inputs) for (void* p : {t->X, t->ysum, t->ydiv}) dev_free(c, p);
  for (void* p : {t->A, t->W, t->S, t->wscr, t->svec_shared ? nullptr : t->svec, (void*)t->dmu, (void*)t->fnorm}) dev_free(c, p);
  for (void* p : {(void*)t->dF, (void*)t->dtmp}) dev_free(c, p);
  delete t;
}

static int ensure_task_workspace(hbo_ctx* c, int dtype, TaskHost* t, bool need_S, int naug_cols) {
  const size_t es = esize(dtype);
  const size_t ld = (size_t)t->ld;
  if (!t->A) HIPCHK(c, dev_alloc(c, &t->A, (size_t)(t->npad + HBO_TILE) * ld * es));
  if (!t->W) {   // a W that served the same shape before still has its zeros above the diagonal
    bool reused = false;
    HIPCHK(c, dev_alloc(c, &t->W, (size_t)t->npad * ld * es, dtype == HBO_F64 ? 2 : 1, &reused));
    if (!reused) HIPCHK(c, hipMemsetAsync(t->W, 0, (size_t)t->npad * ld * es, c->stream));
  }
  if (need_S && !t->S) HIPCHK(c, dev_alloc(c, &t->S, (size_t)t->npad * ld * es));
  if (!t->wscr) HIPCHK(c, dev_alloc(c, &t->wscr, (size_t)((t->npad + 511) / 512) * ld * es));
  if (t->svec_cols < naug_cols) {
    if (t->svec && !t->svec_shared) { HIPCHK(c, hipStreamSynchronize(c->stream)); dev_free(c, t->svec); }
    t->svec = nullptr; t->svec_shared = false;
    HIPCHK(c, dev_alloc(c, &t->svec, (size_t)t->npad * es * naug_cols));
    HIPCHK(c, hipMemsetAsync(t->svec, 0, (size_t)t->npad * es * naug_cols, c->stream));
    t->svec_cols = naug_cols;
  }
  if (!t->dmu) { HIPCHK(c, dev_alloc(c, (void**)&t->dmu, (size_t)t->npad * sizeof(double))); HIPCHK(c, dev_alloc(c, (void**)&t->fnorm, 2 * sizeof(double))); }
  return HBO_OK;
}

// role of the augmented rows (see TaskDesc): the three training objectives + the posterior cache
enum { ROLE_FACTOR = 100 };

static void fill_desc(TaskDesc& d, TaskHost* t, const hbo_model* m, int dtype, int role) {
  memset(&d, 0, sizeof d);
  d.A = t->A; d.W = t->W; d.S = t->S; d.wscr = t->wscr; d.X = t->X; d.ysum = t->ysum; d.svec = t->svec;
  d.dmu = t->dmu; d.fnorm = t->fnorm;
  const double mm = (double)t->m;
  switch (role) {
    case OBJ_NLL:   // objectives.py:144-156 incl. the (m,m)+scalar broadcast for m > 1
      d.naug = 1; d.e_last = -mm; d.coef_c = 0.5; d.coef_lh = 0.5 * mm * mm;
      d.coef_const = mm * mm * 0.5 * (double)t->n * log(2.0 * M_PI);
      break;
    case OBJ_EKL:   // utils.py:84-106 partial KL: tr(K1^-1 C0) + d^T K1^-1 d + logdet K1
      d.ysum = t->ydiv; d.naug = t->m + 1; d.e_last = 1.0; d.coef_c = 1.0; d.coef_lh = 1.0;
      break;
    case OBJ_EUC:   // utils.py:151-173 |mu0 - mu1| + |C0 - K1|_F  (no factorisation)
      d.ysum = t->ydiv; d.naug = t->m + 1; d.e_last = 1.0;
      break;
    default:        // posterior cache: rows y_a - mu
      d.naug = t->m; d.e_all = -1.0; d.coef_c = 0.5; d.coef_lh = 0.5;
      break;
  }
  d.last_src = d.naug - 1;
  if ((role == OBJ_EKL || role == OBJ_EUC) && t->m + 1 > HBO_TILE) { d.naug = 1; d.last_src = t->m; d.nvec = t->m + 1; }   // see TaskDesc::nvec
  d.n = (int)t->n; d.npad = t->npad; d.nblk = t->nblk; d.m = t->m; d.ld = t->ld;
  const void* last = needs_mlp(m) ? t->feat.acts[m->n_layers - 1] : nullptr;
  d.F = m->kernel_uses_mlp ? last : t->X;
  d.fdim = feature_dim(m);
  d.fmean = mean_feature_dim(m);
  d.Fm = (m->mean_id == HBO_MEAN_LINEAR) ? t->X : (m->mean_id == HBO_MEAN_LINEAR_MLP ? last : nullptr);
  d.dF = t->dF;
  (void)dtype;
}

// ---- GPCache -----------------------------------------------------------------------------
struct hbo_cache {
  int dtype = 0, D = 0, m = 0;
  TaskHost* t = nullptr;
  TaskDesc h_desc; TaskDesc* d_desc = nullptr;
  int* d_info = nullptr; int info = INT_MAX;
  void* resid = nullptr;   // m x npad : y - mu
  void* zvec = nullptr;    // m x npad : z = L^-1 (y - mu), kept for O(N^2) row appends
  // fp32 caches: W = L^-1 split into three bf16 planes for the posterior product (post3.hip), built at the first use
  unsigned short* w3 = nullptr; size_t w3_elems = 0; bool w3_valid = false;
  int w3_planes = 0;                 // 3: bf16x3 planes, 2: fp16 planes scaled by the power of two behind *d_wmax (post2h.hip)
  unsigned int* d_wmax = nullptr;    // device: bits of max |W|
};

static void fill_nan(void* p, size_t count, int dtype) {
  if (dtype == HBO_F64) for (size_t i = 0; i < count; ++i) ((double*)p)[i] = NAN;
  else for (size_t i = 0; i < count; ++i) ((float*)p)[i] = NAN;
}

