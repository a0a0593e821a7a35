// C ABI of libhbo (see include/hbo.h): context, device memory, orchestration of the HIP kernels
// for the GP hot path, profiling with HIP events, and the RCCL all-reduce used by task sharding.
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

thread_local std::string hbo_g_err;

#include "runtime.h"
#include "sched.h"

// ---- context -----------------------------------------------------------------------------
extern "C" const char* hbo_version(void) { return "hbo 0.1 (gfx950)"; }
extern "C" int hbo_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int hbo_device_info(int device, char* name_out, int32_t cap, int32_t* cus, int64_t* mem_bytes) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    (void)hipGetLastError();   // (not sticky: a later HIPCHK(hipGetLastError()) of this thread must not trip over it)
    return fail(nullptr, HBO_ERR_NODEV, "hbo_device_info: no such device");
  }
  if (name_out && cap > 0) { snprintf(name_out, (size_t)cap, "%s (%s)", p.name, p.gcnArchName); }
  if (cus) *cus = p.multiProcessorCount;
  if (mem_bytes) *mem_bytes = (int64_t)p.totalGlobalMem;
  return HBO_OK;
}
extern "C" const char* hbo_last_error(hbo_ctx* ctx) { return ctx ? ctx->err.c_str() : hbo_g_err.c_str(); }

extern "C" int hbo_ctx_create(int device, hbo_ctx** out) {
  if (!out) return fail(nullptr, HBO_ERR_ARG, "hbo_ctx_create: out is null");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(nullptr, HBO_ERR_NODEV, "hbo_ctx_create: no HIP device visible");
  if (device < 0 || device >= n) return fail(nullptr, HBO_ERR_ARG, "hbo_ctx_create: bad device index");
  hbo_ctx* c = new hbo_ctx();
  c->device = device;
  hbo_ctx* nullctx = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreate(&c->stream);
  if (e == hipSuccess) e = hipStreamCreate(&c->stream2);
  if (e == hipSuccess) e = hipStreamCreate(&c->stream4);
  if (e == hipSuccess) e = hbo_malloc(c, (void**)&c->d_model, sizeof(ModelDev));
  if (e == hipSuccess) e = hbo_malloc(c, (void**)&c->d_yield, sizeof(int) * HBO_YIELD_TAB_ENTRIES);
  if (e == hipSuccess) e = hipMemset(c->d_yield, 0, sizeof(int) * HBO_YIELD_TAB_ENTRIES);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_model, sizeof(ModelDev), hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming);
  if (e != hipSuccess) {
    hbo_g_err = std::string("hbo_ctx_create: ") + hipGetErrorString(e);
    delete c;
    (void)nullctx;
    return HBO_ERR_HIP;
  }
  { hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      c->n_cus = prop.multiProcessorCount;
      // parked buffers of freed datasets / caches: at most a quarter of the device memory (option pool_cap_mb)
      c->pool_cap = std::min<size_t>(c->pool_cap, (size_t)prop.totalGlobalMem / 4);
    } }
  *out = c;
  return HBO_OK;
}
extern "C" int hbo_comm_destroy(hbo_ctx* ctx);
extern "C" int hbo_ctx_destroy(hbo_ctx* c) {
  if (!c) return HBO_OK;
  hipSetDevice(c->device);
  hbo_comm_destroy(c);
  prof_begin(c);
  for (int l = 0; l < HBO_MAX_MLP_LAYERS; ++l) { if (c->d_mlp_w[l]) hipFree(c->d_mlp_w[l]); if (c->d_mlp_b[l]) hipFree(c->d_mlp_b[l]); }
  for (hipEvent_t ev : c->prof_events) hipEventDestroy(ev);
  if (c->d_model) hipFree(c->d_model);
  if (c->d_yield) hipFree(c->d_yield);
  if (c->h_model) hipHostFree(c->h_model);
  if (c->hp_stage) hipHostFree(c->hp_stage);
  if (c->ev_upload) hipEventDestroy(c->ev_upload);
  for (auto& kv : c->ws) if (kv.second.first) hipFree(kv.second.first);
  for (auto& kv : c->pool_free) for (void* p : kv.second) hipFree(p);
  c->pool_free.clear(); c->pool_live.clear(); c->pool_bytes = 0;
  for (hipEvent_t ev : c->ev_pool) hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_timed) if (ev) hipEventDestroy(ev);
  if (c->stream4) hipStreamDestroy(c->stream4);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return HBO_OK;
}
extern "C" int hbo_set_option(hbo_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return HBO_ERR_ARG;
  if (!strcmp(name, "potrf_group")) { if (value < 0 || value > 16) return fail(c, HBO_ERR_ARG, "potrf_group in 0..16 (0: auto)"); c->opt_group = (int)value; return HBO_OK; }
  if (!strcmp(name, "overlap_trtri")) { c->opt_overlap_trtri = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "dynamic_tiles")) { c->opt_dynamic_tiles = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "f1_on_chain")) { c->opt_f1_on_chain = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "lookahead")) { c->opt_lookahead = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "cu_yield")) { if (value < 0 || value > 2) return fail(c, HBO_ERR_ARG, "cu_yield in 0..2"); c->opt_cu_yield = (int)value; return HBO_OK; }
  if (!strcmp(name, "trtri_gran")) { if (value < 0) return fail(c, HBO_ERR_ARG, "trtri_gran >= 0"); c->opt_trtri_gran = (int)value; return HBO_OK; }
  if (!strcmp(name, "small_nblk")) { c->opt_small_nblk = (int)value; return HBO_OK; }
  if (!strcmp(name, "lauum_split")) { c->opt_lauum_split = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "pool_cap_mb")) {
    if (value < 0) return fail(c, HBO_ERR_ARG, "pool_cap_mb >= 0");
    c->pool_cap = (size_t)value << 20;
    if (c->pool_bytes > c->pool_cap) {   // trim: release everything parked (simple and rare)
      hipSetDevice(c->device);
      pool_release_all(c);
    }
    return HBO_OK;
  }
  if (!strcmp(name, "bulk_tail")) { c->opt_bulk_tail = (int)value; return HBO_OK; }
  if (!strcmp(name, "post_bf16x3")) { c->opt_post_bf16x3 = value != 0; return HBO_OK; }
  if (!strcmp(name, "trtri_bf16x3")) { c->opt_trtri_bf16x3 = value != 0; return HBO_OK; }
  if (!strcmp(name, "trtri3_min_s")) { if (value < 1 || value > 1024) return fail(c, HBO_ERR_ARG, "trtri3_min_s in 1..1024"); c->opt_trtri3_min_s = (int)value; return HBO_OK; }
  if (!strcmp(name, "syrk3_col")) { c->opt_syrk3_col = value != 0; return HBO_OK; }
  if (!strcmp(name, "syrk3_sep")) { c->opt_syrk3_sep = value != 0; return HBO_OK; }
  if (!strcmp(name, "syrk3_free")) { if (value < 0 || value > 200) return fail(c, HBO_ERR_ARG, "syrk3_free in 0..200"); c->opt_syrk3_free = (int)value; return HBO_OK; }
  if (!strcmp(name, "syrk_bf16x3")) { c->opt_syrk_bf16x3 = value != 0; return HBO_OK; }
  if (!strcmp(name, "post_chunk")) { if (value < 128 || value > 65536) return fail(c, HBO_ERR_ARG, "post_chunk in 128..65536"); c->opt_post_chunk = (int)value; return HBO_OK; }
  if (!strcmp(name, "trtri_at")) { if (value < 0 || value > 63) return fail(c, HBO_ERR_ARG, "trtri_at in 0..63"); c->opt_trtri_at = (int)value; return HBO_OK; }
  if (!strcmp(name, "trtri_small_wgs")) { if (value < 1 || value > 4) return fail(c, HBO_ERR_ARG, "trtri_small_wgs in 1..4"); c->opt_trtri_small_wgs = (int)value; return HBO_OK; }
  if (!strcmp(name, "trtri_free")) { if (value < 0 || value > 200) return fail(c, HBO_ERR_ARG, "trtri_free in 0..200"); c->opt_trtri_free = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag")) { if (value < 0 || value > 2) return fail(c, HBO_ERR_ARG, "dag in 0..2"); c->opt_dag = (int)value; c->dag_broken = 0; return HBO_OK; }
  if (!strcmp(name, "dag_f1_small")) { c->opt_dag_f1_small = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_join")) { c->opt_dag_join = value ? 1 : 0; return HBO_OK; }
  if (!strcmp(name, "dag_reserve")) { if (value < 0 || value > 4) return fail(c, HBO_ERR_ARG, "dag_reserve in 0..4 (CUs per shader engine)"); c->opt_dag_reserve = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_near64")) { if (value < 0 || value > 2) return fail(c, HBO_ERR_ARG, "dag_near64 in 0..2"); c->opt_dag_near64 = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_trtri")) { if (value < 0 || value > 64) return fail(c, HBO_ERR_ARG, "dag_trtri in 0..64"); c->opt_dag_trtri = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_spin_us")) { if (value < 0 || value > 1000) return fail(c, HBO_ERR_ARG, "dag_spin_us in 0..1000"); c->opt_dag_spin_us = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_idle_sleep")) { if (value < 0 || value > 1000) return fail(c, HBO_ERR_ARG, "dag_idle_sleep in 0..1000"); c->opt_dag_idle_sleep = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_dbg")) { c->opt_dag_dbg = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_max_nblk")) { c->opt_dag_max_nblk = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_min_nblk")) { c->opt_dag_min_nblk = (int)value; return HBO_OK; }
  if (!strcmp(name, "dag_timeout_ms")) { if (value < 1 || value > 60000) return fail(c, HBO_ERR_ARG, "dag_timeout_ms in 1..60000"); c->opt_dag_timeout_ms = (int)value; return HBO_OK; }
  if (!strcmp(name, "persist_free")) { if (value < -1 || value > 200) return fail(c, HBO_ERR_ARG, "persist_free in -1..200 (-1: auto)"); c->opt_persist_free = (int)value; return HBO_OK; }
  return fail(c, HBO_ERR_ARG, std::string("unknown option ") + name);
}
extern "C" int hbo_profile_enable(hbo_ctx* c, int level) { if (!c) return HBO_ERR_ARG; c->prof_level = level; return HBO_OK; }
extern "C" int hbo_profile_get(hbo_ctx* c, char names[][32], double* ms, int32_t* launches, int32_t* n) {
  if (!c || !n) return HBO_ERR_ARG;
  int k = (int)std::min<size_t>(c->prof_names.size(), HBO_MAX_PROFILE_STAGES);
  for (int i = 0; i < k; ++i) {
    if (names) { strncpy(names[i], c->prof_names[i].c_str(), 31); names[i][31] = 0; }
    if (ms) ms[i] = c->prof_ms[i];
    if (launches) launches[i] = c->prof_count[i];
  }
  *n = k;
  return HBO_OK;
}

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

// fills ctx->h_model, uploads it and the MLP weights
static int upload_model(hbo_ctx* c, const hbo_model* m) {
  int rc = validate_model(c, m);
  if (rc) return rc;
  // the pinned copy may still be read by the previous upload (calls that return without waiting for the stream)
  HIPCHK(c, hipEventSynchronize(c->ev_upload));
  ModelDev& h = *c->h_model;
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

extern "C" int hbo_grad_layout_of(const hbo_model* m, hbo_grad_layout* out) {
  if (!m || !out) return HBO_ERR_ARG;
  int pos = 0;
  const bool dot = m->kernel_id == HBO_KERNEL_DOT;
  out->lengthscale = dot ? -1 : pos; if (!dot) pos += m->n_lengthscale;
  out->signal_variance = dot ? -1 : pos; if (!dot) pos += 1;
  out->noise_variance = pos++;
  out->constant = (m->mean_id == HBO_MEAN_CONSTANT) ? pos++ : -1;
  out->dot_prod_sigma = dot ? pos++ : -1;
  out->dot_prod_bias = dot ? pos++ : -1;
  const int fm = mean_feature_dim(m);
  out->linear_kernel = fm ? pos : -1; pos += fm;
  out->linear_bias = fm ? pos++ : -1;
  for (int l = 0; l < HBO_MAX_MLP_LAYERS; ++l) { out->mlp_kernel[l] = -1; out->mlp_bias[l] = -1; }
  if (needs_mlp(m)) {
    int fin = m->input_dim;
    for (int l = 0; l < m->n_layers; ++l) {
      out->mlp_kernel[l] = pos; pos += fin * m->features[l];
      out->mlp_bias[l] = pos; pos += m->features[l];
      fin = m->features[l];
    }
  }
  out->total = pos;
  return HBO_OK;
}

// ---- feature pipeline ----------------------------------------------------------------------
// Computes the MLP activations of x (n x D, device) into acts[l] (allocated by the caller: n x f_l)
static void run_mlp(hbo_ctx* c, const hbo_model* m, const void* x, int64_t n, void* const* acts) {
  const void* in = x;
  int fin = m->input_dim;
  for (int l = 0; l < m->n_layers; ++l) {
    launch_dense_tanh(m->dtype, in, c->d_mlp_w[l], c->d_mlp_b[l], acts[l], n, fin, m->features[l], c->stream);
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
  void* A = nullptr; void* W = nullptr; void* S = nullptr; void* wscr = nullptr; void* svec = nullptr; int svec_cols = 0;
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
  // results of one evaluation, one device block = one copy back: [value T][gradient T x out_stride][info T (int)]
  double* d_pack = nullptr; size_t pack_bytes = 0;
  int* d_info = nullptr;
  double* d_nll = nullptr;
  std::vector<TaskDesc> h_desc_dev;   // what d_desc holds
  double* d_partials = nullptr; size_t partials_bytes = 0;
  double* d_gradout = nullptr;
  double* d_mlpgrad = nullptr; size_t mlpgrad_elems = 0;
  bool has_S = false;
};

static void free_task(hbo_ctx* c, TaskHost* t) {
  if (!t) return;
  if (t->owns_inputs) for (void* p : {t->X, t->ysum, t->ydiv}) dev_free(c, p);
  for (void* p : {t->A, t->W, t->S, t->wscr, t->svec, (void*)t->dmu, (void*)t->fnorm}) dev_free(c, p);
  for (void* p : {(void*)t->dF, (void*)t->dtmp}) if (p) hipFree(p);
  delete t;
}

extern "C" int hbo_dataset_free(hbo_ctx* c, hbo_dataset* ds) {
  if (!ds) return HBO_OK;
  if (c) hipSetDevice(c->device);
  // (the buffers go back to the pool while kernels of the last evaluation may still run only if a call returned without
  //  draining its streams -- none does: every entry point synchronises before it returns)
  for (TaskHost* t : ds->tasks) free_task(c, t);
  dev_free(c, ds->d_inputs);
  for (void* p : {(void*)ds->d_desc, (void*)ds->d_pack, (void*)ds->d_partials, (void*)ds->d_mlpgrad}) if (p) hipFree(p);
  delete ds;
  return HBO_OK;
}

extern "C" int hbo_dataset_create(hbo_ctx* c, int dtype, int input_dim, const hbo_task* tasks, int n_tasks,
                                  hbo_dataset** out) {
  if (!c || !out || (n_tasks > 0 && !tasks)) return fail(c, HBO_ERR_ARG, "hbo_dataset_create: null argument");
  if (dtype != HBO_F32 && dtype != HBO_F64) return fail(c, HBO_ERR_ARG, "hbo_dataset_create: bad dtype");
  if (input_dim <= 0 || input_dim > HBO_MAX_FEATURE_DIM) return fail(c, HBO_ERR_ARG, "hbo_dataset_create: bad input_dim");
  HIPCHK(c, hipSetDevice(c->device));
  hbo_dataset* ds = new hbo_dataset();
  ds->dtype = dtype; ds->D = input_dim;
  const size_t es = esize(dtype);
  // All inputs of all tasks travel in ONE block: laid out in a pinned staging buffer (x, the column sums of y, the
  // divergence rows), one host-to-device copy, the tasks point into the block.  (Three synchronous copies per task cost
  // ~1 ms for 24 small tasks -- what an Adam step of GP.train() pays when it re-samples its batch, gp.py:101-111.)
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0;
  for (int k = 0; k < n_tasks; ++k) {
    const hbo_task& tk = tasks[k];
    if (tk.n <= 0) continue;  // objectives.py:184-185: empty sub-datasets are skipped
    if (tk.m <= 0 || !tk.x || !tk.y) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_ARG, "hbo_dataset_create: bad task"); }
    total += al((size_t)tk.n * input_dim * es) + al((size_t)tk.n * es);
    if (tk.m + 1 <= HBO_TILE) total += al((size_t)(tk.m + 1) * tk.n * es);
  }
  unsigned char* stage = nullptr;
  if (total) {
    HIPCHK(c, hipEventSynchronize(c->ev_upload));   // the pinned buffer may still feed an earlier upload
    stage = static_cast<unsigned char*>(pinned_stage(c, total));
    hipError_t e = stage ? dev_alloc(c, &ds->d_inputs, total) : hipErrorOutOfMemory;
    if (e != hipSuccess) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_HIP, std::string("hbo_dataset_create: ") + hipGetErrorString(e)); }
  }
  size_t off = 0;
  for (int k = 0; k < n_tasks; ++k) {
    const hbo_task& tk = tasks[k];
    if (tk.n <= 0) continue;
    TaskHost* t = new TaskHost();
    ds->tasks.push_back(t);
    t->owns_inputs = false;
    t->n = tk.n; t->m = tk.m; t->npad = round_up(tk.n, HBO_TILE); t->nblk = t->npad / HBO_TILE; t->ld = padded_ld(t->npad, dtype);
    t->X = (char*)ds->d_inputs + off;
    memcpy(stage + off, tk.x, (size_t)tk.n * input_dim * es);
    off += al((size_t)tk.n * input_dim * es);
    // ysum (sum over columns, in double then cast)
    t->ysum = (char*)ds->d_inputs + off;
    for (int64_t i = 0; i < tk.n; ++i) {
      double sum = 0;
      for (int a = 0; a < tk.m; ++a) sum += host_elem(tk.y, dtype, i * tk.m + a);
      if (dtype == HBO_F64) ((double*)(stage + off))[i] = sum; else ((float*)(stage + off))[i] = (float)sum;
    }
    off += al((size_t)tk.n * es);
    if (tk.m + 1 <= HBO_TILE) {
      // sample statistics of objectives.py:57-58: mu_data = mean over the m aligned columns, cov_data =
      // (1/m) sum_a yc_a yc_a^T (jnp.cov(bias=True)); kept as its m rank-1 factors.
      t->ydiv = (char*)ds->d_inputs + off;
      unsigned char* yd = stage + off;
      const double rs = 1.0 / sqrt((double)tk.m);
      for (int64_t i = 0; i < tk.n; ++i) {
        double mu0 = 0;
        for (int a = 0; a < tk.m; ++a) mu0 += host_elem(tk.y, dtype, i * tk.m + a);
        mu0 /= tk.m;
        for (int a = 0; a <= tk.m; ++a) {
          const double v = a < tk.m ? (host_elem(tk.y, dtype, i * tk.m + a) - mu0) * rs : -mu0;
          if (dtype == HBO_F64) ((double*)yd)[(size_t)a * tk.n + i] = v; else ((float*)yd)[(size_t)a * tk.n + i] = (float)v;
        }
      }
      off += al((size_t)(tk.m + 1) * tk.n * es);
    }
    ds->max_nblk = std::max(ds->max_nblk, t->nblk);
  }
  if (total) {
    hipError_t e = hipMemcpyAsync(ds->d_inputs, stage, total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipEventRecord(c->ev_upload, c->stream);
    if (e != hipSuccess) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_HIP, std::string("hbo_dataset_create: ") + hipGetErrorString(e)); }
  }
  ds->ntasks = (int)ds->tasks.size();
  // largest tasks first: their tiles are dispatched first
  std::stable_sort(ds->tasks.begin(), ds->tasks.end(), [](TaskHost* a, TaskHost* b) { return a->n > b->n; });
  *out = ds;
  return HBO_OK;
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
    if (t->svec) { HIPCHK(c, hipStreamSynchronize(c->stream)); dev_free(c, t->svec); t->svec = nullptr; }
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
  d.n = (int)t->n; d.npad = t->npad; d.nblk = t->nblk; d.m = t->m; d.ld = t->ld;
  const void* last = needs_mlp(m) ? t->feat.acts[m->n_layers - 1] : nullptr;
  d.F = m->kernel_uses_mlp ? last : t->X;
  d.fdim = feature_dim(m);
  d.fmean = mean_feature_dim(m);
  d.Fm = (m->mean_id == HBO_MEAN_LINEAR) ? t->X : (m->mean_id == HBO_MEAN_LINEAR_MLP ? last : nullptr);
  d.dF = t->dF;
  (void)dtype;
}

extern "C" int hbo_nll(hbo_ctx* c, const hbo_model* m, hbo_dataset* ds, double* nll_sum, double* nll_per_task,
                       double* grad_sum) {
  return hbo_objective(c, m, ds, HBO_OBJ_NLL, nll_sum, nll_per_task, grad_sum);
}

// Task-sharded form (hbo_objective_sharded): the sums over this rank's tasks are formed on the device, all-reduced in place
// over the context's RCCL communicator and copied to the host once.
struct ShardReq { double* count; double* timing; };
int comm_allreduce_device(hbo_ctx* c, double* d_buf, int count, hipStream_t st);   // comm.hip
void launch_shard_reduce(const double* nll, const double* grad, const int* info, int T, int out_stride, const int* map,
                         const double* mlp, const int* mlp_seg, int n_mlp_seg, double* out, int out_count, hipStream_t st);   // gram.hip

static int objective_impl(hbo_ctx* c, const hbo_model* m_in, hbo_dataset* ds, int objective, double* nll_sum,
                          double* nll_per_task, double* grad_sum, const ShardReq* sh) {
  if (!c || !nll_sum || !m_in || (!ds && !sh)) return fail(c, HBO_ERR_ARG, "hbo_objective: null argument");
  if (objective != HBO_OBJ_NLL && objective != HBO_OBJ_EKL && objective != HBO_OBJ_EUC) return fail(c, HBO_ERR_ARG, "hbo_objective: unknown objective id");
  HIPCHK(c, hipSetDevice(c->device));
  hbo_model mcopy = *m_in;
  if (objective != HBO_OBJ_NLL) mcopy.eps = 0.0;   // objectives.py:63-65: cov_model = K + noise I, no jitter
  const hbo_model* m = &mcopy;
  const int obj = objective;
  const bool euc = obj == OBJ_EUC;
  int rc = validate_model(c, m);
  if (rc) return rc;
  if (ds && ds->ntasks > 0 && (m->dtype != ds->dtype || m->input_dim != ds->D)) return fail(c, HBO_ERR_ARG, "hbo_objective: model/dataset dtype or input_dim mismatch");
  hbo_grad_layout lay;
  hbo_grad_layout_of(m, &lay);
  const bool want_grad = grad_sum != nullptr;
  *nll_sum = 0;
  if (want_grad) for (int i = 0; i < lay.total; ++i) grad_sum[i] = 0;
  const int T = ds ? ds->ntasks : 0;
  hipStream_t st = c->stream;
  // sharded: [nll, count, grad] of the whole job, reduced on the device
  const int red_count = 2 + (want_grad ? lay.total : 0);
  auto finish_sharded = [&](double* d_red, hipEvent_t ev0, hipEvent_t ev1) -> int {
    hipEvent_t ev2 = pool_event_timed(c, 2);
    int rc = c->comm ? comm_allreduce_device(c, d_red, red_count, st) : HBO_OK;
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(ev2, st));
    double* stage = static_cast<double*>(pinned_stage(c, sizeof(double) * red_count));
    if (!stage) return fail(c, HBO_ERR_HIP, "hbo_objective_sharded: pinned staging buffer");
    HIPCHK(c, hipMemcpyAsync(stage, d_red, sizeof(double) * red_count, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *nll_sum = stage[0];
    *sh->count = stage[1];
    if (want_grad) for (int i = 0; i < lay.total; ++i) grad_sum[i] = stage[2 + i];
    if (sh->timing) {
      float ms_local = 0, ms_comm = 0;
      hipEventElapsedTime(&ms_local, ev0, ev1); hipEventElapsedTime(&ms_comm, ev1, ev2);
      sh->timing[0] = ms_local; sh->timing[1] = 1e3 * ms_comm;
    }
    return HBO_OK;
  };
  if (T == 0) {
    if (!sh) return HBO_OK;
    // a rank beyond the task count: zeros into the collective
    HIPCHK(c, hipSetDevice(c->device));
    double* d_red = static_cast<double*>(ws_get(c, WS_SHARD_RED, sizeof(double) * red_count));
    if (!d_red) return HBO_ERR_HIP;
    hipEvent_t ev0 = pool_event_timed(c, 0), ev1 = pool_event_timed(c, 1);
    HIPCHK(c, hipEventRecord(ev0, st));
    HIPCHK(c, hipMemsetAsync(d_red, 0, sizeof(double) * red_count, st));
    HIPCHK(c, hipEventRecord(ev1, st));
    return finish_sharded(d_red, ev0, ev1);
  }
  if (obj != OBJ_NLL) for (TaskHost* t : ds->tasks) if (!t->ydiv) return fail(c, HBO_ERR_UNSUPPORTED, "hbo_objective: divergence objectives need m + 1 <= 128 aligned columns");
  const int dtype = ds->dtype;
  prof_begin(c);
  hipEvent_t ev_sh0 = nullptr;
  if (sh) { ev_sh0 = pool_event_timed(c, 0); HIPCHK(c, hipEventRecord(ev_sh0, st)); }
  rc = upload_model(c, m);
  if (rc) return rc;

  // workspaces + descriptors
  ds->h_desc.resize(T);
  for (int k = 0; k < T; ++k) {
    TaskHost* t = ds->tasks[k];
    rc = ensure_task_workspace(c, dtype, t, want_grad && !euc, obj == OBJ_NLL ? 1 : t->m + 1);
    if (rc) return rc;
    if (needs_mlp(m)) { rc = t->feat.ensure(c, m, t->n); if (rc) return rc; }
    if (needs_mlp(m) && want_grad) {
      int maxf = m->input_dim;
      for (int l = 0; l < m->n_layers; ++l) maxf = std::max(maxf, (int)m->features[l]);
      const size_t need = (size_t)t->n * maxf;
      if (t->dF_elems < need) {
        if (t->dF) hipFree(t->dF);
        if (t->dtmp) hipFree(t->dtmp);
        t->dF = t->dtmp = nullptr;
        HIPCHK(c, hbo_malloc(c, (void**)&t->dF, need * sizeof(double)));
        HIPCHK(c, hbo_malloc(c, (void**)&t->dtmp, need * sizeof(double)));
        t->dF_elems = need;
      }
    }
    fill_desc(ds->h_desc[k], t, m, dtype, obj);
  }
  if (!ds->d_desc) HIPCHK(c, hbo_malloc(c, (void**)&ds->d_desc, sizeof(TaskDesc) * T));
  const int out_stride = (m->kernel_id == HBO_KERNEL_DOT ? 0 : m->n_lengthscale) + 6 + mean_feature_dim(m);
  const size_t pack_bytes = sizeof(double) * T * (1 + (size_t)out_stride) + sizeof(int) * T;
  if (ds->pack_bytes < pack_bytes) {
    if (ds->d_pack) hipFree(ds->d_pack);
    ds->d_pack = nullptr; ds->pack_bytes = 0;
    HIPCHK(c, hbo_malloc(c, (void**)&ds->d_pack, pack_bytes));
    ds->pack_bytes = pack_bytes;
  }
  ds->d_nll = ds->d_pack; ds->d_gradout = ds->d_pack + T; ds->d_info = reinterpret_cast<int*>(ds->d_pack + T + (size_t)T * out_stride);
  // small transfers go through one pinned buffer: descriptors up (only when they changed), results down in one copy
  unsigned char* stage = static_cast<unsigned char*>(pinned_stage(c, std::max(sizeof(TaskDesc) * T, pack_bytes)));
  if (!stage) return fail(c, HBO_ERR_HIP, "hbo_objective: pinned staging buffer");
  if (ds->h_desc_dev.size() != (size_t)T || memcmp(ds->h_desc_dev.data(), ds->h_desc.data(), sizeof(TaskDesc) * T) != 0) {
    HIPCHK(c, hipEventSynchronize(c->ev_upload));
    memcpy(stage, ds->h_desc.data(), sizeof(TaskDesc) * T);
    HIPCHK(c, hipMemcpyAsync(ds->d_desc, stage, sizeof(TaskDesc) * T, hipMemcpyHostToDevice, st));
    ds->h_desc_dev = ds->h_desc;
  }
  HIPCHK(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ds->d_info), INT_MAX, T, st));

  const int max_nblk = ds->max_nblk, max_npad = max_nblk * HBO_TILE;
  {
    ProfScope ps(c, "features", 1);
    if (needs_mlp(m)) for (int k = 0; k < T; ++k) run_mlp(c, m, ds->tasks[k]->X, ds->tasks[k]->n, ds->tasks[k]->feat.acts.data());
    launch_aug_rows(dtype, ds->d_desc, T, max_npad, c->d_model, st);
  }
  int max_naug = 1;
  for (int k = 0; k < T; ++k) max_naug = std::max(max_naug, ds->h_desc[k].naug);
  TrtriProgress trtri_pg;
  hipStream_t side = st; hipEvent_t ev_side = nullptr;
  const bool early_trtri = want_grad && c->opt_lookahead && c->opt_overlap_trtri && max_nblk >= 4;
  if (!euc) {
    {
      ProfScope ps(c, "gram", 1);
      GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.tasks = ds->d_desc; g.fdim = feature_dim(m); g.symmetric = 1; g.padded = 1;
      launch_gram(dtype, g, c->d_model, dim3(max_nblk, max_nblk, T), st);
    }
    {
      std::vector<int> h_nblk(T);
      for (int k = 0; k < T; ++k) h_nblk[k] = ds->h_desc[k].nblk;
      c->trtri_host_task = TaskDesc{};
      if (T == 1) c->trtri_host_task = ds->h_desc[0];
      ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, ds->d_desc, T, max_nblk, ds->d_info, early_trtri ? &trtri_pg : nullptr, h_nblk.data());
    }
    // the small reductions (log-determinant + quadratic form now, alpha = W^T z and d nll / d mu after the inverse) run on
    // the idle panel stream beside the inverse and K^-1 = W^T W instead of between them (0.14 ms at cfg 2)
    side = (want_grad && obj == OBJ_NLL && c->opt_lookahead) ? c->stream2 : st;
    if (side != st) { hipEvent_t e = pool_event(c, 2); hipEventRecord(e, st); hipStreamWaitEvent(side, e, 0); }
    { ProfScope ps(c, "nll_reduce", 1, side); launch_nll_reduce(dtype, ds->d_desc, T, ds->d_info, ds->d_nll, side); }
  }

  const int fdim = feature_dim(m);
  const int nacc = grad_nacc(m->kernel_id, fdim);
  const int64_t stride_task = (int64_t)(max_nblk * (max_nblk + 1)) * nacc;   // two half-tile slots per lower tile
  if (want_grad || euc) {   // EUC: the Frobenius norm of the value comes out of the contraction pass
    const size_t pb = sizeof(double) * (stride_task * T + (size_t)HBO_GRAD_PRE_ROWS * nacc * T);   // per-tile partials + their pre-reduction
    if (ds->partials_bytes < pb) { if (ds->d_partials) hipFree(ds->d_partials); HIPCHK(c, hbo_malloc(c, (void**)&ds->d_partials, pb)); ds->partials_bytes = pb; }
    if (!euc) {
      // The tail of the inverse is a chain of small dependent products (the tree over the last panels) before its top-level
      // product: the machine is mostly idle for ~0.5 ms.  The part of K^-1 = W^T W that only needs W11 (final since the
      // overlapped inverse walked the leading blocks) runs beside it on the side stream.
      const int lsplit = (T == 1 && early_trtri && c->opt_lauum_split && max_nblk > c->opt_small_nblk) ? lauum_split_for(trtri_pg.diag, max_nblk) : 0;
      hipEvent_t ev_l1 = nullptr;
      if (lsplit > 0) {
        hipEvent_t e0 = pool_event(c, 0); hipEventRecord(e0, st); hipStreamWaitEvent(c->stream4, e0, 0);
        { ProfScope ps(c, "lauum_early", 1, c->stream4); run_lauum(c, dtype, ds->d_desc, T, max_nblk, lsplit, 1, c->stream4); }
        ev_l1 = pool_event(c, 1); hipEventRecord(ev_l1, c->stream4);
      }
      { ProfScope ps(c, "trtri", 1);
        run_trtri(c, dtype, ds->d_desc, T, max_nblk, &trtri_pg); }
      if (ev_l1) hipStreamWaitEvent(st, ev_l1, 0);
      if (side != st) { hipEvent_t e = pool_event(c, 3); hipEventRecord(e, st); hipStreamWaitEvent(side, e, 0); }   // W is complete
      { ProfScope ps(c, "wt_z", 1, side);
        for (int b = 0; b < max_naug; ++b) launch_wt_z(dtype, ds->d_desc, T, max_nblk, b, b, max_npad, side); }
      if (side != st) {
        launch_dmu(dtype, ds->d_desc, T, obj, side);
        ev_side = pool_event(c, 4); hipEventRecord(ev_side, side);
      }
      { ProfScope ps(c, "lauum", 1); run_lauum(c, dtype, ds->d_desc, T, max_nblk, lsplit, lsplit > 0 ? 2 : 0); }
      if (ev_side) hipStreamWaitEvent(st, ev_side, 0);
    }
    { ProfScope ps(c, "grad_contract", 1);
      if (!ev_side) launch_dmu(dtype, ds->d_desc, T, obj, st);
      launch_grad_contract(dtype, ds->d_desc, T, max_nblk, c->d_model, m->kernel_id, fdim, obj, ds->d_partials, stride_task, st);
      launch_grad_finalize(dtype, ds->d_desc, T, c->d_model, m->kernel_id, fdim, obj, ds->d_partials, stride_task, ds->d_gradout, out_stride, euc ? ds->d_nll : nullptr, st,
                           ds->d_partials + stride_task * T, max_nblk); }
  }
  if (want_grad) {
    if (needs_mlp(m)) {
      // d nll / d features -> MLP backward (hyperbo/gp_utils/basis_functions.py:24-36), summed over tasks
      ProfScope ps(c, "mlp_backward", 1);
      const int L = m->n_layers, flast = m->features[L - 1];
      size_t tot = 0; int fin0 = m->input_dim;
      std::vector<size_t> woff(L), boff(L);
      for (int l = 0; l < L; ++l) { woff[l] = tot; tot += (size_t)fin0 * m->features[l]; boff[l] = tot; tot += m->features[l]; fin0 = m->features[l]; }
      if (ds->mlpgrad_elems < tot) { if (ds->d_mlpgrad) hipFree(ds->d_mlpgrad); HIPCHK(c, hbo_malloc(c, (void**)&ds->d_mlpgrad, tot * sizeof(double))); ds->mlpgrad_elems = tot; }
      HIPCHK(c, hipMemsetAsync(ds->d_mlpgrad, 0, tot * sizeof(double), st));
      for (int k = 0; k < T; ++k) HIPCHK(c, hipMemsetAsync(ds->tasks[k]->dF, 0, (size_t)ds->tasks[k]->n * flast * sizeof(double), st));
      if (m->kernel_uses_mlp) {
        launch_grad_feat(dtype, ds->d_desc, T, max_nblk, c->d_model, m->kernel_id, flast, obj, st);
        if (euc) launch_scale_dF(ds->d_desc, T, (int64_t)max_npad, flast, st);
      }
      if (m->mean_id == HBO_MEAN_LINEAR_MLP) launch_grad_feat_mean(dtype, ds->d_desc, T, (int64_t)max_npad, c->d_model, flast, st);
      for (int k = 0; k < T; ++k) {
        TaskHost* t = ds->tasks[k];
        double* cur = t->dF; double* other = t->dtmp;
        for (int l = L - 1; l >= 0; --l) {
          const int fin = l ? m->features[l - 1] : m->input_dim;
          const void* in = l ? t->feat.acts[l - 1] : t->X;
          launch_dense_bwd(dtype, in, t->feat.acts[l], c->d_mlp_w[l], cur, l ? other : nullptr,
                           ds->d_mlpgrad + woff[l], ds->d_mlpgrad + boff[l], t->n, fin, m->features[l], st);
          std::swap(cur, other);
        }
      }
    }
  }
  if (sh) {
    // [nll, count, grad] of this rank's tasks in the caller's gradient layout, on the device: entry j of a task's gradient block
    // goes to map[j] (the scatter the host loop below does), the MLP gradient -- already summed over the tasks -- by segments
    const int n_ls = m->kernel_id == HBO_KERNEL_DOT ? 0 : m->n_lengthscale;
    const int fm = mean_feature_dim(m);
    std::vector<int> hmap(out_stride + 3 * 2 * HBO_MAX_MLP_LAYERS, -1);
    if (want_grad) {
      for (int d = 0; d < n_ls; ++d) hmap[d] = lay.lengthscale < 0 ? -1 : lay.lengthscale + d;
      hmap[n_ls] = lay.signal_variance; hmap[n_ls + 1] = lay.noise_variance; hmap[n_ls + 2] = lay.constant;
      hmap[n_ls + 3] = lay.dot_prod_sigma; hmap[n_ls + 4] = lay.dot_prod_bias;
      for (int d = 0; d < fm; ++d) hmap[n_ls + 5 + d] = lay.linear_kernel < 0 ? -1 : lay.linear_kernel + d;
      hmap[n_ls + 5 + fm] = lay.linear_bias;
    }
    int nseg = 0;
    if (want_grad && needs_mlp(m)) {
      int pos = 0, fin0 = m->input_dim;
      for (int l = 0; l < m->n_layers; ++l) {
        const int wn = fin0 * m->features[l], bn = m->features[l];
        int* sg = hmap.data() + out_stride + 3 * nseg;
        sg[0] = lay.mlp_kernel[l]; sg[1] = pos; sg[2] = wn; ++nseg; pos += wn;
        sg += 3; sg[0] = lay.mlp_bias[l]; sg[1] = pos; sg[2] = bn; ++nseg; pos += bn;
        fin0 = m->features[l];
      }
    }
    int* d_map = static_cast<int*>(ws_get(c, WS_SHARD_MAP, sizeof(int) * hmap.size()));
    double* d_red = static_cast<double*>(ws_get(c, WS_SHARD_RED, sizeof(double) * red_count));
    if (!d_map || !d_red) return HBO_ERR_HIP;
    HIPCHK(c, hipEventSynchronize(c->ev_upload));
    memcpy(stage, hmap.data(), sizeof(int) * hmap.size());
    HIPCHK(c, hipMemcpyAsync(d_map, stage, sizeof(int) * hmap.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_upload, st));
    launch_shard_reduce(ds->d_nll, want_grad ? ds->d_gradout : nullptr, ds->d_info, T, out_stride, d_map, ds->d_mlpgrad, d_map + out_stride, nseg,
                        d_red, red_count, st);
    hipEvent_t ev1 = pool_event_timed(c, 1);
    HIPCHK(c, hipEventRecord(ev1, st));
    int rcs = finish_sharded(d_red, ev_sh0, ev1);
    HIPCHK(c, hipGetLastError());
    prof_collect(c);
    if (dag_aborted(c)) return objective_impl(c, m_in, ds, objective, nll_sum, nll_per_task, grad_sum, sh);
    if (rcs) return rcs;
    return std::isnan(*nll_sum) ? HBO_NOT_PD : HBO_OK;
  }
  HIPCHK(c, hipMemcpyAsync(stage, ds->d_pack, pack_bytes, hipMemcpyDeviceToHost, st));
  const double* h_nll = reinterpret_cast<const double*>(stage);
  const double* h_grad = h_nll + T;
  const int* h_info = reinterpret_cast<const int*>(h_grad + (size_t)T * out_stride);
  std::vector<double> h_mlp;
  if (want_grad && needs_mlp(m)) {
    h_mlp.resize(ds->mlpgrad_elems);
    HIPCHK(c, hipMemcpyAsync(h_mlp.data(), ds->d_mlpgrad, sizeof(double) * ds->mlpgrad_elems, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipGetLastError());
  prof_collect(c);
  // the resident tile-task schedule ran out of its wall-clock bound (it never has; a hang would be a dead GPU): the context
  // falls back to the launch schedule for good and this evaluation is repeated on it
  if (dag_aborted(c)) return objective_impl(c, m_in, ds, objective, nll_sum, nll_per_task, grad_sum, sh);

  bool notpd = false;
  double total = 0;
  for (int k = 0; k < T; ++k) { total += h_nll[k]; if (h_info[k] != INT_MAX) notpd = true; if (nll_per_task) nll_per_task[k] = h_nll[k]; }
  *nll_sum = total;
  if (want_grad) {
    const int n_ls = m->kernel_id == HBO_KERNEL_DOT ? 0 : m->n_lengthscale;
    const int fm = mean_feature_dim(m);
    for (int k = 0; k < T; ++k) {
      const double* o = h_grad + (size_t)k * out_stride;
      const bool bad = h_info[k] != INT_MAX;
      auto add = [&](int off, double v) { if (off >= 0) grad_sum[off] += bad ? NAN : v; };
      for (int d = 0; d < n_ls; ++d) add(lay.lengthscale < 0 ? -1 : lay.lengthscale + d, o[d]);
      add(lay.signal_variance, o[n_ls]);
      add(lay.noise_variance, o[n_ls + 1]);
      add(lay.constant, o[n_ls + 2]);
      add(lay.dot_prod_sigma, o[n_ls + 3]);
      add(lay.dot_prod_bias, o[n_ls + 4]);
      for (int d = 0; d < fm; ++d) add(lay.linear_kernel + d, o[n_ls + 5 + d]);
      add(lay.linear_bias, o[n_ls + 5 + fm]);
    }
    if (needs_mlp(m)) {
      size_t pos = 0; int fin0 = m->input_dim;
      for (int l = 0; l < m->n_layers; ++l) {
        const size_t wn = (size_t)fin0 * m->features[l], bn = m->features[l];
        for (size_t i = 0; i < wn; ++i) grad_sum[lay.mlp_kernel[l] + i] = notpd ? NAN : h_mlp[pos + i];
        pos += wn;
        for (size_t i = 0; i < bn; ++i) grad_sum[lay.mlp_bias[l] + i] = notpd ? NAN : h_mlp[pos + i];
        pos += bn; fin0 = m->features[l];
      }
    }
  }
  return notpd ? HBO_NOT_PD : HBO_OK;
}
extern "C" int hbo_objective(hbo_ctx* c, const hbo_model* m, hbo_dataset* ds, int objective, double* nll_sum,
                             double* nll_per_task, double* grad_sum) {
  if (!ds) return fail(c, HBO_ERR_ARG, "hbo_objective: null argument");
  return objective_impl(c, m, ds, objective, nll_sum, nll_per_task, grad_sum, nullptr);
}
extern "C" int hbo_objective_sharded(hbo_ctx* c, const hbo_model* m, hbo_dataset* ds, int objective, double* value_sum,
                                     double* count, double* grad_sum, double* timing) {
  if (!count) return fail(c, HBO_ERR_ARG, "hbo_objective_sharded: null argument");
  ShardReq sh{count, timing};
  return objective_impl(c, m, ds, objective, value_sum, nullptr, grad_sum, &sh);
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
};

extern "C" int hbo_cache_free(hbo_ctx* c, hbo_cache* k) {
  if (!k) return HBO_OK;
  if (c) hipSetDevice(c->device);
  free_task(c, k->t);
  for (void* p : {(void*)k->d_desc, (void*)k->d_info, k->resid, k->zvec, (void*)k->w3}) if (p) hipFree(p);
  delete k;
  return HBO_OK;
}

extern "C" int hbo_factor(hbo_ctx* c, const hbo_model* m, const void* x, int64_t n, const void* y, int32_t mcols,
                          hbo_cache** out) {
  if (!c || !out || !x || !y) return fail(c, HBO_ERR_ARG, "hbo_factor: null argument");
  if (n <= 0 || mcols <= 0 || mcols > HBO_TILE) return fail(c, HBO_ERR_ARG, "hbo_factor: need n>0 and 1<=m<=128");
  HIPCHK(c, hipSetDevice(c->device));
  prof_begin(c);
  int rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = m->dtype;
  const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  hbo_cache* k = new hbo_cache();
  k->dtype = dtype; k->D = m->input_dim; k->m = mcols;
  TaskHost* t = k->t = new TaskHost();
  t->n = n; t->m = mcols; t->npad = round_up(n, HBO_TILE); t->nblk = t->npad / HBO_TILE; t->ld = padded_ld(t->npad, dtype);
  auto bail = [&](int code) { hbo_cache_free(c, k); return code; };
#define HIPCHK_K(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); return bail(HBO_ERR_HIP); } } while (0)
  HIPCHK_K(dev_alloc(c, &t->X, (size_t)t->npad * m->input_dim * es));   // capacity npad rows (row appends)
  HIPCHK_K(hipMemcpy(t->X, x, (size_t)n * m->input_dim * es, hipMemcpyHostToDevice));
  // y^T (m x n) so that aug row a = column a of y
  std::vector<unsigned char> yt((size_t)n * mcols * es);
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < mcols; ++a) memcpy(yt.data() + ((size_t)a * n + i) * es, (const unsigned char*)y + ((size_t)i * mcols + a) * es, es);
  HIPCHK_K(dev_alloc(c, &t->ysum, (size_t)n * mcols * es));
  HIPCHK_K(hipMemcpy(t->ysum, yt.data(), (size_t)n * mcols * es, hipMemcpyHostToDevice));
  rc = ensure_task_workspace(c, dtype, t, true, mcols);
  if (rc) return bail(rc);
  if (needs_mlp(m)) { rc = t->feat.ensure(c, m, t->npad); if (rc) return bail(rc); }
  fill_desc(k->h_desc, t, m, dtype, ROLE_FACTOR);
  HIPCHK_K(hbo_malloc(c, (void**)&k->d_desc, sizeof(TaskDesc)));
  HIPCHK_K(hbo_malloc(c, (void**)&k->d_info, sizeof(int)));
  HIPCHK_K(hbo_malloc(c, &k->resid, (size_t)mcols * t->npad * es));
  HIPCHK_K(hbo_malloc(c, &k->zvec, (size_t)mcols * t->npad * es));
  HIPCHK_K(hipMemcpy(k->d_desc, &k->h_desc, sizeof(TaskDesc), hipMemcpyHostToDevice));
  int inf = INT_MAX;
  HIPCHK_K(hipMemcpy(k->d_info, &inf, sizeof(int), hipMemcpyHostToDevice));

  { ProfScope ps(c, "features", 1);
    if (needs_mlp(m)) run_mlp(c, m, t->X, n, t->feat.acts.data());
    launch_aug_rows(dtype, k->d_desc, 1, t->npad, c->d_model, st); }
  HIPCHK_K(hipMemcpy2DAsync(k->resid, (size_t)t->npad * es, (char*)t->A + (size_t)t->npad * t->ld * es, (size_t)t->ld * es, (size_t)t->npad * es, mcols, hipMemcpyDeviceToDevice, st));
  { ProfScope ps(c, "gram", 1);
    GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.tasks = k->d_desc; g.fdim = feature_dim(m); g.symmetric = 1; g.padded = 1;
    launch_gram(dtype, g, c->d_model, dim3(t->nblk, t->nblk, 1), st); }
  // the inverse W = L^-1 (kept for the posterior products) starts beside the panel chain, as in the objective path
  TrtriProgress trtri_pg;
  const bool early_trtri = c->opt_lookahead && c->opt_overlap_trtri && t->nblk >= 4;
  c->trtri_host_task = k->h_desc;
  { ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, k->d_desc, 1, t->nblk, k->d_info, early_trtri ? &trtri_pg : nullptr); }
  HIPCHK_K(hipMemcpy2DAsync(k->zvec, (size_t)t->npad * es, (char*)t->A + (size_t)t->npad * t->ld * es, (size_t)t->ld * es, (size_t)t->npad * es, mcols, hipMemcpyDeviceToDevice, st));
  { ProfScope ps(c, "trtri", 1); run_trtri(c, dtype, k->d_desc, 1, t->nblk, &trtri_pg); }
  { ProfScope ps(c, "wt_z", 1);
    for (int a = 0; a < mcols; ++a) launch_wt_z(dtype, k->d_desc, 1, t->nblk, a, a, t->npad, st); }
  HIPCHK_K(hipMemcpyAsync(&k->info, k->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK_K(hipStreamSynchronize(st));
  HIPCHK_K(hipGetLastError());
  prof_collect(c);
  if (dag_aborted(c)) { hbo_cache_free(c, k); return hbo_factor(c, m, x, n, y, mcols, out); }
#undef HIPCHK_K
  *out = k;
  return k->info != INT_MAX ? HBO_NOT_PD : HBO_OK;
}

// O(N^2) row append (the reference re-factorises from scratch after every BO
// observation, hyperbo/bo_utils/bayesopt.py:186-190, and notes "One can potentially support rank-1
// updates", hyperbo/gp_utils/gp.py:284).  For each new point (x*, y*), with W = L^-1 resident:
//   l = W k(X,x*),  d = sqrt(k(x*,x*) + sigma^2 + eps - l.l),  L' = [[L,0],[l^T,d]],
//   W' = [[W,0],[-(l^T W)/d, 1/d]],  z' = [z; (r* - l.z)/d],  alpha' = [alpha + w' z'_n ; z'_n/d].
// Two triangular mat-vecs on the device, O(n) arithmetic on the host.  Returns HBO_ERR_UNSUPPORTED
// when the padded capacity (npad) is exhausted -- the caller then re-factorises.
extern "C" int hbo_cache_append(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* x_new, int64_t n_new,
                                const void* y_new) {
  if (!c || !k || !x_new || !y_new) return fail(c, HBO_ERR_ARG, "hbo_cache_append: null argument");
  if (n_new <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  TaskHost* t = k->t;
  if (k->dtype != m->dtype || k->D != m->input_dim) return fail(c, HBO_ERR_ARG, "hbo_cache_append: cache/model mismatch");
  if (k->info != INT_MAX) return HBO_NOT_PD;
  if (t->n + n_new > t->npad) return fail(c, HBO_ERR_UNSUPPORTED, "hbo_cache_append: capacity exhausted (re-factorise)");
  k->w3_valid = false;   // W changes: its bf16 planes are rebuilt at the next posterior call
  int rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = k->dtype; const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  const int fdim = feature_dim(m), fm = mean_feature_dim(m), mc = k->m;
  void *d_kx = nullptr, *d_l = nullptr, *d_w = nullptr, *d_mu = nullptr, *d_kd = nullptr;
  auto cleanup = [&]() {};   // ctx-owned scratch
#define HIPCHK_A(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); cleanup(); return HBO_ERR_HIP; } } while (0)
  d_kx = ws_get(c, WS_AP_KX, (size_t)t->npad * es); d_l = ws_get(c, WS_AP_L, (size_t)t->npad * es);
  d_w = ws_get(c, WS_AP_W, (size_t)t->npad * es); d_mu = ws_get(c, WS_AP_MU, 16); d_kd = ws_get(c, WS_AP_KD, 16);
  if (!d_kx || !d_l || !d_w || !d_mu || !d_kd) return HBO_ERR_HIP;
  std::vector<double> l(t->npad), w(t->npad), z((size_t)mc * t->npad), al((size_t)mc * t->npad);
  std::vector<unsigned char> buf((size_t)t->npad * es * std::max(mc, 1));
  auto to_host = [&](const void* dev, std::vector<double>& out, size_t count) -> hipError_t {
    hipError_t e = hipMemcpy(buf.data(), dev, count * es, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < count; ++i) out[i] = host_elem(buf.data(), dtype, (int64_t)i);
    return e;
  };
  auto to_dev = [&](void* dev, const double* src, size_t count) -> hipError_t {
    for (size_t i = 0; i < count; ++i) { if (dtype == HBO_F64) ((double*)buf.data())[i] = src[i]; else ((float*)buf.data())[i] = (float)src[i]; }
    return hipMemcpy(dev, buf.data(), count * es, hipMemcpyHostToDevice);
  };
  HIPCHK_A(to_host(k->zvec, z, (size_t)mc * t->npad));
  HIPCHK_A(to_host(t->svec, al, (size_t)mc * t->npad));
  int status = HBO_OK;
  for (int64_t q = 0; q < n_new && status == HBO_OK; ++q) {
    const int64_t n = t->n;
    // new input row -> X[n], features -> acts[.][n]
    void* xrow = (char*)t->X + (size_t)n * m->input_dim * es;
    HIPCHK_A(hipMemcpyAsync(xrow, (const char*)x_new + (size_t)q * m->input_dim * es, (size_t)m->input_dim * es, hipMemcpyHostToDevice, st));
    const void* flast = nullptr;
    if (needs_mlp(m)) {
      void* rows[HBO_MAX_MLP_LAYERS];
      for (int lyr = 0; lyr < m->n_layers; ++lyr) rows[lyr] = (char*)t->feat.acts[lyr] + (size_t)n * m->features[lyr] * es;
      run_mlp(c, m, xrow, 1, rows);
      flast = rows[m->n_layers - 1];
    }
    const void* Fq = m->kernel_uses_mlp ? flast : xrow;
    const void* Fmq = (m->mean_id == HBO_MEAN_LINEAR) ? xrow : (m->mean_id == HBO_MEAN_LINEAR_MLP ? flast : nullptr);
    launch_mean(dtype, Fmq, 1, fm, c->d_model, d_mu, st);
    launch_kdiag(dtype, Fq, 1, fdim, c->d_model, d_kd, st);
    // k(X, x*)  (n x 1), zero-padded to npad
    HIPCHK_A(hipMemsetAsync(d_kx, 0, (size_t)t->npad * es, st));
    { GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = k->h_desc.F; g.x2 = Fq; g.out = d_kx; g.n1 = n; g.n2 = 1; g.ldo = 1; g.fdim = fdim;
      launch_gram(dtype, g, c->d_model, dim3(1, (unsigned)((n + 127) / 128), 1), st); }
    // l = W kx ; wl = W^T l
    launch_tri_matvec(dtype, t->W, t->ld, t->npad, d_kx, t->npad, 1, 0, d_l, t->npad, st);
    launch_wt_z(dtype, k->d_desc, 1, t->nblk, 0, 0, t->npad, st, d_l, d_w);   // W^T l (two-stage, uses S as scratch)
    HIPCHK_A(hipStreamSynchronize(st));
    HIPCHK_A(to_host(d_l, l, (size_t)t->npad));
    HIPCHK_A(to_host(d_w, w, (size_t)t->npad));
    std::vector<double> one(1);
    HIPCHK_A(to_host(d_mu, one, 1)); const double mu_new = one[0];
    HIPCHK_A(to_host(d_kd, one, 1)); const double kappa = one[0] + m->noise_variance + m->eps;
    double ll = 0;
    for (int64_t i = 0; i < n; ++i) ll += l[i] * l[i];
    const double d2 = kappa - ll;
    if (!(d2 > 0)) { status = HBO_NOT_PD; k->info = (int)n + 1; break; }
    const double d = sqrt(d2);
    for (int64_t i = 0; i < n; ++i) w[i] = -w[i] / d;      // new row of W (columns < n)
    // write row n of L and of W (identity padding row is overwritten)
    l[n] = d; w[n] = 1.0 / d;
    HIPCHK_A(to_dev((char*)t->A + (size_t)n * t->ld * es, l.data(), (size_t)n + 1));
    HIPCHK_A(to_dev((char*)t->W + (size_t)n * t->ld * es, w.data(), (size_t)n + 1));
    for (int a = 0; a < mc; ++a) {
      double* za = z.data() + (size_t)a * t->npad; double* aa = al.data() + (size_t)a * t->npad;
      const double r_new = host_elem(y_new, dtype, q * mc + a) - mu_new;
      double lz = 0;
      for (int64_t i = 0; i < n; ++i) lz += l[i] * za[i];
      const double zn = (r_new - lz) / d;
      za[n] = zn;
      for (int64_t i = 0; i < n; ++i) aa[i] += w[i] * zn;
      aa[n] = zn / d;
      // residual buffer (y - mu) gains the new entry
      double rr = r_new;
      HIPCHK_A(to_dev((char*)k->resid + ((size_t)a * t->npad + n) * es, &rr, 1));
    }
    t->n = n + 1;
    k->h_desc.n = (int)t->n;
  }
  if (status == HBO_OK || status == HBO_NOT_PD) {
    HIPCHK_A(to_dev(k->zvec, z.data(), (size_t)mc * t->npad));
    HIPCHK_A(to_dev(t->svec, al.data(), (size_t)mc * t->npad));
    HIPCHK_A(hipMemcpy(k->d_desc, &k->h_desc, sizeof(TaskDesc), hipMemcpyHostToDevice));
  }
  HIPCHK_A(hipGetLastError());
#undef HIPCHK_A
  cleanup();
  return status;
}

static void fill_nan(void* p, size_t count, int dtype) {
  if (dtype == HBO_F64) for (size_t i = 0; i < count; ++i) ((double*)p)[i] = NAN;
  else for (size_t i = 0; i < count; ++i) ((float*)p)[i] = NAN;
}

extern "C" int hbo_cache_export(hbo_ctx* c, hbo_cache* k, void* chol_out, void* kinvy_out, void* ymu_out) {
  if (!c || !k) return fail(c, HBO_ERR_ARG, "hbo_cache_export: null argument");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t es = esize(k->dtype);
  TaskHost* t = k->t;
  const int64_t n = t->n;
  const bool bad = k->info != INT_MAX;
  if (chol_out) {
    if (bad) fill_nan(chol_out, (size_t)n * n, k->dtype);
    else {
      void* tmp = nullptr;
      HIPCHK(c, hbo_malloc(c, &tmp, (size_t)n * n * es));
      launch_extract_lower(k->dtype, t->A, t->ld, n, tmp, c->stream);
      hipError_t e = hipMemcpyAsync(chol_out, tmp, (size_t)n * n * es, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      hipFree(tmp);
      HIPCHK(c, e);
    }
  }
  std::vector<unsigned char> buf((size_t)k->m * t->npad * es);
  auto export_cols = [&](const void* dev, void* outp, bool nanfill) -> int {
    if (nanfill) { fill_nan(outp, (size_t)n * k->m, k->dtype); return HBO_OK; }
    HIPCHK(c, hipMemcpy(buf.data(), dev, buf.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < k->m; ++a)
      for (int64_t i = 0; i < n; ++i)
        memcpy((unsigned char*)outp + ((size_t)i * k->m + a) * es, buf.data() + ((size_t)a * t->npad + i) * es, es);
    return HBO_OK;
  };
  if (kinvy_out) { int rc = export_cols(t->svec, kinvy_out, bad); if (rc) return rc; }
  if (ymu_out) { int rc = export_cols(k->resid, ymu_out, false); if (rc) return rc; }
  return HBO_OK;
}

// ---- posterior / acquisition ---------------------------------------------------------------
static int posterior(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* xq, int64_t M, int full_cov,
                     void* mu_out, void* var_out, void* acq_out, int acq_id, double param, double add_noise,
                     double scale) {
  if (!c || !xq) return fail(c, HBO_ERR_ARG, "posterior: null argument");
  if (M <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  prof_begin(c);
  int rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = m->dtype;
  if (k && (k->dtype != dtype || k->D != m->input_dim)) return fail(c, HBO_ERR_ARG, "posterior: cache/model mismatch");
  const size_t es = esize(dtype);
  const int fdim = feature_dim(m), fm = mean_feature_dim(m);
  // Candidates are STREAMED: chunks of `CH` queries, so that the cross-Gram workspace (npad x CH) does not grow with M
  // (gp.py:295-305 materialises all of Kxq; at cfg 3 that is 16384 x 65536 fp32 = 4.3 GB).  Two workspaces alternate:
  // upload + features + cross Gram of chunk i+1 run on a second stream beside the triangular product of chunk i; the
  // results of all chunks are gathered in M-sized vectors and come back in one copy.  full_cov keeps a single pass.
  const int64_t CH = full_cov ? 65536 : std::max<int64_t>(c->opt_post_chunk, HBO_TILE);
  if (full_cov && M > CH) { return fail(c, HBO_ERR_UNSUPPORTED, "posterior: full_cov limited to 65536 queries"); }
  const int64_t mc_max = std::min<int64_t>(M, CH);
  const int nbuf = (!full_cov && M > CH) ? 2 : 1;
  const int mpad_max = round_up(mc_max, HBO_TILE);
  const int64_t ldq_max = padded_ld(mpad_max, dtype);
  hipStream_t sa = c->stream, sb = nbuf == 2 ? c->stream2 : c->stream;
  char *d_xq = nullptr, *d_mu0 = nullptr, *d_kd = nullptr, *d_K = nullptr, *d_colsq = nullptr, *d_mupart = nullptr;
  void *d_mu = nullptr, *d_var = nullptr, *d_acq = nullptr, *d_V = nullptr, *d_Kqq = nullptr, *d_cov = nullptr;
  char* fq_acts[HBO_MAX_MLP_LAYERS] = {nullptr};
  size_t fq_stride[HBO_MAX_MLP_LAYERS] = {0};
#define HIPCHK_P(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); return HBO_ERR_HIP; } } while (0)
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t vec_b = al((size_t)mc_max * es);
  // the queries go up in ONE copy (M x D elements: small beside the N x CH workspace): a pageable host-to-device copy
  // inside the chunk loop waits for the products in flight on the other stream -- it serialised the two streams and
  // took cfg 3 from 142 to 197 ms
  { d_xq = (char*)ws_get(c, WS_XQ, (size_t)M * m->input_dim * es); if (!d_xq) return HBO_ERR_HIP; }
  HIPCHK_P(hipMemcpyAsync(d_xq, xq, (size_t)M * m->input_dim * es, hipMemcpyHostToDevice, sa));
  { d_mu0 = (char*)ws_get(c, WS_MU0, vec_b * nbuf); if (!d_mu0) return HBO_ERR_HIP; }
  { d_kd = (char*)ws_get(c, WS_KD, vec_b * nbuf); if (!d_kd) return HBO_ERR_HIP; }
  { d_mu = ws_get(c, WS_MU, (size_t)M * es); if (!d_mu) return HBO_ERR_HIP; }
  { d_var = ws_get(c, WS_VAR, (size_t)M * es); if (!d_var) return HBO_ERR_HIP; }
  if (acq_out) { d_acq = ws_get(c, WS_ACQ, (size_t)M * es); if (!d_acq) return HBO_ERR_HIP; }
  if (needs_mlp(m)) for (int l = 0; l < m->n_layers; ++l) {
    fq_stride[l] = al((size_t)mc_max * m->features[l] * es);
    fq_acts[l] = (char*)ws_get(c, WS_FQ0 + l, fq_stride[l] * nbuf); if (!fq_acts[l]) return HBO_ERR_HIP;
  }
  TaskHost* t = k ? k->t : nullptr;
  size_t K_b = 0, colsq_b = 0;
  if (k) {
    K_b = al((size_t)t->npad * ldq_max * es); colsq_b = al((size_t)t->nblk * ldq_max * es);
    { d_K = (char*)ws_get(c, WS_K, K_b * nbuf); if (!d_K) return HBO_ERR_HIP; }
    { d_colsq = (char*)ws_get(c, WS_COLSQ, colsq_b * nbuf); if (!d_colsq) return HBO_ERR_HIP; }
    { d_mupart = (char*)ws_get(c, WS_MUPART, colsq_b * nbuf); if (!d_mupart) return HBO_ERR_HIP; }
    if (full_cov) { d_V = ws_get(c, WS_V, (size_t)t->npad * ldq_max * es); if (!d_V) return HBO_ERR_HIP; }
  }
  if (full_cov) { { d_Kqq = ws_get(c, WS_KQQ, (size_t)M * M * es); if (!d_Kqq) return HBO_ERR_HIP; } { d_cov = ws_get(c, WS_COV, (size_t)M * M * es); if (!d_cov) return HBO_ERR_HIP; } }
  // fp32: the product runs on the bf16 matrix cores from exact three-way splits of both operands (post3.hip)
  bool use3 = k && dtype == HBO_F32 && c->opt_post_bf16x3 && !full_cov;
  unsigned short* d_K3 = nullptr; size_t k3_b = 0;
  const int nkb = k ? t->npad / 16 : 0;
  if (use3 && !k->w3_valid) {
    // the split copy of W costs 1.5 x its bytes: when the device cannot spare them the fp32-MFMA product takes over
    const size_t elems = (size_t)t->npad * t->npad * 3;
    if (!k->w3 || k->w3_elems != elems) {
      if (k->w3) hipFree(k->w3);
      k->w3 = nullptr; k->w3_elems = 0;
      if (hbo_malloc(c, (void**)&k->w3, elems * sizeof(unsigned short)) != hipSuccess) { (void)hipGetLastError(); k->w3 = nullptr; use3 = false; }
      else k->w3_elems = elems;
    }
    if (use3) {
      ProfScope ps(c, "split_w", 1, sa);
      launch_split3_rows(static_cast<const float*>(t->W), t->ld, t->nblk, k->w3, nkb, sa);
      k->w3_valid = true;
    }
  }
  if (use3) {
    k3_b = al((size_t)mpad_max * t->npad * 3 * sizeof(unsigned short));   // (mpad / 128) x nkb blocks of 3 x 128 x 16
    d_K3 = (unsigned short*)ws_get(c, WS_K3, k3_b * nbuf);
    if (!d_K3) { c->err.clear(); use3 = false; }
  }
  const bool bad = k && k->info != INT_MAX;
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
  size_t evi = 0;
  if (nbuf == 2) {   // the side stream starts behind whatever the main stream still holds (model upload)
    hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sa); hipStreamWaitEvent(sb, e, 0);
  }

  int64_t chunk = 0;
  for (int64_t q0 = 0; q0 < M; q0 += CH, ++chunk) {
    const int b = (int)(chunk % nbuf);
    const int64_t mc = std::min<int64_t>(CH, M - q0);
    const int mpad = round_up(mc, HBO_TILE);
    const int64_t ldq = padded_ld(mpad, dtype);
    char* xq_d = d_xq + (size_t)q0 * m->input_dim * es; char* mu0_d = d_mu0 + b * vec_b; char* kd_d = d_kd + b * vec_b;
    // ---- producer side (sb): inputs, features, prior mean / variance, cross Gram into workspace b ----
    if (ev_free[b]) hipStreamWaitEvent(sb, ev_free[b], 0);   // workspace b was read by the products of chunk - 2
    const void* fq_last = nullptr;
    { ProfScope ps(c, "features", 1, sb);
      if (needs_mlp(m)) {
        void* acts[HBO_MAX_MLP_LAYERS];
        for (int l = 0; l < m->n_layers; ++l) acts[l] = fq_acts[l] + b * fq_stride[l];
        const void* in = xq_d; int fin = m->input_dim;
        for (int l = 0; l < m->n_layers; ++l) { launch_dense_tanh(dtype, in, c->d_mlp_w[l], c->d_mlp_b[l], acts[l], mc, fin, m->features[l], sb); in = acts[l]; fin = m->features[l]; }
        fq_last = acts[m->n_layers - 1];
      } }
    const void* Fq = m->kernel_uses_mlp ? fq_last : xq_d;
    const void* Fmq = (m->mean_id == HBO_MEAN_LINEAR) ? (const void*)xq_d : (m->mean_id == HBO_MEAN_LINEAR_MLP ? fq_last : nullptr);
    launch_mean(dtype, Fmq, mc, fm, c->d_model, mu0_d, sb);
    launch_kdiag(dtype, Fq, mc, fdim, c->d_model, kd_d, sb);
    void* mu_d = (char*)d_mu + (size_t)q0 * es; void* var_d = (char*)d_var + (size_t)q0 * es;
    void* acq_d = d_acq ? (char*)d_acq + (size_t)q0 * es : nullptr;
    if (!k) {  // prior branch (gp.py:275-282)
      HIPCHK_P(hipMemcpyAsync(mu_d, mu0_d, (size_t)mc * es, hipMemcpyDeviceToDevice, sb));
      if (full_cov) {
        GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = Fq; g.x2 = Fq; g.out = d_cov; g.n1 = mc; g.n2 = mc; g.ldo = mc; g.fdim = fdim;
        launch_gram(dtype, g, c->d_model, dim3((unsigned)((mc + 127) / 128), (unsigned)((mc + 127) / 128), 1), sb);
      } else {
        HIPCHK_P(hipMemcpyAsync(var_d, kd_d, (size_t)mc * es, hipMemcpyDeviceToDevice, sb));
      }
      if (acq_out) {   // acquisition on the prior
        PostArgs pa = {}; pa.Kxq = nullptr; pa.n = 0; pa.nblk = 0; pa.ldq = ldq; pa.alpha = nullptr; pa.colsq = nullptr;
        pa.kdiag = kd_d; pa.muq = mu0_d; pa.acq_out = acq_d; pa.M = mc; pa.acq_id = acq_id; pa.param = param; pa.add_noise = add_noise; pa.scale = scale;
        launch_post_epilogue(dtype, pa, sb);
      }
      if (nbuf == 2) { ev_free[b] = pool_event(c, evi++); hipEventRecord(ev_free[b], sb); }
      continue;
    }
    char* K_d = d_K + b * K_b; char* colsq_d = d_colsq + b * colsq_b;
    { ProfScope ps(c, "cross_gram", 1, sb);
      GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = k->h_desc.F; g.x2 = Fq; g.out = K_d; g.n1 = t->n; g.n2 = mc; g.ldo = ldq;
      g.n1pad = t->npad; g.n2pad = mpad; g.fdim = fdim; g.symmetric = 0; g.padded = 1;
      launch_gram(dtype, g, c->d_model, dim3(mpad / HBO_TILE, t->nblk, 1), sb); }
    unsigned short* K3_d = use3 ? d_K3 + (size_t)b * (k3_b / sizeof(unsigned short)) : nullptr;
    if (use3) {
      ProfScope ps(c, "split_kxq", 1, sb);
      launch_split3_transpose(reinterpret_cast<const float*>(K_d), ldq, t->npad, mpad, K3_d, nkb, sb);
    }
    if (nbuf == 2) { ev_ready[b] = pool_event(c, evi++); hipEventRecord(ev_ready[b], sb); hipStreamWaitEvent(sa, ev_ready[b], 0); }
    // ---- consumer side (sa): V = L^-1 Kxq on MFMA (column sums of squares), then mean / variance / acquisition ----
    if (use3) {
      ProfScope ps(c, "post_gemm", 1, sa);
      Post3Args a = {}; a.Wp = k->w3; a.Kp = K3_d; a.nkb = nkb;
      a.colsq = reinterpret_cast<float*>(colsq_d); a.ldc = ldq; a.V = nullptr; a.ldv = 0; a.nblk = t->nblk;
      launch_post3(a, mpad / HBO_TILE, sa);
    } else {
      ProfScope ps(c, "post_gemm", 1, sa);
      GemmArgs a = {}; a.tasks = k->d_desc; a.mode = GEMM_POST; a.B = K_d; a.ldb = ldq; a.V = full_cov ? d_V : nullptr; a.colsq = colsq_d;
      launch_gemm(dtype, a, dim3(mpad / HBO_TILE, t->nblk, 1), sa); }
    { ProfScope ps(c, "post_epilogue", 1, sa);
      PostArgs pa = {}; pa.Kxq = K_d; pa.ldq = ldq; pa.npad = t->npad; pa.n = (int)t->n; pa.nblk = t->nblk; pa.alpha = t->svec; pa.colsq = colsq_d; pa.mupart = d_mupart + b * colsq_b;
      pa.kdiag = kd_d; pa.muq = mu0_d; pa.mu_out = mu_d; pa.var_out = var_d; pa.acq_out = acq_d; pa.M = mc;
      pa.acq_id = acq_id; pa.param = param; pa.add_noise = add_noise; pa.scale = scale;
      launch_post_epilogue(dtype, pa, sa); }
    if (nbuf == 2) { ev_free[b] = pool_event(c, evi++); hipEventRecord(ev_free[b], sa); }
    if (full_cov) {
      GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = Fq; g.x2 = Fq; g.out = d_Kqq; g.n1 = mc; g.n2 = mc; g.ldo = mc; g.fdim = fdim;
      launch_gram(dtype, g, c->d_model, dim3((unsigned)((mc + 127) / 128), (unsigned)((mc + 127) / 128), 1), sa);
      launch_fullcov(dtype, d_V, ldq, t->npad, d_Kqq, mc, d_cov, sa);
    }
  }
  if (nbuf == 2) {   // join: everything the side stream produced (the prior branch runs there entirely)
    hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sb); hipStreamWaitEvent(sa, e, 0);
  }
  if (mu_out) HIPCHK_P(hipMemcpyAsync(mu_out, d_mu, (size_t)M * es, hipMemcpyDeviceToHost, sa));
  if (var_out) {
    if (full_cov) HIPCHK_P(hipMemcpyAsync(var_out, d_cov, (size_t)M * M * es, hipMemcpyDeviceToHost, sa));
    else HIPCHK_P(hipMemcpyAsync(var_out, d_var, (size_t)M * es, hipMemcpyDeviceToHost, sa));
  }
  if (acq_out) HIPCHK_P(hipMemcpyAsync(acq_out, d_acq, (size_t)M * es, hipMemcpyDeviceToHost, sa));
  HIPCHK_P(hipStreamSynchronize(sa));
  if (nbuf == 2) HIPCHK_P(hipStreamSynchronize(sb));
  HIPCHK_P(hipGetLastError());
#undef HIPCHK_P
  prof_collect(c);
  if (bad) {
    if (mu_out) fill_nan(mu_out, (size_t)M, dtype);
    if (var_out) fill_nan(var_out, full_cov ? (size_t)M * M : (size_t)M, dtype);
    if (acq_out) fill_nan(acq_out, (size_t)M, dtype);
    return HBO_NOT_PD;
  }
  return HBO_OK;
}

extern "C" int hbo_predict(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* xq, int64_t M, int full_cov,
                           void* mu_out, void* var_out) {
  return posterior(c, m, k, xq, M, full_cov, mu_out, var_out, nullptr, 0, 0, 0, 1);
}
extern "C" int hbo_acq(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* xq, int64_t M, int acq_id,
                       double param, double add_noise, double scale, void* out) {
  if (acq_id < 0 || acq_id > HBO_ACQ_UCB) return fail(c, HBO_ERR_ARG, "hbo_acq: bad acq_id");
  if (!out) return fail(c, HBO_ERR_ARG, "hbo_acq: out is null");
  return posterior(c, m, k, xq, M, 0, nullptr, nullptr, out, acq_id, param, add_noise, scale);
}

// ---- d acquisition / d x_query: what jaxopt's L-BFGS-B differentiates in bayesopt() (bayesopt.py:116-125) ----
extern "C" int hbo_acq_grad(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* xq, int64_t M, int acq_id,
                            double param, double add_noise, double scale, void* acq_out, double* grad_out) {
  if (!c || !xq || !acq_out || !grad_out || !m) return fail(c, HBO_ERR_ARG, "hbo_acq_grad: null argument");
  if (acq_id < 0 || acq_id > HBO_ACQ_UCB) return fail(c, HBO_ERR_ARG, "hbo_acq_grad: bad acq_id");
  if (M <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = validate_model(c, m);
  if (rc) return rc;
  rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = m->dtype;
  if (k && (k->dtype != dtype || k->D != m->input_dim)) return fail(c, HBO_ERR_ARG, "hbo_acq_grad: cache/model mismatch");
  const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  const int D = m->input_dim, fdim = feature_dim(m), fm = mean_feature_dim(m);
  const bool mlp = needs_mlp(m);
  const int L = m->n_layers, flast = mlp ? m->features[L - 1] : 0;
  TaskHost* t = (k && k->t->n > 0) ? k->t : nullptr;
  const int64_t CH = 1024;   // queries per pass: three [CH][npad] panels of workspace
  const int64_t mc_max = std::min<int64_t>(M, CH);
  int maxf = D;
  for (int l = 0; l < L; ++l) maxf = std::max(maxf, (int)m->features[l]);
  size_t nparam = 1;
  { int fin0 = D; for (int l = 0; l < L; ++l) { nparam = std::max(nparam, (size_t)(fin0 + 1) * m->features[l]); fin0 = m->features[l]; } }
  void* d_xq = ws_get(c, WS_XQ, (size_t)mc_max * D * es);
  void* d_mu0 = ws_get(c, WS_MU0, (size_t)mc_max * es);
  void* d_kd = ws_get(c, WS_KD, (size_t)mc_max * es);
  void* d_acq = ws_get(c, WS_ACQ, (size_t)mc_max * es);
  double* d_gf = (double*)ws_get(c, WS_AG_GF, (size_t)mc_max * fdim * sizeof(double));
  double* d_dmu = (double*)ws_get(c, WS_AG_DMU, (size_t)mc_max * sizeof(double));
  double* d_gx = (double*)ws_get(c, WS_AG_GX, (size_t)mc_max * D * sizeof(double));
  double* d_t0 = (double*)ws_get(c, WS_AG_T0, (size_t)mc_max * maxf * sizeof(double));
  double* d_t1 = (double*)ws_get(c, WS_AG_T1, (size_t)mc_max * maxf * sizeof(double));
  double* d_dw = (double*)ws_get(c, WS_AG_DW, nparam * sizeof(double));   // weight-gradient sink of the shared MLP backward
  if (!d_xq || !d_mu0 || !d_kd || !d_acq || !d_gf || !d_dmu || !d_gx || !d_t0 || !d_t1 || !d_dw) return HBO_ERR_HIP;
  void *d_K = nullptr, *d_L = nullptr, *d_B = nullptr;
  if (t) {
    d_K = ws_get(c, WS_AG_K, (size_t)mc_max * t->npad * es); d_L = ws_get(c, WS_AG_L, (size_t)mc_max * t->npad * es);
    d_B = ws_get(c, WS_AG_B, (size_t)mc_max * t->npad * es);
    if (!d_K || !d_L || !d_B) return HBO_ERR_HIP;
  }
  void* fq_acts[HBO_MAX_MLP_LAYERS] = {nullptr};
  if (mlp) for (int l = 0; l < L; ++l) { fq_acts[l] = ws_get(c, WS_FQ0 + l, (size_t)mc_max * m->features[l] * es); if (!fq_acts[l]) return HBO_ERR_HIP; }
  const bool bad = k && k->info != INT_MAX;
#define HIPCHK_D(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); return HBO_ERR_HIP; } } while (0)
  for (int64_t q0 = 0; q0 < M; q0 += CH) {
    const int64_t mc = std::min<int64_t>(CH, M - q0);
    HIPCHK_D(hipMemcpyAsync(d_xq, (const char*)xq + (size_t)q0 * D * es, (size_t)mc * D * es, hipMemcpyHostToDevice, st));
    const void* fq_last = nullptr;
    if (mlp) { run_mlp(c, m, d_xq, mc, fq_acts); fq_last = fq_acts[L - 1]; }
    const void* Fq = m->kernel_uses_mlp ? fq_last : d_xq;
    const void* Fmq = (m->mean_id == HBO_MEAN_LINEAR) ? d_xq : (m->mean_id == HBO_MEAN_LINEAR_MLP ? fq_last : nullptr);
    launch_mean(dtype, Fmq, mc, fm, c->d_model, d_mu0, st);
    launch_kdiag(dtype, Fq, mc, fdim, c->d_model, d_kd, st);
    if (t) {
      HIPCHK_D(hipMemsetAsync(d_K, 0, (size_t)mc * t->npad * es, st));
      GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = Fq; g.x2 = k->h_desc.F; g.out = d_K; g.n1 = mc; g.n2 = t->n; g.ldo = t->npad; g.fdim = fdim;
      launch_gram(dtype, g, c->d_model, dim3((unsigned)((t->n + 127) / 128), (unsigned)((mc + 127) / 128), 1), st);
      launch_tri_matvec(dtype, t->W, t->ld, t->npad, d_K, t->npad, (int)mc, 0, d_L, t->npad, st);
      launch_tri_matvec(dtype, t->W, t->ld, t->npad, d_L, t->npad, (int)mc, 1, d_B, t->npad, st);
    }
    AcqGradArgs a = {};
    a.Fq = Fq; a.F = t ? k->h_desc.F : nullptr; a.fdim = fdim; a.n = t ? t->n : 0; a.npad = t ? t->npad : 0;
    a.Kq = d_K; a.L = d_L; a.B = d_B; a.alpha = t ? t->svec : nullptr; a.kdiag = d_kd; a.muq = d_mu0;
    a.acq_id = acq_id; a.param = param; a.add_noise = add_noise; a.scale = scale;
    a.acq_out = d_acq; a.gfeat = d_gf; a.dmu = d_dmu; a.M = mc;
    launch_acq_grad(dtype, a, c->d_model, st);
    // assemble d/dx: kernel part (direct or through the MLP) + mean part (mean.py:62-79)
    double* gmlp = nullptr;   // gradient w.r.t. the MLP output
    if (m->kernel_uses_mlp) {
      gmlp = d_gf;
      if (m->mean_id == HBO_MEAN_LINEAR_MLP) launch_acq_grad_mean(d_dmu, c->d_model, mc, flast, gmlp, 1, st);
      HIPCHK_D(hipMemsetAsync(d_gx, 0, (size_t)mc * D * sizeof(double), st));
      if (m->mean_id == HBO_MEAN_LINEAR) launch_acq_grad_mean(d_dmu, c->d_model, mc, D, d_gx, 1, st);
    } else {
      HIPCHK_D(hipMemcpyAsync(d_gx, d_gf, (size_t)mc * D * sizeof(double), hipMemcpyDeviceToDevice, st));
      if (m->mean_id == HBO_MEAN_LINEAR) launch_acq_grad_mean(d_dmu, c->d_model, mc, D, d_gx, 1, st);
      if (m->mean_id == HBO_MEAN_LINEAR_MLP) { gmlp = d_t0; launch_acq_grad_mean(d_dmu, c->d_model, mc, flast, gmlp, 0, st); }
    }
    if (gmlp) {
      double* cur = gmlp; double* other = (gmlp == d_t0) ? d_t1 : d_t0;
      for (int l = L - 1; l >= 0; --l) {
        const int fin = l ? m->features[l - 1] : D;
        const void* in = l ? fq_acts[l - 1] : d_xq;
        launch_dense_bwd(dtype, in, fq_acts[l], c->d_mlp_w[l], cur, other, d_dw, d_dw + (size_t)fin * m->features[l], mc, fin, m->features[l], st);
        cur = other; other = (cur == d_t0) ? d_t1 : d_t0;
      }
      launch_add_inplace(d_gx, cur, mc * D, st);
    }
    HIPCHK_D(hipMemcpyAsync((char*)acq_out + (size_t)q0 * es, d_acq, (size_t)mc * es, hipMemcpyDeviceToHost, st));
    HIPCHK_D(hipMemcpyAsync(grad_out + (size_t)q0 * D, d_gx, (size_t)mc * D * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK_D(hipStreamSynchronize(st));
  }
  HIPCHK_D(hipGetLastError());
#undef HIPCHK_D
  if (bad) {
    fill_nan(acq_out, (size_t)M, dtype);
    for (int64_t i = 0; i < M * D; ++i) grad_out[i] = NAN;
    return HBO_NOT_PD;
  }
  return HBO_OK;
}

// ---- Gram / mean on host arrays --------------------------------------------------------------
extern "C" int hbo_gram(hbo_ctx* c, const hbo_model* m, const void* x1, int64_t n1, const void* x2, int64_t n2,
                        int diag, void* out) {
  if (!c || !x1 || !out) return fail(c, HBO_ERR_ARG, "hbo_gram: null argument");
  if (diag && x2) return fail(c, HBO_ERR_ARG, "hbo_gram: diag requires x2 == NULL (kernel.py:54-57)");
  if (n1 <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = m->dtype; const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  if (!x2) n2 = n1;
  if (n2 <= 0) return HBO_OK;
  void *d1 = nullptr, *d2 = nullptr, *dout = nullptr;
  FeatBuf f1, f2;
  auto cleanup = [&]() { for (void* p : {d1, d2, dout}) if (p) hipFree(p); };
#define HIPCHK_G(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); cleanup(); return HBO_ERR_HIP; } } while (0)
  HIPCHK_G(hbo_malloc(c, &d1, (size_t)n1 * m->input_dim * es));
  HIPCHK_G(hipMemcpyAsync(d1, x1, (size_t)n1 * m->input_dim * es, hipMemcpyHostToDevice, st));
  const void* F1 = d1; const void* F2 = d1;
  if (m->kernel_uses_mlp) { rc = f1.ensure(c, m, n1); if (rc) { cleanup(); return rc; } run_mlp(c, m, d1, n1, f1.acts.data()); F1 = F2 = f1.acts[m->n_layers - 1]; }
  if (x2) {
    HIPCHK_G(hbo_malloc(c, &d2, (size_t)n2 * m->input_dim * es));
    HIPCHK_G(hipMemcpyAsync(d2, x2, (size_t)n2 * m->input_dim * es, hipMemcpyHostToDevice, st));
    F2 = d2;
    if (m->kernel_uses_mlp) { rc = f2.ensure(c, m, n2); if (rc) { cleanup(); return rc; } run_mlp(c, m, d2, n2, f2.acts.data()); F2 = f2.acts[m->n_layers - 1]; }
  }
  const int fdim = feature_dim(m);
  if (diag) {
    HIPCHK_G(hbo_malloc(c, &dout, (size_t)n1 * es));
    launch_kdiag(dtype, F1, n1, fdim, c->d_model, dout, st);
    HIPCHK_G(hipMemcpyAsync(out, dout, (size_t)n1 * es, hipMemcpyDeviceToHost, st));
  } else {
    HIPCHK_G(hbo_malloc(c, &dout, (size_t)n1 * n2 * es));
    GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.x1 = F1; g.x2 = F2; g.out = dout; g.n1 = n1; g.n2 = n2; g.ldo = n2; g.fdim = fdim;
    launch_gram(dtype, g, c->d_model, dim3((unsigned)((n2 + 127) / 128), (unsigned)((n1 + 127) / 128), 1), st);
    HIPCHK_G(hipMemcpyAsync(out, dout, (size_t)n1 * n2 * es, hipMemcpyDeviceToHost, st));
  }
  HIPCHK_G(hipStreamSynchronize(st));
  HIPCHK_G(hipGetLastError());
#undef HIPCHK_G
  cleanup();
  return HBO_OK;
}

extern "C" int hbo_mean(hbo_ctx* c, const hbo_model* m, const void* x, int64_t n, void* out) {
  if (!c || !x || !out) return fail(c, HBO_ERR_ARG, "hbo_mean: null argument");
  if (n <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = upload_model(c, m);
  if (rc) return rc;
  const int dtype = m->dtype; const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  void *dx = nullptr, *dout = nullptr;
  FeatBuf f;
  auto cleanup = [&]() { for (void* p : {dx, dout}) if (p) hipFree(p); };
  hipError_t e = hbo_malloc(c, &dx, (size_t)n * m->input_dim * es);
  if (e == hipSuccess) e = hbo_malloc(c, &dout, (size_t)n * es);
  if (e == hipSuccess) e = hipMemcpyAsync(dx, x, (size_t)n * m->input_dim * es, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) { cleanup(); return fail(c, HBO_ERR_HIP, hipGetErrorString(e)); }
  const void* Fm = (m->mean_id == HBO_MEAN_LINEAR) ? dx : nullptr;
  if (m->mean_id == HBO_MEAN_LINEAR_MLP) { rc = f.ensure(c, m, n); if (rc) { cleanup(); return rc; } run_mlp(c, m, dx, n, f.acts.data()); Fm = f.acts[m->n_layers - 1]; }
  launch_mean(dtype, Fm, n, mean_feature_dim(m), c->d_model, dout, st);
  e = hipMemcpyAsync(out, dout, (size_t)n * es, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipGetLastError();
  cleanup();
  if (e != hipSuccess) return fail(c, HBO_ERR_HIP, hipGetErrorString(e));
  return HBO_OK;
}

// ---- dense SPD building block ---------------------------------------------------------------
extern "C" int hbo_spd_solve(hbo_ctx* c, int dtype, const void* a, int64_t n, const void* b, int32_t mcols,
                             void* chol_out, void* inv_out, void* x_out, double* logdet_half) {
  if (!c || !a) return fail(c, HBO_ERR_ARG, "hbo_spd_solve: null argument");
  if (n <= 0) return fail(c, HBO_ERR_ARG, "hbo_spd_solve: n must be positive");
  if (b && (mcols <= 0 || mcols > HBO_TILE)) return fail(c, HBO_ERR_ARG, "hbo_spd_solve: 1 <= m <= 128");
  if (dtype != HBO_F32 && dtype != HBO_F64) return fail(c, HBO_ERR_ARG, "hbo_spd_solve: bad dtype");
  HIPCHK(c, hipSetDevice(c->device));
  prof_begin(c);
  const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  TaskHost* t = new TaskHost();
  t->n = n; t->m = b ? mcols : 1; t->npad = round_up(n, HBO_TILE); t->nblk = t->npad / HBO_TILE; t->ld = padded_ld(t->npad, dtype);
  void *d_a = nullptr, *d_b = nullptr, *d_tmp = nullptr; TaskDesc* d_desc = nullptr; int* d_info = nullptr;
  auto cleanup = [&]() { free_task(c, t); for (void* p : {d_a, d_b, d_tmp, (void*)d_desc, (void*)d_info}) if (p) hipFree(p); };
#define HIPCHK_S(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); cleanup(); return HBO_ERR_HIP; } } while (0)
  const bool need_inv = inv_out != nullptr || x_out != nullptr;
  { int rc = ensure_task_workspace(c, dtype, t, need_inv, t->m); if (rc) { cleanup(); return rc; } }
  HIPCHK_S(hbo_malloc(c, &d_a, (size_t)n * n * es));
  HIPCHK_S(hipMemcpyAsync(d_a, a, (size_t)n * n * es, hipMemcpyHostToDevice, st));
  if (b) { HIPCHK_S(hbo_malloc(c, &d_b, (size_t)n * mcols * es)); HIPCHK_S(hipMemcpyAsync(d_b, b, (size_t)n * mcols * es, hipMemcpyHostToDevice, st)); }
  launch_fill_spd(dtype, d_a, n, t->A, t->ld, t->npad, st);
  launch_set_aug(dtype, d_b, n, b ? mcols : 0, t->A, t->ld, t->npad, st);
  TaskDesc h; memset(&h, 0, sizeof h);
  h.A = t->A; h.W = t->W; h.S = t->S; h.wscr = t->wscr; h.svec = t->svec; h.n = (int)n; h.npad = t->npad; h.nblk = t->nblk; h.m = t->m; h.ld = t->ld;
  h.naug = t->m;
  HIPCHK_S(hbo_malloc(c, (void**)&d_desc, sizeof h));
  HIPCHK_S(hbo_malloc(c, (void**)&d_info, sizeof(int)));
  int inf = INT_MAX;
  HIPCHK_S(hipMemcpyAsync(d_desc, &h, sizeof h, hipMemcpyHostToDevice, st));
  HIPCHK_S(hipMemcpyAsync(d_info, &inf, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK_S(hipStreamSynchronize(st));
  c->trtri_host_task = h;
  { ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, d_desc, 1, t->nblk, d_info); }
  if (need_inv) {
    { ProfScope ps(c, "trtri", 1); run_trtri(c, dtype, d_desc, 1, t->nblk); }
    if (x_out && b) for (int col = 0; col < mcols; ++col) launch_wt_z(dtype, d_desc, 1, t->nblk, col, col, t->npad, st);
    if (inv_out) { ProfScope ps(c, "lauum", 1); run_lauum(c, dtype, d_desc, 1, t->nblk); }
  }
  HIPCHK_S(hipMemcpyAsync(&inf, d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK_S(hbo_malloc(c, &d_tmp, (size_t)n * n * es));
  std::vector<unsigned char> hchol;
  if (chol_out || logdet_half) {
    launch_extract_lower(dtype, t->A, t->ld, n, d_tmp, st);
    void* dst = chol_out;
    if (!dst) { hchol.resize((size_t)n * n * es); dst = hchol.data(); }
    HIPCHK_S(hipMemcpyAsync(dst, d_tmp, (size_t)n * n * es, hipMemcpyDeviceToHost, st));
    HIPCHK_S(hipStreamSynchronize(st));
    if (logdet_half) { double s = 0; for (int64_t i = 0; i < n; ++i) s += log(host_elem(dst, dtype, i * n + i)); *logdet_half = s; }
  }
  if (inv_out) {
    launch_symmetrize_from_lower(dtype, t->S, t->ld, n, d_tmp, st);
    HIPCHK_S(hipMemcpyAsync(inv_out, d_tmp, (size_t)n * n * es, hipMemcpyDeviceToHost, st));
  }
  HIPCHK_S(hipStreamSynchronize(st));
  if (x_out && b) {
    std::vector<unsigned char> buf((size_t)mcols * t->npad * es);
    HIPCHK_S(hipMemcpy(buf.data(), t->svec, buf.size(), hipMemcpyDeviceToHost));
    for (int col = 0; col < mcols; ++col)
      for (int64_t i = 0; i < n; ++i)
        memcpy((unsigned char*)x_out + ((size_t)i * mcols + col) * es, buf.data() + ((size_t)col * t->npad + i) * es, es);
  }
  HIPCHK_S(hipGetLastError());
#undef HIPCHK_S
  prof_collect(c);
  const bool bad = inf != INT_MAX;
  if (bad) {
    if (chol_out) fill_nan(chol_out, (size_t)n * n, dtype);
    if (inv_out) fill_nan(inv_out, (size_t)n * n, dtype);
    if (x_out && b) fill_nan(x_out, (size_t)n * mcols, dtype);
    if (logdet_half) *logdet_half = NAN;
  }
  cleanup();
  return bad ? HBO_NOT_PD : HBO_OK;
}
