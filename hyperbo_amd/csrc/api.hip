// C ABI of libhbo (see include/hbo.h): context, device memory, orchestration of the HIP kernels
// for the GP hot path, profiling with HIP events, and the RCCL all-reduce used by task sharding.
#include "api_internal.h"

thread_local std::string hbo_g_err;

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
  if (e == hipSuccess) e = hipStreamCreate(&c->stream3);
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
      c->lds_per_block = (int)std::min<size_t>(prop.sharedMemPerBlock, (size_t)1 << 30);
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
  for (hipEvent_t ev : c->ev_pool_sweep) hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_timed) if (ev) hipEventDestroy(ev);
  if (c->stream4) hipStreamDestroy(c->stream4);
  if (c->stream3) hipStreamDestroy(c->stream3);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return HBO_OK;
}
// The six options of the boundary (include/hbo.h).
extern "C" int hbo_set_option(hbo_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return HBO_ERR_ARG;
  if (!strcmp(name, "potrf_group")) { if (value < 0 || value > 16) return fail(c, HBO_ERR_ARG, "potrf_group in 0..16 (0: auto)"); c->opt_group = (int)value; return HBO_OK; }
  if (!strcmp(name, "lookahead")) { if (value < 0 || value > 2) return fail(c, HBO_ERR_ARG, "lookahead in 0..2"); c->opt_lookahead = (int)value; return HBO_OK; }
  if (!strcmp(name, "small_nblk")) { c->opt_small_nblk = (int)value; return HBO_OK; }
  if (!strcmp(name, "pool_cap_mb")) {
    if (value < 0) return fail(c, HBO_ERR_ARG, "pool_cap_mb >= 0");
    c->pool_cap = (size_t)value << 20;
    if (c->pool_bytes > c->pool_cap) {   // trim: release everything parked (simple and rare)
      hipSetDevice(c->device);
      pool_release_all(c);
    }
    return HBO_OK;
  }
  if (!strcmp(name, "post_chunk")) { if (value < 128 || value > 65536) return fail(c, HBO_ERR_ARG, "post_chunk in 128..65536"); c->opt_post_chunk = (int)value; return HBO_OK; }
  if (!strcmp(name, "bf16x3")) { c->opt_post_bf16x3 = c->opt_syrk_bf16x3 = c->opt_trtri_bf16x3 = c->opt_lauum_bf16x3 = value != 0; return HBO_OK; }
  return fail(c, HBO_ERR_ARG, std::string("unknown option ") + name);
}
// Measurement hooks (include/hbo_tune.h): placement and overlap knobs of the schedules, for the A/B tools under tools/ and
// the scheduling sweep of the tests.  Every non-default value was measured equal or worse; none changes a result.
extern "C" int hbo_tune(hbo_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return HBO_ERR_ARG;
  struct Knob { const char* name; int hbo_ctx::*field; int64_t lo, hi; };
  static const Knob knobs[] = {
      {"overlap_trtri", &hbo_ctx::opt_overlap_trtri, 0, 1}, {"cu_yield", &hbo_ctx::opt_cu_yield, 0, 2},
      {"persist_free", &hbo_ctx::opt_persist_free, -1, 200}, {"trtri_at", &hbo_ctx::opt_trtri_at, 0, 63},
      {"trtri_free", &hbo_ctx::opt_trtri_free, 0, 200}, {"lauum_persist", &hbo_ctx::opt_lauum_persist, 0, 128}, {"split_f1", &hbo_ctx::opt_split_f1, 0, 2}, {"f2_split", &hbo_ctx::opt_f2_split, 0, 2}, {"poison", &hbo_ctx::opt_poison, 0, 1},
      {"sweep", &hbo_ctx::opt_sweep, 0, 2}, {"sweep_qs", &hbo_ctx::opt_sweep_qs, 0, 16}, {"sweep_side", &hbo_ctx::opt_sweep_side, 0, 1}, {"sweep_free", &hbo_ctx::opt_sweep_free, 1, 200}, {"sweep_big", &hbo_ctx::opt_sweep_big, 0, 1 << 30}, {"batch_bg", &hbo_ctx::opt_batch_bg, -1, 2},
      {"post_bf16x3", &hbo_ctx::opt_post_bf16x3, 0, 1}, {"post_f16x2", &hbo_ctx::opt_post_f16x2, 0, 1}, {"chol_f16x2", &hbo_ctx::opt_chol_f16x2, 0, 1}, {"group_inner", &hbo_ctx::opt_group_inner, -1, 16}, {"syrk_bf16x3", &hbo_ctx::opt_syrk_bf16x3, 0, 1},
      {"trtri_bf16x3", &hbo_ctx::opt_trtri_bf16x3, 0, 1}, {"lauum_bf16x3", &hbo_ctx::opt_lauum_bf16x3, 0, 1}, {"trtri3_min_s", &hbo_ctx::opt_trtri3_min_s, 1, 1024},
      {"fault_shard", &hbo_ctx::opt_fault_shard, 0, 2}, {"small_fused", &hbo_ctx::opt_small_fused, 0, 1}, {"post_serial", &hbo_ctx::opt_post_serial, 0, 1},
      {"syrk3_col", &hbo_ctx::opt_syrk3_col, 0, 1}, {"syrk3_sep", &hbo_ctx::opt_syrk3_sep, 0, 1}, {"syrk3_free", &hbo_ctx::opt_syrk3_free, 0, 200},
  };
  if (!strcmp(name, "gram_mfma")) {   // process-wide: fp32 Gram matrices with at least `value` features on the matrix cores (gram.hip: gram_mfma_kernel); 0: never
    if (value < 0 || value > 4096) return fail(c, HBO_ERR_ARG, "gram_mfma in 0..4096");
    gram_set_mfma_min_features((int)value);
    return HBO_OK;
  }
  for (const Knob& k : knobs)
    if (!strcmp(name, k.name)) {
      if (value < k.lo || value > k.hi) return fail(c, HBO_ERR_ARG, std::string(name) + " out of range");
      c->*(k.field) = (int)value;
      return HBO_OK;
    }
  return hbo_set_option(c, name, value);   // (the tools pass every name through one entry point)
}
// Sustained fp64 MFMA rate of THIS device, measured now (include/hbo_tune.h): every SIMD of every CU runs two waves of
// back-to-back independent v_mfma_f64_16x16x4_f64 (4 accumulators per wave, no memory traffic) for ~`ms` milliseconds between two
// HIP events.  What bench.py reports beside the 78.6 TFLOP/s datasheet figure: the denominator a kernel can actually reach at the
// clock the chip holds under an fp64 MFMA load.  (Four accumulators per wave, the loop in assembly: see the kernel.)
namespace {
typedef double probe_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_probe_kernel(double* out, int iters) {
  probe_d4 acc[4];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (probe_d4){0, 0, 0, 0};
  // the whole loop is ONE assembly block: around a builtin (or a per-instruction asm) inside a C loop the register allocator parks
  // the loop-carried accumulators in VGPRs and copies them to AGPRs and back every iteration (16 v_accvgpr moves per MFMA)
  int n = iters;
  asm volatile(
      "1:\n"
      "v_mfma_f64_16x16x4_f64 %0, %5, %6, %0\n"
      "v_mfma_f64_16x16x4_f64 %1, %5, %6, %1\n"
      "v_mfma_f64_16x16x4_f64 %2, %5, %6, %2\n"
      "v_mfma_f64_16x16x4_f64 %3, %5, %6, %3\n"
      "s_sub_u32 %4, %4, 1\n"
      "s_cmp_lg_u32 %4, 0\n"
      "s_cbranch_scc1 1b\n"
      "s_nop 15\n"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+s"(n)
      : "v"(a), "v"(b)
      : "scc");
  double sacc = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}
}  // namespace
extern "C" int hbo_mfma_peak_probe(hbo_ctx* c, double ms, double* tflops_out) {
  if (!c || !tflops_out || !(ms > 0) || ms > 1000) return fail(c, HBO_ERR_ARG, "hbo_mfma_peak_probe: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  const int blocks = 2 * c->n_cus;   // two 4-wave workgroups per CU: two waves per SIMD
  double* d_out = static_cast<double*>(ws_get(c, WS_COUNTERS + 1000, sizeof(double) * 256 * blocks));
  if (!d_out) return HBO_ERR_HIP;
  hipEvent_t e0 = pool_event_timed(c, 0), e1 = pool_event_timed(c, 1);
  double best = 0;
  int iters = 8192;
  for (int rep = 0; rep < 3; ++rep) {   // the first pass sizes the second and third to ~ms
    HIPCHK(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, c->stream, d_out, iters);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float el = 0;
    HIPCHK(c, hipEventElapsedTime(&el, e0, e1));
    const double flops = (double)blocks * 4 * (double)iters * 4 * 2048.0;   // waves x MFMAs x 16*16*4*2
    if (rep > 0) best = std::max(best, flops / (el * 1e-3) / 1e12);
    if (rep == 0) iters = (int)std::min(4.0e6, std::max(1024.0, iters * ms / std::max((double)el, 1e-3)));
  }
  HIPCHK(c, hipGetLastError());
  *tflops_out = best;
  return HBO_OK;
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

extern "C" int hbo_dataset_free(hbo_ctx* c, hbo_dataset* ds) {
  if (!ds) return HBO_OK;
  if (c) hipSetDevice(c->device);
  // (the buffers go back to the pool while kernels of the last evaluation may still run only if a call returned without
  //  draining its streams -- none does: every entry point synchronises before it returns)
  for (TaskHost* t : ds->tasks) free_task(c, t);
  dev_free(c, ds->d_inputs);
  dev_free(c, ds->d_svec);
  for (void* p : {(void*)ds->d_desc, (void*)ds->d_pack, (void*)ds->d_partials, (void*)ds->d_mlpgrad, (void*)ds->d_mlp}) dev_free(c, p);
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
    total += al((size_t)(tk.m + 1) * tk.n * es);
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
    {
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

namespace {
struct GatherTask { const void* sx; const void* sys; const void* syd; void* dx; void* dys; void* dyd; int64_t n_src, n_dst, idx_off; int m, has_idx; };
// row r of task blockIdx.y of the sub-sample = row idx[idx_off + r] (or r) of the resident task: inputs, column sum of y, divergence rows
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherTask* __restrict__ tasks, const int32_t* __restrict__ idx, int D) {
  const GatherTask t = tasks[blockIdx.y];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= t.n_dst) return;
  const int64_t s = t.has_idx ? (int64_t)idx[t.idx_off + r] : r;
  const T* sx = static_cast<const T*>(t.sx) + s * D; T* dx = static_cast<T*>(t.dx) + r * D;
  for (int d = 0; d < D; ++d) dx[d] = sx[d];
  static_cast<T*>(t.dys)[r] = static_cast<const T*>(t.sys)[s];
  if (t.syd) for (int a = 0; a <= t.m; ++a) static_cast<T*>(t.dyd)[(int64_t)a * t.n_dst + r] = static_cast<const T*>(t.syd)[(int64_t)a * t.n_src + s];
}
}  // namespace

extern "C" int hbo_dataset_subsample(hbo_ctx* c, const hbo_dataset* src, const int64_t* counts, const int32_t* idx, hbo_dataset** out) {
  if (!c || !src || !counts || !out) return fail(c, HBO_ERR_ARG, "hbo_dataset_subsample: null argument");
  HIPCHK(c, hipSetDevice(c->device));
  const int T = src->ntasks, dtype = src->dtype, D = src->D;
  const size_t es = esize(dtype);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t total = 0; int64_t nidx = 0, max_dst = 0;
  for (int k = 0; k < T; ++k) {
    const TaskHost* t = src->tasks[k];
    const int64_t nd = counts[k] < 0 ? t->n : counts[k];
    if (nd <= 0 || nd > t->n) return fail(c, HBO_ERR_ARG, "hbo_dataset_subsample: counts[k] must be in 1..n_k (or negative: the whole task)");
    if (counts[k] >= 0) { if (!idx) return fail(c, HBO_ERR_ARG, "hbo_dataset_subsample: idx is null"); nidx += nd; }
    total += al((size_t)nd * D * es) + al((size_t)nd * es) + (t->ydiv ? al((size_t)(t->m + 1) * nd * es) : 0);
    max_dst = std::max(max_dst, nd);
  }
  hbo_dataset* ds = new hbo_dataset();
  ds->dtype = dtype; ds->D = D;
  if (T == 0) { *out = ds; return HBO_OK; }
  // descriptors + indices through the pinned staging buffer, one copy
  const size_t desc_b = al(sizeof(GatherTask) * T), idx_b = al(sizeof(int32_t) * (size_t)std::max<int64_t>(nidx, 1));
  HIPCHK(c, hipEventSynchronize(c->ev_upload));
  unsigned char* stage = static_cast<unsigned char*>(pinned_stage(c, desc_b + idx_b));
  unsigned char* d_args = static_cast<unsigned char*>(ws_get(c, WS_GATHER, desc_b + idx_b));
  hipError_t e = (stage && d_args) ? dev_alloc(c, &ds->d_inputs, total) : hipErrorOutOfMemory;
  if (e != hipSuccess) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_HIP, std::string("hbo_dataset_subsample: ") + hipGetErrorString(e)); }
  GatherTask* g = reinterpret_cast<GatherTask*>(stage);
  size_t off = 0; int64_t ioff = 0;
  for (int k = 0; k < T; ++k) {
    const TaskHost* s = src->tasks[k];
    const int64_t nd = counts[k] < 0 ? s->n : counts[k];
    TaskHost* t = new TaskHost();
    ds->tasks.push_back(t);
    t->owns_inputs = false;
    t->n = nd; t->m = s->m; t->npad = round_up(nd, HBO_TILE); t->nblk = t->npad / HBO_TILE; t->ld = padded_ld(t->npad, dtype);
    t->X = (char*)ds->d_inputs + off; off += al((size_t)nd * D * es);
    t->ysum = (char*)ds->d_inputs + off; off += al((size_t)nd * es);
    if (s->ydiv) { t->ydiv = (char*)ds->d_inputs + off; off += al((size_t)(s->m + 1) * nd * es); }
    g[k] = GatherTask{s->X, s->ysum, s->ydiv, t->X, t->ysum, t->ydiv, s->n, nd, ioff, s->m, counts[k] >= 0 ? 1 : 0};
    if (counts[k] >= 0) {
      for (int64_t r = 0; r < nd; ++r) if (idx[ioff + r] < 0 || idx[ioff + r] >= s->n) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_ARG, "hbo_dataset_subsample: index out of range"); }
      ioff += nd;
    }
    ds->max_nblk = std::max(ds->max_nblk, t->nblk);
  }
  if (nidx) memcpy(stage + desc_b, idx, sizeof(int32_t) * (size_t)nidx);
  hipStream_t st = c->stream;
  e = hipMemcpyAsync(d_args, stage, desc_b + idx_b, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipEventRecord(c->ev_upload, st);
  if (e == hipSuccess) {
    const dim3 grid((unsigned)((max_dst + 255) / 256), (unsigned)T);
    if (dtype == HBO_F64) hipLaunchKernelGGL(gather_rows_kernel<double>, grid, dim3(256), 0, st, reinterpret_cast<const GatherTask*>(d_args), reinterpret_cast<const int32_t*>(d_args + desc_b), D);
    else hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const GatherTask*>(d_args), reinterpret_cast<const int32_t*>(d_args + desc_b), D);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { hbo_dataset_free(c, ds); return fail(c, HBO_ERR_HIP, std::string("hbo_dataset_subsample: ") + hipGetErrorString(e)); }
  ds->ntasks = (int)ds->tasks.size();
  std::stable_sort(ds->tasks.begin(), ds->tasks.end(), [](TaskHost* a, TaskHost* b) { return a->n > b->n; });
  *out = ds;
  return HBO_OK;
}

extern "C" int hbo_cache_free(hbo_ctx* c, hbo_cache* k) {
  if (!k) return HBO_OK;
  if (c) hipSetDevice(c->device);
  free_task(c, k->t);
  for (void* p : {(void*)k->d_desc, (void*)k->d_info, k->resid, k->zvec, (void*)k->w3, (void*)k->d_wmax}) if (p) hipFree(p);
  delete k;
  return HBO_OK;
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
// x = L^-T L^-1 b with a factor the caller already holds (linalg.py:139-145, the `cached_cholesky=` branch): the factor and the
// right-hand sides go up, two substitution sweeps run on the device (trisolve.hip), the solution comes back.
void launch_chol_solve(int dtype, const void* L, int64_t n, void* rhs, int64_t npad, int m, hipStream_t st);   // trisolve.hip
extern "C" int hbo_chol_solve(hbo_ctx* c, int dtype, const void* chol_lower, int64_t n, const void* b, int32_t mcols, void* x_out) {
  if (!c || !chol_lower || !b || !x_out) return fail(c, HBO_ERR_ARG, "hbo_chol_solve: null argument");
  if (n <= 0 || mcols <= 0) return fail(c, HBO_ERR_ARG, "hbo_chol_solve: n and m must be positive");
  if (dtype != HBO_F32 && dtype != HBO_F64) return fail(c, HBO_ERR_ARG, "hbo_chol_solve: bad dtype");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  const int64_t npad = round_up(n, 64);
  void *d_l = nullptr, *d_r = nullptr;
  auto cleanup = [&]() { for (void* p : {d_l, d_r}) if (p) hipFree(p); };
#define HIPCHK_S(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); cleanup(); return HBO_ERR_HIP; } } while (0)
  // right-hand sides as rows (m x npad, zero padding)
  std::vector<unsigned char> rt((size_t)mcols * npad * es, 0);
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < mcols; ++a) memcpy(rt.data() + ((size_t)a * npad + i) * es, (const unsigned char*)b + ((size_t)i * mcols + a) * es, es);
  HIPCHK_S(hbo_malloc(c, &d_l, (size_t)n * n * es));
  HIPCHK_S(hbo_malloc(c, &d_r, rt.size()));
  HIPCHK_S(hipMemcpyAsync(d_l, chol_lower, (size_t)n * n * es, hipMemcpyHostToDevice, st));
  HIPCHK_S(hipMemcpyAsync(d_r, rt.data(), rt.size(), hipMemcpyHostToDevice, st));
  launch_chol_solve(dtype, d_l, n, d_r, npad, mcols, st);
  HIPCHK_S(hipMemcpyAsync(rt.data(), d_r, rt.size(), hipMemcpyDeviceToHost, st));
  HIPCHK_S(hipStreamSynchronize(st));
  HIPCHK_S(hipGetLastError());
#undef HIPCHK_S
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < mcols; ++a) memcpy((unsigned char*)x_out + ((size_t)i * mcols + a) * es, rt.data() + ((size_t)a * npad + i) * es, es);
  cleanup();
  return HBO_OK;
}

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
