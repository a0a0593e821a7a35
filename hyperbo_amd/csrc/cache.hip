// GPCache side of the C ABI: hbo_factor (hyperbo/basics/linalg.py:72-110 behind gp.py:540-560), the O(N^2) row append,
// and everything that reads a cache -- hbo_predict (gp.py:242-305), hbo_acq (acfun.py:36-142), hbo_acq_grad (bayesopt.py:116-125).
#include "api_internal.h"

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

  if (c->opt_poison) launch_poison(dtype, k->d_desc, 1, t->npad, st);
  { ProfScope ps(c, "features", 1);
    if (needs_mlp(m)) run_mlp(c, m, t->X, n, t->feat.acts.data());
    launch_aug_rows(dtype, k->d_desc, 1, t->npad, c->d_model, st); }
  HIPCHK_K(hipMemcpy2DAsync(k->resid, (size_t)t->npad * es, (char*)t->A + (size_t)t->npad * t->ld * es, (size_t)t->ld * es, (size_t)t->npad * es, mcols, hipMemcpyDeviceToDevice, st));
  { ProfScope ps(c, "gram", 1);
    GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.tasks = k->d_desc; g.fdim = feature_dim(m); g.symmetric = 1; g.padded = 1;
    launch_gram(dtype, g, c->d_model, dim3(t->nblk, t->nblk, 1), st); }
  // the inverse W = L^-1 (kept for the posterior products) starts beside the panel chain, as in the objective path
  TrtriProgress trtri_pg;
  const bool early_trtri = use_lookahead(c, 1, t->nblk) && c->opt_overlap_trtri && t->nblk >= 4;
  c->trtri_host_task = k->h_desc;
  CholBoundScope bound_scope(c, chol_diag_bound_of(m));
  { ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, k->d_desc, 1, t->nblk, k->d_info, early_trtri ? &trtri_pg : nullptr); }
  HIPCHK_K(hipMemcpy2DAsync(k->zvec, (size_t)t->npad * es, (char*)t->A + (size_t)t->npad * t->ld * es, (size_t)t->ld * es, (size_t)t->npad * es, mcols, hipMemcpyDeviceToDevice, st));
  { ProfScope ps(c, "trtri", 1); run_trtri(c, dtype, k->d_desc, 1, t->nblk, &trtri_pg); }
  { ProfScope ps(c, "wt_z", 1);
    for (int a = 0; a < mcols; ++a) launch_wt_z(dtype, k->d_desc, 1, t->nblk, a, a, t->npad, st); }
  HIPCHK_K(hipMemcpyAsync(&k->info, k->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK_K(hipStreamSynchronize(st));
  HIPCHK_K(hipGetLastError());
  prof_collect(c);
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
namespace {
// One new observation joins a cached factorisation (GP.update_sub_dataset(is_append=True) + setup_predictor, gp.py:426-452,540-560),
// entirely on the device: with l = W k(X, x*) and wl = W^T l from the two triangular mat-vecs before it, ONE workgroup
//   d = sqrt(k(x*, x*) + noise + eps - l.l)        (not positive: *fail_at = n + 1, nothing is written)
//   row n of L = [l, d],   row n of W = [-wl / d, 1 / d]
//   per column a of y:  z_a[n] = (y*_a - mu(x*) - l.z_a) / d,   alpha_a += W[n, :] z_a[n],   resid_a[n] = y*_a - mu(x*)
// Sums in fp64 for both dtypes (as the host loop this replaces did).
template <typename T>
__global__ __launch_bounds__(1024) void append_row_kernel(T* L, T* W, int64_t ld, int64_t n, const T* l, const T* wl, const T* mu_new,
                                                          const T* kdiag, double kadd, const T* y_new, int mc, int64_t npad, T* z,
                                                          T* alpha, T* resid, int* fail_at) {
  __shared__ double red[16];
  __shared__ double bc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto block_sum = [&](double v) -> double {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();               // red[] / bc of the previous call are no longer read
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (tid == 0) { double s = 0; for (int k = 0; k < 16; ++k) s += red[k]; bc = s; }
    __syncthreads();
    return bc;
  };
  if (*fail_at) return;            // an earlier row of this call failed (uniform)
  double s = 0;
  for (int64_t i = tid; i < n; i += 1024) { const double v = (double)l[i]; s += v * v; }
  const double d2 = (double)kdiag[0] + kadd - block_sum(s);
  if (!(d2 > 0)) { if (tid == 0) *fail_at = (int)n + 1; return; }
  const double d = sqrt(d2);
  for (int64_t i = tid; i < n; i += 1024) { L[n * ld + i] = l[i]; W[n * ld + i] = (T)(-(double)wl[i] / d); }
  if (tid == 0) { L[n * ld + n] = (T)d; W[n * ld + n] = (T)(1.0 / d); }
  for (int a = 0; a < mc; ++a) {
    T* za = z + (int64_t)a * npad; T* aa = alpha + (int64_t)a * npad;
    double lz = 0;
    for (int64_t i = tid; i < n; i += 1024) lz += (double)l[i] * (double)za[i];
    lz = block_sum(lz);
    const double r_new = (double)y_new[a] - (double)mu_new[0];
    const double zn = (r_new - lz) / d;
    for (int64_t i = tid; i < n; i += 1024) aa[i] = (T)((double)aa[i] + (-(double)wl[i] / d) * zn);
    if (tid == 0) { za[n] = (T)zn; aa[n] = (T)(zn / d); resid[(int64_t)a * npad + n] = (T)r_new; }
  }
}
}  // namespace

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
#define HIPCHK_A(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); return HBO_ERR_HIP; } } while (0)
  void* d_kx = ws_get(c, WS_AP_KX, (size_t)t->npad * es); void* d_l = ws_get(c, WS_AP_L, (size_t)t->npad * es);
  void* d_w = ws_get(c, WS_AP_W, (size_t)t->npad * es); void* d_mu = ws_get(c, WS_AP_MU, 16); void* d_kd = ws_get(c, WS_AP_KD, 16);
  // per call: the new targets (n_new x m) and the failure word
  unsigned char* d_y = static_cast<unsigned char*>(ws_get(c, WS_AP_Y, (size_t)n_new * mc * es + 16));
  if (!d_kx || !d_l || !d_w || !d_mu || !d_kd || !d_y) return HBO_ERR_HIP;
  int* d_fail = reinterpret_cast<int*>(d_y + (((size_t)n_new * mc * es + 15) & ~(size_t)15));
  HIPCHK_A(hipMemsetAsync(d_fail, 0, sizeof(int), st));
  HIPCHK_A(hipMemcpyAsync(d_y, y_new, (size_t)n_new * mc * es, hipMemcpyHostToDevice, st));
  const int64_t n0 = t->n;
  for (int64_t q = 0; q < n_new; ++q) {
    const int64_t n = n0 + q;
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
    // l = W kx ; wl = W^T l ; then the new rows and the updated z, alpha in one workgroup (no host round trip)
    launch_tri_matvec(dtype, t->W, t->ld, t->npad, d_kx, t->npad, 1, 0, d_l, t->npad, st);
    launch_wt_z(dtype, k->d_desc, 1, t->nblk, 0, 0, t->npad, st, d_l, d_w);   // W^T l (two-stage, uses its own scratch)
    const double kadd = m->noise_variance + m->eps;
    if (dtype == HBO_F64)
      hipLaunchKernelGGL(append_row_kernel<double>, dim3(1), dim3(1024), 0, st, (double*)t->A, (double*)t->W, t->ld, n, (const double*)d_l,
                         (const double*)d_w, (const double*)d_mu, (const double*)d_kd, kadd, (const double*)d_y + q * mc, mc, (int64_t)t->npad,
                         (double*)k->zvec, (double*)t->svec, (double*)k->resid, d_fail);
    else
      hipLaunchKernelGGL(append_row_kernel<float>, dim3(1), dim3(1024), 0, st, (float*)t->A, (float*)t->W, t->ld, n, (const float*)d_l,
                         (const float*)d_w, (const float*)d_mu, (const float*)d_kd, kadd, (const float*)d_y + q * mc, mc, (int64_t)t->npad,
                         (float*)k->zvec, (float*)t->svec, (float*)k->resid, d_fail);
  }
  int failed_at = 0;
  HIPCHK_A(hipMemcpyAsync(&failed_at, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK_A(hipStreamSynchronize(st));
  int status = HBO_OK;
  if (failed_at) { status = HBO_NOT_PD; k->info = failed_at; t->n = failed_at - 1; }   // the rows before it were appended
  else t->n = n0 + n_new;
  k->h_desc.n = (int)t->n;
  HIPCHK_A(hipMemcpy(k->d_desc, &k->h_desc, sizeof(TaskDesc), hipMemcpyHostToDevice));
  HIPCHK_A(hipGetLastError());
#undef HIPCHK_A
  return status;
}

// ---- posterior / acquisition ---------------------------------------------------------------
// `ov` (hbo_acq_samples): the model and its MLP weights are already on the device at ov->md / ov->mlp_w / ov->mlp_b, the queries at
// ov->xq_dev; the acquisition values go to ov->acq_dev and stay there -- nothing is uploaded, copied back or waited for, so that
// the posteriors of many parameter samples queue up behind one another on the stream.
// `lane` > 0 (single-chunk calls only): the pass runs on the context's side stream `lane` with its own set of workspaces, so that
// the latency-bound launch chains of different samples overlap.
struct PosteriorOverride { const ModelDev* md; void* const* mlp_w; void* const* mlp_b; const void* xq_dev; void* acq_dev; int lane; };
static int posterior(hbo_ctx* c, const hbo_model* m, hbo_cache* k, const void* xq, int64_t M, int full_cov,
                     void* mu_out, void* var_out, void* acq_out, int acq_id, double param, double add_noise,
                     double scale, const PosteriorOverride* ov = nullptr) {
  if (!c || !xq) return fail(c, HBO_ERR_ARG, "posterior: null argument");
  if (M <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (!ov) prof_begin(c);
  int rc = ov ? validate_model(c, m) : upload_model(c, m);
  if (rc) return rc;
  const ModelDev* const md = ov ? ov->md : c->d_model;
  void* const* const mw = ov ? ov->mlp_w : c->d_mlp_w;
  void* const* const mb = ov ? ov->mlp_b : c->d_mlp_b;
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
  const int lane = (ov && nbuf == 1) ? ov->lane : 0;
  const int wso = 4096 * lane;   // workspace slots of this lane
  hipStream_t sa = lane == 1 ? c->stream2 : (lane == 2 ? c->stream4 : c->stream), sb = (nbuf == 2 && !c->opt_post_serial) ? c->stream2 : sa;
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
  if (ov) d_xq = (char*)const_cast<void*>(ov->xq_dev);
  else {
    d_xq = (char*)ws_get(c, WS_XQ, (size_t)M * m->input_dim * es); if (!d_xq) return HBO_ERR_HIP;
    HIPCHK_P(hipMemcpyAsync(d_xq, xq, (size_t)M * m->input_dim * es, hipMemcpyHostToDevice, sa));
  }
  { d_mu0 = (char*)ws_get(c, WS_MU0 + wso, vec_b * nbuf); if (!d_mu0) return HBO_ERR_HIP; }
  { d_kd = (char*)ws_get(c, WS_KD + wso, vec_b * nbuf); if (!d_kd) return HBO_ERR_HIP; }
  { d_mu = ws_get(c, WS_MU + wso, (size_t)M * es); if (!d_mu) return HBO_ERR_HIP; }
  { d_var = ws_get(c, WS_VAR + wso, (size_t)M * es); if (!d_var) return HBO_ERR_HIP; }
  if (ov) d_acq = ov->acq_dev;
  else if (acq_out) { d_acq = ws_get(c, WS_ACQ + wso, (size_t)M * es); if (!d_acq) return HBO_ERR_HIP; }
  if (needs_mlp(m)) for (int l = 0; l < m->n_layers; ++l) {
    fq_stride[l] = al((size_t)mc_max * m->features[l] * es);
    fq_acts[l] = (char*)ws_get(c, WS_FQ0 + l + wso, fq_stride[l] * nbuf); if (!fq_acts[l]) return HBO_ERR_HIP;
  }
  TaskHost* t = k ? k->t : nullptr;
  size_t K_b = 0, colsq_b = 0;
  if (k) {
    K_b = al((size_t)t->npad * ldq_max * es); colsq_b = al((size_t)t->nblk * ldq_max * es);
    { d_K = (char*)ws_get(c, WS_K + wso, K_b * nbuf); if (!d_K) return HBO_ERR_HIP; }
    { d_colsq = (char*)ws_get(c, WS_COLSQ + wso, colsq_b * nbuf); if (!d_colsq) return HBO_ERR_HIP; }
    { d_mupart = (char*)ws_get(c, WS_MUPART + wso, colsq_b * nbuf); if (!d_mupart) return HBO_ERR_HIP; }
    if (full_cov) { d_V = ws_get(c, WS_V + wso, (size_t)t->npad * ldq_max * es); if (!d_V) return HBO_ERR_HIP; }
  }
  // full covariance: Kqq goes into an mpad x ldq buffer (the candidates' padded leading dimension) and V^T V is subtracted in
  // place by a GEMM (gemm.hip: GEMM_VTV); with no cache (prior branch) the M x M Gram is the answer
  if (full_cov) {
    if (k) { d_Kqq = ws_get(c, WS_KQQ + wso, (size_t)mpad_max * ldq_max * es); if (!d_Kqq) return HBO_ERR_HIP; }
    { d_cov = ws_get(c, WS_COV + wso, (size_t)M * M * es); if (!d_cov) return HBO_ERR_HIP; }
  }
  // Few candidates (a BO step asks for tens of them): one workgroup per 128-row tile of W would walk a K range of up to N alone
  // (N = 8192, 64 queries: 1.1 ms for 8.6 GFLOP); the K range is cut into chunks of `kchunk` blocks instead, one workgroup per
  // (row tile, chunk), partial products to a workspace, summed and squared by a second small kernel (0.1-0.2 ms).
  int kchunk = 0;
  void* d_vpart = nullptr;
  // (decided on the TOTAL number of candidates, not on the chunk: every post_chunk then gives the same bits)
  if (k && !full_cov && t->nblk >= 2 && (int64_t)((M + HBO_TILE - 1) / HBO_TILE) * t->nblk < 2 * c->n_cus) {
    kchunk = std::max(2, std::min(8, t->nblk / 8));   // N = 8100: 1 / 2 / 4 / 8 / 16 blocks per chunk: 0.81 / 0.48 / 0.35 / 0.35 / 0.35 ms; N = 2000: 2 / 4 / 8: 0.11 / 0.11 / 0.17
    // (below 8 blocks one block per chunk: the lone 128-tile of the last row block of an N = 512 cache ran its K = 512 alone on a
    //  CU for 75 us -- an HGP acquisition over 32 samples spent half its time there)
    if (t->nblk < 8) kchunk = 1;
    const int nch_max = (t->nblk + kchunk - 1) / kchunk;
    d_vpart = ws_get(c, WS_VPART + wso, (size_t)nch_max * t->npad * ldq_max * es);
    if (!d_vpart) { c->err.clear(); kchunk = 0; }
  }
  // fp32: the product runs on the bf16 matrix cores from exact three-way splits of both operands (post3.hip)
  bool use3 = k && dtype == HBO_F32 && c->opt_post_bf16x3 && !full_cov && kchunk == 0;
  unsigned short* d_K3 = nullptr; size_t k3_b = 0;
  const int nkb = k ? t->npad / 16 : 0;
  // stationary covariances (|k| <= signal variance: the cross Gram's scale is known without a pass over it): two-way fp16 split,
  // three MFMAs per product instead of six (post2h.hip)
  const bool use2h = use3 && c->opt_post_f16x2 && m->kernel_id != HBO_KERNEL_DOT;
  const int planes = use2h ? 2 : 3;
  const float kscale = use2h ? post2h_scale_for(m->signal_variance) : 1.f;
  if (use3 && k->w3_valid && k->w3_planes != planes) k->w3_valid = false;
  if (use3 && !k->w3_valid) {
    // the split copy of W costs 1.5 x its bytes (fp16: 1 x): when the device cannot spare them the fp32-MFMA product takes over
    const size_t elems = (size_t)t->npad * t->npad * planes;
    if (use2h && !k->d_wmax && hbo_malloc(c, (void**)&k->d_wmax, sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); k->d_wmax = nullptr; use3 = false; }
    if (use3 && (!k->w3 || k->w3_elems != elems)) {
      if (k->w3) hipFree(k->w3);
      k->w3 = nullptr; k->w3_elems = 0;
      if (hbo_malloc(c, (void**)&k->w3, elems * sizeof(unsigned short)) != hipSuccess) { (void)hipGetLastError(); k->w3 = nullptr; use3 = false; }
      else k->w3_elems = elems;
    }
    if (use3) {
      ProfScope ps(c, "split_w", 1, sa);
      if (use2h) {
        launch_absmax_lower(static_cast<const float*>(t->W), t->ld, t->nblk, k->d_wmax, sa);
        launch_split2h_rows(static_cast<const float*>(t->W), t->ld, t->nblk, k->w3, nkb, k->d_wmax, sa);
      } else {
        launch_split3_rows(static_cast<const float*>(t->W), t->ld, t->nblk, k->w3, nkb, sa);
      }
      k->w3_valid = true; k->w3_planes = planes;
    }
  }
  if (use3) {
    k3_b = al((size_t)mpad_max * t->npad * planes * sizeof(unsigned short));   // (mpad / 128) x nkb blocks of `planes` x 128 x 16
    d_K3 = (unsigned short*)ws_get(c, WS_K3 + wso, k3_b * nbuf);
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
        for (int l = 0; l < m->n_layers; ++l) { launch_dense_tanh(dtype, in, mw[l], mb[l], acts[l], mc, fin, m->features[l], sb); in = acts[l]; fin = m->features[l]; }
        fq_last = acts[m->n_layers - 1];
      } }
    const void* Fq = m->kernel_uses_mlp ? fq_last : xq_d;
    const void* Fmq = (m->mean_id == HBO_MEAN_LINEAR) ? (const void*)xq_d : (m->mean_id == HBO_MEAN_LINEAR_MLP ? fq_last : nullptr);
    launch_mean(dtype, Fmq, mc, fm, md, mu0_d, sb);
    launch_kdiag(dtype, Fq, mc, fdim, md, kd_d, sb);
    void* mu_d = (char*)d_mu + (size_t)q0 * es; void* var_d = (char*)d_var + (size_t)q0 * es;
    void* acq_d = d_acq ? (char*)d_acq + (size_t)q0 * es : nullptr;
    if (!k) {  // prior branch (gp.py:275-282)
      HIPCHK_P(hipMemcpyAsync(mu_d, mu0_d, (size_t)mc * es, hipMemcpyDeviceToDevice, sb));
      if (full_cov) {
        GramArgs g = {}; g.kernel_id = m->kernel_id; g.x1 = Fq; g.x2 = Fq; g.out = d_cov; g.n1 = mc; g.n2 = mc; g.ldo = mc; g.fdim = fdim;
        launch_gram(dtype, g, md, dim3((unsigned)((mc + 127) / 128), (unsigned)((mc + 127) / 128), 1), sb);
      } else {
        HIPCHK_P(hipMemcpyAsync(var_d, kd_d, (size_t)mc * es, hipMemcpyDeviceToDevice, sb));
      }
      if (d_acq) {   // acquisition on the prior
        PostArgs pa = {}; pa.Kxq = nullptr; pa.n = 0; pa.nblk = 0; pa.ldq = ldq; pa.alpha = nullptr; pa.colsq = nullptr;
        pa.kdiag = kd_d; pa.muq = mu0_d; pa.acq_out = acq_d; pa.M = mc; pa.acq_id = acq_id; pa.param = param; pa.add_noise = add_noise; pa.scale = scale;
        launch_post_epilogue(dtype, pa, sb);
      }
      if (nbuf == 2) { ev_free[b] = pool_event(c, evi++); hipEventRecord(ev_free[b], sb); }
      continue;
    }
    char* K_d = d_K + b * K_b; char* colsq_d = d_colsq + b * colsq_b;
    { ProfScope ps(c, "cross_gram", 1, sb);
      GramArgs g = {}; g.kernel_id = m->kernel_id; g.x1 = k->h_desc.F; g.x2 = Fq; g.out = K_d; g.n1 = t->n; g.n2 = mc; g.ldo = ldq;
      g.n1pad = t->npad; g.n2pad = mpad; g.fdim = fdim; g.symmetric = 0; g.padded = 1;
      // the producer of a streamed posterior runs BESIDE the product of the previous chunk, in the slots its resident grid leaves: there the
      // matrix-core form (62 KB of LDS per workgroup, the product's own MFMA pipes) is the slower one -- cfg 3: EI 58.5 ms with the direct
      // form, 59.0 with it, although alone it takes 0.33 ms per chunk against 0.58 (round 6)
      g.direct_form = nbuf == 2;
      launch_gram(dtype, g, md, dim3(mpad / HBO_TILE, t->nblk, 1), sb); }
    unsigned short* K3_d = use3 ? d_K3 + (size_t)b * (k3_b / sizeof(unsigned short)) : nullptr;
    if (use3) {
      ProfScope ps(c, "split_kxq", 1, sb);
      if (use2h) launch_split2h_transpose(reinterpret_cast<const float*>(K_d), ldq, t->npad, mpad, K3_d, nkb, kscale, sb);
      else launch_split3_transpose(reinterpret_cast<const float*>(K_d), ldq, t->npad, mpad, K3_d, nkb, sb);
    }
    if (nbuf == 2) { ev_ready[b] = pool_event(c, evi++); hipEventRecord(ev_ready[b], sb); hipStreamWaitEvent(sa, ev_ready[b], 0); }
    // ---- consumer side (sa): V = L^-1 Kxq on MFMA (column sums of squares), then mean / variance / acquisition ----
    if (use3 && use2h) {
      ProfScope ps(c, "post_gemm", 1, sa);
      Post2hArgs a = {}; a.Wp = k->w3; a.Kp = K3_d; a.nkb = nkb; a.wmax_bits = k->d_wmax; a.kscale = kscale;
      a.colsq = reinterpret_cast<float*>(colsq_d); a.ldc = ldq; a.nblk = t->nblk;
      if (c->opt_lauum_persist && !ov) {
        int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
        if (counters) { a.work_counter = counters + HBO_N_COUNTERS - 8 + (b & 1); hipMemsetAsync(a.work_counter, 0, sizeof(int), sa); }
      }
      launch_post2h(a, mpad / HBO_TILE, sa);
    } else if (use3) {
      ProfScope ps(c, "post_gemm", 1, sa);
      Post3Args a = {}; a.Wp = k->w3; a.Kp = K3_d; a.nkb = nkb;
      a.colsq = reinterpret_cast<float*>(colsq_d); a.ldc = ldq; a.V = nullptr; a.ldv = 0; a.nblk = t->nblk;
      if (c->opt_lauum_persist && !ov) {   // (a resident grid with a tile counter for the large products; one counter per chunk in flight)
        int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
        if (counters) { a.work_counter = counters + HBO_N_COUNTERS - 8 + (b & 1); hipMemsetAsync(a.work_counter, 0, sizeof(int), sa); }
      }
      launch_post3(a, mpad / HBO_TILE, sa);
    } else {
      ProfScope ps(c, "post_gemm", 1, sa);
      GemmArgs a = {}; a.tasks = k->d_desc; a.mode = GEMM_POST; a.B = K_d; a.ldb = ldq; a.V = full_cov ? d_V : nullptr; a.colsq = colsq_d;
      if (kchunk > 0) {
        int pairs = 0;
        for (int i = 0; i < t->nblk; ++i) pairs += (i + kchunk) / kchunk;
        a.kchunk = kchunk; a.V = d_vpart; a.colsq = nullptr;
        launch_gemm(dtype, a, dim3(mpad / HBO_TILE, pairs, 1), sa);
        launch_post_colsq_split(dtype, d_vpart, t->npad, ldq, mpad, t->nblk, kchunk, colsq_d, sa);
      } else {
        const int64_t tiles = (int64_t)(mpad / HBO_TILE) * t->nblk;
        if (c->opt_lauum_persist && !ov && tiles > 4 * c->n_cus) {   // (as the bf16 form above)
          int* counters = (int*)ws_get(c, WS_COUNTERS, sizeof(int) * HBO_N_COUNTERS);
          if (counters) {
            a.work_counter = counters + HBO_N_COUNTERS - 8 + (b & 1); hipMemsetAsync(a.work_counter, 0, sizeof(int), sa);
            a.persistent = 2 * c->n_cus;
          }
        }
        launch_gemm(dtype, a, dim3(mpad / HBO_TILE, t->nblk, 1), sa);
      } }
    { ProfScope ps(c, "post_epilogue", 1, sa);
      PostArgs pa = {}; pa.Kxq = K_d; pa.ldq = ldq; pa.npad = t->npad; pa.n = (int)t->n; pa.nblk = t->nblk; pa.alpha = t->svec; pa.colsq = colsq_d; pa.mupart = d_mupart + b * colsq_b;
      pa.kdiag = kd_d; pa.muq = mu0_d; pa.mu_out = mu_d; pa.var_out = var_d; pa.acq_out = acq_d; pa.M = mc;
      pa.acq_id = acq_id; pa.param = param; pa.add_noise = add_noise; pa.scale = scale;
      launch_post_epilogue(dtype, pa, sa); }
    if (nbuf == 2) { ev_free[b] = pool_event(c, evi++); hipEventRecord(ev_free[b], sa); }
    if (full_cov) {
      ProfScope ps(c, "full_cov", 1, sa);
      GramArgs g = {}; g.kernel_id = m->kernel_id; g.x1 = Fq; g.x2 = Fq; g.out = d_Kqq; g.n1 = mc; g.n2 = mc; g.ldo = ldq; g.fdim = fdim;
      launch_gram(dtype, g, md, dim3((unsigned)((mc + 127) / 128), (unsigned)((mc + 127) / 128), 1), sa);
      // (columns of V beyond the candidates are zero: Kxq is zero-padded; the padded part of the Kqq buffer is never copied out)
      GemmArgs a = {}; a.tasks = k->d_desc; a.mode = GEMM_VTV; a.B = d_V; a.ldb = ldq; a.V = d_Kqq;
      launch_gemm(dtype, a, dim3(mpad / HBO_TILE, mpad / HBO_TILE, 1), sa);
    }
  }
  if (nbuf == 2) {   // join: everything the side stream produced (the prior branch runs there entirely)
    hipEvent_t e = pool_event(c, evi++); hipEventRecord(e, sb); hipStreamWaitEvent(sa, e, 0);
  }
  if (ov) return (k && k->info != INT_MAX) ? HBO_NOT_PD : HBO_OK;   // (the caller copies back and waits once for all samples)
  if (mu_out) HIPCHK_P(hipMemcpyAsync(mu_out, d_mu, (size_t)M * es, hipMemcpyDeviceToHost, sa));
  if (var_out) {
    // (dense on the device first: a pitched copy to pageable host memory goes row by row)
    if (full_cov && k) HIPCHK_P(hipMemcpy2DAsync(d_cov, (size_t)M * es, d_Kqq, (size_t)ldq_max * es, (size_t)M * es, (size_t)M, hipMemcpyDeviceToDevice, sa));
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

// ---- S hyper-parameter samples of one model family as ONE batch ------------------------------------------------------------
// hyperbo/bo_utils/acfun.py:72-82 evaluates an acquisition function on an HGP by looping `predict` over the model-parameter
// samples (gp.py:666-682: every sample re-factorises the same observations under its own hyper-parameters); with jax that loop is
// what one would `vmap`.  Here the S Gram matrices are built and factorised as one batch of S tasks -- the kernels that depend on
// the model read one ModelDev per task (GramArgs::model_stride, launch_aug_rows) -- the inverses and alpha = K^-1 (y - mu)
// likewise, and the S posteriors + acquisition epilogues then queue up on the stream without a host round trip between them:
// one upload, one copy back, one wait.  out: [S, M] acquisition values (model dtype), row s = sample s.
extern "C" int hbo_acq_samples(hbo_ctx* c, const hbo_model* models, int32_t S, const void* x, int64_t n, const void* y, int32_t mcols,
                               const void* xq, int64_t M, int acq_id, const double* params, const double* add_noise, double scale,
                               void* out) {
  if (!c || !models || !x || !y || !xq || !out || !params || !add_noise) return fail(c, HBO_ERR_ARG, "hbo_acq_samples: null argument");
  if (S <= 0 || S > 4096) return fail(c, HBO_ERR_ARG, "hbo_acq_samples: 1 <= S <= 4096");
  if (n <= 0 || mcols <= 0 || mcols > HBO_TILE) return fail(c, HBO_ERR_ARG, "hbo_acq_samples: need n>0 and 1<=m<=128");
  if (acq_id < 0 || acq_id > HBO_ACQ_UCB) return fail(c, HBO_ERR_ARG, "hbo_acq_samples: bad acq_id");
  if (M <= 0) return HBO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const hbo_model* m0 = &models[0];
  for (int s = 0; s < S; ++s) {
    const hbo_model* m = &models[s];
    int rc = validate_model(c, m);
    if (rc) return rc;
    bool same = m->dtype == m0->dtype && m->kernel_id == m0->kernel_id && m->mean_id == m0->mean_id && m->input_dim == m0->input_dim &&
                m->kernel_uses_mlp == m0->kernel_uses_mlp && m->n_lengthscale == m0->n_lengthscale && needs_mlp(m) == needs_mlp(m0);
    if (same && needs_mlp(m0)) { same = m->n_layers == m0->n_layers; for (int l = 0; same && l < m0->n_layers; ++l) same = m->features[l] == m0->features[l]; }
    if (!same) return fail(c, HBO_ERR_ARG, "hbo_acq_samples: the samples must share dtype, covariance, mean and MLP architecture");
  }
  prof_begin(c);
  const int dtype = m0->dtype;
  const size_t es = esize(dtype);
  hipStream_t st = c->stream;
  const int D = m0->input_dim;
  const bool mlp = needs_mlp(m0);
  const int L = mlp ? m0->n_layers : 0;
  std::vector<hbo_cache*> ks(S, nullptr);
  auto cleanup = [&]() { for (hbo_cache* k : ks) if (k) hbo_cache_free(c, k); };
#define HIPCHK_S(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { c->err = std::string(#call " failed: ") + hipGetErrorString(e__); cleanup(); return HBO_ERR_HIP; } } while (0)
#define RCCHK_S(call) do { int rc__ = (call); if (rc__) { cleanup(); return rc__; } } while (0)
  // ---- the S models, their MLP weights, the queries: device copies that live for the whole call
  ModelDev* d_models = static_cast<ModelDev*>(ws_get(c, WS_SMP_MODELS, sizeof(ModelDev) * S));
  TaskDesc* d_batch = static_cast<TaskDesc*>(ws_get(c, WS_SMP_DESC, sizeof(TaskDesc) * S));
  int* d_infos = static_cast<int*>(ws_get(c, WS_SMP_INFO, sizeof(int) * S));
  char* d_acq = static_cast<char*>(ws_get(c, WS_SMP_ACQ, (size_t)S * M * es));
  char* d_xq = static_cast<char*>(ws_get(c, WS_SMP_XQ, (size_t)M * D * es));
  if (!d_models || !d_batch || !d_infos || !d_acq || !d_xq) return HBO_ERR_HIP;
  std::vector<ModelDev> h_models(S);
  for (int s = 0; s < S; ++s) fill_model_dev(h_models[s], &models[s]);
  HIPCHK_S(hipMemcpyAsync(d_models, h_models.data(), sizeof(ModelDev) * S, hipMemcpyHostToDevice, st));
  HIPCHK_S(hipMemcpyAsync(d_xq, xq, (size_t)M * D * es, hipMemcpyHostToDevice, st));
  std::vector<void*> w_dev((size_t)S * HBO_MAX_MLP_LAYERS, nullptr), b_dev((size_t)S * HBO_MAX_MLP_LAYERS, nullptr);
  if (mlp) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t per = 0; int fin = D;
    for (int l = 0; l < L; ++l) { per += al((size_t)fin * m0->features[l] * es) + al((size_t)m0->features[l] * es); fin = m0->features[l]; }
    char* blk = static_cast<char*>(ws_get(c, WS_SMP_MLP, per * S));
    if (!blk) return HBO_ERR_HIP;
    for (int s = 0; s < S; ++s) {
      char* p = blk + per * s; fin = D;
      for (int l = 0; l < L; ++l) {
        const size_t wb = (size_t)fin * m0->features[l] * es, bb = (size_t)m0->features[l] * es;
        w_dev[(size_t)s * HBO_MAX_MLP_LAYERS + l] = p; HIPCHK_S(hipMemcpyAsync(p, models[s].mlp_kernel[l], wb, hipMemcpyHostToDevice, st)); p += al(wb);
        b_dev[(size_t)s * HBO_MAX_MLP_LAYERS + l] = p; HIPCHK_S(hipMemcpyAsync(p, models[s].mlp_bias[l], bb, hipMemcpyHostToDevice, st)); p += al(bb);
        fin = m0->features[l];
      }
    }
  }
  // ---- S caches over the same observations (each a complete hbo_cache: everything that reads one works on them)
  const int npad = round_up(n, HBO_TILE), nblk = npad / HBO_TILE;
  std::vector<unsigned char> yt((size_t)n * mcols * es);
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < mcols; ++a) memcpy(yt.data() + ((size_t)a * n + i) * es, (const unsigned char*)y + ((size_t)i * mcols + a) * es, es);
  std::vector<TaskDesc> h_batch(S);
  for (int s = 0; s < S; ++s) {
    hbo_cache* k = ks[s] = new hbo_cache();
    k->dtype = dtype; k->D = D; k->m = mcols;
    TaskHost* t = k->t = new TaskHost();
    t->n = n; t->m = mcols; t->npad = npad; t->nblk = nblk; t->ld = padded_ld(npad, dtype);
    HIPCHK_S(dev_alloc(c, &t->X, (size_t)npad * D * es));
    HIPCHK_S(dev_alloc(c, &t->ysum, (size_t)n * mcols * es));
    if (s == 0) {
      HIPCHK_S(hipMemcpyAsync(t->X, x, (size_t)n * D * es, hipMemcpyHostToDevice, st));
      HIPCHK_S(hipMemcpyAsync(t->ysum, yt.data(), (size_t)n * mcols * es, hipMemcpyHostToDevice, st));
    } else {
      HIPCHK_S(hipMemcpyAsync(t->X, ks[0]->t->X, (size_t)n * D * es, hipMemcpyDeviceToDevice, st));
      HIPCHK_S(hipMemcpyAsync(t->ysum, ks[0]->t->ysum, (size_t)n * mcols * es, hipMemcpyDeviceToDevice, st));
    }
    RCCHK_S(ensure_task_workspace(c, dtype, t, true, mcols));
    if (mlp) RCCHK_S(t->feat.ensure(c, m0, npad));
    fill_desc(k->h_desc, t, &models[s], dtype, ROLE_FACTOR);
    h_batch[s] = k->h_desc;
    HIPCHK_S(hbo_malloc(c, (void**)&k->d_desc, sizeof(TaskDesc)));
    HIPCHK_S(hbo_malloc(c, (void**)&k->d_info, sizeof(int)));
    HIPCHK_S(hbo_malloc(c, &k->resid, (size_t)mcols * npad * es));
    HIPCHK_S(hbo_malloc(c, &k->zvec, (size_t)mcols * npad * es));
    HIPCHK_S(hipMemcpyAsync(k->d_desc, &k->h_desc, sizeof(TaskDesc), hipMemcpyHostToDevice, st));
  }
  HIPCHK_S(hipMemcpyAsync(d_batch, h_batch.data(), sizeof(TaskDesc) * S, hipMemcpyHostToDevice, st));
  HIPCHK_S(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_infos), INT_MAX, S, st));
  // ---- one batched pipeline: features (per sample: its own weights), residual rows, Gram, factorisation, inverse, alpha
  if (c->opt_poison) launch_poison(dtype, d_batch, S, npad, st);
  { ProfScope ps(c, "features", 1);
    if (mlp) for (int s = 0; s < S; ++s) run_mlp(c, m0, ks[s]->t->X, n, ks[s]->t->feat.acts.data(), &w_dev[(size_t)s * HBO_MAX_MLP_LAYERS], &b_dev[(size_t)s * HBO_MAX_MLP_LAYERS]);
    launch_aug_rows(dtype, d_batch, S, npad, d_models, st, 1); }
  { ProfScope ps(c, "gram", 1);
    GramArgs g = {}; g.kernel_id = m0->kernel_id; g.tasks = d_batch; g.fdim = feature_dim(m0); g.symmetric = 1; g.padded = 1; g.model_stride = 1;
    launch_gram(dtype, g, d_models, dim3(nblk, nblk, S), st); }
  TrtriProgress trtri_pg;
  const bool early_trtri = use_lookahead(c, S, nblk) && c->opt_overlap_trtri && nblk >= 4;
  c->trtri_host_task = S == 1 ? h_batch[0] : TaskDesc{};
  double bound_all = chol_diag_bound_of(&models[0]);
  for (int s = 1; s < S; ++s) { const double b = chol_diag_bound_of(&models[s]); bound_all = (b > 0 && bound_all > 0) ? std::max(bound_all, b) : 0.0; }
  CholBoundScope bound_scope(c, bound_all);
  { ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, d_batch, S, nblk, d_infos, early_trtri ? &trtri_pg : nullptr); }
  { ProfScope ps(c, "trtri", 1); run_trtri(c, dtype, d_batch, S, nblk, &trtri_pg); }
  { ProfScope ps(c, "wt_z", 1);
    for (int a = 0; a < mcols; ++a) launch_wt_z(dtype, d_batch, S, nblk, a, a, npad, st); }
  std::vector<int> h_infos(S, INT_MAX);
  HIPCHK_S(hipMemcpyAsync(h_infos.data(), d_infos, sizeof(int) * S, hipMemcpyDeviceToHost, st));
  HIPCHK_S(hipStreamSynchronize(st));   // (the samples' info words decide NaN rows below; the caller's x / y / weights have been consumed)
  // ---- S posteriors + acquisition epilogues: three independent launch chains (main stream + the two side streams, each with its
  //      own workspaces) -- a pass is ~7 latency-bound launches of 10-20 us, and the passes of different samples share nothing
  const int lanes = (M <= std::max<int64_t>(c->opt_post_chunk, HBO_TILE)) ? 3 : 1;
  hipStream_t lane_stream[3] = {st, c->stream2, c->stream4};
  if (lanes > 1) { hipEvent_t e = pool_event(c, 0); hipEventRecord(e, st); hipStreamWaitEvent(c->stream2, e, 0); hipStreamWaitEvent(c->stream4, e, 0); }
  bool any_bad = false;
  for (int s = 0; s < S; ++s) {
    ks[s]->info = h_infos[s];
    PosteriorOverride ov = {d_models + s, mlp ? &w_dev[(size_t)s * HBO_MAX_MLP_LAYERS] : nullptr, mlp ? &b_dev[(size_t)s * HBO_MAX_MLP_LAYERS] : nullptr, d_xq,
                            d_acq + (size_t)s * M * es, lanes > 1 ? s % lanes : 0};
    int rc = posterior(c, &models[s], ks[s], xq, M, 0, nullptr, nullptr, out, acq_id, params[s], add_noise[s], scale, &ov);
    if (rc == HBO_NOT_PD) any_bad = true;
    else if (rc) { for (hipStream_t q : lane_stream) hipStreamSynchronize(q); cleanup(); return rc; }
  }
  for (int l = 1; l < lanes; ++l) { hipEvent_t e = pool_event(c, (size_t)l); hipEventRecord(e, lane_stream[l]); hipStreamWaitEvent(st, e, 0); }
  HIPCHK_S(hipMemcpyAsync(out, d_acq, (size_t)S * M * es, hipMemcpyDeviceToHost, st));
  HIPCHK_S(hipStreamSynchronize(st));
  HIPCHK_S(hipStreamSynchronize(c->stream2));
  HIPCHK_S(hipGetLastError());
  prof_collect(c);
  for (int s = 0; s < S; ++s) if (h_infos[s] != INT_MAX) fill_nan((char*)out + (size_t)s * M * es, (size_t)M, dtype);
  cleanup();
#undef HIPCHK_S
#undef RCCHK_S
  return any_bad ? HBO_NOT_PD : HBO_OK;
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
