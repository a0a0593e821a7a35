// The objectives of the C ABI: hbo_nll / hbo_objective (hyperbo/gp_utils/objectives.py:29-210 -- NLL, EKL, Euclid, value and
// gradient in one pass: features -> Gram -> blocked Cholesky with the augmented rows -> inverse -> K^-1 -> contraction) and its
// task-sharded form hbo_objective_sharded (device-side reduction + one in-place all-reduce).
#include "api_internal.h"
#include <chrono>

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

// The sharded form never leaves its peers alone in the collective: objective_local does everything up to (not including) the
// all-reduce and hands back the device buffer [nll, count, grad]; whatever it returns, objective_impl then takes part in the ONE
// all-reduce of the evaluation -- with NaN in every slot after a local failure (the peers see a NaN objective, not a hang) --
// and only if not even that buffer can be had does it abort the communicator, so that the peers' collective fails too.
struct ShardOut { double* d_red = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr; int red_count = 0; };
int comm_abort(hbo_ctx* c);   // comm.hip

static int objective_local(hbo_ctx* c, const hbo_model* m_in, hbo_dataset* ds, int objective, double* nll_sum,
                           double* nll_per_task, double* grad_sum, const ShardReq* sh, ShardOut* so) {
  if (!c || !nll_sum || !m_in || (!ds && !sh)) return fail(c, HBO_ERR_ARG, "hbo_objective: null argument");
  if (objective != HBO_OBJ_NLL && objective != HBO_OBJ_EKL && objective != HBO_OBJ_EUC) return fail(c, HBO_ERR_ARG, "hbo_objective: unknown objective id");
  HIPCHK(c, hipSetDevice(c->device));
  hbo_model mcopy = *m_in;
  if (objective != HBO_OBJ_NLL) mcopy.eps = 0.0;   // objectives.py:63-65: cov_model = K + noise I, no jitter
  const hbo_model* m = &mcopy;
  CholBoundScope bound_scope(c, 0.0);   // (set where the factorisation starts; cleared on every way out)
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
  if (so) so->red_count = red_count;
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
    so->d_red = d_red; so->ev0 = ev0; so->ev1 = ev1;
    return HBO_OK;
  }
  if (obj != OBJ_NLL) for (TaskHost* t : ds->tasks) if (!t->ydiv) return fail(c, HBO_ERR_ARG, "hbo_objective: the dataset carries no divergence rows");
  // tasks with more than 127 aligned columns: their data rows do not ride in the augmented tile-row (TaskDesc::nvec)
  bool extras = false;
  if (obj != OBJ_NLL) for (TaskHost* t : ds->tasks) if (t->m + 1 > HBO_TILE) extras = true;
  const int dtype = ds->dtype;
  prof_begin(c);
  const auto host_t0 = std::chrono::steady_clock::now();   // host time spent queueing this evaluation (stage "host_enqueue", level 1)
  hipEvent_t ev_sh0 = nullptr;
  if (sh) { ev_sh0 = pool_event_timed(c, 0); HIPCHK(c, hipEventRecord(ev_sh0, st)); }
  rc = upload_model(c, m);
  if (rc) return rc;

  // workspaces + descriptors.  A batch that takes the single-workgroup evaluation (small.hip) without an MLP keeps its matrices in
  // LDS: no A / W / S / alpha buffers at all (a freshly sub-sampled batch per Adam step paid ~150 pool operations and two fills for
  // buffers its one launch never reads); the leaf inverses that leaf_cholesky4 also stores go to one shared scratch block
  // (the single-workgroup evaluation needs 132 KB (fp64) / 68 KB (fp32) of LDS in one workgroup: a device that cannot give it -- any
  //  ARCH other than gfx950 the Makefile is pointed at -- takes the blocked pipeline instead of failing at launch)
  const bool small_ok = c->opt_small_fused && ds && small_eval_lds(ds->dtype) <= c->lds_per_block;
  const bool lds_only = obj == OBJ_NLL && ds->max_nblk == 1 && small_ok && !needs_mlp(m);
  void* small_scratch = nullptr;
  if (lds_only) {
    small_scratch = ws_get(c, WS_SMALL_W, (size_t)HBO_TILE * padded_ld(HBO_TILE, ds->dtype) * esize(ds->dtype));
    if (!small_scratch) return HBO_ERR_HIP;
  }
  if (!lds_only && !ds->d_svec && T > 1) {
    // first evaluation of this dataset: every task's alpha vector from ONE zeroed block
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t es = esize(ds->dtype);
    size_t tot = 0;
    for (TaskHost* t : ds->tasks) tot += al((size_t)t->npad * es * (obj == OBJ_NLL ? 1 : t->m + 1));
    bool fresh = true;
    for (TaskHost* t : ds->tasks) if (t->svec) fresh = false;
    if (fresh && dev_alloc(c, &ds->d_svec, tot) == hipSuccess) {
      HIPCHK(c, hipMemsetAsync(ds->d_svec, 0, tot, st));
      size_t off = 0;
      for (TaskHost* t : ds->tasks) {
        const int cols = obj == OBJ_NLL ? 1 : t->m + 1;
        t->svec = (char*)ds->d_svec + off; t->svec_cols = cols; t->svec_shared = true;
        off += al((size_t)t->npad * es * cols);
      }
    } else { (void)hipGetLastError(); ds->d_svec = nullptr; }
  }
  ds->h_desc.resize(T);
  for (int k = 0; k < T; ++k) {
    TaskHost* t = ds->tasks[k];
    if (!lds_only) {
      // (a batch with extra rows runs the inverse over ALL its tasks, value-only calls too: GEMM_TRTRI_A stores S21 of every task with
      //  two or more blocks, so every task of such a batch needs S -- not only the ones with more than 127 aligned columns)
      rc = ensure_task_workspace(c, dtype, t, (want_grad || extras) && !euc, obj == OBJ_NLL ? 1 : t->m + 1);
      if (rc) return rc;
    }
    if (needs_mlp(m)) { rc = t->feat.ensure(c, m, t->n); if (rc) return rc; }
    if (needs_mlp(m) && want_grad) {
      int maxf = m->input_dim;
      for (int l = 0; l < m->n_layers; ++l) maxf = std::max(maxf, (int)m->features[l]);
      const size_t need = (size_t)t->n * maxf;
      if (t->dF_elems < need) {
        if (t->dF) dev_free(c, t->dF);
        if (t->dtmp) dev_free(c, t->dtmp);
        t->dF = t->dtmp = nullptr;
        HIPCHK(c, dev_alloc(c, (void**)&t->dF, need * sizeof(double)));
        HIPCHK(c, dev_alloc(c, (void**)&t->dtmp, need * sizeof(double)));
        t->dF_elems = need;
      }
    }
    fill_desc(ds->h_desc[k], t, m, dtype, obj);
    if (lds_only && !ds->h_desc[k].W) ds->h_desc[k].W = small_scratch;   // (every task writes its leaf inverses there: never read)
  }
  int64_t max_n = 0;
  for (int k = 0; k < T; ++k) max_n = std::max<int64_t>(max_n, ds->tasks[k]->n);
  if (needs_mlp(m)) {
    // per-task pointers of the batched MLP passes (mlp.hip); uploaded when an allocation moved
    std::vector<MlpTaskDev> hm(T);
    for (int k = 0; k < T; ++k) {
      TaskHost* t = ds->tasks[k];
      memset(&hm[k], 0, sizeof(MlpTaskDev));
      hm[k].x = t->X; hm[k].n = t->n; hm[k].dF = t->dF; hm[k].dtmp = t->dtmp;
      for (int l = 0; l < m->n_layers; ++l) hm[k].acts[l] = t->feat.acts[l];
    }
    if (!ds->d_mlp) HIPCHK(c, dev_alloc(c, (void**)&ds->d_mlp, sizeof(MlpTaskDev) * T));
    if (ds->h_mlp_dev.size() != (size_t)T || memcmp(ds->h_mlp_dev.data(), hm.data(), sizeof(MlpTaskDev) * T) != 0) {
      HIPCHK(c, hipStreamSynchronize(st));   // (the previous table may still be read by queued work; happens once per dataset)
      ds->h_mlp_dev = hm;
      HIPCHK(c, hipMemcpyAsync(ds->d_mlp, ds->h_mlp_dev.data(), sizeof(MlpTaskDev) * T, hipMemcpyHostToDevice, st));
    }
  }
  // (per-dataset buffers come from the context's pool: a fresh batch per Adam step paid three hipMalloc + three synchronising
  //  hipFree for its descriptors, results and gradient partials)
  if (!ds->d_desc) HIPCHK(c, dev_alloc(c, (void**)&ds->d_desc, sizeof(TaskDesc) * T));
  const int out_stride = (m->kernel_id == HBO_KERNEL_DOT ? 0 : m->n_lengthscale) + 6 + mean_feature_dim(m);
  const size_t pack_bytes = sizeof(double) * T * (1 + (size_t)out_stride) + sizeof(int) * T;
  if (ds->pack_bytes < pack_bytes) {
    if (ds->d_pack) dev_free(c, ds->d_pack);
    ds->d_pack = nullptr; ds->pack_bytes = 0;
    HIPCHK(c, dev_alloc(c, (void**)&ds->d_pack, pack_bytes));
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
    HIPCHK(c, hipEventRecord(c->ev_upload, st));   // the pinned buffer is written again further down (sharded: the scatter map)
    ds->h_desc_dev = ds->h_desc;
  }
  // (a batch that takes the single-workgroup evaluation sets its info words itself: one launch less on a 60 us path)
  if (!(obj == OBJ_NLL && ds->max_nblk == 1 && small_ok)) HIPCHK(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ds->d_info), INT_MAX, T, st));

  const int max_nblk = ds->max_nblk, max_npad = max_nblk * HBO_TILE;
  // every task fits one 128-block (the reference's training regime: sub-sampled tasks of 50-100 points): ONE launch, one workgroup
  // per task does Gram -> factorisation -> inverse -> K^-1 -> contraction in LDS (small.hip) instead of the 13 launches below
  const bool fused_small = obj == OBJ_NLL && max_nblk == 1 && small_ok;
  auto mlp_forward = [&]() {   // the basis of every task, one launch per layer
    int fin = m->input_dim;
    for (int l = 0; l < m->n_layers; ++l) {
      launch_mlp_forward_batch(dtype, ds->d_mlp, T, max_n, l, c->d_mlp_w[l], c->d_mlp_b[l], fin, m->features[l], st);
      fin = m->features[l];
    }
  };
  if (fused_small) {
    if (needs_mlp(m)) { ProfScope ps(c, "features", 1); mlp_forward(); }
    ProfScope ps(c, "small_eval", 1);
    launch_small_eval(dtype, ds->d_desc, T, c->d_model, m->kernel_id, feature_dim(m), ds->d_info, ds->d_nll,
                      want_grad ? ds->d_gradout : nullptr, out_stride, want_grad && needs_mlp(m), st);
  } else {
  if (c->opt_poison) launch_poison(dtype, ds->d_desc, T, max_npad, st);
  {
    ProfScope ps(c, "features", 1);
    if (needs_mlp(m)) mlp_forward();
    launch_aug_rows(dtype, ds->d_desc, T, max_npad, c->d_model, st);
  }
  int max_naug = 1;
  for (int k = 0; k < T; ++k) max_naug = std::max(max_naug, ds->h_desc[k].naug);
  TrtriProgress trtri_pg;
  SweepState sweep_st;
  hipStream_t side = st; hipEvent_t ev_side = nullptr;
  const bool la = use_lookahead(c, T, max_nblk);
  // the inverse and K^-1 behind the panel chain, row group by row group (sched.hip:sweep_advance) -- or the block-recursive
  // inverse started beside the chain and K^-1 = W^T W after it
  const bool sweep = want_grad && !euc && use_sweep(c, dtype, T, max_nblk);
  if (sweep) sweep_st.qs = c->opt_sweep_qs > 0 ? c->opt_sweep_qs : sweep_group(T, max_nblk);
  const bool early_trtri = !sweep && want_grad && la && c->opt_overlap_trtri && max_nblk >= 4;
  if (!euc) {
    {
      ProfScope ps(c, "gram", 1);
      GramArgs g = {}; g.kernel_id = c->h_model->kernel_id; g.tasks = ds->d_desc; g.fdim = feature_dim(m); g.symmetric = 1; g.padded = 1;
      launch_gram(dtype, g, c->d_model, dim3(max_nblk, max_nblk, T), st);
    }
    {
      c->trtri_host_task = TaskDesc{};
      if (T == 1) c->trtri_host_task = ds->h_desc[0];
      c->chol_diag_bound = chol_diag_bound_of(m);   // (cleared where the evaluation leaves: bound_scope)
      ProfScope ps(c, "potrf", 1); run_potrf(c, dtype, ds->d_desc, T, max_nblk, ds->d_info, early_trtri ? &trtri_pg : nullptr, sweep ? &sweep_st : nullptr);
    }
    // the small reductions (log-determinant + quadratic form now, alpha = W^T z and d nll / d mu after the inverse) run on
    // the idle panel stream beside the inverse and K^-1 = W^T W instead of between them (0.14 ms at cfg 2)
    side = (want_grad && obj == OBJ_NLL && la) ? c->stream2 : st;
    if (side != st) { hipEvent_t e = pool_event(c, 2); hipEventRecord(e, st); hipStreamWaitEvent(side, e, 0); }
    { ProfScope ps(c, "nll_reduce", 1, side); launch_nll_reduce(dtype, ds->d_desc, T, ds->d_info, ds->d_nll, side); }
  }

  // the data rows of tasks with more than 127 aligned columns, as outer-product vectors in svec columns 1..m (objectives.py:29-106
  // has no limit on m).  EUC: the rows themselves.  EKL: tr(K1^-1 C0) = sum_b |W row_b|^2 and alpha_b = W^T W row_b from the explicit
  // inverse (two triangular products over all m rows), the quadratic forms added to the task's value.  Main stream, W complete.
  auto extra_rows = [&]() -> int {
    // one scratch buffer for the largest of them, taken BEFORE the loop: growing it between two tasks would free memory that the
    // first task's queued products still read
    size_t zbytes = 0;
    if (!euc) for (int k = 0; k < T; ++k) if (ds->h_desc[k].nvec) zbytes = std::max(zbytes, (size_t)ds->tasks[k]->m * ds->tasks[k]->npad * esize(dtype));
    void* const zb = zbytes ? ws_get(c, WS_EXTRA_Z, zbytes) : nullptr;
    if (zbytes && !zb) return HBO_ERR_HIP;
    for (int k = 0; k < T; ++k) {
      if (!ds->h_desc[k].nvec) continue;
      TaskHost* t = ds->tasks[k];
      const size_t es = esize(dtype);
      char* cols = static_cast<char*>(t->svec) + (size_t)t->npad * es;   // column 1
      launch_expand_rows(dtype, t->ydiv, t->n, t->npad, cols, t->m, st);
      if (euc) continue;
      launch_tri_matvec(dtype, t->W, t->ld, t->npad, cols, t->npad, t->m, 0, zb, t->npad, st);
      launch_add_sumsq(dtype, zb, t->npad, t->m, ds->h_desc[k].coef_c, ds->d_nll + k, st);
      launch_tri_matvec(dtype, t->W, t->ld, t->npad, zb, t->npad, t->m, 1, cols, t->npad, st);
    }
    return HBO_OK;
  };
  if (extras && !euc && !want_grad) {   // value only: the inverse is needed for the extra rows all the same
    { ProfScope ps(c, "trtri", 1); run_trtri(c, dtype, ds->d_desc, T, max_nblk, &trtri_pg); }
    ProfScope ps(c, "extra_rows", 1);
    rc = extra_rows(); if (rc) return rc;
  }
  const int fdim = feature_dim(m);
  const int nacc = grad_nacc(m->kernel_id, fdim);
  const int64_t stride_task = (int64_t)(max_nblk * (max_nblk + 1)) * nacc;   // two half-tile slots per lower tile
  if (want_grad || euc) {   // EUC: the Frobenius norm of the value comes out of the contraction pass
    const size_t pb = sizeof(double) * (stride_task * T + (size_t)HBO_GRAD_PRE_ROWS * nacc * T);   // per-tile partials + their pre-reduction
    if (ds->partials_bytes < pb) { if (ds->d_partials) dev_free(c, ds->d_partials); HIPCHK(c, dev_alloc(c, (void**)&ds->d_partials, pb)); ds->partials_bytes = pb; }
    if (!euc && sweep) {
      // what the chain left: the last row group(s).  W is complete behind the last group's rows (event), K^-1 behind its update
      hipEvent_t e = side != st ? pool_event(c, 3) : nullptr;
      { ProfScope ps(c, "sweep_tail", 1);
        sweep_advance(c, dtype, ds->d_desc, T, max_nblk, max_nblk, st, sweep_st, e); }
      if (side != st) hipStreamWaitEvent(side, e, 0);
      { ProfScope ps(c, "wt_z", 1, side);
        for (int b = 0; b < max_naug; ++b) launch_wt_z(dtype, ds->d_desc, T, max_nblk, b, b, max_npad, side); }
      if (side != st) {
        launch_dmu(dtype, ds->d_desc, T, obj, side);
        ev_side = pool_event(c, 4); hipEventRecord(ev_side, side);
        hipStreamWaitEvent(st, ev_side, 0);
      }
    } else if (!euc) {
      { ProfScope ps(c, "trtri", 1);
        run_trtri(c, dtype, ds->d_desc, T, max_nblk, &trtri_pg); }
      if (side != st) { hipEvent_t e = pool_event(c, 3); hipEventRecord(e, st); hipStreamWaitEvent(side, e, 0); }   // W is complete
      { ProfScope ps(c, "wt_z", 1, side);
        for (int b = 0; b < max_naug; ++b) launch_wt_z(dtype, ds->d_desc, T, max_nblk, b, b, max_npad, side); }
      if (side != st) {
        launch_dmu(dtype, ds->d_desc, T, obj, side);
        ev_side = pool_event(c, 4); hipEventRecord(ev_side, side);
      }
      { ProfScope ps(c, "lauum", 1); run_lauum(c, dtype, ds->d_desc, T, max_nblk); }
      if (ev_side) hipStreamWaitEvent(st, ev_side, 0);
    }
    if (extras) { ProfScope ps(c, "extra_rows", 1); rc = extra_rows(); if (rc) return rc; }
    { ProfScope ps(c, "grad_contract", 1);
      if (!ev_side) launch_dmu(dtype, ds->d_desc, T, obj, st);
      launch_grad_contract(dtype, ds->d_desc, T, max_nblk, c->d_model, m->kernel_id, fdim, obj, ds->d_partials, stride_task, st);
      launch_grad_finalize(dtype, ds->d_desc, T, c->d_model, m->kernel_id, fdim, obj, ds->d_partials, stride_task, ds->d_gradout, out_stride, euc ? ds->d_nll : nullptr, st,
                           ds->d_partials + stride_task * T, max_nblk); }
  }
  }   // !fused_small
  if (want_grad && needs_mlp(m)) {
    // d nll / d features -> MLP backward (hyperbo/gp_utils/basis_functions.py:24-36), summed over tasks; every pass one launch
    // for the whole batch (mlp.hip)
    ProfScope ps(c, "mlp_backward", 1);
    const int L = m->n_layers, flast = m->features[L - 1];
    size_t tot = 0; int fin0 = m->input_dim;
    std::vector<size_t> woff(L), boff(L);
    for (int l = 0; l < L; ++l) { woff[l] = tot; tot += (size_t)fin0 * m->features[l]; boff[l] = tot; tot += m->features[l]; fin0 = m->features[l]; }
    if (ds->mlpgrad_elems < tot) { if (ds->d_mlpgrad) dev_free(c, ds->d_mlpgrad); HIPCHK(c, dev_alloc(c, (void**)&ds->d_mlpgrad, tot * sizeof(double))); ds->mlpgrad_elems = tot; }
    HIPCHK(c, hipMemsetAsync(ds->d_mlpgrad, 0, tot * sizeof(double), st));
    launch_mlp_zero_dF_batch(ds->d_mlp, T, max_n, flast, st);
    if (m->kernel_uses_mlp) {
      launch_grad_feat(dtype, ds->d_desc, T, max_nblk, c->d_model, m->kernel_id, flast, obj, st);
      if (euc) launch_scale_dF(ds->d_desc, T, (int64_t)max_npad, flast, st);
    }
    if (m->mean_id == HBO_MEAN_LINEAR_MLP) launch_grad_feat_mean(dtype, ds->d_desc, T, (int64_t)max_npad, c->d_model, flast, st);
    for (int l = L - 1; l >= 0; --l) {
      const int fin = l ? m->features[l - 1] : m->input_dim;
      launch_dense_bwd_batch(dtype, ds->d_mlp, T, max_n, l, ((L - 1 - l) % 2) == 0, c->d_mlp_w[l], ds->d_mlpgrad + woff[l], ds->d_mlpgrad + boff[l],
                             fin, m->features[l], l > 0, st);
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
    HIPCHK(c, hipGetLastError());
    so->d_red = d_red; so->ev0 = ev_sh0; so->ev1 = ev1;
    return HBO_OK;
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
  const double host_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipGetLastError());
  prof_collect(c);
  if (c->prof_level >= 1) { c->prof_names.push_back("host_enqueue"); c->prof_ms.push_back(host_enqueue_ms); c->prof_count.push_back(1); }

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
static int objective_impl(hbo_ctx* c, const hbo_model* m_in, hbo_dataset* ds, int objective, double* nll_sum,
                          double* nll_per_task, double* grad_sum, const ShardReq* sh) {
  if (!sh) return objective_local(c, m_in, ds, objective, nll_sum, nll_per_task, grad_sum, nullptr, nullptr);
  if (c && c->comm_aborted)
    return fail(c, HBO_ERR_COMM, "hbo_objective_sharded: the communicator was aborted after a failed evaluation; call hbo_comm_init again");
  ShardOut so;
  int rc_local = objective_local(c, m_in, ds, objective, nll_sum, nll_per_task, grad_sum, sh, &so);
  if (!c || !nll_sum || !m_in) return rc_local;   // argument errors are the same on every rank: nobody reaches the collective
  // one-shot fault injection (hbo_tune "fault_shard", tests only): 1 = this rank's local part counts as failed (NaN contribution
  // path below), 2 = and it cannot even produce the NaN buffer (abort path)
  const int fault = c->opt_fault_shard; c->opt_fault_shard = 0;
  if (fault && rc_local == HBO_OK) { rc_local = fail(c, HBO_ERR_HIP, "hbo_objective_sharded: injected local failure (fault_shard)"); so.d_red = nullptr; }
  hipStream_t st = c->stream;
  int red_count = so.red_count;
  if (red_count <= 0) {   // failed before the model was looked at: the layout still fixes the length the peers reduce
    hbo_grad_layout lay;
    if (hbo_grad_layout_of(m_in, &lay) != HBO_OK) return rc_local;
    red_count = 2 + (grad_sum ? lay.total : 0);
  }
  if (rc_local != HBO_OK || !so.d_red) {
    // local failure: NaN into every slot of the collective (0x7FF800007FF80000 is a quiet NaN).  The failed pipeline may have left
    // launches queued on the side streams (look-ahead panels, the early inverse): join them before the workspaces are reused
    (void)hipGetLastError();
    for (hipStream_t q : {c->stream2, c->stream3, c->stream4}) if (q) (void)hipStreamSynchronize(q);
    (void)hipGetLastError();
    so.d_red = static_cast<double*>(ws_get(c, WS_SHARD_RED, sizeof(double) * red_count));
    so.ev0 = pool_event_timed(c, 0); so.ev1 = pool_event_timed(c, 1);
    const bool ok = fault != 2 && so.d_red && hipEventRecord(so.ev0, st) == hipSuccess &&
                    hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(so.d_red), 0x7FF80000, 2 * (size_t)red_count, st) == hipSuccess &&
                    hipEventRecord(so.ev1, st) == hipSuccess;
    if (!ok) { if (c->comm) comm_abort(c); return rc_local ? rc_local : HBO_ERR_HIP; }
  }
  hipEvent_t ev2 = pool_event_timed(c, 2);
  int rc = c->comm ? comm_allreduce_device(c, so.d_red, red_count, st) : HBO_OK;
  if (rc) return rc_local ? rc_local : rc;
  HIPCHK(c, hipEventRecord(ev2, st));
  double* stage = static_cast<double*>(pinned_stage(c, sizeof(double) * red_count));
  if (!stage) return fail(c, HBO_ERR_HIP, "hbo_objective_sharded: pinned staging buffer");
  HIPCHK(c, hipEventSynchronize(c->ev_upload));
  HIPCHK(c, hipMemcpyAsync(stage, so.d_red, sizeof(double) * red_count, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipGetLastError());
  prof_collect(c);
  *nll_sum = stage[0];
  *sh->count = stage[1];
  if (grad_sum) for (int i = 0; i < red_count - 2; ++i) grad_sum[i] = stage[2 + i];
  if (sh->timing) {
    float ms_local = 0, ms_comm = 0;
    hipEventElapsedTime(&ms_local, so.ev0, so.ev1); hipEventElapsedTime(&ms_comm, so.ev1, ev2);
    sh->timing[0] = ms_local; sh->timing[1] = 1e3 * ms_comm;
  }
  if (rc_local) return rc_local;
  return std::isnan(*nll_sum) ? HBO_NOT_PD : HBO_OK;
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
